#!/bin/bash
# Builds cgat_amd/libcgat_hip.so for gfx950 (MI355X).  hipcc cross-compiles without a GPU.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
OUT="$HERE/libcgat_hip.so"
SRCS=(api gemm gemmsplit bilinear wgradc edgez edgebwd rowsdw collate optim rowops segment plan layers chain rowprog)
OBJS=()
PIDS=()
mkdir -p "$HERE/csrc/build"
# objects are stale when the extra flags they were compiled with differ from this call's (a dev tool's -D... ablation must
# never survive into a later default build: staleness by mtime alone kept the flagged object)
FLAGS_STAMP="$HERE/csrc/build/.hipcc_flags"
if [ ! -f "$FLAGS_STAMP" ] || [ "$(cat "$FLAGS_STAMP")" != "${CGAT_HIPCC_FLAGS}" ]; then
  rm -f "$HERE"/csrc/build/*.o
  printf '%s' "${CGAT_HIPCC_FLAGS}" > "$FLAGS_STAMP"
fi
for s in "${SRCS[@]}"; do
  src="$HERE/csrc/$s.hip"; obj="$HERE/csrc/build/$s.o"
  if [ ! -f "$obj" ] || [ "$src" -nt "$obj" ] || [ "$HERE/csrc/kernels.h" -nt "$obj" ] || [ "$HERE/csrc/common.h" -nt "$obj" ] || [ "$HERE/csrc/mfma_bf16.h" -nt "$obj" ] || [ "$HERE/csrc/wgrad_batch.h" -nt "$obj" ] || [ "$HERE/../include/cgat_hip.h" -nt "$obj" ]; then
    extra=""
    # MFMA kernels with VALU epilogues: no SLP packing into v_pk_*_f32 (see the note in csrc/edgez.hip)
    if [ "$s" = edgez ] || [ "$s" = edgebwd ] || [ "$s" = bilinear ] || [ "$s" = wgradc ] || [ "$s" = chain ] || [ "$s" = rowsdw ] || [ "$s" = gemmsplit ]; then extra="-fno-slp-vectorize"; fi
    rm -f "$obj"     # a failed compile must not leave a stale object behind for the link step
    "$HIPCC" --offload-arch=gfx950 -O3 -std=c++17 -fPIC $extra -c "$src" -o "$obj" ${CGAT_HIPCC_FLAGS} &
    PIDS+=($!)
  fi
  OBJS+=("$obj")
done
for pid in "${PIDS[@]}"; do
  wait "$pid" || { echo "build_lib.sh: a compile step failed" >&2; exit 1; }
done
"$HIPCC" --offload-arch=gfx950 -shared -fPIC -o "$OUT" "${OBJS[@]}"
echo "built $OUT"
