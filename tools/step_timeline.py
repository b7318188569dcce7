#!/usr/bin/env python3
"""One layer step (forward + backward) as a timeline, from the rocprofv3 --kernel-trace of the bench command
(tools/collect_profiles.sh: <dir>/stats/*/*_kernel_trace.csv): the last full step, times in ms from the start of the
per-edge forward kernel; kernels under 20 us are folded into the count column of the next listed kernel.
    python tools/step_timeline.py gpurun_out/r04final/stats profiles/r04_final_step_timeline.txt"""
import csv, glob, re, sys

src, dst = sys.argv[1], sys.argv[2]
path = sorted(glob.glob(src + "/*/*_kernel_trace.csv"))[0]
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Queue_Id"])
             for r in csv.DictReader(open(path))), key=lambda e: e[0])
first = [i for i, e in enumerate(ev) if e[2].startswith("void edge_z_kernel<6, true") or e[2].startswith("void edge_z6w_kernel")
         or e[2].startswith("void edge_zx_kernel")]
assert len(first) >= 3, "needs at least three steps in the trace"
a, b = first[-2], first[-1]
step = ev[a:b]
t0 = step[0][0]
queues = sorted({e[3] for e in step})


def short(n):
    n = re.sub(r"^void ", "", n)
    return re.sub(r"\(.*", "", n)


out = ["# One layer step (forward + backward) of the final round-5 build, default mode f16x3c, serial order (no side stream in",
       "# the 24-bit modes): rocprofv3 --kernel-trace of `bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-exclusive-pass",
       "# --no-extra-legs` (tools/collect_profiles.sh, tools/step_timeline.py); the last full step, times in ms from the start of",
       "# the per-edge forward kernel; kernels under 20 us are folded into the count column of the next listed kernel.",
       f"# wall = {(step[-1][1] - t0) / 1e6:.3f} ms, {len(step)} launches, queues {queues}; busy = "
       f"{sum(e[1] - e[0] for e in step) / 1e6:.3f} ms of kernel time, gaps = "
       f"{sum(max(0, step[i + 1][0] - step[i][1]) for i in range(len(step) - 1)) / 1e6:.3f} ms",
       "#  start     end    dur   queue  (+small)  kernel"]
small = 0
for s, e, n, q in step:
    if e - s < 20000:
        small += 1
        continue
    out.append(f"{(s - t0) / 1e6:8.3f} {(e - t0) / 1e6:8.3f} {(e - s) / 1e6:6.3f}   {queues.index(q)}     {small:4d}     {short(n)}")
    small = 0
open(dst, "w").write("\n".join(out) + "\n")
print("\n".join(out[:8]))
