// Shared host-side helpers of libcgat_hip: error handling, launch timing registry.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define CGAT_OK 0
#define CGAT_ERR_ARG 1
#define CGAT_ERR_HIP 2
#define CGAT_ERR_WORKSPACE 3
#define CGAT_ERR_UNSUPPORTED 4

void cgat_set_error(const char* fmt, ...);

#define CGAT_CHECK_ARG(cond, ...)        \
  do {                                   \
    if (!(cond)) {                       \
      cgat_set_error(__VA_ARGS__);       \
      return CGAT_ERR_ARG;               \
    }                                    \
  } while (0)

#define CGAT_HIP(expr)                                                                   \
  do {                                                                                   \
    hipError_t _e = (expr);                                                              \
    if (_e != hipSuccess) {                                                              \
      cgat_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
      return CGAT_ERR_HIP;                                                               \
    }                                                                                    \
  } while (0)

#define CGAT_TRY(expr)            \
  do {                            \
    int _rc = (expr);             \
    if (_rc != CGAT_OK) return _rc; \
  } while (0)

// ---- launch timing (HIP events on the launch stream; enabled by cgat_prof_enable) ----
struct ProfScope {
  int slot;
  hipStream_t stream;
  ProfScope(const char* tag, hipStream_t s);
  ~ProfScope();
};
#define CGAT_PROF(tag, stream) ProfScope _prof_scope_##__LINE__(tag, stream)

// every kernel launch of the library passes through CGAT_LAUNCH_CHECK: a process-wide launch counter (cgat_prof_launches)
extern unsigned long long g_cgat_launches;
#define CGAT_LAUNCH_CHECK()                                                  \
  do {                                                                       \
    __atomic_fetch_add(&g_cgat_launches, 1ull, __ATOMIC_RELAXED);            \
    hipError_t _e = hipGetLastError();                                       \
    if (_e != hipSuccess) {                                                  \
      cgat_set_error("%s:%d: kernel launch -> %s", __FILE__, __LINE__, hipGetErrorString(_e)); \
      return CGAT_ERR_HIP;                                                   \
    }                                                                        \
  } while (0)

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// bump allocator over a caller-provided device workspace
struct Workspace {
  char* base;
  size_t cap, off;
  bool ok;
  Workspace(void* p, size_t bytes) : base((char*)p), cap(bytes), off(0), ok(true) {}
  template <typename T>
  T* take(size_t n) {
    size_t bytes = (n * sizeof(T) + 255) & ~(size_t)255;
    if (off + bytes > cap) {
      ok = false;
      off += bytes;
      return nullptr;
    }
    T* r = (T*)(base + off);
    off += bytes;
    return r;
  }
};
static inline size_t ws_round(size_t n_elems, size_t elem) { return (n_elems * elem + 255) & ~(size_t)255; }
