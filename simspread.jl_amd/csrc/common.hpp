// Shared host-side plumbing of libsimspread_hip: error reporting, device buffers, the
// per-process context (one GPU per process), event-based stage timing.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <atomic>
#include <cstring>
#include <string>
#include <vector>

#include "simspread_hip.h"

namespace ss {

// ---------------------------------------------------------------- errors
std::string& last_error();
int fail(int code, const char* fmt, ...);

#define SS_HIP(call)                                                                  \
  do {                                                                                \
    hipError_t _e = (call);                                                           \
    if (_e != hipSuccess) {                                                           \
      int _c = (_e == hipErrorOutOfMemory) ? SS_ENOMEM : SS_EHIP;                     \
      return ss::fail(_c, "%s:%d %s -> %s", __FILE__, __LINE__, #call,                \
                      hipGetErrorString(_e));                                         \
    }                                                                                 \
  } while (0)

#define SS_TRY(expr)                \
  do {                              \
    int _rc = (expr);               \
    if (_rc != SS_OK) return _rc;   \
  } while (0)

// ---------------------------------------------------------------- context
struct Timing {
  // events of the last timed call; resolved lazily by ss_timing_last()
  std::vector<hipEvent_t> pool;
  size_t used = 0;
  struct Span { int stage; hipEvent_t a, b; };
  std::vector<Span> spans;
  double extra[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  double resolved[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  bool dirty = false;
  bool hold = false;  // ss_timing_hold: calls accumulate instead of replacing one another
  unsigned generation = 0;
  Timing();   // every thread's Timing is listed in a registry, so that ss_shutdown can destroy the events of ALL threads
  ~Timing();  // thread exit: destroy this thread's events (if their context is still alive) and leave the registry
};

struct Ctx {
  bool inited = false;
  int device = -1;
  hipStream_t stream = nullptr;      // the stream everything is enqueued on
  hipStream_t own_stream = nullptr;  // created by ss_init; `stream` may point at a caller's stream instead
  int num_cu = 256;
  size_t lds_per_block = 160 * 1024;
  unsigned generation = 0;           // bumped by ss_shutdown: events created before belong to a dead context
};
Ctx& ctx();
// Timings and the kernel-path note of the last call are per host thread (like ss_last_error): calls from
// several threads on different handles overlap and must not share event lists.
Timing& timing();
std::string& path_note();            // kernels the last predict / spmm call of this thread went through
void path_add(const char* tag);
int require_init();

// stage ids for Timing::Span / ss_timing_last
enum { ST_TOTAL = 0, ST_TRANSFER = 1, ST_SPMM = 2, ST_EPILOGUE = 3, ST_H2D = 4, ST_D2H = 5,
       ST_NSPMM = 6, ST_NTRANSFER = 7 };

void timing_begin_call();
int timing_mark(hipEvent_t* ev);                 // record an event from the pool on the stream
void timing_span(int stage, hipEvent_t a, hipEvent_t b);
void timing_count(int stage, double inc);

// ---------------------------------------------------------------- device memory
template <class T>
struct DevBuf {
  T* p = nullptr;
  size_t n = 0;
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  DevBuf(DevBuf&& o) noexcept : p(o.p), n(o.n) { o.p = nullptr; o.n = 0; }
  DevBuf& operator=(DevBuf&& o) noexcept {
    if (this != &o) { release(); p = o.p; n = o.n; o.p = nullptr; o.n = 0; }
    return *this;
  }
  ~DevBuf() { release(); }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr; n = 0;
  }
  int alloc(size_t count) {
    release();
    if (count == 0) count = 1;  // keep a valid pointer for empty operands
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&p), count * sizeof(T));
    if (e != hipSuccess) {
      p = nullptr;
      return fail(e == hipErrorOutOfMemory ? SS_ENOMEM : SS_EHIP, "hipMalloc(%zu bytes): %s",
                  count * sizeof(T), hipGetErrorString(e));
    }
    n = count;
    return SS_OK;
  }
};

// copy `count` elements from a caller buffer (host or device) to the device, async on the stream
template <class T>
int upload(T* dst, const T* src, size_t count, int mem) {
  if (count == 0) return SS_OK;
  SS_HIP(hipMemcpyAsync(dst, src, count * sizeof(T),
                        mem == SS_MEM_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice,
                        ctx().stream));
  return SS_OK;
}

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

}  // namespace ss
