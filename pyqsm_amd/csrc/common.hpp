// common.hpp — per-device context (stream, scratch arena, event timers) and
// error plumbing shared by every translation unit of libpyqsm_hip.so.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/pyqsm_hip.h"

namespace pyqsm {

void set_error(const char* fmt, ...);

// Host buffers the library hands to its caller (released with pyqsm_free). Buffers of a
// megabyte or more come page-locked from a small pool (context.hip): the device-to-host copy
// into them, and the host-to-device copy when the caller passes them back (the Laplacian goes
// straight into the contraction solve), run at link speed instead of through the driver's
// staging of pageable memory. Falls back to malloc; nullptr when that fails too.
void* out_alloc(size_t bytes);
void out_free(void* p);
int fail(int code, const char* fmt, ...);

#define PQ_HIP(expr)                                                                   \
  do {                                                                                 \
    hipError_t e__ = (expr);                                                           \
    if (e__ != hipSuccess)                                                             \
      return ::pyqsm::fail(e__ == hipErrorOutOfMemory ? PYQSM_ENOMEM : PYQSM_EHIP,     \
                           "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__),     \
                           __FILE__, __LINE__);                                        \
  } while (0)

#define PQ_TRY(expr)         \
  do {                       \
    int r__ = (expr);        \
    if (r__ != 0) return r__; \
  } while (0)

// Grow-only scratch memory. alloc() bumps inside the current chunk and adds a
// chunk when it runs out; reset() folds several chunks into one of the summed
// size so that a repeated call of the same shape allocates nothing.
class Arena {
 public:
  int alloc(size_t bytes, void** out);
  template <typename T>
  int get(size_t count, T** out) {
    return alloc(count * sizeof(T), reinterpret_cast<void**>(out));
  }
  int reset();
  void destroy();
  // Position of the allocator; rewind(mark) releases everything allocated after mark() (chunks
  // added meanwhile stay and are reused). For long calls that loop over large temporaries.
  struct Mark {
    size_t chunk, used;
  };
  Mark mark() const;
  void rewind(const Mark& m);

 private:
  struct Chunk {
    char* base;
    size_t size;
    size_t used;
  };
  std::vector<Chunk> chunks_;
  size_t cur_ = 0;  // chunk allocations currently come from
};

struct Timer {
  double ms = 0.0;
  int64_t launches = 0;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
  std::vector<int> weights;  // launches represented by each pending pair
};

struct Ctx {
  int device = -1;
  hipStream_t stream = nullptr;
  Arena arena;
  std::mutex mu;  // calls on one device serialise
  int prof = 0;  // 0 off, 1 phase timers, 2 also single kernels inside the solver loops
  std::map<std::string, Timer> timers;
  std::vector<hipEvent_t> event_pool;
  int cu_count = 256;
};

// Returns the context for `device`, creating it on first use. nullptr + error
// message when the device does not exist.
Ctx* ctx_for(int device);

// RAII bracket: records a start/stop event pair on the stream under `name`
// when profiling is enabled; otherwise does nothing.
class ProfScope {
 public:
  ProfScope(Ctx* c, const char* name, int weight = 1, int level = 1);
  ~ProfScope();

 private:
  Ctx* c_;
  Timer* t_ = nullptr;
  hipEvent_t start_ = nullptr;
};

// roctx range around a C-ABI entry point (tracing): shows up in
// `rocprofv3 --marker-trace`. librocprofiler-sdk-roctx is looked up once at run time;
// without it (or without a tool attached) a range costs two indirect calls.
class ApiRange {
 public:
  explicit ApiRange(const char* name);
  ~ApiRange();

 private:
  bool on_;
};
#define PQ_API_RANGE(name) ::pyqsm::ApiRange api_range__(name)

// Destroys the RCCL communicators (multi.hip); called by pyqsm_shutdown.
void comm_shutdown();

inline int ceil_div(int64_t a, int64_t b) { return static_cast<int>((a + b - 1) / b); }

// Device-resident pieces other translation units chain (skeleton.hip runs the whole contraction
// loop in HBM): kNN, the point-cloud Laplacian, the contraction solve.
struct LapOut {  // all pointers into the context arena
  int32_t *indptr, *indices;
  double *vals, *mass;
  int32_t nnz;
};
int laplacian_device(Ctx* c, const double* d_xyz, int64_t n, const int64_t* seg_start /*host, may be null*/,
                     int64_t n_seg, int32_t k, double moll, LapOut* out);

// Exclusive prefix sum of n int32 values, in place, on the stream (scan.hip).
int exclusive_scan_i32(Ctx* c, int32_t* data, int64_t n);

// Stable sort of n (key, value) pairs by the low `bits` bits of the key (scan.hip: radix passes
// without global atomics, so the order is reproducible). *keys / *vals are the inputs and, on
// return, point at the sorted arrays (the inputs themselves or arena buffers of the same size).
int stable_sort_pairs_u32(Ctx* c, uint32_t** keys, int32_t** vals, int64_t n, int bits);

}  // namespace pyqsm
