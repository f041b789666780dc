// scan.hip — device-wide exclusive prefix sum of int32 (three-phase: per-block
// reduce, recursive scan of the block sums, per-block scan + offset). HBM-bound:
// 2 reads + 1 write of the array.
#include "common.hpp"

namespace pyqsm {

static constexpr int kScanThreads = 256;
static constexpr int kScanItems = 8;
static constexpr int kScanTile = kScanThreads * kScanItems;  // 2048 per block

__device__ __forceinline__ int32_t wave_incl_scan(int32_t v) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    int32_t t = __shfl_up(v, off, 64);
    if (lane >= off) v += t;
  }
  return v;
}

// Exclusive scan of one value per thread across a 256-thread block; returns the
// exclusive prefix and writes the block total to *total.
__device__ __forceinline__ int32_t block_excl_scan(int32_t v, int32_t* total) {
  __shared__ int32_t wsum[kScanThreads / 64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int32_t incl = wave_incl_scan(v);
  if (lane == 63) wsum[w] = incl;
  __syncthreads();
  int32_t base = 0, tot = 0;
#pragma unroll
  for (int i = 0; i < kScanThreads / 64; ++i) {
    if (i < w) base += wsum[i];
    tot += wsum[i];
  }
  *total = tot;
  __syncthreads();
  return base + incl - v;
}

__global__ __launch_bounds__(kScanThreads) void k_scan_reduce(const int32_t* __restrict__ in,
                                                              int64_t n,
                                                              int32_t* __restrict__ sums) {
  const int64_t base = int64_t(blockIdx.x) * kScanTile;
  int32_t s = 0;
#pragma unroll
  for (int k = 0; k < kScanItems; ++k) {
    int64_t i = base + int64_t(k) * kScanThreads + threadIdx.x;
    if (i < n) s += in[i];
  }
  int32_t tot;
  (void)block_excl_scan(s, &tot);
  if (threadIdx.x == 0) sums[blockIdx.x] = tot;
}

__global__ __launch_bounds__(kScanThreads) void k_scan_apply(int32_t* __restrict__ data, int64_t n,
                                                             const int32_t* __restrict__ offs) {
  // Each thread owns kScanItems consecutive elements of the tile.
  const int64_t base = int64_t(blockIdx.x) * kScanTile + int64_t(threadIdx.x) * kScanItems;
  int32_t v[kScanItems];
  int32_t s = 0;
#pragma unroll
  for (int k = 0; k < kScanItems; ++k) {
    v[k] = (base + k < n) ? data[base + k] : 0;
    s += v[k];
  }
  int32_t tot;
  int32_t run = block_excl_scan(s, &tot) + (offs ? offs[blockIdx.x] : 0);
#pragma unroll
  for (int k = 0; k < kScanItems; ++k) {
    if (base + k < n) data[base + k] = run;
    run += v[k];
  }
}

int exclusive_scan_i32(Ctx* c, int32_t* data, int64_t n) {
  if (n <= 0) return 0;
  const int64_t nb = (n + kScanTile - 1) / kScanTile;
  if (nb == 1) {
    hipLaunchKernelGGL(k_scan_apply, dim3(1), dim3(kScanThreads), 0, c->stream, data, n,
                       static_cast<const int32_t*>(nullptr));
    PQ_HIP(hipGetLastError());
    return 0;
  }
  int32_t* sums = nullptr;
  PQ_TRY(c->arena.get(size_t(nb), &sums));
  hipLaunchKernelGGL(k_scan_reduce, dim3(unsigned(nb)), dim3(kScanThreads), 0, c->stream, data, n,
                     sums);
  PQ_HIP(hipGetLastError());
  PQ_TRY(exclusive_scan_i32(c, sums, nb));
  hipLaunchKernelGGL(k_scan_apply, dim3(unsigned(nb)), dim3(kScanThreads), 0, c->stream, data, n,
                     sums);
  PQ_HIP(hipGetLastError());
  return 0;
}

}  // namespace pyqsm
