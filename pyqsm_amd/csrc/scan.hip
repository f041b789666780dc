// scan.hip — device-wide exclusive prefix sum of int32 (three-phase: per-block
// reduce, recursive scan of the block sums, per-block scan + offset). HBM-bound:
// 2 reads + 1 write of the array, all as whole cache lines per wave.
#include <utility>

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

// Each thread owns two groups of four consecutive elements, one in each half of the tile,
// loaded and stored as 16-byte vectors: a wave's access is 1 KB contiguous = 8 cache lines.
// (Eight consecutive scalars per thread made every one of the 8 loads of a wave span 2 KB:
// 128 line accesses for 16 lines of data, and 2.1 TB/s on a 16 M-cell grid.)
__global__ __launch_bounds__(kScanThreads) void k_scan_apply(int32_t* __restrict__ data, int64_t n,
                                                             const int32_t* __restrict__ offs) {
  const int64_t tile = int64_t(blockIdx.x) * kScanTile;
  int4 v[2];
  int32_t s[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int64_t i = tile + int64_t(j * kScanThreads + threadIdx.x) * 4;
    if (i + 3 < n) {
      v[j] = *reinterpret_cast<const int4*>(data + i);
    } else {
      v[j].x = i < n ? data[i] : 0;
      v[j].y = i + 1 < n ? data[i + 1] : 0;
      v[j].z = i + 2 < n ? data[i + 2] : 0;
      v[j].w = 0;
    }
    s[j] = v[j].x + v[j].y + v[j].z + v[j].w;
  }
  int32_t tot0, tot1;
  const int32_t r0 = block_excl_scan(s[0], &tot0);
  const int32_t r1 = block_excl_scan(s[1], &tot1);
  const int32_t off = offs ? offs[blockIdx.x] : 0;
  const int32_t run[2] = {off + r0, off + tot0 + r1};
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int64_t i = tile + int64_t(j * kScanThreads + threadIdx.x) * 4;
    int4 o;
    o.x = run[j];
    o.y = o.x + v[j].x;
    o.z = o.y + v[j].y;
    o.w = o.z + v[j].z;
    if (i + 3 < n) {
      *reinterpret_cast<int4*>(data + i) = o;
    } else {
      if (i < n) data[i] = o.x;
      if (i + 1 < n) data[i + 1] = o.y;
      if (i + 2 < n) data[i + 2] = o.z;
    }
  }
}

// the same for arrays that do not start on a 16-byte boundary
__global__ __launch_bounds__(kScanThreads) void k_scan_apply_unaligned(int32_t* __restrict__ data, int64_t n,
                                                                       const int32_t* __restrict__ offs) {
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

static void launch_apply(Ctx* c, int64_t nb, int32_t* data, int64_t n, const int32_t* offs) {
  if ((reinterpret_cast<uintptr_t>(data) & 15) == 0)
    hipLaunchKernelGGL(k_scan_apply, dim3(unsigned(nb)), dim3(kScanThreads), 0, c->stream, data, n, offs);
  else
    hipLaunchKernelGGL(k_scan_apply_unaligned, dim3(unsigned(nb)), dim3(kScanThreads), 0, c->stream, data, n,
                       offs);
}

int exclusive_scan_i32(Ctx* c, int32_t* data, int64_t n) {
  if (n <= 0) return 0;
  const int64_t nb = (n + kScanTile - 1) / kScanTile;
  if (nb == 1) {
    launch_apply(c, 1, data, n, nullptr);
    PQ_HIP(hipGetLastError());
    return 0;
  }
  int32_t* sums = nullptr;
  PQ_TRY(c->arena.get(size_t(nb), &sums));
  hipLaunchKernelGGL(k_scan_reduce, dim3(unsigned(nb)), dim3(kScanThreads), 0, c->stream, data, n,
                     sums);
  PQ_HIP(hipGetLastError());
  PQ_TRY(exclusive_scan_i32(c, sums, nb));
  launch_apply(c, nb, data, n, sums);
  PQ_HIP(hipGetLastError());
  return 0;
}

// ---- stable radix sort of (key, value) pairs ----------------------------------------------------
// Least-significant-digit passes of 8 bits; each pass is a per-tile digit histogram, the scan
// above over [digit][tile], and a scatter in which a pair's place among the equal digits of its
// tile follows its position in the tile (ballot ranks inside a wave, waves and rounds in order).
// No global atomics: the result is THE stable order, the same bits on every run — which is what
// the contraction solve needs from its spatial ordering (lbc.hip), where an order that depends
// on the scheduling of atomics ends up in the rounding of every dot product.

static constexpr int kSortRounds = 8;                        // rounds of 256 pairs per block
static constexpr int kSortTile = kScanThreads * kSortRounds;  // 2048

__global__ __launch_bounds__(256) void k_radix_hist(const uint32_t* __restrict__ keys, int64_t n,
                                                    int shift, int32_t* __restrict__ bh, int nb) {
  __shared__ int h[256];
  h[threadIdx.x] = 0;
  __syncthreads();
  const int64_t base = int64_t(blockIdx.x) * kSortTile;
#pragma unroll
  for (int r = 0; r < kSortRounds; ++r) {
    const int64_t i = base + r * 256 + threadIdx.x;
    if (i < n) atomicAdd(&h[(keys[i] >> shift) & 255u], 1);  // LDS; counts do not depend on the order
  }
  __syncthreads();
  bh[size_t(threadIdx.x) * nb + blockIdx.x] = h[threadIdx.x];
}

__global__ __launch_bounds__(256) void k_radix_scatter(const uint32_t* __restrict__ keys,
                                                       const int32_t* __restrict__ vals, int64_t n,
                                                       int shift, const int32_t* __restrict__ bh, int nb,
                                                       uint32_t* __restrict__ keys_out,
                                                       int32_t* __restrict__ vals_out) {
  __shared__ int base[256];     // next free place of each digit for this tile
  __shared__ int wcnt[4][256];  // digit counts of the four waves in the current round
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  base[threadIdx.x] = bh[size_t(threadIdx.x) * nb + blockIdx.x];
  const int64_t tile = int64_t(blockIdx.x) * kSortTile;
  for (int r = 0; r < kSortRounds; ++r) {
#pragma unroll
    for (int q = 0; q < 4; ++q) wcnt[q][threadIdx.x] = 0;
    __syncthreads();
    const int64_t i = tile + r * 256 + threadIdx.x;
    const bool live = i < n;
    const uint32_t key = live ? keys[i] : 0u;
    const int val = live ? vals[i] : 0;
    const int digit = int((key >> shift) & 255u);
    unsigned long long peers = __ballot(live);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const bool bit = (digit >> b) & 1;
      const unsigned long long bal = __ballot(bit);
      peers &= bit ? bal : ~bal;
    }
    const int rank = __popcll(peers & ((1ull << lane) - 1ull));
    if (live && rank == 0) wcnt[w][digit] = __popcll(peers);
    __syncthreads();
    if (live) {
      int off = 0;
      for (int q = 0; q < w; ++q) off += wcnt[q][digit];
      const int pos = base[digit] + off + rank;
      keys_out[pos] = key;
      vals_out[pos] = val;
    }
    __syncthreads();
    base[threadIdx.x] += (wcnt[0][threadIdx.x] + wcnt[1][threadIdx.x]) +
                         (wcnt[2][threadIdx.x] + wcnt[3][threadIdx.x]);
    __syncthreads();
  }
}

int stable_sort_pairs_u32(Ctx* c, uint32_t** keys, int32_t** vals, int64_t n, int bits) {
  if (n <= 1 || bits <= 0) return 0;
  if (n > 0x7FFFFF00LL) return fail(PYQSM_ERANGE, "more than 2^31 pairs to sort");
  const int nb = int((n + kSortTile - 1) / kSortTile);
  uint32_t* k2 = nullptr;
  int32_t *v2 = nullptr, *bh = nullptr;
  PQ_TRY(c->arena.get(size_t(n), &k2));
  PQ_TRY(c->arena.get(size_t(n), &v2));
  PQ_TRY(c->arena.get(size_t(256) * nb + 1, &bh));
  uint32_t *ka = *keys, *kb = k2;
  int32_t *va = *vals, *vb = v2;
  for (int shift = 0; shift < bits; shift += 8) {
    hipLaunchKernelGGL(k_radix_hist, dim3(unsigned(nb)), dim3(256), 0, c->stream, ka, n, shift, bh, nb);
    PQ_HIP(hipGetLastError());
    PQ_TRY(exclusive_scan_i32(c, bh, int64_t(256) * nb));
    hipLaunchKernelGGL(k_radix_scatter, dim3(unsigned(nb)), dim3(256), 0, c->stream, ka, va, n, shift, bh,
                       nb, kb, vb);
    PQ_HIP(hipGetLastError());
    std::swap(ka, kb);
    std::swap(va, vb);
  }
  *keys = ka;
  *vals = va;
  return 0;
}

}  // namespace pyqsm
