// sparse.hpp — CSR handle shared by the contraction solve and its multilevel
// preconditioner.
#pragma once
#include <algorithm>

#include "common.hpp"

namespace pyqsm {

struct DevCsr {
  int32_t *indptr, *indices;
  double* vals;
};

// ---- reproducible reductions ----------------------------------------------------------------
// The dot products of the solver are NOT accumulated with atomics: an fp64 atomicAdd per block
// makes the order of the additions, and with it the last bits of every CG scalar, a matter of
// scheduling, and the contraction loop amplifies those bits into 1e-4 relative differences of the
// skeleton between two runs on the same input. Instead every block of a reducing kernel stores
// its three sums into its own slot of a partial array [3][kPart], and every block of a consuming
// kernel adds the kPart partials up again in one fixed order (12 loads per thread and a block
// reduction; the partials sit in L2). Same input, same bits, whatever the scheduling. It is also
// faster: kPart same-address atomics per launch were served one at a time (~9 ns each, measured:
// 1024 blocks 33.6 us, 4096 blocks 57.6 us for the same pass).
static constexpr int kPart = 1024;  // slots per column; a reducing kernel's grid is at most this

// Grid of a kernel that ends in reduce3_part (its row loop strides by the grid).
inline unsigned reduce_grid(int64_t n) { return unsigned(std::min<int64_t>(std::max<int64_t>(ceil_div(n, 256), 1), kPart)); }

// Block reduction of three per-thread sums into this block's slot of part[3][kPart]; block 0
// clears the slots past the grid. 256 threads per block, gridDim.x <= kPart. Two calls in one
// kernel need a __syncthreads() between them.
__device__ __forceinline__ void reduce3_part(double v0, double v1, double v2, double* __restrict__ part) {
  __shared__ double red[3][4];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    v0 += __shfl_down(v0, off, 64);
    v1 += __shfl_down(v1, off, 64);
    v2 += __shfl_down(v2, off, 64);
  }
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) {
    red[0][w] = v0;
    red[1][w] = v1;
    red[2][w] = v2;
  }
  __syncthreads();
  if (threadIdx.x < 3)
    part[threadIdx.x * kPart + blockIdx.x] = (red[threadIdx.x][0] + red[threadIdx.x][1]) +
                                             (red[threadIdx.x][2] + red[threadIdx.x][3]);
  if (blockIdx.x == 0)
    for (int s = int(gridDim.x) + int(threadIdx.x); s < kPart; s += 256)
      part[s] = part[kPart + s] = part[2 * kPart + s] = 0.0;
}

// The three totals of a partial array: the same bits in every thread of every block. Every
// thread of the (256-thread) block must call it.
__device__ __forceinline__ void part_total3(const double* __restrict__ part, double out[3]) {
  __shared__ double tot[3][4];
  const int t = threadIdx.x;
  double s[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const double* p = part + k * kPart;
    s[k] = (p[t] + p[t + 256]) + (p[t + 512] + p[t + 768]);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    s[0] += __shfl_down(s[0], off, 64);
    s[1] += __shfl_down(s[1], off, 64);
    s[2] += __shfl_down(s[2], off, 64);
  }
  __syncthreads();  // an earlier call's readers are done with tot
  if ((t & 63) == 0) {
    tot[0][t >> 6] = s[0];
    tot[1][t >> 6] = s[1];
    tot[2][t >> 6] = s[2];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 3; ++k) out[k] = (tot[k][0] + tot[k][1]) + (tot[k][2] + tot[k][3]);
}
static_assert(kPart == 1024, "part_total3 adds four partials per thread of a 256-thread block");

// The totals on the host, in a fixed order as well (24 KB per array; the stream is drained).
int part_totals_host(Ctx* c, const double* const* parts, int m, double (*out)[3]);

// Aggregation multigrid for B = c*L + diag(wh) (amg.hip). Opaque to callers.
struct AmgHierarchy;

// Builds the hierarchy for B = diag(cw)*L + diag(wh) of the n-point system (cw [n], constant
// along every edge of L). All device
// memory comes from the context arena (valid until the next arena reset).
int amg_build(Ctx* c, const DevCsr& L, int n, const double* cw, const double* wh, AmgHierarchy** out);
void amg_destroy(AmgHierarchy* h);
int amg_levels(const AmgHierarchy* h);

// x = M^-1 b for three columns: one symmetric V(1,1) cycle (l1-Jacobi smoothing,
// piecewise-constant strength-based aggregation, dense solve on the coarsest level).
// When `dot` is given it is a partial array [3][kPart] that receives b . x per column (fused
// into the last kernel, see reduce3_part).
int amg_vcycle(Ctx* c, AmgHierarchy* h, const double* b, double* x, double* dot = nullptr);
// Floats per row of an fp32 solver vector: (x, y, z, 0), so that a row is one 16-byte access.
static constexpr int kVecStride = 4;

// The same with fp32 vectors of kVecStride floats per row (no conversion pass), and the fp32
// copy of B the cycle uses on its finest level (for a CG that runs in fp32 altogether).
int amg_vcycle_f32(Ctx* c, AmgHierarchy* h, const float* b, float* x, double* dot = nullptr);
void amg_fine_matrix(const AmgHierarchy* h, const int32_t** indptr, const int32_t** indices,
                     const float** vals);

}  // namespace pyqsm
