// sparse.hpp — CSR handle shared by the contraction solve and its multilevel
// preconditioner.
#pragma once
#include "common.hpp"

namespace pyqsm {

struct DevCsr {
  int32_t *indptr, *indices;
  double* vals;
};

// Block reduction of three partial sums followed by one fp64 atomic per column.
__device__ __forceinline__ void reduce3_atomic(double v0, double v1, double v2, double* out) {
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
  if (threadIdx.x < 3) {
    double t = (red[threadIdx.x][0] + red[threadIdx.x][1]) +
               (red[threadIdx.x][2] + red[threadIdx.x][3]);
    atomicAdd(out + threadIdx.x, t);
  }
}

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
// When `dot` is given, dot[0..2] += b . x per column (fused into the last kernel).
int amg_vcycle(Ctx* c, AmgHierarchy* h, const double* b, double* x, double* dot = nullptr);
// Floats per row of an fp32 solver vector: (x, y, z, 0), so that a row is one 16-byte access.
static constexpr int kVecStride = 4;

// The same with fp32 vectors of kVecStride floats per row (no conversion pass), and the fp32
// copy of B the cycle uses on its finest level (for a CG that runs in fp32 altogether).
int amg_vcycle_f32(Ctx* c, AmgHierarchy* h, const float* b, float* x, double* dot = nullptr);
void amg_fine_matrix(const AmgHierarchy* h, const int32_t** indptr, const int32_t** indices,
                     const float** vals);

}  // namespace pyqsm
