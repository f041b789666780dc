// sparse.hpp — CSR handle shared by the contraction solve and its multilevel
// preconditioner.
#pragma once
#include "common.hpp"

namespace pyqsm {

struct DevCsr {
  int32_t *indptr, *indices;
  double* vals;
};

// Aggregation multigrid for B = c*L + diag(wh) (amg.hip). Opaque to callers.
struct AmgHierarchy;

// Builds the hierarchy for the n-point system; `xyz` (f64 [n,3], device) are the
// current positions and only steer which points are aggregated together. All
// device memory comes from the context arena (valid until the next arena reset).
int amg_build(Ctx* c, const DevCsr& L, int n, double cw, const double* wh, const double* xyz,
              AmgHierarchy** out);
void amg_destroy(AmgHierarchy* h);
int amg_levels(const AmgHierarchy* h);

// x = M^-1 b for three columns: one symmetric V(1,1) cycle (l1-Jacobi smoothing,
// piecewise-constant aggregation, dense solve on the coarsest level).
int amg_vcycle(Ctx* c, AmgHierarchy* h, const double* b, double* x);

}  // namespace pyqsm
