// stubs.hip — entry points declared in include/pyqsm_hip.h whose kernels are not
// written yet. They fail loudly (no CPU fallback). Each function moves to its own
// translation unit when implemented.
#include "common.hpp"
using namespace pyqsm;
#define NOT_YET(name) return fail(PYQSM_EINVAL, name ": not implemented in this build")
extern "C" {
#ifndef HAVE_KNN
int pyqsm_knn(const double*, int64_t, int32_t, int32_t, int32_t*, double*, int32_t) { NOT_YET("pyqsm_knn"); }
int pyqsm_knn_dev(const double*, int64_t, int32_t, int32_t, int32_t*, double*, int32_t) { NOT_YET("pyqsm_knn_dev"); }
#endif
#ifndef HAVE_RANSAC
int pyqsm_ransac(const double*, int64_t, const int64_t*, int64_t, int32_t, double, double*, double*, double*, int64_t*, int64_t*, int64_t*, int32_t) { NOT_YET("pyqsm_ransac"); }
int pyqsm_ransac_models(const double*, int64_t, const int64_t*, int64_t, double*, int32_t) { NOT_YET("pyqsm_ransac_models"); }
int pyqsm_ransac_count(const double*, int64_t, const double*, int64_t, int32_t, double, int32_t*, int32_t) { NOT_YET("pyqsm_ransac_count"); }
#endif
#ifndef HAVE_LBC
int pyqsm_lbc_solve(const int32_t*, const int32_t*, const double*, int64_t, const double*, const double*, const double*, double, int32_t, double*, int32_t*, double*, int32_t) { NOT_YET("pyqsm_lbc_solve"); }
int pyqsm_spmv3(const int32_t*, const int32_t*, const double*, int64_t, const double*, double*, int32_t) { NOT_YET("pyqsm_spmv3"); }
int pyqsm_clamp(double*, int64_t, const double*, const double*, int32_t) { NOT_YET("pyqsm_clamp"); }
#endif
#ifndef HAVE_LAPLACIAN
int pyqsm_pc_laplacian(const double*, int64_t, int32_t, double, int64_t*, int32_t**, int32_t**, double**, double*, int32_t) { NOT_YET("pyqsm_pc_laplacian"); }
#endif
}
