// skeleton.hip — the Laplacian-contraction loop of pyQSM/geometry/skeletonize.py:240-373
// (extract_skeleton) with every array resident in HBM for the whole loop: point-cloud Laplacian
// -> contraction solve -> clamp -> weight update -> next Laplacian, 20 times, without the
// matrix or the points crossing PCIe in between (the Python loop moves ~200 MB per step at one
// million points and spends ~8 % of the run in NumPy).
//
// One call contracts one cloud or SEVERAL stacked clouds (segments): one Laplacian build and
// one block-diagonal solve per step serve all of them while weights (:264-265, :329-335), the
// clamp box (:291-296), the volume ratio and the termination (:279, :349, :353) stay per cloud.
// The loop's quirks are kept: the positional weights are updated with the mass of the Laplacian
// just USED, and the volume ratio compares that mass with the first one (it lags one step).
#include <chrono>
#include "sparse.hpp"

#include <cmath>

namespace pyqsm {

int lbc_solve_device(Ctx* c, const DevCsr& L, int64_t n, const double* wl, bool wl_edge_const,
                     const double* wh, const double* pts, double rtol, int32_t max_it, double* x,
                     int32_t* iters, double resid[3]);

// segment of every point (segments are contiguous: bisection over the offsets)
__global__ __launch_bounds__(256) void k_seg_of(int n, int S, const int32_t* __restrict__ seg_start,
                                                int32_t* __restrict__ seg_of) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  int lo = 0, hi = S;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (seg_start[mid] <= i) lo = mid; else hi = mid;
  }
  seg_of[i] = lo;
}

__global__ __launch_bounds__(256) void k_fill_by_seg(int n, const int32_t* __restrict__ seg_of,
                                                     const double* __restrict__ per_seg,
                                                     double* __restrict__ out) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = per_seg[seg_of[i]];
}

__global__ __launch_bounds__(256) void k_fill_const(int n, double v, double* __restrict__ out) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = v;
}

// flags[s]: bit 0 = some coordinate of the new positions differs from the current ones
// ((new == cur).all() of :287 is false), bit 1 = some coordinate of the new positions is not NaN
// (least_squares_sparse returns the input when EVERYTHING is NaN, :177-179)
__global__ __launch_bounds__(256) void k_seg_flags(int n, const int32_t* __restrict__ seg_of,
                                                   const double* __restrict__ nw,
                                                   const double* __restrict__ cur,
                                                   int32_t* __restrict__ flags) {
  int i = blockIdx.x * 256 + threadIdx.x;
  int f = 0, sg = -1;
  if (i < n) {
    sg = seg_of[i];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const double a = nw[3 * size_t(i) + k], b = cur[3 * size_t(i) + k];
      if (!(a == b)) f |= 1;
      if (a == a) f |= 2;
    }
  }
  // One atomic per WAVE at most, none once the bits are there: every point OR-ing its bits into
  // its cloud's word was a million atomics on ONE address for a single cloud — served one at a
  // time, 11 ms per contraction step, 0.23 s of a 2.8 s loop.
  const unsigned long long live = __ballot(i < n);
  if (live == 0ull) return;
  const int lead = __ffsll(live) - 1;
  const int s0 = __shfl(sg, lead, 64);
  if (__ballot(i < n && sg != s0) == 0ull) {  // the usual wave: one cloud
    const int wf = (__ballot(f & 1) != 0ull ? 1 : 0) | (__ballot(f & 2) != 0ull ? 2 : 0);
    if ((threadIdx.x & 63) == lead && wf && (flags[s0] & wf) != wf) atomicOr(&flags[s0], wf);
  } else if (f && (flags[sg] & f) != f) {
    atomicOr(&flags[sg], f);
  }
}

// :291-307 for the clouds still active: clamp into the cloud's box, shift = cur - new,
// total += shift, cur = new. Inactive clouds keep their state (shift 0).
__global__ __launch_bounds__(256) void k_step(int n, const int32_t* __restrict__ seg_of,
                                              const int32_t* __restrict__ active,
                                              const double* __restrict__ lo /*[S,3]*/,
                                              const double* __restrict__ hi,
                                              const double* __restrict__ nw, double* __restrict__ cur,
                                              double* __restrict__ total,
                                              double* __restrict__ shift /*may be null*/) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int s = seg_of[i];
  const bool on = active[s] != 0;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const size_t q = 3 * size_t(i) + k;
    double sh = 0.0;
    if (on) {
      double v = nw[q];
      const double l = lo[3 * s + k], h = hi[3 * s + k];
      if (v < l) v = l;  // NaN stays NaN, as in the reference's comparisons
      if (v > h) v = h;
      sh = cur[q] - v;
      total[q] += sh;
      cur[q] = v;
    }
    if (shift) shift[q] = sh;
  }
}

// :331,335 positional weights of the active clouds: wh *= sqrt(M0 / M_used), clipped
__global__ __launch_bounds__(256) void k_wh_update(int n, const int32_t* __restrict__ seg_of,
                                                   const int32_t* __restrict__ active,
                                                   const double* __restrict__ m0,
                                                   const double* __restrict__ m_used,
                                                   double max_attraction, double* __restrict__ wh) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n || !active[seg_of[i]]) return;
  double w = wh[i] * sqrt(m0[i] / m_used[i]);
  w = w < 0.1 ? 0.1 : w;             // np.clip(x, 0.1, max): NaN passes through
  w = w > max_attraction ? max_attraction : w;
  wh[i] = w;
}

struct DevBlock {  // hipMalloc'ed memory of one call, released on every exit path
  std::vector<void*> ptrs;
  ~DevBlock() {
    for (void* p : ptrs) (void)hipFree(p);
  }
  template <typename T>
  int get(size_t count, T** out) {
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, (count ? count : 1) * sizeof(T));
    if (e != hipSuccess) return fail(PYQSM_ENOMEM, "hipMalloc(%zu) failed: %s", count * sizeof(T), hipGetErrorString(e));
    ptrs.push_back(p);
    *out = static_cast<T*>(p);
    return 0;
  }
  void drop(void* p) {
    for (auto& q : ptrs)
      if (q == p) {
        (void)hipFree(q);
        q = nullptr;
      }
  }
};

}  // namespace pyqsm

using namespace pyqsm;

// NumPy's pairwise summation (numpy/_core/src/umath/loops_utils.h.src, @TYPE@_pairwise_sum, as
// published: below 8 values a plain loop; up to 128 values eight running sums over the multiples
// of 8, combined as ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)), then the rest one by one; above that the
// array is split at n/2 rounded down to a multiple of 8). Restated here, not linked: the point is
// the ORDER of the additions.
static double np_pairwise_sum(const double* a, int64_t n) {
  if (n < 8) {
    double res = 0.0;
    for (int64_t i = 0; i < n; ++i) res += a[i];
    return res;
  }
  if (n <= 128) {
    double r[8];
    for (int j = 0; j < 8; ++j) r[j] = a[j];
    int64_t i = 8;
    for (; i < n - (n % 8); i += 8)
      for (int j = 0; j < 8; ++j) r[j] += a[i + j];
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res += a[i];
    return res;
  }
  int64_t n2 = n / 2;
  n2 -= n2 % 8;
  return np_pairwise_sum(a, n2) + np_pairwise_sum(a + n2, n - n2);
}

extern "C" {

int pyqsm_mean_f64(const double* v, int64_t n, double* out) {
  if (!out) return fail(PYQSM_EINVAL, "null output");
  if (n <= 0 || !v) {
    *out = std::nan("");
    return 0;
  }
  // np.add.reduce hands its inner loop at most one buffer (8192 elements, np.getbufsize()'s
  // default) at a time and adds the pieces' pairwise sums up in order (NumPy 2.2 checked: the
  // whole-array pairwise sum differs from np.sum in the last bit for some n > 8192, this does not)
  double acc = 0.0;
  for (int64_t b = 0; b < n; b += 8192) acc += np_pairwise_sum(v + b, std::min<int64_t>(8192, n - b));
  *out = acc / double(n);
  return 0;
}

int pyqsm_extract_skeleton(const double* xyz, int64_t n, const int64_t* seg_start, int64_t n_seg,
                           int32_t k, double moll, int32_t max_iter, double termination_ratio,
                           double contraction_factor, double attraction_factor,
                           double max_contraction, double max_attraction, const double* lo,
                           const double* hi, double rtol, int32_t solver_max_it, double* out_pts,
                           double* total_shift, double* steps, int32_t* n_steps,
                           int32_t* solve_iters, double* solve_resid, uint8_t* solve_ok,
                           int32_t* n_solves, int32_t device) {
  PQ_API_RANGE("pyqsm_extract_skeleton");
  if (n_solves) *n_solves = 0;
  if (n < 0 || n_seg < 1 || max_iter < 0) return fail(PYQSM_EINVAL, "negative size");
  if (n > 0x7FFFFF00LL) return fail(PYQSM_ERANGE, "more than 2^31 points");
  if (!lo || !hi || !n_steps || (n > 0 && (!xyz || !out_pts || !total_shift)))
    return fail(PYQSM_EINVAL, "pyqsm_extract_skeleton: NULL pointer");
  if (k < 3 || k > 64) return fail(PYQSM_ERANGE, "n_neighbors must be in [3, 64]");
  if (!(moll >= 0) || !(rtol > 0)) return fail(PYQSM_EINVAL, "mollify factor >= 0 and rtol > 0 required");
  const int S = int(n_seg);
  std::vector<int64_t> one{0, n};
  if (S == 1 && !seg_start) seg_start = one.data();
  if (!seg_start || seg_start[0] != 0 || seg_start[S] != n)
    return fail(PYQSM_EINVAL, "seg_start must run from 0 to n");
  for (int s = 0; s < S; ++s) {
    if (seg_start[s + 1] <= seg_start[s]) return fail(PYQSM_EINVAL, "empty or unordered segment");
    n_steps[s] = 0;
  }
  if (n == 0) return 0;
  Ctx* c = ctx_for(device);
  if (!c) return PYQSM_ENODEV;
  std::lock_guard<std::mutex> lk(c->mu);
  c->arena.reset();
  // host-side wall clock of the loop's phases (PYQSM_LBC_TRACE)
  const bool trace_t = getenv("PYQSM_LBC_TRACE") != nullptr;
  double t_acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};  // setup, build, solve, means, step, tail, flags, kstep, copy
  auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double t_mark = now();
  auto lap = [&](int k) {
    const double t = now();
    t_acc[k] += t - t_mark;
    t_mark = t;
  };
  const int N = int(n);
  const dim3 gn(ceil_div(n, 256)), blk(256);
  DevBlock mem;
  double *cur, *nw, *total, *wh, *wl, *m0, *m_used, *shift = nullptr, *d_lo, *d_hi, *d_segval;
  int32_t *seg_of, *d_seg_start, *d_active, *d_flags;
  PQ_TRY(mem.get(size_t(n) * 3, &cur));
  PQ_TRY(mem.get(size_t(n) * 3, &nw));
  PQ_TRY(mem.get(size_t(n) * 3, &total));
  PQ_TRY(mem.get(size_t(n), &wh));
  PQ_TRY(mem.get(size_t(n), &wl));
  PQ_TRY(mem.get(size_t(n), &m0));
  PQ_TRY(mem.get(size_t(n), &m_used));
  if (steps) PQ_TRY(mem.get(size_t(n) * 3, &shift));
  PQ_TRY(mem.get(size_t(S) * 3, &d_lo));
  PQ_TRY(mem.get(size_t(S) * 3, &d_hi));
  PQ_TRY(mem.get(size_t(S), &d_segval));
  PQ_TRY(mem.get(size_t(n), &seg_of));
  PQ_TRY(mem.get(size_t(S) + 1, &d_seg_start));
  PQ_TRY(mem.get(size_t(S), &d_active));
  PQ_TRY(mem.get(size_t(S), &d_flags));
  std::vector<int32_t> h_start(size_t(S) + 1);
  for (int s = 0; s <= S; ++s) h_start[size_t(s)] = int32_t(seg_start[s]);
  PQ_HIP(hipMemcpyAsync(cur, xyz, size_t(n) * 24, hipMemcpyHostToDevice, c->stream));
  PQ_HIP(hipMemcpyAsync(d_lo, lo, size_t(S) * 24, hipMemcpyHostToDevice, c->stream));
  PQ_HIP(hipMemcpyAsync(d_hi, hi, size_t(S) * 24, hipMemcpyHostToDevice, c->stream));
  PQ_HIP(hipMemcpyAsync(d_seg_start, h_start.data(), (size_t(S) + 1) * 4, hipMemcpyHostToDevice, c->stream));
  PQ_HIP(hipMemsetAsync(total, 0, size_t(n) * 24, c->stream));
  hipLaunchKernelGGL(k_seg_of, gn, blk, 0, c->stream, N, S, d_seg_start, seg_of);
  hipLaunchKernelGGL(k_fill_const, gn, blk, 0, c->stream, N, attraction_factor, wh);  // :264
  PQ_HIP(hipGetLastError());
  PQ_HIP(hipStreamSynchronize(c->stream));  // h_start may go; inputs are on the device

  // the Laplacian of the current points, kept outside the arena between build and solve
  DevCsr Lp{nullptr, nullptr, nullptr};
  size_t cap = 0;
  PQ_TRY(mem.get(size_t(n) + 1, &Lp.indptr));
  const Arena::Mark base = c->arena.mark();
  auto build = [&]() -> int {
    LapOut lo_;
    PQ_TRY(laplacian_device(c, cur, n, seg_start, S, k, moll, &lo_));
    if (size_t(lo_.nnz) + 1 > cap) {
      if (Lp.indices) mem.drop(Lp.indices);
      if (Lp.vals) mem.drop(Lp.vals);
      cap = size_t(double(lo_.nnz) * 1.25) + 1024;
      PQ_TRY(mem.get(cap, &Lp.indices));
      PQ_TRY(mem.get(cap, &Lp.vals));
    }
    PQ_HIP(hipMemcpyAsync(Lp.indptr, lo_.indptr, (size_t(n) + 1) * 4, hipMemcpyDeviceToDevice, c->stream));
    PQ_HIP(hipMemcpyAsync(Lp.indices, lo_.indices, size_t(lo_.nnz) * 4, hipMemcpyDeviceToDevice, c->stream));
    PQ_HIP(hipMemcpyAsync(Lp.vals, lo_.vals, size_t(lo_.nnz) * 8, hipMemcpyDeviceToDevice, c->stream));
    PQ_HIP(hipMemcpyAsync(m_used, lo_.mass, size_t(n) * 8, hipMemcpyDeviceToDevice, c->stream));
    PQ_HIP(hipStreamSynchronize(c->stream));
    c->arena.rewind(base);
    return 0;
  };
  std::vector<double> mean0(static_cast<size_t>(S)), mean_used(static_cast<size_t>(S)),
      wl_seg(static_cast<size_t>(S)), vr(static_cast<size_t>(S), 1.0);
  std::vector<int32_t> active(static_cast<size_t>(S), 1), iteration(static_cast<size_t>(S), 0),
      h_flags(static_cast<size_t>(S));
  // per-cloud means with NumPy's summation order (pyqsm_mean_f64: what np.mean of the Python loop
  // returns, bit for bit); 8 bytes per point come to the host for it, once per step
  std::vector<double> h_vec(static_cast<size_t>(n));
  auto seg_means = [&](const double* v, std::vector<double>& out) -> int {
    PQ_HIP(hipMemcpyAsync(h_vec.data(), v, size_t(n) * 8, hipMemcpyDeviceToHost, c->stream));
    PQ_HIP(hipStreamSynchronize(c->stream));
    for (int s = 0; s < S; ++s)
      pyqsm_mean_f64(h_vec.data() + seg_start[s], seg_start[s + 1] - seg_start[s], &out[size_t(s)]);
    return 0;
  };
  lap(0);
  PQ_TRY(build());                                                                    // :253-255
  lap(1);
  PQ_HIP(hipMemcpyAsync(m0, m_used, size_t(n) * 8, hipMemcpyDeviceToDevice, c->stream));  // M_list[0]
  PQ_TRY(seg_means(m0, mean0));
  lap(3);
  for (int s = 0; s < S; ++s)
    wl_seg[size_t(s)] = contraction_factor * 1000.0 * std::sqrt(mean0[size_t(s)]);     // :265
  int step = 0;
  auto any_active = [&]() {
    for (int s = 0; s < S; ++s)
      if (active[size_t(s)]) return true;
    return false;
  };
  const int step_cap = max_iter > 0 ? max_iter : 1;  // the reference always runs its first pass
  while (any_active() && step < step_cap) {
    for (int s = 0; s < S; ++s)
      if (!(vr[size_t(s)] > termination_ratio)) active[size_t(s)] = 0;                 // :279
    if (!any_active()) break;
    // ---- solve (:281-285) ------------------------------------------------------------
    PQ_HIP(hipMemcpyAsync(d_segval, wl_seg.data(), size_t(S) * 8, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_fill_by_seg, gn, blk, 0, c->stream, N, seg_of, d_segval, wl);
    PQ_HIP(hipGetLastError());
    int32_t it = 0;
    double rs[3] = {0, 0, 0};
    int rc = lbc_solve_device(c, Lp, n, wl, true, wh, cur, rtol, solver_max_it, nw, &it, rs);
    if (rc != 0 && rc != PYQSM_ENOCONV) return rc;
    PQ_HIP(hipStreamSynchronize(c->stream));
    lap(2);
    c->arena.rewind(base);
    if (solve_iters) solve_iters[step] = it;
    if (solve_resid) solve_resid[step] = std::max(rs[0], std::max(rs[1], rs[2]));
    if (solve_ok) solve_ok[step] = rc == 0;
    // ---- unchanged / all-NaN clouds stop (:177-179, :287-289) ------------------------
    PQ_HIP(hipMemsetAsync(d_flags, 0, size_t(S) * 4, c->stream));
    hipLaunchKernelGGL(k_seg_flags, gn, blk, 0, c->stream, N, seg_of, nw, cur, d_flags);
    PQ_HIP(hipGetLastError());
    PQ_HIP(hipMemcpyAsync(h_flags.data(), d_flags, size_t(S) * 4, hipMemcpyDeviceToHost, c->stream));
    PQ_HIP(hipStreamSynchronize(c->stream));
    lap(6);
    for (int s = 0; s < S; ++s)
      if (active[size_t(s)] && (!(h_flags[size_t(s)] & 1) || !(h_flags[size_t(s)] & 2))) active[size_t(s)] = 0;
    ++step;
    if (n_solves) *n_solves = step;
    if (!any_active()) {
      // the reference breaks before recording anything for this step
      if (steps) memset(steps + size_t(step - 1) * size_t(n) * 3, 0, size_t(n) * 24);
      break;
    }
    // ---- clamp, shift, accumulate (:291-307) -----------------------------------------
    PQ_HIP(hipMemcpyAsync(d_active, active.data(), size_t(S) * 4, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_step, gn, blk, 0, c->stream, N, seg_of, d_active, d_lo, d_hi, nw, cur, total, shift);
    lap(7);
    if (steps)
      PQ_HIP(hipMemcpyAsync(steps + size_t(step - 1) * size_t(n) * 3, shift, size_t(n) * 24,
                            hipMemcpyDeviceToHost, c->stream));
    lap(8);
    // ---- weights (:329-335) with the mass of the Laplacian just used ---------------------
    hipLaunchKernelGGL(k_wh_update, gn, blk, 0, c->stream, N, seg_of, d_active, m0, m_used, max_attraction, wh);
    PQ_HIP(hipGetLastError());
    lap(4);
    PQ_TRY(seg_means(m_used, mean_used));                                              // M_list[-1] of :337
    lap(3);
    for (int s = 0; s < S; ++s)
      if (active[size_t(s)]) {
        double w = wl_seg[size_t(s)] * contraction_factor;
        w = w < 0.1 ? 0.1 : w;
        w = w > max_contraction ? max_contraction : w;
        wl_seg[size_t(s)] = w;
        ++iteration[size_t(s)];
        ++n_steps[s];
      }
    PQ_TRY(build());                                                                  // :341-343
    lap(1);
    for (int s = 0; s < S; ++s)
      if (active[size_t(s)]) {
        vr[size_t(s)] = mean_used[size_t(s)] / mean0[size_t(s)];                       // :349 (lags a step)
        if (iteration[size_t(s)] >= max_iter) active[size_t(s)] = 0;                   // :353-360
      }
  }
  lap(4);
  PQ_HIP(hipMemcpyAsync(out_pts, cur, size_t(n) * 24, hipMemcpyDeviceToHost, c->stream));
  PQ_HIP(hipMemcpyAsync(total_shift, total, size_t(n) * 24, hipMemcpyDeviceToHost, c->stream));
  PQ_HIP(hipStreamSynchronize(c->stream));
  lap(5);
  if (trace_t)
    fprintf(stderr, "extract_skeleton host clock: setup %.3f build %.3f solve %.3f means %.3f step %.3f tail %.3f s (flags %.3f kstep %.3f copy %.3f)\n",
            t_acc[0], t_acc[1], t_acc[2], t_acc[3], t_acc[4], t_acc[5], t_acc[6], t_acc[7], t_acc[8]);
  return 0;
}

}  // extern "C"
