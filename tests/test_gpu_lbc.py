"""HIP contraction solve vs SciPy spsolve fixtures (skeletonize.py:148-180);
tolerance 1e-5 relative on vertex positions (BASELINE.json north_star)."""
import glob
import os

import numpy as np
import pytest
import scipy.sparse as sp
from scipy.sparse import csr_matrix

import oracle
from pyqsm_amd import hip

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
RTOL = 1e-5


def _load(path):
    g = np.load(path)
    n = len(g["points"])
    L = csr_matrix((g["data"], g["indices"], g["indptr"]), shape=(n, n))
    return g, L


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "lbc_*.npz"))))
def test_solve_matches_spsolve_fixture(gpu, path):
    g, L = _load(path)
    x, iters, resid, ok = hip.lbc_solve(L, g["wl"], g["wh"], g["points"], rtol=1e-10,
                                        max_it=200000, device=gpu)
    assert ok, (iters, resid)
    sol = g["solution"]
    scale = np.abs(sol).max()
    assert np.abs(x - sol).max() <= RTOL * scale
    # and the oracle restatement run live agrees with its own fixture
    live = oracle.least_squares_sparse(g["points"], L, g["wl"], g["wh"])
    assert np.abs(live - sol).max() <= 1e-9 * scale


def test_spmv3_matches_scipy(gpu):
    g, L = _load(sorted(glob.glob(os.path.join(GOLD, "lbc_*.npz")))[0])
    x = np.random.default_rng(0).normal(size=g["points"].shape)
    y = hip.spmv3(L, x, device=gpu)
    ref = L @ x
    assert np.abs(y - ref).max() <= 1e-12 * np.abs(ref).max()


def test_clamp_matches_reference_loop(gpu):
    rng = np.random.default_rng(1)
    pts = rng.normal(0, 2, (1000, 3))
    lo, hi = np.array([-1.0, -0.5, 0.0]), np.array([1.0, 0.5, 3.0])
    want = pts.copy()
    for p in want:                                     # skeletonize.py:291-296
        for i in range(3):
            if p[i] < lo[i]:
                p[i] = lo[i]
            if p[i] > hi[i]:
                p[i] = hi[i]
    got = hip.clamp(pts.copy(), lo, hi, device=gpu)
    assert np.array_equal(got, want)


def test_max_it_reports_no_convergence(gpu):
    g, L = _load(sorted(glob.glob(os.path.join(GOLD, "lbc_*.npz")))[1])
    x, iters, resid, ok = hip.lbc_solve(L, g["wl"], g["wh"], g["points"], rtol=1e-14, max_it=25,
                                        device=gpu)
    assert not ok and 25 <= iters <= 30 and np.all(np.isfinite(x))


def _small_system(n=1500, seed=3):
    from pyqsm_amd import synth
    P = synth.forest(n, seed=seed)
    L, M = oracle.point_cloud_laplacian(P, 12, 1e-6)
    return P, L.tocsr(), np.asarray(M)


@pytest.mark.parametrize("path_env", [{}, {"PYQSM_AMG": "0"}, {"PYQSM_LBC_SORT": "0"}, {"PYQSM_NO_GRAPH": "1"}])
def test_solver_variants_agree_with_spsolve(gpu, monkeypatch, path_env):
    """Multigrid (default), Jacobi-PCG B-solves, unsorted unknowns, plain launches: every variant
    must reproduce the SciPy direct solve of the same system to the parity bound."""
    for k, v in path_env.items():
        monkeypatch.setenv(k, v)
    P, L, M = _small_system(6000)
    wl = np.full(len(P), 3e3 * np.sqrt(M.mean()))
    rng = np.random.default_rng(0)
    wh = 3.0 * np.sqrt(M.mean() / M) ** rng.uniform(0.5, 1.5, len(P))     # rough positional weights
    x, iters, resid, ok = hip.lbc_solve(L, wl, wh, P, rtol=1e-9, max_it=500000, device=gpu)
    ref = oracle.least_squares_sparse(P, L, wl, wh)
    assert ok, (iters, resid)
    assert np.abs(x - ref).max() <= RTOL * np.abs(ref).max()


def test_non_uniform_laplacian_weights_take_the_general_path(gpu):
    """extract_skeleton never produces per-point wl, least_squares_sparse accepts them
    (skeletonize.py:160-164): Jacobi-PCG on A itself, residual-based stop."""
    P, L, M = _small_system(1200)
    rng = np.random.default_rng(1)
    wl = rng.uniform(0.5, 2.0, len(P))
    wh = rng.uniform(1.0, 3.0, len(P))
    x, iters, resid, ok = hip.lbc_solve(L, wl, wh, P, rtol=1e-11, max_it=400000, device=gpu)
    ref = oracle.least_squares_sparse(P, L, wl, wh)
    assert np.abs(x - ref).max() <= RTOL * np.abs(ref).max()


@pytest.mark.parametrize("n_each", [1500, 6000])
def test_block_diagonal_system_with_a_weight_per_block(gpu, n_each):
    """Several clouds stacked into one block-diagonal system, each with its own contraction
    weight: wl is constant along every edge of L, which is all the fast path (B^2-preconditioned
    CG, multigrid B-solves; sorted unknowns from 4096 rows on) needs. Same answer as the direct
    solve of the stacked system and as solving the blocks one by one."""
    from scipy.sparse import block_diag
    blocks = [_small_system(n_each, seed=s) for s in (3, 4, 5)]
    P = np.concatenate([b[0] + [40.0 * k, 0, 0] for k, b in enumerate(blocks)])
    L = block_diag([b[1] for b in blocks], format="csr")
    M = np.concatenate([b[2] for b in blocks])
    wl = np.concatenate([np.full(len(b[0]), f * 1e3 * np.sqrt(b[2].mean())) for f, b in zip((3.0, 9.0, 27.0), blocks)])
    wh = 3.0 * np.sqrt(M.mean() / M)
    x, iters, resid, ok = hip.lbc_solve(L, wl, wh, P, rtol=1e-9, max_it=500000, device=gpu)
    assert ok, (iters, resid)
    ref = oracle.least_squares_sparse(P, L, wl, wh)
    assert np.abs(x - ref).max() <= RTOL * np.abs(ref).max()
    assert iters < 3000                       # (the plain Jacobi-PCG path needs tens of thousands)
    lo = 0
    for b in blocks:                          # the blocks one by one
        hi = lo + len(b[0])
        xb, _, _, okb = hip.lbc_solve(b[1], wl[lo:hi], wh[lo:hi], P[lo:hi], rtol=1e-9, max_it=500000, device=gpu)
        assert okb and np.abs(xb - x[lo:hi]).max() <= RTOL * np.abs(ref).max()
        lo = hi


def test_riccati_weighted_preconditioner_same_solution_fewer_iterations(gpu, monkeypatch):
    """The preconditioner's Riccati weights (lbc.hip: k_riccati_f) change how
    fast the outer iteration converges, not what it converges to: on the system of the third
    contraction of a loop — the first with a rough W_H — the solve with PYQSM_RICCATI=1 agrees
    with the plain one to the solver's tolerance, both end with ok = True, and it takes fewer
    iterations; with a uniform W_H the Riccati equation is solved by W_H itself and the two modes
    return the same bits."""
    from pyqsm_amd import synth
    from pyqsm_amd.geometry import skeletonize as sk
    P = synth.forest(30_000, seed=4)
    systems = []
    inner = sk.least_squares_sparse

    def capture(pts, L, laplacian_weighting, positional_weighting, **kw):
        systems.append((L.copy(), laplacian_weighting.copy(), positional_weighting.copy(), pts.copy()))
        return inner(pts=pts, L=L, laplacian_weighting=laplacian_weighting,
                     positional_weighting=positional_weighting, **kw)

    monkeypatch.setattr(sk, "least_squares_sparse", capture)
    sk.extract_skeleton(P, max_iter=3, termination_ratio=0.0, contraction_factor=3, attraction_factor=3)
    monkeypatch.undo()
    assert len(systems) == 3
    out = {}
    for step in (0, 2):
        L, wl, wh, pts = systems[step]
        for mode in ("0", "1"):
            monkeypatch.setenv("PYQSM_RICCATI", mode)
            out[step, mode] = hip.lbc_solve(L, wl, wh, pts, rtol=1e-8, device=gpu)
        monkeypatch.delenv("PYQSM_RICCATI")
    x0, it0, _, ok0 = out[0, "0"]
    x1, it1, _, ok1 = out[0, "1"]
    assert ok0 and ok1 and it0 == it1 and np.array_equal(x0, x1)          # uniform W_H: F = 0
    x0, it0, _, ok0 = out[2, "0"]
    x1, it1, _, ok1 = out[2, "1"]
    wh = systems[2][2]
    assert wh.max() / wh.min() > 100.0                                    # rough W_H
    assert ok0 and ok1
    scale = np.abs(x0).max()
    print(f"rough W_H: plain {it0} iterations, Riccati {it1}; difference {np.abs(x0 - x1).max() / scale:.1e}")
    assert np.abs(x0 - x1).max() <= 1e-6 * scale
    assert it1 < it0


def test_dense_coarsest_level_made_on_the_device(gpu, monkeypatch, capfd):
    """The multigrid stops coarsening at <= 1024 rows and solves that level with a dense inverse made on
    the device (amg.hip: k_gj_pivot / panels / update, k_dense_mv); PYQSM_AMG_DENSE_MAX=96 is the
    earlier hierarchy (down to <= 96 rows, inverse by host Cholesky). Same system, both ways: the
    hierarchies differ as stated, both solves converge, the solutions agree to the solver's tolerance
    and the iteration counts stay close (the coarse solve is exact either way)."""
    from pyqsm_amd import synth
    from pyqsm_amd.geometry import skeletonize as sk
    P = synth.forest(60_000, seed=6)
    L, M = sk.point_cloud_laplacian(P, mollify_factor=1e-6, n_neighbors=20, device=gpu)
    wl = np.full(len(P), 3 * 1e3 * np.sqrt(np.mean(M.diagonal())))
    wh = np.full(len(P), 3.0)
    monkeypatch.setenv("PYQSM_LBC_TRACE", "1")
    out, last = {}, {}
    for dmax in ("96", "1024"):
        monkeypatch.setenv("PYQSM_AMG_DENSE_MAX", dmax)
        capfd.readouterr()
        out[dmax] = hip.lbc_solve(L, wl, wh, P, rtol=1e-8, device=gpu)
        levels = [ln for ln in capfd.readouterr().err.splitlines() if ln.startswith("multigrid levels:")]
        last[dmax] = int(levels[0].split()[-1])
    assert last["96"] <= 96 < last["1024"] <= 1024
    (x0, it0, _, ok0), (x1, it1, _, ok1) = out["96"], out["1024"]
    assert ok0 and ok1
    scale = np.abs(x0).max()
    print(f"coarsest {last['96']} rows: {it0} iterations; coarsest {last['1024']} rows: {it1}; "
          f"difference {np.abs(x0 - x1).max() / scale:.1e}")
    assert np.abs(x0 - x1).max() <= 1e-6 * scale
    assert abs(it1 - it0) <= 0.15 * it0


def test_diverging_fp32_inner_solve_falls_back_to_the_fp64_operator(gpu, monkeypatch, capfd):
    """On a collapsed cloud with a very large W_L the fp32 multigrid-CG solves of the
    preconditioner B^-1 B^-1 diverge (c L_ii ~ 1e9 next to W_H = 0.1 is beyond what fp32 rows
    resolve; DESIGN.md section 6, "fp32 breakdown"). lbc.hip's ladder notices within 16
    iterations and repeats the application with the fp64 operator, so the solve still ends with
    ok = True and agrees with the solve that never used fp32 (PYQSM_LBC_F32=0). The true residual
    of A = W_L L^T L W_L + W_H^2 (pyQSM/geometry/skeletonize.py:134-137) cannot be the check here:
    with entries of 1e24 its evaluation in fp64 carries more rounding than |b|."""
    from pyqsm_amd import synth
    from pyqsm_amd.geometry import skeletonize as sk
    P = synth.forest(30_000, seed=4)
    systems = []
    inner = sk.least_squares_sparse

    def capture(pts, L, laplacian_weighting, positional_weighting, **kw):
        systems.append((L.copy(), laplacian_weighting.copy(), positional_weighting.copy(), pts.copy()))
        return inner(pts=pts, L=L, laplacian_weighting=laplacian_weighting,
                     positional_weighting=positional_weighting, **kw)

    monkeypatch.setattr(sk, "least_squares_sparse", capture)
    sk.extract_skeleton(P, max_iter=16, termination_ratio=0.0, contraction_factor=7, attraction_factor=3)
    monkeypatch.undo()
    L, wl, wh, pts = systems[-1]
    wl = wl * 1000.0
    monkeypatch.setenv("PYQSM_LBC_TRACE", "1")
    capfd.readouterr()
    x, iters, resid, ok = hip.lbc_solve(L, wl, wh, pts, rtol=1e-8, max_it=20_000, device=gpu)
    err = capfd.readouterr().err
    assert "fp64 operator from here" in err, "the fp32 inner solve was expected to break down on this system"
    monkeypatch.delenv("PYQSM_LBC_TRACE")
    monkeypatch.setenv("PYQSM_LBC_F32", "0")
    x64, iters64, _, ok64 = hip.lbc_solve(L, wl, wh, pts, rtol=1e-8, max_it=20_000, device=gpu)
    assert ok and ok64 and np.isfinite(x).all()
    diff = np.abs(x - x64).max() / np.abs(x64).max()
    print(f"ladder {iters} iterations, fp64 only {iters64}; difference {diff:.1e}")
    assert diff <= 1e-6
    assert iters <= iters64 + 200           # the failed fp32 attempt costs tens of iterations, not thousands
