"""BASELINE.json configs[4] at its stated sizes, stage by stage against independent checks
(pyQSM/qsm_generation.py:182-316 is the caller this pipeline stands for):

* the 5 M-point scan's DBSCAN against the sequential C oracle (bit-exact labels and core set);
* one 50 k-point tree through the whole contraction loop with the invariants of
  tests/test_gpu_config3.py (host-side residuals, symmetric Laplacians, no ENOCONV, bounds);
* RANSAC circles on that tree's 0.5 m stem slices (H = 1000, seed 2) against the NumPy
  restatement of pyransac3d (inlier sets bit-exact);
* 5 sun angles of the canopy light simulation at full mesh size, a sample of each against the
  brute-force oracle.
The N-GPU form of the pipeline (clusters dealt round-robin, rays sharded through RCCL) is
examples/config5_pipeline.py --gpus N; here the one-GPU forms of the same calls run."""
import numpy as np
import pytest

import oracle
from pyqsm_amd import hip, synth
from pyqsm_amd.math_utils import fit
from pyqsm_amd.viz import ray_casting as rc
from tests.test_gpu_config3 import _assert_invariants, _run_config3

pytestmark = pytest.mark.gpu


def test_dbscan_5m_points_vs_oracle(gpu):
    P = synth.forest(5_000_000)
    lab, core = hip.dbscan(P, 0.1, 10, device=gpu)
    lab0, core0 = oracle.dbscan(P, 0.1, 10)
    assert np.array_equal(core, core0) and np.array_equal(lab, lab0)
    assert lab.max() + 1 == 100                                   # one cluster per tree
    # the wrapper's post-processing on the same labels (fit.py:224-250): core samples only
    uniq, idxs, noise = fit.cluster_DBSCAN(np.arange(len(P)), P, 0.1, 10, device=gpu)
    assert len(idxs) == 100 and len(noise) == int((lab == -1).sum())
    assert sum(len(i) for i in idxs) == int(core.sum())


def test_one_tree_contracts_with_invariants(gpu, monkeypatch):
    out = _run_config3(50_000, 20, 3, monkeypatch)
    _assert_invariants(*out, iters=20)


def test_ransac_on_stem_slices(gpu):
    P = synth.forest(50_000)
    fits = 0
    for z0 in np.arange(0.5, 5.5, 0.5):
        sl = P[(P[:, 2] >= z0) & (P[:, 2] < z0 + 0.5)]
        sl = sl[np.hypot(sl[:, 0] - np.median(sl[:, 0]), sl[:, 1] - np.median(sl[:, 1])) < 0.6]
        if len(sl) < 50:
            continue
        samples = fit.draw_samples(len(sl), 1000, seed=2)
        mesh, _, inl, r, axis = fit.fit_shape_RANSAC(pts=sl.copy(), shape="circle", threshold=0.04,
                                                     max_radius=0.3 * 1.75, samples=samples, device=gpu)
        flat = sl.copy()
        flat[:, 2] = 0
        c0, a0, r0, inl0, _ = oracle.ransac_fit(flat, samples, "circle", 0.04)
        assert mesh is not None and np.array_equal(inl, inl0) and abs(r - r0) < 1e-9
        assert abs(r - 0.30) < 0.02                               # the trunk's radius
        fits += 1
    assert fits >= 8


def test_light_simulation_five_sun_angles(gpu):
    verts, tris = synth.canopy_mesh(500_000)
    rng = np.random.default_rng(5)
    for az in (45.0, 90.0, 135.0, 180.0, 225.0):
        rays = synth.sun_rays(verts, 2_000_000, elevation_deg=60.0, azimuth_deg=az)
        ans = rc.cast_rays((verts, tris), rays=rays, device=gpu)
        sample = rng.choice(len(rays), 800, replace=False)
        t0, p0, _ = oracle.cast_rays(verts, tris, rays[sample])
        assert np.array_equal(ans["t_hit"][sample], t0) and np.array_equal(ans["primitive_ids"][sample], p0)
        assert 0.4 < ans["hit"].mean() < 0.9


def test_whole_pipeline_at_stated_sizes(gpu):
    """configs[4] as ONE run at its stated sizes on one GPU (the driver of examples/config5_pipeline.py,
    which is what bench.py's `config5` section times): 5 M points -> DBSCAN -> all 100 trees through
    extract_skeleton_batch (20 contractions) -> every 0.5 m stem slice of every tree through
    fit_shape_RANSAC_batch (H = 1000) -> 5 sun angles x 10 M rays. Stands for the per-cluster calls of
    pyQSM/qsm_generation.py:182-316. Checks: cluster lists against the sequential C oracle; every
    solve of every tree ok; three sampled trees against the per-tree loop; sampled slices' inlier sets
    against the NumPy restatement of pyransac3d; 2 000 sampled rays per sun angle against the
    brute-force oracle at the full 10 M."""
    from examples import config5_pipeline as c5
    from pyqsm_amd.geometry import skeletonize as sk
    out, kept = c5.run(scale=1.0, skeleton_iters=20, max_trees=100, engine="native", keep=True)
    print({k: v for k, v in out.items() if k.endswith("_s") or k in ("clusters", "ransac_slices", "ransac_fits")})
    P = kept["pts"]
    assert out["points"] == 5_000_000 and out["rays"] == 50_000_000 and out["tris"] == 500_000
    # --- stage 1: the wrapper's core-sample lists (fit.py:243-246) from the oracle's labels
    lab0, core0 = oracle.dbscan(P, 0.1, 10)
    assert out["clusters"] == 100 == lab0.max() + 1
    want = {int(np.flatnonzero((lab0 == k) & core0)[0]): np.flatnonzero((lab0 == k) & core0) for k in range(100)}
    for idx in kept["idxs"]:
        assert np.array_equal(np.sort(idx), want[int(np.min(idx))])
    assert np.array_equal(np.sort(kept["noise"]), np.flatnonzero((lab0 == -1) & ~core0))
    # --- stage 2: 100 trees, 20 contractions each, no solve without convergence
    sks = kept["skeletons"]
    assert len(sks) == 100
    for pc, total, steps in sks:
        assert len(steps) == 20 and len(pc.solve_log) == 20 and all(q["ok"] for q in pc.solve_log)
        assert np.isfinite(pc.points).all() and 0.1 < np.linalg.norm(total, axis=1).mean() < 0.3
    rng = np.random.default_rng(4)
    first = later = 0.0
    for j in rng.choice(100, 3, replace=False):
        tree = P[kept["idxs"][j]]
        pc1, total1, steps1 = sk.extract_skeleton(tree, max_iter=20, termination_ratio=0.0, device=gpu)
        scale = np.abs(tree).max()
        first = max(first, np.abs(steps1[0] - sks[j][2][0]).max() / scale)
        later = max(later, np.abs(pc1.points - sks[j][0].points).max() / scale)
    print(f"batch vs per-tree loop: first contraction {first:.1e}, after 20 contractions {later:.1e}")
    # the first contraction solves the same systems (1e-5 is north_star's tolerance; measured ~1e-8);
    # after twenty the two differ by the loop's response to the last bits of every solve — the batch
    # shares its CG scalars among the clouds of a group (tests/test_gpu_batch.py measures that response
    # on a one-ulp copy: up to ~1e-4 relative) — so the end state is held to 1e-3 of the scene size
    assert first <= 1e-5
    assert later <= 1e-3
    # --- stage 3: every slice was fitted; sampled slices against oracle.ransac_fit
    per_tree = kept["slices"]
    assert len(per_tree) == 100 and out["ransac_slices"] >= 900 and out["ransac_fits"] >= 0.95 * out["ransac_slices"]
    checked = 0
    for j in rng.choice(100, 6, replace=False):
        fits, slices, smp = per_tree[j]
        for q in rng.choice(len(slices), 2, replace=False):
            flat = slices[q].copy()
            flat[:, 2] = 0.0
            c0, a0, r0, inl0, _ = oracle.ransac_fit(flat, smp[q], "circle", 0.04)
            mesh, _, inl, r, axis = fits[q]
            assert mesh is not None and np.array_equal(inl, inl0) and abs(r - r0) < 1e-9
            checked += 1
    assert checked == 12 and abs(out["ransac_median_radius_m"] - 0.30) < 0.02
    # --- stage 4: 2 000 sampled rays per sun angle at the full 10 M, bit for bit
    verts, tris = kept["mesh"]
    for az, t_hit, prim in kept["hits"]:
        assert t_hit.shape == (10_000_000,)
        rays = synth.sun_rays(verts, kept["n_rays_per_angle"], elevation_deg=60.0, azimuth_deg=az)
        sample = rng.choice(len(rays), 2000, replace=False)
        t0, p0, _ = oracle.cast_rays(verts, tris, rays[sample])
        assert np.array_equal(t_hit[sample], t0) and np.array_equal(prim[sample], p0)
        assert 0.4 < np.isfinite(t_hit).mean() < 0.9
