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
