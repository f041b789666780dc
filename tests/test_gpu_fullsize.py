"""Parity at BASELINE.json's full single-GPU sizes.

Where the CPU oracle finishes in seconds (DBSCAN, kNN at 1 M points) the comparison is
direct; for the 10 M x 500 k ray sweep it goes through size-independent properties:
the culled and the brute-force GPU paths must agree bit for bit on the whole batch, and
a sample of the rays is checked against the brute-force oracle."""
import numpy as np
import pytest

import oracle
from pyqsm_amd import hip, synth

pytestmark = pytest.mark.gpu


def test_dbscan_1m_points_vs_oracle(gpu):
    P = synth.forest(1_000_000)
    lab, core = hip.dbscan(P, 0.1, 10, device=gpu)
    lab0, core0 = oracle.dbscan(P, 0.1, 10)
    assert np.array_equal(core, core0) and np.array_equal(lab, lab0)
    assert lab.max() + 1 == 20                              # one cluster per tree
    # relabelling invariance: a permuted cloud yields the same partition
    perm = np.random.default_rng(0).permutation(len(P))
    lab_p, core_p = hip.dbscan(P[perm], 0.1, 10, device=gpu)
    assert np.array_equal(core_p, core[perm])
    same_noise = (lab_p == -1) == (lab[perm] == -1)
    assert same_noise.all()
    pairs = np.unique(np.stack([lab[perm][lab_p >= 0], lab_p[lab_p >= 0]], 1), axis=0)
    assert len(pairs) == 20                                 # a bijection between label sets


def test_knn_1m_points_vs_oracle(gpu):
    P = synth.forest(1_000_000)
    idx, d2 = hip.knn(P, 20, True, device=gpu)
    idx0, d20 = oracle.knn(P, 20, True)
    assert np.array_equal(d2, d20) and np.array_equal(idx, idx0)
    assert np.all(np.diff(d2, axis=1) >= 0)                 # sorted rows


def test_ray_sweep_10m_rays_500k_triangles(gpu, monkeypatch):
    verts, tris = synth.canopy_mesh(500_000)
    rays = synth.sun_rays(verts, 10_000_000)
    monkeypatch.setenv("PYQSM_RAY_CULL", "1")
    t_c, p_c, uv_c = hip.cast_rays(verts, tris, rays, device=gpu)
    monkeypatch.setenv("PYQSM_RAY_CULL", "0")
    t_b, p_b, uv_b = hip.cast_rays(verts, tris, rays, device=gpu)
    assert np.array_equal(t_c, t_b) and np.array_equal(p_c, p_b) and np.array_equal(uv_c, uv_b)
    sample = np.random.default_rng(1).choice(len(rays), 3000, replace=False)
    t0, p0, uv0 = oracle.cast_rays(verts, tris, rays[sample])
    assert np.array_equal(t_b[sample], t0) and np.array_equal(p_b[sample], p0)
    assert np.array_equal(uv_b[sample], uv0)
    hit = np.isfinite(t_b)
    assert 0.5 < hit.mean() < 0.75 and np.all(p_b[~hit] == 0xFFFFFFFF)
    # every reported hit point lies on its triangle's plane (t, u, v are consistent)
    h = np.flatnonzero(hit)[::5000]
    tri = verts[tris[p_b[h]]].astype(np.float64)
    on_ray = rays[h, :3].astype(np.float64) + rays[h, 3:].astype(np.float64) * t_b[h, None]
    bary = ((1 - uv_b[h, 0] - uv_b[h, 1])[:, None] * tri[:, 0] + uv_b[h, 0][:, None] * tri[:, 1]
            + uv_b[h, 1][:, None] * tri[:, 2])
    assert np.abs(on_ray - bary).max() < 1e-4
