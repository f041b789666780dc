"""Scratch timing of the kernels with HBM-resident inputs (development aid)."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
from pyqsm_amd import hip, synth, _lib
_lib.require_gpu(0)
what = sys.argv[1] if len(sys.argv) > 1 else "all"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
hip.prof_enable(True)
P = synth.forest(n)
d_xyz = hip.DeviceBuffer.from_array(P)
if what in ("all", "dbscan"):
    d_lab = hip.DeviceBuffer(n * 8); d_core = hip.DeviceBuffer(n)
    for it in range(3):
        hip.prof_reset()
        t = time.time(); nc = hip.dbscan_dev(d_xyz.ptr, n, 0.1, 10, d_lab.ptr, d_core.ptr, want_count=True); hip.sync(); dt = time.time() - t
        print(f"dbscan n={n} clusters={nc} wall {dt*1e3:.2f} ms -> {n/dt/1e6:.1f} Mpts/s", {k: round(hip.prof_get(k)[0], 3) for k in ("dbscan_bin", "dbscan_core", "dbscan_union", "dbscan_label", "dbscan_total")}, flush=True)
if what in ("all", "knn"):
    k = 20
    d_idx = hip.DeviceBuffer(n * k * 4); d_d2 = hip.DeviceBuffer(n * k * 8)
    for it in range(3):
        hip.prof_reset()
        t = time.time(); hip.knn_dev(d_xyz.ptr, n, k, True, d_idx.ptr, d_d2.ptr); hip.sync(); dt = time.time() - t
        print(f"knn n={n} k={k} wall {dt*1e3:.2f} ms -> {n/dt/1e6:.1f} Mpts/s", {kk: (round(hip.prof_get(kk)[0], 3), hip.prof_get(kk)[1]) for kk in ("knn_bin", "knn_search", "knn_retry", "knn_total")}, flush=True)
if what in ("all", "lap"):
    m = min(n, 200_000)
    for it in range(2):
        hip.prof_reset()
        t = time.time(); (ip, ix, dv), mass = hip.pc_laplacian(P[:m], 20, 1e-6); dt = time.time() - t
        print(f"laplacian n={m} wall {dt*1e3:.1f} ms nnz/row {len(ix)/m:.2f}", {kk: round(hip.prof_get(kk)[0], 3) for kk in ("lap_knn", "lap_fans", "lap_assemble")}, flush=True)
if what in ("all", "rays"):
    T = 500_000; R = int(sys.argv[3]) if len(sys.argv) > 3 else 2_000_000
    verts, tris = synth.canopy_mesh(T)
    rays = synth.sun_rays(verts, R)
    mesh = hip.DeviceMesh(verts, tris)
    d_rays = hip.DeviceBuffer.from_array(rays); d_t = hip.DeviceBuffer(R * 4); d_p = hip.DeviceBuffer(R * 4)
    for it in range(2):
        hip.prof_reset()
        t = time.time(); hip.cast_rays_dev(mesh, d_rays.ptr, R, d_t.ptr, d_p.ptr); hip.sync(); dt = time.time() - t
        th = d_t.download((R,), np.float32)
        print(f"rays R={R} T={T} wall {dt*1e3:.1f} ms -> {R*T/dt/1e6:.3e} Mray-tri/s hits={np.isfinite(th).mean():.3f}", hip.prof_get("cast_rays"), hip.prof_get("cast_rays_culled"), flush=True)
