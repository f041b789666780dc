"""dev aid: time cast_rays on the reference's pinhole camera (1280 x 950) over the canopy mesh"""
import os, sys, time, numpy as np
sys.path.insert(0, '.')
from pyqsm_amd import hip, synth, _lib
from pyqsm_amd.viz.ray_casting import create_rays_pinhole
_lib.require_gpu(0)
verts, tris = synth.canopy_mesh(500_000)
c = 0.5 * (verts.min(0) + verts.max(0))
rays = create_rays_pinhole(90.0, c, c + [0, 0, 10], (0, 1, -1), 1280, 950).reshape(-1, 6)
mesh = hip.DeviceMesh(verts, tris, 0)
d_rays = hip.DeviceBuffer.from_array(rays, 0)
d_t = hip.DeviceBuffer(len(rays) * 4, 0); d_p = hip.DeviceBuffer(len(rays) * 4, 0)
out = {}
for flag in ("1", "0"):
    os.environ["PYQSM_RAY_CULL"] = flag
    hip.cast_rays_dev(mesh, d_rays.ptr, len(rays), d_t.ptr, d_p.ptr); hip.sync(0)
    t0 = time.perf_counter()
    for _ in range(3): hip.cast_rays_dev(mesh, d_rays.ptr, len(rays), d_t.ptr, d_p.ptr)
    hip.sync(0)
    dt = (time.perf_counter() - t0) / 3
    out[flag] = (d_t.download((len(rays),), np.float32), d_p.download((len(rays),), np.uint32))
    print('cull', flag, '%.2f ms' % (dt * 1e3), 'hit fraction %.3f' % np.isfinite(out[flag][0]).mean())
print('identical', np.array_equal(out["1"][0], out["0"][0]) and np.array_equal(out["1"][1], out["0"][1]))
