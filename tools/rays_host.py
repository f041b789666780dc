"""dev aid: host-side profile of cast_rays on 10 M sun rays x 500 k triangles (config 5's light stage)"""
import cProfile, pstats, io, sys, time
sys.path.insert(0, '.')
import numpy as np
from pyqsm_amd import synth, _lib
from pyqsm_amd.viz.ray_casting import cast_rays
_lib.require_gpu(0)
verts, tris = synth.canopy_mesh(500_000)
t = time.perf_counter(); rays = synth.sun_rays(verts, 10_000_000, elevation_deg=60.0, azimuth_deg=45.0); print('sun_rays %.0f ms' % ((time.perf_counter() - t) * 1e3), rays.dtype, rays.shape)
cast_rays((verts, tris), rays=rays)
pr = cProfile.Profile(); pr.enable()
t = time.perf_counter(); ans = cast_rays((verts, tris), rays=rays); print('cast_rays %.0f ms' % ((time.perf_counter() - t) * 1e3))
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(18); print(s.getvalue()[:3000])
