"""dev aid: interception_layers on config 5's light stage (10 M sun rays x 500 k triangles)"""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from pyqsm_amd import synth, _lib
from pyqsm_amd.viz.ray_casting import interception_layers
_lib.require_gpu(0)
T, R = int(sys.argv[1]) if len(sys.argv) > 1 else 500_000, int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
verts, tris = synth.canopy_mesh(T)
rays = synth.sun_rays(verts, R, elevation_deg=60.0, azimuth_deg=45.0)
interception_layers((verts, tris), rays[:1000])
t = time.perf_counter(); areas, layer = interception_layers((verts, tris), rays); dt = time.perf_counter() - t
print('%d rounds in %.0f ms (%.1f ms per round); layer sizes %s; never hit %d' % (len(areas), dt * 1e3, dt * 1e3 / max(1, len(areas)), np.bincount(layer[layer >= 0])[:8].tolist(), int((layer < 0).sum())))
