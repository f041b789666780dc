"""ctypes binding of libpyqsm_hip.so (the C-ABI declared in include/pyqsm_hip.h).

There is no CPU fallback: if the shared library is missing or a call fails, the
caller gets an exception, never a silently different code path.
"""
from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libpyqsm_hip.so")

i64, i32, u32, dbl = ctypes.c_int64, ctypes.c_int32, ctypes.c_uint32, ctypes.c_double
vp = ctypes.c_void_p

# name -> (restype, argtypes); every symbol include/pyqsm_hip.h declares
SIGNATURES = {
    "pyqsm_device_count": (ctypes.c_int, []),
    "pyqsm_init": (ctypes.c_int, [ctypes.c_int]),
    "pyqsm_shutdown": (ctypes.c_int, []),
    "pyqsm_last_error": (ctypes.c_char_p, []),
    "pyqsm_version": (ctypes.c_char_p, []),
    "pyqsm_sync": (ctypes.c_int, [ctypes.c_int]),
    "pyqsm_stream": (vp, [ctypes.c_int]),
    "pyqsm_dev_malloc": (ctypes.c_int, [ctypes.c_int, ctypes.c_size_t, ctypes.POINTER(vp)]),
    "pyqsm_dev_free": (ctypes.c_int, [ctypes.c_int, vp]),
    "pyqsm_h2d": (ctypes.c_int, [ctypes.c_int, vp, vp, ctypes.c_size_t]),
    "pyqsm_d2h": (ctypes.c_int, [ctypes.c_int, vp, vp, ctypes.c_size_t]),
    "pyqsm_free": (None, [vp]),
    "pyqsm_host_alloc": (vp, [ctypes.c_size_t]),
    "pyqsm_prof_enable": (ctypes.c_int, [ctypes.c_int, ctypes.c_int]),
    "pyqsm_prof_reset": (ctypes.c_int, [ctypes.c_int]),
    "pyqsm_prof_get": (ctypes.c_int, [ctypes.c_int, ctypes.c_char_p, ctypes.POINTER(dbl),
                                      ctypes.POINTER(i64)]),
    "pyqsm_cast_rays": (ctypes.c_int, [vp, i64, vp, i64, vp, i64, vp, vp, vp, i32]),
    "pyqsm_expand_tris_dev": (ctypes.c_int, [vp, i64, vp, i64, vp, i32]),
    "pyqsm_cast_rays_dev": (ctypes.c_int, [vp, i64, vp, i64, vp, vp, vp, i32]),
    "pyqsm_list_intersections": (ctypes.c_int, [vp, i64, vp, i64, vp, i64, vp, vp, vp, vp, vp,
                                                i64, ctypes.POINTER(i64), i32]),
    "pyqsm_point_mesh_distance": (ctypes.c_int, [vp, i64, vp, i64, vp, i64, vp, vp, i32]),
    "pyqsm_cast_rays_multi": (ctypes.c_int, [vp, i64, vp, i64, vp, i64, vp, vp, vp, i32]),
    "pyqsm_shard_bounds": (ctypes.c_int, [i64, i32, i32, ctypes.POINTER(i64), ctypes.POINTER(i64)]),
    "pyqsm_comm_unique_id": (ctypes.c_int, [vp]),
    "pyqsm_comm_init_rank": (ctypes.c_int, [vp, i32, i32, i32]),
    "pyqsm_comm_finalize": (ctypes.c_int, []),
    "pyqsm_comm_info": (ctypes.c_int, [ctypes.POINTER(i32), ctypes.POINTER(i32),
                                       ctypes.POINTER(i32)]),
    "pyqsm_comm_broadcast_dev": (ctypes.c_int, [vp, i64, i32]),
    "pyqsm_comm_all_gather_dev": (ctypes.c_int, [vp, vp, i64]),
    "pyqsm_comm_all_reduce_max": (ctypes.c_int, [ctypes.POINTER(dbl)]),
    "pyqsm_dbscan": (ctypes.c_int, [vp, i64, dbl, i32, vp, vp, i32]),
    "pyqsm_dbscan_dev": (ctypes.c_int, [vp, i64, dbl, i32, vp, vp, ctypes.POINTER(i64), i32]),
    "pyqsm_dbscan_ex": (ctypes.c_int, [vp, i64, dbl, i32, i32, vp, vp, i32]),
    "pyqsm_dbscan_dev_ex": (ctypes.c_int, [vp, i64, dbl, i32, i32, vp, vp, ctypes.POINTER(i64), i32]),
    "pyqsm_knn": (ctypes.c_int, [vp, i64, i32, i32, vp, vp, i32]),
    "pyqsm_knn_dev": (ctypes.c_int, [vp, i64, i32, i32, vp, vp, i32]),
    "pyqsm_ransac": (ctypes.c_int, [vp, i64, vp, i64, i32, dbl, vp, vp, ctypes.POINTER(dbl), vp,
                                    ctypes.POINTER(i64), ctypes.POINTER(i64), i32]),
    "pyqsm_ransac_models": (ctypes.c_int, [vp, i64, vp, i64, vp, i32]),
    "pyqsm_ransac_count": (ctypes.c_int, [vp, i64, vp, i64, i32, dbl, vp, i32]),
    "pyqsm_ransac_batch": (ctypes.c_int, [vp, i64, vp, i64, vp, i64, i32, dbl, vp, vp, vp, vp, vp, vp, i32]),
    "pyqsm_lbc_solve": (ctypes.c_int, [vp, vp, vp, i64, vp, vp, vp, dbl, i32, vp,
                                       ctypes.POINTER(i32), vp, i32]),
    "pyqsm_spmv3": (ctypes.c_int, [vp, vp, vp, i64, vp, vp, i32]),
    "pyqsm_clamp": (ctypes.c_int, [vp, i64, vp, vp, i32]),
    "pyqsm_ball_query": (ctypes.c_int, [vp, i64, vp, dbl, vp, ctypes.POINTER(i64), i32]),
    "pyqsm_radius_mark": (ctypes.c_int, [vp, i64, vp, i64, dbl, i32, vp, vp, i32]),
    "pyqsm_radius_label": (ctypes.c_int, [vp, i64, vp, i64, vp, dbl, i32, vp, vp, i32]),
    "pyqsm_radius_knn": (ctypes.c_int, [vp, i64, vp, i64, dbl, i32, vp, vp, i32]),
    "pyqsm_fps": (ctypes.c_int, [vp, i64, i64, i64, vp, i32]),
    "pyqsm_pc_laplacian": (ctypes.c_int, [vp, i64, i32, dbl, ctypes.POINTER(i64),
                                          ctypes.POINTER(vp), ctypes.POINTER(vp),
                                          ctypes.POINTER(vp), vp, i32]),
    "pyqsm_extract_skeleton": (ctypes.c_int, [vp, i64, vp, i64, i32, dbl, i32, dbl, dbl, dbl, dbl, dbl, vp, vp,
                                              dbl, i32, vp, vp, vp, vp, vp, vp, vp, ctypes.POINTER(i32), i32]),
    "pyqsm_pc_laplacian_seg": (ctypes.c_int, [vp, i64, vp, i64, i32, dbl, ctypes.POINTER(i64),
                                              ctypes.POINTER(vp), ctypes.POINTER(vp),
                                              ctypes.POINTER(vp), vp, i32]),
    "pyqsm_mean_f64": (ctypes.c_int, [vp, i64, ctypes.POINTER(dbl)]),
    "pyqsm_extreme_points": (ctypes.c_int, [vp, i64, vp, i32, vp, i32]),
    "pyqsm_outside_halfspaces": (ctypes.c_int, [vp, i64, vp, i32, dbl, vp, ctypes.POINTER(i64), i32]),
}

_lib = None


class PyQSMHipError(RuntimeError):
    """A libpyqsm_hip call returned a negative status."""

    def __init__(self, code: int, message: str):
        super().__init__(f"libpyqsm_hip error {code}: {message}")
        self.code = code


def load() -> ctypes.CDLL:
    """Load the library and bind every declared symbol. Raises if the shared
    object has not been built (run ``python -c 'import __graft_entry__ as g; g.build()'``
    or ``make -C pyqsm_amd/csrc``)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: the HIP library has not been built and there is no CPU "
            "fallback. Build it with `make -C pyqsm_amd/csrc`.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(code: int) -> None:
    if code != 0:
        msg = load().pyqsm_last_error()
        raise PyQSMHipError(code, msg.decode("utf-8", "replace") if msg else "")


def device_count() -> int:
    return int(load().pyqsm_device_count())


def logical_device_count() -> int:
    """What a multi-GPU driver may address: the visible GPUs, or — with PYQSM_MULTI_FAKE_RANKS=N
    (csrc/multi.hip: N logical ranks on one device, how one-GPU boxes run the N > 1 code) — N."""
    fake = os.environ.get("PYQSM_MULTI_FAKE_RANKS", "")
    return int(fake) if fake.isdigit() and int(fake) > 0 else device_count()


def physical_device(logical: int) -> int:
    """The real device behind a logical device index (identity unless PYQSM_MULTI_FAKE_RANKS is set)."""
    fake = os.environ.get("PYQSM_MULTI_FAKE_RANKS", "")
    if fake.isdigit() and int(fake) > 0:
        return int(logical) % max(1, device_count())
    return int(logical)


def require_gpu(device: int = 0) -> None:
    """Raise unless `device` is a usable GPU."""
    check(load().pyqsm_init(int(device)))
