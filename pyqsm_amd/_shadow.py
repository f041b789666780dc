"""Fall-through for the modules of this package that share a name with a pyQSM module.

pyQSM imports its own code flat (``from math_utils.fit import cluster_DBSCAN, fit_shape_RANSAC,
kmeans`` — pyQSM/qsm_generation.py:19-20; ``from utils.io import convert_las`` —
pyQSM/pipeline.py:8). With this directory AHEAD of pyQSM's on ``sys.path``:

* ``geometry/``, ``math_utils/``, ``viz/`` and ``utils/`` hold no ``__init__.py`` on either side,
  so they are namespace packages whose portions merge: ``math_utils.fit`` is this package's,
  ``utils.io`` / ``geometry.mesh_processing`` / ``viz.color`` are still pyQSM's;
* a module that exists on BOTH sides (``math_utils/fit.py``, ``geometry/point_cloud_processing.py``,
  ``qsm_generation.py`` …) resolves to this package's, which only restates the hot-path functions.
  Every other name of the shadowed module (``kmeans``, ``clean_cloud``, ``crop_by_percentile``,
  ``get_shape``, ``get_angles`` …) is looked up, on first use, in the NEXT same-named file on
  ``sys.path`` through the module-level ``__getattr__`` installed here (PEP 562; ``from m import
  name`` honours it). That file is executed once under a private module name; its own flat
  imports see the same merged packages, so the engines it reaches are the HIP ones as well.

Nothing is looked up unless a name is missing, so without pyQSM on the path this costs nothing.
"""
from __future__ import annotations

import importlib.util
import os
import sys
import threading

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
_lock = threading.RLock()
_loaded: dict = {}          # relative path -> module, or None when no other file exists


def _relative_file(module_name: str) -> str:
    parts = module_name.split(".")
    if parts[0] == "pyqsm_amd":
        parts = parts[1:]
    return os.path.join(*parts) + ".py"


def shadowed_module(module_name: str):
    """The next file on ``sys.path`` with this module's relative path (not this package's own),
    executed as ``_pyqsm_shadowed.<name>``; None when there is none."""
    rel = _relative_file(module_name)
    with _lock:
        if rel in _loaded:
            return _loaded[rel]
        _loaded[rel] = None                       # re-entrant lookups while the file executes see "none"
        for entry in list(sys.path):
            base = os.path.abspath(entry or os.getcwd())
            if base == _PKG_DIR:
                continue
            path = os.path.join(base, rel)
            if not os.path.isfile(path) or os.path.abspath(path).startswith(_PKG_DIR + os.sep):
                continue
            name = "_pyqsm_shadowed." + rel[:-3].replace(os.sep, ".")
            spec = importlib.util.spec_from_file_location(name, path)
            mod = importlib.util.module_from_spec(spec)
            sys.modules[name] = mod
            try:
                spec.loader.exec_module(mod)
            except BaseException:
                sys.modules.pop(name, None)
                _loaded.pop(rel, None)
                raise
            _loaded[rel] = mod
            break
        return _loaded[rel]


def fall_through(module_name: str):
    """A module-level ``__getattr__`` for ``module_name``: names this package does not define come
    from the shadowed file; AttributeError (as without it) when there is none or it lacks them."""
    def __getattr__(attr: str):
        if attr.startswith("__") and attr.endswith("__"):
            raise AttributeError(f"module {module_name!r} has no attribute {attr!r}")
        other = shadowed_module(module_name)
        if other is not None and hasattr(other, attr):
            return getattr(other, attr)
        raise AttributeError(f"module {module_name!r} has no attribute {attr!r}"
                             + ("" if other is None else f" (nor has {other.__file__})"))
    return __getattr__


HOT_FUNCTIONS = {
    # module (flat name) -> the names this package replaces there
    "math_utils.fit": ("cluster_DBSCAN", "fit_shape_RANSAC", "z_align_and_fit", "choose_and_cluster"),
    "geometry.point_cloud_processing": ("cluster_plus", "cluster_and_get_largest"),
    "geometry.skeletonize": ("extract_skeleton", "least_squares_sparse", "extract_topology",
                             "extract_skeletal_graph", "simplify_graph", "skeleton_to_QSM"),
    "geometry.reconstruction": ("get_neighbors_kdtree",),
    "viz.ray_casting": ("cast_rays", "sparse_cast_w_intersections", "get_points_inside_mesh",
                        "project_to_image", "raycast_to_pcd", "mri"),
    "utils.lib_integration": ("find_neighbors_in_ball", "get_neighbors_in_tree"),
    "tree_isolation": ("extend_seed_clusters",),
    "qsm_generation": ("fit_cyl_to_cluster",),
}


def install(verbose: bool = False) -> dict:
    """Patch the HIP wrappers into a pyQSM that is imported ALREADY (or laid out so that its own
    modules win on ``sys.path``): for every hot function, every module in ``sys.modules`` that
    holds the reference's function object under that name — the defining module and everybody who
    did ``from … import`` — is rebound to this package's function. Modules imported later are
    covered by putting this directory first on ``sys.path`` (done here too).
    Returns {"module.name": number of bindings replaced}."""
    import importlib
    if _PKG_DIR not in sys.path:
        sys.path.insert(0, _PKG_DIR)
    elif sys.path[0] != _PKG_DIR:
        sys.path.remove(_PKG_DIR)
        sys.path.insert(0, _PKG_DIR)
    replaced = {}
    for mod_name, names in HOT_FUNCTIONS.items():
        ours = importlib.import_module("pyqsm_amd." + mod_name)
        theirs = sys.modules.get(mod_name)
        if theirs is None or os.path.abspath(getattr(theirs, "__file__", "") or "").startswith(_PKG_DIR + os.sep):
            continue                                    # not imported, or already this package's
        for name in names:
            new = getattr(ours, name, None)
            old = theirs.__dict__.get(name)
            if new is None or old is None or old is new:
                continue
            count = 0
            for m in list(sys.modules.values()):
                d = getattr(m, "__dict__", None)
                if d is not None and d.get(name) is old:
                    d[name] = new
                    count += 1
            replaced[f"{mod_name}.{name}"] = count
            if verbose:
                print(f"pyqsm_amd.install: {mod_name}.{name} -> HIP ({count} bindings)")
    return replaced
