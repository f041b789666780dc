"""Route A of INTEGRATION.md: pyqsm_amd/ ahead of pyQSM's directory on sys.path.

pyQSM's geometry/, math_utils/, viz/, utils/ are namespace packages (no __init__.py) and its
drivers import far more from them than the hot path (pyQSM/pipeline.py:8-10,
pyQSM/qsm_generation.py:17-52, pyQSM/canopy_metrics.py:19-25). This test lays out a STAND-IN
tree with that shape in a temporary directory — builder-written stub modules whose functions
return markers, not copies of reference files — and checks, in a fresh interpreter, that

* every sibling import of the stand-in drivers still resolves (to the stand-in's stubs),
* the hot functions resolve to this package's HIP wrappers everywhere, including inside the
  stand-in's own modules that are reached through the fall-through,
* names a shadowed module does not restate (kmeans, clean_cloud, get_angles, sphere_step …)
  come from the stand-in's module of the same name,
* PY_QSM_LOG_CONFIG is honoured and, without it, the stand-in's own log.yml keeps configuring
  the `calc` logger (pyQSM/set_config.py:16-42),
* pyqsm_amd.install() patches a pyQSM that was imported first.
No GPU is touched: only import resolution is exercised."""
import json
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "pyqsm_amd")


def _stub(names, origin):
    return "".join(f"def {n}(*a, **k):\n    return ('standin', '{origin}.{n}')\n\n" for n in names)


STANDIN = {
    "set_config.py": "import logging\nconfig = {'standin': True}\nlog = logging.getLogger('calc')\n",
    "log.yml": textwrap.dedent("""\
        version: 1
        disable_existing_loggers: false
        handlers:
            console:
                class: utils.log_utils.ConsoleHandler
                level: ERROR
        loggers:
            calc:
                level: DEBUG
                handlers: [console]
                propagate: no
        """),
    "pipeline.py": textwrap.dedent("""\
        from set_config import log
        from utils.io import convert_las, np_to_o3d, load
        from canopy_metrics import get_downsample
        from utils.general import list_if

        def loop_over_files(func, *args, **kwargs):
            return func({'seed': 1, 'src': None}, *args, **kwargs)
        """),
    "canopy_metrics.py": textwrap.dedent("""\
        from geometry.skeletonize import extract_skeleton, extract_topology
        from viz.ray_casting import project_pcd
        from set_config import config, log

        def get_downsample(*a, **k):
            return ('standin', 'canopy_metrics.get_downsample')

        def get_shift(file_content, *a, **k):
            return extract_skeleton
        """),
    "qsm_generation.py": textwrap.dedent("""\
        from geometry.skeletonize import extract_skeleton, extract_topology
        from tree_isolation import extend_seed_clusters
        from math_utils.fit import cluster_DBSCAN, fit_shape_RANSAC, kmeans
        from math_utils.fit import choose_and_cluster, cluster_DBSCAN, fit_shape_RANSAC, kmeans
        from utils.io import save, load, save_line_set
        from utils.lib_integration import find_neighbors_in_ball
        from viz.color import split_on_percentile
        from viz.viz_utils import color_continuous_map
        from math_utils.general import (get_angles, get_center, get_radius, rotation_matrix_from_arr,
                                        unit_vector, get_percentile)
        from set_config import config
        from geometry.point_cloud_processing import (cluster_plus, crop_by_percentile, filter_by_norm,
                                                     clean_cloud, crop, get_shape, create_one_or_many_pcds,
                                                     orientation_from_norms, get_ball_mesh)
        from viz.viz_utils import iter_draw, draw
        from geometry.point_cloud_processing import join_pcds
        from geometry.reconstruction import get_neighbors_kdtree
        from tree_isolation import pcds_from_extend_seed_file
        from geometry.mesh_processing import map_density
        from viz.plotting import plot_dist_dist

        def fit_cyl_to_cluster(*a, **k):
            return ('standin', 'qsm_generation.fit_cyl_to_cluster')

        def sphere_step(*a, **k):
            return {'dbscan': cluster_DBSCAN, 'ransac': fit_shape_RANSAC, 'kmeans': kmeans,
                    'ball': find_neighbors_in_ball, 'clean': clean_cloud, 'plus': cluster_plus}
        """),
    "tree_isolation.py": _stub(["extend_seed_clusters", "pcds_from_extend_seed_file"], "tree_isolation"),
    "utils/io.py": _stub(["convert_las", "np_to_o3d", "load", "save", "save_line_set"], "utils.io"),
    "utils/general.py": _stub(["list_if"], "utils.general"),
    "utils/lib_integration.py": _stub(["find_neighbors_in_ball", "pts_to_cloud"], "utils.lib_integration"),
    "utils/log_utils.py": "import logging\n\nclass ConsoleHandler(logging.StreamHandler):\n    pass\n",
    "geometry/skeletonize.py": _stub(["extract_skeleton", "extract_topology"], "geometry.skeletonize"),
    "geometry/point_cloud_processing.py": _stub(
        ["cluster_plus", "crop_by_percentile", "filter_by_norm", "clean_cloud", "crop", "get_shape",
         "create_one_or_many_pcds", "orientation_from_norms", "get_ball_mesh", "join_pcds"],
        "geometry.point_cloud_processing"),
    "geometry/reconstruction.py": _stub(["get_neighbors_kdtree"], "geometry.reconstruction"),
    "geometry/mesh_processing.py": _stub(["map_density"], "geometry.mesh_processing"),
    "math_utils/fit.py": _stub(["cluster_DBSCAN", "fit_shape_RANSAC", "kmeans", "choose_and_cluster"],
                               "math_utils.fit"),
    "math_utils/general.py": _stub(["get_angles", "get_center", "get_radius", "rotation_matrix_from_arr",
                                    "unit_vector", "get_percentile"], "math_utils.general"),
    "viz/color.py": _stub(["split_on_percentile"], "viz.color"),
    "viz/viz_utils.py": _stub(["color_continuous_map", "iter_draw", "draw"], "viz.viz_utils"),
    "viz/plotting.py": _stub(["plot_dist_dist"], "viz.plotting"),
    "viz/ray_casting.py": _stub(["project_pcd", "cast_rays"], "viz.ray_casting"),
}


def _lay_out(tmp_path):
    base = tmp_path / "standin_pyqsm"
    for rel, text in STANDIN.items():
        f = base / rel
        f.parent.mkdir(parents=True, exist_ok=True)
        f.write_text(text)
    return str(base)


def _run(code, pythonpath, extra_env=None):
    env = {k: v for k, v in os.environ.items() if not k.startswith("PY_QSM")}
    env["PYTHONPATH"] = os.pathsep.join(pythonpath)
    env.update(extra_env or {})
    r = subprocess.run([sys.executable, "-c", textwrap.dedent(code)], capture_output=True, text=True,
                       env=env, cwd="/tmp", timeout=300)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


CHECK = """
    import json, logging, os
    import pipeline, qsm_generation, canopy_metrics, set_config
    def where(f):
        return f.__code__.co_filename if hasattr(f, '__code__') else None
    step = qsm_generation.sphere_step()          # the stand-in's own function, reached by fall-through
    print(json.dumps({
        'pipeline_siblings': [pipeline.convert_las(), pipeline.load(), pipeline.list_if(),
                              pipeline.get_downsample()],
        'hot': {name: where(f) for name, f in {
            'cluster_DBSCAN': qsm_generation.cluster_DBSCAN, 'fit_shape_RANSAC': qsm_generation.fit_shape_RANSAC,
            'cluster_plus': qsm_generation.cluster_plus, 'extract_skeleton': qsm_generation.extract_skeleton,
            'extract_topology': qsm_generation.extract_topology,
            'extend_seed_clusters': qsm_generation.extend_seed_clusters,
            'find_neighbors_in_ball': qsm_generation.find_neighbors_in_ball,
            'get_neighbors_kdtree': qsm_generation.get_neighbors_kdtree,
            'canopy_extract_skeleton': canopy_metrics.get_shift({}),
            'choose_and_cluster': qsm_generation.choose_and_cluster,
            'fit_cyl_to_cluster': qsm_generation.fit_cyl_to_cluster}.items()},
        'inside_standin': {k: where(v) for k, v in step.items()},
        'fallthrough': [qsm_generation.kmeans(), qsm_generation.clean_cloud(), qsm_generation.get_shape(),
                        qsm_generation.crop_by_percentile(), qsm_generation.get_angles(),
                        qsm_generation.map_density(), qsm_generation.pcds_from_extend_seed_file(),
                        canopy_metrics.project_pcd(), qsm_generation.save()],
        'cast_rays': where(__import__('viz.ray_casting', fromlist=['cast_rays']).cast_rays),
        'config_has_dbscan': 'dbscan' in set_config.config,
        'log_config_file': set_config.log_config_file, 'log_applied': set_config.log_config_applied,
        'calc_level': logging.getLogger('calc').level,
        'calc_handlers': [type(h).__name__ for h in logging.getLogger('calc').handlers],
        'pipeline_log_name': pipeline.log.name,
    }))
"""


def test_every_sibling_import_resolves_and_the_hot_path_is_ours(tmp_path):
    standin = _lay_out(tmp_path)
    got = _run(CHECK, [PKG, standin])
    assert got["pipeline_siblings"] == [["standin", "utils.io.convert_las"], ["standin", "utils.io.load"],
                                        ["standin", "utils.general.list_if"],
                                        ["standin", "canopy_metrics.get_downsample"]]
    for name, path in got["hot"].items():
        assert path and path.startswith(PKG + os.sep), (name, path)
    # the stand-in's sphere_step lives in ITS qsm_generation.py; what it calls is ours
    for name in ("dbscan", "ransac", "ball", "plus"):
        assert got["inside_standin"][name].startswith(PKG + os.sep), got["inside_standin"]
    for name in ("kmeans", "clean"):
        assert got["inside_standin"][name].startswith(standin + os.sep), got["inside_standin"]
    assert [m[1] for m in got["fallthrough"]] == [
        "math_utils.fit.kmeans", "geometry.point_cloud_processing.clean_cloud",
        "geometry.point_cloud_processing.get_shape", "geometry.point_cloud_processing.crop_by_percentile",
        "math_utils.general.get_angles", "geometry.mesh_processing.map_density",
        "tree_isolation.pcds_from_extend_seed_file", "viz.ray_casting.project_pcd", "utils.io.save"]
    assert got["cast_rays"].startswith(PKG + os.sep)
    assert got["config_has_dbscan"]
    # no PY_QSM_LOG_CONFIG: the stand-in's own log.yml still does the bootstrap (its handler class
    # lives in the stand-in's utils/ portion of the merged namespace package)
    assert got["log_config_file"] == os.path.join(standin, "log.yml") and got["log_applied"]
    assert got["calc_level"] == 10 and got["calc_handlers"] == ["ConsoleHandler"]
    assert got["pipeline_log_name"] == "calc"


def test_py_qsm_log_config_is_honoured(tmp_path):
    standin = _lay_out(tmp_path)
    cfg = tmp_path / "mylog.yml"
    cfg.write_text(textwrap.dedent("""\
        version: 1
        disable_existing_loggers: false
        handlers:
            f:
                class: logging.FileHandler
                filename: %s
        loggers:
            calc:
                level: ERROR
                handlers: [f]
        """ % (tmp_path / "calc.log")))
    got = _run(CHECK, [PKG, standin], {"PY_QSM_LOG_CONFIG": str(cfg)})
    assert got["log_config_file"] == str(cfg) and got["log_applied"]
    assert got["calc_level"] == 40 and got["calc_handlers"] == ["FileHandler"]
    # without pyQSM on the path: the packaged log.yml; a broken file: an error line, defaults kept
    code = """
        import json, logging
        from pyqsm_amd import set_config
        print(json.dumps({'file': set_config.log_config_file, 'applied': set_config.log_config_applied,
                          'handlers': [type(h).__name__ for h in logging.getLogger('calc').handlers]}))
    """
    alone = _run(code, [ROOT])
    assert alone["file"] == os.path.join(PKG, "log.yml") and alone["applied"]
    assert alone["handlers"] == ["StreamHandler"]
    bad = tmp_path / "bad.yml"
    bad.write_text("version: 1\nhandlers: {x: {class: no.such.Handler}}\nroot: {handlers: [x]}\n")
    broken = _run(code, [ROOT], {"PY_QSM_LOG_CONFIG": str(bad)})
    assert broken["file"] == str(bad) and not broken["applied"]


def test_install_patches_a_pyqsm_that_was_imported_first(tmp_path):
    standin = _lay_out(tmp_path)
    code = """
        import json
        import qsm_generation, canopy_metrics                 # all stand-in: pyqsm_amd is not ahead
        import math_utils.fit
        before = qsm_generation.cluster_DBSCAN()
        import pyqsm_amd
        replaced = pyqsm_amd.install()
        def where(f):
            return f.__code__.co_filename
        print(json.dumps({'before': before, 'replaced': replaced,
                          'after': [where(qsm_generation.cluster_DBSCAN), where(math_utils.fit.cluster_DBSCAN),
                                    where(qsm_generation.fit_shape_RANSAC), where(canopy_metrics.extract_skeleton),
                                    where(qsm_generation.cluster_plus), where(qsm_generation.fit_cyl_to_cluster)],
                          'kept': qsm_generation.kmeans()}))
    """
    got = _run(code, [standin, ROOT])
    assert got["before"] == ["standin", "math_utils.fit.cluster_DBSCAN"]
    assert all(p.startswith(PKG + os.sep) for p in got["after"]), got["after"]
    assert got["replaced"]["math_utils.fit.cluster_DBSCAN"] >= 2         # the module and its importer
    assert got["replaced"]["geometry.skeletonize.extract_skeleton"] >= 3   # + canopy_metrics
    assert got["kept"] == ["standin", "math_utils.fit.kmeans"]
