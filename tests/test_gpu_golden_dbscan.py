"""HIP DBSCAN against outputs of scikit-learn itself (tests/golden/dbscan_*.npz)."""
import glob
import os

import numpy as np
import pytest

from pyqsm_amd import hip

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "dbscan_*.npz"))))
def test_labels_and_core_set_bit_exact(gpu, path):
    g = np.load(path)
    lab, core = hip.dbscan(g["points"], float(g["eps"]), int(g["min_pts"]), device=gpu)
    assert np.array_equal(lab, g["labels"])
    assert np.array_equal(np.flatnonzero(core), g["core"])
