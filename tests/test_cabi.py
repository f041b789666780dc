"""The C-ABI library loads on a machine without a GPU, exports every symbol
include/pyqsm_hip.h declares, and fails loudly (no CPU fallback) when asked to compute."""
import ctypes
import os
import re

import numpy as np
import pytest

from pyqsm_amd import _lib, hip

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "pyqsm_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pyqsm_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_table_agree():
    assert _declared() == sorted(_lib.SIGNATURES)


def test_every_declared_symbol_is_exported():
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in _declared():
        assert hasattr(lib, name), name


def test_load_and_version():
    lib = _lib.load()
    assert lib.pyqsm_version().startswith(b"pyqsm_hip")
    assert _lib.device_count() >= 0


@pytest.mark.skipif(_lib.device_count() > 0, reason="checks the no-GPU failure mode")
def test_no_gpu_means_an_error_not_a_fallback():
    with pytest.raises(_lib.PyQSMHipError) as e:
        hip.dbscan(np.zeros((10, 3)), 0.1, 3)
    assert e.value.code == -3
    with pytest.raises(_lib.PyQSMHipError):
        hip.cast_rays(np.zeros((3, 3), np.float32), np.array([[0, 1, 2]], np.int32),
                      np.zeros((4, 6), np.float32))
    with pytest.raises(_lib.PyQSMHipError) as e:
        hip.cast_rays_multi(np.zeros((3, 3), np.float32), np.array([[0, 1, 2]], np.int32),
                            np.zeros((4, 6), np.float32), n_devices=0)
    assert e.value.code == -3


def test_product_never_imports_the_oracle():
    """pyqsm_amd/ must not reference oracle/ (the oracle is test infrastructure)."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "pyqsm_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(import|from)\s+oracle\b", src, flags=re.M), f
                assert "liboracle" not in src, f
