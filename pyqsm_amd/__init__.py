"""pyqsm_amd — MI355X (gfx950) implementation of pyQSM's point-cloud geometry hot path.

Host code is plain Python + NumPy calling hand-written HIP kernels through the
C-ABI of ``libpyqsm_hip.so`` (include/pyqsm_hip.h) with ctypes. The sub-packages
``math_utils``, ``geometry`` and ``viz`` mirror pyQSM's flat module layout so that
putting this directory on ``sys.path`` gives ``from math_utils.fit import
cluster_DBSCAN`` etc. the HIP implementation.
"""
__version__ = "0.1.0"

from ._shadow import install  # noqa: E402,F401  (patch the HIP wrappers into an imported pyQSM)
