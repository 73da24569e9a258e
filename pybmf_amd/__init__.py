"""pybmf_amd -- MI355X-native drop-in for PyBMF's continuous-relaxation multiplicative-update hot path.

Host-side mirror of the reference's class surface (``BinaryMFPenalty``, ``WNMF``, ``BinaryMFThreshold`` with
PyBMF's ``fit()/evaluate()`` protocol) over hand-written gfx950 HIP kernels in ``csrc/`` reached through the
C ABI of ``include/bmf_hip.h``.  Importing the package loads ``csrc/libbmf_hip.so`` and fails loudly if it is
missing: there is no CPU fallback.
"""
from . import _lib  # noqa: F401  (loads libbmf_hip.so; raises if absent)

__version__ = "0.1.0"
