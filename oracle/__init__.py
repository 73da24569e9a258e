"""CPU oracle for the PyBMF continuous-relaxation MU hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``pybmf_amd/`` may import this package.
Allowed callers: ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` -- and there only as the checker / the timed CPU baseline,
never as the product path.

Parity status: PINNED.  Every function in ``pybmf_oracle`` is checked in
``tests/test_oracle_golden.py`` against fixtures in ``tests/golden/`` that were
produced by importing the reference (``/root/reference``, tag 2024_10_08) in the
build container with ``tests/golden/make_golden.py`` (committed).
"""
from .pybmf_oracle import *  # noqa: F401,F403
