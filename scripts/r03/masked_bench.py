"""The masked multiplicative update (W = 'mask' on a negative-sampled matrix) at MovieLens-1M shape, k = 16, as bench.py's
secondary.widened_engines.masked_penalty runs it: iterations/s, and where an iteration goes (host call time vs a stream synchronisation
after each phase).  Measurement aid."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pybmf_amd import _lib as L  # noqa: E402
from pybmf_amd.engine import BitMatrix, MaskedMUEngine, SparseObs  # noqa: E402

rs = np.random.RandomState(0)
mm, nn, kk = 6040, 3706, 16
pu, pv = rs.pareto(1.2, mm) + 1, rs.pareto(1.2, nn) + 1
P = np.outer(pu / pu.sum(), pv / pv.sum())
ones = rs.rand(mm, nn) < np.minimum(P * 1_000_209, 1.0)
neg = (rs.rand(mm, nn) < ones.mean()) & ~ones
r, c = np.nonzero(ones | neg)
S = SparseObs(r, c, ones[r, c].astype(np.float32), None, (mm, nn))
eng = MaskedMUEngine(S, kk, L.MODE_PENALTY, bits=BitMatrix(ones.astype(np.uint8), "cuda:0"))
eng.load_factors(np.abs(rs.standard_normal((mm, kk))) * 0.2, np.abs(rs.standard_normal((nn, kk))) * 0.2)
eng.prepare()
for i in range(5):
    eng.update(1.02 ** i)
    eng.scalars(1.0)
torch.cuda.synchronize()
n = 200
t0 = time.perf_counter()
for i in range(n):
    eng.update(1.02 ** i)
    eng.scalars(1.0)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
# the same with the host time of the calls alone (nothing waited for until the end)
t0 = time.perf_counter()
for i in range(n):
    eng.update(1.0)
t_host_update = (time.perf_counter() - t0) / n
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(n):
    eng.update(1.0)
torch.cuda.synchronize()
t_dev_update = (time.perf_counter() - t0) / n
print(f"masked penalty {mm}x{nn}, {len(r)} cells, k={kk}: {1 / dt:.0f} it/s = {1e6 * dt:.0f} us per iteration (update + scalars); "
      f"update alone: host {1e6 * t_host_update:.0f} us of calls, {1e6 * t_dev_update:.0f} us with the device")
