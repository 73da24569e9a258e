"""Masked path at MovieLens-1M shape (6040 x 3706, ~1e6 observed cells incl. sampled negatives, k = 16): time per iteration."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pybmf_amd import _lib as L
from pybmf_amd.engine import BitMatrix, MaskedMUEngine, SparseObs

rs = np.random.RandomState(0)
m, n, k = 6040, 3706, 16
pu, pv = rs.pareto(1.2, m) + 1, rs.pareto(1.2, n) + 1
P = np.outer(pu / pu.sum(), pv / pv.sum())
ones = rs.rand(m, n) < np.minimum(P * 1_000_209, 1.0)
neg = (rs.rand(m, n) < ones.mean()) & ~ones            # as many sampled negatives as positives
obs = ones | neg
r, c = np.nonzero(obs)
vals = ones[r, c].astype(np.float32)
print("observed cells:", len(r), "ones:", int(ones.sum()))
S = SparseObs(r, c, vals, None, (m, n))
B = BitMatrix(ones.astype(np.uint8), "cuda:0")
for with_scores in (False, True):
    eng = MaskedMUEngine(S, k, L.MODE_PENALTY, bits=B if with_scores else None)
    U0 = np.abs(rs.standard_normal((m, k))) * 0.2
    V0 = np.abs(rs.standard_normal((n, k))) * 0.2
    eng.load_factors(U0, V0)
    eng.prepare()
    for _ in range(3):
        eng.update(1.0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    iters = 50
    for i in range(iters):
        eng.update(1.0 * 1.02 ** i)
        if with_scores:
            eng.scalars(1.0)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    print(f"with whole-matrix scores each iteration={with_scores}: {dt*1e6:.0f} us/iteration = {1/dt:.0f} it/s; "
          f"gather traffic/iteration ~ {2*len(r)*2*32*4/1e6:.0f} MB")
