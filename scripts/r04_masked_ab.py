"""Masked BinaryMF-Penalty at MovieLens-1M shape: the loop as one C call per iteration with the scalars read one iteration late
(bmf_masked_iterate) against the stepwise update() + scalars() protocol.  The workload of bench.py's secondary.widened_engines.masked_penalty."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pybmf_amd import _lib as L
from pybmf_amd.engine import BitMatrix, MaskedMUEngine, SparseObs
rs = np.random.RandomState(0)
mm, nn, kk = 6040, 3706, 16
pu, pv = rs.pareto(1.2, mm) + 1, rs.pareto(1.2, nn) + 1
P = np.outer(pu / pu.sum(), pv / pv.sum())
ones = rs.rand(mm, nn) < np.minimum(P * 1_000_209, 1.0)
neg = (rs.rand(mm, nn) < ones.mean()) & ~ones
r, c = np.nonzero(ones | neg)
S = SparseObs(r, c, ones[r, c].astype(np.float32), None, (mm, nn))
U0, V0 = np.abs(rs.standard_normal((mm, kk))) * 0.2, np.abs(rs.standard_normal((nn, kk))) * 0.2
for mode in ("c-loop", "stepwise"):
    eng = MaskedMUEngine(S, kk, L.MODE_PENALTY, bits=BitMatrix(ones.astype(np.uint8), "cuda:0"))
    eng.load_factors(U0, V0)
    eng.prepare()
    warm, iters = 3, 200
    if mode == "c-loop":
        eng.iterate(0, 1.0, update=False)
        for i in range(1, warm + 1):
            eng.iterate(i, 1.0); eng.row(i - 1, 1.0)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(warm + 1, warm + 1 + iters):
            eng.iterate(i, 1.0); h = eng.row(i - 1, 1.0)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / iters
    else:
        for i in range(warm):
            eng.update(1.0); eng.scalars(1.0)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(iters):
            eng.update(1.0); h = eng.scalars(1.0)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / iters
    print(f"{mode}: {1/dt:.0f} it/s ({dt*1e3:.4f} ms/iteration), error {h[0]:.6e}")
