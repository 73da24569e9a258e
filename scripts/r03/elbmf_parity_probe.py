"""ELBMF's one-call loop against the fp64 oracle loop at a mid size (default 20000 x 5000, k = 64, 30 iterations): relative distance of
the factors and of every log column, iteration by iteration.  Measurement aid (the golden tests are small: k = 6).
usage: elbmf_parity_probe.py [m n k iters beta]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import oracle as orc  # noqa: E402
from pybmf_amd import _lib as L  # noqa: E402
from pybmf_amd.engine import BitMatrix  # noqa: E402
from pybmf_amd.palm import PalmEngine  # noqa: E402

a = sys.argv[1:]
m, n, k, T = (int(a[i]) if len(a) > i else d for i, d in enumerate((20000, 5000, 64, 30)))
beta = float(a[4]) if len(a) > 4 else 0.0
rs = np.random.RandomState(31)
X = ((rs.rand(m, 24) < 0.12).astype(np.float32) @ (rs.rand(24, n) < 0.12).astype(np.float32) > 0).astype(np.uint8)
X ^= (rs.rand(m, n) < 0.01).astype(np.uint8)
U0, V0 = rs.rand(m, k) * 0.3, rs.rand(n, k) * 0.3
l1, l2, growth = 0.01, 0.02, 1.05
eng = PalmEngine(BitMatrix(X, "cuda:0"), k, L.PALM_ELBMF, beta=beta)
eng.load_factors(U0, V0)
rel = lambda p, q: float(np.linalg.norm(p - q) / max(np.linalg.norm(q), 1e-300))  # noqa: E731
# the oracle step by step (its fit function keeps only the final factors): reuse its update function
U, V, Ul, Vl = U0.copy(), V0.copy(), U0.copy(), V0.copy()
Xf = X.astype(np.float64)
worst = 0.0
for t in range(T):
    r1, r2 = l1, l2 * growth ** t
    eng.iterate(t, r1, r2, r1, r2)
    Un, Ul2 = orc.elbmf_update(Xf, U, V, None, r1, r2, beta, Ul, reassoc=True)
    Vn, Vl2 = orc.elbmf_update(Xf.T, V, U, None, r1, r2, beta, Vl, reassoc=True)
    U, V, Ul, Vl = Un, Vn, Ul2, Vl2
    err = float(((Xf - U @ V.T) ** 2).sum())
    got = eng.row(t)
    Ug, Vg = eng.factors()
    du, dv = rel(Ug, U), rel(Vg, V)
    worst = max(worst, du, dv)
    flips = int(((Ug > 0.5) != (U > 0.5)).sum() + ((Vg > 0.5) != (V > 0.5)).sum())
    print(f"iter {t:3d}: rel U {du:.2e} V {dv:.2e}  err GPU {got[0]:.6e} oracle {err:.6e} (rel {abs(got[0] - err) / err:.1e})  thresholded entries that differ: {flips}", flush=True)
print(f"worst relative distance over {T} iterations: {worst:.2e} (gate 1e-4)")
