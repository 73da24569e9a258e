"""The masked multiplicative update (W = 'mask' on a negative-sampled matrix, MovieLens-1M shape, k = 16) against the fp64 oracle
(literal association over the dense mask), every iteration: factors, rec_error over the observed cells, whole-matrix counts.
Measurement aid (the golden sets g7 / g9 are small).  usage: masked_parity_probe.py [iters]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import oracle as orc  # noqa: E402
from pybmf_amd import _lib as L  # noqa: E402
from pybmf_amd.engine import BitMatrix, MaskedMUEngine, SparseObs  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rs = np.random.RandomState(0)
mm, nn, kk = 6040, 3706, 16
pu, pv = rs.pareto(1.2, mm) + 1, rs.pareto(1.2, nn) + 1
P = np.outer(pu / pu.sum(), pv / pv.sum())
ones = rs.rand(mm, nn) < np.minimum(P * 1_000_209, 1.0)
neg = (rs.rand(mm, nn) < ones.mean()) & ~ones
r, c = np.nonzero(ones | neg)
S = SparseObs(r, c, ones[r, c].astype(np.float32), None, (mm, nn))
eng = MaskedMUEngine(S, kk, L.MODE_PENALTY, bits=BitMatrix(ones.astype(np.uint8), "cuda:0"))
U0, V0 = np.abs(rs.standard_normal((mm, kk))) * 0.2, np.abs(rs.standard_normal((nn, kk))) * 0.2
eng.load_factors(U0, V0)
eng.prepare()
X = ones.astype(np.float64)
W = (ones | neg).astype(np.float64)
U, V = U0.copy(), V0.copy()
rel = lambda p, q: float(np.linalg.norm(p - q) / max(np.linalg.norm(q), 1e-300))  # noqa: E731
worst = 0.0
for t in range(T):
    reg = 1.02 ** t
    eng.update(reg)
    err, rec, rg, rmse, mae, cnt = eng.scalars(reg)
    V = orc.penalty_update_V(X, W, U, V, reg)
    U = orc.penalty_update_U(X, W, U, V, reg)
    e_o, rec_o, rg_o = orc.penalty_errors(X, W, U, V, reg)
    Ug, Vg = eng.factors()
    du, dv = rel(Ug, U), rel(Vg, V)
    worst = max(worst, du, dv)
    tp, fp, fn, tn = orc.confusion_counts(X, orc.boolean_product(U, V, 0.5, 0.5))
    print(f"iter {t:3d}: rel U {du:.2e} V {dv:.2e}  rec_error rel {abs(rec - rec_o) / rec_o:.1e}  reg_error rel {abs(rg - rg_o) / max(rg_o, 1e-300):.1e}  "
          f"counts {'equal' if tuple(cnt) == (tp, fp, fn, tn) else f'GPU {cnt} oracle {(tp, fp, fn, tn)}'}", flush=True)
print(f"worst relative distance over {T} iterations: {worst:.2e} (gate 1e-4)")
