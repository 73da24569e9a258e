"""BinaryMFThreshold with its DEFAULT W='mask' on a negative-sampled csr at MovieLens-1M shape (k = 16): the objective runs over the
observed cells (csrc/thresh64.hip: bmf_thresh_transform64 + bmf_masked_thresh64).  Outer iterations/s and the time of one evaluation."""
import contextlib
import io
import os
import sys
import time

import numpy as np
from scipy.sparse import csr_matrix

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pybmf_amd.models import BinaryMFThreshold, WNMF  # noqa: E402

rs = np.random.RandomState(11)
m, n, k = 6040, 3706, 16
pu, pv = rs.pareto(1.2, m) + 1, rs.pareto(1.2, n) + 1
P = np.outer(pu / pu.sum(), pv / pv.sum())
ones = rs.rand(m, n) < np.minimum(P * 1_000_209, 1.0)
neg = (rs.rand(m, n) < ones.mean()) & ~ones
r, c = np.nonzero(ones | neg)
Xs = csr_matrix((ones[r, c].astype(np.float64), (r, c)), shape=(m, n))   # explicit zeros = the sampled negatives
FIT = dict(task="reconstruction", show_logs=False, show_result=False, save_model=False)
with contextlib.redirect_stdout(io.StringIO()):
    w = WNMF(k=k, W="mask", init_method="normal", max_iter=20, seed=5)
    w.fit(Xs.copy(), **FIT)
    model = BinaryMFThreshold(k=k, U=w.U.copy(), V=w.V.copy(), u=0.3, v=0.3, lamda=10, min_diff=1e-3, max_iter=30)   # W='mask' by default
    calls = {"F": 0, "dF": 0}
    F0, dF0 = model.F, model.dF
    model.F = lambda x: (calls.__setitem__("F", calls["F"] + 1), F0(x))[1]
    model.dF = lambda x: (calls.__setitem__("dF", calls["dF"] + 1), dF0(x))[1]
    t0 = time.perf_counter()
    model.fit(Xs.copy(), **FIT)
    dt = time.perf_counter() - t0
    t1 = time.perf_counter()
    for _ in range(50):
        F0([0.3, 0.3])
    tF = (time.perf_counter() - t1) / 50
    t1 = time.perf_counter()
    for _ in range(50):
        dF0([0.3, 0.3])
    tdF = (time.perf_counter() - t1) / 50
print(f"C5 under W='mask' ({len(r)} observed cells) 6040x3706 k=16: {model.n_iter} outer iterations in {dt:.3f} s = {model.n_iter / dt:.1f} it/s; "
      f"{calls['F']} F + {calls['dF']} dF evaluations; F {tF * 1e6:.0f} us, dF {tdF * 1e6:.0f} us per call; u={model.u:.4f} v={model.v:.4f}")
