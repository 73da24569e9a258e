"""BASELINE config #5: BinaryMFThreshold line search at MovieLens-1M shape (6040 x 3706, k=16) on a shape/density-matched
stand-in.  Reports outer iterations/s, F/dF evaluations/s and the time of one evaluation."""
import sys, os, time, contextlib, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pybmf_amd.models import BinaryMFThreshold, WNMF

rs = np.random.RandomState(11)
m, n, k = 6040, 3706, 16
pu, pv = rs.pareto(1.2, m) + 1, rs.pareto(1.2, n) + 1
P = np.outer(pu / pu.sum(), pv / pv.sum())
s = 1_000_209.0
for _ in range(60):
    s *= 1_000_209.0 / np.minimum(P * s, 1.0).sum()
X = (rs.rand(m, n) < np.minimum(P * s, 1.0)).astype(np.uint8)   # (sums to 1 000 209 expected ones: bench.py::secondary_c5)
FIT = dict(task="reconstruction", show_logs=False, show_result=False, save_model=False)
with contextlib.redirect_stdout(io.StringIO()):
    w = WNMF(k=k, W="full", init_method="normal", max_iter=20, seed=5)
    w.fit(X, **FIT)
    model = BinaryMFThreshold(k=k, U=w.U.copy(), V=w.V.copy(), W="full", u=0.3, v=0.3, lamda=10, min_diff=1e-3, max_iter=30)
    calls = {"F": 0, "dF": 0}
    F0, dF0 = model.F, model.dF
    model.F = lambda x: (calls.__setitem__("F", calls["F"] + 1), F0(x))[1]
    model.dF = lambda x: (calls.__setitem__("dF", calls["dF"] + 1), dF0(x))[1]
    t0 = time.perf_counter()
    model.fit(X, **FIT)
    dt = time.perf_counter() - t0
    t1 = time.perf_counter()
    for _ in range(50):
        F0([0.3, 0.3])
    tF = (time.perf_counter() - t1) / 50
    t1 = time.perf_counter()
    for _ in range(50):
        dF0([0.3, 0.3])
    tdF = (time.perf_counter() - t1) / 50
print(f"C5 thresholding 6040x3706 k=16: {model.n_iter} outer iterations in {dt:.3f} s (incl. upload) = {model.n_iter / dt:.1f} it/s; "
      f"{calls['F']} F + {calls['dF']} dF evaluations; F {tF * 1e6:.0f} us, dF {tdF * 1e6:.0f} us per call; u={model.u:.4f} v={model.v:.4f}")
