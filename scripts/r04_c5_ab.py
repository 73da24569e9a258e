"""Config #5 A/B (round 4): the thresholding line search at MovieLens-1M shape with the batched trace-form objective
(csrc/thresh_trace.hip, the default) against the tile product of rounds 2-3 (BMF_THRESH_TRACE=0), same data, same search path.
Prints outer iterations / s of the whole fit() and the final (u, v, F), which must agree."""
import sys, os, time, contextlib, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pybmf_amd.models import BinaryMFThreshold, WNMF

rs = np.random.RandomState(11)
m, n, k = 6040, 3706, 16
pu, pv = rs.pareto(1.2, m) + 1, rs.pareto(1.2, n) + 1
P = np.outer(pu / pu.sum(), pv / pv.sum())
s = 1_000_209.0
for _ in range(60):
    s *= 1_000_209.0 / np.minimum(P * s, 1.0).sum()
X = (rs.rand(m, n) < np.minimum(P * s, 1.0)).astype(np.uint8)
FIT = dict(task="reconstruction", show_logs=False, show_result=False, save_model=False)
with contextlib.redirect_stdout(io.StringIO()):
    w = WNMF(k=k, W="full", init_method="normal", max_iter=20, seed=5)
    w.fit(X, **FIT)
res = {}
for lam in (10, 100):
    for trace in ("1", "0", "1", "0"):
        os.environ["BMF_THRESH_TRACE"] = trace
        with contextlib.redirect_stdout(io.StringIO()):
            model = BinaryMFThreshold(k=k, U=w.U.copy(), V=w.V.copy(), W="full", u=0.3, v=0.3, lamda=lam, min_diff=1e-3, max_iter=30)
            t0 = time.perf_counter()
            model.fit(X, **FIT)
            dt = time.perf_counter() - t0
        rows = model.logs["updates"]
        Fs = np.asarray(rows[("", "", "F")], dtype=np.float64)
        us = np.asarray(rows[("", "", "u")], dtype=np.float64)
        print(f"lamda {lam} trace={trace}: {model.n_iter} outer iterations in {dt * 1e3:.1f} ms = {model.n_iter / dt:.0f} it/s; u={model.u:.9f} v={model.v:.9f} F={Fs[-1]:.6f} rows={len(Fs)}")
        res.setdefault(lam, []).append((len(Fs), us, Fs))
    a, b = res[lam][0], res[lam][1]
    assert a[0] == b[0], "row counts differ"
    print(f"   trace vs tile product: max |du| {np.abs(a[1] - b[1]).max():.2e}, max rel dF {np.abs(a[2] / b[2] - 1).max():.2e}")
