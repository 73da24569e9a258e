"""Where a config-#5 fit spends its time with the batched trace-form objective: setup, one batch of P points, one gradient."""
import sys, os, time, contextlib, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
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
    model = BinaryMFThreshold(k=k, U=w.U.copy(), V=w.V.copy(), W="full", u=0.3, v=0.3, lamda=10, min_diff=1e-3, max_iter=30)
    model.fit(X, **FIT)
for rep in range(2):
    t0 = time.perf_counter(); model._upload_factors(); torch.cuda.synchronize(); t1 = time.perf_counter()
    print(f"_upload_factors (incl. _setup_trace): {(t1 - t0) * 1e3:.2f} ms")
    t0 = time.perf_counter(); model._setup_trace(); torch.cuda.synchronize(); t1 = time.perf_counter()
    print(f"_setup_trace alone: {(t1 - t0) * 1e3:.2f} ms")
for P_, grad in ((1, False), (1, True), (8, False), (24, False), (32, False), (8, True), (32, True)):
    pts = [(0.3 + 1e-3 * i, 0.3 + 2e-3 * i) for i in range(P_)]
    for _ in range(3):
        model._eval_trace(pts, grad)
    t0 = time.perf_counter()
    for _ in range(20):
        model._eval_trace(pts, grad)
    dt = (time.perf_counter() - t0) / 20
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
    for a, b in ev:
        a.record(); model._eval_trace(pts, grad); b.record()
    torch.cuda.synchronize()
    print(f"P={P_:2d} grad={int(grad)}: wall {dt * 1e6:7.1f} us per batch; GPU events {sorted(a.elapsed_time(b) for a, b in ev)[5] * 1e3:7.1f} us")
import cProfile, pstats
with contextlib.redirect_stdout(io.StringIO()):
    model = BinaryMFThreshold(k=k, U=w.U.copy(), V=w.V.copy(), W="full", u=0.3, v=0.3, lamda=10, min_diff=1e-3, max_iter=30)
    pr = cProfile.Profile(); pr.enable(); model.fit(X, **FIT); pr.disable()
st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(22)
