"""Round 5: config #2's C-side loop with the contractions on the bf16 matrix instruction (both operands split three ways) against the
exact-fp32 instruction: iteration time, the two passes alone, and the trajectories side by side.  Measurement aid."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pybmf_amd import _lib as L
from pybmf_amd.engine import RealMatrix, RealMUEngine

m, n, k = 20000, 5000, 32
rs = np.random.RandomState(0)
X = ((rs.rand(m, 32) @ rs.rand(32, n)) / 32).astype(np.float32) + 0.01 * rs.rand(m, n).astype(np.float32)
R = RealMatrix(X, "cuda:0")
r2 = np.random.RandomState(2024)
avg = np.sqrt(X.mean() / k)
V0 = np.abs(avg * r2.standard_normal((n, k)))
U0 = np.abs(avg * r2.standard_normal((m, k)))
iters = 60
logs, facs = {}, {}
for rep in range(2):
    for bf3 in (False, True):
        eng = RealMUEngine(R, k, with_mae=True, bf16x3=bf3)
        eng.load_factors(U0, V0)
        eng.device_loop(max_iter=iters + 8)
        eng.run(1, 4)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.run(4, 4 + iters)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / iters
        log, stop = eng.read_log()
        logs[bf3] = log
        facs[bf3] = eng.factors()
        print(f"bf16x3={bf3}: {1e3 * dt:.4f} ms per iteration = {1 / dt:.0f} it/s; 0.8 GB -> {0.8 / dt / 1e3:.2f} TB/s; error {log[-1, 1]:.6f} RMSE {log[-1, 5]:.3e} MAE {log[-1, 6]:.3e}")
a, b = logs[False], logs[True]
print("log rows, max relative difference (error, RMSE, MAE):", [float(np.abs(a[:, c] - b[:, c]).max() / np.abs(a[:, c]).max()) for c in (1, 5, 6)])
print("factors after", 3 + iters, "iterations, relative Frobenius difference U, V:",
      [float(np.linalg.norm(facs[True][i] - facs[False][i]) / np.linalg.norm(facs[False][i])) for i in (0, 1)])
# the two passes alone
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
eng = RealMUEngine(R, k, with_mae=True, bf16x3=True)
eng.load_factors(U0, V0)
eng.device_loop(max_iter=4)
Xt, XTt = R.tiled()
calls = [("xf_f32_tiled (X V)", lambda: L.lib.bmf_xf_f32_tiled(L.ptr(Xt), R.m_pad, R.n_pad, L.ptr(eng.VT), 32, L.ptr(eng.Mslab), R.m_pad * 32, eng.splits_xv, st)),
         ("xf_f32_tiled_bf3 (X V)", lambda: L.lib.bmf_xf_f32_tiled_bf3(L.ptr(Xt), R.m_pad, R.n_pad, L.ptr(eng._VT3), L.ptr(eng.Mslab), R.m_pad * 32, eng.splits_xv, st)),
         ("xf_f32_tiled_resid (X^T U + sums)", lambda: L.lib.bmf_xf_f32_tiled_resid(L.ptr(XTt), R.n_pad, R.m_pad, L.ptr(eng.UT), L.ptr(eng._Urf), L.ptr(eng.V), 32, L.ptr(eng.Nslab),
                                                                                   R.n_pad * 32, eng.splits_xtu, L.ptr(eng.sums), st)),
         ("xf_f32_tiled_resid_bf3 (X^T U + sums)", lambda: L.lib.bmf_xf_f32_tiled_resid_bf3(L.ptr(XTt), R.n_pad, R.m_pad, L.ptr(eng._UT3), L.ptr(eng._Urf), L.ptr(eng.V),
                                                                                           L.ptr(eng.Nslab), R.n_pad * 32, eng.splits_xtu, L.ptr(eng.sums), st))]
for rep in range(2):
    for name, call in calls:
        for _ in range(3):
            L.check(call())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            L.check(call())
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 20
        print(f"{name}: {us:.1f} us per launch = {R.m_pad * R.n_pad * 4.0 / us / 1e6:.2f} TB/s")
