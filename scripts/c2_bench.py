"""BASELINE config #2: WNMF multiplicative updates on a 20000 x 5000 dense fp32 X, k=32 (SURVEY 8d recipe).  it/s of the
Python-driven device loop (update + scalars per iteration)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pybmf_amd.engine import RealMatrix, RealMUEngine

m, n, k = 20000, 5000, 32
rs = np.random.RandomState(0)
X = ((rs.rand(m, 32) @ rs.rand(32, n)) / 32).astype(np.float32) + 0.01 * rs.rand(m, n).astype(np.float32)
R = RealMatrix(X, "cuda:0")
eng = RealMUEngine(R, k, with_mae=bool(int(os.environ.get("MAE", "1"))))
r2 = np.random.RandomState(2024)
avg = np.sqrt(X.mean() / k)
V0 = np.abs(avg * r2.standard_normal((n, k)))
U0 = np.abs(avg * r2.standard_normal((m, k)))
eng.load_factors(U0, V0)
for _ in range(3):
    eng.update(); eng.scalars()
torch.cuda.synchronize()
its = 30
t0 = time.perf_counter()
for _ in range(its):
    eng.update()
    e = eng.scalars()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / its
print(f"C2 WNMF 20000x5000 k=32 fp32: {1/dt:.1f} it/s ({dt*1e3:.3f} ms/iteration, error {e[0]:.4f}); X = {X.nbytes/1e6:.0f} MB read 3x per iteration "
      f"=> {3*X.nbytes/dt/1e12:.2f} TB/s")
