"""BASELINE config #2 in the C-side loop (bmf_wnmf_real_run): WNMF on a 20000 x 5000 dense fp32 X, k = 32, error + RMSE + MAE every
iteration.  Prints it/s; the workload of bench.py's secondary.c2_wnmf_real."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pybmf_amd.engine import RealMatrix, RealMUEngine

m, n, k = 20000, 5000, 32
rs = np.random.RandomState(0)
X = ((rs.rand(m, 32) @ rs.rand(32, n)) / 32).astype(np.float32) + 0.01 * rs.rand(m, n).astype(np.float32)
R = RealMatrix(X, "cuda:0")
its, warm = int(os.environ.get("ITERS", "30")), 5
eng = RealMUEngine(R, k, with_mae=bool(int(os.environ.get("MAE", "1"))))
r2 = np.random.RandomState(2024)
avg = np.sqrt(X.mean() / k)
eng.load_factors(np.abs(avg * r2.standard_normal((m, k))), np.abs(avg * r2.standard_normal((n, k))))
eng.device_loop(max_iter=its + warm + 2)
eng.run(1, 1 + warm)
torch.cuda.synchronize()
t0 = time.perf_counter()
eng.run(1 + warm, 1 + warm + its)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / its
log = eng.read_log()
passes = 2 if getattr(eng, "_Urf", None) is not None else 3
print(f"C2 WNMF 20000x5000 k=32 fp32, C loop: {1/dt:.1f} it/s ({dt*1e3:.3f} ms/iteration); X = {X.nbytes/1e6:.0f} MB read {passes}x per iteration "
      f"=> {passes*X.nbytes/dt/1e12:.2f} TB/s; log rows {len(log[0]) if isinstance(log, tuple) else len(log)}")
