"""BASELINE config #3 through the drop-in class: BinaryMFPenalty.fit() on a 100 000 x 20 000 Boolean HOST array (2 GB of uint8),
k = 64, 30 iterations -- wall time of the whole call (upload, packing on the device, loop with all scores, log tables, factors
back to the host)."""
import contextlib, io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pybmf_amd.models import BinaryMFPenalty

m, n, k = 100_000, 20_000, 64
rs = np.random.RandomState(1000)
A = (rs.rand(m, k) < 0.067).astype(np.uint8)
B = (rs.rand(n, k) < 0.067).astype(np.uint8)
t0 = time.perf_counter()
X = np.empty((m, n), dtype=np.uint8)
for r0 in range(0, m, 10_000):          # planted Boolean product, built in row chunks on the host
    X[r0:r0 + 10_000] = (A[r0:r0 + 10_000].astype(np.float32) @ B.T.astype(np.float32)) > 0
print(f"host matrix built in {time.perf_counter() - t0:.1f} s, density {X.mean():.3f}")
kw = dict(task="reconstruction", show_logs=False, show_result=False, save_model=False)
for rep in range(2):
    with contextlib.redirect_stdout(io.StringIO()):
        mdl = BinaryMFPenalty(k=k, W="full", reg=1, reg_growth=1.02, init_method="normal", normalize_method="balance", max_iter=29, seed=2024)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        mdl.fit(X, **kw)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    row = mdl.logs["updates"].iloc[-1]
    print(f"C3 fit() from a host array: {dt:.2f} s for {mdl.n_iter} updates with all scores (error {float(row[('', '', 'error')]):.1f}, "
          f"F1 {float(mdl.logs['boolean'].iloc[-1][('train', 0, 'F1')]):.4f})")
