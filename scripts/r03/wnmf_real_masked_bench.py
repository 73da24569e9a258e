"""WNMF on real-valued ratings (1..5) under W = 'mask' at MovieLens-1M shape, fit(train, val, test, task='prediction'): the classic
recommender use of the reference's WNMF.  Seconds per fit; run under rocprofv3 --kernel-trace --stats to see the kernels."""
import contextlib
import io
import os
import sys
import time

import numpy as np
from scipy.sparse import csr_matrix

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pybmf_amd.models import WNMF  # noqa: E402

rs = np.random.RandomState(11)
m, n, k = 6040, 3706, 16
pu, pv = rs.pareto(1.2, m) + 1, rs.pareto(1.2, n) + 1
P = np.outer(pu / pu.sum(), pv / pv.sum())
ones = rs.rand(m, n) < np.minimum(P * 1_000_209, 1.0)
r, c = np.nonzero(ones)
v = rs.randint(1, 6, size=len(r)).astype(np.float64)
part = rs.rand(len(r))
sets = []
for lo, hi in ((0.0, 0.8), (0.8, 0.9), (0.9, 1.0)):
    sel = (part >= lo) & (part < hi)
    sets.append(csr_matrix((v[sel], (r[sel], c[sel])), shape=(m, n)))
best = None
for rep in range(2):
    with contextlib.redirect_stdout(io.StringIO()):
        mdl = WNMF(k=k, W="mask", init_method="normal", seed=5, max_iter=40, tol=0.0, min_diff=0.0)
        t0 = time.perf_counter()
        mdl.fit(sets[0].copy(), sets[1].copy(), sets[2].copy(), task="prediction", show_logs=False, show_result=False, save_model=False)
        dt = time.perf_counter() - t0
    best = dt if best is None else min(best, dt)
print(f"WNMF on ratings, W='mask', {sets[0].nnz} / {sets[1].nnz} / {sets[2].nnz} entries, k={k}: {mdl.n_iter} iterations in {best:.3f} s", flush=True)
