"""The recommender workflow of the reference's examples at MovieLens-1M shape: a negative-sampled matrix split into train / val / test
(csr with explicit zeros), `fit(X_train, X_val, X_test, task='prediction')` -- every iteration scores all three sets over their entries.
BinaryMFPenalty (W = 'mask'), k = 16.  Seconds per fit and per iteration.  Run under rocprofv3 --kernel-trace --stats to see the kernels."""
import contextlib
import io
import os
import sys
import time

import numpy as np
from scipy.sparse import csr_matrix

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pybmf_amd.models import BinaryMFPenalty, WNMF  # noqa: E402

rs = np.random.RandomState(11)
m, n, k = 6040, 3706, 16
pu, pv = rs.pareto(1.2, m) + 1, rs.pareto(1.2, n) + 1
P = np.outer(pu / pu.sum(), pv / pv.sum())
ones = rs.rand(m, n) < np.minimum(P * 1_000_209, 1.0)
neg = (rs.rand(m, n) < ones.mean()) & ~ones
r, c = np.nonzero(ones | neg)
v = ones[r, c].astype(np.float64)
part = rs.rand(len(r))
sets = []
for lo, hi in ((0.0, 0.8), (0.8, 0.9), (0.9, 1.0)):
    sel = (part >= lo) & (part < hi)
    sets.append(csr_matrix((v[sel], (r[sel], c[sel])), shape=(m, n)))
X_train, X_val, X_test = sets
for name, cls, kw in (("BinaryMFPenalty", BinaryMFPenalty, dict(reg=1.0, reg_growth=1.05, max_iter=40, tol=0.0, min_diff=0.0)),
                      ("WNMF", WNMF, dict(max_iter=40, tol=0.0, min_diff=0.0))):
    best = None
    for rep in range(2):
        with contextlib.redirect_stdout(io.StringIO()):
            mdl = cls(k=k, W="mask", init_method="normal", seed=5, **kw)
            t0 = time.perf_counter()
            mdl.fit(X_train.copy(), X_val.copy(), X_test.copy(), task="prediction", show_logs=False, show_result=False, save_model=False)
            dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    print(f"{name} fit(train, val, test, task='prediction'), {X_train.nnz} / {X_val.nnz} / {X_test.nnz} entries, k={k}: {mdl.n_iter} iterations in {best:.3f} s "
          f"= {1e3 * best / max(mdl.n_iter, 1):.2f} ms per iteration incl. upload and the three score sets", flush=True)
