"""cProfile of one BinaryMFPenalty.fit(train, val, test, task='prediction') at MovieLens-1M shape (scripts/r03/prediction_bench.py's data)."""
import cProfile
import contextlib
import io
import os
import pstats
import sys

import numpy as np
from scipy.sparse import csr_matrix

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pybmf_amd.models import BinaryMFPenalty  # noqa: E402

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


def fit():
    with contextlib.redirect_stdout(io.StringIO()):
        mdl = BinaryMFPenalty(k=k, W="mask", init_method="normal", seed=5, reg=1.0, reg_growth=1.05, max_iter=40, tol=0.0, min_diff=0.0)
        mdl.fit(sets[0].copy(), sets[1].copy(), sets[2].copy(), task="prediction", show_logs=False, show_result=False, save_model=False)


fit()
pr = cProfile.Profile()
pr.enable()
fit()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(28)
