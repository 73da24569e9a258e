"""cProfile of BinaryMFThreshold.fit at the C5 shape: where does the host time go?"""
import sys, os, cProfile, pstats, contextlib, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from pybmf_amd.models import BinaryMFThreshold, WNMF
rs = np.random.RandomState(11)
m, n, k = 6040, 3706, 16
pu, pv = rs.pareto(1.2, m) + 1, rs.pareto(1.2, n) + 1
P = np.outer(pu / pu.sum(), pv / pv.sum())
X = (rs.rand(m, n) < np.minimum(P * 1_000_209, 1.0)).astype(np.uint8)
FIT = dict(task="reconstruction", show_logs=False, show_result=False, save_model=False)
with contextlib.redirect_stdout(io.StringIO()):
    w = WNMF(k=k, W="full", init_method="normal", max_iter=20, seed=5)
    w.fit(X, **FIT)
    for rep in range(2):
        model = BinaryMFThreshold(k=k, U=w.U.copy(), V=w.V.copy(), W="full", u=0.3, v=0.3, lamda=10, min_diff=1e-3, max_iter=30)
        pr = cProfile.Profile()
        pr.enable()
        model.fit(X, **FIT)
        pr.disable()
st = pstats.Stats(pr, stream=sys.stdout)
st.sort_stats("cumulative").print_stats(35)
