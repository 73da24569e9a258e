import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np
np.set_printoptions(precision=9, linewidth=200)
from test_edge_cases_gpu import run_engine, oracle_run
m = n = k = 1
rs = np.random.RandomState(m * 1000 + n + k)
X = (rs.rand(m, n) < 0.4).astype(np.uint8)
U0 = np.abs(rs.standard_normal((m, k))) * 0.3 + 1e-3
V0 = np.abs(rs.standard_normal((n, k))) * 0.3 + 1e-3
regs = [0.5 * 1.2 ** i for i in range(5)]
ref = oracle_run(X, U0, V0, 0.5, 1.2, 5)
want = np.array(ref["updates"])
print("X", X, "U0", U0, "V0", V0)
for panel in ("f16", "i8"):
    L, log, U, V = run_engine(X, U0, V0, regs, panel=panel)
    got = log[:, [L.LOG_ERROR, L.LOG_REC, L.LOG_REGERR, L.LOG_RMSE, L.LOG_MAE]]
    print(panel, "U", U, ref["U"], "V", V, ref["V"])
    print(got - want[:, [1, 2, 4, 5, 6]])
