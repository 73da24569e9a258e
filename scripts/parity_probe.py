"""Drift of the GPU trajectory from the fp64 CPU oracle at a size between config #1 and C3 (default 20000 x 20000, k=64,
same generator/density/schedule as bench.py), per operand format.  Oracle = re-associated fp64 updates (oracle/, test
infrastructure); this script is a measurement aid, not part of the product."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import oracle as orc
from bench import host_init
from pybmf_amd import _lib as L
from pybmf_amd.engine import BitMatrix, MUEngine
from pybmf_amd.generators import PlantedBooleanOnDevice

m, n, k = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (20_000, 20_000, 64)))
iters = [int(x) for x in os.environ.get("ITERS", "1,5,10,20,35,60,100").split(",")]
dev = torch.device("cuda:0")
gen = PlantedBooleanOnDevice(m, n, k, density=(0.067, 0.067), seed=1000, noise=(0.05, 0.01), noise_seed=2000, device=dev)
X = BitMatrix(gen, dev)
regs = [min(1.0 * 1.02 ** i, 1e10) for i in range(max(iters))]
runs = {}
U0 = V0 = None
for name, panel, terms in (("i8x3", "i8", 3), ("bf16x3", "bf16", 3), ("f16x2", "f16", 2)):
    eng = MUEngine(X, k=k, mode=L.MODE_PENALTY, terms=terms, with_mae=False, tol=0.0, max_iter=200, panel=panel)
    if U0 is None:
        U0, V0 = host_init(eng.sum_x / (float(m) * n), m, n, k, seed=2024)
    eng.load_factors(U0, V0)
    eng.prepare(regs[0])
    snaps, done = {}, 0
    for it in iters:
        eng.run(regs[done:it], it0=done + 1)
        done = it
        snaps[it] = eng.factors()
    runs[name] = snaps
    del eng
Xh = X.rows_dense_u8(0, m).astype(np.float64)
U, V = U0.copy(), V0.copy()
rel = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))
t0 = time.time()
print("iter  " + "  ".join(f"{nm:>24s}" for nm in runs), flush=True)
for it in range(1, max(iters) + 1):
    V = orc.penalty_update_V_reassoc(Xh, U, V, regs[it - 1])
    U = orc.penalty_update_U_reassoc(Xh, U, V, regs[it - 1])
    if it in iters:
        print(f"{it:4d}  " + "  ".join(f"U {rel(runs[nm][it][0], U):.2e} V {rel(runs[nm][it][1], V):.2e}" for nm in runs)
              + f"   ({time.time() - t0:.0f} s)", flush=True)
