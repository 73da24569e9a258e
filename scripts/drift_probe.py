"""How fast do operand-format differences grow along the C3 trajectory?  Runs the same schedule with bf16x3 / f16x2 /
bf16x2 operands and with bf16x3 from an initial state perturbed by 1e-7 (relative, random): the last one is the
amplification any fp32-level difference (e.g. another BLAS summation order) undergoes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from bench import host_init
from pybmf_amd import _lib as L
from pybmf_amd.engine import BitMatrix, MUEngine
from pybmf_amd.generators import PlantedBooleanOnDevice

m, n, k = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (100_000, 20_000, 64)))
dev = torch.device("cuda:0")
gen = PlantedBooleanOnDevice(m, n, k, density=(0.067, 0.067), seed=1000, noise=(0.05, 0.01), noise_seed=2000, device=dev)
X = BitMatrix(gen, dev)
iters = [1, 2, 5, 10, 20, 35]
regs = [1.0 * 1.02 ** i for i in range(max(iters))]
U0, V0 = None, None
runs = {}
for name, panel, terms, eps in (("bf16x3", "bf16", 3, 0.0), ("f16x2", "f16", 2, 0.0), ("bf16x2", "bf16", 2, 0.0), ("bf16x3+1e-7", "bf16", 3, 1e-7)):
    eng = MUEngine(X, k=k, mode=L.MODE_PENALTY, terms=terms, with_mae=False, tol=0.0, max_iter=100, panel=panel)
    if U0 is None:
        U0, V0 = host_init(eng.sum_x / (float(m) * n), m, n, k, seed=2024)
    rs = np.random.RandomState(1)
    Ua = U0 * (1.0 + eps * rs.standard_normal(U0.shape))
    eng.load_factors(Ua, V0)
    eng.prepare(regs[0])
    snaps, done = {}, 0
    for it in iters:
        eng.run(regs[done:it], it0=done + 1)
        done = it
        snaps[it] = eng.factors()
    runs[name] = snaps
    del eng
base = runs["bf16x3"]
rel = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))
print("iter  " + "  ".join(f"{nm:>24s}" for nm in list(runs)[1:]))
for it in iters:
    print(f"{it:4d}  " + "  ".join(f"U {rel(runs[nm][it][0], base[it][0]):.2e} V {rel(runs[nm][it][1], base[it][1]):.2e}" for nm in list(runs)[1:]))
