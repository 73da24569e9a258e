"""How often the predicted digit-plane scales are off along the headline trajectory: the loop stepped one iteration per call, the flag
words of the column-scale check (scale[3 kp : 4 kp]) read after each, and the time of each iteration (events)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from pybmf_amd import _lib as L
from pybmf_amd.engine import BitMatrix, MUEngine
from pybmf_amd.generators import PlantedBooleanOnDevice
m, n, k = 100_000, 20_000, 64
dev = torch.device("cuda", 0)
gen = PlantedBooleanOnDevice(m, n, k, density=(0.067, 0.067), seed=1000, noise=(0.05, 0.01), noise_seed=2000, device=dev)
X = BitMatrix(gen, dev); del gen
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 80
eng = MUEngine(X, k=k, mode=L.MODE_PENALTY, terms=3, with_mae=False, tol=-1.0, min_diff=0.0, max_iter=iters + 2, panel="i8")
U0, V0 = bench.host_init(eng.sum_x / (float(m) * n), m, n, k, seed=2024)
eng.load_factors(U0, V0)
regs, r = [], 1.0
for _ in range(iters + 1):
    regs.append(r); r = min(r * 1.02, 1e10)
eng.prepare(regs[0])
kp = eng.kp
rows = []
for i in range(iters):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    eng.run([regs[i]], it0=1 + i)
    e1.record()
    torch.cuda.synchronize()
    fu = int((eng.scaleU[3 * kp:4 * kp] != 0).sum().item()); fv = int((eng.scaleV[3 * kp:4 * kp] != 0).sum().item())
    rows.append((i + 1, e0.elapsed_time(e1), fu, fv))
for a in range(0, iters, 10):
    print(" ".join(f"[{it}: {ms:.3f} ms U{fu} V{fv}]" for it, ms, fu, fv in rows[a:a + 10]))
