"""Where the host time of one ELBMF iteration goes when the loop is driven one C call per iteration (PalmEngine.iterate / .row):
per-call host times of iterate() and row(), GPU time between events.  usage: r04_palm_wait_probe.py [iters]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pybmf_amd import _lib as L
from pybmf_amd.engine import BitMatrix
from pybmf_amd.generators import PlantedBooleanOnDevice
from pybmf_amd.palm import PalmEngine
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
m, n, k = 100_000, 20_000, 64
gen = PlantedBooleanOnDevice(m, n, k, density=(0.067, 0.067), seed=1000, noise=(0.05, 0.01), noise_seed=2000, device="cuda:0")
X = BitMatrix(gen, "cuda:0")
eng = PalmEngine(X, k, L.PALM_ELBMF, beta=0.0, panel="i8")
rs = np.random.RandomState(3)
eng.load_factors(rs.rand(m, k) * 0.2, rs.rand(n, k) * 0.2)
sched = lambda i: (0.01, 0.02 * 1.02 ** i, 0.01, 0.02 * 1.02 ** i)
for i in range(3):
    eng.iterate(i, *sched(i))
eng.row(2)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t_it, t_row = [], []
e0.record()
t0 = time.perf_counter()
eng.iterate(3, *sched(3))
for i in range(3, 3 + iters):
    a = time.perf_counter()
    eng.iterate(i + 1, *sched(i + 1))
    b = time.perf_counter()
    eng.row(i)
    c = time.perf_counter()
    t_it.append(b - a); t_row.append(c - b)
e1.record()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"threads torch {torch.get_num_threads()} OMP {os.environ.get('OMP_NUM_THREADS')}: wall {1e3 * dt / (iters + 1):.3f} ms/iteration, GPU span {e0.elapsed_time(e1) / (iters + 1):.3f} ms/iteration; "
      f"slowest iterate() calls (index: ms) {[(int(j), round(1e3 * t_it[j], 2)) for j in np.argsort(t_it)[-3:][::-1]]}; "
      f"iterate() median {1e3 * np.median(t_it):.3f} ms max {1e3 * max(t_it):.3f}; row() median {1e3 * np.median(t_row):.3f} ms max {1e3 * max(t_row):.3f}")
