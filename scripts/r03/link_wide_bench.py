"""The link models (PNLPF, WNMF-KL) and the rank-128 engine at the headline shape, as bench.py's secondary.widened_engines runs them:
one line per engine.  Run under rocprofv3 --kernel-trace --stats to see where their iterations go.  usage: link_wide_bench.py [pnlpf|kl|wide ...]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pybmf_amd import _lib as L  # noqa: E402
from pybmf_amd.engine import BitMatrix, LinkMUEngine  # noqa: E402
from pybmf_amd.generators import PlantedBooleanOnDevice  # noqa: E402

which = sys.argv[1:] or ["pnlpf", "kl", "wide"]
m, n, k = 100_000, 20_000, 64
gen = PlantedBooleanOnDevice(m, n, k, density=(0.067, 0.067), seed=1000, noise=(0.05, 0.01), noise_seed=2000, device="cuda:0")
X = BitMatrix(gen, "cuda:0")
rs = np.random.RandomState(3)
avg = np.sqrt(X.sum_local / (float(m) * n) / k)
U0, V0 = np.abs(avg * rs.standard_normal((m, k))) + 1e-6, np.abs(avg * rs.standard_normal((n, k))) + 1e-6


def timed(fn, iters, warm=1):
    for i in range(warm):
        fn(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(warm, warm + iters):
        fn(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


for name, link, mode in (("pnlpf", L.LINK_SIGMOID, L.MODE_PENALTY), ("kl", L.LINK_KL, L.MODE_WNMF)):
    if name not in which:
        continue
    eng = LinkMUEngine(X, k, link, mode, lamda=10.0)
    eng.load_factors(U0, V0)
    eng.prepare()

    def it(i, eng=eng):
        eng.update(1.0)
        eng.scalars(1.0)
    dt = timed(it, 4)
    print(f"{name}: {1e3 * dt:.2f} ms per iteration (update pair + scalars) = {1 / dt:.1f} it/s", flush=True)
    del eng
if "wide" in which:
    from pybmf_amd.wide import WideMUEngine
    kw = 128
    avg = np.sqrt(X.sum_local / (float(m) * n) / kw)
    eng = WideMUEngine(X, kw, L.MODE_PENALTY, with_mae=True)
    eng.load_factors(np.abs(avg * rs.standard_normal((m, kw))) + 1e-6, np.abs(avg * rs.standard_normal((n, kw))) + 1e-6)
    eng.prepare()

    def wit(i):
        eng.update(1.02 ** i)
        eng.scalars(1.02 ** i)
    dt = timed(wit, 6)
    print(f"rank 128: {1e3 * dt:.2f} ms per iteration = {1 / dt:.1f} it/s", flush=True)
