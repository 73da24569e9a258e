"""ELBMF's iPALM loop (PyBMF/models/ELBMF.py:107-163) on the GPU engine at the headline shape: iterations/s, Python-driven as in the
model class (two proximal steps, two refreshes = panel + Gram + norms + bits GEMM each, one read-back of the scalars).
usage: python scripts/palm_bench.py [m n k iters]   (BMF_PALM_PANEL=i8|f16)"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pybmf_amd import _lib as L  # noqa: E402
from pybmf_amd.engine import BitMatrix  # noqa: E402
from pybmf_amd.generators import PlantedBooleanOnDevice  # noqa: E402
from pybmf_amd.palm import PalmEngine  # noqa: E402


def run(m=100_000, n=20_000, k=64, iters=30, panel="i8", device="cuda:0"):
    gen = PlantedBooleanOnDevice(m, n, k, density=(0.067, 0.067), seed=1000, noise=(0.05, 0.01), noise_seed=2000, device=device)
    X = BitMatrix(gen, device)
    eng = PalmEngine(X, k, L.PALM_ELBMF, beta=0.0, panel=panel)
    rs = np.random.RandomState(3)
    eng.load_factors(rs.rand(m, k) * 0.2, rs.rand(n, k) * 0.2)
    l1, l2, growth = 0.01, 0.02, 1.02

    def it(i):
        a, b = l1, l2 * growth ** i
        eng.step("U", a, b, a, b)
        eng.step("V", a, b, a, b)
        eng.refresh("U")
        eng.refresh("V")
        return eng.scalars()

    if os.environ.get("BMF_PALM_LOOP", "c") == "python":
        for i in range(3):
            it(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(3, 3 + iters):
            err, ug, vg, cnt = it(i)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    else:   # one C call per iteration, the scalars of iteration t read while t + 1 runs (how ELBMF.iPALM drives it)
        sched = lambda i: (l1, l2 * growth ** i, l1, l2 * growth ** i)   # noqa: E731
        warm = 12   # (>= 8: every process but the first on a box stalls ~37 ms ONCE inside its ~7th iteration -- a one-off of the runtime, scripts/r04_palm_wait_probe.py)
        for i in range(warm):
            eng.iterate(i, *sched(i))
        eng.row(warm - 1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.iterate(warm, *sched(warm))
        for i in range(warm, warm + iters):
            eng.iterate(i + 1, *sched(i + 1))
            err, ug, vg, cnt = eng.row(i)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        dt *= iters / (iters + 1)   # (iters + 1 iterations ran inside the window)
    if os.environ.get("BMF_LIB", "").endswith("normcount.so"):
        print("squarings of the last norms launch (U, V):", float(eng.normsU[1]), float(eng.normsV[1]))
    return {"config": f"ELBMF iPALM, {m}x{n} Boolean, k={k}, beta=0, operands {panel}", "iterations_per_s": iters / dt, "ms_per_iteration": 1e3 * dt / iters,
            "final_error": float(err), "counts_TP_FP_FN_TN": [int(c) for c in cnt]}


if __name__ == "__main__":
    a = [int(v) for v in sys.argv[1:5]]
    print(run(*a, panel=os.environ.get("BMF_PALM_PANEL", "i8")))
