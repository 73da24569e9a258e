"""Is the link pass limited by what the matrix pipe may draw?  The same launches with real factors and with all-zero factors (every MFMA then
multiplies zeros: same instruction stream, same memory traffic, almost no switching in the multipliers)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import host_init
from pybmf_amd import _lib as L
from pybmf_amd.engine import BitMatrix, LinkMUEngine
from pybmf_amd.generators import PlantedBooleanOnDevice
m, n, k = 100_000, 20_000, 64
dev = torch.device("cuda:0")
X = BitMatrix(PlantedBooleanOnDevice(m, n, k, density=(0.067, 0.067), seed=1000, noise=(0.05, 0.01), noise_seed=2000, device=dev), dev)
U0, V0 = host_init(X.sum_local / (float(m) * n), m, n, k, seed=2024)
for name, link, mode in (("sigmoid", L.LINK_SIGMOID, L.MODE_PENALTY), ("KL", L.LINK_KL, L.MODE_WNMF)):
    for what, (U, V) in (("real factors", (U0, V0)), ("zero factors", (np.zeros_like(U0), np.zeros_like(V0)))):
        eng = LinkMUEngine(X, k, link, mode, lamda=10.0)
        eng.load_factors(U, V); eng.prepare()
        def side():   # the V-side pass only, on the operands as loaded (no epilogue: the factors stay what they are)
            with torch.cuda.device(dev):
                L.check(L.lib.bmf_link_pass16(L.ptr(X.bits), eng.m_pad, X.ldx, m, n, L.ptr(eng.wsU), L.ptr(eng.wsV), eng.n_pad, eng.kp, link, 10.0,
                                              L.ptr(eng.numU), L.ptr(eng.denU_slabs) if eng.denU_slabs is not None else None, eng.m_pad * eng.kp, eng.splitsU, None))
        for _ in range(3): side()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): side()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
        print(f"{name}, {what}: U-side pass {dt*1e3:.3f} ms")
        del eng
