"""Timing of the tile-fused link pass (PNLPF / WNMF-KL) at the C3 shape: ms per update pair and fraction of the fp32-MFMA peak.
usage: python scripts/link_bench.py [m n k]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from bench import host_init
from pybmf_amd import _lib as L
from pybmf_amd.engine import BitMatrix, LinkMUEngine
from pybmf_amd.generators import PlantedBooleanOnDevice

m, n, k = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (100_000, 20_000, 64)))
dev = torch.device("cuda:0")
gen = PlantedBooleanOnDevice(m, n, k, density=(0.067, 0.067), seed=1000, noise=(0.05, 0.01), noise_seed=2000, device=dev)
X = BitMatrix(gen, dev)
U0, V0 = host_init(X.sum_local / (float(m) * n), m, n, k, seed=2024)
PEAK = 157.3
for name, link, mode, flops_per_side in (("PNLPF (sigmoid link)", L.LINK_SIGMOID, L.MODE_PENALTY, 6.0 * m * n * k),
                                         ("WNMF Kullback-Leibler", L.LINK_KL, L.MODE_WNMF, 4.0 * m * n * k)):
    eng = LinkMUEngine(X, k, link, mode, lamda=10.0)
    eng.load_factors(U0, V0)
    eng.prepare()
    eng.update(1.0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    its = 5
    for _ in range(its):
        eng.update(1.0)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / its
    t0 = time.perf_counter()
    for _ in range(its):
        eng.scalars(1.0)
    ds = (time.perf_counter() - t0) / its
    tf = 2 * flops_per_side / dt / 1e12
    print(f"{name}: update (V then U) {dt * 1e3:.2f} ms = {tf:.1f} TFLOP/s algorithmic = {tf / PEAK:.2f} of the fp32-MFMA peak; "
          f"scalars pass {ds * 1e3:.2f} ms; splits U/V = {eng.splitsU}/{eng.splitsV}")
    del eng
