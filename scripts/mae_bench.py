"""Time bmf_mae_sum (split-bf16 MFMA pass for the MAE column) at the bench size.  usage: [BMF_LIB=...] python scripts/mae_bench.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pybmf_amd import _lib as L  # noqa: E402

m, n, k = (int(os.environ.get(v, d)) for v, d in (("M", 100_000), ("N", 20_000), ("K", 64)))
d = torch.device("cuda:0")
kp = 32 if k <= 32 else 64
m_pad, n_pad = -(-m // 512) * 512, -(-n // 512) * 512   # the padding of engine.BitMatrix
g = torch.Generator(device=d).manual_seed(1)
xt = torch.randint(-2**31, 2**31 - 1, (n_pad, m_pad // 32), dtype=torch.int32, device=d, generator=g)
xt &= torch.randint(-2**31, 2**31 - 1, xt.shape, dtype=torch.int32, device=d, generator=g)   # density 1/4
U = torch.rand((m_pad, kp), device=d, generator=g) * 0.2
V = torch.rand((n_pad, kp), device=d, generator=g) * 0.2
ws = torch.zeros(2 * (m_pad + n_pad) * kp, dtype=torch.int16, device=d)
out = torch.zeros(1, dtype=torch.float64, device=d)
st = L.C.c_void_p(torch.cuda.current_stream().cuda_stream)


tiled = os.environ.get("BMF_MAE_TILED", "0") == "1"
if tiled:
    xtt = torch.empty_like(xt)
    L.check(L.lib.bmf_tile_bits(L.ptr(xt), n_pad, m_pad // 32, m_pad // 32, L.ptr(xtt), st))


def call():
    if tiled:
        L.check(L.lib.bmf_mae_sum_tiled(L.ptr(xtt), m_pad // 32, m_pad, n_pad, L.ptr(U), L.ptr(V), kp, L.ptr(ws), L.ptr(out), st))
    else:
        L.check(L.lib.bmf_mae_sum(L.ptr(xt), m_pad // 32, m_pad, n_pad, L.ptr(U), L.ptr(V), kp, L.ptr(ws), L.ptr(out), st))


for _ in range(3):
    call()
torch.cuda.synchronize()
out.zero_()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = 20
e0.record()
for _ in range(reps):
    call()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
print(f"{os.environ.get('BMF_LIB', 'libbmf_hip.so')}{' [X^T tiled]' if tiled else ''}: bmf_mae_sum {m}x{n} k={k}: {ms:.4f} ms per call (incl. the two split kernels), "
      f"{6.0 * m * n * k / ms / 1e9:.0f} TFLOP/s hardware, sum/call = {float(out.item()) / reps:.9e}")
