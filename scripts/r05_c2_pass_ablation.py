"""Round 5: the bf16 x 3 X V pass of config #2 (csrc/xf_f32.hip) with one ingredient removed (flavours: wrong results on purpose).
usage: BMF_LIB=libbmf_c2_<flavour>.so python scripts/r05_c2_pass_ablation.py"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pybmf_amd import _lib as L
from pybmf_amd.engine import RealMatrix

m, n, k = 20000, 5000, 32
rs = np.random.RandomState(0)
X = ((rs.rand(m, 32) @ rs.rand(32, n)) / 32).astype(np.float32) + 0.01 * rs.rand(m, n).astype(np.float32)
R = RealMatrix(X, "cuda:0")
dev = torch.device("cuda:0")
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
V = torch.rand((R.n_pad, 32), dtype=torch.float32, device=dev)
V3 = torch.empty(R.n_pad * 48, dtype=torch.int32, device=dev)
L.check(L.lib.bmf_frag_bf16x3(L.ptr(V), R.n_pad, L.ptr(V3), st))
Vf = torch.empty(R.n_pad * 32, dtype=torch.float32, device=dev)
L.check(L.lib.bmf_frag_f32(L.ptr(V), R.n_pad, 32, L.ptr(Vf), st))
Xt = R.tiled()[0]
splits = 13
out = torch.zeros((splits, R.m_pad, 32), dtype=torch.float32, device=dev)
res = []
for name, call in (("bf16 x 3", lambda: L.lib.bmf_xf_f32_tiled_bf3(L.ptr(Xt), R.m_pad, R.n_pad, L.ptr(V3), L.ptr(out), R.m_pad * 32, splits, st)),
                   ("exact fp32", lambda: L.lib.bmf_xf_f32_tiled(L.ptr(Xt), R.m_pad, R.n_pad, L.ptr(Vf), 32, L.ptr(out), R.m_pad * 32, splits, st))):
    for _ in range(5):
        L.check(call())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30):
        L.check(call())
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 30
    res.append(f"{name} {us:.1f} us ({R.m_pad * R.n_pad * 4.0 / us / 1e6:.2f} TB/s)")
print(os.environ.get("BMF_LIB", "libbmf_hip.so"), "|", " | ".join(res))
