"""Times the int8 bits GEMM alone at the headline shape (both orientations), for A/B runs of kernel flavours (BMF_LIB=...).
Measurement aid, not part of the product.  usage: gemm_i8_microbench.py [launches=40] [limbs=3]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pybmf_amd import _lib as L
from pybmf_amd.engine import BitMatrix, xf_slots_i8
from pybmf_amd.generators import PlantedBooleanOnDevice

n_launch = int(sys.argv[1]) if len(sys.argv) > 1 else 40
limbs = int(sys.argv[2]) if len(sys.argv) > 2 else 3
m, n, k = (int(v) for v in os.environ.get("SHAPE", "100000,20000,64").split(","))
kp = 32 if k <= 32 else 64
dev = torch.device("cuda:0")
X = BitMatrix(PlantedBooleanOnDevice(m, n, k, density=(0.067, 0.067), seed=1000, noise=(0.05, 0.01), noise_seed=2000, device=dev), dev)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
res = {}
tiled = int(os.environ.get("TILED", "1"))
xt = X.tiled() if tiled else (X.bits, X.bits_t)
for name, bits, rows_pad, ldw, red_pad in (("XV", xt[0], X.m_pad, X.ldx, X.n_pad), ("XtU", xt[1], X.n_pad, X.ldxt, X.m_pad)):
    F64 = torch.rand((red_pad, kp), dtype=torch.float64, device=dev)
    F32 = F64.float()
    panel = torch.zeros((limbs, kp, red_pad), dtype=torch.int8, device=dev)
    scale = torch.zeros(2 * kp, dtype=torch.float32, device=dev)
    ws = torch.zeros(red_pad // 128 * kp, dtype=torch.float32, device=dev)
    L.check(L.lib.bmf_make_panel_i8(L.ptr(F64), L.ptr(F32), red_pad, kp, kp, limbs, L.ptr(panel), red_pad, L.ptr(ws), L.ptr(scale), st))
    splits = xf_slots_i8(rows_pad, red_pad, kp)
    out = torch.zeros((splits, rows_pad, kp), dtype=torch.float32, device=dev)
    args = (L.ptr(bits), rows_pad, ldw, red_pad // 32, L.ptr(panel), red_pad, limbs, L.ptr(scale[kp:]), kp, L.ptr(out), rows_pad * kp, splits, tiled, st)
    for _ in range(5):
        L.check(L.lib.bmf_xf_bits_i8(*args))
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_launch)]
    for a, b in evs:
        a.record()
        L.check(L.lib.bmf_xf_bits_i8(*args))
        b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    res[name] = (ts[len(ts) // 2] * 1e3, ts[0] * 1e3)
print(f"[tiled {tiled}]", end=" ")
print(f"[occupancy {L.lib.bmf_xf_bits_i8_occupancy(limbs)} WG/CU]", os.environ.get("BMF_LIB", "libbmf_hip.so"), " ".join(f"{nm}: median {v[0]:.1f} us min {v[1]:.1f} us" for nm, v in res.items()),
      f"| mean of medians {sum(v[0] for v in res.values()) / 2:.1f} us")
