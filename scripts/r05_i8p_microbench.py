"""Round 5: the anti-phase eight-wave bits GEMM (csrc/xf_bits_i8p.hip, variant 4) beside the default kernel at the headline shape:
sums of slabs equal to fp32 rounding, launch times of both orientations.  Measurement aid.  usage: r05_i8p_microbench.py [launches=30]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pybmf_amd import _lib as L
from pybmf_amd.engine import BitMatrix, xf_slots_i8
from pybmf_amd.generators import PlantedBooleanOnDevice

n_launch = int(sys.argv[1]) if len(sys.argv) > 1 else 30
m, n, k = (int(v) for v in os.environ.get("SHAPE", "100000,20000,64").split(","))
dev = torch.device("cuda:0")
X = BitMatrix(PlantedBooleanOnDevice(m, n, k, density=(0.067, 0.067), seed=1000, noise=(0.05, 0.01), noise_seed=2000, device=dev), dev)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
kp = 64


def timed(fn, reps=n_launch):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in evs:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return ts[len(ts) // 2] * 1e3


tiled = X.tiled()
for name, tb, rows_pad, ldw, red_pad in (("XV", tiled[0], X.m_pad, X.ldx, X.n_pad), ("XtU", tiled[1], X.n_pad, X.ldxt, X.m_pad)):
    g = torch.Generator(device=dev)
    g.manual_seed(11)
    F64 = torch.rand((red_pad, kp), dtype=torch.float64, device=dev, generator=g)
    F64[:, 3] *= 1e-3
    panel = torch.zeros((3, kp, red_pad), dtype=torch.int8, device=dev)
    scale = torch.zeros(2 * kp, dtype=torch.float32, device=dev)
    ws = torch.zeros(red_pad // 128 * kp, dtype=torch.float32, device=dev)
    L.check(L.lib.bmf_make_panel_i8(L.ptr(F64), L.ptr(F64.float()), red_pad, kp, kp, 3, L.ptr(panel), red_pad, L.ptr(ws), L.ptr(scale), st))
    res = {}
    for variant in (0, 4, 0, 4):
        L.check(min(0, L.lib.bmf_xf_bits_i8_variant(variant)))
        splits = xf_slots_i8(rows_pad, red_pad, kp)
        out = torch.zeros((splits, rows_pad, kp), dtype=torch.float32, device=dev)
        args = (L.ptr(tb), rows_pad, ldw, red_pad // 32, L.ptr(panel), red_pad, 3, L.ptr(scale[kp:]), kp, L.ptr(out), rows_pad * kp, splits, 1, st)
        L.check(L.lib.bmf_xf_bits_i8(*args))
        torch.cuda.synchronize()
        tot = out.double().sum(0)
        t = timed(lambda: L.check(L.lib.bmf_xf_bits_i8(*args)))
        if variant in res:
            print(f"[{name}] variant {variant}: {t:.1f} us (second pass)")
        else:
            res[variant] = tot
            print(f"[{name}] variant {variant}: {t:.1f} us, {splits} slab slots")
    L.lib.bmf_xf_bits_i8_variant(0)
    rel = float(((res[4] - res[0]).abs().max() / res[0].abs().max()).item())
    print(f"[{name}] variant 4 vs 0, sum of slabs: max |diff| / max = {rel:.3e}")
    assert rel < 3e-7
