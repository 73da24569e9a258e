"""Times the factor-update epilogue alone at the headline shape: the plain form + the stand-alone int8 plane builder, against the
form that emits the planes itself.  Measurement aid.  usage: epi_bench.py [rows=100352] [kp=64] [splits=2]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from pybmf_amd import _lib as L

rows_pad = int(sys.argv[1]) if len(sys.argv) > 1 else 100352
kp = int(sys.argv[2]) if len(sys.argv) > 2 else 64
splits = int(sys.argv[3]) if len(sys.argv) > 3 else 2
d = torch.device("cuda:0")
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
g = torch.Generator(device=d).manual_seed(1)
F64 = torch.rand((rows_pad, kp), dtype=torch.float64, device=d, generator=g) * 0.5
F = F64.float()
num = torch.rand((splits, rows_pad, kp), dtype=torch.float32, device=d, generator=g) * 3
G = torch.rand((kp, kp), dtype=torch.float32, device=d, generator=g) * 50
G = (G + G.t()).contiguous()
rowbits = torch.zeros(rows_pad, dtype=torch.int64, device=d)
colbits = torch.zeros((kp, rows_pad // 32), dtype=torch.int32, device=d)
partials = torch.zeros((rows_pad // 128, 2), dtype=torch.float64, device=d)
blockmax = torch.zeros((rows_pad // 128, kp), dtype=torch.float32, device=d)
planes = torch.zeros((3, kp, rows_pad), dtype=torch.int8, device=d)
scale = torch.zeros(2 * kp, dtype=torch.float32, device=d)
ws = torch.zeros(rows_pad // 128 * kp, dtype=torch.float32, device=d)
L.check(L.lib.bmf_make_panel_i8(L.ptr(F64), L.ptr(F), rows_pad, kp, kp, 3, L.ptr(planes), rows_pad, L.ptr(ws), L.ptr(scale), st))


def args(with_planes):
    a = L.EpilogueArgs()
    a.F64, a.F, a.rows_pad, a.rows, a.k, a.kp = F64.data_ptr(), F.data_ptr(), rows_pad, rows_pad - 3, kp, kp
    a.num, a.slab_stride, a.splits = num.data_ptr(), rows_pad * kp, splits
    a.G, a.reg, a.mode, a.thr, a.terms = G.data_ptr(), 1.0, 1, 0.5, 0
    a.panel, a.ldp, a.rowbits, a.colbits, a.ldcb = 0, rows_pad, rowbits.data_ptr(), colbits.data_ptr(), rows_pad // 32
    a.partials, a.stop, a.blockmax = partials.data_ptr(), 0, blockmax.data_ptr()
    if with_planes:
        a.planes, a.plane_scale, a.limbs = planes.data_ptr(), scale.data_ptr(), 3
    return a


def timeit(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
    return ts[len(ts) // 2], ts[0]


a0, a1 = args(False), args(True)
t_plain = timeit(lambda: L.check(L.lib.bmf_mu_epilogue(C.byref(a0), st)))
t_build = timeit(lambda: L.check(L.lib.bmf_make_panel_i8(L.ptr(F64), L.ptr(F), rows_pad, kp, kp, 3, L.ptr(planes), rows_pad, L.ptr(ws), L.ptr(scale), st)))
t_fused = timeit(lambda: L.check(L.lib.bmf_mu_epilogue(C.byref(a1), st)))
mb = rows_pad * kp * (8 + 8 + 4 + 4 + 4 * splits + 3) / 1e6
print(f"rows_pad {rows_pad} kp {kp} splits {splits}: plain epilogue median {t_plain[0]:.1f} us (min {t_plain[1]:.1f}) + stand-alone builder (3 kernels) {t_build[0]:.1f} us; "
      f"fused epilogue {t_fused[0]:.1f} us (min {t_fused[1]:.1f}) = {mb / t_fused[0] / 1e3:.2f} TB/s of its {mb:.0f} MB")

if hasattr(L.lib, "bmf_debug_epi_stamps"):   # diagnostic flavour (-DBMF_EPI_STAMP): phases of wave 0 of workgroup 0 in the last fused launch
    buf = (C.c_ulonglong * 16)()
    assert L.lib.bmf_debug_epi_stamps(buf) == 0
    t = list(buf)
    ghz = (t[12] - t[0]) / max(1, (t[15] - t[14])) * 0.1
    names = ["stop word", "operand + first chunks landed", "F G issued/waited"] + [f"chunk {q}" for q in range(8)] + ["planes / bit-columns / block sums"]
    print(f"[epi stamps] clock {ghz:.2f} GHz; total {(t[12] - t[0]) / ghz / 1e3:.2f} us: " +
          "; ".join(f"{nm} {(t[i + 1] - t[i]) / ghz / 1e3:.2f}" for i, nm in enumerate(names)))
