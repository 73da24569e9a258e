"""bmf_xf_f32_tiled / bmf_residual_sums_f32_tiled at the shapes of BASELINE config #2 (20000 x 5000 fp32, k = 32): us per launch, TB/s.
BMF_LIB selects an experimental flavour."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pybmf_amd import _lib as L
d = torch.device("cuda", 0)
m_pad, n_pad, kp = int(os.environ.get("M_PAD", "20096")), int(os.environ.get("N_PAD", "5120")), int(os.environ.get("KP", "32"))
X = torch.rand((m_pad, n_pad), device=d)
Xt = torch.empty(m_pad * n_pad, device=d)
L.check(L.lib.bmf_tile_f32(L.ptr(X), m_pad, n_pad, n_pad, L.ptr(Xt), None))
U, V = torch.rand((m_pad, kp), device=d), torch.rand((n_pad, kp), device=d)
Vfrag, Vrf = torch.empty(n_pad * kp, device=d), torch.empty(n_pad * kp, device=d)
L.check(L.lib.bmf_frag_f32(L.ptr(V), n_pad, kp, L.ptr(Vfrag), None))
L.check(L.lib.bmf_frag_rows_f32(L.ptr(V), n_pad, kp, L.ptr(Vrf), None))
sums = torch.zeros(4, dtype=torch.float64, device=d)
for splits in [int(v) for v in os.environ.get("SPLITS", "2,4,8").split(",")]:
    out = torch.zeros((splits, m_pad, kp), device=d)
    def run():
        L.check(L.lib.bmf_xf_f32_tiled(L.ptr(Xt), m_pad, n_pad, L.ptr(Vfrag), kp, L.ptr(out), m_pad * kp, splits, None))
    for _ in range(5): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 50
    print(f"xf_f32_tiled {m_pad}x{n_pad} kp={kp} splits={splits}: {us:.1f} us, {m_pad*n_pad*4/us/1e6:.2f} TB/s")
    if os.environ.get("CHECK", "0") == "1":   # against a float64 product
        want = X.double() @ V.double()
        err = ((out.double().sum(0) - want).abs().max() / want.abs().max()).item()
        print(f"  check vs fp64 product: max rel err {err:.2e}")
        assert err < 1e-5
if kp == 32:   # the fused pass: out = X^T U with the residual sums of the cells riding along (bmf_xf_f32_tiled_resid)
    XT = X.t().contiguous()
    XTt = torch.empty(m_pad * n_pad, device=d)
    L.check(L.lib.bmf_tile_f32(L.ptr(XT), n_pad, m_pad, m_pad, L.ptr(XTt), None))
    Ufrag, Urf = torch.empty(m_pad * kp, device=d), torch.empty(m_pad * kp, dtype=torch.int32, device=d)
    L.check(L.lib.bmf_frag_f32(L.ptr(U), m_pad, kp, L.ptr(Ufrag), None))
    L.check(L.lib.bmf_frag_rows_bf16(L.ptr(U), m_pad, kp, L.ptr(Urf), None))
    for splits in [int(v) for v in os.environ.get("SPLITS_T", "6,9,12,13").split(",")]:
        out = torch.zeros((splits, n_pad, kp), device=d)
        def run():
            L.check(L.lib.bmf_xf_f32_tiled_resid(L.ptr(XTt), n_pad, m_pad, L.ptr(Ufrag), L.ptr(Urf), L.ptr(V), kp, L.ptr(out), n_pad * kp, splits,
                                                 L.ptr(sums), None))
        for _ in range(5): run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): run()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 50
        print(f"xf_f32_tiled_resid {n_pad}x{m_pad} kp={kp} splits={splits}: {us:.1f} us, {m_pad*n_pad*4/us/1e6:.2f} TB/s")
    del XT, XTt
def rr():
    L.check(L.lib.bmf_residual_sums_f32_tiled(L.ptr(Xt), m_pad, n_pad, L.ptr(U), L.ptr(Vrf), kp, L.ptr(sums), None))
for _ in range(5): rr()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): rr()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / 50
print(f"residual tiled: {us:.1f} us, {m_pad*n_pad*4/us/1e6:.2f} TB/s")
if os.environ.get("STREAM_REF", "0") == "1":   # what plain streaming kernels of the framework reach on the same buffer
    Y = torch.empty_like(Xt)
    for name, fn, nbytes in (("torch sum (read)", lambda: Xt.sum(), Xt.numel() * 4), ("torch copy (read + write)", lambda: Y.copy_(Xt), Xt.numel() * 8)):
        for _ in range(5): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): fn()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 50
        print(f"{name}: {us:.1f} us, {nbytes/us/1e6:.2f} TB/s")
