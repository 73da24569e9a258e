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
    dg = os.environ.get("DIGITS", "")   # power probe: overwrite the digit planes with a chosen distribution (results then mean nothing)
    if dg:
        g = torch.Generator(device=dev); g.manual_seed(7)
        lo_, hi_ = {"full": (-128, 128), "pos7": (0, 128), "pos6": (0, 64), "pos4": (0, 16), "neg7": (-128, 0), "zero": (0, 1), "one": (1, 2), "m1": (-1, 0)}[dg]
        panel.copy_(torch.randint(lo_, hi_, panel.shape, generator=g, device=dev, dtype=torch.int8))
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
    if hasattr(L.lib, "bmf_debug_read_stamps"):   # diagnostic flavour (-DBMF_EXP_STAMP): per-workgroup stamps of the last launch
        import numpy as np
        buf = (C.c_ulonglong * 2048)()
        assert L.lib.bmf_debug_read_stamps(buf) == 0
        st_ = np.array(buf, dtype=np.uint64).reshape(512, 4)
        st_ = st_[st_[:, 1] > 0]
        r0, r1, cyc = st_[:, 0].astype(np.float64), st_[:, 1].astype(np.float64), st_[:, 2].astype(np.float64)
        xcc = (st_[:, 3] >> np.uint64(32)).astype(np.int64) & 0xf
        hw = st_[:, 3].astype(np.int64) & 0xffffffff
        t0 = r0.min()
        dur = (r1 - r0) * 0.01   # us (100 MHz)
        print(f"[stamps {name}] {len(st_)} workgroups; kernel span {(r1.max() - t0) * 0.01:.1f} us; start skew max {(r0.max() - t0) * 0.01:.2f} us "
              f"(p50 {np.median(r0 - t0) * 0.01:.2f}); duration mean {dur.mean():.1f} min {dur.min():.1f} p50 {np.median(dur):.1f} p95 {np.percentile(dur, 95):.1f} max {dur.max():.1f} us; "
              f"end: first {(r1.min() - t0) * 0.01:.1f} last {(r1.max() - t0) * 0.01:.1f} us; clock {np.median(cyc / (r1 - r0)) * 0.1:.3f} GHz (min {(cyc / (r1 - r0)).min() * 0.1:.3f} max {(cyc / (r1 - r0)).max() * 0.1:.3f})")
        for x in sorted(set(xcc.tolist())):
            sel = xcc == x
            print(f"   xcc {x}: {sel.sum()} wgs, mean duration {dur[sel].mean():.1f} us, max {dur[sel].max():.1f}, start p50 {np.median(r0[sel] - t0) * 0.01:.2f} us, clock {np.median((cyc / (r1 - r0))[sel]) * 0.1:.3f} GHz")
        bidx = np.nonzero(np.array(buf, dtype=np.uint64).reshape(512, 4)[:, 1] > 0)[0]
        halves_ = kp // 32
        half_of = (bidx >> 3) % halves_
        for nm, sel in (("blockIdx < 256", bidx < 256), ("blockIdx >= 256", bidx >= 256), ("column half 0", half_of == 0), ("column half 1", half_of == 1)):
            if sel.any():
                print(f"   {nm}: {sel.sum()} wgs, duration mean {dur[sel].mean():.1f} p50 {np.median(dur[sel]):.1f} min {dur[sel].min():.1f} max {dur[sel].max():.1f} us")
        os.makedirs("gpurun_out/stamps", exist_ok=True)
        np.save(f"gpurun_out/stamps/{name}_{m}.npy", np.column_stack([bidx, st_.astype(np.int64)]))
        cu = {}
        pairs = {}
        for b_, x, h, d_ in zip(bidx.tolist(), xcc.tolist(), hw.tolist(), dur.tolist()):
            pairs.setdefault((x, (h >> 8) & 0xff), []).append((b_, d_))
        two = [sorted(v) for v in pairs.values() if len(v) == 2]
        if two:
            lo = np.array([v[0][1] for v in two]); hi = np.array([v[1][1] for v in two])
            print(f"   CUs with two workgroups: {len(two)}; lower blockIdx of the pair: mean {lo.mean():.1f} us, higher: mean {hi.mean():.1f} us; "
                  f"pairs where the lower blockIdx finished first: {(lo < hi).sum()}; blockIdx distance of a pair: median {np.median([v[1][0] - v[0][0] for v in two]):.0f}")
        for x, h, d_ in zip(xcc.tolist(), hw.tolist(), dur.tolist()):
            cu.setdefault((x, (h >> 8) & 0xff), []).append(d_)   # HW_ID bits 8..15: cu, sh, se
        import collections
        print(f"   distinct (xcc, se, sh, cu) ids {len(cu)}; workgroups per CU: {dict(collections.Counter(len(v) for v in cu.values()))}")
print(f"[tiled {tiled}]", end=" ")
print(f"[occupancy {L.lib.bmf_xf_bits_i8_occupancy(limbs)} WG/CU]", os.environ.get("BMF_LIB", "libbmf_hip.so"), " ".join(f"{nm}: median {v[0]:.1f} us min {v[1]:.1f} us" for nm, v in res.items()),
      f"| mean of medians {sum(v[0] for v in res.values()) / 2:.1f} us")
