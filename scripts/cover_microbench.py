"""Cost model of the cover kernel: time vs. number of set factor bits per row (C3 shape)."""
import ctypes as C
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pybmf_amd import _lib as L

d = torch.device("cuda:0")
m_pad, n_pad, kp = 100352, 20480, 64
ldx = n_pad // 32
X = torch.randint(-2**31, 2**31 - 1, (m_pad, ldx), dtype=torch.int32, device=d)
vcol = torch.randint(-2**31, 2**31 - 1, (kp, ldx), dtype=torch.int32, device=d)
counts = torch.zeros(4, dtype=torch.int64, device=d)
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
rs = np.random.RandomState(0)
for p in (0, 1, 2, 4, 8, 16, 32, 64):
    u = np.zeros(m_pad, np.uint64)
    for _ in range(p):
        pass
    bits = np.zeros((m_pad, 64), np.uint8)
    if p:
        idx = np.argsort(rs.rand(m_pad, 64), axis=1)[:, :p]
        np.put_along_axis(bits, idx, 1, axis=1)
    u = np.packbits(bits, axis=1, bitorder="little").view(np.uint64).ravel()
    ud = torch.from_numpy(u.view(np.int64)).to(d)
    for _ in range(2):
        L.check(L.lib.bmf_cover_count(L.ptr(X), m_pad, ldx, ldx, L.ptr(ud), L.ptr(vcol), ldx, kp, L.ptr(counts), None, s))
    torch.cuda.synchronize()
    if os.environ.get("COLD"):   # every launch after 512 MiB of other traffic: X comes from HBM, as in the iteration loop
        flush = torch.zeros(512 << 20, dtype=torch.uint8, device=d)
        ts = []
        for _ in range(10):
            flush_sum = flush.view(torch.int32).sum()   # a READ of 512 MiB: evicts X without leaving dirty lines behind
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            L.check(L.lib.bmf_cover_count(L.ptr(X), m_pad, ldx, ldx, L.ptr(ud), L.ptr(vcol), ldx, kp, L.ptr(counts), None, s))
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        print(f"popc={p:2d}: {sorted(ts)[len(ts) // 2]:8.1f} us (cold)")
        continue
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        L.check(L.lib.bmf_cover_count(L.ptr(X), m_pad, ldx, ldx, L.ptr(ud), L.ptr(vcol), ldx, kp, L.ptr(counts), None, s))
    e1.record()
    torch.cuda.synchronize()
    print(f"popc={p:2d}: {e0.elapsed_time(e1) / 10 * 1e3:8.1f} us")
