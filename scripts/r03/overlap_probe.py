"""Does the cover count overlap with the U epilogue when the two are put on two streams?  Both stream ~0.25 GB and each reaches
about half of the HBM peak alone; the epilogue's second round (784 blocks on 512 slots) leaves half the chip idle.  Measurement aid.
usage: overlap_probe.py [rows=100352] [n=20000]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pybmf_amd import _lib as L

rows_pad = int(sys.argv[1]) if len(sys.argv) > 1 else 100352
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
kp, splits = 64, 2
words = (n + 127) // 128 * 4
d = torch.device("cuda:0")
s1 = torch.cuda.current_stream()
s2 = torch.cuda.Stream()
p1, p2 = C.c_void_p(s1.cuda_stream), C.c_void_p(s2.cuda_stream)
g = torch.Generator(device=d).manual_seed(1)
F64 = torch.rand((rows_pad, kp), dtype=torch.float64, device=d, generator=g) * 0.5
F = F64.float()
num = torch.rand((splits, rows_pad, kp), dtype=torch.float32, device=d, generator=g) * 3
G = torch.rand((kp, kp), dtype=torch.float32, device=d, generator=g) * 50
G = (G + G.t()).contiguous()
rowbits_new = torch.zeros(rows_pad, dtype=torch.int64, device=d)
colbits_new = torch.zeros((kp, rows_pad // 32), dtype=torch.int32, device=d)
partials = torch.zeros((rows_pad // 128, 2), dtype=torch.float64, device=d)
blockmax = torch.zeros((rows_pad // 128, kp), dtype=torch.float32, device=d)
planes = torch.zeros((3, kp, rows_pad), dtype=torch.int8, device=d)
scale = torch.zeros(2 * kp, dtype=torch.float32, device=d)
ws = torch.zeros(rows_pad // 128 * kp, dtype=torch.float32, device=d)
L.check(L.lib.bmf_make_panel_i8(L.ptr(F64), L.ptr(F), rows_pad, kp, kp, 3, L.ptr(planes), rows_pad, L.ptr(ws), L.ptr(scale), p1))

# cover inputs: X at ~10 % density, four factor bits per row, V columns at ~6 %
Xbits = torch.randint(-2**31, 2**31 - 1, (rows_pad, words), dtype=torch.int32, device=d, generator=g)
Xbits &= torch.randint(-2**31, 2**31 - 1, (rows_pad, words), dtype=torch.int32, device=d, generator=g)
Xbits &= torch.randint(-2**31, 2**31 - 1, (rows_pad, words), dtype=torch.int32, device=d, generator=g)
rowbits = torch.zeros(rows_pad, dtype=torch.int64, device=d)
for _ in range(4):
    rowbits |= torch.ones(rows_pad, dtype=torch.int64, device=d) << torch.randint(0, 63, (rows_pad,), device=d, generator=g)
colbits = torch.randint(-2**31, 2**31 - 1, (kp, words), dtype=torch.int32, device=d, generator=g)
for _ in range(3):
    colbits &= torch.randint(-2**31, 2**31 - 1, (kp, words), dtype=torch.int32, device=d, generator=g)
counts = torch.zeros(4, dtype=torch.int64, device=d)

a = L.EpilogueArgs()
a.F64, a.F, a.rows_pad, a.rows, a.k, a.kp = F64.data_ptr(), F.data_ptr(), rows_pad, rows_pad - 3, kp, kp
a.num, a.slab_stride, a.splits = num.data_ptr(), rows_pad * kp, splits
a.G, a.reg, a.mode, a.thr, a.terms = G.data_ptr(), 1.0, 1, 0.5, 0
a.panel, a.ldp, a.rowbits, a.colbits, a.ldcb = 0, rows_pad, rowbits_new.data_ptr(), colbits_new.data_ptr(), rows_pad // 32
a.partials, a.stop, a.blockmax = partials.data_ptr(), 0, blockmax.data_ptr()
a.planes, a.plane_scale, a.limbs = planes.data_ptr(), scale.data_ptr(), 3


def epi(st):
    L.check(L.lib.bmf_mu_epilogue(C.byref(a), st))


def cover(st):
    L.check(L.lib.bmf_cover_count(L.ptr(Xbits), rows_pad, words, words, L.ptr(rowbits), L.ptr(colbits), words, kp, L.ptr(counts), None, st))


def timeit(fn, reps=9, inner=40):
    # `inner` repetitions inside one timed region, behind a long filler kernel so that the host has enqueued them all
    # before the first one starts (a Python launch costs more than these kernels run)
    filler = torch.empty(1 << 28, dtype=torch.float32, device=d)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        for _ in range(4):
            filler.add_(1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s1)
        for _ in range(inner):
            fn()
        e1.record(s1)
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / inner)
    ts.sort()
    return ts[len(ts) // 2], ts[0]


def seq(e0=None):
    epi(p1)
    cover(p1)


def only_epi(e0=None):
    epi(p1)


def only_cover(e0=None):
    cover(p1)


def two(first):
    def run(e0=None):
        if e0 is None:
            e0 = torch.cuda.Event()
            e0.record(s1)
        s2.wait_event(e0)
        if first == "epi":
            epi(p1)
            cover(p2)
        else:
            cover(p2)
            epi(p1)
        e2 = torch.cuda.Event()
        e2.record(s2)
        s1.wait_event(e2)
    return run


for name, fn in (("epilogue alone", only_epi), ("cover alone", only_cover), ("one stream, back to back", seq),
                 ("two streams, epilogue launched first", two("epi")), ("two streams, cover launched first", two("cover"))):
    med, mn = timeit(fn)
    print(f"{name:40s} median {med:6.1f} us   min {mn:6.1f} us", flush=True)

# do two streams overlap at all?  a plain streaming kernel (torch add_ over 64 MB) beside the cover count / the epilogue
small = torch.empty(1 << 25, dtype=torch.float32, device=d)


def add_alone(e0=None):
    small.add_(1.0)


def pair(kernel):
    def run(e0=None):
        e = torch.cuda.Event()
        e.record(s1)
        s2.wait_event(e)
        kernel(p2)
        small.add_(1.0)
        e2 = torch.cuda.Event()
        e2.record(s2)
        s1.wait_event(e2)
    return run


def seq_pair(kernel):
    def run(e0=None):
        kernel(p1)
        small.add_(1.0)
    return run


for name, fn in (("add_ (256 MB of traffic) alone", add_alone), ("cover then add_, one stream", seq_pair(cover)),
                 ("cover || add_, two streams", pair(cover)), ("epilogue then add_, one stream", seq_pair(epi)),
                 ("epilogue || add_, two streams", pair(epi))):
    med, mn = timeit(fn)
    print(f"{name:40s} median {med:6.1f} us   min {mn:6.1f} us", flush=True)

# the MAE pass is bound by vector / matrix issue, the epilogue and the cover count by HBM: do THEY overlap?
n_pad = (n + 255) // 256 * 256
XT = torch.randint(-2**31, 2**31 - 1, (n_pad, rows_pad // 32), dtype=torch.int32, device=d, generator=g)
XT &= torch.randint(-2**31, 2**31 - 1, (n_pad, rows_pad // 32), dtype=torch.int32, device=d, generator=g)
Uf = torch.rand((rows_pad, kp), dtype=torch.float32, device=d, generator=g) * 0.3
Vf = torch.rand((n_pad, kp), dtype=torch.float32, device=d, generator=g) * 0.3
wsm = torch.zeros(2 * (rows_pad + n_pad) * kp, dtype=torch.int16, device=d)
msum = torch.zeros(2, dtype=torch.float64, device=d)


def mae(st):
    L.check(L.lib.bmf_mae_sum_ex(L.ptr(XT), rows_pad // 32, rows_pad, n_pad, L.ptr(Uf), L.ptr(Vf), kp, L.ptr(wsm), L.ptr(msum), 1, st))


def both(st):
    epi(st)
    cover(st)


def seq2(k1, k2):
    def run(e0=None):
        k1(p1)
        k2(p1)
    return run


def par2(k_side, k_main):
    def run(e0=None):
        e = torch.cuda.Event()
        e.record(s1)
        s2.wait_event(e)
        k_side(p2)
        k_main(p1)
        e2 = torch.cuda.Event()
        e2.record(s2)
        s1.wait_event(e2)
    return run


for name, fn in (("MAE alone", seq2(mae, lambda st: None)), ("MAE then epilogue, one stream", seq2(mae, epi)),
                 ("MAE || epilogue", par2(mae, epi)), ("MAE then cover, one stream", seq2(mae, cover)),
                 ("MAE || cover", par2(mae, cover)), ("MAE then epilogue + cover, one stream", seq2(mae, both)),
                 ("MAE || epilogue + cover", par2(mae, both)), ("epilogue + cover || MAE (MAE on the main stream)", par2(both, mae))):
    med, mn = timeit(fn, inner=20)
    print(f"{name:50s} median {med:6.1f} us   min {mn:6.1f} us", flush=True)
