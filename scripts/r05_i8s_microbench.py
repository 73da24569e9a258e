"""Round 5: the 2:4-sparse bits GEMM (csrc/xf_bits_i8s.hip) beside the dense one at the headline shape -- exactness and time.

For both orientations (X V, X^T U):
  * overflow census of the bench's planted matrix (bmf_s24_count): share of cells, concentration in the densest rows;
  * S24 pack of ALL rows, sparse kernel vs the dense kernel on the same matrix with the overflow ones cleared (bitwise: same plan,
    same slices, both exact), sparse + overflow pass vs the dense kernel on the original matrix (sum of slabs);
  * times: dense, sparse (all rows), overflow pass; and the split form -- the rows with the most overflow to the dense kernel,
    the rest to the sparse one -- for a few dense shares.
Measurement aid, not part of the product.  usage: r05_i8s_microbench.py [launches=30]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pybmf_amd import _lib as L
from pybmf_amd.engine import BitMatrix, xf_slots_i8
from pybmf_amd.generators import PlantedBooleanOnDevice

n_launch = int(sys.argv[1]) if len(sys.argv) > 1 else 30
m, n, k = (int(v) for v in os.environ.get("SHAPE", "100000,20000,64").split(","))
shares = [float(v) for v in os.environ.get("DENSE_SHARES", "0,0.1,0.2,0.3").split(",")]
kp = 32 if k <= 32 else 64
dev = torch.device("cuda:0")
dens = float(os.environ.get("DENSITY", "0.067"))
X = BitMatrix(PlantedBooleanOnDevice(m, n, k, density=(dens, dens), seed=1000, noise=(0.05, 0.01), noise_seed=2000, device=dev), dev)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
FORM = int(os.environ.get("FORM", "0"))   # 0: four waves / 256-row tiles; 1: eight waves / 512-row tiles, DMA roles split
L.check(min(0, L.lib.bmf_xf_bits_i8s_form(FORM)))
TILE = 512 if FORM else 256
print(f"form {FORM}; shape {m} x {n}, k = {k}; ones {X.sum_local} = {X.sum_local / (m * n):.4f} of the cells; sparse kernel occupancy {L.lib.bmf_xf_bits_i8s_occupancy()} WG/CU")


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


def round_up(v, q):
    return (v + q - 1) // q * q


def pack(bits, rows_sel, red_words, want_kept):
    """S24 of the selected rows (rows_sel: int32 tensor of source rows, or None = all rows of `bits`) + the overflow CSR"""
    nsel = bits.shape[0] if rows_sel is None else int(rows_sel.numel())
    rows_pad_s = round_up(max(nsel, 1), TILE)
    rowsel = None
    if rows_sel is not None:
        rowsel = torch.full((rows_pad_s,), -1, dtype=torch.int32, device=dev)
        rowsel[:nsel] = rows_sel
    counts = torch.zeros(bits.shape[0], dtype=torch.int32, device=dev)
    L.check(L.lib.bmf_s24_count(L.ptr(bits), bits.shape[0], bits.shape[1], red_words, L.ptr(counts), st))
    pc = torch.zeros(rows_pad_s, dtype=torch.int64, device=dev)
    pc[:nsel] = (counts if rows_sel is None else counts[rows_sel.long()]).long()
    ovf_ptr = torch.zeros(rows_pad_s + 1, dtype=torch.int64, device=dev)
    ovf_ptr[1:] = torch.cumsum(pc, 0)
    n_ovf = int(ovf_ptr[-1].item())
    ovf_idx = torch.zeros(max(n_ovf, 1), dtype=torch.int32, device=dev)
    cursor = torch.zeros(rows_pad_s, dtype=torch.int32, device=dev)
    s24 = torch.empty(L.lib.bmf_s24_bytes(rows_pad_s, red_words) // 4, dtype=torch.int32, device=dev)
    kept = torch.zeros((rows_pad_s, bits.shape[1]), dtype=torch.int32, device=dev) if want_kept else None
    L.check(L.lib.bmf_s24_pack(L.ptr(bits), bits.shape[1], red_words, L.ptr(rowsel), rows_pad_s, L.ptr(s24), L.ptr(ovf_ptr), L.ptr(cursor),
                               L.ptr(ovf_idx), L.ptr(kept), bits.shape[1], st))
    torch.cuda.synchronize()
    assert torch.equal(cursor.long(), pc), "the packer's overflow cursors must end at the rows' counts"
    return dict(s24=s24, rows_pad_s=rows_pad_s, rowsel=rowsel, ovf_ptr=ovf_ptr, ovf_idx=ovf_idx, n_ovf=n_ovf, kept=kept, counts=counts, nsel=nsel)


for name, bits, rows, rows_pad, ldw, red_pad in (("XV", X.bits, X.m, X.m_pad, X.ldx, X.n_pad), ("XtU", X.bits_t, X.n, X.n_pad, X.ldxt, X.m_pad)):
    red_words = red_pad // 32
    g = torch.Generator(device=dev)
    g.manual_seed(11)
    F64 = torch.rand((red_pad, kp), dtype=torch.float64, device=dev, generator=g)
    F64[:, 3] *= 1e-3
    F32 = F64.float()
    panel = torch.zeros((3, kp, red_pad), dtype=torch.int8, device=dev)
    scale = torch.zeros(2 * kp, dtype=torch.float32, device=dev)
    ws = torch.zeros(red_pad // 128 * kp, dtype=torch.float32, device=dev)
    L.check(L.lib.bmf_make_panel_i8(L.ptr(F64), L.ptr(F32), red_pad, kp, kp, 3, L.ptr(panel), red_pad, L.ptr(ws), L.ptr(scale), st))
    colscale = scale[kp:]

    if os.environ.get("MODE") == "time":   # kernel flavours (ablations: wrong results on purpose): the sparse kernel's time only
        s24 = torch.empty(L.lib.bmf_s24_bytes(rows_pad, red_words) // 4, dtype=torch.int32, device=dev)
        L.check(L.lib.bmf_s24_pack(L.ptr(bits), ldw, red_words, None, rows_pad, L.ptr(s24), None, None, None, None, 0, st))
        splits = L.lib.bmf_xf_bits_i8s_slots(rows_pad, red_words, kp)
        out_s = torch.zeros((splits, rows_pad, kp), dtype=torch.float32, device=dev)
        t = timed(lambda: L.check(L.lib.bmf_xf_bits_i8s(L.ptr(s24), rows_pad, red_words, L.ptr(panel), red_pad, L.ptr(colscale), kp, L.ptr(out_s), rows_pad * kp, splits, None, st)))
        line = f"[{name}] {os.environ.get('BMF_LIB', 'libbmf_hip.so')}: sparse kernel, all rows {t:.1f} us"
        if os.environ.get("WITH_DENSE") == "1":   # the dense kernel on the same matrix beside it (PMC passes compare the two)
            tiled = torch.empty_like(bits)
            L.check(L.lib.bmf_tile_bits(L.ptr(bits), rows_pad, ldw, red_words, L.ptr(tiled), st))
            sp_d = xf_slots_i8(rows_pad, red_pad, kp)
            out_d = torch.zeros((sp_d, rows_pad, kp), dtype=torch.float32, device=dev)
            td = timed(lambda: L.check(L.lib.bmf_xf_bits_i8(L.ptr(tiled), rows_pad, ldw, red_words, L.ptr(panel), red_pad, 3, L.ptr(colscale), kp, L.ptr(out_d), rows_pad * kp, sp_d, 1, st)))
            line += f" | dense kernel {td:.1f} us"
        print(line)
        continue
    # ---- census ----
    P = pack(bits, None, red_words, True)
    cnt = P["counts"][:rows].long()
    order = torch.argsort(cnt, descending=True)
    cs = torch.cumsum(cnt[order], 0).double() / max(int(cnt.sum().item()), 1)
    print(f"[{name}] overflow ones {P['n_ovf']} = {P['n_ovf'] / (rows * (red_pad)):.5f} of the cells; per row mean {cnt.double().mean().item():.1f} max {int(cnt.max().item())}; "
          f"share held by the densest 5 / 10 / 20 / 30 % of the rows: " + " / ".join(f"{cs[int(rows * f) - 1].item():.2f}" for f in (0.05, 0.1, 0.2, 0.3)))

    # ---- exactness: all rows ----
    splits = max(xf_slots_i8(rows_pad, red_pad, kp), L.lib.bmf_xf_bits_i8s_slots(rows_pad, red_words, kp))
    tiled = torch.empty_like(bits)
    L.check(L.lib.bmf_tile_bits(L.ptr(bits), rows_pad, ldw, red_words, L.ptr(tiled), st))
    out_d = torch.zeros((splits, rows_pad, kp), dtype=torch.float32, device=dev)
    dense_args = (L.ptr(tiled), rows_pad, ldw, red_words, L.ptr(panel), red_pad, 3, L.ptr(colscale), kp, L.ptr(out_d), rows_pad * kp, splits, 1, st)
    L.check(L.lib.bmf_xf_bits_i8(*dense_args))
    kept_tiled = torch.empty_like(bits)
    L.check(L.lib.bmf_tile_bits(L.ptr(P["kept"]), rows_pad, ldw, red_words, L.ptr(kept_tiled), st))
    out_k = torch.zeros_like(out_d)
    L.check(L.lib.bmf_xf_bits_i8(L.ptr(kept_tiled), rows_pad, ldw, red_words, L.ptr(panel), red_pad, 3, L.ptr(colscale), kp, L.ptr(out_k), rows_pad * kp, splits, 1, st))
    out_s = torch.full_like(out_d, 7.0)
    sparse_args = (L.ptr(P["s24"]), rows_pad, red_words, L.ptr(panel), red_pad, L.ptr(colscale), kp, L.ptr(out_s), rows_pad * kp, splits, None, st)
    L.check(L.lib.bmf_xf_bits_i8s(*sparse_args))
    torch.cuda.synchronize()
    same = torch.equal(out_s, out_k)
    if FORM:   # other row tiles, other slices: the slabs are cut differently -- compare their sums (each slab is an exact sum rounded once)
        sk, ss = out_k.double().sum(0), out_s.double().sum(0)
        rel_k = float(((ss - sk).abs().max() / sk.abs().max()).item())
        same = rel_k < 3e-7
        print(f"[{name}] sparse kernel (form 1) vs dense kernel on the matrix without its overflow ones, sum of slabs: max |diff| / max = {rel_k:.3e}")
    else:
        print(f"[{name}] sparse kernel vs dense kernel on the matrix without its overflow ones: {'BITWISE EQUAL' if same else 'DIFFERENT'}"
              f" (max |diff| {float((out_s - out_k).abs().max().item()):.3e})")
    ovf_args = (L.ptr(P["ovf_ptr"]), L.ptr(P["ovf_idx"]), None, rows_pad, L.ptr(F64), kp, L.ptr(colscale), kp, L.ptr(out_s), st)
    L.check(L.lib.bmf_s24_overflow(*ovf_args))
    torch.cuda.synchronize()
    full_s, full_d = out_s.double().sum(0), out_d.double().sum(0)
    rel = float(((full_s - full_d).abs().max() / full_d.abs().max()).item())
    print(f"[{name}] sparse + overflow pass vs dense kernel on the whole matrix: max |diff| / max = {rel:.3e}")
    assert same and rel < 1e-6

    # ---- times ----
    t_dense = timed(lambda: L.check(L.lib.bmf_xf_bits_i8(*dense_args)))
    t_sparse = timed(lambda: L.check(L.lib.bmf_xf_bits_i8s(*sparse_args)))
    t_ovf = timed(lambda: L.check(L.lib.bmf_s24_overflow(*ovf_args)))
    print(f"[{name}] dense kernel {t_dense:.1f} us | sparse kernel, all rows {t_sparse:.1f} us ({t_dense / t_sparse:.2f} x) | overflow pass, all rows "
          f"({P['n_ovf']} ones) {t_ovf:.1f} us")
    del out_k, kept_tiled, P

    # ---- the split form: the rows with the most overflow to the dense kernel ----
    for share in shares:
        if share <= 0:
            continue
        nd = int(rows * share) // 512 * 512
        rows_d = order[:nd].to(torch.int32)
        rows_s = torch.sort(order[nd:])[0].to(torch.int32)
        Ps = pack(bits, rows_s, red_words, False)
        nd_pad = round_up(nd, 512)
        bits_d = torch.zeros((nd_pad, ldw), dtype=torch.int32, device=dev)
        bits_d[:nd] = bits[rows_d.long()]
        tiled_d = torch.empty_like(bits_d)
        L.check(L.lib.bmf_tile_bits(L.ptr(bits_d), nd_pad, ldw, red_words, L.ptr(tiled_d), st))
        sp_d = xf_slots_i8(nd_pad, red_pad, kp)
        out_dd = torch.zeros((sp_d, nd_pad, kp), dtype=torch.float32, device=dev)
        a_d = (L.ptr(tiled_d), nd_pad, ldw, red_words, L.ptr(panel), red_pad, 3, L.ptr(colscale), kp, L.ptr(out_dd), nd_pad * kp, sp_d, 1, st)
        sp_s = max(splits, L.lib.bmf_xf_bits_i8s_slots(Ps["rows_pad_s"], red_words, kp))
        out_ss = torch.zeros((sp_s, rows_pad, kp), dtype=torch.float32, device=dev)
        a_s = (L.ptr(Ps["s24"]), Ps["rows_pad_s"], red_words, L.ptr(panel), red_pad, L.ptr(colscale), kp, L.ptr(out_ss), rows_pad * kp, sp_s, L.ptr(Ps["rowsel"]), st)
        a_o = (L.ptr(Ps["ovf_ptr"]), L.ptr(Ps["ovf_idx"]), L.ptr(Ps["rowsel"]), Ps["rows_pad_s"], L.ptr(F64), kp, L.ptr(colscale), kp, L.ptr(out_ss), st)
        L.check(L.lib.bmf_xf_bits_i8(*a_d))
        L.check(L.lib.bmf_xf_bits_i8s(*a_s))
        L.check(L.lib.bmf_s24_overflow(*a_o))
        torch.cuda.synchronize()
        full = out_ss.double().sum(0)
        full[rows_d.long()] = out_dd.double().sum(0)[:nd]
        rel = float(((full - full_d).abs().max() / full_d.abs().max()).item())
        td, ts, to = (timed(lambda: L.check(L.lib.bmf_xf_bits_i8(*a_d))), timed(lambda: L.check(L.lib.bmf_xf_bits_i8s(*a_s))),
                      timed(lambda: L.check(L.lib.bmf_s24_overflow(*a_o))))

        def all3():
            L.check(L.lib.bmf_xf_bits_i8(*a_d))
            L.check(L.lib.bmf_xf_bits_i8s(*a_s))
            L.check(L.lib.bmf_s24_overflow(*a_o))
        t3 = timed(all3)
        print(f"[{name}] dense share {share:.2f}: {nd} rows dense {td:.1f} us + {Ps['nsel']} rows sparse {ts:.1f} us + overflow ({Ps['n_ovf']} ones) {to:.1f} us"
              f" = {td + ts + to:.1f} us; the three launches back to back {t3:.1f} us (dense kernel alone {t_dense:.1f}); max |diff| / max vs dense {rel:.2e}")
        assert rel < 1e-6
        del Ps, bits_d, tiled_d, out_dd, out_ss
