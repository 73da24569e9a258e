"""The 2:4-sparse bits GEMM (csrc/xf_bits_i8s.hip) against the dense int8 bits GEMM and the host encoder.

  * bmf_s24_pack on the device == the C encoder of csrc/s24.h compiled for the host (tests/test_s24_format.py checks that one against
    a model of the instruction), bit for bit; the overflow list = exactly the ones the form drops;
  * bmf_xf_bits_i8s on the S24 form == bmf_xf_bits_i8 on the bit matrix without its overflow ones, slab for slab (both exact);
  * + bmf_s24_overflow == the dense kernel on the whole matrix (sum of slabs, one fp32 rounding apart), also through a row selection
    with the remaining rows on the dense kernel, and for a 50 %-dense matrix where nearly every group overflows.
Replaces multiply(W, X) @ V / multiply(W, X).T @ U (PyBMF/models/BinaryMFPenalty.py:139,154)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rup(v, q):
    return (v + q - 1) // q * q


@pytest.fixture(scope="module")
def shim(tmp_path_factory):
    so = tmp_path_factory.mktemp("s24") / "s24_shim.so"
    subprocess.run(["gcc", "-O2", "-shared", "-fPIC", os.path.join(ROOT, "tests", "csrc", "s24_shim.c"), "-o", str(so)], check=True)
    lib = C.CDLL(str(so))
    lib.s24_encode_pair.restype = C.c_uint
    return lib


def host_pack(shim, bits_np):
    """S24 of a host bit matrix (rows % 256 == 0, words % 16 == 0) with the shared C encoder; returns (s24 words, kept, per-row extra)"""
    rows, words = bits_np.shape
    groups = words // 16
    s24 = np.zeros((rows // 256, groups, 24576 // 4), np.uint32)
    kept = np.zeros_like(bits_np)
    extra = np.zeros(rows, np.int64)
    idx = np.zeros((2, 4), np.uint32)
    val = np.zeros((2, 2), np.uint32)
    k2 = np.zeros((2, 4), np.uint32)
    for row in range(rows):
        tile, r = divmod(row, 256)
        for grp in range(groups):
            blk = s24[tile, grp]
            for h in range(2):
                w = np.ascontiguousarray(bits_np[row, 16 * grp + 8 * h: 16 * grp + 8 * h + 8].reshape(2, 4))
                extra[row] += shim.s24_encode_pair(w.ctypes.data_as(C.c_void_p), idx.ctypes.data_as(C.c_void_p), val.ctypes.data_as(C.c_void_p),
                                                   k2.ctypes.data_as(C.c_void_p))
                for ai in range(2):
                    a = h + 2 * ai
                    blk[(r * 64 + a * 16) // 4: (r * 64 + a * 16) // 4 + 4] = idx[ai]
                    blk[(16384 + r * 32 + a * 8) // 4: (16384 + r * 32 + a * 8) // 4 + 2] = val[ai]
                kept[row, 16 * grp + 8 * h: 16 * grp + 8 * h + 8] = k2.reshape(8)
    return s24.reshape(-1), kept, extra


def random_bits(rows, cols, rows_pad, cols_pad, density, seed, row_scale=None):
    rs = np.random.RandomState(seed)
    p = np.full((rows, 1), density)
    if row_scale is not None:
        p = p * row_scale.reshape(-1, 1)
    x = (rs.rand(rows, cols) < p).astype(np.uint8)
    full = np.zeros((rows_pad, cols_pad), np.uint8)
    full[:rows, :cols] = x
    return np.packbits(full, axis=1, bitorder="little").view(np.uint32)


def device_pack(L, bits, rowsel_src, red_words, want_kept=False):
    dev = bits.device
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    nsel = bits.shape[0] if rowsel_src is None else int(rowsel_src.numel())
    rows_pad_s = rup(max(nsel, 1), 256)
    rowsel = None
    if rowsel_src is not None:
        rowsel = torch.full((rows_pad_s,), -1, dtype=torch.int32, device=dev)
        rowsel[:nsel] = rowsel_src
    counts = torch.zeros(bits.shape[0], dtype=torch.int32, device=dev)
    L.check(L.lib.bmf_s24_count(L.ptr(bits), bits.shape[0], bits.shape[1], red_words, L.ptr(counts), st))
    pc = torch.zeros(rows_pad_s, dtype=torch.int64, device=dev)
    pc[:nsel] = (counts if rowsel_src is None else counts[rowsel_src.long()]).long()
    ovf_ptr = torch.zeros(rows_pad_s + 1, dtype=torch.int64, device=dev)
    ovf_ptr[1:] = torch.cumsum(pc, 0)
    n_ovf = int(ovf_ptr[-1].item())
    ovf_idx = torch.full((max(n_ovf, 1),), -1, dtype=torch.int32, device=dev)
    cursor = torch.zeros(rows_pad_s, dtype=torch.int32, device=dev)
    nbytes = L.lib.bmf_s24_bytes(rows_pad_s, red_words)
    assert nbytes == rows_pad_s // 256 * (red_words // 16) * 24576
    s24 = torch.zeros(nbytes // 4, dtype=torch.int32, device=dev)
    kept = torch.zeros((rows_pad_s, bits.shape[1]), dtype=torch.int32, device=dev) if want_kept else None
    L.check(L.lib.bmf_s24_pack(L.ptr(bits), bits.shape[1], red_words, L.ptr(rowsel), rows_pad_s, L.ptr(s24), L.ptr(ovf_ptr), L.ptr(cursor),
                               L.ptr(ovf_idx), L.ptr(kept), bits.shape[1], st))
    torch.cuda.synchronize()
    assert torch.equal(cursor.long(), pc)
    return dict(s24=s24, rows_pad_s=rows_pad_s, rowsel=rowsel, ovf_ptr=ovf_ptr, ovf_idx=ovf_idx, n_ovf=n_ovf, kept=kept, counts=counts, nsel=nsel)


def make_planes(L, red_pad, kp, seed):
    dev = torch.device("cuda:0")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    F64 = torch.rand((red_pad, kp), dtype=torch.float64, device=dev, generator=g)
    F64[:, 1] *= 1e-4
    F64[:, 2] = 0.0
    F32 = F64.float()
    panel = torch.zeros((3, kp, red_pad), dtype=torch.int8, device=dev)
    scale = torch.zeros(2 * kp, dtype=torch.float32, device=dev)
    ws = torch.zeros(red_pad // 128 * kp, dtype=torch.float32, device=dev)
    L.check(L.lib.bmf_make_panel_i8(L.ptr(F64), L.ptr(F32), red_pad, kp, kp, 3, L.ptr(panel), red_pad, L.ptr(ws), L.ptr(scale), st))
    return F64, panel, scale[kp:]


def test_device_packer_is_the_host_encoder(shim):
    from pybmf_amd import _lib as L
    rows, cols = 300, 1000
    rows_pad, cols_pad = 512, 1024
    bits_np = random_bits(rows, cols, rows_pad, cols_pad, 0.25, seed=3)
    bits = torch.from_numpy(bits_np.view(np.int32)).cuda()
    P = device_pack(L, bits, None, cols_pad // 32, want_kept=True)
    s24_h, kept_h, extra_h = host_pack(shim, bits_np)
    assert np.array_equal(P["s24"].cpu().numpy().view(np.uint32), s24_h)
    assert np.array_equal(P["kept"].cpu().numpy().view(np.uint32), kept_h)
    assert np.array_equal(P["counts"].cpu().numpy().astype(np.int64), extra_h) and extra_h.sum() > 1000
    # the overflow list: exactly the ones the form drops, as reduction indices (cl = 128 g + 32 t + bit inside a 512-block)
    ptr, idx = P["ovf_ptr"].cpu().numpy(), P["ovf_idx"].cpu().numpy()
    dropped = bits_np & ~kept_h
    for row in (0, 1, 17, 255, 256, 299):
        want = []
        for wi in np.nonzero(dropped[row])[0]:
            blk, w = divmod(int(wi), 16)
            g, t = divmod(w, 4)
            want += [512 * blk + 128 * g + 32 * t + b for b in range(32) if (int(dropped[row, wi]) >> b) & 1]
        assert sorted(idx[ptr[row]:ptr[row + 1]].tolist()) == sorted(want)
    # the same reduction index means the same cell of the plain layout: word 16 blk + 4 g + t, bit
    assert ptr[rows_pad] == extra_h.sum()


@pytest.mark.parametrize("rows,red,kp,density", [(700, 2000, 64, 0.08), (256, 512, 32, 0.3), (1300, 5000, 64, 0.5), (513, 1500, 64, 0.02), (300, 900, 32, 1.0)])
def test_sparse_kernel_and_overflow_pass_are_exact(rows, red, kp, density):
    from pybmf_amd import _lib as L
    from pybmf_amd.engine import xf_slots_i8
    dev = torch.device("cuda:0")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    rows_pad, red_pad = rup(rows, 512), rup(red, 512)
    red_words = red_pad // 32
    bits = torch.from_numpy(random_bits(rows, red, rows_pad, red_pad, density, seed=rows).view(np.int32)).to(dev)
    F64, panel, colscale = make_planes(L, red_pad, kp, seed=red)
    P = device_pack(L, bits, None, red_words, want_kept=True)
    splits = max(xf_slots_i8(rows_pad, red_pad, kp), L.lib.bmf_xf_bits_i8s_slots(rows_pad, red_words, kp))

    def dense(b):
        out = torch.full((splits, rows_pad, kp), 3.0, dtype=torch.float32, device=dev)
        L.check(L.lib.bmf_xf_bits_i8(L.ptr(b), rows_pad, b.shape[1], red_words, L.ptr(panel), red_pad, 3, L.ptr(colscale), kp, L.ptr(out),
                                     rows_pad * kp, splits, 0, st))
        return out
    out_d, out_k = dense(bits), dense(P["kept"])
    out_s = torch.full((splits, rows_pad, kp), 5.0, dtype=torch.float32, device=dev)
    L.check(L.lib.bmf_xf_bits_i8s(L.ptr(P["s24"]), rows_pad, red_words, L.ptr(panel), red_pad, L.ptr(colscale), kp, L.ptr(out_s), rows_pad * kp, splits,
                                  None, st))
    torch.cuda.synchronize()
    assert torch.equal(out_s, out_k)                      # same plan, same slices, both exact: slab for slab
    L.check(L.lib.bmf_s24_overflow(L.ptr(P["ovf_ptr"]), L.ptr(P["ovf_idx"]), None, rows_pad, L.ptr(F64), kp, L.ptr(colscale), kp, L.ptr(out_s), st))
    torch.cuda.synchronize()
    got, want = out_s.double().sum(0), out_d.double().sum(0)
    # exact integer product of X with the quantised factor, recomputed on the host in int64
    q = torch.round(torch.clamp(F64 / colscale.double(), -8355711.0, 8355711.0)).to(torch.int64).cpu().numpy()
    xb = np.unpackbits(bits.cpu().numpy().view(np.uint8), axis=1, bitorder="little")[:, :red_pad]
    # plain bit column c = 32 * word + bit holds reduction index 512 blk + 128 g + 32 t + bit with word = 16 blk + 4 g + t: the same number
    exact = (xb[:rows].astype(np.int64) @ q) * colscale.double().cpu().numpy()
    np.testing.assert_allclose(want.cpu().numpy()[:rows], exact, rtol=3e-7, atol=0)
    np.testing.assert_allclose(got.cpu().numpy()[:rows], exact, rtol=3e-7, atol=0)
    assert float(got[rows:].abs().max()) == 0.0 if rows < rows_pad else True
    if density >= 0.5:
        assert P["n_ovf"] > 0.05 * rows * red        # nearly every group overflows: the form still adds up


@pytest.mark.parametrize("form", [1, 2])
@pytest.mark.parametrize("rows,red,kp,density", [(1300, 5000, 64, 0.08), (512, 512, 32, 0.3), (700, 2000, 64, 1.0)])
def test_eight_wave_forms_are_exact(rows, red, kp, density, form):
    """bmf_xf_bits_i8s_form(1): one workgroup of eight waves per CU, 512-row tiles, plane and S24 pieces fetched by different waves, the S24
    words double-buffered by group; form 2: the same with the two wave groups in anti-phase (one issues only matrix instructions while the
    other loads).  Other tiles and slices than the dense kernel's, so the SUM of the slabs is compared: with the dense kernel on the kept
    bits, and with the exact integer product."""
    from pybmf_amd import _lib as L
    from pybmf_amd.engine import xf_slots_i8
    dev = torch.device("cuda:0")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    rows_pad, red_pad = rup(rows, 512), rup(red, 512)
    red_words = red_pad // 32
    bits = torch.from_numpy(random_bits(rows, red, rows_pad, red_pad, density, seed=rows + 1).view(np.int32)).to(dev)
    F64, panel, colscale = make_planes(L, red_pad, kp, seed=red + 1)
    P = device_pack(L, bits, None, red_words, want_kept=True)
    assert L.lib.bmf_xf_bits_i8s_form(form) == 0     # (returns the form before)
    try:
        assert L.lib.bmf_xf_bits_i8s_occupancy() == 1
        splits = L.lib.bmf_xf_bits_i8s_slots(rows_pad, red_words, kp)
        out_s = torch.full((splits, rows_pad, kp), 5.0, dtype=torch.float32, device=dev)
        L.check(L.lib.bmf_xf_bits_i8s(L.ptr(P["s24"]), rows_pad, red_words, L.ptr(panel), red_pad, L.ptr(colscale), kp, L.ptr(out_s), rows_pad * kp,
                                      splits, None, st))
        # 256-row packed forms that are not whole 512-row tiles are refused in this form
        assert L.lib.bmf_xf_bits_i8s(L.ptr(P["s24"]), rows_pad + 256, red_words, L.ptr(panel), red_pad, L.ptr(colscale), kp, L.ptr(out_s),
                                     rows_pad * kp, splits, None, st) == -1
    finally:
        assert L.lib.bmf_xf_bits_i8s_form(0) == form and L.lib.bmf_xf_bits_i8s_form(-1) == 0
    sp_d = xf_slots_i8(rows_pad, red_pad, kp)
    out_k = torch.zeros((sp_d, rows_pad, kp), dtype=torch.float32, device=dev)
    L.check(L.lib.bmf_xf_bits_i8(L.ptr(P["kept"]), rows_pad, bits.shape[1], red_words, L.ptr(panel), red_pad, 3, L.ptr(colscale), kp, L.ptr(out_k),
                                 rows_pad * kp, sp_d, 0, st))
    torch.cuda.synchronize()
    got, want = out_s.double().sum(0).cpu().numpy(), out_k.double().sum(0).cpu().numpy()
    np.testing.assert_allclose(got, want, rtol=3e-7, atol=0)
    q = torch.round(torch.clamp(F64 / colscale.double(), -8355711.0, 8355711.0)).to(torch.int64).cpu().numpy()
    xk = np.unpackbits(P["kept"].cpu().numpy().view(np.uint8), axis=1, bitorder="little")[:, :red_pad]
    exact = (xk[:rows].astype(np.int64) @ q) * colscale.double().cpu().numpy()
    np.testing.assert_allclose(got[:rows], exact, rtol=3e-7, atol=0)
    assert float(np.abs(got[rows:]).max(initial=0.0)) == 0.0


def test_row_selection_splits_the_work_between_the_two_kernels():
    """The rows with the most overflow ones go to the dense kernel (a compacted bit matrix), the rest to the sparse kernel through a row
    map; together they are the dense kernel on everything."""
    from pybmf_amd import _lib as L
    from pybmf_amd.engine import xf_slots_i8
    dev = torch.device("cuda:0")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    rows, red, kp = 1500, 3000, 64
    rows_pad, red_pad = rup(rows, 512), rup(red, 512)
    red_words = red_pad // 32
    rs = np.random.RandomState(5)
    scale = np.where(rs.rand(rows) < 0.2, 6.0, 1.0)      # a fifth of the rows six times denser
    bits = torch.from_numpy(random_bits(rows, red, rows_pad, red_pad, 0.06, seed=9, row_scale=scale).view(np.int32)).to(dev)
    F64, panel, colscale = make_planes(L, red_pad, kp, seed=4)
    counts = torch.zeros(rows_pad, dtype=torch.int32, device=dev)
    L.check(L.lib.bmf_s24_count(L.ptr(bits), rows_pad, bits.shape[1], red_words, L.ptr(counts), st))
    order = torch.argsort(counts[:rows].long(), descending=True)
    nd = 256
    rows_d, rows_s = order[:nd].to(torch.int32), torch.sort(order[nd:])[0].to(torch.int32)
    assert float(counts[rows_d.long()].double().mean()) > 20 * float(counts[rows_s.long()].double().mean() + 1e-9)
    P = device_pack(L, bits, rows_s, red_words)
    splits = max(xf_slots_i8(rows_pad, red_pad, kp), L.lib.bmf_xf_bits_i8s_slots(P["rows_pad_s"], red_words, kp))
    out = torch.full((splits, rows_pad, kp), 9.0, dtype=torch.float32, device=dev)
    L.check(L.lib.bmf_xf_bits_i8s(L.ptr(P["s24"]), P["rows_pad_s"], red_words, L.ptr(panel), red_pad, L.ptr(colscale), kp, L.ptr(out), rows_pad * kp,
                                  splits, L.ptr(P["rowsel"]), st))
    L.check(L.lib.bmf_s24_overflow(L.ptr(P["ovf_ptr"]), L.ptr(P["ovf_idx"]), L.ptr(P["rowsel"]), P["rows_pad_s"], L.ptr(F64), kp, L.ptr(colscale), kp,
                                   L.ptr(out), st))
    nd_pad = rup(nd, 512)
    bits_d = torch.zeros((nd_pad, bits.shape[1]), dtype=torch.int32, device=dev)
    bits_d[:nd] = bits[rows_d.long()]
    sp_d = xf_slots_i8(nd_pad, red_pad, kp)
    out_dd = torch.zeros((sp_d, nd_pad, kp), dtype=torch.float32, device=dev)
    L.check(L.lib.bmf_xf_bits_i8(L.ptr(bits_d), nd_pad, bits.shape[1], red_words, L.ptr(panel), red_pad, 3, L.ptr(colscale), kp, L.ptr(out_dd),
                                 nd_pad * kp, sp_d, 0, st))
    sp_a = xf_slots_i8(rows_pad, red_pad, kp)
    out_all = torch.zeros((sp_a, rows_pad, kp), dtype=torch.float32, device=dev)
    L.check(L.lib.bmf_xf_bits_i8(L.ptr(bits), rows_pad, bits.shape[1], red_words, L.ptr(panel), red_pad, 3, L.ptr(colscale), kp, L.ptr(out_all),
                                 rows_pad * kp, sp_a, 0, st))
    torch.cuda.synchronize()
    # rows the selection does not cover were not touched (still the fill value in every slot)
    assert bool((out[:, rows_d.long()] == 9.0).all()) and bool((out[:, rows:] == 9.0).all())
    full = out.double().sum(0)
    full[rows_d.long()] = out_dd.double().sum(0)[:nd]
    want = out_all.double().sum(0)
    np.testing.assert_allclose(full[:rows].cpu().numpy(), want[:rows].cpu().numpy(), rtol=3e-7, atol=0)


def test_bad_arguments_are_refused():
    from pybmf_amd import _lib as L
    lib = L.lib
    assert lib.bmf_s24_bytes(255, 16) == -1 and lib.bmf_s24_bytes(256, 15) == -1
    assert lib.bmf_xf_bits_i8s_slots(256, 16, 48) < 0
    assert lib.bmf_xf_bits_i8s(None, 256, 16, None, 512, None, 64, None, 256 * 64, 1, None, None) == -1 and b"null pointer" in lib.bmf_last_error()
    assert lib.bmf_s24_pack(None, 16, 16, None, 256, None, None, None, None, None, 0, None) == -1
    assert lib.bmf_xf_bits_i8s_occupancy() == 2
