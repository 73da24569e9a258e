"""Per-kernel parity on the GPU: every entry point of libbmf_hip.so against NumPy / the CPU oracle, called
through the C ABI (ctypes), at small odd shapes."""
import ctypes as C
import json
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

import oracle as orc  # noqa: E402


@pytest.fixture(scope="module")
def env():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from pybmf_amd import _lib as L
    from pybmf_amd import engine as E
    return L, E, torch.device("cuda:0")


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def dev(a, device, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.to(device)


def host_panel_value(panel, rows_pad, kp, terms, pos):
    """Reconstruct F[r][j] = sum_t panel[t][j][128*b + pos[cl]] from a device panel (int16 bits of bf16)."""
    p = panel.cpu().numpy().view(np.uint16).astype(np.uint32) << 16
    p = p.view(np.float32).astype(np.float64)  # [T][kp][ldp]
    r = np.arange(rows_pad)
    idx = (r // 128) * 128 + pos[r % 128]
    return p[:, :, idx].sum(0).T  # rows_pad x kp


@pytest.fixture(scope="module")
def pos(env):
    L, E, d = env
    return np.array([L.lib.bmf_panel_pos(i) for i in range(128)])


def test_panel_pos_is_a_permutation(pos):
    assert sorted(pos.tolist()) == list(range(128))


def test_pack_and_popcount(env):
    L, E, d = env
    rs = np.random.RandomState(0)
    for (m, n) in [(300, 777), (64, 64), (1000, 500), (513, 129)]:
        X = (rs.rand(m, n) < 0.3).astype(np.uint8)
        B = E.BitMatrix(X, d, chunk_rows=128)
        assert B.sum_local == int(X.sum())
        assert np.array_equal(B.to_dense_u8(), X)
        bt = B.bits_t[:n].cpu().numpy().view(np.uint8)
        assert np.array_equal(np.unpackbits(bt, axis=1, bitorder="little")[:, :m], X.T)
        # padding is zero
        assert int(B.bits[m:].abs().sum().item()) == 0 and int(B.bits_t[n:].abs().sum().item()) == 0
    # float / sparse inputs, nonzero = 1
    import scipy.sparse as sp
    X = (rs.rand(100, 90) < 0.2).astype(np.float64)
    assert np.array_equal(E.BitMatrix(sp.csr_matrix(X), d).to_dense_u8(), X.astype(np.uint8))
    assert np.array_equal(E.BitMatrix(torch.from_numpy(X), d).to_dense_u8(), X.astype(np.uint8))


@pytest.mark.parametrize("kp,terms", [(32, 1), (32, 3), (64, 2), (64, 3)])
def test_make_panel_roundtrip(env, pos, kp, terms):
    L, E, d = env
    rs = np.random.RandomState(1)
    rows_pad = 512
    F = (np.abs(rs.standard_normal((rows_pad, kp))) * 10.0 ** rs.uniform(-8, 0, (rows_pad, kp))).astype(np.float32)
    Fd = dev(F, d)
    panel = torch.zeros((terms, kp, rows_pad), dtype=torch.int16, device=d)
    L.check(L.lib.bmf_make_panel(L.ptr(Fd), rows_pad, kp, kp, terms, L.ptr(panel), rows_pad, stream()))
    got = host_panel_value(panel, rows_pad, kp, terms, pos)
    if terms == 3:
        assert np.array_equal(got.astype(np.float32), F)  # three bf16 addends reproduce fp32 exactly
    else:
        tol = 2.0 ** (-8 * terms)
        np.testing.assert_allclose(got, F, rtol=tol)


@pytest.mark.parametrize("kp,terms,extra", [(32, 1, 0), (32, 2, 3), (64, 1, 1), (64, 3, 0), (64, 3, 2), (64, 2, 0)])
def test_xf_bits_exact_on_integers(env, kp, terms, extra):
    """0/1 bits times small integers: every product and partial sum is exact in bf16/fp32, so any layout or
    indexing error shows up as an exact mismatch.  Asymmetric data on purpose."""
    L, E, d = env
    rs = np.random.RandomState(2)
    rows, red = 700, 1000
    X = (rs.rand(rows, red) < 0.35).astype(np.uint8)
    B = E.BitMatrix(X, d)
    F = rs.randint(0, 8, size=(red, kp)).astype(np.float32)
    F[:, 0] = np.arange(red) % 7  # column structure
    red_pad = B.n_pad
    Fp = np.zeros((red_pad, kp), np.float32)
    Fp[:red] = F
    Fd = dev(Fp, d)
    panel = torch.zeros((terms, kp, red_pad), dtype=torch.int16, device=d)
    L.check(L.lib.bmf_make_panel(L.ptr(Fd), red_pad, kp, kp, terms, L.ptr(panel), red_pad, stream()))
    splits = E.xf_slots(B.m_pad, red_pad, terms, kp) + extra  # surplus slabs must come back as zeros
    out = torch.full((splits, B.m_pad, kp), -1.0, dtype=torch.float32, device=d)
    L.check(L.lib.bmf_xf_bits(L.ptr(B.bits), B.m_pad, B.ldx, red_pad // 32, L.ptr(panel), red_pad, terms, kp, L.ptr(out),
                              B.m_pad * kp, splits, stream()))
    assert L.lib.bmf_xf_bits(L.ptr(B.bits), B.m_pad, B.ldx, red_pad // 32, L.ptr(panel), red_pad, terms, kp, L.ptr(out),
                             B.m_pad * kp, 0, stream()) == L.lib.bmf_xf_bits(None, 0, 0, 0, None, 0, 0, 0, None, 0, 0, None) == -1
    got = out.sum(0).cpu().numpy()
    want = X.astype(np.float64) @ F.astype(np.float64)
    assert np.array_equal(got[:rows], want)
    assert not got[rows:].any()
    # and the transposed orientation:  X^T @ G
    G = rs.randint(0, 5, size=(rows, kp)).astype(np.float32)
    Gp = np.zeros((B.m_pad, kp), np.float32)
    Gp[:rows] = G
    panel2 = torch.zeros((terms, kp, B.m_pad), dtype=torch.int16, device=d)
    Gd = dev(Gp, d)
    L.check(L.lib.bmf_make_panel(L.ptr(Gd), B.m_pad, kp, kp, terms, L.ptr(panel2), B.m_pad, stream()))
    splits2 = E.xf_slots(B.n_pad, B.m_pad, terms, kp)
    out2 = torch.full((splits2, B.n_pad, kp), -1.0, dtype=torch.float32, device=d)
    L.check(L.lib.bmf_xf_bits(L.ptr(B.bits_t), B.n_pad, B.ldxt, B.m_pad // 32, L.ptr(panel2), B.m_pad, terms, kp, L.ptr(out2),
                              B.n_pad * kp, splits2, stream()))
    assert np.array_equal(out2.sum(0).cpu().numpy()[:red], X.T.astype(np.float64) @ G.astype(np.float64))


@pytest.mark.parametrize("terms,tol", [(3, 2e-6), (2, 3e-5), (1, 8e-3)])
def test_xf_bits_real_factors(env, terms, tol):
    L, E, d = env
    rs = np.random.RandomState(3)
    rows, red, kp = 900, 1300, 64
    X = (rs.rand(rows, red) < 0.25).astype(np.uint8)
    B = E.BitMatrix(X, d)
    F = np.zeros((B.n_pad, kp), np.float32)
    F[:red] = np.abs(rs.standard_normal((red, kp))).astype(np.float32) * 10.0 ** rs.uniform(-4, 0, (red, kp))
    panel = torch.zeros((terms, kp, B.n_pad), dtype=torch.int16, device=d)
    Fd = dev(F, d)
    L.check(L.lib.bmf_make_panel(L.ptr(Fd), B.n_pad, kp, kp, terms, L.ptr(panel), B.n_pad, stream()))
    splits = E.xf_slots(B.m_pad, B.n_pad, terms, kp)
    out = torch.full((splits, B.m_pad, kp), -1.0, dtype=torch.float32, device=d)
    L.check(L.lib.bmf_xf_bits(L.ptr(B.bits), B.m_pad, B.ldx, B.n_pad // 32, L.ptr(panel), B.n_pad, terms, kp, L.ptr(out),
                              B.m_pad * kp, splits, stream()))
    got = out.sum(0).double().cpu().numpy()[:rows]
    want = X.astype(np.float64) @ F[:red].astype(np.float64)
    err = np.linalg.norm(got - want) / np.linalg.norm(want)
    assert err < tol, err


def f16_panel(L, Fd, rows_pad, kp, d):
    panel = torch.zeros((2, kp, rows_pad), dtype=torch.int16, device=d)
    ws = torch.full((rows_pad // 128 * kp,), -7.0, dtype=torch.float32, device=d)  # scratch: no initial state required
    scale = torch.zeros((2 * kp,), dtype=torch.float32, device=d)
    L.check(L.lib.bmf_make_panel_f16(L.ptr(Fd), rows_pad, kp, kp, L.ptr(panel), rows_pad, L.ptr(ws), L.ptr(scale), stream()))
    return panel, scale, ws


@pytest.mark.parametrize("kp,rows_pad", [(32, 512), (64, 1536), (64, 512 * 300)])
def test_make_panel_f16(env, pos, kp, rows_pad):
    """Two fp16 addends of the column-scaled factor: scales are the exact powers of two that put the column maximum in
    [2^14, 2^15); hi + lo reproduces F to 2^-22 of the column scale (and exactly 0 stays 0)."""
    L, E, d = env
    rs = np.random.RandomState(5)
    F = (np.abs(rs.standard_normal((rows_pad, kp))) * 10.0 ** rs.uniform(-6, 0, (rows_pad, kp))).astype(np.float32)
    F[:, 1] *= 1e-12      # a column that lives at 1e-12
    F[:, 2] = 0.0         # an all-zero column
    F[:, 3] *= 3e4        # and a large one
    F[7, :] = 0.0
    panel, scale, ws = f16_panel(L, dev(F, d), rows_pad, kp, d)
    sc = scale.cpu().numpy().astype(np.float64)
    cmax = np.abs(F).max(0).astype(np.float64)
    for c in range(kp):
        if cmax[c] == 0:
            assert sc[c] == 1.0 and sc[kp + c] == 0.5
        else:
            assert 2.0 ** 14 <= cmax[c] * sc[c] < 2.0 ** 15 and np.log2(sc[c]) == np.round(np.log2(sc[c]))
            assert sc[kp + c] == 0.5 / sc[c]
    p = panel.cpu().numpy().view(np.float16).astype(np.float64)  # [2][kp][rows_pad]
    r = np.arange(rows_pad)
    idx = (r // 128) * 128 + pos[r % 128]
    got = p[:, :, idx].sum(0).T / sc[:kp]
    err = np.abs(got - F) / np.maximum(cmax, 1e-300)
    assert err.max() < 2.0 ** -22, err.max()
    assert not got[7].any() and not got[:, 2].any()
    # a second call on the same workspace gives the same panel
    panel2 = torch.zeros_like(panel)
    L.check(L.lib.bmf_make_panel_f16(L.ptr(dev(F, d)), rows_pad, kp, kp, L.ptr(panel2), rows_pad, L.ptr(ws), L.ptr(scale), stream()))
    assert torch.equal(panel, panel2)


@pytest.mark.parametrize("kp", [32, 64])
def test_xf_bits_f16(env, kp):
    """bits x fp16 panel: exact on small integers (layout check), ~1e-7 on real factors with wildly different columns."""
    L, E, d = env
    rs = np.random.RandomState(6)
    rows, red = 700, 1000
    X = (rs.rand(rows, red) < 0.35).astype(np.uint8)
    B = E.BitMatrix(X, d)
    red_pad = B.n_pad
    F = np.zeros((red_pad, kp), np.float32)
    F[:red] = rs.randint(0, 8, size=(red, kp))
    F[:red, 0] = np.arange(red) % 7
    panel, scale, _ = f16_panel(L, dev(F, d), red_pad, kp, d)
    splits = E.xf_slots(B.m_pad, red_pad, 2, kp) + 1
    out = torch.full((splits, B.m_pad, kp), -1.0, dtype=torch.float32, device=d)
    L.check(L.lib.bmf_xf_bits_f16(L.ptr(B.bits), B.m_pad, B.ldx, red_pad // 32, L.ptr(panel), red_pad, L.ptr(scale[kp:]), kp,
                                  L.ptr(out), B.m_pad * kp, splits, stream()))
    got = out.sum(0).cpu().numpy()
    assert np.array_equal(got[:rows], X.astype(np.float64) @ F[:red].astype(np.float64))
    assert not got[rows:].any()
    assert L.lib.bmf_xf_bits_f16(L.ptr(B.bits), B.m_pad, B.ldx, red_pad // 32, L.ptr(panel), red_pad, None, kp, L.ptr(out),
                                 B.m_pad * kp, splits, stream()) == -1
    # real factors; column magnitudes spread over 12 decades
    F[:red] = np.abs(rs.standard_normal((red, kp))).astype(np.float32) * 10.0 ** rs.uniform(-3, 0, (red, kp))
    F *= (10.0 ** rs.uniform(-9, 3, kp)).astype(np.float32)[None, :]
    panel, scale, _ = f16_panel(L, dev(F, d), red_pad, kp, d)
    L.check(L.lib.bmf_xf_bits_f16(L.ptr(B.bits), B.m_pad, B.ldx, red_pad // 32, L.ptr(panel), red_pad, L.ptr(scale[kp:]), kp,
                                  L.ptr(out), B.m_pad * kp, splits, stream()))
    got = out.sum(0).double().cpu().numpy()[:rows]
    want = X.astype(np.float64) @ F[:red].astype(np.float64)
    err = np.linalg.norm(got - want, axis=0) / np.linalg.norm(want, axis=0)   # per column: every scale must be right
    assert err.max() < 5e-7, err.max()


@pytest.mark.parametrize("kp,red_pad", [(32, 504), (64, 504), (32, 512), (64, 640), (32, 1344), (64, 1344)])
def test_xf_f32(env, kp, red_pad):
    """red_pad % 64 == 0 takes the LDS-ring kernel (64-row tiles, stages of 64 reduction indices; the split counts below give
    workgroups 1, 2, 3, many and zero stages), anything else the direct one."""
    L, E, d = env
    rs = np.random.RandomState(4)
    rows, red = 300, red_pad - 12 if red_pad > 640 else 500
    rows_pad = 384
    A = np.zeros((rows_pad, red_pad), np.float32)
    A[:rows, :red] = rs.rand(rows, red)
    FT = np.zeros((kp, red_pad), np.float32)
    FT[:, :red] = rs.rand(kp, red)
    Ad, FTd = dev(A, d), dev(FT, d)  # keep references: ptr() of a temporary would dangle
    for splits in (1, 3, 7, 8, 11):
        out = torch.full((splits, rows_pad, kp), np.nan, dtype=torch.float32, device=d)   # every slab must be written
        L.check(L.lib.bmf_xf_f32(L.ptr(Ad), rows_pad, red_pad, red_pad, L.ptr(FTd), red_pad, kp, L.ptr(out),
                                 rows_pad * kp, splits, stream()))
        got = out.sum(0).double().cpu().numpy()
        want = A.astype(np.float64) @ FT.T.astype(np.float64)
        assert np.linalg.norm(got - want) / np.linalg.norm(want) < 1e-6


@pytest.mark.parametrize("kp,rows_pad,blocks", [(32, 512, 3), (64, 1024, 16), (64, 2560, 7), (64, 128, 64), (64, 130, 5), (32, 20096, 256), (64, 20480, 256),
                                                 (64, 100352, 256), (64, 100352, 1000)])
def test_gram(env, kp, rows_pad, blocks):
    """(128 rows on 64 blocks: most waves own no row pair; 130 rows on 5 blocks: partial last groups; the headline shapes: 49 / 10 row pairs
    per wave through the ring of four load groups.)"""
    L, E, d = env
    rs = np.random.RandomState(5)
    F = (rs.rand(rows_pad, kp) - 0.25).astype(np.float32)
    F[rows_pad - min(37, rows_pad // 4):] = 0
    slabs = torch.zeros((blocks, kp, kp), dtype=torch.float32, device=d)
    g32 = torch.zeros((kp, kp), dtype=torch.float32, device=d)
    g64 = torch.zeros((kp, kp), dtype=torch.float64, device=d)
    Fd = dev(F, d)
    L.check(L.lib.bmf_gram_partial(L.ptr(Fd), rows_pad, kp, kp, L.ptr(slabs), blocks, stream()))
    L.check(L.lib.bmf_reduce_slabs(L.ptr(slabs), kp * kp, blocks, kp * kp, L.ptr(g32), L.ptr(g64), stream()))
    want = F.astype(np.float64).T @ F.astype(np.float64)
    np.testing.assert_allclose(g64.cpu().numpy(), want, rtol=2e-6, atol=2e-6 * np.abs(want).max())
    np.testing.assert_allclose(g32.cpu().numpy(), want, rtol=2e-6, atol=2e-6 * np.abs(want).max())
    g = g64.cpu().numpy()
    assert np.array_equal(g, g.T)   # the tile below the diagonal is the mirror of the one above it, bit for bit


def run_epilogue(L, d, F, rows, k, kp, num, G, reg, mode, thr, terms):
    rows_pad = F.shape[0]
    F64d = dev(F.astype(np.float64), d)
    Fd = F64d.float()  # the fp32 shadow must equal (float)F64 on entry: it is the MFMA operand of F G
    a = L.EpilogueArgs()
    numd = None if num is None else dev(num, d)
    Gd = dev(G, d)
    panel = torch.zeros((terms, kp, rows_pad), dtype=torch.int16, device=d)
    rowbits = torch.zeros((rows_pad,), dtype=torch.int64, device=d)
    colbits = torch.zeros((kp, rows_pad // 32), dtype=torch.int32, device=d)
    partials = torch.zeros((rows_pad // 128, 2), dtype=torch.float64, device=d)
    a.F64, a.F, a.rows_pad, a.rows, a.k, a.kp = F64d.data_ptr(), Fd.data_ptr(), rows_pad, rows, k, kp
    a.num = 0 if numd is None else numd.data_ptr()
    a.slab_stride, a.splits = rows_pad * kp, (1 if num is None else num.shape[0])
    a.G, a.reg, a.mode, a.thr, a.terms = Gd.data_ptr(), reg, mode, thr, terms
    a.panel, a.ldp, a.rowbits, a.colbits, a.ldcb = panel.data_ptr(), rows_pad, rowbits.data_ptr(), colbits.data_ptr(), rows_pad // 32
    a.partials, a.stop = partials.data_ptr(), 0
    L.check(L.lib.bmf_mu_epilogue(C.byref(a), stream()))
    assert np.array_equal(Fd.cpu().numpy(), F64d.cpu().numpy().astype(np.float32))  # shadow = rounded master
    return Fd.cpu().numpy(), panel, rowbits.cpu().numpy(), colbits.cpu().numpy(), partials.cpu().numpy()


@pytest.mark.parametrize("k,kp,reg,mode", [(5, 32, 1.0, 1), (8, 32, 0.0, 1), (64, 64, 3.5, 1), (40, 64, 1e3, 1), (12, 32, 0.0, 2)])
def test_mu_epilogue(env, pos, k, kp, reg, mode):
    L, E, d = env
    rs = np.random.RandomState(6)
    rows, rows_pad, splits, terms = 333, 512, 3, 3
    F = np.zeros((rows_pad, kp), np.float32)
    F[:rows, :k] = np.abs(rs.standard_normal((rows, k))) * 0.5
    F[5, :k] = 0.0  # an all-zero row: denom == 0 -> eps (reg = 0) and F == 0 -> eps
    other = np.abs(rs.standard_normal((200, k))) * 0.5
    G = np.zeros((kp, kp), np.float32)
    G[:k, :k] = (other.T @ other).astype(np.float32)
    num = np.zeros((splits, rows_pad, kp), np.float32)
    num[:, :rows, :k] = rs.rand(splits, rows, k) * 3
    if mode == 1 and reg == 0.0:
        num[:, 7, :k] = 0.0  # zero numerator -> F_new == 0 -> eps clamp
    Fn, panel, rowbits, colbits, partials = run_epilogue(L, d, F, rows, k, kp, num, G, reg, mode, 0.5, terms)

    f = F[:rows, :k].astype(np.float64)
    nm = num[:, :rows, :k].astype(np.float64).sum(0)
    den = f @ G[:k, :k].astype(np.float64)
    if mode == 1:
        nume = nm + 3 * reg * f ** 2
        den = den + 2 * reg * f ** 3 + reg * f
    else:
        nume = nm
    den[den == 0] = orc.EPS
    want = f * (nume / den)
    if mode == 1:
        want[want == 0] = orc.EPS
    got = Fn[:rows, :k].astype(np.float64)
    np.testing.assert_allclose(got, want, rtol=3e-6, atol=1e-30)
    assert not Fn[rows:].any() and not Fn[:, k:].any()
    if mode == 1:
        assert (Fn[5, :k] == np.float32(orc.EPS)).all()
    # thresholded bits (both forms) agree with the fp32 factor the kernel produced
    want_bits = Fn[:rows, :k] > 0.5
    rb = rowbits.view(np.uint64)
    got_bits = ((rb[:rows, None] >> np.arange(k, dtype=np.uint64)[None, :]) & np.uint64(1)).astype(bool)
    assert np.array_equal(got_bits, want_bits)
    assert not rb[rows:].any()
    cb = np.unpackbits(colbits.view(np.uint8), axis=1, bitorder="little")  # [kp][rows_pad]
    assert np.array_equal(cb[:k, :rows].astype(bool), want_bits.T) and not cb[:, rows:].any() and not cb[k:].any()
    # partial sums
    fn64 = Fn.astype(np.float64)
    np.testing.assert_allclose(partials[:, 0].sum(), ((fn64 ** 2 - fn64) ** 2).sum(), rtol=1e-5)
    np.testing.assert_allclose(partials[:, 1].sum(), (fn64 * num.astype(np.float64).sum(0)).sum(), rtol=1e-5)
    # the panel holds the new factor exactly (3 addends)
    assert np.array_equal(host_panel_value(panel, rows_pad, kp, terms, pos).astype(np.float32), Fn)


@pytest.mark.parametrize("k,kp,reg,mode,limbs", [(64, 64, 1.5, 1, 3), (40, 64, 0.0, 2, 3), (12, 32, 2.0, 1, 3), (20, 32, 0.0, 0, 2), (64, 64, 1.0, 1, 2), (33, 64, 0.0, 0, 3)])
def test_mu_epilogue_emits_the_int8_digit_planes(env, k, kp, reg, mode, limbs):
    """bmf_mu_epilogue with `planes`: same update, bits, partial sums and column maxima as the plain form, and the digit planes it
    writes with a given column scale are byte for byte what bmf_make_panel_i8 builds from the updated factor with that scale
    (csrc/epilogue.hip: mu_epilogue_i8_kernel; the layout argument is in its header comment)."""
    L, E, d = env
    rs = np.random.RandomState(16)
    rows, rows_pad, splits = 900, 1024, 2
    F = np.zeros((rows_pad, kp), np.float64)
    F[:rows, :k] = np.abs(rs.standard_normal((rows, k))) * 10.0 ** rs.uniform(-3, 0, (rows, k))
    F[5, :k] = 0.0
    other = np.abs(rs.standard_normal((300, k))) * 0.5
    G = np.zeros((kp, kp), np.float32)
    G[:k, :k] = (other.T @ other).astype(np.float32)
    num = np.zeros((splits, rows_pad, kp), np.float32)
    num[:, :rows, :k] = rs.rand(splits, rows, k) * 3

    def run(with_planes, scale_vec=None):
        a = L.EpilogueArgs()
        F64d, Fd = dev(F, d), dev(F.astype(np.float32), d)
        numd, Gd = dev(num, d), dev(G, d)
        rowbits = torch.zeros(rows_pad, dtype=torch.int64, device=d)
        colbits = torch.zeros((kp, rows_pad // 32), dtype=torch.int32, device=d)
        partials = torch.zeros((rows_pad // 128, 2), dtype=torch.float64, device=d)
        blockmax = torch.zeros((rows_pad // 128, kp), dtype=torch.float32, device=d)
        planes = torch.full((limbs, kp, rows_pad), 77, dtype=torch.int8, device=d)
        a.F64, a.F, a.rows_pad, a.rows, a.k, a.kp = F64d.data_ptr(), Fd.data_ptr(), rows_pad, rows, k, kp
        a.num, a.slab_stride, a.splits = (numd.data_ptr() if mode != 0 else 0), rows_pad * kp, splits
        a.G, a.reg, a.mode, a.thr, a.terms = Gd.data_ptr(), reg, mode, 0.5, 0
        a.panel, a.ldp, a.rowbits, a.colbits, a.ldcb = 0, rows_pad, rowbits.data_ptr(), colbits.data_ptr(), rows_pad // 32
        a.partials, a.stop, a.blockmax = partials.data_ptr(), 0, blockmax.data_ptr()
        if with_planes:
            sc = dev(scale_vec.astype(np.float32), d)
            a.planes, a.plane_scale, a.limbs = planes.data_ptr(), sc.data_ptr(), limbs
        L.check(L.lib.bmf_mu_epilogue(C.byref(a), stream()))
        return (F64d.cpu().numpy(), Fd.cpu().numpy(), rowbits.cpu().numpy(), colbits.cpu().numpy(), partials.cpu().numpy(), blockmax.cpu().numpy(),
                planes.cpu().numpy(), F64d, Fd)

    ref = run(False)
    # the exact column scales of the UPDATED factor, from the stand-alone builder; planes from the builder = the reference planes
    want_planes = torch.zeros((limbs, kp, rows_pad), dtype=torch.int8, device=d)
    ws = torch.zeros(rows_pad // 128 * kp, dtype=torch.float32, device=d)
    scale = torch.zeros(2 * kp, dtype=torch.float32, device=d)
    L.check(L.lib.bmf_make_panel_i8(L.ptr(ref[7]), L.ptr(ref[8]), rows_pad, kp, kp, limbs, L.ptr(want_planes), rows_pad, L.ptr(ws), L.ptr(scale), stream()))
    sc = scale.cpu().numpy()[:kp]
    got = run(True, sc)
    for i in range(6):   # factors (fp64 + shadow), bits, partial sums, column maxima: identical arithmetic
        if i == 4:
            np.testing.assert_allclose(got[i], ref[i], rtol=1e-14)
        else:
            assert np.array_equal(got[i], ref[i]), i
    assert np.array_equal(got[6], want_planes.cpu().numpy())
    # with a guarded (one bit lower) scale the planes are the builder's for THAT scale: check by value, q = rint(F 2^(e-1))
    got2 = run(True, sc * 0.5)
    pl = got2[6].astype(np.int64)
    r = np.arange(rows_pad)
    posi8 = np.array([L.lib.bmf_panel_pos_i8(int(c)) for c in range(512)])
    idx = (r // 512) * 512 + posi8[r % 512]
    q = sum(pl[l][:, idx].T * 256 ** l for l in range(limbs)) * (256 if limbs == 2 else 1)
    want_q = np.rint(np.clip(got2[0] * (sc * 0.5)[None, :].astype(np.float64), -8355711.0, 8355711.0))
    if limbs == 2:
        want_q = np.floor((want_q + 128) / 256) * 256
    assert np.array_equal(q, want_q.astype(np.int64))


def test_mu_epilogue_prepare_mode(env, pos):
    L, E, d = env
    rs = np.random.RandomState(7)
    rows, rows_pad, k, kp = 130, 256, 9, 32
    F = np.zeros((rows_pad, kp), np.float32)
    F[:rows, :k] = rs.rand(rows, k)
    G = np.zeros((kp, kp), np.float32)
    Fn, panel, rowbits, colbits, partials = run_epilogue(L, d, F, rows, k, kp, None, G, 0.0, 0, 0.3, 2)
    assert np.array_equal(Fn, F)
    got = host_panel_value(panel, rows_pad, kp, 2, pos)
    np.testing.assert_allclose(got, F, rtol=2 ** -15)
    assert partials[:, 1].sum() == 0.0


@pytest.mark.parametrize("m,n,k,du,dv", [(300, 500, 8, 0.3, 0.3), (1000, 9000, 64, 0.05, 0.1), (257, 129, 33, 0.9, 0.9), (64, 64, 1, 0.5, 0.5)])
def test_cover_count_bit_exact(env, m, n, k, du, dv):
    L, E, d = env
    rs = np.random.RandomState(8)
    kp = 32 if k <= 32 else 64
    X = (rs.rand(m, n) < 0.3).astype(np.uint8)
    Ub = (rs.rand(m, k) < du)
    Vb = (rs.rand(n, k) < dv)
    Ub[3] = False
    B = E.BitMatrix(X, d)
    w = (1 << np.arange(k, dtype=np.uint64))
    ubits = np.zeros(B.m_pad, np.uint64)
    ubits[:m] = (Ub.astype(np.uint64) * w).sum(1)
    vcol = np.zeros((kp, B.n_pad), np.uint8)
    vcol[:k, :n] = Vb.T
    vcolbits = np.packbits(vcol, axis=1, bitorder="little").view(np.int32)
    counts = torch.zeros(4, dtype=torch.int64, device=d)
    ud, vd = dev(ubits.view(np.int64), d), dev(vcolbits, d)
    L.check(L.lib.bmf_cover_count(L.ptr(B.bits), B.m_pad, B.ldx, B.n_pad // 32, L.ptr(ud), L.ptr(vd), B.n_pad // 32, kp,
                                  L.ptr(counts), None, stream()))
    pd = orc.boolean_product(Ub.astype(np.int64), Vb.astype(np.int64))
    tp, fp, fn, tn = orc.confusion_counts(X.astype(np.int64), pd)
    got = counts.cpu().numpy()
    assert (int(got[0]), int(got[1])) == (tp, fp)


@pytest.mark.parametrize("m,n,k", [(300, 500, 8), (130, 70, 40), (1000, 333, 64)])
def test_residual_sums(env, m, n, k):
    L, E, d = env
    rs = np.random.RandomState(9)
    kp = 32 if k <= 32 else 64
    X = (rs.rand(m, n) < 0.3).astype(np.uint8)
    B = E.BitMatrix(X, d)
    U = np.zeros((B.m_pad, kp), np.float32)
    V = np.zeros((B.n_pad, kp), np.float32)
    U[:m, :k] = rs.rand(m, k) * 0.4
    V[:n, :k] = rs.rand(n, k) * 0.4
    sums = torch.zeros(4, dtype=torch.float64, device=d)
    Ud, Vd = dev(U, d), dev(V, d)
    L.check(L.lib.bmf_residual_sums(L.ptr(B.bits), B.m_pad, B.ldx, m, n, L.ptr(Ud), L.ptr(Vd), kp, L.ptr(sums), None, stream()))
    R = X.astype(np.float64) - U[:m].astype(np.float64) @ V[:n].astype(np.float64).T
    got = sums.cpu().numpy()
    np.testing.assert_allclose(got[0], np.abs(R).sum(), rtol=1e-6)
    np.testing.assert_allclose(got[1], (R ** 2).sum(), rtol=1e-6)


def test_thresh_eval_against_golden(env, golden_dir):
    L, E, d = env
    z = np.load(os.path.join(golden_dir, "g4_threshold.npz"))
    meta = json.load(open(os.path.join(golden_dir, "g4_threshold.json")))
    shape = z["shape"]
    X = np.unpackbits(z["X_bits"], axis=1, bitorder="little")[:, : shape[1]]
    m, n = X.shape
    k, kp = 16, 32
    B = E.BitMatrix(X, d)
    U = np.zeros((B.m_pad, kp), np.float32)
    V = np.zeros((B.n_pad, kp), np.float32)
    U[:m, :k], V[:n, :k] = z["U"], z["V"]
    Ud, Vd = dev(U, d), dev(V, d)
    work = torch.zeros(((2 * B.m_pad + 2 * B.n_pad) * kp,), dtype=torch.float32, device=d)
    out = torch.zeros(4, dtype=torch.float64, device=d)
    # the factors were rounded to fp32 on upload: compare against the oracle on the same rounded factors (tight)
    U32, V32 = U[:m, :k].astype(np.float64), V[:n, :k].astype(np.float64)
    Xf = X.astype(np.float64)
    for lam in (10, 100):
        for i, u in enumerate(meta["grid_u"]):
            for j, v in enumerate(meta["grid_v"]):
                L.check(L.lib.bmf_thresh_eval(L.ptr(B.bits), B.m_pad, B.ldx, m, n, L.ptr(Ud), B.n_pad, L.ptr(Vd), k, kp, u, v,
                                              float(lam), 1, L.ptr(work), L.ptr(out), stream()))
                o = out.cpu().numpy()
                F_same = orc.thresh_F(Xf, None, U32, V32, u, v, lam)
                dF_same = orc.thresh_dF(Xf, None, U32, V32, u, v, lam)
                assert 0.5 * o[1] == pytest.approx(F_same, rel=2e-6)
                scale = np.abs(dF_same).max() + 1.0
                assert abs(o[2] - dF_same[0]) < 2e-5 * scale and abs(o[3] - dF_same[1]) < 2e-5 * scale
                # and against the reference's own fp64 numbers (golden), looser: fp32 factor rounding
                assert 0.5 * o[1] == pytest.approx(z[f"F_grid_lam{lam}"][i, j], rel=1e-4)


def _trace_inputs(L, X, d):
    """The ones of X as the segment list bmf_thresh_trace64 takes (<= 128 cells of one row each, longest first)."""
    m = X.shape[0]
    rows, cols = np.nonzero(X)
    counts = np.bincount(rows, minlength=m)
    starts = np.cumsum(counts) - counts
    seg = [(i, starts[i] + a, min(128, counts[i] - a)) for i in range(m) for a in range(0, counts[i], 128)]
    seg.sort(key=lambda t: -t[2])
    seg_row = dev(np.array([t[0] for t in seg], np.int32), d)
    seg_beg = dev(np.array([t[1] for t in seg], np.int64), d)
    seg_len = dev(np.array([t[2] for t in seg], np.int32), d)
    return seg_row, seg_beg, seg_len, len(seg), dev(cols.astype(np.int32), d), float(len(rows))


def _trace_eval(L, d, X, U, V, pts, lam, grad):
    m, n = X.shape
    k = U.shape[1]
    kp = 32 if k <= 32 else 64
    Ud = torch.zeros((m + 5, kp), dtype=torch.float64, device=d)
    Vd = torch.zeros((n + 3, kp), dtype=torch.float64, device=d)
    Ud[:m, :k], Vd[:n, :k] = torch.from_numpy(U).to(d), torch.from_numpy(V).to(d)
    seg_row, seg_beg, seg_len, nseg, idx, sx = _trace_inputs(L, X, d)
    maxp = L.lib.bmf_thresh_trace64_max_pairs(k)
    work = torch.zeros(int(L.lib.bmf_thresh_trace64_work(m, n, k, maxp)), dtype=torch.float64, device=d)
    out_host = torch.zeros(4 * maxp + 1, dtype=torch.float64).pin_memory()
    res, seq = [], 0.0
    for a in range(0, len(pts), maxp):
        part = pts[a:a + maxp]
        uv = (C.c_double * (2 * len(part)))(*[x for p_ in part for x in p_])
        seq += 1.0
        L.check(L.lib.bmf_thresh_trace64(L.ptr(seg_row), L.ptr(seg_beg), L.ptr(seg_len), nseg, L.ptr(idx), m, n, L.ptr(Ud), L.ptr(Vd), kp, k, uv,
                                         len(part), float(lam), sx, int(grad), L.ptr(work), L.ptr(out_host), seq, stream()))
        torch.cuda.synchronize()
        o = out_host.numpy()
        assert o[4 * len(part)] == seq            # the sequence word is written last
        assert (o[0:4 * len(part):4] == seq).all()   # and every pair carries the call's stamp behind its results
        res += [(0.5 * o[4 * i + 1], o[4 * i + 2], o[4 * i + 3]) for i in range(len(part))]
    return res


def test_thresh_trace64_against_golden(env, golden_dir):
    """The batched trace-form objective (csrc/thresh_trace.hip) on the reference's own grid (golden g4): F to 1e-11, dF to 1e-9 of
    its scale, all 25 points of a grid in one call each for F and for F + dF."""
    L, E, d = env
    z = np.load(os.path.join(golden_dir, "g4_threshold.npz"))
    meta = json.load(open(os.path.join(golden_dir, "g4_threshold.json")))
    X = np.unpackbits(z["X_bits"], axis=1, bitorder="little")[:, : z["shape"][1]]
    pts = [(u, v) for u in meta["grid_u"] for v in meta["grid_v"]]
    for lam in (10, 100):
        Fg, dFg = z[f"F_grid_lam{lam}"].ravel(), z[f"dF_grid_lam{lam}"].reshape(-1, 2)
        got_f = _trace_eval(L, d, X, z["U"], z["V"], pts, lam, False)
        got_g = _trace_eval(L, d, X, z["U"], z["V"], pts, lam, True)
        np.testing.assert_allclose([g[0] for g in got_f], Fg, rtol=1e-11)
        np.testing.assert_allclose([g[0] for g in got_g], Fg, rtol=1e-11)
        assert np.array_equal([g[0] for g in got_f], [g[0] for g in got_g])     # F does not depend on whether the gradient rides along
        scale = np.abs(dFg).max()
        assert np.abs(np.array([g[1:] for g in got_g]) - dFg).max() <= 1e-9 * scale


@pytest.mark.parametrize("m,n,k,dens", [(70, 45, 5, 0.3), (300, 333, 20, 0.1), (260, 200, 40, 0.5), (33, 700, 64, 0.9), (50, 40, 16, 0.0)])
def test_thresh_trace64_shapes(env, m, n, k, dens):
    """16, 32 and 64 lanes per pair, rows longer than one segment, an empty matrix, batches that need several calls: against the oracle."""
    L, E, d = env
    rs = np.random.RandomState(k)
    X = (rs.rand(m, n) < dens).astype(np.uint8)
    if dens > 0:
        X[0, :] = 1
        X[3, :] = 0
    else:
        X[1, 2] = 1   # (one cell: the list must not be empty)
    U, V = rs.rand(m, k), rs.rand(n, k)
    pts = [(0.1 + 0.8 * rs.rand(), 0.1 + 0.8 * rs.rand()) for _ in range(37)]
    Xf = X.astype(np.float64)
    for lam, grad in ((10, True), (100, False)):
        got = _trace_eval(L, d, X, U, V, pts, lam, grad)
        for (u, v), g in zip(pts, got):
            assert g[0] == pytest.approx(orc.thresh_F(Xf, None, U, V, u, v, lam), rel=1e-11, abs=1e-9)
            if grad:
                want = orc.thresh_dF(Xf, None, U, V, u, v, lam)
                assert np.abs(np.array(g[1:]) - want).max() <= 1e-9 * (np.abs(want).max() + 1.0)


@pytest.mark.parametrize("m,n,k", [(70, 45, 5), (300, 333, 40), (129, 31, 64)])
def test_real_product(env, m, n, k):
    from pybmf_amd.device_ops import real_product, product_csr
    rs = np.random.RandomState(10)
    U, V = rs.rand(m, k), rs.rand(n, k)
    got = real_product(U, V)
    assert got.shape == (m, n)
    np.testing.assert_allclose(got, U @ V.T, rtol=2e-6)
    assert np.allclose(np.asarray(product_csr(U, V, boolean=False).todense()), U @ V.T, rtol=2e-6)
    Ub, Vb = (U > 0.7).astype(np.int64), (V > 0.7).astype(np.int64)
    assert np.array_equal(np.asarray(product_csr(Ub, Vb, boolean=True).todense()), orc.boolean_product(Ub, Vb))


@pytest.mark.parametrize("m,n,k", [(300, 200, 8), (1000, 700, 40), (513, 129, 64)])
def test_mae_sum_bf16_mfma(env, m, n, k):
    """sum |X - U V^T| on the split-bf16 MFMA against NumPy fp64 and against the exact-fp32 residual pass."""
    L, E, d = env
    rs = np.random.RandomState(m + k)
    X = (rs.rand(m, n) < 0.3).astype(np.uint8)
    B = E.BitMatrix(X, d)
    kp = 32 if k <= 32 else 64
    U = np.zeros((B.m_pad, kp), np.float32)
    V = np.zeros((B.n_pad, kp), np.float32)
    U[:m, :k] = np.abs(rs.standard_normal((m, k))) * 0.4 * 10.0 ** rs.uniform(-3, 0, (m, k))
    V[:n, :k] = np.abs(rs.standard_normal((n, k))) * 0.4
    Ud, Vd = dev(U, d), dev(V, d)
    ws = torch.zeros(2 * (B.m_pad + B.n_pad) * kp, dtype=torch.int16, device=d)
    out = torch.zeros(1, dtype=torch.float64, device=d)
    L.check(L.lib.bmf_mae_sum(L.ptr(B.bits_t), B.ldxt, B.m_pad, B.n_pad, L.ptr(Ud), L.ptr(Vd), kp, L.ptr(ws), L.ptr(out), stream()))
    want = np.abs(X.astype(np.float64) - U[:m].astype(np.float64) @ V[:n].astype(np.float64).T).sum()
    got = float(out.item())
    assert got == pytest.approx(want, rel=2e-6)
    sums = torch.zeros(4, dtype=torch.float64, device=d)
    L.check(L.lib.bmf_residual_sums(L.ptr(B.bits), B.m_pad, B.ldx, m, n, L.ptr(Ud), L.ptr(Vd), kp, L.ptr(sums), None, stream()))
    assert got == pytest.approx(float(sums[0].item()), rel=2e-6)
    assert L.lib.bmf_mae_sum(L.ptr(B.bits_t), B.ldxt, B.m_pad + 1, B.n_pad, L.ptr(Ud), L.ptr(Vd), kp, L.ptr(ws), L.ptr(out), stream()) == -1
    # both operand precisions, spelled out: three bf16 products (per-cell accuracy) and ONE fp16 product (a third of the MFMA
    # work; what the size rule picks from 2^24 cells on: its per-cell error mostly averages out in the sum)
    for one, tol in ((0, 2e-6), (1, 2e-5)):
        out.zero_()
        L.check(L.lib.bmf_mae_sum_ex(L.ptr(B.bits_t), B.ldxt, B.m_pad, B.n_pad, L.ptr(Ud), L.ptr(Vd), kp, L.ptr(ws), L.ptr(out), one, stream()))
        assert float(out.item()) == pytest.approx(want, rel=tol), (one, float(out.item()) / want - 1)
    # the single-product pass reading X^T from its tiled copy (what the iteration driver does when the int8 GEMM has one)
    tiled = torch.empty_like(B.bits_t)
    L.check(L.lib.bmf_tile_bits(L.ptr(B.bits_t), B.n_pad, B.ldxt, B.ldxt, L.ptr(tiled), stream()))
    out.zero_()
    L.check(L.lib.bmf_mae_sum_tiled(L.ptr(tiled), B.ldxt, B.m_pad, B.n_pad, L.ptr(Ud), L.ptr(Vd), kp, L.ptr(ws), L.ptr(out), stream()))
    assert float(out.item()) == pytest.approx(want, rel=2e-5), float(out.item()) / want - 1


def test_panel_pos_i8_is_a_permutation(env):
    L, _, _ = env
    assert sorted(L.lib.bmf_panel_pos_i8(i) for i in range(512)) == list(range(512))
    assert L.lib.bmf_panel_pos_i8(512) == -1


def i8_quantise(F64, limbs):
    """Host restatement of bmf_make_panel_i8: q = rint(F 2^e), column maxima (of the fp32 shadow) into [2^22, 0.996 * 2^23], or
    [2^21, 2^22) when they would land above that; limbs = 2 rounds q to a multiple of 256.  Returns the value the digit planes
    represent, per entry."""
    m = np.abs(F64.astype(np.float32)).max(axis=0)
    fr, ex = np.frexp(np.where(m > 0, m, np.float32(1.0)))
    e = np.where(m > 0, np.where(fr > np.float32(0.99599), 22, 23) - ex, 0)
    q = np.rint(F64 * 2.0 ** e[None, :])
    if limbs == 2:
        q = np.floor((q + 128) / 256) * 256
    return q * 2.0 ** -e[None, :]


@pytest.fixture
def i8_variant(request, env):
    """Runs a test on one variant of the 64-column bits GEMM (bmf_xf_bits_i8_variant) and restores the default afterwards."""
    L = env[0]
    prev = L.lib.bmf_xf_bits_i8_variant(request.param)
    assert prev >= 0
    yield request.param
    L.lib.bmf_xf_bits_i8_variant(prev)


@pytest.mark.parametrize("i8_variant", [0, 1, 2, 4], indirect=True)
@pytest.mark.parametrize("kp,limbs,rows,red", [(64, 3, 700, 1000), (32, 3, 1500, 700), (64, 2, 513, 384), (32, 2, 40, 4100), (64, 3, 3000, 20000)])
def test_xf_bits_i8_is_exact(env, i8_variant, kp, limbs, rows, red):
    """bits x int8 digit planes: the product of X with the QUANTISED factor, exactly (int32 accumulation, fp64 recombination, one
    rounding to fp32 per slab), whatever the column magnitudes -- on each kernel variant (csrc/xf_bits_i8.hip, xf_bits_i8w.hip; 4 = the
    anti-phase eight-wave kernel of xf_bits_i8p.hip: tiled bit matrix, three planes)."""
    L, E, d = env
    if kp == 32 and i8_variant != 0:
        pytest.skip("the 64-column variants serve kp = 64 only")
    if i8_variant == 4 and limbs != 3:
        pytest.skip("variant 4 takes three digit planes")
    rs = np.random.RandomState(16)
    X = (rs.rand(rows, red) < 0.3).astype(np.uint8)
    B = E.BitMatrix(X, d)
    red_pad = B.n_pad
    F64 = np.zeros((red_pad, kp))
    F64[:red] = np.abs(rs.standard_normal((red, kp))) * 10.0 ** rs.uniform(-3, 0, (red, kp))
    F64 *= (10.0 ** rs.uniform(-9, 3, kp))[None, :]
    F64[:red, 1] *= -1.0          # signed digits
    F64[: red // 2, 2] = 0.0
    Fd, F32 = dev(F64, d), dev(F64.astype(np.float32), d)
    panel = torch.zeros((limbs, kp, red_pad), dtype=torch.int8, device=d)
    scale = torch.zeros(2 * kp, dtype=torch.float32, device=d)
    ws = torch.zeros(red_pad // 128 * kp, dtype=torch.float32, device=d)
    L.check(L.lib.bmf_make_panel_i8(L.ptr(Fd), L.ptr(F32), red_pad, kp, kp, limbs, L.ptr(panel), red_pad, L.ptr(ws), L.ptr(scale), stream()))
    # the planes, read back and un-permuted, are the balanced digits of the quantised factor
    pos = np.array([L.lib.bmf_panel_pos_i8(i) for i in range(512)])
    P = panel.cpu().numpy().astype(np.float64).reshape(limbs, kp, red_pad // 512, 512)[:, :, :, pos].reshape(limbs, kp, red_pad)
    sc = scale.cpu().numpy().astype(np.float64)
    value = sum(P[l] * 256.0 ** l for l in range(limbs)).T * sc[kp:][None, :]
    want_q = i8_quantise(F64, limbs)
    assert np.array_equal(value, want_q)
    rel = np.abs(want_q - F64).max(axis=0) / np.abs(F64).max(axis=0)
    assert rel.max() <= (2.0 ** -22 if limbs == 3 else 2.0 ** -14)   # half a unit of the last digit kept
    splits = E.xf_slots_i8(B.m_pad, red_pad, kp) + 1
    out = torch.full((splits, B.m_pad, kp), -1.0, dtype=torch.float32, device=d)
    tiled = B.tiled()[0]
    words = B.bits.cpu().numpy().reshape(B.m_pad // 256, 256, B.ldx // 16, 16).transpose(0, 2, 1, 3)
    assert np.array_equal(tiled.cpu().numpy().ravel(), words.ravel())
    if i8_variant == 4:   # reads the tiled copy only, and says so
        assert L.lib.bmf_xf_bits_i8(L.ptr(B.bits), B.m_pad, B.ldx, red_pad // 32, L.ptr(panel), red_pad, limbs, L.ptr(scale[kp:]), kp,
                                    L.ptr(out), B.m_pad * kp, splits, 0, stream()) == -1 and b"tiled" in L.lib.bmf_last_error()
        L.check(L.lib.bmf_xf_bits_i8(L.ptr(tiled), B.m_pad, B.ldx, red_pad // 32, L.ptr(panel), red_pad, limbs, L.ptr(scale[kp:]), kp,
                                     L.ptr(out), B.m_pad * kp, splits, 1, stream()))
    else:
        L.check(L.lib.bmf_xf_bits_i8(L.ptr(B.bits), B.m_pad, B.ldx, red_pad // 32, L.ptr(panel), red_pad, limbs, L.ptr(scale[kp:]), kp,
                                     L.ptr(out), B.m_pad * kp, splits, 0, stream()))
        # the tiled layout of the bit matrix (bmf_tile_bits) gives bit-identical slabs
        out2 = torch.full_like(out, -1.0)
        L.check(L.lib.bmf_xf_bits_i8(L.ptr(tiled), B.m_pad, B.ldx, red_pad // 32, L.ptr(panel), red_pad, limbs, L.ptr(scale[kp:]), kp,
                                     L.ptr(out2), B.m_pad * kp, splits, 1, stream()))
        assert torch.equal(out, out2)
    slabs = out.cpu().numpy()
    assert not slabs[:, rows:].any()
    exact = X.astype(np.float64) @ want_q[:red]
    got = slabs.astype(np.float64).sum(0)[:rows]
    nz = (slabs != 0).any(axis=(1, 2)).sum()
    # every slab holds an exactly computed partial sum rounded once to fp32
    assert np.abs(got - exact).max() <= nz * 2.0 ** -24 * np.abs(exact).max() * 1.0001 + 0.0
    err = np.linalg.norm(got - exact, axis=0) / np.maximum(np.linalg.norm(exact, axis=0), 1e-300)
    assert err.max() < 1e-7, err.max()
    assert L.lib.bmf_xf_bits_i8(L.ptr(B.bits), B.m_pad, B.ldx, red_pad // 32, L.ptr(panel), red_pad, 4, L.ptr(scale[kp:]), kp, L.ptr(out),
                                B.m_pad * kp, splits, 0, stream()) == -1


def test_thresh_eval64_against_golden(env, golden_dir):
    """The fp64 thresholding objective (csrc/thresh64.hip) against the REFERENCE's own F on the golden grid and the oracle's dF,
    from the fp64 factors: 1e-10 (round 1's fp32 path: F 1e-4, dF only against fp32-rounded factors)."""
    L, E, d = env
    z = np.load(os.path.join(golden_dir, "g4_threshold.npz"))
    meta = json.load(open(os.path.join(golden_dir, "g4_threshold.json")))
    X = np.unpackbits(z["X_bits"], axis=1, bitorder="little")[:, : int(z["shape"][1])]
    m, n = X.shape
    k, kp = 16, 32
    B = E.BitMatrix(X, d)
    U = np.zeros((B.m_pad, kp))
    V = np.zeros((B.n_pad, kp))
    U[:m, :k], V[:n, :k] = z["U"], z["V"]
    Ud, Vd = dev(U, d), dev(V, d)
    work = torch.zeros((int(L.lib.bmf_thresh_eval64_work(B.m_pad, B.n_pad, kp)),), dtype=torch.float64, device=d)
    out = torch.zeros(4, dtype=torch.float64, device=d)
    Xf = X.astype(np.float64)
    for lam in (10, 100):
        for i, u in enumerate(meta["grid_u"]):
            for j, v in enumerate(meta["grid_v"]):
                L.check(L.lib.bmf_thresh_eval64(L.ptr(B.bits), B.m_pad, B.ldx, m, n, L.ptr(Ud), B.n_pad, L.ptr(Vd), k, kp, u, v,
                                                float(lam), 1, L.ptr(work), L.ptr(out), stream()))
                o = out.cpu().numpy()
                assert 0.5 * o[1] == pytest.approx(z[f"F_grid_lam{lam}"][i, j], rel=1e-11)
                dF = orc.thresh_dF(Xf, None, z["U"], z["V"], u, v, lam)
                np.testing.assert_allclose(o[2:4], dF, rtol=1e-9, atol=1e-9 * np.abs(dF).max())
                assert o[0] == pytest.approx(np.abs(Xf - orc.stable_sigmoid((z["U"] - u) * lam) @ orc.stable_sigmoid((z["V"] - v) * lam).T).sum(), rel=1e-11)
    # deterministic: bit-identical when repeated
    o1 = out.clone()
    L.check(L.lib.bmf_thresh_eval64(L.ptr(B.bits), B.m_pad, B.ldx, m, n, L.ptr(Ud), B.n_pad, L.ptr(Vd), k, kp, meta["grid_u"][-1], meta["grid_v"][-1],
                                    100.0, 1, L.ptr(work), L.ptr(out), stream()))
    assert torch.equal(o1, out)


@pytest.mark.gpu
def test_integration_md_stub_runs_as_written():
    """The ctypes stub of INTEGRATION.md section 2 (both code blocks, executed verbatim) computes X @ V."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "INTEGRATION.md")).read()
    sec = text[text.index("## 2. The ctypes stub"):text.index("## 3. Entry point")]
    blocks = re.findall(r"```python\n(.*?)```", sec, flags=re.S)
    assert len(blocks) == 2
    so = os.path.join(root, "pybmf_amd", "csrc", "libbmf_hip.so")
    rs = np.random.RandomState(5)
    X = (rs.rand(700, 1300) < 0.2).astype(np.float64)
    V = rs.rand(1300, 10)
    ns = {}
    exec(blocks[0].replace('"libbmf_hip.so"', repr(so)), ns)
    got = ns["x_times_v"](X, V)
    ref = X @ V
    assert np.abs(got - ref).max() <= 2e-6 * np.abs(ref).max()
    # second block: continues inside x_times_v's scope in the document; give it the same names
    m, n = X.shape
    k, kp = 10, 32
    m_pad, n_pad = 1024, 1536
    import torch
    dev = "cuda:0"
    bits = torch.zeros((m_pad, n_pad // 32), dtype=torch.int32, device=dev)
    Xd = torch.from_numpy(np.ascontiguousarray(X != 0).view(np.uint8)).to(dev)
    ns["_chk"](ns["_lib"].bmf_pack_rows_u8(Xd.data_ptr(), m, n, n, bits.data_ptr(), n_pad // 32, None))
    ns.update(dict(m=m, n=n, k=k, kp=kp, m_pad=m_pad, n_pad=n_pad, dev=dev, bits=bits, V=V))
    exec(blocks[1], ns)
    assert np.abs(ns["num"] - ref).max() <= 3e-7 * np.abs(ref).max()


@pytest.mark.gpu
@pytest.mark.parametrize("kp", [32, 64])
def test_tiled_real_matrix_kernels(env, kp):
    """bmf_tile_f32 (the layout formula of the header), bmf_xf_f32_tiled == bmf_xf_f32 bit for bit, and the residual pass over
    the tiled copy against NumPy fp64 and against the row-major kernel."""
    L, E, d = env
    rs = np.random.RandomState(11)
    rows, red, rows_pad, red_pad = 300, 1000, 384, 1024
    A = np.zeros((rows_pad, red_pad), np.float32)
    A[:rows, :red] = rs.rand(rows, red)
    Ad = dev(A, d)
    tiled = torch.empty(rows_pad * red_pad, dtype=torch.float32, device=d)
    L.check(L.lib.bmf_tile_f32(L.ptr(Ad), rows_pad, red_pad, red_pad, L.ptr(tiled), stream()))
    t = tiled.cpu().numpy().reshape(rows_pad // 64, red_pad // 64, 2, 2, 32, 8, 4)   # [tile][st][kh][rw][rl][c][e]
    want = np.empty_like(t)
    A7 = A.reshape(rows_pad // 64, 2, 32, red_pad // 64, 2, 8, 4)                     # [tile][rw][rl][st][kh][chunk][e]
    for rl in range(32):
        want[:, :, :, :, rl, :, :] = A7[:, :, rl][:, :, :, :, np.arange(8) ^ ((rl >> 1) & 7), :].transpose(0, 2, 3, 1, 4, 5)
    assert np.array_equal(t, want)

    FT = np.zeros((kp, red_pad), np.float32)
    FT[:, :red] = rs.rand(kp, red)
    FTd = dev(FT, d)
    Fd = dev(np.ascontiguousarray(FT.T), d)                 # the factor itself: red_pad x kp
    frag = torch.empty(red_pad * kp, dtype=torch.float32, device=d)
    L.check(L.lib.bmf_frag_f32(L.ptr(Fd), red_pad, kp, L.ptr(frag), stream()))
    NT = kp // 32
    fr = frag.cpu().numpy().reshape(red_pad // 64, 2, 4, NT, 2, 32, 4)       # [st][kh][u][nt][h][r][t]
    F = FT.T
    want_fr = np.empty_like(fr)
    for kh in range(2):
        for u in range(4):
            for nt in range(NT):
                for h in range(2):
                    for t in range(4):
                        want_fr[:, kh, u, nt, h, :, t] = F.reshape(red_pad // 64, 64, kp)[:, 32 * kh + 8 * u + 4 * h + t, 32 * nt:32 * nt + 32]
    assert np.array_equal(fr, want_fr)
    for splits in (1, 3, 5):
        o1 = torch.zeros((splits, rows_pad, kp), dtype=torch.float32, device=d)
        o2 = torch.full((splits, rows_pad, kp), np.nan, dtype=torch.float32, device=d)
        L.check(L.lib.bmf_xf_f32(L.ptr(Ad), rows_pad, red_pad, red_pad, L.ptr(FTd), red_pad, kp, L.ptr(o1), rows_pad * kp, splits, stream()))
        L.check(L.lib.bmf_xf_f32_tiled(L.ptr(tiled), rows_pad, red_pad, L.ptr(frag), kp, L.ptr(o2), rows_pad * kp, splits, stream()))
        assert torch.equal(o1, o2)

    # residual: X = A (rows x red real cells), U: rows_pad x kp, V: red_pad x kp, zero padded
    k = kp - 5
    U = np.zeros((rows_pad, kp), np.float32)
    V = np.zeros((red_pad, kp), np.float32)
    U[:rows, :k] = rs.rand(rows, k) * 0.2
    V[:red, :k] = rs.rand(red, k) * 0.2
    Ud, Vd = dev(U, d), dev(V, d)
    Vrf = torch.empty(red_pad * kp, dtype=torch.float32, device=d)
    L.check(L.lib.bmf_frag_rows_f32(L.ptr(Vd), red_pad, kp, L.ptr(Vrf), stream()))
    vr = Vrf.cpu().numpy().reshape(red_pad // 64, 2, kp // 8, 2, 32, 4)      # [st][cw][q][h][r][t]
    V4 = V.reshape(red_pad // 64, 2, 32, 2, kp // 8, 4)                       # [st][cw][r][h][q][t]
    assert np.array_equal(vr, V4.transpose(0, 1, 4, 3, 2, 5))
    s1 = torch.zeros(4, dtype=torch.float64, device=d)
    s2 = torch.zeros(4, dtype=torch.float64, device=d)
    L.check(L.lib.bmf_residual_sums_f32_tiled(L.ptr(tiled), rows_pad, red_pad, L.ptr(Ud), L.ptr(Vrf), kp, L.ptr(s1), stream()))
    L.check(L.lib.bmf_residual_sums_f32(L.ptr(Ad), rows_pad, red_pad, rows, red, L.ptr(Ud), L.ptr(Vd), kp, L.ptr(s2), stream()))
    R = A[:rows, :red].astype(np.float64) - U[:rows].astype(np.float64) @ V[:red].astype(np.float64).T
    got, old = s1.cpu().numpy(), s2.cpu().numpy()
    assert abs(got[0] - np.abs(R).sum()) <= 1e-6 * np.abs(R).sum(), (got[0], np.abs(R).sum())
    assert abs(got[1] - (R * R).sum()) <= 1e-6 * (R * R).sum()
    assert abs(got[0] - old[0]) <= 1e-6 * old[0] and abs(got[1] - old[1]) <= 1e-6 * old[1]
    # the sums are added to what the buffer holds
    L.check(L.lib.bmf_residual_sums_f32_tiled(L.ptr(tiled), rows_pad, red_pad, L.ptr(Ud), L.ptr(Vrf), kp, L.ptr(s1), stream()))
    assert abs(s1.cpu().numpy()[0] - 2 * got[0]) <= 1e-9 * got[0]
    # the contraction A V with the residual sums of A - U V^T folded into the same pass (what opens the next iteration of the
    # real-valued WNMF loop with A = X^T): the product bit for bit the plain tiled kernel's, the sums those of the residual pass
    if kp == 32:
        fragV = torch.empty(red_pad * kp, dtype=torch.float32, device=d)
        L.check(L.lib.bmf_frag_f32(L.ptr(Vd), red_pad, kp, L.ptr(fragV), stream()))
        Vrb = torch.empty(red_pad * kp, dtype=torch.int32, device=d)    # bf16 hi / lo pairs in row-fragment order
        L.check(L.lib.bmf_frag_rows_bf16(L.ptr(Vd), red_pad, kp, L.ptr(Vrb), stream()))
        vb = Vrb.cpu().numpy().view(np.uint16).reshape(red_pad // 64, 2, 2, 2, 2, 32, 8)     # [st][kh][ks][hi/lo][h][r][e]
        as_f32 = lambda b: (b.astype(np.uint32) << 16).view(np.float32)  # noqa: E731
        V6 = V.reshape(red_pad // 64, 2, 32, 2, 2, 8).transpose(0, 1, 3, 4, 2, 5)           # [st][kh][ks][h][r][e]
        hi, lo = as_f32(vb[:, :, :, 0]), as_f32(vb[:, :, :, 1])
        assert np.abs(hi + lo - V6).max() <= 2.0 ** -16 * np.abs(V6).max()
        for splits in (1, 3):
            o1 = torch.full((splits, rows_pad, kp), np.nan, dtype=torch.float32, device=d)
            o2 = torch.full((splits, rows_pad, kp), np.nan, dtype=torch.float32, device=d)
            s3 = torch.zeros(4, dtype=torch.float64, device=d)
            L.check(L.lib.bmf_xf_f32_tiled(L.ptr(tiled), rows_pad, red_pad, L.ptr(fragV), kp, L.ptr(o1), rows_pad * kp, splits, stream()))
            L.check(L.lib.bmf_xf_f32_tiled_resid(L.ptr(tiled), rows_pad, red_pad, L.ptr(fragV), L.ptr(Vrb), L.ptr(Ud), kp, L.ptr(o2), rows_pad * kp, splits,
                                                 L.ptr(s3), stream()))
            assert torch.equal(o1, o2)
            f = s3.cpu().numpy()
            assert abs(f[0] - got[0]) <= 1e-6 * got[0] and abs(f[1] - got[1]) <= 1e-6 * got[1], (f, got)
        # round 5: the same contractions on the bf16 matrix instruction, both operands split three ways (x = hi + mid + lo EXACTLY, by
        # truncation; six products per k-step): the fp32 product to 2^-23, checked against the fp64 product and the exact-fp32 kernel
        V3 = torch.empty(red_pad * 48, dtype=torch.int32, device=d)
        L.check(L.lib.bmf_frag_bf16x3(L.ptr(Vd), red_pad, L.ptr(V3), stream()))
        v3 = V3.cpu().numpy().view(np.uint16).reshape(red_pad // 64, 2, 2, 3, 2, 32, 8)        # [st][kh][ks][hi/mid/lo][h][r][j]
        parts = [as_f32(v3[:, :, :, q]).astype(np.float64) for q in range(3)]
        # element j of k-step ks of lane (r, h): row 64 st + 32 kh + 8 (2 ks + (j >> 2)) + 4 h + (j & 3) of the factor, column r
        V7 = V.reshape(red_pad // 64, 2, 2, 2, 2, 4, kp)                                       # [st][kh][ks][j >> 2][h][j & 3][r]
        V7 = V7.transpose(0, 1, 2, 4, 6, 3, 5).reshape(red_pad // 64, 2, 2, 2, 32, 8)          # [st][kh][ks][h][r][j]
        assert np.array_equal((parts[0] + parts[1] + parts[2]).astype(np.float32), V7)          # exact three-way split
        assert (np.abs(parts[1]) <= 2.0 ** -7 * np.abs(parts[0]) + 1e-300).all() and (np.abs(parts[2]) <= 2.0 ** -14 * np.abs(parts[0]) + 1e-300).all()
        exact = A.astype(np.float64) @ V.astype(np.float64)
        for splits in (1, 3):
            o1 = torch.full((splits, rows_pad, kp), np.nan, dtype=torch.float32, device=d)
            o3 = torch.full((splits, rows_pad, kp), np.nan, dtype=torch.float32, device=d)
            o4 = torch.full((splits, rows_pad, kp), np.nan, dtype=torch.float32, device=d)
            s4 = torch.zeros(4, dtype=torch.float64, device=d)
            L.check(L.lib.bmf_xf_f32_tiled(L.ptr(tiled), rows_pad, red_pad, L.ptr(fragV), kp, L.ptr(o1), rows_pad * kp, splits, stream()))
            L.check(L.lib.bmf_xf_f32_tiled_bf3(L.ptr(tiled), rows_pad, red_pad, L.ptr(V3), L.ptr(o3), rows_pad * kp, splits, stream()))
            L.check(L.lib.bmf_xf_f32_tiled_resid_bf3(L.ptr(tiled), rows_pad, red_pad, L.ptr(V3), L.ptr(Vrb), L.ptr(Ud), L.ptr(o4), rows_pad * kp, splits,
                                                     L.ptr(s4), stream()))
            assert torch.equal(o3, o4)
            g1, g3 = o1.double().sum(0).cpu().numpy(), o3.double().sum(0).cpu().numpy()
            scale = np.abs(A).astype(np.float64) @ np.abs(V).astype(np.float64)               # sum of |terms|: what rounding errors are relative to
            e3, e1 = float((np.abs(g3 - exact) / (scale + 1e-30)).max()), float((np.abs(g1 - exact) / (scale + 1e-30)).max())
            assert e3 <= 3e-6 and e1 <= 3e-6 and e3 <= 2.0 * e1 + 2e-7, (e3, e1)   # no worse than the exact-fp32 instruction's own accumulation
            f = s4.cpu().numpy()
            assert abs(f[0] - got[0]) <= 1e-6 * got[0] and abs(f[1] - got[1]) <= 1e-6 * got[1], (f, got)
    else:
        assert L.lib.bmf_xf_f32_tiled_resid(L.ptr(tiled), rows_pad, red_pad, L.ptr(frag), L.ptr(Vrf), L.ptr(Ud), kp, L.ptr(s1), rows_pad * kp, 1,
                                            L.ptr(s1), stream()) == -1
