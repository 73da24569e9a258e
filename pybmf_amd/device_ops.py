"""One-shot device operations behind the reference's module-level helpers (update_U / update_V / error,
get_prediction*, metrics) -- each call uploads its operands, runs the HIP kernels and returns host objects.

These exist so that code written against ``PyBMF.models.BinaryMFPenalty.update_U`` & co. keeps working; the fitted
models never round-trip like this (their loop stays on the device, see engine.MUEngine).
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch
from scipy.sparse import csr_matrix

from . import _lib as L
from ._lib import lib, check, ptr
from .engine import BitMatrix, MaskedMUEngine, MUEngine, SparseObs, _stream, require_gpu, round_up

DEFAULT_DEVICE = "cuda:0"


def _bits_of(Fb: np.ndarray, rows_pad: int):
    """Boolean factor (rows x k, k <= 64) -> (rowbits int64[rows_pad], colbits int32[kp][rows_pad/32]) on the host."""
    rows, k = Fb.shape
    kp = 32 if k <= 32 else 64
    w = (1 << np.arange(k, dtype=np.uint64))
    rowbits = np.zeros(rows_pad, np.uint64)
    rowbits[:rows] = (Fb.astype(np.uint64) * w).sum(1)
    col = np.zeros((kp, rows_pad), np.uint8)
    col[:k, :rows] = Fb.T
    colbits = np.packbits(col, axis=1, bitorder="little").view(np.int32)
    return rowbits.view(np.int64), colbits, kp


def _threshold(F, t, ts):
    F = np.asarray(F)
    if ts is not None:
        assert len(ts) == F.shape[1]
        return F > np.asarray(ts)[None, :]
    if t is not None:
        return F > t
    return F != 0


def boolean_product_bits(Ub: np.ndarray, Vb: np.ndarray, device=DEFAULT_DEVICE) -> torch.Tensor:
    """Bits of min(1, Ub @ Vb^T) on the device: int32 [m_pad][n_pad/32]."""
    dev = require_gpu(device)
    m, k = Ub.shape
    n = Vb.shape[0]
    if k > L.MAX_KP:   # OR of the products of the 64-column blocks (the kernel takes one 64-bit word of factor bits per row)
        out = None
        for c0 in range(0, k, L.MAX_KP):
            blk = boolean_product_bits(Ub[:, c0:c0 + L.MAX_KP], Vb[:, c0:c0 + L.MAX_KP], device)
            out = blk if out is None else out.bitwise_or_(blk)
        return out
    m_pad, n_pad = round_up(m, 64), round_up(n, 128)
    rb, _, kp = _bits_of(Ub, m_pad)
    _, cb, _ = _bits_of(Vb, n_pad)
    with torch.cuda.device(dev):
        rbd, cbd = torch.from_numpy(rb).to(dev), torch.from_numpy(np.ascontiguousarray(cb)).to(dev)
        out = torch.zeros((m_pad, n_pad // 32), dtype=torch.int32, device=dev)
        check(lib.bmf_boolean_product_bits(ptr(rbd), m_pad, ptr(cbd), n_pad // 32, kp, n_pad // 32, ptr(out), n_pad // 32,
                                           _stream()), "bmf_boolean_product_bits")
        torch.cuda.synchronize()
    return out


def bits_to_csr(bits: torch.Tensor, m: int, n: int, dtype=np.int64) -> csr_matrix:
    b = bits[:m].cpu().numpy().view(np.uint8)
    return csr_matrix(np.unpackbits(b, axis=1, bitorder="little")[:, :n].astype(dtype))


def boolean_product_csr(U, V, u=None, v=None, us=None, vs=None, device=DEFAULT_DEVICE) -> csr_matrix:
    Ub, Vb = _threshold(U, u, us), _threshold(V, v, vs)
    return bits_to_csr(boolean_product_bits(Ub, Vb, device), Ub.shape[0], Vb.shape[0])


def real_product(U, V, device=DEFAULT_DEVICE) -> np.ndarray:
    """U @ V^T as a dense fp64 host array, computed by the exact-fp32 MFMA product kernel (bmf_real_product)."""
    dev = require_gpu(device)
    U, V = np.asarray(U), np.asarray(V)
    m, k = U.shape
    n = V.shape[0]
    if k > L.MAX_KP:   # sum of the products of the 64-column blocks
        return sum(real_product(U[:, c0:c0 + L.MAX_KP], V[:, c0:c0 + L.MAX_KP], device) for c0 in range(0, k, L.MAX_KP))
    kp = 32 if k <= 32 else 64
    m_pad, n_pad = round_up(m, 128), round_up(n, 32)
    with torch.cuda.device(dev):
        Ud = torch.zeros((m_pad, kp), dtype=torch.float32, device=dev)
        Vd = torch.zeros((n_pad, kp), dtype=torch.float32, device=dev)
        Ud[:m, :k] = torch.from_numpy(np.ascontiguousarray(U, dtype=np.float32)).to(dev)
        Vd[:n, :k] = torch.from_numpy(np.ascontiguousarray(V, dtype=np.float32)).to(dev)
        out = torch.empty((m, n), dtype=torch.float32, device=dev)
        check(lib.bmf_real_product(ptr(Ud), m_pad, m, ptr(Vd), n_pad, n, kp, ptr(out), n, _stream()), "bmf_real_product")
        return out.cpu().numpy().astype(np.float64)


def product_csr(U, V, boolean=True, device=DEFAULT_DEVICE) -> csr_matrix:
    """U @ V^T as csr: Boolean product of the given 0/1 factors, or the real-valued product (end-of-fit materialisation
    of ``X_pd`` that the reference API promises; not part of the iteration)."""
    if boolean:
        return boolean_product_csr(U, V, device=device)
    return csr_matrix(real_product(U, V, device))


def confusion_counts(X_gt, X_pd, device=DEFAULT_DEVICE):
    """(TP, FP, FN, TN) of two Boolean matrices of equal shape, whole matrix, on the device
    (utils/metrics.py:56-77 for task='reconstruction')."""
    dev = require_gpu(device)
    G, P = BitMatrix(X_gt, dev), BitMatrix(X_pd, dev)
    assert (G.m, G.n) == (P.m, P.n)
    with torch.cuda.device(dev):
        both = G.bits & P.bits
        only_p = P.bits & ~G.bits
        cnt = torch.zeros(2, dtype=torch.int64, device=dev)
        check(lib.bmf_popcount(ptr(both), G.m_pad, G.ldx, G.ldx, ptr(cnt[0:1]), _stream()), "bmf_popcount")
        check(lib.bmf_popcount(ptr(only_p), G.m_pad, G.ldx, G.ldx, ptr(cnt[1:2]), _stream()), "bmf_popcount")
        tp, fp = (int(x) for x in cnt.cpu().numpy())
    fn = G.sum_local - tp
    tn = G.m * G.n - tp - fp - fn
    return tp, fp, fn, tn


def weighted_sqdiff(X_gt, X_pd, W=None, device=DEFAULT_DEVICE, chunk_cells: int = 1 << 24) -> float:
    """sum(W o (X_gt - X_pd)^2) for explicit m x n matrices (ndarray / np.matrix / scipy sparse), fp64 on the device
    (bmf_sqdiff_sum), fed in row chunks.  W: None = all ones."""
    from .engine import require_gpu
    dev = require_gpu(device)
    m, n = X_gt.shape
    if tuple(X_pd.shape) != (m, n) or (W is not None and tuple(W.shape) != (m, n)):
        raise ValueError("X_gt, X_pd and W must have the same shape")

    def rows(A, a, b):
        blk = A[a:b]
        blk = blk.toarray() if hasattr(blk, "toarray") else np.asarray(blk)
        return torch.from_numpy(np.ascontiguousarray(blk, dtype=np.float64)).to(dev)

    step = max(1, chunk_cells // max(n, 1))
    with torch.cuda.device(dev):
        out = torch.zeros(1, dtype=torch.float64, device=dev)
        work = torch.zeros(int(lib.bmf_sqdiff_work()), dtype=torch.float64, device=dev)
        for a in range(0, m, step):
            b = min(a + step, m)
            A, B = rows(X_gt, a, b), rows(X_pd, a, b)
            Wc = rows(W, a, b) if W is not None else None
            check(lib.bmf_sqdiff_sum(ptr(A), ptr(B), ptr(Wc), A.numel(), ptr(work), ptr(out), _stream()), "bmf_sqdiff_sum")
        return float(out.item())


class OneStep:
    """X, U, V on the device for a single multiplicative update / error evaluation with an explicit `reg`."""

    def __init__(self, X, U, V, mode=L.MODE_PENALTY, device=DEFAULT_DEVICE, terms=3):
        U, V = np.asarray(U, dtype=np.float64), np.asarray(V, dtype=np.float64)
        self.B = BitMatrix(X, device)
        self.eng = MUEngine(self.B, k=U.shape[1], mode=mode, terms=terms, with_mae=False, tol=-1.0, min_diff=-1.0, max_iter=4)
        self.eng.load_factors(U, V)

    def _epilogue(self, which: str, reg: float):
        e, st = self.eng, self.eng.st
        a = L.EpilogueArgs()
        if which == "V":
            F64, F, rows_pad, rows, num, splits, G = e.V64, e.V, st.n_pad, st.n, e.Nslab, e.splits_xtu, e.GU
            panel, rb, cb, ldcb, part, thr = e.Vpanel, e.vbits, e.vcolbits, st.ldvc, e.partV, st.thr_v
        else:
            F64, F, rows_pad, rows, num, splits, G = e.U64, e.U, st.m_pad, st.m, e.Mslab, e.splits_xv, e.GV
            panel, rb, cb, ldcb, part, thr = e.Upanel, e.ubits, e.ucolbits, st.lduc, e.partU, st.thr_u
        a.F64, a.F, a.rows_pad, a.rows, a.k, a.kp = F64.data_ptr(), F.data_ptr(), rows_pad, rows, e.k, e.kp
        a.num, a.slab_stride, a.splits = num.data_ptr(), rows_pad * e.kp, splits
        a.G, a.reg, a.mode, a.thr, a.terms = G.data_ptr(), float(reg), e.mode, thr, e.terms
        a.panel, a.ldp, a.rowbits, a.colbits, a.ldcb = panel.data_ptr(), rows_pad, rb.data_ptr(), cb.data_ptr(), ldcb
        a.partials, a.stop = part.data_ptr(), 0
        check(lib.bmf_mu_epilogue(C.byref(a), _stream()), "bmf_mu_epilogue")

    def _gram(self, F, rows_pad, out32):
        e = self.eng
        kk = e.kp * e.kp
        check(lib.bmf_gram_partial(ptr(F), rows_pad, e.kp, e.kp, ptr(e.gram_slabs), e.gram_blocks, _stream()), "bmf_gram_partial")
        check(lib.bmf_reduce_slabs(ptr(e.gram_slabs), kk, e.gram_blocks, kk, ptr(out32), None, _stream()), "bmf_reduce_slabs")

    def _panel(self, F, rows_pad, panel):
        e = self.eng
        check(lib.bmf_make_panel(ptr(F), rows_pad, e.kp, e.kp, e.terms, ptr(panel), rows_pad, _stream()), "bmf_make_panel")

    def update_V(self, reg):
        e, st, B = self.eng, self.eng.st, self.B
        self._panel(e.U, st.m_pad, e.Upanel)
        self._gram(e.U, st.m_pad, e.GU)
        check(lib.bmf_xf_bits(ptr(B.bits_t), st.n_pad, B.ldxt, st.m_pad // 32, ptr(e.Upanel), st.m_pad, e.terms, e.kp, ptr(e.Nslab),
                              st.n_pad * e.kp, e.splits_xtu, _stream()), "bmf_xf_bits")
        self._epilogue("V", reg)
        return e.factors()[1]

    def update_U(self, reg):
        e, st, B = self.eng, self.eng.st, self.B
        self._panel(e.V, st.n_pad, e.Vpanel)
        self._gram(e.V, st.n_pad, e.GV)
        check(lib.bmf_xf_bits(ptr(B.bits), st.m_pad, B.ldx, st.n_pad // 32, ptr(e.Vpanel), st.n_pad, e.terms, e.kp, ptr(e.Mslab),
                              st.m_pad * e.kp, e.splits_xv, _stream()), "bmf_xf_bits")
        self._epilogue("U", reg)
        return e.factors()[0]

    def errors(self, reg):
        """(error, rec_error, reg_error) for the current factors."""
        e = self.eng
        e.log.zero_()
        e.stop.zero_()
        e.prepare(float(reg))
        log, _ = e.read_log()
        return float(log[0, L.LOG_ERROR]), float(log[0, L.LOG_REC]), float(log[0, L.LOG_REGERR])

    def residual_sums(self):
        """(sum |X - U V^T|, sum (X - U V^T)^2) over the real cells."""
        e, B = self.eng, self.B
        sums = torch.zeros(4, dtype=torch.float64, device=B.device)
        check(lib.bmf_residual_sums(ptr(B.bits), B.m_pad, B.ldx, B.m, B.n, ptr(e.U), ptr(e.V), e.kp, ptr(sums), None, _stream()),
              "bmf_residual_sums")
        s = sums.cpu().numpy()
        return float(s[0]), float(s[1])


class MaskedOneStep:
    """The same single steps under a mask / weight matrix W (the m x n product cannot be re-associated: SDDMM + SpMM over the
    cells with W != 0, csrc/masked.hip).  W and X are host matrices of equal shape."""

    def __init__(self, X, W, U, V, mode=L.MODE_PENALTY, device=DEFAULT_DEVICE, link=0, lamda=10.0):
        from scipy.sparse import coo_matrix, issparse
        U, V = np.asarray(U, dtype=np.float64), np.asarray(V, dtype=np.float64)
        Wc = coo_matrix(W)
        Wc.eliminate_zeros()
        if Wc.shape != tuple(X.shape):
            raise ValueError("W must have the shape of X")
        Xs = X.tocsr() if issparse(X) else np.asarray(X)
        vals = np.asarray(Xs[Wc.row, Wc.col]).ravel()
        obs = SparseObs(Wc.row, Wc.col, vals, Wc.data, X.shape, device)
        bits = BitMatrix(X, device) if link else None   # (the link engine scores against the Boolean X)
        self.eng = MaskedMUEngine(obs, U.shape[1], mode, bits=bits, link=link, lamda=lamda)
        self.eng.load_factors(U, V)

    def update_U(self, reg):
        e = self.eng
        with torch.cuda.device(e.device):
            e._pass(e.obs.csr, e.m, e.U, e.V, e.numU, e.denU, None)
            e._epilogue("U", e.mode, float(reg))
        return e.factors()[0]

    def update_V(self, reg):
        e = self.eng
        with torch.cuda.device(e.device):
            e._pass(e.obs.csc, e.n, e.V, e.U, e.numV, e.denV, None)
            e._epilogue("V", e.mode, float(reg))
        return e.factors()[1]

    def errors(self, reg):
        """(error, rec_error, reg_error) for the current factors."""
        self.eng.prepare()
        return self.eng.scalars(float(reg))[:3]
