"""Multiplicative updates for a rank 64 < k <= 128 (the reference has no rank limit: PyBMF/models/BinaryMFPenalty.py:32).

Every kernel of the k <= 64 path is written for one 64-bit word of factor bits per row.  A wider factor is held here as two BLOCKS
of 64 columns, F = [F_0 | F_1]; what is per column runs per block through the kernels that exist -- the bits GEMMs X V_b and
X^T U_b on the int8 digit planes, the plane builder, the fp64 element-wise update (``bmf_mu_epilogue`` with a precomputed
denominator) -- and what couples the blocks is in csrc/wide.hip: the re-associated denominator den_b = sum_b' F_b' G[b'][b]
(``bmf_fg_f32``), the cross blocks of the Gram matrices (``bmf_gram_cross``), the cover count over all 128 factors
(``bmf_cover_count_wide``) and the residual sums with the full product (``bmf_resid_sums_wide``).

Python-driven, the interface of ``engine.MaskedMUEngine`` (prepare / update / scalars, and the pipelined iterate / row): a
correctness row -- no BASELINE configuration has k > 64 -- not a tuned path.  Boolean X, one GPU.  ``WideMUEngine``: the all-ones
mask (re-associated dense path); ``WideMaskedMUEngine``: W = 'mask' / a weight matrix (contractions over the observed cells,
``bmf_masked_pass_wide``).  X_val / X_test are scored by ``engine.ObservedScorer`` / ``WholeScorer``, which take the block lists.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib as L
from ._lib import lib, check, ptr
from .engine import BitMatrix, _stream, xf_slots_i8

MAX_K_WIDE = 128
BK = 64   # columns of a block


class WideMUEngine:
    def __init__(self, X: BitMatrix, k: int, mode: int = L.MODE_PENALTY, with_mae: bool = True, thr=(0.5, 0.5)):
        if not (BK < k <= MAX_K_WIDE):
            raise NotImplementedError(f"k={k}: the two-block engine takes {BK} < k <= {MAX_K_WIDE}")
        if mode not in (L.MODE_PENALTY, L.MODE_WNMF):
            raise ValueError("mode must be MODE_PENALTY or MODE_WNMF")
        self.X, self.k, self.mode, self.with_mae, self.thr = X, int(k), int(mode), bool(with_mae), thr
        self.nb = nb = 2
        self.kp = BK   # (columns of a block: what the scorers are told; they recognise the block lists)
        self.kb = [BK, self.k - BK]          # real columns of each block
        dev = self.device = X.device
        mp, np_ = X.m_pad, X.n_pad
        self.m, self.n = X.m, X.n
        z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=dev)  # noqa: E731
        blocks = lambda rows, dt: [z((rows, BK), dt) for _ in range(nb)]  # noqa: E731
        self.U64, self.V64 = blocks(mp, torch.float64), blocks(np_, torch.float64)
        self.U, self.V = blocks(mp, torch.float32), blocks(np_, torch.float32)
        self.denU, self.denV = blocks(mp, torch.float32), blocks(np_, torch.float32)
        self.Upanel = [z((3, BK, mp), torch.int8) for _ in range(nb)]
        self.Vpanel = [z((3, BK, np_), torch.int8) for _ in range(nb)]
        self.scaleU, self.scaleV = [z((2 * BK,), torch.float32) for _ in range(nb)], [z((2 * BK,), torch.float32) for _ in range(nb)]
        self._ws = z((max(mp, np_) // 128 * BK,), torch.float32)
        with torch.cuda.device(dev):
            self.splits_xv, self.splits_xtu = xf_slots_i8(mp, np_, BK), xf_slots_i8(np_, mp, BK)
            self._tiled = X.tiled()
        self.Mslab = [z((self.splits_xv, mp, BK), torch.float32) for _ in range(nb)]
        self.Nslab = [z((self.splits_xtu, np_, BK), torch.float32) for _ in range(nb)]
        self.partU, self.partV = [z((mp // 128, 2), torch.float64) for _ in range(nb)], [z((np_ // 128, 2), torch.float64) for _ in range(nb)]
        self.ubits, self.vbits = [z((mp,), torch.int64) for _ in range(nb)], [z((np_,), torch.int64) for _ in range(nb)]
        self.ucolbits = [z((BK, mp // 32), torch.int32) for _ in range(nb)]
        self.vcolbits = [z((BK, np_ // 32), torch.int32) for _ in range(nb)]
        self.gram_blocks = int(min(256, max(1, max(mp, np_) // 256)))
        self._gslabs = z((self.gram_blocks, BK, BK), torch.float32)
        g = lambda dt: [[z((BK, BK), dt) for _ in range(nb)] for _ in range(nb)]  # noqa: E731
        self.GU, self.GV, self.GU64, self.GV64 = g(torch.float32), g(torch.float32), g(torch.float64), g(torch.float64)
        self.counts = z((2,), torch.int64)
        self.sums = z((2,), torch.float64)
        self._mae_ws = z(((mp + np_) * 2 * BK,), torch.int16) if self.with_mae else None
        self._scal = z((8,), torch.float64)
        self.sum_x = float(X.sum_local)

    # ---- factors ------------------------------------------------------------------------------------------------------------
    def load_factors(self, U0, V0):
        assert U0.shape == (self.m, self.k) and V0.shape == (self.n, self.k), (U0.shape, V0.shape)
        for F64, F, F0, rows in ((self.U64, self.U, U0, self.m), (self.V64, self.V, V0, self.n)):
            for b in range(self.nb):
                F64[b].zero_()
                F64[b][:rows, : self.kb[b]] = torch.from_numpy(np.ascontiguousarray(F0[:, BK * b: BK * b + self.kb[b]], dtype=np.float64)).to(self.device)
                F[b].copy_(F64[b])

    def factors(self):
        U = torch.cat([self.U64[b][: self.m, : self.kb[b]] for b in range(self.nb)], dim=1).cpu().numpy()
        V = torch.cat([self.V64[b][: self.n, : self.kb[b]] for b in range(self.nb)], dim=1).cpu().numpy()
        return U, V

    # ---- the pieces ---------------------------------------------------------------------------------------------------------
    def _side(self, which):
        X = self.X
        if which == "U":
            return dict(F64=self.U64, F=self.U, rows_pad=X.m_pad, rows=X.m, num=self.Mslab, splits=self.splits_xv, den=self.denU, part=self.partU,
                        rb=self.ubits, cb=self.ucolbits, thr=self.thr[0], panel=self.Upanel, scale=self.scaleU, G=self.GU, G64=self.GU64,
                        Gother=self.GV)
        return dict(F64=self.V64, F=self.V, rows_pad=X.n_pad, rows=X.n, num=self.Nslab, splits=self.splits_xtu, den=self.denV, part=self.partV,
                    rb=self.vbits, cb=self.vcolbits, thr=self.thr[1], panel=self.Vpanel, scale=self.scaleV, G=self.GV, G64=self.GV64,
                    Gother=self.GU)

    def _denominators(self, which):
        """den_b = sum_a F_a G[a][b] with G the Gram matrix of the OTHER factor, for both blocks, from the factor as it stands (all of it
        before any block is updated: the reference updates every column from the old factor, BinaryMFPenalty.py:139-148)."""
        s = self._side(which)
        st = _stream()
        for b in range(self.nb):
            for a in range(self.nb):
                check(lib.bmf_fg_f32(ptr(s["F"][a]), s["rows_pad"], ptr(s["Gother"][a][b]), BK, ptr(s["den"][b]), int(a > 0), st), "bmf_fg_f32")

    def _epilogue(self, which, b, mode, reg, with_num=True):
        s = self._side(which)
        a = L.EpilogueArgs()
        a.F64, a.F, a.rows_pad, a.rows, a.k, a.kp = s["F64"][b].data_ptr(), s["F"][b].data_ptr(), s["rows_pad"], s["rows"], self.kb[b], BK
        a.num, a.slab_stride, a.splits = (s["num"][b].data_ptr() if with_num else 0), s["rows_pad"] * BK, s["splits"]
        a.G, a.den, a.reg, a.mode, a.thr, a.terms = 0, s["den"][b].data_ptr(), float(reg), mode, float(s["thr"]), 0
        a.panel, a.ldp, a.rowbits, a.colbits, a.ldcb = 0, s["rows_pad"], s["rb"][b].data_ptr(), s["cb"][b].data_ptr(), s["rows_pad"] // 32
        a.partials, a.stop = s["part"][b].data_ptr(), 0
        check(lib.bmf_mu_epilogue(C.byref(a), _stream()), "bmf_mu_epilogue")

    def _refresh(self, which):
        """Everything derived from a factor after it changed: digit planes of both blocks, the four blocks of its Gram matrix, and the
        big contraction that uses it (X V_b for V, X^T U_b for U)."""
        s, X, st = self._side(which), self.X, _stream()
        kk = BK * BK
        for b in range(self.nb):
            check(lib.bmf_make_panel_i8(ptr(s["F64"][b]), ptr(s["F"][b]), s["rows_pad"], BK, BK, 3, ptr(s["panel"][b]), s["rows_pad"], ptr(self._ws),
                                        ptr(s["scale"][b]), st), "bmf_make_panel_i8")
        for a in range(self.nb):
            for b in range(a, self.nb):
                check(lib.bmf_gram_cross(ptr(s["F"][a]), ptr(s["F"][b]), s["rows_pad"], ptr(self._gslabs), self.gram_blocks, st), "bmf_gram_cross")
                check(lib.bmf_reduce_slabs(ptr(self._gslabs), kk, self.gram_blocks, kk, ptr(s["G"][a][b]), ptr(s["G64"][a][b]), st), "bmf_reduce_slabs")
                if a != b:   # G[b][a] = G[a][b]^T
                    s["G"][b][a].copy_(s["G"][a][b].t())
                    s["G64"][b][a].copy_(s["G64"][a][b].t())
        for b in range(self.nb):
            if which == "V":
                check(lib.bmf_xf_bits_i8(ptr(self._tiled[0]), X.m_pad, X.ldx, X.n_pad // 32, ptr(self.Vpanel[b]), X.n_pad, 3, ptr(self.scaleV[b][BK:]), BK,
                                         ptr(self.Mslab[b]), X.m_pad * BK, self.splits_xv, 1, st), "bmf_xf_bits_i8")
            else:
                check(lib.bmf_xf_bits_i8(ptr(self._tiled[1]), X.n_pad, X.ldxt, X.m_pad // 32, ptr(self.Upanel[b]), X.m_pad, 3, ptr(self.scaleU[b][BK:]), BK,
                                         ptr(self.Nslab[b]), X.n_pad * BK, self.splits_xtu, 1, st), "bmf_xf_bits_i8")

    # ---- the loop body ------------------------------------------------------------------------------------------------------
    def prepare(self):
        """Iteration-0 bookkeeping (BinaryMFPenalty.py:68-75): shadows, bits and regulariser sums of the initial factors, <U0, X V0>,
        both Gram matrices and X^T U0 for the first V update."""
        with torch.cuda.device(self.device):
            for b in range(self.nb):
                self._epilogue("V", b, L.MODE_PREPARE, 0.0, with_num=False)
            self._refresh("V")
            for b in range(self.nb):
                self._epilogue("U", b, L.MODE_PREPARE, 0.0)
            self._refresh("U")

    def update(self, reg):
        """V then U (Gauss-Seidel between the factors, every column of a factor from its old value), regulariser `reg`."""
        with torch.cuda.device(self.device):
            for which in ("V", "U"):
                self._denominators(which)
                for b in range(self.nb):
                    self._epilogue(which, b, self.mode, reg)
                self._refresh(which)

    # ---- the loop without a host round trip per iteration (round 5): iteration t + 1 is enqueued before the scalars of t are read ----
    LOG_ROWS = 8

    def can_pipeline(self):
        return True

    def iterate(self, it: int, reg: float, update: bool = True):
        """Enqueue iteration `it`: keep the current iterate, update (unless update = False: log row 0), gather the scalars into a pinned
        host row with an asynchronous copy + event.  ``row(it, reg)`` waits for that row only (the protocol of MaskedMUEngine /
        LinkMUEngine, driven by BinaryMFPenalty._fit_masked / WNMF._fit_masked)."""
        with torch.cuda.device(self.device):
            if getattr(self, "_rows_host", None) is None:
                self._rows_host = torch.zeros((self.LOG_ROWS, 8), dtype=torch.float64).pin_memory()
                self._events = [None] * self.LOG_ROWS
                self._Up = [torch.empty_like(t) for t in self.U64]
                self._Vp = [torch.empty_like(t) for t in self.V64]
            if update:
                for b in range(self.nb):
                    self._Up[b].copy_(self.U64[b], non_blocking=True)
                    self._Vp[b].copy_(self.V64[b], non_blocking=True)
                self.update(reg)
            self._gather_scalars()
            slot = it % self.LOG_ROWS
            self._rows_host[slot].copy_(self._scal, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            self._events[slot] = (it, ev)

    def row(self, it: int, reg: float):
        slot = it % self.LOG_ROWS
        if self._events[slot] is None or self._events[slot][0] != it:
            raise RuntimeError(f"row {it} is not available")
        self._events[slot][1].synchronize()
        return self._decode(self._rows_host[slot].numpy().copy(), reg)

    def previous_factors(self):
        """The iterate before the last enqueued update."""
        U = torch.cat([self._Up[b][: self.m, : self.kb[b]] for b in range(self.nb)], dim=1).cpu().numpy()
        V = torch.cat([self._Vp[b][: self.n, : self.kb[b]] for b in range(self.nb)], dim=1).cpu().numpy()
        return U, V

    def scalars(self, reg):
        """(error, rec_error, reg_error, RMSE, MAE, (TP, FP, FN, TN)) of the current state; one synchronising read.  rec_error in
        the trace form 1/2 (sum X - 2 <U, X V> + <U^T U, V^T V>) like the k <= 64 loop (api.hip::finalize_body)."""
        with torch.cuda.device(self.device):
            self._gather_scalars()
            h = self._scal.cpu().numpy()
        return self._decode(h, reg)

    def _gather_scalars(self):
        """The eight scalars of the current state into self._scal (device), nothing read back."""
        X = self.X
        if True:
            st = _stream()
            out = self._scal
            out.zero_()
            out[0] = sum(p[:, 1].sum() for p in self.partU)                      # <U, X V>
            out[1] = sum((self.GU64[a][b] * self.GV64[a][b]).sum() for a in range(self.nb) for b in range(self.nb))
            out[2] = sum(p[:, 0].sum() for p in self.partU)
            out[3] = sum(p[:, 0].sum() for p in self.partV)
            self.counts.zero_()
            check(lib.bmf_cover_count_wide(ptr(X.bits), X.m_pad, X.ldx, X.n_pad // 32, ptr(self.ubits[0]), ptr(self.ubits[1]), ptr(self.vcolbits[0]),
                                           ptr(self.vcolbits[1]), X.n_pad // 32, ptr(self.counts), st), "bmf_cover_count_wide")
            out[4:6] = self.counts.double()
            if self.with_mae:
                self.sums.zero_()
                check(lib.bmf_resid_sums_wide(ptr(self._tiled[1]), X.ldxt, X.m_pad, X.n_pad, ptr(self.U[0]), ptr(self.U[1]), ptr(self.V[0]), ptr(self.V[1]),
                                              ptr(self._mae_ws), ptr(self.sums), 1, st), "bmf_resid_sums_wide")
                out[6] = self.sums[0]

    def _decode(self, h, reg):
        cells = float(self.m) * float(self.n)
        rec = 0.5 * (self.sum_x - 2.0 * float(h[0]) + float(h[1]))
        rg = float(reg) * (0.5 * float(h[2]) + 0.5 * float(h[3])) if self.mode == L.MODE_PENALTY else 0.0
        rmse = float(np.sqrt(max(2.0 * rec, 0.0) / cells))
        mae = float(h[6]) / cells if self.with_mae else float("nan")
        tp, fp = int(h[4]), int(h[5])
        fn = int(self.sum_x) - tp
        return rec + rg, rec, rg, rmse, mae, (tp, fp, fn, self.m * self.n - tp - fp - fn)

    def residual_sums(self):
        """(sum |X - U V^T|, sum (X - U V^T)^2) of the current factors by the direct pass (one fp16 product per cell)."""
        X = self.X
        with torch.cuda.device(self.device):
            if self._mae_ws is None:
                self._mae_ws = torch.zeros(((X.m_pad + X.n_pad) * 2 * BK,), dtype=torch.int16, device=self.device)
            self.sums.zero_()
            check(lib.bmf_resid_sums_wide(ptr(self._tiled[1]), X.ldxt, X.m_pad, X.n_pad, ptr(self.U[0]), ptr(self.U[1]), ptr(self.V[0]), ptr(self.V[1]),
                                          ptr(self._mae_ws), ptr(self.sums), 1, _stream()), "bmf_resid_sums_wide")
            s = self.sums.cpu().numpy()
        return float(s[0]), float(s[1])


class WideMaskedMUEngine:
    """The masked updates (W = 'mask' or a weight matrix: PyBMF/models/BinaryMFPenalty.py:139-142,154-157, WNMF.py:98-106) for a
    rank 64 < k <= 128: ``engine.MaskedMUEngine`` with every factor as two 64-column blocks.  The pass over the observed cells takes
    the dot product over both blocks and leaves numerators / denominators per block (``bmf_masked_pass_wide``); the fp64 element-wise
    update runs per block (the denominators already carry the coupling).  Whole-matrix scores of a Boolean X (task =
    'reconstruction') from ``bmf_resid_sums_wide`` and ``bmf_cover_count_wide``.  Stepped from Python, one read-back per iteration;
    one GPU; no link."""

    def __init__(self, obs, k: int, mode: int, bits: BitMatrix = None, with_mae: bool = True, thr=(0.5, 0.5)):
        if not (BK < k <= MAX_K_WIDE):
            raise NotImplementedError(f"k={k}: the two-block engine takes {BK} < k <= {MAX_K_WIDE}")
        if mode not in (L.MODE_PENALTY, L.MODE_WNMF):
            raise ValueError("mode must be MODE_PENALTY or MODE_WNMF")
        from .engine import round_up
        self.obs, self.k, self.mode, self.bits, self.with_mae, self.thr = obs, int(k), int(mode), bits, bool(with_mae), thr
        self.nb, self.kp = 2, BK
        self.kb = [BK, self.k - BK]
        dev = self.device = obs.device
        self.m, self.n = obs.m, obs.n
        mp, np_ = self.m_pad, self.n_pad = round_up(max(self.m, 1), L.ROW_PAD), round_up(self.n, L.ROW_PAD)
        if bits is not None:
            assert (bits.m_pad, bits.n_pad) == (mp, np_)
        self.sum_x = float(bits.sum_local) if bits is not None else None
        z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=dev)  # noqa: E731
        blocks = lambda rows, dt: [z((rows, BK), dt) for _ in range(2)]  # noqa: E731
        self.U64, self.V64 = blocks(mp, torch.float64), blocks(np_, torch.float64)
        self.U, self.V = blocks(mp, torch.float32), blocks(np_, torch.float32)
        self.numU, self.denU, self.numV, self.denV = blocks(mp, torch.float32), blocks(mp, torch.float32), blocks(np_, torch.float32), blocks(np_, torch.float32)
        self.partU, self.partV = [z((mp // 128, 2), torch.float64) for _ in range(2)], [z((np_ // 128, 2), torch.float64) for _ in range(2)]
        self.ubits, self.vbits = [z((mp,), torch.int64) for _ in range(2)], [z((np_,), torch.int64) for _ in range(2)]
        self.ucolbits = [z((BK, mp // 32), torch.int32) for _ in range(2)]
        self.vcolbits = [z((BK, np_ // 32), torch.int32) for _ in range(2)]
        self.sums, self.sums2 = z((4,), torch.float64), z((2,), torch.float64)
        self.counts = z((2,), torch.int64)
        self._scal = z((8,), torch.float64)
        self._mae_ws = None
        self._tiled_t = None

    def load_factors(self, U0, V0):
        assert U0.shape == (self.m, self.k) and V0.shape == (self.n, self.k), (U0.shape, V0.shape)
        for F64, F, F0, rows in ((self.U64, self.U, U0, self.m), (self.V64, self.V, V0, self.n)):
            for b in range(2):
                F64[b].zero_()
                F64[b][:rows, : self.kb[b]] = torch.from_numpy(np.ascontiguousarray(F0[:, BK * b: BK * b + self.kb[b]], dtype=np.float64)).to(self.device)
                F[b].copy_(F64[b])

    def factors(self):
        U = torch.cat([self.U64[b][: self.m, : self.kb[b]] for b in range(2)], dim=1).cpu().numpy()
        V = torch.cat([self.V64[b][: self.n, : self.kb[b]] for b in range(2)], dim=1).cpu().numpy()
        return U, V

    def _pass(self, ls, rows, Fself, Fother, num, den, sums):
        if sums is not None:
            sums.zero_()
        if ls.get("part_wide") is None:
            ls["part_wide"] = [torch.zeros((max(ls["nseg"], 1), 2, BK), dtype=torch.float32, device=self.device) for _ in range(2)]
        p0, p1 = ls["part_wide"]
        check(lib.bmf_masked_pass_wide(ptr(ls["ptr"]), ptr(ls["idx"]), ptr(ls["val"]), ptr(ls["wgt"]), rows, ptr(ls["seg_row"]), ptr(ls["seg_beg"]),
                                       ls["nseg"], ptr(ls["row_seg_ptr"]), ptr(Fself[0]), ptr(Fself[1]), ptr(Fother[0]), ptr(Fother[1]), ptr(p0), ptr(p1),
                                       ptr(num[0]), ptr(num[1]), ptr(den[0]), ptr(den[1]), ptr(sums), _stream()), "bmf_masked_pass_wide")

    def _epilogue(self, which, b, mode, reg):
        if which == "V":
            F64, F, rows_pad, rows, num, den = self.V64, self.V, self.n_pad, self.n, self.numV, self.denV
            rb, cb, part, thr = self.vbits, self.vcolbits, self.partV, self.thr[1]
        else:
            F64, F, rows_pad, rows, num, den = self.U64, self.U, self.m_pad, self.m, self.numU, self.denU
            rb, cb, part, thr = self.ubits, self.ucolbits, self.partU, self.thr[0]
        a = L.EpilogueArgs()
        a.F64, a.F, a.rows_pad, a.rows, a.k, a.kp = F64[b].data_ptr(), F[b].data_ptr(), rows_pad, rows, self.kb[b], BK
        a.num, a.slab_stride, a.splits = (0 if mode == L.MODE_PREPARE else num[b].data_ptr()), rows_pad * BK, 1
        a.G, a.den, a.reg, a.mode, a.thr, a.terms = 0, den[b].data_ptr(), float(reg), mode, float(thr), 0
        a.panel, a.ldp, a.rowbits, a.colbits, a.ldcb = 0, rows_pad, rb[b].data_ptr(), cb[b].data_ptr(), rows_pad // 32
        a.partials, a.stop = part[b].data_ptr(), 0
        check(lib.bmf_mu_epilogue(C.byref(a), _stream()), "bmf_mu_epilogue")

    def prepare(self):
        """Shadows, bits and regulariser partials of the initial factors; numerators of the first V update + rec_error."""
        with torch.cuda.device(self.device):
            for which in ("V", "U"):
                for b in range(2):
                    self._epilogue(which, b, L.MODE_PREPARE, 0.0)
            self._pass(self.obs.csc, self.n, self.V, self.U, self.numV, self.denV, self.sums)

    def update(self, reg):
        """V then U (Gauss-Seidel between the factors; both blocks of a factor from the same pass, i.e. from its old value), then the
        pass that prepares the next V update and measures rec_error of the new state."""
        with torch.cuda.device(self.device):
            for b in range(2):
                self._epilogue("V", b, self.mode, reg)
            self._pass(self.obs.csr, self.m, self.U, self.V, self.numU, self.denU, None)
            for b in range(2):
                self._epilogue("U", b, self.mode, reg)
            self._pass(self.obs.csc, self.n, self.V, self.U, self.numV, self.denV, self.sums)

    def scalars(self, reg):
        """(error, rec_error, reg_error, RMSE, MAE, (TP, FP, FN, TN) or None): rec_error over the observed cells (0.5 sum W o (X - U V^T)^2),
        RMSE / MAE / counts over the whole matrix (task='reconstruction') when the Boolean matrix was given."""
        cells = float(self.m) * float(self.n)
        with torch.cuda.device(self.device):
            out = self._scal
            out.zero_()
            out[0] = self.sums[0]
            out[1] = sum(p[:, 0].sum() for p in self.partU)
            out[2] = sum(p[:, 0].sum() for p in self.partV)
            if self.bits is not None:
                B, st = self.bits, _stream()
                if self._mae_ws is None:
                    self._mae_ws = torch.zeros(((B.m_pad + B.n_pad) * 2 * BK,), dtype=torch.int16, device=self.device)
                    self._tiled_t = B.tiled()[1]
                self.sums2.zero_()
                check(lib.bmf_resid_sums_wide(ptr(self._tiled_t), B.ldxt, B.m_pad, B.n_pad, ptr(self.U[0]), ptr(self.U[1]), ptr(self.V[0]), ptr(self.V[1]),
                                              ptr(self._mae_ws), ptr(self.sums2), 1, st), "bmf_resid_sums_wide")
                self.counts.zero_()
                check(lib.bmf_cover_count_wide(ptr(B.bits), B.m_pad, B.ldx, B.n_pad // 32, ptr(self.ubits[0]), ptr(self.ubits[1]), ptr(self.vcolbits[0]),
                                               ptr(self.vcolbits[1]), B.n_pad // 32, ptr(self.counts), st), "bmf_cover_count_wide")
                out[3:5] = self.sums2
                out[5:7] = self.counts.double()
            h = out.cpu().numpy()
        rec = 0.5 * float(h[0])
        rg = float(reg) * (0.5 * float(h[1]) + 0.5 * float(h[2])) if self.mode == L.MODE_PENALTY else 0.0
        rmse = mae = float("nan")
        counts = None
        if self.bits is not None:
            rmse, mae = float(np.sqrt(h[4] / cells)), float(h[3] / cells)
            tp, fp = int(h[5]), int(h[6])
            fn = int(self.sum_x) - tp
            counts = (tp, fp, fn, self.m * self.n - tp - fp - fn)
        return rec + rg, rec, rg, rmse, mae, counts
