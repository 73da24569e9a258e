"""Device state and step driver of the proximal (PALM / iPALM) models ELBMF and PRIMP.

The gradient of 1/2 ||X - U V^T||_F^2 in U is U (V^T V) - X V: the bits GEMM and the k x k Gram of the multiplicative-update
path give both; the step itself is ``bmf_palm_epilogue`` (csrc/palm.hip) with the step size from ``bmf_sym_norms`` -- nothing
visits the host inside a step.  The loop is driven from Python with one read-back per iteration (the stopping rules of
PyBMF/models/ELBMF.py:157-160 and PRIMP.py:128-129 look at a scalar every iteration).

Boolean X.  Under the all-ones mask ``multiply(W, U V^T - X) V`` re-associates to ``U (V^T V) - X V``; with a mask / weight matrix
(``obs``: engine.SparseObs, the cells with W != 0) the two contractions run over the observed cells only (``bmf_masked_pass`` at the
extrapolated point) and the epilogue takes their difference (ELBMF.py:190).
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np
import torch

from . import _lib as L
from ._lib import lib, check, ptr
from .engine import BitMatrix, _stream, xf_slots, xf_slots_i8


class PalmEngine:
    def __init__(self, X: BitMatrix, k: int, variant: int, beta: float = 0.0, thr=(0.5, 0.5), obs=None, panel: str = "i8"):
        """panel: operand format of the two bits GEMMs -- 'i8' (three int8 digit planes on the integer MFMA, exact accumulation: the
        format of the multiplicative-update loop) or 'f16' (two fp16 addends, fp32 accumulation: round 1's)."""
        if panel not in ("i8", "f16"):
            raise ValueError(f"panel={panel!r}: 'i8' or 'f16'")
        self.panel = panel
        if not (1 <= k <= L.MAX_KP):
            raise NotImplementedError(f"k={k}: this build supports 1 <= k <= {L.MAX_KP}")
        self.X, self.k, self.variant, self.beta, self.thr = X, int(k), int(variant), float(beta), thr
        self.norm_kind = L.NORM_SPECTRAL if variant == L.PALM_ELBMF else L.NORM_FROBENIUS
        self.kp = kp = 32 if k <= 32 else 64
        dev = self.device = X.device
        mp, np_ = X.m_pad, X.n_pad
        z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=dev)  # noqa: E731
        self.U64, self.V64, self.Up64, self.Vp64 = z((mp, kp), torch.float64), z((np_, kp), torch.float64), z((mp, kp), torch.float64), z((np_, kp), torch.float64)
        self.U, self.V = z((mp, kp), torch.float32), z((np_, kp), torch.float32)
        if panel == "i8":
            self.Upanel, self.Vpanel = z((3, kp, mp), torch.int8), z((3, kp, np_), torch.int8)
        else:
            self.Upanel, self.Vpanel = z((2, kp, mp), torch.int16), z((2, kp, np_), torch.int16)
        self.scaleU, self.scaleV = z((4 * kp,), torch.float32), z((4 * kp,), torch.float32)   # (2 kp used by the stand-alone builder, 4 kp by the predicted-scale step)
        self.wsU, self.wsV = z((mp // 128 * kp,), torch.float32), z((np_ // 128 * kp,), torch.float32)
        with torch.cuda.device(dev):
            if panel == "i8":
                self.splits_xv, self.splits_xtu = xf_slots_i8(mp, np_, kp), xf_slots_i8(np_, mp, kp)
                self._tiled = X.tiled()   # (X, X^T) in the layout the int8 GEMM streams best
            else:
                self.splits_xv, self.splits_xtu = xf_slots(mp, np_, 2, kp), xf_slots(np_, mp, 2, kp)
        self.Mslab, self.Nslab = z((self.splits_xv, mp, kp), torch.float32), z((self.splits_xtu, np_, kp), torch.float32)
        self.gram_blocks = int(min(256, max(1, max(mp, np_) // 256)))
        self.gram_slabs = z((self.gram_blocks, kp, kp), torch.float32)
        self.GU, self.GV = z((kp, kp), torch.float32), z((kp, kp), torch.float32)
        self.GU64, self.GV64 = z((kp * kp,), torch.float64), z((kp * kp,), torch.float64)
        self.normsU, self.normsV = z((2,), torch.float64), z((2,), torch.float64)
        self.partU, self.partV = z((mp // 128,), torch.float64), z((np_ // 128,), torch.float64)
        self.dot_blocks = max(1024, mp // 128)   # (also holds the per-block <U, X V> partials of the plane-emitting U step)
        self.dotpart = z((self.dot_blocks,), torch.float64)
        self.ubits, self.vbits = z((mp,), torch.int64), z((np_,), torch.int64)
        self.ucolbits, self.vcolbits = z((kp, mp // 32), torch.int32), z((kp, np_ // 32), torch.int32)
        self.counts = z((4,), torch.int64)
        self._scal = z((8,), torch.float64)
        self.sum_x = float(X.sum_local)
        # mask / weight matrix: observed-cell lists + (num, den) of the masked pass and the extrapolated point as fp32
        self.obs = obs
        if obs is not None:
            assert (obs.m, obs.n) == (X.m, X.n)
            self.numU, self.denU, self.numV, self.denV = (z((r, kp), torch.float32) for r in (mp, mp, np_, np_))
            self.FeU, self.FeV = z((mp, kp), torch.float32), z((np_, kp), torch.float32)
            self._grad_ready = {"U": False, "V": False}

    # ---- the ELBMF loop body as ONE C call per iteration (bmf_palm_iterate), and the scalars of an iteration read back late ----
    LOG_ROWS = 8

    def _state(self):
        """bmf_palm_state over this engine's buffers (all-ones mask, int8 operands)."""
        if getattr(self, "_st", None) is not None:
            return self._st
        if self.obs is not None or self.panel != "i8":
            raise NotImplementedError("bmf_palm_iterate / bmf_primp_iterate: the loops under the all-ones mask on the int8 operands")
        X, kp, dev = self.X, self.kp, self.device
        with torch.cuda.device(dev):   # the log rows are written by the scalars kernel straight into pinned host memory: no copy in the stream
            self._log_host = torch.zeros((self.LOG_ROWS, 8), dtype=torch.float64).pin_memory()
        st = L.PalmState()
        st.struct_bytes = C.sizeof(L.PalmState)
        st.m, st.n, st.k, st.kp, st.variant, st.norm_kind = X.m, X.n, self.k, kp, self.variant, self.norm_kind
        st.splits_xv, st.splits_xtu, st.gram_blocks, st.dot_blocks, st.log_rows = self.splits_xv, self.splits_xtu, self.gram_blocks, self.dot_blocks, self.LOG_ROWS
        st.m_pad, st.n_pad, st.Xbits, st.ldx = X.m_pad, X.n_pad, X.bits.data_ptr(), X.ldx
        st.Xtiled, st.XTtiled = self._tiled[0].data_ptr(), self._tiled[1].data_ptr()
        for name in ("U64", "V64", "Up64", "Vp64", "U", "V", "Upanel", "Vpanel", "scaleU", "scaleV", "wsU", "wsV", "Mslab", "Nslab", "gram_slabs",
                     "GU", "GV", "GU64", "GV64", "normsU", "normsV", "partU", "partV", "dotpart", "ubits", "vbits", "ucolbits", "vcolbits", "counts"):
            setattr(st, name, getattr(self, name).data_ptr())
        st.log = self._log_host.data_ptr()
        st.beta, st.thr_u, st.thr_v = self.beta, float(self.thr[0]), float(self.thr[1])
        self._st = st
        self._events = [None] * self.LOG_ROWS
        self._last = -1
        self._lag = lib.bmf_palm_row_lag(C.byref(st)) if self.variant == L.PALM_ELBMF else 0
        if self._lag < 0:
            check(self._lag, "bmf_palm_row_lag")
        return st

    def can_pipeline(self):
        """Can the loop run as one C call per iteration with its scalars read one iteration late?"""
        return self.obs is None and self.panel == "i8"

    # ---- PRIMP's loop body as ONE C call per iteration (bmf_primp_iterate) ----
    def primp_iterate(self, it: int, l1: float, l2: float):
        """Enqueue iteration `it` of PRIMP's loop (U step, what derives from the new U, V step, what derives from the new V, the two
        sums of the objective into pinned host memory); ``primp_row(it)`` waits for that row only."""
        st = self._state()
        with torch.cuda.device(self.device):
            check(lib.bmf_primp_iterate(C.byref(st), int(it), float(l1), float(l2), _stream()), "bmf_primp_iterate")
            self._mark(it)
        self._last = it

    def primp_row(self, it: int) -> float:
        """||X - U V^T||_F^2 after iteration `it`."""
        slot = it % self.LOG_ROWS
        if self._events[slot] is None or self._events[slot][0] != it:
            raise RuntimeError(f"row {it} is not available (last iteration enqueued: {self._last})")
        self._events[slot][1].synchronize()
        return self._decode(self._log_host[slot].numpy().copy(), False)[0]

    def keep(self):
        """Snapshot the current factors (device copies, in stream order): what a loop that has already enqueued the next iteration
        returns when its stopping rule fires on this one."""
        if getattr(self, "_Uk", None) is None:
            self._Uk, self._Vk = torch.empty_like(self.U64), torch.empty_like(self.V64)
        with torch.cuda.device(self.device):
            self._Uk.copy_(self.U64, non_blocking=True)
            self._Vk.copy_(self.V64, non_blocking=True)

    def kept_factors(self):
        X = self.X
        return self._Uk[: X.m, : self.k].cpu().numpy(), self._Vk[: X.n, : self.k].cpu().numpy()

    def iterate(self, it: int, l1: float, l2: float, gap_l1: float, gap_l2: float):
        """Enqueue iteration `it` of ELBMF's loop (both steps, everything derived, the log row, which lands in pinned host memory);
        ``row(it)`` waits for that row only.  With beta = 0 the cross term <U, X V> of row `it` falls out of the U step of iteration
        `it + 1` (bmf_palm_row_lag): the row is then complete after that step -- or after ``row(it)`` has asked for it explicitly when no
        further iteration was enqueued.  At most LOG_ROWS - 2 iterations may be outstanding."""
        st = self._state()
        args = (C.byref(st), int(it), float(l1), float(l2), float(gap_l1), float(gap_l2))
        with torch.cuda.device(self.device):
            s = _stream()
            check(lib.bmf_palm_iterate(*args, 1, s), "bmf_palm_iterate")
            if self._lag and it > 0:
                self._mark(it - 1)
            check(lib.bmf_palm_iterate(*args, 2, s), "bmf_palm_iterate")
            if not self._lag:
                self._mark(it)
        self._last = it

    def _mark(self, it):
        ev = torch.cuda.Event()
        ev.record()
        self._events[it % self.LOG_ROWS] = (it, ev)

    def row(self, it: int):
        """(err, U gap, V gap, (TP, FP, FN, TN)) of iteration `it`, as ``scalars`` returns them."""
        slot = it % self.LOG_ROWS
        if self._events[slot] is None or self._events[slot][0] != it:
            if not (self._lag and it == self._last):
                raise RuntimeError(f"row {it} is not available (last iteration enqueued: {self._last})")
            with torch.cuda.device(self.device):   # the last iteration: no later U step will complete its row
                check(lib.bmf_palm_finish_row(C.byref(self._st), int(it), _stream()), "bmf_palm_finish_row")
                self._mark(it)
        self._events[slot][1].synchronize()
        return self._decode(self._log_host[slot].numpy().copy(), True)

    def previous_factors(self):
        """The iterate before the current one (Up64 / Vp64): what a loop that ran one iteration past its stopping rule returns."""
        X = self.X
        return self.Up64[: X.m, : self.k].cpu().numpy(), self.Vp64[: X.n, : self.k].cpu().numpy()

    def _decode(self, h, with_counts):
        X = self.X
        err = self.sum_x - 2.0 * float(h[0]) + float(h[1])
        counts = None
        if with_counts:
            tp, fp = int(h[4]), int(h[5])
            fn = int(self.sum_x) - tp
            counts = (tp, fp, fn, X.m * X.n - tp - fp - fn)
        return err, float(h[2]), float(h[3]), counts

    def _side(self, which):
        X = self.X
        if which == "U":
            return dict(F64=self.U64, P64=self.Up64, F=self.U, rows_pad=X.m_pad, rows=X.m, panel=self.Upanel, scale=self.scaleU, ws=self.wsU,
                        G=self.GU, G64=self.GU64, norms=self.normsU, part=self.partU, rb=self.ubits, cb=self.ucolbits, thr=self.thr[0],
                        # what this factor's STEP consumes: X V and the Gram of V
                        num=self.Mslab, splits=self.splits_xv, Go=self.GV, norms_o=self.normsV,
                        # what a REFRESH of this factor produces: X^T U
                        bits=X.bits_t, rp=X.n_pad, ldw=X.ldxt, red_words=X.m_pad // 32, out=self.Nslab, osplits=self.splits_xtu)
        return dict(F64=self.V64, P64=self.Vp64, F=self.V, rows_pad=X.n_pad, rows=X.n, panel=self.Vpanel, scale=self.scaleV, ws=self.wsV,
                    G=self.GV, G64=self.GV64, norms=self.normsV, part=self.partV, rb=self.vbits, cb=self.vcolbits, thr=self.thr[1],
                    num=self.Nslab, splits=self.splits_xtu, Go=self.GU, norms_o=self.normsU,
                    bits=X.bits, rp=X.m_pad, ldw=X.ldx, red_words=X.n_pad // 32, out=self.Mslab, osplits=self.splits_xv)

    def load_factors(self, U0, V0, U_prev=None, V_prev=None):
        """Initial factors; ``U_prev / V_prev``: the iterate before them (inertial term), default = the same (ELBMF.py:111)."""
        X = self.X
        for F64, P64, F0, Fp, rows in ((self.U64, self.Up64, U0, U_prev, X.m), (self.V64, self.Vp64, V0, V_prev, X.n)):
            F64.zero_()
            P64.zero_()
            F64[:rows, : self.k] = torch.from_numpy(np.ascontiguousarray(F0, dtype=np.float64)).to(self.device)
            P64[:rows, : self.k] = torch.from_numpy(np.ascontiguousarray(F0 if Fp is None else Fp, dtype=np.float64)).to(self.device)
        self.U.copy_(self.U64)
        self.V.copy_(self.V64)
        with torch.cuda.device(self.device):
            self.refresh("U")
            self.refresh("V")

    def factors(self):
        X = self.X
        return self.U64[: X.m, : self.k].cpu().numpy(), self.V64[: X.n, : self.k].cpu().numpy()

    def refresh(self, which):
        """Everything derived from factor `which` after it changed: its fp16 panel, its Gram (fp32 + fp64), the two norms of
        the Gram, and the big contraction that uses it (X V for V, X^T U for U)."""
        s = self._side(which)
        kp, kk = self.kp, self.kp * self.kp
        with torch.cuda.device(self.device):
            st = _stream()
            if self.panel == "i8":
                check(lib.bmf_make_panel_i8(ptr(s["F64"]), ptr(s["F"]), s["rows_pad"], kp, kp, 3, ptr(s["panel"]), s["rows_pad"], ptr(s["ws"]),
                                            ptr(s["scale"]), st), "bmf_make_panel_i8")
            else:
                check(lib.bmf_make_panel_f16(ptr(s["F"]), s["rows_pad"], kp, kp, ptr(s["panel"]), s["rows_pad"], ptr(s["ws"]), ptr(s["scale"]), st),
                      "bmf_make_panel_f16")
            check(lib.bmf_gram_partial(ptr(s["F"]), s["rows_pad"], kp, kp, ptr(self.gram_slabs), self.gram_blocks, st), "bmf_gram_partial")
            check(lib.bmf_reduce_slabs(ptr(self.gram_slabs), kk, self.gram_blocks, kk, ptr(s["G"]), ptr(s["G64"]), st), "bmf_reduce_slabs")
            check(lib.bmf_sym_norms(ptr(s["G64"]), kp, ptr(s["norms"]), st), "bmf_sym_norms")
            if self.panel == "i8":
                tiled = self._tiled[1] if which == "U" else self._tiled[0]   # a refresh of U runs X^T U
                check(lib.bmf_xf_bits_i8(ptr(tiled), s["rp"], s["ldw"], s["red_words"], ptr(s["panel"]), s["rows_pad"], 3, ptr(s["scale"][kp:]), kp,
                                         ptr(s["out"]), s["rp"] * kp, s["osplits"], 1, st), "bmf_xf_bits_i8")
            else:
                check(lib.bmf_xf_bits_f16(ptr(s["bits"]), s["rp"], s["ldw"], s["red_words"], ptr(s["panel"]), s["rows_pad"], ptr(s["scale"][kp:]), kp,
                                          ptr(s["out"]), s["rp"] * kp, s["osplits"], st), "bmf_xf_bits_f16")

    def masked_grad(self, which):
        """(W o X) G and (W o (Fe G^T)) G for factor `which` over the observed cells, Fe = its extrapolated point, G = the CURRENT
        other factor (ELBMF.py:188-190).  ELBMF's loop is Jacobi -- the V step sees the old U -- so its caller runs this for both
        factors before either step(); step() runs it itself when the caller has not."""
        assert self.obs is not None
        s = self._side(which)
        ls, rows = (self.obs.csr, self.X.m) if which == "U" else (self.obs.csc, self.X.n)
        Fe, num, den, other = (self.FeU, self.numU, self.denU, self.V) if which == "U" else (self.FeV, self.numV, self.denV, self.U)
        with torch.cuda.device(self.device):
            st = _stream()
            check(lib.bmf_palm_extrapolate(ptr(s["F64"]), ptr(s["P64"]), self.beta, s["rows_pad"] * self.kp, ptr(Fe), st), "bmf_palm_extrapolate")
            if ls.get("part") is None or ls["part"].shape[2] != self.kp:
                ls["part"] = torch.zeros((max(ls["nseg"], 1), 2, self.kp), dtype=torch.float32, device=self.device)
            check(lib.bmf_masked_pass(ptr(ls["ptr"]), ptr(ls["idx"]), ptr(ls["val"]), ptr(ls["wgt"]), rows, ptr(ls["seg_row"]),
                                      ptr(ls["seg_beg"]), ls["nseg"], ptr(ls["row_seg_ptr"]), ptr(Fe), ptr(other), self.kp,
                                      ptr(ls["part"]), ptr(num), ptr(den), None, st), "bmf_masked_pass")
        self._grad_ready[which] = True

    def step(self, which, l1: float, l2: float, gap_l1: float = 0.0, gap_l2: float = 0.0, advance_prev: bool = True):
        """One proximal step of factor `which` from the CURRENT contraction / Gram of the other factor."""
        s = self._side(which)
        if self.obs is not None and not self._grad_ready[which]:
            self.masked_grad(which)
        a = L.PalmArgs()
        a.F64, a.Fprev64, a.F = s["F64"].data_ptr(), s["P64"].data_ptr(), s["F"].data_ptr()
        a.rows_pad, a.rows, a.k, a.kp = s["rows_pad"], s["rows"], self.k, self.kp
        a.splits, a.num, a.slab_stride = s["splits"], s["num"].data_ptr(), s["rows_pad"] * self.kp
        a.G, a.norms, a.norm_kind, a.variant = s["Go"].data_ptr(), s["norms_o"].data_ptr(), self.norm_kind, self.variant
        a.beta, a.l1, a.l2, a.gap_l1, a.gap_l2 = self.beta, float(l1), float(l2), float(gap_l1), float(gap_l2)
        a.advance_prev, a.thr = int(advance_prev), float(s["thr"])
        a.rowbits, a.colbits, a.ldcb = s["rb"].data_ptr(), s["cb"].data_ptr(), s["rows_pad"] // 32
        a.partials, a.blockmax, a.stop = s["part"].data_ptr(), None, None
        if self.obs is not None:   # gradient = den - num of the masked pass (one array each)
            num, den = (self.numU, self.denU) if which == "U" else (self.numV, self.denV)
            a.splits, a.num, a.den = 1, num.data_ptr(), den.data_ptr()
            self._grad_ready[which] = False
        with torch.cuda.device(self.device):
            check(lib.bmf_palm_epilogue(C.byref(a), _stream()), "bmf_palm_epilogue")

    def scalars(self, with_counts=True):
        """(||X - U V^T||_F^2, U gap, V gap, (TP, FP, FN, TN) or None) of the current state; one synchronising read.
        Needs Mslab = X V and both Grams of the current factors (i.e. both sides refreshed)."""
        X = self.X
        with torch.cuda.device(self.device):
            st = _stream()
            n = X.m_pad * self.kp
            check(lib.bmf_dot_slabs(ptr(self.U64), ptr(self.Mslab), n, self.splits_xv, n, ptr(self.dotpart), self.dot_blocks, st), "bmf_dot_slabs")
            if with_counts:   # (the counters are zero: load_factors / the previous call reset them)
                check(lib.bmf_cover_count(ptr(X.bits), X.m_pad, X.ldx, X.n_pad // 32, ptr(self.ubits), ptr(self.vcolbits), X.n_pad // 32,
                                          self.kp, ptr(self.counts), None, st), "bmf_cover_count")
            check(lib.bmf_palm_scalars(ptr(self.dotpart), self.dot_blocks, ptr(self.GU64), ptr(self.GV64), self.kp * self.kp, ptr(self.partU),
                                       self.partU.numel(), ptr(self.partV), self.partV.numel(), ptr(self.counts) if with_counts else None,
                                       ptr(self._scal), st), "bmf_palm_scalars")
            h = self._scal.cpu().numpy()
        return self._decode(h, with_counts)
