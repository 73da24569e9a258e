"""BinaryMFThreshold -- learn the two scalar thresholds (u, v) that binarise given real factors U, V, by Wolfe line
search on the sigmoid-smoothed reconstruction error.  Drop-in for ``PyBMF/models/BinaryMFThreshold.py``.

F(u, v) and its gradient are each ONE tile-fused pass on the GPU (csrc/residual.hip, ``bmf_thresh_eval``): the m x n
product of the sigmoid-transformed factors is never materialised.  The line search (``solvers/line_search.py``) and the
outer loop (:82-147) are host control flow, as in the reference.
"""
from __future__ import annotations


import ctypes as C
import os
import time

import numpy as np

from ..solvers import line_search, limit_step_size
from ..utils import ismat
from .ContinuousModel import ContinuousModel


class BinaryMFThreshold(ContinuousModel):
    def __init__(self, k, U, V, W='mask', u=0.5, v=0.5, lamda=100, solver="line-search", min_diff=1e-3, max_iter=100,
                 init_method='custom', normalize_method=None, seed=None):
        self.check_params(k=k, U=U, V=V, W=W, u=u, v=v, lamda=lamda, solver=solver, min_diff=min_diff, max_iter=max_iter,
                          init_method=init_method, normalize_method=normalize_method, seed=seed)

    def check_params(self, **kwargs):
        super().check_params(**kwargs)
        assert self.solver in ['line-search']
        assert self.init_method in ['custom']
        assert self.normalize_method in ['balance', 'matrixwise-normalize', 'columnwise-normalize', 'matrixwise-mapping',
                                         'columnwise-mapping', None]
        assert ismat(self.W) or self.W in ['mask', 'full']

    def fit(self, X_train, X_val=None, X_test=None, **kwargs):
        super().fit(X_train, X_val, X_test, **kwargs)
        self._fit()
        self.X_pd = None  # Boolean product at the learnt (u, v), built on first access
        self.finish(show_logs=self.show_logs, save_model=self.save_model, show_result=self.show_result)

    def _make_X_pd(self):
        from ..device_ops import boolean_product_csr
        return boolean_product_csr(self.U, self.V, u=self.u, v=self.v, device=self.device)

    def _cover_counts(self):
        """TP / FP / FN / TN at the current (u, v): the k-bit words of the thresholded factors are built ON THE DEVICE from the fp64
        factors the search already keeps there (the generic route thresholds and packs them on the host: 0.25 ms of NumPy per outer
        iteration), then one cover-count launch and a 16-byte read-back."""
        import torch
        from .. import _lib as L
        from .._lib import lib, check, ptr
        Ud = getattr(self, "_Ud", None)
        if not self._boolean:
            return super()._cover_counts()   # real-valued data: the arithmetic "confusion sums" of the host-side U, V (unchanged by the search)
        if (Ud is None or getattr(self, "_log_buffer", None) is None or self.k > L.MAX_KP or getattr(self, "_sharded", False)
                or getattr(self, "_rows", (0, self.m)) != (0, self.m)):
            return super()._cover_counts()   # (outside fit() the host-side U, V are the truth: they may have been replaced since)
        B, dev, kp = self._bits, self._bits.device, self._kp
        with torch.cuda.device(dev):
            cache = getattr(self, "_bit_consts", None)
            if cache is None or cache[0].device != Ud.device:
                cache = (torch.from_numpy(np.array([1 << c for c in range(self.k)], dtype=np.uint64).view(np.int64)).to(dev),   # (bit 63 wraps: uint64 viewed as int64)
                         torch.tensor([1 << b for b in range(32)], dtype=torch.int64, device=dev),
                         torch.zeros(B.m_pad, dtype=torch.int64, device=dev), torch.zeros((kp, B.n_pad), dtype=torch.int64, device=dev),
                         torch.zeros(2, dtype=torch.int64, device=dev))
                self._bit_consts = cache
            wk, w32, ub, vwide, cnt = cache
            ub[: self.m] = ((Ud[: self.m, : self.k] > float(self.u)).to(torch.int64) * wk).sum(1)
            vwide[: self.k, : self.n] = (self._Vd[: self.n, : self.k] > float(self.v)).t().to(torch.int64)
            vcb = (vwide.view(kp, B.n_pad // 32, 32) * w32).sum(-1).to(torch.int32)
            cnt.zero_()
            check(lib.bmf_cover_count(ptr(B.bits), B.m_pad, B.ldx, B.n_pad // 32, ptr(ub), ptr(vcb), B.n_pad // 32, kp, ptr(cnt), None,
                                      C.c_void_p(torch.cuda.current_stream().cuda_stream)), "bmf_cover_count")   # (the stream of the torch operations above)
            tp, fp = (int(x) for x in cnt.cpu().numpy())
        fn = B.sum_local - tp
        return tp, fp, fn, self.m * self.n - tp - fp - fn

    def threshold_to_x(self):
        return np.array([self.u, self.v])

    def x_to_threshold(self, x_last):
        self.u, self.v = x_last[0], x_last[1]

    def x_bounds(self):
        eps = 1e-6
        return (np.array([self.U.min() + eps, self.V.min() + eps]), np.array([self.U.max() - eps, self.V.max() - eps]))

    def evaluate_with_threshold(self, n_iter, new_fval):
        self.X_pd = None
        self.evaluate(df_name='updates', head_info={'iter': n_iter, 'u': self.u, 'v': self.v, 'F': new_fval})

    # ---- device side ------------------------------------------------------------------------------------------
    def _upload_factors(self):
        import torch
        B = self._bits
        self._kp = 32 if self.k <= 32 else 64
        dev = B.device
        # fp64 on the device: the line search compares F values that differ by min_diff = 1e-3 on F ~ 1e4 (csrc/thresh64.hip)
        from .._lib import lib
        self._Ud = torch.zeros((B.m_pad, self._kp), dtype=torch.float64, device=dev)
        self._Vd = torch.zeros((B.n_pad, self._kp), dtype=torch.float64, device=dev)
        self._Ud[: self.m, : self.k] = torch.from_numpy(np.ascontiguousarray(self.U, dtype=np.float64)).to(dev)
        self._Vd[: self.n, : self.k] = torch.from_numpy(np.ascontiguousarray(self.V, dtype=np.float64)).to(dev)
        self._mblocks = 1024
        n_work = int(lib.bmf_thresh_eval64_work(B.m_pad, B.n_pad, self._kp))
        self._work = torch.zeros((max(n_work, (2 * B.m_pad + 2 * B.n_pad) * self._kp + 4 * self._mblocks),), dtype=torch.float64, device=dev)
        self._out = torch.zeros(4, dtype=torch.float64, device=dev)
        # stream handle and device index looked up once per fit: an evaluation is ~50 us of kernels, torch's current_stream() /
        # device-context bookkeeping was 17 ms of a 190-ms fit
        import ctypes as C
        with torch.cuda.device(dev):
            self._stream_obj = torch.cuda.current_stream()
        self._stream_ptr = C.c_void_p(self._stream_obj.cuda_stream)
        self._dev_index = dev.index if dev.index is not None else torch.cuda.current_device()
        # the four result sums land in pinned host memory straight from the last kernel (mapped: the device writes through the same
        # pointer), so an evaluation ends with a stream synchronisation instead of a device-to-host copy
        # (allocated with the model's device current: the pinned allocation is mapped for THAT device, the kernel writes through the
        # host pointer)
        with torch.cuda.device(dev):
            self._out_host = torch.zeros(4, dtype=torch.float64).pin_memory()
        self._out_np = self._out_host.numpy()
        # "not delivered yet" marker of the result words: a signalling NaN whose payload is the evaluation's number.  No arithmetic
        # produces that bit pattern (a computed NaN is quiet), so a result that IS NaN ends the wait like any other value -- the wait
        # compares bit patterns, not "is it still NaN".
        self._out_u64 = self._out_np.view(np.uint64)
        self._eval_no = 0
        self._poll = True   # wait for the result words in pinned memory (bounded) instead of a stream synchronisation: -15 us per evaluation
        self._F_memo, self._dF_memo = {}, {}
        self._setup_trace()
        # Stream contract of the dense evaluation: every launch of this fit goes to the stream that was current HERE and each
        # evaluation synchronises that stream before it returns its numbers, so the results do not depend on what the caller's
        # current stream is at evaluation time; the factors above were uploaded from pageable host memory (synchronous copies),
        # so nothing enqueued elsewhere has to be ordered before an evaluation.

    # ---- trace form, batched (csrc/thresh_trace.hip): the all-ones mask only ---------------------------------------------
    def _setup_trace(self):
        """The ones of X as a row list on the device + workspace for bmf_thresh_trace64; self._trace stays None when the objective
        runs over a list of observed cells (W = 'mask' / weights) or BMF_THRESH_TRACE=0 asks for the tile product."""
        import torch
        from .._lib import lib
        self._trace = None
        if getattr(self, "_obs", None) is not None or os.environ.get("BMF_THRESH_TRACE", "1") == "0":
            return
        B = self._bits
        dev = B.device
        with torch.cuda.device(dev):
            # (the list of the ones costs ~16 bytes per one while it is built: a dense or very large X keeps the tile-product objective)
            if int(B.sum_local) * 16 > min(torch.cuda.mem_get_info(dev)[0] // 4, 4 << 30):
                return
            shifts = torch.arange(32, dtype=torch.int32, device=dev)
            step = max(1, (1 << 26) // max(1, B.bits.shape[1] * 32))   # rows per chunk: the unpacked 0 / 1 chunk stays under 256 MB
            rows_l, cols_l = [], []
            for a in range(0, self.m, step):
                b = min(self.m, a + step)
                cells = ((B.bits[a:b].unsqueeze(-1) >> shifts) & 1).reshape(b - a, -1)[:, : self.n]   # rows a..b of X as 0 / 1
                rc = torch.nonzero(cells)                                                             # row-major order
                rows_l.append(rc[:, 0] + a)
                cols_l.append(rc[:, 1].to(torch.int32))
                del cells, rc
            rows_all = torch.cat(rows_l) if rows_l else torch.zeros(0, dtype=torch.int64, device=dev)
            idx = (torch.cat(cols_l) if cols_l else torch.zeros(0, dtype=torch.int32, device=dev)).contiguous()
            counts = torch.bincount(rows_all, minlength=self.m)
            starts = torch.cumsum(counts, 0) - counts
            # segments of <= 128 cells of one row (a wave's unit of work), the longest first
            SEG = 128
            nseg_row = (counts + SEG - 1) // SEG
            seg_row = torch.repeat_interleave(torch.arange(self.m, device=dev), nseg_row)
            first = torch.cumsum(nseg_row, 0) - nseg_row
            within = torch.arange(seg_row.numel(), device=dev) - first[seg_row]
            seg_beg = starts[seg_row] + within * SEG
            seg_len = torch.minimum(counts[seg_row] - within * SEG, torch.tensor(SEG, device=dev))
            o = torch.argsort(seg_len, descending=True, stable=True)
            seg_row, seg_beg, seg_len = seg_row[o].to(torch.int32).contiguous(), seg_beg[o].contiguous(), seg_len[o].to(torch.int32).contiguous()
            nseg = int(seg_row.numel())
            if idx.numel() == 0 or nseg == 0 or nseg > 8 * self.m + 64:
                return
            max_pairs = int(lib.bmf_thresh_trace64_max_pairs(self.k))
            # The workspace grows with the pairs evaluated per call (~ 2 pairs (m + n + 2) k doubles + Grams + cell partials: 1.6 GB at
            # 100k x 100k, k = 32, 32 pairs).  Keep it within a budget -- a quarter of the free device memory, at most 2 GiB -- by
            # evaluating fewer pairs per call; when even one pair does not fit, the tile-product objective (its workspace is 2 (m + n) k
            # doubles) takes the fit instead of an out-of-memory error.  (advisor, round 4)
            free_b, _ = torch.cuda.mem_get_info(dev)
            budget = min(free_b // 4, 2 << 30)
            n_work = int(lib.bmf_thresh_trace64_work(self.m, self.n, self.k, max_pairs))
            while max_pairs > 1 and n_work * 8 > budget:
                max_pairs //= 2
                n_work = int(lib.bmf_thresh_trace64_work(self.m, self.n, self.k, max_pairs))
            if n_work <= 0 or n_work * 8 > budget:
                return
            try:
                work = torch.zeros(n_work, dtype=torch.float64, device=dev)
            except torch.cuda.OutOfMemoryError:
                return
            out_host = torch.zeros(4 * max_pairs + 1, dtype=torch.float64).pin_memory()
        self._trace = {"seg_row": seg_row, "seg_beg": seg_beg, "seg_len": seg_len, "nseg": nseg, "idx": idx, "work": work, "out_host": out_host, "out": out_host.numpy(), "max_pairs": max_pairs,
                       "seq": 0.0, "sum_x": float(idx.numel()), "last_hit": 8}

    def _eval_trace(self, points, want_grad):
        """F (and dF) at every point of `points` in one enqueue; fills the memo tables."""
        from .._lib import lib, check, ptr
        tr = self._trace
        pts = [(float(p_[0]), float(p_[1])) for p_ in points]
        for a in range(0, len(pts), tr["max_pairs"]):
            part = pts[a:a + tr["max_pairs"]]
            uv = (C.c_double * (2 * len(part)))(*[x for pr in part for x in pr])
            tr["seq"] += 1.0
            seq, out = tr["seq"], tr["out"]
            check(lib.bmf_thresh_trace64(ptr(tr["seg_row"]), ptr(tr["seg_beg"]), ptr(tr["seg_len"]), tr["nseg"], ptr(tr["idx"]), self.m, self.n, ptr(self._Ud), ptr(self._Vd), self._kp, self.k, uv,
                                         len(part), float(self.lamda), tr["sum_x"], int(want_grad), ptr(tr["work"]), ptr(tr["out_host"]),
                                         seq, self._stream_ptr), "bmf_thresh_trace64")
            # every pair's word 0 and the word after the last pair carry this call's sequence number, each written behind the results it
            # vouches for: wait for ALL of them (bounded), else for the stream.  (Waiting for the last word alone relied on writes to
            # host memory from different blocks becoming visible in fence order; once in ~10^5 calls a slot still held the previous
            # call's value -- a nearby trial point -- and the search took a slightly different path.)
            if self._poll:
                stamps = out[0:4 * len(part) + 1:4]
                deadline = time.perf_counter() + 0.002
                while not (stamps == seq).all():
                    if time.perf_counter() > deadline:
                        self._stream_obj.synchronize()
                        break
            else:
                self._stream_obj.synchronize()
            for i, key in enumerate(part):
                self._F_memo[key] = float(0.5 * out[4 * i + 1])
                if want_grad:
                    self._dF_memo[key] = np.array([out[4 * i + 2], out[4 * i + 3]])

    def _prefetch(self, points):
        """line_search's announcement of its next trial steps: evaluate F at the ones not known yet, in one batch."""
        todo = [p_ for p_ in points if (float(p_[0]), float(p_[1])) not in self._F_memo]
        if todo:
            self._eval_trace(todo, False)

    def _eval(self, params, want_grad):
        import torch
        from .._lib import lib, check, ptr
        from ..engine import _stream
        B = self._bits
        u, v = float(params[0]), float(params[1])
        if getattr(self, "_obs", None) is not None:
            return self._eval_masked(u, v, want_grad)
        if torch.cuda.current_device() != self._dev_index:
            with torch.cuda.device(self._dev_index):
                return self._eval_dense(u, v, want_grad)
        return self._eval_dense(u, v, want_grad)

    def _mark_pending(self, words):
        """Fill the result words with this evaluation's "not delivered yet" pattern; returns it."""
        self._eval_no = (self._eval_no + 1) & 0xFFFFFFFF
        pending = np.uint64(0x7FF4DEAD00000000 | self._eval_no)   # exponent all ones, quiet bit clear, payload = the evaluation's number
        self._out_u64[words] = pending
        return pending

    def _wait_words(self, pending, words, stream):
        """Wait until none of the result words holds `pending` any more (at most 2 ms, then for the stream)."""
        w = self._out_u64[words]
        deadline = None
        while (w == pending).any():
            if deadline is None:
                deadline = time.perf_counter() + 0.002
            elif time.perf_counter() > deadline:
                stream.synchronize()
                break

    def _eval_dense(self, u, v, want_grad):
        from .._lib import lib, check, ptr
        B = self._bits
        out = self._out_np
        if self._poll:
            pending = self._mark_pending(slice(0, 4))   # (the previous evaluation has delivered: nothing is in flight)
        check(lib.bmf_thresh_eval64(ptr(B.bits), B.m_pad, B.ldx, self.m, self.n, ptr(self._Ud), B.n_pad, ptr(self._Vd), self.k,
                                    self._kp, u, v, float(self.lamda), int(want_grad), ptr(self._work), ptr(self._out_host),
                                    self._stream_ptr), "bmf_thresh_eval64")
        if self._poll:
            # The last kernel writes its four sums into this pinned (host-coherent) array: wait for the four words themselves instead of
            # for the stream -- a stream synchronisation costs ~15 us of wake-up latency per evaluation, a third of the kernel time, and a
            # Wolfe search is a chain of ~25 dependent evaluations.  Bounded: pinned memory that turns out not to be host-coherent ends
            # in the stream synchronisation after 2 ms.
            self._wait_words(pending, slice(0, 4), self._stream_obj)
            return out.copy()
        self._stream_obj.synchronize()
        return out.copy()

    def _eval_masked(self, u, v, want_grad):
        """F / dF over the observed cells only (W = 'mask' on unstored cells, or weights): transform + sparse pass."""
        import torch
        from .._lib import lib, check, ptr
        from ..engine import _stream
        B, ls, kp = self._bits, self._obs.csr, self._kp
        with torch.cuda.device(B.device):
            mp, np_ = B.m_pad * kp, B.n_pad * kp
            Us, dUs, Vs, dVs, part = (self._work[0:mp], self._work[mp:2 * mp], self._work[2 * mp:2 * mp + np_],
                                      self._work[2 * mp + np_:2 * mp + 2 * np_], self._work[2 * mp + 2 * np_:])
            s = _stream()
            check(lib.bmf_thresh_transform64(ptr(self._Ud), B.m_pad, self.m, self.k, kp, u, float(self.lamda), ptr(Us),
                                             ptr(dUs) if want_grad else None, s), "bmf_thresh_transform64")
            check(lib.bmf_thresh_transform64(ptr(self._Vd), B.n_pad, self.n, self.k, kp, v, float(self.lamda), ptr(Vs),
                                             ptr(dVs) if want_grad else None, s), "bmf_thresh_transform64")
            # the three sums land in words 1..3 of the pinned result array (word 0 is unused here), written -- not accumulated -- by the
            # last launch; the host waits for those words, as in the dense evaluation
            out = self._out_np
            out[0] = 0.0
            pending = self._mark_pending(slice(1, 4))
            check(lib.bmf_masked_thresh64_k(ptr(ls["ptr"]), ptr(ls["idx"]), ptr(ls["val"]), ptr(ls["wgt"]), ptr(ls["seg_row"]),
                                            ptr(ls["seg_beg"]), ls["nseg"], ptr(Us), ptr(dUs) if want_grad else None, ptr(Vs),
                                            ptr(dVs) if want_grad else None, kp, self.k, ptr(part), self._mblocks,
                                            C.c_void_p(self._out_host.data_ptr() + 8), s), "bmf_masked_thresh64_k")
            launch_stream = torch.cuda.current_stream()   # the stream the launches above went to (_stream())
            if self._poll:
                self._wait_words(pending, slice(1, 4), launch_stream)
            else:
                launch_stream.synchronize()
            return out.copy()   # [unused, sum (w r)^2, g1, g2]: same slots as the dense path

    def F(self, params):
        """0.5 * || X - sigmoid(lamda (U - u)) sigmoid(lamda (V - v))^T ||_F^2   (:150-171).  A point is evaluated once per fit: the
        search asks for the accepted point again as `new_fval`, after `limit_step_size` and as `fk` of the next search
        (line_search.py:35,64, BinaryMFThreshold.py:109-118 of the reference)."""
        key = (float(params[0]), float(params[1]))
        memo = getattr(self, "_F_memo", None)
        if memo is not None and key in memo:
            return memo[key]
        if getattr(self, "_trace", None) is not None:
            self._eval_trace([key], False)
            return self._F_memo[key]
        val = float(0.5 * self._eval(params, False)[1])
        if memo is not None:
            memo[key] = val
        return val

    def dF(self, params):
        """The 2-vector the reference calls dF (:174-207); memoised like F."""
        key = (float(params[0]), float(params[1]))
        memo = getattr(self, "_dF_memo", None)
        if memo is not None and key in memo:
            return memo[key].copy()
        if getattr(self, "_trace", None) is not None:
            self._eval_trace([key], True)
            return self._dF_memo[key].copy()
        o = self._eval(params, True)
        val = np.array([o[2], o[3]])
        if memo is not None:
            memo[key] = val.copy()
            self._F_memo.setdefault(key, float(0.5 * o[1]))
        return val

    def dXdx(self, X, x):
        """lamda * sigmoid'(lamda (X - x)) (:211-227), in the overflow-free form lamda * s * (1 - s)."""
        z = (np.asarray(X, dtype=np.float64) - x) * self.lamda
        s = np.where(z >= 0, 1.0 / (1.0 + np.exp(-np.abs(z))), np.exp(-np.abs(z)) / (1.0 + np.exp(-np.abs(z))))
        return self.lamda * s * (1.0 - s)

    def _fit(self):
        self._upload_factors()
        self._log_buffer = {}
        try:
            self._search()
        finally:
            self._flush_logs()
            self._log_buffer = None

    def _search(self):
        n_iter = 0
        x_last = self.threshold_to_x()
        p_last = -self.dF(x_last)
        new_fval = self.F(x_last)
        self.evaluate_with_threshold(n_iter, new_fval)
        improving = True
        while improving:
            n_iter += 1
            xk, pk = x_last, p_last
            tr = getattr(self, "_trace", None)
            if tr is not None:
                # announce the trial chain of this search (known beforehand) so that it is ONE batched evaluation; its length follows the
                # step at which the previous search ended (a speed heuristic only: F values and decisions do not depend on it)
                n_before = len(self._F_memo)
                alpha, fc, gc, new_fval, old_fval, new_slope = line_search(f=self.F, myfprime=self.dF, xk=xk, pk=pk, maxiter=50, prefetch=self._prefetch,
                                                                            chain=min(tr["max_pairs"], tr["last_hit"] + 2))
                tr["last_hit"] = max(4, fc - 3)
                del n_before
            else:
                alpha, fc, gc, new_fval, old_fval, new_slope = line_search(f=self.F, myfprime=self.dF, xk=xk, pk=pk, maxiter=50)
            if alpha is None:
                print("[W] Search direction is not a descent direction.")
                break
            x_last = xk + alpha * pk
            x_min, x_max = self.x_bounds()
            x_last, alpha = limit_step_size(x_min=x_min, x_max=x_max, x_last=x_last, xk=xk, pk=pk, alpha=alpha)
            p_last = -self.dF(x_last)
            new_fval = self.F(x_last)
            diff = np.abs(new_fval - old_fval)
            self.print_msg("  Wolfe line search iter         : {}".format(n_iter))
            self.print_msg("    num of function evals        : {}".format(fc))
            self.print_msg("    num of gradient evals        : {}".format(gc))
            self.print_msg("    function value update        : {:.3f} -> {:.3f}".format(old_fval, new_fval))
            self.x_to_threshold(x_last)
            self.evaluate_with_threshold(n_iter, new_fval)
            improving = self.early_stop(n_iter=n_iter, diff=diff)
        self.n_iter = n_iter
