"""Shared start-up of the continuous-relaxation models (``PyBMF/models/ContinuousModel.py:14-203``):
mask handling, random factor initialisation (V is drawn before U), 'balance' normalisation, zeros -> eps, and the
hand-over of X to the GPU as a bit matrix.
"""
from __future__ import annotations

import numpy as np

from .. import _lib as L
from ..utils import ismat, to_dense
from .BaseModel import BaseModel

EPS = float(np.finfo(np.float64).eps)


class ContinuousModel(BaseModel):
    def __init__(self):
        raise NotImplementedError("This is a template class.")

    # ---- overridable knobs of this build (settable like any other parameter, e.g. fit(..., device='cuda:1')) ----
    device = "cuda:0"
    panel = "i8"       # operand format of the two bits GEMMs: 'i8' = `terms` planes of signed 8-bit digits of the column-scaled
                       # factor on the integer MFMA with exact int32 accumulation (default; drift from the fp64 reference at
                       # 100k x 20k an order of magnitude below the floating-point formats, profiles/r02_parity_trace_c3_*),
                       # 'f16' = two column-scaled fp16 addends (22 significant bits), 'bf16' = `terms` bf16 addends
    terms = 3          # digit planes (i8: 3 = 24-bit factor) / bf16 addends (3 = fp32-exact operands, 2 = 16 bits)
    with_mae = True    # run the residual pass that MAE needs (off: MAE column is NaN, RMSE/rec_error unaffected)

    def init_model(self):
        self._start_timer()
        self._make_name()
        self._init_logs()
        if not (hasattr(self, "init_method") and self.init_method == "custom"):
            self._init_factors()
        self._to_device()
        self._make_scorers()
        self.init_W()
        self.init_UV()
        self.normalize_UV()
        self._to_dense()
        self._to_float()
        if getattr(self, "solver", None) == "mu" and getattr(self, "U", None) is not None and getattr(self, "V", None) is not None:
            self.U[self.U == 0] = EPS
            self.V[self.V == 0] = EPS
        self._same_start_on_every_rank()

    # ---- row sharding over the GPUs of one node ----------------------------------------------------
    _sharded = False

    def _same_start_on_every_rank(self):
        """Sharded fit: the initial factors are rank 0's (with seed=None every process would draw its own)."""
        if not self._sharded or getattr(self, "U", None) is None or getattr(self, "V", None) is None:
            return
        import torch
        import torch.distributed as dist
        dev = self.device if dist.get_backend() == "nccl" else "cpu"
        for name in ("U", "V"):
            t = torch.from_numpy(np.ascontiguousarray(getattr(self, name), dtype=np.float64)).to(dev)
            dist.broadcast(t, src=0)
            setattr(self, name, t.cpu().numpy())

    def _shard_plan(self):
        """Decide whether this fit is row-sharded: a torch.distributed process group with more than one rank is up (one
        process per GPU, e.g. under torchrun; every rank calls fit() with the SAME arguments) and the model is BinaryMFPenalty,
        WNMF or PNLPF on a Boolean matrix without extra data sets (any mask the unsharded fit accepts).  Rank p then keeps rows [lo, hi) of X and of U, V is
        replicated, and each iteration exchanges two buffers (pybmf_amd/sharding.py).  After the fit every rank holds the
        full U, V and identical logs.  Anything else runs unsharded (identically on every rank)."""
        self._sharded, self._rows = False, (0, self.m)
        try:
            import torch.distributed as dist
        except ImportError:
            return
        if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
            return
        if "device" not in self.__dict__:     # not set by the caller: this process's current device (torch.cuda.set_device),
            import torch                      # whether or not the fit ends up sharded -- never every rank on cuda:0
            if torch.cuda.is_available():
                self.device = f"cuda:{torch.cuda.current_device()}"
        if type(self).__name__ not in ("BinaryMFPenalty", "WNMF", "PNLPF"):
            return
        if self.X_val is not None or self.X_test is not None:
            return
        if getattr(self, "task", None) == "prediction":   # scores over stored entries: not part of the exchange
            return
        if self.m < 64 * dist.get_world_size():           # shards are cut at multiples of 32 rows: keep at least two groups per rank
            return
        import torch
        from ..sharding import shard_rows
        self._sharded = True
        self._rows = shard_rows(self.m, dist.get_rank(), dist.get_world_size())
        if "device" not in self.__dict__:     # not set by the caller: this process's current device (torch.cuda.set_device)
            self.device = f"cuda:{torch.cuda.current_device()}"

    def _sum_over_ranks(self, values):
        """Element-wise sum over the ranks of a sharded fit (float64); identity otherwise."""
        if not self._sharded:
            return [float(v) for v in values]
        import torch
        import torch.distributed as dist
        t = torch.tensor([float(v) for v in values], dtype=torch.float64, device=self.device if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t)
        return [float(v) for v in t.cpu().numpy()]

    def _max_over_ranks(self, value):
        """Maximum over the ranks of a sharded fit; identity otherwise (every rank must take the same decision)."""
        if not self._sharded:
            return value
        import torch
        import torch.distributed as dist
        t = torch.tensor([float(value)], dtype=torch.float64, device=self.device if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def _gather_rows(self, F_local):
        """All ranks' row shards of a factor, concatenated (every rank gets the full matrix); identity when unsharded."""
        if not self._sharded:
            return F_local
        import torch
        import torch.distributed as dist
        from ..sharding import shard_rows
        world = dist.get_world_size()
        sizes = [hi - lo for lo, hi in (shard_rows(self.m, r, world) for r in range(world))]
        dev = self.device if dist.get_backend() == "nccl" else "cpu"
        buf = torch.zeros((max(sizes), F_local.shape[1]), dtype=torch.float64, device=dev)
        buf[: F_local.shape[0]] = torch.from_numpy(np.ascontiguousarray(F_local, dtype=np.float64)).to(dev)
        parts = [torch.empty_like(buf) for _ in range(world)]
        dist.all_gather(parts, buf)
        return np.concatenate([p[:sz].cpu().numpy() for p, sz in zip(parts, sizes)])

    # the models that also take real-valued data (the reference casts whatever it is given to float64 and runs,
    # ContinuousModel.py:188-203); WNMF has its own hand-over (dense fp32-MFMA path)
    _REAL_OK = ("BinaryMFPenalty", "PNLPF", "BinaryMFThreshold")
    _boolean = True
    _all_cells = False

    def _to_device(self):
        """X_train -> bits in HBM (both orientations).  Values other than 0 / 1: the matrix goes to the device as fp32 (engine.RealMatrix)
        and the fit runs over its cells on the masked kernels (init_W); never silently binarised."""
        from ..engine import BitMatrix
        X = self._X_input
        # uint8 arrays / tensors go to the device as they are and the packer reports the largest byte it saw: the "values are 0 / 1"
        # check then costs nothing (a host pass over a 100k x 20k array is 0.1 s, a third of a 30-update fit)
        on_device = (isinstance(X, np.ndarray) and X.dtype == np.uint8) or (hasattr(X, "dtype") and str(X.dtype) == "torch.uint8")
        self._boolean, self._all_cells = True, False
        if not on_device and not self._values_are_boolean(X):
            return self._to_device_real(X)
        self._shard_plan()
        lo, hi = self._rows
        self._bits = BitMatrix(X, self.device, row_lo=lo, row_hi=hi)
        if on_device and self._max_over_ranks(self._bits.max_u8) > 1:
            raise NotImplementedError("this model's GPU path takes a Boolean (0/1) matrix as uint8; pass other values as a float array")
        self._x_mean = self._sum_over_ranks([self._bits.sum_local])[0] / (float(self.m) * float(self.n))

    def _to_device_real(self, X):
        from ..engine import RealMatrix
        if type(self).__name__ not in self._REAL_OK:
            raise NotImplementedError("this model's GPU path takes a Boolean (0/1) matrix")
        import torch
        if isinstance(X, torch.Tensor):
            X = X.detach().cpu().numpy()
        host = np.asarray(X.todense()) if hasattr(X, "todense") else np.asarray(X)
        host = np.ascontiguousarray(host, dtype=np.float64)
        if not np.isfinite(host).all():
            raise TypeError("NaN is found in prediction.")   # (what the reference's first evaluate() raises on such data)
        self._boolean = False
        self._sharded, self._rows = False, (0, self.m)   # real-valued data for these models: one GPU
        self._real = RealMatrix(host, self.device)
        # (what the Boolean path keeps in `_bits` besides the bits: the padded shape and the device)
        self._bits = _PadDims(self.m, self.n, self._real.device)
        self._real_host = host
        self._x_mean = float(host.mean())
        self._sum_x_real = float(host.sum())

    @staticmethod
    def _values_are_boolean(X) -> bool:
        """Are all values 0 / 1?  Boolean-ness is a property of the VALUES (any dtype)."""
        import torch
        if isinstance(X, torch.Tensor):
            return bool(((X == 0) | (X == 1)).all().item())
        if hasattr(X, "data") and hasattr(X, "tocsr"):
            return (not X.nnz) or ContinuousModel._values_are_boolean(np.asarray(X.data))
        if isinstance(X, np.ndarray):
            # (np.isin sorts: 50 ms on a 6040 x 3706 uint8 matrix -- a quarter of a whole config-#5 fit)
            if not X.size or X.dtype.kind == "b":
                return True
            if X.dtype.kind == "u":
                return int(X.max()) <= 1
            if X.dtype.kind == "i":
                return int(X.min()) >= 0 and int(X.max()) <= 1
            if X.dtype.kind == "f":
                return np.count_nonzero((X != 0) & (X != 1)) == 0
            return False
        return True   # lazy row sources (generators) produce bits by construction

    @staticmethod
    def _check_boolean(X):
        """Anything but 0 / 1 is refused, never silently binarised."""
        if not ContinuousModel._values_are_boolean(X):
            raise NotImplementedError("this model's GPU path takes a Boolean (0/1) matrix")

    def init_W(self):
        """'full' = all-ones mask: never materialised (re-associated dense path).  'mask' = the pattern of STORED entries of
        the csr training matrix, explicit zeros included (ContinuousModel.py:52-55); an explicit matrix = weights.  Anything
        that is not the all-ones mask is turned into a device list of observed cells (engine.SparseObs) and the fit runs
        on the masked kernels (bmf_masked_pass)."""
        self._obs = None
        self._mask_is_pattern = False
        if not hasattr(self, "W"):
            return
        assert (isinstance(self.W, str) and self.W in ["mask", "full"]) or ismat(self.W)
        if isinstance(self.W, str) and self.W == "full":
            if self._wants_cell_list():
                self._observe_all_cells()
            return
        from scipy.sparse import coo_matrix, issparse
        from ..engine import SparseObs
        if isinstance(self.W, str):  # 'mask'
            if not issparse(self.X_train):
                raise NotImplementedError("W='mask' needs a host matrix (ndarray / scipy sparse) to take the stored pattern from")
            if self.X_train.nnz == self.m * self.n:
                return  # every cell is stored: the mask is the all-ones matrix
            if getattr(self, "beta_loss", "frobenius") == "kullback-leibler":
                self._mask_is_pattern = True  # WNMF-KL: only the objective sees the mask (bits of the pattern, WNMF._fit_kl)
                return
            coo = self.X_train.tocoo()
            rows, cols, vals, wgts = coo.row, coo.col, coo.data, None
        else:
            Wc = coo_matrix(self.W)
            if Wc.shape != (self.m, self.n):
                raise ValueError("W must have the shape of X_train")
            if Wc.nnz == self.m * self.n and (Wc.data == 1).all():
                return
            rows, cols, wgts = Wc.row, Wc.col, Wc.data
            Xd = self.X_train if not issparse(self.X_train) else self.X_train.tocsr()
            vals = np.asarray(Xd[rows, cols]).ravel()
        if self._sharded:   # this rank's rows of the observation list (the masked engine sums the V side over the ranks)
            lo, hi = self._rows
            rows = np.asarray(rows)
            keep = (rows >= lo) & (rows < hi)
            rows, cols, vals = rows[keep] - lo, np.asarray(cols)[keep], np.asarray(vals)[keep]
            wgts = None if wgts is None else np.asarray(wgts)[keep]
            self._obs = SparseObs(rows, cols, vals, wgts, (hi - lo, self.n), self.device)
            return
        self._obs = SparseObs(rows, cols, vals, wgts, (self.m, self.n), self.device)

    MAX_REAL_CELLS = 50_000_000

    def _wants_cell_list(self):
        """Does the all-ones mask need the list of all cells?  Real-valued data on the models whose dense kernels contract bits."""
        return not self._boolean

    def _observe_all_cells(self):
        """A real-valued X under the all-ones mask: the penalty / link / thresholding models have no re-associated dense path for it
        (their dense kernels contract BITS), so every cell becomes an observed cell of weight 1 and the fit runs on the masked kernels
        (bmf_masked_pass and friends: SDDMM + SpMM over a cell list) -- the reference's literal arithmetic, cell for cell."""
        from ..engine import SparseObs
        cells = self.m * self.n
        if cells > self.MAX_REAL_CELLS:
            raise NotImplementedError(f"real-valued X of {self.m} x {self.n} cells under W='full': the cell-list kernels take up to "
                                      f"{self.MAX_REAL_CELLS} cells (WNMF's Frobenius loss has the dense fp32 path for larger data)")
        rows = np.repeat(np.arange(self.m, dtype=np.int64), self.n)
        cols = np.tile(np.arange(self.n, dtype=np.int64), self.m)
        self._obs = SparseObs(rows, cols, self._real_host.ravel(), None, (self.m, self.n), self.device)
        self._all_cells = True

    def init_UV(self):
        if not hasattr(self, "init_method"):
            return
        if self.init_method == "normal":
            avg = np.sqrt(self._x_mean / self.k)
            V = avg * self.rng.standard_normal(size=(self.n, self.k))
            U = avg * self.rng.standard_normal(size=(self.m, self.k))
            self.U, self.V = np.abs(U), np.abs(V)
        elif self.init_method == "uniform":
            avg = np.sqrt(self._x_mean / self.k)
            self.V = self.rng.uniform(low=0, high=avg * 2, size=(self.n, self.k))
            self.U = self.rng.uniform(low=0, high=avg * 2, size=(self.m, self.k))
        elif self.init_method == "custom":
            assert getattr(self, "U", None) is not None and getattr(self, "V", None) is not None
            self.U, self.V = np.array(to_dense(self.U), dtype=np.float64), np.array(to_dense(self.V), dtype=np.float64)

    def normalize_UV(self):
        method = getattr(self, "normalize_method", None)
        if not hasattr(self, "normalize_method") or method is None:
            return
        lo_hi = lambda: (self.U.min(), self.U.max(), self.V.min(), self.V.max())  # noqa: E731
        before = lo_hi()
        if method == "balance":
            dU, dV = np.sqrt(self.U.max(axis=0)), np.sqrt(self.V.max(axis=0))
            for i in range(self.k):
                self.U[:, i] = self.U[:, i] * dV[i] / dU[i]
                self.V[:, i] = self.V[:, i] * dU[i] / dV[i]
        elif method == "matrixwise-normalize":
            self.U, self.V = self.U / self.U.max(), self.V / self.V.max()
        elif method == "columnwise-normalize":
            self.U, self.V = self.U / self.U.max(axis=0), self.V / self.V.max(axis=0)
        elif method == "matrixwise-mapping":
            self.U, self.V = unique_values_mapping(self.U), unique_values_mapping(self.V)
        elif method == "columnwise-mapping":
            for i in range(self.k):
                self.U[:, i] = unique_values_mapping(self.U[:, i])
                self.V[:, i] = unique_values_mapping(self.V[:, i])
        else:
            raise ValueError(f"normalize_method={method!r}")
        print("[I] Normalized from: U: [{:.4f}, {:.4f}], V: [{:.4f}, {:.4f}]".format(*before))
        print("[I]              to: U: [{:.4f}, {:.4f}], V: [{:.4f}, {:.4f}]".format(*lo_hi()))

    def _to_dense(self):
        for name in ("U", "V"):
            if getattr(self, name, None) is not None:
                setattr(self, name, to_dense(getattr(self, name)))

    def _to_float(self):
        for name in ("U", "V"):
            if getattr(self, name, None) is not None:
                setattr(self, name, np.asarray(getattr(self, name)).astype(np.float64))

    @staticmethod
    def _check_nan(values):
        """The reference refuses a prediction that contains NaN (utils/metrics.py:29-30, raised from the first evaluate());
        here the device loop has already run, so the check is made on what it logged."""
        if np.isnan(np.asarray(values, dtype=np.float64)).any():
            raise TypeError("NaN is found in prediction.")

    # ---- scoring from the device ------------------------------------------------------------------------------
    def _make_scorers(self):
        """What evaluate() measures on each data set besides the whole training matrix (which the engines score
        themselves): task='prediction' -> the entries of train / val / test (engine.ObservedScorer);
        task='reconstruction' -> the whole val / test matrices (engine.WholeScorer).

        Reference behaviour kept: the continuous models densify X_train / X_val / X_test in init_model
        (ContinuousModel.py:167-182 _to_dense), so the triplets that eval() gathers under task='prediction'
        (utils/evaluate_utils.py:33, to_triplet of a dense array) are the NON-ZERO cells only -- explicit zeros of a
        negative-sampled csr take part in the mask W (built before _to_dense) but not in these scores."""
        from ..engine import ObservedScorer, WholeScorer
        self._scorers = {}
        task = getattr(self, "task", None)
        if task not in ("prediction", "reconstruction"):
            return
        if task == "prediction":
            if not hasattr(self.X_train, "tocoo"):
                raise NotImplementedError("task='prediction' needs a host training matrix to take the stored entries from")

        def nonzeros(X):
            X = X.copy()
            X.eliminate_zeros()
            return X
        if task == "prediction" and self._train_scored_by_entries:
            self._scorers["train"] = ObservedScorer(nonzeros(self.X_train), self.device)
        for name in ("val", "test"):
            X = getattr(self, "X_" + name)
            if X is not None:
                self._scorers[name] = ObservedScorer(nonzeros(X), self.device) if task == "prediction" else WholeScorer(X, self.device)

    _train_scored_by_entries = True

    @staticmethod
    def _metric_values(metrics, rmse_mae, counts):
        from ..utils import scores_from_counts
        out = {}
        if rmse_mae is not None:
            out["RMSE"], out["MAE"] = rmse_mae
        if counts is not None:
            tp, fp, fn, tn = counts[:4]
            r, p, a, f1 = scores_from_counts(*counts)
            out.update({"TP": tp, "FP": fp, "FN": fn, "TN": tn, "Recall": r, "Precision": p, "Accuracy": a, "F1": f1,
                        "TPR": r, "PPV": p, "ACC": a})
        return [out.get(mt) for mt in metrics]

    def _engine_scores(self, eng, want_real=True, want_boolean=True, link=None, lamda=0.0):
        """{set name: (rmse_mae or None, counts or None)} for every extra scorer, from the engine's device state.  link / lamda: the
        real-valued prediction is sigmoid(lamda (U V^T - 1/2)) (PNLPF) instead of U V^T."""
        out = {}
        for name, sc in self._scorers.items():
            rm = sc.real(eng.U, eng.V, eng.kp, link=link, lamda=lamda) if want_real else None
            cn = sc.boolean(eng.ubits, eng.vbits, eng.vcolbits, eng.kp) if want_boolean else None
            out[name] = (rm, cn)
        return out

    def _score(self, name, metrics):
        """Values for `metrics` on data set `name` at the model's current host-side U, V (and thresholds)."""
        sc = getattr(self, "_scorers", {}).get(name)
        if sc is None:
            if name != "train":
                raise ValueError(f"no {name} data was given to fit()")
            return self._score_train(metrics)
        import torch
        from ..device_ops import _bits_of
        from ..engine import round_up
        from .. import _lib as L
        want_real = any(mt in ("RMSE", "MAE") for mt in metrics)
        want_bool = any(mt not in ("RMSE", "MAE") for mt in metrics)
        kp = 32 if self.k <= 32 else 64
        dev = sc.device
        m_pad, n_pad = round_up(self.m, L.ROW_PAD), round_up(self.n, L.ROW_PAD)
        rm = cn = None
        if self.k > L.MAX_KP:   # two 64-column blocks per factor (pybmf_amd/wide.py): the scorers take the block lists
            with torch.cuda.device(dev):
                def blocks(F, rows, rows_pad):
                    out = []
                    for b in range(2):
                        t = torch.zeros((rows_pad, 64), dtype=torch.float32, device=dev)
                        part = np.ascontiguousarray(np.asarray(F)[:, 64 * b: 64 * b + 64], dtype=np.float32)
                        t[:rows, : part.shape[1]] = torch.from_numpy(part).to(dev)
                        out.append(t)
                    return out
                if want_real:
                    rm = sc.real(blocks(self.U, self.m, m_pad), blocks(self.V, self.n, n_pad), 64)
                if want_bool:
                    u, v = self._thresholds()
                    ub, vb, vcb = [], [], []
                    for b in range(2):
                        rb_u, _, _ = _bits_of(np.asarray(self.U)[:, 64 * b: 64 * b + 64] > u, m_pad)
                        rb_v, cb_v, _ = _bits_of(np.asarray(self.V)[:, 64 * b: 64 * b + 64] > v, n_pad)
                        ub.append(torch.from_numpy(rb_u).to(dev))
                        vb.append(torch.from_numpy(rb_v).to(dev))
                        cbp = np.zeros((64, cb_v.shape[1]), dtype=cb_v.dtype)
                        cbp[: cb_v.shape[0]] = cb_v
                        vcb.append(torch.from_numpy(np.ascontiguousarray(cbp)).to(dev))
                    cn = sc.boolean(ub, vb, vcb, 64)
            return self._metric_values(metrics, rm, cn)
        with torch.cuda.device(dev):
            if want_real:
                Ud = torch.zeros((m_pad, kp), dtype=torch.float32, device=dev)
                Vd = torch.zeros((n_pad, kp), dtype=torch.float32, device=dev)
                Ud[: self.m, : self.k] = torch.from_numpy(np.ascontiguousarray(self.U, dtype=np.float32)).to(dev)
                Vd[: self.n, : self.k] = torch.from_numpy(np.ascontiguousarray(self.V, dtype=np.float32)).to(dev)
                rm = sc.real(Ud, Vd, kp)
            if want_bool:
                u, v = self._thresholds()
                rb_u, _, _ = _bits_of(np.asarray(self.U) > u, m_pad)
                rb_v, cb_v, _ = _bits_of(np.asarray(self.V) > v, n_pad)
                ub, vb = torch.from_numpy(rb_u).to(dev), torch.from_numpy(rb_v).to(dev)
                vcb = torch.from_numpy(np.ascontiguousarray(cb_v)).to(dev)
                cn = sc.boolean(ub, vb, vcb, kp)
        return self._metric_values(metrics, rm, cn)

    def _score_train(self, metrics):
        """Values for `metrics` at the model's current U, V (and thresholds): Boolean scores from the cover-count kernel,
        RMSE / MAE from the residual pass."""
        from ..utils import scores_from_counts
        out = {}
        if any(mt in ("RMSE", "MAE") for mt in metrics):
            s_abs, s_sq = self._residual_sums()
            cells = float(self.m) * float(self.n)
            out["RMSE"], out["MAE"] = float(np.sqrt(s_sq / cells)), float(s_abs / cells)
        if any(mt not in ("RMSE", "MAE") for mt in metrics):
            cnt = self._cover_counts()
            tp, fp, fn, tn = cnt[:4]
            r, p, a, f1 = scores_from_counts(*cnt)
            out.update({"TP": tp, "FP": fp, "FN": fn, "TN": tn, "Recall": r, "Precision": p, "Accuracy": a, "F1": f1,
                        "TPR": r, "PPV": p, "ACC": a})
        return [out.get(mt) for mt in metrics]

    def _thresholds(self):
        return getattr(self, "u", 0.5), getattr(self, "v", 0.5)

    def _cover_counts(self):
        from ..device_ops import boolean_product_bits
        import torch
        from .._lib import lib, check, ptr
        from ..engine import _stream
        u, v = self._thresholds()
        if not self._boolean:
            return self._real_confusion(np.asarray(self.U) > u, np.asarray(self.V) > v)
        B = self._bits
        lo, hi = getattr(self, "_rows", (0, self.m))
        if self.k <= L.MAX_KP:
            # ONE launch of the cover-count kernel on the k-bit words of the thresholded factors (the Boolean product is never
            # materialised) and a 16-byte read-back; the product-bits + popcount route below was 0.7 ms of launches, bit operations
            # and copies per call -- a third of an outer iteration of the thresholding search
            from ..device_ops import _bits_of
            rb_u, _, kp = _bits_of(np.asarray(self.U)[lo:hi] > u, B.m_pad)
            _, cb_v, _ = _bits_of(np.asarray(self.V) > v, B.n_pad)
            with torch.cuda.device(B.device):
                ub = torch.from_numpy(rb_u).to(B.device)
                vcb = torch.from_numpy(np.ascontiguousarray(cb_v)).to(B.device)
                cnt = torch.zeros(2, dtype=torch.int64, device=B.device)
                check(lib.bmf_cover_count(ptr(B.bits), B.m_pad, B.ldx, B.n_pad // 32, ptr(ub), ptr(vcb), B.n_pad // 32, kp, ptr(cnt), None, _stream()),
                      "bmf_cover_count")
                tp, fp = (int(x) for x in cnt.cpu().numpy())
        else:
          with torch.cuda.device(B.device):
            # product bits of the thresholded factors, then TP = |X & pd|, FP = |~X & pd| by popcount
            pd = boolean_product_bits(np.asarray(self.U)[lo:hi] > u, self.V > v, B.device)
            pdb = torch.zeros_like(B.bits)
            r, c = min(pd.shape[0], pdb.shape[0]), min(pd.shape[1], pdb.shape[1])
            pdb[:r, :c] = pd[:r, :c]
            cnt = torch.zeros(2, dtype=torch.int64, device=B.device)
            tp_bits, fp_bits = B.bits & pdb, pdb & ~B.bits
            check(lib.bmf_popcount(ptr(tp_bits), B.m_pad, B.ldx, B.ldx, ptr(cnt[0:1]), _stream()), "bmf_popcount")
            check(lib.bmf_popcount(ptr(fp_bits), B.m_pad, B.ldx, B.ldx, ptr(cnt[1:2]), _stream()), "bmf_popcount")
            tp, fp = (int(x) for x in cnt.cpu().numpy())
        tp, fp, fn = (int(round(x)) for x in self._sum_over_ranks([tp, fp, B.sum_local - tp]))   # exact: counts < 2^53
        return tp, fp, fn, self.m * self.n - tp - fp - fn

    def _real_confusion(self, Ub, Vb):
        """(TP, FP, FN, TN, (sum gt, sum pd, cells)) of the real-valued training matrix against the Boolean product of the thresholded
        factors Ub, Vb (bool arrays): the reference's arithmetic on two csr matrices (utils/metrics.py:56-77), bmf_real_confusion."""
        import torch
        from .._lib import lib, check, ptr
        from ..device_ops import _bits_of
        from ..engine import _stream
        R = self._real
        rb_u, _, _ = _bits_of(Ub, R.m_pad)
        rb_v, _, _ = _bits_of(Vb, R.n_pad)
        with torch.cuda.device(R.device):
            ub, vb = torch.from_numpy(rb_u).to(R.device), torch.from_numpy(rb_v).to(R.device)
            out = torch.zeros(6, dtype=torch.float64, device=R.device)
            check(lib.bmf_real_confusion(ptr(R.X), R.n_pad, self.m, self.n, ptr(ub), ptr(vb), ptr(out), _stream()), "bmf_real_confusion")
            c = [float(x) for x in out.cpu().numpy()]
        return c[0], c[1], c[2], c[3], (c[4], c[5], float(self.m) * float(self.n))

    @staticmethod
    def _counts_of(engine_counts, cells):
        """The engine's count tuple as the models carry it: 4 integers (Boolean X), or 4 real sums + (sum gt, sum pd, cells)."""
        if engine_counts is not None and len(engine_counts) == 6:
            c = engine_counts
            return c[0], c[1], c[2], c[3], (c[4], c[5], float(cells))
        return engine_counts

    def _wide_engine(self, mode):
        """The two-block engines for a rank 64 < k <= 128 (pybmf_amd/wide.py): one GPU, Boolean data; the all-ones mask on the
        re-associated dense path, W = 'mask' / a weight matrix on the cell lists.  X_val / X_test are scored like at k <= 64 (the scorers
        take the block lists)."""
        from ..wide import MAX_K_WIDE, WideMaskedMUEngine, WideMUEngine
        if self.k > MAX_K_WIDE:
            raise NotImplementedError(f"k={self.k}: this build supports k <= {MAX_K_WIDE}")
        if self._sharded or not self._boolean:
            raise NotImplementedError(f"k={self.k}: a rank above {L.MAX_KP} runs on one GPU, on Boolean (0/1) data")
        if getattr(self, "_obs", None) is not None:
            return WideMaskedMUEngine(self._obs, self.k, mode, bits=self._bits, with_mae=self.with_mae)
        return WideMUEngine(self._bits, self.k, mode, with_mae=self.with_mae)

    def _residual_sums(self):
        import torch
        from .._lib import lib, check, ptr
        from ..engine import _stream, round_up
        B = self._bits
        if self.k > L.MAX_KP:   # two 64-column blocks per factor, one fp16 product per cell (bmf_resid_sums_wide)
            from ..wide import BK
            lo, hi = getattr(self, "_rows", (0, self.m))
            with torch.cuda.device(B.device):
                blocks = []
                for F, rows, rows_pad in ((np.asarray(self.U)[lo:hi], hi - lo, B.m_pad), (np.asarray(self.V), self.n, B.n_pad)):
                    for c0 in (0, BK):
                        t = torch.zeros((rows_pad, BK), dtype=torch.float32, device=B.device)
                        t[:rows, : min(BK, self.k - c0)] = torch.from_numpy(np.ascontiguousarray(F[:, c0:c0 + BK], dtype=np.float32)).to(B.device)
                        blocks.append(t)
                ws = torch.zeros(((B.m_pad + B.n_pad) * 2 * BK,), dtype=torch.int16, device=B.device)
                sums = torch.zeros(2, dtype=torch.float64, device=B.device)
                check(lib.bmf_resid_sums_wide(ptr(B.bits_t), B.ldxt, B.m_pad, B.n_pad, ptr(blocks[0]), ptr(blocks[1]), ptr(blocks[2]), ptr(blocks[3]),
                                              ptr(ws), ptr(sums), 0, _stream()), "bmf_resid_sums_wide")
                s = sums.cpu().numpy()
            s_abs, s_sq = self._sum_over_ranks([s[0], s[1]])
            return s_abs, s_sq
        kp = 32 if self.k <= 32 else 64
        if not self._boolean:
            R = self._real
            with torch.cuda.device(R.device):
                Ud = torch.zeros((R.m_pad, kp), dtype=torch.float32, device=R.device)
                Vd = torch.zeros((R.n_pad, kp), dtype=torch.float32, device=R.device)
                Ud[: self.m, : self.k] = torch.from_numpy(np.ascontiguousarray(self.U, dtype=np.float32)).to(R.device)
                Vd[: self.n, : self.k] = torch.from_numpy(np.ascontiguousarray(self.V, dtype=np.float32)).to(R.device)
                sums = torch.zeros(4, dtype=torch.float64, device=R.device)
                check(lib.bmf_residual_sums_f32(ptr(R.X), R.m_pad, R.n_pad, self.m, self.n, ptr(Ud), ptr(Vd), kp, ptr(sums), _stream()),
                      "bmf_residual_sums_f32")
                s = sums.cpu().numpy()
            return float(s[0]), float(s[1])
        with torch.cuda.device(B.device):
            Ud = torch.zeros((B.m_pad, kp), dtype=torch.float32, device=B.device)
            Vd = torch.zeros((B.n_pad, kp), dtype=torch.float32, device=B.device)
            lo, hi = getattr(self, "_rows", (0, self.m))
            Ud[: hi - lo, : self.k] = torch.from_numpy(np.ascontiguousarray(np.asarray(self.U)[lo:hi], dtype=np.float32)).to(B.device)
            Vd[: self.n, : self.k] = torch.from_numpy(np.ascontiguousarray(self.V, dtype=np.float32)).to(B.device)
            sums = torch.zeros(4, dtype=torch.float64, device=B.device)
            check(lib.bmf_residual_sums(ptr(B.bits), B.m_pad, B.ldx, B.m, B.n, ptr(Ud), ptr(Vd), kp, ptr(sums), None, _stream()),
                  "bmf_residual_sums")
            s = sums.cpu().numpy()
        s_abs, s_sq = self._sum_over_ranks([s[0], s[1]])
        return s_abs, s_sq


class _PadDims:
    """The shape bookkeeping of an engine.BitMatrix without the bits: what a real-valued fit hands to code that only needs the padded
    shape and the device."""

    def __init__(self, m, n, device):
        from ..engine import round_up
        self.m, self.n, self.device = int(m), int(n), device
        self.m_pad, self.n_pad = round_up(max(self.m, 1), L.ROW_PAD), round_up(self.n, L.ROW_PAD)


def unique_values_mapping(arr):
    """Map every value to (its rank among the distinct values) / (number of distinct values), i.e. onto an arithmetic
    sequence in [0, 1) (PyBMF/models/ContinuousModel.py:225-231)."""
    arr = np.asarray(arr, dtype=np.float64)
    uniq, inverse = np.unique(arr, return_inverse=True)
    return (inverse.reshape(arr.shape) / len(uniq)).astype(np.float64)
