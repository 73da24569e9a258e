"""WNMF -- weighted non-negative matrix factorisation by multiplicative updates (Frobenius loss), on the GPU.

Drop-in for ``PyBMF.models.WNMF`` (``PyBMF/models/WNMF.py``), Frobenius loss.  With the all-ones mask (W='full', or
W='mask' on a matrix whose stored pattern is the whole matrix) a Boolean X runs on the bit kernels (same engine as
BinaryMFPenalty with the WNMF update rule) and a real-valued X on the fp32-MFMA GEMM; with a proper mask (W='mask' on a
csr with unstored cells, or a weight matrix) the contractions run over the observed cells (engine.MaskedMUEngine).
The Kullback-Leibler loss (W='full' or 'mask', Boolean X) runs on the tile-fused link kernels (csrc/link.hip).

Reference quirk not reproduced: ``WNMF.error`` (:133-144) overwrites exact zeros of X_train and of U V^T with eps in
place before taking the difference, and the updates that follow see those eps values.  Wherever a row or column has any
observed non-zero cell that is an O(1e-16) perturbation, far below the 1e-4 gate.  It decides the result only for a block
of observed cells that are all zero and share no row or column with observed non-zero data (possible under a sparse mask):
the reference leaves such rows / columns at arbitrary O(1) / O(eps) values whose products are ~eps (the rank-one fit of an
all-eps block, which needs the fp64 exponent range), here they collapse to exactly 0 -- the predictions agree to 1e-16.
"""
from __future__ import annotations

import numpy as np

from .. import _lib as L
from ..utils import header, record_many
from .ContinuousModel import ContinuousModel


class WNMF(ContinuousModel):
    def __init__(self, k, U=None, V=None, W='mask', beta_loss='frobenius', init_method='normal', solver='mu', tol=0.0,
                 min_diff=0.0, max_iter=30, seed=None):
        self.check_params(k=k, U=U, V=V, W=W, beta_loss=beta_loss, init_method=init_method, solver=solver, tol=tol,
                          min_diff=min_diff, max_iter=max_iter, seed=seed)

    def check_params(self, **kwargs):
        super().check_params(**kwargs)
        assert self.beta_loss in ['frobenius', 'kullback-leibler']
        assert self.solver in ['mu']
        assert self.init_method in ['uniform', 'normal', 'custom']

    def fit(self, X_train, X_val=None, X_test=None, **kwargs):
        super().fit(X_train, X_val, X_test, **kwargs)
        self._fit()
        self.X_pd = None  # real-valued U V^T, built on first access
        self.finish(show_logs=self.show_logs, save_model=self.save_model, show_result=self.show_result)

    def _make_X_pd(self):
        from ..device_ops import product_csr
        return product_csr(self.U, self.V, boolean=False, device=self.device)

    # Reference behaviour kept: WNMF.error() (:133-144) writes eps into every zero cell of the (densified) X_train before the
    # first evaluate(), so under task='prediction' every cell of the training matrix is an "entry" and the train scores
    # are whole-matrix scores (val / test are untouched and scored at their non-zero entries).
    _train_scored_by_entries = False

    def _to_device(self):
        from ..engine import BitMatrix, RealMatrix
        X = self._X_input
        host = np.asarray(X.todense()) if hasattr(X, "todense") else X
        # Boolean-ness is decided from the VALUES, not the dtype: an integer matrix of ratings 1..5 is real-valued data (the
        # reference casts X to float64 and fits the values, WNMF.py:40-47); device tensors are inspected on the device
        import torch
        if isinstance(host, torch.Tensor):
            self._boolean = bool(((host == 0) | (host == 1)).all().item())
            if not self._boolean:
                host = host.detach().cpu().numpy()
        elif isinstance(host, np.ndarray) or isinstance(host, (list, tuple)) or hasattr(host, "__array__"):
            host = np.asarray(host)
            self._boolean = bool(host.dtype.kind in "biuf" and self._values_are_boolean(host))
        else:
            # a lazy row source (shape + row slicing, e.g. generators.PlantedBooleanOnDevice, which load_dataset keeps as it is):
            # np.asarray would make a 0-d object array of it.  Such sources produce bits by construction, as
            # ContinuousModel._values_are_boolean says for every other model.
            self._boolean = self._values_are_boolean(host)
        self._sharded, self._rows = False, (0, self.m)
        self._all_cells = False
        if self._boolean:
            self._shard_plan()
            lo, hi = self._rows
            self._bits = BitMatrix(X, self.device, row_lo=lo, row_hi=hi)
            self._x_mean = self._sum_over_ranks([self._bits.sum_local])[0] / (float(self.m) * float(self.n))
        else:
            if isinstance(getattr(self, "W", None), str) and self.W == "full":   # (the masked kernels shard Boolean matrices only)
                self._shard_plan()
            lo, hi = self._rows
            self._real = RealMatrix(host[lo:hi], self.device)
            self._x_mean = float(np.asarray(host, dtype=np.float64).mean())
            if self.beta_loss == 'kullback-leibler':   # no dense real-valued KL kernels: the cells as a list (init_W, _fit_kl)
                self._sharded, self._rows = False, (0, self.m)
                self._real = RealMatrix(host, self.device)
                self._real_host = np.ascontiguousarray(host, dtype=np.float64)

    def _wants_cell_list(self):
        return not self._boolean and self.beta_loss == 'kullback-leibler'

    def _fit(self):
        if getattr(self, "task", None) is None:
            raise AttributeError(f"'{type(self).__name__}' object has no attribute 'task'")
        self._extras = []
        if self.beta_loss == 'kullback-leibler':
            rows = self._fit_kl()
        elif getattr(self, "_obs", None) is not None and self.k <= L.MAX_KP:
            rows = self._fit_masked()
        elif self.k > L.MAX_KP:   # two 64-column blocks per factor (pybmf_amd/wide.py), W = 'full' or a mask / weights; Boolean data
            rows = self._fit_masked(self._wide_engine(L.MODE_WNMF))
        else:
            rows = self._fit_boolean() if self._boolean else self._fit_real()
        self._check_nan([r[1] for r in rows])
        extras = self._extras if self._scorers else None
        lrows = []
        for i, (it, err, rmse, mae) in enumerate(rows):
            head = {'iter': int(it), 'error': err}
            sets = {'train': (rmse, mae)}
            if extras is not None:
                sets.update({nm: rm for nm, (rm, _) in extras[i].items()})
            cols, vals = header(list(head.keys()), levels=3), list(head.values())
            for nm in ('train', 'val', 'test'):
                if nm in sets:
                    cols += [(nm, 0, 'RMSE'), (nm, 0, 'MAE')]
                    vals += list(sets[nm])
            lrows.append(vals)
        record_many(self.logs, 'updates', cols, lrows)
        self.n_iter = int(rows[-1][0])
        diff = abs(rows[-2][1] - rows[-1][1]) if len(rows) > 1 else None
        self.early_stop(error=rows[-1][1], diff=diff, n_iter=self.n_iter)

    def _fit_boolean(self):
        from ..engine import MUEngine
        eng = self._eng = MUEngine(self._bits, k=self.k, mode=L.MODE_WNMF, terms=self.terms, with_mae=self.with_mae, panel=self.panel,
                                   tol=float(self.tol), min_diff=float(self.min_diff), max_iter=int(self.max_iter), sharded=self._sharded)
        lo, hi = self._rows
        eng.load_factors(self.U[lo:hi], self.V)
        eng.prepare(0.0)
        if not self._scorers:
            eng.run([0.0] * (self.max_iter + 1), it0=1)
        else:
            self._note(eng)
            for it in range(1, self.max_iter + 2):
                eng.run([0.0], it0=it)
                self._note(eng)
                if int(eng.stop.item()):
                    break
        try:
            log, _ = eng.read_log()
            U_local, self.V = eng.factors()
        finally:
            eng.close()
        self.U = self._gather_rows(U_local)
        return [(r[L.LOG_ITER], r[L.LOG_ERROR], r[L.LOG_RMSE], r[L.LOG_MAE]) for r in log]

    def _fit_real(self):
        from ..engine import RealMUEngine
        eng = self._eng = RealMUEngine(self._real, self.k, with_mae=self.with_mae, sharded=self._sharded, m_total=self.m)
        lo, hi = self._rows
        eng.load_factors(self.U[lo:hi], self.V)
        if not self._sharded and not self._scorers:
            # the whole loop on the device: one C call enqueues max_iter + 1 iterations, the stopping rule raises a device flag
            eng.device_loop(int(self.max_iter), tol=float(self.tol), min_diff=float(self.min_diff))
            eng.run(1, int(self.max_iter) + 2)
            log, _ = eng.read_log()
            self._check_nan(log[:, [L.LOG_ERROR]])
            self.U, self.V = eng.factors()
            return [(int(r[L.LOG_ITER]), float(r[L.LOG_ERROR]), float(r[L.LOG_RMSE]), float(r[L.LOG_MAE])) for r in log]
        rows = []
        n_iter = 0
        err_old, rmse, mae = eng.scalars()
        rows.append((n_iter, err_old, rmse, mae))
        self._note(eng)
        improving = True
        while improving:
            n_iter += 1
            eng.update()
            err, rmse, mae = eng.scalars()
            self._note(eng)
            diff = abs(err_old - err)
            err_old = err
            rows.append((n_iter, err, rmse, mae))
            improving = self.early_stop(error=err_old, diff=diff, n_iter=n_iter, verbose=False)
        U_local, self.V = eng.factors()
        self.U = self._gather_rows(U_local)
        return rows

    def _fit_masked(self, eng=None):
        """W = 'mask' (stored pattern) or weights: contractions over the observed cells (bmf_masked_pass).  `eng`: another engine with
        the prepare / update / scalars protocol (the two-block engine of a rank above 64)."""
        from ..engine import MaskedMUEngine
        if eng is None:
            eng = MaskedMUEngine(self._obs, self.k, L.MODE_WNMF, bits=self._bits if self._boolean else None,
                                 real=None if self._boolean else self._real, with_mae=self.with_mae,
                                 sharded=self._sharded, m_total=self.m)
        self._eng = eng
        lo, hi = self._rows
        eng.load_factors(self.U[lo:hi], self.V)
        eng.prepare()
        rows = []
        n_iter = 0
        if not self._scorers and getattr(eng, "can_pipeline", lambda: False)():
            # one C call per iteration, iteration t + 1 enqueued before the scalars of t are read (see BinaryMFPenalty._fit_masked)
            eng.iterate(0, 0.0, update=False)
            eng.iterate(1, 0.0)
            err_old, _, _, rmse, mae, _ = eng.row(0, 0.0)
            rows.append((n_iter, err_old, rmse, mae))
            while True:
                n_iter += 1
                eng.iterate(n_iter + 1, 0.0)
                err, _, _, rmse, mae, _ = eng.row(n_iter, 0.0)
                diff = abs(err_old - err)
                err_old = err
                rows.append((n_iter, err, rmse, mae))
                if not self.early_stop(error=err_old, diff=diff, n_iter=n_iter, verbose=False):
                    break
            U_local, self.V = eng.previous_factors()
            eng.load_factors(U_local, self.V)
            self.U = self._gather_rows(U_local)
            return rows
        err_old, _, _, rmse, mae, _ = eng.scalars(0.0)
        rows.append((n_iter, err_old, rmse, mae))
        self._note(eng)
        improving = True
        while improving:
            n_iter += 1
            eng.update(0.0)
            err, _, _, rmse, mae, _ = eng.scalars(0.0)
            self._note(eng)
            diff = abs(err_old - err)
            err_old = err
            rows.append((n_iter, err, rmse, mae))
            improving = self.early_stop(error=err_old, diff=diff, n_iter=n_iter, verbose=False)
        U_local, self.V = eng.factors()
        self.U = self._gather_rows(U_local)
        return rows

    def _fit_kl(self):
        """beta_loss='kullback-leibler' (WNMF.py:111-129, error :143-145): tile-fused (X / U V^T) V passes, Boolean X."""
        from ..engine import BitMatrix, LinkMUEngine
        if not self._boolean:
            # real-valued data (WNMF.py:111-129 on whatever it is given): numerators (W o X / U V^T) F over a list of cells -- every cell
            # under the all-ones mask, the cells with W != 0 under a weight matrix --, denominators = the column sums of the other
            # factor, the objective over the same cells (:143-145); RMSE / MAE of U V^T over the whole matrix (fp32 residual pass)
            if getattr(self, "_obs", None) is None or self._sharded:
                raise NotImplementedError("the Kullback-Leibler loss on real-valued data: W='full' or a weight matrix, one GPU")
            from ..engine import MaskedMUEngine
            return self._fit_masked(MaskedMUEngine(self._obs, self.k, L.MODE_WNMF, real=self._real, with_mae=self.with_mae, link=L.LINK_KL,
                                                   all_cells=self._all_cells, m_total=self.m))
        obs_bits = None
        if getattr(self, "_obs", None) is not None:
            # a weight matrix: the numerators (W o X / U V^T) F run over the cells with W != 0, the denominators O F are the column sums
            # of the other factor (WNMF.py:111-129), the objective sum W o (X log(X / UV) - X + UV) over those cells (:143-145)
            from ..engine import MaskedMUEngine
            return self._fit_masked(MaskedMUEngine(self._obs, self.k, L.MODE_WNMF, bits=self._bits, with_mae=self.with_mae, link=L.LINK_KL,
                                                   sharded=self._sharded, m_total=self.m))
        if getattr(self, "_mask_is_pattern", False):
            # W='mask': the stored pattern contains every non-zero of X, so W o X = X and the updates are those of the all-ones
            # mask (the reference's denominators use the all-ones matrix O, not W); only the objective is restricted to the
            # observed cells.
            from scipy.sparse import csr_matrix
            Xs = self.X_train.tocsr()
            pattern = csr_matrix((np.ones(Xs.nnz, dtype=np.uint8), Xs.indices, Xs.indptr), shape=Xs.shape)
            lo, hi = self._rows
            obs_bits = BitMatrix(pattern, self.device, row_lo=lo, row_hi=hi)
        eng = self._eng = LinkMUEngine(self._bits, self.k, L.LINK_KL, L.MODE_WNMF, obs_bits=obs_bits, sharded=self._sharded)
        lo, hi = self._rows
        eng.load_factors(self.U[lo:hi], self.V)
        eng.prepare()
        rows = []
        n_iter = 0
        if not self._scorers and eng.can_pipeline():
            # one C call per iteration (bmf_link_iterate), iteration t + 1 enqueued before the scalars of t are read (see
            # BinaryMFPenalty._fit_masked): the loop overshoots its stopping rule by one iteration and returns the iterate before
            eng.iterate(0, 0.0, update=False)
            eng.iterate(1, 0.0)
            err_old, _, _, rmse, mae, _ = eng.row(0, 0.0)
            rows.append((n_iter, err_old, rmse, mae))
            while True:
                n_iter += 1
                eng.iterate(n_iter + 1, 0.0)
                err, _, _, rmse, mae, _ = eng.row(n_iter, 0.0)
                diff = abs(err_old - err)
                err_old = err
                rows.append((n_iter, err, rmse, mae))
                if not self.early_stop(error=err_old, diff=diff, n_iter=n_iter, verbose=False):
                    break
            U_local, self.V = eng.previous_factors()
            eng.load_factors(U_local, self.V)
            self.U = self._gather_rows(U_local)
            return rows
        err_old, _, _, rmse, mae, _ = eng.scalars(0.0)
        rows.append((n_iter, err_old, rmse, mae))
        self._note(eng)
        improving = True
        while improving:
            n_iter += 1
            eng.update(0.0)
            err, _, _, rmse, mae, _ = eng.scalars(0.0)
            self._note(eng)
            diff = abs(err_old - err)
            err_old = err
            rows.append((n_iter, err, rmse, mae))
            improving = self.early_stop(error=err_old, diff=diff, n_iter=n_iter, verbose=False)
        U_local, self.V = eng.factors()
        self.U = self._gather_rows(U_local)
        return rows

    def _note(self, eng):
        """RMSE / MAE of the extra data sets (val / test; train under task='prediction') at the engine's current state."""
        if self._scorers:
            self._extras.append(self._engine_scores(eng, want_boolean=False))

    def update(self):
        raise NotImplementedError("WNMF.update() is folded into the device loop; call fit()")

    def error(self):
        """0.5 * || X - U V^T ||_F^2 for the current factors."""
        if self._boolean:
            from ..device_ops import OneStep
            return 0.5 * OneStep(self._X_input, self.U, self.V, mode=L.MODE_WNMF).residual_sums()[1]
        lo, hi = self._rows
        self._eng.load_factors(self.U[lo:hi], self.V)
        return self._eng.scalars()[0]

    def _residual_sums(self):
        if self._boolean:
            return super()._residual_sums()
        lo, hi = self._rows
        self._eng.load_factors(self.U[lo:hi], self.V)
        s_abs, s_sq = self._sum_over_ranks(self._eng._residual())
        return s_abs, s_sq
