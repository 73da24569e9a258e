"""BinaryMFPenalty -- penalty-function Boolean matrix factorisation by multiplicative updates, on the GPU.

Drop-in for ``PyBMF.models.BinaryMFPenalty`` (``PyBMF/models/BinaryMFPenalty.py``): same constructor, ``fit()``,
attributes (``U, V, X_pd, logs['updates'], logs['boolean'], reg, ...``) and module-level ``update_U / update_V /
error / rec_error / reg_error``.  The loop of ``_fit`` (:61-115) runs as one enqueued sequence of HIP kernels per
iteration (engine.MUEngine / csrc/api.hip); its early-stop rule is evaluated on the device; the two log tables are
assembled once at the end from the device log.

    min 1/2 ||X - U V^T||_F^2 + 1/2 reg ||U^2 - U||_F^2 + 1/2 reg ||V^2 - V||_F^2        (Zhang et al.)
"""
from __future__ import annotations

import numpy as np

from .. import _lib as L
from ..utils import header, record_many, scores_from_counts
from .ContinuousModel import ContinuousModel


LOG_SUM_GT, LOG_SUM_PD = 14, 15   # spare columns of a log row (L.LOG_COLS = 16): sum gt, sum pd of a real-valued training matrix


class BinaryMFPenalty(ContinuousModel):
    def __init__(self, k, U=None, V=None, W='full', beta_loss="frobenius", solver="mu", reg=2.0, reg_growth=3, max_reg=1e10,
                 tol=0.01, min_diff=0.0, max_iter=100, init_method='custom', normalize_method='balance', seed=None):
        self.check_params(k=k, U=U, V=V, W=W, reg=reg, beta_loss=beta_loss, solver=solver, reg_growth=reg_growth,
                          max_reg=max_reg, tol=tol, min_diff=min_diff, max_iter=max_iter, init_method=init_method,
                          normalize_method=normalize_method, seed=seed)

    def check_params(self, **kwargs):
        super().check_params(**kwargs)
        assert self.beta_loss in ['frobenius']
        assert self.solver in ['mu']
        assert self.init_method in ['normal', 'uniform', 'custom']
        assert self.normalize_method in ['balance', None]
        self.reg, self.reg_growth, self.max_reg = np.float64(self.reg), np.float64(self.reg_growth), np.float64(self.max_reg)

    def fit(self, X_train, X_val=None, X_test=None, **kwargs):
        super().fit(X_train, X_val, X_test, **kwargs)
        self._fit()
        self.X_pd = None  # Boolean product at (0.5, 0.5), built on first access (see BaseModel.X_pd)
        self.finish(show_logs=self.show_logs, save_model=self.save_model, show_result=self.show_result)

    def _make_X_pd(self):
        from ..device_ops import boolean_product_csr
        return boolean_product_csr(self.U, self.V, u=0.5, v=0.5, device=self.device)

    # ---- the loop ---------------------------------------------------------------------------------------------
    def _engine(self, mode=L.MODE_PENALTY):
        from ..engine import MUEngine
        return MUEngine(self._bits, k=self.k, mode=mode, terms=self.terms, with_mae=self.with_mae, thr=(0.5, 0.5), panel=self.panel,
                        tol=float(self.tol), min_diff=float(self.min_diff), max_iter=int(self.max_iter), sharded=self._sharded)

    def _fit(self):
        """Multiplicative updates of V then U (Gauss-Seidel), log rows 0 .. n_iter, geometric growth of `reg`."""
        if getattr(self, "task", None) is None:
            raise AttributeError(f"'{type(self).__name__}' object has no attribute 'task'")
        if getattr(self, "_obs", None) is not None and self.k <= L.MAX_KP:
            return self._fit_masked()
        if self.k > L.MAX_KP:   # two 64-column blocks per factor (pybmf_amd/wide.py), on the masked loop's protocol (mask or not)
            return self._fit_masked(self._wide_engine(L.MODE_PENALTY))
        eng = self._eng = self._engine()
        lo, hi = self._rows
        eng.load_factors(self.U[lo:hi], self.V)
        # reg used by update t is reg0 * growth^(t-1), capped (BinaryMFPenalty.py:115); computed like the reference does
        regs, r = [], self.reg
        for _ in range(self.max_iter + 1):
            regs.append(float(r))
            r = min(r * self.reg_growth, self.max_reg)
        eng.prepare(regs[0])
        extras = None
        if not self._scorers:
            eng.run(regs, it0=1)            # the whole loop is enqueued by one C call
        else:                               # extra data sets to score per iteration: step from Python
            extras = [self._engine_scores(eng)]
            for it in range(1, len(regs) + 1):
                eng.run([regs[it - 1]], it0=it)
                extras.append(self._engine_scores(eng))
                if int(eng.stop.item()):
                    break
        try:
            log, stop = eng.read_log()
            U_local, self.V = eng.factors()
        finally:
            eng.close()   # (a host-callback communicator is freed here; an RCCL one stays in the module cache: engine.shutdown_comms)
        self.U = self._gather_rows(U_local)
        n_iter = int(log[-1, L.LOG_ITER])
        self._log_to_frames(log, extras)
        self._stop_reason(log[-1], n_iter)
        # self.reg after fit is one growth step past the last update (the reference grows it after early_stop)
        r = self.reg
        for _ in range(n_iter):
            r = min(r * self.reg_growth, self.max_reg)
        self.reg = r
        self.n_iter = n_iter

    def _fit_masked(self, eng=None):
        """Same loop on the masked kernels (W = 'mask' / weights): contractions over the observed cells only.  `eng`: another engine
        with the prepare / update / scalars protocol (the two-block engine of a rank above 64)."""
        from ..engine import MaskedMUEngine
        if eng is None:
            # (a real-valued X: scores from the fp32 copy and the reference's arithmetic "confusion sums" instead of bit kernels)
            eng = MaskedMUEngine(self._obs, self.k, L.MODE_PENALTY, bits=self._bits if self._boolean else None,
                                 real=None if self._boolean else self._real, real_counts=not self._boolean, all_cells=self._all_cells,
                                 with_mae=self.with_mae, sharded=self._sharded, m_total=self.m)
        self._eng = eng
        lo, hi = self._rows
        eng.load_factors(self.U[lo:hi], self.V)
        eng.prepare()
        rows = []
        extras = [] if self._scorers else None
        n_iter = 0

        def log_row(it, reg):
            err, rec, rg, rmse, mae, cnt = eng.scalars(reg)
            if extras is not None:
                extras.append(self._engine_scores(eng))
            r = np.zeros(L.LOG_COLS)
            r[[L.LOG_ITER, L.LOG_ERROR, L.LOG_REC, L.LOG_REG, L.LOG_REGERR, L.LOG_RMSE, L.LOG_MAE]] = it, err, rec, reg, rg, rmse, mae
            r[L.LOG_TP:L.LOG_TN + 1] = cnt[:4]
            if len(cnt) == 6:   # real-valued X: sum gt, sum pd ride in the spare columns (see _log_to_frames)
                r[LOG_SUM_GT], r[LOG_SUM_PD] = cnt[4], cnt[5]
            rows.append(r)
            return rg
        if extras is None and getattr(eng, "can_pipeline", lambda: False)():
            # Whole iterations enqueued by one C call each (bmf_masked_iterate); iteration t + 1 is enqueued BEFORE the scalars of t are
            # read, so the device never waits for the host.  The loop runs one iteration past its stopping rule; the engine keeps the
            # iterate before (the regulariser schedule does not depend on the scalars).  Same rows, same decisions.
            def row(it, reg, h):
                err, rec, rg, rmse, mae, cnt = h
                r = np.zeros(L.LOG_COLS)
                r[[L.LOG_ITER, L.LOG_ERROR, L.LOG_REC, L.LOG_REG, L.LOG_REGERR, L.LOG_RMSE, L.LOG_MAE]] = it, err, rec, reg, rg, rmse, mae
                r[L.LOG_TP:L.LOG_TN + 1] = cnt
                rows.append(r)
                return rg
            reg = float(self.reg)
            eng.iterate(0, reg, update=False)
            eng.iterate(1, reg)
            rg_old = row(0, reg, eng.row(0, reg))
            while True:
                n_iter += 1
                reg_next = min(reg * self.reg_growth, self.max_reg)
                eng.iterate(n_iter + 1, reg_next)
                rg = row(n_iter, reg, eng.row(n_iter, reg))
                diff = abs(rg_old - rg)
                rg_old = rg
                improving = self.early_stop(error=rg_old, diff=diff, n_iter=n_iter, verbose=False)
                self.reg = reg_next
                if not improving:
                    break
                reg = reg_next
            U_local, self.V = eng.previous_factors()
            eng.load_factors(U_local, self.V)
        else:
            rg_old = log_row(0, float(self.reg))
            improving = True
            while improving:
                n_iter += 1
                eng.update(float(self.reg))
                rg = log_row(n_iter, float(self.reg))
                diff = abs(rg_old - rg)
                rg_old = rg
                improving = self.early_stop(error=rg_old, diff=diff, n_iter=n_iter, verbose=False)
                self.reg = min(self.reg * self.reg_growth, self.max_reg)
            U_local, self.V = eng.factors()
        self.U = self._gather_rows(U_local)
        log = np.array(rows)
        self._log_to_frames(log, extras)
        self.early_stop(error=float(log[-1, L.LOG_REGERR]), diff=self._last_diff, n_iter=n_iter)
        self.n_iter = n_iter

    def _stop_reason(self, last, n_iter):
        self.early_stop(error=float(last[L.LOG_REGERR]), diff=self._last_diff, n_iter=n_iter)

    def _log_to_frames(self, log, extras=None):
        """logs['updates'] / logs['boolean'] with the reference's 3-level columns.  `extras[i]` holds, per
        extra data set, ((RMSE, MAE), (TP, FP, FN, TN)) of log row i (val / test, and train under task='prediction')."""
        self._check_nan(log[:, [L.LOG_ERROR, L.LOG_REC, L.LOG_REGERR]])
        rg = log[:, L.LOG_REGERR]
        self._last_diff = abs(rg[-2] - rg[-1]) if len(rg) > 1 else None
        self.counts = []
        urows, brows = [], []
        for i, row in enumerate(log):
            head = {'iter': int(row[L.LOG_ITER]), 'error': row[L.LOG_ERROR], 'rec_error': row[L.LOG_REC],
                    'reg': float(row[L.LOG_REG]), 'reg_error': row[L.LOG_REGERR]}
            if self._boolean:
                cnt_train = tuple(int(row[c]) for c in (L.LOG_TP, L.LOG_FP, L.LOG_FN, L.LOG_TN))
            else:   # real-valued X: the reference's arithmetic sums, with their own denominators (utils.scores_from_counts)
                cnt_train = tuple(float(row[c]) for c in (L.LOG_TP, L.LOG_FP, L.LOG_FN, L.LOG_TN)) + (
                    (float(row[LOG_SUM_GT]), float(row[LOG_SUM_PD]), float(self.m) * float(self.n)),)
            sets = {'train': ((row[L.LOG_RMSE], row[L.LOG_MAE]), cnt_train)}
            if extras is not None:
                sets.update(extras[i])
            names = [nm for nm in ('train', 'val', 'test') if nm in sets]
            cols, vals = header(list(head.keys()), levels=3), list(head.values())
            bcols, bvals = [], []
            for nm in names:
                (rmse, mae), cnt = sets[nm]
                cols += [(nm, 0, 'RMSE'), (nm, 0, 'MAE')]
                vals += [rmse, mae]
                bcols += [(nm, 0, mt) for mt in ('Recall', 'Precision', 'Accuracy', 'F1')]
                bvals += list(scores_from_counts(*cnt))
            urows.append(vals)
            brows.append(bvals)
            self.counts.append(sets['train'][1])
        record_many(self.logs, 'updates', cols, urows)   # (the same columns on every row of one fit)
        record_many(self.logs, 'boolean', bcols, brows)

    def get_prediction(self):
        from ..device_ops import product_csr
        return product_csr(self.U, self.V, boolean=False, device=self.device)

    def update_U(self):
        self.U = update_U(X=self._X_input, W=None, U=self.U, V=self.V, reg=self.reg)

    def update_V(self):
        self.V = update_V(X=self._X_input, W=None, U=self.U, V=self.V, reg=self.reg)


# ---- module-level arithmetic, importable like the reference's (PNLPF does `from .BinaryMFPenalty import error, ...`) ----
def _is_full(W):
    if W is None:
        return True
    if hasattr(W, "nnz"):
        return bool(W.nnz == W.shape[0] * W.shape[1] and (W.data == 1).all())
    return bool((np.asarray(W) == 1).all())


def _check_full(W, X):
    """For the callers that only have the all-ones-mask kernels (PNLPF's link passes)."""
    if not _is_full(W):
        raise NotImplementedError("only the all-ones mask (W='full') is supported")


def _one_step(X, W, U, V):
    """The all-ones mask (W None or every entry 1) takes the re-associated dense path, anything else the masked one."""
    from ..device_ops import MaskedOneStep, OneStep
    if not _is_full(W):
        return MaskedOneStep(X, W, U, V)
    return OneStep(X, U, V)


def update_U(X, W, U, V, reg, solver='mu', beta_loss='frobenius'):
    """One multiplicative update of U on the GPU (PyBMF/models/BinaryMFPenalty.py:136-148)."""
    return _one_step(X, W, U, V).update_U(float(reg))


def update_V(X, W, U, V, reg, solver='mu', beta_loss='frobenius'):
    """One multiplicative update of V on the GPU (PyBMF/models/BinaryMFPenalty.py:151-163)."""
    return _one_step(X, W, U, V).update_V(float(reg))


def error(X_gt, X_pd, W, U, V, reg):
    """(error, rec_error, reg_error) (BinaryMFPenalty.py:166-172).  With ``X_pd=None`` the reconstruction term comes from the
    factors (trace form; U V^T is never materialised).  An explicit ``X_pd`` is honoured as the reference honours it -- PNLPF's
    inherited loop passes its sigmoid-link prediction here (PNLPF.py:1,50-58), which is NOT U V^T: rec_error is then
    0.5 * sum(W o (X_gt - X_pd)^2) of the matrices given, summed on the device."""
    if X_pd is None:
        return tuple(_one_step(X_gt, W, U, V).errors(float(reg)))
    rec = rec_error(X_gt, X_pd, W)
    reg_err = float(reg) * (reg_error(U) + reg_error(V))
    return rec + reg_err, rec, reg_err


def rec_error(X_gt, X_pd, W, U=None, V=None):
    """0.5 * sum(W o (X_gt - X_pd)^2) (BinaryMFPenalty.py:175-179).  ``X_pd`` may be None when the factors of the prediction are
    given instead (``U=, V=``: trace form, the product is never formed)."""
    if X_pd is None:
        if U is None or V is None:
            raise TypeError("rec_error needs X_pd, or the factors U= and V= of the prediction")
        return _one_step(X_gt, W, U, V).errors(0.0)[1]
    from ..device_ops import weighted_sqdiff
    Wd = None if _is_full(W) else (W if hasattr(W, "toarray") else np.asarray(W))
    return 0.5 * weighted_sqdiff(X_gt, X_pd, Wd)


def reg_error(X):
    """0.5 * sum((X^2 - X)^2) of one factor (BinaryMFPenalty.py:182-186); O(rows * k) host arithmetic."""
    X = np.asarray(X, dtype=np.float64)
    return float(0.5 * np.sum(np.power(np.power(X, 2) - X, 2)))
