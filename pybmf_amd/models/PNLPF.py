"""PNLPF -- BinaryMFPenalty with a sigmoid link on the product (post-nonlinear penalty function), on the GPU.

Drop-in for ``PyBMF.models.PNLPF`` (``PyBMF/models/PNLPF.py``):

    min 1/2 ||X - sigmoid(link_lamda (U V^T - 1/2))||_F^2 + 1/2 reg ||U^2 - U||_F^2 + 1/2 reg ||V^2 - V||_F^2

The loop is the inherited ``BinaryMFPenalty._fit`` (BinaryMFPenalty.py:61-115); the two updates (:61-91) cannot be
re-associated, so each is one tile-fused pass over X (csrc/link.hip, ``bmf_link_pass``) followed by the shared fp64
epilogue.  Under a mask / weight matrix the contractions run over the observed cells instead (csrc/masked.hip,
``bmf_masked_link_pass``).
"""
from __future__ import annotations

import numpy as np

from .. import _lib as L
from .BinaryMFPenalty import BinaryMFPenalty


class PNLPF(BinaryMFPenalty):
    def __init__(self, k, U=None, V=None, W='full', reg=2.0, beta_loss="frobenius", solver="mu", link_lamda=10, reg_growth=3,
                 max_reg=1e10, tol=0.01, min_diff=0.0, max_iter=100, init_method='custom', normalize_method='balance', seed=None):
        self.check_params(k=k, U=U, V=V, W=W, reg=reg, beta_loss=beta_loss, solver=solver, link_lamda=link_lamda,
                          reg_growth=reg_growth, max_reg=max_reg, tol=tol, min_diff=min_diff, max_iter=max_iter,
                          init_method=init_method, normalize_method=normalize_method, seed=seed)

    def _link_engine(self):
        from ..engine import LinkMUEngine, MaskedMUEngine
        if getattr(self, "_obs", None) is not None:
            # W = 'mask' on a csr with unstored cells, or a weight matrix: both contractions of an update run over the observed cells
            # (multiply(W, multiply(X, d_sig)) @ V and multiply(W, multiply(sig, d_sig)) @ V, PNLPF.py:65-68,81-84), rec_error over
            # them too (the inherited error(): 0.5 sum W o (X - sigmoid(S))^2); RMSE / MAE / Boolean scores stay whole-matrix
            if not self._boolean and not self._all_cells:
                raise NotImplementedError("PNLPF on real-valued data under a mask / weight matrix: whole-matrix scores against the link "
                                          "prediction need every cell (W='full'), or Boolean data")
            return MaskedMUEngine(self._obs, self.k, L.MODE_PENALTY, bits=self._bits if self._boolean else None,
                                  real=None if self._boolean else self._real, real_counts=not self._boolean, all_cells=self._all_cells,
                                  link=L.LINK_SIGMOID, lamda=float(self.link_lamda), sharded=self._sharded, m_total=self.m)
        return LinkMUEngine(self._bits, self.k, L.LINK_SIGMOID, L.MODE_PENALTY, lamda=float(self.link_lamda), sharded=self._sharded)

    def _fit(self):
        if getattr(self, "task", None) is None:
            raise AttributeError(f"'{type(self).__name__}' object has no attribute 'task'")
        eng = self._eng = self._link_engine()
        lo, hi = self._rows
        eng.load_factors(self.U[lo:hi], self.V)
        eng.prepare()
        rows = []
        n_iter = 0
        # X_val / X_test, and the training entries under task='prediction': scored every iteration like the training matrix (the
        # inherited loop, BinaryMFPenalty.py:71,97 -> BaseModel.evaluate :209-257), RMSE / MAE against the LINK prediction (:51-58)
        extras = [] if self._scorers else None

        def log_row(it, reg):
            err, rec, rg, rmse, mae, cnt = eng.scalars(reg)
            if extras is not None:
                extras.append(self._engine_scores(eng, link=L.LINK_SIGMOID, lamda=float(self.link_lamda)))
            r = np.zeros(L.LOG_COLS)
            r[[L.LOG_ITER, L.LOG_ERROR, L.LOG_REC, L.LOG_REG, L.LOG_REGERR, L.LOG_RMSE, L.LOG_MAE]] = it, err, rec, reg, rg, rmse, mae
            r[L.LOG_TP:L.LOG_TN + 1] = cnt[:4]
            if len(cnt) == 6:
                from .BinaryMFPenalty import LOG_SUM_GT, LOG_SUM_PD
                r[LOG_SUM_GT], r[LOG_SUM_PD] = cnt[4], cnt[5]
            rows.append(r)
            return rg
        if extras is None and eng.can_pipeline():
            # Whole iterations enqueued by one C call each (bmf_link_iterate); iteration t + 1 is enqueued BEFORE the scalars of t are
            # read, so the device never waits for the host (BinaryMFPenalty._fit_masked has the same loop on the masked kernels).  The
            # loop runs one iteration past its stopping rule; the engine keeps the iterate before.  Same rows, same decisions.
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

    def get_prediction(self):
        return get_prediction_with_sigmoid(U=self.U, V=self.V, link_lamda=self.link_lamda)

    def update_U(self):
        self.U = update_U(X=self._X_input, W=None, U=self.U, V=self.V, reg=self.reg, link_lamda=self.link_lamda)

    def update_V(self):
        self.V = update_V(X=self._X_input, W=None, U=self.U, V=self.V, reg=self.reg, link_lamda=self.link_lamda)


def get_prediction_with_sigmoid(U, V, link_lamda):
    """sigmoid(link_lamda (U V^T - 1/2)) as a dense host array (PNLPF.py:54-58); end-of-fit utility, not part of the loop."""
    from ..device_ops import real_product
    S = (real_product(U, V) - 0.5) * link_lamda
    out = np.empty_like(S)
    pos = S >= 0
    out[pos] = 1.0 / (1.0 + np.exp(-S[pos]))
    e = np.exp(S[~pos])
    out[~pos] = e / (1.0 + e)
    return out


def _one_step(X, W, U, V, reg, link_lamda, which):
    from ..engine import BitMatrix, LinkMUEngine
    from .BinaryMFPenalty import _is_full
    from .ContinuousModel import ContinuousModel
    ContinuousModel._check_boolean(X)
    U, V = np.asarray(U, dtype=np.float64), np.asarray(V, dtype=np.float64)
    if not _is_full(W):   # a mask / weight matrix: the contractions over the cells with W != 0 (PNLPF.py:65-68,81-84)
        from ..device_ops import MaskedOneStep
        step = MaskedOneStep(X, W, U, V, mode=L.MODE_PENALTY, link=L.LINK_SIGMOID, lamda=float(link_lamda))
        return step.update_U(float(reg)) if which == "U" else step.update_V(float(reg))
    eng = LinkMUEngine(BitMatrix(X, "cuda:0"), U.shape[1], L.LINK_SIGMOID, L.MODE_PENALTY, lamda=float(link_lamda))
    eng.load_factors(U, V)
    eng.prepare()
    import torch
    with torch.cuda.device(eng.device):
        eng._side(which, float(reg))
    return eng.factors()[0 if which == "U" else 1]


def update_U(X, W, U, V, reg, link_lamda, solver='mu', beta_loss='frobenius'):
    """One multiplicative update of U on the GPU (PyBMF/models/PNLPF.py:61-75)."""
    return _one_step(X, W, U, V, reg, link_lamda, "U")


def update_V(X, W, U, V, reg, link_lamda, solver='mu', beta_loss='frobenius'):
    """One multiplicative update of V on the GPU (PyBMF/models/PNLPF.py:77-91)."""
    return _one_step(X, W, U, V, reg, link_lamda, "V")
