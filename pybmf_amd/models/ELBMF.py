"""ELBMF -- elastic-net regularised Boolean matrix factorisation by proximal alternating linearised minimisation (PALM /
iPALM; Dalleiger & Vreeken, NeurIPS 2022), on the GPU.

Counterpart of ``PyBMF/models/ELBMF.py``: same constructor, the ``iPALM`` loop (:110-163) and the module-level
``update_U / prox / get_integrality_gap`` (:166-210).  The reference class does not run as shipped -- ``init_model`` calls
``normalize_UV(method="normalize")`` (:73) while ``ContinuousModel.normalize_UV`` takes no argument -- and the module is
commented out of ``PyBMF/models/__init__.py``; the loop and the step functions do run and are what is matched
(``tests/golden/g14_palm.*``, made by driving them with ``init_model``'s other steps done by hand).  Here ``init_model``
applies 'matrixwise-normalize' (both factors into [0, 1]) only to randomly initialised factors; custom factors are used as
given.

Per iteration (Jacobi: the V step sees the OLD U, ELBMF.py:124-125): one proximal gradient step per factor from X V,
X^T U and the two k x k Grams (``palm.PalmEngine``), then ||X - U V^T||_F^2 by the trace form, the integrality gap and
the Boolean scores of the factors thresholded at 0.5.
"""
from __future__ import annotations

import os

import numpy as np

from .. import _lib as L
from ..utils import header, record_many, scores_from_counts
from .ContinuousModel import ContinuousModel, EPS


class ELBMF(ContinuousModel):
    def __init__(self, k, U=None, V=None, W='full', init_method='custom', reg_l1=0.01, reg_l2=0.02, reg_growth=1.02, rounding=False,
                 beta=0.0, tol=0.0, max_iter=1000, min_diff=1e-8, seed=None):
        self.check_params(k=k, U=U, V=V, W=W, init_method=init_method, reg_l1=reg_l1, reg_l2=reg_l2, reg_growth=reg_growth,
                          rounding=rounding, beta=beta, tol=tol, max_iter=max_iter, min_diff=min_diff, seed=seed)

    def fit(self, X_train, X_val=None, X_test=None, **kwargs):
        super().fit(X_train, X_val, X_test, **kwargs)
        self._fit()
        self.X_pd = None
        self.finish(show_logs=self.show_logs, save_model=self.save_model, show_result=self.show_result)

    def _make_X_pd(self):
        from ..device_ops import boolean_product_csr
        return boolean_product_csr(self.U, self.V, u=0.5, v=0.5, device=self.device)

    def init_model(self):
        if getattr(self, "init_method", "custom") != "custom":
            self.normalize_method = "matrixwise-normalize"
        super().init_model()
        # W = 'mask' / a weight matrix: ContinuousModel.init_W has turned the cells with W != 0 into a device list (self._obs) and the
        # gradient runs over those (multiply(W, U V^T - X) V, ELBMF.py:190); the error and the scores stay whole-matrix (:128, :144)
        self.U[self.U == 0] = EPS
        self.V[self.V == 0] = EPS

    def _fit(self):
        self.iPALM()

    def iPALM(self):
        from ..palm import PalmEngine
        if getattr(self, "task", None) is None:
            raise AttributeError(f"'{type(self).__name__}' object has no attribute 'task'")
        obs = getattr(self, "_obs", None)
        eng = self._eng = PalmEngine(self._bits, self.k, L.PALM_ELBMF, beta=float(self.beta), obs=obs)
        eng.load_factors(self.U, self.V)
        rows, self.counts = [], []
        state = {"gap": np.inf}
        # X_val / X_test (and the training entries under task='prediction') are scored every iteration like the training matrix
        # (ELBMF.py:143 -> BaseModel.evaluate :209-257): Boolean scores only (metrics ERR, Accuracy, Recall, Precision, F1)
        extras = [] if self._scorers else None
        sched = lambda i: (self.reg_l1, self.reg_l2 * (self.reg_growth ** i))   # noqa: E731  (ELBMF.py:122)

        def finish(n_iter, vals):
            """The part of an iteration that looks at its scalars: log row, NaN guard, stopping rule (ELBMF.py:128-160)."""
            reg_l1, reg_l2 = sched(n_iter)
            err, U_gap, V_gap, cnt = vals
            self._check_nan(np.array([[err, U_gap, V_gap]]))
            gap, gap_last = U_gap + V_gap, state["gap"]
            state["gap"] = gap
            rec, prec, acc, f1 = scores_from_counts(*cnt)
            rows.append([n_iter, reg_l1, reg_l2, gap, U_gap, V_gap, err, 1.0 - acc, acc, rec, prec, f1])
            self.counts.append(cnt)
            return self.early_stop(error=gap, diff=abs(gap - gap_last), n_iter=n_iter)

        n_iter = 0
        if obs is None and extras is None and os.environ.get("BMF_PALM_LOOP", "c") != "python":
            # One C call per iteration (bmf_palm_iterate), and iteration t + 1 is enqueued BEFORE the host reads the scalars of t: the
            # device never waits for the stopping rule.  When the rule fires at t, t + 1 has already run -- its `previous iterate`
            # (what ELBMF calls U_last, :124) is the factor pair of t, which is what the loop returns.
            eng.iterate(0, *sched(0), *sched(0))
            while True:
                eng.iterate(n_iter + 1, *sched(n_iter + 1), *sched(n_iter + 1))
                improving = finish(n_iter, eng.row(n_iter))
                n_iter += 1
                if not improving:
                    break
            self.U, self.V = eng.previous_factors()
        else:
            improving = True
            while improving:
                reg_l1, reg_l2 = sched(n_iter)
                if obs is not None:   # (both masked gradients from the state of the previous iteration, before either factor moves)
                    eng.masked_grad("U")
                    eng.masked_grad("V")
                eng.step("U", reg_l1, reg_l2, reg_l1, reg_l2)     # both steps read the state of the previous iteration
                eng.step("V", reg_l1, reg_l2, reg_l1, reg_l2)
                eng.refresh("U")
                eng.refresh("V")
                vals = eng.scalars()
                if extras is not None:   # (the stepwise loop: the engine's bits are those of THIS iteration when its scalars are read)
                    extras.append(self._engine_scores(eng, want_real=False))
                improving = finish(n_iter, vals)
                n_iter += 1
            self.U, self.V = eng.factors()
        if self.rounding:
            self.U, self.V = (self.U > 0.5).astype(np.float64), (self.V > 0.5).astype(np.float64)
        self.n_iter = n_iter
        cols = header(['iter', 'reg_l1', 'reg_l2', 'gap', 'U_gap', 'V_gap', 'error'], levels=3)
        names = ('ERR', 'Accuracy', 'Recall', 'Precision', 'F1')
        cols += [('train', 0, mt) for mt in names]
        if extras is not None:
            for i, ex in enumerate(extras):
                if "train" in ex:   # task='prediction': the training columns are scores over the stored training entries
                    rec, prec, acc, f1 = scores_from_counts(*ex["train"][1])
                    rows[i][7:12] = [1.0 - acc, acc, rec, prec, f1]
                for nm in ("val", "test"):
                    if nm in ex:
                        rec, prec, acc, f1 = scores_from_counts(*ex[nm][1])
                        rows[i] += [1.0 - acc, acc, rec, prec, f1]
            for nm in ("val", "test"):
                if nm in extras[0]:
                    cols += [(nm, 0, mt) for mt in names]
        record_many(self.logs, 'updates', cols, rows)


# ---- module-level step functions, importable like the reference's ---------------------------------------------------------
def get_integrality_gap(U, reg_l1, reg_l2):
    """Elastic-net distance of a factor to {0, 1} (ELBMF.py:166-174); O(rows * k) host arithmetic."""
    U = np.asarray(U, dtype=np.float64)
    dist = np.where(U < 0.5, np.abs(U), np.abs(U - 1))
    return (reg_l1 * dist + reg_l2 * dist ** 2).sum()


def prox(U, kai, lamda):
    """Proximal operator of the elastic-net penalty (ELBMF.py:199-210); O(rows * k) host arithmetic."""
    U = np.asarray(U, dtype=np.float64)
    P = np.where(U <= 0.5, U - kai * np.sign(U), U - kai * np.sign(U - 1) + lamda) / (1 + lamda)
    P[P < 0] = 0
    return P


def update_U(X, U, V, W, reg_l1, reg_l2, beta, U_last, device="cuda:0"):
    """One Gauss-Seidel step for U on the GPU (ELBMF.py:177-196); call it with X.T, V, U for V.  Returns (U_new, U)."""
    from ..engine import BitMatrix
    from ..palm import PalmEngine
    from .BinaryMFPenalty import _is_full
    from .ContinuousModel import ContinuousModel
    ContinuousModel._check_boolean(X)   # anything but 0 / 1 is refused, never silently binarised
    U, V = np.asarray(U, dtype=np.float64), np.asarray(V, dtype=np.float64)
    obs = None
    if not _is_full(W):   # a mask / weight matrix: the gradient runs over the cells with W != 0 (ELBMF.py:190)
        from scipy.sparse import coo_matrix
        from ..engine import SparseObs
        Wc = coo_matrix(W)
        Wc.eliminate_zeros()
        if Wc.shape != tuple(X.shape):
            raise ValueError("W must have the shape of X")
        Xd = np.asarray(X.todense()) if hasattr(X, "todense") else np.asarray(X)
        obs = SparseObs(Wc.row, Wc.col, np.asarray(Xd[Wc.row, Wc.col], dtype=np.float64).ravel(), Wc.data, X.shape, device)
    eng = PalmEngine(BitMatrix(X, device), U.shape[1], L.PALM_ELBMF, beta=float(beta), obs=obs)
    eng.load_factors(U, V, U_prev=np.asarray(U_last, dtype=np.float64))
    eng.step("U", reg_l1, reg_l2)
    return eng.factors()[0], U
