"""PRIMP -- proximal alternating linearised minimisation with a [0, 1] box (Hess et al.), on the GPU.

Counterpart of ``PyBMF/models/PRIMP.py``: same constructor; ``_fit`` runs ``primp`` (:134-160): factors drawn with
``torch.rand`` after ``torch.manual_seed(seed)``, ``elbmf_ipalm`` (:91-131) with ``l2reg = 0``, then rounding.  The reference
class does not run as shipped (``_fit`` calls ``.toarray()`` on the already densified ``X_train``, :29; the module is commented
out of ``PyBMF/models/__init__.py``); its module-level functions do and are what is matched (``tests/golden/g14_palm.*``).

Two behaviours of the reference loop are kept because parity demands them: the inertial term extrapolates from the INITIAL
factors for the whole run (``Uold = U`` inside ``elbmf_step_ipalm`` rebinds a local, :76), and each step applies the prox
twice (``proxelbmfnn`` then ``_proxelbmfnn``, :84-87).  The reference runs in fp32 torch tensors; here the factors are fp64
masters with fp32-accurate contractions (within 1e-4 of either).
"""
from __future__ import annotations

import numpy as np

from .. import _lib as L
from ..utils import header, record_many
from .ContinuousModel import ContinuousModel


class PRIMP(ContinuousModel):
    def __init__(self, k, reg=0.01, reg_growth=1.02, max_iter=1000, min_diff=1e-8, beta=1e-4, seed=None):
        self.check_params(k=k, reg=reg, reg_growth=reg_growth, max_iter=max_iter, min_diff=min_diff, beta=beta, seed=seed)

    def fit(self, X_train, X_val=None, X_test=None, **kwargs):
        super().fit(X_train, X_val, X_test, **kwargs)
        self._fit()
        self.X_pd = None
        self.evaluate(df_name='boolean')
        self.finish(show_logs=self.show_logs, save_model=self.save_model, show_result=self.show_result)

    def _make_X_pd(self):
        from ..device_ops import boolean_product_csr
        return boolean_product_csr(self.U, self.V, u=0.5, v=0.5, device=self.device)

    def init_model(self):
        import torch
        self.init_method = "custom"
        if self.seed is not None:
            torch.manual_seed(int(self.seed))         # primp(): PRIMP.py:147-149 (CPU generator, fp32, U then V as k x n)
        m, n = self.m, self.n
        U0, Vt0 = torch.rand(m, self.k, dtype=torch.float32), torch.rand(self.k, n, dtype=torch.float32)
        self.U, self.V = U0.double().numpy(), Vt0.double().numpy().T.copy()
        super().init_model()
        if self.X_val is not None or self.X_test is not None:
            raise NotImplementedError("PRIMP on the GPU scores the training matrix only")

    def _fit(self):
        if getattr(self, "task", None) is None:
            raise AttributeError(f"'{type(self).__name__}' object has no attribute 'task'")
        self.U, self.V, self.fns = elbmf_ipalm(self._bits, self.U, self.V, self.reg, 0.0, lambda t: self.reg_growth ** t,
                                               int(self.max_iter), float(self.min_diff), float(self.beta))
        # with_rounding (PRIMP.py:155-158): proxelbmfnn(F, 0.5, 0).round() == (F > 0.5) for F in [.., 1]
        self.U, self.V = (self.U > 0.5).astype(np.float64), (self.V > 0.5).astype(np.float64)
        rows = [[t, fn] for t, fn in enumerate(self.fns)]
        record_many(self.logs, 'updates', header(['iter', 'error'], levels=3), rows)


def elbmf_ipalm(X, U, V, l1reg, l2reg, regularization_rate, maxiter, tolerance, beta, callback=None, device="cuda:0"):
    """The loop of PRIMP.py:91-131 on the GPU.  ``X``: a Boolean matrix (array / sparse / engine.BitMatrix); ``V`` is n x k here
    (the reference carries k x n).  Returns (U, V, [||X - U V^T||_F^2 per iteration])."""
    from ..engine import BitMatrix
    from ..palm import PalmEngine
    bits = X if isinstance(X, BitMatrix) else BitMatrix(X, device)
    U, V = np.asarray(U, dtype=np.float64), np.asarray(V, dtype=np.float64)
    eng = PalmEngine(bits, U.shape[1], L.PALM_PRIMP, beta=float(beta))
    eng.load_factors(U, V)            # the anchors of the inertial term stay the initial factors (advance_prev=False below)
    fn, fns = np.inf, []
    for t in range(int(maxiter)):
        tau = regularization_rate(t)
        eng.step("U", l1reg, l2reg * tau, advance_prev=False)
        eng.refresh("U")              # Gauss-Seidel: the V step sees the new U (PRIMP.py:114-115)
        eng.step("V", l1reg, l2reg * tau, advance_prev=False)
        eng.refresh("V")
        fn0, (fn, _, _, _) = fn, eng.scalars(with_counts=False)
        fns.append(fn)
        if callback is not None:
            callback(t, *eng.factors(), fn)
        if abs(fn - fn0) < tolerance:
            break
    U, V = eng.factors()
    return U, V, fns
