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
        super().init_model()   # (X_val / X_test get their scorers there: the final evaluate() scores them like the training matrix)

    def _fit(self):
        if getattr(self, "task", None) is None:
            raise AttributeError(f"'{type(self).__name__}' object has no attribute 'task'")
        self.U, self.V, self.fns = _ipalm_run(self._bits, self.U, self.V, self.reg, 0.0, lambda t: self.reg_growth ** t,
                                              int(self.max_iter), float(self.min_diff), float(self.beta))
        # with_rounding (PRIMP.py:155-158): proxelbmfnn(F, 0.5, 0).round() == (F > 0.5) for F in [.., 1]
        self.U, self.V = (self.U > 0.5).astype(np.float64), (self.V > 0.5).astype(np.float64)
        rows = [[t, fn] for t, fn in enumerate(self.fns)]
        record_many(self.logs, 'updates', header(['iter', 'error'], levels=3), rows)


# ---- module-level functions under the reference's names and signatures (PyBMF/models/PRIMP.py:51-160) ------------------------
# The reference works on torch tensors and carries V as k x n.  Each function takes torch tensors or NumPy arrays and returns the
# kind it was given (tensors: dtype and device of the input); the contractions run in libbmf_hip.so on `device`.

def _is_tensor(x):
    import torch
    return isinstance(x, torch.Tensor)


def _np64(x):
    return x.detach().cpu().double().numpy() if _is_tensor(x) else np.asarray(x, dtype=np.float64)


def _like(arr, ref):
    """`arr` (ndarray) as the kind of `ref`."""
    if _is_tensor(ref):
        import torch
        return torch.from_numpy(np.ascontiguousarray(arr)).to(dtype=ref.dtype if ref.dtype.is_floating_point else torch.float64, device=ref.device)
    return arr


def proxelbmf(x, k, l):  # noqa: E741  (the reference's argument names)
    """Proximal operator of the elastic-net "binary" penalty (PRIMP.py:55-56); element-wise, tensor in -> tensor out."""
    if _is_tensor(x):
        import torch
        return torch.where(x <= 0.5, x - k * torch.sign(x), x - k * torch.sign(x - 1) + l) / (1 + l)
    x = np.asarray(x, dtype=np.float64)
    return np.where(x <= 0.5, x - k * np.sign(x), x - k * np.sign(x - 1) + l) / (1 + l)


def proxelbmfbox(x, k, l):  # noqa: E741
    """PRIMP.py:59-60"""
    p = proxelbmf(x, k, l)
    return p.clamp(0, 1) if _is_tensor(p) else np.clip(p, 0, 1)


def proxelbmfnn(x, k, l):  # noqa: E741
    """PRIMP.py:63-64"""
    p = proxelbmf(x, k, l)
    return p.clamp_min(0) if _is_tensor(p) else np.maximum(p, 0)


def _proxelbmfnn(x, k, l):  # noqa: E741
    """PRIMP.py:51-52"""
    p = proxelbmf(x, k, l)
    return p.clamp_max(1) if _is_tensor(p) else np.minimum(p, 1)


def integrality_gap_elastic(e, l1reg, l2reg):
    """PRIMP.py:67-68"""
    if _is_tensor(e):
        import torch
        return torch.min(l1reg * e.abs() + l2reg * e ** 2, l1reg * (e - 1).abs() + l2reg * (e - 1) ** 2).sum()
    e = np.asarray(e, dtype=np.float64)
    return np.minimum(l1reg * np.abs(e) + l2reg * e ** 2, l1reg * np.abs(e - 1) + l2reg * (e - 1) ** 2).sum()


def _bits_of(X, device):
    from ..engine import BitMatrix
    if isinstance(X, BitMatrix):
        return X
    if _is_tensor(X):
        ContinuousModel._check_boolean(X.detach().cpu().numpy())
    else:
        ContinuousModel._check_boolean(X)
    return BitMatrix(X, device)


def elbmf_step_ipalm(X, U, V, Uold, l1reg, l2reg, tau, beta, device="cuda:0"):
    """One inertial proximal step of U on the GPU (PRIMP.py:71-88): ``V`` is k x n, ``Uold`` the anchor of the inertial term
    (ignored when beta == 0); call it with ``X.T, V.T, U.T, Vold`` for the other factor, as the reference's loop does.  Returns the
    new U; unlike the reference it does not also modify the ``U`` it was given in place."""
    from ..palm import PalmEngine
    Un, Vn = _np64(U), np.ascontiguousarray(_np64(V).T)
    eng = PalmEngine(_bits_of(X, device), Un.shape[1], L.PALM_PRIMP, beta=float(beta))
    eng.load_factors(Un, Vn, U_prev=None if (Uold is None or beta == 0) else _np64(Uold))
    eng.step("U", float(l1reg), float(l2reg) * float(tau), advance_prev=False)
    return _like(eng.factors()[0], U)


def elbmf_ipalm(X, U, V, l1reg, l2reg, regularization_rate, maxiter, tolerance, beta, callback=None, device="cuda:0"):
    """The loop of PRIMP.py:91-131 on the GPU with the reference's signature: ``V`` is k x n, ``callback(t, U, V, fn)`` gets V as
    k x n, the result is ``(U, V)``."""
    cb = None
    if callback is not None:
        cb = lambda t, Uc, Vc, fn: callback(t, _like(Uc, U), _like(np.ascontiguousarray(Vc.T), V), fn)  # noqa: E731
    Un, Vn, _ = _ipalm_run(_bits_of(X, device), _np64(U), np.ascontiguousarray(_np64(V).T), l1reg, l2reg, regularization_rate, maxiter,
                           tolerance, beta, cb)
    return _like(Un, U), _like(np.ascontiguousarray(Vn.T), V)


def primp(X, n_components, l1reg=0.01, l2reg=0, regularization_rate=lambda t: 1.02 ** t, maxiter=3000, tolerance=1e-8, beta=0.0001,
          callback=None, with_rounding=True, seed=None, device="cuda:0"):
    """PRIMP.py:133-160: factors drawn with torch.rand (after torch.manual_seed(seed)) in X's dtype, elbmf_ipalm, rounding.
    Returns (U, V) with V as k x n, tensors when X is a tensor."""
    import torch
    if seed is not None:
        torch.manual_seed(seed)
    dt = X.dtype if (_is_tensor(X) and X.dtype.is_floating_point) else torch.float32
    m, n = X.shape
    U, V = torch.rand(m, n_components, dtype=dt), torch.rand(n_components, n, dtype=dt)
    U, V = elbmf_ipalm(X, U, V, l1reg, l2reg, regularization_rate, maxiter, tolerance, beta, callback, device=device)
    if with_rounding:
        U, V = proxelbmfnn(U, 0.5, l2reg * 1e12).round(), proxelbmfnn(V, 0.5, l2reg * 1e12).round()
    if not _is_tensor(X):
        U, V = U.double().numpy(), V.double().numpy()
    return U, V


def _ipalm_run(bits, U, V, l1reg, l2reg, regularization_rate, maxiter, tolerance, beta, callback=None):
    """The loop itself.  ``bits``: engine.BitMatrix; ``V`` is n x k here; ``callback(t, U, V[n x k], fn)``.  Returns (U, V, [||X - U
    V^T||_F^2 per iteration])."""
    from ..palm import PalmEngine
    U, V = np.asarray(U, dtype=np.float64), np.asarray(V, dtype=np.float64)
    eng = PalmEngine(bits, U.shape[1], L.PALM_PRIMP, beta=float(beta))
    eng.load_factors(U, V)            # the anchors of the inertial term stay the initial factors (advance_prev=False below)
    fn, fns = np.inf, []
    maxiter = int(maxiter)
    if callback is None and maxiter > 0 and eng.can_pipeline():
        # One C call per iteration (bmf_primp_iterate), the objective read one iteration late: iteration t + 1 is enqueued BEFORE the
        # host waits for row t, so the device never waits for the stopping rule (PRIMP.py:128-129).  When the rule fires at t, t + 1
        # has run; the pair of t was snapshotted on the device before it did (same iterations, same factors as the stepwise loop:
        # tests/test_palm_gpu.py).
        eng.primp_iterate(0, l1reg, l2reg * regularization_rate(0))
        for t in range(maxiter):
            more = t + 1 < maxiter
            if more:
                eng.keep()
                eng.primp_iterate(t + 1, l1reg, l2reg * regularization_rate(t + 1))
            fn0, fn = fn, eng.primp_row(t)
            fns.append(fn)
            if abs(fn - fn0) < tolerance:
                U, V = eng.kept_factors() if more else eng.factors()
                return U, V, fns
        U, V = eng.factors()
        return U, V, fns
    for t in range(maxiter):
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
