"""NumPy fp64 restatement of PyBMF's continuous-relaxation hot path (CPU oracle).

TEST INFRASTRUCTURE ONLY -- see ``oracle/__init__.py``.  Parity: PINNED by
``tests/golden/*.npz|json`` (generated from the imported reference by
``tests/golden/make_golden.py``) and checked in ``tests/test_oracle_golden.py``.

All ``file:line`` citations are relative to ``/root/reference`` (PyBMF tag
2024_10_08).  Nothing here is copied from the reference: each routine restates
the arithmetic (operation order included, because fp64 round-off is part of the
pinned trajectory) in plain dense NumPy.

Naming follows the reference: X (m x n data), U (m x k), V (n x k), W (m x n
mask; ``None`` means the all-ones mask of ``W='full'``).
"""
from __future__ import annotations

import numpy as np

EPS = float(np.finfo(np.float64).eps)  # 2.220446049250313e-16

__all__ = [
    "EPS",
    "planted_factor", "synthetic_boolean", "flip_noise",
    "init_factors", "balance_factors", "zeros_to_eps", "unique_values_mapping", "normalize_factors",
    "penalty_update_V", "penalty_update_U", "penalty_update_V_reassoc", "penalty_update_U_reassoc",
    "penalty_errors", "reg_term", "rec_term",
    "real_product", "boolean_product", "boolean_product_blas", "confusion_counts", "boolean_scores", "rmse_mae", "is_boolean_valued", "real_confusion", "real_scores", "matrix_scores",
    "penalty_fit", "wnmf_update", "wnmf_error", "wnmf_fit",
    "stable_sigmoid", "thresh_F", "thresh_dF", "thresh_dXdx", "wolfe_search", "clip_step", "threshold_fit",
    "should_continue", "entry_scores", "confusion_counts_axis", "weighted_error", "coverage_score", "description_length",
    "wnmf_kl_update", "wnmf_kl_error", "wnmf_kl_fit", "pnlpf_prediction", "pnlpf_update_U", "pnlpf_update_V", "pnlpf_fit",
    "elbmf_integrality_gap", "elbmf_prox", "elbmf_step_size", "elbmf_update", "elbmf_fit",
    "primp_prox", "primp_step", "primp_ipalm", "primp_round",
]


# --------------------------------------------------------------------------------------
# synthetic input (generators/SyntheticMatrixGenerator.py:28-70, generators/BaseGenerator.py:158-215,
# utils/generator_utils.py:5-49)
# --------------------------------------------------------------------------------------
def planted_factor(rows: int, k: int, density: float, rng: np.random.RandomState) -> np.ndarray:
    """One planted Boolean factor (SyntheticMatrixGenerator.py:48-70).

    Column c gets a guaranteed block of ones in rows [c*l, (c+1)*l) with l = ceil(rows/100) and
    Bernoulli(density) entries in the tail rows [k*l, rows).  One ``rng.binomial`` call per column,
    in column order -- the draw order is part of the contract.
    """
    F = np.zeros((rows, k), dtype=np.int64)
    l = int(np.ceil(rows / 100))
    for c in range(k):
        F[c * l:(c + 1) * l, c] = 1
        F[k * l:rows, c] = rng.binomial(size=rows - k * l, n=1, p=density)
    return F


def synthetic_boolean(m: int, n: int, k: int, density=(0.2, 0.2), seed: int = 0):
    """X = min(U V^T, 1) with shuffled planted factors; returns (X, U, V, rng).

    Draw order (SyntheticMatrixGenerator.py:32-46, BaseGenerator.py:158-169, generator_utils.py:5-18):
    all of U's columns, all of V's columns, ``rng.rand(m)`` for the row order of U, ``rng.rand(n)`` for V.
    """
    rng = np.random.RandomState(seed)
    U = planted_factor(m, k, density[0], rng)
    V = planted_factor(n, k, density[1], rng)
    U = U[rng.rand(m).argsort(), :]
    V = V[rng.rand(n).argsort(), :]
    X = np.minimum(U @ V.T, 1).astype(np.int64)
    return X, U, V, rng


def flip_noise(X: np.ndarray, noise=(0.0, 0.0), seed: int | None = None, rng=None) -> np.ndarray:
    """Flip 1->0 with p_pos then 0->1 with p_neg (generator_utils.py:30-49, BaseGenerator.py:202-215).

    Two full-shape ``rng.binomial`` draws, in this order; re-seeded when ``seed`` is given.
    """
    if seed is not None:
        rng = np.random.RandomState(seed)
    p_pos, p_neg = noise
    X = np.asarray(X)
    d = rng.binomial(size=X.shape, n=1, p=p_pos)
    X = np.maximum(X - d, 0)
    d = rng.binomial(size=X.shape, n=1, p=p_neg)
    X = np.minimum(X + d, 1)
    return X.astype(np.int64)


# --------------------------------------------------------------------------------------
# model initialisation (models/ContinuousModel.py:14-148,188-203)
# --------------------------------------------------------------------------------------
def init_factors(X: np.ndarray, k: int, method: str, rng: np.random.RandomState):
    """Random initial factors; **V is drawn before U** (ContinuousModel.py:66-80)."""
    m, n = X.shape
    avg = np.sqrt(np.asarray(X, dtype=np.float64).mean() / k)
    if method == "normal":
        V = avg * rng.standard_normal(size=(n, k))
        U = avg * rng.standard_normal(size=(m, k))
        return np.abs(U), np.abs(V)
    if method == "uniform":
        V = rng.uniform(low=0, high=avg * 2, size=(n, k))
        U = rng.uniform(low=0, high=avg * 2, size=(m, k))
        return U, V
    raise ValueError(method)


def balance_factors(U: np.ndarray, V: np.ndarray):
    """'balance' normalisation (ContinuousModel.py:117-123): per column i,
    U[:,i] *= dV[i]/dU[i]; V[:,i] *= dU[i]/dV[i] with d = sqrt(column max); multiply first, then divide."""
    U = np.array(U, dtype=np.float64)
    V = np.array(V, dtype=np.float64)
    dU = np.sqrt(U.max(axis=0))
    dV = np.sqrt(V.max(axis=0))
    for i in range(U.shape[1]):
        U[:, i] = U[:, i] * dV[i] / dU[i]
        V[:, i] = V[:, i] * dU[i] / dV[i]
    return U, V


def unique_values_mapping(arr: np.ndarray) -> np.ndarray:
    """Each value -> (its rank among the distinct values) / (number of distinct values), in [0, 1)
    (ContinuousModel.py:225-231: dict from np.unique, applied element by element)."""
    arr = np.asarray(arr, dtype=np.float64)
    uniq = np.unique(arr)
    lookup = {float(val): idx / len(uniq) for idx, val in enumerate(uniq)}
    out = np.empty(arr.shape, dtype=np.float64)
    for pos, val in np.ndenumerate(arr):
        out[pos] = lookup[float(val)]
    return out


def normalize_factors(U: np.ndarray, V: np.ndarray, method):
    """normalize_UV (ContinuousModel.py:87-148) for every normalize_method."""
    U = np.array(U, dtype=np.float64)
    V = np.array(V, dtype=np.float64)
    if method is None:
        return U, V
    if method == "balance":
        return balance_factors(U, V)
    if method == "matrixwise-normalize":
        return U / U.max(), V / V.max()
    if method == "columnwise-normalize":
        for i in range(U.shape[1]):
            U[:, i] = U[:, i] / U[:, i].max()
            V[:, i] = V[:, i] / V[:, i].max()
        return U, V
    if method == "matrixwise-mapping":
        return unique_values_mapping(U), unique_values_mapping(V)
    if method == "columnwise-mapping":
        for i in range(U.shape[1]):
            U[:, i] = unique_values_mapping(U[:, i])
            V[:, i] = unique_values_mapping(V[:, i])
        return U, V
    raise ValueError(method)


def zeros_to_eps(F: np.ndarray) -> np.ndarray:
    """solver=='mu': exact zeros in the initial factors become eps (ContinuousModel.py:33-36)."""
    F = np.array(F, dtype=np.float64)
    F[F == 0] = EPS
    return F


# --------------------------------------------------------------------------------------
# BinaryMFPenalty arithmetic (models/BinaryMFPenalty.py:136-186)
# --------------------------------------------------------------------------------------
def _masked(W, A):
    return A if W is None else W * A


def penalty_update_V(X, W, U, V, reg):
    """Literal association of BinaryMFPenalty.py:151-163 (materialises U V^T)."""
    num = _masked(W, X).T @ U
    num = num + 3 * reg * np.power(V, 2)
    den = _masked(W, U @ V.T).T @ U
    den = den + (2 * reg * np.power(V, 3) + reg * V)
    den[den == 0] = EPS
    Vn = V * (num / den)
    Vn[Vn == 0] = EPS
    return Vn


def penalty_update_U(X, W, U, V, reg):
    """Literal association of BinaryMFPenalty.py:136-148."""
    num = _masked(W, X) @ V
    num = num + 3 * reg * np.power(U, 2)
    den = _masked(W, U @ V.T) @ V
    den = den + (2 * reg * np.power(U, 3) + reg * U)
    den[den == 0] = EPS
    Un = U * (num / den)
    Un[Un == 0] = EPS
    return Un


def penalty_update_V_reassoc(X, U, V, reg):
    """Same update with W='full' re-associated: (U V^T)^T U == V (U^T U).  Agrees with the literal form
    to ~1e-15; this is the association the HIP path uses."""
    num = X.T @ U + 3 * reg * np.power(V, 2)
    den = V @ (U.T @ U) + (2 * reg * np.power(V, 3) + reg * V)
    den[den == 0] = EPS
    Vn = V * (num / den)
    Vn[Vn == 0] = EPS
    return Vn


def penalty_update_U_reassoc(X, U, V, reg):
    num = X @ V + 3 * reg * np.power(U, 2)
    den = U @ (V.T @ V) + (2 * reg * np.power(U, 3) + reg * U)
    den[den == 0] = EPS
    Un = U * (num / den)
    Un[Un == 0] = EPS
    return Un


def reg_term(F) -> float:
    """0.5 * sum((F^2 - F)^2)  (BinaryMFPenalty.py:182-186)."""
    return float(0.5 * np.sum(np.power(np.power(F, 2) - F, 2)))


def rec_term(X, X_pd, W=None) -> float:
    """0.5 * sum(W o (X - X_pd)^2)  (BinaryMFPenalty.py:175-179)."""
    return float(0.5 * np.sum(_masked(W, np.power(X - X_pd, 2))))


def penalty_errors(X, W, U, V, reg):
    """(error, rec_error, reg_error) as BinaryMFPenalty.py:166-172."""
    rec = rec_term(X, U @ V.T, W)
    rg = float(reg * (reg_term(U) + reg_term(V)))
    return rec + rg, rec, rg


# --------------------------------------------------------------------------------------
# predictions and metrics (utils/common.py:64-151, utils/metrics.py:8-170, utils/evaluate_utils.py:12-54)
# --------------------------------------------------------------------------------------
def real_product(U, V):
    """get_prediction(U, V, boolean=False) (common.py:98-107) as a dense array."""
    return np.asarray(U) @ np.asarray(V).T


def boolean_product(U, V, u=None, v=None, us=None, vs=None):
    """get_prediction_with_threshold (common.py:110-151): min(1, (U>u) @ (V>v)^T), strict '>'.
    ``us``/``vs`` give one threshold per factor column and take precedence over ``u``/``v``."""
    U = np.asarray(U)
    V = np.asarray(V)
    if us is not None:
        assert len(us) == U.shape[1]
        Ub = (U > np.asarray(us)[None, :]).astype(np.int64)
    elif u is not None:
        Ub = (U > u).astype(np.int64)
    else:
        Ub = U
    if vs is not None:
        assert len(vs) == V.shape[1]
        Vb = (V > np.asarray(vs)[None, :]).astype(np.int64)
    elif v is not None:
        Vb = (V > v).astype(np.int64)
    else:
        Vb = V
    return np.minimum(Ub @ Vb.T, 1).astype(np.int64)


def boolean_product_blas(U, V, u=0.5, v=0.5):
    """The same Boolean product as ``boolean_product(U, V, u, v)`` (scalar thresholds) as a bool array, through a float32 BLAS GEMM
    -- counts up to k <= 2^24 are exact in fp32 -- instead of the int64 matmul NumPy runs without BLAS.  Used by the timed
    "best CPU formulation" of bench.py; checked against ``boolean_product`` in tests/test_oracle_golden.py."""
    return ((U > u).astype(np.float32) @ (V > v).astype(np.float32).T) > 0


def confusion_counts(gt, pd):
    """(TP, FP, FN, TN) over the whole matrix (task='reconstruction'; metrics.py:56-77)."""
    gt = np.asarray(gt)
    pd = np.asarray(pd)
    tp = int(np.sum(gt * pd))
    fp = int(np.sum(np.maximum(pd - gt, 0)))
    fn = int(np.sum(np.maximum(gt - pd, 0)))
    tn = int(np.sum((1 - gt) * (1 - pd)))
    return tp, fp, fn, tn


def is_boolean_valued(X) -> bool:
    X = np.asarray(X)
    return bool(((X == 0) | (X == 1)).all())


def real_confusion(gt, pd):
    """The same four sums for a REAL-valued ground truth, unrounded, plus sum gt and sum pd: what utils/metrics.py:56-77 computes on two
    csr matrices under task='reconstruction' (TP = sum gt pd, FP = sum max(pd - gt, 0), FN = sum max(gt - pd, 0), TN = TP of the
    inverted pair = sum (1 - gt)(1 - pd))."""
    gt = np.asarray(gt, dtype=np.float64)
    pd = np.asarray(pd, dtype=np.float64)
    return (float(np.sum(gt * pd)), float(np.sum(np.maximum(pd - gt, 0))), float(np.sum(np.maximum(gt - pd, 0))),
            float(np.sum((1 - gt) * (1 - pd))), float(gt.sum()), float(pd.sum()))


def real_scores(tp, fp, fn, tn, sum_gt, sum_pd, cells):
    """(Recall, Precision, Accuracy, F1) as metrics.py:79-135 forms them from those sums: TP / sum gt, TP / sum pd, (TP + TN) / cells."""
    recall = np.float64(tp) / sum_gt if sum_gt > 0 else 0
    precision = np.float64(tp) / sum_pd if sum_pd > 0 else 0
    accuracy = (np.float64(tp) + np.float64(tn)) / cells
    s = precision + recall
    f1 = 2 * precision * recall / s if s > 0 else 0
    return float(recall), float(precision), float(accuracy), float(f1)


def matrix_scores(X, pd):
    """(counts, (Recall, Precision, Accuracy, F1)) of a whole-matrix comparison, Boolean or real-valued ground truth."""
    if is_boolean_valued(X):
        c = confusion_counts(np.asarray(X).astype(np.int64), pd)
        return c, boolean_scores(*c)
    c = real_confusion(X, pd)
    return c, real_scores(*c, cells=float(np.asarray(X).size))


def confusion_counts_axis(gt, pd, axis=None):
    """(TP, FP, FN, TN) with the reference's ``axis`` (metrics.py:56-77: ``.sum(axis=axis)``; 0 = per column, 1 = per row)."""
    gt, pd = np.asarray(gt, dtype=np.int64), np.asarray(pd, dtype=np.int64)
    tp = (gt * pd).sum(axis=axis)
    fp = np.maximum(pd - gt, 0).sum(axis=axis)
    fn = np.maximum(gt - pd, 0).sum(axis=axis)
    tn = ((1 - gt) * (1 - pd)).sum(axis=axis)
    return tp, fp, fn, tn


def weighted_error(gt, pd, w_fp=0.5, w_fn=None, axis=None):
    """w_fp FP + w_fn FN (metrics.py:182-186)."""
    w_fn = 1 - w_fp if w_fn is None else w_fn
    tp, fp, fn, tn = confusion_counts_axis(gt, pd, axis)
    return w_fp * fp + w_fn * fn


def coverage_score(gt, pd, w_fp=0.5, w_fn=None, axis=None):
    """-w_fp FP + w_fn TP (metrics.py:189-201)."""
    w_fn = 1 - w_fp if w_fn is None else w_fn
    tp, fp, fn, tn = confusion_counts_axis(gt, pd, axis)
    return -w_fp * fp + w_fn * tp


def description_length(gt, U, V, pd=None, w_model=1.0, w_fp=1.0, w_fn=1.0):
    """w_model (|U| + |V|) + w_fp FP + w_fn FN, pd = the Boolean product of U, V unless given (metrics.py:173-179)."""
    U, V = np.asarray(U), np.asarray(V)
    pd = np.minimum(U.astype(np.int64) @ V.astype(np.int64).T, 1) if pd is None else pd
    tp, fp, fn, tn = confusion_counts_axis(gt, pd, None)
    return w_model * (U.sum() + V.sum()) + w_fp * fp + w_fn * fn


def boolean_scores(tp: int, fp: int, fn: int, tn: int):
    """(Recall, Precision, Accuracy, F1) from the four counts with the reference's edge rules
    (metrics.py:79-135): Recall=0 if sum(gt)==0, Precision=0 if sum(pd)==0, F1=0 if P+R==0."""
    n_gt = tp + fn
    n_pd = tp + fp
    total = tp + fp + fn + tn
    recall = np.float64(tp) / n_gt if n_gt > 0 else 0
    precision = np.float64(tp) / n_pd if n_pd > 0 else 0
    accuracy = np.float64(tp + tn) / total
    s = precision + recall
    f1 = 2 * precision * recall / s if s > 0 else 0
    return float(recall), float(precision), float(accuracy), float(f1)


def rmse_mae(gt, pd):
    """(RMSE, MAE) over the whole matrix (metrics.py:149-160)."""
    d = np.asarray(gt, dtype=np.float64) - np.asarray(pd, dtype=np.float64)
    N = d.shape[0] * d.shape[1]
    return float(np.sqrt(np.power(d, 2).sum() / N)), float(np.abs(d).sum() / N)


def entry_scores(rows, cols, vals, U, V, u=None, v=None):
    """task='prediction' (utils/evaluate_utils.py:32-44): the prediction is gathered at the triplets of the ground truth and
    the metrics (utils/metrics.py) run on the two 1-D vectors.  With thresholds -> (TP, FP, FN, TN) of the Boolean product;
    without -> (RMSE, MAE) of U V^T.  The continuous models densify their data sets first (ContinuousModel.py:167-182), so
    the triplets their evaluate() sees are the NON-ZERO cells: the caller passes exactly those."""
    rows, cols = np.asarray(rows), np.asarray(cols)
    gt = np.asarray(vals, dtype=np.float64)
    if u is None:
        pd = np.einsum("ij,ij->i", np.asarray(U, dtype=np.float64)[rows], np.asarray(V, dtype=np.float64)[cols])
        d = gt - pd
        return float(np.sqrt(np.power(d, 2).sum() / len(d))), float(np.abs(d).sum() / len(d))
    pd = ((np.asarray(U)[rows] > u) & (np.asarray(V)[cols] > v)).any(axis=1).astype(np.float64)
    tp = int(np.logical_and(gt, pd).sum())
    fp = int(np.maximum(pd - gt, 0).sum())
    fn = int(np.maximum(gt - pd, 0).sum())
    tn = int(np.logical_and(1 - gt, 1 - pd).sum())
    return tp, fp, fn, tn


def should_continue(model: dict, error=None, diff=None, n_iter=None) -> bool:
    """early_stop() (BaseModelTools.py:299-343) for the three criteria the hot path uses.
    ``model`` carries the optional keys tol / max_iter / min_diff."""
    go = True
    if error is not None and "tol" in model and error <= model["tol"]:
        go = False
    if n_iter is not None and "max_iter" in model and n_iter > model["max_iter"]:
        go = False
    if diff is not None and "min_diff" in model and diff < model["min_diff"]:
        go = False
    return go


# --------------------------------------------------------------------------------------
# BinaryMFPenalty.fit (models/BinaryMFPenalty.py:49-115)
# --------------------------------------------------------------------------------------
def penalty_fit(X, k, U=None, V=None, reg=2.0, reg_growth=3.0, max_reg=1e10, tol=0.01, min_diff=0.0,
                max_iter=100, init_method="custom", normalize_method="balance", seed=None,
                literal=True, use_mask=False, W=None):
    """Whole ``BinaryMFPenalty(...).fit(X, task='reconstruction')`` trajectory with W='full'.

    Returns a dict with final U, V, final ``reg`` and the two log tables as lists of rows:
    ``updates`` rows = (iter, error, rec_error, reg, reg_error, RMSE, MAE),
    ``boolean`` rows = (Recall, Precision, Accuracy, F1), ``counts`` rows = (TP, FP, FN, TN).
    ``literal`` selects the reference's association; ``use_mask`` also multiplies by an explicit
    all-ones mask like the reference does (same numbers, more memory traffic; used for CPU timing).
    ``W``: a dense 0/1 (or weight) mask for W='mask' / an explicit W (ContinuousModel.py:39-63); it enters the two
    updates and rec_error, while RMSE / MAE / Boolean scores stay whole-matrix (task='reconstruction').
    """
    X = np.asarray(X, dtype=np.float64)
    m, n = X.shape
    rng = np.random.RandomState(seed)
    if init_method == "custom":
        U0, V0 = np.array(U, dtype=np.float64), np.array(V, dtype=np.float64)
    else:
        U0, V0 = init_factors(X, k, init_method, rng)
    if normalize_method == "balance":
        U0, V0 = balance_factors(U0, V0)
    U, V = zeros_to_eps(U0), zeros_to_eps(V0)
    if W is not None:
        W = np.asarray(W, dtype=np.float64)
        assert literal, "a general mask has no re-associated form"
    elif use_mask:
        W = np.ones((m, n))
    reg, reg_growth, max_reg = np.float64(reg), np.float64(reg_growth), np.float64(max_reg)
    ctl = {"tol": tol, "max_iter": max_iter, "min_diff": min_diff}

    updates, boolean, counts = [], [], []

    def log_rows(it, err, rec, rg):
        rmse, mae = rmse_mae(X, real_product(U, V))
        updates.append((it, err, rec, float(reg), rg, rmse, mae))
        c, sc = matrix_scores(X, boolean_product(U, V, 0.5, 0.5))
        counts.append(c)
        boolean.append(sc)

    n_iter = 0
    err_old, rec_old, rg_old = penalty_errors(X, W, U, V, reg)
    log_rows(n_iter, err_old, rec_old, rg_old)
    improving = True
    while improving:
        n_iter += 1
        if literal:
            V = penalty_update_V(X, W, U, V, reg)
            U = penalty_update_U(X, W, U, V, reg)
        else:
            V = penalty_update_V_reassoc(X, U, V, reg)
            U = penalty_update_U_reassoc(X, U, V, reg)
        err, rec, rg = penalty_errors(X, W, U, V, reg)
        diff = abs(rg_old - rg)
        err_old, rec_old, rg_old = err, rec, rg
        log_rows(n_iter, err, rec, rg)
        improving = should_continue(ctl, error=rg_old, diff=diff, n_iter=n_iter)
        reg = min(reg * reg_growth, max_reg)
    return {"U": U, "V": V, "U0": U0, "V0": V0, "reg": float(reg), "n_iter": n_iter,
            "updates": updates, "boolean": boolean, "counts": counts}


# --------------------------------------------------------------------------------------
# WNMF (models/WNMF.py:51-150), Frobenius loss
# --------------------------------------------------------------------------------------
def wnmf_update(X, W, U, V):
    """One Frobenius MU sweep, V then U (WNMF.py:96-109).  No penalty terms and **no** factor==0 -> eps
    clamp; only ``denom == 0 -> eps``."""
    num = _masked(W, X).T @ U
    den = _masked(W, U @ V.T).T @ U
    den[den == 0] = EPS
    V = V * (num / den)
    num = _masked(W, X) @ V
    den = _masked(W, U @ V.T) @ V
    den[den == 0] = EPS
    U = U * (num / den)
    return U, V


def wnmf_error(X, W, U, V):
    """WNMF.error (WNMF.py:133-144).  Mutates ``X`` **in place**: exact zeros become eps, as the
    reference does through its ``X_gt = self.X_train`` alias."""
    X_pd = U @ V.T
    X[X == 0] = EPS
    X_pd[X_pd == 0] = EPS
    return float(0.5 * np.sum(_masked(W, np.power(X - X_pd, 2))))


def wnmf_fit(X, k, U=None, V=None, W=None, tol=0.0, min_diff=0.0, max_iter=30, init_method="normal", seed=None):
    """``WNMF(...).fit(X, task='reconstruction')`` with a dense mask ``W`` (None = all ones).
    Returns U, V, the mutated X and rows (iter, error, RMSE, MAE)."""
    X = np.array(X, dtype=np.float64)
    rng = np.random.RandomState(seed)
    if init_method == "custom":
        U, V = np.array(U, dtype=np.float64), np.array(V, dtype=np.float64)
    else:
        U, V = init_factors(X, k, init_method, rng)
    U, V = zeros_to_eps(U), zeros_to_eps(V)
    ctl = {"tol": tol, "max_iter": max_iter, "min_diff": min_diff}
    rows = []
    n_iter = 0
    err_old = wnmf_error(X, W, U, V)
    rows.append((n_iter, err_old) + rmse_mae(X, real_product(U, V)))
    improving = True
    while improving:
        n_iter += 1
        U, V = wnmf_update(X, W, U, V)
        err = wnmf_error(X, W, U, V)
        diff = abs(err_old - err)
        err_old = err
        rows.append((n_iter, err) + rmse_mae(X, real_product(U, V)))
        improving = should_continue(ctl, error=err_old, diff=diff, n_iter=n_iter)
    return {"U": U, "V": V, "X": X, "updates": rows, "n_iter": n_iter}


def wnmf_kl_update(X, W, U, V):
    """One Kullback-Leibler MU sweep, V then U (WNMF.py:111-129): num = (WX / UV)^T U, denom = O^T U with O the all-ones
    matrix (NOT the mask), denom == 0 -> eps; no clamp of the factor."""
    WX = _masked(W, X)
    O = np.ones(X.shape)
    num = (WX / (U @ V.T)).T @ U
    den = O.T @ U
    den[den == 0] = EPS
    V = V * (num / den)
    num = (WX / (U @ V.T)) @ V
    den = O @ V
    den[den == 0] = EPS
    U = U * (num / den)
    return U, V


def wnmf_kl_error(X, W, U, V):
    """WNMF.error for beta_loss='kullback-leibler' (WNMF.py:133-147): zeros of X and of U V^T become eps IN PLACE first,
    then sum(W o (X log(X / X_pd) - X + X_pd))."""
    X_pd = U @ V.T
    X[X == 0] = EPS
    X_pd[X_pd == 0] = EPS
    return float(np.sum(_masked(W, X * np.log(X / X_pd) - X + X_pd)))


def wnmf_kl_fit(X, k, U=None, V=None, W=None, tol=0.0, min_diff=0.0, max_iter=30, init_method="normal", seed=None):
    """``WNMF(beta_loss='kullback-leibler').fit(X, task='reconstruction')``; rows (iter, error, RMSE, MAE)."""
    X = np.array(X, dtype=np.float64)
    rng = np.random.RandomState(seed)
    if init_method == "custom":
        U, V = np.array(U, dtype=np.float64), np.array(V, dtype=np.float64)
    else:
        U, V = init_factors(X, k, init_method, rng)
    U, V = zeros_to_eps(U), zeros_to_eps(V)
    ctl = {"tol": tol, "max_iter": max_iter, "min_diff": min_diff}
    rows = []
    n_iter = 0
    err_old = wnmf_kl_error(X, W, U, V)
    rows.append((n_iter, err_old) + rmse_mae(X, real_product(U, V)))
    improving = True
    while improving:
        n_iter += 1
        U, V = wnmf_kl_update(X, W, U, V)
        err = wnmf_kl_error(X, W, U, V)
        diff = abs(err_old - err)
        err_old = err
        rows.append((n_iter, err) + rmse_mae(X, real_product(U, V)))
        improving = should_continue(ctl, error=err_old, diff=diff, n_iter=n_iter)
    return {"U": U, "V": V, "X": X, "updates": rows, "n_iter": n_iter}


# --------------------------------------------------------------------------------------
# PNLPF (models/PNLPF.py): BinaryMFPenalty's loop with a sigmoid link on the product
# --------------------------------------------------------------------------------------
def pnlpf_prediction(U, V, link_lamda):
    """sigmoid(link_lamda (U V^T - 1/2)) (PNLPF.py:54-58)."""
    return stable_sigmoid((U @ V.T - 0.5) * link_lamda)


def _pnlpf_parts(U, V, link_lamda):
    sig = pnlpf_prediction(U, V, link_lamda)
    return sig, sig * (1 - sig)


def pnlpf_update_U(X, W, U, V, reg, link_lamda):
    """PNLPF.py:61-75."""
    sig, d = _pnlpf_parts(U, V, link_lamda)
    num = link_lamda * _masked(W, X * d) @ V + 3 * reg * np.power(U, 2)
    den = link_lamda * _masked(W, sig * d) @ V + 2 * reg * np.power(U, 3) + reg * U
    den[den == 0] = EPS
    Un = U * (num / den)
    Un[Un == 0] = EPS
    return Un


def pnlpf_update_V(X, W, U, V, reg, link_lamda):
    """PNLPF.py:77-91."""
    sig, d = _pnlpf_parts(U, V, link_lamda)
    num = link_lamda * _masked(W, X * d).T @ U + 3 * reg * np.power(V, 2)
    den = link_lamda * _masked(W, sig * d).T @ U + 2 * reg * np.power(V, 3) + reg * V
    den[den == 0] = EPS
    Vn = V * (num / den)
    Vn[Vn == 0] = EPS
    return Vn


def pnlpf_fit(X, k, U=None, V=None, reg=2.0, link_lamda=10, reg_growth=3.0, max_reg=1e10, tol=0.01, min_diff=0.0,
              max_iter=100, init_method="custom", normalize_method="balance", seed=None, W=None, trace=False):
    """``PNLPF(...).fit(X, task='reconstruction')``: the inherited BinaryMFPenalty._fit (BinaryMFPenalty.py:61-115) with
    PNLPF's update_U / update_V / get_prediction.  Same return layout as penalty_fit; RMSE / MAE / rec_error are measured
    against the sigmoid prediction, the Boolean scores against the factors thresholded at 0.5.  trace: also return the
    factor pair of every log row (what the per-iteration scores of extra data sets are computed from)."""
    X = np.asarray(X, dtype=np.float64)
    rng = np.random.RandomState(seed)
    if init_method == "custom":
        U0, V0 = np.array(U, dtype=np.float64), np.array(V, dtype=np.float64)
    else:
        U0, V0 = init_factors(X, k, init_method, rng)
    if normalize_method == "balance":
        U0, V0 = balance_factors(U0, V0)
    U, V = zeros_to_eps(U0), zeros_to_eps(V0)
    reg, reg_growth, max_reg = np.float64(reg), np.float64(reg_growth), np.float64(max_reg)
    ctl = {"tol": tol, "max_iter": max_iter, "min_diff": min_diff}
    updates, boolean, counts, states = [], [], [], []

    def errors():
        rec = rec_term(X, pnlpf_prediction(U, V, link_lamda), W)
        rg = float(reg) * (reg_term(U) + reg_term(V))
        return rec + rg, rec, rg

    def log_rows(it, err, rec, rg):
        rmse, mae = rmse_mae(X, pnlpf_prediction(U, V, link_lamda))
        if trace:
            states.append((U.copy(), V.copy()))
        updates.append((it, err, rec, float(reg), rg, rmse, mae))
        c, sc = matrix_scores(X, boolean_product(U, V, 0.5, 0.5))
        counts.append(c)
        boolean.append(sc)

    n_iter = 0
    err_old, rec_old, rg_old = errors()
    log_rows(n_iter, err_old, rec_old, rg_old)
    improving = True
    while improving:
        n_iter += 1
        V = pnlpf_update_V(X, W, U, V, reg, link_lamda)
        U = pnlpf_update_U(X, W, U, V, reg, link_lamda)
        err, rec, rg = errors()
        diff = abs(rg_old - rg)
        err_old, rec_old, rg_old = err, rec, rg
        log_rows(n_iter, err, rec, rg)
        improving = should_continue(ctl, error=rg_old, diff=diff, n_iter=n_iter)
        reg = min(reg * reg_growth, max_reg)
    return {"U": U, "V": V, "U0": U0, "V0": V0, "reg": float(reg), "n_iter": n_iter,
            "updates": updates, "boolean": boolean, "counts": counts, **({"trace": states} if trace else {})}


# --------------------------------------------------------------------------------------
# BinaryMFThreshold (models/BinaryMFThreshold.py:62-227, solvers/line_search.py:4-104, utils/common.py:81-89)
# --------------------------------------------------------------------------------------
def stable_sigmoid(Z):
    """Piecewise-stable logistic (common.py:81-89)."""
    Z = np.asarray(Z, dtype=np.float64)
    Y = np.zeros(Z.shape)
    pos = Z >= 0
    Y[pos] = 1.0 / (1.0 + np.exp(-Z[pos]))
    ez = np.exp(Z[~pos])
    Y[~pos] = ez / (1 + ez)
    return Y


def thresh_F(X, W, U, V, u, v, lamda):
    """F(u,v) = 0.5 * sum((W o (X - s(lam(U-u)) s(lam(V-v))^T))^2)  (BinaryMFThreshold.py:150-171)."""
    Us = stable_sigmoid((U - u) * lamda)
    Vs = stable_sigmoid((V - v) * lamda)
    R = X - Us @ Vs.T
    return float(0.5 * np.sum(np.power(_masked(W, R), 2)))


def thresh_dXdx(F, x, lamda):
    """lam * exp(-lam (F-x)) * sigmoid(lam (F-x))^2  (BinaryMFThreshold.py:211-227; overflow-prone form kept)."""
    d = F - x
    with np.errstate(over="ignore", invalid="ignore"):
        return np.exp(-lamda * d) * lamda * stable_sigmoid(d * lamda) ** 2


def thresh_dF(X, W, U, V, u, v, lamda):
    """The 2-vector the reference calls dF (BinaryMFThreshold.py:174-207).  Note its sign convention:
    R = W o (X - X_pd) is used as 'dFdX' and dXdx is d(sigmoid)/d(-x), so the two signs cancel."""
    Us = stable_sigmoid((U - u) * lamda)
    Vs = stable_sigmoid((V - v) * lamda)
    R = _masked(W, X - Us @ Vs.T)
    dFdU = R @ Vs
    dFdu = dFdU * thresh_dXdx(U, u, lamda)
    dFdV = Us.T @ R
    dFdv = dFdV * thresh_dXdx(V, v, lamda).T
    return np.array([np.sum(dFdu), np.sum(dFdv)])


def wolfe_search(f, fprime, xk, pk, maxiter=1000, c1=0.1, c2=0.4):
    """Bracketing / bisection Wolfe search (solvers/line_search.py:4-67).
    Returns (alpha, fc, gc, new_fval, old_fval, new_slope)."""
    alpha = 2
    a, b = 0, 10
    it = 0
    fk = f(xk)
    gk = fprime(xk)
    fc, gc = 1, 1
    x = xk
    while it <= maxiter:
        it += 1
        x = xk + alpha * pk
        armijo = f(x) - fk <= alpha * c1 * np.dot(gk, pk)
        fc += 1
        if armijo:
            curvature = np.dot(fprime(x), pk) >= c2 * np.dot(gk, pk)
            gc += 1
            if curvature:
                break
            if b < 10:
                a = alpha
                alpha = (a + b) / 2
            else:
                alpha = alpha * 1.2
        else:
            b = alpha
            alpha = (a + b) / 2
    new_fval, new_slope = f(x), fprime(x)
    return alpha, fc + 1, gc + 1, new_fval, fk, new_slope


def clip_step(x_min, x_max, x_last, xk, alpha, pk):
    """limit_step_size (solvers/line_search.py:70-104)."""
    if (x_last <= x_max).all() and (x_last >= x_min).all():
        return x_last, alpha
    a_new = alpha
    for i in range(len(x_last)):
        a_tmp = a_new
        if x_last[i] > x_max[i]:
            a_tmp = (x_max[i] - xk[i]) / pk[i]
        if x_last[i] < x_min[i]:
            a_tmp = (x_min[i] - xk[i]) / pk[i]
        if a_tmp < a_new:
            a_new = a_tmp
    return xk + a_new * pk, a_new


def threshold_fit(X, U, V, W=None, u=0.5, v=0.5, lamda=100, min_diff=1e-3, max_iter=100):
    """``BinaryMFThreshold(...).fit`` (BinaryMFThreshold.py:82-147).  Rows = (iter, u, v, F, TP, FP, FN, TN);
    also returns the F/dF evaluation counts."""
    X = np.asarray(X, dtype=np.float64)
    U = np.asarray(U, dtype=np.float64)
    V = np.asarray(V, dtype=np.float64)
    calls = {"F": 0, "dF": 0}

    def f(x):
        calls["F"] += 1
        return thresh_F(X, W, U, V, x[0], x[1], lamda)

    def g(x):
        calls["dF"] += 1
        return thresh_dF(X, W, U, V, x[0], x[1], lamda)

    ctl = {"max_iter": max_iter, "min_diff": min_diff}
    Xi = X.astype(np.int64) if is_boolean_valued(X) else np.asarray(X, dtype=np.float64)
    confusion = confusion_counts if is_boolean_valued(X) else (lambda gt, pd: real_confusion(gt, pd))
    rows = []
    n_iter = 0
    x_last = np.array([u, v])
    p_last = -g(x_last)
    new_fval = f(x_last)
    rows.append((n_iter, float(x_last[0]), float(x_last[1]), new_fval) +
                confusion(Xi, boolean_product(U, V, x_last[0], x_last[1])))
    improving = True
    while improving:
        n_iter += 1
        xk, pk = x_last, p_last
        alpha, fc, gc, new_fval, old_fval, new_slope = wolfe_search(f, g, xk, pk, maxiter=50)
        x_last = xk + alpha * pk
        eps = 1e-6
        x_min = np.array([U.min() + eps, V.min() + eps])
        x_max = np.array([U.max() - eps, V.max() - eps])
        x_last, alpha = clip_step(x_min, x_max, x_last, xk, alpha, pk)
        p_last = -g(x_last)
        new_fval = f(x_last)
        diff = np.abs(new_fval - old_fval)
        rows.append((n_iter, float(x_last[0]), float(x_last[1]), new_fval) +
                    confusion(Xi, boolean_product(U, V, x_last[0], x_last[1])))
        improving = should_continue(ctl, n_iter=n_iter, diff=diff)
    return {"u": float(x_last[0]), "v": float(x_last[1]), "rows": rows, "calls": calls, "n_iter": n_iter}


# --------------------------------------------------------------------------------------
# ELBMF: elastic-net proximal updates, PALM / iPALM (models/ELBMF.py:110-210)
# The class fails as shipped (init_model calls normalize_UV(method=...), ELBMF.py:73 vs ContinuousModel.py:87); the
# module-level step functions and the iPALM loop body run and are what is restated (golden g14).
# --------------------------------------------------------------------------------------
def elbmf_integrality_gap(F, reg_l1, reg_l2) -> float:
    """sum of reg_l1 d + reg_l2 d^2, d = distance to the nearer of {0, 1} (ELBMF.py:166-174)."""
    dist = np.where(F < 0.5, np.abs(F), np.abs(F - 1))
    return float((reg_l1 * dist + reg_l2 * dist ** 2).sum())


def elbmf_prox(F, kai, lamda):
    """Proximal operator of the elastic-net penalty, negatives clamped to 0 (ELBMF.py:199-210)."""
    P = np.where(F <= 0.5, F - kai * np.sign(F), F - kai * np.sign(F - 1) + lamda)
    P = P / (1 + lamda)
    P[P < 0] = 0
    return P


def elbmf_step_size(G, beta, norm="spectral"):
    """eta from the Lipschitz constant of the gradient: 1 / (1.1 L) (PALM) or 2 (1 - beta) / (1 + 2 beta) / L (iPALM),
    L = max(||G||, 1e-4), spectral norm in ELBMF (ELBMF.py:184-185), Frobenius in PRIMP (PRIMP.py:73-80)."""
    L = max(float(np.linalg.norm(G, ord=2 if norm == "spectral" else "fro")), 1e-4)
    return 1 / (1.1 * L) if beta == 0 else 2 * (1 - beta) / (1 + 2 * beta) / L


def elbmf_update(X, F, G, W, reg_l1, reg_l2, beta, F_last, reassoc=False):
    """One Gauss-Seidel step for factor F against the other factor G (ELBMF.py:177-196; call it with X.T, V, U for V).
    Returns (F_new, F) -- the second value is what the caller keeps as `F_last`.  ``W=None`` = all-ones mask;
    ``reassoc``: (F G^T - X) G = F (G^T G) - X G, the association the HIP path uses."""
    F_cur, F_before = F, F_last
    eta = elbmf_step_size(G.T @ G, beta, "spectral")
    Fe = F_cur + beta * (F_cur - F_before)
    if reassoc:
        assert W is None
        grad = Fe @ (G.T @ G) - X @ G
    else:
        R = Fe @ G.T - X
        grad = (R if W is None else np.multiply(W, R)) @ G
    Fn = elbmf_prox(Fe - eta * grad, reg_l1 * eta, reg_l2 * eta)
    return Fn, F_cur


def elbmf_fit(X, U, V, W=None, reg_l1=0.01, reg_l2=0.02, reg_growth=1.02, beta=0.0, tol=0.0, max_iter=1000, min_diff=1e-8,
              reassoc=False):
    """The iPALM loop (ELBMF.py:110-163) from given initial factors.  Rows of ``updates``:
    (iter, reg_l1, reg_l2, gap, U_gap, V_gap, error); ``counts``: (TP, FP, FN, TN) of the factors thresholded at 0.5;
    ``scores``: (ERR, Accuracy, Recall, Precision, F1) as logged by evaluate(metrics=[...]) (ELBMF.py:144-155)."""
    X = np.asarray(X, dtype=np.float64)
    U, V = np.array(U, dtype=np.float64), np.array(V, dtype=np.float64)
    U_last, V_last = U.copy(), V.copy()
    ctl = {"tol": tol, "max_iter": max_iter, "min_diff": min_diff}
    gap = np.inf
    updates, counts, scores = [], [], []
    WT = None if W is None else W.T
    n_iter, improving = 0, True
    while improving:
        l1, l2 = reg_l1, reg_l2 * (reg_growth ** n_iter)
        # the V step sees the OLD U (the loop passes self.U, which is only replaced after both steps, ELBMF.py:124-125,135)
        U_new, U_last = elbmf_update(X, U, V, W, l1, l2, beta, U_last, reassoc)
        V_new, V_last = elbmf_update(X.T, V, U, WT, l1, l2, beta, V_last, reassoc)
        U, V = U_new, V_new
        err = float(np.power(X - U @ V.T, 2).sum())
        gu, gv = elbmf_integrality_gap(U, l1, l2), elbmf_integrality_gap(V, l1, l2)
        gap, gap_last = gu + gv, gap
        updates.append((n_iter, l1, l2, gap, gu, gv, err))
        c = confusion_counts(X.astype(np.int64), boolean_product(U, V, 0.5, 0.5))
        counts.append(c)
        rec, prec, acc, f1 = boolean_scores(*c)
        scores.append((1.0 - acc, acc, rec, prec, f1))
        improving = should_continue(ctl, error=gap, diff=abs(gap - gap_last), n_iter=n_iter)
        n_iter += 1
    return {"U": U, "V": V, "U_last": U_last, "V_last": V_last, "updates": updates, "counts": counts, "scores": scores}


# --------------------------------------------------------------------------------------
# PRIMP (models/PRIMP.py:51-160).  The class fails as shipped (_fit calls .toarray() on the already densified X_train,
# PRIMP.py:29); the module-level functions run (torch CPU tensors in the reference; NumPy here, in the dtype of the inputs).
# PRIMP stores V as k x n.
# --------------------------------------------------------------------------------------
def primp_prox(x, k, l):
    """proxelbmf (PRIMP.py:55-56): no clamp."""
    return np.where(x <= 0.5, x - k * np.sign(x), x - k * np.sign(x - 1) + l) / (1 + l)


def primp_step(X, U, Vt, U_anchor, l1reg, l2reg, tau, beta):
    """elbmf_step_ipalm (PRIMP.py:71-88).  ``Vt`` is k x n.  The inertial term extrapolates from ``U_anchor``, which the
    reference's loop never advances (its `Uold = U` rebinds a local), so the anchor stays the INITIAL factor.  Two prox
    applications: proxelbmfnn (max with 0) then _proxelbmfnn (min with 1, no lower clamp)."""
    VVt, XVt = Vt @ Vt.T, X @ Vt.T
    L = max(float(np.linalg.norm(VVt)), 1e-4)
    if beta != 0:
        U = U + beta * (U - U_anchor)
        step = 2 * (1 - beta) / (1 + 2 * beta) / L
    else:
        step = 1 / (1.1 * L)
    dt = U.dtype.type
    U = U - (U @ VVt - XVt) * dt(step)
    U = np.maximum(primp_prox(U, dt(l1reg * step), dt(l2reg * tau * step)), dt(0))
    U = np.minimum(primp_prox(U, dt(l1reg * step), dt(l2reg * tau * step)), dt(1))
    return U


def primp_ipalm(X, U, Vt, l1reg, l2reg, reg_growth, maxiter, tolerance, beta):
    """elbmf_ipalm (PRIMP.py:91-131): returns (U, Vt, [||X - U Vt||_F^2 per iteration])."""
    Ua, Va = (U.copy(), Vt.T.copy()) if beta != 0 else (None, None)
    fn, fns = np.inf, []
    for t in range(maxiter):
        tau = reg_growth ** t
        U = primp_step(X, U, Vt, Ua, l1reg, l2reg, tau, beta)
        Vt = primp_step(X.T, Vt.T, U.T, Va, l1reg, l2reg, tau, beta).T
        fn0, fn = fn, float(np.linalg.norm(X - U @ Vt) ** 2)
        fns.append(fn)
        if abs(fn - fn0) < tolerance:
            break
    return U, Vt, fns


def primp_round(F, l2reg=0.0):
    """with_rounding (PRIMP.py:155-158): proxelbmfnn(F, 0.5, l2reg * 1e12).round()."""
    return np.round(np.maximum(primp_prox(F, 0.5, l2reg * 1e12), 0))
