"""Host-side helpers with the names the reference exposes in ``PyBMF.utils`` for this path.

Only thin plumbing lives here (container conversion, log tables, score formulas on integer counts); every m x n
reduction is done on the GPU by libbmf_hip.  Citations are to /root/reference/PyBMF.
"""
from __future__ import annotations

import os
import re
import warnings

import numpy as np
import pandas as pd
from scipy.sparse import coo_matrix, csc_matrix, csr_matrix, issparse, lil_matrix, spmatrix

__all__ = ["to_sparse", "to_dense", "ismat", "isnum", "binarize", "header", "record", "record_many", "ignore_warnings",
           "scores_from_counts", "get_cache_path", "_make_name", "get_prediction", "get_prediction_with_threshold",
           "check_sparse", "to_triplet", "multiply", "dot", "matmul", "add", "subtract", "power", "sigmoid", "d_sigmoid"]


def to_sparse(X, type="csr"):
    """utils/sparse_utils.py:5-20"""
    kinds = {"coo": coo_matrix, "csr": csr_matrix, "csc": csc_matrix, "lil": lil_matrix}
    assert type in kinds, "Matrix type not available"
    return kinds[type](X)


def to_dense(X, squeeze=False):
    """utils/sparse_utils.py:23-36 (keep_nan variant not needed on this path)"""
    if issparse(X):
        X = X.toarray()
    elif isinstance(X, np.matrix):
        X = np.asarray(X)
    return X.squeeze() if squeeze else X


def ismat(X):
    """utils/boolean_utils.py:151-159"""
    return isinstance(X, (np.ndarray, spmatrix))


def isnum(X):
    return isinstance(X, (int, float))


def binarize(X, threshold=0.5):
    """Heaviside step with a strict '>' (utils/common.py:64-79)."""
    Y = (X > threshold).astype(int)
    return to_sparse(Y, type=X.format) if isinstance(X, spmatrix) else Y


def header(names, levels, depth=None):
    """Multi-level column tuples, the name sits on level `depth` (utils/evaluate_utils.py:85-98)."""
    depth = levels if depth is None else depth
    out = []
    for name in names:
        t = [""] * levels
        t[depth - 1] = name
        out.append(tuple(t))
    return out


def record(df_dict, df_name, columns, records, verbose=False):
    """Append one timestamped row to logs[df_name], creating the table on first use (utils/evaluate_utils.py:57-82)."""
    if df_name not in df_dict:
        if isinstance(columns[0], tuple):
            cols = pd.MultiIndex.from_tuples(header(["time"], levels=len(columns[0])) + list(columns))
        else:
            cols = ["time"] + list(columns)
        df_dict[df_name] = pd.DataFrame(columns=cols)
    stamp = [pd.Timestamp.now().strftime("%d/%m/%y %I:%M:%S")]
    df = df_dict[df_name]
    df.loc[len(df.index)] = stamp + list(records)
    if verbose:
        print(df.tail())


def record_many(df_dict, df_name, columns, rows):
    """`record` for a whole trajectory at once: the rows a fit() logged on the device become one table append instead of one
    pandas append per row (which copies the table every time: 12 of the 19 ms of a config-#1 fit)."""
    rows = [list(r) for r in rows]
    if not rows:
        return
    stamp = pd.Timestamp.now().strftime("%d/%m/%y %I:%M:%S")
    if df_name not in df_dict:   # the whole table in one constructor call (an empty table + a first .loc row + a concat cost 2 ms more)
        if isinstance(columns[0], tuple):
            cols = pd.MultiIndex.from_tuples(header(["time"], levels=len(columns[0])) + list(columns))
        else:
            cols = ["time"] + list(columns)
        df_dict[df_name] = pd.DataFrame([[stamp] + r for r in rows], columns=cols, dtype=object)
        return
    df = df_dict[df_name]
    new = pd.DataFrame([[stamp] + r for r in rows], columns=df.columns, index=range(len(df.index), len(df.index) + len(rows)), dtype=object)
    df_dict[df_name] = pd.concat([df, new])


def ignore_warnings(func):
    """utils/decorator_utils.py:16-24"""
    def inner(*args, **kwargs):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            return func(*args, **kwargs)
    inner.__name__ = getattr(func, "__name__", "inner")
    return inner


def scores_from_counts(tp, fp, fn, tn, sums=None):
    """Recall, Precision, Accuracy, F1 from the four integer counts with the reference's edge rules
    (utils/metrics.py:79-135): a ratio with an empty denominator is 0."""
    if sums is not None:
        # real-valued ground truth: the four "counts" are the reference's arithmetic sums over two csr matrices (utils/metrics.py:56-77:
        # TP = sum gt pd, TN = sum (1 - gt)(1 - pd)), and the denominators are sum gt, sum pd and the number of cells (:79-117)
        n_gt, n_pd, cells = (float(v) for v in sums)
        recall = np.float64(tp) / n_gt if n_gt > 0 else 0
        precision = np.float64(tp) / n_pd if n_pd > 0 else 0
        accuracy = (np.float64(tp) + np.float64(tn)) / cells
        s = precision + recall
        return recall, precision, accuracy, (2 * precision * recall / s if s > 0 else 0)
    tp, fp, fn, tn = int(tp), int(fp), int(fn), int(tn)
    n_gt, n_pd = tp + fn, tp + fp
    recall = np.float64(tp) / n_gt if n_gt > 0 else 0
    precision = np.float64(tp) / n_pd if n_pd > 0 else 0
    accuracy = np.float64(tp + tn) / (tp + fp + fn + tn)
    s = precision + recall
    f1 = 2 * precision * recall / s if s > 0 else 0
    return recall, precision, accuracy, f1


def get_cache_path(relative_path=None):
    """~/.pybmf/<relative_path>, directories created (utils/download_utils.py:75-90)."""
    root = os.path.join(os.path.expanduser("~"), ".pybmf")
    full = root if relative_path is None else os.path.join(root, relative_path)
    os.makedirs(os.path.dirname(full) if relative_path else full, exist_ok=True)
    return full, root


def _make_name(model=None, model_name=None, format="%Y-%m-%d %H-%M-%S-%f "):
    """Timestamp + class name (utils/dataframe_utils.py:179-205)."""
    if model_name is None:
        model_name = type(model).__name__ if model is not None else "model"
        model_name = re.sub(r"[^0-9A-Za-z]", "", model_name)
    return pd.Timestamp.now().strftime(format) + model_name


def get_prediction(U, V, boolean=True, sparse=True):
    """Real or Boolean product of two host factors as a csr matrix (utils/common.py:98-107).  Host convenience for
    small matrices (the fitted models materialise their own X_pd on the GPU)."""
    from ..device_ops import product_csr
    return product_csr(np.asarray(to_dense(U)), np.asarray(to_dense(V)), boolean=boolean)


def get_prediction_with_threshold(U, V, u=None, v=None, us=None, vs=None, sparse=True):
    """min(1, (U > u) @ (V > v)^T) as csr, computed with the bit kernels (utils/common.py:110-151)."""
    from ..device_ops import boolean_product_csr
    return boolean_product_csr(np.asarray(to_dense(U)), np.asarray(to_dense(V)), u=u, v=v, us=us, vs=vs)


# ---- container helpers of utils/boolean_utils.py / sparse_utils.py / common.py: host plumbing with the reference's names and
# ---- dispatch rules (dense in -> dense out, sparse in -> sparse out)
def check_sparse(X, sparse=None):
    """utils/sparse_utils.py:49-55"""
    if sparse is True and not issparse(X):
        return to_sparse(X)
    if sparse is False and issparse(X):
        return to_dense(X)
    return X


def to_triplet(X):
    """(rows, cols, values) of the stored entries (utils/sparse_utils.py:39-46)"""
    coo = coo_matrix(X)
    return np.asarray(coo.row, dtype=int), np.asarray(coo.col, dtype=int), np.asarray(coo.data, dtype=float)


def multiply(U, V, sparse=None, boolean=False):
    """Element-wise product (utils/boolean_utils.py:6-33)."""
    if ismat(U) and ismat(V):
        assert U.shape == V.shape, "U and V should have the same shape"
        if issparse(U) or issparse(V) or sparse:
            X = csr_matrix(U).multiply(csr_matrix(V))
        else:
            X = np.logical_and(U, V).astype(int) if boolean else np.multiply(U, V)
    else:
        X = U * V
    return check_sparse(X, sparse=sparse)


def dot(u, v, boolean=False):
    """Inner product of two vectors, OR-of-ANDs when boolean (utils/boolean_utils.py:36-58)."""
    if issparse(u) or issparse(v):
        u, v = csr_matrix(u), csr_matrix(v)
        assert u.shape == v.shape, "U and V should have the same shape"
        x = u.multiply(v).sum()
        return (x > 0).astype(int) if boolean else x   # NumPy integer, like the reference
    assert np.shape(u) == np.shape(v), "U and V should have the same shape"
    return np.any(np.logical_and(u, v), axis=-1).astype(int) if boolean else np.dot(u, v)


def matmul(U, V, sparse=None, boolean=False):
    """Matrix product, min(1, .) when boolean (utils/boolean_utils.py:61-84).  Host container helper for small operands; the
    m x n Boolean product of factors lives on the GPU (device_ops.boolean_product_bits / get_prediction_with_threshold)."""
    sparse = bool(sparse or issparse(U) or issparse(V))
    assert U.shape[1] == V.shape[0], "U and V should be multiplicable"
    if sparse:
        X = csr_matrix(U) @ csr_matrix(V)
        X = X.minimum(1).astype(int) if boolean else X
    else:
        X = U @ V
        X = np.minimum(X, 1).astype(int) if boolean else X
    return check_sparse(X, sparse=sparse)


def add(X, Y, sparse=None, boolean=False):
    """utils/boolean_utils.py:87-107"""
    if isnum(X) or isnum(Y):
        X = to_dense(X) if issparse(X) else X
        Y = to_dense(Y) if issparse(Y) else Y
    Z = np.add(X, Y).astype(bool).astype(float) if boolean else X + Y
    return check_sparse(Z, sparse=bool(sparse or issparse(X) or issparse(Y)))


def subtract(X, Y, sparse=False, boolean=False):
    """utils/boolean_utils.py:110-133: matrix - constant as in the reference; matrix - matrix, which the reference's version cannot
    reach (it only defines its operands in the constant branch: UnboundLocalError), does the evident thing here."""
    Xd = to_dense(X) if issparse(X) and (isnum(X) or isnum(Y)) else X
    Yd = to_dense(Y) if issparse(Y) and (isnum(X) or isnum(Y)) else Y
    Z = np.subtract(Xd, Yd).astype(bool).astype(float) if boolean else Xd - Yd
    return check_sparse(Z, sparse=bool(sparse or issparse(Xd) or issparse(Yd)))


def power(X, n):
    """utils/boolean_utils.py:136-142"""
    return X.power(n).astype(np.float64) if issparse(X) else np.power(X, n).astype(np.float64)


def sigmoid(X):
    """Piecewise-stable logistic (utils/common.py:82-89)."""
    if issparse(X):   # (the reference fails on the masked assignment with the same exception type)
        raise NotImplementedError("sigmoid takes a dense array")
    X = np.asarray(X, dtype=np.float64)
    Y = np.empty(X.shape)
    pos = X >= 0
    Y[pos] = 1.0 / (1.0 + np.exp(-X[pos]))
    e = np.exp(X[~pos])
    Y[~pos] = e / (1.0 + e)
    return Y


def d_sigmoid(X):
    """sigmoid'(X) = sigmoid(X) (1 - sigmoid(X)) (utils/common.py:92-95)"""
    Y = sigmoid(X)
    return Y * (1 - Y)


def __getattr__(name):
    # the device-backed metrics live in utils/metrics.py and pull in torch + the HIP library: import them on first use
    if name in ("TP", "FP", "TN", "FN", "TPR", "PPV", "ACC", "ERR", "F1", "weighted_error", "coverage_score", "description_length",
                "get_metrics", "confusion"):
        from . import metrics
        return getattr(metrics, name)
    raise AttributeError(name)
