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

__all__ = ["to_sparse", "to_dense", "ismat", "isnum", "binarize", "header", "record", "ignore_warnings",
           "scores_from_counts", "get_cache_path", "_make_name", "get_prediction", "get_prediction_with_threshold"]


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


def ignore_warnings(func):
    """utils/decorator_utils.py:16-24"""
    def inner(*args, **kwargs):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            return func(*args, **kwargs)
    inner.__name__ = getattr(func, "__name__", "inner")
    return inner


def scores_from_counts(tp, fp, fn, tn):
    """Recall, Precision, Accuracy, F1 from the four integer counts with the reference's edge rules
    (utils/metrics.py:79-135): a ratio with an empty denominator is 0."""
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


def __getattr__(name):
    # the device-backed metrics live in utils/metrics.py and pull in torch + the HIP library: import them on first use
    if name in ("TP", "FP", "TN", "FN", "TPR", "PPV", "ACC", "ERR", "F1", "weighted_error", "coverage_score", "description_length",
                "get_metrics", "confusion"):
        from . import metrics
        return getattr(metrics, name)
    raise AttributeError(name)
