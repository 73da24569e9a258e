"""Cover scores on the GPU with the names and signatures of ``PyBMF/utils/metrics.py``: the confusion
counts TP / FP / TN / FN (overall, or per row / per column with ``axis``), the ratios built on them, and the three
coverage costs the heuristic models rank candidates with -- ``coverage_score``, ``weighted_error``, ``description_length``.

Inputs are Boolean matrices (ndarray / scipy sparse / engine.BitMatrix).  Every count is an integer popcount over bit
matrices in HBM (``bmf_confusion_rows`` / ``bmf_popcount`` / ``bmf_cover_count``); nothing m x n is formed on the host.
``axis`` follows the reference: ``axis=0`` sums over rows (one value per column), ``axis=1`` one value per row.
"""
from __future__ import annotations

import numpy as np
import torch

from .. import _lib as L
from .._lib import check, lib, ptr

DEFAULT_DEVICE = "cuda:0"


def _bits(X, device=DEFAULT_DEVICE):
    from ..engine import BitMatrix
    return X if isinstance(X, BitMatrix) else BitMatrix(X, device)


def _stream():
    import ctypes as C
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def confusion(gt, pd, axis=None, device=DEFAULT_DEVICE):
    """(TP, FP, FN, TN) as python ints (axis=None) or int64 arrays (axis=0: per column, axis=1: per row)."""
    G, P = _bits(gt, device), _bits(pd, device)
    assert (G.m, G.n) == (P.m, P.n), "gt and pd must have the same shape"
    if axis not in (None, 0, 1):
        raise ValueError("axis must be None, 0 or 1")
    with torch.cuda.device(G.device):
        if axis == 0:   # per column of X = per row of X^T
            g, p, rows, words, ldg, ldp, other = G.bits_t, P.bits_t, G.n, G.m_pad // 32, G.ldxt, P.ldxt, G.m
        else:
            g, p, rows, words, ldg, ldp, other = G.bits, P.bits, G.m, G.n_pad // 32, G.ldx, P.ldx, G.n
        tp = torch.zeros(rows, dtype=torch.int32, device=G.device)
        fp = torch.zeros(rows, dtype=torch.int32, device=G.device)
        check(lib.bmf_confusion_rows(ptr(g), ldg, ptr(p), ldp, rows, words, ptr(tp), ptr(fp), _stream()), "bmf_confusion_rows")
        ones = torch.zeros(rows, dtype=torch.int32, device=G.device)   # row sums of gt: |G_r and G_r|
        dummy = torch.zeros(rows, dtype=torch.int32, device=G.device)
        check(lib.bmf_confusion_rows(ptr(g), ldg, ptr(g), ldg, rows, words, ptr(ones), ptr(dummy), _stream()), "bmf_confusion_rows")
        tp, fp, ones = (t.cpu().numpy().astype(np.int64) for t in (tp, fp, ones))
    fn = ones - tp
    tn = other - tp - fp - fn
    if axis is None:
        return int(tp.sum()), int(fp.sum()), int(fn.sum()), int(tn.sum())
    return tp, fp, fn, tn


def TP(gt, pd, axis=None):
    """utils/metrics.py:56-58"""
    return confusion(gt, pd, axis)[0]


def FP(gt, pd, axis=None):
    """utils/metrics.py:61-68"""
    return confusion(gt, pd, axis)[1]


def FN(gt, pd, axis=None):
    """utils/metrics.py:75-76"""
    return confusion(gt, pd, axis)[2]


def TN(gt, pd, axis=None):
    """utils/metrics.py:71-72"""
    return confusion(gt, pd, axis)[3]


def _ratio(num, den):
    return num / den if den > 0 else 0


def TPR(gt, pd, axis=None):
    """Recall (utils/metrics.py:79-83; like the reference, the ratio metrics are defined for axis=None)."""
    tp, fp, fn, tn = confusion(gt, pd, None)
    return _ratio(tp, tp + fn)


def PPV(gt, pd, axis=None):
    """Precision (utils/metrics.py:105-109)."""
    tp, fp, fn, tn = confusion(gt, pd, None)
    return _ratio(tp, tp + fp)


def ACC(gt, pd, axis=None):
    """Accuracy (utils/metrics.py:112-119): (TP + TN) / n with n = the number of cells (axis=None) or pd.shape[axis]."""
    tp, fp, fn, tn = confusion(gt, pd, axis)
    G = _bits(gt)
    n = G.m * G.n if axis is None else (G.m, G.n)[axis]
    return (tp + tn) / n


def ERR(gt, pd, axis=None):
    return 1 - ACC(gt, pd, axis)


def F1(gt, pd, axis=None):
    """utils/metrics.py:128-139"""
    p, r = PPV(gt, pd), TPR(gt, pd)
    return 2 * p * r / (p + r) if p + r > 0 else 0


def weighted_error(gt, pd, w_fp=0.5, w_fn=None, axis=None):
    """Coverage cost to be minimised: w_fp FP + w_fn FN (utils/metrics.py:182-186)."""
    w_fn = 1 - w_fp if w_fn is None else w_fn
    tp, fp, fn, tn = confusion(gt, pd, axis)
    return w_fp * fp + w_fn * fn


def coverage_score(gt, pd, w_fp=0.5, w_fn=None, axis=None):
    """Coverage score to be maximised: -w_fp FP + w_fn TP (utils/metrics.py:189-201)."""
    w_fn = 1 - w_fp if w_fn is None else w_fn
    tp, fp, fn, tn = confusion(gt, pd, axis)
    return -w_fp * fp + w_fn * tp


def description_length(gt, U, V, pd=None, w_model=1.0, w_fp=1.0, w_fn=1.0, device=DEFAULT_DEVICE):
    """w_model (|U| + |V|) + w_fp FP + w_fn FN (utils/metrics.py:173-179).  With ``pd=None`` the Boolean product of the 0/1
    factors is never materialised: FP / FN come from the cover-count kernel."""
    Ub = np.asarray(U.todense() if hasattr(U, "todense") else U) != 0
    Vb = np.asarray(V.todense() if hasattr(V, "todense") else V) != 0
    model = int(Ub.sum()) + int(Vb.sum())
    if pd is not None:
        tp, fp, fn, tn = confusion(gt, pd, None, device)
    else:
        from ..device_ops import _bits_of
        G = _bits(gt, device)
        k = Ub.shape[1]
        if k > L.MAX_KP:
            raise NotImplementedError(f"k={k}: this build supports k <= {L.MAX_KP}")
        rb, _, kp = _bits_of(Ub, G.m_pad)
        _, cb, _ = _bits_of(Vb, G.n_pad)
        with torch.cuda.device(G.device):
            rbd, cbd = torch.from_numpy(rb).to(G.device), torch.from_numpy(np.ascontiguousarray(cb)).to(G.device)
            cnt = torch.zeros(4, dtype=torch.int64, device=G.device)
            check(lib.bmf_cover_count(ptr(G.bits), G.m_pad, G.ldx, G.n_pad // 32, ptr(rbd), ptr(cbd), G.n_pad // 32, kp, ptr(cnt),
                                      None, _stream()), "bmf_cover_count")
            tp, fp = (int(x) for x in cnt[:2].cpu().numpy())
        fn = G.sum_local - tp
    return w_model * model + w_fp * fp + w_fn * fn


def get_metrics(gt, pd, metrics, axis=None):
    """utils/metrics.py:8-53 for the Boolean metrics (one device pass, then the formulas)."""
    table = {"TP": TP, "FP": FP, "TN": TN, "FN": FN, "TPR": TPR, "PPV": PPV, "ACC": ACC, "ERR": ERR, "F1": F1,
             "Recall": TPR, "Precision": PPV, "Accuracy": ACC, "Error": ERR}
    G, P = _bits(gt), _bits(pd)
    return [table[m](G, P, axis) if m in table else None for m in metrics]
