"""Differential checks of the HOST-side pieces against the reference itself, imported from /root/reference when that exists
(the build container; never on the GPU box -- the tests skip there).  Golden vectors pin fixed cases; these run the two
implementations side by side on randomised inputs: the synthetic generator (bit-identical matrices), the line search and
its step limiter, the stopping rule with its messages, and the container helpers of utils.  Nothing here touches the GPU."""
import contextlib
import io
import itertools
import os
import sys
import warnings

import numpy as np
import pytest
from scipy.sparse import csc_matrix, csr_matrix, issparse, lil_matrix

HERE = os.path.dirname(os.path.abspath(__file__))
pytestmark = pytest.mark.skipif(not os.path.isdir("/root/reference/PyBMF"), reason="the reference is only mounted in the build container")


@pytest.fixture(scope="module")
def ref():
    sys.path.insert(0, os.path.join(HERE, "golden"))
    import make_golden as mg
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        mg.load_reference()
    return sys.modules


def outcome(fn, *args, **kw):
    buf = io.StringIO()
    try:
        with contextlib.redirect_stdout(buf), warnings.catch_warnings():
            warnings.simplefilter("ignore")
            return fn(*args, **kw), buf.getvalue(), None
    except Exception as e:  # noqa: BLE001  (the exception TYPE is part of the comparison)
        return None, buf.getvalue(), type(e).__name__


def dense(x):
    return np.asarray(x.todense()) if issparse(x) else np.asarray(x)


def test_generator_is_bit_identical(ref):
    from PyBMF.generators import SyntheticMatrixGenerator as R
    from pybmf_amd.generators import SyntheticMatrixGenerator as M
    rs = np.random.RandomState(5)
    for _ in range(25):
        m, n, k = int(rs.randint(1, 300)), int(rs.randint(1, 300)), int(rs.randint(1, 12))
        dens = [float(rs.uniform(0.02, 0.6)), float(rs.uniform(0.02, 0.6))]
        seed, nseed = int(rs.randint(0, 10 ** 6)), int(rs.randint(0, 10 ** 6))
        noise = [float(rs.uniform(0, 0.2)), float(rs.uniform(0, 0.1))]
        got = []
        for G in (R, M):
            def run():
                g = G(m=m, n=n, k=k, density=dens)
                g.generate(seed=seed)
                clean = dense(g.X).astype(np.uint8)
                g.add_noise(noise=noise, seed=nseed)
                return clean, dense(g.X).astype(np.uint8), dense(g.U).astype(np.uint8), dense(g.V).astype(np.uint8)
            got.append(outcome(run))
        (a, _, ea), (b, _, eb) = got
        assert ea == eb, (m, n, k, ea, eb)
        if ea is None:
            assert all(np.array_equal(x, y) for x, y in zip(a, b)), (m, n, k, dens, seed, nseed, noise)


def test_line_search_and_step_limit(ref):
    Rm, Mm = ref["PyBMF.solvers.line_search"], __import__("pybmf_amd.solvers.line_search", fromlist=["x"])
    rs = np.random.RandomState(0)
    for t in range(200):
        A = rs.rand(2, 2)
        A = A @ A.T + 0.1 * np.eye(2)
        b = rs.randn(2)
        f, g = [(lambda x: 0.5 * x @ A @ x - b @ x, lambda x: A @ x - b), (lambda x: np.sum(np.cosh(x - b)), lambda x: np.sinh(x - b)),
                (lambda x: np.sum((x - b) ** 4) + 0.1 * np.sum(x ** 2), lambda x: 4 * (x - b) ** 3 + 0.2 * x)][t % 3]
        xk = rs.rand(2)
        pk = -g(xk)
        (r, _, er), (m, _, em) = outcome(Rm.line_search, f, g, xk, pk), outcome(Mm.line_search, f, g, xk, pk)
        assert er == em
        if er is None:
            for x, y in zip(r, m):
                assert (x is None and y is None) or np.allclose(np.asarray(x, float), np.asarray(y, float), rtol=1e-12, equal_nan=True)
    flat = lambda z: np.concatenate([np.ravel(np.asarray(q, float)) for q in (z if isinstance(z, (tuple, list)) else [z])])  # noqa: E731
    for _ in range(1000):
        x_min, x_max = sorted(rs.rand(2) * 2 - 0.5)
        args = (x_min, x_max, rs.rand(2), rs.rand(2) * 1.5 - 0.25, rs.rand() * 3, rs.randn(2))
        (r, _, er), (m, _, em) = outcome(Rm.limit_step_size, *args), outcome(Mm.limit_step_size, *args)
        assert er == em and (er is not None or np.allclose(flat(r), flat(m), rtol=1e-13, equal_nan=True)), args


def test_stopping_rule_and_its_messages(ref):
    from PyBMF.models.BaseModelTools import BaseModelTools as R
    from pybmf_amd.models.BaseModelTools import BaseModelTools as M

    class RP(R):
        def __init__(self):
            pass

    class MP(M):
        def __init__(self):
            pass
    rs = np.random.RandomState(3)
    for _ in range(2000):
        attrs, kw = {}, {}
        if rs.rand() < 0.7:
            attrs["tol"] = float(rs.choice([0.0, 0.01, 1.0, -1.0]))
        if rs.rand() < 0.7:
            attrs["min_diff"] = float(rs.choice([0.0, 1e-3, 0.5]))
        if rs.rand() < 0.7:
            attrs["max_iter"] = int(rs.choice([0, 1, 5, 100]))
        if rs.rand() < 0.3:
            attrs["k"] = int(rs.choice([1, 3]))
        if rs.rand() < 0.7:
            kw["error"] = float(rs.choice([0.0, 0.005, 0.5, 2.0]))
        if rs.rand() < 0.7:
            kw["diff"] = float(rs.choice([0.0, 1e-4, 0.1, 1.0]))
        if rs.rand() < 0.7:
            kw["n_iter"] = int(rs.choice([0, 1, 2, 6, 101]))
        if rs.rand() < 0.2:
            kw["n_factor"] = int(rs.choice([1, 3, 4]))
        res = []
        for P in (RP, MP):
            p = P()
            for a, v in attrs.items():
                setattr(p, a, v)
            res.append(outcome(p.early_stop, **kw))
        assert res[0] == res[1], (attrs, kw, res)


def test_container_helpers(ref):
    import PyBMF.utils as R
    import pybmf_amd.utils as M
    rs = np.random.RandomState(0)

    def forms(A):
        return [A, csr_matrix(A), lil_matrix(A), csc_matrix(A)]

    def same(name, fr, fm, *args, **kw):
        (r, _, er), (m, _, em) = outcome(fr, *args, **kw), outcome(fm, *args, **kw)
        assert er == em, (name, er, em)
        if er is None:
            assert type(r).__name__ == type(m).__name__ or (issparse(r) and issparse(m) and r.format == m.format), (name, type(r), type(m))
            assert np.allclose(dense(r).astype(float), dense(m).astype(float), equal_nan=True), name
    A, B, C = ((rs.rand(*s) < 0.4).astype(float) for s in ((7, 5), (7, 5), (5, 6)))
    for a, b in itertools.product(forms(A), forms(B)):
        for boolean in (False, True):
            same("multiply", R.multiply, M.multiply, a, b, boolean=boolean)
            same("add", R.add, M.add, a, b, boolean=boolean)
    for a, c in itertools.product(forms(A), forms(C)):
        for boolean, sparse in itertools.product((False, True), (False, True)):
            same("matmul", R.matmul, M.matmul, a, c, boolean=boolean, sparse=sparse)
    u, v = A[:, 0], B[:, 0]
    for a, b in itertools.product([u, csr_matrix(u), lil_matrix(u)], [v, csr_matrix(v)]):
        for boolean in (False, True):
            same("dot", R.dot, M.dot, a, b, boolean=boolean)
    for a in forms(rs.rand(7, 5)):
        same("power", R.power, M.power, a, 2)
        same("sigmoid", R.sigmoid, M.sigmoid, a)
        same("d_sigmoid", R.d_sigmoid, M.d_sigmoid, a)
        same("to_dense", R.to_dense, M.to_dense, a)
        same("binarize", R.binarize, M.binarize, a, 0.5)
        same("to_triplet", lambda x: np.array(R.to_triplet(x)), lambda x: np.array(M.to_triplet(x)), a)
        for fmt in ("csr", "lil", "csc"):
            same("to_sparse", R.to_sparse, M.to_sparse, a, fmt)
    for a in (A, csr_matrix(A)):      # matrix - constant (matrix - matrix cannot be reached in the reference: see utils.subtract)
        same("subtract", R.subtract, M.subtract, a, 1.0)
        same("subtract", R.subtract, M.subtract, 1.0, a)


# ---- the oracle itself against the reference's classes, on random small problems -------------------------------------------
FITKW = dict(task="reconstruction", show_logs=False, show_result=False, save_model=False)


def frame(df):
    return np.array([[float(v) for v in r[1:]] for r in df.values.tolist()])


def quiet_fit(model, *data, **kw):
    with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        model.fit(*data, **{**FITKW, **kw})
    return model


def random_problem(rs, real=False):
    m, n, k = int(rs.randint(2, 60)), int(rs.randint(2, 50)), int(rs.randint(1, 8))
    if real:
        X = rs.rand(m, max(k, 2)) @ rs.rand(max(k, 2), n) / max(k, 2) + 0.01 * rs.rand(m, n)
        X[rs.rand(m, n) < 0.1] = 0.0
    else:
        X = (rs.rand(m, n) < rs.uniform(0.15, 0.7)).astype(np.float64)
        X[rs.randint(m), rs.randint(n)] = 1.0
    U0 = np.abs(rs.standard_normal((m, k))) * 0.3 + 1e-3
    V0 = np.abs(rs.standard_normal((n, k))) * 0.3 + 1e-3
    return m, n, k, X, U0, V0


def random_mask(rs, X, weights):
    """A mask that keeps every row and column of X observed with a non-zero somewhere (see WNMF.py docstring in pybmf_amd)."""
    m, n = X.shape
    W = (rs.rand(m, n) < rs.uniform(0.4, 0.9)).astype(np.float64)
    W[X != 0] = 1.0
    if weights:
        W = W * rs.choice([0.5, 1.0, 3.0], size=(m, n))
    return W


def test_oracle_penalty_fit_against_the_reference(ref):
    import oracle as orc
    from PyBMF.models import BinaryMFPenalty
    rs = np.random.RandomState(11)
    for t in range(12):
        m, n, k, X, U0, V0 = random_problem(rs)
        reg, growth, iters = float(rs.choice([0.0, 0.5, 2.0])), float(rs.choice([1.0, 1.3, 3.0])), int(rs.randint(1, 5))
        mode = t % 2   # all-ones mask / W='mask' on a csr with explicit zeros.  (A weight MATRIX cannot be given to the reference:
        #                  its `self.W in ['mask', 'full']` check raises on an ndarray and on scipy sparse alike.)
        W = None if mode == 0 else random_mask(rs, X, weights=False)
        if mode == 1:
            r, c = np.nonzero(W)
            data, Wref = csr_matrix((X[r, c], (r, c)), shape=(m, n)), "mask"
        else:
            data, Wref = X.copy(), "full"
        mdl = quiet_fit(BinaryMFPenalty(k=k, U=U0.copy(), V=V0.copy(), W=Wref, reg=reg, reg_growth=growth, init_method="custom",
                                        normalize_method="balance", max_iter=iters, tol=-1.0), data)
        got = orc.penalty_fit(X * (W != 0) if W is not None else X, k=k, U=U0.copy(), V=V0.copy(), reg=reg, reg_growth=growth,
                              init_method="custom", normalize_method="balance", max_iter=iters, tol=-1.0, W=W)
        np.testing.assert_allclose(got["U"], np.asarray(mdl.U), rtol=1e-9, atol=1e-300, err_msg=str((t, mode)))
        np.testing.assert_allclose(got["V"], np.asarray(mdl.V), rtol=1e-9, atol=1e-300)
        np.testing.assert_allclose(np.array(got["updates"]), frame(mdl.logs["updates"]), rtol=1e-9)
        np.testing.assert_allclose(np.array(got["boolean"]), frame(mdl.logs["boolean"]), rtol=1e-13, atol=0)


def test_oracle_wnmf_fits_against_the_reference(ref):
    import oracle as orc
    from PyBMF.models import WNMF
    rs = np.random.RandomState(12)
    for t in range(12):
        m, n, k, X, U0, V0 = random_problem(rs, real=(t % 2 == 0))
        iters = int(rs.randint(1, 5))
        W = None if t % 3 == 0 else random_mask(rs, X, weights=False)
        if W is None:
            data, Wref = X.copy(), "full"
        else:
            r, c = np.nonzero(W)
            data, Wref = csr_matrix((X[r, c], (r, c)), shape=(m, n)), "mask"
        mdl = quiet_fit(WNMF(k=k, U=U0.copy(), V=V0.copy(), W=Wref, init_method="custom", max_iter=iters, tol=-1.0), data)
        got = orc.wnmf_fit(X.copy(), k, U=U0.copy(), V=V0.copy(), W=W, max_iter=iters, init_method="custom", tol=-1.0)
        np.testing.assert_allclose(got["U"], np.asarray(mdl.U), rtol=1e-9, atol=1e-300, err_msg=str(t))
        np.testing.assert_allclose(got["V"], np.asarray(mdl.V), rtol=1e-9, atol=1e-300)
        np.testing.assert_allclose(np.array(got["updates"]), frame(mdl.logs["updates"]), rtol=1e-9)
    for t in range(6):   # Kullback-Leibler, Boolean X with every row and column non-empty
        m, n, k, X, U0, V0 = random_problem(rs)
        X[np.arange(m), rs.randint(n, size=m)] = 1.0
        X[rs.randint(m, size=n), np.arange(n)] = 1.0
        iters = int(rs.randint(1, 4))
        masked = t % 2 == 1
        mdl = quiet_fit(WNMF(k=k, U=U0.copy(), V=V0.copy(), W="mask" if masked else "full", beta_loss="kullback-leibler", init_method="custom",
                             max_iter=iters), X.copy())
        got = orc.wnmf_kl_fit(X.copy(), k, U=U0.copy(), V=V0.copy(), W=(X != 0).astype(np.float64) if masked else None, max_iter=iters,
                              init_method="custom")
        np.testing.assert_allclose(got["U"], np.asarray(mdl.U), rtol=1e-9, atol=1e-300, err_msg=str(t))
        np.testing.assert_allclose(np.array(got["updates"]), frame(mdl.logs["updates"]), rtol=1e-9)


def test_oracle_pnlpf_and_threshold_against_the_reference(ref):
    import oracle as orc
    from PyBMF.models import PNLPF, BinaryMFThreshold
    rs = np.random.RandomState(13)
    for t in range(6):
        m, n, k, X, U0, V0 = random_problem(rs)
        iters, lam = int(rs.randint(1, 4)), float(rs.choice([5, 10, 20]))
        mdl = quiet_fit(PNLPF(k=k, U=U0.copy(), V=V0.copy(), W="full", reg=1.0, link_lamda=lam, reg_growth=1.2, init_method="custom",
                              normalize_method="balance", max_iter=iters, tol=-1.0), X.copy())
        got = orc.pnlpf_fit(X, k=k, U=U0.copy(), V=V0.copy(), reg=1.0, link_lamda=lam, reg_growth=1.2, init_method="custom",
                            normalize_method="balance", max_iter=iters, tol=-1.0)
        np.testing.assert_allclose(got["U"], np.asarray(mdl.U), rtol=1e-9, atol=1e-300, err_msg=str(t))
        np.testing.assert_allclose(np.array(got["updates"]), frame(mdl.logs["updates"]), rtol=1e-9)
        np.testing.assert_allclose(np.array(got["boolean"]), frame(mdl.logs["boolean"]), rtol=1e-13, atol=0)
    for t in range(6):
        m, n, k, X, _, _ = random_problem(rs)
        U, V = rs.rand(m, k), rs.rand(n, k)
        lam, u0, v0 = float(rs.choice([5, 10, 50])), float(rs.uniform(0.2, 0.8)), float(rs.uniform(0.2, 0.8))
        mdl = quiet_fit(BinaryMFThreshold(k=k, U=U.copy(), V=V.copy(), W="full", u=u0, v=v0, lamda=lam, min_diff=1e-3, max_iter=15), X.copy())
        got = orc.threshold_fit(X, U, V, None, u=u0, v=v0, lamda=lam, min_diff=1e-3, max_iter=15)
        assert got["u"] == pytest.approx(float(mdl.u), rel=1e-9) and got["v"] == pytest.approx(float(mdl.v), rel=1e-9), t
        np.testing.assert_allclose(np.array([r[:4] for r in got["rows"]]), frame(mdl.logs["updates"])[:, :4], rtol=1e-9)


def test_oracle_entry_scores_against_the_reference(ref):
    """task='prediction' with val / test sets: the columns the reference logs equal the oracle's entry scorer on the reference's
    own factors (non-zero cells of each densified set; WNMF's train set: every cell, after its in-place eps write)."""
    import oracle as orc
    from PyBMF.models import BinaryMFPenalty, WNMF
    rs = np.random.RandomState(14)
    for t in range(8):
        m, n, k, X, U0, V0 = random_problem(rs)
        part = rs.randint(0, 4, size=(m, n))
        sets = {}
        for nm, sel in (("train", part < 2), ("val", part == 2), ("test", part == 3)):
            r, c = np.nonzero(sel)
            sets[nm] = csr_matrix((X[r, c], (r, c)), shape=(m, n))
        if any(s.nnz == 0 or s.data.sum() == 0 for s in sets.values()):
            continue
        if t % 2 == 0:
            mdl = BinaryMFPenalty(k=k, U=U0.copy(), V=V0.copy(), W="mask", reg=0.5, reg_growth=1.3, init_method="custom", normalize_method=None,
                                  max_iter=2, tol=-1.0)
        else:
            mdl = WNMF(k=k, U=U0.copy(), V=V0.copy(), W="mask", init_method="custom", max_iter=2, tol=-1.0)
        quiet_fit(mdl, sets["train"].copy(), sets["val"].copy(), sets["test"].copy(), task="prediction")
        U, V = np.asarray(mdl.U), np.asarray(mdl.V)
        cols = [tuple(str(x) for x in c) for c in mdl.logs["updates"].columns]
        row = mdl.logs["updates"].values.tolist()[-1]
        for nm in ("train", "val", "test"):
            coo = sets[nm].tocoo()
            keep = coo.data != 0
            if nm == "train" and t % 2 == 1:   # WNMF: whole matrix
                want = orc.rmse_mae(sets[nm].toarray(), U @ V.T)
            else:
                want = orc.entry_scores(coo.row[keep], coo.col[keep], coo.data[keep], U, V)
            assert float(row[cols.index((nm, "0", "RMSE"))]) == pytest.approx(want[0], rel=1e-9), (t, nm)
            assert float(row[cols.index((nm, "0", "MAE"))]) == pytest.approx(want[1], rel=1e-9), (t, nm)
        if t % 2 == 0:
            bcols = [tuple(str(x) for x in c) for c in mdl.logs["boolean"].columns]
            brow = mdl.logs["boolean"].values.tolist()[-1]
            for nm in ("train", "val", "test"):
                coo = sets[nm].tocoo()
                keep = coo.data != 0
                want = orc.boolean_scores(*orc.entry_scores(coo.row[keep], coo.col[keep], coo.data[keep], U, V, 0.5, 0.5))
                got = [float(brow[bcols.index((nm, "0", mt))]) for mt in ("Recall", "Precision", "Accuracy", "F1")]
                np.testing.assert_allclose(got, want, rtol=1e-13)


def test_oracle_metrics_against_the_reference(ref):
    import oracle as orc
    import PyBMF.utils as R
    rs = np.random.RandomState(15)
    for t in range(40):
        m, n, k = int(rs.randint(1, 40)), int(rs.randint(1, 40)), int(rs.randint(1, 7))
        G = (rs.rand(m, n) < rs.choice([0.0, 0.3, 0.7, 1.0])).astype(np.int64)
        Ub, Vb = (rs.rand(m, k) < 0.3).astype(np.int64), (rs.rand(n, k) < 0.3).astype(np.int64)
        U, V = rs.rand(m, k), rs.rand(n, k)
        u, v = float(rs.uniform(0.2, 0.8)), float(rs.uniform(0.2, 0.8))
        (pr, _, er) = outcome(R.get_prediction_with_threshold, U=U, V=V, u=u, v=v)
        assert er is None
        np.testing.assert_array_equal(dense(pr), orc.boolean_product(U, V, u, v))
        P = orc.boolean_product(Ub, Vb)
        Gs, Ps = csr_matrix(G), csr_matrix(P)
        for ax in (None, 0, 1):
            want = [np.asarray(outcome(f, Gs, Ps, axis=ax)[0]).ravel() for f in (R.TP, R.FP, R.FN, R.TN)]
            got = orc.confusion_counts_axis(G, P, axis=ax)
            for a, b in zip(got, want):
                np.testing.assert_array_equal(np.asarray(a).ravel(), b)
            for w_fp in (0.5, 0.2):
                (cs, _, e1) = outcome(R.coverage_score, gt=Gs, pd=Ps, w_fp=w_fp, axis=ax)
                (we, _, e2) = outcome(R.weighted_error, gt=Gs, pd=Ps, w_fp=w_fp, axis=ax)
                assert e1 is None and e2 is None
                np.testing.assert_allclose(np.asarray(cs, float).ravel(), np.asarray(orc.coverage_score(G, P, w_fp=w_fp, axis=ax), float).ravel(), rtol=1e-14)
                np.testing.assert_allclose(np.asarray(we, float).ravel(), np.asarray(orc.weighted_error(G, P, w_fp=w_fp, axis=ax), float).ravel(), rtol=1e-14)
        (dl, _, e3) = outcome(R.description_length, gt=Gs, U=csr_matrix(Ub), V=csr_matrix(Vb), w_model=0.7, w_fp=2.0, w_fn=3.0)
        assert e3 is None and float(dl) == pytest.approx(orc.description_length(G, Ub, Vb, w_model=0.7, w_fp=2.0, w_fn=3.0), rel=1e-14)
        names = ["TP", "FP", "TN", "FN", "Recall", "Precision", "Accuracy", "F1", "RMSE", "MAE"]
        (vals, _, e4) = outcome(R.get_metrics, gt=Gs, pd=Ps, metrics=names)
        assert e4 is None
        tp, fp, fn, tn = orc.confusion_counts(G, P)
        rmse, mae = orc.rmse_mae(G, P)
        mine = [tp, fp, tn, fn, *orc.boolean_scores(tp, fp, fn, tn), rmse, mae]
        np.testing.assert_allclose(np.array(vals, float), np.array(mine, float), rtol=1e-14)
