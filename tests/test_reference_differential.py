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
