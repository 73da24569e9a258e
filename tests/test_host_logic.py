"""Host-side logic of the drop-in layer (no GPU needed): parameter / config plumbing, stopping rule, log tables, score
formulas, the line search and the host generator -- against golden vectors from the reference and the CPU oracle."""
import contextlib
import hashlib
import io
import json
import os

import numpy as np
import pytest

import oracle as orc
from pybmf_amd.generators import SyntheticMatrixGenerator
from pybmf_amd.models import BinaryMFPenalty, BinaryMFThreshold, WNMF
from pybmf_amd.solvers import limit_step_size, line_search
from pybmf_amd.utils import header, record, scores_from_counts, binarize, to_sparse


@contextlib.contextmanager
def captured():
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        yield buf


def test_constructor_defaults_match_reference_signatures():
    with captured():
        p = BinaryMFPenalty(k=3)
        w = WNMF(k=3)
        t = BinaryMFThreshold(k=3, U=np.ones((4, 3)), V=np.ones((5, 3)))
    assert (p.W, p.beta_loss, p.solver, float(p.reg), float(p.reg_growth), float(p.max_reg), p.tol, p.min_diff, p.max_iter,
            p.init_method, p.normalize_method) == ("full", "frobenius", "mu", 2.0, 3.0, 1e10, 0.01, 0.0, 100, "custom", "balance")
    assert isinstance(p.reg, np.float64) and p.U is None and p.V is None
    assert (w.W, w.beta_loss, w.init_method, w.solver, w.tol, w.min_diff, w.max_iter) == ("mask", "frobenius", "normal", "mu", 0.0, 0.0, 30)
    assert (t.W, t.u, t.v, t.lamda, t.solver, t.min_diff, t.max_iter, t.init_method, t.normalize_method) == \
           ("mask", 0.5, 0.5, 100, "line-search", 1e-3, 100, "custom", None)


def test_set_params_and_config_semantics():
    with captured() as out:
        m = BinaryMFPenalty(k=4, seed=7, init_method="normal")
    text = out.getvalue()
    assert "[I] k            : 4" in text and "[I] seed         : 7" in text and "[I] verbose      : False" in text
    assert not hasattr(m, "task")                       # `task` exists only once given (fit(task=...))
    assert m.rng.rand() == np.random.RandomState(7).rand()
    assert (m.show_logs, m.save_model, m.show_result, m.scaling, m.pixels) == (True, True, True, 1.0, 2)
    with captured() as out:
        m.check_params(task="reconstruction", show_logs=False, max_iter=5, display=True, scaling=2.0)
    assert m.task == "reconstruction" and m.show_logs is False and m.save_model is True and m.max_iter == 5
    assert m.display is True and m.scaling == 2.0
    with captured():
        m.check_params(seed=None)                        # an existing seed / rng is kept
    assert m.seed == 7
    with captured(), pytest.raises(AssertionError):
        m.check_params(task="ranking")
    # matrices are echoed by shape, lists by length
    with captured() as out:
        m.set_params(U=np.zeros((3, 2)), us=[1, 2, 3])
    assert "(3, 2)" in out.getvalue() and ": 3" in out.getvalue()


def test_early_stop_rule_and_messages():
    with captured():
        m = BinaryMFPenalty(k=2, tol=0.5, min_diff=0.1, max_iter=3)
    cases = [(dict(error=1.0, diff=1.0, n_iter=1), True, ""),
             (dict(error=0.5, diff=1.0, n_iter=1), False, "Error <= tolerance"),
             (dict(error=1.0, diff=1.0, n_iter=3), True, ""),
             (dict(error=1.0, diff=1.0, n_iter=4), False, "Reach maximum iteration"),
             (dict(error=1.0, diff=0.05, n_iter=1), False, "Difference lower than threshold")]
    for kw, want, msg in cases:
        with captured() as out:
            assert m.early_stop(**kw) is want
        assert (msg in out.getvalue()) if msg else out.getvalue() == ""
        with captured():
            assert m.early_stop(**kw) == orc.should_continue({"tol": 0.5, "min_diff": 0.1, "max_iter": 3}, **kw)
    with captured() as out:
        assert m.early_stop(msg="forced") is False
    assert "[W] Stopped in advance: forced" in out.getvalue()


def test_header_and_record_schema():
    assert header(["time", "k", "score"], levels=3, depth=2) == [("", "time", ""), ("", "k", ""), ("", "score", "")]
    logs = {}
    cols = header(["iter", "error"], levels=3) + [("train", 0, "RMSE")]
    record(logs, "updates", cols, [0, 1.5, 0.25])
    record(logs, "updates", cols, [1, 1.0, 0.20])
    df = logs["updates"]
    assert list(df.columns) == [("", "", "time"), ("", "", "iter"), ("", "", "error"), ("train", 0, "RMSE")]
    assert df.shape == (2, 4) and isinstance(df.iloc[0, 0], str) and df.iloc[1, 2] == 1.0
    import re
    assert re.fullmatch(r"\d\d/\d\d/\d\d \d\d:\d\d:\d\d", df.iloc[0, 0])


def test_scores_from_counts_edge_rules(golden_dir):
    cases = json.load(open(os.path.join(golden_dir, "g5_metrics.json")))
    for c in cases:
        m = c["metrics"]
        got = scores_from_counts(m["TP"], m["FP"], m["FN"], m["TN"])
        assert tuple(float(x) for x in got) == (m["Recall"], m["Precision"], m["Accuracy"], m["F1"])
    assert scores_from_counts(0, 0, 5, 5)[:2] == (0.0, 0) and scores_from_counts(0, 0, 0, 9) == (0, 0, 1.0, 0)
    assert np.array_equal(binarize(np.array([0.5, 0.5000001, 0.4]), 0.5), [0, 1, 0])
    assert binarize(to_sparse(np.array([[0.6, 0.1]])), 0.5).format == "csr"


def test_line_search_matches_oracle_search():
    A = np.array([[3.0, 0.5], [0.5, 1.0]])
    b = np.array([1.0, -2.0])
    f = lambda x: float(0.5 * x @ A @ x - b @ x + 0.1 * np.sin(3 * x[0]))       # noqa: E731
    g = lambda x: A @ x - b + np.array([0.3 * np.cos(3 * x[0]), 0.0])             # noqa: E731
    for x0 in ([2.0, 2.0], [-1.0, 0.5], [0.1, -3.0]):
        xk = np.array(x0)
        pk = -g(xk)
        mine = line_search(f, g, xk, pk, maxiter=50)
        ref = orc.wolfe_search(f, g, xk, pk, maxiter=50)
        assert mine[0] == ref[0] and mine[1:3] == ref[1:3]
        assert mine[3] == ref[3] and mine[4] == ref[4] and np.array_equal(mine[5], ref[5])
    xk, pk = np.array([0.2, 0.3]), np.array([1.0, -1.0])
    for x_last, alpha in ([np.array([0.9, -0.4]), 0.7], [np.array([0.5, 0.0]), 0.3]):
        got = limit_step_size(np.array([0.0, 0.0]), np.array([0.8, 1.0]), x_last, xk, alpha, pk)
        want = orc.clip_step(np.array([0.0, 0.0]), np.array([0.8, 1.0]), x_last, xk, alpha, pk)
        assert np.array_equal(got[0], want[0]) and got[1] == want[1]


def test_host_generator_is_bit_identical_to_the_reference(golden_dir):
    g6 = json.load(open(os.path.join(golden_dir, "g6_generator.json")))
    for c in g6["generator"]:
        gen = SyntheticMatrixGenerator(m=c["m"], n=c["n"], k=c["k"], density=c["density"])
        gen.generate(seed=c["seed"])
        assert int(gen.X.sum()) == c["sum_clean"]
        assert hashlib.sha256(np.packbits(gen.X, axis=1, bitorder="little").tobytes()).hexdigest() == c["sha_clean"]
        gen.add_noise(noise=c["noise"], seed=c["noise_seed"])
        assert int(gen.X.sum()) == c["sum_noisy"]
        assert hashlib.sha256(np.packbits(gen.X, axis=1, bitorder="little").tobytes()).hexdigest() == c["sha_noisy"]


def test_init_and_balance_match_reference_draw_order(golden_dir):
    """init_UV / normalize_UV on the host, without touching the GPU (the bit upload is stubbed out)."""
    g6 = json.load(open(os.path.join(golden_dir, "g6_generator.json")))
    X = (np.random.RandomState(0).rand(60, 40) < 0.3).astype(np.float64)
    for method in ("normal", "uniform"):
        with captured():
            mdl = BinaryMFPenalty(k=4, W="full", init_method=method, normalize_method=None, seed=2024)
            mdl.check_params(task="reconstruction")
            mdl.m, mdl.n, mdl._x_mean = 60, 40, X.mean()
            mdl.init_UV()
        np.testing.assert_allclose(mdl.U.ravel()[:8], g6["init"][method]["U_head"], rtol=1e-12)
        np.testing.assert_allclose(mdl.V.ravel()[:8], g6["init"][method]["V_head"], rtol=1e-12)
    U0, V0 = orc.init_factors(X, 4, "normal", np.random.RandomState(5))
    with captured():
        mdl = BinaryMFPenalty(k=4, U=U0.copy(), V=V0.copy(), init_method="custom", normalize_method="balance")
        mdl.normalize_UV()
    Ub, Vb = orc.balance_factors(U0, V0)
    assert np.array_equal(mdl.U, Ub) and np.array_equal(mdl.V, Vb)


def test_container_helpers_follow_the_reference_dispatch():
    """utils/boolean_utils.py helpers: dense in -> dense out, sparse in -> sparse out, Boolean semantics."""
    import numpy as np
    from scipy.sparse import csr_matrix, issparse
    from pybmf_amd import utils as u
    rs = np.random.RandomState(0)
    A = (rs.rand(7, 5) < 0.4).astype(int)
    B = (rs.rand(7, 5) < 0.4).astype(int)
    C = (rs.rand(5, 6) < 0.4).astype(int)
    assert np.array_equal(u.multiply(A, B, boolean=True), A & B) and issparse(u.multiply(csr_matrix(A), B))
    assert np.array_equal(u.matmul(A, C, boolean=True), np.minimum(A @ C, 1))
    S = u.matmul(csr_matrix(A), csr_matrix(C), boolean=True)
    assert issparse(S) and np.array_equal(S.toarray(), np.minimum(A @ C, 1))
    assert np.array_equal(u.matmul(A.astype(float), C.astype(float)), A @ C)
    assert u.dot(A[0], B[0], boolean=True) == int((A[0] & B[0]).any()) and u.dot(A[0], B[0]) == A[0] @ B[0]
    assert np.array_equal(u.subtract(A.astype(float), 0.5), A - 0.5) and np.array_equal(u.power(A, 2), A.astype(float) ** 2)
    assert np.array_equal(u.add(A, B, boolean=True), ((A + B) > 0).astype(float))
    z = np.array([-800.0, -1.0, 0.0, 1.0, 800.0])
    np.testing.assert_allclose(u.sigmoid(z), [0.0, 1 / (1 + np.e), 0.5, np.e / (1 + np.e), 1.0], atol=1e-15)
    np.testing.assert_allclose(u.d_sigmoid(z), u.sigmoid(z) * (1 - u.sigmoid(z)))
    r, c, d = u.to_triplet(csr_matrix(A))
    assert np.array_equal(A[r, c], d) and len(d) == A.sum()


def test_normalize_uv_every_method_matches_the_reference():
    """ContinuousModel.normalize_UV (host code, runs before anything touches the GPU) against reference golden g12."""
    import os
    from pybmf_amd.models.ContinuousModel import ContinuousModel, unique_values_mapping
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "g12_normalize.npz"))

    class Probe(ContinuousModel):
        def __init__(self):
            pass

    for method in ("balance", "matrixwise-normalize", "columnwise-normalize", "matrixwise-mapping", "columnwise-mapping", None):
        m = Probe()
        m.k, m.U, m.V, m.normalize_method = 6, z["U0"].copy(), z["V0"].copy(), method
        m.normalize_UV()
        want_u, want_v = (z["U0"], z["V0"]) if method is None else (z[f"U_{method}"], z[f"V_{method}"])
        np.testing.assert_allclose(m.U, want_u, rtol=1e-15, atol=0)
        np.testing.assert_allclose(m.V, want_v, rtol=1e-15, atol=0)
    m = Probe()
    m.k, m.U, m.V, m.normalize_method = 6, z["U0"].copy(), z["V0"].copy(), "no-such-method"
    with pytest.raises(ValueError):
        m.normalize_UV()
    a = np.array([[0.5, 0.1], [0.1, 0.9]])
    np.testing.assert_array_equal(unique_values_mapping(a), np.array([[1 / 3, 0.0], [0.0, 2 / 3]]))


def test_boolean_ness_is_decided_from_values_not_dtype():
    """An integer matrix with values outside {0, 1} (ratings 1..5) is real-valued data: the Boolean-only models refuse it instead
    of silently binarising it (the reference casts X to float64 and fits the values)."""
    import scipy.sparse as sp
    from pybmf_amd.models.ContinuousModel import ContinuousModel
    ok = [np.eye(4, dtype=np.int64), np.eye(4, dtype=np.uint8), np.eye(4), np.eye(4, dtype=bool), sp.csr_matrix(np.eye(4, dtype=np.int32))]
    bad = [np.eye(4, dtype=np.int64) * 3, np.eye(4) * 0.5, sp.csr_matrix(np.eye(4, dtype=np.int64) * 5), np.full((2, 2), -1, dtype=np.int8)]
    for X in ok:
        ContinuousModel._check_boolean(X)
    for X in bad:
        with pytest.raises(NotImplementedError, match="Boolean"):
            ContinuousModel._check_boolean(X)


def test_movielens_loader_follows_the_reference_recipe(tmp_path):
    """bench.py's config-#5 loader (used when ~/.pybmf/data/movielens/ml-1m/ratings.dat exists): rows / columns are the sorted
    distinct user / item ids, a cell is rating > 0.5 (PyBMF/datasets/MovieLensData.py:36-40, 62-91)."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    p = tmp_path / "ratings.dat"
    p.write_text("5::10::3::978300760\n1::20::5::978302109\n5::20::1::978301968\n3::7::4::978300275\n")
    X = bench.load_movielens_1m(str(p))
    assert X.dtype == np.uint8 and X.tolist() == [[0, 0, 1], [1, 0, 0], [0, 1, 1]]      # users 1, 3, 5 x items 7, 10, 20
    assert bench.load_movielens_1m(str(tmp_path / "absent.dat")) is None
