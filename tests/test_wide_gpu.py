"""Rank 64 < k <= 128 (two 64-column blocks per factor, pybmf_amd/wide.py + csrc/wide.hip) against the fp64 oracle: the reference
has no rank limit (PyBMF/models/BinaryMFPenalty.py:32), its updates (:136-163; WNMF.py:96-109) and scores are restated in
oracle/pybmf_oracle.py.  Gates as everywhere: factors 1e-4 norm-wise, scalars 1e-4 relative, Boolean counts exact."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

import oracle as orc  # noqa: E402


@pytest.fixture(scope="module")
def env():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from pybmf_amd import _lib as L
    from pybmf_amd import engine as E
    from pybmf_amd import wide as W
    return L, E, W


def relf(a, b):
    return np.linalg.norm(a - b) / np.linalg.norm(b)


def problem(m, n, k, seed):
    X, _, _, _ = orc.synthetic_boolean(m, n, min(k, 12), (0.25, 0.25), seed=seed)
    X = orc.flip_noise(X, (0.05, 0.02), seed=seed + 1).astype(np.float64)
    rs = np.random.RandomState(seed + 2)
    avg = np.sqrt(X.mean() / k)
    U0, V0 = np.abs(avg * rs.standard_normal((m, k))), np.abs(avg * rs.standard_normal((n, k)))
    U0[3, :] = 0.0   # a zero row and a zero column: the eps paths
    V0[:, k - 1] = 0.0
    return X, orc.zeros_to_eps(U0), orc.zeros_to_eps(V0)


def oracle_scalars(X, U, V, reg, penalty):
    if penalty:
        err, rec, rg = orc.penalty_errors(X, None, U, V, reg)
    else:
        rec = orc.rec_term(X, U @ V.T)
        err, rg = rec, 0.0
    rmse, mae = orc.rmse_mae(X, U @ V.T)
    cnt = orc.confusion_counts(X, orc.boolean_product(U, V, 0.5, 0.5))
    return err, rec, rg, rmse, mae, tuple(int(c) for c in cnt)


@pytest.mark.parametrize("m,n,k,penalty", [(700, 450, 100, True), (513, 300, 128, True), (640, 200, 65, False), (400, 600, 97, False)])
def test_wide_trajectory_matches_the_oracle(env, m, n, k, penalty):
    L, E, W = env
    X, U, V = problem(m, n, k, seed=m + k)
    eng = W.WideMUEngine(E.BitMatrix(X.astype(np.uint8), "cuda:0"), k, L.MODE_PENALTY if penalty else L.MODE_WNMF, with_mae=True)
    eng.load_factors(U, V)
    eng.prepare()
    reg = 1.0

    def check(it):
        got = eng.scalars(reg)
        want = oracle_scalars(X, U, V, reg, penalty)
        for g, w, name in zip(got[:5], want[:5], ("error", "rec_error", "reg_error", "RMSE", "MAE")):
            assert g == pytest.approx(w, rel=1e-4, abs=1e-9), (it, name, g, w)
        assert got[5] == want[5], (it, got[5], want[5])

    check(0)
    for it in range(1, 6):
        eng.update(reg)
        if penalty:
            V = orc.penalty_update_V_reassoc(X, U, V, reg)
            U = orc.penalty_update_U_reassoc(X, U, V, reg)
        else:
            U, V = orc.wnmf_update(X, None, U, V)
        Ug, Vg = eng.factors()
        assert Ug.shape == (m, k) and Vg.shape == (n, k)
        assert relf(Ug, U) < 1e-5 and relf(Vg, V) < 1e-5, (it, relf(Ug, U), relf(Vg, V))
        check(it)
        reg *= 1.3
    s_abs, s_sq = eng.residual_sums()
    R = X - U @ V.T
    assert s_abs == pytest.approx(np.abs(R).sum(), rel=1e-4) and s_sq == pytest.approx((R * R).sum(), rel=1e-4)


def test_wide_engine_refuses_other_ranks(env):
    L, E, W = env
    B = E.BitMatrix(np.eye(64, dtype=np.uint8), "cuda:0")
    for k in (64, 129):
        with pytest.raises(NotImplementedError):
            W.WideMUEngine(B, k)


# ---- the drop-in classes at a rank above 64 -------------------------------------------------------------------------------------
import contextlib  # noqa: E402
import io  # noqa: E402

FIT = dict(task="reconstruction", show_logs=False, show_result=False, save_model=False)


def frame_values(df):
    return np.array([[float(v) for v in row[1:]] for row in df.values.tolist()])  # drop the 'time' column


def test_classes_fit_at_a_rank_above_64(env):
    """BinaryMFPenalty and WNMF with 64 < k <= 128: whole fit() against the oracle's fit (log tables, factors, X_pd, evaluate)."""
    from pybmf_amd.models import BinaryMFPenalty, WNMF
    X, _, _, _ = orc.synthetic_boolean(500, 380, 10, (0.25, 0.25), seed=21)
    X = orc.flip_noise(X, (0.05, 0.02), seed=22)
    k = 80
    ref = orc.penalty_fit(X, k=k, reg=1.0, reg_growth=1.3, init_method="normal", normalize_method="balance", max_iter=6, seed=9, literal=False)
    with contextlib.redirect_stdout(io.StringIO()):
        model = BinaryMFPenalty(k=k, W="full", reg=1.0, reg_growth=1.3, init_method="normal", normalize_method="balance", max_iter=6, seed=9)
        model.fit(X.astype(np.uint8), **FIT)
    np.testing.assert_allclose(frame_values(model.logs["updates"]), np.array(ref["updates"]), rtol=1e-4)
    np.testing.assert_allclose(frame_values(model.logs["boolean"]), np.array(ref["boolean"]), rtol=1e-12)
    assert relf(model.U, ref["U"]) < 1e-4 and relf(model.V, ref["V"]) < 1e-4
    assert model.U.shape == (500, k) and model.n_iter == ref["n_iter"]
    assert [tuple(c) for c in model.counts] == [tuple(c) for c in ref["counts"]]
    assert np.array_equal(np.asarray(model.X_pd.todense()), orc.boolean_product(ref["U"], ref["V"], 0.5, 0.5))
    rmse, mae = orc.rmse_mae(X, ref["U"] @ ref["V"].T)
    got = model._score_train(["RMSE", "MAE", "TP", "FP"])
    assert got[0] == pytest.approx(rmse, rel=1e-4) and got[1] == pytest.approx(mae, rel=1e-4)
    assert tuple(got[2:]) == tuple(ref["counts"][-1][:2])

    refw = orc.wnmf_fit(X.astype(np.float64), k=70, W=None, max_iter=5, init_method="normal", seed=7)
    with contextlib.redirect_stdout(io.StringIO()):
        w = WNMF(k=70, W="full", init_method="normal", max_iter=5, seed=7)
        w.fit(X.astype(np.uint8), **FIT)
    np.testing.assert_allclose(frame_values(w.logs["updates"]), np.array(refw["updates"]), rtol=1e-4)
    assert relf(w.U, refw["U"]) < 1e-4 and relf(w.V, refw["V"]) < 1e-4
    assert np.allclose(np.asarray(w.X_pd.todense()), refw["U"] @ refw["V"].T, rtol=1e-3, atol=1e-5)

    # what stays refused says so: a rank above 128 (masks and extra data sets at 64 < k <= 128: tests/test_wide_extras_gpu.py)
    with contextlib.redirect_stdout(io.StringIO()):
        with pytest.raises(NotImplementedError, match="k <= 128"):
            BinaryMFPenalty(k=130, W="full", init_method="normal", max_iter=2, seed=1).fit(X.astype(np.uint8), **FIT)


def test_wide_loop_without_host_round_trips_takes_the_same_path(env, monkeypatch):
    """Round 5: at 64 < k <= 128 the loop enqueues iteration t + 1 before it reads the scalars of t (WideMUEngine.iterate / row /
    previous_factors, the protocol of the masked and link engines) -- same rows, same stopping iteration, same factors as the stepwise
    loop with a read-back per iteration, incl. a run that its stopping rule ends early."""
    from pybmf_amd.models import BinaryMFPenalty
    from pybmf_amd.wide import WideMUEngine
    X, _, _, _ = orc.synthetic_boolean(400, 300, 9, (0.25, 0.25), seed=31)
    X = orc.flip_noise(X, (0.05, 0.02), seed=32).astype(np.uint8)
    base = dict(k=100, W="full", reg=1.0, reg_growth=1.3, init_method="normal", normalize_method="balance", seed=5)

    def run(pipelined, **kw):
        monkeypatch.setattr(WideMUEngine, "can_pipeline", lambda self, p=pipelined: p)
        with contextlib.redirect_stdout(io.StringIO()):
            mdl = BinaryMFPenalty(**base, **kw)
            mdl.fit(X, **FIT)
        return frame_values(mdl.logs["updates"]), frame_values(mdl.logs["boolean"]), mdl.U.copy(), mdl.V.copy(), mdl.n_iter, float(mdl.reg)
    probe = run(False, max_iter=12, tol=0.0)
    reg_err = probe[0][:, 4]
    t_stop = int(np.argmin(reg_err[1:9])) + 1               # an iteration the rule "reg_error <= tol" can fire on, well before max_iter
    tol_early = float(reg_err[t_stop]) * (1 + 1e-9)
    for kw in (dict(max_iter=6, tol=0.0), dict(max_iter=12, tol=tol_early)):
        a, b = run(True, **kw), run(False, **kw)
        assert a[4] == b[4] and a[5] == b[5]
        if kw["tol"] > 0:
            assert a[4] <= t_stop < 12                       # stopped early, by the rule
        assert np.array_equal(a[0][:, [0, 3]], b[0][:, [0, 3]])           # iter, reg
        np.testing.assert_allclose(a[0], b[0], rtol=1e-12)                  # (the MAE / RMSE sums are fp64 atomics)
        assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3])
