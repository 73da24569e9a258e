"""The drop-in classes, used the way the reference's notebooks use PyBMF, against golden vectors made by the reference
(tests/golden/make_golden.py) and against the CPU oracle."""
import contextlib
import io
import json
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

import oracle as orc  # noqa: E402

FIT = dict(task="reconstruction", show_logs=False, show_result=False, save_model=False)


@pytest.fixture(scope="module", autouse=True)
def need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")


@contextlib.contextmanager
def quiet():
    with contextlib.redirect_stdout(io.StringIO()):
        yield


def relf(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / np.linalg.norm(np.asarray(b))


def unpack(bits, shape):
    return np.unpackbits(np.asarray(bits, dtype=np.uint8), axis=1, bitorder="little")[:, : shape[1]]


def frame_values(df):
    return np.array([[float(v) for v in row[1:]] for row in df.values.tolist()])  # drop the 'time' column


def test_binarymfpenalty_fit_matches_reference(golden_dir):
    from pybmf_amd.models import BinaryMFPenalty
    z = np.load(os.path.join(golden_dir, "g1_penalty_c1.npz"))
    meta = json.load(open(os.path.join(golden_dir, "g1_penalty_c1.json")))
    X = unpack(z["X_bits"], z["shape"])
    with quiet():
        model = BinaryMFPenalty(k=8, U=None, V=None, W="full", reg=1, reg_growth=1.02, init_method="normal",
                                normalize_method="balance", max_iter=20, seed=2024)
        model.fit(X, **FIT)
    # log schema and values
    up, bo = model.logs["updates"], model.logs["boolean"]
    assert [tuple(str(x) for x in c) for c in up.columns][1:] == [tuple(c) for c in meta["updates"]["columns"]]
    assert [tuple(str(x) for x in c) for c in bo.columns][1:] == [tuple(c) for c in meta["boolean"]["columns"]]
    assert up.columns[0] == ("", "", "time") and isinstance(up.iloc[0, 0], str)
    np.testing.assert_allclose(frame_values(up), np.array(meta["updates"]["rows"]), rtol=1e-4)
    np.testing.assert_allclose(frame_values(bo), np.array(meta["boolean"]["rows"]), rtol=1e-15, atol=0)
    # factors, final reg, prediction
    assert model.U.dtype == np.float64 and model.U.shape == (1000, 8) and model.V.shape == (500, 8)
    assert relf(model.U, z["U_final"]) < 1e-4 and relf(model.V, z["V_final"]) < 1e-4
    assert float(model.reg) == pytest.approx(meta["final_reg"], rel=1e-15)
    assert list(model.counts[-1]) == meta["final_counts_TP_FP_FN_TN"]
    pd_ = model.X_pd
    assert pd_.shape == (1000, 500) and pd_.format == "csr"
    assert orc.confusion_counts(X.astype(np.int64), np.asarray(pd_.todense())) == tuple(meta["final_counts_TP_FP_FN_TN"])
    # attributes the reference leaves behind
    for attr in ("U", "V", "W", "X_train", "X_val", "X_test", "beta_loss", "display", "init_method", "k", "logs", "m", "max_iter",
                 "max_reg", "min_diff", "n", "name", "normalize_method", "pixels", "reg", "reg_growth", "rng", "save_model",
                 "scaling", "seed", "show_logs", "show_result", "solver", "task", "time", "tol", "verbose"):
        assert hasattr(model, attr), attr
    assert isinstance(model.time, str) and model.time.endswith("s")
    # evaluate() stays usable after fit and appends a row with the reference's column layout
    with quiet():
        model.evaluate(df_name="boolean")
        model.evaluate(df_name="extra", head_info={"iter": 99}, metrics=["RMSE", "MAE", "F1"])
    assert len(model.logs["boolean"]) == 23
    np.testing.assert_allclose(frame_values(model.logs["boolean"])[-1], meta["boolean"]["rows"][-1], rtol=1e-15)
    ex = model.logs["extra"]
    assert list(ex.columns)[1:] == [("", "", "iter"), ("train", 0, "RMSE"), ("train", 0, "MAE"), ("train", 0, "F1")]
    np.testing.assert_allclose(frame_values(ex)[0][1:3], meta["updates"]["rows"][-1][5:7], rtol=1e-4)


def test_binarymfpenalty_errors_like_reference():
    from pybmf_amd.models import BinaryMFPenalty
    X = (np.random.RandomState(0).rand(80, 60) < 0.3).astype(np.uint8)
    with quiet():
        with pytest.raises(AssertionError):
            BinaryMFPenalty(k=4, solver="als")
        with pytest.raises(AssertionError):
            BinaryMFPenalty(k=4, init_method="nndsvd")
        m = BinaryMFPenalty(k=4, init_method="normal", seed=1)
        with pytest.raises(TypeError, match="Missing training data"):
            m.fit(None, **FIT)
        with pytest.raises(AttributeError, match="task"):
            BinaryMFPenalty(k=4, init_method="normal", seed=1).fit(X, show_logs=False, show_result=False, save_model=False)
        with pytest.raises(AssertionError):
            BinaryMFPenalty(k=4, init_method="normal", seed=1).fit(X, task="ranking")
        # (values other than 0 / 1 are fitted as the reference fits them since round 5: tests/test_real_valued_gpu.py; bytes above 1 in a
        # uint8 array -- the "bits by construction" fast path -- are still refused rather than binarised)
        with pytest.raises(NotImplementedError, match="uint8"):
            BinaryMFPenalty(k=4, init_method="normal", seed=1).fit(X * 3, **FIT)
        with pytest.raises(NotImplementedError, match="k <= 128"):   # (64 < k <= 128 runs on the two-block engine: tests/test_wide_gpu.py)
            BinaryMFPenalty(k=129, init_method="normal", seed=1).fit(X, **FIT)


def test_fit_kwargs_override_and_custom_init(golden_dir):
    """fit(**kwargs) may override any parameter (BaseModel.check_params runs again); init_method='custom' resumes."""
    from pybmf_amd.models import BinaryMFPenalty
    z = np.load(os.path.join(golden_dir, "g1_penalty_c1.npz"))
    X = unpack(z["X_bits"], z["shape"])
    with quiet():
        model = BinaryMFPenalty(k=8, U=z["U0"], V=z["V0"], W="full", reg=1, reg_growth=1.02, init_method="custom",
                                normalize_method=None, max_iter=100)
        model.fit(X, max_iter=0, **FIT)  # exactly one update: n_iter = 1 > max_iter = 0
    assert model.max_iter == 0 and len(model.logs["updates"]) == 2
    assert relf(model.V, z["V1"]) < 2e-6 and relf(model.U, z["U1"]) < 2e-6


def test_module_level_updates(golden_dir):
    from pybmf_amd.models.BinaryMFPenalty import error, rec_error, reg_error, update_U, update_V
    z = np.load(os.path.join(golden_dir, "g2_penalty_steps.npz"))
    for case in range(3):
        X = z[f"c{case}_X"]
        U, V = z[f"c{case}_U"], z[f"c{case}_V"]
        for reg in (0.0, 1.0, 1e3):
            tag = f"c{case}_r{reg:g}"
            V1 = update_V(X=X, W=np.ones(X.shape), U=U, V=V, reg=np.float64(reg))
            assert relf(V1, z[tag + "_V1"]) < 5e-6
            U1 = update_U(X=X, W=None, U=U, V=z[tag + "_V1"], reg=reg)
            assert relf(U1, z[tag + "_U1"]) < 5e-6
            e = error(X_gt=X, X_pd=None, W=None, U=z[tag + "_U1"], V=z[tag + "_V1"], reg=reg)
            np.testing.assert_allclose(e, z[tag + "_err"], rtol=2e-5)
            assert rec_error(X, None, None, U=z[tag + "_U1"], V=z[tag + "_V1"]) == pytest.approx(z[tag + "_err"][1], rel=2e-5)
        assert reg_error(U) == pytest.approx(orc.reg_term(U), rel=1e-12)


def test_error_honours_an_explicit_prediction():
    """error(X_gt, X_pd, W, U, V, reg) with an X_pd that is NOT U V^T -- what the reference's PNLPF passes through the inherited
    loop (PyBMF/models/PNLPF.py:1,50-58): the reconstruction term is taken from the matrices given
    (PyBMF/models/BinaryMFPenalty.py:166-179), never silently from the factors."""
    from scipy.sparse import csr_matrix
    from pybmf_amd.models.BinaryMFPenalty import error, rec_error
    rs = np.random.RandomState(4)
    m, n, k = 333, 217, 5
    X = (rs.rand(m, n) < 0.2).astype(np.float64)
    U, V = rs.rand(m, k), rs.rand(n, k)
    P = 1.0 / (1.0 + np.exp(-10.0 * (U @ V.T - 0.5)))        # PNLPF's get_prediction_with_sigmoid
    W = (rs.rand(m, n) < 0.7) * rs.choice([0.5, 1.0, 2.0], size=(m, n))
    for Wa, Wn in ((None, np.ones((m, n))), (np.ones((m, n)), np.ones((m, n))), (W, W), (csr_matrix(W), W)):
        want_rec = 0.5 * float((Wn * (X - P) ** 2).sum())
        want_reg = 3.0 * (orc.reg_term(U) + orc.reg_term(V))
        for Pa in (P, np.asmatrix(P), csr_matrix(P)):
            assert rec_error(X, Pa, Wa) == pytest.approx(want_rec, rel=1e-12)
            np.testing.assert_allclose(error(csr_matrix(X), Pa, Wa, U, V, 3.0), (want_rec + want_reg, want_rec, want_reg), rtol=1e-12)
    # the factor form still exists (X_pd=None), and with X_pd = U V^T both agree
    assert rec_error(X, U @ V.T, None) == pytest.approx(rec_error(X, None, None, U=U, V=V), rel=2e-5)
    with pytest.raises(TypeError):
        rec_error(X, None, None)
    with pytest.raises(ValueError):
        rec_error(X, P[:-1], None)


def test_wnmf_on_a_lazy_row_source():
    """load_dataset keeps a lazy row source (shape + row slicing) as X_train; WNMF must treat it as Boolean like every other
    model and pack it, not np.asarray() it into a 0-d object array (round-2 advisor finding)."""
    from pybmf_amd.generators import PlantedBooleanOnDevice
    from pybmf_amd.models import WNMF
    gen = PlantedBooleanOnDevice(700, 300, 6, density=(0.2, 0.2), seed=5, noise=(0.05, 0.01), noise_seed=6, device="cuda:0")
    dense = gen[0:700].cpu().numpy()
    with quiet():
        a = WNMF(k=6, W="full", init_method="normal", max_iter=6, seed=7)
        a.fit(gen, **FIT)
        b = WNMF(k=6, W="full", init_method="normal", max_iter=6, seed=7)
        b.fit(dense, **FIT)
    assert a._boolean and b._boolean
    assert np.array_equal(a.U, b.U) and np.array_equal(a.V, b.V)


def test_wnmf_boolean_full_mask():
    from pybmf_amd.models import WNMF
    X, _, _, _ = orc.synthetic_boolean(400, 300, 6, (0.2, 0.2), seed=5)
    X = orc.flip_noise(X, (0.05, 0.01), seed=6)
    ref = orc.wnmf_fit(X.astype(np.float64), k=6, W=None, max_iter=15, init_method="normal", seed=7)
    with quiet():
        model = WNMF(k=6, W="full", init_method="normal", max_iter=15, seed=7)
        model.fit(X.astype(np.uint8), **FIT)
    got = frame_values(model.logs["updates"])
    assert list(model.logs["updates"].columns)[1:] == [("", "", "iter"), ("", "", "error"), ("train", 0, "RMSE"), ("train", 0, "MAE")]
    np.testing.assert_allclose(got, np.array(ref["updates"]), rtol=1e-4)
    assert relf(model.U, ref["U"]) < 1e-4 and relf(model.V, ref["V"]) < 1e-4
    assert np.allclose(np.asarray(model.X_pd.todense()), ref["U"] @ ref["V"].T, rtol=1e-3, atol=1e-5)
    # default W='mask' on a dense Boolean array = the pattern of its ones (csr drops the zeros): the masked kernels
    refm = orc.wnmf_fit(X.astype(np.float64), k=6, W=(X != 0).astype(np.float64), max_iter=4, init_method="normal", seed=7)
    with quiet():
        mm = WNMF(k=6, init_method="normal", max_iter=4, seed=7)
        mm.fit(X.astype(np.uint8), **FIT)
    np.testing.assert_allclose(frame_values(mm.logs["updates"]), np.array(refm["updates"]), rtol=1e-4)
    assert relf(mm.U, refm["U"]) < 1e-4
    # the Kullback-Leibler loss against the oracle (reference golden: tests/test_link_gpu.py)
    refk = orc.wnmf_kl_fit(X.astype(np.float64), k=6, max_iter=5, init_method="normal", seed=7)
    with quiet():
        mk = WNMF(k=6, W="full", beta_loss="kullback-leibler", init_method="normal", max_iter=5, seed=7)
        mk.fit(X.astype(np.uint8), **FIT)
    np.testing.assert_allclose(frame_values(mk.logs["updates"]), np.array(refk["updates"]), rtol=1e-4)
    assert relf(mk.U, refk["U"]) < 1e-4


def test_wnmf_real_loop_on_the_bf16_instruction(golden_dir):
    """RealMUEngine(bf16x3=True): X V and X^T U of the C-side loop on v_mfma_f32_32x32x16_bf16 with both operands split three ways
    (csrc/xf_f32.hip: bmf_xf_f32_tiled_bf3, the factor's frag3 order out of real_update_kernel) -- the trajectory of the exact-fp32 loop."""
    from pybmf_amd.engine import RealMatrix, RealMUEngine
    z = np.load(os.path.join(golden_dir, "g3_wnmf.npz"))
    X = z["X"].astype(np.float32)
    rs = np.random.RandomState(5)
    k = 7
    U0, V0 = rs.rand(X.shape[0], k) + 0.1, rs.rand(X.shape[1], k) + 0.1
    res = {}
    for bf3 in (False, True):
        eng = RealMUEngine(RealMatrix(X, "cuda:0"), k, with_mae=True, bf16x3=bf3)
        eng.load_factors(U0, V0)
        eng.device_loop(max_iter=14)
        eng.run(1, 13)
        log, stop = eng.read_log()
        assert (getattr(eng, "_UT3", None) is not None) == bf3
        res[bf3] = (log[:13, [1, 5, 6]], eng.factors())
    np.testing.assert_allclose(res[True][0], res[False][0], rtol=2e-6)
    assert relf(res[True][1][0], res[False][1][0]) < 2e-5 and relf(res[True][1][1], res[False][1][1]) < 2e-5


def test_wnmf_real_matches_reference(golden_dir):
    from pybmf_amd.models import WNMF
    z = np.load(os.path.join(golden_dir, "g3_wnmf.npz"))
    meta = json.load(open(os.path.join(golden_dir, "g3_wnmf.json")))
    p = meta["params"]
    with quiet():
        model = WNMF(k=p["k"], W="full", init_method=p["init_method"], max_iter=p["max_iter"], seed=p["seed"])
        model.fit(z["X"].copy(), **FIT)
    np.testing.assert_allclose(frame_values(model.logs["updates"]), np.array(meta["full"]["rows"]), rtol=1e-4)
    assert relf(model.U, z["full_U"]) < 1e-4 and relf(model.V, z["full_V"]) < 1e-4
    # default W='mask' is accepted when the stored pattern is the whole matrix (no exact zeros)
    Xd = z["X"].copy()
    Xd[Xd == 0] = 0.123
    ref = orc.wnmf_fit(Xd, k=p["k"], W=None, max_iter=5, init_method="normal", seed=3)
    with quiet():
        m2 = WNMF(k=p["k"], init_method="normal", max_iter=5, seed=3)
        m2.fit(Xd, **FIT)
    np.testing.assert_allclose(frame_values(m2.logs["updates"]), np.array(ref["updates"]), rtol=1e-4)


def test_binarymfthreshold_matches_reference(golden_dir):
    from pybmf_amd.models import BinaryMFThreshold
    z = np.load(os.path.join(golden_dir, "g4_threshold.npz"))
    meta = json.load(open(os.path.join(golden_dir, "g4_threshold.json")))
    X = unpack(z["X_bits"], z["shape"])
    for lam in (10, 100):
        g = meta[f"lam{lam}"]
        with quiet():
            model = BinaryMFThreshold(k=16, U=z["U"].copy(), V=z["V"].copy(), W="full", u=0.5, v=0.5, lamda=lam, min_diff=1e-3,
                                      max_iter=100)
            model.fit(X, **FIT)
        rows = frame_values(model.logs["updates"])
        ref = np.array(g["rows"]["rows"])
        assert list(model.logs["updates"].columns)[1:5] == [("", "", "iter"), ("", "", "u"), ("", "", "v"), ("", "", "F")]
        # the search is a chain of comparisons on F values: evaluated in fp64 on the device (csrc/thresh64.hip) it takes the
        # reference's decisions -- same number of outer iterations, same (u, v, F) on every row
        nc = min(len(rows), len(ref))
        bad = np.nonzero(np.abs(rows[:nc, 1:4] - ref[:nc, 1:4]).max(1) > 1e-6 * np.abs(ref[:nc, 1:4]).max(1) + 1e-9)[0]
        first = int(bad[0]) if len(bad) else nc
        assert len(rows) == len(ref), (lam, len(rows), len(ref), "first differing row", first, rows[max(first - 1, 0):first + 2, :4].tolist(),
                                       ref[max(first - 1, 0):first + 2, :4].tolist(), "trace" if model._trace is not None else "tile product",
                                       None if model._trace is None else model._trace["max_pairs"])
        np.testing.assert_allclose(rows[:, :4], ref[:, :4], rtol=1e-6, atol=1e-9)
        np.testing.assert_allclose(rows[:, 4:], ref[:, 4:], rtol=1e-12, atol=1e-15)      # Boolean scores at those thresholds: exact counts
        assert model.u == pytest.approx(g["u"], abs=1e-6) and model.v == pytest.approx(g["v"], abs=1e-6)
        assert model.F([0.4, 0.55]) == pytest.approx(z[f"F_grid_lam{lam}"][2, 3], rel=1e-4)


def test_binarymfthreshold_trace_and_tile_objectives_take_the_same_path(golden_dir, monkeypatch):
    """W = 'full': the batched trace-form objective (the default, csrc/thresh_trace.hip, one launch per Wolfe chain) and the tile
    product of rounds 2-3 (BMF_THRESH_TRACE=0, one launch per evaluation) give the same rows, (u, v, F) to 1e-12."""
    from pybmf_amd.models import BinaryMFThreshold
    z = np.load(os.path.join(golden_dir, "g4_threshold.npz"))
    X = unpack(z["X_bits"], z["shape"])
    rows = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("BMF_THRESH_TRACE", flag)
        with quiet():
            model = BinaryMFThreshold(k=16, U=z["U"].copy(), V=z["V"].copy(), W="full", u=0.3, v=0.6, lamda=10, min_diff=1e-3, max_iter=40)
            model.fit(X, **FIT)
        assert (model._trace is not None) == (flag == "1")
        rows[flag] = frame_values(model.logs["updates"])
    assert rows["1"].shape == rows["0"].shape
    np.testing.assert_allclose(rows["1"], rows["0"], rtol=1e-12, atol=1e-14)


def test_binarymfthreshold_trace_workspace_is_bounded(golden_dir, monkeypatch):
    """The trace-form workspace is sized within a memory budget (a quarter of the free device memory, at most 2 GiB): a tight budget
    means fewer (u, v) pairs per call -- same search path --, no budget at all the tile-product objective instead of an out-of-memory error."""
    import torch as th
    from pybmf_amd.models import BinaryMFThreshold
    z = np.load(os.path.join(golden_dir, "g4_threshold.npz"))
    X = unpack(z["X_bits"], z["shape"])
    kw = dict(k=16, W="full", u=0.3, v=0.6, lamda=10, min_diff=1e-3, max_iter=12)
    with quiet():
        full = BinaryMFThreshold(U=z["U"].copy(), V=z["V"].copy(), **kw)
        full.fit(X, **FIT)
    want, pairs_full = frame_values(full.logs["updates"]), full._trace["max_pairs"]
    real_info = th.cuda.mem_get_info
    for free_bytes, expect in ((4 * 1_500_000, "fewer pairs"), (4 * 200_000, "tile product")):
        # (enough for the list of the ones, 16 bytes each, in the second case only through the workspace check)
        monkeypatch.setattr(th.cuda, "mem_get_info", lambda dev=None, fb=free_bytes: (fb + 64 * int(X.sum()), real_info(dev)[1]))
        with quiet():
            mdl = BinaryMFThreshold(U=z["U"].copy(), V=z["V"].copy(), **kw)
            mdl.fit(X, **FIT)
        if expect == "fewer pairs":
            assert mdl._trace is not None and 1 <= mdl._trace["max_pairs"] < pairs_full
        else:
            assert mdl._trace is None
        got = frame_values(mdl.logs["updates"])
        assert got.shape == want.shape
        np.testing.assert_allclose(got, want, rtol=1e-10, atol=1e-12)
    monkeypatch.setattr(th.cuda, "mem_get_info", real_info)


@pytest.mark.parametrize("method", ["balance", "matrixwise-normalize", "columnwise-normalize", "matrixwise-mapping", "columnwise-mapping"])
def test_binarymfthreshold_normalize_methods(golden_dir, method):
    """Every normalize_method of the reference (ContinuousModel.py:87-148) ahead of the line search: reference golden g12."""
    from pybmf_amd.models import BinaryMFThreshold
    z = np.load(os.path.join(golden_dir, "g12_normalize.npz"))
    g = json.load(open(os.path.join(golden_dir, "g12_normalize.json")))[method]
    with quiet():
        model = BinaryMFThreshold(k=6, U=z["U0"].copy(), V=z["V0"].copy(), W="full", u=0.4, v=0.4, lamda=10, min_diff=1e-3, max_iter=8,
                                  normalize_method=method)
        model.fit(z["X"], **FIT)
    np.testing.assert_allclose(model.U, z[f"U_{method}"], rtol=1e-15)
    np.testing.assert_allclose(model.V, z[f"V_{method}"], rtol=1e-15)
    rows, ref = frame_values(model.logs["updates"]), np.array(g["rows"]["rows"])
    assert len(rows) == len(ref), (len(rows), len(ref))
    np.testing.assert_allclose(rows[:, :4], ref[:, :4], rtol=1e-6, atol=1e-9)
    assert model.u == pytest.approx(g["u"], abs=1e-6) and model.v == pytest.approx(g["v"], abs=1e-6)


def test_nan_in_the_factors_is_refused_like_the_reference():
    """utils/metrics.py:29-30: TypeError("NaN is found in prediction.")"""
    from pybmf_amd.models import BinaryMFPenalty, WNMF
    X, _, _, _ = orc.synthetic_boolean(200, 150, 4, (0.2, 0.2), seed=5)
    rs = np.random.RandomState(0)
    U0, V0 = rs.rand(200, 4), rs.rand(150, 4)
    U0[3, 1] = np.nan
    with quiet():
        with pytest.raises(TypeError, match="NaN is found in prediction"):
            BinaryMFPenalty(k=4, U=U0.copy(), V=V0.copy(), W="full", reg=1.0, reg_growth=1.1, init_method="custom", normalize_method=None,
                            max_iter=3).fit(X.astype(np.uint8), **FIT)
        with pytest.raises(TypeError, match="NaN is found in prediction"):
            WNMF(k=4, U=U0.copy(), V=V0.copy(), W="full", init_method="custom", max_iter=3).fit(X.astype(np.uint8), **FIT)


def test_default_config_saves_a_loadable_checkpoint(tmp_path, monkeypatch):
    """fit() with the reference's default config (save_model=True, show_logs / show_result on): the pickle holds factors,
    logs and parameters, no device handles (models/BaseModelTools.py:239-259)."""
    import pickle
    from pybmf_amd.models import BinaryMFPenalty
    monkeypatch.setenv("HOME", str(tmp_path))
    X, _, _, _ = orc.synthetic_boolean(300, 200, 5, (0.2, 0.2), seed=5)
    with quiet():
        mdl = BinaryMFPenalty(k=5, W="full", reg=1.0, reg_growth=1.1, init_method="normal", max_iter=4, seed=3)
        mdl.fit(X.astype(np.uint8), task="reconstruction")
    assert os.path.exists(mdl.pickle_path)
    with open(mdl.pickle_path, "rb") as fh:
        data = pickle.load(fh)
    assert np.array_equal(data["U"], mdl.U) and np.array_equal(data["V"], mdl.V)
    assert set(data["logs"]) == {"updates", "boolean"} and data["k"] == 5 and float(data["reg"]) == float(mdl.reg)
    assert isinstance(mdl.time, str) and mdl.name.endswith("BinaryMFPenalty")


def test_fit_accepts_every_container_of_a_boolean_matrix():
    """ndarray of any dtype, np.matrix, every scipy.sparse format, torch tensors on the host or on the device (the reference
    takes ndarray / scipy.sparse through to_sparse, models/BaseModel.py:146): the same fit from each."""
    from scipy.sparse import coo_matrix, csc_matrix, csr_matrix, dok_matrix, lil_matrix
    from pybmf_amd.models import BinaryMFPenalty
    rs = np.random.RandomState(0)
    X = rs.rand(130, 90) < 0.3
    want = None
    for data in (X, X.astype(np.int8), X.astype(np.float32), np.asmatrix(X.astype(np.float64)), csr_matrix(X.astype(np.float64)),
                 csc_matrix(X.astype(np.float64)), coo_matrix(X.astype(np.int64)), lil_matrix(X.astype(np.float64)), dok_matrix(X.astype(np.float32)),
                 torch.from_numpy(X.astype(np.uint8)), torch.from_numpy(X.astype(np.float32)).cuda()):
        with quiet():
            p = BinaryMFPenalty(k=5, W="full", reg=1.0, reg_growth=1.1, init_method="normal", max_iter=4, seed=2)
            p.fit(data, **FIT)
        got = frame_values(p.logs["updates"])
        want = got if want is None else want
        np.testing.assert_allclose(got, want, rtol=1e-12, err_msg=type(data).__name__)


def test_other_models_and_options_accept_the_same_containers():
    """WNMF (real-valued and Boolean), BinaryMFThreshold and PNLPF through sparse formats, weight matrices in any sparse format,
    factors given as sparse matrices, NumPy integer parameters, extra data sets given as coo / ndarray."""
    from scipy.sparse import coo_matrix, csc_matrix, csr_matrix, lil_matrix
    from pybmf_amd.models import BinaryMFPenalty, BinaryMFThreshold, PNLPF, WNMF
    rs = np.random.RandomState(0)
    X = (rs.rand(130, 90) < 0.3).astype(np.float64)
    R = (rs.rand(130, 6) @ rs.rand(6, 90)) / 6

    def last_error(model, *data):
        with quiet():
            model.fit(*data, **FIT)
        return float(model.logs["updates"].values[-1][2])

    wn = lambda **kw: WNMF(k=5, init_method="normal", max_iter=3, seed=2, **kw)  # noqa: E731
    base = last_error(wn(W="full"), R)
    assert last_error(wn(W="full"), csc_matrix(R)) == pytest.approx(base, rel=1e-12)
    assert last_error(wn(W="full"), coo_matrix(R)) == pytest.approx(base, rel=1e-12)
    assert last_error(wn(), R) == pytest.approx(base, rel=1e-12)          # default W='mask' on a matrix without zeros
    Wm = (rs.rand(130, 90) < 0.5) * 2.0
    assert last_error(wn(W=coo_matrix(Wm)), X) == pytest.approx(last_error(wn(W=lil_matrix(Wm)), X), rel=1e-12)
    assert last_error(wn(W=Wm), X) == pytest.approx(last_error(wn(W=csr_matrix(Wm)), X), rel=1e-12)
    U, V = rs.rand(130, 5), rs.rand(90, 5)
    thr = lambda U_, V_, data, **kw: (lambda t: (quiet_fit(t, data), (t.u, t.v))[1])(  # noqa: E731
        BinaryMFThreshold(k=5, U=U_, V=V_, lamda=10, max_iter=3, **kw))
    assert thr(U.copy(), V.copy(), coo_matrix(X)) == pytest.approx(thr(U.copy(), V.copy(), csr_matrix(X)), rel=1e-12)
    assert thr(lil_matrix(U), csr_matrix(V), X, W="full") == pytest.approx(thr(U.copy(), V.copy(), X, W="full"), rel=1e-12)
    pen = lambda **kw: BinaryMFPenalty(W="full", reg=1, init_method="normal", **kw)  # noqa: E731
    assert last_error(pen(k=np.int64(5), max_iter=np.int32(3), seed=np.int64(2)), X) == pytest.approx(last_error(pen(k=5, max_iter=3, seed=2), X), rel=1e-12)
    assert last_error(pen(k=5, max_iter=3, seed=2), X, coo_matrix(X), X.astype(np.int8)) == pytest.approx(last_error(pen(k=5, max_iter=3, seed=2), X), rel=1e-12)
    pn = lambda: PNLPF(k=5, W="full", reg=1, init_method="normal", max_iter=3, seed=2)  # noqa: E731
    assert last_error(pn(), coo_matrix(X)) == pytest.approx(last_error(pn(), X), rel=1e-12)
    for model in (pen(k=5, max_iter=0, seed=2), WNMF(k=5, W="full", init_method="normal", max_iter=0, seed=2)):
        with quiet():
            model.fit(X, **FIT)
        assert len(model.logs["updates"]) == 2   # the loop runs once before `n_iter > max_iter` is looked at, as in the reference


def quiet_fit(model, *data):
    with quiet():
        model.fit(*data, **FIT)
    return model


def test_wnmf_takes_integer_ratings_as_real_values():
    """an int64 matrix of ratings (values 0..5, as the reference's MovieLens loader yields) is fitted on its VALUES,
    like the reference does after its cast to float64 -- not on the binarised pattern."""
    from pybmf_amd.models import WNMF, BinaryMFPenalty
    rs = np.random.RandomState(4)
    m, n, k = 90, 70, 5
    R = (rs.rand(m, n) < 0.4) * rs.randint(1, 6, size=(m, n))
    R = R.astype(np.int64)
    ref = orc.wnmf_fit(R.astype(np.float64), k=k, W=np.ones((m, n)), max_iter=6, init_method="normal", seed=11)
    with quiet():
        w = WNMF(k=k, W="full", init_method="normal", max_iter=6, seed=11)
        w.fit(R, **FIT)
    assert w._boolean is False
    assert relf(w.U, ref["U"]) < 1e-4 and relf(w.V, ref["V"]) < 1e-4
    rows = np.array([[float(v) for v in r[1:]] for r in w.logs["updates"].values.tolist()])
    assert rows[-1, 1] == pytest.approx(ref["updates"][-1][1], rel=1e-4)
    # BinaryMFPenalty takes the VALUES too (round 5; the reference casts to float64 and runs): against the oracle's restatement
    refp = orc.penalty_fit(R.astype(np.float64), k=k, reg=1.0, reg_growth=1.1, init_method="normal", normalize_method="balance", max_iter=4, seed=1)
    with quiet():
        p = BinaryMFPenalty(k=k, reg=1.0, reg_growth=1.1, init_method="normal", normalize_method="balance", max_iter=4, seed=1)
        p.fit(R, **FIT)
    assert p._boolean is False and relf(p.U, refp["U"]) < 1e-4 and relf(p.V, refp["V"]) < 1e-4
    # the same values as a 0/1 pattern ARE Boolean, whatever the dtype
    with quiet():
        w2 = WNMF(k=k, W="full", init_method="normal", max_iter=2, seed=11)
        w2.fit((R > 0).astype(np.int64), **FIT)
    assert w2._boolean is True


def test_sparse_obs_merges_duplicate_coordinates():
    """a coo mask with repeated (row, col) entries must not corrupt the observation lists."""
    from pybmf_amd.engine import SparseObs
    rows, cols = np.array([0, 1, 1, 2, 1]), np.array([0, 2, 2, 1, 0])
    vals, wgts = np.array([1.0, 1.0, 1.0, 0.0, 1.0]), np.array([1.0, 0.5, 0.25, 2.0, 1.0])
    obs = SparseObs(rows, cols, vals, wgts, (3, 4), "cuda:0")
    assert obs.nnz == 4
    csr = obs.csr
    got = sorted(zip(np.repeat(np.arange(3), np.diff(csr["ptr"].cpu().numpy())).tolist(), csr["idx"].cpu().numpy().tolist(),
                     csr["val"].cpu().numpy().tolist(), csr["wgt"].cpu().numpy().tolist()))
    assert got == [(0, 0, 1.0, 1.0), (1, 0, 1.0, 1.0), (1, 2, 2.0, 0.75), (2, 1, 0.0, 2.0)]


@pytest.mark.gpu
def test_uint8_matrix_with_other_values_is_refused_by_the_device_side_check():
    """uint8 inputs are uploaded as they are and checked for 'values are 0 / 1' by the packer (no host pass over the array):
    the Boolean-only models must still refuse them, host arrays and tensors alike."""
    import torch
    from pybmf_amd.models import BinaryMFPenalty
    rs = np.random.RandomState(3)
    X = (rs.rand(300, 200) < 0.2).astype(np.uint8)
    fit = dict(task="reconstruction", show_logs=False, show_result=False, save_model=False)
    BinaryMFPenalty(k=4, W="full", reg=1.0, init_method="normal", max_iter=2, seed=1).fit(X, **fit)   # fine
    X[17, 5] = 3
    for bad in (X, torch.from_numpy(X)):
        with pytest.raises(NotImplementedError, match="Boolean"):
            BinaryMFPenalty(k=4, W="full", reg=1.0, init_method="normal", max_iter=2, seed=1).fit(bad, **fit)


def test_binarymfthreshold_at_rank_64_packs_factor_bit_63():
    """The cover count of the thresholding model packs the thresholded factors on the device, one k-bit word per row: at k = 64 the weight
    of column 63 is 2^63, which does not fit a signed 64-bit Python-to-torch conversion (found by the property test; regression)."""
    from pybmf_amd.models import BinaryMFThreshold
    rs = np.random.RandomState(64)
    m, n, k = 90, 70, 64
    X = (rs.rand(m, n) < 0.3).astype(np.uint8)
    U, V = rs.rand(m, k) * 0.3, rs.rand(n, k) * 0.3
    U[:, 63] = rs.rand(m) * 1.5      # column 63 is the one that decides cells
    V[:, 63] = rs.rand(n) * 1.5
    with quiet():
        t = BinaryMFThreshold(k=k, U=U.copy(), V=V.copy(), u=0.5, v=0.5, lamda=10, max_iter=2, init_method="custom", normalize_method=None)
        t.fit(X.copy(), **FIT)
    rows = frame_values(t.logs["updates"])     # iter, u, v, F, then Recall / Precision / Accuracy / F1 of the train set
    assert len(rows) >= 2
    for row in rows:
        u, v = row[1], row[2]
        pd = orc.boolean_product((t.U > u).astype(np.int64), (t.V > v).astype(np.int64))
        tp, fp, fn, tn = orc.confusion_counts(X.astype(np.int64), pd)
        want = [tp / max(tp + fn, 1), tp / max(tp + fp, 1), (tp + tn) / (m * n)]
        np.testing.assert_allclose(row[4:7], want, rtol=1e-12)
