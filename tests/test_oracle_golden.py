"""The CPU oracle against the golden vectors produced by the reference (tests/golden/make_golden.py).

This pins the oracle: every number here came out of PyBMF @ 2024_10_08 itself.
"""
import hashlib
import json
import os

import numpy as np
import pytest

import oracle as orc

RT = dict(rtol=1e-12, atol=0)


def unpack(bits, shape):
    return np.unpackbits(np.asarray(bits, dtype=np.uint8), axis=1, bitorder="little")[:, :shape[1]]


@pytest.fixture(scope="module")
def g1(golden_dir):
    z = np.load(os.path.join(golden_dir, "g1_penalty_c1.npz"))
    meta = json.load(open(os.path.join(golden_dir, "g1_penalty_c1.json")))
    X = unpack(z["X_bits"], z["shape"])
    return z, meta, X


def test_generator_bit_exact(golden_dir):
    g6 = json.load(open(os.path.join(golden_dir, "g6_generator.json")))
    for c in g6["generator"]:
        X, _, _, rng = orc.synthetic_boolean(c["m"], c["n"], c["k"], c["density"], c["seed"])
        assert int(X.sum()) == c["sum_clean"]
        sha = hashlib.sha256(np.packbits(X.astype(np.uint8), axis=1, bitorder="little").tobytes()).hexdigest()
        assert sha == c["sha_clean"]
        Xn = orc.flip_noise(X, c["noise"], seed=c["noise_seed"])
        assert int(Xn.sum()) == c["sum_noisy"]
        sha = hashlib.sha256(np.packbits(Xn.astype(np.uint8), axis=1, bitorder="little").tobytes()).hexdigest()
        assert sha == c["sha_noisy"]


def test_init_draw_order(golden_dir):
    g6 = json.load(open(os.path.join(golden_dir, "g6_generator.json")))
    X = (np.random.RandomState(0).rand(60, 40) < 0.3).astype(np.float64)
    for method in ("normal", "uniform"):
        U, V = orc.init_factors(X, 4, method, np.random.RandomState(2024))
        np.testing.assert_allclose(U.ravel()[:8], g6["init"][method]["U_head"], **RT)
        np.testing.assert_allclose(V.ravel()[:8], g6["init"][method]["V_head"], **RT)


def test_c1_input_matches_generator(g1):
    z, meta, X = g1
    g = meta["generator"]
    Xc, _, _, _ = orc.synthetic_boolean(g["m"], g["n"], g["k"], g["density"], g["seed"])
    Xn = orc.flip_noise(Xc, g["noise"], seed=g["noise_seed"])
    assert int(Xc.sum()) == meta["sum_clean"] == 135135
    assert int(Xn.sum()) == meta["sum_noisy"] == 132199
    assert np.array_equal(Xn, X)


def test_c1_single_step(g1):
    z, meta, X = g1
    X = X.astype(np.float64)
    V1 = orc.penalty_update_V(X, None, z["U0"], z["V0"], np.float64(1.0))
    U1 = orc.penalty_update_U(X, None, z["U0"], V1, np.float64(1.0))
    np.testing.assert_allclose(V1, z["V1"], **RT)
    np.testing.assert_allclose(U1, z["U1"], **RT)
    # the re-associated form the HIP path uses is the same update to round-off
    V1r = orc.penalty_update_V_reassoc(X, z["U0"], z["V0"], 1.0)
    U1r = orc.penalty_update_U_reassoc(X, z["U0"], V1r, 1.0)
    assert np.linalg.norm(V1r - z["V1"]) / np.linalg.norm(z["V1"]) < 1e-13
    assert np.linalg.norm(U1r - z["U1"]) / np.linalg.norm(z["U1"]) < 1e-13


@pytest.mark.parametrize("literal", [True, False])
def test_c1_trajectory(g1, literal):
    z, meta, X = g1
    p = meta["params"]
    res = orc.penalty_fit(X, k=p["k"], reg=p["reg"], reg_growth=p["reg_growth"], init_method=p["init_method"],
                          normalize_method=p["normalize_method"], max_iter=p["max_iter"], seed=p["seed"], literal=literal)
    tol = 1e-11 if literal else 1e-9
    np.testing.assert_allclose(orc.zeros_to_eps(res["U0"]), z["U0"], **RT)
    np.testing.assert_allclose(orc.zeros_to_eps(res["V0"]), z["V0"], **RT)
    np.testing.assert_allclose(res["U"], z["U_final"], rtol=tol, atol=1e-300)
    np.testing.assert_allclose(res["V"], z["V_final"], rtol=tol, atol=1e-300)
    assert res["n_iter"] == 21 and len(res["updates"]) == 22
    np.testing.assert_allclose(np.array(res["updates"]), np.array(meta["updates"]["rows"]), rtol=tol)
    np.testing.assert_allclose(np.array(res["boolean"]), np.array(meta["boolean"]["rows"]), rtol=1e-15, atol=0)
    assert list(res["counts"][-1]) == meta["final_counts_TP_FP_FN_TN"] == [110012, 6743, 22187, 361058]
    assert res["reg"] == pytest.approx(meta["final_reg"], rel=1e-15)


def test_penalty_steps_edge_cases(golden_dir):
    z = np.load(os.path.join(golden_dir, "g2_penalty_steps.npz"))
    for case in range(3):
        X = z[f"c{case}_X"].astype(np.float64)
        U, V = z[f"c{case}_U"], z[f"c{case}_V"]
        for reg in (0.0, 1.0, 1e3):
            tag = f"c{case}_r{reg:g}"
            V1 = orc.penalty_update_V(X, np.ones(X.shape), U, V, np.float64(reg))
            U1 = orc.penalty_update_U(X, np.ones(X.shape), U, V1, np.float64(reg))
            np.testing.assert_allclose(V1, z[tag + "_V1"], **RT)
            np.testing.assert_allclose(U1, z[tag + "_U1"], **RT)
            np.testing.assert_allclose(orc.penalty_errors(X, None, U1, V1, reg), z[tag + "_err"], rtol=1e-12)
    # case 0 has an all-zero column in U: with reg=0 the V update hits denom==0 -> eps and V==0 -> eps
    assert (z["c0_r0_V1"][:, 2] == orc.EPS).all()


def test_wnmf(golden_dir):
    z = np.load(os.path.join(golden_dir, "g3_wnmf.npz"))
    meta = json.load(open(os.path.join(golden_dir, "g3_wnmf.json")))
    p = meta["params"]
    for wname in ("full", "mask"):
        X = z["X"].copy()
        W = None if wname == "full" else (X != 0).astype(np.float64)
        res = orc.wnmf_fit(X, k=p["k"], W=W, max_iter=p["max_iter"], init_method=p["init_method"], seed=p["seed"])
        np.testing.assert_allclose(res["U"], z[f"{wname}_U"], rtol=1e-10, atol=1e-300)
        np.testing.assert_allclose(res["V"], z[f"{wname}_V"], rtol=1e-10, atol=1e-300)
        np.testing.assert_allclose(np.array(res["updates"]), np.array(meta[wname]["rows"]), rtol=1e-10)
        # in-place eps quirk (WNMF.py:136-139): the training matrix has no exact zeros afterwards
        assert (res["X"] != 0).all() and np.array_equal(res["X"], z[f"{wname}_X_after"])


def test_threshold_objective_and_search(golden_dir):
    z = np.load(os.path.join(golden_dir, "g4_threshold.npz"))
    meta = json.load(open(os.path.join(golden_dir, "g4_threshold.json")))
    X = unpack(z["X_bits"], z["shape"]).astype(np.float64)
    U, V = z["U"], z["V"]
    for lam in (10, 100):
        for i, u in enumerate(meta["grid_u"]):
            for j, v in enumerate(meta["grid_v"]):
                assert orc.thresh_F(X, None, U, V, u, v, lam) == pytest.approx(z[f"F_grid_lam{lam}"][i, j], rel=1e-12)
                np.testing.assert_allclose(orc.thresh_dF(X, None, U, V, u, v, lam), z[f"dF_grid_lam{lam}"][i, j],
                                           rtol=1e-9, atol=1e-9)
        res = orc.threshold_fit(X, U, V, None, u=0.5, v=0.5, lamda=lam, min_diff=1e-3, max_iter=100)
        g = meta[f"lam{lam}"]
        assert res["calls"] == g["calls"]
        rows = np.array(g["rows"]["rows"])
        mine = np.array([r[:4] for r in res["rows"]])
        np.testing.assert_allclose(mine, rows[:, :4], rtol=1e-9)
        scores = np.array([orc.boolean_scores(*r[4:]) for r in res["rows"]])
        np.testing.assert_allclose(scores, rows[:, 4:], rtol=1e-15)
        assert res["u"] == pytest.approx(g["u"], rel=1e-9) and res["v"] == pytest.approx(g["v"], rel=1e-9)


NORMALIZE_METHODS = ("balance", "matrixwise-normalize", "columnwise-normalize", "matrixwise-mapping", "columnwise-mapping")


def test_normalize_methods(golden_dir):
    """normalize_UV for every normalize_method, and the line search that follows it (reference golden g12)."""
    z = np.load(os.path.join(golden_dir, "g12_normalize.npz"))
    meta = json.load(open(os.path.join(golden_dir, "g12_normalize.json")))
    X = z["X"].astype(np.float64)
    for method in NORMALIZE_METHODS:
        U, V = orc.normalize_factors(z["U0"], z["V0"], method)
        np.testing.assert_allclose(U, z[f"U_{method}"], rtol=1e-15, atol=0)
        np.testing.assert_allclose(V, z[f"V_{method}"], rtol=1e-15, atol=0)
        res = orc.threshold_fit(X, U, V, None, u=0.4, v=0.4, lamda=10, min_diff=1e-3, max_iter=8)
        rows = np.array(meta[method]["rows"]["rows"])
        np.testing.assert_allclose(np.array([r[:4] for r in res["rows"]]), rows[:, :4], rtol=1e-9)
        assert res["u"] == pytest.approx(meta[method]["u"], rel=1e-9) and res["v"] == pytest.approx(meta[method]["v"], rel=1e-9)


def test_metrics(golden_dir):
    cases = json.load(open(os.path.join(golden_dir, "g5_metrics.json")))
    for c in cases:
        gt = unpack(c["gt_bits"], c["shape"]).astype(np.int64)
        pd = unpack(c["pd_bits"], c["shape"]).astype(np.int64)
        tp, fp, fn, tn = orc.confusion_counts(gt, pd)
        m = c["metrics"]
        assert (tp, fp, tn, fn) == (m["TP"], m["FP"], m["TN"], m["FN"])
        r, p, a, f1 = orc.boolean_scores(tp, fp, fn, tn)
        assert (r, p, a, f1) == (m["Recall"], m["Precision"], m["Accuracy"], m["F1"])
        rmse, mae = orc.rmse_mae(gt, pd)
        assert rmse == pytest.approx(m["RMSE"], rel=1e-15) and mae == pytest.approx(m["MAE"], rel=1e-15)


def test_masked_updates(golden_dir):
    """W='mask' on a csr with explicit zeros (observed cells = stored entries): BinaryMFPenalty and WNMF."""
    z = np.load(os.path.join(golden_dir, "g7_masked.npz"))
    meta = json.load(open(os.path.join(golden_dir, "g7_masked.json")))
    m, n = z["shape"]
    X = np.zeros((m, n))
    W = np.zeros((m, n))
    X[z["rows"], z["cols"]] = z["vals"]
    W[z["rows"], z["cols"]] = 1.0
    res = orc.penalty_fit(X, k=6, U=z["p_U0"], V=z["p_V0"], W=W, reg=1.0, reg_growth=1.3, init_method="custom",
                          normalize_method=None, max_iter=7)
    np.testing.assert_allclose(res["U"], z["p_U"], rtol=1e-10, atol=1e-300)
    np.testing.assert_allclose(res["V"], z["p_V"], rtol=1e-10, atol=1e-300)
    np.testing.assert_allclose(np.array(res["updates"]), np.array(meta["penalty"]["updates"]["rows"]), rtol=1e-10)
    np.testing.assert_allclose(np.array(res["boolean"]), np.array(meta["penalty"]["boolean"]["rows"]), rtol=1e-14, atol=0)
    assert res["reg"] == pytest.approx(meta["penalty"]["final_reg"], rel=1e-15)
    w = orc.wnmf_fit(X, k=6, U=z["w_U0"], V=z["w_V0"], W=W, init_method="custom", max_iter=7)
    np.testing.assert_allclose(w["U"], z["w_U"], rtol=1e-10, atol=1e-300)
    np.testing.assert_allclose(np.array(w["updates"]), np.array(meta["wnmf"]["updates"]["rows"]), rtol=1e-10)
    # a column / row without observed cells is driven to exactly 0 by WNMF (0/eps) and to eps by the penalty model
    assert (w["V"][9] == 0).all() and (w["U"][5] == 0).all()
    assert (res["V"][9] > 0).all()


def test_pnlpf_under_a_mask(golden_dir):
    """PNLPF with W='mask' on a csr with explicit zeros (the class) and with a real weight matrix (module-level steps): golden g16."""
    z = np.load(os.path.join(golden_dir, "g16_pnlpf_masked.npz"))
    meta = json.load(open(os.path.join(golden_dir, "g16_pnlpf_masked.json")))
    m, n = z["shape"]
    X = np.zeros((m, n)); W = np.zeros((m, n))
    X[z["rows"], z["cols"]] = z["vals"]
    W[z["rows"], z["cols"]] = 1.0
    p = meta["pnlpf"]["params"]
    res = orc.pnlpf_fit(X, k=p["k"], U=z["p_U0"], V=z["p_V0"], W=W, reg=p["reg"], reg_growth=p["reg_growth"], link_lamda=p["link_lamda"],
                        init_method="custom", normalize_method=None, max_iter=p["max_iter"])
    np.testing.assert_allclose(res["U"], z["p_U"], rtol=1e-9, atol=1e-300)
    np.testing.assert_allclose(res["V"], z["p_V"], rtol=1e-9, atol=1e-300)
    np.testing.assert_allclose(np.array(res["updates"]), np.array(meta["pnlpf"]["updates"]["rows"]), rtol=1e-9)
    np.testing.assert_allclose(np.array(res["boolean"]), np.array(meta["pnlpf"]["boolean"]["rows"]), rtol=1e-14, atol=0)
    Xd = np.unpackbits(z["Xd"], axis=1)[:, :n].astype(np.float64)
    for i, st in enumerate(meta["steps"]):
        Vn = orc.pnlpf_update_V(Xd, z["Wr"], z["s_U"], z["s_V"], st["reg"], st["link_lamda"])
        Un = orc.pnlpf_update_U(Xd, z["Wr"], z["s_U"], Vn, st["reg"], st["link_lamda"])
        np.testing.assert_allclose(Vn, z[f"step{i}_V"], rtol=1e-11, atol=1e-300)
        np.testing.assert_allclose(Un, z[f"step{i}_U"], rtol=1e-11, atol=1e-300)


def test_masked_threshold_objective(golden_dir):
    z7 = np.load(os.path.join(golden_dir, "g7_masked.npz"))
    z8 = np.load(os.path.join(golden_dir, "g8_threshold_masked.npz"))
    meta = json.load(open(os.path.join(golden_dir, "g8_threshold_masked.json")))
    m, n = z7["shape"]
    X = np.zeros((m, n)); W = np.zeros((m, n))
    X[z7["rows"], z7["cols"]] = z7["vals"]
    W[z7["rows"], z7["cols"]] = 1.0
    U, V = z7["w_U"], z7["w_V"]
    for i, a in enumerate(z8["grid"]):
        for j, b in enumerate(z8["grid"]):
            assert orc.thresh_F(X, W, U, V, a, b, 10) == pytest.approx(z8["F_grid"][i, j], rel=1e-12)
            np.testing.assert_allclose(orc.thresh_dF(X, W, U, V, a, b, 10), z8["dF_grid"][i, j], rtol=1e-9, atol=1e-9)
    res = orc.threshold_fit(X, U, V, W, u=0.3, v=0.3, lamda=10, min_diff=1e-3, max_iter=40)
    rows = np.array(meta["rows"]["rows"])
    np.testing.assert_allclose(np.array([r[:4] for r in res["rows"]]), rows[:, :4], rtol=1e-9)
    assert res["u"] == pytest.approx(meta["u"], rel=1e-9)


def test_rank_above_64_under_a_mask(golden_dir):
    """Golden g20 (k = 72, the reference has no rank limit): the masked penalty and WNMF fits on a csr with explicit zeros, every log row;
    the final factors are stored in fp32 (1e-6)."""
    z = np.load(os.path.join(golden_dir, "g20_wide_rank.npz"))
    meta = json.load(open(os.path.join(golden_dir, "g20_wide_rank.json")))
    m, n = z["shape"]
    X = np.zeros((m, n)); W = np.zeros((m, n))
    X[z["train_rows"], z["train_cols"]] = z["train_vals"]
    W[z["train_rows"], z["train_cols"]] = 1.0
    p = meta["params"]["penalty"]
    ref = meta["penalty_mask_reconstruction"]
    res = orc.penalty_fit(X, k=p["k"], U=z["pen_U0"], V=z["pen_V0"], W=W, reg=p["reg"], reg_growth=p["reg_growth"], init_method="custom",
                          normalize_method=None, max_iter=p["max_iter"])
    np.testing.assert_allclose(res["U"], z["penalty_mask_reconstruction_U"], rtol=1e-6, atol=1e-30)
    np.testing.assert_allclose(res["V"], z["penalty_mask_reconstruction_V"], rtol=1e-6, atol=1e-30)
    np.testing.assert_allclose(np.array(res["updates"]), np.array(ref["updates"]["rows"]), rtol=1e-10)
    np.testing.assert_allclose(np.array(res["boolean"]), np.array(ref["boolean"]["rows"]), rtol=1e-14, atol=0)
    assert res["reg"] == pytest.approx(ref["final_reg"], rel=1e-15)
    pw = meta["params"]["wnmf"]
    w = orc.wnmf_fit(X, k=pw["k"], U=z["wnmf_U0"], V=z["wnmf_V0"], W=W, init_method="custom", max_iter=pw["max_iter"])
    np.testing.assert_allclose(w["U"], z["wnmf_mask_reconstruction_U"], rtol=1e-6, atol=1e-30)
    np.testing.assert_allclose(np.array(w["updates"]), np.array(meta["wnmf_mask_reconstruction"]["updates"]["rows"]), rtol=1e-10)
    # task='prediction' with val / test: the trajectory is the same masked fit; the logged scores of the last row follow from the final
    # factors through the entry-wise scorer (non-zero entries of each set)
    up = meta["penalty_prediction"]["updates"]
    col = {tuple(c): i for i, c in enumerate(up["columns"])}
    for name in ("val", "test"):
        keep = z[name + "_vals"] != 0
        rmse, mae = orc.entry_scores(z[name + "_rows"][keep], z[name + "_cols"][keep], z[name + "_vals"][keep], res["U"], res["V"])
        assert rmse == pytest.approx(up["rows"][-1][col[(name, "0", "RMSE")]], rel=1e-9)
        assert mae == pytest.approx(up["rows"][-1][col[(name, "0", "MAE")]], rel=1e-9)


def test_prediction_task_scores(golden_dir):
    """fit(X_train, X_val, X_test, task='prediction'): the logged scores of the last row follow from the final factors through
    the entry-wise scorer; under task='reconstruction' val / test are scored as whole matrices."""
    z = np.load(os.path.join(golden_dir, "g9_prediction.npz"))
    meta = json.load(open(os.path.join(golden_dir, "g9_prediction.json")))
    m, n = z["shape"]

    def entries(name):  # the non-zero entries of the set
        keep = z[name + "_vals"] != 0
        return z[name + "_rows"][keep], z[name + "_cols"][keep], z[name + "_vals"][keep]

    def dense(name):
        X = np.zeros((m, n))
        X[z[name + "_rows"], z[name + "_cols"]] = z[name + "_vals"]
        return X

    up, bo = meta["penalty_prediction"]["updates"], meta["penalty_prediction"]["boolean"]
    col = {tuple(c): i for i, c in enumerate(up["columns"])}
    bcol = {tuple(c): i for i, c in enumerate(bo["columns"])}
    for name in ("train", "val", "test"):
        rmse, mae = orc.entry_scores(*entries(name), z["p_U"], z["p_V"])
        assert rmse == pytest.approx(up["rows"][-1][col[(name, "0", "RMSE")]], rel=1e-12)
        assert mae == pytest.approx(up["rows"][-1][col[(name, "0", "MAE")]], rel=1e-12)
        sc = orc.boolean_scores(*orc.entry_scores(*entries(name), z["p_U"], z["p_V"], 0.5, 0.5))
        ref = [bo["rows"][-1][bcol[(name, "0", mt)]] for mt in ("Recall", "Precision", "Accuracy", "F1")]
        np.testing.assert_allclose(sc, ref, rtol=1e-14)
    # WNMF: train is scored over the whole matrix (its error() fills the zero cells with eps), val / test by entries
    up = meta["wnmf_prediction"]["updates"]
    col = {tuple(c): i for i, c in enumerate(up["columns"])}
    rmse, mae = orc.rmse_mae(dense("train"), orc.real_product(z["w_U"], z["w_V"]))
    assert rmse == pytest.approx(up["rows"][-1][col[("train", "0", "RMSE")]], rel=1e-10)
    for name in ("val", "test"):
        rmse, mae = orc.entry_scores(*entries(name), z["w_U"], z["w_V"])
        assert rmse == pytest.approx(up["rows"][-1][col[(name, "0", "RMSE")]], rel=1e-12)
        assert mae == pytest.approx(up["rows"][-1][col[(name, "0", "MAE")]], rel=1e-12)
    # thresholds learnt on the WNMF factors
    th = meta["threshold_prediction"]
    col = {tuple(c): i for i, c in enumerate(th["updates"]["columns"])}
    for name in ("train", "val", "test"):
        sc = orc.boolean_scores(*orc.entry_scores(*entries(name), z["w_U"], z["w_V"], th["u"], th["v"]))
        ref = [th["updates"]["rows"][-1][col[(name, "0", mt)]] for mt in ("Recall", "Precision", "Accuracy", "F1")]
        np.testing.assert_allclose(sc, ref, rtol=1e-14)
    # reconstruction: whole-matrix scores of each set; the trajectory itself is the W='full' penalty fit on X_train
    up, bo = meta["penalty_reconstruction"]["updates"], meta["penalty_reconstruction"]["boolean"]
    col = {tuple(c): i for i, c in enumerate(up["columns"])}
    bcol = {tuple(c): i for i, c in enumerate(bo["columns"])}
    res = orc.penalty_fit(dense("train"), k=5, U=z["r_U0"], V=z["r_V0"], reg=1.0, reg_growth=1.3, init_method="custom",
                          normalize_method=None, max_iter=5)
    np.testing.assert_allclose(res["U"], z["r_U"], rtol=1e-10, atol=1e-300)
    for name in ("train", "val", "test"):
        rmse, mae = orc.rmse_mae(dense(name), orc.real_product(z["r_U"], z["r_V"]))
        assert rmse == pytest.approx(up["rows"][-1][col[(name, "0", "RMSE")]], rel=1e-10)
        assert mae == pytest.approx(up["rows"][-1][col[(name, "0", "MAE")]], rel=1e-10)
        sc = orc.boolean_scores(*orc.confusion_counts(dense(name), orc.boolean_product(z["r_U"], z["r_V"], 0.5, 0.5)))
        ref = [bo["rows"][-1][bcol[(name, "0", mt)]] for mt in ("Recall", "Precision", "Accuracy", "F1")]
        np.testing.assert_allclose(sc, ref, rtol=1e-14)


def test_link_models(golden_dir):
    """PNLPF (sigmoid link) and WNMF with the Kullback-Leibler loss: whole trajectories against the reference (g10)."""
    z = np.load(os.path.join(golden_dir, "g10_link_models.npz"))
    meta = json.load(open(os.path.join(golden_dir, "g10_link_models.json")))
    m, n = z["shape"]
    X = np.unpackbits(z["X"], axis=1)[:, :n].astype(np.float64)
    p = meta["pnlpf"]["params"]
    res = orc.pnlpf_fit(X, k=p["k"], U=z["p_U0"], V=z["p_V0"], reg=p["reg"], reg_growth=p["reg_growth"],
                        link_lamda=p["link_lamda"], init_method="custom", normalize_method=None, max_iter=p["max_iter"])
    np.testing.assert_allclose(res["U"], z["p_U"], rtol=1e-10, atol=1e-300)
    np.testing.assert_allclose(res["V"], z["p_V"], rtol=1e-10, atol=1e-300)
    np.testing.assert_allclose(np.array(res["updates"]), np.array(meta["pnlpf"]["updates"]["rows"]), rtol=1e-10)
    np.testing.assert_allclose(np.array(res["boolean"]), np.array(meta["pnlpf"]["boolean"]["rows"]), rtol=1e-14, atol=0)
    assert res["reg"] == pytest.approx(meta["pnlpf"]["final_reg"], rel=1e-15)
    w = orc.wnmf_kl_fit(X, k=p["k"], U=z["w_U0"], V=z["w_V0"], init_method="custom", max_iter=9)
    np.testing.assert_allclose(w["U"], z["w_U"], rtol=1e-10, atol=1e-300)
    np.testing.assert_allclose(w["V"], z["w_V"], rtol=1e-10, atol=1e-300)
    np.testing.assert_allclose(np.array(w["updates"]), np.array(meta["wnmf_kl"]["updates"]["rows"]), rtol=1e-10)


def test_kl_with_the_default_mask(golden_dir):
    """WNMF-KL under W='mask' (reference golden g13): dense Boolean X (pattern = non-zeros) and a csr with explicit zeros."""
    z10 = np.load(os.path.join(golden_dir, "g10_link_models.npz"))
    z = np.load(os.path.join(golden_dir, "g13_kl_mask.npz"))
    meta = json.load(open(os.path.join(golden_dir, "g13_kl_mask.json")))
    m, n = z10["shape"]
    X = np.unpackbits(z["X"], axis=1)[:, :n].astype(np.float64)
    Wcsr = np.zeros((m, n))
    Wcsr[z["rows"], z["cols"]] = 1.0
    for tag, W in (("dense", (X != 0).astype(np.float64)), ("csr", Wcsr)):
        w = orc.wnmf_kl_fit(X.copy(), k=6, U=z10["w_U0"], V=z10["w_V0"], W=W, init_method="custom", max_iter=6)
        np.testing.assert_allclose(w["U"], z[tag + "_U"], rtol=1e-10, atol=1e-300)
        np.testing.assert_allclose(w["V"], z[tag + "_V"], rtol=1e-10, atol=1e-300)
        np.testing.assert_allclose(np.array(w["updates"]), np.array(meta[tag]["updates"]["rows"]), rtol=1e-10)


def test_kl_with_a_weight_matrix(golden_dir):
    """WNMF-KL under a real weight matrix (reference golden g18): the oracle's restatement of WNMF.py:111-129,143-145."""
    z10 = np.load(os.path.join(golden_dir, "g10_link_models.npz"))
    z = np.load(os.path.join(golden_dir, "g18_kl_weights.npz"))
    ref = json.load(open(os.path.join(golden_dir, "g18_kl_weights.json")))["updates"]
    m, n = z10["shape"]
    X = np.unpackbits(z10["X"], axis=1)[:, :n].astype(np.float64)
    w = orc.wnmf_kl_fit(X.copy(), k=6, U=z10["w_U0"], V=z10["w_V0"], W=z["Wr"], init_method="custom", max_iter=6)
    np.testing.assert_allclose(w["U"], z["U"], rtol=1e-10, atol=1e-300)
    np.testing.assert_allclose(w["V"], z["V"], rtol=1e-10, atol=1e-300)
    np.testing.assert_allclose(np.array(w["updates"]), np.array(ref["rows"]), rtol=1e-10)


def test_real_valued_data(golden_dir):
    """Training data that is not 0 / 1 (reference golden g19: the reference casts whatever it is given to float64 and runs,
    ContinuousModel.py:188-203): BinaryMFPenalty under W='full' / 'mask' / a weight matrix, PNLPF, WNMF-KL, BinaryMFThreshold -- the
    trajectories, and the reference's arithmetic "confusion" metrics on a real-valued ground truth (utils/metrics.py:56-135), which the
    oracle restates as real_confusion / real_scores."""
    z = np.load(os.path.join(golden_dir, "g19_real_valued.npz"))
    meta = json.load(open(os.path.join(golden_dir, "g19_real_valued.json")))
    pen = meta["params"]["penalty"]
    kw = dict(k=pen["k"], reg=pen["reg"], reg_growth=pen["reg_growth"], init_method="custom", normalize_method=None, max_iter=pen["max_iter"])
    for tag, X in (("x01", z["X01"]), ("x3", z["X3"])):
        assert not orc.is_boolean_valued(X)
        res = orc.penalty_fit(X, U=z[f"pen_{tag}_U0"], V=z[f"pen_{tag}_V0"], **kw)
        np.testing.assert_allclose(res["U"], z[f"pen_{tag}_U"], rtol=1e-9, atol=1e-300)
        np.testing.assert_allclose(np.array(res["updates"]), np.array(meta[f"pen_{tag}"]["updates"]["rows"]), rtol=1e-9)
        np.testing.assert_allclose(np.array(res["boolean"]), np.array(meta[f"pen_{tag}"]["boolean"]["rows"]), rtol=1e-11, atol=0)
        assert res["reg"] == pytest.approx(meta[f"pen_{tag}"]["final_reg"], rel=1e-15)
        q = orc.pnlpf_fit(X, U=z[f"pnlpf_{tag}_U0"], V=z[f"pnlpf_{tag}_V0"], link_lamda=meta["params"]["link_lamda"], **kw)
        np.testing.assert_allclose(q["U"], z[f"pnlpf_{tag}_U"], rtol=1e-8, atol=1e-300)
        np.testing.assert_allclose(np.array(q["updates"]), np.array(meta[f"pnlpf_{tag}"]["updates"]["rows"]), rtol=1e-8)
        np.testing.assert_allclose(np.array(q["boolean"]), np.array(meta[f"pnlpf_{tag}"]["boolean"]["rows"]), rtol=1e-10, atol=0)
        w = orc.wnmf_kl_fit(X, k=pen["k"], U=z[f"kl_{tag}_U0"], V=z[f"kl_{tag}_V0"], init_method="custom", max_iter=pen["max_iter"])
        np.testing.assert_allclose(w["U"], z[f"kl_{tag}_U"], rtol=1e-9, atol=1e-300)
        np.testing.assert_allclose(np.array(w["updates"]), np.array(meta[f"kl_{tag}"]["updates"]["rows"]), rtol=1e-9)
        th = meta["params"]["threshold"]
        U, V = z[f"pen_{tag}_U"], z[f"pen_{tag}_V"]
        assert orc.thresh_F(X, None, U, V, th["u"], th["v"], th["lamda"]) == pytest.approx(meta[f"thr_{tag}"]["F0"], rel=1e-12)
        np.testing.assert_allclose(orc.thresh_dF(X, None, U, V, th["u"], th["v"], th["lamda"]), z[f"thr_{tag}_dF0"], rtol=1e-9)
        t = orc.threshold_fit(X, U, V, None, u=th["u"], v=th["v"], lamda=th["lamda"], min_diff=th["min_diff"], max_iter=th["max_iter"])
        rows = np.array(meta[f"thr_{tag}"]["rows"]["rows"])
        np.testing.assert_allclose(np.array([r[:4] for r in t["rows"]]), rows[:, :4], rtol=1e-8)
        sc = np.array([orc.real_scores(*r[4:], cells=float(X.size)) for r in t["rows"]])
        np.testing.assert_allclose(sc, rows[:, 4:], rtol=1e-10)
    # the masks on the [0, 1] data
    X = z["X01"]
    W = np.zeros_like(X)
    W[z["mask_rows"], z["mask_cols"]] = 1.0
    for tag, Wm, Xm in (("pen_mask", W, X * W), ("pen_wgt", z["Wr"], X)):
        res = orc.penalty_fit(Xm, U=z[f"{tag}_U0"], V=z[f"{tag}_V0"], W=Wm, **kw)
        np.testing.assert_allclose(res["U"], z[f"{tag}_U"], rtol=1e-9, atol=1e-300)
        np.testing.assert_allclose(np.array(res["updates"]), np.array(meta[tag]["updates"]["rows"]), rtol=1e-9)
        np.testing.assert_allclose(np.array(res["boolean"]), np.array(meta[tag]["boolean"]["rows"]), rtol=1e-11, atol=0)


def test_val_and_test_sets_of_the_link_and_proximal_models(golden_dir):
    """Reference golden g17 against the oracle: PNLPF's per-iteration scores of X_val / X_test under task='prediction' -- RMSE / MAE over
    the NON-ZERO cells of each set against the sigmoid-link prediction, Boolean scores over the same cells (BaseModel.evaluate :209-257
    through the inherited loop) -- from the oracle's PNLPF trajectory."""
    z9 = np.load(os.path.join(golden_dir, "g9_prediction.npz"))
    z = np.load(os.path.join(golden_dir, "g17_val_test_sets.npz"))
    g = json.load(open(os.path.join(golden_dir, "g17_val_test_sets.json")))["pnlpf_prediction"]
    m, n = (int(v) for v in z9["shape"])
    X = np.zeros((m, n))
    W = np.zeros((m, n))
    X[z9["train_rows"], z9["train_cols"]] = z9["train_vals"]
    W[z9["train_rows"], z9["train_cols"]] = 1.0
    fit = orc.pnlpf_fit(X, k=5, U=z["pnlpf_prediction_U0"], V=z["pnlpf_prediction_V0"], W=W, reg=1.0, reg_growth=1.2, link_lamda=10,
                        init_method="custom", normalize_method=None, max_iter=6, trace=True)
    cols = [tuple(c) for c in g["updates"]["columns"]]
    rows = np.array(g["updates"]["rows"], dtype=np.float64)
    np.testing.assert_allclose(fit["U"], z["pnlpf_prediction_U"], rtol=1e-9)
    for name in ("val", "test"):
        r, c, v = z9[name + "_rows"], z9[name + "_cols"], z9[name + "_vals"].astype(np.float64)
        keep = v != 0     # the continuous models densify their data sets: the entries eval() sees are the non-zero cells (g9)
        jr, jm = cols.index((name, "0", "RMSE")), cols.index((name, "0", "MAE"))
        for it, (U, V) in enumerate(fit["trace"]):
            pd = orc.pnlpf_prediction(U, V, 10)[r[keep], c[keep]]
            d = v[keep] - pd
            assert np.sqrt((d ** 2).mean()) == pytest.approx(rows[it, jr], rel=1e-9)
            assert np.abs(d).mean() == pytest.approx(rows[it, jm], rel=1e-9)


def _g11_cases(golden_dir):
    for c in json.load(open(os.path.join(golden_dir, "g11_cover_scores.json"))):
        m, n, k = c["shape"]
        gt = np.unpackbits(np.array(c["gt"], dtype=np.uint8), axis=1)[:, :n].astype(np.int64)
        U, V = np.array(c["U"]), np.array(c["V"])
        yield c, gt, U, V, np.minimum(U @ V.T, 1)


def test_cover_scores(golden_dir):
    """Confusion counts with axis, coverage_score, weighted_error, description_length (reference golden g11)."""
    for c, gt, U, V, pd in _g11_cases(golden_dir):
        for ax in (None, 0, 1):
            ref = c["all" if ax is None else f"axis{ax}"]
            tp, fp, fn, tn = orc.confusion_counts_axis(gt, pd, ax)
            for nm, v in (("TP", tp), ("FP", fp), ("FN", fn), ("TN", tn)):
                assert np.array_equal(np.asarray(v, dtype=float), np.asarray(ref[nm]))
            np.testing.assert_allclose(orc.coverage_score(gt, pd, axis=ax), ref["coverage_score_0.5"], rtol=1e-15)
            np.testing.assert_allclose(orc.coverage_score(gt, pd, w_fp=0.3, axis=ax), ref["coverage_score_0.3"], rtol=1e-15)
            np.testing.assert_allclose(orc.weighted_error(gt, pd, w_fp=0.2, w_fn=0.7, axis=ax), ref["weighted_error_0.2_0.7"], rtol=1e-15)
        assert orc.description_length(gt, U, V) == c["description_length"]
        assert orc.description_length(gt, U, V, pd=pd, w_model=0.5, w_fp=2.0, w_fn=3.0) == c["description_length_w"]


# ---- ELBMF / PRIMP (golden g14) -----------------------------------------------------------------------
def _g14(golden_dir):
    z = np.load(os.path.join(golden_dir, "g14_palm.npz"))
    meta = json.load(open(os.path.join(golden_dir, "g14_palm.json")))
    m, n, k = (int(v) for v in z["shape"])
    X = np.unpackbits(z["X"], axis=1)[:, :n].astype(np.float64)
    return z, meta, X


def test_elbmf_elementwise_pieces(golden_dir):
    z, meta, _ = _g14(golden_dir)
    for i, (kai, lam) in enumerate(meta["prox_params"]):
        np.testing.assert_allclose(orc.elbmf_prox(z["prox_in"].copy(), kai, lam), z[f"prox_out_{i}"], rtol=0, atol=1e-15)
    for g in meta["gap"]:
        assert orc.elbmf_integrality_gap(z["prox_in"], g["reg_l1"], g["reg_l2"]) == pytest.approx(g["value"], rel=1e-14)
    assert meta["crashes_as_shipped"]["ELBMF"].startswith("TypeError") and meta["crashes_as_shipped"]["PRIMP"].startswith("AttributeError")


def test_elbmf_single_steps(golden_dir):
    z, meta, X = _g14(golden_dir)
    W = np.ones_like(X)
    for i, p in enumerate(meta["steps"]):
        for reassoc in (False, True):
            Un, Ul = orc.elbmf_update(X, z["U0"], z["V0"], None if reassoc else W, p["reg_l1"], p["reg_l2"], p["beta"], z["U_prev"], reassoc)
            Vn, _ = orc.elbmf_update(X.T, z["V0"], Un, None if reassoc else W.T, p["reg_l1"], p["reg_l2"], p["beta"], z["V0"], reassoc)
            tol = 1e-11 if reassoc else 1e-13
            np.testing.assert_allclose(Un, z[f"step{i}_U"], rtol=tol, atol=tol)
            np.testing.assert_allclose(Vn, z[f"step{i}_V"], rtol=tol, atol=tol)
            assert np.array_equal(Ul, z["U0"])


@pytest.mark.parametrize("tag", ["palm", "ipalm"])
def test_elbmf_loop(golden_dir, tag):
    z, meta, X = _g14(golden_dir)
    g = meta[tag]
    for reassoc in (False, True):
        res = orc.elbmf_fit(X, z["U0"], z["V0"], None, reg_l1=0.01, reg_l2=0.02, reg_growth=1.05, beta=g["beta"], tol=0.0,
                            max_iter=g["max_iter"], min_diff=1e-8, reassoc=reassoc)
        want = np.array(g["updates"]["rows"], dtype=np.float64)
        got = np.array([list(u) + list(s) for u, s in zip(res["updates"], res["scores"])])
        assert got.shape == want.shape
        np.testing.assert_allclose(got, want, rtol=1e-9)
        np.testing.assert_allclose(res["U"], z[f"{tag}_U"], rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(res["V"], z[f"{tag}_V"], rtol=1e-9, atol=1e-12)
        assert list(res["counts"][-1]) == g["counts"]


def test_elbmf_under_a_mask(golden_dir):
    """ELBMF's gradient under a mask / weight matrix, multiply(W, U V^T - X) V (PyBMF/models/ELBMF.py:177-196): the module-level step
    with a 0/1 mask and with real weights, and the class's loop with W='mask' on a csr with explicit zeros (golden g15)."""
    z = np.load(os.path.join(golden_dir, "g15_elbmf_masked.npz"))
    meta = json.load(open(os.path.join(golden_dir, "g15_elbmf_masked.json")))
    m, n, k = (int(v) for v in z["shape"])
    X = np.unpackbits(z["X"], axis=1)[:, :n].astype(np.float64)
    W01 = np.unpackbits(z["W01"], axis=1)[:, :n].astype(np.float64)
    Ws = {"W01": W01, "Wr": z["Wr"]}
    for i, p in enumerate(meta["steps"]):
        W = Ws[p["W"]]
        Un, Ul = orc.elbmf_update(X, z["U0"], z["V0"], W, p["reg_l1"], p["reg_l2"], p["beta"], z["U_prev"])
        Vn, _ = orc.elbmf_update(X.T, z["V0"], z["U0"], W.T, p["reg_l1"], p["reg_l2"], p["beta"], z["V0"])
        np.testing.assert_allclose(Un, z[f"mstep{i}_U"], rtol=1e-12, atol=1e-14)
        np.testing.assert_allclose(Vn, z[f"mstep{i}_V"], rtol=1e-12, atol=1e-14)
        assert Ul is z["U0"] or np.array_equal(Ul, z["U0"])
    for tag in ("mpalm", "mipalm"):
        g = meta[tag]
        res = orc.elbmf_fit(X * W01, z["U0"], z["V0"], W01, reg_l1=0.01, reg_l2=0.02, reg_growth=1.05, beta=g["beta"], max_iter=g["max_iter"],
                            min_diff=1e-8, tol=0.0)
        want = np.array(g["updates"]["rows"], dtype=np.float64)
        got = np.array([list(u) + list(s) for u, s in zip(res["updates"], res["scores"])])
        assert got.shape == want.shape
        np.testing.assert_allclose(got, want, rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(res["U"], z[f"{tag}_U"], rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(res["V"], z[f"{tag}_V"], rtol=1e-9, atol=1e-12)
        assert list(res["counts"][-1]) == g["counts"]


def test_primp_steps_and_runs(golden_dir):
    z, meta, X = _g14(golden_dir)
    for i, p in enumerate(meta["primp_steps"]):
        Un = orc.primp_step(X, z["U0"], np.ascontiguousarray(z["V0"].T), z["U_prev"], p["l1reg"], p["l2reg"], p["tau"], p["beta"])
        np.testing.assert_allclose(Un, z[f"pstep{i}_U"], rtol=1e-12, atol=1e-14)
    for tag, dt, tol in (("primp64", np.float64, 1e-9), ("primp64_b0", np.float64, 1e-9), ("primp32", np.float32, 2e-4)):
        g = meta[tag]
        U, Vt, fns = orc.primp_ipalm(X.astype(dt), z["U0"].astype(dt), np.ascontiguousarray(z["V0"].T).astype(dt), 0.01, 0.0, 1.02,
                                     g["maxiter"], 1e-8, g["beta"])
        np.testing.assert_allclose(U, z[f"{tag}_U"], rtol=tol, atol=tol)
        np.testing.assert_allclose(Vt, z[f"{tag}_Vt"], rtol=tol, atol=tol)
        assert fns[-1] == pytest.approx(g["fn_final"], rel=max(tol, 1e-9))
        if dt is np.float64:
            assert np.array_equal(orc.primp_round(U).astype(np.uint8), z[f"{tag}_Ur"])
            assert np.array_equal(orc.primp_round(Vt).astype(np.uint8), z[f"{tag}_Vtr"])
    # the reference's own fp32 run (what PRIMP._fit would do) stays within the 1e-4 gate of its fp64 run
    assert np.linalg.norm(z["primp32_U"] - z["primp64_U"]) / np.linalg.norm(z["primp64_U"]) < 1e-4


def test_boolean_product_blas_equals_the_restated_product():
    rs = np.random.RandomState(8)
    for m, n, k in ((40, 30, 5), (300, 200, 64), (7, 9, 1)):
        U, V = rs.rand(m, k), rs.rand(n, k)
        for u, v in ((0.5, 0.5), (0.9, 0.2), (0.0, 1.0)):
            assert np.array_equal(orc.boolean_product_blas(U, V, u, v), orc.boolean_product(U, V, u, v).astype(bool))
