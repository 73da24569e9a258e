"""Masked multiplicative updates (W = 'mask' on a csr with unstored cells, or a weight matrix): SURVEY 8f rank 1.
Golden vectors from the reference (g7: BinaryMFPenalty / WNMF on a csr with explicit zeros; g3: WNMF 'mask' on real data)."""
import contextlib
import io
import json
import os

import numpy as np
import pytest
from scipy.sparse import csr_matrix

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


def frame_values(df):
    return np.array([[float(v) for v in row[1:]] for row in df.values.tolist()])


@pytest.fixture(scope="module")
def g7(golden_dir):
    z = np.load(os.path.join(golden_dir, "g7_masked.npz"))
    meta = json.load(open(os.path.join(golden_dir, "g7_masked.json")))
    m, n = z["shape"]
    X = csr_matrix((z["vals"].astype(np.float64), (z["rows"], z["cols"])), shape=(m, n))  # explicit zeros stay stored
    assert X.nnz == len(z["rows"])
    return z, meta, X


def test_masked_pass_kernel_against_numpy(g7):
    import ctypes as C
    from pybmf_amd import _lib as L
    from pybmf_amd.engine import SparseObs
    z, meta, X = g7
    m, n = X.shape
    rs = np.random.RandomState(0)
    for k, kp in ((6, 32), (40, 64)):
        wg = rs.rand(X.nnz).astype(np.float32) + 0.5
        obs = SparseObs(z["rows"], z["cols"], z["vals"], wg, (m, n))
        assert obs.csr["nseg"] == m - 1 and obs.csc["nseg"] == n - 1  # one segment per non-empty row / column here
        U = np.zeros((m, kp), np.float32)
        V = np.zeros((n, kp), np.float32)
        U[:, :k], V[:, :k] = rs.rand(m, k), rs.rand(n, k)
        Ud, Vd = torch.from_numpy(U).cuda(), torch.from_numpy(V).cuda()
        num, den = torch.zeros((m, kp), device="cuda"), torch.zeros((m, kp), device="cuda")
        sums = torch.zeros(2, dtype=torch.float64, device="cuda")
        s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        def run(ls, rows, Fs, Fo, nm, dn, sm):
            part = torch.zeros((max(ls["nseg"], 1), 2, kp), dtype=torch.float32, device="cuda")
            L.check(L.lib.bmf_masked_pass(L.ptr(ls["ptr"]), L.ptr(ls["idx"]), L.ptr(ls["val"]), L.ptr(ls["wgt"]), rows,
                                          L.ptr(ls["seg_row"]), L.ptr(ls["seg_beg"]), ls["nseg"], L.ptr(ls["row_seg_ptr"]), L.ptr(Fs),
                                          L.ptr(Fo), kp, L.ptr(part), L.ptr(nm), L.ptr(dn), L.ptr(sm) if sm is not None else None, s))
        run(obs.csr, m, Ud, Vd, num, den, sums)
        W = np.zeros((m, n)); Xd = np.zeros((m, n))
        W[z["rows"], z["cols"]] = wg
        Xd[z["rows"], z["cols"]] = z["vals"]
        P = U.astype(np.float64) @ V.astype(np.float64).T
        np.testing.assert_allclose(num.cpu().numpy(), (W * Xd) @ V, rtol=2e-5, atol=1e-6)
        np.testing.assert_allclose(den.cpu().numpy(), (W * P) @ V, rtol=2e-5, atol=1e-6)
        got = sums.cpu().numpy()
        assert got[0] == pytest.approx((W * (Xd - P) ** 2).sum(), rel=1e-5) and got[1] == pytest.approx((W * np.abs(Xd - P)).sum(), rel=1e-5)
        # the transposed orientation through the CSC list
        numv, denv = torch.zeros((n, kp), device="cuda"), torch.zeros((n, kp), device="cuda")
        run(obs.csc, n, Vd, Ud, numv, denv, None)
        np.testing.assert_allclose(numv.cpu().numpy(), (W * Xd).T @ U, rtol=2e-5, atol=1e-6)
        np.testing.assert_allclose(denv.cpu().numpy(), (W * P).T @ U, rtol=2e-5, atol=1e-6)


def test_penalty_mask_matches_reference(g7):
    from pybmf_amd.models import BinaryMFPenalty
    z, meta, X = g7
    with quiet():
        mdl = BinaryMFPenalty(k=6, W="mask", reg=1.0, reg_growth=1.3, init_method="normal", normalize_method="balance",
                              max_iter=7, seed=4)
        mdl.fit(X.copy(), **FIT)
    np.testing.assert_allclose(frame_values(mdl.logs["updates"]), np.array(meta["penalty"]["updates"]["rows"]), rtol=1e-4)
    np.testing.assert_allclose(frame_values(mdl.logs["boolean"]), np.array(meta["penalty"]["boolean"]["rows"]), rtol=1e-15, atol=0)
    assert relf(mdl.U, z["p_U"]) < 1e-4 and relf(mdl.V, z["p_V"]) < 1e-4
    assert float(mdl.reg) == pytest.approx(meta["penalty"]["final_reg"], rel=1e-15)


def test_wnmf_mask_matches_reference(g7, golden_dir):
    from pybmf_amd.models import WNMF
    z, meta, X = g7
    with quiet():
        w = WNMF(k=6, W="mask", init_method="normal", max_iter=7, seed=4)
        w.fit(X.copy(), **FIT)
    np.testing.assert_allclose(frame_values(w.logs["updates"]), np.array(meta["wnmf"]["updates"]["rows"]), rtol=1e-4)
    assert relf(w.U, z["w_U"]) < 1e-4 and relf(w.V, z["w_V"]) < 1e-4
    assert (w.V[9] == 0).all() and (w.U[5] == 0).all()   # unobserved row / column: 0 / eps = 0 exactly, as in the reference
    # real-valued data with exact zeros, W='mask' = its nonzero pattern (g3)
    z3 = np.load(os.path.join(golden_dir, "g3_wnmf.npz"))
    m3 = json.load(open(os.path.join(golden_dir, "g3_wnmf.json")))
    p = m3["params"]
    with quiet():
        w3 = WNMF(k=p["k"], W="mask", init_method=p["init_method"], max_iter=p["max_iter"], seed=p["seed"])
        w3.fit(z3["X"].copy(), **FIT)
    np.testing.assert_allclose(frame_values(w3.logs["updates"]), np.array(m3["mask"]["rows"]), rtol=1e-4)
    assert relf(w3.U, z3["mask_U"]) < 1e-4 and relf(w3.V, z3["mask_V"]) < 1e-4


def test_explicit_weight_matrix_against_oracle(g7):
    from pybmf_amd.models import BinaryMFPenalty
    z, meta, X = g7
    m, n = X.shape
    rs = np.random.RandomState(3)
    Wm = (rs.rand(m, n) < 0.4) * (0.5 + rs.rand(m, n))      # weights in [0.5, 1.5) on 40 % of the cells
    Xd = np.zeros((m, n)); Xd[z["rows"], z["cols"]] = z["vals"]
    ref = orc.penalty_fit(Xd, k=6, U=z["p_U0"], V=z["p_V0"], W=Wm, reg=0.5, reg_growth=1.2, init_method="custom",
                          normalize_method=None, max_iter=5)
    with quiet():
        mdl = BinaryMFPenalty(k=6, U=z["p_U0"].copy(), V=z["p_V0"].copy(), W=Wm, reg=0.5, reg_growth=1.2, init_method="custom",
                              normalize_method=None, max_iter=5)
        mdl.fit(Xd, **FIT)
    np.testing.assert_allclose(frame_values(mdl.logs["updates"]), np.array(ref["updates"]), rtol=1e-4)
    assert relf(mdl.U, ref["U"]) < 1e-4 and relf(mdl.V, ref["V"]) < 1e-4
    assert mdl.counts[-1] == tuple(ref["counts"][-1])


@pytest.mark.parametrize("k,kcols", [(9, 32), (9, 9), (16, 16), (20, 20), (32, 32)])
def test_masked_pass_long_rows_are_segmented(k, kcols):
    """Power-law shape: one row observes every column, one column every row -> several 64-cell segments per row/column.  kcols: what
    the pass is told about the factor's width -- 32 = one cell per step (bmf_masked_pass), <= 16 / <= 32 = four / two cells per step."""
    import ctypes as C
    from pybmf_amd import _lib as L
    from pybmf_amd.engine import SparseObs
    rs = np.random.RandomState(5)
    m, n, kp = 300, 517, 32
    obs_mask = rs.rand(m, n) < 0.05
    obs_mask[7, :] = True
    obs_mask[:, 100] = True
    obs_mask[11, :] = False
    r, c = np.nonzero(obs_mask)
    vals = (rs.rand(len(r)) < 0.4).astype(np.float32)
    S = SparseObs(r, c, vals, None, (m, n))
    assert S.csr["nseg"] == int(np.ceil(obs_mask.sum(1) / 64).sum()) and S.csr["nseg"] > m
    U = np.zeros((m, kp), np.float32); V = np.zeros((n, kp), np.float32)
    U[:, :k], V[:, :k] = rs.rand(m, k), rs.rand(n, k)
    Ud, Vd = torch.from_numpy(U).cuda(), torch.from_numpy(V).cuda()
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    Xd = np.zeros((m, n)); Xd[r, c] = vals
    P = U.astype(np.float64) @ V.astype(np.float64).T
    for ls, rows, Fs, Fo, want_num, want_den in ((S.csr, m, Ud, Vd, (obs_mask * Xd) @ V, (obs_mask * P) @ V),
                                                  (S.csc, n, Vd, Ud, (obs_mask * Xd).T @ U, (obs_mask * P).T @ U)):
        num, den = torch.full((rows, kp), -1.0, device="cuda"), torch.full((rows, kp), -1.0, device="cuda")
        part = torch.zeros((ls["nseg"], 2, kp), dtype=torch.float32, device="cuda")
        sums = torch.zeros(2, dtype=torch.float64, device="cuda")
        if kcols == 32 and k < 32:
            L.check(L.lib.bmf_masked_pass(L.ptr(ls["ptr"]), L.ptr(ls["idx"]), L.ptr(ls["val"]), None, rows, L.ptr(ls["seg_row"]),
                                          L.ptr(ls["seg_beg"]), ls["nseg"], L.ptr(ls["row_seg_ptr"]), L.ptr(Fs), L.ptr(Fo), kp, L.ptr(part),
                                          L.ptr(num), L.ptr(den), L.ptr(sums), s))
        else:
            L.check(L.lib.bmf_masked_link_pass_k(L.ptr(ls["ptr"]), L.ptr(ls["idx"]), L.ptr(ls["val"]), None, rows, L.ptr(ls["seg_row"]),
                                                 L.ptr(ls["seg_beg"]), ls["nseg"], L.ptr(ls["row_seg_ptr"]), L.ptr(Fs), L.ptr(Fo), kp, kcols,
                                                 L.ptr(part), L.ptr(num), L.ptr(den), L.ptr(sums), 0, 0.0, s))
        np.testing.assert_allclose(num.cpu().numpy(), want_num, rtol=2e-5, atol=1e-6)
        np.testing.assert_allclose(den.cpu().numpy(), want_den, rtol=2e-5, atol=1e-6)
        assert float(sums[0]) == pytest.approx((obs_mask * (Xd - P) ** 2).sum(), rel=1e-5)
        assert float(sums[1]) == pytest.approx((obs_mask * np.abs(Xd - P)).sum(), rel=1e-5)
    assert (num.cpu().numpy() >= 0).all()  # outputs fully overwritten: no -1 left, the empty row is zero
    assert (den.cpu().numpy() >= 0).all()


def test_threshold_default_mask_matches_reference(g7, golden_dir):
    """BinaryMFThreshold with its default W='mask' on a csr with explicit zeros (reference golden g8)."""
    from pybmf_amd.models import BinaryMFThreshold
    z, meta7, X = g7
    z8 = np.load(os.path.join(golden_dir, "g8_threshold_masked.npz"))
    meta = json.load(open(os.path.join(golden_dir, "g8_threshold_masked.json")))
    with quiet():
        mdl = BinaryMFThreshold(k=6, U=z["w_U"].copy(), V=z["w_V"].copy(), u=0.3, v=0.3, lamda=10, min_diff=1e-3, max_iter=40)
        mdl.fit(X.copy(), **FIT)
    for i, a in enumerate(z8["grid"]):
        for j, b in enumerate(z8["grid"]):
            assert mdl.F([a, b]) == pytest.approx(z8["F_grid"][i, j], rel=1e-4)
            want = z8["dF_grid"][i, j]
            np.testing.assert_allclose(mdl.dF([a, b]), want, rtol=1e-3, atol=1e-3 * np.abs(z8["dF_grid"]).max())
    rows = frame_values(mdl.logs["updates"])
    ref = np.array(meta["rows"]["rows"])
    assert len(rows) == len(ref), (len(rows), len(ref))      # fp64 objective on the device: the reference's decisions, row by row
    np.testing.assert_allclose(rows[:, :4], ref[:, :4], rtol=1e-6, atol=1e-9)
    assert mdl.u == pytest.approx(meta["u"], abs=1e-6) and mdl.v == pytest.approx(meta["v"], abs=1e-6)
    assert rows[-1, 3] == pytest.approx(ref[-1, 3], rel=1e-3)


def test_module_level_steps_with_a_weight_matrix():
    """update_V / update_U / error / rec_error as importable functions (PNLPF-style callers) under a weight matrix W --
    zeros (unobserved cells) and non-unit weights -- against the oracle's literal arithmetic."""
    from pybmf_amd.models.BinaryMFPenalty import update_U, update_V, error, rec_error
    rs = np.random.RandomState(11)
    m, n, k = 90, 70, 5
    X = (rs.rand(m, n) < 0.3).astype(np.float64)
    W = rs.choice([0.0, 1.0, 2.5], size=(m, n), p=[0.4, 0.4, 0.2])
    W[7, :] = 0.0            # an unobserved row
    U, V = rs.rand(m, k) * 0.6 + 0.01, rs.rand(n, k) * 0.6 + 0.01
    for reg in (0.0, 1.5):
        V1 = update_V(X, W, U, V, reg)
        assert relf(V1, orc.penalty_update_V(X, W, U, V, reg)) < 2e-6
        U1 = update_U(X, csr_matrix(W), U, V1, reg)
        assert relf(U1, orc.penalty_update_U(X, W, U, V1, reg)) < 2e-6
        want = orc.penalty_errors(X, W, U1, V1, reg)
        np.testing.assert_allclose(error(X, None, W, U1, V1, reg), want, rtol=2e-6)
        assert rec_error(X, None, W, U=U1, V=V1) == pytest.approx(want[1], rel=2e-6)
    # the all-ones mask given explicitly still takes the dense path
    assert relf(update_V(X, np.ones((m, n)), U, V, 1.0), orc.penalty_update_V(X, None, U, V, 1.0)) < 2e-6


def test_masked_fit_on_a_power_law_pattern():
    """A recommender-shaped observation pattern -- a few very long rows and columns (hundreds of 64-cell segments), many
    short ones, some empty -- through WNMF and BinaryMFPenalty with W='mask' against the oracle."""
    from pybmf_amd.models import BinaryMFPenalty, WNMF
    rs = np.random.RandomState(17)
    m, n, k = 3001, 2003, 20
    pr = 1.0 / np.arange(1, m + 1) ** 0.8
    pc = 1.0 / np.arange(1, n + 1) ** 0.8
    P = np.minimum(1.0, 60.0 * np.outer(pr / pr.max(), pc / pc.max()))
    obs = rs.rand(m, n) < P
    vals = (rs.rand(m, n) < 0.5).astype(np.float64)
    r, c = np.nonzero(obs)
    X = csr_matrix((vals[r, c], (r, c)), shape=(m, n))          # explicit zeros stay stored
    assert X.nnz == len(r) and obs.sum(1).max() > 640 and (obs.sum(1) == 0).any()
    W = obs.astype(np.float64)
    Xd = vals * W
    U0 = np.abs(rs.standard_normal((m, k))) * 0.3 + 1e-3
    V0 = np.abs(rs.standard_normal((n, k))) * 0.3 + 1e-3
    ref = orc.penalty_fit(Xd, k=k, U=U0.copy(), V=V0.copy(), reg=1.0, reg_growth=1.3, init_method="custom", normalize_method=None,
                          max_iter=2, tol=-1.0, W=W)
    with quiet():
        p = BinaryMFPenalty(k=k, U=U0.copy(), V=V0.copy(), W="mask", reg=1.0, reg_growth=1.3, init_method="custom", normalize_method=None,
                            max_iter=2, tol=-1.0)
        p.fit(X.copy(), **FIT)
    assert relf(p.U, ref["U"]) < 1e-5 and relf(p.V, ref["V"]) < 1e-5
    np.testing.assert_allclose(frame_values(p.logs["updates"])[:, :5], np.array(ref["updates"])[:, :5], rtol=1e-5)
    refw = orc.wnmf_fit(Xd.copy(), k, U=U0.copy(), V=V0.copy(), W=W, max_iter=2, init_method="custom")
    with quiet():
        w = WNMF(k=k, U=U0.copy(), V=V0.copy(), W="mask", init_method="custom", max_iter=2)
        w.fit(X.copy(), **FIT)
    live_r, live_c = (W * Xd).sum(1) > 0, (W * Xd).sum(0) > 0   # (rows / columns of observed zeros only: see WNMF.py docstring)
    assert relf(w.U[live_r], refw["U"][live_r]) < 1e-5 and relf(w.V[live_c], refw["V"][live_c]) < 1e-5


def test_masked_loop_in_c_calls_takes_the_same_path_as_the_stepwise_loop(monkeypatch):
    """BinaryMFPenalty and WNMF under W='mask' enqueue whole iterations by one C call each (bmf_masked_iterate) and read the scalars of
    iteration t while t + 1 runs, so the loop overshoots its stopping rule by one iteration and returns the iterate before.  Same
    kernels in the same order as the stepwise loop: stopping iteration and factors must be identical, bit for bit; the log rows to 1e-12."""
    from pybmf_amd.models import BinaryMFPenalty, WNMF
    rs = np.random.RandomState(5)
    m, n, k = 700, 500, 12
    obs = rs.rand(m, n) < 0.2
    vals = (rs.rand(m, n) < 0.4).astype(np.float64)
    r, c = np.nonzero(obs)
    X = csr_matrix((vals[r, c], (r, c)), shape=(m, n))
    U0 = np.abs(rs.standard_normal((m, k))) * 0.3 + 1e-3
    V0 = np.abs(rs.standard_normal((n, k))) * 0.3 + 1e-3
    out = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("BMF_MASKED_PIPELINE", flag)
        with quiet():
            p = BinaryMFPenalty(k=k, U=U0.copy(), V=V0.copy(), W="mask", reg=1.0, reg_growth=1.5, init_method="custom", normalize_method=None,
                                max_iter=40, tol=0.0, min_diff=1e-3)
            p.fit(X.copy(), **FIT)
            w = WNMF(k=k, U=U0.copy(), V=V0.copy(), W="mask", init_method="custom", max_iter=7)
            w.fit(X.copy(), **FIT)
        out[flag] = (p.U.copy(), p.V.copy(), frame_values(p.logs["updates"]), p.n_iter, float(p.reg),
                     w.U.copy(), w.V.copy(), frame_values(w.logs["updates"]))
    a, b = out["1"], out["0"]
    assert a[3] == b[3] and 2 <= a[3] <= 41 and a[4] == b[4]   # (n_iter > max_iter ends the reference loop: 41)
    for i in (0, 1, 5, 6):    # factors: bit for bit
        np.testing.assert_array_equal(np.asarray(a[i]), np.asarray(b[i]))
    for i in (2, 7):          # log rows: the residual sums are fp64 atomic accumulations (order-dependent in the last bits)
        np.testing.assert_allclose(np.asarray(a[i]), np.asarray(b[i]), rtol=1e-12, atol=0)


def test_masked_loop_in_c_calls_on_a_real_valued_matrix(monkeypatch):
    """The same equality for WNMF under W='mask' on REAL-valued data (the whole-matrix scores then run on the fp32 copy of X,
    bmf_masked_loop.Xreal), with an early stop before max_iter so that the overshoot-and-return-the-previous-iterate path is the one
    that ends the fit."""
    from pybmf_amd.models import WNMF
    rs = np.random.RandomState(11)
    m, n, k = 400, 260, 8
    obs = rs.rand(m, n) < 0.3
    vals = rs.rand(m, n) * 4 + 0.5
    r, c = np.nonzero(obs)
    X = csr_matrix((vals[r, c], (r, c)), shape=(m, n))
    U0 = np.abs(rs.standard_normal((m, k))) * 0.5 + 1e-2
    V0 = np.abs(rs.standard_normal((n, k))) * 0.5 + 1e-2
    out = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("BMF_MASKED_PIPELINE", flag)
        with quiet():
            w = WNMF(k=k, U=U0.copy(), V=V0.copy(), W="mask", init_method="custom", max_iter=200, min_diff=5.0)
            w.fit(X.copy(), **FIT)
        out[flag] = (w.U.copy(), w.V.copy(), frame_values(w.logs["updates"]), w.n_iter)
    a, b = out["1"], out["0"]
    assert a[3] == b[3] and 2 <= a[3] < 200   # stopped by min_diff, at the same iteration
    np.testing.assert_array_equal(a[0], b[0])
    np.testing.assert_array_equal(a[1], b[1])
    np.testing.assert_allclose(a[2], b[2], rtol=1e-12, atol=0)
