"""SURVEY 8f rank 2 on the GPU: the proximal (PALM / iPALM) steps of ELBMF and PRIMP against reference golden g14
(PyBMF/models/ELBMF.py:110-210, PyBMF/models/PRIMP.py:51-160) and against the oracle on other shapes.
Gate: 1e-4 relative on factors and logged scalars, Boolean counts exact."""
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
    with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
        yield


def relf(a, b):
    return float(np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(np.asarray(b)), 1e-300))


@pytest.fixture(scope="module")
def g14(golden_dir):
    z = np.load(os.path.join(golden_dir, "g14_palm.npz"))
    meta = json.load(open(os.path.join(golden_dir, "g14_palm.json")))
    n = int(z["shape"][1])
    X = np.unpackbits(z["X"], axis=1)[:, :n].astype(np.uint8)
    return z, meta, X


def test_sym_norms_against_numpy():
    from pybmf_amd import _lib as L
    rs = np.random.RandomState(3)
    cases = []
    for k, kp in ((5, 32), (32, 32), (33, 64), (64, 64)):
        A = rs.rand(300, k)
        cases.append((A.T @ A, kp))                                  # a Gram of a positive factor: one dominant eigenvalue
        B = rs.standard_normal((k + 3, k))
        cases.append((B.T @ B, kp))                                  # no dominant direction
    cases.append((np.eye(7) * 2.5, 32))                              # all eigenvalues equal
    Q, _ = np.linalg.qr(rs.standard_normal((20, 20)))
    cases.append((Q @ np.diag([3.0, 3.0 - 1e-9] + [1.0] * 18) @ Q.T, 32))   # a nearly degenerate pair on top
    cases.append((np.outer(np.arange(1, 9.0), np.arange(1, 9.0)), 32))      # rank one
    cases.append((np.zeros((4, 4)), 32))                             # the zero matrix
    for G, kp in cases:
        k = G.shape[0]
        Gp = np.zeros((kp, kp))
        Gp[:k, :k] = G
        d = torch.from_numpy(Gp).cuda()
        out = torch.zeros(2, dtype=torch.float64, device="cuda")
        L.check(L.lib.bmf_sym_norms(L.ptr(d), kp, L.ptr(out), None))
        got = out.cpu().numpy()
        want = (np.linalg.norm(G, ord=2), np.linalg.norm(G))
        assert got[0] == pytest.approx(want[0], rel=1e-9, abs=1e-300), (k, kp)
        assert got[1] == pytest.approx(want[1], rel=1e-12, abs=1e-300)


def test_elbmf_single_steps(g14):
    from pybmf_amd.models.ELBMF import update_U, prox, get_integrality_gap
    z, meta, X = g14
    for i, p in enumerate(meta["steps"]):
        Un, Ul = update_U(X, z["U0"], z["V0"], None, p["reg_l1"], p["reg_l2"], p["beta"], z["U_prev"])
        assert relf(Un, z[f"step{i}_U"]) < 1e-5, (i, relf(Un, z[f"step{i}_U"]))
        assert np.array_equal(Ul, z["U0"])
        Vn, _ = update_U(np.ascontiguousarray(X.T), z["V0"], z[f"step{i}_U"], None, p["reg_l1"], p["reg_l2"], p["beta"], z["V0"])
        assert relf(Vn, z[f"step{i}_V"]) < 1e-5
    for i, (kai, lam) in enumerate(meta["prox_params"]):
        np.testing.assert_allclose(prox(z["prox_in"], kai, lam), z[f"prox_out_{i}"], atol=1e-15)
    for g in meta["gap"]:
        assert get_integrality_gap(z["prox_in"], g["reg_l1"], g["reg_l2"]) == pytest.approx(g["value"], rel=1e-14)


@pytest.mark.parametrize("tag", ["palm", "ipalm"])
def test_elbmf_class_matches_reference_loop(g14, tag):
    from pybmf_amd.models import ELBMF
    z, meta, X = g14
    g = meta[tag]
    with quiet():
        mdl = ELBMF(k=6, U=z["U0"].copy(), V=z["V0"].copy(), W="full", init_method="custom", reg_l1=0.01, reg_l2=0.02, reg_growth=1.05,
                    beta=g["beta"], max_iter=g["max_iter"], min_diff=1e-8, tol=0.0)
        mdl.fit(X, **FIT)
    want = np.array(g["updates"]["rows"], dtype=np.float64)
    got = np.array([[float(v) for v in r[1:]] for r in mdl.logs["updates"].values.tolist()])
    assert [tuple(str(x) for x in c) for c in mdl.logs["updates"].columns][1:] == [tuple(c) for c in g["updates"]["columns"]]
    assert got.shape == want.shape
    np.testing.assert_allclose(got[:, :7], want[:, :7], rtol=1e-4)
    np.testing.assert_allclose(got[:, 7:], want[:, 7:], rtol=1e-12, atol=1e-15)      # scores from exact integer counts
    assert relf(mdl.U, z[f"{tag}_U"]) < 1e-4 and relf(mdl.V, z[f"{tag}_V"]) < 1e-4
    assert list(mdl.counts[-1]) == g["counts"]
    assert mdl.X_pd.sum() == g["counts"][0] + g["counts"][1]


def _fixture_rand(monkeypatch, z):
    """torch.rand replaced by the reference's own draws for primp(seed=3) (g14: `primp_seed3_U0`, `primp_seed3_Vt0`, made with
    torch.manual_seed(3)): the comparison below then does not depend on which torch build generated the fixture."""
    draws = [torch.from_numpy(z["primp_seed3_U0"].copy()), torch.from_numpy(z["primp_seed3_Vt0"].copy())]

    def fake_rand(*shape, dtype=None, **kw):
        t = draws.pop(0)
        assert tuple(t.shape) == tuple(shape), (t.shape, shape)
        return t.to(dtype) if dtype is not None else t
    monkeypatch.setattr(torch, "rand", fake_rand)


def test_primp_loop_and_class(g14, monkeypatch):
    from pybmf_amd.models import PRIMP
    from pybmf_amd.models.PRIMP import elbmf_ipalm
    z, meta, X = g14
    for tag in ("primp64", "primp64_b0", "primp32"):
        g = meta[tag]
        seen = []
        # the reference's signature (PRIMP.py:91-131): V is k x n, the result is (U, V), the callback sees (t, U, V[k x n], fn)
        U, Vt = elbmf_ipalm(X, z["U0"], np.ascontiguousarray(z["V0"].T), 0.01, 0.0, lambda t: 1.02 ** t, g["maxiter"], 1e-8, g["beta"],
                            lambda t, Uc, Vc, fn: seen.append((t, Uc.shape, Vc.shape, float(fn))))
        assert Vt.shape == z[f"{tag}_Vt"].shape
        assert relf(U, z[f"{tag}_U"]) < 1e-4 and relf(Vt, z[f"{tag}_Vt"]) < 1e-4, (tag, relf(U, z[f"{tag}_U"]))
        assert seen[-1][3] == pytest.approx(g["fn_final"], rel=1e-4) and seen[0][1:3] == (U.shape, Vt.shape)
        if tag != "primp32":
            # rounding: identical wherever the factor is not within 1e-4 of the 0.5 threshold
            far = np.abs(z[f"{tag}_U"] - 0.5) > 1e-4
            assert np.array_equal((U > 0.5)[far], z[f"{tag}_Ur"].astype(bool)[far])
    # tensors in -> tensors out, dtype kept
    Ut, Vtt = elbmf_ipalm(torch.from_numpy(X.astype(np.float32)), torch.from_numpy(z["U0"]).float(), torch.from_numpy(z["V0"].T.copy()).float(),
                          0.01, 0, lambda t: 1.02 ** t, 3, 1e-8, 1e-4, None)
    assert isinstance(Ut, torch.Tensor) and Ut.dtype == torch.float32 and tuple(Vtt.shape) == (6, 100)
    _fixture_rand(monkeypatch, z)
    with quiet():
        mdl = PRIMP(k=6, reg=0.01, reg_growth=1.02, max_iter=25, min_diff=1e-8, beta=1e-4, seed=3)
        mdl.fit(X, **FIT)
    agree_u = (mdl.U.astype(np.uint8) == z["primp_seed3_Ur"]).mean()
    agree_v = (mdl.V.T.astype(np.uint8) == z["primp_seed3_Vtr"]).mean()
    assert agree_u > 0.999 and agree_v > 0.999, (agree_u, agree_v)
    assert set(np.unique(mdl.U)) <= {0.0, 1.0} and len(mdl.logs["boolean"]) == 1


def test_primp_loop_in_c_calls_takes_the_same_path_as_the_stepwise_loop(g14):
    """PRIMP's loop as one C call per iteration (bmf_primp_iterate: the objective read one iteration late, the pair of the stopping
    iteration snapshotted on the device) against the stepwise loop (a callback forces it): same number of iterations, same objective
    values, same factors -- with a tolerance that stops the run early, with one that never does, and against the reference's final
    factors (g14; PyBMF/models/PRIMP.py:96-131)."""
    from pybmf_amd.engine import BitMatrix
    from pybmf_amd.models.PRIMP import _ipalm_run
    z, meta, X = g14
    bits = BitMatrix(X, "cuda:0")
    for tag, tol in (("primp64", 1e-8), ("primp64_b0", 1e-8), ("primp64", 30.0), ("primp64_b0", 1e3)):
        g = meta[tag]
        args = (bits, z["U0"], z["V0"], 0.01, 0.0, lambda t: 1.02 ** t, g["maxiter"], tol, g["beta"])
        seen = []
        Us, Vs, fs = _ipalm_run(*args, lambda t, Uc, Vc, fn: seen.append(t))      # stepwise (a callback wants every iterate)
        Uc, Vc, fc = _ipalm_run(*args, None)                                       # one C call per iteration
        assert len(fc) == len(fs) == len(seen) and (tol < 1.0 or len(fc) < g["maxiter"]), (tag, tol, len(fc), len(fs))
        np.testing.assert_allclose(fc, fs, rtol=1e-12)
        assert np.array_equal(Uc, Us) and np.array_equal(Vc, Vs)
        if tol < 1.0:
            assert relf(Uc, z[f"{tag}_U"]) < 1e-4 and relf(Vc.T, z[f"{tag}_Vt"]) < 1e-4 and fc[-1] == pytest.approx(g["fn_final"], rel=1e-4)


def test_primp_module_functions_against_reference_fixtures(g14, monkeypatch):
    """The module-level surface of PyBMF/models/PRIMP.py:51-160 under its own names and signatures: single steps on the HIP path
    against the reference's `elbmf_step_ipalm` outputs (g14 `pstep0..2`), `primp()` end to end, the element-wise helpers."""
    import importlib
    P = importlib.import_module("pybmf_amd.models.PRIMP")   # (the package attribute of that name is the class)
    z, meta, X = g14
    Vt0 = np.ascontiguousarray(z["V0"].T)
    for i, p in enumerate(meta["primp_steps"]):
        Un = P.elbmf_step_ipalm(X, z["U0"], Vt0, z["U_prev"], p["l1reg"], p["l2reg"], p["tau"], p["beta"])
        assert relf(Un, z[f"pstep{i}_U"]) < 1e-5, (i, relf(Un, z[f"pstep{i}_U"]))
        assert Un.max() <= 1.0   # (the second prox of a step clamps from above only, PRIMP.py:51-52: slightly negative entries are the reference's)
    # the other factor through the transposed call, as the reference's loop does it (PRIMP.py:115)
    Vn_t = P.elbmf_step_ipalm(X.T, z["V0"], np.ascontiguousarray(z["U0"].T), None, 0.01, 0.0, 1.0, 0.0)
    want = orc.primp_step(np.ascontiguousarray(X.T).astype(np.float64), z["V0"], np.ascontiguousarray(z["U0"].T), None, 0.01, 0.0, 1.0, 0.0)
    assert relf(Vn_t, want) < 1e-5
    # primp(): same draws as the reference's run with seed 3 -> the same rounded factors (up to cells within 1e-4 of the threshold)
    _fixture_rand(monkeypatch, z)
    Ur, Vtr = P.primp(torch.from_numpy(X.astype(np.float32)), 6, l1reg=0.01, maxiter=25, beta=1e-4, seed=3)
    assert isinstance(Ur, torch.Tensor) and tuple(Vtr.shape) == (6, 100)
    assert (Ur.numpy().astype(np.uint8) == z["primp_seed3_Ur"]).mean() > 0.999 and (Vtr.numpy().astype(np.uint8) == z["primp_seed3_Vtr"]).mean() > 0.999
    # element-wise helpers: NumPy and torch forms agree with each other and with their definitions
    x = np.linspace(-0.5, 1.5, 41)
    for fn, clip in ((P.proxelbmf, None), (P.proxelbmfnn, (0, None)), (P._proxelbmfnn, (None, 1)), (P.proxelbmfbox, (0, 1))):
        a, b = fn(x, 0.05, 0.2), fn(torch.from_numpy(x), 0.05, 0.2).numpy()
        base = np.where(x <= 0.5, x - 0.05 * np.sign(x), x - 0.05 * np.sign(x - 1) + 0.2) / 1.2
        np.testing.assert_allclose(a, b, rtol=0, atol=1e-15)
        np.testing.assert_allclose(a, base if clip is None else np.clip(base, *clip), rtol=0, atol=1e-15)
    g = P.integrality_gap_elastic(x, 0.3, 0.7)
    assert g == pytest.approx(float(P.integrality_gap_elastic(torch.from_numpy(x), 0.3, 0.7)), rel=1e-14)
    assert g == pytest.approx(np.minimum(0.3 * np.abs(x) + 0.7 * x ** 2, 0.3 * np.abs(x - 1) + 0.7 * (x - 1) ** 2).sum(), rel=1e-14)
    with pytest.raises(NotImplementedError):   # never silently binarised
        P.elbmf_step_ipalm(X * 3, z["U0"], Vt0, None, 0.01, 0.0, 1.0, 0.0)


def test_elbmf_under_a_mask_against_reference_golden(golden_dir):
    """ELBMF with a mask / weight matrix on the HIP path (bmf_palm_extrapolate + bmf_masked_pass + bmf_palm_epilogue with `den`)
    against the reference's own numbers (golden g15): module-level steps with a 0/1 mask and with real weights, beta = 0 and > 0,
    and the class with W='mask' on a csr whose stored entries (explicit zeros included) are the observed cells."""
    from scipy.sparse import csr_matrix
    from pybmf_amd.models import ELBMF
    from pybmf_amd.models.ELBMF import update_U
    z = np.load(os.path.join(golden_dir, "g15_elbmf_masked.npz"))
    meta = json.load(open(os.path.join(golden_dir, "g15_elbmf_masked.json")))
    m, n, k = (int(v) for v in z["shape"])
    X = np.unpackbits(z["X"], axis=1)[:, :n].astype(np.uint8)
    W01 = np.unpackbits(z["W01"], axis=1)[:, :n].astype(np.float64)
    Ws = {"W01": W01, "Wr": z["Wr"]}
    for i, p in enumerate(meta["steps"]):
        W = Ws[p["W"]]
        Un, Ul = update_U(X, z["U0"], z["V0"], W, p["reg_l1"], p["reg_l2"], p["beta"], z["U_prev"])
        Vn, _ = update_U(np.ascontiguousarray(X.T), z["V0"], z["U0"], np.ascontiguousarray(W.T), p["reg_l1"], p["reg_l2"], p["beta"], z["V0"])
        assert relf(Un, z[f"mstep{i}_U"]) < 1e-5 and relf(Vn, z[f"mstep{i}_V"]) < 1e-5, (i, relf(Un, z[f"mstep{i}_U"]), relf(Vn, z[f"mstep{i}_V"]))
        assert np.array_equal(Ul, z["U0"])
    r, c = np.nonzero(W01)
    Xs = csr_matrix((X[r, c].astype(np.float64), (r, c)), shape=X.shape)
    for tag in ("mpalm", "mipalm"):
        g = meta[tag]
        with quiet():
            mdl = ELBMF(k=k, U=z["U0"].copy(), V=z["V0"].copy(), W="mask", init_method="custom", reg_l1=0.01, reg_l2=0.02, reg_growth=1.05,
                        beta=g["beta"], max_iter=g["max_iter"], min_diff=1e-8, tol=0.0)
            mdl.fit(Xs, **FIT)
        want = np.array(g["updates"]["rows"], dtype=np.float64)
        got = np.array([[float(v) for v in row[1:]] for row in mdl.logs["updates"].values.tolist()])
        assert got.shape == want.shape
        np.testing.assert_allclose(got[:, :7], want[:, :7], rtol=1e-4)
        np.testing.assert_allclose(got[:, 7:], want[:, 7:], rtol=1e-12, atol=1e-15)      # scores from exact integer counts
        assert relf(mdl.U, z[f"{tag}_U"]) < 1e-4 and relf(mdl.V, z[f"{tag}_V"]) < 1e-4
        assert list(mdl.counts[-1]) == g["counts"]


def test_elbmf_mid_size_against_oracle():
    """Another shape (k > 32, ragged), beta > 0, a few iterations: factors, error, gaps and counts against the oracle."""
    from pybmf_amd import _lib as L
    from pybmf_amd.engine import BitMatrix
    from pybmf_amd.palm import PalmEngine
    rs = np.random.RandomState(9)
    m, n, k = 700, 437, 40
    X = ((rs.rand(m, 8) < 0.2).astype(int) @ (rs.rand(8, n) < 0.2).astype(int) > 0).astype(np.uint8)
    U0, V0 = rs.rand(m, k) * 0.4, rs.rand(n, k) * 0.4
    res = orc.elbmf_fit(X, U0, V0, None, reg_l1=0.02, reg_l2=0.05, reg_growth=1.1, beta=0.15, max_iter=6, min_diff=0.0, reassoc=True)
    eng = PalmEngine(BitMatrix(X, "cuda:0"), k, L.PALM_ELBMF, beta=0.15)
    eng.load_factors(U0, V0)
    for t, want in enumerate(res["updates"]):
        l1, l2 = 0.02, 0.05 * 1.1 ** t
        eng.step("U", l1, l2, l1, l2)
        eng.step("V", l1, l2, l1, l2)
        eng.refresh("U")
        eng.refresh("V")
        err, gu, gv, cnt = eng.scalars()
        assert err == pytest.approx(want[6], rel=1e-4) and gu == pytest.approx(want[4], rel=1e-4) and gv == pytest.approx(want[5], rel=1e-4)
        assert cnt == tuple(res["counts"][t])
    U, V = eng.factors()
    assert relf(U, res["U"]) < 1e-4 and relf(V, res["V"]) < 1e-4


@pytest.mark.parametrize("m,n,k,beta", [(700, 437, 40, 0.15), (1300, 520, 64, 0.0), (300, 200, 7, 0.0)])
def test_elbmf_one_call_iteration_matches_the_stepwise_engine(m, n, k, beta):
    """bmf_palm_iterate (one C call per iteration, scalars read back late) against the step / refresh / scalars calls it replaces, and
    the factors a loop returns when it has run one iteration past its stopping rule (the previous iterate)."""
    from pybmf_amd import _lib as L
    from pybmf_amd.engine import BitMatrix
    from pybmf_amd.palm import PalmEngine
    rs = np.random.RandomState(19)
    X = ((rs.rand(m, 8) < 0.2).astype(int) @ (rs.rand(8, n) < 0.2).astype(int) > 0).astype(np.uint8)
    U0, V0 = rs.rand(m, k) * 0.4, rs.rand(n, k) * 0.4
    B = BitMatrix(X, "cuda:0")
    a, b = PalmEngine(B, k, L.PALM_ELBMF, beta=beta), PalmEngine(B, k, L.PALM_ELBMF, beta=beta)
    a.load_factors(U0, V0)
    b.load_factors(U0, V0)
    T = 7
    sched = lambda t: (0.02, 0.05 * 1.1 ** t)   # noqa: E731
    want, before_last = [], None
    for t in range(T):
        l1, l2 = sched(t)
        if t == T - 1:
            before_last = a.factors()
        a.step("U", l1, l2, l1, l2)
        a.step("V", l1, l2, l1, l2)
        a.refresh("U")
        a.refresh("V")
        want.append(a.scalars())
    for t in range(T):   # two iterations outstanding, as the model class drives it
        b.iterate(t, *sched(t), *sched(t))
        if t >= 1:
            got = b.row(t - 1)
            assert got[3] == want[t - 1][3]
            np.testing.assert_allclose(got[:3], want[t - 1][:3], rtol=1e-6)
    got = b.row(T - 1)
    assert got[3] == want[T - 1][3]
    np.testing.assert_allclose(got[:3], want[T - 1][:3], rtol=1e-6)
    Ua, Va = a.factors()
    Ub, Vb = b.factors()
    assert relf(Ub, Ua) < 1e-6 and relf(Vb, Va) < 1e-6
    Up, Vp = b.previous_factors()
    assert relf(Up, before_last[0]) < 1e-6 and relf(Vp, before_last[1]) < 1e-6


def test_elbmf_class_loops_agree(monkeypatch):
    """ELBMF.fit through the one-call loop (default) and through the stepwise Python loop: same number of iterations, same log, same factors."""
    from pybmf_amd.models import ELBMF
    rs = np.random.RandomState(23)
    m, n, k = 600, 380, 12
    X = ((rs.rand(m, 6) < 0.25).astype(int) @ (rs.rand(6, n) < 0.25).astype(int) > 0).astype(np.uint8)
    U0, V0 = rs.rand(m, k) * 0.5, rs.rand(n, k) * 0.5
    out = {}
    for loop in ("c", "python"):
        monkeypatch.setenv("BMF_PALM_LOOP", loop)
        with quiet():
            mdl = ELBMF(k=k, U=U0.copy(), V=V0.copy(), W="full", init_method="custom", reg_l1=0.01, reg_l2=0.02, reg_growth=1.3, beta=0.0,
                        max_iter=200, min_diff=1e-3, tol=0.0)
            mdl.fit(X, **FIT)
        log = np.array([[float(v) for v in r[1:]] for r in mdl.logs["updates"].values.tolist()])
        out[loop] = (mdl.n_iter, log, mdl.U, mdl.V, list(mdl.counts[-1]))
    assert out["c"][0] == out["python"][0] and 3 < out["c"][0] < 200
    np.testing.assert_allclose(out["c"][1], out["python"][1], rtol=1e-6, atol=1e-12)
    assert relf(out["c"][2], out["python"][2]) < 1e-6 and relf(out["c"][3], out["python"][3]) < 1e-6
    assert out["c"][4] == out["python"][4]
