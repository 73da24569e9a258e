"""updates through an element-wise link -- PNLPF (sigmoid) and WNMF with the Kullback-Leibler loss.
Kernel parity against NumPy fp64 / the oracle, model parity against the reference golden g10."""
import contextlib
import ctypes as C
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


def frame_values(df):
    return np.array([[float(v) for v in row[1:]] for row in df.values.tolist()])


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


@pytest.fixture(scope="module")
def g10(golden_dir):
    z = np.load(os.path.join(golden_dir, "g10_link_models.npz"))
    meta = json.load(open(os.path.join(golden_dir, "g10_link_models.json")))
    m, n = z["shape"]
    X = np.unpackbits(z["X"], axis=1)[:, :n].astype(np.float64)
    return z, meta, X


@pytest.mark.parametrize("m,n,k", [(210, 150, 6), (130, 700, 40), (1, 1, 1), (515, 33, 32)])
def test_link_pass_and_sums_against_numpy(m, n, k):
    from pybmf_amd import _lib as L
    from pybmf_amd.engine import BitMatrix, LinkMUEngine
    rs = np.random.RandomState(m + n + k)
    X = (rs.rand(m, n) < 0.3).astype(np.float64)
    U = np.abs(rs.standard_normal((m, k))) * 0.4 + 1e-3
    V = np.abs(rs.standard_normal((n, k))) * 0.4 + 1e-3
    lam = 7.0
    for link, mode in ((L.LINK_SIGMOID, L.MODE_PENALTY), (L.LINK_KL, L.MODE_WNMF)):
        eng = LinkMUEngine(BitMatrix(X.astype(np.uint8), "cuda:0"), k, link, mode, lamda=lam, mfma="f32")
        eng.load_factors(U, V)
        eng.prepare()
        Uf, Vf = U.astype(np.float32).astype(np.float64), V.astype(np.float32).astype(np.float64)  # what the kernels see
        P = Uf @ Vf.T
        with torch.cuda.device(eng.device):
            Xb = eng.X
            stride = eng.m_pad * eng.kp
            L.check(L.lib.bmf_link_pass(L.ptr(Xb.bits), eng.m_pad, Xb.ldx, m, n, L.ptr(eng.U), L.ptr(eng.V), eng.n_pad, eng.kp, link, lam,
                                        L.ptr(eng.numU), L.ptr(eng.denU_slabs), stride, eng.splitsU, stream()))
            num = eng.numU.double().sum(0).cpu().numpy()[:m, :k]
            if link == L.LINK_SIGMOID:
                sig = orc.stable_sigmoid((P - 0.5) * lam)
                d = sig * (1 - sig)
                np.testing.assert_allclose(num, lam * (X * d) @ Vf, rtol=2e-5, atol=1e-6)
                den = eng.denU_slabs.double().sum(0).cpu().numpy()[:m, :k]
                np.testing.assert_allclose(den, lam * (sig * d) @ Vf, rtol=2e-5, atol=1e-6)
                want = (np.abs(X - sig).sum(), ((X - sig) ** 2).sum())
            else:
                np.testing.assert_allclose(num, (X / P) @ Vf, rtol=2e-5, atol=1e-6)
                want = (np.abs(X - P).sum(), ((X - P) ** 2).sum(), np.where(X > 0, -np.log(P) - 1 + P, P).sum())
            assert not eng.numU.sum(0)[m:].any()   # padded rows stay zero
            # the transposed orientation (V's pass)
            strideV = eng.n_pad * eng.kp
            L.check(L.lib.bmf_link_pass(L.ptr(Xb.bits_t), eng.n_pad, Xb.ldxt, n, m, L.ptr(eng.V), L.ptr(eng.U), eng.m_pad, eng.kp, link, lam,
                                        L.ptr(eng.numV), L.ptr(eng.denV_slabs), strideV, eng.splitsV, stream()))
            numV = eng.numV.double().sum(0).cpu().numpy()[:n, :k]
            ref = lam * (X * d).T @ Uf if link == L.LINK_SIGMOID else (X / P).T @ Uf
            np.testing.assert_allclose(numV, ref, rtol=2e-5, atol=1e-6)
            sums = torch.zeros(4, dtype=torch.float64, device=eng.device)
            L.check(L.lib.bmf_link_sums(L.ptr(Xb.bits), eng.m_pad, Xb.ldx, m, n, L.ptr(eng.U), L.ptr(eng.V), eng.n_pad, eng.kp, link, lam,
                                        None, L.ptr(sums), stream()))
            got = sums.cpu().numpy()
        np.testing.assert_allclose(got[: len(want)], want, rtol=2e-5)
        assert L.lib.bmf_link_pass(L.ptr(Xb.bits), eng.m_pad, Xb.ldx, m, n, L.ptr(eng.U), L.ptr(eng.V), eng.n_pad, eng.kp, 9, lam,
                                   L.ptr(eng.numU), None, stride, eng.splitsU, stream()) == -1


@pytest.mark.parametrize("m,n,k,unbalanced", [(210, 150, 6, False), (130, 700, 40, False), (515, 33, 32, False), (210, 150, 6, True), (300, 260, 64, True)])
def test_link_pass16_against_numpy(m, n, k, unbalanced):
    """The 16-bit MFMA flavour of the pass: same contractions, products right to 2^-16.  `unbalanced`: the column pair that carries most of P is
    tiny in U and huge in V (and another one the other way round) -- the fp16 hi / lo operands of P are scaled per column PAIR
    (bmf_link_split_pair), one scale per factor would leave those columns of U with 11 bits."""
    from pybmf_amd import _lib as L
    from pybmf_amd.engine import BitMatrix, LinkMUEngine
    rs = np.random.RandomState(m + n + k)
    X = (rs.rand(m, n) < 0.3).astype(np.float64)
    U = np.abs(rs.standard_normal((m, k))) * 0.4 + 1e-3
    V = np.abs(rs.standard_normal((n, k))) * 0.4 + 1e-3
    if unbalanced:
        U[:, 0] *= 3e-5; V[:, 0] *= 1e5      # U V^T is dominated by this pair (x 3)
        U[:, k - 1] *= 2e4; V[:, k - 1] *= 4e-5
        U[:, 1] = 0.0                         # and a dead column
    lam = 7.0
    for link, mode in ((L.LINK_SIGMOID, L.MODE_PENALTY), (L.LINK_KL, L.MODE_WNMF)):
        eng = LinkMUEngine(BitMatrix(X.astype(np.uint8), "cuda:0"), k, link, mode, lamda=lam, mfma="bf16")
        eng.load_factors(U, V)
        eng.prepare()
        Uf, Vf = U.astype(np.float32).astype(np.float64), V.astype(np.float32).astype(np.float64)
        P = Uf @ Vf.T
        with torch.cuda.device(eng.device):
            Xb = eng.X
            L.check(L.lib.bmf_link_pass16(L.ptr(Xb.bits), eng.m_pad, Xb.ldx, m, n, L.ptr(eng.wsU), L.ptr(eng.wsV), eng.n_pad, eng.kp, link,
                                          lam, L.ptr(eng.numU), L.ptr(eng.denU_slabs), eng.m_pad * eng.kp, eng.splitsU, stream()))
            L.check(L.lib.bmf_link_pass16(L.ptr(Xb.bits_t), eng.n_pad, Xb.ldxt, n, m, L.ptr(eng.wsV), L.ptr(eng.wsU), eng.m_pad, eng.kp, link,
                                          lam, L.ptr(eng.numV), L.ptr(eng.denV_slabs), eng.n_pad * eng.kp, eng.splitsV, stream()))
            numU = eng.numU.double().sum(0).cpu().numpy()[:m, :k]
            numV = eng.numV.double().sum(0).cpu().numpy()[:n, :k]
            assert not eng.numU.sum(0)[m:].any()
        if link == L.LINK_SIGMOID:
            sig = orc.stable_sigmoid((P - 0.5) * lam)
            d = sig * (1 - sig)
            wantU, wantV = lam * (X * d) @ Vf, lam * (X * d).T @ Uf
            denU = eng.denU_slabs.double().sum(0).cpu().numpy()[:m, :k]
            assert relf(denU, lam * (sig * d) @ Vf) < 2e-5
        else:
            wantU, wantV = (X / P) @ Vf, (X / P).T @ Uf
        assert relf(numU, wantU) < 2e-5 and relf(numV, wantV) < 2e-5, (relf(numU, wantU), relf(numV, wantV))


def test_pnlpf_matches_reference(g10):
    from pybmf_amd.models import PNLPF
    z, meta, X = g10
    p = meta["pnlpf"]["params"]
    with quiet():
        mdl = PNLPF(k=p["k"], U=z["p_U0"].copy(), V=z["p_V0"].copy(), W="full", reg=p["reg"], reg_growth=p["reg_growth"],
                    link_lamda=p["link_lamda"], init_method="custom", normalize_method=None, max_iter=p["max_iter"])
        mdl.fit(X.copy(), **FIT)
    assert relf(mdl.U, z["p_U"]) < 1e-4 and relf(mdl.V, z["p_V"]) < 1e-4
    print(f"PNLPF drift after {p['max_iter'] + 1} updates: U {relf(mdl.U, z['p_U']):.2e} V {relf(mdl.V, z['p_V']):.2e}")
    np.testing.assert_allclose(frame_values(mdl.logs["updates"]), np.array(meta["pnlpf"]["updates"]["rows"]), rtol=1e-4)
    np.testing.assert_allclose(frame_values(mdl.logs["boolean"]), np.array(meta["pnlpf"]["boolean"]["rows"]), rtol=1e-12, atol=0)
    assert float(mdl.reg) == pytest.approx(meta["pnlpf"]["final_reg"], rel=1e-12)
    # the seeded start of the reference run is reproduced too (init 'normal' + 'balance')
    with quiet():
        m2 = PNLPF(k=p["k"], W="full", reg=p["reg"], reg_growth=p["reg_growth"], link_lamda=p["link_lamda"], init_method="normal",
                   normalize_method="balance", max_iter=p["max_iter"], seed=5)
        m2.fit(X.copy(), **FIT)
    assert relf(m2.U, z["p_U"]) < 1e-4
    # module-level one-shot updates (PNLPF.py:61-91)
    from pybmf_amd.models.PNLPF import update_U, update_V
    V1 = update_V(X, None, z["p_U0"], z["p_V0"], 1.0, 10)
    assert relf(V1, orc.pnlpf_update_V(X, None, z["p_U0"], z["p_V0"], 1.0, 10)) < 5e-6
    U1 = update_U(X, None, z["p_U0"], V1, 1.0, 10)
    assert relf(U1, orc.pnlpf_update_U(X, None, z["p_U0"], V1, 1.0, 10)) < 5e-6


def test_pnlpf_under_a_mask_matches_reference(golden_dir):
    """PNLPF with W='mask' on a csr with explicit zeros -- the contractions of both updates over the observed cells
    (bmf_masked_link_pass), rec_error over them, scores whole-matrix -- and the module-level steps with a real weight matrix,
    against the reference's numbers (golden g16)."""
    from scipy.sparse import csr_matrix
    from pybmf_amd.models import PNLPF
    from pybmf_amd.models.PNLPF import update_U, update_V
    z = np.load(os.path.join(golden_dir, "g16_pnlpf_masked.npz"))
    meta = json.load(open(os.path.join(golden_dir, "g16_pnlpf_masked.json")))
    m, n = (int(v) for v in z["shape"])
    Xs = csr_matrix((z["vals"].astype(np.float64), (z["rows"], z["cols"])), shape=(m, n))
    p = meta["pnlpf"]["params"]
    with quiet():
        mdl = PNLPF(k=p["k"], U=z["p_U0"].copy(), V=z["p_V0"].copy(), W="mask", reg=p["reg"], reg_growth=p["reg_growth"],
                    link_lamda=p["link_lamda"], init_method="custom", normalize_method=None, max_iter=p["max_iter"])
        mdl.fit(Xs, **FIT)
    assert relf(mdl.U, z["p_U"]) < 1e-4 and relf(mdl.V, z["p_V"]) < 1e-4, (relf(mdl.U, z["p_U"]), relf(mdl.V, z["p_V"]))
    np.testing.assert_allclose(frame_values(mdl.logs["updates"]), np.array(meta["pnlpf"]["updates"]["rows"]), rtol=1e-4)
    np.testing.assert_allclose(frame_values(mdl.logs["boolean"]), np.array(meta["pnlpf"]["boolean"]["rows"]), rtol=1e-12, atol=0)
    Xd = np.unpackbits(z["Xd"], axis=1)[:, :n].astype(np.uint8)
    for i, st in enumerate(meta["steps"]):
        Vn = update_V(Xd, z["Wr"], z["s_U"], z["s_V"], st["reg"], st["link_lamda"])
        Un = update_U(Xd, z["Wr"], z["s_U"], z[f"step{i}_V"], st["reg"], st["link_lamda"])
        assert relf(Vn, z[f"step{i}_V"]) < 1e-5 and relf(Un, z[f"step{i}_U"]) < 1e-5, (i, relf(Vn, z[f"step{i}_V"]), relf(Un, z[f"step{i}_U"]))


def test_wnmf_kl_matches_reference(g10):
    from pybmf_amd.models import WNMF
    z, meta, X = g10
    with quiet():
        w = WNMF(k=6, U=z["w_U0"].copy(), V=z["w_V0"].copy(), W="full", beta_loss="kullback-leibler", init_method="custom", max_iter=9)
        w.fit(X.copy(), **FIT)
    print(f"WNMF-KL drift: U {relf(w.U, z['w_U']):.2e} V {relf(w.V, z['w_V']):.2e}")
    assert relf(w.U, z["w_U"]) < 1e-4 and relf(w.V, z["w_V"]) < 1e-4
    np.testing.assert_allclose(frame_values(w.logs["updates"]), np.array(meta["wnmf_kl"]["updates"]["rows"]), rtol=1e-4)
    with quiet():
        w2 = WNMF(k=6, W="full", beta_loss="kullback-leibler", init_method="normal", max_iter=9, seed=5)
        w2.fit(X.copy(), **FIT)   # the seeded start of the reference run
    assert relf(w2.U, z["w_U"]) < 1e-4


def test_link_models_refuse_what_they_do_not_cover(g10):
    from scipy.sparse import csr_matrix
    from pybmf_amd.models import PNLPF, WNMF
    z, meta, X = g10
    Xs = csr_matrix(X)   # unstored zeros -> W='mask' is a proper mask
    with quiet():
        # (a weight matrix on the Kullback-Leibler loss runs since round 4: test_wnmf_kl_with_a_weight_matrix)
        # (extra data sets on PNLPF are scored since round 4: tests/test_prediction_gpu.py::test_pnlpf_scores_val_and_test_sets)
        p = PNLPF(k=6, W="full", reg=1.0, init_method="normal", max_iter=3, seed=5)
        p.fit(X.copy(), X_val=Xs, **FIT)
        assert ("val", 0, "RMSE") in list(p.logs["updates"].columns)


def test_wnmf_kl_with_the_default_mask(golden_dir):
    """WNMF(beta_loss='kullback-leibler') with the reference's default W='mask' (reference golden g13): on a dense Boolean
    matrix and on a csr with explicit zeros.  The factors follow the all-ones-mask updates (W o X = X, denominators use the
    all-ones matrix), the objective is summed over the observed cells."""
    from scipy.sparse import csr_matrix
    from pybmf_amd.models import WNMF
    z10 = np.load(os.path.join(golden_dir, "g10_link_models.npz"))
    z = np.load(os.path.join(golden_dir, "g13_kl_mask.npz"))
    meta = json.load(open(os.path.join(golden_dir, "g13_kl_mask.json")))
    m, n = z10["shape"]
    X = np.unpackbits(z["X"], axis=1)[:, :n].astype(np.float64)
    Xs = csr_matrix((X[z["rows"], z["cols"]], (z["rows"], z["cols"])), shape=(m, n))
    assert Xs.nnz == len(z["rows"])   # explicit zeros stay stored
    for tag, data in (("dense", X), ("csr", Xs)):
        with quiet():
            w = WNMF(k=6, U=z10["w_U0"].copy(), V=z10["w_V0"].copy(), W="mask", beta_loss="kullback-leibler", init_method="custom",
                     max_iter=6)
            w.fit(data.copy(), **FIT)
        assert relf(w.U, z[tag + "_U"]) < 1e-4 and relf(w.V, z[tag + "_V"]) < 1e-4
        np.testing.assert_allclose(frame_values(w.logs["updates"]), np.array(meta[tag]["updates"]["rows"]), rtol=1e-4)


@pytest.mark.parametrize("task", ["reconstruction", "prediction"])
def test_wnmf_kl_scores_val_and_test_sets(task):
    """fit(X_train, X_val, X_test) with the Kullback-Leibler loss: the extra columns are the scores of U V^T on each set --
    whole matrix under 'reconstruction', its non-zero entries under 'prediction' (the densified data sets of the reference)."""
    from scipy.sparse import csr_matrix
    from pybmf_amd.models import WNMF
    rs = np.random.RandomState(3)
    m, n, k = 120, 90, 5
    X = (rs.rand(m, n) < 0.3).astype(np.float64)
    part = rs.randint(0, 3, size=(m, n))
    sets = {}
    for nm, sel in (("train", part == 0), ("val", part == 1), ("test", part == 2)):
        r, c = np.nonzero(sel)
        sets[nm] = csr_matrix((X[r, c], (r, c)), shape=(m, n))
    with quiet():
        w = WNMF(k=k, W="mask", beta_loss="kullback-leibler", init_method="normal", max_iter=3, seed=1)
        w.fit(sets["train"].copy(), sets["val"].copy(), sets["test"].copy(), task=task, show_logs=False, show_result=False, save_model=False)
    cols = [tuple(str(x) for x in c) for c in w.logs["updates"].columns]
    row = w.logs["updates"].values.tolist()[-1]
    for nm in ("val", "test"):
        if task == "reconstruction":
            want = orc.rmse_mae(sets[nm].toarray(), w.U @ w.V.T)
        else:
            coo = sets[nm].tocoo()
            keep = coo.data != 0
            want = orc.entry_scores(coo.row[keep], coo.col[keep], coo.data[keep], w.U, w.V)
        assert float(row[cols.index((nm, "0", "RMSE"))]) == pytest.approx(want[0], rel=1e-5)
        assert float(row[cols.index((nm, "0", "MAE"))]) == pytest.approx(want[1], rel=1e-5)


@pytest.mark.parametrize("m,n,k", [(3001, 2003, 20), (1031, 4100, 64)])
def test_link_models_medium_odd_shapes(m, n, k):
    """PNLPF and WNMF-KL at shapes with many 32-row / 32-column tiles and ragged edges, column slabs split over several
    workgroups: two updates against the oracle."""
    from pybmf_amd.models import PNLPF, WNMF
    rs = np.random.RandomState(m + n)
    X = (rs.rand(m, n) < 0.3).astype(np.float64)
    U0 = np.abs(rs.standard_normal((m, k))) * 0.3 + 1e-3
    V0 = np.abs(rs.standard_normal((n, k))) * 0.3 + 1e-3
    ref = orc.pnlpf_fit(X, k=k, U=U0.copy(), V=V0.copy(), reg=1.0, link_lamda=10, reg_growth=1.2, init_method="custom",
                        normalize_method=None, max_iter=1, tol=-1.0)
    with quiet():
        p = PNLPF(k=k, U=U0.copy(), V=V0.copy(), W="full", reg=1.0, link_lamda=10, reg_growth=1.2, init_method="custom",
                  normalize_method=None, max_iter=1, tol=-1.0)
        p.fit(X.copy(), **FIT)
    assert relf(p.U, ref["U"]) < 2e-5 and relf(p.V, ref["V"]) < 2e-5, (relf(p.U, ref["U"]), relf(p.V, ref["V"]))
    np.testing.assert_allclose(frame_values(p.logs["updates"]), np.array(ref["updates"]), rtol=1e-4)
    np.testing.assert_allclose(frame_values(p.logs["boolean"]), np.array(ref["boolean"]), rtol=1e-12)
    refk = orc.wnmf_kl_fit(X.copy(), k, U=U0.copy(), V=V0.copy(), W=None, max_iter=1, init_method="custom")
    with quiet():
        w = WNMF(k=k, U=U0.copy(), V=V0.copy(), W="full", beta_loss="kullback-leibler", init_method="custom", max_iter=1)
        w.fit(X.copy(), **FIT)
    assert relf(w.U, refk["U"]) < 2e-5 and relf(w.V, refk["V"]) < 2e-5, (relf(w.U, refk["U"]), relf(w.V, refk["V"]))
    np.testing.assert_allclose(frame_values(w.logs["updates"]), np.array(refk["updates"]), rtol=1e-4)


def test_wnmf_kl_with_a_weight_matrix(golden_dir):
    """WNMF(beta_loss='kullback-leibler') under a REAL weight matrix (reference golden g18, WNMF.py:111-129,143-145): numerators over the
    cells with W != 0 (csrc/masked.hip, BMF_LINK_KL), denominators = column sums of the other factor, objective over those cells."""
    from pybmf_amd.models import WNMF
    z10 = np.load(os.path.join(golden_dir, "g10_link_models.npz"))
    z = np.load(os.path.join(golden_dir, "g18_kl_weights.npz"))
    ref = json.load(open(os.path.join(golden_dir, "g18_kl_weights.json")))["updates"]
    m, n = (int(v) for v in z10["shape"])
    X = np.unpackbits(z10["X"], axis=1)[:, :n].astype(np.float64)
    with quiet():
        w = WNMF(k=6, U=z10["w_U0"].copy(), V=z10["w_V0"].copy(), W=z["Wr"].copy(), beta_loss="kullback-leibler", init_method="custom", max_iter=6)
        w.fit(X.copy(), **FIT)
    assert relf(w.U, z["U"]) < 1e-4 and relf(w.V, z["V"]) < 1e-4
    rows = np.array([[float(v) for v in r[1:]] for r in w.logs["updates"].values.tolist()])
    np.testing.assert_allclose(rows, np.array(ref["rows"], dtype=np.float64), rtol=1e-4)


def test_link_loops_in_c_calls_take_the_same_path_as_the_stepwise_loops(monkeypatch):
    """PNLPF and WNMF-KL enqueue whole iterations by one C call each (bmf_link_iterate) and read the scalars of iteration t while t + 1
    runs; the loop overshoots its stopping rule by one iteration and returns the iterate before.  Same kernels in the same order as the
    stepwise loop: stopping iteration and factors bit for bit, log rows to 1e-12 (their sums are fp64 atomic accumulations)."""
    from pybmf_amd.models import PNLPF, WNMF
    rs = np.random.RandomState(21)
    m, n, k = 420, 310, 10
    X = (rs.rand(m, n) < 0.25).astype(np.uint8)
    U0 = np.abs(rs.standard_normal((m, k))) * 0.3 + 1e-3
    V0 = np.abs(rs.standard_normal((n, k))) * 0.3 + 1e-3
    out = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("BMF_LINK_PIPELINE", flag)
        with quiet():
            p = PNLPF(k=k, U=U0.copy(), V=V0.copy(), W="full", reg=1.0, reg_growth=1.3, link_lamda=10, init_method="custom", normalize_method=None,
                      max_iter=12, tol=0.0)
            p.fit(X.copy(), **FIT)
            w = WNMF(k=k, U=U0.copy(), V=V0.copy(), W="full", beta_loss="kullback-leibler", init_method="custom", max_iter=60, min_diff=20.0)
            w.fit(X.copy(), **FIT)
        out[flag] = (p.U.copy(), p.V.copy(), frame_values(p.logs["updates"]), p.n_iter, float(p.reg),
                     w.U.copy(), w.V.copy(), frame_values(w.logs["updates"]), w.n_iter)
    a, b = out["1"], out["0"]
    assert a[3] == b[3] == 13 and a[4] == b[4]            # n_iter > max_iter ends the reference loop
    assert a[8] == b[8] and 2 <= a[8] <= 61               # the same stopping iteration (min_diff or max_iter)
    for i in (0, 1, 5, 6):
        np.testing.assert_array_equal(np.asarray(a[i]), np.asarray(b[i]))
    for i in (2, 7):
        np.testing.assert_allclose(np.asarray(a[i]), np.asarray(b[i]), rtol=1e-12, atol=0)
