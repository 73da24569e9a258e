"""Randomised (hypothesis, derandomised) checks of every engine behind the model classes against the oracle: ragged shapes,
k from 1 to 64, densities from almost empty to almost full, weights, unobserved rows.  Complements the golden-vector tests
(fixed shapes) -- the fixed-shape suites cannot see a padding or indexing mistake that only shows at, say, n = 1."""
import contextlib
import io

import numpy as np
import pytest
from scipy.sparse import csr_matrix

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

import oracle as orc  # noqa: E402

hyp = pytest.importorskip("hypothesis")
from hypothesis import HealthCheck, assume, given, settings, strategies as st  # noqa: E402

FIT = dict(task="reconstruction", show_logs=False, show_result=False, save_model=False)
SETTINGS = dict(deadline=None, suppress_health_check=list(HealthCheck), derandomize=True)


@pytest.fixture(scope="module", autouse=True)
def need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")


@contextlib.contextmanager
def quiet():
    with contextlib.redirect_stdout(io.StringIO()):
        yield


def relf(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(np.asarray(b)), 1e-300)


def factors(rs, m, n, k, scale=0.3):
    return np.abs(rs.standard_normal((m, k))) * scale + 1e-3, np.abs(rs.standard_normal((n, k))) * scale + 1e-3


def test_wnmf_real_valued_matrix():
    """WNMF on a real-valued dense X (config #2's path: fp32 X in HBM, LDS-staged contraction), all-ones mask."""
    from pybmf_amd.models import WNMF

    @settings(max_examples=60, **SETTINGS)
    @given(m=st.integers(2, 400), n=st.integers(2, 400), k=st.integers(1, 64), seed=st.integers(0, 10_000))
    def check(m, n, k, seed):
        rs = np.random.RandomState(seed)
        X = (rs.rand(m, max(k, 2)) @ rs.rand(max(k, 2), n) / max(k, 2) + 0.01 * rs.rand(m, n)).astype(np.float32).astype(np.float64)
        U0, V0 = factors(rs, m, n, k)
        # tol = -1: an exact fit (a 1 x n matrix with k = 1 after one update) has error 0.0 in fp64 and ~1e-8 on the fp32
        # contractions, so "error <= tol" with the default tol = 0 would stop the two runs at different iterations
        ref = orc.wnmf_fit(X.copy(), k, U=U0.copy(), V=V0.copy(), W=None, max_iter=3, init_method="custom", tol=-1.0)
        with quiet():
            w = WNMF(k=k, U=U0.copy(), V=V0.copy(), W="full", init_method="custom", max_iter=3, tol=-1.0)
            w.fit(X.copy(), **FIT)
        assert relf(w.U, ref["U"]) < 1e-4 and relf(w.V, ref["V"]) < 1e-4, (m, n, k, relf(w.U, ref["U"]), relf(w.V, ref["V"]))
        rows = np.array([[float(v) for v in r[1:]] for r in w.logs["updates"].values.tolist()])
        want = np.array(ref["updates"])[:, :4]
        scale = np.array([1.0, float((X ** 2).sum()), float(np.sqrt((X ** 2).mean())), float(np.abs(X).mean())])
        assert rows.shape == want.shape
        # (the error is a difference of O(|X|^2) terms evaluated from fp32 contractions: an absolute floor of ~1e-7 |X|^2, which
        # RMSE = sqrt(error / cells) turns into ~3e-4 rms(X) when the fit is nearly exact)
        assert (np.abs(rows[:, :4] - want) <= 2e-4 * np.abs(want) + np.array([0, 2e-6, 1e-3, 1e-4]) * scale).all(), (rows, want)

    check()


def test_masked_updates_with_weights():
    """BinaryMFPenalty and WNMF under a weight matrix with unobserved cells / rows (SDDMM + SpMM path)."""
    from pybmf_amd.models import BinaryMFPenalty, WNMF

    @settings(max_examples=60, **SETTINGS)
    @given(m=st.integers(1, 300), n=st.integers(1, 300), k=st.integers(1, 64), dens=st.floats(0.05, 0.9), obs=st.floats(0.05, 1.0),
           reg=st.sampled_from([0.0, 0.05, 1.0, 10.0]), seed=st.integers(0, 10_000))
    def check(m, n, k, dens, obs, reg, seed):
        rs = np.random.RandomState(seed)
        X = (rs.rand(m, n) < dens).astype(np.float64)
        W = (rs.rand(m, n) < obs) * rs.choice([1.0, 0.5, 3.0], size=(m, n))
        if m > 2:
            W[rs.randint(m), :] = 0.0
        # an all-zero observed matrix is the one place where the reference's in-place "0 -> eps" on X_train (WNMF.py:136-139)
        # decides the result (the factors of an all-eps matrix); everywhere else it is a 1e-16 perturbation
        assume((W * X).sum() > 0)
        U0, V0 = factors(rs, m, n, k)
        ref = orc.penalty_fit(X, k=k, U=U0.copy(), V=V0.copy(), reg=reg, reg_growth=1.3, init_method="custom", normalize_method=None,
                              max_iter=2, tol=-1.0, W=W)
        with quiet():
            mdl = BinaryMFPenalty(k=k, U=U0.copy(), V=V0.copy(), W=csr_matrix(W), reg=reg, reg_growth=1.3, init_method="custom",
                                  normalize_method=None, max_iter=2, tol=-1.0)
            mdl.fit(X.copy(), **FIT)
        assert relf(mdl.U, ref["U"]) < 1e-4 and relf(mdl.V, ref["V"]) < 1e-4, (m, n, k, relf(mdl.U, ref["U"]), relf(mdl.V, ref["V"]))
        refw = orc.wnmf_fit(X.copy(), k, U=U0.copy(), V=V0.copy(), W=W, max_iter=2, init_method="custom")
        with quiet():
            w = WNMF(k=k, U=U0.copy(), V=V0.copy(), W=W.copy(), init_method="custom", max_iter=2)
            w.fit(X.copy(), **FIT)
        # rows / columns whose observed cells are all zero are where the reference's in-place "0 -> eps" on X_train decides the
        # factors (WNMF.py docstring here): compare the others, and the predictions on every observed cell
        live_r, live_c = (W * X).sum(1) > 0, (W * X).sum(0) > 0
        assert relf(w.U[live_r], refw["U"][live_r]) < 1e-4 and relf(w.V[live_c], refw["V"][live_c]) < 1e-4, (m, n, k)
        P, Pr = (w.U @ w.V.T)[W != 0], (refw["U"] @ refw["V"].T)[W != 0]
        assert np.abs(P - Pr).max() <= 1e-4 * max(1.0, np.abs(Pr).max()), (m, n, k)

    check()


def test_threshold_objective_and_gradient():
    """F and dF of BinaryMFThreshold at random thresholds, all-ones mask and the stored-entries mask."""
    from pybmf_amd.models import BinaryMFThreshold

    @settings(max_examples=60, **SETTINGS)
    @given(m=st.integers(1, 300), n=st.integers(1, 300), k=st.integers(1, 64), dens=st.floats(0.05, 0.9), u=st.floats(0.05, 0.95),
           v=st.floats(0.05, 0.95), lam=st.sampled_from([1.0, 10.0, 100.0]), masked=st.booleans(), seed=st.integers(0, 10_000))
    def check(m, n, k, dens, u, v, lam, masked, seed):
        rs = np.random.RandomState(seed)
        X = (rs.rand(m, n) < dens).astype(np.float64)
        U0, V0 = rs.rand(m, k), rs.rand(n, k)
        W = None
        Xin = X
        if masked:
            W = (rs.rand(m, n) < 0.6).astype(np.float64)
            if W.sum() == 0:
                W[0, 0] = 1.0
            r, c = np.nonzero(W)
            Xin = csr_matrix((X[r, c], (r, c)), shape=(m, n))  # explicit zeros stay stored: they are observed cells
        with quiet():
            mdl = BinaryMFThreshold(k=k, U=U0.copy(), V=V0.copy(), W="mask" if masked else "full", u=0.5, v=0.5, lamda=lam, max_iter=1)
            mdl.fit(Xin, **FIT)
        wantF = orc.thresh_F(X, W, U0, V0, u, v, lam)
        assert mdl.F([u, v]) == pytest.approx(wantF, rel=2e-4, abs=1e-6), (m, n, k, masked)
        wantG = orc.thresh_dF(X, W, U0, V0, u, v, lam)
        np.testing.assert_allclose(mdl.dF([u, v]), wantG, rtol=2e-3, atol=2e-4 * max(1.0, float(np.abs(wantG).max())))

    check()


def test_link_models():
    """PNLPF and WNMF with the Kullback-Leibler loss (tile-fused passes through an element-wise link)."""
    from pybmf_amd.models import PNLPF, WNMF

    @settings(max_examples=40, **SETTINGS)
    @given(m=st.integers(1, 300), n=st.integers(1, 300), k=st.integers(1, 64), dens=st.floats(0.05, 0.9), seed=st.integers(0, 10_000))
    def check(m, n, k, dens, seed):
        rs = np.random.RandomState(seed)
        X = (rs.rand(m, n) < dens).astype(np.float64)
        assume(X.sum() > 0)
        U0, V0 = factors(rs, m, n, k)
        ref = orc.pnlpf_fit(X, k=k, U=U0.copy(), V=V0.copy(), reg=1.0, link_lamda=10, reg_growth=1.2, init_method="custom",
                            normalize_method=None, max_iter=2, tol=-1.0)
        with quiet():
            p = PNLPF(k=k, U=U0.copy(), V=V0.copy(), W="full", reg=1.0, link_lamda=10, reg_growth=1.2, init_method="custom",
                      normalize_method=None, max_iter=2, tol=-1.0)
            p.fit(X.copy(), **FIT)
        assert relf(p.U, ref["U"]) < 1e-4 and relf(p.V, ref["V"]) < 1e-4, (m, n, k, relf(p.U, ref["U"]), relf(p.V, ref["V"]))
        refk = orc.wnmf_kl_fit(X.copy(), k, U=U0.copy(), V=V0.copy(), W=None, max_iter=2, init_method="custom")
        with quiet():
            w = WNMF(k=k, U=U0.copy(), V=V0.copy(), W="full", beta_loss="kullback-leibler", init_method="custom", max_iter=2)
            w.fit(X.copy(), **FIT)
        assert relf(w.U, refk["U"]) < 1e-4 and relf(w.V, refk["V"]) < 1e-4, (m, n, k, relf(w.U, refk["U"]), relf(w.V, refk["V"]))

    check()


def test_cover_score_api():
    """TP / FP / FN / TN with and without `axis`, coverage_score, description_length on bit matrices of any shape."""
    from pybmf_amd import utils as u

    @settings(max_examples=60, **SETTINGS)
    @given(m=st.integers(1, 500), n=st.integers(1, 500), k=st.integers(1, 64), dg=st.floats(0.0, 1.0), dp=st.floats(0.0, 1.0),
           seed=st.integers(0, 10_000))
    def check(m, n, k, dg, dp, seed):
        rs = np.random.RandomState(seed)
        G = csr_matrix((rs.rand(m, n) < dg).astype(np.int64))
        Ub, Vb = (rs.rand(m, k) < dp * 0.3).astype(np.int64), (rs.rand(n, k) < 0.3).astype(np.int64)
        P = csr_matrix(orc.boolean_product(Ub, Vb))
        Gd, Pd = G.toarray(), P.toarray()
        for ax in (None, 0, 1):
            want = orc.confusion_counts_axis(Gd, Pd, axis=ax)
            got = (u.TP(G, P, axis=ax), u.FP(G, P, axis=ax), u.FN(G, P, axis=ax), u.TN(G, P, axis=ax))
            for a, b in zip(got, want):
                np.testing.assert_array_equal(np.asarray(a).ravel(), np.asarray(b).ravel())
        assert u.coverage_score(G, P, w_fp=0.3) == pytest.approx(orc.coverage_score(Gd, Pd, w_fp=0.3), rel=1e-15)
        assert u.description_length(G, csr_matrix(Ub), csr_matrix(Vb)) == orc.description_length(Gd, Ub, Vb)

    check()


def test_prediction_task_scores():
    """fit(X_train, X_val, X_test, task='prediction') on negative-sampled csr splits of any shape: the val / test columns of the
    last log row against the oracle's entry scorer evaluated on the fitted factors (non-zero cells of each set, as the
    reference's densified data sets have it)."""
    from pybmf_amd.models import BinaryMFPenalty

    @settings(max_examples=30, **SETTINGS)
    @given(m=st.integers(2, 250), n=st.integers(2, 250), k=st.integers(1, 64), dens=st.floats(0.1, 0.8), seed=st.integers(0, 10_000))
    def check(m, n, k, dens, seed):
        rs = np.random.RandomState(seed)
        X = (rs.rand(m, n) < dens).astype(np.float64)
        part = rs.randint(0, 4, size=(m, n))            # 0, 1: train, 2: val, 3: test -- every cell observed in one set
        sets = {}
        for nm, sel in (("train", part < 2), ("val", part == 2), ("test", part == 3)):
            r, c = np.nonzero(sel)
            sets[nm] = csr_matrix((X[r, c], (r, c)), shape=(m, n))   # explicit zeros stay stored
        assume(all(s.nnz > 0 and s.data.sum() > 0 for s in sets.values()))
        U0, V0 = factors(rs, m, n, k)
        with quiet():
            mdl = BinaryMFPenalty(k=k, U=U0.copy(), V=V0.copy(), W="mask", reg=0.5, reg_growth=1.3, init_method="custom",
                                  normalize_method=None, max_iter=2, tol=-1.0)
            mdl.fit(sets["train"].copy(), sets["val"].copy(), sets["test"].copy(), task="prediction", show_logs=False, show_result=False,
                    save_model=False)
        ucols = [tuple(str(x) for x in c) for c in mdl.logs["updates"].columns]
        bcols = [tuple(str(x) for x in c) for c in mdl.logs["boolean"].columns]
        urow, brow = mdl.logs["updates"].values.tolist()[-1], mdl.logs["boolean"].values.tolist()[-1]
        for nm in ("train", "val", "test"):
            coo = sets[nm].tocoo()
            keep = coo.data != 0
            r, c, vals = coo.row[keep], coo.col[keep], coo.data[keep]
            rmse, mae = orc.entry_scores(r, c, vals, mdl.U, mdl.V)
            assert float(urow[ucols.index((nm, "0", "RMSE"))]) == pytest.approx(rmse, rel=1e-4, abs=1e-7)
            assert float(urow[ucols.index((nm, "0", "MAE"))]) == pytest.approx(mae, rel=1e-4, abs=1e-7)
            want = orc.boolean_scores(*orc.entry_scores(r, c, vals, mdl.U, mdl.V, 0.5, 0.5))
            got = [float(brow[bcols.index((nm, "0", mt))]) for mt in ("Recall", "Precision", "Accuracy", "F1")]
            # (an entry within 1e-7 of the threshold may fall on the other side in fp32; none does at these sizes)
            np.testing.assert_allclose(got, want, rtol=1e-12)

    check()


def test_bit_matrix_from_any_container():
    """BitMatrix (X and X^T as bits in HBM) from every container fit() accepts, contiguous or not, any shape, any row shard."""
    from scipy.sparse import coo_matrix, csc_matrix
    from pybmf_amd.engine import BitMatrix

    @settings(max_examples=60, **SETTINGS)
    @given(m=st.integers(1, 600), n=st.integers(1, 600), dens=st.floats(0.0, 1.0), kind=st.integers(0, 7), seed=st.integers(0, 10_000),
           cut=st.floats(0.0, 1.0))
    def check(m, n, dens, kind, seed, cut):
        rs = np.random.RandomState(seed)
        D = (rs.rand(m, n) < dens)
        want = D.astype(np.uint8)
        X = [lambda: D, lambda: D.astype(np.float64) * 2.5, lambda: np.asfortranarray(D.astype(np.int32)),
             lambda: np.repeat(D, 2, axis=1)[:, ::2].astype(np.float32),           # a strided view
             lambda: csr_matrix(D.astype(np.float64)), lambda: csc_matrix(D.astype(np.int8)), lambda: coo_matrix(D.astype(np.float32)).tocsr(),
             lambda: torch.from_numpy(D.astype(np.float32)).t().contiguous().t()][kind]()   # a transposed torch view
        lo = int(cut * m) // 2
        hi = max(lo, m - int((1.0 - cut) * m) // 3)
        B = BitMatrix(X, "cuda:0", row_lo=lo, row_hi=hi, chunk_rows=128)
        assert (B.m, B.n, B.sum_local) == (hi - lo, n, int(want[lo:hi].sum()))
        np.testing.assert_array_equal(B.to_dense_u8(), want[lo:hi])
        bt = B.bits_t[:n].cpu().numpy().view(np.uint8)
        np.testing.assert_array_equal(np.unpackbits(bt, axis=1, bitorder="little")[:, : hi - lo], want[lo:hi].T)
        assert not B.bits[B.m:].any() and not B.bits_t[n:].any()   # the padding stays zero

    check()
