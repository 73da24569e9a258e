"""Real-valued (not 0 / 1) training data on the models whose GPU path was Boolean-only until round 5 -- reference golden g19
(the reference casts whatever it is given to float64 and runs: PyBMF/models/ContinuousModel.py:188-203).

BinaryMFPenalty under W='full' / 'mask' / a weight matrix, PNLPF under W='full', WNMF with the Kullback-Leibler loss under W='full'
(models/WNMF.py:111-129), BinaryMFThreshold under W='full': log tables (incl. the reference's arithmetic "confusion" metrics on a
real-valued ground truth, utils/metrics.py:56-135) and factors against the reference, gate 1e-4; the confusion kernel against NumPy."""
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
def g19(golden_dir):
    return np.load(os.path.join(golden_dir, "g19_real_valued.npz")), json.load(open(os.path.join(golden_dir, "g19_real_valued.json")))


def test_real_confusion_kernel_against_numpy():
    import ctypes as C
    from pybmf_amd import _lib as L
    from pybmf_amd.device_ops import _bits_of
    from pybmf_amd.engine import RealMatrix
    rs = np.random.RandomState(2)
    m, n, k = 301, 517, 40
    X = rs.rand(m, n) * 2.5 * (rs.rand(m, n) < 0.4)
    Ub, Vb = rs.rand(m, k) < 0.08, rs.rand(n, k) < 0.08
    R = RealMatrix(X, "cuda:0")
    rb_u, _, _ = _bits_of(Ub, R.m_pad)
    rb_v, _, _ = _bits_of(Vb, R.n_pad)
    out = torch.zeros(6, dtype=torch.float64, device="cuda:0")
    ub, vb = torch.from_numpy(rb_u).cuda(), torch.from_numpy(rb_v).cuda()   # (kept alive: a temporary's block would be reused by the next upload)
    L.check(L.lib.bmf_real_confusion(L.ptr(R.X), R.n_pad, m, n, L.ptr(ub), L.ptr(vb), L.ptr(out), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    pd = (Ub.astype(np.int64) @ Vb.T.astype(np.int64)) > 0
    want = orc.real_confusion(X.astype(np.float32).astype(np.float64), pd)
    np.testing.assert_allclose(out.cpu().numpy(), np.array(want), rtol=1e-12)
    assert want[5] > 100 and abs(want[0] + want[2] - want[4]) > 1.0   # values above 1: TP + FN is not sum gt -- the arithmetic form matters


@pytest.mark.parametrize("tag", ["x01", "x3"])
def test_penalty_and_pnlpf_on_real_valued_data(g19, tag):
    from pybmf_amd.models import PNLPF, BinaryMFPenalty
    z, meta = g19
    X = z["X01"] if tag == "x01" else z["X3"]
    pen = meta["params"]["penalty"]
    with quiet():
        p = BinaryMFPenalty(W="full", **pen)
        p.fit(X.copy(), **FIT)
    ref = meta[f"pen_{tag}"]
    np.testing.assert_allclose(frame_values(p.logs["updates"]), np.array(ref["updates"]["rows"]), rtol=1e-4)
    np.testing.assert_allclose(frame_values(p.logs["boolean"]), np.array(ref["boolean"]["rows"]), rtol=1e-4)
    assert relf(p.U, z[f"pen_{tag}_U"]) < 1e-4 and relf(p.V, z[f"pen_{tag}_V"]) < 1e-4
    assert float(p.reg) == pytest.approx(ref["final_reg"], rel=1e-15)
    # evaluate() afterwards: the same arithmetic metrics from the host-side factors
    with quiet():
        p.evaluate(df_name="after", metrics=["Recall", "Precision", "Accuracy", "F1", "RMSE", "MAE"])
    row = frame_values(p.logs["after"])[-1]
    np.testing.assert_allclose(row[:4], np.array(ref["boolean"]["rows"])[-1], rtol=1e-4)
    np.testing.assert_allclose(row[4:6], np.array(ref["updates"]["rows"])[-1][5:7], rtol=1e-4)
    with quiet():
        q = PNLPF(W="full", link_lamda=meta["params"]["link_lamda"], **pen)
        q.fit(X.copy(), **FIT)
    ref = meta[f"pnlpf_{tag}"]
    np.testing.assert_allclose(frame_values(q.logs["updates"]), np.array(ref["updates"]["rows"]), rtol=1e-4)
    np.testing.assert_allclose(frame_values(q.logs["boolean"]), np.array(ref["boolean"]["rows"]), rtol=1e-4)
    assert relf(q.U, z[f"pnlpf_{tag}_U"]) < 1e-4 and relf(q.V, z[f"pnlpf_{tag}_V"]) < 1e-4


@pytest.mark.parametrize("tag", ["x01", "x3"])
def test_wnmf_kl_on_real_valued_data(g19, tag):
    from pybmf_amd.models import WNMF
    z, meta = g19
    X = z["X01"] if tag == "x01" else z["X3"]
    with quiet():
        w = WNMF(k=meta["params"]["penalty"]["k"], W="full", beta_loss="kullback-leibler", init_method="normal", max_iter=8, seed=7)
        w.fit(X.copy(), **FIT)
    np.testing.assert_allclose(frame_values(w.logs["updates"]), np.array(meta[f"kl_{tag}"]["updates"]["rows"]), rtol=1e-4)
    assert relf(w.U, z[f"kl_{tag}_U"]) < 1e-4 and relf(w.V, z[f"kl_{tag}_V"]) < 1e-4


@pytest.mark.parametrize("tag", ["x01", "x3"])
def test_threshold_on_real_valued_data(g19, tag):
    from pybmf_amd.models import BinaryMFThreshold
    z, meta = g19
    X = z["X01"] if tag == "x01" else z["X3"]
    th, ref = meta["params"]["threshold"], meta[f"thr_{tag}"]
    with quiet():
        t = BinaryMFThreshold(k=meta["params"]["penalty"]["k"], U=z[f"pen_{tag}_U"].copy(), V=z[f"pen_{tag}_V"].copy(), W="full", u=th["u"], v=th["v"],
                              lamda=th["lamda"], min_diff=th["min_diff"], max_iter=th["max_iter"])
        t.fit(X.copy(), **FIT)
    assert t.F([th["u"], th["v"]]) == pytest.approx(ref["F0"], rel=1e-6)
    np.testing.assert_allclose(t.dF([th["u"], th["v"]]), z[f"thr_{tag}_dF0"], rtol=1e-4, atol=1e-4 * np.abs(z[f"thr_{tag}_dF0"]).max())
    rows, want = frame_values(t.logs["updates"]), np.array(ref["rows"]["rows"])
    assert len(rows) == len(want)
    np.testing.assert_allclose(rows[:, :4], want[:, :4], rtol=1e-5, atol=1e-8)      # iter, u, v, F: the reference's search path
    np.testing.assert_allclose(rows[:, 4:], want[:, 4:], rtol=1e-4)                   # Recall, Precision, Accuracy, F1 (arithmetic form)
    assert t.u == pytest.approx(ref["u"], abs=1e-6) and t.v == pytest.approx(ref["v"], abs=1e-6)


def test_penalty_on_real_valued_data_under_masks(g19):
    from pybmf_amd.models import BinaryMFPenalty
    z, meta = g19
    X = z["X01"]
    pen = meta["params"]["penalty"]
    Xm = csr_matrix((X[z["mask_rows"], z["mask_cols"]], (z["mask_rows"], z["mask_cols"])), shape=X.shape)   # explicit zeros stay stored
    assert Xm.nnz == len(z["mask_rows"])
    for tag, kw, data in (("pen_mask", dict(W="mask"), Xm), ("pen_wgt", dict(W=z["Wr"]), X)):
        with quiet():
            p = BinaryMFPenalty(**kw, **pen)
            p.fit(data.copy(), **FIT)
        np.testing.assert_allclose(frame_values(p.logs["updates"]), np.array(meta[tag]["updates"]["rows"]), rtol=1e-4)
        np.testing.assert_allclose(frame_values(p.logs["boolean"]), np.array(meta[tag]["boolean"]["rows"]), rtol=1e-4)
        assert relf(p.U, z[f"{tag}_U"]) < 1e-4 and relf(p.V, z[f"{tag}_V"]) < 1e-4


@pytest.mark.parametrize("m,n,k,seed", [(300, 200, 40, 1), (130, 257, 17, 2), (64, 33, 64, 3)])
def test_real_valued_fits_against_the_oracle_on_other_shapes(m, n, k, seed):
    """Beyond the golden shapes (k = 5): a padded rank of 64, ragged shapes, values above 1 -- against the oracle's restatement
    (pinned by g19 on the CPU), which draws the same initial factors from the same seed."""
    from pybmf_amd.models import PNLPF, WNMF, BinaryMFPenalty
    rs = np.random.RandomState(seed)
    X = (rs.rand(m, 6) * (rs.rand(m, 6) < 0.5)) @ (rs.rand(6, n) * (rs.rand(6, n) < 0.5)) * 1.7
    X[X < 0.1] = 0.0
    kw = dict(k=k, reg=0.7, reg_growth=1.2, init_method="normal", normalize_method="balance", max_iter=5, seed=seed)
    ref = orc.penalty_fit(X, **kw)
    with quiet():
        p = BinaryMFPenalty(W="full", **kw)
        p.fit(X.copy(), **FIT)
    np.testing.assert_allclose(frame_values(p.logs["updates"]), np.array(ref["updates"]), rtol=1e-4)
    np.testing.assert_allclose(frame_values(p.logs["boolean"]), np.array(ref["boolean"]), rtol=1e-4, atol=1e-9)
    assert relf(p.U, ref["U"]) < 1e-4 and relf(p.V, ref["V"]) < 1e-4
    refq = orc.pnlpf_fit(X, link_lamda=8, **kw)
    with quiet():
        q = PNLPF(W="full", link_lamda=8, **kw)
        q.fit(X.copy(), **FIT)
    np.testing.assert_allclose(frame_values(q.logs["updates"]), np.array(refq["updates"]), rtol=2e-4)
    assert relf(q.U, refq["U"]) < 2e-4 and relf(q.V, refq["V"]) < 2e-4
    refw = orc.wnmf_kl_fit(X, k=k, max_iter=5, init_method="normal", seed=seed)
    with quiet():
        w = WNMF(k=k, W="full", beta_loss="kullback-leibler", init_method="normal", max_iter=5, seed=seed)
        w.fit(X.copy(), **FIT)
    np.testing.assert_allclose(frame_values(w.logs["updates"]), np.array(refw["updates"]), rtol=1e-4)
    assert relf(w.U, refw["U"]) < 1e-4 and relf(w.V, refw["V"]) < 1e-4


def test_what_is_still_refused_says_so(g19):
    """No silent binarisation, no silent fallback: the combinations without a GPU path raise with their reason."""
    from pybmf_amd.models import ELBMF, PNLPF, BinaryMFPenalty
    z, meta = g19
    X = z["X01"]
    pen = meta["params"]["penalty"]
    with quiet():
        with pytest.raises(NotImplementedError, match="link"):
            PNLPF(W=z["Wr"], link_lamda=10, **pen).fit(X.copy(), **FIT)
        big = BinaryMFPenalty(W="full", **pen)
        big.MAX_REAL_CELLS = 100
        with pytest.raises(NotImplementedError, match="cell-list kernels take up to 100 cells"):
            big.fit(X.copy(), **FIT)
        with pytest.raises(NotImplementedError, match="Boolean"):
            ELBMF(k=5, U=z["pen_x01_U0"].copy(), V=z["pen_x01_V0"].copy(), W="full", init_method="custom").fit(X.copy(), **FIT)
