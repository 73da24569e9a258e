"""the cover-score API of utils/metrics.py on the GPU, bit-exact against the reference golden g11."""
import json
import os

import numpy as np
import pytest
from scipy.sparse import csr_matrix

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

import oracle as orc  # noqa: E402


@pytest.fixture(scope="module", autouse=True)
def need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")


def cases(golden_dir):
    for c in json.load(open(os.path.join(golden_dir, "g11_cover_scores.json"))):
        m, n, k = c["shape"]
        gt = np.unpackbits(np.array(c["gt"], dtype=np.uint8), axis=1)[:, :n].astype(np.int64)
        U, V = np.array(c["U"]), np.array(c["V"])
        yield c, gt, U, V, np.minimum(U @ V.T, 1)


def test_counts_and_scores_match_reference(golden_dir):
    from pybmf_amd import utils as u
    for c, gt, U, V, pd in cases(golden_dir):
        G, P = csr_matrix(gt), csr_matrix(pd)
        for ax in (None, 0, 1):
            ref = c["all" if ax is None else f"axis{ax}"]
            for nm in ("TP", "FP", "FN", "TN"):
                assert np.array_equal(np.asarray(getattr(u, nm)(G, P, axis=ax), dtype=float), np.asarray(ref[nm])), (nm, ax)
            np.testing.assert_allclose(np.asarray(u.ACC(G, P, axis=ax), dtype=float), ref["ACC"], rtol=1e-15)
            np.testing.assert_allclose(u.coverage_score(G, P, axis=ax), ref["coverage_score_0.5"], rtol=1e-15)
            np.testing.assert_allclose(u.coverage_score(gt, pd, w_fp=0.3, axis=ax), ref["coverage_score_0.3"], rtol=1e-15)
            np.testing.assert_allclose(u.weighted_error(G, P, w_fp=0.2, w_fn=0.7, axis=ax), ref["weighted_error_0.2_0.7"], rtol=1e-15)
        assert u.description_length(G, csr_matrix(U), csr_matrix(V)) == c["description_length"]      # product never materialised
        assert u.description_length(G, U, V, pd=P, w_model=0.5, w_fp=2.0, w_fn=3.0) == c["description_length_w"]
        tp, fp, fn, tn = u.confusion(G, P)
        r, p_, a, f1 = orc.boolean_scores(tp, fp, fn, tn)
        assert u.get_metrics(G, P, ["Recall", "Precision", "Accuracy", "F1", "TP"]) == [r, p_, a, f1, tp]


def test_large_random_against_numpy():
    from pybmf_amd import utils as u
    rs = np.random.RandomState(4)
    gt = (rs.rand(3000, 5000) < 0.1).astype(np.uint8)
    pd = (rs.rand(3000, 5000) < 0.2).astype(np.uint8)
    for ax in (None, 0, 1):
        want = orc.confusion_counts_axis(gt, pd, ax)
        got = u.confusion(gt, pd, ax)
        for w, g in zip(want, got):
            assert np.array_equal(np.asarray(w), np.asarray(g))
