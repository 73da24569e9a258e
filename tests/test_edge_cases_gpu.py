"""Edge cases of the hot path on the GPU: degenerate / ragged inputs, extreme k, the other BASELINE.json configurations
at their full sizes (through properties where the oracle would be too slow)."""
import contextlib
import io

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
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(np.asarray(b)), 1e-300)


def run_engine(X, U0, V0, regs, mode=None, terms=3, with_mae=True, **kw):
    from pybmf_amd import _lib as L
    from pybmf_amd.engine import BitMatrix, MUEngine
    eng = MUEngine(BitMatrix(X, "cuda:0"), k=U0.shape[1], mode=L.MODE_PENALTY if mode is None else mode, terms=terms,
                   with_mae=with_mae, max_iter=len(regs) + 1, tol=-1.0, **kw)
    eng.load_factors(U0, V0)
    eng.prepare(regs[0])
    eng.run(regs, it0=1)
    log, stop = eng.read_log()
    U, V = eng.factors()
    return L, log, U, V


def oracle_run(X, U0, V0, reg, growth, iters):
    return orc.penalty_fit(X, k=U0.shape[1], U=U0, V=V0, reg=reg, reg_growth=growth, init_method="custom",
                           normalize_method=None, max_iter=iters - 1, tol=-1.0, literal=False)


@pytest.mark.parametrize("panel", ["bf16", "f16", "i8"])
@pytest.mark.parametrize("m,n,k", [(1, 1, 1), (3, 70, 2), (33, 31, 1), (65, 129, 7), (200, 40, 32), (130, 260, 33), (90, 50, 64)])
def test_ragged_shapes_and_extreme_k(m, n, k, panel):
    rs = np.random.RandomState(m * 1000 + n + k)
    X = (rs.rand(m, n) < 0.4).astype(np.uint8)
    U0 = np.abs(rs.standard_normal((m, k))) * 0.3 + 1e-3
    V0 = np.abs(rs.standard_normal((n, k))) * 0.3 + 1e-3
    regs = [0.5 * 1.2 ** i for i in range(5)]
    L, log, U, V = run_engine(X, U0, V0, regs, panel=panel)
    ref = oracle_run(X, U0, V0, 0.5, 1.2, 5)
    assert relf(U, ref["U"]) < 1e-4 and relf(V, ref["V"]) < 1e-4
    want = np.array(ref["updates"])
    # rec_error / RMSE / MAE are differences against X evaluated from fp32-accurate products: when the fit is almost
    # exact (tiny matrices) they lose relative accuracy to cancellation, hence the absolute floor scaled by sum(X)
    np.testing.assert_allclose(log[:, [L.LOG_ERROR, L.LOG_REC, L.LOG_REGERR, L.LOG_RMSE, L.LOG_MAE]], want[:, [1, 2, 4, 5, 6]],
                               rtol=1e-4, atol=1e-5 * max(1.0, float(X.sum())))
    assert [tuple(int(v) for v in r[L.LOG_TP:L.LOG_TN + 1]) for r in log] == [tuple(c) for c in ref["counts"]]


@pytest.mark.parametrize("panel", ["bf16", "f16", "i8"])
def test_all_zero_and_all_one_matrices(panel):
    rs = np.random.RandomState(1)
    for fill in (0, 1):
        X = np.full((70, 45), fill, dtype=np.uint8)
        U0, V0 = rs.rand(70, 4) + 0.01, rs.rand(45, 4) + 0.01
        regs = [1.0] * 4
        L, log, U, V = run_engine(X, U0, V0, regs, panel=panel)
        ref = oracle_run(X, U0, V0, 1.0, 1.0, 4)
        assert np.isfinite(log[:, :7]).all()
        if fill == 0:
            # X = 0: both numerators vanish; with reg > 0 the factors shrink but stay positive (never exactly 0: eps clamp)
            assert (U > 0).all() and (V > 0).all()
        assert relf(U, ref["U"]) < 1e-4 and relf(V, ref["V"]) < 1e-4
        assert tuple(int(v) for v in log[-1, L.LOG_TP:L.LOG_TN + 1]) == tuple(ref["counts"][-1])
        assert log[-1, L.LOG_TP] + log[-1, L.LOG_FN] == X.sum()


@pytest.mark.parametrize("panel", ["bf16", "f16", "i8"])
def test_empty_rows_columns_and_zero_factor_entries(panel):
    """Rows/columns of X without a single one, and exact zeros in the initial factors (the solver turns them into eps
    before the loop, models/ContinuousModel.py:33-36; the engine must keep eps-sized entries alive like the reference)."""
    rs = np.random.RandomState(2)
    X = (rs.rand(150, 90) < 0.3).astype(np.uint8)
    X[10:20] = 0
    X[:, 5:9] = 0
    U0, V0 = rs.rand(150, 6) * 0.5, rs.rand(90, 6) * 0.5
    U0[:, 2] = 0.0
    V0[7] = 0.0
    U0, V0 = orc.zeros_to_eps(U0), orc.zeros_to_eps(V0)
    regs = [0.0, 0.0, 1.0, 2.0]
    L, log, U, V = run_engine(X, U0, V0, regs, panel=panel)
    ref_u, ref_v = U0.copy(), V0.copy()
    for r in regs:
        ref_v = orc.penalty_update_V_reassoc(X.astype(np.float64), ref_u, ref_v, r)
        ref_u = orc.penalty_update_U_reassoc(X.astype(np.float64), ref_u, ref_v, r)
    assert relf(U, ref_u) < 1e-4 and relf(V, ref_v) < 1e-4
    assert np.isfinite(U).all() and (U > 0).all() and (V > 0).all()


def test_threshold_bits_at_the_boundary():
    """binarize is a strict '>' on the fp64 factor (utils/common.py:64-79): entries exactly equal to 0.5, and one ulp
    either side, must land on the reference's side."""
    from pybmf_amd.device_ops import boolean_product_csr
    U = np.array([[0.5, np.nextafter(0.5, 1)], [np.nextafter(0.5, 0), 0.7]])
    V = np.array([[0.9, 0.0], [0.0, 0.9], [0.5, 0.5]])
    got = np.asarray(boolean_product_csr(U, V, u=0.5, v=0.5).todense())
    assert np.array_equal(got, orc.boolean_product(U, V, 0.5, 0.5))
    got = np.asarray(boolean_product_csr(U, V, us=[0.4, 0.6], vs=[0.8, 0.95]).todense())
    assert np.array_equal(got, orc.boolean_product(U, V, us=[0.4, 0.6], vs=[0.8, 0.95]))


def test_config2_wnmf_20k_x_5k_k32_real():
    """BASELINE configs[1]: WNMF multiplicative update, 20k x 5k dense fp32, k = 32 (SURVEY 8d recipe), full size, against
    the re-associated fp64 oracle for 6 iterations."""
    from pybmf_amd.models import WNMF
    rs = np.random.RandomState(0)
    m, n, k = 20000, 5000, 32
    X = ((rs.rand(m, 32) @ rs.rand(32, n)) / 32).astype(np.float32) + (0.01 * rs.rand(m, n)).astype(np.float32)
    with quiet():
        model = WNMF(k=k, W="full", init_method="normal", seed=2024, max_iter=5)
        model.fit(X, **FIT)
    X64 = X.astype(np.float64)
    U, V = orc.init_factors(X64, k, "normal", np.random.RandomState(2024))
    errs = [0.5 * ((X64 - U @ V.T) ** 2).sum()]
    for _ in range(6):
        V = V * ((X64.T @ U) / (V @ (U.T @ U)))
        U = U * ((X64 @ V) / (U @ (V.T @ V)))
        errs.append(0.5 * ((X64 - U @ V.T) ** 2).sum())
    got = np.array([[float(v) for v in row[1:]] for row in model.logs["updates"].values.tolist()])
    assert got.shape[0] == 7
    np.testing.assert_allclose(got[:, 1], errs, rtol=1e-4)
    assert relf(model.U, U) < 1e-4 and relf(model.V, V) < 1e-4
    cells = float(m) * n
    assert got[-1, 2] == pytest.approx(np.sqrt(2 * errs[-1] / cells), rel=1e-4)
    assert got[-1, 3] == pytest.approx(np.abs(X64 - U @ V.T).sum() / cells, rel=1e-4)


def test_config5_threshold_movielens_shape():
    """BASELINE configs[4] at MovieLens-1M shape (6040 x 3706, k = 16) on a shape/density-matched stand-in (the dataset needs
    a download): F and dF at several (u, v) against the oracle, then a full line search that must decrease F."""
    from pybmf_amd.models import BinaryMFThreshold
    rs = np.random.RandomState(11)
    m, n, k = 6040, 3706, 16
    pu, pv = rs.pareto(1.2, m) + 1, rs.pareto(1.2, n) + 1
    P = np.outer(pu / pu.sum(), pv / pv.sum())
    # cell probabilities min(s P, 1) with s iterated until they SUM to the data set's 1 000 209 ones (bench.py::secondary_c5: clipping
    # the popular rows / columns at 1 loses mass -- with s = 1 000 209 the stand-in has 0.54 M ones, half the density of the config)
    s = 1_000_209.0
    for _ in range(60):
        s *= 1_000_209.0 / np.minimum(P * s, 1.0).sum()
    X = (rs.rand(m, n) < np.minimum(P * s, 1.0)).astype(np.uint8)
    assert abs(int(X.sum()) - 1_000_209) < 5_000
    res = orc.wnmf_fit(X.astype(np.float64), k=k, max_iter=20, init_method="normal", seed=5)
    U, V = res["U"], res["V"]
    with quiet():
        model = BinaryMFThreshold(k=k, U=U.copy(), V=V.copy(), W="full", u=0.3, v=0.3, lamda=10, min_diff=1e-3, max_iter=30)
        model.fit(X, **FIT)
    Xf = X.astype(np.float64)
    U32, V32 = U.astype(np.float32).astype(np.float64), V.astype(np.float32).astype(np.float64)
    for (u, v) in [(0.3, 0.3), (0.12, 0.4), (model.u, model.v)]:
        assert model.F([u, v]) == pytest.approx(orc.thresh_F(Xf, None, U32, V32, u, v, 10), rel=2e-6)
        want = orc.thresh_dF(Xf, None, U32, V32, u, v, 10)
        np.testing.assert_allclose(model.dF([u, v]), want, rtol=1e-4, atol=1e-4 * np.abs(want).max())
    rows = np.array([[float(x) for x in r[1:]] for r in model.logs["updates"].values.tolist()])
    assert rows[-1, 3] < rows[0, 3] and (np.diff(rows[:, 3]) <= 1e-6 * rows[0, 3]).all()
    tp, fp, fn, tn = orc.confusion_counts(X.astype(np.int64), orc.boolean_product(U, V, model.u, model.v))
    assert rows[-1, 4:] == pytest.approx(orc.boolean_scores(tp, fp, fn, tn), rel=1e-12)


def test_config3_full_size_properties():
    """BASELINE configs[2] (100k x 20k, k = 64) is too large for the oracle; check size-independent properties after a few
    iterations: TP + FN = sum(X) exactly, all four counts sum to m*n, the trace-form rec_error equals the direct residual
    pass and NumPy fp64 on a row sample, 2- and 3-addend operands agree, the loop is deterministic run to run."""
    import ctypes as C
    from pybmf_amd import _lib as L
    from pybmf_amd.engine import BitMatrix, MUEngine
    from pybmf_amd.generators import PlantedBooleanOnDevice
    m, n, k = 100_000, 20_000, 64
    X = BitMatrix(PlantedBooleanOnDevice(m, n, k, density=(0.067, 0.067), seed=1000, noise=(0.05, 0.01), noise_seed=2000), "cuda:0")
    assert 0.05 < X.sum_local / (m * n) < 0.3   # 7.5 %: 64 % of the rows of U sit in the planted blocks, 36 % are Bernoulli tail
    rng = np.random.RandomState(2024)
    avg = np.sqrt(X.sum_local / (m * n) / k)
    V0 = np.abs(avg * rng.standard_normal((n, k)))
    U0 = np.abs(avg * rng.standard_normal((m, k)))
    regs = [1.02 ** i for i in range(4)]
    outs = []
    for terms in (3, 3, 2):
        eng = MUEngine(X, k=k, mode=L.MODE_PENALTY, terms=terms, with_mae=True, max_iter=6)
        eng.load_factors(U0, V0)
        eng.prepare(regs[0])
        eng.run(regs, it0=1)
        log, stop = eng.read_log()
        outs.append((log, *eng.factors()))
    log, U, V = outs[0]
    assert stop == 0 and log.shape[0] == 5 and np.isfinite(log[:, :11]).all()
    for row in log:
        assert row[L.LOG_TP] + row[L.LOG_FN] == X.sum_local
        assert row[L.LOG_TP] + row[L.LOG_FP] + row[L.LOG_FN] + row[L.LOG_TN] == float(m) * n
    assert (np.diff(log[:, L.LOG_REC]) < 0).all()          # MU decreases the reconstruction error here
    # bitwise deterministic run to run (slab sums in fixed order, integer counts); the MAE column alone comes from fp64
    # atomics of the residual pass and may differ in its last bits
    keep = [c for c in range(log.shape[1]) if c != L.LOG_MAE]
    assert np.array_equal(outs[1][0][:, keep], log[:, keep]) and np.array_equal(outs[1][1], U) and np.array_equal(outs[1][2], V)
    np.testing.assert_allclose(outs[1][0][:, L.LOG_MAE], log[:, L.LOG_MAE], rtol=1e-12)
    assert relf(outs[2][1], U) < 1e-5 and relf(outs[2][2], V) < 1e-5         # 2 vs 3 bf16 addends
    # RMSE (trace form) vs the residual pass that produced MAE: rec_error = 0.5 * sum R^2
    sums = torch.zeros(4, dtype=torch.float64, device="cuda:0")
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    L.check(L.lib.bmf_residual_sums(L.ptr(X.bits), X.m_pad, X.ldx, m, n, L.ptr(eng.U), L.ptr(eng.V), eng.kp, L.ptr(sums), None, s))
    # (eng is the last, 2-addend engine)
    assert 0.5 * float(sums[1]) == pytest.approx(outs[2][0][-1, L.LOG_REC], rel=1e-6)
    rs_ = 512
    Xs = X.rows_dense_u8(0, rs_).astype(np.float64)
    host = ((Xs - outs[2][1][:rs_] @ outs[2][2].T) ** 2).sum()
    sums.zero_()
    L.check(L.lib.bmf_residual_sums(L.ptr(X.bits), X.m_pad, X.ldx, rs_, n, L.ptr(eng.U), L.ptr(eng.V), eng.kp, L.ptr(sums), None, s))
    assert float(sums[1]) == pytest.approx(host, rel=1e-6)


def test_random_shapes_property():
    """Randomised shapes, densities, operand formats and schedules (hypothesis): three updates against the oracle, Boolean
    counts of the GPU's own factors bit-exact against NumPy."""
    from hypothesis import assume, given, settings, strategies as st, HealthCheck

    @settings(max_examples=60, deadline=None, suppress_health_check=list(HealthCheck), derandomize=True)
    @given(m=st.integers(1, 700), n=st.integers(1, 700), k=st.integers(1, 64), dens=st.floats(0.02, 0.9),
           panel=st.sampled_from(["f16", "bf16", "i8"]), reg=st.one_of(st.just(0.0), st.floats(0.01, 50.0)), seed=st.integers(0, 10_000))
    def check(m, n, k, dens, panel, reg, seed):
        rs = np.random.RandomState(seed)
        X = (rs.rand(m, n) < dens).astype(np.uint8)
        # (an all-zero X with a vanishing reg sends the factors to ~1e-88, where "relative error" compares underflow behaviour)
        assume(X.any())
        U0 = np.abs(rs.standard_normal((m, k))) * 0.3 + 1e-3
        V0 = np.abs(rs.standard_normal((n, k))) * 0.3 + 1e-3
        regs = [reg * 1.3 ** i for i in range(3)]
        L, log, U, V = run_engine(X, U0, V0, regs, panel=panel)
        ref = orc.penalty_fit(X, k=k, U=U0, V=V0, reg=reg, reg_growth=1.3, init_method="custom", normalize_method=None,
                              max_iter=2, tol=-1.0, literal=False)
        assert relf(U, ref["U"]) < 1e-4 and relf(V, ref["V"]) < 1e-4, (m, n, k, panel, relf(U, ref["U"]), relf(V, ref["V"]))
        pd = orc.boolean_product(U, V, 0.5, 0.5)
        want = orc.confusion_counts(X.astype(np.int64), pd)
        got = tuple(int(log[-1, c]) for c in (L.LOG_TP, L.LOG_FP, L.LOG_FN, L.LOG_TN))
        assert got == want, (m, n, k, got, want)

    check()


@pytest.mark.parametrize("m,n,k,panel", [(5003, 2999, 37, "f16"), (2999, 5003, 64, "bf16"), (9001, 1031, 5, "f16"), (5003, 2999, 64, "i8"), (1031, 9001, 20, "i8")])
def test_medium_odd_shapes_many_row_tiles(m, n, k, panel):
    """Several 512-row tiles with a ragged last one in both orientations, stream-K slices that start and end inside a tile, k
    that is not a multiple of anything: three updates against the oracle, counts exact."""
    rs = np.random.RandomState(m + n + k)
    X = (rs.rand(m, n) < 0.25).astype(np.uint8)
    U0 = np.abs(rs.standard_normal((m, k))) * 0.2 + 1e-3
    V0 = np.abs(rs.standard_normal((n, k))) * 0.2 + 1e-3
    regs = [0.7 * 1.2 ** i for i in range(3)]
    L, log, U, V = run_engine(X, U0, V0, regs, panel=panel)
    ref = orc.penalty_fit(X, k=k, U=U0, V=V0, reg=0.7, reg_growth=1.2, init_method="custom", normalize_method=None, max_iter=2,
                          tol=-1.0, literal=False)
    assert relf(U, ref["U"]) < 1e-5 and relf(V, ref["V"]) < 1e-5, (relf(U, ref["U"]), relf(V, ref["V"]))
    want = np.array(ref["updates"])
    got = log[:, [L.LOG_ITER, L.LOG_ERROR, L.LOG_REC, L.LOG_REG, L.LOG_REGERR, L.LOG_RMSE, L.LOG_MAE]]
    np.testing.assert_allclose(got, want, rtol=1e-5)
    pd = orc.boolean_product(U, V, 0.5, 0.5)
    assert tuple(int(log[-1, c]) for c in (L.LOG_TP, L.LOG_FP, L.LOG_FN, L.LOG_TN)) == orc.confusion_counts(X.astype(np.int64), pd)


@pytest.mark.parametrize("m,n,k", [(3001, 2003, 20), (2048, 4160, 64), (5000, 777, 33)])
def test_wnmf_real_valued_medium_shapes(m, n, k):
    """WNMF on a real-valued dense X at shapes that take the LDS-staged contraction in one orientation and the direct one in
    the other (reduction length a multiple of 64 or not), several row tiles: three updates against the oracle."""
    from pybmf_amd.models import WNMF
    rs = np.random.RandomState(m + k)
    X = (rs.rand(m, 16) @ rs.rand(16, n) / 16 + 0.01 * rs.rand(m, n)).astype(np.float32).astype(np.float64)
    U0 = np.abs(rs.standard_normal((m, k))) * 0.3 + 1e-3
    V0 = np.abs(rs.standard_normal((n, k))) * 0.3 + 1e-3
    ref = orc.wnmf_fit(X.copy(), k, U=U0.copy(), V=V0.copy(), W=None, max_iter=2, init_method="custom", tol=-1.0)
    with quiet():
        w = WNMF(k=k, U=U0.copy(), V=V0.copy(), W="full", init_method="custom", max_iter=2, tol=-1.0)
        w.fit(X.copy(), **FIT)
    assert relf(w.U, ref["U"]) < 1e-5 and relf(w.V, ref["V"]) < 1e-5, (relf(w.U, ref["U"]), relf(w.V, ref["V"]))
    rows = np.array([[float(v) for v in r[1:]] for r in w.logs["updates"].values.tolist()])
    np.testing.assert_allclose(rows[:, :4], np.array(ref["updates"])[:, :4], rtol=2e-4)
