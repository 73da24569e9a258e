"""Row-sharded HIP engine, rehearsed on ONE GPU: 2 (and 3) ranks share cuda:0 and exchange through `gloo` (RCCL refuses
two ranks on one device; the exchange protocol is backend-agnostic, see pybmf_amd/sharding.py).  The sharded run must
reproduce the single-engine run: same log rows, same V on every rank, U = concatenation of the shards."""
import os
import socket

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

import oracle as orc  # noqa: E402


def same_run(a, b):
    """Two runs of the same kernels and the same all-reduces: factors and log bit for bit -- except the MAE column, whose sum is
    accumulated with fp64 atomics in arrival order (the only floating-point atomics of the loop).  Raises with what differs."""
    from pybmf_amd import _lib as L
    cols = [c for c in range(a["log"].shape[1]) if c != L.LOG_MAE]
    for name, x, y in (("U", a["U"], b["U"]), ("V", a["V"], b["V"]), ("log", a["log"][:, cols], b["log"][:, cols])):
        if not np.array_equal(x, y):
            bad = np.argwhere(x != y)
            raise AssertionError(f"{name} differs in {len(bad)} entries, first at {bad[0].tolist()}: {x[tuple(bad[0])]!r} vs {y[tuple(bad[0])]!r}; "
                                 f"max rel {np.max(np.abs(x - y) / np.maximum(np.abs(y), 1e-300)):.3e}")
    np.testing.assert_allclose(a["log"][:, L.LOG_MAE], b["log"][:, L.LOG_MAE], rtol=1e-12, atol=0.0)
    return True


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def worker(rank, world, port, X, U0, V0, regs, out_dir, panel, blocked=True, loop="c", backend="gloo", overlap=None):
    import torch.distributed as dist
    # the scalar part of a step under the numerator's all-reduce (csrc/api.hip::exchange_phase): the library decides by shard size;
    # these small problems force it on or off (read once per process, before the first step)
    if overlap is None:
        os.environ.pop("BMF_EXCHANGE_OVERLAP", None)
    else:
        os.environ["BMF_EXCHANGE_OVERLAP"] = "1" if overlap else "0"
    # the two-block X^T U exchange is chosen from measured all-reduce times (RCCL only); these small problems force it on or off
    if blocked is None:   # decided from measured all-reduce / GEMM times (RCCL communicators only)
        os.environ.pop("BMF_XTU_BLOCKS", None)
    else:
        os.environ["BMF_XTU_BLOCKS"] = "2" if blocked else "1"
    # "c": the loop and its collectives are enqueued by bmf_penalty_run_sharded (over gloo: through the host-callback communicator);
    # "python": the host-driven reference protocol, sharding.ExchangeLoop
    os.environ["BMF_SHARDED_LOOP"] = "c" if loop == "refused" else loop
    from pybmf_amd import _lib as L
    from pybmf_amd.engine import BitMatrix, MUEngine, shard_rows
    if loop == "refused":   # RCCL refuses the communicator on this rank: the ranks must agree on the host-driven protocol
        import pybmf_amd.engine as E
        E.lib.bmf_comm_create = lambda *a: -1
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    if backend == "nccl":
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lo, hi = shard_rows(X.shape[0], rank, world)
        B = BitMatrix(X, "cuda:0", row_lo=lo, row_hi=hi)
        eng = MUEngine(B, k=U0.shape[1], mode=L.MODE_PENALTY, terms=3, with_mae=True, max_iter=len(regs) + 1, sharded=True, panel=panel)
        assert (eng._comm is not None) == (loop == "c")
        if loop == "refused":
            assert "c_loop_refused" in eng.exchange_plan and eng.exchange_plan["loop"].startswith("python")
        eng.load_factors(U0[lo:hi], V0)
        eng.prepare(regs[0])
        eng.comm_timing(True)
        eng.run(regs, it0=1, **({"poll_every": 3} if loop == "c" else {}))
        timing = eng.comm_timing(False)
        log, stop = eng.read_log()
        U, V = eng.factors()
        if blocked is None:
            assert eng.exchange_plan["decided_by"] == "measured" and eng.exchange_plan["allreduce_numerator_ms"] >= 0.0, eng.exchange_plan
        else:
            assert eng.n_blocks() == (2 if blocked and panel == "i8" and eng.kp == 64 else 1)
        assert timing["steps_timed"] == len(regs) and timing["exposed_comm_ms_per_step"] >= 0.0
        assert "all-reduce" in eng.exchange_description() and eng.exchange_plan["xtu_blocks"] == eng.n_blocks()
        import ctypes as C
        overlaps = L.lib.bmf_exchange_overlaps(C.byref(eng.st), eng._comm) if eng._comm else -1
        np.savez(os.path.join(out_dir, f"r{rank}{loop}.npz"), U=U, V=V, log=log, stop=stop, overlaps=overlaps, m_pad=B.m_pad)
        eng.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,panel,m,k,blocked,overlap", [(2, "bf16", 1500, 12, True, None), (3, "f16", 1500, 12, True, None), (2, "f16", 97, 12, True, None),
                                                              (3, "f16", 40, 12, True, None), (2, "i8", 1500, 40, True, None), (3, "i8", 1100, 64, True, None),
                                                              (2, "i8", 700, 12, True, None), (2, "i8", 1100, 64, False, None),
                                                              (3, "i8", 1100, 64, True, True), (2, "i8", 1100, 64, False, True), (2, "f16", 900, 12, True, True)])
def test_sharded_engine_matches_single(tmp_path, world, panel, m, k, blocked, overlap):
    """(m = 40 on three ranks: shards of 32, 8 and 0 rows -- refused by every rank together.  int8 panels with k > 32: X^T U goes
    out in two 32-column blocks, block-major exchange buffer.)"""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import torch.multiprocessing as mp
    from pybmf_amd import _lib as L
    from pybmf_amd.engine import BitMatrix, MUEngine
    X, _, _, _ = orc.synthetic_boolean(m, 700, 12, (0.15, 0.15), seed=41)
    X = orc.flip_noise(X, (0.05, 0.01), seed=42).astype(np.uint8)
    U0, V0 = orc.init_factors(X, k, "normal", np.random.RandomState(8))
    U0, V0 = orc.balance_factors(U0, V0)
    U0, V0 = orc.zeros_to_eps(U0), orc.zeros_to_eps(V0)
    regs = [1.0 * 1.05 ** i for i in range(8)]

    eng = MUEngine(BitMatrix(X, "cuda:0"), k=k, mode=L.MODE_PENALTY, terms=3, with_mae=True, max_iter=len(regs) + 1, panel=panel)
    eng.load_factors(U0, V0)
    eng.prepare(regs[0])
    eng.run(regs, it0=1)
    log1, _ = eng.read_log()
    U1, V1 = eng.factors()

    if m < 32 * (world - 1) + 1:   # some rank would hold no rows: every rank refuses together
        with pytest.raises(Exception, match="would hold no rows"):
            mp.spawn(worker, args=(world, free_port(), X, U0, V0, regs, str(tmp_path), panel, blocked), nprocs=world, join=True)
        return
    for loop in ("c", "python"):   # (overlap: the C loop with the scalar part of a step under the numerator's all-reduce, still bit for bit the protocol)
        mp.spawn(worker, args=(world, free_port(), X, U0, V0, regs, str(tmp_path), panel, blocked, loop, "gloo", overlap), nprocs=world, join=True)
    parts = [np.load(os.path.join(tmp_path, f"r{r}c.npz")) for r in range(world)]
    # the C-side loop (bmf_penalty_run_sharded) and the host-driven reference protocol issue the same kernels and the same
    # all-reduces: bitwise-equal factors and logs
    for r in range(world):
        q = np.load(os.path.join(tmp_path, f"r{r}python.npz"))
        assert same_run(q, parts[r])
    U = np.concatenate([p["U"] for p in parts])
    rel = lambda a, b: np.linalg.norm(a - b) / np.linalg.norm(b)  # noqa: E731
    assert rel(U, U1) < 2e-6
    for p in parts:
        assert rel(p["V"], V1) < 2e-6
        assert np.array_equal(p["V"], parts[0]["V"])  # replicated state is bitwise identical across ranks
        np.testing.assert_allclose(p["log"][:, :7], log1[:, :7], rtol=2e-6)
        assert np.array_equal(p["log"][:, L.LOG_TP:L.LOG_TN + 1], log1[:, L.LOG_TP:L.LOG_TN + 1])  # integer counts exact
    ref = orc.penalty_fit(X, k=k, U=U0, V=V0, reg=1.0, reg_growth=1.05, init_method="custom", normalize_method=None,
                          max_iter=len(regs) - 1, tol=-1.0, literal=False)
    assert rel(U, ref["U"]) < 1e-4 and rel(parts[0]["V"], ref["V"]) < 1e-4


def test_uneven_shards_take_the_same_exchange_form(tmp_path):
    """Two ranks whose padded shard sizes straddle the threshold of the overlap decision (m = 15 380: 7 712 and 7 668 rows, padded
    8 192 and 7 680; the MAE pass is on, so the local rule would say "overlap" on rank 0 only): the decision is taken from the
    largest shard, both ranks issue the same collectives, and the run is the single-engine run."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import torch.multiprocessing as mp
    from pybmf_amd import _lib as L
    from pybmf_amd.engine import BitMatrix, MUEngine
    m, k, world = 15380, 12, 2
    X, _, _, _ = orc.synthetic_boolean(m, 300, 12, (0.15, 0.15), seed=43)
    X = orc.flip_noise(X, (0.05, 0.01), seed=44).astype(np.uint8)
    U0, V0 = orc.init_factors(X, k, "normal", np.random.RandomState(9))
    U0, V0 = orc.balance_factors(U0, V0)
    U0, V0 = orc.zeros_to_eps(U0), orc.zeros_to_eps(V0)
    regs = [1.0 * 1.05 ** i for i in range(5)]
    eng = MUEngine(BitMatrix(X, "cuda:0"), k=k, mode=L.MODE_PENALTY, terms=3, with_mae=True, max_iter=len(regs) + 1, panel="i8")
    eng.load_factors(U0, V0)
    eng.prepare(regs[0])
    eng.run(regs, it0=1)
    log1, _ = eng.read_log()
    U1, V1 = eng.factors()
    mp.spawn(worker, args=(world, free_port(), X, U0, V0, regs, str(tmp_path), "i8", False, "c", "gloo", None), nprocs=world, join=True)
    parts = [np.load(os.path.join(tmp_path, f"r{r}c.npz")) for r in range(world)]
    assert sorted(int(p["m_pad"]) for p in parts) == [7680, 8192]           # the case the advisor described
    assert int(parts[0]["overlaps"]) == int(parts[1]["overlaps"]) == 1      # ... and both ranks overlap (largest shard >= 8192, MAE on)
    rel = lambda a, b: np.linalg.norm(a - b) / np.linalg.norm(b)  # noqa: E731
    assert rel(np.concatenate([p["U"] for p in parts]), U1) < 2e-6 and rel(parts[0]["V"], V1) < 2e-6
    assert np.array_equal(parts[0]["V"], parts[1]["V"])
    # (the RMSE / MAE columns come from the residual pass, whose sums are fp64 atomics in arrival order: 1e-12, not bitwise)
    # one log on both ranks, bit for bit (but for the MAE column: fp64 atomics in arrival order) -- which needs V^T V summed in the same
    # order on both, i.e. a Gram slab count taken from the largest shard, not from the local one
    cols = [c for c in range(parts[0]["log"].shape[1]) if c != L.LOG_MAE]
    assert np.array_equal(parts[0]["log"][:, cols], parts[1]["log"][:, cols])
    np.testing.assert_allclose(parts[0]["log"][:, L.LOG_MAE], parts[1]["log"][:, L.LOG_MAE], rtol=1e-12, atol=0.0)
    assert np.array_equal(parts[0]["log"][:, L.LOG_TP:L.LOG_TN + 1], parts[1]["log"][:, L.LOG_TP:L.LOG_TN + 1])
    np.testing.assert_allclose(parts[0]["log"][:, :7], log1[:, :7], rtol=2e-6)


def test_c_loop_on_rccl_with_one_rank(tmp_path):
    """The RCCL leaf of the C-side loop: bmf_comm_create (ncclCommInitRank through the unique id that travels over the group),
    the measured blocked / unblocked decision, bmf_penalty_prepare_sharded / bmf_penalty_run_sharded with the collectives on the
    side stream -- with ONE rank, which is what a one-GPU box allows.  Same numbers as the unsharded loop, bit for bit, and as
    the host-driven protocol on torch's own RCCL communicator."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import torch.multiprocessing as mp
    from pybmf_amd import _lib as L
    from pybmf_amd.engine import BitMatrix, MUEngine
    m, k = 1300, 64
    X, _, _, _ = orc.synthetic_boolean(m, 700, 12, (0.15, 0.15), seed=43)
    X = orc.flip_noise(X, (0.05, 0.01), seed=44).astype(np.uint8)
    U0, V0 = orc.init_factors(X, k, "normal", np.random.RandomState(8))
    U0, V0 = orc.balance_factors(U0, V0)
    U0, V0 = orc.zeros_to_eps(U0), orc.zeros_to_eps(V0)
    regs = [1.0 * 1.05 ** i for i in range(8)]
    eng = MUEngine(BitMatrix(X, "cuda:0"), k=k, mode=L.MODE_PENALTY, terms=3, with_mae=True, max_iter=len(regs) + 1, panel="i8")
    eng.load_factors(U0, V0)
    eng.prepare(regs[0])
    eng.run(regs, it0=1)
    log1, _ = eng.read_log()
    U1, V1 = eng.factors()
    for blocked in (False, True, None):
        for loop in (("c",) if blocked is None else ("c", "python", "refused") if blocked else ("c", "python")):
            mp.spawn(worker, args=(1, free_port(), X, U0, V0, regs, str(tmp_path), "i8", blocked, loop, "nccl"), nprocs=1, join=True)
        c = np.load(os.path.join(tmp_path, "r0c.npz"))
        if blocked is not None:
            q = np.load(os.path.join(tmp_path, "r0python.npz"))
            assert same_run(c, q)
        if blocked:   # a refused RCCL communicator: every rank falls back to the host-driven protocol, same numbers
            assert same_run(np.load(os.path.join(tmp_path, "r0refused.npz")), q)
        if blocked is False:   # one launch of X^T U: the very kernels of the unsharded loop
            assert np.array_equal(c["U"], U1) and np.array_equal(c["V"], V1)
        rel = lambda a, b: np.linalg.norm(a - b) / np.linalg.norm(b)  # noqa: E731
        assert rel(c["U"], U1) < 2e-6 and rel(c["V"], V1) < 2e-6
        np.testing.assert_allclose(c["log"][:, :7], log1[:, :7], rtol=2e-6)


def c4_init(mean_x, m, n, k, seed):
    """init_method='normal' + normalize_method='balance' + zeros -> eps (PyBMF/models/ContinuousModel.py:66-75,117-123,33-36), on the host."""
    rng = np.random.RandomState(seed)
    avg = np.sqrt(mean_x / k)
    V = np.abs(avg * rng.standard_normal(size=(n, k)))
    U = np.abs(avg * rng.standard_normal(size=(m, k)))
    dU, dV = np.sqrt(U.max(axis=0)), np.sqrt(V.max(axis=0))
    U, V = U * dV / dU, V * dU / dV
    eps = np.finfo(np.float64).eps
    U[U == 0] = eps
    V[V == 0] = eps
    return U, V


def c4_engine(m, n, k, regs, rank, world, sharded):
    from pybmf_amd import _lib as L
    from pybmf_amd.engine import BitMatrix, MUEngine, shard_rows
    from pybmf_amd.generators import PlantedBooleanOnDevice
    dev = torch.device("cuda", 0)
    gen = PlantedBooleanOnDevice(m, n, k, density=(0.067, 0.067), seed=1000, noise=(0.05, 0.01), noise_seed=2000, device=dev)
    lo, hi = shard_rows(m, rank, world)
    B = BitMatrix(gen, dev, row_lo=lo, row_hi=hi)
    del gen
    eng = MUEngine(B, k=k, mode=L.MODE_PENALTY, terms=3, with_mae=True, max_iter=len(regs) + 2, sharded=sharded, panel="i8")
    U0, V0 = c4_init(eng.sum_x / (float(m) * n), m, n, k, seed=2024)
    eng.load_factors(U0[lo:hi], V0)
    eng.prepare(regs[0])
    eng.run(regs, it0=1)
    log, stop = eng.read_log()
    U, V = eng.factors()
    return eng, U, V, log, stop


def c4_worker(rank, world, port, m, n, k, regs, out_dir):
    import torch.distributed as dist
    for v in ("BMF_EXCHANGE_OVERLAP", "BMF_XTU_BLOCKS"):
        os.environ.pop(v, None)   # the library's own plan for this shard size
    os.environ["BMF_SHARDED_LOOP"] = "c"
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        eng, U, V, log, stop = c4_engine(m, n, k, regs, rank, world, sharded=True)
        assert eng._comm is not None and eng.exchange_plan["loop"].startswith("C")
        # one more sharded update from this state, for the oracle check of the step itself (the caller recomputes sampled rows /
        # columns of it in fp64 from the state before the step)
        reg_x = regs[-1] * 1.02
        eng.run([reg_x], it0=len(regs) + 1)
        U_next, V_next = eng.factors()
        np.savez(os.path.join(out_dir, f"c4r{rank}.npz"), U=U, V=V, log=log, stop=stop, U_next=U_next, V_next=V_next, reg_x=reg_x)
        eng.close()
    finally:
        dist.destroy_process_group()


def test_config4_full_size_four_row_shards_on_one_gpu(tmp_path):
    """BASELINE config #4 (100 000 x 20 000 Boolean, k = 64, row-sharded; reference loop PyBMF/models/BinaryMFPenalty.py:81-115) at
    FULL size, as far as a one-GPU box allows: four ranks share cuda:0, each holds ~25 000 rows, and the C-side sharded loop
    (bmf_penalty_run_sharded) exchanges through the host-callback communicator over gloo.  Against the unsharded loop on the same
    matrix: U (concatenated shards) and V to 2e-6, every scalar of the log to 2e-6, the cover counts exactly, V bit for bit the same
    on all ranks.  (The unsharded loop is tied to the fp64 oracle at this size by tests/test_c3_parity_gpu.py.)"""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import torch.multiprocessing as mp
    from pybmf_amd import _lib as L
    m, n, k, world = 100_000, 20_000, 64, 4
    regs = [1.0 * 1.02 ** i for i in range(6)]
    eng, U1, V1, log1, stop1 = c4_engine(m, n, k, regs, 0, 1, sharded=False)
    assert stop1 == 0 and log1.shape[0] == len(regs) + 1
    # rows / columns of X for the oracle check of the sharded step below (the whole matrix lives only as bits on the device)
    from c3_lockstep import unpack_cols, unpack_rows
    rs = np.random.RandomState(7)
    I, J = np.sort(rs.choice(m, size=1024, replace=False)), np.sort(rs.choice(n, size=256, replace=False))
    XI, XJ = unpack_rows(eng.X, I), unpack_cols(eng.X, J)
    del eng
    torch.cuda.empty_cache()
    mp.spawn(c4_worker, args=(world, free_port(), m, n, k, regs, str(tmp_path)), nprocs=world, join=True)
    parts = [np.load(os.path.join(tmp_path, f"c4r{r}.npz")) for r in range(world)]
    assert [p["U"].shape[0] for p in parts] == [25_024, 24_992, 24_992, 24_992]   # shard_rows: boundaries on multiples of 32 rows
    U = np.concatenate([p["U"] for p in parts])
    rel = lambda a, b: np.linalg.norm(a - b) / np.linalg.norm(b)  # noqa: E731
    assert rel(U, U1) < 2e-6, rel(U, U1)
    for p in parts:
        assert int(p["stop"]) == 0
        assert rel(p["V"], V1) < 2e-6
        assert np.array_equal(p["V"], parts[0]["V"])
        assert np.array_equal(p["log"][:, [c for c in range(p["log"].shape[1]) if c != L.LOG_MAE]],
                              parts[0]["log"][:, [c for c in range(p["log"].shape[1]) if c != L.LOG_MAE]])   # one log, on every rank
        np.testing.assert_allclose(p["log"][:, :7], log1[:, :7], rtol=2e-6)
        assert np.array_equal(p["log"][:, L.LOG_TP:L.LOG_TN + 1], log1[:, L.LOG_TP:L.LOG_TN + 1])  # integer counts exact
    # The sharded step against the fp64 oracle DIRECTLY (not through the unsharded loop): one more update of the four-rank run,
    # recomputed on sampled columns (V: a sum over ALL ranks' rows -- the exchange is in it) and sampled rows (U) from the state before it
    # (reference: update_V / update_U, PyBMF/models/BinaryMFPenalty.py:136-163, re-associated form).
    reg_x = float(parts[0]["reg_x"])
    U_next, V_next = np.concatenate([p["U_next"] for p in parts]), parts[0]["V_next"]
    Vs = orc.penalty_update_V_reassoc(XJ, U, parts[0]["V"][J], reg_x)
    Us = orc.penalty_update_U_reassoc(XI, U[I], V_next, reg_x)
    assert rel(V_next[J], Vs) < 1e-4 and rel(U_next[I], Us) < 1e-4, (rel(V_next[J], Vs), rel(U_next[I], Us))
    print(f"[c4 oracle step] sharded update vs fp64 oracle on {len(J)} columns / {len(I)} rows: rel V {rel(V_next[J], Vs):.2e}, rel U {rel(U_next[I], Us):.2e}")


def masked_inputs(X):
    """A csr with explicit zeros and unstored cells (W='mask'), and a weight matrix with zeros."""
    from scipy.sparse import csr_matrix
    rs = np.random.RandomState(9)
    keep = (rs.rand(*X.shape) < 0.5) | (X != 0)
    r, c = np.nonzero(keep)
    Xs = csr_matrix((X[r, c].astype(np.float64), (r, c)), shape=X.shape)
    Wm = (rs.rand(*X.shape) < 0.6) * rs.choice([0.5, 1.0, 2.0], size=X.shape)
    return Xs, Wm


def real_input(X):
    """A real-valued matrix of the same shape (WNMF's fp32 path)."""
    rs = np.random.RandomState(10)
    return (rs.rand(X.shape[0], 6) @ rs.rand(6, X.shape[1]) / 6 + 0.01 * rs.rand(*X.shape)).astype(np.float32).astype(np.float64)


def model_worker(rank, world, port, X, out_dir):
    """Every rank runs the SAME script (as under torchrun): the drop-in classes shard the rows themselves."""
    import contextlib
    import io
    import torch.distributed as dist
    from pybmf_amd.models import BinaryMFPenalty, PNLPF, WNMF
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        fit = dict(task="reconstruction", show_logs=False, show_result=False, save_model=False)
        with contextlib.redirect_stdout(io.StringIO()):
            p = BinaryMFPenalty(k=7, W="full", reg=1.0, reg_growth=1.1, init_method="normal", normalize_method="balance", max_iter=6, seed=3)
            p.fit(X, **fit)
            w = WNMF(k=7, W="full", init_method="normal", max_iter=5, seed=3)
            w.fit(X, **fit)
            free = WNMF(k=7, W="full", init_method="normal", max_iter=2, seed=None)   # unseeded: rank 0's draw is everyone's
            free.fit(X, **fit)
            Xs, Wm = masked_inputs(X)
            pm = BinaryMFPenalty(k=7, W="mask", reg=1.0, reg_growth=1.1, init_method="normal", normalize_method="balance", max_iter=4, seed=3)
            pm.fit(Xs, **fit)
            ww = WNMF(k=7, W=Wm, init_method="normal", max_iter=4, seed=3)
            ww.fit(X, **fit)
            pl = PNLPF(k=7, W="full", reg=1.0, reg_growth=1.1, link_lamda=10, init_method="normal", normalize_method="balance", max_iter=3, seed=3)
            pl.fit(X, **fit)
            kl = WNMF(k=7, W="mask", beta_loss="kullback-leibler", init_method="normal", max_iter=3, seed=3)
            kl.fit(Xs, **fit)
            klw = WNMF(k=7, W=Wm, beta_loss="kullback-leibler", init_method="normal", max_iter=3, seed=3)   # KL under a weight matrix (round 5)
            klw.fit(X, **fit)
            wr = WNMF(k=7, W="full", init_method="normal", max_iter=4, seed=3)
            wr.fit(real_input(X), **fit)
            wr_sums = wr._residual_sums()
            tp = p._cover_counts()
            rs = p._residual_sums()
        assert pm._sharded and ww._sharded and pm._obs.m < X.shape[0] and pl._sharded and kl._sharded and wr._sharded and klw._sharded
        assert p._sharded and w._sharded and p._bits.m < X.shape[0]
        np.savez(os.path.join(out_dir, f"m{rank}.npz"), pU=p.U, pV=p.V, wU=w.U, wV=w.V, freeU=free.U, freeV=free.V, pmU=pm.U, pmV=pm.V, wwU=ww.U, wwV=ww.V, plU=pl.U, plV=pl.V, klU=kl.U, klV=kl.V, klwU=klw.U, klwV=klw.V,
                 klw_updates=np.array([[float(v) for v in r[1:]] for r in klw.logs["updates"].values.tolist()]), wrU=wr.U, wrV=wr.V, wr_sums=np.array(wr_sums),
                 wr_updates=np.array([[float(v) for v in r[1:]] for r in wr.logs["updates"].values.tolist()]),
                 pl_updates=np.array([[float(v) for v in r[1:]] for r in pl.logs["updates"].values.tolist()]),
                 pl_boolean=np.array([[float(v) for v in r[1:]] for r in pl.logs["boolean"].values.tolist()]),
                 kl_updates=np.array([[float(v) for v in r[1:]] for r in kl.logs["updates"].values.tolist()]),
                 pm_updates=np.array([[float(v) for v in r[1:]] for r in pm.logs["updates"].values.tolist()]),
                 pm_boolean=np.array([[float(v) for v in r[1:]] for r in pm.logs["boolean"].values.tolist()]),
                 ww_updates=np.array([[float(v) for v in r[1:]] for r in ww.logs["updates"].values.tolist()]), counts=np.array(tp), sums=np.array(rs),
                 p_updates=np.array([[float(v) for v in r[1:]] for r in p.logs["updates"].values.tolist()]),
                 p_boolean=np.array([[float(v) for v in r[1:]] for r in p.logs["boolean"].values.tolist()]),
                 w_updates=np.array([[float(v) for v in r[1:]] for r in w.logs["updates"].values.tolist()]))
    finally:
        dist.destroy_process_group()


def test_model_classes_shard_their_rows_under_a_process_group(tmp_path):
    """BinaryMFPenalty / WNMF .fit() called identically on every rank of a torch.distributed group (SURVEY 8e through the
    drop-in surface): each rank keeps its row shard on the device, and ends with the full factors and the logs of the
    single-process fit."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import contextlib
    import io
    import torch.multiprocessing as mp
    from pybmf_amd.models import BinaryMFPenalty, PNLPF, WNMF
    X, _, _, _ = orc.synthetic_boolean(1100, 600, 7, (0.2, 0.2), seed=51)
    X = orc.flip_noise(X, (0.05, 0.01), seed=52).astype(np.uint8)
    fit = dict(task="reconstruction", show_logs=False, show_result=False, save_model=False)
    with contextlib.redirect_stdout(io.StringIO()):
        p = BinaryMFPenalty(k=7, W="full", reg=1.0, reg_growth=1.1, init_method="normal", normalize_method="balance", max_iter=6, seed=3)
        p.fit(X, **fit)
        w = WNMF(k=7, W="full", init_method="normal", max_iter=5, seed=3)
        w.fit(X, **fit)
        Xs, Wm = masked_inputs(X)
        pm = BinaryMFPenalty(k=7, W="mask", reg=1.0, reg_growth=1.1, init_method="normal", normalize_method="balance", max_iter=4, seed=3)
        pm.fit(Xs, **fit)
        ww = WNMF(k=7, W=Wm, init_method="normal", max_iter=4, seed=3)
        ww.fit(X, **fit)
        pl = PNLPF(k=7, W="full", reg=1.0, reg_growth=1.1, link_lamda=10, init_method="normal", normalize_method="balance", max_iter=3, seed=3)
        pl.fit(X, **fit)
        kl = WNMF(k=7, W="mask", beta_loss="kullback-leibler", init_method="normal", max_iter=3, seed=3)
        kl.fit(Xs, **fit)
        klw = WNMF(k=7, W=Wm, beta_loss="kullback-leibler", init_method="normal", max_iter=3, seed=3)
        klw.fit(X, **fit)
        wr = WNMF(k=7, W="full", init_method="normal", max_iter=4, seed=3)
        wr.fit(real_input(X), **fit)
    assert not p._sharded
    world = 2
    mp.spawn(model_worker, args=(world, free_port(), X, str(tmp_path)), nprocs=world, join=True)
    rel = lambda a, b: np.linalg.norm(a - b) / np.linalg.norm(b)  # noqa: E731
    frame = lambda df: np.array([[float(v) for v in r[1:]] for r in df.values.tolist()])  # noqa: E731
    for r in range(world):
        z = np.load(os.path.join(tmp_path, f"m{r}.npz"))
        assert z["pU"].shape == p.U.shape and rel(z["pU"], p.U) < 2e-6 and rel(z["pV"], p.V) < 2e-6
        assert rel(z["wU"], w.U) < 2e-6 and rel(z["wV"], w.V) < 2e-6
        np.testing.assert_allclose(z["p_updates"], frame(p.logs["updates"]), rtol=2e-6)
        np.testing.assert_allclose(z["p_boolean"], frame(p.logs["boolean"]), rtol=1e-12)
        np.testing.assert_allclose(z["w_updates"], frame(w.logs["updates"]), rtol=2e-6)
        assert rel(z["pmU"], pm.U) < 2e-6 and rel(z["pmV"], pm.V) < 2e-6 and rel(z["wwU"], ww.U) < 2e-6 and rel(z["wwV"], ww.V) < 2e-6
        np.testing.assert_allclose(z["pm_updates"], frame(pm.logs["updates"]), rtol=5e-6)
        np.testing.assert_allclose(z["pm_boolean"], frame(pm.logs["boolean"]), rtol=1e-12)
        np.testing.assert_allclose(z["ww_updates"], frame(ww.logs["updates"]), rtol=5e-6)
        assert rel(z["plU"], pl.U) < 5e-6 and rel(z["plV"], pl.V) < 5e-6 and rel(z["klU"], kl.U) < 5e-6 and rel(z["klV"], kl.V) < 5e-6
        np.testing.assert_allclose(z["pl_updates"], frame(pl.logs["updates"]), rtol=1e-5)
        np.testing.assert_allclose(z["pl_boolean"], frame(pl.logs["boolean"]), rtol=1e-12)
        np.testing.assert_allclose(z["kl_updates"], frame(kl.logs["updates"]), rtol=1e-5)
        # WNMF-KL under a weight matrix, row-sharded (the V-side denominator = the column sums of U over all ranks' rows)
        assert rel(z["klwU"], klw.U) < 5e-6 and rel(z["klwV"], klw.V) < 5e-6
        np.testing.assert_allclose(z["klw_updates"], frame(klw.logs["updates"]), rtol=1e-5)
        assert rel(z["wrU"], wr.U) < 5e-6 and rel(z["wrV"], wr.V) < 5e-6
        np.testing.assert_allclose(z["wr_updates"], frame(wr.logs["updates"]), rtol=2e-5)
        np.testing.assert_allclose(z["wr_sums"], np.array(wr._residual_sums()), rtol=1e-5)
        assert tuple(z["counts"]) == tuple(p._cover_counts())
        z0 = np.load(os.path.join(tmp_path, "m0.npz"))
        assert np.array_equal(z["freeV"], z0["freeV"]) and np.array_equal(z["freeU"], z0["freeU"]) and np.isfinite(z["freeU"]).all()
        np.testing.assert_allclose(z["sums"], np.array(p._residual_sums()), rtol=1e-6)
