"""The N > 1 path on CPU: world_size-2 (and 3) `gloo` runs of the sharding protocol (pybmf_amd/sharding.py) with a NumPy
backend standing in for the HIP kernels, checked against the unsharded CPU oracle.

What this covers: shard_rows, which buffers are exchanged and that a plain SUM all-reduce of them is sufficient
(X_p^T U_p, U_p^T U_p, the partial scalars and TP/FP), that V stays replicated bit-for-bit, the log assembled from the
reduced block, and the one-exchange-point-per-iteration ordering (numerator of the NEXT V update travels with the scalars of
THIS iteration; X^T U goes out in two column blocks, the first grouped with the fp64 block, so that one collective can run
under the other block's GEMM).  The HIP backend (engine.MUEngine) implements the same protocol; its arithmetic is checked
on the GPU by tests/test_*_gpu.py.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle as orc
from pybmf_amd.sharding import ExchangeLoop, shard_rows

EPS = orc.EPS


class NumpyBackend(ExchangeLoop):
    """fp64 NumPy stand-in with the dataflow of csrc/api.hip (sweep + finalize)."""

    def __init__(self, Xp, n, k, U0p, V0, sum_x, cells, sharded, nblocks=2):
        self.Xp, self.k = Xp.astype(np.float64), k
        self.U, self.V = U0p.copy(), V0.copy()
        self.sum_x, self.cells, self.sharded = sum_x, cells, sharded
        h = (k + 1) // 2
        self.nblocks = nblocks if k >= 2 else 1
        self.Nblocks = [torch.zeros((n, h), dtype=torch.float64), torch.zeros((n, k - h), dtype=torch.float64)] if self.nblocks == 2 else \
            [torch.zeros((n, k), dtype=torch.float64)]
        self.comm = torch.zeros(8 + k * k, dtype=torch.float64)
        self.rows = []

    # two column blocks of X^T U, like the HIP backend at kp = 64 (block-major exchange buffer)
    def n_blocks(self):
        return self.nblocks

    def _cols(self, b):
        if self.nblocks == 1:
            return slice(0, self.k)
        h = (self.k + 1) // 2
        return slice(0, h) if b == 0 else slice(h, self.k)

    def exchange_block(self, b):
        return self.Nblocks[b]

    def exchange_scalars(self):
        return self.comm

    def _scalar_part(self, M):
        """Everything of the new (U, V) that goes into the fp64 block (the end of the head phase)."""
        U, V, Xp = self.U, self.V, self.Xp
        self.GV = V.T @ V
        self.regV = float(((V ** 2 - V) ** 2).sum())
        pd = orc.boolean_product(U, V, 0.5, 0.5)
        tp, fp, _, _ = orc.confusion_counts(Xp.astype(np.int64), pd)
        c = self.comm.numpy()
        c[:] = 0
        c[0], c[1], c[2], c[3] = (U * M).sum(), ((U ** 2 - U) ** 2).sum(), tp, fp
        c[8:] = (U.T @ U).ravel()

    def local_xtu_block(self, b):
        self.Nblocks[b].copy_(torch.from_numpy(np.ascontiguousarray((self.Xp.T @ self.U)[:, self._cols(b)])))

    def local_prepare(self):
        self._scalar_part(self.Xp @ self.V)
        for b in range(self.n_blocks()):
            self.local_xtu_block(b)

    def local_update_head(self, reg):
        k = self.k
        N = np.concatenate([t.numpy() for t in self.Nblocks], axis=1)   # reduced by the previous exchange
        GU = self.comm.numpy()[8:].reshape(k, k).copy()
        V = self.V
        den = V @ GU + (2 * reg * V ** 3 + reg * V)
        den[den == 0] = EPS
        V = V * ((N + 3 * reg * V ** 2) / den)
        V[V == 0] = EPS
        self.V = V
        M = self.Xp @ V
        U = self.U
        den = U @ (V.T @ V) + (2 * reg * U ** 3 + reg * U)
        den[den == 0] = EPS
        U = U * ((M + 3 * reg * U ** 2) / den)
        U[U == 0] = EPS
        self.U = U
        self._scalar_part(M)

    def finalize(self, it, reg):
        c = self.comm.numpy()
        GU = c[8:]
        rec = 0.5 * (self.sum_x - 2 * c[0] + float((GU * self.GV.ravel()).sum()))
        rg = reg * (0.5 * c[1] + 0.5 * self.regV)
        tp, fp = c[2], c[3]
        fn = self.sum_x - tp
        self.rows.append((it, rec + rg, rec, reg, rg, tp, fp, fn, self.cells - tp - fp - fn))


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def worker(rank, world, port, X, U0, V0, regs, out_dir, nblocks=2):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        m, n = X.shape
        lo, hi = shard_rows(m, rank, world)
        t = torch.tensor([float(X[lo:hi].sum())], dtype=torch.float64)
        dist.all_reduce(t)
        be = NumpyBackend(X[lo:hi], n, U0.shape[1], U0[lo:hi], V0, float(t.item()), float(m) * n, sharded=True, nblocks=nblocks)
        be.prepare(regs[0])
        be.run(regs, it0=1)
        np.savez(os.path.join(out_dir, f"r{rank}.npz"), U=be.U, V=be.V, rows=np.array(be.rows), lo=lo, hi=hi)
    finally:
        dist.destroy_process_group()


def test_shard_rows_partition():
    for m, world in [(1000, 2), (100000, 8), (33, 4), (64, 3), (5, 8)]:
        spans = [shard_rows(m, r, world) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == m
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        assert all(lo % 32 == 0 for lo, hi in spans if lo < m)
        sizes = [hi - lo for lo, hi in spans]
        assert max(sizes) - min(s for s in sizes) <= 32 + 31 or world > m // 32


@pytest.mark.parametrize("world,nblocks", [(2, 2), (3, 2), (2, 1)])
def test_sharded_loop_equals_unsharded_oracle(tmp_path, world, nblocks):
    X, _, _, _ = orc.synthetic_boolean(330, 140, 5, (0.2, 0.2), seed=21)
    X = orc.flip_noise(X, (0.05, 0.01), seed=22)
    k, iters = 5, 6
    U0, V0 = orc.init_factors(X, k, "normal", np.random.RandomState(4))
    U0, V0 = orc.balance_factors(U0, V0)
    U0, V0 = orc.zeros_to_eps(U0), orc.zeros_to_eps(V0)
    regs = [1.0 * 1.1 ** i for i in range(iters)]
    mp.get_context("spawn")
    mp.spawn(worker, args=(world, free_port(), X, U0, V0, regs, str(tmp_path), nblocks), nprocs=world, join=True)

    ref = orc.penalty_fit(X, k=k, U=U0, V=V0, reg=1.0, reg_growth=1.1, init_method="custom", normalize_method=None,
                          max_iter=iters - 1, tol=-1.0, literal=False)
    parts = [np.load(os.path.join(tmp_path, f"r{r}.npz")) for r in range(world)]
    U = np.concatenate([p["U"] for p in parts])
    np.testing.assert_allclose(U, ref["U"], rtol=1e-10, atol=1e-300)
    for p in parts:
        np.testing.assert_allclose(p["V"], ref["V"], rtol=1e-10, atol=1e-300)
        assert np.array_equal(p["V"], parts[0]["V"])            # replicated state stays bitwise identical
        assert np.array_equal(p["rows"], parts[0]["rows"])      # every rank logs the same rows
    rows = parts[0]["rows"]
    want = np.array([u[:5] for u in ref["updates"]])
    np.testing.assert_allclose(rows[:, :5], want, rtol=1e-9)
    assert [tuple(int(v) for v in r[5:]) for r in rows] == [tuple(c) for c in ref["counts"]]


def test_shard_rows_partitions_any_matrix():
    """shard_rows for every (m, world): contiguous, disjoint, covering, cut at multiples of 32 rows (the bit columns of X^T of a
    shard are whole words), sizes within one 32-row group of each other -- also when there are more ranks than groups."""
    hypothesis = pytest.importorskip("hypothesis")
    from hypothesis import given, settings, strategies as st

    @settings(max_examples=300, deadline=None, derandomize=True)
    @given(m=st.integers(1, 200_000), world=st.integers(1, 16))
    def check(m, world):
        cuts = [shard_rows(m, r, world) for r in range(world)]
        assert cuts[0][0] == 0 and cuts[-1][1] == m
        for (lo, hi), (lo2, _) in zip(cuts, cuts[1:]):
            assert lo <= hi == lo2
        for lo, hi in cuts:
            assert lo == hi == m or (lo % 32 == 0 and (hi % 32 == 0 or hi == m))   # (more ranks than 32-row groups: empty shards at the end)
        groups = [-(-(hi - lo) // 32) for lo, hi in cuts]
        assert max(groups) - min(groups) <= 1

    check()


def test_run_reads_the_stop_flag_one_poll_period_late_and_leaves_at_a_fixed_iteration():
    """ExchangeLoop.run: the probe started at one poll point is read at the next, so the loop ends at a point that depends only
    on the iteration the flag was raised at (every rank must leave at the same iteration)."""
    class Dummy(ExchangeLoop):
        sharded = True

        def __init__(self, stop_at):
            self.done, self.stop_at, self.probes = 0, stop_at, []

        def step(self, it, reg):
            self.done += 1

        def stopped(self):
            self.probes.append(self.done)
            return self.done >= self.stop_at

    for stop_at, expect in [(10, 24), (8, 16), (1, 16), (17, 32), (1000, 64)]:
        d = Dummy(stop_at)
        d.run([1.0] * 64, it0=1, poll_every=8)
        assert d.done == expect, (stop_at, d.done)
    d = Dummy(1)
    d.run([1.0] * 64, it0=1, poll_every=0)   # no polling: all iterations enqueued
    assert d.done == 64 and d.probes == []
