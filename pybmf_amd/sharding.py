"""Row sharding of the multiplicative-update loop over the GPUs of one node.

Rank p holds rows [lo, hi) of X (bits, both orientations) and of U; V is replicated.  Per iteration the only exchange
is a sum over ranks of two buffers:

    exchange_f32 : X_p^T U_p                     (n_pad x kp fp32)   -- numerator of the next V update
    exchange_f64 : [sum U_p o (X_p V), sum (U_p^2 - U_p)^2, TP_p, FP_p, sum|R_p|, sum R_p^2, -, -, U_p^T U_p (kp x kp)]

issued with ``torch.distributed.all_reduce`` (RCCL over xGMI under the "nccl" backend; "gloo" on CPU in the tests).
Everything else (V update, V^T V, log row, stopping rule) is computed redundantly and identically on every rank.

``ExchangeLoop`` is the host-side protocol; a backend provides ``local_prepare() / local_update(reg) /
finalize(it, reg)`` and the two exchange tensors.  ``engine.MUEngine`` is the HIP backend; the CPU tests plug in a
NumPy backend to check the protocol under gloo without a GPU.
"""
from __future__ import annotations

from typing import Tuple


def shard_rows(m: int, rank: int, world: int) -> Tuple[int, int]:
    """Row range [lo, hi) of rank `rank`: contiguous blocks of whole 32-row groups, sizes differ by at most one group
    (so every rank's bit columns of X^T are whole words)."""
    groups = (m + 31) // 32
    base, extra = divmod(groups, world)
    g_lo = rank * base + min(rank, extra)
    g_hi = g_lo + base + (1 if rank < extra else 0)
    return min(g_lo * 32, m), min(g_hi * 32, m)


class ExchangeLoop:
    """prepare / step / run in terms of a backend's local phases and ONE exchange point per iteration.

    Iteration dataflow under row sharding (rank p holds rows of X and of U; V replicated):

        head        V update (needs the summed X^T U and U^T U of the previous exchange), V^T V, X_p V, U_p update, and the
                    scalar part: U_p^T U_p, cover counts, MAE sums, partial error terms  ->  the fp64 block is complete
        X^T U       in column blocks: X_p^T U_p[:, block b]  ->  exchange block b is complete
        exchange    all-reduce(block 0 + the fp64 block) is started as soon as block 0 is enqueued and runs under the GEMM of
                    block 1; all-reduce(block 1) follows; both are awaited before the log row is finalised

    so the numerator of the NEXT V update travels with the scalars of THIS iteration.  A backend that keeps X^T U in
    one block (small shards: the split costs more than it hides) sends the fp64 block first, under the X^T U GEMM, and the
    numerator after it.
    """

    sharded: bool = False
    group = None
    _timing = None
    _coalesce = None

    # backend protocol -------------------------------------------------------------------------------------------
    def local_prepare(self):
        """Iteration-0 bookkeeping with the same products as an update (all exchange buffers complete afterwards)."""
        raise NotImplementedError

    def local_update(self, reg: float):
        """Whole local iteration."""
        self.local_update_head(reg)
        for b in range(self.n_blocks()):
            self.local_xtu_block(b)

    def local_update_head(self, reg: float):
        """V update .. U update and the scalar part: afterwards the fp64 exchange block is complete."""
        raise NotImplementedError

    def local_xtu_block(self, b: int):
        """X_p^T U_p for column block b: afterwards exchange block b is complete."""
        raise NotImplementedError

    def n_blocks(self) -> int:
        return 1

    def exchange_block(self, b: int):
        """fp32 tensor (contiguous) holding column block b of X_p^T U_p."""
        raise NotImplementedError

    def exchange_scalars(self):
        """fp64 tensor: [sum U_p o (X_p V), sum (U_p^2 - U_p)^2, TP_p, FP_p, sum|R_p|, sum R_p^2, -, -, U_p^T U_p (kp x kp)]."""
        raise NotImplementedError

    def finalize(self, it: int, reg: float):
        raise NotImplementedError

    def stopped(self) -> bool:
        """Has the stopping rule fired (identical on every rank: it is evaluated on all-reduced values)?  May synchronise."""
        return False

    # exchange ---------------------------------------------------------------------------------------------------
    def _all_reduce_async(self, bufs):
        """Sum `bufs` over the ranks, asynchronously; RCCL ("nccl"): one grouped launch for all of them."""
        import torch.distributed as dist
        if self._coalesce is None:   # (decided once: this runs several times per iteration of a host-paced loop)
            # usable = RCCL backend and this torch has the (private) coalescing context manager with the signature used below.
            # Decided HERE, before any collective is live: an exception raised while collectives are being issued must propagate
            # (re-issuing them after a partial failure would sum buffers twice or desynchronise the ranks).
            usable = dist.get_backend(self.group) == "nccl" and hasattr(dist, "_coalescing_manager")
            if usable:
                import inspect
                try:
                    params = inspect.signature(dist._coalescing_manager).parameters
                    usable = all(p in params for p in ("group", "device", "async_ops"))
                except (TypeError, ValueError):
                    usable = False
            self._coalesce = usable
        if self._coalesce and len(bufs) > 1:
            # (torch coalesces tensors of ONE dtype per launch -- "Tensors must have identical type" otherwise: the fp32 numerator
            # blocks go out together, the fp64 scalar block on its own)
            handles, by_dtype = [], {}
            for buf in bufs:
                by_dtype.setdefault(buf.dtype, []).append(buf)
            for same in by_dtype.values():
                if len(same) == 1:
                    handles.append(dist.all_reduce(same[0], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
                    continue
                with dist._coalescing_manager(group=self.group, device=same[0].device, async_ops=True) as cm:
                    for buf in same:
                        dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group)
                handles.append(cm)
            return handles
        return [dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True) for buf in bufs]

    def exchange(self):
        """Blocking form (iteration 0)."""
        if not self.sharded:
            return
        bufs = [self.exchange_block(b) for b in range(self.n_blocks())] + [self.exchange_scalars()]
        for h in self._all_reduce_async(bufs):
            h.wait()

    def prepare(self, reg0: float):
        """Iteration-0 bookkeeping: everything derived from the initial factors + log row 0."""
        self.local_prepare()
        self.exchange()
        self.finalize(0, float(reg0))

    def step(self, it: int, reg: float):
        """One full iteration `it` (>= 1) with regulariser `reg`, including its log row."""
        if not self.sharded:
            self.local_update(float(reg))
        else:
            self.local_update_head(float(reg))
            pending = []
            t = self._timing
            if t is not None:
                t["issue"].append(self._event())
            nb = self.n_blocks()
            if nb == 1:   # X^T U in one piece: the scalars go first and travel under its GEMM; the numerator follows, exposed
                pending += self._all_reduce_async([self.exchange_scalars()])
            for b in range(nb):
                self.local_xtu_block(b)
                pending += self._all_reduce_async([self.exchange_block(b)] + ([self.exchange_scalars()] if (b == 0 and nb > 1) else []))
            if t is not None:
                t["before_wait"].append(self._event())
            for h in pending:
                h.wait()
            if t is not None:
                t["after_wait"].append(self._event())
        self.finalize(int(it), float(reg))

    def stop_probe(self):
        """Start reading the stop flag as of the iterations enqueued so far; the handle goes to stop_probe_result()."""
        return self.stopped()

    def stop_probe_result(self, probe) -> bool:
        return bool(probe)

    def run(self, regs, it0: int = 1, poll_every: int = 8):
        """Iterations it0 ..; when sharded the loop is host-driven, so the device-side stop flag is looked at every `poll_every`
        iterations and the remaining iterations -- no-ops on the device -- are not enqueued.  The look is one poll period late:
        a probe is started at one poll point and read at the next, by which time it has long landed, so the host never drains
        the queue.  Every rank reads the flag as of the SAME iteration (it derives from all-reduced values), so all ranks leave
        the loop at the same point -- a rank that left earlier would leave the others waiting in a collective."""
        probe = None
        for i, r in enumerate(regs):
            self.step(it0 + i, r)
            if self.sharded and poll_every and (i + 1) % poll_every == 0 and i + 1 < len(regs):
                if probe is not None and self.stop_probe_result(probe):
                    break
                probe = self.stop_probe()

    # timing of the exchange (bench.py) -----------------------------------------------------------------------------
    def _event(self):
        import torch
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    def comm_timing(self, on: bool):
        """on=True: start recording; on=False: stop and return the per-step means (ms): `exposed_comm_ms_per_step` = what the
        compute stream waited for the collectives, `xtu_and_exchange_ms_per_step` = from the start of the X^T U blocks to the
        end of the exchange."""
        if on:
            self._timing = {"issue": [], "before_wait": [], "after_wait": []}
            return None
        t, self._timing = self._timing, None
        if not t or not t["issue"]:
            return {}
        import torch
        torch.cuda.synchronize()
        n = len(t["issue"])
        exposed = sum(a.elapsed_time(b) for a, b in zip(t["before_wait"], t["after_wait"])) / n
        span = sum(a.elapsed_time(b) for a, b in zip(t["issue"], t["after_wait"])) / n
        return {"exposed_comm_ms_per_step": exposed, "xtu_and_exchange_ms_per_step": span, "steps_timed": n}

    def exchange_description(self) -> str:
        blocks = [self.exchange_block(b) for b in range(self.n_blocks())]
        sc = self.exchange_scalars()
        if len(blocks) == 1:
            return (f"per step: all-reduce(SUM) of {sc.numel() * sc.element_size()} B ({sc.dtype}) of scalars / U^T U, issued before the X^T U GEMM and "
                    f"running under it, then all-reduce(SUM) of the {blocks[0].numel() * blocks[0].element_size()} B ({blocks[0].dtype}) numerator X^T U")
        return (f"{len(blocks)} all-reduce(SUM) of {blocks[0].numel() * blocks[0].element_size()} B ({blocks[0].dtype}) per step, the first grouped "
                f"with {sc.numel() * sc.element_size()} B ({sc.dtype}) of scalars / U^T U; block 0 runs under the X^T U GEMM of block 1")
