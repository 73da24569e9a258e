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
    """prepare / step / run in terms of a backend's local phases and one exchange per iteration."""

    sharded: bool = False
    group = None

    # backend protocol -------------------------------------------------------------------------------------------
    def local_prepare(self):
        raise NotImplementedError

    def local_update(self, reg: float):
        """Whole local iteration (head + tail)."""
        self.local_update_head(reg)
        self.local_update_tail()

    def local_update_head(self, reg: float):
        """V update .. X^T U: afterwards the first exchange buffer (X_p^T U_p) is complete."""
        raise NotImplementedError

    def local_update_tail(self):
        """U^T U, cover counts, scalar partials: afterwards the second exchange buffer is complete."""
        raise NotImplementedError

    def finalize(self, it: int, reg: float):
        raise NotImplementedError

    def exchange_buffers(self):
        raise NotImplementedError

    # loop -------------------------------------------------------------------------------------------------------
    def exchange(self):
        if not self.sharded:
            return
        import torch.distributed as dist
        for buf in self.exchange_buffers():
            dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group)

    def prepare(self, reg0: float):
        """Iteration-0 bookkeeping: everything derived from the initial factors + log row 0."""
        self.local_prepare()
        self.exchange()
        self.finalize(0, float(reg0))

    def step(self, it: int, reg: float):
        """One full iteration `it` (>= 1) with regulariser `reg`, including its log row.  When sharded, the large
        all-reduce (X^T U) is started as soon as the head is enqueued and overlaps the tail kernels."""
        if not self.sharded:
            self.local_update(float(reg))
        else:
            import torch.distributed as dist
            big, small = self.exchange_buffers()
            self.local_update_head(float(reg))
            pending = dist.all_reduce(big, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            self.local_update_tail()
            dist.all_reduce(small, op=dist.ReduceOp.SUM, group=self.group)
            pending.wait()
        self.finalize(int(it), float(reg))

    def run(self, regs, it0: int = 1):
        for i, r in enumerate(regs):
            self.step(it0 + i, r)
