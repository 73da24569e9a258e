"""Device-side state and iteration driver for the multiplicative-update hot path.

PyTorch is used for device memory, streams and ``torch.distributed`` only; all arithmetic happens in
``libbmf_hip.so`` (``pybmf_amd/_lib.py``).  Nothing in here falls back to the CPU: constructing a
``BitMatrix`` or ``MUEngine`` without a GPU raises.

Sharding: rank p of P holds a contiguous block of rows of X (as bits, both orientations)
and the matching rows of U; V is replicated.  One iteration needs one exchange: the sum over ranks of
``X_p^T U_p`` (n x k, fp32) and of a small fp64 block (``U_p^T U_p``, three scalars, TP/FP).
"""
from __future__ import annotations

import ctypes as C
import time
import os
from typing import Optional, Tuple

import numpy as np
import torch

from . import _lib as L
from ._lib import lib, check, ptr
from .sharding import ExchangeLoop, shard_rows  # noqa: F401  (shard_rows re-exported)


def round_up(x: int, m: int) -> int:
    return (x + m - 1) // m * m


def xf_slots(rows_pad: int, red_pad: int, terms: int, kp: int) -> int:
    """Slab slots bmf_xf_bits needs for this shape on the current device (stream-K decomposition)."""
    n = lib.bmf_xf_bits_slots(rows_pad, red_pad // 32, terms, kp)
    if n < 1:
        check(n, "bmf_xf_bits_slots")
    return int(n)


def xf_slots_i8(rows_pad: int, red_pad: int, kp: int) -> int:
    """Slab slots bmf_xf_bits_i8 needs for this shape on the current device."""
    n = lib.bmf_xf_bits_i8_slots(rows_pad, red_pad // 32, kp)
    if n < 1:
        check(n, "bmf_xf_bits_i8_slots")
    return int(n)


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


import contextlib  # noqa: E402
_NULL_CTX = contextlib.nullcontext()


def require_gpu(device) -> torch.device:
    device = torch.device(device)
    if device.type != "cuda" or not torch.cuda.is_available():
        raise RuntimeError("pybmf_amd runs on an AMD GPU (device 'cuda:N' under ROCm); no CPU path exists")
    return device


class BitMatrix:
    """A Boolean m x n matrix resident in HBM as bits, in both orientations (X and X^T), zero padded.

    ``X`` may be a NumPy array, a SciPy sparse matrix or a torch tensor (any device); nonzero = 1.
    Only rows [row_lo, row_hi) are kept (this rank's shard)."""

    def __init__(self, X, device="cuda:0", row_lo: int = 0, row_hi: Optional[int] = None, chunk_rows: int = 8192):
        self.device = require_gpu(device)
        m_total, n = X.shape
        row_hi = m_total if row_hi is None else row_hi
        assert 0 <= row_lo <= row_hi <= m_total
        self.m_total, self.row_lo, self.row_hi = int(m_total), int(row_lo), int(row_hi)
        self.m, self.n = int(row_hi - row_lo), int(n)
        self.m_pad = round_up(max(self.m, 1), L.ROW_PAD)
        self.n_pad = round_up(self.n, L.ROW_PAD)
        self.ldx, self.ldxt = self.n_pad // 32, self.m_pad // 32
        self.bits = torch.zeros((self.m_pad, self.ldx), dtype=torch.int32, device=self.device)
        self.bits_t = torch.zeros((self.n_pad, self.ldxt), dtype=torch.int32, device=self.device)
        assert chunk_rows % 64 == 0
        # largest byte seen in chunks that were uploaded as they are (uint8 sources; everything else arrives as `!= 0`): lets the
        # caller check "values are 0 / 1" on the device instead of with a pass over a host array of gigabytes
        self.max_u8 = 0
        vmax = None
        with torch.cuda.device(self.device):
            for r0 in range(0, self.m, chunk_rows):
                r1 = min(r0 + chunk_rows, self.m)
                xc = self._chunk_u8(X, row_lo + r0, row_lo + r1)
                if xc.numel():
                    cm = xc.max()
                    vmax = cm if vmax is None else torch.maximum(vmax, cm)
                self._pack(xc, self.bits[r0:r1])
                xt = xc.t().contiguous()
                self._pack(xt, self.bits_t[:, r0 // 32:])
            cnt = torch.zeros(1, dtype=torch.int64, device=self.device)
            check(lib.bmf_popcount(ptr(self.bits), self.m_pad, self.ldx, self.ldx, ptr(cnt), _stream()), "bmf_popcount")
            self.sum_local = int(cnt.item())
            if vmax is not None:
                self.max_u8 = int(vmax.item())

    def tiled(self):
        """(X bits, X^T bits) in the layout the int8 GEMM streams best (bmf_tile_bits): made on first use, kept."""
        if getattr(self, "_tiled", None) is None:
            with torch.cuda.device(self.device):
                t = torch.empty_like(self.bits)
                tt = torch.empty_like(self.bits_t)
                check(lib.bmf_tile_bits(ptr(self.bits), self.m_pad, self.ldx, self.ldx, ptr(t), _stream()), "bmf_tile_bits")
                check(lib.bmf_tile_bits(ptr(self.bits_t), self.n_pad, self.ldxt, self.ldxt, ptr(tt), _stream()), "bmf_tile_bits")
            self._tiled = (t, tt)
        return self._tiled

    def _chunk_u8(self, X, a: int, b: int) -> torch.Tensor:
        xc = X[a:b]
        if isinstance(xc, torch.Tensor):
            if xc.dtype != torch.uint8:
                xc = (xc != 0).to(torch.uint8)
            return xc.to(self.device).contiguous()
        if hasattr(xc, "tocoo") and hasattr(xc, "todense"):
            # scipy sparse: only the coordinates of the non-zero stored entries travel (explicit zeros are zeros), and the dense 0 / 1
            # chunk is made on the device -- densifying on the host was a float64 matrix of the chunk's size, a compare and an upload
            # of one byte per cell (20 ms of a 90-ms fit at MovieLens-1M shape; 1.3 GB per chunk at 8192 x 20 000)
            xc = xc.tocsr()          # (a row slice: a new object, safe to canonicalise in place)
            xc.sum_duplicates()      # what todense() would have added up
            co = xc.tocoo()
            keep = co.data != 0
            rr = torch.from_numpy(np.ascontiguousarray(co.row[keep], dtype=np.int64)).to(self.device)
            cc = torch.from_numpy(np.ascontiguousarray(co.col[keep], dtype=np.int64)).to(self.device)
            dense = torch.zeros(xc.shape, dtype=torch.uint8, device=self.device)
            dense[rr, cc] = 1
            return dense
        if hasattr(xc, "todense"):  # other sparse containers
            xc = np.asarray(xc.todense())
        xc = np.asarray(xc)
        xc = np.ascontiguousarray(xc) if xc.dtype == np.uint8 else np.ascontiguousarray(xc != 0).view(np.uint8)
        return torch.from_numpy(xc).to(self.device)

    def rows_dense_u8(self, a: int, b: int) -> np.ndarray:
        """Rows [a, b) of this shard as dense uint8 on the host."""
        bts = self.bits[a:b].cpu().numpy().view(np.uint8)
        return np.unpackbits(bts, axis=1, bitorder="little")[:, : self.n]

    def _pack(self, x_u8: torch.Tensor, out_rows: torch.Tensor):
        rows, cols = x_u8.shape
        if rows == 0 or cols == 0:
            return
        # a single-row tensor reports whatever stride its producer left behind (torch ignores strides of size-1 dims)
        ldx = x_u8.stride(0) if rows > 1 else cols
        check(lib.bmf_pack_rows_u8(ptr(x_u8), rows, cols, ldx, ptr(out_rows), out_rows.stride(0), _stream()),
              "bmf_pack_rows_u8")

    def to_dense_u8(self) -> np.ndarray:
        """This shard as a dense uint8 matrix on the host (test helper; unpacks on the CPU)."""
        b = self.bits[: self.m].cpu().numpy().view(np.uint8)
        return np.unpackbits(b, axis=1, bitorder="little")[:, : self.n]


# RCCL communicators of this process, one per (process group, device): created by the first sharded engine that needs one, shared by
# the later ones, destroyed ONLY by shutdown_comms() -- never by an engine's close() or finaliser: ncclCommDestroy after
# torch.distributed.destroy_process_group(), or at interpreter shutdown, can hang or fault, and a repeated fit() should not pay
# ncclCommInitRank again.  _XTU_PLANS: the measured blocked / unblocked decision of a shape on a communicator, likewise kept.
_RCCL_COMMS: dict = {}
_XTU_PLANS: dict = {}


def _comm_key(group, device):
    dev = torch.device(device)
    return (id(group) if group is not None else "WORLD", dev.index if dev.index is not None else torch.cuda.current_device())


def shutdown_comms():
    """Destroy the cached RCCL communicators.  Call it while the process group is still alive (before
    torch.distributed.destroy_process_group()); afterwards it only forgets them."""
    import torch.distributed as dist
    alive = dist.is_available() and dist.is_initialized()
    for key, (h, _, _) in list(_RCCL_COMMS.items()):
        if alive:
            lib.bmf_comm_destroy(h)
        _RCCL_COMMS.pop(key, None)
    _XTU_PLANS.clear()


class MUEngine(ExchangeLoop):
    """Multiplicative-update engine on a BitMatrix: owns the factors, panels, workspaces and the log.

    ``mode``: L.MODE_PENALTY (BinaryMFPenalty) or L.MODE_WNMF.  ``terms``: addends per factor entry in the big
    contractions.  ``panel``: 'bf16' (terms = 3: fp32-exact operands, 2: 16 significant bits) or 'f16' (two fp16 addends
    of the column-scaled factor: 22 significant bits relative to the column maximum, 2/3 of the MFMA work of bf16 x 3).
    ``group``: a torch.distributed process group when X is row-sharded; ``None`` = single GPU."""

    def __init__(self, X: BitMatrix, k: int, mode: int = L.MODE_PENALTY, terms: int = 3, with_mae: bool = True,
                 thr=(0.5, 0.5), tol: float = 0.01, min_diff: float = 0.0, max_iter: int = 100, sharded: bool = False,
                 group=None, panel: str = "bf16", mae: str = "bf16"):
        if not (1 <= k <= L.MAX_KP):
            raise NotImplementedError(f"k={k}: this build supports 1 <= k <= {L.MAX_KP}")
        if panel not in ("bf16", "f16", "i8"):
            raise ValueError("panel must be 'bf16', 'f16' or 'i8'")
        self.panel = panel
        if panel == "f16":
            terms = 2
        if panel == "i8" and terms not in (2, 3):
            raise ValueError("panel='i8' takes terms = 3 (23-bit factor) or 2 (15-bit)")
        self.X, self.k, self.mode, self.terms, self.with_mae = X, int(k), int(mode), int(terms), bool(with_mae)
        self.kp = 32 if k <= 32 else 64
        self.max_iter = int(max_iter)
        self.sharded, self.group = bool(sharded), group
        dev = X.device
        self.device = dev
        self._dev_index = dev.index if dev.index is not None else torch.cuda.current_device()
        m_pad, n_pad, kp, T = X.m_pad, X.n_pad, self.kp, self.terms
        z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=dev)  # noqa: E731
        self.U64, self.V64 = z((m_pad, kp), torch.float64), z((n_pad, kp), torch.float64)  # master factors
        self.U, self.V = z((m_pad, kp), torch.float32), z((n_pad, kp), torch.float32)      # fp32 shadows
        self.Upanel, self.Vpanel = z((T, kp, m_pad), torch.int16), z((T, kp, n_pad), torch.int16)
        with torch.cuda.device(dev):
            if panel == "i8":
                # (X^T U is also launched one 32-column block at a time when sharded: a different stream-K cut)
                self.splits_xv, self.splits_xtu = xf_slots_i8(m_pad, n_pad, kp), max(xf_slots_i8(n_pad, m_pad, kp), xf_slots_i8(n_pad, m_pad, 32))
            else:
                self.splits_xv, self.splits_xtu = xf_slots(m_pad, n_pad, T, kp), xf_slots(n_pad, m_pad, T, kp)
        self.Mslab = z((self.splits_xv, m_pad, kp), torch.float32)
        self.Nslab = z((self.splits_xtu, n_pad, kp), torch.float32)
        # the fp32 exchange buffer X^T U.  Sharded with int8 panels and kp = 64 it can be kept in two 32-column blocks
        # ([2][n_pad][32]) so that each block is one contiguous all-reduce that runs under the GEMM of the other; whether that
        # pays is decided below from measured times (_choose_xtu_blocks), once the communicator and the state exist.
        self.nred_blocks = 1
        self._nred_flat = z((n_pad * kp,), torch.float32)
        self.Nred = self._nred_flat.view(1, n_pad, kp)
        # the padded rows of the LARGEST shard: what every per-shard-size decision that must come out the same on all ranks is taken
        # from (shards differ by up to 32 rows and pad to 512, so the local m_pad can differ between ranks) -- the form of the exchange
        # (different collectives), and the number of Gram slabs: V^T V is computed by every rank for itself, and a different slab count
        # is a different summation order, i.e. log rows (and, in a tie, a stop decision) that differ between ranks by an fp32 rounding
        if self.sharded:
            import torch.distributed as dist
            t = torch.tensor([m_pad], dtype=torch.int64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
            self.m_pad_max, self.world = int(t.item()), dist.get_world_size(self.group)
        else:
            self.m_pad_max, self.world = m_pad, 1
        self.gram_blocks = int(min(256, max(1, max(self.m_pad_max, n_pad) // 256)))
        self.gram_slabs = z((self.gram_blocks, kp, kp), torch.float32)
        self.GU, self.GV = z((kp, kp), torch.float32), z((kp, kp), torch.float32)
        self.comm = z((8 + kp * kp,), torch.float64)
        self.GV64 = z((kp * kp,), torch.float64)
        self.partU, self.partV = z((m_pad // 128, 2), torch.float64), z((n_pad // 128, 2), torch.float64)
        self.scal = z((8,), torch.float64)
        self.ubits, self.vbits = z((m_pad,), torch.int64), z((n_pad,), torch.int64)
        self.ucolbits, self.vcolbits = z((kp, m_pad // 32), torch.int32), z((kp, n_pad // 32), torch.int32)
        self.counts = z((4,), torch.int64)
        self.log_rows = self.max_iter + 2
        self.log = z((self.log_rows, L.LOG_COLS), torch.float64)
        self.stop = z((1,), torch.int32)
        self._stop_host, self._stop_turn = None, 0
        self.scaleU, self.scaleV = z((4 * kp,), torch.float32), z((4 * kp,), torch.float32)   # (int8 planes: 4 * kp, see bmf_hip.h)
        self.panel_ws = z((max(m_pad, n_pad) // 128 * kp,), torch.float32)
        # MAE pass: 'bf16' = split-bf16 MFMA (bmf_mae_sum), 'f32' = the exact-fp32 residual pass
        self.mae_ws = z((2 * (m_pad + n_pad) * kp,), torch.int16) if (self.with_mae and mae == "bf16") else None

        sum_x = float(X.sum_local)
        if self.sharded:
            import torch.distributed as dist
            # (the second entry: does any rank hold no rows?  Every rank learns it from the same collective and refuses together --
            # a rank raising alone would leave the others waiting in the next all-reduce)
            t = torch.tensor([sum_x, 1.0 if X.m == 0 else 0.0], dtype=torch.float64, device=dev)
            dist.all_reduce(t, group=self.group)
            sum_x, empties = (float(v) for v in t.cpu().numpy())
            if empties > 0:
                raise ValueError(f"row sharding: {int(empties)} rank(s) would hold no rows (m_total = {X.m_total}; shards are cut at "
                                 "multiples of 32 rows) -- use fewer ranks")
        self.sum_x = sum_x

        st = L.PenaltyState()
        st.struct_bytes = C.sizeof(L.PenaltyState)
        st.m, st.n, st.k, st.kp, st.terms = X.m, X.n, self.k, kp, T
        st.mode, st.with_mae = self.mode, int(self.with_mae)
        st.m_pad, st.n_pad = m_pad, n_pad
        st.Xbits, st.ldx, st.XTbits, st.ldxt = X.bits.data_ptr(), X.ldx, X.bits_t.data_ptr(), X.ldxt
        st.U64, st.V64 = self.U64.data_ptr(), self.V64.data_ptr()
        st.U, st.V, st.Upanel, st.Vpanel = (t.data_ptr() for t in (self.U, self.V, self.Upanel, self.Vpanel))
        st.Mslab, st.splits_xv = self.Mslab.data_ptr(), self.splits_xv
        st.Nslab, st.splits_xtu = self.Nslab.data_ptr(), self.splits_xtu
        st.Nred = self.Nred.data_ptr()
        st.gram_slabs, st.gram_blocks = self.gram_slabs.data_ptr(), self.gram_blocks
        st.GU, st.GV, st.comm, st.GV64 = (t.data_ptr() for t in (self.GU, self.GV, self.comm, self.GV64))
        st.partU, st.partV, st.scal = self.partU.data_ptr(), self.partV.data_ptr(), self.scal.data_ptr()
        st.ubits, st.ucolbits, st.lduc = self.ubits.data_ptr(), self.ucolbits.data_ptr(), m_pad // 32
        st.vbits, st.vcolbits, st.ldvc = self.vbits.data_ptr(), self.vcolbits.data_ptr(), n_pad // 32
        st.counts, st.log, st.log_rows, st.stop = self.counts.data_ptr(), self.log.data_ptr(), self.log_rows, self.stop.data_ptr()
        st.sum_x, st.cells = self.sum_x, float(X.m_total) * float(X.n)
        st.tol, st.min_diff = float(tol), float(min_diff)
        st.thr_u, st.thr_v = float(thr[0]), float(thr[1])
        st.panel_kind = {"f16": L.PANEL_F16, "i8": L.PANEL_I8, "bf16": L.PANEL_BF16}[panel]
        st.scaleU, st.scaleV, st.panel_ws = self.scaleU.data_ptr(), self.scaleV.data_ptr(), self.panel_ws.data_ptr()
        st.mae_ws = self.mae_ws.data_ptr() if self.mae_ws is not None else None
        st.nred_blocks = self.nred_blocks
        # (0 = the library's rule on this state; with several ranks the rule is evaluated on the largest shard: same answer everywhere)
        st.exchange_overlap = (2 if lib.bmf_exchange_overlap_rule(int(self.with_mae), self.m_pad_max) else 1) if self.world > 1 else 0
        if panel == "i8":
            self._xt = X.tiled()
            st.Xtiled, st.XTtiled = self._xt[0].data_ptr(), self._xt[1].data_ptr()
        self.st = st
        # Row-sharded: the loop, collectives included, is enqueued from C (bmf_penalty_run_sharded) through a communicator object
        # -- RCCL called directly when the group's backend is "nccl", the group's own all_reduce as a host callback otherwise
        # (gloo: the tests).  BMF_SHARDED_LOOP=python keeps the host-driven reference protocol (sharding.ExchangeLoop).
        self._comm, self._cb, self._cb_error, self.exchange_plan = None, None, None, {}
        if self.sharded:
            if os.environ.get("BMF_SHARDED_LOOP", "c") != "python":
                self._make_comm()
            self._choose_xtu_blocks()
            # the stop-flag probes' pinned buffers, now and not at the first probe of a timed run (a pinned allocation is milliseconds)
            with torch.cuda.device(self.device):
                self._stop_host = [torch.zeros(1, dtype=self.stop.dtype).pin_memory() for _ in range(2)]

    # ---- communicator and exchange plan (row-sharded runs) ------------------------------------------------------------------
    def _make_comm(self):
        import sys
        import torch.distributed as dist
        world, rank = dist.get_world_size(self.group), dist.get_rank(self.group)
        h = C.c_void_p()
        self._comm_cached = False
        with torch.cuda.device(self.device):
            if dist.get_backend(self.group) == "nccl":
                # One RCCL communicator per (process group, device), made once and kept (module cache): a fit() does not pay
                # ncclCommInitRank again, and nothing destroys a communicator from a finaliser (see close()).
                key = _comm_key(self.group, self.device)
                if key in _RCCL_COMMS:
                    # (a cached communicator is only as good as the process group it was made on: after destroy_process_group() and a
                    # new init -- another world size, other ranks -- or when a collected subgroup's id() is reused, the handle is
                    # stale.  It is forgotten, not destroyed: ncclCommDestroy without its peers can hang.)
                    h_, world_, rank_ = _RCCL_COMMS[key]
                    if (world_, rank_) == (world, rank):
                        self._comm, self._comm_cached = h_, True
                        return
                    _RCCL_COMMS.pop(key)
                    for k_ in [k_ for k_ in _XTU_PLANS if k_[0] == key]:
                        _XTU_PLANS.pop(k_)

                def fall_back(why):
                    print(f"[pybmf_amd] rank {rank}: no RCCL communicator of our own ({why}); "
                          "falling back to the host-driven exchange over torch.distributed", file=sys.stderr, flush=True)
                    self._comm_fallback = why

                # 1. Can every rank load RCCL at all?  Agreed on BEFORE anyone enters ncclCommInitRank: that call is a collective, and
                #    a rank that failed earlier (dlopen, a missing symbol) would leave the others blocked inside it.
                avail = int(lib.bmf_comm_available())
                why = (lib.bmf_last_error() or b"").decode() if avail != 1 else ""
                ok = torch.tensor([1 if avail == 1 else 0], dtype=torch.int32, device=self.device)
                dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=self.group)
                if int(ok.item()) != 1:
                    return fall_back(why or "RCCL cannot be loaded on another rank")
                # 2. rank 0 makes the unique id; it travels once through the existing group (129th byte: "rank 0 succeeded", so that a
                #    failure there is seen by every rank instead of leaving the others in the broadcast)
                buf = (C.c_ubyte * (L.COMM_ID_BYTES + 1))()
                if rank == 0:
                    buf[L.COMM_ID_BYTES] = 1 if lib.bmf_comm_unique_id(buf) == L.BMF_OK else 0
                t = torch.tensor(list(buf), dtype=torch.uint8, device=self.device)
                dist.broadcast(t, src=dist.get_global_rank(self.group, 0) if self.group is not None else 0, group=self.group)
                raw = bytes(t.cpu().numpy().tolist())
                if raw[L.COMM_ID_BYTES] != 1:
                    msg = lib.bmf_last_error()
                    return fall_back(f"bmf_comm_unique_id failed on rank 0: {msg.decode() if (rank == 0 and msg) else 'see rank 0'}")
                # 3. the collective bootstrap.  Every rank is known to get this far; if one of them still fails INSIDE RCCL the others
                #    learn it from the agreement below (RCCL's own bootstrap timeout applies to a rank that never arrives).
                if os.environ.get("BMF_DEBUG_COMM"):
                    print(f"[pybmf_amd] rank {rank}: entering ncclCommInitRank (world {world})", file=sys.stderr, flush=True)
                rc = lib.bmf_comm_create(raw[:L.COMM_ID_BYTES], world, rank, C.byref(h))
                why = (lib.bmf_last_error() or b"").decode() if rc != L.BMF_OK else ""
                ok = torch.tensor([1 if rc == L.BMF_OK else 0], dtype=torch.int32, device=self.device)
                dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=self.group)
                if int(ok.item()) != 1:
                    if rc == L.BMF_OK:
                        lib.bmf_comm_destroy(h)
                    return fall_back(why or "bmf_comm_create failed on another rank")
                _RCCL_COMMS[key] = (h, world, rank)
                self._comm_cached = True
            else:
                self._cb = L.ALLREDUCE_FN(self._host_allreduce)   # (kept alive with the engine)
                check(lib.bmf_comm_create_host(self._cb, None, world, rank, C.byref(h)), "bmf_comm_create_host")
        self._comm = h

    def _host_allreduce(self, user, buf, count, dtype, stream):
        """bmf_allreduce_fn over the torch.distributed group (gloo): sums `count` elements at device pointer `buf`, ordered on `stream`."""
        try:
            import torch.distributed as dist
            for t in (self._nred_flat, self.comm):
                lo = t.data_ptr()
                off = (buf - lo) // t.element_size()
                if lo <= buf and off + count <= t.numel() and (dtype == L.DTYPE_F64) == (t.dtype == torch.float64):
                    view = t[off:off + count]
                    break
            else:
                raise ValueError(f"all-reduce of an unknown buffer {buf:#x} ({count} elements)")
            # (stream 0 = the legacy default stream, which is torch's default stream: nothing to switch to)
            ctx = torch.cuda.stream(torch.cuda.ExternalStream(stream, device=self.device)) if stream else _NULL_CTX
            with ctx:
                dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group)
            return 0
        except Exception as e:  # noqa: BLE001  (must not propagate through the C frame; re-raised by _check_cb)
            self._cb_error = e
            return 1

    def _check_cb(self, rc, what):
        if rc != L.BMF_OK and self._cb_error is not None:
            e, self._cb_error = self._cb_error, None
            raise e
        check(rc, what)

    def _time_ms(self, fn, reps=5, warm=2):
        for _ in range(warm):
            fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        b.synchronize()
        return a.elapsed_time(b) / reps

    def _plan_key(self):
        """Key of a measured exchange plan.  Rank-invariant on purpose: the measuring branch is collective, so a rank that found a
        cached plan while another did not would leave that one alone in its all-reduces -- and the local m_pad can differ between
        ranks and between two fits of nearly the same size (shards pad to 512 rows)."""
        return (_comm_key(self.group, self.device), self.X.m_total, self.world, self.m_pad_max, self.X.n_pad, self.kp, self.with_mae)

    def _choose_xtu_blocks(self):
        """X^T U in one launch or in two 32-column blocks (kp = 64, int8 planes)?  Two blocks hide the all-reduce of the first under
        the GEMM of the second but cost two grid fills / drains and two slab reductions.  Decided from times measured here, on
        this shape and this communicator (each rank measures, the ranks take the maximum and so the same decision):
            one block : exposed = all-reduce(whole numerator)                    (grouped with the scalars, on the compute stream)
            two blocks: exposed = (two launches - one launch) + all-reduce(half) + max(0, all-reduce(half) - GEMM(half)) + the
                        side stream's event fences (~20 us)
        BMF_XTU_BLOCKS=1|2 overrides."""
        import torch.distributed as dist
        kp, n_pad = self.kp, self.X.n_pad
        can_block = self.panel == "i8" and kp == 64
        plan = {"loop": "C (bmf_penalty_run_sharded)" if self._comm else "python (sharding.ExchangeLoop)"}
        if getattr(self, "_comm_fallback", None):
            plan["c_loop_refused"] = self._comm_fallback
        forced = os.environ.get("BMF_XTU_BLOCKS")
        nb = 1
        if can_block and forced in ("1", "2"):
            nb, plan["decided_by"] = int(forced), "BMF_XTU_BLOCKS"
        elif can_block and self._comm and dist.get_backend(self.group) == "nccl" and self._plan_key() in _XTU_PLANS:
            cached = _XTU_PLANS[self._plan_key()]   # (same problem on the same communicator: measured once)
            nb = cached["xtu_blocks"]
            plan.update({k_: v_ for k_, v_ in cached.items() if k_ not in ("loop", "c_loop_refused")})
            plan["decided_by"] = "measured (cached)"
        elif can_block and self._comm and dist.get_backend(self.group) == "nccl":
            st, n32 = self.st, n_pad * kp
            with torch.cuda.device(self.device):
                s = _stream()
                st.nred_blocks = 1
                t_one = self._time_ms(lambda: check(lib.bmf_penalty_update_xtu(C.byref(st), -1, s), "bmf_penalty_update_xtu"))
                st.nred_blocks = 2
                t_two = self._time_ms(lambda: (check(lib.bmf_penalty_update_xtu(C.byref(st), 0, s), "bmf_penalty_update_xtu"),
                                               check(lib.bmf_penalty_update_xtu(C.byref(st), 1, s), "bmf_penalty_update_xtu")))
                ar = lambda cnt: self._time_ms(lambda: check(lib.bmf_allreduce(self._comm, ptr(self._nred_flat), cnt, None, 0, s), "bmf_allreduce"))  # noqa: E731
                ar_full, ar_half = ar(n32), ar(n32 // 2)
                t = torch.tensor([t_one, t_two, ar_full, ar_half], dtype=torch.float64, device=self.device)
                dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
                t_one, t_two, ar_full, ar_half = (float(v) for v in t.cpu().numpy())
                self._nred_flat.zero_()
            fence_ms = 0.02   # the side stream's event fences (measured with one rank, where the collectives are free: 17-19 us per step)
            e1 = ar_full
            e2 = max(0.0, t_two - t_one) + ar_half + max(0.0, ar_half - 0.5 * t_two) + fence_ms
            nb = 2 if e2 < e1 else 1
            plan.update(decided_by="measured", xtu_one_launch_ms=t_one, xtu_two_blocks_ms=t_two, allreduce_numerator_ms=ar_full,
                        allreduce_half_ms=ar_half, exposed_estimate_one_block_ms=e1, exposed_estimate_two_blocks_ms=e2)
        else:
            plan["decided_by"] = "default (one block)"
        self.nred_blocks = self.st.nred_blocks = nb
        self.Nred = self._nred_flat.view(nb, n_pad, kp // nb)
        plan["xtu_blocks"] = nb
        if plan.get("decided_by") == "measured":
            _XTU_PLANS[self._plan_key()] = dict(plan)
        self.exchange_plan = plan

    def close(self):
        """Let go of the communicator.  A host-callback communicator (its side stream and events) is freed here; an RCCL communicator
        belongs to the module cache and stays (shutdown_comms() destroys it while the process group is alive).  Idempotent."""
        comm, self._comm = getattr(self, "_comm", None), None
        if comm and not getattr(self, "_comm_cached", False):
            lib.bmf_comm_destroy(comm)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def __del__(self):
        # (only the host-callback kind: nothing RCCL is ever destroyed from a finaliser)
        try:
            self.close()
        except Exception:  # noqa: BLE001  (interpreter shutdown)
            pass

    # ---- factors -----------------------------------------------------------------------------------------
    def load_factors(self, U0: np.ndarray, V0: np.ndarray):
        """Upload this rank's rows of U (m_local x k) and the full V (n x k); resets log and stop flag."""
        X = self.X
        assert tuple(U0.shape) == (X.m, self.k) and tuple(V0.shape) == (X.n, self.k), (U0.shape, V0.shape)
        self.U64.zero_()
        self.V64.zero_()
        # (torch tensors -- e.g. factors staged in HBM beforehand -- are copied device-side: no PCIe transfer, no idle GPU)
        as_dev = lambda F: (F if torch.is_tensor(F) else torch.from_numpy(np.ascontiguousarray(F, dtype=np.float64))).to(self.device, torch.float64)
        self.U64[: X.m, : self.k] = as_dev(U0)
        self.V64[: X.n, : self.k] = as_dev(V0)
        self.U.copy_(self.U64)  # shadows (the PREPARE sweep rewrites them too)
        self.V.copy_(self.V64)
        self.log.zero_()
        self.stop.zero_()
        self.counts.zero_()
        self.scal.zero_()
        self.scaleU.zero_()   # no prediction yet: the first epilogue's digit planes are rebuilt with the exact scale
        self.scaleV.zero_()

    def factors(self) -> Tuple[np.ndarray, np.ndarray]:
        X = self.X
        return (self.U64[: X.m, : self.k].cpu().numpy(), self.V64[: X.n, : self.k].cpu().numpy())

    # ---- iteration (backend protocol of sharding.ExchangeLoop) ---------------------------------------------------
    def n_blocks(self):
        return self.nred_blocks

    def exchange_block(self, b):
        return self.Nred[b]

    def exchange_scalars(self):
        return self.comm

    def stopped(self):
        with torch.cuda.device(self.device):
            return int(self.stop.item()) != 0

    def stop_probe(self):
        """Asynchronous copy of the stop flag to pinned host memory + an event (ExchangeLoop.run reads it one poll period later)."""
        with torch.cuda.device(self.device):
            if self._stop_host is None:
                self._stop_host = [torch.zeros(1, dtype=self.stop.dtype).pin_memory() for _ in range(2)]
            host = self._stop_host[self._stop_turn]
            self._stop_turn ^= 1
            host.copy_(self.stop, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            return host, ev

    def stop_probe_result(self, probe):
        host, ev = probe
        ev.synchronize()
        return int(host[0]) != 0

    # every launch goes to the CURRENT stream of the engine's own device (an engine built on cuda:1 while cuda:0 is current
    # must not enqueue on cuda:0's stream against cuda:1 pointers).  Entering a device context costs several microseconds of
    # host time per call -- the sharded loop is host-paced at small shards -- so it is only entered when the device is not current.
    def _on_device(self):
        return _NULL_CTX if torch.cuda.current_device() == self._dev_index else torch.cuda.device(self.device)

    def local_prepare(self):
        with self._on_device():
            check(lib.bmf_penalty_prepare(C.byref(self.st), _stream()), "bmf_penalty_prepare")

    def local_update(self, reg: float):
        with self._on_device():
            check(lib.bmf_penalty_update(C.byref(self.st), float(reg), _stream()), "bmf_penalty_update")

    def local_update_head(self, reg: float):
        with self._on_device():
            check(lib.bmf_penalty_update_head(C.byref(self.st), float(reg), _stream()), "bmf_penalty_update_head")

    def local_xtu_block(self, b: int):
        with self._on_device():
            check(lib.bmf_penalty_update_xtu(C.byref(self.st), int(b), _stream()), "bmf_penalty_update_xtu")

    def finalize(self, it: int, reg: float):
        with self._on_device():
            check(lib.bmf_penalty_finalize(C.byref(self.st), int(it), float(reg), self.max_iter, _stream()), "bmf_penalty_finalize")

    def prepare(self, reg0: float):
        if self.sharded and self._comm:
            with torch.cuda.device(self.device):
                self._check_cb(lib.bmf_penalty_prepare_sharded(C.byref(self.st), self._comm, float(reg0), self.max_iter, _stream()),
                               "bmf_penalty_prepare_sharded")
        else:
            super().prepare(reg0)

    def step(self, it: int, reg: float):
        if self.sharded and self._comm:
            self.run([reg], it0=it)
        else:
            super().step(it, reg)

    def run(self, regs, it0: int = 1, poll_every: int = 16):
        """Iterations it0 .. it0+len(regs)-1.  Single GPU: one C call enqueues all of them (no Python in the loop).  Row-sharded: one
        C call per `poll_every` iterations, collectives included; between calls the device-side stop flag is looked at WITHOUT
        draining the queue -- an asynchronous copy is started at one poll point and read at the next -- and the remaining
        iterations (no-ops on the device, but their collectives would still run) are not enqueued.  The flag derives from
        all-reduced values, so every rank reads the same value at the same point and leaves the loop together."""
        regs = [float(r) for r in regs]
        if not regs:
            return
        if not self.sharded:
            arr = (C.c_double * len(regs))(*regs)
            with torch.cuda.device(self.device):
                check(lib.bmf_penalty_run(C.byref(self.st), it0, it0 + len(regs), arr, self.max_iter, _stream()), "bmf_penalty_run")
        elif self._comm:
            with torch.cuda.device(self.device):
                s, probe, i = _stream(), None, 0
                while i < len(regs):
                    chunk = regs[i:i + poll_every] if poll_every else regs[i:]
                    arr = (C.c_double * len(chunk))(*chunk)
                    self._check_cb(lib.bmf_penalty_run_sharded(C.byref(self.st), self._comm, it0 + i, it0 + i + len(chunk), arr, self.max_iter, s),
                                   "bmf_penalty_run_sharded")
                    i += len(chunk)
                    if i < len(regs):
                        if probe is not None and self.stop_probe_result(probe):
                            break
                        probe = self.stop_probe()
        else:
            with torch.cuda.device(self.device):
                super().run(regs, it0)

    def comm_timing(self, on: bool):
        if not (self.sharded and self._comm):
            return super().comm_timing(on)
        if on:
            check(lib.bmf_comm_timing(self._comm, 4096), "bmf_comm_timing")
            return None
        n, ex, sp = C.c_int32(0), C.c_double(0.0), C.c_double(0.0)
        check(lib.bmf_comm_timing_read(self._comm, C.byref(n), C.byref(ex), C.byref(sp)), "bmf_comm_timing_read")
        check(lib.bmf_comm_timing(self._comm, 0), "bmf_comm_timing")
        if n.value == 0:
            return {}
        return {"exposed_comm_ms_per_step": ex.value / n.value, "xtu_and_exchange_ms_per_step": sp.value / n.value, "steps_timed": n.value}

    def exchange_description(self) -> str:
        if not (self.sharded and self._comm):
            return super().exchange_description()
        n32, n64 = self._nred_flat.numel() * 4, self.comm.numel() * 8
        if self.nred_blocks == 2:
            return (f"per step, issued from C on a side stream: X^T U block 0 -> grouped all-reduce(SUM) of {n32 // 2} B (fp32 numerator block 0) + {n64} B "
                    f"(fp64 scalars / U^T U) under the GEMM of block 1 -> all-reduce(SUM) of {n32 // 2} B (block 1)")
        if int(lib.bmf_exchange_overlaps(C.byref(self.st), self._comm)) == 1:
            return (f"per step, issued from C: X^T U GEMM -> all-reduce(SUM) of the {n32} B fp32 numerator on a side stream, the scalar part of the step "
                    f"(cover count, MAE, gather) under it on the compute stream -> all-reduce(SUM) of {n64} B (fp64 scalars / U^T U)")
        return (f"per step, issued from C on the compute stream after the X^T U GEMM: ONE grouped RCCL launch = all-reduce(SUM) of the {n32} B fp32 "
                f"numerator X^T U + all-reduce(SUM) of {n64} B (fp64 scalars / U^T U)")

    def read_log(self) -> Tuple[np.ndarray, int]:
        """(valid log rows, stop iteration or 0); synchronises."""
        with torch.cuda.device(self.device):
            log = self.log.cpu().numpy()
            stop = int(self.stop.item())
        valid = log[:, L.LOG_VALID] > 0
        return log[valid], stop


class RealMatrix:
    """A real-valued (fp32) m x n matrix in HBM in both orientations, zero padded -- the WNMF input when X is not
    Boolean.  Rows padded to 128, the contiguous (reduction) dimension to 32."""

    def __init__(self, X, device="cuda:0"):
        self.device = require_gpu(device)
        if hasattr(X, "todense"):
            X = np.asarray(X.todense())
        Xt = X if isinstance(X, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(X, dtype=np.float32))
        self.m, self.n = int(Xt.shape[0]), int(Xt.shape[1])
        self.m_pad, self.n_pad = round_up(self.m, 128), round_up(self.n, 128)
        self.X = torch.zeros((self.m_pad, self.n_pad), dtype=torch.float32, device=self.device)
        self.X[: self.m, : self.n] = Xt.to(self.device, dtype=torch.float32)
        self.XT = self.X.t().contiguous()
        self._tiled = None

    def tiled(self):
        """(Xtiled, XTtiled): both orientations as contiguous 64 x 64 blocks in the LDS image of the ring kernels (bmf_tile_f32) --
        what the contractions and the residual pass stream; built once."""
        if self._tiled is None:
            with torch.cuda.device(self.device):
                out = []
                for A, rows_pad, red in ((self.X, self.m_pad, self.n_pad), (self.XT, self.n_pad, self.m_pad)):
                    t = torch.empty((rows_pad * red,), dtype=torch.float32, device=self.device)
                    check(lib.bmf_tile_f32(ptr(A), rows_pad, red, red, ptr(t), _stream()), "bmf_tile_f32")
                    out.append(t)
                self._tiled = tuple(out)
        return self._tiled


class RealMUEngine:
    """WNMF (Frobenius, all-ones mask) multiplicative updates on a real-valued X (PyBMF/models/WNMF.py:96-144).

    Same kernels as the Boolean engine except for the two big contractions, which use the exact-fp32 MFMA GEMM
    (bmf_xf_f32).  The loop is driven from Python, one iteration = ~10 launches; the scalars of an iteration are read
    back once per iteration for the log and the stopping rule (this is the small-matrix secondary path, config #2)."""

    def __init__(self, X: RealMatrix, k: int, with_mae: bool = True, sharded: bool = False, group=None, m_total: Optional[int] = None,
                 bf16x3: bool = False):
        """``sharded``: X holds this rank's rows, U is local, V replicated; X^T U, U^T U and the scalar sums are summed over the
        ranks of `group` (torch.distributed).  ``m_total``: rows of the whole matrix."""
        if not (1 <= k <= L.MAX_KP):
            raise NotImplementedError(f"k={k}: this build supports 1 <= k <= {L.MAX_KP}")
        self.X, self.k, self.with_mae = X, int(k), bool(with_mae)
        # C-side loop at k <= 32: contractions on the bf16 matrix instruction with both operands split three ways instead of the exact-fp32
        # instruction -- same results to 2^-23, same speed (the passes are bound by the LDS-DMA stream, not by the matrix pipe:
        # profiles/r05_c2_bf3_ab.txt), so off by default
        self.bf16x3 = bool(bf16x3)
        self.sharded, self.group = bool(sharded), group
        self.m_total = int(m_total) if m_total is not None else X.m
        self.kp = kp = 32 if k <= 32 else 64
        dev = self.device = X.device
        z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=dev)  # noqa: E731
        m_pad, n_pad = X.m_pad, X.n_pad
        self.U64, self.V64 = z((m_pad, kp), torch.float64), z((n_pad, kp), torch.float64)
        self.U, self.V = z((m_pad, kp), torch.float32), z((n_pad, kp), torch.float32)
        import os
        target = 1024  # workgroups per GEMM launch (row tiles x reduction splits)
        # (the GEMM tiles rows by 64; every workgroup should get at least ~8 stages of 64 reduction indices)
        def splits_for(tiles, red):
            # 512 workgroups are resident (two per CU); a launch of just over 512 or 1024 leaves a last round of a few workgroups
            # running alone (20000 x 5000: X^T U with 13 splits = 1040 workgroups 119.5 us, with 12 = 960 workgroups 114.8 us)
            s = max(1, min(red // 512, -(-target // tiles)))
            if s > 1 and (tiles * s) % 512 < 128 and tiles * (s - 1) >= 768:
                s -= 1
            return s
        self.splits_xv, self.splits_xtu = splits_for(m_pad // 64, n_pad), splits_for(n_pad // 64, m_pad)
        self.Mslab, self.Nslab = z((self.splits_xv, m_pad, kp), torch.float32), z((self.splits_xtu, n_pad, kp), torch.float32)
        # the factors in the fragment orders of the tiled kernels (bmf_frag_f32 / bmf_frag_rows_f32), rebuilt before every use
        self._Ufrag, self._Vfrag, self._Vrf = z((m_pad * kp,), torch.float32), z((n_pad * kp,), torch.float32), z((n_pad * kp,), torch.float32)
        # (the fused update of the C-side loop leaves one Gram slab per workgroup, one workgroup per 128 rows up to this many)
        self.gram_blocks = int(min(1024, max(1, max(m_pad, n_pad) // 128)))
        self.gram_slabs = z((self.gram_blocks, kp, kp), torch.float32)
        self.GU, self.GV = z((kp, kp), torch.float32), z((kp, kp), torch.float32)
        self.GU64, self.GV64 = z((kp * kp,), torch.float64), z((kp * kp,), torch.float64)
        self.partU, self.partV = z((m_pad // 128, 2), torch.float64), z((n_pad // 128, 2), torch.float64)
        # outputs of the shared epilogue that this path does not use
        self._panel = z((1, kp, max(m_pad, n_pad)), torch.int16)
        self._rowbits = z((max(m_pad, n_pad),), torch.int64)
        self._colbits = z((kp, max(m_pad, n_pad) // 32), torch.int32)
        self.sums = z((4,), torch.float64)
        self._scal = z((4,), torch.float64)
        self.Nred = z((n_pad, kp), torch.float32) if self.sharded else None
        with torch.cuda.device(dev):
            self.sum_x2 = self._residual(zero_factors=True)[1]  # sum X^2 = residual pass against U = V = 0
        if self.sharded:
            self.sum_x2 = self._sum_ranks(torch.tensor([self.sum_x2], dtype=torch.float64, device=dev))[0]

    # ---- the C-side loop (csrc/wnmf_real.hip): one call enqueues whole iterations, stopping rule on the device -----------------
    def device_loop(self, max_iter: int, tol: float = 0.0, min_diff: float = 0.0):
        """Prepare the state of bmf_wnmf_real_run (unsharded fits) and write log row 0."""
        assert not self.sharded
        X, kp, dev = self.X, self.kp, self.device
        z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=dev)  # noqa: E731
        self.UT, self.VT = z((kp, X.m_pad), torch.float32), z((kp, X.n_pad), torch.float32)
        self.log_rows = int(max_iter) + 2
        self.log, self.stop = z((self.log_rows, L.LOG_COLS), torch.float64), z((1,), torch.int32)
        self._scal8 = z((8,), torch.float64)
        self.max_iter = int(max_iter)
        st = L.WnmfRealState()
        st.struct_bytes = C.sizeof(L.WnmfRealState)
        st.m, st.n, st.k, st.kp, st.with_mae = X.m, X.n, self.k, kp, int(self.with_mae)
        st.m_pad, st.n_pad = X.m_pad, X.n_pad
        st.X, st.XT = X.X.data_ptr(), X.XT.data_ptr()
        st.Xtiled, st.XTtiled = (t.data_ptr() for t in X.tiled())
        st.Vrf = self._Vrf.data_ptr()
        if kp == 32 and self.with_mae:   # the residual sums then ride in the X^T U pass (bmf_xf_f32_tiled_resid)
            self._Urf = z((X.m_pad * kp,), torch.float32)
            st.Urf = self._Urf.data_ptr()
        if kp == 32 and self.bf16x3:
            # the factors' bf16 x 3 orders (rebuilt by every fused update): X V and X^T U then run on the bf16 matrix instruction with both
            # operands split three ways -- the fp32 product to 2^-23 at 6 / 16 of the exact-fp32 instruction's matrix-pipe time
            self._UT3, self._VT3 = z((X.m_pad * 48,), torch.int32), z((X.n_pad * 48,), torch.int32)
            st.UT3, st.VT3 = self._UT3.data_ptr(), self._VT3.data_ptr()
        st.U64, st.V64, st.U, st.V, st.UT, st.VT = (t.data_ptr() for t in (self.U64, self.V64, self.U, self.V, self.UT, self.VT))
        st.Mslab, st.splits_xv, st.Nslab, st.splits_xtu = self.Mslab.data_ptr(), self.splits_xv, self.Nslab.data_ptr(), self.splits_xtu
        st.gram_slabs, st.gram_blocks = self.gram_slabs.data_ptr(), self.gram_blocks
        st.GU, st.GV, st.GU64, st.GV64 = (t.data_ptr() for t in (self.GU, self.GV, self.GU64, self.GV64))
        st.partU, st.partV = self.partU.data_ptr(), self.partV.data_ptr()
        st.rowbits, st.colbits, st.ldcb = self._rowbits.data_ptr(), self._colbits.data_ptr(), self._colbits.shape[1]
        st.sums, st.scal, st.log, st.log_rows, st.stop = self.sums.data_ptr(), self._scal8.data_ptr(), self.log.data_ptr(), self.log_rows, self.stop.data_ptr()
        st.sum_x2, st.cells, st.tol, st.min_diff = float(self.sum_x2), float(X.m) * float(X.n), float(tol), float(min_diff)
        self.st = st
        with torch.cuda.device(dev):
            check(lib.bmf_wnmf_real_prepare(C.byref(st), _stream()), "bmf_wnmf_real_prepare")

    def run(self, it0: int, it1: int):
        """Iterations it0 .. it1 - 1, enqueued by one C call (no host round trip)."""
        with torch.cuda.device(self.device):
            check(lib.bmf_wnmf_real_run(C.byref(self.st), int(it0), int(it1), self.max_iter, _stream()), "bmf_wnmf_real_run")

    def read_log(self):
        with torch.cuda.device(self.device):
            log = self.log.cpu().numpy()
            stop = int(self.stop.item())
        return log[log[:, L.LOG_VALID] > 0], stop

    def _sum_ranks(self, t):
        """Element-wise sum of a device tensor over the ranks (in place); returns it on the host."""
        import torch.distributed as dist
        dist.all_reduce(t, group=self.group)
        return [float(v) for v in t.cpu().numpy().ravel()]

    def _gram_u(self):
        """U^T U; sharded: of all ranks' rows."""
        self._gram(self.U, self.X.m_pad, self.GU, self.GU64)
        if self.sharded:
            import torch.distributed as dist
            dist.all_reduce(self.GU64, group=self.group)
            self.GU.copy_(self.GU64.view(self.kp, self.kp))

    def load_factors(self, U0, V0):
        X = self.X
        self.U64.zero_()
        self.V64.zero_()
        self.U64[: X.m, : self.k] = torch.from_numpy(np.ascontiguousarray(U0, dtype=np.float64)).to(self.device)
        self.V64[: X.n, : self.k] = torch.from_numpy(np.ascontiguousarray(V0, dtype=np.float64)).to(self.device)
        self.U.copy_(self.U64)  # shadows (the PREPARE sweep rewrites them too)
        self.V.copy_(self.V64)

    def factors(self):
        X = self.X
        return self.U64[: X.m, : self.k].cpu().numpy(), self.V64[: X.n, : self.k].cpu().numpy()   # the fp64 masters, like every engine

    def _gram(self, F, rows_pad, out32, out64):
        kk = self.kp * self.kp
        check(lib.bmf_gram_partial(ptr(F), rows_pad, self.kp, self.kp, ptr(self.gram_slabs), self.gram_blocks, _stream()), "bmf_gram_partial")
        check(lib.bmf_reduce_slabs(ptr(self.gram_slabs), kk, self.gram_blocks, kk, ptr(out32), ptr(out64), _stream()), "bmf_reduce_slabs")

    def _epilogue(self, F64, F, rows_pad, rows, num, splits, G, part, mode):
        a = L.EpilogueArgs()
        a.F64, a.F, a.rows_pad, a.rows, a.k, a.kp = F64.data_ptr(), F.data_ptr(), rows_pad, rows, self.k, self.kp
        a.num, a.slab_stride, a.splits = num.data_ptr(), rows_pad * self.kp, splits
        a.G, a.reg, a.mode, a.thr, a.terms = G.data_ptr(), 0.0, mode, 0.5, 1
        a.panel, a.ldp, a.rowbits, a.colbits, a.ldcb = (self._panel.data_ptr(), self._panel.shape[2], self._rowbits.data_ptr(),
                                                        self._colbits.data_ptr(), self._colbits.shape[1])
        a.partials, a.stop = part.data_ptr(), 0
        check(lib.bmf_mu_epilogue(C.byref(a), _stream()), "bmf_mu_epilogue")

    def _xv(self):
        X = self.X
        check(lib.bmf_frag_f32(ptr(self.V), X.n_pad, self.kp, ptr(self._Vfrag), _stream()), "bmf_frag_f32")
        check(lib.bmf_xf_f32_tiled(ptr(X.tiled()[0]), X.m_pad, X.n_pad, ptr(self._Vfrag), self.kp, ptr(self.Mslab),
                                   X.m_pad * self.kp, self.splits_xv, _stream()), "bmf_xf_f32_tiled")

    def _xtu(self):
        X = self.X
        check(lib.bmf_frag_f32(ptr(self.U), X.m_pad, self.kp, ptr(self._Ufrag), _stream()), "bmf_frag_f32")
        check(lib.bmf_xf_f32_tiled(ptr(X.tiled()[1]), X.n_pad, X.m_pad, ptr(self._Ufrag), self.kp, ptr(self.Nslab),
                                   X.n_pad * self.kp, self.splits_xtu, _stream()), "bmf_xf_f32_tiled")

    def _residual(self, zero_factors=False):
        X = self.X
        self.sums.zero_()
        U = torch.zeros_like(self.U) if zero_factors else self.U
        V = torch.zeros_like(self.V) if zero_factors else self.V
        check(lib.bmf_frag_rows_f32(ptr(V), X.n_pad, self.kp, ptr(self._Vrf), _stream()), "bmf_frag_rows_f32")
        check(lib.bmf_residual_sums_f32_tiled(ptr(X.tiled()[0]), X.m_pad, X.n_pad, ptr(U), ptr(self._Vrf), self.kp, ptr(self.sums), _stream()),
              "bmf_residual_sums_f32_tiled")
        s = self.sums.cpu().numpy()
        return float(s[0]), float(s[1])

    def scalars(self):
        """(error, RMSE, MAE) of the current factors; error by the trace form from X V, U^T U, V^T V."""
        X = self.X
        with torch.cuda.device(self.device):
            self._gram(self.V, X.n_pad, self.GV, self.GV64)
            self._xv()
            self._epilogue(self.U64, self.U, X.m_pad, X.m, self.Mslab, self.splits_xv, self.GV, self.partU, L.MODE_PREPARE)
            self._gram_u()
            out = self._scal  # one synchronising read for everything
            out.zero_()
            out[0] = self.partU[:, 1].sum()
            if self.with_mae:
                self.sums.zero_()
                check(lib.bmf_frag_rows_f32(ptr(self.V), X.n_pad, self.kp, ptr(self._Vrf), _stream()), "bmf_frag_rows_f32")
                check(lib.bmf_residual_sums_f32_tiled(ptr(X.tiled()[0]), X.m_pad, X.n_pad, ptr(self.U), ptr(self._Vrf), self.kp, ptr(self.sums),
                                                      _stream()), "bmf_residual_sums_f32_tiled")
                out[2] = self.sums[0]
            if self.sharded:   # <U, X V> and sum |R| are sums over the ranks' rows; <U^T U, V^T V> uses the summed Gram matrix
                import torch.distributed as dist
                dist.all_reduce(out, group=self.group)
            out[1] = (self.GU64 * self.GV64).sum()
            h = out.cpu().numpy()
        err = 0.5 * (self.sum_x2 - 2.0 * float(h[0]) + float(h[1]))
        cells = float(self.m_total) * float(X.n)
        mae = float(h[2]) / cells if self.with_mae else float("nan")
        return err, float(np.sqrt(max(2.0 * err, 0.0) / cells)), mae

    def update(self):
        """V then U (Gauss-Seidel), WNMF.py:96-109."""
        X = self.X
        with torch.cuda.device(self.device):
            self._gram_u()
            self._xtu()
            if self.sharded:   # X_p^T U_p summed over the ranks: the numerator of the V update
                import torch.distributed as dist
                stride = X.n_pad * self.kp
                check(lib.bmf_reduce_slabs(ptr(self.Nslab), stride, self.splits_xtu, stride, ptr(self.Nred), None, _stream()), "bmf_reduce_slabs")
                dist.all_reduce(self.Nred, group=self.group)
                self._epilogue(self.V64, self.V, X.n_pad, X.n, self.Nred, 1, self.GU, self.partV, L.MODE_WNMF)
            else:
                self._epilogue(self.V64, self.V, X.n_pad, X.n, self.Nslab, self.splits_xtu, self.GU, self.partV, L.MODE_WNMF)
            self._gram(self.V, X.n_pad, self.GV, self.GV64)
            self._xv()
            self._epilogue(self.U64, self.U, X.m_pad, X.m, self.Mslab, self.splits_xv, self.GV, self.partU, L.MODE_WNMF)


class SparseObs:
    """The observed cells of X (W = 'mask' or a weight matrix) on the device as CSR and CSC lists: (row, col, x, w).

    ``rows / cols / vals`` list every observed cell once (explicit zeros included); ``wgts`` = None means weight 1."""

    def __init__(self, rows, cols, vals, wgts, shape, device="cuda:0"):
        from scipy.sparse import coo_matrix
        self.device = require_gpu(device)
        self.m, self.n = int(shape[0]), int(shape[1])
        rows, cols = np.asarray(rows, dtype=np.int64), np.asarray(cols, dtype=np.int64)
        # every observed cell must be listed ONCE: scipy sums duplicate coordinates when it builds csr / csc, which would
        # turn the position markers below into garbage.  Repeated (row, col) pairs are merged here the way scipy would
        # (values and weights summed), which is what the reference's csr arithmetic does with them.
        key = rows * int(shape[1]) + cols
        if key.size < 2 or bool(np.all(key[1:] > key[:-1])):   # row-major and strictly increasing (a canonical csr / coo): nothing repeats
            uniq = inv = None
        else:
            uniq, inv = np.unique(key, return_inverse=True)
        if uniq is not None and uniq.size != key.size:
            vals = np.bincount(inv, weights=np.asarray(vals, dtype=np.float64), minlength=uniq.size)
            if wgts is not None:
                wgts = np.bincount(inv, weights=np.asarray(wgts, dtype=np.float64), minlength=uniq.size)
            rows, cols = uniq // int(shape[1]), uniq % int(shape[1])
        self.nnz = int(rows.size)
        # carry (value, weight) through the two orderings via the position of each cell in the input list
        pos = np.arange(1, self.nnz + 1, dtype=np.float64)  # never 0: coo -> csr keeps every cell
        csr = coo_matrix((pos, (rows, cols)), shape=shape).tocsr()
        csc = coo_matrix((pos, (rows, cols)), shape=shape).tocsc().T.tocsr()  # the same lists, by column of X (rows of X^T)
        assert csr.nnz == self.nnz and csc.nnz == self.nnz
        vals = np.asarray(vals, dtype=np.float32)
        wg = None if wgts is None else np.asarray(wgts, dtype=np.float32)

        def up(a, dt):
            return torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).to(self.device)

        def pack(mat):
            order = mat.data.astype(np.int64) - 1
            # segments of <= 64 consecutive cells of one row (load balance on power-law rows, csrc/masked.hip)
            lens = np.diff(mat.indptr)
            nsegs = (lens + 63) // 64
            row_seg_ptr = np.concatenate([[0], np.cumsum(nsegs)]).astype(np.int64)
            seg_row = np.repeat(np.arange(mat.shape[0], dtype=np.int32), nsegs)
            within = np.arange(row_seg_ptr[-1], dtype=np.int64) - row_seg_ptr[seg_row]
            seg_beg = mat.indptr[seg_row].astype(np.int64) + 64 * within
            cell_row = np.repeat(np.arange(mat.shape[0], dtype=np.int32), lens)
            return dict(ptr=up(mat.indptr, np.int64), idx=up(mat.indices, np.int32), val=up(vals[order], np.float32),
                        cell_row=up(cell_row, np.int32),
                        wgt=None if wg is None else up(wg[order], np.float32), seg_row=up(seg_row, np.int32),
                        seg_beg=up(seg_beg, np.int64), row_seg_ptr=up(row_seg_ptr, np.int64), nseg=int(row_seg_ptr[-1]),
                        rows=int(mat.shape[0]))
        self.csr = pack(csr)   # rows of X: cells of row i with their column indices
        self.csc = pack(csc)   # columns of X: cells of column j with their row indices


class ObservedScorer:
    """Scores over the STORED entries of one data set (task='prediction': utils/evaluate_utils.py:32-44 gathers the
    prediction at the triplets of X_gt, explicit zeros included, then utils/metrics.py on the two 1-D vectors)."""

    def __init__(self, X, device="cuda:0"):
        from scipy.sparse import coo_matrix
        coo = X.tocoo() if hasattr(X, "tocoo") else coo_matrix(X)
        self.m, self.n = coo.shape
        self.nnz = int(coo.nnz)
        self.device = require_gpu(device)
        self.obs = SparseObs(coo.row, coo.col, coo.data, None, coo.shape, device) if self.nnz else None
        self._scratch = {}
        if self.nnz:
            self.sums = torch.zeros(4, dtype=torch.float64, device=self.device)
            self.counts = torch.zeros(4, dtype=torch.int64, device=self.device)

    def real(self, U, V, kp, link=None, lamda=0.0):
        """(RMSE, MAE) of U V^T -- or, with link = L.LINK_SIGMOID, of sigmoid(lamda (U V^T - 1/2)), PNLPF's prediction
        (PyBMF/models/PNLPF.py:51-58) -- against the stored values; U, V: device fp32 [>= rows][kp]."""
        if not self.nnz:
            return float("nan"), float("nan")
        ls = self.obs.csr
        if isinstance(U, (list, tuple)):   # 64 < k <= 128: two blocks of 64 columns per factor (pybmf_amd/wide.py)
            if link:
                raise NotImplementedError("scores against a link prediction take k <= 64")
            with torch.cuda.device(self.device):
                if "wide" not in self._scratch:
                    z = lambda *shape: torch.zeros(shape, dtype=torch.float32, device=self.device)  # noqa: E731
                    self._scratch["wide"] = [z(self.m, 64) for _ in range(4)] + [z(max(ls["nseg"], 1), 2, 64) for _ in range(2)]
                n0, n1, d0, d1, p0, p1 = self._scratch["wide"]
                self.sums.zero_()
                check(lib.bmf_masked_pass_wide(ptr(ls["ptr"]), ptr(ls["idx"]), ptr(ls["val"]), None, self.m, ptr(ls["seg_row"]), ptr(ls["seg_beg"]),
                                               ls["nseg"], ptr(ls["row_seg_ptr"]), ptr(U[0]), ptr(U[1]), ptr(V[0]), ptr(V[1]), ptr(p0), ptr(p1),
                                               ptr(n0), ptr(n1), ptr(d0), ptr(d1), ptr(self.sums), _stream()), "bmf_masked_pass_wide")
                s = self.sums.cpu().numpy()
            return float(np.sqrt(s[0] / self.nnz)), float(s[1] / self.nnz)
        with torch.cuda.device(self.device):
            if kp not in self._scratch:
                self._scratch[kp] = (torch.zeros((self.m, kp), dtype=torch.float32, device=self.device),
                                     torch.zeros((self.m, kp), dtype=torch.float32, device=self.device),
                                     torch.zeros((max(ls["nseg"], 1), 2, kp), dtype=torch.float32, device=self.device))
            num, den, part = self._scratch[kp]
            self.sums.zero_()
            if link:
                check(lib.bmf_masked_link_pass(ptr(ls["ptr"]), ptr(ls["idx"]), ptr(ls["val"]), None, self.m, ptr(ls["seg_row"]),
                                               ptr(ls["seg_beg"]), ls["nseg"], ptr(ls["row_seg_ptr"]), ptr(U), ptr(V), kp, ptr(part),
                                               ptr(num), ptr(den), ptr(self.sums), int(link), float(lamda), _stream()), "bmf_masked_link_pass")
            else:
                check(lib.bmf_masked_pass(ptr(ls["ptr"]), ptr(ls["idx"]), ptr(ls["val"]), None, self.m, ptr(ls["seg_row"]),
                                          ptr(ls["seg_beg"]), ls["nseg"], ptr(ls["row_seg_ptr"]), ptr(U), ptr(V), kp, ptr(part),
                                          ptr(num), ptr(den), ptr(self.sums), _stream()), "bmf_masked_pass")
            s = self.sums.cpu().numpy()
        return float(np.sqrt(s[0] / self.nnz)), float(s[1] / self.nnz)

    def boolean(self, ubits, vbits, vcolbits=None, kp=None):
        """(TP, FP, FN, TN) of the Boolean product of the thresholded factors at the stored entries."""
        if not self.nnz:
            return 0, 0, 0, 0
        ls = self.obs.csr
        with torch.cuda.device(self.device):
            self.counts.zero_()
            if isinstance(ubits, (list, tuple)):   # 64 < k <= 128: two k-bit words per factor row
                check(lib.bmf_masked_counts_wide(ptr(ls["cell_row"]), ptr(ls["idx"]), ptr(ls["val"]), self.nnz, ptr(ubits[0]), ptr(ubits[1]),
                                                 ptr(vbits[0]), ptr(vbits[1]), ptr(self.counts), _stream()), "bmf_masked_counts_wide")
                return tuple(int(x) for x in self.counts.cpu().numpy())
            check(lib.bmf_masked_counts(ptr(ls["cell_row"]), ptr(ls["idx"]), ptr(ls["val"]), self.nnz, ptr(ubits), ptr(vbits),
                                        ptr(self.counts), _stream()), "bmf_masked_counts")
            return tuple(int(x) for x in self.counts.cpu().numpy())

    @property
    def cells(self):
        return self.nnz


class WholeScorer:
    """Scores of one data set over the WHOLE matrix, unstored cells counting as zeros (task='reconstruction',
    utils/evaluate_utils.py:46-51): the dense residual pass and the cover count on that set's own bits."""

    def __init__(self, X, device="cuda:0"):
        host = X
        arr = np.asarray(X.todense()) if hasattr(X, "todense") else np.asarray(X)
        self.boolean_data = bool(np.isin(arr, (0, 1)).all())
        self.device = require_gpu(device)
        self.m, self.n = arr.shape
        if self.boolean_data:
            self.bits, self.realm = BitMatrix(host, device), None
        else:
            self.bits, self.realm = None, RealMatrix(arr, device)
        self.sums = torch.zeros(4, dtype=torch.float64, device=self.device)
        self.counts = torch.zeros(4, dtype=torch.int64, device=self.device)

    def real(self, U, V, kp, link=None, lamda=0.0):
        cells = float(self.m) * float(self.n)
        if isinstance(U, (list, tuple)):   # 64 < k <= 128: the residual sums with the product over both blocks (bmf_resid_sums_wide)
            if link or self.bits is None:
                raise NotImplementedError("a rank above 64 scores Boolean (0/1) data sets against U V^T only")
            B = self.bits
            with torch.cuda.device(self.device):
                if getattr(self, "_wide_ws", None) is None:
                    self._wide_ws = torch.zeros(((B.m_pad + B.n_pad) * 2 * 64,), dtype=torch.int16, device=self.device)
                    self._tiled_t = B.tiled()[1]
                self.sums.zero_()
                check(lib.bmf_resid_sums_wide(ptr(self._tiled_t), B.ldxt, B.m_pad, B.n_pad, ptr(U[0]), ptr(U[1]), ptr(V[0]), ptr(V[1]),
                                              ptr(self._wide_ws), ptr(self.sums), 1, _stream()), "bmf_resid_sums_wide")
                s = self.sums.cpu().numpy()
            return float(np.sqrt(s[1] / cells)), float(s[0] / cells)
        with torch.cuda.device(self.device):
            self.sums.zero_()
            if link:
                if self.bits is None:
                    raise NotImplementedError("scores against a link prediction need a Boolean (0/1) data set")
                B = self.bits
                assert U.shape[0] >= B.m_pad and V.shape[0] >= B.n_pad
                check(lib.bmf_link_sums(ptr(B.bits), B.m_pad, B.ldx, self.m, self.n, ptr(U), ptr(V), B.n_pad, kp, int(link), float(lamda), None,
                                        ptr(self.sums), _stream()), "bmf_link_sums")
            elif self.bits is not None:
                B = self.bits
                assert U.shape[0] >= B.m_pad and V.shape[0] >= B.n_pad
                check(lib.bmf_residual_sums(ptr(B.bits), B.m_pad, B.ldx, self.m, self.n, ptr(U), ptr(V), kp, ptr(self.sums), None,
                                            _stream()), "bmf_residual_sums")
            else:
                R = self.realm
                assert U.shape[0] >= R.m_pad and V.shape[0] >= R.n_pad
                check(lib.bmf_residual_sums_f32(ptr(R.X), R.m_pad, R.n_pad, self.m, self.n, ptr(U), ptr(V), kp, ptr(self.sums),
                                                _stream()), "bmf_residual_sums_f32")
            s = self.sums.cpu().numpy()
        return float(np.sqrt(s[1] / cells)), float(s[0] / cells)

    def boolean(self, ubits, vbits, vcolbits, kp):
        if self.bits is None:
            raise NotImplementedError("Boolean scores need a Boolean (0/1) data set")
        B = self.bits
        with torch.cuda.device(self.device):
            self.counts.zero_()
            if isinstance(ubits, (list, tuple)):   # 64 < k <= 128
                check(lib.bmf_cover_count_wide(ptr(B.bits), B.m_pad, B.ldx, B.n_pad // 32, ptr(ubits[0]), ptr(ubits[1]), ptr(vcolbits[0]),
                                               ptr(vcolbits[1]), B.n_pad // 32, ptr(self.counts), _stream()), "bmf_cover_count_wide")
            else:
                check(lib.bmf_cover_count(ptr(B.bits), B.m_pad, B.ldx, B.n_pad // 32, ptr(ubits), ptr(vcolbits), B.n_pad // 32, kp,
                                          ptr(self.counts), None, _stream()), "bmf_cover_count")
            tp, fp = (int(v) for v in self.counts[:2].cpu().numpy())
        fn = B.sum_local - tp
        return tp, fp, fn, self.m * self.n - tp - fp - fn

    @property
    def cells(self):
        return self.m * self.n


class MaskedMUEngine:
    """Multiplicative updates with a general mask / weight matrix: the contractions run over the
    observed cells only (bmf_masked_pass, CSR for U and CSC for V), the element-wise update is the shared fp64 epilogue
    fed with (num, den).  Whole-matrix scores (task='reconstruction': RMSE / MAE / Boolean counts treat unobserved cells
    as zeros, utils/evaluate_utils.py:46-51) use the dense kernels on `bits` (Boolean X) or `real` (real-valued X).
    The loop is driven from Python, scalars are read back once per iteration."""

    def __init__(self, obs: SparseObs, k: int, mode: int, bits: Optional[BitMatrix] = None, real: Optional["RealMatrix"] = None,
                 with_mae: bool = True, thr=(0.5, 0.5), sharded: bool = False, group=None, m_total: Optional[int] = None,
                 link: int = 0, lamda: float = 10.0, real_counts: bool = False, all_cells: bool = False):
        """``sharded``: `obs` (and `bits`) hold this rank's rows only, U is local, V replicated; the partial V-side numerators /
        denominators and the scalars are summed over the ranks of `group` (torch.distributed).  ``m_total``: rows of the whole
        matrix (for the means).  ``real_counts`` (with `real`): the Boolean scores of a REAL-valued X, i.e. the reference's arithmetic
        "confusion sums" (bmf_real_confusion) instead of counts.  ``all_cells``: `obs` lists every cell with weight 1 (a real-valued X
        under the all-ones mask, which has no re-associated dense path for the penalty / link models): the sums of the pass then ARE the
        whole-matrix sums, also against a link prediction."""
        if not (1 <= k <= L.MAX_KP):
            raise NotImplementedError(f"k={k}: this build supports 1 <= k <= {L.MAX_KP}")
        self.obs, self.k, self.mode, self.bits, self.real, self.with_mae, self.thr = obs, int(k), int(mode), bits, real, with_mae, thr
        # link = L.LINK_SIGMOID: PNLPF under a mask (the product goes through sigmoid(lamda (. - 1/2)) inside the pass and the scores)
        self.link, self.lamda = int(link), float(lamda)
        # link = L.LINK_KL: WNMF's Kullback-Leibler updates under a weight matrix (numerator over the observed cells, denominator = the
        # column sums of the other factor: the reference's all-ones matrix O, WNMF.py:117-126)
        self.real_counts, self.all_cells = bool(real_counts), bool(all_cells)
        if self.link not in (0, L.LINK_SIGMOID, L.LINK_KL) or (self.link == L.LINK_SIGMOID and bits is None and not self.all_cells) or (
                self.link and bits is None and real is None):
            raise NotImplementedError("the masked engine takes link = 0, LINK_SIGMOID or LINK_KL; whole-matrix scores against a link prediction "
                                      "need a Boolean (0/1) X, or a real-valued X whose every cell is observed")
        if self.real_counts and real is None:
            raise ValueError("real_counts needs the real-valued matrix (`real`)")
        # (link = LINK_KL row-sharded: the V-side denominator is the column sum of U over ALL ranks' rows -- every rank fills denV with
        # its local column sums and _sum_v_side adds the ranks' buffers, like the numerators)
        self.kp = kp = 32 if k <= 32 else 64
        dev = self.device = obs.device
        self.m, self.n = obs.m, obs.n
        self.sharded, self.group = bool(sharded), group
        self.m_total = int(m_total) if m_total is not None else self.m
        if self.sharded and real is not None:
            raise NotImplementedError("row sharding of the masked updates takes a Boolean matrix")
        self.m_pad, self.n_pad = round_up(max(self.m, 1), L.ROW_PAD), round_up(self.n, L.ROW_PAD)
        if bits is not None:
            assert (bits.m_pad, bits.n_pad) == (self.m_pad, self.n_pad)
        self.sum_x = None
        if bits is not None:
            self.sum_x = float(bits.sum_local)
            if self.sharded:
                import torch.distributed as dist
                t = torch.tensor([self.sum_x], dtype=torch.float64, device=dev)
                dist.all_reduce(t, group=self.group)
                self.sum_x = float(t.item())
        z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=dev)  # noqa: E731
        mp, np_ = self.m_pad, self.n_pad
        self.U64, self.V64 = z((mp, kp), torch.float64), z((np_, kp), torch.float64)
        self.U, self.V = z((mp, kp), torch.float32), z((np_, kp), torch.float32)
        self.numU, self.denU, self.numV, self.denV = (z((r, kp), torch.float32) for r in (mp, mp, np_, np_))
        self.partU, self.partV = z((mp // 128, 2), torch.float64), z((np_ // 128, 2), torch.float64)
        self.Upanel, self.Vpanel = z((1, kp, mp), torch.int16), z((1, kp, np_), torch.int16)  # by-products, unused here
        self.ubits, self.vbits = z((mp,), torch.int64), z((np_,), torch.int64)
        self.ucolbits, self.vcolbits = z((kp, mp // 32), torch.int32), z((kp, np_ // 32), torch.int32)
        self.sums, self.sums2 = z((4,), torch.float64), z((4,), torch.float64)
        self.counts = z((4,), torch.int64)
        self._scal = z((8,), torch.float64)
        self.conf = z((6,), torch.float64) if self.real_counts else None

    def load_factors(self, U0, V0):
        self.U64.zero_()
        self.V64.zero_()
        self.U64[: self.m, : self.k] = torch.from_numpy(np.ascontiguousarray(U0, dtype=np.float64)).to(self.device)
        self.V64[: self.n, : self.k] = torch.from_numpy(np.ascontiguousarray(V0, dtype=np.float64)).to(self.device)
        self.U.copy_(self.U64)
        self.V.copy_(self.V64)

    def factors(self):
        return self.U64[: self.m, : self.k].cpu().numpy(), self.V64[: self.n, : self.k].cpu().numpy()

    def _pass(self, ls, rows, Fself, Fother, num, den, sums):
        if sums is not None:
            sums.zero_()
        if ls.get("part") is None:
            ls["part"] = torch.zeros((max(ls["nseg"], 1), 2, self.kp), dtype=torch.float32, device=self.device)
        # (k real columns: the padding columns of both shadows are zero, so a wave may take several cells per step when k <= 16 / 32)
        check(lib.bmf_masked_link_pass_k(ptr(ls["ptr"]), ptr(ls["idx"]), ptr(ls["val"]), ptr(ls["wgt"]), rows, ptr(ls["seg_row"]),
                                         ptr(ls["seg_beg"]), ls["nseg"], ptr(ls["row_seg_ptr"]), ptr(Fself), ptr(Fother), self.kp, self.k,
                                         ptr(ls["part"]), ptr(num), ptr(den), ptr(sums), self.link, self.lamda, _stream()), "bmf_masked_link_pass_k")
        if self.link == L.LINK_KL:   # denominator O F_other: the column sums of the other factor (fp64 master), the same for every row
            F64 = self.V64 if Fother is self.V else self.U64
            den.copy_(F64.sum(0).float().unsqueeze(0).expand_as(den))

    def _epilogue_args(self, which, mode, reg):
        a = L.EpilogueArgs()
        if which == "V":
            F64, F, rows_pad, rows, num, den = self.V64, self.V, self.n_pad, self.n, self.numV, self.denV
            panel, rb, cb, part, thr = self.Vpanel, self.vbits, self.vcolbits, self.partV, self.thr[1]
        else:
            F64, F, rows_pad, rows, num, den = self.U64, self.U, self.m_pad, self.m, self.numU, self.denU
            panel, rb, cb, part, thr = self.Upanel, self.ubits, self.ucolbits, self.partU, self.thr[0]
        a.F64, a.F, a.rows_pad, a.rows, a.k, a.kp = F64.data_ptr(), F.data_ptr(), rows_pad, rows, self.k, self.kp
        a.num, a.slab_stride, a.splits = (0 if mode == L.MODE_PREPARE else num.data_ptr()), rows_pad * self.kp, 1
        a.G, a.den, a.reg, a.mode, a.thr, a.terms = 0, den.data_ptr(), float(reg), mode, float(thr), 1
        a.panel, a.ldp, a.rowbits, a.colbits, a.ldcb = panel.data_ptr(), rows_pad, rb.data_ptr(), cb.data_ptr(), rows_pad // 32
        a.partials, a.stop = part.data_ptr(), 0
        return a

    def _epilogue(self, which, mode, reg):
        a = self._epilogue_args(which, mode, reg)
        check(lib.bmf_mu_epilogue(C.byref(a), _stream()), "bmf_mu_epilogue")

    # ---- whole iterations enqueued by one C call each (bmf_masked_iterate), scalars read one iteration late ---------------------
    LOG_ROWS = 8

    def can_pipeline(self):
        """One rank, no link or the sigmoid link (the Kullback-Leibler denominator is made with torch ops between the kernels)."""
        import os
        return (not self.sharded and self.link in (0, L.LINK_SIGMOID) and not self.real_counts and not (self.link and self.real is not None)
                and os.environ.get("BMF_MASKED_PIPELINE", "1") != "0")   # (A/B switch)

    def _side_args(self, ls, rows):
        if ls.get("part") is None:
            ls["part"] = torch.zeros((max(ls["nseg"], 1), 2, self.kp), dtype=torch.float32, device=self.device)
        sd = L.MaskedSide()
        sd.ptr, sd.idx, sd.val, sd.wgt = (ls[k].data_ptr() if ls[k] is not None else 0 for k in ("ptr", "idx", "val", "wgt"))
        sd.seg_row, sd.seg_beg, sd.row_seg_ptr, sd.part = ls["seg_row"].data_ptr(), ls["seg_beg"].data_ptr(), ls["row_seg_ptr"].data_ptr(), ls["part"].data_ptr()
        sd.rows, sd.nseg = rows, ls["nseg"]
        return sd

    def _loop_state(self):
        if getattr(self, "_loop", None) is not None:
            return self._loop
        assert self.can_pipeline()
        st = L.MaskedLoop()
        st.struct_bytes = C.sizeof(L.MaskedLoop)
        st.m, st.n, st.k, st.kp, st.link, st.lamda = self.m, self.n, self.k, self.kp, self.link, self.lamda
        st.csr, st.csc = self._side_args(self.obs.csr, self.m), self._side_args(self.obs.csc, self.n)
        st.epiU, st.epiV = self._epilogue_args("U", self.mode, 0.0), self._epilogue_args("V", self.mode, 0.0)
        self.Up64, self.Vp64 = torch.zeros_like(self.U64), torch.zeros_like(self.V64)
        st.sums, st.Up64, st.Vp64 = self.sums.data_ptr(), self.Up64.data_ptr(), self.Vp64.data_ptr()
        if self.bits is not None:
            B = self.bits
            st.Xbits, st.x_m_pad, st.ldx, st.x_n_pad = B.bits.data_ptr(), B.m_pad, B.ldx, B.n_pad
        elif self.real is not None:
            R = self.real
            st.Xreal, st.r_m_pad, st.r_n_pad = R.X.data_ptr(), R.m_pad, R.n_pad
        st.sums2, st.counts, st.nbU, st.nbV = self.sums2.data_ptr(), self.counts.data_ptr(), self.partU.shape[0], self.partV.shape[0]
        self.sums2.zero_()
        self.counts.zero_()
        self._rows_host = torch.zeros((self.LOG_ROWS, 8), dtype=torch.float64).pin_memory()
        self._events = [None] * self.LOG_ROWS
        self._loop = st
        return st

    def iterate(self, it: int, reg: float, update: bool = True):
        """Enqueue iteration `it` (update = False: only the scalars of the current state, log row 0); ``row(it, reg)`` waits for its row.
        At most LOG_ROWS - 1 iterations may be outstanding."""
        st = self._loop_state()
        slot = it % self.LOG_ROWS
        with torch.cuda.device(self.device):
            check(lib.bmf_masked_iterate(C.byref(st), float(reg), int(bool(update)), C.c_void_p(self._rows_host[slot].data_ptr()), _stream()),
                  "bmf_masked_iterate")
            ev = torch.cuda.Event()
            ev.record()
        self._events[slot] = (it, ev)

    def row(self, it: int, reg: float):
        """The scalars of iteration `it` as ``scalars(reg)`` returns them (an event wait, no polling)."""
        slot = it % self.LOG_ROWS
        if self._events[slot] is None or self._events[slot][0] != it:
            raise RuntimeError(f"row {it} is not available")
        self._events[slot][1].synchronize()
        h = self._rows_host[slot].numpy().copy()
        return self._decode_scalars(h, reg, self.bits is not None or self.real is not None, float(self.m_total) * float(self.n))

    def previous_factors(self):
        """The iterate before the last enqueued update: what a loop that ran one iteration past its stopping rule returns."""
        return self.Up64[: self.m, : self.k].cpu().numpy(), self.Vp64[: self.n, : self.k].cpu().numpy()

    def prepare(self):
        """Shadows, bits and regulariser partials of the initial factors; numerators of the first V update + rec_error."""
        with torch.cuda.device(self.device):
            self._epilogue("V", L.MODE_PREPARE, 0.0)
            self._epilogue("U", L.MODE_PREPARE, 0.0)
            self._pass(self.obs.csc, self.n, self.V, self.U, self.numV, self.denV, self.sums)
            self._sum_v_side()

    def update(self, reg):
        """V then U (Gauss-Seidel) with regulariser `reg`, then the pass that prepares the next V update and measures
        rec_error of the new state."""
        with torch.cuda.device(self.device):
            self._epilogue("V", self.mode, reg)
            self._pass(self.obs.csr, self.m, self.U, self.V, self.numU, self.denU, None)
            self._epilogue("U", self.mode, reg)
            self._pass(self.obs.csc, self.n, self.V, self.U, self.numV, self.denV, self.sums)
            self._sum_v_side()

    def _sum_v_side(self):
        """Sharded: every rank has seen only its rows' cells of each column -- the V-side numerators / denominators and the
        residual sums are sums over the ranks (the one exchange of an iteration)."""
        if not self.sharded:
            return
        import torch.distributed as dist
        for buf in (self.numV, self.denV, self.sums):
            dist.all_reduce(buf, group=self.group)

    def scalars(self, reg):
        """(error, rec_error, reg_error, RMSE, MAE, (TP, FP, FN, TN) or None) of the current state.  Everything is gathered
        into one device vector and read back ONCE (a synchronising read costs more than the kernels at MovieLens size)."""
        if not self.sharded:
            return self._scalars_one_launch(reg)
        with torch.cuda.device(self.device):
            cells = float(self.m_total) * float(self.n)
            out = self._scal
            out.zero_()
            out[0] = self.sums[0]
            out[1] = self.partU[:, 0].sum()
            out[2] = self.partV[:, 0].sum()
            have_scores = True
            if self.bits is not None:
                B = self.bits
                self.sums2.zero_()
                if self.link:   # whole-matrix RMSE / MAE against the link prediction (PNLPF.get_prediction, PNLPF.py:50-58)
                    check(lib.bmf_link_sums(ptr(B.bits), B.m_pad, B.ldx, self.m, self.n, ptr(self.U), ptr(self.V), B.n_pad, self.kp, self.link,
                                            self.lamda, None, ptr(self.sums2), _stream()), "bmf_link_sums")
                else:
                    check(lib.bmf_residual_sums(ptr(B.bits), B.m_pad, B.ldx, self.m, self.n, ptr(self.U), ptr(self.V), self.kp,
                                                ptr(self.sums2), None, _stream()), "bmf_residual_sums")
                self.counts.zero_()
                check(lib.bmf_cover_count(ptr(B.bits), B.m_pad, B.ldx, B.n_pad // 32, ptr(self.ubits), ptr(self.vcolbits),
                                          B.n_pad // 32, self.kp, ptr(self.counts), None, _stream()), "bmf_cover_count")
                out[3:5] = self.sums2[:2]
                out[5:7] = self.counts[:2].double()   # exact: counts < 2^53
            elif self.real is not None:
                R = self.real
                self.sums2.zero_()
                Up, Vp = self.U[: R.m_pad], self.V[: R.n_pad]
                check(lib.bmf_residual_sums_f32(ptr(R.X), R.m_pad, R.n_pad, self.m, self.n, ptr(Up), ptr(Vp), self.kp, ptr(self.sums2),
                                                _stream()), "bmf_residual_sums_f32")
                out[3:5] = self.sums2[:2]
            else:
                have_scores = False
            if self.sharded:   # entries 0 (already summed) and 2 (V is replicated) are the same on every rank; the rest is local
                import torch.distributed as dist
                loc = out[[1, 3, 4, 5, 6]].clone()
                dist.all_reduce(loc, group=self.group)
                out[[1, 3, 4, 5, 6]] = loc
            h = out.cpu().numpy()
        return self._decode_scalars(h, reg, have_scores, cells)

    def _decode_scalars(self, h, reg, have_scores, cells):
        rec = 0.5 * float(h[0])
        rg = float(reg) * (0.5 * float(h[1]) + 0.5 * float(h[2])) if self.mode == L.MODE_PENALTY else 0.0
        rmse = mae = float("nan")
        counts = None
        if have_scores:
            rmse, mae = float(np.sqrt(h[4] / cells)), float(h[3] / cells)
        if self.bits is not None:
            tp, fp = int(h[5]), int(h[6])
            fn = int(self.sum_x) - tp
            counts = (tp, fp, fn, self.m_total * self.n - tp - fp - fn)
        elif self.real_counts:
            counts = self._real_confusion()
        return rec + rg, rec, rg, rmse, mae, counts

    def _real_confusion(self):
        """(TP, FP, FN, TN, sum gt, sum pd) of the real-valued X against the Boolean product of the thresholded factors, as the
        reference's metrics compute them on two csr matrices (bmf_real_confusion); one more synchronising read of six doubles."""
        R = self.real
        with torch.cuda.device(self.device):
            self.conf.zero_()
            check(lib.bmf_real_confusion(ptr(R.X), R.n_pad, self.m, self.n, ptr(self.ubits), ptr(self.vbits), ptr(self.conf), _stream()),
                  "bmf_real_confusion")
            return tuple(float(v) for v in self.conf.cpu().numpy())

    def _whole_sums_real(self, s):
        """Whole-matrix sum |x - p|, sum (x - p)^2 of a real-valued X into sums2: against U V^T by the dense fp32 residual pass; against
        a link prediction from the sums of the pass itself (every cell observed with weight 1: they are the whole-matrix sums)."""
        R = self.real
        if self.link == L.LINK_SIGMOID:
            self.sums2[0] = self.sums[1]
            self.sums2[1] = self.sums[0]
            return
        Up, Vp = self.U[: R.m_pad], self.V[: R.n_pad]
        check(lib.bmf_residual_sums_f32(ptr(R.X), R.m_pad, R.n_pad, self.m, self.n, ptr(Up), ptr(Vp), self.kp, ptr(self.sums2), s),
              "bmf_residual_sums_f32")

    def _scalars_one_launch(self, reg):
        """One rank: the whole-matrix sums and the cover count, then ONE gather launch (bmf_masked_scalars, which also resets the
        accumulators for the next call) writing into pinned host memory, and one stream synchronisation -- instead of a dozen small
        torch launches and a blocking copy."""
        with torch.cuda.device(self.device):
            if getattr(self, "_scal_host", None) is None:
                self._scal_host = torch.zeros(8, dtype=torch.float64).pin_memory()
                self._scal_np = self._scal_host.numpy()
                self.sums2.zero_()
                self.counts.zero_()
            s = _stream()
            have_scores, sums2, counts = True, None, None
            if self.bits is not None:
                B = self.bits
                if self.link:   # whole-matrix RMSE / MAE against the link prediction (PNLPF.get_prediction, PNLPF.py:50-58)
                    check(lib.bmf_link_sums(ptr(B.bits), B.m_pad, B.ldx, self.m, self.n, ptr(self.U), ptr(self.V), B.n_pad, self.kp, self.link,
                                            self.lamda, None, ptr(self.sums2), s), "bmf_link_sums")
                else:
                    check(lib.bmf_residual_sums(ptr(B.bits), B.m_pad, B.ldx, self.m, self.n, ptr(self.U), ptr(self.V), self.kp,
                                                ptr(self.sums2), None, s), "bmf_residual_sums")
                check(lib.bmf_cover_count(ptr(B.bits), B.m_pad, B.ldx, B.n_pad // 32, ptr(self.ubits), ptr(self.vcolbits),
                                          B.n_pad // 32, self.kp, ptr(self.counts), None, s), "bmf_cover_count")
                sums2, counts = ptr(self.sums2), ptr(self.counts)
            elif self.real is not None:
                self._whole_sums_real(s)
                sums2 = ptr(self.sums2)
            else:
                have_scores = False
            out = self._scal_np
            # Every one of the eight words is pre-set to a "not delivered yet" bit pattern (a signalling NaN whose payload counts the
            # calls: no arithmetic produces it, so a sum that IS NaN ends the wait like any other value) and the wait is for ALL of them
            # to have changed -- not for the last-written word alone, which would rely on the device's writes to host memory becoming
            # visible in program order.
            self._scal_no = (getattr(self, "_scal_no", 0) + 1) & 0xFFFFFFFF
            bits = out.view(np.uint64)
            pending = np.uint64(0x7FF4DEAD00000000 | self._scal_no)
            bits[:] = pending
            check(lib.bmf_masked_scalars(ptr(self.sums), ptr(self.partU), self.partU.shape[0], ptr(self.partV), self.partV.shape[0], sums2, counts,
                                         C.c_void_p(self._scal_host.data_ptr()), s), "bmf_masked_scalars")
            deadline = None
            while (bits == pending).any():   # (a stream synchronisation costs ~15 us of wake-up latency; after 20 ms fall back to it)
                if deadline is None:
                    deadline = time.perf_counter() + 0.02
                elif time.perf_counter() > deadline:
                    torch.cuda.current_stream().synchronize()
                    break
            h = out.copy()
        return self._decode_scalars(h, reg, have_scores, float(self.m_total) * float(self.n))


class LinkMUEngine:
    """Multiplicative updates whose m x n product passes through an element-wise link before it is contracted again
    (SURVEY 8f rank 3): PNLPF (sigmoid link, models/PNLPF.py) and WNMF with the Kullback-Leibler loss (models/WNMF.py:111-129),
    Boolean X, all-ones mask (KL also with W='mask', see `obs_bits`).  The two contractions of an update are one tile-fused pass
    (bmf_link_pass); the element-wise
    update is the shared fp64 epilogue fed with (num slabs, den).  The loop is driven from Python, scalars are read back once
    per iteration."""

    def __init__(self, bits: BitMatrix, k: int, link: int, mode: int, lamda: float = 10.0, thr=(0.5, 0.5), mfma: str = "bf16",
                 obs_bits: Optional[BitMatrix] = None, sharded: bool = False, group=None):
        """``mfma``: 'bf16' = the passes on the split-bf16 MFMA (bmf_link_pass16, products right to 2^-16), 'f32' = exact fp32 MFMA.
        ``obs_bits`` (KL only): the observed cells of a W='mask' fit.  Every non-zero of X is observed, so W o X = X and the
        updates are those of the all-ones mask (their denominators use the all-ones matrix, WNMF.py:113,118,125); only the
        objective is summed over the observed cells."""
        if not (1 <= k <= L.MAX_KP):
            raise NotImplementedError(f"k={k}: this build supports 1 <= k <= {L.MAX_KP}")
        if mfma not in ("bf16", "f32"):
            raise ValueError("mfma must be 'bf16' or 'f32'")
        self.mfma = mfma
        if obs_bits is not None:
            assert link == L.LINK_KL and (obs_bits.m_pad, obs_bits.n_pad, obs_bits.ldx) == (bits.m_pad, bits.n_pad, bits.ldx)
        self.obs_bits = obs_bits
        # sharded: `bits` holds this rank's rows, U is local, V replicated; the V-side contractions and the scalars are summed over
        # the ranks of `group` (torch.distributed)
        self.sharded, self.group = bool(sharded), group
        self.X, self.k, self.link, self.mode, self.lamda, self.thr = bits, int(k), int(link), int(mode), float(lamda), thr
        self.kp = kp = 32 if k <= 32 else 64
        dev = self.device = bits.device
        self.m, self.n, self.m_pad, self.n_pad = bits.m, bits.n, bits.m_pad, bits.n_pad
        z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=dev)  # noqa: E731
        mp, np_ = self.m_pad, self.n_pad
        self.U64, self.V64 = z((mp, kp), torch.float64), z((np_, kp), torch.float64)
        self.U, self.V = z((mp, kp), torch.float32), z((np_, kp), torch.float32)
        self.splitsU, self.splitsV = int(lib.bmf_link_splits(self.m, self.n)), int(lib.bmf_link_splits(self.n, self.m))
        self.numU, self.numV = z((self.splitsU, mp, kp), torch.float32), z((self.splitsV, np_, kp), torch.float32)
        sig = self.link == L.LINK_SIGMOID
        self.denU_slabs = z((self.splitsU, mp, kp), torch.float32) if sig else None
        self.denV_slabs = z((self.splitsV, np_, kp), torch.float32) if sig else None
        self.denU, self.denV = z((mp, kp), torch.float32), z((np_, kp), torch.float32)
        self.colsum = z((kp,), torch.float32)
        self.partU, self.partV = z((mp // 128, 2), torch.float64), z((np_ // 128, 2), torch.float64)
        self.ubits, self.vbits = z((mp,), torch.int64), z((np_,), torch.int64)
        self.ucolbits, self.vcolbits = z((kp, mp // 32), torch.int32), z((kp, np_ // 32), torch.int32)
        self.sums = z((4,), torch.float64)
        self.counts = z((4,), torch.int64)
        self._scal = z((8,), torch.float64)
        self.sum_x, self.m_total = float(bits.sum_local), bits.m_total
        if self.sharded:
            import torch.distributed as dist
            t = torch.tensor([self.sum_x], dtype=torch.float64, device=dev)
            dist.all_reduce(t, group=self.group)
            self.sum_x = float(t.item())
            self.numV_red = z((np_, kp), torch.float32)
        # bf16 copies of the factors for the split-bf16 pass: [row-major hi | mid | lo | permuted hi | lo]
        self.wsU = z((5 * mp * kp,), torch.int16) if mfma == "bf16" else None
        self.wsV = z((5 * np_ * kp,), torch.int16) if mfma == "bf16" else None

    def _split(self, which):
        if self.mfma != "bf16":
            return
        # (both factors' copies: the per-column scales of the fp16 operands belong to the PAIR, bmf_link_split_pair)
        check(lib.bmf_link_split_pair(ptr(self.U), self.m_pad, ptr(self.V), self.n_pad, self.kp, ptr(self.wsU), ptr(self.wsV), _stream()),
              "bmf_link_split_pair")

    def load_factors(self, U0, V0):
        self.U64.zero_()
        self.V64.zero_()
        self.U64[: self.m, : self.k] = torch.from_numpy(np.ascontiguousarray(U0, dtype=np.float64)).to(self.device)
        self.V64[: self.n, : self.k] = torch.from_numpy(np.ascontiguousarray(V0, dtype=np.float64)).to(self.device)
        self.U.copy_(self.U64)
        self.V.copy_(self.V64)

    def factors(self):
        return self.U64[: self.m, : self.k].cpu().numpy(), self.V64[: self.n, : self.k].cpu().numpy()

    def _epilogue_args(self, which, mode, reg, num=None, splits=None):
        a = L.EpilogueArgs()
        if which == "V":
            F64, F, rows_pad, rows, num0, splits0, den = self.V64, self.V, self.n_pad, self.n, self.numV, self.splitsV, self.denV
            rb, cb, part, thr = self.vbits, self.vcolbits, self.partV, self.thr[1]
        else:
            F64, F, rows_pad, rows, num0, splits0, den = self.U64, self.U, self.m_pad, self.m, self.numU, self.splitsU, self.denU
            rb, cb, part, thr = self.ubits, self.ucolbits, self.partU, self.thr[0]
        num, splits = (num0 if num is None else num), (splits0 if splits is None else splits)
        a.F64, a.F, a.rows_pad, a.rows, a.k, a.kp = F64.data_ptr(), F.data_ptr(), rows_pad, rows, self.k, self.kp
        a.num, a.slab_stride, a.splits = (0 if mode == L.MODE_PREPARE else num.data_ptr()), rows_pad * self.kp, splits
        a.G, a.den, a.reg, a.mode, a.thr, a.terms = 0, den.data_ptr(), float(reg), mode, float(thr), 0
        a.panel, a.ldp, a.rowbits, a.colbits, a.ldcb = 0, rows_pad, rb.data_ptr(), cb.data_ptr(), rows_pad // 32
        a.partials, a.stop, a.blockmax = part.data_ptr(), 0, 0
        return a

    def _epilogue(self, which, mode, reg, num=None, splits=None):
        a = self._epilogue_args(which, mode, reg, num, splits)
        check(lib.bmf_mu_epilogue(C.byref(a), _stream()), "bmf_mu_epilogue")

    # ---- whole iterations enqueued by one C call each (bmf_link_iterate), scalars read one iteration late -----------------------
    LOG_ROWS = 8

    def can_pipeline(self):
        """One rank, the 16-bit MFMA flavour."""
        return not self.sharded and self.mfma == "bf16" and os.environ.get("BMF_LINK_PIPELINE", "1") != "0"   # (A/B switch)

    def _loop_state(self):
        if getattr(self, "_loop", None) is not None:
            return self._loop
        assert self.can_pipeline()
        X = self.X
        st = L.LinkLoop()
        st.struct_bytes = C.sizeof(L.LinkLoop)
        st.m, st.n, st.k, st.kp, st.link, st.splitsU, st.splitsV, st.lamda = self.m, self.n, self.k, self.kp, self.link, self.splitsU, self.splitsV, self.lamda
        st.Xbits, st.XTbits, st.m_pad, st.n_pad, st.ldx, st.ldxt = X.bits.data_ptr(), X.bits_t.data_ptr(), self.m_pad, self.n_pad, X.ldx, X.ldxt
        st.wsU, st.wsV, st.numU, st.numV = self.wsU.data_ptr(), self.wsV.data_ptr(), self.numU.data_ptr(), self.numV.data_ptr()
        if self.link == L.LINK_SIGMOID:
            st.denU_slabs, st.denV_slabs = self.denU_slabs.data_ptr(), self.denV_slabs.data_ptr()
        st.colsum = self.colsum.data_ptr()
        st.epiU, st.epiV = self._epilogue_args("U", self.mode, 0.0), self._epilogue_args("V", self.mode, 0.0)
        self.Up64, self.Vp64 = torch.zeros_like(self.U64), torch.zeros_like(self.V64)
        st.Up64, st.Vp64, st.sums, st.counts = self.Up64.data_ptr(), self.Vp64.data_ptr(), self.sums.data_ptr(), self.counts.data_ptr()
        if self.obs_bits is not None:
            st.Obits = self.obs_bits.bits.data_ptr()
        st.nbU, st.nbV = self.partU.shape[0], self.partV.shape[0]
        self._rows_host = torch.zeros((self.LOG_ROWS, 8), dtype=torch.float64).pin_memory()
        self._events = [None] * self.LOG_ROWS
        self._loop = st
        return st

    def iterate(self, it: int, reg: float, update: bool = True):
        """Enqueue iteration `it` (update = False: only the scalars of the current state, log row 0); ``row(it, reg)`` waits for its row."""
        st = self._loop_state()
        slot = it % self.LOG_ROWS
        with torch.cuda.device(self.device):
            check(lib.bmf_link_iterate(C.byref(st), float(reg), int(bool(update)), C.c_void_p(self._rows_host[slot].data_ptr()), _stream()),
                  "bmf_link_iterate")
            ev = torch.cuda.Event()
            ev.record()
        self._events[slot] = (it, ev)

    def row(self, it: int, reg: float):
        """The scalars of iteration `it` as ``scalars(reg)`` returns them (an event wait, no polling)."""
        slot = it % self.LOG_ROWS
        if self._events[slot] is None or self._events[slot][0] != it:
            raise RuntimeError(f"row {it} is not available")
        self._events[slot][1].synchronize()
        h = self._rows_host[slot].numpy().copy()
        return self._decode(s=(h[3], h[4], h[0]), tp=int(h[5]), fp=int(h[6]), pu=float(h[1]), pv=float(h[2]), reg=reg)

    def previous_factors(self):
        """The iterate before the last enqueued update: what a loop that ran one iteration past its stopping rule returns."""
        return self.Up64[: self.m, : self.k].cpu().numpy(), self.Vp64[: self.n, : self.k].cpu().numpy()

    def _decode(self, s, tp, fp, pu, pv, reg):
        cells = float(self.m_total) * float(self.n)
        fn = int(self.sum_x) - tp
        counts = (tp, fp, fn, self.m_total * self.n - tp - fp - fn)
        rmse, mae = float(np.sqrt(s[1] / cells)), float(s[0] / cells)
        if self.link == L.LINK_KL:
            return float(s[2]), float(s[2]), 0.0, rmse, mae, counts
        rec = 0.5 * float(s[1])
        rg = float(reg) * (0.5 * pu + 0.5 * pv)
        return rec + rg, rec, rg, rmse, mae, counts

    def _side(self, which, reg):
        X = self.X
        if which == "V":
            bits, rows_pad, ldx, rows, cols, Fs, Fo, opad = X.bits_t, self.n_pad, X.ldxt, self.n, self.m, self.V, self.U, self.m_pad
            num, den_slabs, den, splits, orows = self.numV, self.denV_slabs, self.denV, self.splitsV, self.m
        else:
            bits, rows_pad, ldx, rows, cols, Fs, Fo, opad = X.bits, self.m_pad, X.ldx, self.m, self.n, self.U, self.V, self.n_pad
            num, den_slabs, den, splits, orows = self.numU, self.denU_slabs, self.denU, self.splitsU, self.n
        stride = rows_pad * self.kp
        if self.mfma == "bf16":
            ws_s, ws_o = (self.wsV, self.wsU) if which == "V" else (self.wsU, self.wsV)
            check(lib.bmf_link_pass16(ptr(bits), rows_pad, ldx, rows, cols, ptr(ws_s), ptr(ws_o), opad, self.kp, self.link, self.lamda,
                                      ptr(num), ptr(den_slabs), stride, splits, _stream()), "bmf_link_pass16")
        else:
            check(lib.bmf_link_pass(ptr(bits), rows_pad, ldx, rows, cols, ptr(Fs), ptr(Fo), opad, self.kp, self.link, self.lamda,
                                    ptr(num), ptr(den_slabs), stride, splits, _stream()), "bmf_link_pass")
        if self.link == L.LINK_SIGMOID:
            check(lib.bmf_reduce_slabs(ptr(den_slabs), stride, splits, stride, ptr(den), None, _stream()), "bmf_reduce_slabs")
        else:  # KL: the denominator is the column-sum vector of the other factor, the same for every row
            check(lib.bmf_colsum_fill(ptr(Fo), orows, self.kp, ptr(self.colsum), ptr(den), rows_pad, _stream()), "bmf_colsum_fill")
        if self.sharded and which == "V":
            # each rank has contracted over its own rows only: the V-side numerator (slabs summed locally first) and denominator
            # are sums over the ranks -- the exchange of an iteration
            import torch.distributed as dist
            check(lib.bmf_reduce_slabs(ptr(num), stride, splits, stride, ptr(self.numV_red), None, _stream()), "bmf_reduce_slabs")
            dist.all_reduce(self.numV_red, group=self.group)
            dist.all_reduce(den, group=self.group)
            self._epilogue(which, self.mode, reg, num=self.numV_red, splits=1)
        else:
            self._epilogue(which, self.mode, reg)
        self._split(which)

    def prepare(self):
        with torch.cuda.device(self.device):
            self._epilogue("V", L.MODE_PREPARE, 0.0)
            self._epilogue("U", L.MODE_PREPARE, 0.0)
            self._split("V")

    def update(self, reg):
        """V then U (Gauss-Seidel): U's pass sees the new V."""
        with torch.cuda.device(self.device):
            self._side("V", reg)
            self._side("U", reg)

    def scalars(self, reg):
        """(error, rec_error, reg_error, RMSE, MAE, (TP, FP, FN, TN)) of the current state."""
        X = self.X
        with torch.cuda.device(self.device):
            self.sums.zero_()
            ob = ptr(self.obs_bits.bits) if self.obs_bits is not None else None
            if self.mfma == "bf16":
                check(lib.bmf_link_sums16(ptr(X.bits), self.m_pad, X.ldx, self.m, self.n, ptr(self.wsU), ptr(self.wsV), self.n_pad,
                                          self.kp, self.link, self.lamda, ob, ptr(self.sums), _stream()), "bmf_link_sums16")
            else:
                check(lib.bmf_link_sums(ptr(X.bits), self.m_pad, X.ldx, self.m, self.n, ptr(self.U), ptr(self.V), self.n_pad, self.kp,
                                        self.link, self.lamda, ob, ptr(self.sums), _stream()), "bmf_link_sums")
            self.counts.zero_()
            check(lib.bmf_cover_count(ptr(X.bits), X.m_pad, X.ldx, X.n_pad // 32, ptr(self.ubits), ptr(self.vcolbits),
                                      X.n_pad // 32, self.kp, ptr(self.counts), None, _stream()), "bmf_cover_count")
            out = self._scal   # one synchronising read for everything
            out[0:3] = self.sums[:3]
            out[3:5] = self.counts[:2].double()
            out[5] = self.partU[:, 0].sum()
            out[6] = self.partV[:, 0].sum()
            if self.sharded:   # everything but the V partial (entry 6, replicated) is a sum over the ranks' rows
                import torch.distributed as dist
                loc = out[0:6].clone()
                dist.all_reduce(loc, group=self.group)
                out[0:6] = loc
            hv = out.cpu().numpy()
            s = hv[0:3]
            tp, fp = int(hv[3]), int(hv[4])
            pu, pv = float(hv[5]), float(hv[6])
        return self._decode(s, tp, fp, pu, pv, reg)
