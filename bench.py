#!/usr/bin/env python3
"""Headline benchmark: BinaryMF-Penalty multiplicative-update iterations/s on a 100 000 x 20 000 Boolean X, k = 64.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" is one full iteration of PyBMF's loop body (models/BinaryMFPenalty.py:82-115) with MAE switched off
(the north-star four-GEMM definition): V update, U update, error / rec_error / reg_error scalars,
Boolean cover counts, regulariser growth and the device-side early-stop test.  X lives in HBM as bits before the
timed region starts.  N > 1 row-shards X (strong scaling: the problem is fixed) with two all-reduces per step.

Prints ONE JSON line (rank 0) with the driver's contract plus `roofline` (dominant kernel = the bits GEMM, timed
with HIP events on its own stream inside the timed region) and `cpu_baseline` (the NumPy oracle, literal reference
association, on a bounded row sample, rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BF16_DENSE_PEAK_TFLOPS = 2500.0   # MI355X_MICROARCH.md: ~2.5 PF dense bf16 MFMA
FP32_MFMA_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md: fp32-input MFMA = fp32 vector peak
HBM_PEAK_BYTES = 8.0e12           # MI355X_MICROARCH.md: HBM3E ~8 TB/s


def host_init(mean_x, m, n, k, seed):
    """init_method='normal' + normalize_method='balance' + zeros -> eps, as models/ContinuousModel.py:66-75,117-123,33-36
    (V drawn before U from one RandomState(seed))."""
    rng = np.random.RandomState(seed)
    avg = np.sqrt(mean_x / k)
    V = np.abs(avg * rng.standard_normal(size=(n, k)))
    U = np.abs(avg * rng.standard_normal(size=(m, k)))
    dU, dV = np.sqrt(U.max(axis=0)), np.sqrt(V.max(axis=0))
    U = U * dV / dU
    V = V * dU / dV
    eps = np.finfo(np.float64).eps
    U[U == 0] = eps
    V[V == 0] = eps
    return U, V


def cpu_baseline(Xs, U0s, V0, reg, m_full, iters=3):
    """The oracle's update on a row sample; per-iteration time scales linearly in m (n, k unchanged), so it/s at full size =
    it/s * m_s / m.  Two variants: the literal association (reference operation order incl. the all-ones mask
    multiply) -- the reported baseline -- and the re-associated one (the association the HIP path uses).  Median of `iters`
    warm iterations each."""
    import oracle as orc
    Xf = Xs.astype(np.float64)
    Xi = Xs.astype(np.int64)
    W = np.ones_like(Xf)

    def timed(update_V, update_U, errors):
        U, V = U0s.copy(), V0.copy()
        V = update_V(U, V)  # warm
        U = update_U(U, V)
        ts = []
        for _ in range(iters):
            t0 = time.perf_counter()
            V = update_V(U, V)
            U = update_U(U, V)
            errors(U, V)
            orc.confusion_counts(Xi, orc.boolean_product(U, V, 0.5, 0.5))
            ts.append(time.perf_counter() - t0)
        return float(np.median(ts))

    r = np.float64(reg)
    t_lit = timed(lambda U, V: orc.penalty_update_V(Xf, W, U, V, r), lambda U, V: orc.penalty_update_U(Xf, W, U, V, r),
                  lambda U, V: orc.penalty_errors(Xf, W, U, V, reg))
    sx = float(Xf.sum())

    def trace_errors(U, V):  # rec_error without the m x n product: 0.5 (sum X - 2 <X V, U> + <U^T U, V^T V>) for Boolean X
        return 0.5 * (sx - 2.0 * float(((Xf @ V) * U).sum()) + float(((U.T @ U) * (V.T @ V)).sum())) + reg * (orc.reg_term(U) + orc.reg_term(V))
    t_re = timed(lambda U, V: orc.penalty_update_V_reassoc(Xf, U, V, r), lambda U, V: orc.penalty_update_U_reassoc(Xf, U, V, r), trace_errors)
    scale = Xs.shape[0] / m_full
    return (1.0 / t_lit) * scale, t_lit, (1.0 / t_re) * scale, t_re


def few_blas_threads():
    """The small NumPy cross-checks between the GPU legs must not wake an OpenBLAS pool of one thread per host core: on a box whose
    cgroup grants a fraction of those cores the spinning workers delayed the wake-up of the thread waiting on the GPU by tens of
    milliseconds (seen as 2-3x slower secondary legs, at random).  The CPU baseline at the end uses every core it can get."""
    try:
        from threadpoolctl import threadpool_limits
        return threadpool_limits(limits=4)
    except ImportError:
        import contextlib
        return contextlib.nullcontext()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--m", type=int, default=100_000)
    ap.add_argument("--n", type=int, default=20_000)
    ap.add_argument("--k", type=int, default=64)
    ap.add_argument("--operands", default="i8x3", choices=["i8x3", "i8x2", "f16x2", "bf16x3", "bf16x2"],
                    help="factor operand format of the two bits GEMMs: 3 / 2 planes of int8 digits with exact int32 accumulation "
                         "(23 / 15 significant bits), two column-scaled fp16 addends (22 bits), or 3 / 2 bf16 addends (24 / 16 bits)")
    ap.add_argument("--mae", type=int, default=0, help="1: also run the residual (MAE) pass every step")
    ap.add_argument("--secondary", type=int, default=1, help="0: skip the secondary legs (with_mae, updates_only) -- profiling runs")
    ap.add_argument("--cpu-rows", type=int, default=4096, help="row sample of the CPU baseline (0 = skip)")
    ap.add_argument("--alt-operands", default="f16x2", choices=["none", "i8x3", "i8x2", "f16x2", "bf16x3", "bf16x2"],
                    help="also time the loop with this operand format (N=1 only)")
    args = ap.parse_args()
    opnd = {"f16x2": ("f16", 2), "bf16x3": ("bf16", 3), "bf16x2": ("bf16", 2), "i8x3": ("i8", 3), "i8x2": ("i8", 2)}
    args.panel, args.terms = opnd[args.operands]

    import torch
    import torch.distributed as dist
    torch.set_num_threads(4)   # host-side tensor ops here are tiny; a wide OpenMP pool spinning after them only competes with the GPU wait
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    # BMF_BENCH_REHEARSAL=1: functional rehearsal of the N > 1 code path on a box with ONE GPU -- every rank on cuda:0, exchange
    # over gloo (RCCL refuses two ranks on one device).  The numbers of such a run mean nothing.
    rehearsal = os.environ.get("BMF_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local = 0
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    # BMF_FORCE_SHARDED=1 rehearses the multi-GPU code path (RCCL init, all-reduces, Python-driven loop) with one rank
    sharded = world > 1 or os.environ.get("BMF_FORCE_SHARDED") == "1"
    if sharded:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    from pybmf_amd import _lib as L
    from pybmf_amd.engine import BitMatrix, MUEngine, shard_rows
    from pybmf_amd.generators import PlantedBooleanOnDevice

    m, n, k = args.m, args.n, args.k
    K, W = args.steps, args.warmup
    # planted factors, density 0.067 so that X has ~25 % ones at k = 64, noise [0.05, 0.01]
    dens = 0.067 if k >= 32 else 0.2
    gen = PlantedBooleanOnDevice(m, n, k, density=(dens, dens), seed=1000, noise=(0.05, 0.01), noise_seed=2000, device=device)
    lo, hi = shard_rows(m, rank, world)
    X = BitMatrix(gen, device, row_lo=lo, row_hi=hi)
    del gen
    reg0, growth, max_reg = 1.0, 1.02, 1e10
    max_iter = W + K + 1
    eng = MUEngine(X, k=k, mode=L.MODE_PENALTY, terms=args.terms, with_mae=bool(args.mae), tol=float(os.environ.get('BMF_BENCH_TOL', '0.01')), min_diff=0.0,
                   max_iter=max_iter, sharded=sharded, panel=args.panel)
    U0, V0 = host_init(eng.sum_x / (float(m) * n), m, n, k, seed=2024)
    eng.load_factors(U0[lo:hi], V0)
    regs, r = [], np.float64(reg0)
    for _ in range(W + K):
        regs.append(float(r))
        r = min(r * np.float64(growth), np.float64(max_reg))

    def barrier():
        torch.cuda.synchronize()
        if sharded:
            dist.barrier()
            torch.cuda.synchronize()

    eng.prepare(regs[0])
    eng.run(regs[:W], it0=1)
    barrier()
    L.check(L.lib.bmf_timer_enable(2 * K + 8))
    L.check(L.lib.bmf_timer_stride(3))   # sample 1 launch in 3 (alternates between X V and X^T U): event pairs cost stream time
    t0 = time.perf_counter()
    eng.run(regs[W:], it0=1 + W)
    barrier()
    dt = time.perf_counter() - t0
    import ctypes as C
    n_launch, gemm_ms = C.c_int(0), C.c_double(0.0)
    L.check(L.lib.bmf_timer_read(C.byref(n_launch), C.byref(gemm_ms)))
    L.check(L.lib.bmf_timer_disable())
    if sharded:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    log, stop = eng.read_log()
    if os.environ.get("BMF_NO_CHECK") != "1":  # (timing-only kernel experiments produce wrong numbers on purpose)
        assert log.shape[0] == 1 + W + K and stop == 0, (log.shape, stop)
        assert np.isfinite(log[:, :6]).all()
    last = log[-1]

    # independent check of the logged (trace-form) rec_error: direct residual pass on the GPU, and NumPy fp64 on the
    # first rows of rank 0's shard
    chk = {}
    sums = torch.zeros(4, dtype=torch.float64, device=device)
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    L.check(L.lib.bmf_residual_sums(L.ptr(X.bits), X.m_pad, X.ldx, X.m, X.n, L.ptr(eng.U), L.ptr(eng.V), eng.kp, L.ptr(sums),
                                    None, stream))
    if sharded:
        dist.all_reduce(sums)
    direct = 0.5 * float(sums[1].item())
    chk["rec_error_trace_vs_direct_rel"] = abs(direct - last[L.LOG_REC]) / direct
    if rank == 0:
        rs = min(1024, X.m)
        Uh, Vh = eng.factors()
        Xs = X.rows_dense_u8(0, rs).astype(np.float64)
        with few_blas_threads():
            host = float(((Xs - Uh[:rs] @ Vh.T) ** 2).sum())
        sums.zero_()
        L.check(L.lib.bmf_residual_sums(L.ptr(X.bits), X.m_pad, X.ldx, rs, X.n, L.ptr(eng.U), L.ptr(eng.V), eng.kp, L.ptr(sums),
                                        None, stream))
        chk["residual_gpu_vs_numpy_fp64_rel"] = abs(float(sums[1].item()) - host) / host

    if rank != 0:
        if sharded:
            dist.destroy_process_group()
        return

    its = K / dt
    traffic = None  # HBM-side bytes per launch from the committed PMC passes (separate rocprofv3 --pmc runs)
    pmc = os.path.join(ROOT, "profiles", "r01_pmc_xf_bits.json")
    if os.path.exists(pmc) and (m, n, k, world) == (100_000, 20_000, 64, 1):
        pj = json.load(open(pmc))
        if pj.get("operands", "bf16x3") == args.operands:
            traffic = pj.get("traffic_bytes_per_launch")
    launches = max(n_launch.value, 1)
    avg_ms = gemm_ms.value / launches
    # algorithmic flops of one bits-GEMM launch on this rank: 2 * m_local * n * k (X V and X^T U are the same count)
    flops_launch = 2.0 * X.m * n * k
    achieved = flops_launch / (avg_ms * 1e-3) / 1e12
    bytes_launch = X.m * n / 8.0 + 0.5 * (X.m + n) * k * (2.0 * args.terms + 4.0)
    out = {
        "metric": "MU iterations/sec (BinaryMF-Penalty, 100k x 20k Boolean, k=64)" if (m, n, k) == (100_000, 20_000, 64)
                  else f"MU iterations/sec (BinaryMF-Penalty, {m}x{n} Boolean, k={k})",
        "value": its, "unit": "iterations/s", "n_gpus": world, "steps": K, "warmup": W,
        "ms_per_step": 1e3 * dt / K, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": f"{args.operands} operands (bits x split-{args.panel} MFMA), fp32 accumulate, fp64 master factors and scalars",
        "data": "synthetic (planted Boolean factors + flip noise, generated on device; SURVEY 8d)",
        "config": {"workload": f"BinaryMF-Penalty MU, {m}x{n} dense Boolean X, k={k}, reg=1 growth=1.02, init normal+balance seed 2024",
                   "mae_pass": bool(args.mae), "operands": args.operands, "row_sharding": f"{world} x {X.m} rows",
                   "splits_xv": eng.splits_xv, "splits_xtu": eng.splits_xtu},
        "roofline": {"kernel": "xf_bits_kernel (X V and X^T U)", "bound": "mfma", "achieved": achieved,
                     "peak": BF16_DENSE_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / BF16_DENSE_PEAK_TFLOPS,
                     "traffic": traffic, "traffic_source": "profiles/r01_pmc_xf_bits.json (FETCH_SIZE x2 + WRITE_SIZE, fabric side incl. Infinity-Cache hits)" if traffic else None, "launches_timed": launches, "avg_launch_ms": avg_ms,
                     "algorithmic_flops_per_launch": flops_launch, "hw_flops_factor": args.terms,
                     "frac_of_fp32_mfma_peak": achieved / FP32_MFMA_PEAK_TFLOPS,
                     # the other roofline of SURVEY 8d: algorithmic bytes of one launch (X as bits once, the factor panel, the
                     # fp32 result; mean of the X V and X^T U launches) against HBM
                     "algorithmic_bytes_per_launch": bytes_launch, "frac_of_hbm_peak": bytes_launch / (avg_ms * 1e-3) / HBM_PEAK_BYTES,
                     "gemm_share_of_step": 2.0 * avg_ms * 1e-3 * K / dt},
        "iteration_vs_fp32_mfma_roofline": its / (FP32_MFMA_PEAK_TFLOPS * 1e12 * world / (4.0 * m * n * k + 4.0 * (m + n) * k * k)),
        "final": {"iter": int(last[L.LOG_ITER]), "error": last[L.LOG_ERROR], "rec_error": last[L.LOG_REC],
                  "reg_error": last[L.LOG_REGERR], "TP": int(last[L.LOG_TP]), "FP": int(last[L.LOG_FP])},
        "checks": chk,
    }
    if world == 1 and not sharded and args.alt_operands not in ("none", args.operands):
        # secondary number, same data and schedule, outside the timed region of `value`: another operand format
        # (bf16x3 = fp32-exact operands; the difference of the final factors between the two runs is reported)
        panel2, terms2 = opnd[args.alt_operands]
        eng2 = MUEngine(X, k=k, mode=L.MODE_PENALTY, terms=terms2, with_mae=bool(args.mae), tol=0.01, min_diff=0.0,
                        max_iter=max_iter, panel=panel2)
        eng2.load_factors(U0[lo:hi], V0)
        eng2.prepare(regs[0])
        eng2.run(regs[:W], it0=1)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        eng2.run(regs[W:], it0=1 + W)
        torch.cuda.synchronize()
        dt2 = time.perf_counter() - t1
        log2, _ = eng2.read_log()
        U2, V2 = eng2.factors()
        Uh, Vh = eng.factors()
        with few_blas_threads():
            du, dv = float(np.linalg.norm(U2 - Uh) / np.linalg.norm(Uh)), float(np.linalg.norm(V2 - Vh) / np.linalg.norm(Vh))
        out["alt"] = {"operands": args.alt_operands, "value": K / dt2, "ms_per_step": 1e3 * dt2 / K,
                      "rel_diff_U_vs_main": du, "rel_diff_V_vs_main": dv,
                      "rel_diff_rec_error_vs_main": float(abs(log2[-1, L.LOG_REC] / last[L.LOG_REC] - 1.0))}
        del eng2
    if world == 1 and not sharded and not args.mae and args.secondary:
        # the reference's loop also logs MAE every iteration (BinaryMFPenalty.py:71,97): the same loop with the MAE pass on
        # (split-bf16 MFMA, csrc/mae.hip), outside the timed region of `value` ("a second line reports the MAE-on rate")
        eng3 = MUEngine(X, k=k, mode=L.MODE_PENALTY, terms=args.terms, with_mae=True, tol=float(os.environ.get('BMF_BENCH_TOL', '0.01')),
                        min_diff=0.0, max_iter=max_iter, panel=args.panel)
        eng3.load_factors(U0[lo:hi], V0)
        eng3.prepare(regs[0])
        eng3.run(regs[:W], it0=1)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        eng3.run(regs[W:], it0=1 + W)
        t_enq = time.perf_counter() - t1
        torch.cuda.synchronize()
        dt3 = time.perf_counter() - t1
        if os.environ.get("BMF_BENCH_DEBUG"):
            print(f"[debug] with_mae leg: enqueue {1e3 * t_enq:.1f} ms, total {1e3 * dt3:.1f} ms", file=sys.stderr)
        log3, _ = eng3.read_log()
        out["with_mae"] = {"value": K / dt3, "ms_per_step": 1e3 * dt3 / K, "MAE": float(log3[-1, L.LOG_MAE]),
                           "RMSE": float(log3[-1, L.LOG_RMSE])}
        del eng3
        # "updates only" (SURVEY 8d): V- and U-update with their error terms, no Boolean cover count, no MAE -- what the CPU
        # number above would be compared with if the scores were left out
        eng4 = MUEngine(X, k=k, mode=L.MODE_PENALTY, terms=args.terms, with_mae=False, tol=float(os.environ.get('BMF_BENCH_TOL', '0.01')),
                        min_diff=0.0, max_iter=max_iter, panel=args.panel)
        eng4.st.updates_only = 1
        eng4.load_factors(U0[lo:hi], V0)
        eng4.prepare(regs[0])
        eng4.run(regs[:W], it0=1)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        eng4.run(regs[W:], it0=1 + W)
        t_enq = time.perf_counter() - t1
        torch.cuda.synchronize()
        dt4 = time.perf_counter() - t1
        if os.environ.get("BMF_BENCH_DEBUG"):
            print(f"[debug] updates_only leg: enqueue {1e3 * t_enq:.1f} ms, total {1e3 * dt4:.1f} ms", file=sys.stderr)
        log4, _ = eng4.read_log()
        out["updates_only"] = {"value": K / dt4, "ms_per_step": 1e3 * dt4 / K,
                               "rel_diff_error_vs_main": abs(float(log4[-1, L.LOG_ERROR]) - last[L.LOG_ERROR]) / last[L.LOG_ERROR]}
        del eng4
    if world == 1 and args.cpu_rows > 0:
        rs = min(args.cpu_rows, X.m)
        Xs = X.rows_dense_u8(0, rs)
        v, t, v_re, t_re = cpu_baseline(Xs, U0[:rs], V0, reg0, m)
        try:
            from threadpoolctl import threadpool_info
            thr = max([p.get("num_threads", 1) for p in threadpool_info()] or [os.cpu_count()])
        except Exception:
            thr = os.cpu_count()
        out["cpu_baseline"] = {"value": v, "unit": "iterations/s", "cores": int(thr), "kind": "port",
                               "sample": f"first {rs} of {m} rows (n={n}, k={k} unchanged), literal reference association "
                                         f"incl. all-ones mask, fp64 NumPy/OpenBLAS, median of 3 full iterations "
                                         f"({t:.2f} s each), scaled by {rs}/{m}",
                               "reassociated": {"value": v_re, "seconds_per_sample_iteration": t_re,
                                                "note": "same sample, the association the HIP path uses (V (U^T U) instead of "
                                                        "(U V^T)^T U, trace-form rec_error)"}}
    print(json.dumps(out))
    if sharded:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
