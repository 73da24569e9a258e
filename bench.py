#!/usr/bin/env python3
"""Headline benchmark: BinaryMF-Penalty multiplicative-update iterations/s on a 100 000 x 20 000 Boolean X, k = 64.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" is one full iteration of PyBMF's loop body (models/BinaryMFPenalty.py:82-115) with MAE switched off
(the north-star four-GEMM definition): V update, U update, error / rec_error / reg_error scalars,
Boolean cover counts, regulariser growth and the device-side early-stop test.  X lives in HBM as bits before the
timed region starts.  N > 1 row-shards X (strong scaling: the problem is fixed) with one exchange per step.

Prints ONE JSON line (rank 0) with the driver's contract plus
  `roofline`      dominant kernel = the bits GEMM, timed with HIP events on its own stream inside the timed region; `traffic` =
                  fabric-side bytes per launch from rocprofv3 PMC passes of THIS build, made by a child process after the timed
                  region (separate --pmc runs, FETCH_SIZE x 2 + WRITE_SIZE as MI355X_MICROARCH.md prescribes);
  `cpu_baseline`  the NumPy oracle, literal reference association, rank 0, N = 1 only, on the largest row sample that host RAM
                  and a ~25 s budget allow;
  `checks`        the GPU against itself (trace-form vs direct residual) AND against the fp64 oracle: one more update from the
                  final state re-computed exactly on a row / column sample (`oracle_step_rel_U/V`);
  `secondary`     the other BASELINE.json configurations, driver-timed: C1 fit() ms, C2 WNMF it/s, C5 line-search it/s;
  `repeat`        three more timed legs of K steps (outside `value`) so that the rate can be corroborated.
"""
import argparse
import contextlib
import ctypes as C
import io
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# MI355X_MICROARCH.md, Matrix cores: dense peaks.  I8 runs at twice the BF16 rate per clock (no spec line of its own).
MFMA_PEAK_TFLOPS = {"i8": 5000.0, "f16": 2500.0, "bf16": 2500.0}
FP32_MFMA_PEAK_TFLOPS = 157.3     # fp32-input MFMA = fp32 vector peak (the north star's reference roofline)
HBM_PEAK_BYTES = 8.0e12           # HBM3E ~8 TB/s
OPND = {"f16x2": ("f16", 2), "bf16x3": ("bf16", 3), "bf16x2": ("bf16", 2), "i8x3": ("i8", 3), "i8x2": ("i8", 2)}


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


def few_blas_threads():
    """The small NumPy cross-checks between the GPU legs must not wake an OpenBLAS pool of one thread per host core: on a box whose
    cgroup grants a fraction of those cores the spinning workers delayed the wake-up of the thread waiting on the GPU by tens of
    milliseconds.  The CPU baseline at the end uses every core it can get."""
    try:
        from threadpoolctl import threadpool_limits
        return threadpool_limits(limits=4)
    except ImportError:
        return contextlib.nullcontext()


def blas_threads():
    try:
        from threadpoolctl import threadpool_info
        return max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:
        return None


# ---------------------------------------------------------------------------------------------------------------------
# CPU baseline (oracle = test infrastructure; here only as the timed CPU reference, outside every GPU timing)
# ---------------------------------------------------------------------------------------------------------------------
def cpu_baseline(X, U0, V0, reg, m_full, budget_s=25.0):
    """The oracle's iteration on the first rows of X; time is linear in the row count (n, k fixed), so it/s at full size =
    it/s * rows / m.  Literal association = the reference's operation order incl. the all-ones mask multiply (five m x n fp64
    temporaries); the row count is the largest that fits half the available host RAM and the time budget, found from a short
    calibration run.  Also the re-associated "best CPU" variant (the association the HIP path uses, trace-form error, Boolean
    product by a float BLAS GEMM)."""
    import oracle as orc
    try:
        import psutil
        avail = float(psutil.virtual_memory().available)
    except Exception:
        avail = 16e9
    n, k = V0.shape

    def literal_iter(Xf, Xi, W, U, V):
        V = orc.penalty_update_V(Xf, W, U, V, np.float64(reg))
        U = orc.penalty_update_U(Xf, W, U, V, np.float64(reg))
        orc.penalty_errors(Xf, W, U, V, reg)
        orc.confusion_counts(Xi, orc.boolean_product(U, V, 0.5, 0.5))
        return U, V

    def reassoc_iter(Xf, Xb, sx, U, V):
        V = orc.penalty_update_V_reassoc(Xf, U, V, np.float64(reg))
        U = orc.penalty_update_U_reassoc(Xf, U, V, np.float64(reg))
        _ = 0.5 * (sx - 2.0 * float(((Xf @ V) * U).sum()) + float(((U.T @ U) * (V.T @ V)).sum())) + reg * (orc.reg_term(U) + orc.reg_term(V))
        pd = orc.boolean_product_blas(U, V, 0.5, 0.5)
        tp = int(np.count_nonzero(pd & Xb))
        _ = (tp, int(np.count_nonzero(pd)) - tp)
        return U, V

    def run(rows, fn, prep, iters):
        Xs = X.rows_dense_u8(0, rows)
        args = prep(Xs)
        U, V = U0[:rows].copy(), V0.copy()
        U, V = fn(*args, U, V)   # warm
        ts = []
        for _ in range(iters):
            t0 = time.perf_counter()
            U, V = fn(*args, U, V)
            ts.append(time.perf_counter() - t0)
        return float(np.median(ts))

    prep_lit = lambda Xs: (Xs.astype(np.float64), Xs.astype(np.int64), np.ones(Xs.shape))  # noqa: E731
    prep_re = lambda Xs: (Xs.astype(np.float64), Xs.astype(bool), float(Xs.sum()))         # noqa: E731
    cal = min(2048, X.m)
    t_cal = run(cal, literal_iter, prep_lit, 1)
    per_row = t_cal / cal
    rows_ram = int(0.5 * avail / (6 * 8.0 * n))          # X, W, W o X, U V^T, W o (U V^T) + slack, fp64
    rows_time = int(budget_s / 4.0 / max(per_row, 1e-9))  # one warm + three timed iterations
    rows = max(cal, min(X.m, rows_ram, rows_time))
    t_lit = run(rows, literal_iter, prep_lit, 3) if rows > cal else t_cal
    rows_re = min(X.m, max(rows, min(int(0.5 * avail / (3 * 8.0 * n)), 8 * rows)))
    t_re = run(rows_re, reassoc_iter, prep_re, 3)
    return {"value": (1.0 / t_lit) * rows / m_full, "unit": "iterations/s", "cores": os.cpu_count(), "kind": "port",
            "blas_threads": blas_threads(), "host_ram_available_gb": round(avail / 1e9, 1),
            "sample": f"first {rows} of {m_full} rows (n={n}, k={k} unchanged): the largest sample within half the available host RAM "
                      f"({rows_ram} rows) and a {budget_s:.0f} s budget ({rows_time} rows at {per_row * 1e3:.3f} ms per row, calibrated on {cal} "
                      f"rows); literal reference association incl. all-ones mask, fp64 NumPy/OpenBLAS, median of 3 iterations "
                      f"({t_lit:.2f} s each), scaled by rows/m",
            "reassociated": {"value": (1.0 / t_re) * rows_re / m_full, "rows": rows_re, "seconds_per_sample_iteration": t_re,
                             "note": "the association the HIP path uses (V (U^T U) instead of (U V^T)^T U), trace-form rec_error, Boolean "
                                     "product by a float32 BLAS GEMM > 0: the best CPU formulation of the same iteration"}}


# ---------------------------------------------------------------------------------------------------------------------
# oracle check of one update at full size, exactly, on a sample (the updates are independent per row / per column)
# ---------------------------------------------------------------------------------------------------------------------
def oracle_step_check(X, eng, reg, it, n_rows=1024, n_cols=256, seed=3):
    """One more update from the engine's CURRENT state, re-computed in fp64 by the oracle on a sample: V_new[J] from X[:, J],
    U_old, V_old[J]; U_new[I] from X[I, :], U_old[I] and the engine's V_new (PyBMF/models/BinaryMFPenalty.py:136-163).
    Returns the relative Frobenius distances (V sample, U sample).  Advances the engine by one iteration."""
    import torch
    import oracle as orc
    rs = np.random.RandomState(seed)
    I = np.sort(rs.choice(X.m, size=min(n_rows, X.m), replace=False))
    J = np.sort(rs.choice(X.n, size=min(n_cols, X.n), replace=False))
    U_old, V_old = eng.factors()
    eng.run([reg], it0=it)
    U_new, V_new = eng.factors()
    bj = X.bits_t[torch.from_numpy(J).to(X.device)].cpu().numpy().view(np.uint8)
    XJ = np.ascontiguousarray(np.unpackbits(bj, axis=1, bitorder="little")[:, : X.m].T.astype(np.float64))
    bi = X.bits[torch.from_numpy(I).to(X.device)].cpu().numpy().view(np.uint8)
    XI = np.unpackbits(bi, axis=1, bitorder="little")[:, : X.n].astype(np.float64)
    with few_blas_threads():
        Vs = orc.penalty_update_V_reassoc(XJ, U_old, V_old[J], reg)
        Us = orc.penalty_update_U_reassoc(XI, U_old[I], V_new, reg)
        rel = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))  # noqa: E731
        return rel(V_new[J], Vs), rel(U_new[I], Us)


# ---------------------------------------------------------------------------------------------------------------------
# fabric traffic of the dominant kernel: rocprofv3 PMC passes of this very build, in child processes
# ---------------------------------------------------------------------------------------------------------------------
def measure_traffic(args, timeout_s=150):
    """FETCH_SIZE and WRITE_SIZE of the bits GEMM, mean per launch, from two separate `rocprofv3 --pmc` child runs of this script
    (`--pmc-child`: the timed loop only).  gfx950: FETCH_SIZE counts 64 B per 128-B request -> x 2 (MI355X_MICROARCH.md, HBM);
    both counters are in KiB.  Returns (bytes per launch, note) or (None, reason)."""
    import csv
    import glob
    exe = shutil.which("rocprofv3")
    if exe is None:
        return None, "rocprofv3 not found"
    tot = {}
    tmp = tempfile.mkdtemp(prefix="bmf_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, counter)
            cmd = [exe, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", out, "--", sys.executable,
                   os.path.join(ROOT, "bench.py"), "--pmc-child", "--steps", "4", "--warmup", "1", "--operands", args.operands,
                   "--m", str(args.m), "--n", str(args.n), "--k", str(args.k)]
            r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=timeout_s)
            vals = []
            for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if "xf_bits" in row["Kernel_Name"] and row["Counter_Name"] == counter:
                        vals.append(float(row["Counter_Value"]))
            if not vals:
                return None, f"no {counter} rows (rc {r.returncode}): {r.stderr[-200:]}"
            tot[counter] = sum(vals) / len(vals)
        return (2.0 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1024.0, (
            f"live: rocprofv3 --pmc child runs of this build (FETCH_SIZE {tot['FETCH_SIZE']:.4g} KiB x 2 + WRITE_SIZE "
            f"{tot['WRITE_SIZE']:.4g} KiB per launch, fabric side incl. Infinity-Cache hits)")
    except Exception as e:  # noqa: BLE001
        return None, f"{type(e).__name__}: {e}"
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


# ---------------------------------------------------------------------------------------------------------------------
# secondary configurations (BASELINE.json configs[0], [1], [4]); N = 1 only, outside the timed region of `value`
# ---------------------------------------------------------------------------------------------------------------------
FIT = dict(task="reconstruction", show_logs=False, show_result=False, save_model=False)


def secondary_c1():
    """configs[0]: BinaryMFPenalty.fit() on the 1000 x 500 generator matrix, k = 8, 21 updates, whole call (upload, packing, loop,
    log tables).  The reference needs 4.6 s for it."""
    import torch
    from pybmf_amd.generators import SyntheticMatrixGenerator
    from pybmf_amd.models import BinaryMFPenalty
    gen = SyntheticMatrixGenerator(m=1000, n=500, k=8, density=[0.2, 0.2])
    gen.generate(seed=1000)
    gen.add_noise(noise=[0.05, 0.01], seed=2000)
    times = []
    for _ in range(4):
        with contextlib.redirect_stdout(io.StringIO()):
            mdl = BinaryMFPenalty(k=8, W="full", reg=1, reg_growth=1.02, init_method="normal", normalize_method="balance", max_iter=20, seed=2024)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            mdl.fit(gen.X, **FIT)
            torch.cuda.synchronize()
            times.append(time.perf_counter() - t0)
    row = mdl.logs["updates"].iloc[-1]
    return {"config": "BinaryMF-Penalty fit(), 1000x500 generator matrix, k=8, 21 updates incl. upload and log tables",
            "fit_ms": 1e3 * min(times[1:]), "iterations_per_s": 21 / min(times[1:]), "final_error": float(row[("", "", "error")]),
            "reference_final_error": 11619.10566320379, "counts_TP_FP_FN_TN": [int(c) for c in mdl.counts[-1]],
            "reference_counts": [110012, 6743, 22187, 361058]}


def secondary_c2(iters=30):
    """configs[1]: WNMF multiplicative updates, 20 000 x 5 000 dense fp32 X, k = 32 (SURVEY 8d recipe); per iteration X is read
    twice (X V; X^T U with the residual sums for RMSE / MAE folded into the same pass): HBM-bound."""
    import torch
    from pybmf_amd.engine import RealMatrix, RealMUEngine
    m, n, k = 20000, 5000, 32
    rs = np.random.RandomState(0)
    X = ((rs.rand(m, 32) @ rs.rand(32, n)) / 32).astype(np.float32) + 0.01 * rs.rand(m, n).astype(np.float32)
    eng = RealMUEngine(RealMatrix(X, "cuda:0"), k, with_mae=True)
    r2 = np.random.RandomState(2024)
    avg = np.sqrt(X.mean() / k)
    V0 = np.abs(avg * r2.standard_normal((n, k)))
    U0 = np.abs(avg * r2.standard_normal((m, k)))
    eng.load_factors(U0, V0)
    eng.device_loop(max_iter=iters + 8)     # the C-side loop (bmf_wnmf_real_run): no host round trip inside
    eng.run(1, 4)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.run(4, 4 + iters)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    log, stop = eng.read_log()
    assert stop == 0 and log.shape[0] == 4 + iters, (stop, log.shape)
    e = (log[-1, 1], log[-1, 5], log[-1, 6])
    fused = getattr(eng, "_Urf", None) is not None
    passes = 2 if fused else 3
    bytes_it = float(passes) * X.nbytes
    # the two kernels that read X, on their own (HIP events on the launch stream, 20 launches each)
    from pybmf_amd import _lib as L
    R = eng.X
    Xt = R.tiled()[0]
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    kern = {}
    calls = [("xf_f32_tiled", lambda: L.lib.bmf_xf_f32_tiled(L.ptr(Xt), R.m_pad, R.n_pad, L.ptr(eng._Vfrag), eng.kp, L.ptr(eng.Mslab),
                                                            R.m_pad * eng.kp, eng.splits_xv, stream)),
             ("residual_sums_f32_tiled", lambda: L.lib.bmf_residual_sums_f32_tiled(L.ptr(Xt), R.m_pad, R.n_pad, L.ptr(eng.U), L.ptr(eng._Vrf),
                                                                                  eng.kp, L.ptr(eng.sums), stream))]
    if fused:   # X^T U with the residual sums in the same pass (what the loop runs instead of the two above on X^T / X)
        XTt = R.tiled()[1]
        calls.append(("xf_f32_tiled_resid", lambda: L.lib.bmf_xf_f32_tiled_resid(L.ptr(XTt), R.n_pad, R.m_pad, L.ptr(eng.UT), L.ptr(eng._Urf), L.ptr(eng.V),
                                                                                eng.kp, L.ptr(eng.Nslab), R.n_pad * eng.kp, eng.splits_xtu, L.ptr(eng.sums), stream)))
    for name, call in calls:
        for _ in range(3):
            L.check(call())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            L.check(call())
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 20
        kern[name] = {"us_per_launch": us, "GB_per_s": R.m_pad * R.n_pad * 4.0 / us / 1e3, "frac_of_hbm_peak": R.m_pad * R.n_pad * 4.0 / (us * 1e-6) / HBM_PEAK_BYTES}
    return {"config": "WNMF MU, 20000x5000 dense fp32, k=32, error + RMSE + MAE every iteration", "iterations_per_s": 1.0 / dt,
            "ms_per_iteration": 1e3 * dt, "error": float(e[0]),
            "roofline": {"bound": "hbm", "algorithmic_bytes_per_iteration": bytes_it, "achieved": bytes_it / dt / 1e9, "peak": HBM_PEAK_BYTES / 1e9,
                         "unit": "GB/s", "frac": bytes_it / dt / HBM_PEAK_BYTES,
                         "passes_over_X": passes,
                         "note": f"whole iteration ({passes} passes over X + epilogues, Grams, fragment re-ordering, finalize) against HBM; the passes alone: `kernels`"},
            "kernels": kern}


ML1M_PATH = os.path.join(os.path.expanduser("~"), ".pybmf", "data", "movielens", "ml-1m", "ratings.dat")


def load_movielens_1m(path=ML1M_PATH):
    """MovieLens-1M as the reference builds it (PyBMF/datasets/MovieLensData.py:36-40, 62-91): `uid::iid::rating::timestamp`
    lines, rows = sorted distinct user ids, columns = sorted distinct item ids, cell = rating > 0.5 (every rating is 1..5, so
    every rated cell is a one).  Returns a dense uint8 matrix (6040 x 3706 for the real file) or None when the file is absent."""
    if not os.path.isfile(path):
        return None
    uid, iid, rating = [], [], []
    with open(path, encoding="latin1") as f:
        for line in f:
            parts = line.rstrip("\n").split("::")
            if len(parts) < 3:
                continue
            uid.append(int(parts[0]))
            iid.append(int(parts[1]))
            rating.append(int(parts[2]))
    uid, iid, rating = np.asarray(uid), np.asarray(iid), np.asarray(rating)
    rows = np.unique(uid, return_inverse=True)[1]
    cols = np.unique(iid, return_inverse=True)[1]
    X = np.zeros((rows.max() + 1, cols.max() + 1), dtype=np.uint8)
    X[rows, cols] = (rating > 0.5).astype(np.uint8)
    return X


def secondary_c5():
    """configs[4]: BinaryMFThreshold line search on MovieLens-1M (6040 x 3706, k = 16): the real data when `ratings.dat` is in
    ~/.pybmf/data/movielens/ml-1m/ (where the reference's downloader puts it), else a shape / density-matched stand-in (the data
    set is a download and there is no network); factors from 20 WNMF updates.  `data` says which one ran."""
    from pybmf_amd.models import BinaryMFThreshold, WNMF
    rs = np.random.RandomState(11)
    m, n, k = 6040, 3706, 16
    X = load_movielens_1m()
    data = f"MovieLens-1M from {ML1M_PATH}, rating > 0.5" if X is not None else "stand-in: power-law row / column popularity, 1 000 209 expected ones (ratings.dat not present)"
    if X is None:
        pu, pv = rs.pareto(1.2, m) + 1, rs.pareto(1.2, n) + 1
        P = np.outer(pu / pu.sum(), pv / pv.sum())
        # cell probabilities min(s P, 1) with s chosen so that they SUM to 1 000 209 (clipping the popular rows / columns at 1 loses
        # mass: with s = 1 000 209 the round-3 stand-in had 536 438 ones, half the density it claimed)
        s = 1_000_209.0
        for _ in range(60):
            s *= 1_000_209.0 / np.minimum(P * s, 1.0).sum()
        X = (rs.rand(m, n) < np.minimum(P * s, 1.0)).astype(np.uint8)
    m, n = X.shape
    with contextlib.redirect_stdout(io.StringIO()):
        w = WNMF(k=k, W="full", init_method="normal", max_iter=20, seed=5)
        w.fit(X, **FIT)
        best = None
        for _ in range(2):
            model = BinaryMFThreshold(k=k, U=w.U.copy(), V=w.V.copy(), W="full", u=0.3, v=0.3, lamda=10, min_diff=1e-3, max_iter=30)
            t0 = time.perf_counter()
            model.fit(X, **FIT)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
    return {"config": f"BinaryMF-Thresholding line search, {m}x{n} (MovieLens-1M shape), k=16, lamda=10, whole fit() incl. upload", "data": data,
            "ones": int(X.sum()), "outer_iterations": int(model.n_iter), "fit_s": best, "iterations_per_s": model.n_iter / best, "u": float(model.u), "v": float(model.v)}


def secondary_widened(X, U0, V0):
    """One rate per widened engine so that they are measured rows, not only correctness rows: ELBMF's iPALM loop, the
    PNLPF and WNMF-KL update pairs (tile-fused link pass) on the headline bit matrix, the masked update at MovieLens-1M shape.
    All Python-driven loops, as their model classes drive them."""
    import torch
    from pybmf_amd import _lib as L
    from pybmf_amd.engine import BitMatrix, LinkMUEngine, MaskedMUEngine, SparseObs
    from pybmf_amd.palm import PalmEngine
    out = {}
    m, n, k = X.m, X.n, U0.shape[1]

    def roof(bound, per_it, dt, peak, unit, what):
        """Whole-iteration roofline of a widened engine: ALGORITHMIC flops (or bytes) of one iteration / its wall time, against the
        peak of the unit the dominant kernel runs on (no per-kernel timing here: these loops are timed as their model classes drive them)."""
        ach = per_it / dt / (1e12 if unit == "TFLOP/s" else 1e9)
        return {"bound": bound, "achieved": ach, "peak": peak, "unit": unit, "frac": ach / peak, "algorithmic_per_iteration": per_it,
                "scope": "whole iteration", "counts": what, "traffic": None}

    def timed(fn, iters, warm=2):
        for i in range(warm):
            fn(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(warm, warm + iters):
            fn(i)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / iters

    # ELBMF (PyBMF/models/ELBMF.py:107-163): two proximal steps, two refreshes (int8 planes, Gram, spectral norm, bits GEMM), scalars
    eng = PalmEngine(X, k, L.PALM_ELBMF, beta=0.0)
    eng.load_factors(U0, V0)
    res = {}

    # driven as ELBMF.iPALM drives it: one C call per iteration (bmf_palm_iterate), the scalars of iteration t read while t + 1 runs
    sched = lambda i: (0.01, 0.02 * 1.02 ** i, 0.01, 0.02 * 1.02 ** i)   # noqa: E731
    warm, iters = 3, 20
    for i in range(warm):
        eng.iterate(i, *sched(i))
    eng.row(warm - 1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.iterate(warm, *sched(warm))
    for i in range(warm, warm + iters):
        eng.iterate(i + 1, *sched(i + 1))
        res["s"] = eng.row(i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / (iters + 1)
    out["elbmf_ipalm"] = {"config": f"ELBMF iPALM loop, {m}x{n} Boolean, k={k}, beta=0, int8 x3 operands, scores every iteration",
                          "iterations_per_s": 1.0 / dt, "ms_per_iteration": 1e3 * dt, "error": float(res["s"][0]),
                          "roofline": roof("mfma", 4.0 * m * n * k, dt, MFMA_PEAK_TFLOPS["i8"], "TFLOP/s",
                                           "the two bits GEMMs X V and X^T U (2 m n k each), as the headline counts them; the int8 x3 "
                                           "digit planes issue 3 MFMA passes per algorithmic flop")}
    del eng
    # PRIMP's loop (Gauss-Seidel, anchored inertial term), driven as models/PRIMP.py drives it since round 5: one C call per iteration
    # (bmf_primp_iterate), the objective of iteration t read while t + 1 runs, the pair of t snapshotted on the device
    eng = PalmEngine(X, k, L.PALM_PRIMP, beta=1e-4)
    eng.load_factors(U0, V0)
    psched = lambda i: (0.01, 0.02 * 1.02 ** i)   # noqa: E731
    for i in range(warm):
        eng.primp_iterate(i, *psched(i))
    eng.primp_row(warm - 1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.primp_iterate(warm, *psched(warm))
    for i in range(warm, warm + iters):
        eng.keep()
        eng.primp_iterate(i + 1, *psched(i + 1))
        res["p"] = eng.primp_row(i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / (iters + 1)
    out["primp_ipalm"] = {"config": f"PRIMP loop (Gauss-Seidel proximal steps, beta=1e-4), {m}x{n} Boolean, k={k}, int8 x3 operands, objective every iteration",
                          "iterations_per_s": 1.0 / dt, "ms_per_iteration": 1e3 * dt, "objective": float(res["p"]),
                          "roofline": roof("mfma", 4.0 * m * n * k, dt, MFMA_PEAK_TFLOPS["i8"], "TFLOP/s",
                                           "the two bits GEMMs X^T U and X V (2 m n k each); the steps do not emit the digit planes (anchored "
                                           "inertial form): a stand-alone plane builder per factor rides in the iteration")}
    del eng
    for name, link, mode in (("pnlpf", L.LINK_SIGMOID, L.MODE_PENALTY), ("wnmf_kl", L.LINK_KL, L.MODE_WNMF)):
        eng = LinkMUEngine(X, k, link, mode, lamda=10.0)
        eng.load_factors(U0, V0)
        eng.prepare()

        # driven as the model classes drive it: one C call per iteration (bmf_link_iterate), the scalars of iteration t read while t + 1 runs
        eng.iterate(0, 1.0, update=False)
        eng.iterate(1, 1.0)
        eng.row(0, 1.0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(2, 6):
            eng.iterate(i, 1.0)
            eng.row(i - 1, 1.0)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 4
        # per factor update P (2 m n k) and one (KL) or two (sigmoid) contractions with the other factor (2 m n k each); the scalar pass P again
        flops_it = (2 * (6.0 if name == "pnlpf" else 4.0) + 2.0) * m * n * k
        out[name] = {"config": f"{'PNLPF (sigmoid link)' if name == 'pnlpf' else 'WNMF Kullback-Leibler'} update pair + scalar pass, {m}x{n} Boolean, k={k}",
                     "iterations_per_s": 1.0 / dt, "ms_per_iteration": 1e3 * dt,
                     "roofline": roof("mfma", flops_it, dt, MFMA_PEAK_TFLOPS["bf16"], "TFLOP/s",
                                      "P = U V^T and the contractions of the two tile-fused passes + P of the scalar pass, algorithmic "
                                      "(fp32-equivalent) flops; executed on the bf16 MFMA with split operands: 6 products for P, 3 per "
                                      "contraction -- the pipe is 50-56 % busy, the vector unit as much (profiles/r04_pmc_link.md)")}
        del eng
    # rank 128 on the two-block engine (pybmf_amd/wide.py): V and U update + every score of a log row, Python-driven
    from pybmf_amd.wide import WideMUEngine
    kw = 128
    rsw = np.random.RandomState(7)
    avg = np.sqrt(X.sum_local / (float(m) * n) / kw)
    eng = WideMUEngine(X, kw, L.MODE_PENALTY, with_mae=True)
    eng.load_factors(np.abs(avg * rsw.standard_normal((m, kw))) + 1e-6, np.abs(avg * rsw.standard_normal((n, kw))) + 1e-6)
    eng.prepare()

    # driven as the model classes drive it since round 5: iteration t + 1 enqueued before the scalars of t are read (no host round trip)
    eng.iterate(0, 1.0, update=False)
    eng.iterate(1, 1.02)
    eng.row(0, 1.0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n_w = 6
    for i in range(1, 1 + n_w):
        eng.iterate(i + 1, 1.02 ** (i + 1))
        res["w"] = eng.row(i, 1.02 ** i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n_w
    out["penalty_k128"] = {"config": f"BinaryMF-Penalty MU at rank 128 (two 64-column blocks per factor), {m}x{n} Boolean, all scores incl. MAE every iteration",
                           "iterations_per_s": 1.0 / dt, "ms_per_iteration": 1e3 * dt, "error": float(res["w"][0]),
                           "roofline": roof("mfma", 4.0 * m * n * kw, dt, MFMA_PEAK_TFLOPS["i8"], "TFLOP/s",
                                            "the bits GEMMs X V and X^T U at rank 128 (2 m n k each), two 64-column blocks per factor")}
    del eng
    # masked update (W = 'mask' on a negative-sampled csr) at MovieLens-1M shape, k = 16
    rs = np.random.RandomState(0)
    mm, nn, kk = 6040, 3706, 16
    pu, pv = rs.pareto(1.2, mm) + 1, rs.pareto(1.2, nn) + 1
    P = np.outer(pu / pu.sum(), pv / pv.sum())
    ones = rs.rand(mm, nn) < np.minimum(P * 1_000_209, 1.0)
    neg = (rs.rand(mm, nn) < ones.mean()) & ~ones
    r, c = np.nonzero(ones | neg)
    S = SparseObs(r, c, ones[r, c].astype(np.float32), None, (mm, nn))
    eng = MaskedMUEngine(S, kk, L.MODE_PENALTY, bits=BitMatrix(ones.astype(np.uint8), "cuda:0"))
    eng.load_factors(np.abs(rs.standard_normal((mm, kk))) * 0.2, np.abs(rs.standard_normal((nn, kk))) * 0.2)
    eng.prepare()

    # driven as BinaryMFPenalty._fit_masked drives it: one C call per iteration (bmf_masked_iterate), the scalars of iteration t read while
    # t + 1 runs
    warm, iters = 3, 30
    eng.iterate(0, 1.0, update=False)
    for i in range(1, warm + 1):
        eng.iterate(i, 1.02 ** i)
        eng.row(i - 1, 1.0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(warm + 1, warm + 1 + iters):
        eng.iterate(i, 1.02 ** i)
        res["m"] = eng.row(i - 1, 1.0)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    # per observed cell and factor update: the cell record (index 4 B, value 4 B) and one row of the other factor (kp fp32, from L2 after
    # the first touch); per update the factor itself (fp64 master read + written, fp32 shadow written) -- two updates per iteration
    cells_obs, kp_m = len(r), 32
    bytes_it = 2.0 * cells_obs * (8 + 4 * kp_m) + 2.0 * (mm + nn) * kp_m * (8 + 8 + 4) / 2
    out["masked_penalty"] = {"config": f"BinaryMF-Penalty under W='mask', {mm}x{nn}, {len(r)} observed cells, k={kk}, whole-matrix scores every iteration",
                             "iterations_per_s": 1.0 / dt, "ms_per_iteration": 1e3 * dt,
                             "roofline": roof("hbm", bytes_it, dt, HBM_PEAK_BYTES / 1e9, "GB/s",
                                              "cell records + gathered factor rows of the two masked passes + the factor updates; at this "
                                              "size (0.7 M cells, 12 MB) the iteration is launch- and latency-bound, not bandwidth-bound")}
    return out


@contextlib.contextmanager
def stdout_to_stderr():
    """Everything written to file descriptor 1 inside the block -- by Python or by C libraries through stdio -- goes to stderr."""
    libc = C.CDLL(None)
    sys.stdout.flush()
    libc.fflush(None)
    saved = os.dup(1)
    os.dup2(2, 1)
    try:
        yield
    finally:
        sys.stdout.flush()
        libc.fflush(None)    # a pipe is fully buffered: what C code printed must leave its buffer while fd 1 still is stderr
        os.dup2(saved, 1)
        os.close(saved)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--m", type=int, default=100_000)
    ap.add_argument("--n", type=int, default=20_000)
    ap.add_argument("--k", type=int, default=64)
    ap.add_argument("--operands", default="i8x3", choices=sorted(OPND),
                    help="factor operand format of the two bits GEMMs: 3 / 2 planes of int8 digits with exact int32 accumulation "
                         "(24 / 16 significant bits), two column-scaled fp16 addends (22 bits), or 3 / 2 bf16 addends (24 / 16 bits)")
    ap.add_argument("--mae", type=int, default=0, help="1: also run the residual (MAE) pass every step")
    ap.add_argument("--secondary", type=int, default=1, help="0: skip the secondary legs and configurations (profiling runs)")
    ap.add_argument("--cpu-rows", type=int, default=-1, help="0: skip the CPU baseline; -1: largest sample within RAM and time budget")
    ap.add_argument("--traffic", type=int, default=1, help="0: skip the rocprofv3 child runs that measure the GEMM's fabric traffic")
    ap.add_argument("--alt-operands", default="f16x2", choices=["none"] + sorted(OPND), help="also time the loop with this format (N=1 only)")
    ap.add_argument("--preheat", type=int, default=150,
                    help="untimed iterations of the same loop before the run proper: a fresh process reaches its steady rate only after ~60 ms of GPU "
                         "work (power state / clocks; measured: the SAME 20 iterations of the trajectory run 7 %% faster when 80 iterations went before). "
                         "The factors are reloaded afterwards; the first K of these iterations are timed too and reported as `cold_start`. 0: off")
    ap.add_argument("--sustained", type=int, default=5000,
                    help="iterations of the `sustained` leg (outside `value`): the same loop with the stopping rule disabled, long enough (>= 3 s of GPU "
                         "time) for an external sampler to see the GPU busy; 0: off")
    ap.add_argument("--repeat", type=int, default=-1, help="extra timed legs of K steps after the timed region (default: 3 with --secondary 1, else 0)")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--i8-variant", type=int, default=0,
                    help="kernel variant of the whole-factor int8 bits GEMM (bmf_xf_bits_i8_variant; A/B runs: 0 = the default, 4 = the "
                         "anti-phase eight-wave kernel of csrc/xf_bits_i8p.hip)")
    args = ap.parse_args()
    args.panel, args.terms = OPND[args.operands]
    if args.i8_variant:
        from pybmf_amd import _lib as _L
        _L.check(min(0, _L.lib.bmf_xf_bits_i8_variant(args.i8_variant)))

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
    # BMF_FORCE_SHARDED=1 rehearses the multi-GPU code path (RCCL communicator, all-reduces, the C-side sharded loop) with one rank
    sharded = world > 1 or os.environ.get("BMF_FORCE_SHARDED") == "1"
    if sharded:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        # RCCL prints a banner (HIP / ROCm version, host, library path) to STDOUT when its communicator comes up; the contract
        # is ONE JSON line there, so fd 1 points at stderr until the first collective has run
        with stdout_to_stderr():
            if rehearsal:
                dist.init_process_group("gloo", rank=rank, world_size=world)
            else:
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
            t = torch.ones(1, device=device)
            dist.all_reduce(t)
            torch.cuda.synchronize()
            assert float(t.item()) == float(world)

    from pybmf_amd import _lib as L
    from pybmf_amd.engine import BitMatrix, MUEngine, shard_rows
    from pybmf_amd.generators import PlantedBooleanOnDevice

    m, n, k = args.m, args.n, args.k
    K, W = args.steps, args.warmup
    extra_legs = 0 if (args.pmc_child or not args.secondary) else 3
    if args.repeat >= 0 and not args.pmc_child:
        extra_legs = args.repeat
    # planted factors, density 0.067, noise [0.05, 0.01]
    dens = 0.067 if k >= 32 else 0.2
    gen = PlantedBooleanOnDevice(m, n, k, density=(dens, dens), seed=1000, noise=(0.05, 0.01), noise_seed=2000, device=device)
    lo, hi = shard_rows(m, rank, world)
    X = BitMatrix(gen, device, row_lo=lo, row_hi=hi)
    del gen
    reg0, growth, max_reg = 1.0, 1.02, 1e10
    comm_leg = 1 if (sharded and not args.pmc_child) else 0   # one more leg of K steps with the exchange timed by events (outside `value`)
    n_iter_total = W + K * (1 + extra_legs + comm_leg) + 1    # + the oracle-checked extra update
    preheat = 0 if args.pmc_child else max(0, args.preheat)
    if preheat:
        preheat = max(preheat, W + K)
    max_iter = max(n_iter_total, preheat) + 1
    tol = float(os.environ.get("BMF_BENCH_TOL", "0.01"))
    eng = MUEngine(X, k=k, mode=L.MODE_PENALTY, terms=args.terms, with_mae=bool(args.mae), tol=tol, min_diff=0.0,
                   max_iter=max_iter, sharded=sharded, panel=args.panel)
    U0, V0 = host_init(eng.sum_x / (float(m) * n), m, n, k, seed=2024)
    U0d, V0d = torch.from_numpy(U0[lo:hi]).to(device), torch.from_numpy(V0).to(device)   # the initial factors, resident in HBM
    eng.load_factors(U0d, V0d)
    regs, r = [], np.float64(reg0)
    for _ in range(max(n_iter_total, preheat)):
        regs.append(float(r))
        r = min(r * np.float64(growth), np.float64(max_reg))

    def barrier():
        torch.cuda.synchronize()
        if sharded:
            dist.barrier()
            torch.cuda.synchronize()

    def timed_leg(first, count, it0):
        barrier()
        t0 = time.perf_counter()
        eng.run(regs[first:first + count], it0=it0)
        barrier()
        dt = time.perf_counter() - t0
        if sharded:
            t = torch.tensor([dt], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    cold = None
    if preheat:   # untimed for `value`; its own first K steps after W are what a cold process delivers
        eng.prepare(regs[0])
        eng.run(regs[:W], it0=1)
        cold = timed_leg(W, K, 1 + W)
        eng.run(regs[W + K:preheat], it0=1 + W + K)
        barrier()
        eng.load_factors(U0d, V0d)         # fresh start, device-side (no PCIe wait in which the clocks fall back): log, stop flag and counters are reset too
    eng.prepare(regs[0])
    eng.run(regs[:W], it0=1)
    barrier()
    L.check(L.lib.bmf_timer_enable(2 * K + 8))
    L.check(L.lib.bmf_timer_stride(3))   # sample 1 launch in 3 (alternates between X V and X^T U): event pairs cost stream time
    dt = timed_leg(W, K, 1 + W)          # THE timed region: exactly K steps
    n_launch, gemm_ms = C.c_int(0), C.c_double(0.0)
    L.check(L.lib.bmf_timer_read(C.byref(n_launch), C.byref(gemm_ms)))
    L.check(L.lib.bmf_timer_disable())
    if args.pmc_child:
        return
    repeat = [timed_leg(W + K * (1 + j), K, 1 + W + K * (1 + j)) for j in range(extra_legs)]
    done = W + K * (1 + extra_legs)
    comm = None
    if comm_leg:   # the exchange, timed by events inside the C loop (three events per step: kept out of `value`)
        eng.comm_timing(True)
        dt_c = timed_leg(done, K, 1 + done)
        comm = eng.comm_timing(False) or {}
        comm["ms_per_step_in_this_leg"] = 1e3 * dt_c / K
        done += K

    local_floor = None
    if comm_leg:   # every rank: its own shard through the unsharded loop (no collectives inside), max over ranks
        eng_l = MUEngine(X, k=k, mode=L.MODE_PENALTY, terms=args.terms, with_mae=bool(args.mae), tol=tol, min_diff=0.0, max_iter=W + K + 2,
                         sharded=False, panel=args.panel)
        eng_l.load_factors(U0[lo:hi], V0)
        eng_l.prepare(regs[0])
        eng_l.run(regs[:W], it0=1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng_l.run(regs[W:W + K], it0=1 + W)
        torch.cuda.synchronize()
        tl = torch.tensor([(time.perf_counter() - t0) * 1e3 / K], dtype=torch.float64, device=device)
        dist.all_reduce(tl, op=dist.ReduceOp.MAX)
        local_floor = float(tl.item())
        del eng_l

    log, stop = eng.read_log()
    if os.environ.get("BMF_NO_CHECK") != "1":  # (timing-only kernel experiments produce wrong numbers on purpose)
        assert log.shape[0] == 1 + done and stop == 0, (log.shape, stop)
        assert np.isfinite(log[:, :6]).all()
    last = log[W + K]    # the log row at the end of the timed region

    # checks: (a) the logged trace-form rec_error against a direct residual pass on the GPU and NumPy fp64 on rank 0's first rows;
    # (b) against the ORACLE: one more update from the final state, exact on a row / column sample
    chk = {}
    sums = torch.zeros(4, dtype=torch.float64, device=device)
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    L.check(L.lib.bmf_residual_sums(L.ptr(X.bits), X.m_pad, X.ldx, X.m, X.n, L.ptr(eng.U), L.ptr(eng.V), eng.kp, L.ptr(sums), None, stream))
    if sharded:
        dist.all_reduce(sums)
    direct = 0.5 * float(sums[1].item())
    chk["rec_error_trace_vs_direct_rel"] = abs(direct - log[-1, L.LOG_REC]) / direct
    if rank == 0:
        rs = min(1024, X.m)
        Uh, Vh = eng.factors()
        Xs = X.rows_dense_u8(0, rs).astype(np.float64)
        with few_blas_threads():
            host = float(((Xs - Uh[:rs] @ Vh.T) ** 2).sum())
        sums.zero_()
        L.check(L.lib.bmf_residual_sums(L.ptr(X.bits), X.m_pad, X.ldx, rs, X.n, L.ptr(eng.U), L.ptr(eng.V), eng.kp, L.ptr(sums), None, stream))
        chk["residual_gpu_vs_numpy_fp64_rel"] = abs(float(sums[1].item()) - host) / host
    if not sharded:
        rv, ru = oracle_step_check(X, eng, regs[done], 1 + done)
        chk["oracle_step_rel_V"], chk["oracle_step_rel_U"] = rv, ru
        chk["oracle_step"] = (f"update {1 + done} from the GPU's own state re-computed by the fp64 oracle on 256 columns (V) and 1024 rows (U) "
                              "of the full problem; gate 1e-4 (PyBMF/models/BinaryMFPenalty.py:136-163)")
        if os.environ.get("BMF_NO_CHECK") != "1":
            assert rv <= 1e-4 and ru <= 1e-4, (rv, ru)

    # sustained leg (all ranks): the same loop, stopping rule disabled, long enough for an external sampler to see the GPU busy
    sus = None
    if args.sustained > 0:
        n_s = int(args.sustained)
        regs_s, r = [], np.float64(reg0)
        for _ in range(n_s + 1):
            regs_s.append(float(r))
            r = min(r * np.float64(growth), np.float64(max_reg))
        eng_s = MUEngine(X, k=k, mode=L.MODE_PENALTY, terms=args.terms, with_mae=bool(args.mae), tol=-1.0, min_diff=0.0,
                         max_iter=n_s + 2, sharded=sharded, panel=args.panel)
        eng_s.load_factors(U0[lo:hi], V0)
        eng_s.prepare(regs_s[0])
        barrier()
        t0 = time.perf_counter()
        eng_s.run(regs_s[:n_s], it0=1)
        barrier()
        dt_s = time.perf_counter() - t0
        if sharded:
            t = torch.tensor([dt_s], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt_s = float(t.item())
        log_s, stop_s = eng_s.read_log()
        assert log_s.shape[0] == 1 + n_s and stop_s == 0 and np.isfinite(log_s[:, :6]).all(), (log_s.shape, stop_s)
        sus = {"iterations": n_s, "seconds": dt_s, "value": n_s / dt_s, "ms_per_step": 1e3 * dt_s / n_s,
               "final": {"error": float(log_s[-1, L.LOG_ERROR]), "rec_error": float(log_s[-1, L.LOG_REC]), "reg": regs_s[n_s - 1],
                         "TP": int(log_s[-1, L.LOG_TP]), "FP": int(log_s[-1, L.LOG_FP])},
               "note": f"{n_s} consecutive iterations from a fresh start in one run, tol disabled, reg growing 1.02x per step to its cap 1e10 "
                       "(outside `value`); the factors turn Boolean on the way, which makes the data-dependent cover count cheaper"}
        if not sharded:
            rv, ru = oracle_step_check(X, eng_s, regs_s[n_s], 1 + n_s)
            sus["oracle_step_rel_V"], sus["oracle_step_rel_U"] = rv, ru
            if os.environ.get("BMF_NO_CHECK") != "1":
                assert rv <= 1e-4 and ru <= 1e-4, (rv, ru)
        if sharded:
            eng_s.close()
        del eng_s

    if rank != 0:
        if sharded:
            eng.close()
            from pybmf_amd.engine import shutdown_comms
            shutdown_comms()   # (the cached RCCL communicator, while the process group is alive)
            dist.destroy_process_group()
        return

    its = K / dt
    launches = max(n_launch.value, 1)
    avg_ms = gemm_ms.value / launches
    # algorithmic flops of one bits-GEMM launch on this rank: 2 * m_local * n * k (X V and X^T U are the same count)
    flops_launch = 2.0 * X.m * n * k
    achieved = flops_launch / (avg_ms * 1e-3) / 1e12
    limb_bytes = 1.0 if args.panel == "i8" else 2.0
    bytes_launch = X.m * n / 8.0 + 0.5 * (X.m + n) * k * (limb_bytes * args.terms + 4.0)
    peak = MFMA_PEAK_TFLOPS[args.panel]
    traffic, traffic_note = (None, "skipped")
    profiled = "rocprof" in os.environ.get("LD_PRELOAD", "").lower() or any(k.startswith(("ROCPROF", "ROCPROFILER")) for k in os.environ)
    if args.traffic and world == 1 and not sharded and profiled:
        traffic_note = "skipped: this run is itself under a profiler (no profiler children inside a profiled process)"
    elif args.traffic and world == 1 and not sharded:
        traffic, traffic_note = measure_traffic(args)
    out = {
        "metric": "MU iterations/sec (BinaryMF-Penalty, 100k x 20k Boolean, k=64)" if (m, n, k) == (100_000, 20_000, 64)
                  else f"MU iterations/sec (BinaryMF-Penalty, {m}x{n} Boolean, k={k})",
        "value": its, "unit": "iterations/s", "n_gpus": world, "steps": K, "warmup": W,
        "ms_per_step": 1e3 * dt / K, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": ("i8 (bits x int8 digit planes on the integer MFMA, exact int32 accumulation); fp64 master factors and scalars"
                  if args.panel == "i8" else
                  f"{args.panel} (bits x split-{args.panel} MFMA, fp32 accumulate); fp64 master factors and scalars"),
        "data": "synthetic (planted Boolean factors + flip noise, generated on device; SURVEY 8d)",
        "config": {"workload": f"BinaryMF-Penalty MU, {m}x{n} dense Boolean X, k={k}, reg=1 growth=1.02, init normal+balance seed 2024",
                   "mae_pass": bool(args.mae), "operands": args.operands, "row_sharding": f"{world} x {X.m} rows",
                   "splits_xv": eng.splits_xv, "splits_xtu": eng.splits_xtu,
                   "k_limit": "the tuned path takes k <= 64 (one 64-bit word of factor bits per row); 64 < k <= 128 runs BinaryMFPenalty / WNMF on a two-block engine (pybmf_amd/wide.py, Python-driven), larger k raises NotImplementedError"},
        "roofline": {"kernel": "xf_bits_i8_kernel (X V and X^T U)" if args.panel == "i8" else "xf_bits_kernel (X V and X^T U)",
                     "bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                     "peak_note": f"dense {args.panel} MFMA peak of MI355X_MICROARCH.md (i8 = 2 x the bf16 rate per clock); the kernel issues "
                                  f"{args.terms} MFMA passes per algorithmic flop (hw_flops_factor)",
                     "traffic": traffic, "traffic_source": traffic_note, "launches_timed": launches, "avg_launch_ms": avg_ms,
                     "algorithmic_flops_per_launch": flops_launch, "hw_flops_factor": args.terms,
                     "frac_of_f16_mfma_peak": achieved / MFMA_PEAK_TFLOPS["f16"],
                     "mfma_pipe_busy_estimate": args.terms * achieved / peak,
                     "frac_of_fp32_mfma_peak": achieved / FP32_MFMA_PEAK_TFLOPS,
                     # the other roofline of SURVEY 8d: algorithmic bytes of one launch (X as bits once, the factor panel, the
                     # fp32 result; mean of the X V and X^T U launches) against HBM
                     "algorithmic_bytes_per_launch": bytes_launch, "frac_of_hbm_peak": bytes_launch / (avg_ms * 1e-3) / HBM_PEAK_BYTES,
                     "traffic_over_algorithmic": (traffic / bytes_launch) if traffic else None,
                     "gemm_share_of_step": 2.0 * avg_ms * 1e-3 * K / dt},
        "iteration_vs_fp32_mfma_roofline": its / (FP32_MFMA_PEAK_TFLOPS * 1e12 * world / (4.0 * m * n * k + 4.0 * (m + n) * k * k)),
        "final": {"iter": int(last[L.LOG_ITER]), "error": last[L.LOG_ERROR], "rec_error": last[L.LOG_REC],
                  "reg_error": last[L.LOG_REGERR], "TP": int(last[L.LOG_TP]), "FP": int(last[L.LOG_FP])},
        "checks": chk,
    }
    if cold is not None:
        out["cold_start"] = {"value": K / cold, "ms_per_step": 1e3 * cold / K,
                             "note": f"the same W + K steps in the cold process, before the {preheat} untimed pre-heating iterations after which the factors were "
                                     "reloaded and the run proper started (`value`); identical data and trajectory: the difference is the GPU's power / clock state"}
        out["config"]["preheat_iterations"] = preheat
    if repeat:
        rates = sorted([its] + [K / t for t in repeat])
        out["repeat"] = {"legs_of_K_steps": [K / t for t in repeat], "median_incl_value": rates[len(rates) // 2],
                         "note": "further timed legs of K steps each, continuing the same run (outside `value`)"}
        # `value` is ONE window of K steps (12 ms at the default K): the noisiest number of the line.  The median over it and the repeat
        # legs (lower middle for an even count) stands beside it; `value` itself stays what the contract says.
        out["value_median_of_legs"] = rates[(len(rates) - 1) // 2]
    if sharded:
        d = {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "rank0_device": f"cuda:{local}",
             "device": torch.cuda.get_device_name(device), "exchange": eng.exchange_description(), "plan": eng.exchange_plan, **(comm or {})}
        # the diagnosis of the N > 1 step in one line: what a rank computes per step, what of the exchange is NOT hidden under it, and
        # the same shard through the unsharded loop (no exchange, the log row fused into the slab sum) as the floor
        if comm and "exposed_comm_ms_per_step" in comm:
            d["per_rank_compute_ms"] = comm["ms_per_step_in_this_leg"] - comm["exposed_comm_ms_per_step"]
        if local_floor is not None:
            d["unsharded_same_shard_ms_per_step"] = local_floor
            d["ideal_vs_achieved"] = {"ideal_ms_per_step": local_floor, "achieved_ms_per_step": 1e3 * dt / K, "ratio": local_floor / (1e3 * dt / K),
                                      "note": "ideal = this rank's rows through the single-GPU loop, no exchange (max over ranks); achieved = `ms_per_step`"}
        out["distributed"] = d
    if world == 1 and not sharded and args.secondary:
        sec = {}
        # another operand format, same data and schedule; the difference of the final factors between the two runs is reported
        if args.alt_operands not in ("none", args.operands):
            panel2, terms2 = OPND[args.alt_operands]
            eng2 = MUEngine(X, k=k, mode=L.MODE_PENALTY, terms=terms2, with_mae=bool(args.mae), tol=tol, min_diff=0.0, max_iter=max_iter, panel=panel2)
            eng2.load_factors(U0[lo:hi], V0)
            eng2.prepare(regs[0])
            eng2.run(regs[:W], it0=1)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            eng2.run(regs[W:W + K], it0=1 + W)
            torch.cuda.synchronize()
            dt2 = time.perf_counter() - t1
            log2, _ = eng2.read_log()
            U2, V2 = eng2.factors()
            # the main engine has moved on: compare at iteration W + K through a fresh run of the main format
            eng1 = MUEngine(X, k=k, mode=L.MODE_PENALTY, terms=args.terms, with_mae=bool(args.mae), tol=tol, min_diff=0.0, max_iter=max_iter, panel=args.panel)
            eng1.load_factors(U0[lo:hi], V0)
            eng1.prepare(regs[0])
            eng1.run(regs[:W + K], it0=1)
            U1, V1 = eng1.factors()
            with few_blas_threads():
                du, dv = float(np.linalg.norm(U2 - U1) / np.linalg.norm(U1)), float(np.linalg.norm(V2 - V1) / np.linalg.norm(V1))
            out["alt"] = {"operands": args.alt_operands, "value": K / dt2, "ms_per_step": 1e3 * dt2 / K,
                          "rel_diff_U_vs_main": du, "rel_diff_V_vs_main": dv,
                          "rel_diff_rec_error_vs_main": float(abs(log2[-1, L.LOG_REC] / last[L.LOG_REC] - 1.0))}
            del eng2, eng1
        if not args.mae:
            # the reference's loop also logs MAE every iteration (BinaryMFPenalty.py:71,97) -- the DEFAULT of the model classes
            for name, kw in (("with_mae", dict(with_mae=True)), ("updates_only", dict(with_mae=False))):
                e3 = MUEngine(X, k=k, mode=L.MODE_PENALTY, terms=args.terms, tol=tol, min_diff=0.0, max_iter=max_iter, panel=args.panel, **kw)
                if name == "updates_only":   # V- and U-update with their error terms, no Boolean cover count, no MAE
                    e3.st.updates_only = 1
                e3.load_factors(U0[lo:hi], V0)
                e3.prepare(regs[0])
                e3.run(regs[:W], it0=1)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                e3.run(regs[W:W + K], it0=1 + W)
                torch.cuda.synchronize()
                dt3 = time.perf_counter() - t1
                log3, _ = e3.read_log()
                out[name] = {"value": K / dt3, "ms_per_step": 1e3 * dt3 / K}
                if name == "with_mae":
                    out[name].update(MAE=float(log3[-1, L.LOG_MAE]), RMSE=float(log3[-1, L.LOG_RMSE]))
                else:
                    out[name]["rel_diff_error_vs_main"] = abs(float(log3[-1, L.LOG_ERROR]) - last[L.LOG_ERROR]) / last[L.LOG_ERROR]
                del e3
        for name, fn in (("c1_penalty_fit", secondary_c1), ("c2_wnmf_real", secondary_c2), ("c5_threshold_line_search", secondary_c5)):
            try:
                sec[name] = fn()
            except Exception as e:  # noqa: BLE001  (a secondary number must not take the headline line down)
                sec[name] = {"error": f"{type(e).__name__}: {e}"}
        try:
            sec["widened_engines"] = secondary_widened(X, U0, V0)
        except Exception as e:  # noqa: BLE001
            sec["widened_engines"] = {"error": f"{type(e).__name__}: {e}"}
        out["secondary"] = sec
    if world == 1 and args.cpu_rows != 0:
        out["cpu_baseline"] = cpu_baseline(X, U0, V0, reg0, m)
    if sus is not None:
        out["sustained"] = sus
    print(json.dumps(out))
    if sharded:
        eng.close()
        from pybmf_amd.engine import shutdown_comms
        shutdown_comms()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
