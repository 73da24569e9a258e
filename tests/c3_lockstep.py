"""GPU trajectory vs the fp64 oracle at BASELINE.json configs[2] (100 000 x 20 000 Boolean, k = 64), iteration by iteration.

Shared by ``tests/test_c3_parity_gpu.py`` (the gate) and ``scripts/parity_trace.py`` (the committed per-iteration trace under
``profiles/``).  Test infrastructure: imports ``oracle``; nothing under ``pybmf_amd/`` imports this.

Reference loop being matched: PyBMF/models/BinaryMFPenalty.py:81-115 (update_V, update_U :136-163; the oracle's re-associated
form agrees with the literal one to 1e-15, tests/test_oracle_golden.py::test_c1_single_step).
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import oracle as orc  # noqa: E402

OPERANDS = {"f16x2": ("f16", 2), "bf16x3": ("bf16", 3), "bf16x2": ("bf16", 2), "i8x3": ("i8", 3), "i8x2": ("i8", 2)}


def rel(a, b):
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def bench_problem(m=100_000, n=20_000, k=64, device="cuda:0", n_iter=100):
    """X (bits on the device), the initial factors and the regulariser schedule of bench.py."""
    from bench import host_init
    from pybmf_amd.engine import BitMatrix
    from pybmf_amd.generators import PlantedBooleanOnDevice
    dens = 0.067 if k >= 32 else 0.2
    gen = PlantedBooleanOnDevice(m, n, k, density=(dens, dens), seed=1000, noise=(0.05, 0.01), noise_seed=2000, device=device)
    X = BitMatrix(gen, device)
    del gen
    U0, V0 = host_init(X.sum_local / (float(m) * n), m, n, k, seed=2024)
    regs, r = [], np.float64(1.0)
    for _ in range(n_iter):
        regs.append(float(r))
        r = min(r * np.float64(1.02), np.float64(1e10))
    return X, U0, V0, regs


def host_ram_available() -> float:
    try:
        import psutil
        return float(psutil.virtual_memory().available)
    except Exception:
        return 0.0


def boolean_counts_host(X, U, V, u=0.5, v=0.5, chunk=8192):
    """(TP, FP) of min(1, (U > u)(V > v)^T) against the bits of X, in row chunks (utils/common.py:110-151, metrics.py:56-68)."""
    Vb = (V > v).astype(np.float32)
    tp = fp = 0
    for a in range(0, X.m, chunk):
        b = min(a + chunk, X.m)
        cover = ((U[a:b] > u).astype(np.float32) @ Vb.T) > 0
        xs = X.rows_dense_u8(a, b).astype(bool)
        tp += int(np.count_nonzero(cover & xs))
        fp += int(np.count_nonzero(cover & ~xs))
    return tp, fp


def lockstep(X, U0, V0, regs, n_iter, operands=("f16x2",), scalars_every=10, out=None, with_mae=False):
    """Run one engine per operand format and the oracle side by side from the same initial factors.

    Yields ``(iteration, {operands: (rel_U, rel_V)}, extras)``; ``extras`` carries the scalar / count comparisons on the
    iterations where they are made.  Needs the fp64 X on the host (8 B per cell)."""
    import torch
    from pybmf_amd import _lib as L
    from pybmf_amd.engine import MUEngine
    k = U0.shape[1]
    engines = {}
    for name in operands:
        panel, terms = OPERANDS[name]
        eng = MUEngine(X, k=k, mode=L.MODE_PENALTY, terms=terms, with_mae=with_mae, tol=-1.0, min_diff=0.0, max_iter=n_iter + 1, panel=panel)
        eng.load_factors(U0, V0)
        eng.prepare(regs[0])
        engines[name] = eng
    Xh = X.rows_dense_u8(0, X.m).astype(np.float64)
    sum_x = float(X.sum_local)
    U, V = U0.copy(), V0.copy()
    t0 = time.time()
    for it in range(1, n_iter + 1):
        reg = regs[it - 1]
        for eng in engines.values():
            eng.run([reg], it0=it)
        V = orc.penalty_update_V_reassoc(Xh, U, V, reg)
        U = orc.penalty_update_U_reassoc(Xh, U, V, reg)
        res, extras = {}, {}
        facs = {}
        for name, eng in engines.items():
            Ug, Vg = eng.factors()
            facs[name] = (Ug, Vg)
            res[name] = (rel(Ug, U), rel(Vg, V))
        if scalars_every and (it % scalars_every == 0 or it == n_iter or it == 1):
            # oracle scalars in the trace form (fp64): rec = 0.5 (sum X - 2 <X V, U> + <U^T U, V^T V>), BinaryMFPenalty.py:166-186
            rec = 0.5 * (sum_x - 2.0 * float(((Xh @ V) * U).sum()) + float(((U.T @ U) * (V.T @ V)).sum()))
            reg_err = reg * (orc.reg_term(U) + orc.reg_term(V))
            for name, eng in engines.items():
                log, _ = eng.read_log()
                row = log[log[:, L.LOG_ITER] == it][0]
                extras[name] = {"rec_rel": abs(row[L.LOG_REC] / rec - 1.0), "reg_err_rel": abs(row[L.LOG_REGERR] / reg_err - 1.0),
                                "error_rel": abs(row[L.LOG_ERROR] / (rec + reg_err) - 1.0)}
                if with_mae and (it == n_iter or it == 1):
                    # MAE of the ORACLE's factors (utils/metrics.py:156-160), in row chunks
                    if "mae" not in extras:
                        tot = 0.0
                        for a in range(0, X.m, 8192):
                            tot += float(np.abs(Xh[a:a + 8192] - U[a:a + 8192] @ V.T).sum())
                        extras["mae"] = tot / (float(X.m) * X.n)
                    extras[name]["mae_rel"] = abs(row[L.LOG_MAE] / extras["mae"] - 1.0)
                    extras[name]["rmse_rel"] = abs(row[L.LOG_RMSE] / np.sqrt(2.0 * rec / (float(X.m) * X.n)) - 1.0)
                if it == n_iter or it == 1:
                    Ug, Vg = facs[name]
                    tp, fp = boolean_counts_host(X, Ug, Vg)
                    extras[name]["counts_gpu"] = (int(row[L.LOG_TP]), int(row[L.LOG_FP]))
                    extras[name]["counts_host"] = (tp, fp)
                    # how close the nearest factor entry sits to the 0.5 threshold ("with the margin logged")
                    extras[name]["min_margin"] = float(min(np.abs(Ug - 0.5).min(), np.abs(Vg - 0.5).min()))
        if out is not None:
            line = f"{it:4d}  " + "  ".join(f"{nm} U {res[nm][0]:.3e} V {res[nm][1]:.3e}" for nm in res) + f"   ({time.time() - t0:.0f} s)"
            for nm, e in extras.items():
                if not isinstance(e, dict):
                    continue
                line += f"\n        {nm}: " + ", ".join(f"{kk}={vv:.2e}" if isinstance(vv, float) else f"{kk}={vv}" for kk, vv in e.items())
            print(line, file=out, flush=True)
        yield it, res, extras
    del engines
    torch.cuda.synchronize()


def sampled_step_check(X, eng, reg, it, n_rows=2048, n_cols=512, seed=7):
    """One update from the engine's CURRENT state, checked exactly on a sample: the V update is independent per column of X and the
    U update per row, so rows of the new factors can be recomputed in fp64 from the state before the step
    (V_new[J] from X[:, J], U_old, V_old[J]; U_new[I] from X[I, :], U_old[I], the engine's own V_new).
    Returns (rel_V_on_sample, rel_U_on_sample).  Advances the engine by one iteration."""
    rs = np.random.RandomState(seed)
    I = np.sort(rs.choice(X.m, size=min(n_rows, X.m), replace=False))
    J = np.sort(rs.choice(X.n, size=min(n_cols, X.n), replace=False))
    U_old, V_old = eng.factors()
    eng.run([reg], it0=it)
    U_new, V_new = eng.factors()
    XJ = unpack_cols(X, J)                     # m x |J|
    XI = unpack_rows(X, I)                     # |I| x n
    Vs = orc.penalty_update_V_reassoc(XJ, U_old, V_old[J], reg)
    Us = orc.penalty_update_U_reassoc(XI, U_old[I], V_new, reg)
    return rel(V_new[J], Vs), rel(U_new[I], Us)


def unpack_rows(X, I):
    import torch
    b = X.bits[torch.from_numpy(np.asarray(I, dtype=np.int64)).to(X.device)].cpu().numpy().view(np.uint8)
    return np.unpackbits(b, axis=1, bitorder="little")[:, : X.n].astype(np.float64)


def unpack_cols(X, J):
    import torch
    b = X.bits_t[torch.from_numpy(np.asarray(J, dtype=np.int64)).to(X.device)].cpu().numpy().view(np.uint8)
    return np.ascontiguousarray(np.unpackbits(b, axis=1, bitorder="little")[:, : X.m].T.astype(np.float64))
