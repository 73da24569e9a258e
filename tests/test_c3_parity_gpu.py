"""Oracle parity at the HEADLINE configuration (BASELINE.json configs[2]: 100 000 x 20 000 Boolean, k = 64, the generator,
initialisation and regulariser schedule of bench.py), default operands, EVERY iteration through the drift peak.

Gates (north star / SURVEY 8d): ||dU||_F/||U||_F and ||dV||_F/||V||_F <= 1e-4 against the fp64 oracle trajectory at every
iteration; error / rec_error / reg_error <= 1e-4 relative where compared; TP, FP bit-exact for the GPU's own factors.
Reference loop: PyBMF/models/BinaryMFPenalty.py:81-115, updates :136-163.

Which oracle: the RE-ASSOCIATED form of the updates (oracle.penalty_update_*_reassoc: (U V^T)^T U = V (U^T U); the literal form
would build the 16-GB m x n product twice per iteration).  The two forms are tied to each other in fp64 at config #1, where the
literal form is affordable (tests/test_oracle_golden.py::test_c1_single_step: 1e-15), and the literal form is what the reference's
golden vectors pin; at this size the comparison is therefore with the restatement, not with a run of the reference itself.
All N_ITER iterations are compared, none sampled: the worst iteration moves between boxes (31 .. 42 so far).

The oracle needs the fp64 X on the host (16 GB).  If the box cannot hold it the test falls back to exact single-step checks
on a row / column sample from the GPU's own state (the updates are independent per row / per column) and says so.
C3_PARITY_ITERS overrides the number of iterations (default 64: the drift peaks between iterations 35 and 59, profiles/)."""
import os
import sys

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from c3_lockstep import bench_problem, host_ram_available, lockstep, sampled_step_check  # noqa: E402

GATE = 1e-4
N_ITER = int(os.environ.get("C3_PARITY_ITERS", "64"))


@pytest.fixture(scope="module", autouse=True)
def need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")


def test_c3_trajectory_matches_oracle_every_iteration(capsys):
    m, n, k = 100_000, 20_000, 64
    X, U0, V0, regs = bench_problem(m, n, k, n_iter=N_ITER)
    need = 8.0 * m * n + 6e9
    if host_ram_available() < need:
        # sampled fallback: one-step parity from the GPU's own state at every iteration
        from pybmf_amd import _lib as L
        from pybmf_amd.engine import MUEngine
        eng = MUEngine(X, k=k, mode=L.MODE_PENALTY, with_mae=False, tol=-1.0, max_iter=N_ITER + 1, panel="i8", terms=3)
        eng.load_factors(U0, V0)
        eng.prepare(regs[0])
        worst = 0.0
        for it in range(1, N_ITER + 1):
            rv, ru = sampled_step_check(X, eng, regs[it - 1], it, seed=it)
            worst = max(worst, rv, ru)
            assert rv <= GATE and ru <= GATE, (it, rv, ru)
        with capsys.disabled():
            print(f"\n[c3 parity] host RAM < {need / 1e9:.0f} GB: sampled single-step fallback, worst {worst:.2e} over {N_ITER} iterations",
                  file=sys.stderr)
        return
    worst = {"U": (0.0, 0), "V": (0.0, 0)}
    worst_f16 = 0.0
    # (the fp16 x 2 operands -- round 1's format, kept for A/B behind panel='f16' -- ride along in the same oracle loop and are held to
    # the gate for the first F16_ITERS iterations: a full-size guard for the format the default is compared with)
    F16_ITERS = 16
    for it, res, extras in lockstep(X, U0, V0, regs, N_ITER, operands=("i8x3", "f16x2"), scalars_every=16, with_mae=True):
        ru, rv = res["i8x3"]
        assert ru <= GATE and rv <= GATE, f"iteration {it}: rel U {ru:.3e}, rel V {rv:.3e} (gate {GATE})"
        if it <= F16_ITERS:
            fu, fv = res["f16x2"]
            assert fu <= GATE and fv <= GATE, f"iteration {it}, fp16 x 2 operands: rel U {fu:.3e}, rel V {fv:.3e} (gate {GATE})"
            worst_f16 = max(worst_f16, fu, fv)
            ef = extras.get("f16x2")
            if ef:
                assert ef["rec_rel"] <= GATE and ef["reg_err_rel"] <= GATE and ef["error_rel"] <= GATE, (it, ef)
        if ru > worst["U"][0]:
            worst["U"] = (ru, it)
        if rv > worst["V"][0]:
            worst["V"] = (rv, it)
        e = extras.get("i8x3")
        if e:
            assert e["rec_rel"] <= GATE and e["reg_err_rel"] <= GATE and e["error_rel"] <= GATE, (it, e)
            if "mae_rel" in e:   # the MAE / RMSE columns of the reference's log (BinaryMFPenalty.py:97, metrics.py:138-160)
                assert e["mae_rel"] <= GATE and e["rmse_rel"] <= GATE, (it, e)
            if "counts_gpu" in e:
                assert e["counts_gpu"] == e["counts_host"], (it, e)
    with capsys.disabled():
        print(f"\n[c3 parity] {N_ITER} iterations vs the fp64 oracle: worst rel U {worst['U'][0]:.2e} (iteration {worst['U'][1]}), "
              f"rel V {worst['V'][0]:.2e} (iteration {worst['V'][1]}); fp16 x 2 operands over the first {F16_ITERS}: worst {worst_f16:.2e}; gate {GATE}",
              file=sys.stderr)


def test_c3_sampled_step_check_agrees_with_itself():
    """The sampled single-step checker (the fallback above, and bench.py's `checks.oracle_step_rel_*`) on a mid-size problem where the
    full oracle step is cheap: both must see the same, small, error."""
    import oracle as orc
    from pybmf_amd import _lib as L
    from pybmf_amd.engine import MUEngine
    X, U0, V0, regs = bench_problem(6000, 3000, 64, n_iter=4)
    eng = MUEngine(X, k=64, mode=L.MODE_PENALTY, with_mae=False, tol=-1.0, max_iter=8, panel="i8", terms=3)
    eng.load_factors(U0, V0)
    eng.prepare(regs[0])
    eng.run(regs[:2], it0=1)
    U_old, V_old = eng.factors()
    rv, ru = sampled_step_check(X, eng, regs[2], 3, n_rows=500, n_cols=300)
    Xh = X.rows_dense_u8(0, X.m).astype(np.float64)
    V_new = orc.penalty_update_V_reassoc(Xh, U_old, V_old, regs[2])
    U_new = orc.penalty_update_U_reassoc(Xh, U_old, V_new, regs[2])
    Ug, Vg = eng.factors()
    full_v = np.linalg.norm(Vg - V_new) / np.linalg.norm(V_new)
    full_u = np.linalg.norm(Ug - U_new) / np.linalg.norm(U_new)
    assert rv < 5e-6 and ru < 5e-6 and full_v < 5e-6 and full_u < 5e-6, (rv, ru, full_v, full_u)
    assert 0.2 < (rv + 1e-12) / (full_v + 1e-12) < 5.0
