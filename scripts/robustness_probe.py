"""How far does the fp32 GPU trajectory stay from the fp64 oracle under the reference's DEFAULT hyper-parameters
(reg=2, reg_growth=3: lambda reaches 1e10 after ~21 iterations), and with 2 vs 3 bf16 addends?"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle as orc
from pybmf_amd import _lib as L
from pybmf_amd.engine import BitMatrix, MUEngine

z = np.load("tests/golden/g1_penalty_c1.npz")
X = np.unpackbits(z["X_bits"], axis=1, bitorder="little")[:, : z["shape"][1]]
def rel(a, b): return np.linalg.norm(a - b) / np.linalg.norm(b)
for (reg, growth, iters) in [(1.0, 1.02, 100), (2.0, 3.0, 40), (1.0, 1.3, 60)]:
    ref = orc.penalty_fit(X, k=8, U=z["U0"], V=z["V0"], reg=reg, reg_growth=growth, init_method="custom", normalize_method=None,
                          max_iter=iters - 1, tol=-1.0, literal=False)
    for terms in (3, 2):
        eng = MUEngine(BitMatrix(X, "cuda:0"), k=8, mode=L.MODE_PENALTY, terms=terms, with_mae=False, tol=-1.0, max_iter=iters + 1)
        eng.load_factors(z["U0"], z["V0"])
        regs, r = [], np.float64(reg)
        for _ in range(iters):
            regs.append(float(r)); r = min(r * growth, 1e10)
        eng.prepare(regs[0]); eng.run(regs, it0=1)
        log, stop = eng.read_log(); U, V = eng.factors()
        want = np.array(ref["updates"])
        flipsU = int(((U > 0.5) != (ref["U"] > 0.5)).sum()); flipsV = int(((V > 0.5) != (ref["V"] > 0.5)).sum())
        cnt = tuple(int(log[-1, c]) for c in (L.LOG_TP, L.LOG_FP))
        print(f"reg={reg} growth={growth} iters={iters} terms={terms}: relU={rel(U, ref['U']):.2e} relV={rel(V, ref['V']):.2e} "
              f"max rel err(error col)={np.abs(log[:,1]/want[:,1]-1).max():.2e} flips U/V={flipsU}/{flipsV} "
              f"counts gpu={cnt} ref={ref['counts'][-1][:2]} margin={min(np.abs(ref['U']-0.5).min(), np.abs(ref['V']-0.5).min()):.1e}")

# default schedule WITH the tol-based stop: does the GPU path stop at the same iteration as the fp64 reference?
ref = orc.penalty_fit(X, k=8, U=z["U0"], V=z["V0"], reg=2.0, reg_growth=3.0, init_method="custom", normalize_method=None,
                      max_iter=100, tol=0.01, literal=False)
for terms in (3, 2):
    eng = MUEngine(BitMatrix(X, "cuda:0"), k=8, mode=L.MODE_PENALTY, terms=terms, with_mae=False, tol=0.01, max_iter=100)
    eng.load_factors(z["U0"], z["V0"])
    regs, r = [], np.float64(2.0)
    for _ in range(101):
        regs.append(float(r)); r = min(r * 3.0, 1e10)
    eng.prepare(regs[0]); eng.run(regs, it0=1)
    log, stop = eng.read_log(); U, V = eng.factors()
    want = np.array(ref["updates"])
    n = min(len(log), len(want))
    print(f"default schedule, tol=0.01, terms={terms}: stop gpu={stop} ref={ref['n_iter']} rows {len(log)}/{len(want)} relU={rel(U, ref['U']):.2e} "
          f"relV={rel(V, ref['V']):.2e} max rel err error={np.abs(log[:n,1]/want[:n,1]-1).max():.2e} reg_err={np.abs(log[1:n,4]/want[1:n,4]-1).max():.2e}")
