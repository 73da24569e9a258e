"""CPU emulation: how far is sum|X - UV^T| from the exact value when both factors are rounded to OCP fp8 e4m3 (symmetric per-column
power-of-two scaling) / to one int8 digit / to fp16?  Factors from the oracle trajectory of the bench's model at a reduced size."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import oracle as orc

def e4m3(a):
    a = np.asarray(a, np.float64); s = np.sign(a); x = np.abs(a)
    x = np.minimum(x, 448.0)
    e = np.floor(np.log2(np.where(x > 0, x, 1.0))); e = np.maximum(e, -6.0)
    q = 2.0 ** (e - 3)
    return s * np.rint(x / q) * q

def colscale(U, V):
    mu, mv = np.abs(U).max(0), np.abs(V).max(0)
    mu[mu == 0] = 1; mv[mv == 0] = 1
    # put both column maxima near 2^7 (e4m3 max 448)
    eu = 7 - np.ceil(np.log2(mu)); ev = 7 - np.ceil(np.log2(mv))
    return 2.0 ** eu, 2.0 ** ev

m, n, k = int(sys.argv[1]), int(sys.argv[2]), 64
X, _, _, _ = orc.synthetic_boolean(m, n, k, (0.05, 0.05), seed=5)
X = orc.flip_noise(X, (0.02, 0.005), seed=6).astype(np.float64)
rng = np.random.RandomState(2024)
U, V = orc.init_factors(X, k, "normal", rng)
U, V = orc.balance_factors(U, V)
U, V = orc.zeros_to_eps(U), orc.zeros_to_eps(V)
reg = 1.0
print("density", X.mean())
for it in range(1, 41):
    V = orc.penalty_update_V_reassoc(X, U, V, reg)
    U = orc.penalty_update_U_reassoc(X, U, V, reg)
    reg *= 1.02
    if it in (1, 2, 3, 5, 8, 12, 16, 20, 25, 30, 35, 40):
        P = U @ V.T
        exact = np.abs(X - P).sum()
        su, sv = colscale(U, V)
        U8, V8 = e4m3(U * su) / su, e4m3(V * sv) / sv
        f8 = np.abs(X - U8 @ V8.T).sum()
        f8u = np.abs(X - U8 @ V.T).sum()
        U16, V16 = U.astype(np.float16).astype(np.float64), V.astype(np.float16).astype(np.float64)
        f16 = np.abs(X - U16 @ V16.T).sum()
        qi = lambda F: np.rint(F / np.abs(F).max(0) * 127) * np.abs(F).max(0) / 127
        i8 = np.abs(X - qi(U) @ qi(V).T).sum()
        print(f"it {it:3d} MAE {exact / X.size:.5f}  fp8/fp8 {f8 / exact - 1:+.2e}  fp8(U only) {f8u / exact - 1:+.2e}  int8 {i8 / exact - 1:+.2e}  fp16 {f16 / exact - 1:+.2e}", flush=True)
