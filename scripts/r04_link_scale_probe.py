import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
from pybmf_amd import _lib as L
from pybmf_amd.engine import BitMatrix, LinkMUEngine
m, n, k = 300, 260, 64
rs = np.random.RandomState(m + n + k)
X = (rs.rand(m, n) < 0.3).astype(np.float64)
U = np.abs(rs.standard_normal((m, k))) * 0.4 + 1e-3
V = np.abs(rs.standard_normal((n, k))) * 0.4 + 1e-3
U[:, 0] *= 3e-5; V[:, 0] *= 1e5; U[:, k - 1] *= 2e4; V[:, k - 1] *= 4e-5; U[:, 1] = 0.0
relf = lambda a, b: np.linalg.norm(a - b) / np.linalg.norm(b)
for how in ("pair", "single"):
    eng = LinkMUEngine(BitMatrix(X.astype(np.uint8), "cuda:0"), k, L.LINK_KL, L.MODE_WNMF, mfma="bf16")
    eng.load_factors(U, V); eng.prepare()
    if how == "single":
        L.check(L.lib.bmf_link_split(L.ptr(eng.U), eng.m_pad, eng.kp, L.ptr(eng.wsU), None)); L.check(L.lib.bmf_link_split(L.ptr(eng.V), eng.n_pad, eng.kp, L.ptr(eng.wsV), None))
    Xb = eng.X
    L.check(L.lib.bmf_link_pass16(L.ptr(Xb.bits), eng.m_pad, Xb.ldx, m, n, L.ptr(eng.wsU), L.ptr(eng.wsV), eng.n_pad, eng.kp, L.LINK_KL, 7.0, L.ptr(eng.numU), None, eng.m_pad * eng.kp, eng.splitsU, None))
    torch.cuda.synchronize()
    Uf, Vf = U.astype(np.float32).astype(np.float64), V.astype(np.float32).astype(np.float64)
    want = (X / (Uf @ Vf.T)) @ Vf
    print(how, "scales: relative error of the KL numerator", relf(eng.numU.double().sum(0).cpu().numpy()[:m, :k], want))
