"""Where does bmf_xf_f32 differ from A @ FT^T?  (error by split count, rows, columns)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from pybmf_amd import _lib as L
d = torch.device("cuda", 0)
rs = np.random.RandomState(4)
for kp, red_pad in [(32, 512), (64, 640), (32, 1344)]:
    rows_pad = 384
    A = np.zeros((rows_pad, red_pad), np.float32); A[:300, :red_pad - 12] = rs.rand(300, red_pad - 12)
    FT = np.zeros((kp, red_pad), np.float32); FT[:, :red_pad - 12] = rs.rand(kp, red_pad - 12)
    Ad, FTd = torch.from_numpy(A).to(d), torch.from_numpy(FT).to(d)
    want = A.astype(np.float64) @ FT.T.astype(np.float64)
    for splits in (1, 3, 8):
        out = torch.full((splits, rows_pad, kp), np.nan, dtype=torch.float32, device=d)
        L.check(L.lib.bmf_xf_f32(L.ptr(Ad), rows_pad, red_pad, red_pad, L.ptr(FTd), red_pad, kp, L.ptr(out), rows_pad * kp, splits, None))
        torch.cuda.synchronize()
        got = out.sum(0).double().cpu().numpy()
        err = np.abs(got - want)
        print(f"kp {kp} red_pad {red_pad} splits {splits}: rel {np.linalg.norm(got - want) / np.linalg.norm(want):.3e} nan {np.isnan(got).sum()} "
              f"bad rows {np.where(err.max(1) > 1e-3)[0][:12]} bad cols {np.where(err.max(0) > 1e-3)[0][:12]}")
        if splits == 1 and np.nanmax(err) > 1e-3:
            i, j = np.unravel_index(np.nanargmax(err), err.shape)
            print("   worst", i, j, got[i, j], want[i, j], " ratio", got[i, j] / want[i, j])
            # which part of the reduction is missing / doubled?  compare against per-64 stage partial sums
            parts = np.array([A[i, 64 * s:64 * s + 64].astype(np.float64) @ FT[j, 64 * s:64 * s + 64] for s in range(red_pad // 64)])
            print("   stage parts", np.round(parts, 3), "sum", parts.sum())
