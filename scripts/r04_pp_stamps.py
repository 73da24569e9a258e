"""Phase stamps of link_pass16pp_kernel (diagnostic flavour libbmf_ppstamp.so, -DBMF_PP_STAMP): where a phase's cycles go for one wave of
each group of workgroup (0, 0).  usage: BMF_LIB=libbmf_ppstamp.so python scripts/r04_pp_stamps.py"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import host_init
from pybmf_amd import _lib as L
from pybmf_amd.engine import BitMatrix, LinkMUEngine
from pybmf_amd.generators import PlantedBooleanOnDevice
m, n, k = 100_000, 20_000, 64
dev = torch.device("cuda:0")
X = BitMatrix(PlantedBooleanOnDevice(m, n, k, density=(0.067, 0.067), seed=1000, noise=(0.05, 0.01), noise_seed=2000, device=dev), dev)
U0, V0 = host_init(X.sum_local / (float(m) * n), m, n, k, seed=2024)
for name, link, mode in (("sigmoid", L.LINK_SIGMOID, L.MODE_PENALTY), ("KL", L.LINK_KL, L.MODE_WNMF)):
    eng = LinkMUEngine(X, k, link, mode, lamda=10.0)
    eng.load_factors(U0, V0); eng.prepare()
    for _ in range(3): eng.update(1.0)
    torch.cuda.synchronize()
    buf = np.zeros((2, 512, 4), dtype=np.uint64)
    fn = L.lib.bmf_debug_pp_stamps; fn.restype = C.c_int; fn.argtypes = [C.c_void_p]
    assert fn(buf.ctypes.data_as(C.c_void_p)) == 0
    s = buf.astype(np.int64)
    print(name, 'HW_ID of waves 0-3:', [hex(int(v)) for v in s[0, 511]], 'SIMD', [int(v >> 4) & 3 for v in s[0, 511]], '; waves 4-7:', [hex(int(v)) for v in s[1, 511]], 'SIMD', [int(v >> 4) & 3 for v in s[1, 511]])
    c = s[0, 510]
    print(f"{name}: workgroup (0,0) ran {(c[3]-c[1])*10:.0f} ns = {c[2]-c[0]} s_memtime ticks -> {(c[2]-c[0])/((c[3]-c[1])*10):.3f} ticks per ns")
    for g, nm in ((0, "group A (M then V)"), (1, "group B (V then M)")):
        t = s[g, 20:180]   # steady state
        first, bar1, second, bar2 = t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2], np.r_[t[1:, 0] - t[:-1, 3], 0]
        period = np.diff(t[:, 0])
        print(f"{name} {nm}: period {np.median(period):.0f} cycles; first half work {np.median(first):.0f}, barrier {np.median(bar1):.0f}, "
              f"second half work {np.median(second):.0f}, barrier {np.median(bar2[:-1]):.0f}")
    del eng
