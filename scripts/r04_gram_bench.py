"""bmf_gram_partial alone at the shapes of the headline loop (U: 100352 x 64, V: 20480 x 64; 256 blocks): us per launch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pybmf_amd import _lib as L
d = torch.device("cuda", 0)
for rows_pad, kp, blocks in ((100352, 64, 256), (20480, 64, 256), (12544, 64, 256), (20096, 32, 256)):
    F = torch.rand((rows_pad, kp), device=d)
    slabs = torch.zeros((blocks, kp, kp), device=d)
    run = lambda: L.check(L.lib.bmf_gram_partial(L.ptr(F), rows_pad, kp, kp, L.ptr(slabs), blocks, None))
    for _ in range(5): run()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(40)]
    for a, b in ev:
        a.record(); run(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
    want = F.double().t() @ F.double()
    err = ((slabs.double().sum(0) - want).abs().max() / want.abs().max()).item()
    print(f"gram_partial {rows_pad} x {kp}, {blocks} blocks: median {ts[20]:.1f} us min {ts[0]:.1f} us ({rows_pad * kp * 4 / ts[20] / 1e6:.2f} TB/s); max rel err vs fp64 {err:.1e}")
