"""Is the 3-rank blocked run reproducible?  Runs the C-side loop twice and the host-driven protocol twice (gloo, three ranks on one GPU)
and compares the four results pairwise, and each with the single-engine run.  Diagnostic for tests/test_sharded_gpu.py."""
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np
import torch.multiprocessing as mp

import oracle as orc
import test_sharded_gpu as T

if __name__ == "__main__":
    world, panel, m, k = 3, "i8", 1100, 64
    blocked = os.environ.get("PROBE_BLOCKED", "1") == "1"
    X, _, _, _ = orc.synthetic_boolean(m, 700, 12, (0.15, 0.15), seed=41)
    X = orc.flip_noise(X, (0.05, 0.01), seed=42).astype(np.uint8)
    U0, V0 = orc.init_factors(X, k, "normal", np.random.RandomState(8))
    U0, V0 = orc.balance_factors(U0, V0)
    U0, V0 = orc.zeros_to_eps(U0), orc.zeros_to_eps(V0)
    regs = [1.0 * 1.05 ** i for i in range(8)]
    runs = {}
    for tag, loop in (("c1", "c"), ("c2", "c"), ("p1", "python"), ("p2", "python")):
        d = tempfile.mkdtemp()
        mp.spawn(T.worker, args=(world, T.free_port(), X, U0, V0, regs, d, panel, blocked, loop), nprocs=world, join=True)
        parts = [np.load(os.path.join(d, f"r{r}{loop}.npz")) for r in range(world)]
        runs[tag] = (np.concatenate([p["U"] for p in parts]), parts[0]["V"], parts[0]["log"])
    rel = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))
    for a in runs:
        for b in runs:
            if a < b:
                print(f"{a} vs {b}: U identical {np.array_equal(runs[a][0], runs[b][0])} (rel {rel(runs[a][0], runs[b][0]):.2e}), V identical {np.array_equal(runs[a][1], runs[b][1])}, "
                      f"log[:, :6] identical {np.array_equal(runs[a][2][:, :6], runs[b][2][:, :6])}")
