"""Per-iteration drift of the GPU trajectory from the fp64 oracle at the headline configuration (100 000 x 20 000, k = 64,
bench.py's generator / init / schedule), for several operand formats at once.  Measurement aid (imports tests/c3_lockstep.py and
the oracle); its output is committed under profiles/.

    python scripts/parity_trace.py [n_iter=100] [operands=f16x2,bf16x3] > gpurun_out/parity_trace.txt
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from c3_lockstep import bench_problem, lockstep  # noqa: E402

n_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 100
ops = tuple((sys.argv[2] if len(sys.argv) > 2 else "f16x2,bf16x3").split(","))
m, n, k = (int(v) for v in os.environ.get("SHAPE", "100000,20000,64").split(","))
X, U0, V0, regs = bench_problem(m, n, k, n_iter=n_iter)
print(f"# {m} x {n}, k = {k}; rel = ||GPU - oracle||_F / ||oracle||_F per iteration; operands {ops}", flush=True)
worst = {o: [0.0, 0, 0.0, 0] for o in ops}
for it, res, extras in lockstep(X, U0, V0, regs, n_iter, operands=ops, scalars_every=10, out=sys.stdout):
    for o in ops:
        if res[o][0] > worst[o][0]:
            worst[o][0:2] = [res[o][0], it]
        if res[o][1] > worst[o][2]:
            worst[o][2:4] = [res[o][1], it]
for o in ops:
    print(f"# worst {o}: U {worst[o][0]:.3e} at iteration {worst[o][1]}, V {worst[o][2]:.3e} at iteration {worst[o][3]}")
