"""Is the Python-driven sharded loop host-bound?  One rank (BMF_FORCE_SHARDED-style: RCCL initialised, world 1), rows = M:
time to ENQUEUE `steps` iterations vs time until the GPU has finished them.  usage: M=12500 python scripts/shard_host_probe.py"""
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pybmf_amd import _lib as L  # noqa: E402
from pybmf_amd.engine import BitMatrix, MUEngine  # noqa: E402
from pybmf_amd.generators import PlantedBooleanOnDevice  # noqa: E402

m, n, k, steps = int(os.environ.get("M", 12500)), 20000, 64, int(os.environ.get("STEPS", 200))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29544")
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
gen = PlantedBooleanOnDevice(m, n, k, density=(0.067, 0.067), seed=1000, noise=(0.05, 0.01), noise_seed=2000, device=dev)
X = BitMatrix(gen, dev)
eng = MUEngine(X, k=k, mode=L.MODE_PENALTY, with_mae=False, tol=-1.0, min_diff=-1.0, max_iter=steps + 10, sharded=True, panel="f16")
rs = np.random.RandomState(0)
eng.load_factors(rs.rand(m, k) * 0.2, rs.rand(n, k) * 0.2)
eng.prepare(1.0)
eng.run([1.0] * 5, it0=1)
torch.cuda.synchronize()
t0 = time.perf_counter()
eng.run([1.0] * steps, it0=6)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"m={m}: enqueue {1e6 * (t1 - t0) / steps:.1f} us/iter (host), finished {1e6 * (t2 - t0) / steps:.1f} us/iter (GPU)")
dist.destroy_process_group()
