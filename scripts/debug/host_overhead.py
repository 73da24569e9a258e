"""Host-side cost of one sharded iteration (Python-driven: head, X^T U blocks, all-reduces, finalize), phase by phase.
BMF_FORCE_SHARDED-style: one rank, nccl.  usage: python scripts/debug/host_overhead.py [m]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, torch.distributed as dist
m = int(sys.argv[1]) if len(sys.argv) > 1 else 12500
n, k = 20000, 64
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
from pybmf_amd import _lib as L
from pybmf_amd.engine import BitMatrix, MUEngine
from pybmf_amd.generators import PlantedBooleanOnDevice
from bench import host_init
gen = PlantedBooleanOnDevice(m, n, k, density=(0.067, 0.067), seed=1000, noise=(0.05, 0.01), noise_seed=2000, device=dev)
X = BitMatrix(gen, dev)
eng = MUEngine(X, k=k, mode=L.MODE_PENALTY, terms=3, with_mae=False, tol=0.0, min_diff=0.0, max_iter=2000, sharded=True, panel="i8")
U0, V0 = host_init(eng.sum_x / (float(m) * n), m, n, k, seed=2024)
eng.load_factors(U0, V0)
eng.prepare(1.0)
for i in range(10):
    eng.step(1 + i, 1.0)
torch.cuda.synchronize()
N = 200
SYNC_EACH = os.environ.get('SYNC_EACH', '1') == '1'   # 1: drain the queue after every step, so the phases carry no back-pressure
t_enq_acc = 0.0
acc = dict(head=0.0, xtu=0.0, ar=0.0, wait=0.0, fin=0.0)
pc = time.perf_counter
t_all = pc()
for i in range(N):
    t_step = pc()
    t0 = pc(); eng.local_update_head(1.0); t1 = pc(); acc["head"] += t1 - t0
    pend = []
    for b in range(eng.n_blocks()):
        t0 = pc(); eng.local_xtu_block(b); t1 = pc(); acc["xtu"] += t1 - t0
        pend += eng._all_reduce_async([eng.exchange_block(b)] + ([eng.exchange_scalars()] if b == 0 else [])); t2 = pc(); acc["ar"] += t2 - t1
    t0 = pc()
    for h in pend: h.wait()
    t1 = pc(); acc["wait"] += t1 - t0
    eng.finalize(11 + i, 1.0); t2 = pc(); acc["fin"] += t2 - t1
    if SYNC_EACH:
        t_enq_acc += pc() - t_step
        torch.cuda.synchronize()
t_enq = t_enq_acc if SYNC_EACH else pc() - t_all
torch.cuda.synchronize()
t_tot = pc() - t_all
print(f"m={m}: host enqueue {1e3*t_enq/N:.4f} ms/step, with final sync {1e3*t_tot/N:.4f} ms/step")
print({k_: round(1e6 * v / N, 1) for k_, v in acc.items()}, "us/step")
dist.destroy_process_group()
