#!/bin/bash
# Functional rehearsal of the driver's N > 1 launch on a one-GPU box: every rank on cuda:0, exchange over gloo (numbers mean nothing)
set -o pipefail
for n in 2 3; do
  echo "== N=$n"
  BMF_BENCH_REHEARSAL=1 timeout -k 10 280 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29510 + n)) \
     bench.py --gpus $n --steps 20 --warmup 5 --sustained 40 --preheat 30 2> gpurun_out/r04_rehearsal_n$n.err | tee gpurun_out/r04_rehearsal_n$n.json || { tail -30 gpurun_out/r04_rehearsal_n$n.err; exit 1; }
done
