#!/bin/bash
# functional rehearsal of the N = 2 bench path on one GPU (gloo; the numbers mean nothing)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/s38
BMF_BENCH_REHEARSAL=1 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 5 --warmup 2 --cpu-rows 0 --preheat 0 > gpurun_out/s38/bench2.json 2> gpurun_out/s38/bench2.err
echo "rc $?"; tail -c 1500 gpurun_out/s38/bench2.json; tail -5 gpurun_out/s38/bench2.err
