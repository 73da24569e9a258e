#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/s41; rm -rf gpurun_out/s41/prof
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/s41/prof -- python3 $GRAFT_REPO_ROOT/bench.py --m 12500 --steps 60 --warmup 5 --cpu-rows 0 --traffic 0 --secondary 0 --sustained 0 > $GRAFT_REPO_ROOT/gpurun_out/s41/bench.log 2>&1
cd $GRAFT_REPO_ROOT
cp $(ls -t gpurun_out/s41/prof/*/*kernel_stats.csv | head -1) gpurun_out/s41/kernel_stats_m12500.csv
cp $(ls -t gpurun_out/s41/prof/*/*kernel_trace.csv | head -1) gpurun_out/s41/kernel_trace_m12500.csv
