#!/bin/bash
set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/s30

rm -rf gpurun_out/s30/prof
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/s30/prof -- python3 $GRAFT_REPO_ROOT/scripts/palm_bench.py > $GRAFT_REPO_ROOT/gpurun_out/s30/bench.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(ls -t gpurun_out/s30/prof/*/*kernel_stats.csv | head -1); cp $f gpurun_out/s30/kernel_stats_elbmf.csv; head -20 $f | cut -c1-160
cp $(ls -t gpurun_out/s30/prof/*/*kernel_trace.csv | head -1) gpurun_out/s30/kernel_trace_elbmf.csv
