#!/bin/bash
set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/s33
timeout -k 10 200 python scripts/r03/masked_bench.py
rm -rf gpurun_out/s33/prof
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/s33/prof -- python3 $GRAFT_REPO_ROOT/scripts/r03/masked_bench.py > $GRAFT_REPO_ROOT/gpurun_out/s33/bench.log 2>&1
cd $GRAFT_REPO_ROOT
cp $(ls -t gpurun_out/s33/prof/*/*kernel_stats.csv | head -1) gpurun_out/s33/kernel_stats_masked.csv
