#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/s43; rm -rf gpurun_out/s43/prof
timeout -k 10 300 python scripts/r03/prediction_bench.py 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/s43/prof -- python3 $GRAFT_REPO_ROOT/scripts/r03/prediction_bench.py > $GRAFT_REPO_ROOT/gpurun_out/s43/bench.log 2>&1
cd $GRAFT_REPO_ROOT
cp $(ls -t gpurun_out/s43/prof/*/*kernel_stats.csv | head -1) gpurun_out/s43/kernel_stats_prediction.csv
