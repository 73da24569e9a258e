#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/s45; rm -rf gpurun_out/s45/prof
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/s45/prof -- python3 $GRAFT_REPO_ROOT/scripts/r03/wnmf_real_masked_bench.py > $GRAFT_REPO_ROOT/gpurun_out/s45/bench.log 2>&1
cd $GRAFT_REPO_ROOT
grep "WNMF on ratings\|Error\|error" gpurun_out/s45/bench.log | head -5
cp $(ls -t gpurun_out/s45/prof/*/*kernel_stats.csv | head -1) gpurun_out/s45/kernel_stats.csv
