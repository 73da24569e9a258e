#!/bin/bash
set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/s39
for w in pnlpf kl wide; do
  rm -rf gpurun_out/s39/prof_$w
  (cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/s39/prof_$w -- python3 $GRAFT_REPO_ROOT/scripts/r03/link_wide_bench.py $w > $GRAFT_REPO_ROOT/gpurun_out/s39/$w.log 2>&1)
  grep -h "it/s" gpurun_out/s39/$w.log
  cp $(ls -t gpurun_out/s39/prof_$w/*/*kernel_stats.csv | head -1) gpurun_out/s39/kernel_stats_$w.csv
done
