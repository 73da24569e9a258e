#!/bin/bash
# The judged artefacts of a round in one go: default bench line, rocprofv3 kernel stats of the same command, PMC passes of the GEMM.
# usage (on the GPU box): scripts/final_profiles.sh TAG   -> gpurun_out/TAG/{bench.json, stats/, pmc/}
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$1; mkdir -p $OUT
python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
cut -c1-300 $OUT/bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --cpu-rows 0 --traffic 0 > $OUT/bench_under_profiler.log 2>&1
cp $OUT/stats/*/*kernel_stats.csv $OUT/kernel_stats.csv
scripts/trace_i8.sh $1/headline_only --traffic 0 > $OUT/headline_only_summary.txt 2>&1
cp $OUT/headline_only/*/*kernel_stats.csv $OUT/kernel_stats_headline_only.csv
cat $OUT/headline_only_summary.txt | head -8
scripts/pmc_i8.sh $1/pmc > /dev/null 2>&1
cat $OUT/pmc/summary.md | grep "xf_bits_i8" | head -20
