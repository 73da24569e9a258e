#!/bin/bash
# Round 4: the artefacts under profiles/ that DESIGN.md, profiles/README.md and the bench line cite, from the final build (run on the GPU
# box via gpurun; the copies into profiles/ are made afterwards from gpurun_out/final4/).   r04_final_profiles.sh [part ...]   parts: bench stats shards c2 c5 widened parity
set -o pipefail
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/final4; mkdir -p $OUT
export TMPDIR=/tmp
PARTS="${@:-bench stats shards c2 c5 widened parity}"
has() { [[ " $PARTS " == *" $1 "* ]]; }
stats_table() {   # kernel_stats.csv -> one line per kernel of this library
python3 - <<PY
import csv, glob
f = glob.glob("$1/*/*kernel_stats.csv")[0]
print("== $1")
for row in csv.DictReader(open(f)):
    n = row["Name"]
    if "anonymous" in n and "at::" not in n:
        print("%-62s calls %5s avg %8.1f us min %8.1f max %8.1f" % (n.split("(anonymous namespace)::")[1][:60], row["Calls"], float(row["AverageNs"]) / 1e3, float(row["MinNs"]) / 1e3, float(row["MaxNs"]) / 1e3))
PY
}
if has bench; then   # the default bench line (what the driver runs), twice
  timeout -k 10 550 python bench.py > $OUT/bench_default_a.json 2> $OUT/bench_default_a.err; echo "bench a rc $?"
  timeout -k 10 550 python bench.py > $OUT/bench_default_b.json 2> $OUT/bench_default_b.err; echo "bench b rc $?"
fi
if has stats; then   # kernel statistics of the headline loop alone and with the MAE pass
  rm -rf $OUT/prof_headline $OUT/prof_mae
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_headline -- python3 $GRAFT_REPO_ROOT/bench.py --steps 60 --warmup 5 --cpu-rows 0 --traffic 0 --secondary 0 --sustained 0 > $GRAFT_REPO_ROOT/$OUT/prof_headline.log 2>&1)
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_mae -- python3 $GRAFT_REPO_ROOT/bench.py --steps 60 --warmup 5 --cpu-rows 0 --traffic 0 --secondary 0 --sustained 0 --mae 1 > $GRAFT_REPO_ROOT/$OUT/prof_mae.log 2>&1)
  stats_table $OUT/prof_headline | tee $OUT/prof_headline.txt; stats_table $OUT/prof_mae | tee $OUT/prof_mae.txt
fi
if has shards; then scripts/r04_shard_sizes.sh | tee $OUT/shard_sizes.txt; fi
if has c2; then
  rm -rf $OUT/prof_c2
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_c2 -- python3 $GRAFT_REPO_ROOT/scripts/c2_loop.py > $GRAFT_REPO_ROOT/$OUT/prof_c2.log 2>&1)
  tail -1 $OUT/prof_c2.log; stats_table $OUT/prof_c2 | tee $OUT/prof_c2.txt
fi
if has c5; then timeout -k 10 200 python scripts/c5_bench.py 2>&1 | tail -4 | tee $OUT/c5.txt; fi
if has widened; then
  timeout -k 10 200 python scripts/link_bench.py 2>&1 | grep TFLOP | tee $OUT/link_bench.txt
  timeout -k 10 120 python scripts/r04_masked_ab.py 2>&1 | tail -2 | tee $OUT/masked_ab.txt
  timeout -k 10 200 python scripts/palm_bench.py 2>&1 | tail -2 | tee $OUT/palm_bench.txt
fi
if has parity; then python -m pytest tests/test_c3_parity_gpu.py -m gpu -q -s 2>&1 | grep -E "c3 parity|passed|failed" | tee $OUT/parity_c3.txt; fi
