#!/bin/bash
# Round 5: kernel statistics of the headline loop at a given row count (rocprofv3 --kernel-trace --stats): r05_kernel_stats.sh M [NAME] [ENV=VAL ...]
#   -> gpurun_out/r05/kernel_stats_NAME.csv (+ a one-line-per-kernel table on stdout)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
M=${1:-100000}; NAME=${2:-m$M}; shift; shift
for kv in "$@"; do export "$kv"; done
OUT=gpurun_out/r05/prof_$NAME; rm -rf $OUT; mkdir -p $OUT
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT -- python3 $GRAFT_REPO_ROOT/bench.py --m $M --steps 60 --warmup 5 --cpu-rows 0 --traffic 0 --secondary 0 --sustained 0 > $GRAFT_REPO_ROOT/$OUT.log 2>&1)
cp $OUT/*/*kernel_stats.csv gpurun_out/r05/kernel_stats_$NAME.csv
python3 - <<PY
import csv
print("== $NAME (m = $M)")
for row in csv.DictReader(open("gpurun_out/r05/kernel_stats_$NAME.csv")):
    n = row["Name"]
    if "anonymous" in n and "at::" not in n and int(row["Calls"]) > 20:
        print("%-62s calls %5s avg %8.1f us min %8.1f max %8.1f" % (n.split("(anonymous namespace)::")[1][:60], row["Calls"], float(row["AverageNs"]) / 1e3, float(row["MinNs"]) / 1e3, float(row["MaxNs"]) / 1e3))
PY
grep "^{" $OUT.log | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench line: value', d['value'], 'ms_per_step', d['ms_per_step'], 'repeat', d.get('repeat',{}).get('legs_of_K_steps'))"
