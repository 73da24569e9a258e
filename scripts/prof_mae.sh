#!/bin/bash
# kernel stats of the bench with the MAE pass in the loop.  usage: prof_mae.sh OUTNAME
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$1; rm -rf $OUT; mkdir -p $(dirname $OUT)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 bench.py --steps 20 --warmup 3 --cpu-rows 0 --alt-operands none --secondary 0 --traffic 0 --mae 1 > $OUT.log 2>&1
grep -h metric $OUT.log | cut -c1-160
python3 - <<PY
import csv,glob
f=glob.glob("$OUT/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    n=r['Name']
    if 'anonymous' in n:
        print(f"{n.split('(anonymous namespace)::')[1][:60]:60s} calls={r['Calls']:>4} avg_us={float(r['AverageNs'])/1e3:9.1f} min={float(r['MinNs'])/1e3:8.1f} max={float(r['MaxNs'])/1e3:8.1f}")
PY
