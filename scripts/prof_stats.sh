#!/bin/bash
# rocprofv3 kernel-trace stats of the bench; prints our kernels' averages.  usage: prof_stats.sh OUTNAME [bench args]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$1; shift; rm -rf $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 bench.py --steps 20 --warmup 3 --cpu-rows 0 --alt-operands none "$@" > $OUT.log 2>&1
grep -h metric $OUT.log | cut -c1-200
python3 - <<PY
import csv,glob
f=glob.glob("$OUT/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    n=r['Name']
    if 'anonymous' in n:
        print(f"{n.split('(anonymous namespace)::')[1][:48]:48s} calls={r['Calls']:>4} avg_us={float(r['AverageNs'])/1e3:9.1f} min={float(r['MinNs'])/1e3:8.1f} max={float(r['MaxNs'])/1e3:8.1f}")
PY
