#!/bin/bash
# rocprofv3 kernel-trace stats of BASELINE config #2 (WNMF, 20000 x 5000 fp32, k = 32) in the C-side loop.  usage: prof_c2.sh OUTNAME
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$1; rm -rf $OUT; mkdir -p $(dirname $OUT)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 scripts/c2_loop.py > $OUT.log 2>&1
cat $OUT.log | tail -3
python3 - <<PY
import csv,glob
f=glob.glob("$OUT/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    n=r['Name']
    if 'anonymous' in n:
        print(f"{n.split('(anonymous namespace)::')[1][:60]:60s} calls={r['Calls']:>4} avg_us={float(r['AverageNs'])/1e3:9.1f} min={float(r['MinNs'])/1e3:8.1f} max={float(r['MaxNs'])/1e3:8.1f}")
PY
