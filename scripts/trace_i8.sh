#!/bin/bash
# kernel-trace of the bench: per-dispatch durations of the bits GEMM in launch order (X V and X^T U alternate).  usage: trace_i8.sh OUT [bench args]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$1; shift; rm -rf $OUT; mkdir -p $(dirname $OUT)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 bench.py --steps 20 --warmup 3 --cpu-rows 0 --alt-operands none --secondary 0 "$@" > $OUT.log 2>&1
grep -h metric $OUT.log | cut -c1-150
python3 - <<PY
import csv,glob,collections
f=glob.glob("$OUT/*/*kernel_trace.csv")[0]
rows=sorted(csv.DictReader(open(f)), key=lambda r:int(r['Start_Timestamp']))
g=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in rows if 'xf_bits' in r['Kernel_Name']]
g=g[8:]
print("bits GEMM launches:", len(g), "even (X V) avg us %.1f  odd (X^T U) avg us %.1f" % (sum(g[0::2])/len(g[0::2]), sum(g[1::2])/len(g[1::2])))
f=glob.glob("$OUT/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    n=r['Name']
    if 'anonymous' in n:
        print(f"{n.split('(anonymous namespace)::')[1][:60]:60s} calls={r['Calls']:>4} avg_us={float(r['AverageNs'])/1e3:9.1f} min={float(r['MinNs'])/1e3:8.1f} max={float(r['MaxNs'])/1e3:8.1f}")
PY
