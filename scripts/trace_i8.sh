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
# launch order with --secondary 0 --alt-operands none: prepare (2), P pre-heating iterations (2 each; the first W + K of them are the
# cold start), prepare (2), W warm-up, K timed, the oracle-checked extra update (2)
W, K, P = 3, 20, 150
cold = g[2 + 2 * W: 2 + 2 * (W + K)]
t0 = 2 + 2 * P + 2 + 2 * W
timed = g[t0:t0 + 2 * K]
print("bits GEMM launches:", len(g))
print("  cold start  (first %d timed-shape launches of the process): avg %.1f us" % (len(cold), sum(cold) / max(len(cold), 1)))
print("  timed region (%d launches after %d pre-heating iterations):   avg %.1f us   X V %.1f  X^T U %.1f" %
      (len(timed), P, sum(timed) / max(len(timed), 1), sum(timed[0::2]) / max(len(timed[0::2]), 1), sum(timed[1::2]) / max(len(timed[1::2]), 1)))
f=glob.glob("$OUT/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    n=r['Name']
    if 'anonymous' in n:
        print(f"{n.split('(anonymous namespace)::')[1][:60]:60s} calls={r['Calls']:>4} avg_us={float(r['AverageNs'])/1e3:9.1f} min={float(r['MinNs'])/1e3:8.1f} max={float(r['MaxNs'])/1e3:8.1f}")
PY
