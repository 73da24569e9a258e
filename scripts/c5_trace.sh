#!/bin/bash
# kernel-trace of the C5 (thresholding line search) leg of bench.py
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r02/c5_trace; rm -rf $OUT; mkdir -p gpurun_out/r02
cat > /tmp/c5_run.py <<PY
import sys; sys.path.insert(0, "$GRAFT_REPO_ROOT")
import bench, json
print(json.dumps(bench.secondary_c5()))
PY
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 /tmp/c5_run.py > $OUT.log 2>&1
tail -1 $OUT.log | cut -c1-300
python3 - <<PY
import csv,glob
f=glob.glob("$OUT/*/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:14]:
    n=r['Name']
    print(f"{n.split('(anonymous namespace)::')[1][:60]:60s} calls={r['Calls']:>5} avg_us={float(r['AverageNs'])/1e3:9.1f} total_ms={float(r['TotalDurationNs'])/1e6:8.1f}")
PY
