#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/s11; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for lead in 0 50 100 150; do
  BMF_I8_LEAD=$lead rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/f_$lead -- python3 $GRAFT_REPO_ROOT/scripts/gemm_i8_microbench.py 6 > $GRAFT_REPO_ROOT/$OUT/f_$lead.log 2>&1
done
cd $GRAFT_REPO_ROOT
python - <<PY
import csv, glob, collections
for lead in ("0", "50", "100", "150"):
    rows = collections.defaultdict(dict)
    for f in glob.glob("$OUT/f_%s/*/*counter_collection.csv" % lead):
        for r in csv.DictReader(open(f)):
            if "xf_bits_i8" in r["Kernel_Name"]:
                rows[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
    ids = sorted(rows); half = len(ids) // 2
    for name, sel in (("XV", ids[:half]), ("XtU", ids[half:])):
        v = [rows[i]["FETCH_SIZE"] for i in sel]
        print("lead", lead, name, "FETCH_SIZE x2 = %.0f MB" % (sum(v) / len(v) * 2 * 1024 / 1e6))
PY
for lead in 0 50 100; do
  echo "== lead $lead"
  BMF_I8_LEAD=$lead timeout -k 10 200 python scripts/gemm_i8_microbench.py 40 2>&1 | tail -1
done
