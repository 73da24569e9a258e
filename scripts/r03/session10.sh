#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/s10; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for sh in 0.5 0.63; do
  BMF_I8_OLD_SHARE=$sh rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/f_$sh -- python3 $GRAFT_REPO_ROOT/scripts/gemm_i8_microbench.py 6 > $GRAFT_REPO_ROOT/$OUT/f_$sh.log 2>&1
  BMF_I8_OLD_SHARE=$sh rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/t_$sh -- python3 $GRAFT_REPO_ROOT/scripts/gemm_i8_microbench.py 6 > $GRAFT_REPO_ROOT/$OUT/t_$sh.log 2>&1
done
cd $GRAFT_REPO_ROOT
python - <<PY
import csv, glob, collections
for sh in ("0.5", "0.63"):
    for kind in ("f", "t"):
        rows = collections.defaultdict(dict)
        for f in glob.glob("$OUT/%s_%s/*/*counter_collection.csv" % (kind, sh)):
            for r in csv.DictReader(open(f)):
                if "xf_bits_i8" in r["Kernel_Name"]:
                    rows[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
        ids = sorted(rows)
        # the microbench launches XV 5 + 6 times, then XtU 5 + 6 times
        half = len(ids) // 2
        for name, sel in (("XV", ids[:half]), ("XtU", ids[half:])):
            agg = collections.defaultdict(list)
            for i in sel:
                for k, v in rows[i].items():
                    agg[k].append(v)
            print(sh, kind, name, {k: round(sum(v) / len(v) / 1e3, 1) for k, v in agg.items()}, "(thousands; FETCH_SIZE in KiB: x2 x1024 = bytes)")
PY
