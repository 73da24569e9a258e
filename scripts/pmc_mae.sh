#!/bin/bash
# PMC passes for the MAE pass (mae_kernel): MFMA busy cycles, fabric-side fetch bytes.  Counters in their own runs with
# --kernel-trace only (one TCC counter group per pass).  usage (GPU box): bash scripts/pmc_mae.sh
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmcmae2; rm -rf $OUT; mkdir -p $OUT
ARGS="bench.py --steps 6 --warmup 2 --cpu-rows 0 --alt-operands none --mae 1"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $OUT/sq -- python3 $ARGS > $OUT/sq.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- python3 $ARGS > $OUT/fetch.log 2>&1
python3 - <<PY
import csv, glob, collections
dur = {}
for d in ("sq", "fetch"):
    agg = collections.defaultdict(list)
    f = glob.glob("$OUT/%s/*/*counter_collection.csv" % d)[0]
    t = glob.glob("$OUT/%s/*/*kernel_trace.csv" % d)[0]
    long_ids = set()
    for r in csv.DictReader(open(t)):
        if "mae_kernel" in r["Kernel_Name"] and int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) > 100000:
            long_ids.add(r["Dispatch_Id"]); dur.setdefault(d, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for r in csv.DictReader(open(f)):
        if "mae_kernel" in r["Kernel_Name"] and r["Dispatch_Id"] in long_ids:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in sorted(agg.items()):
        print(f"{d:6s} {k:28s} mean per dispatch {sum(v)/len(v):.4e}  ({len(v)} dispatches, avg {sum(dur[d])/len(dur[d]):.1f} us under the profiler)")
PY
