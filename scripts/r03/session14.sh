#!/bin/bash
# MAE pass: what clock does the chip hold, how busy is the matrix pipe?  PMC on the microbench (full kernel and two ablations)
set -o pipefail
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/s14; rm -rf $OUT; mkdir -p $OUT
cd /tmp
for l in hip m32_ONLY_MFMA m32_NO_DMA; do
  BMF_LIB=libbmf_$l.so rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/$l -- python3 $GRAFT_REPO_ROOT/scripts/mae_bench.py > $OUT/$l.log 2>&1
done
cd $GRAFT_REPO_ROOT
python3 - <<PY
import csv, glob, collections
for l in ("hip", "m32_ONLY_MFMA", "m32_NO_DMA"):
    f = glob.glob("$OUT/%s/*/*counter_collection.csv" % l)[0]
    t = glob.glob("$OUT/%s/*/*kernel_trace.csv" % l)[0]
    dur = {}
    for r in csv.DictReader(open(t)):
        if "mae32" in r["Kernel_Name"]:
            dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "mae32" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    us = sum(dur.values()) / len(dur)
    m = {k: sum(v) / len(v) for k, v in agg.items()}
    cyc = m["GRBM_GUI_ACTIVE"] / 8
    print(f"{l:16s} {us:7.1f} us  span {cyc/1e3:.0f} k cycles = {cyc/us/1e3:.2f} GHz  MFMA busy {m['SQ_VALU_MFMA_BUSY_CYCLES']/1024/cyc*100:.1f} %  VALU insts {m['SQ_INSTS_VALU']:.3e}  wave-cycles {m['SQ_WAVE_CYCLES']:.3e} wait_inst {m['SQ_WAIT_INST_ANY']/m['SQ_WAVE_CYCLES']*100:.0f} % wait_any {m['SQ_WAIT_ANY']/m['SQ_WAVE_CYCLES']*100:.0f} % active {m['SQ_ACTIVE_INST_ANY']/m['SQ_WAVE_CYCLES']*100:.0f} %")
PY
