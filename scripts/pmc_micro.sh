#!/bin/bash
# PMC counters of the GEMM microbench for kernel flavours: pmc_micro.sh "flavours" -> clock, MFMA busy share, waits
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r02
for f in $1; do
  lib=libbmf_$f.so; [ "$f" = base ] && lib=libbmf_hip.so
  OUT=gpurun_out/r02/pmc_micro_$f; rm -rf $OUT
  BMF_LIB=$lib rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT -- python3 scripts/gemm_i8_microbench.py 10 > $OUT.log 2>&1
  python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(list); dur=[]
for fn in glob.glob("$OUT/*/*counter_collection.csv"):
    for r in csv.DictReader(open(fn)):
        if 'xf_bits_i8' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
for fn in glob.glob("$OUT/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(fn)):
        if 'xf_bits_i8' in r['Kernel_Name']: dur.append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
m={k:sum(v)/len(v) for k,v in acc.items()}
us=sum(dur)/len(dur)
cyc=m['GRBM_GUI_ACTIVE']/8
print(f"$f: {us:.1f} us (profiled), clock {cyc/us/1e3:.2f} GHz, MFMA busy {m['SQ_VALU_MFMA_BUSY_CYCLES']/(cyc*1024):.3f} of SIMD cycles, "
      f"wave cycles: wait_any {m['SQ_WAIT_ANY']/m['SQ_WAVE_CYCLES']:.2f} wait_inst {m['SQ_WAIT_INST_ANY']/m['SQ_WAVE_CYCLES']:.2f} active {m['SQ_ACTIVE_INST_ANY']/m['SQ_WAVE_CYCLES']:.2f}")
PY
done
