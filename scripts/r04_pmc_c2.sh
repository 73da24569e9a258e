#!/bin/bash
# Config #2 (WNMF on a real-valued 20000 x 5000 X, k = 32): kernel statistics and PMC passes of the C-side loop (separate --pmc runs,
# --kernel-trace only).  r04_pmc_c2.sh OUTNAME
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r04_c2/${1:-a}; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 scripts/c2_loop.py > $OUT/stats.log 2>&1
i=0
for P in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT/p$i -- python3 scripts/c2_loop.py > $OUT/p$i.log 2>&1 || { echo "pass $i failed"; tail -3 $OUT/p$i.log; }
done
tail -1 $OUT/stats.log
python3 - <<PY
import csv,glob
f=glob.glob("$OUT/stats/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    n=r['Name']
    if 'anonymous' in n and 'at::' not in n:
        print("%-70s calls %5s avg %8.1f us" % (n.split('(anonymous namespace)::')[1][:68], r['Calls'], float(r['AverageNs'])/1e3))
PY
python3 scripts/pmc_summary.py $OUT | grep "xf_f32\|epilogue\|gram" 
