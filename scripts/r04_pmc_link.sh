#!/bin/bash
# PNLPF / WNMF-KL tile-fused link passes at the headline shape: kernel statistics and SQ counters (separate --pmc runs, --kernel-trace only).
# r04_pmc_link.sh OUTNAME
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r04_link/${1:-a}; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 scripts/link_bench.py > $OUT/stats.log 2>&1
grep "TFLOP" $OUT/stats.log
i=0
for P in "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
         "SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT/p$i -- python3 scripts/link_bench.py > $OUT/p$i.log 2>&1 || { echo "pass $i failed"; tail -3 $OUT/p$i.log; }
done
python3 - <<PY
import csv,glob
f=glob.glob("$OUT/stats/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    n=r['Name']
    if 'anonymous' in n and 'at::' not in n:
        print("%-70s calls %5s avg %8.1f us" % (n.split('(anonymous namespace)::')[1][:68], r['Calls'], float(r['AverageNs'])/1e3))
PY
python3 scripts/pmc_summary.py $OUT | grep "link_"
