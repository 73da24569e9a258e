#!/bin/bash
# Round 3, GPU session 1: full GPU test suite, the C-side sharded loop against the unsharded loop at the shard sizes of 1 / 4 / 8
# GPUs (RCCL with one rank), SQ counters of the int8 GEMM for the stall attribution, the default bench line.
set -o pipefail
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/s1; mkdir -p $OUT
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc $?" | tee -a $OUT/pytest.log
tail -5 $OUT/pytest.log
for mm in 12500 25000 100000; do
  for mode in 0 1; do
    BMF_FORCE_SHARDED=$mode timeout -k 10 300 python bench.py --m $mm --steps 30 --warmup 5 --cpu-rows 0 --traffic 0 --secondary 0 --sustained 0 2>$OUT/bench_m${mm}_s${mode}.err | tail -1 > $OUT/bench_m${mm}_s${mode}.json
    python - <<PY
import json
try:
    d = json.load(open("$OUT/bench_m${mm}_s${mode}.json"))
    print("m=$mm sharded=$mode: %.4f ms/step, %.1f it/s, gemm %.1f us" % (d["ms_per_step"], d["value"], 1e3 * d["roofline"]["avg_launch_ms"]), d.get("distributed", {}).get("plan"), {k: v for k, v in d.get("distributed", {}).items() if "ms" in k})
except Exception as e:
    print("m=$mm sharded=$mode FAILED", e)
PY
  done
done
(cd /tmp && rocprofv3 -L > $GRAFT_REPO_ROOT/$OUT/counters.txt 2>&1)
grep -c "" $OUT/counters.txt
PM="python3 scripts/gemm_i8_microbench.py 12"
cd /tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc/a -- python3 $GRAFT_REPO_ROOT/scripts/gemm_i8_microbench.py 12 > $GRAFT_REPO_ROOT/$OUT/pmc_a.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_BUSY_CU_CYCLES SQ_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc/b -- python3 $GRAFT_REPO_ROOT/scripts/gemm_i8_microbench.py 12 > $GRAFT_REPO_ROOT/$OUT/pmc_b.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc/c -- python3 $GRAFT_REPO_ROOT/scripts/gemm_i8_microbench.py 12 > $GRAFT_REPO_ROOT/$OUT/pmc_c.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_I8 SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_WAVES SQ_IFETCH SQ_WAIT_IFETCH GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc/d -- python3 $GRAFT_REPO_ROOT/scripts/gemm_i8_microbench.py 12 > $GRAFT_REPO_ROOT/$OUT/pmc_d.log 2>&1
cd $GRAFT_REPO_ROOT
python3 scripts/pmc_summary.py $OUT/pmc | grep "xf_bits\|kernel |" > $OUT/pmc_summary.md
cat $OUT/pmc_summary.md
timeout -k 10 400 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc $?"
python - <<PY
import json
d = json.load(open("$OUT/bench_default.json"))
print({k: d[k] for k in ("value", "ms_per_step")}, d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], d.get("sustained"), d.get("with_mae"), d["secondary"]["c2_wnmf_real"]["iterations_per_s"])
PY
