#!/bin/bash
# Round 3: the artefacts under profiles/ that DESIGN.md and the bench line cite, from the final build (run on the GPU box via gpurun).
set -o pipefail
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/final; mkdir -p $OUT
export TMPDIR=/tmp
# 1. the default bench line (what the driver runs), twice
timeout -k 10 500 python bench.py > $OUT/bench_default_a.json 2> $OUT/bench_default_a.err; echo "bench a rc $?"
timeout -k 10 500 python bench.py > $OUT/bench_default_b.json 2> $OUT/bench_default_b.err; echo "bench b rc $?"
# 2. kernel statistics of the headline loop alone and with the MAE pass
cd /tmp
rm -rf $GRAFT_REPO_ROOT/$OUT/prof_headline $GRAFT_REPO_ROOT/$OUT/prof_mae $GRAFT_REPO_ROOT/$OUT/prof_c2 $GRAFT_REPO_ROOT/$OUT/pmc $GRAFT_REPO_ROOT/$OUT/pmc_mae
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_headline -- python3 $GRAFT_REPO_ROOT/bench.py --steps 60 --warmup 5 --cpu-rows 0 --traffic 0 --secondary 0 --sustained 0 > $GRAFT_REPO_ROOT/$OUT/prof_headline.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_mae -- python3 $GRAFT_REPO_ROOT/bench.py --steps 60 --warmup 5 --cpu-rows 0 --traffic 0 --secondary 0 --sustained 0 --mae 1 > $GRAFT_REPO_ROOT/$OUT/prof_mae.log 2>&1
# 3. PMC of the GEMM in the bench loop (real factors), final build
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc/a -- python3 $GRAFT_REPO_ROOT/bench.py --pmc-child --steps 6 --warmup 2 > $GRAFT_REPO_ROOT/$OUT/pmc_a.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc/b -- python3 $GRAFT_REPO_ROOT/bench.py --pmc-child --steps 6 --warmup 2 > $GRAFT_REPO_ROOT/$OUT/pmc_b.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc/c -- python3 $GRAFT_REPO_ROOT/bench.py --pmc-child --steps 6 --warmup 2 > $GRAFT_REPO_ROOT/$OUT/pmc_c.log 2>&1
cd $GRAFT_REPO_ROOT
python3 scripts/pmc_summary.py $OUT/pmc > $OUT/pmc_summary.md
grep "xf_bits\|mu_epilogue\|cover" $OUT/pmc_summary.md
for d in prof_headline prof_mae; do
python - <<PY
import csv, glob
f = glob.glob("$OUT/$d/*/*kernel_stats.csv")[0]
print("== $d")
for row in csv.DictReader(open(f)):
    n = row["Name"]
    if "anonymous" in n and "at::" not in n:
        print("%-62s calls %5s avg %8.1f us min %8.1f max %8.1f" % (n.split("(anonymous namespace)::")[1][:60], row["Calls"], float(row["AverageNs"]) / 1e3, float(row["MinNs"]) / 1e3, float(row["MaxNs"]) / 1e3))
PY
done
# 4. shard sizes: the unsharded C loop against the C-side sharded loop on RCCL with one rank
for mm in 100000 50000 25000 12500; do
  for mode in 0 1; do
    BMF_FORCE_SHARDED=$mode timeout -k 10 300 python bench.py --m $mm --steps 30 --warmup 5 --cpu-rows 0 --traffic 0 --secondary 0 --sustained 0 2>$OUT/shard_m${mm}_s${mode}.err | tail -1 > $OUT/shard_m${mm}_s${mode}.json
    python -c "
import json; d=json.load(open('$OUT/shard_m${mm}_s${mode}.json')); print('m=$mm sharded=$mode: %.4f ms/step %.1f it/s gemm %.1f us' % (d['ms_per_step'], d['value'], 1e3*d['roofline']['avg_launch_ms']), {k: round(v, 4) for k, v in d.get('distributed', {}).items() if 'ms' in k})"
  done
done | tee $OUT/shard_sizes.txt
# 5. config #2 (WNMF on real-valued X, C-side loop): kernel statistics with the residual sums folded into the X^T U pass
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_c2 -- python3 $GRAFT_REPO_ROOT/scripts/c2_loop.py > $GRAFT_REPO_ROOT/$OUT/prof_c2.log 2>&1
# 6. the MAE pass: clock, matrix-pipe busy, instruction counts (microbench, random dense operands)
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc_mae -- python3 $GRAFT_REPO_ROOT/scripts/mae_bench.py > $GRAFT_REPO_ROOT/$OUT/pmc_mae.log 2>&1
cd $GRAFT_REPO_ROOT
tail -2 $OUT/prof_c2.log
python - <<PY
import csv, glob, collections
f = glob.glob("$OUT/prof_c2/*/*kernel_stats.csv")[0]
print("== prof_c2")
for row in csv.DictReader(open(f)):
    n = row["Name"]
    if "anonymous" in n and "at::" not in n:
        print("%-62s calls %5s avg %8.1f us min %8.1f max %8.1f" % (n.split("(anonymous namespace)::")[1][:60], row["Calls"], float(row["AverageNs"]) / 1e3, float(row["MinNs"]) / 1e3, float(row["MaxNs"]) / 1e3))
f = glob.glob("$OUT/pmc_mae/*/*counter_collection.csv")[0]
t = glob.glob("$OUT/pmc_mae/*/*kernel_trace.csv")[0]
dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(t)) if "mae32" in r["Kernel_Name"]]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "mae32" in r["Kernel_Name"]:
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: sum(v) / len(v) for k, v in agg.items()}
us = sum(dur) / len(dur); cyc = m["GRBM_GUI_ACTIVE"] / 8
print("== mae32_kernel (microbench): %.1f us, span %.0f k cycles = %.2f GHz, MFMA busy %.1f %%, VALU instructions (incl. MFMA) %.3e, wave-cycles x4 / span / 3072 slots = %.2f" % (us, cyc / 1e3, cyc / us / 1e3, m["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / cyc * 100, m["SQ_INSTS_VALU"], m["SQ_WAVE_CYCLES"] * 4 / cyc / 3072))
PY
# 6b. the widened loops that were reworked this round: ELBMF's one-call loop and the masked update at MovieLens-1M shape
cd /tmp
rm -rf $GRAFT_REPO_ROOT/$OUT/prof_elbmf $GRAFT_REPO_ROOT/$OUT/prof_masked
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_elbmf -- python3 $GRAFT_REPO_ROOT/scripts/palm_bench.py > $GRAFT_REPO_ROOT/$OUT/prof_elbmf.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_masked -- python3 $GRAFT_REPO_ROOT/scripts/r03/masked_bench.py > $GRAFT_REPO_ROOT/$OUT/prof_masked.log 2>&1
cd $GRAFT_REPO_ROOT
tail -1 $OUT/prof_elbmf.log; tail -1 $OUT/prof_masked.log
# 7. the full-size parity trace
python -m pytest tests/test_c3_parity_gpu.py -m gpu -q -s 2>&1 | grep -E "c3 parity|passed|failed" | tee $OUT/parity_c3.txt
