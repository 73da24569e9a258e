#!/bin/bash
# Round 5: the artefacts under profiles/ that DESIGN.md and the bench line cite, from the final build (run on the GPU box via gpurun; the
# copies into profiles/ are made afterwards from gpurun_out/r05/).   r05_final_profiles.sh [part ...]   parts: bench stats shards tests
set -o pipefail
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r05; mkdir -p $OUT
export TMPDIR=/tmp
PARTS="${@:-bench stats}"
has() { [[ " $PARTS " == *" $1 "* ]]; }
if has bench; then
  timeout -k 10 560 python bench.py > $OUT/bench_default_a.json 2> $OUT/bench_default_a.err; echo "bench a rc $?"
  timeout -k 10 560 python bench.py > $OUT/bench_default_b.json 2> $OUT/bench_default_b.err; echo "bench b rc $?"
fi
if has stats; then
  bash scripts/r05_kernel_stats.sh 100000 headline
  bash scripts/r05_kernel_stats.sh 12500 m12500
fi
if has shards; then
  for m in 100000 50000 25000 12500; do
    echo "== m = $m unsharded"; timeout -k 10 200 python bench.py --m $m --steps 20 --warmup 5 --cpu-rows 0 --traffic 0 --secondary 0 --sustained 0 --repeat 3 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms_per_step', round(d['ms_per_step'],4), 'repeat ms', [round(1e3/v,4) for v in d['repeat']['legs_of_K_steps']])"
    echo "== m = $m C-side sharded loop, world = 1"; BMF_FORCE_SHARDED=1 timeout -k 10 200 python bench.py --m $m --steps 20 --warmup 5 --cpu-rows 0 --traffic 0 --secondary 0 --sustained 0 --repeat 3 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms_per_step', round(d['ms_per_step'],4), 'repeat ms', [round(1e3/v,4) for v in d['repeat']['legs_of_K_steps']])"
  done
fi
if has tests; then python -m pytest tests -m gpu -q -x > $OUT/gpu_suite.log 2>&1; tail -3 $OUT/gpu_suite.log; fi
