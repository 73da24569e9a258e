#!/bin/bash
# What one rank does at the shard sizes of 1 / 2 / 4 / 8 GPUs, on ONE GPU: bench.py unsharded and with the C-side sharded loop forced
# (RCCL with one rank).  Every timed leg of a run is printed, so that a one-off host cost in one of them shows.  r04_shard_sizes.sh [sizes]
for mm in ${1:-100000 50000 25000 12500}; do
  for mode in 0 1; do
    BMF_FORCE_SHARDED=$mode timeout -k 10 200 python bench.py --m $mm --steps 30 --warmup 5 --cpu-rows 0 --traffic 0 --secondary 0 --sustained 0 --repeat 3 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); dd=d.get('distributed',{}) or {}
legs=[1e3/x for x in d.get('repeat',{}).get('legs_of_K_steps',[])]
print('m', $mm, 'forced_sharded', $mode, ': value leg %.4f ms/step;' % d['ms_per_step'], 'repeat legs', ' '.join('%.4f' % x for x in legs), '; cold %.4f' % (1e3/d['cold_start']['value']) if isinstance(d.get('cold_start'),dict) else '', '; gemm launch %.4f ms' % d['roofline']['avg_launch_ms'], '; comm leg', dd.get('ms_per_step_in_this_leg', dd.get('comm_leg_ms_per_step')), 'exposed', dd.get('exposed_comm_ms_per_step'))"
  done
done
