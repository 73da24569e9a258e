#!/bin/bash
# what one rank does at the shard sizes of 2 / 4 / 8 GPUs: bench.py on ONE GPU, sharded code path forced (no peers), and unsharded
for mm in 100000 50000 25000 12500; do
  for mode in 0 1; do
    BMF_FORCE_SHARDED=$mode timeout -k 10 200 python bench.py --m $mm --steps 30 --warmup 5 --cpu-rows 0 --traffic 0 --secondary 0 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); dd=d.get('distributed',{})
print('m', $mm, 'forced_sharded', $mode, round(d['value'],1), 'it/s', round(d['ms_per_step'],4), 'ms; gemm launch', round(d['roofline']['avg_launch_ms'],4), 'ms x', d['roofline']['launches_timed'], '; exposed comm', dd.get('exposed_comm_ms_per_step'), 'xtu+exchange', dd.get('xtu_and_exchange_ms_per_step'), d['config']['splits_xv'], d['config']['splits_xtu'])"
  done
done
