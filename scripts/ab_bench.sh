#!/bin/bash
# A/B several library flavours in one box: usage ab_bench.sh lib1.so lib2.so ...  (two rounds each, interleaved)
for round in 1 2; do for v in "$@"; do BMF_NO_CHECK=${BMF_NO_CHECK:-0} BMF_LIB=$v timeout -k 10 300 python bench.py --steps 30 --warmup 5 --cpu-rows 0 --alt-operands none 2>&1 | grep metric | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$v', round(d['value'],1), 'ms/step', round(d['ms_per_step'],4), 'gemm ms', round(d['roofline']['avg_launch_ms'],4), 'TF', round(d['roofline']['achieved'],1))"; done; done
