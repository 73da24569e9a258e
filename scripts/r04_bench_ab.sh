#!/bin/bash
# in-loop A/B of the bits GEMM variants on the real factors: r04_bench_ab.sh "0 1" -> bench.py (headline only) with BMF_I8_WIDE = each, twice, interleaved
for rep in 1 2; do for w in $1; do
  echo -n "BMF_I8_WIDE=$w "; BMF_I8_WIDE=$w timeout -k 10 300 python bench.py --steps 30 --warmup 5 --cpu-rows 0 --alt-operands none --secondary 0 --traffic 0 --sustained 0 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('value %.1f it/s  ms/step %.4f  gemm launch %.4f ms  frac %.4f  repeat %s' % (d['value'], d['ms_per_step'], r.get('avg_launch_ms',0), r['frac'], d.get('repeat')))"
done; done
