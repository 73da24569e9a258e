#!/bin/bash
# Round 5: static wave priority for the later-dispatched workgroups of the dense bits GEMM (CDNA guide: "static priority for the
# younger half"), with the slice share re-swept around it.  build: flavours of xf_bits_i8.hip; run: the headline loop with each.
# (Since the adoption of priority 1 for the big-slice workgroups at share 0.66, `base` is that; b0s63 is the form before.)
SPECS="${SPECS:-b0s63:-DBMF_EXP_PRIO_BIG=0,-DBMF_EXP_SHARE=63 b0s66:-DBMF_EXP_PRIO_BIG=0 b1s63:-DBMF_EXP_SHARE=63 p1s63:-DBMF_EXP_PRIO_BIG=0,-DBMF_EXP_PRIO_SMALL=1,-DBMF_EXP_SHARE=63 b2s66:-DBMF_EXP_PRIO_BIG=2}"
if [ "$1" = build ]; then
  for sp in $SPECS; do
    name=${sp%%:*}; flags=$(echo ${sp##*:} | tr ',' ' ')
    bash scripts/build_flavour.sh prio_$name "$flags" xf_bits_i8.hip
  done
else
  mkdir -p gpurun_out/r05
  for name in base $(for sp in $SPECS; do echo ${sp%%:*}; done) base; do
    lib=libbmf_prio_$name.so; [ "$name" = base ] && lib=libbmf_hip.so
    BMF_LIB=$lib timeout -k 10 200 python bench.py --steps 30 --warmup 5 --cpu-rows 0 --traffic 0 --secondary 0 --sustained 3000 --repeat 3 2>/dev/null \
      | python3 -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$name', 'value', round(d['value'],1), 'repeat', [round(v,1) for v in d['repeat']['legs_of_K_steps']], 'sustained', round(d.get('sustained',{}).get('value',0),1), 'gemm ms', d['roofline'].get('avg_launch_ms'))"
  done
fi
