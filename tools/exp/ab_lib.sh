#!/bin/bash
# A/B of two builds of the library on ONE lease: the product build against build_alt/libcurdlemsm_alt.so (CURDLE_MSM_LIB),
# alternating, the pipelined headline and an 8-way rank step.  Usage: tools/exp/ab_lib.sh <out-file> [rounds]
O=$1; R=${2:-3}
for i in $(seq 1 $R); do
  for v in base alt; do
    if [ $v = alt ]; then export CURDLE_MSM_LIB=$PWD/build_alt/libcurdlemsm_alt.so; else unset CURDLE_MSM_LIB; fi
    a=$(timeout -k 10 200 python3 bench.py --steps 80 --warmup 10 --no-cpu-baseline --no-verify 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), d['config']['single_call_ms'])")
    b=$(timeout -k 10 200 python3 bench.py --emulate-world 8 --steps 80 --warmup 10 --no-cpu-baseline --no-verify 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step_rank0'],4), round(d['single_call_ms'],4))")
    echo "$i $v whole: $a | rank8: $b" >> $O
  done
done
unset CURDLE_MSM_LIB
cat $O
