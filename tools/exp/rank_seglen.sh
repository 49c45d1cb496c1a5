#!/bin/bash
# One rank of the 8-way window split (kept bases, 6 in flight) under CURDLE_SEG_LEN = positions per accumulate lane (default there: 16 = one round of 131,072 lanes).
for rep in 1 2; do for L in 0 20 24 28 32 40; do
  if [ $L = 0 ]; then unset CURDLE_SEG_LEN; else export CURDLE_SEG_LEN=$L; fi
  timeout -k 10 200 python3 bench.py --emulate-world 8 --steps 120 --warmup 10 --no-cpu-baseline --no-verify 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('L=$L', round(d['ms_per_step_rank0'],4), round(d['single_call_ms'],4), d['kernel_ms_alone'].get('accumulate'), d['kernel_ms_alone'].get('bucket_reduce'))"
done; done
