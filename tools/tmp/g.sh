for cfg in CURDLE_REDUCE_SEG=1 CURDLE_REDUCE_SEG=2 "CURDLE_REDUCE_SEG=2 CURDLE_SEG_LEN=32" "CURDLE_REDUCE_SEG=1 CURDLE_SEG_LEN=32" CURDLE_REDUCE_SEG=8; do
echo "$cfg: $(env $cfg python bench.py --emulate-world 8 --steps 150 --warmup 10 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print(round(d['ms_per_step_rank0'],4), round(d['single_call_ms'],3), {k: v for k, v in d['kernel_ms_alone'].items() if k in ('accumulate','bucket_reduce','window_sum')})")"
done
