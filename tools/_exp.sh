export TMPDIR=/tmp
for a in "" "--emulate-world 8" "--emulate-world 8 --bases-unchanged"; do
python3 bench.py $a --steps 60 --warmup 5 --no-verify --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json
r=json.loads(sys.stdin.read().strip().splitlines()[-1])
if 'emulated_world' in r: print('rank', r['bases_unchanged_flag'], r['ms_per_step_rank0'], r['kernel_ms_alone'].get('accumulate'))
else: print('whole', r['ms_per_step'], r['roofline']['kernel_ms'], r['config']['single_call_ms'])"
done
