for w in 1 2 4; do
if [ $w = 1 ]; then python bench.py --steps 60 --warmup 6 --no-cpu-baseline --no-verify 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('world 1', round(d['ms_per_step'],4), d['config']['single_call_ms'])"
else python bench.py --emulate-world $w --steps 100 --warmup 10 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('world', d['emulated_world'], round(d['ms_per_step_rank0'],4), round(d['single_call_ms'],3))"; fi
done
python tools/sweep.py 262144,524288,1048576 2>/dev/null | cut -c1-62
