mkdir -p gpurun_out/g
python tools/sweep.py 8,128,308,628,1268,2548,4096,8192,16384,32768,65536,131072,262144,524288,1048576 > gpurun_out/g/sweep.txt 2>/dev/null; cut -c1-75 gpurun_out/g/sweep.txt
python tools/bench_verify.py 252 2>/dev/null | head -1 | cut -c1-300
python tools/bench_configs.py 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print({k: round(v['ms'], 3) for k, v in d.items() if 'ms' in v})"
