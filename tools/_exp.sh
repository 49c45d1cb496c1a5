export TMPDIR=/tmp; export CURDLE_BENCH_HOST_REPS=24; O=gpurun_out/r5_hostconv; mkdir -p $O
timeout -k 10 500 python -m pytest tests/test_msm_gpu.py -x -q -m gpu > $O/tests.log 2>&1 || { tail -20 $O/tests.log; exit 1; }
tail -1 $O/tests.log
python3 tools/bench_sync_call.py --variants "X=1;X=2;X=3" 20 19 > $O/s.jsonl 2>$O/err
python3 - <<'E'
import json
for l in open('gpurun_out/r5_hostconv/s.jsonl'):
    r=json.loads(l); print(r['n'], r['variant'], r['median_ms'], r['host_buffers_ms'])
E
FUZZ_SIZES=524288,600000,1048576,700001 timeout -k 10 200 python3 tools/fuzz_msm.py 60 91 > $O/fuzz.txt 2>&1; tail -1 $O/fuzz.txt
