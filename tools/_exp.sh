export TMPDIR=/tmp; export CURDLE_BENCH_HOST_REPS=24; O=gpurun_out/r5_copiers; mkdir -p $O
python3 tools/bench_sync_call.py --variants "X=1;CURDLE_HOST_COPIERS=2;X=2;CURDLE_HOST_COPIERS=2;X=3;CURDLE_HOST_COPIERS=2" 20 19 > $O/s.jsonl 2>$O/err
python3 - <<'E'
import json
for l in open('gpurun_out/r5_copiers/s.jsonl'):
    r=json.loads(l); print(r['n'], r['variant'], r['median_ms'], r['host_buffers_ms'])
E
