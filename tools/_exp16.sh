export TMPDIR=/tmp; O=gpurun_out/r5F; mkdir -p $O
python3 tools/bench_sync_call.py --variants "X=1;CURDLE_SEG_LEN=2;CURDLE_SEG_LEN=3;CURDLE_SEG_LEN=6;CURDLE_WINDOW_BITS=9;CURDLE_WINDOW_BITS=11;CURDLE_WINDOW_BITS=8;CURDLE_REDUCE_SEG=2;CURDLE_SEG_LEN=2,CURDLE_WINDOW_BITS=9;X=2" n=1268 n=2548 n=628 > $O/s.jsonl 2>$O/err
python3 - <<'E'
import json
for l in open('gpurun_out/r5F/s.jsonl'):
    r=json.loads(l); print(r['n'], r['variant'], r['median_ms'], r['min_ms'], r['host_buffers_ms'])
E
