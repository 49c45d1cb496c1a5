export TMPDIR=/tmp; O=gpurun_out/r5o; mkdir -p $O
python3 tools/bench_sync_call.py --variants "X=1;CURDLE_HOST_GRADED=2;X=2;CURDLE_HOST_GRADED=2;CURDLE_HOST_GRADED=2,CURDLE_HOST_CHUNKS=3;CURDLE_HOST_CHUNKS=3;CURDLE_HOST_GRADED=1,CURDLE_HOST_CHUNKS=3" 20 > $O/sync.jsonl 2>$O/sync.err
cat $O/sync.jsonl
