export TMPDIR=/tmp; O=gpurun_out/r5l; mkdir -p $O
timeout -k 10 300 python3 -m pytest tests/test_msm_gpu.py -x -q -m gpu -k "host or chunk or any_curve or full_size" > $O/t.log 2>&1; tail -2 $O/t.log
python3 tools/bench_sync_call.py --variants "X=1;CURDLE_HOST_GRADED=0;CURDLE_HOST_CHUNKS=4;CURDLE_HOST_CHUNKS=6;CURDLE_HOST_CHUNKS=7;X=2;CURDLE_HOST_GRADED=0" 20 > $O/sync.jsonl 2>$O/sync.err
python3 tools/bench_sync_call.py --variants "X=1;CURDLE_HOST_GRADED=0;CURDLE_HOST_CHUNKS=4" 19 > $O/sync19.jsonl 2>>$O/sync.err
cat $O/sync.jsonl $O/sync19.jsonl
