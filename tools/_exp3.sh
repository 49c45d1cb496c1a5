export TMPDIR=/tmp; O=gpurun_out/r5j; mkdir -p $O
timeout -k 10 400 python3 -m pytest tests/test_msm_gpu.py tests/test_abi.py tests/test_device_accumulator.py -x -q -m gpu > $O/t.log 2>&1; tail -2 $O/t.log
python3 tools/bench_sync_call.py --variants "X=1;CURDLE_FRONT=0;CURDLE_FRONT=0,CURDLE_DIRECT_RESULTS=0" n=308 n=1268 n=4096 14 > $O/sync.jsonl 2>$O/sync.err
cat $O/sync.jsonl
