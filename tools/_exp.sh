export TMPDIR=/tmp
timeout -k 10 500 python -m pytest tests/test_msm_gpu.py -x -q -m gpu > gpurun_out/r5_prefetch_tests.log 2>&1 || { tail -20 gpurun_out/r5_prefetch_tests.log; exit 1; }
tail -1 gpurun_out/r5_prefetch_tests.log
python3 tools/sweep.py 308,1268,4096,16384,65536,131072,262144,1048576 2>/dev/null | grep "^n="
python3 tools/bench_sync_call.py --variants "X=1" 20 18 16 14 n=4096 n=1268 2>/dev/null
