export TMPDIR=/tmp; O=gpurun_out/r5h; mkdir -p $O
timeout -k 10 300 python3 -m pytest tests/test_msm_gpu.py tests/test_abi.py -x -q -m gpu > $O/t.log 2>&1; tail -2 $O/t.log
for v in "X=1" "CURDLE_DIRECT_RESULTS=0"; do
  echo "== $v" >> $O/sweep.log
  env $v python3 tools/sweep.py 308,1268,4096,65536,1048576 >> $O/sweep.log 2>&1
done
python3 tools/bench_sync_call.py --variants "X=1;CURDLE_DIRECT_RESULTS=0" 16 17 20 > $O/sync.jsonl 2>$O/sync.err
cut -c1-120 $O/sweep.log; cat $O/sync.jsonl
