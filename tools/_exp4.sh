export TMPDIR=/tmp; O=gpurun_out/r5k; mkdir -p $O
python3 bench.py --mode verify --steps 200 --warmup 20 > $O/verify_line.json 2>/dev/null; cut -c1-600 $O/verify_line.json
CURDLE_VERIFY_TRACE=1 python3 tools/bench_verify.py 252 > $O/verify.log 2> $O/verify_trace.txt; grep "^\[verify\]" $O/verify_trace.txt | tail -8; cut -c1-400 $O/verify.log
