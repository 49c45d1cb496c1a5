export TMPDIR=/tmp; O=gpurun_out/r5n; mkdir -p $O
timeout -k 10 300 python3 -m pytest tests/test_msm_gpu.py -x -q -m gpu -k "host or chunk or any_curve or full_size or scans or two_pass" > $O/t.log 2>&1; tail -2 $O/t.log
python3 tools/bench_sync_call.py --variants "X=1;CURDLE_HOST_GRADED=0;X=2;CURDLE_HOST_GRADED=0;CURDLE_HOST_CHUNKS=4;CURDLE_HOST_CHUNKS=6" 20 > $O/sync.jsonl 2>$O/sync.err
cat $O/sync.jsonl
source tools/_emu_knobs.sh r5n
for m in "--bases-unchanged" ""; do
run default "$m" X=1
run scan2 "$m" CURDLE_SCAN=2
run L32 "$m" CURDLE_SEG_LEN=32
done
cut -c1-330 $O/knobs.txt
