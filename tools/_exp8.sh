export TMPDIR=/tmp; O=gpurun_out/r5q; mkdir -p $O
python3 tools/bench_sync_call.py --variants "X=1;CURDLE_AUX_PRIO=3;X=2;CURDLE_AUX_PRIO=3;CURDLE_AUX_PRIO=1;CURDLE_AUX_PRIO=3,CURDLE_HOST_GRADED=1" 20 > $O/sync.jsonl 2>$O/sync.err
cat $O/sync.jsonl
source tools/_emu_knobs.sh r5q
for rep in 1 2; do
for m in "--bases-unchanged" ""; do
run default "$m" X=1
run aux3 "$m" CURDLE_AUX_PRIO=3
run aux3r0 "$m" CURDLE_AUX_PRIO=3 CURDLE_REDUCE_PRIO=0
done; done
cut -c1-60 $O/knobs.txt
