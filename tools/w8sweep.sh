mkdir -p gpurun_out/w8
out=gpurun_out/w8/emul2.jsonl
: > $out
python bench.py --logn 24 --emulate-world 8 --steps 30 --warmup 4 2>> gpurun_out/w8/sweep.err >> $out || exit 1
python bench.py --logn 24 --emulate-world 8 --split points --steps 30 --warmup 4 2>> gpurun_out/w8/sweep.err >> $out || exit 1
python bench.py --logn 24 --steps 12 --warmup 2 --no-cpu-baseline --no-verify 2>> gpurun_out/w8/sweep.err >> $out || exit 1
python bench.py --logn 22 --emulate-world 8 --split points --steps 60 --warmup 6 2>> gpurun_out/w8/sweep.err >> $out || exit 1
python bench.py --logn 20 --emulate-world 8 --split points --steps 60 --warmup 6 2>> gpurun_out/w8/sweep.err >> $out || exit 1
