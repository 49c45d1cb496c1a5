mkdir -p gpurun_out/w8
python tools/bench_verify.py 252 > gpurun_out/w8/verify.txt 2>> gpurun_out/w8/sweep.err || exit 1
python tools/bench_whisk.py > gpurun_out/w8/whisk.txt 2>> gpurun_out/w8/sweep.err || exit 1
python bench.py --emulate-world 8 --split points --steps 60 --warmup 6 2>> gpurun_out/w8/sweep.err | cut -c1-200 > gpurun_out/w8/pts.txt || exit 1
