mkdir -p gpurun_out/w8
out=gpurun_out/w8/segsweep3.txt
: > $out
python tools/sweep.py >> $out 2>> gpurun_out/w8/sweep.err || exit 1
python tools/sweep.py 4096,8192,16384,32768,131072,524288 >> $out 2>> gpurun_out/w8/sweep.err || exit 1
python tools/bench_configs.py > gpurun_out/w8/configs.json 2>> gpurun_out/w8/sweep.err || exit 1
python tools/bench_verify.py > gpurun_out/w8/verify.txt 2>> gpurun_out/w8/sweep.err || exit 1
