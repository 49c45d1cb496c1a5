#!/bin/bash
# Regenerates the evidence under profiles/ on a GPU box (run through gpurun from the repo root):
#   tests, the default bench line, rocprofv3 kernel stats of the same command, the two PMC
#   passes (separate runs, counters only), the size sweep, BASELINE's configs, end-to-end Verify.
# Everything is written under gpurun_out/; copy what should be judged into profiles/.
set -eo pipefail
export TMPDIR=/tmp
O=$PWD/gpurun_out
mkdir -p $O
timeout -k 10 300 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1
timeout -k 10 400 python bench.py > $O/bench_default.log 2>&1
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats -o stats -- python3 bench.py > $O/prof_stats.log 2>&1
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/prof_fetch -o fetch -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/prof_fetch.log 2>&1
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/prof_write -o write -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/prof_write.log 2>&1
python tools/pmc_summary.py $(find $O/prof_fetch -name '*counter_collection.csv' | head -1) $(find $O/prof_write -name '*counter_collection.csv' | head -1) $O/pmc_traffic.json > $O/pmc_summary.log 2>&1
cp $(find $O/prof_stats -name '*kernel_stats.csv' | head -1) $O/kernel_stats.csv
timeout -k 10 400 python tools/sweep.py > $O/sweep.log 2>&1
timeout -k 10 400 python tools/bench_configs.py 2> $O/configs.err > $O/configs.json
timeout -k 10 300 python tools/bench_verify.py 2> $O/verify.err > $O/verify.log
tail -1 $O/pytest_gpu.log
tail -1 $O/bench_default.log
