#!/bin/bash
# Regenerates the evidence under profiles/ on a GPU box (run through gpurun from the repo root):
#   tests, the default bench line, rocprofv3 kernel stats of the same command, the two PMC
#   passes (separate runs, counters only), the FETCH_SIZE calibration, the size sweep,
#   BASELINE's configs, end-to-end Verify, the multi-GPU emulation table.
# Everything is written under gpurun_out/refresh/; copy what should be judged into profiles/
# (tools/refresh_profiles.sh does NOT touch profiles/ itself).
# Two parts (a gpurun call is at most 20 minutes): `tools/refresh_profiles.sh a` = tests, the bench line,
# rocprofv3 kernel stats, the PMC passes; `... b` = sweeps, configs, the layers around the MSM.  No
# argument: both.
set -eo pipefail
PART=${1:-ab}
export TMPDIR=/tmp
# bench.py raises the hardware-queue count itself, but under rocprofv3 the profiler's preload
# initialises HIP before Python starts: export it here so profiled runs are the benchmarked
# configuration.
export GPU_MAX_HW_QUEUES=16
R=$PWD
O=$PWD/gpurun_out/refresh
mkdir -p $O
cd /tmp
if [[ $PART == *a* ]]; then
timeout -k 10 600 python3 -m pytest $R/tests -x -q -m gpu > $O/pytest_gpu.log 2>&1
timeout -k 10 400 python3 $R/bench.py > $O/bench_default.json 2> $O/bench_default.err
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats -o stats -- python3 $R/bench.py --no-cpu-baseline --no-verify > $O/prof_stats.log 2>&1
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/prof_fetch -o fetch -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-verify > $O/prof_fetch.log 2>&1
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/prof_write -o write -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-verify > $O/prof_write.log 2>&1
python3 $R/tools/pmc_summary.py $(find $O/prof_fetch -name '*counter_collection.csv' | head -1) $(find $O/prof_write -name '*counter_collection.csv' | head -1) $O/pmc_traffic.json > $O/pmc_summary.log 2>&1
cp $(find $O/prof_stats -name '*kernel_stats.csv' | head -1) $O/kernel_stats.csv
# the same command with ONE MSM in flight: launches never overlap, so k_accumulate's AVERAGE duration in this file is the
# isolated duration the line's roofline.kernel_ms is measured as (with four in flight the profiler's durations are spans of
# two overlapping launches: VERDICT r4 had to reach for the minimum)
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats1 -o stats1 -- python3 $R/bench.py --in-flight 1 --no-cpu-baseline --no-verify > $O/prof_stats1.log 2>&1
cp $(find $O/prof_stats1 -name '*kernel_stats.csv' | head -1) $O/kernel_stats_in_flight_1.csv
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/gather_fetch -o gather -- $R/tools/ubench_gather > $O/gather_fetch.log 2>&1
tail -1 $O/pytest_gpu.log
cut -c1-400 $O/bench_default.json
fi
if [[ $PART == *b* ]]; then
timeout -k 10 400 python3 $R/tools/sweep.py > $O/sweep.log 2>/dev/null
timeout -k 10 400 python3 $R/tools/sweep_adversarial.py --out gpurun_out/refresh/adversarial.json > /dev/null 2> $O/adversarial_rows.jsonl
timeout -k 10 400 python3 $R/bench.py --sweep > $O/sweep.json 2> $O/sweep_rows.jsonl
timeout -k 10 300 python3 $R/bench.py --mode whisk-batch --steps 5 --warmup 1 > $O/whisk_batch.json 2> /dev/null
timeout -k 10 300 python3 $R/bench.py --mode verify --steps 200 --warmup 20 > $O/verify_line.json 2> /dev/null
rm -f $O/multi_gpu_emulation.jsonl
# (round 6: a window-split rank keeps its converted bases by DEFAULT -- what bench.py --gpus N runs; --convert-per-call is the gnark-layout-per-call figure)
for w in 2 4 8; do timeout -k 10 200 python3 $R/bench.py --emulate-world $w --steps 60 --warmup 5 --no-cpu-baseline --no-verify 2>/dev/null | tail -1 >> $O/multi_gpu_emulation.jsonl; done
for w in 2 4 8; do timeout -k 10 200 python3 $R/bench.py --emulate-world $w --convert-per-call --steps 60 --warmup 5 --no-cpu-baseline --no-verify 2>/dev/null | tail -1 >> $O/multi_gpu_emulation.jsonl; done
for w in 2 4 8; do timeout -k 10 200 python3 $R/bench.py --emulate-world $w --resident-bases --steps 60 --warmup 5 --no-cpu-baseline --no-verify 2>/dev/null | tail -1 >> $O/multi_gpu_emulation.jsonl; done
timeout -k 10 200 python3 $R/bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-verify 2>/dev/null | tail -1 >> $O/multi_gpu_emulation.jsonl
for lg in 22 24; do timeout -k 10 300 python3 $R/bench.py --logn $lg --steps 10 --warmup 2 --no-cpu-baseline --no-verify 2>/dev/null | tail -1 > $O/bench_2p$lg.json; done
timeout -k 10 400 python3 $R/tools/bench_sync_call.py --variants "DEFAULTS=1" 20 19 18 17 16 14 n=4096 n=1268 n=308 > $O/sync_call.jsonl 2> /dev/null
timeout -k 10 120 python3 $R/tools/bench_h2d.py > $O/h2d.txt 2>&1
timeout -k 10 120 $R/tools/ubench_affine > $O/batched_affine.txt 2>&1
timeout -k 10 400 python3 $R/tools/bench_configs.py 2> $O/configs.err > $O/configs.json
timeout -k 10 300 python3 $R/tools/bench_verify.py 2> $O/verify.err > $O/verify.log
timeout -k 10 300 python3 $R/tools/bench_whisk.py 2> $O/whisk.err > $O/whisk.log
timeout -k 10 200 python3 $R/tools/bench_decode.py 2> /dev/null > $O/decode.log
timeout -k 10 200 python3 $R/tools/bench_scalar_mul_batch.py 2> /dev/null > $O/scalar_mul.log
fi
