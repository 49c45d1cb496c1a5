#!/bin/bash
# What each phase costs a PIPELINED caller, priced by leaving it out (experiment build: tools/exp/build_alt.sh -DCURDLE_EXP_SKIP msm_enqueue).
# Run on a GPU box from the repo root; writes the table to $1.
O=$1
export CURDLE_MSM_LIB=$PWD/build_alt/libcurdlemsm_alt.so
V="CURDLE_DEBUG_SKIP=0;CURDLE_DEBUG_SKIP=1;CURDLE_DEBUG_SKIP=3;CURDLE_DEBUG_SKIP=28;CURDLE_DEBUG_SKIP=8;CURDLE_DEBUG_SKIP=16;CURDLE_DEBUG_SKIP=4;CURDLE_DEBUG_SKIP=31;CURDLE_DEBUG_SKIP=0"
echo "# whole MSM, N = 2^20, 4 in flight (bits: 1 conversion, 2 sort, 4 merge_large, 8 reduce_segments, 16 reduce_level)" > $O
python3 tools/bench_pipeline.py --variants "$V" --logn 20 --in-flight 4 --steps 60 >> $O
echo "# one rank of an 8-way window split (window 0, all 2^20 pairs), 6 in flight, bases converted per call" >> $O
python3 tools/bench_pipeline.py --variants "$V" --logn 20 --in-flight 6 --steps 120 --windows 0:1 >> $O
cat $O
