#!/bin/bash
# A/B of two builds on ONE lease, synchronous calls: the product build against build_alt/libcurdlemsm_alt.so (CURDLE_MSM_LIB), alternating.
# Usage: tools/exp/ab_sync.sh <out-file> [rounds] [sizes...]
O=$1; R=${2:-2}; shift; shift; SIZES=${@:-20 18 17 16 15 14 13 n=4096 n=1268}
for i in $(seq 1 $R); do for v in base alt; do
  if [ $v = alt ]; then export CURDLE_MSM_LIB=$PWD/build_alt/libcurdlemsm_alt.so; else unset CURDLE_MSM_LIB; fi
  python3 tools/bench_sync_call.py --variants "DEFAULTS=1" $SIZES 2>/dev/null | python3 -c "
import json,sys
print('$i $v', ' '.join('%s:%.4f/%.4f' % (json.loads(l)['n'], json.loads(l)['median_ms'], json.loads(l)['host_buffers_ms'] or 0) for l in sys.stdin if l.startswith('{')))" >> $O
done; done
unset CURDLE_MSM_LIB; cat $O
