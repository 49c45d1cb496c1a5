#!/bin/bash
# Experiment builds of the library, never the product: build_alt/libcurdlemsm_alt.so = the product's objects with the
# named translation units recompiled under extra flags.  Usage: tools/exp/build_alt.sh "<flags>" unit [unit ...]
#   tools/exp/build_alt.sh -DCURDLE_EXP_SKIP msm_enqueue       (the phase-skip experiment, profiles/r06_pipeline_phase_costs.txt)
set -e
cd "$(dirname "$0")/../../go-curdleproofs_amd"
make -j8 >/dev/null
FLAGS=$1; shift
mkdir -p ../build_alt/obj
OBJS=$(ls build/*.o)
for u in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unknown-pragmas --offload-arch=gfx950 -munsafe-fp-atomics $FLAGS -c csrc/$u.hip -o ../build_alt/obj/$u.o
  OBJS=$(echo "$OBJS" | grep -v "build/$u.o"); OBJS="$OBJS ../build_alt/obj/$u.o"
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -o ../build_alt/libcurdlemsm_alt.so $OBJS
ls -la ../build_alt/libcurdlemsm_alt.so
