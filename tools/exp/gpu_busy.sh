#!/bin/bash
# GPU share of the two host-bound batch figures (-> profiles/r06_gpu_busy.json): unprofiled wall, the same runner under
# rocprofv3 --kernel-trace, the union of the kernels' intervals, and how throughput follows the host thread count.
# Run on a GPU box from the repo root (tools/bench_batch_busy.py).
set -eo pipefail
export TMPDIR=/tmp GPU_MAX_HW_QUEUES=16
R=$PWD; O=$PWD/gpurun_out/busy; mkdir -p $O; cd /tmp
for m in whisk verify; do
  python3 $R/tools/bench_batch_busy.py run $m 5 2>/dev/null | tail -1 > $O/${m}_plain.json
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace_$m -o t -- python3 $R/tools/bench_batch_busy.py run $m 5 > $O/${m}_profiled.json 2> $O/${m}_prof.err || true
  python3 $R/tools/bench_batch_busy.py parse $(find $O/trace_$m -name '*kernel_trace.csv' | head -1) > $O/${m}_busy.json
  rm -f $O/${m}_threads.jsonl
  for t in 2 4 8 12 16; do BUSY_THREADS=$t python3 $R/tools/bench_batch_busy.py run $m 3 2>/dev/null | tail -1 >> $O/${m}_threads.jsonl; done
  rm -rf $O/trace_$m
done
python3 - <<PY
import json
out = {"_what": "GPU share of the host-bound batch figures (VERDICT r5 item 4): rocprofv3 --kernel-trace of tools/bench_batch_busy.py, "
                "kernels between two marker launches around 5 identical honest steps of 1,024 proofs; union_busy_ms = time with at least one "
                "kernel running; thread_scaling = the same steps at 2..16 host threads (the lease grants 16 cores).  tools/exp/gpu_busy.sh"}
for m in ("whisk", "verify"):
    rd = lambda f: json.loads(open("$O/%s_%s" % (m, f)).read().strip().splitlines()[-1])
    busy = rd("busy.json")
    out[m] = {"unprofiled": rd("plain.json"), "under_profiler": rd("profiled.json"), "trace": busy,
              "gpu_timeline_coverage": busy["gpu_busy_frac_under_profiler"],
              "overlap_factor": round(busy["sum_of_durations_ms"] / busy["union_busy_ms"], 2),
              "thread_scaling": [json.loads(l) for l in open("$O/%s_threads.jsonl" % m)]}
json.dump(out, open("$O/gpu_busy.json", "w"), indent=1)
print(json.dumps({k: (v["gpu_timeline_coverage"], v["overlap_factor"], [(r["host_threads"], r["proofs_per_s"]) for r in v["thread_scaling"]]) for k, v in out.items() if k != "_what"}))
PY
