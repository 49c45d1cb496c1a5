#!/usr/bin/env python3
"""How busy the GPU is during the host-bound batch verifiers, from a kernel trace.

    python tools/bench_batch_busy.py run whisk|verify [steps]          # the runner (put it behind rocprofv3 --kernel-trace)
    python tools/bench_batch_busy.py parse <kernel_trace.csv> [...]     # union of the kernels' intervals between the markers

The runner prepares BASELINE config 5 (1,024 IsValidWhiskShuffleProof verifications, 16 host threads) or the
1,024-proof curdle_verify_batch, warms up, launches a MARKER kernel (k_synth_walk over one point), runs `steps`
identical honest steps, launches the marker again and prints the steps' wall time.  The parser finds the last two
markers in rocprofv3's kernel trace and reports, for the kernels between them: the union of their [start, end]
intervals (the time the GPU had at least one kernel running), the sum of their durations, the span between the
markers, and the busiest kernel names.  HIP events cannot give this: a bracket around a 10 us kernel of a call
whose launches the host issues one by one includes the host's launch gaps (summed brackets of one step came to
1.5x the step's wall time)."""
import csv
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(mode, steps):
    sys.path.insert(0, os.path.join(ROOT, "go-curdleproofs_amd"))
    sys.path.insert(0, ROOT)
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
    import numpy as np
    import torch
    import curdlemsm as cm
    from bench import cm_one_limbs, host_cores
    cm.init(0)
    threads = int(os.environ.get("BUSY_THREADS", "0")) or min(16, host_cores())
    marker = torch.empty((1, 12), dtype=torch.int64, device="cuda:0")
    if mode == "whisk":
        ONE = np.array(cm_one_limbs(), dtype=np.uint64)
        compress = lambda aff: cm.g1_compress(np.concatenate([aff, ONE]))
        crs = cm.CRS(cm.WHISK_ELL, cm.Rand(0))
        sets = []
        for j in range(8):
            r = cm.Rand(10 + j)
            pts = r.get_g1_affines(2 * cm.WHISK_ELL)
            pre = [compress(pts[2 * i]) + compress(pts[2 * i + 1]) for i in range(cm.WHISK_ELL)]
            post, proof = cm.whisk_generate_shuffle_proof(crs, pre, r)
            sets.append((pre, post, proof))
        k = 1024
        batch = cm.PreparedWhiskBatch([sets[i % 8][0] for i in range(k)], [sets[i % 8][1] for i in range(k)],
                                      [sets[i % 8][2] for i in range(k)])
    else:
        ell = 252
        rand = cm.Rand(0)
        crs = cm.CRS(ell, rand)
        distinct = []
        for j in range(8):
            pj = cm.Rand(50 + j).generate_permutation(ell)
            kj = rand.get_fr()
            Rj, Sj = rand.get_g1_affines(ell), rand.get_g1_affines(ell)
            Tj, Uj, Mj, rsj = cm.shuffle_permute_commit(crs, Rj, Sj, pj, kj, rand)
            distinct.append((cm.prove(crs, Rj, Sj, Tj, Uj, Mj, pj, kj, rsj, cm.Rand(60 + j)), Rj, Sj, Tj, Uj, Mj))
        k = 1024
        batch = cm.PreparedVerifyBatch(*[[distinct[i % 8][c] for i in range(k)] for c in range(6)])
    for w in range(2):
        assert all(batch.run(crs, cm.Rand(7 + w), nthreads=threads))
    torch.cuda.synchronize()
    cm.synth_points_walk_device(3, 5, 1, marker.data_ptr())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(steps):
        assert all(batch.run(crs, cm.Rand(100 + s), nthreads=threads))
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    cm.synth_points_walk_device(3, 5, 1, marker.data_ptr())
    torch.cuda.synchronize()
    print(json.dumps({"mode": mode, "proofs_per_step": k, "steps": steps, "host_threads": threads,
                      "ms_per_step": round(wall * 1e3 / steps, 3), "proofs_per_s": round(k * steps / wall, 1)}))


def parse(path):
    rows = list(csv.DictReader(open(path)))
    name_k = next(c for c in rows[0] if c.lower() in ("kernel_name", "name"))
    s_k = next(c for c in rows[0] if c.lower().startswith("start"))
    e_k = next(c for c in rows[0] if c.lower().startswith("end"))
    ev = sorted(((int(r[s_k]), int(r[e_k]), r[name_k]) for r in rows), key=lambda x: x[0])
    marks = [i for i, x in enumerate(ev) if "k_synth_walk" in x[2]]
    if len(marks) < 2:
        raise SystemExit("no two marker kernels in the trace")
    a, b = marks[-2], marks[-1]
    lo, hi = ev[a][1], ev[b][0]
    inside = [x for x in ev[a + 1:b] if x[0] >= lo and x[1] <= hi]
    union, cur_s, cur_e = 0, None, None
    for s, e, _ in inside:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                union += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    if cur_e is not None:
        union += cur_e - cur_s
    by = {}
    for s, e, n in inside:
        short = n.split("(")[0].split("<")[0].replace("void curdle::", "").replace("curdle::", "")
        by[short] = by.get(short, 0) + (e - s)
    top = sorted(by.items(), key=lambda kv: -kv[1])[:8]
    return {"span_ms": round((hi - lo) / 1e6, 3), "union_busy_ms": round(union / 1e6, 3),
            "sum_of_durations_ms": round(sum(e - s for s, e, _ in inside) / 1e6, 3), "kernels": len(inside),
            "gpu_busy_frac_under_profiler": round(union / (hi - lo), 4),
            "top_kernels_ms": {k: round(v / 1e6, 3) for k, v in top}}


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 5)
    else:
        for p in sys.argv[2:]:
            print(json.dumps(dict(parse(p), trace=os.path.basename(p))))
