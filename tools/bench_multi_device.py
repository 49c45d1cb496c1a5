#!/usr/bin/env python3
"""ONE process driving several GPUs through the C ABI (curdle_init_devices): BASELINE configs 4 and
5 the way a Go host would reach them.  Prints one JSON object:
  * N = 2^logn MSM over inputs resident on every device: curdle_msm_g1_replicated by Pippenger
    windows and by point ranges (one host thread per device, partials summed on the host), next to
    the same MSM on device 0 alone -- results compared bit for bit;
  * the same MSM from HOST slices through curdle_msm_g1 (split by point ranges over the devices);
  * 1,024 IsValidWhiskShuffleProof verifications in one batch call, sharded over the devices.
    python tools/bench_multi_device.py [--devices 0,1,...] [--logn 20]
Default: every visible device.  `--devices 0,0` puts two contexts on one GPU (a rehearsal of the
code path, not a measurement).  bench.py runs this as a CHILD process, with a timeout, when more
than one GPU is visible to a one-rank run, so that nothing here can take the bench line down."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "go-curdleproofs_amd"))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--devices", default="")
    ap.add_argument("--logn", type=int, default=20)
    ap.add_argument("--proofs", type=int, default=1024)
    args = ap.parse_args()
    import numpy as np
    import torch
    import curdlemsm as cm
    from bench import cm_one_limbs, host_cores, uniform_scalars
    devices = [int(x) for x in args.devices.split(",")] if args.devices else list(range(torch.cuda.device_count()))
    D = len(devices)
    cm.init_devices(devices)
    n = 1 << args.logn
    sc = uniform_scalars(np.random.default_rng(2), n)
    d_pts, d_sc = [], []
    for i, dev in enumerate(devices):
        cm.set_device(i)
        p = torch.empty((n, 12), dtype=torch.int64, device=f"cuda:{dev}")
        cm.synth_points_walk_device(12345, 6789, n, p.data_ptr())
        d_pts.append(p)
        d_sc.append(torch.from_numpy(sc.view(np.int64)).to(f"cuda:{dev}"))
    cm.set_device(0)
    for dev in set(devices):
        torch.cuda.synchronize(dev)

    def timed(fn, reps=9):
        fn()
        fn()
        ts = []
        for _ in range(reps):
            t = time.perf_counter()
            r = fn()
            ts.append((time.perf_counter() - t) * 1e3)
        return r, float(np.median(ts)), float(min(ts))

    ref, one_ms, _ = timed(lambda: cm.msm_g1_device(d_pts[0].data_ptr(), d_sc[0].data_ptr(), n))
    out = {"devices": devices, "n_pairs": n, "one_device_ms": round(one_ms, 4), "results_equal": True}
    pp, sp = [p.data_ptr() for p in d_pts], [s.data_ptr() for s in d_sc]
    for name, split in (("windows", cm.SPLIT_WINDOWS), ("points", cm.SPLIT_POINTS)):
        r, med, best = timed(lambda: cm.msm_g1_replicated(pp, sp, n, split))
        out[f"replicated_{name}_ms"] = round(med, 4)
        out[f"replicated_{name}_speedup"] = round(one_ms / med, 3)
        out["results_equal"] = out["results_equal"] and bool((r == ref).all())
    pts_h = d_pts[0].cpu().numpy().view(np.uint64)
    # no selection on this thread: a thread that called set_device(i >= 0) keeps its host-buffer MSMs on that ONE
    # device, and this figure is "all devices" (review of round 4: it was measured from a thread pinned to device 0)
    cm.set_device(-1)
    spread0 = cm.stat_spread_calls()
    r, med, best = timed(lambda: cm.msm_g1(pts_h, sc), reps=5)
    out["host_slices_spread_over_devices"] = bool(cm.stat_spread_calls() > spread0) or D == 1
    cm.set_device(0)
    out["host_slices_all_devices_ms"] = round(med, 4)
    out["results_equal"] = out["results_equal"] and bool((r == ref).all())
    # config 5: one batch call, sharded over the devices
    ONE = np.array(cm_one_limbs(), dtype=np.uint64)
    compress = lambda aff: cm.g1_compress(np.concatenate([aff, ONE]))
    crs = cm.CRS(cm.WHISK_ELL, cm.Rand(0))
    sets = []
    for j in range(8):
        rr = cm.Rand(10 + j)
        pts = rr.get_g1_affines(2 * cm.WHISK_ELL)
        pre = [compress(pts[2 * i]) + compress(pts[2 * i + 1]) for i in range(cm.WHISK_ELL)]
        post, proof = cm.whisk_generate_shuffle_proof(crs, pre, rr)
        sets.append((pre, post, proof))
    k = args.proofs
    batch = cm.PreparedWhiskBatch([sets[i % 8][0] for i in range(k)], [sets[i % 8][1] for i in range(k)],
                                  [sets[i % 8][2] for i in range(k)])
    threads = min(16 * D, host_cores())
    ok = all(batch.run(crs, cm.Rand(3), nthreads=threads))
    ts = []
    for rep in range(3):
        t = time.perf_counter()
        ok = ok and all(batch.run(crs, cm.Rand(4 + rep), nthreads=threads))
        ts.append((time.perf_counter() - t) * 1e3)
    out["whisk_batch"] = {"proofs": k, "host_threads": threads, "ms_per_batch": [round(t, 2) for t in ts],
                          "proofs_per_s": round(k / min(ts) * 1e3, 1), "all_accepted": bool(ok)}
    cm.set_device(0)
    del d_pts, d_sc
    cm.shutdown()
    print(json.dumps(out), flush=True)
    return 0 if out["results_equal"] and ok else 1


if __name__ == "__main__":
    sys.exit(main())
