#!/usr/bin/env python3
"""Experiment (round 5): would ONE synchronous mid-size MSM be faster as two window halves in flight?
The bucket reduction of a 2^16-pair call is a chain of dependent additions (0.24 ms of 0.60) that leaves the multipliers idle;
with the windows split in two submits the reduction of the first half runs under the accumulation of the second.  This tool
measures that with the public entry points only: msm_g1_device (one call) against submit(high windows) + submit(low windows) +
wait + wait + g1_sum, on the same resident inputs, results compared.
Usage: python tools/exp_split_call.py [logn ...]   (default 14 15 16 17 18)"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "go-curdleproofs_amd"))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")


def main():
    import numpy as np
    import torch
    import curdlemsm as cm
    from bench import uniform_scalars
    cm.init(0)
    for a in (sys.argv[1:] or ["14", "15", "16", "17", "18"]):
        n = 1 << int(a)
        d_pts = torch.empty((n, 12), dtype=torch.int64, device="cuda:0")
        cm.synth_points_walk_device(12345, 6789, n, d_pts.data_ptr())
        sc = uniform_scalars(np.random.default_rng(2), n)
        d_sc = torch.from_numpy(sc.view(np.int64)).to("cuda:0")
        torch.cuda.synchronize()
        W = cm.num_windows(n)
        flags = cm.MSM_BASES_UNCHANGED
        ref = cm.msm_g1_device(d_pts.data_ptr(), d_sc.data_ptr(), n, flags=flags)

        def one():
            return cm.msm_g1_device(d_pts.data_ptr(), d_sc.data_ptr(), n, flags=flags)

        def split(parts, order):
            cuts = [W * i // parts for i in range(parts + 1)]
            rng = [(cuts[i], cuts[i + 1]) for i in range(parts)]
            if order == "high_first":
                rng = rng[::-1]
            ts = [cm.msm_g1_device_submit(d_pts.data_ptr(), d_sc.data_ptr(), n, 0, b, e, flags=flags) for b, e in rng]
            return cm.g1_sum(np.stack([cm.msm_wait(t) for t in ts]))

        def timed(f, reps=40):
            for _ in range(5):
                r = f()
            lat = []
            for _ in range(reps):
                torch.cuda.synchronize()
                t = time.perf_counter()
                r = f()
                lat.append((time.perf_counter() - t) * 1e3)
            return r, float(np.median(lat)), min(lat)

        line = {"n": n, "W": W}
        r, med, mn = timed(one)
        line["one_call_ms"] = round(med, 4)
        for parts in (2, 3):
            for order in ("high_first", "low_first"):
                r2, med, mn = timed(lambda: split(parts, order))
                line[f"split{parts}_{order}_ms"] = round(med, 4)
                # Jacobian representations differ; compare through another sum with the negated reference is overkill here:
                # g1_sum([x]) normalises nothing either, so compare affine x = X/Z^2 via the library's compress.
                line[f"split{parts}_{order}_equal"] = bool(cm.g1_compress(r2) == cm.g1_compress(ref))
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
