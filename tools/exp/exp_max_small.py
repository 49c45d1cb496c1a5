#!/usr/bin/env python3
"""Experiment (round 6): the fragment count beyond which a bucket goes through k_merge_large (MsmPlan::max_small),
on uniform inputs over the size table and on the adversarial families at 2^12 / 2^16.  Ran against a build that had an
experiment knob MAX_SMALL (profiles/r06_max_small.txt); the knob was removed with the decision (16; 8 up to 4,096 pairs),
so this script is the record of how the table was made, not something the product build can replay."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("go-curdleproofs_amd", "oracle/py", "", "tools"):
    sys.path.insert(0, os.path.join(ROOT, p))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import torch
import curdlemsm as cm
import adversarial_inputs as adv
from bench import uniform_scalars
cm.init(0)
nmax = 1 << 20
d_pts = torch.empty((nmax, 12), dtype=torch.int64, device="cuda:0")
cm.synth_points_walk_device(12345, 6789, nmax, d_pts.data_ptr())
uni = uniform_scalars(np.random.default_rng(2), nmax)

def timed(pp, sp, n, reps):
    for _ in range(3):
        r = cm.msm_g1_device(pp, sp, n)
    lat = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t = time.perf_counter()
        cm.msm_g1_device(pp, sp, n)
        lat.append((time.perf_counter() - t) * 1e3)
    return r, float(np.median(lat))

cases = [("uniform", n) for n in (1268, 4096, 8192, 16384, 32768, 65536, 1 << 18, 1 << 20)] + \
        [(f, 4096) for f in ("all_equal", "distinct_64", "hot_window", "half_equal")] + \
        [(f, 65536) for f in ("all_equal", "distinct_64", "hot_window", "half_equal")]
for fam, n in cases:
    sc, dead = adv.make_family(fam, n, uni, window_bits=cm.window_bits(n))
    d_sc = torch.from_numpy(np.ascontiguousarray(sc).view(np.int64)).to("cuda:0")
    row, ref = {}, None
    for rep in range(2):
        for ms in (0, 12, 8, 6, 4):
            cm.plan_override("MAX_SMALL", ms if ms else None)
            r, t = timed(d_pts.data_ptr(), d_sc.data_ptr(), n, 25 if n <= 65536 else 9)
            ref = r if ref is None else ref
            assert (r == ref).all(), (fam, n, ms)
            row.setdefault(ms or 16, []).append(round(t, 4))
    cm.plan_override("MAX_SMALL", None)
    print(fam, n, row, flush=True)
