#!/usr/bin/env python3
"""Experiment (round 6): is k_accumulate's slowdown on a hot bucket a matter of the stride between the
lanes' positions?  Families hot_window (2^20) and all_equal (2^16) under SEG_LEN overrides."""
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
for fam, logn, Ls in (("hot_window", 20, [0, 127, 120, 96, 64]), ("all_equal", 16, [0, 9, 11, 16, 5]), ("uniform", 20, [0, 127]),
                      ("all_equal", 20, [0, 127])):
    n = 1 << logn
    d_pts = torch.empty((n, 12), dtype=torch.int64, device="cuda:0")
    cm.synth_points_walk_device(12345, 6789, n, d_pts.data_ptr())
    uni = uniform_scalars(np.random.default_rng(2), n)
    sc, dead = adv.make_family(fam, n, uni, window_bits=cm.window_bits(n))
    d_sc = torch.from_numpy(np.ascontiguousarray(sc).view(np.int64)).to("cuda:0")
    ref = None
    for L in Ls:
        cm.plan_override("SEG_LEN", L if L else None)
        cm.profile_enable(1)
        ks = {}
        for _ in range(4):
            r = cm.msm_g1_device(d_pts.data_ptr(), d_sc.data_ptr(), n)
            for a, b in cm.profile_last()["kernels"].items():
                ks.setdefault(a, []).append(b)
        cm.profile_enable(0)
        if ref is None:
            ref = r
        print(fam, logn, "L=", L or "default", "same" if (r == ref).all() else "DIFFERENT",
              {a: round(float(np.mean(b[1:])), 4) for a, b in ks.items() if a in ("accumulate", "merge_large", "bucket_reduce")}, flush=True)
    cm.plan_override("SEG_LEN", None)
