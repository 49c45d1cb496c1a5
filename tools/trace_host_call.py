#!/usr/bin/env python3
"""A few curdle_msm_g1 calls from host buffers, for a rocprofv3 timeline:
   rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d out -- python3 tools/trace_host_call.py 20 4"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "go-curdleproofs_amd"))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import curdlemsm as cm
from bench import uniform_scalars

logn = int(sys.argv[1]) if len(sys.argv) > 1 else 20
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
cm.init(0)
n = 1 << logn
d_pts = torch.empty((n, 12), dtype=torch.int64, device="cuda:0")
cm.synth_points_walk_device(12345, 6789, n, d_pts.data_ptr())
pts = d_pts.cpu().numpy().view(np.uint64).copy()
sc = uniform_scalars(np.random.default_rng(2), n)
torch.cuda.synchronize()
for _ in range(reps):
    t = time.perf_counter()
    cm.msm_g1(pts, sc)
    print(f"{(time.perf_counter() - t) * 1e3:.3f} ms", flush=True)
    time.sleep(0.01)
