#!/usr/bin/env python3
"""A few synchronous MSM calls, for a rocprofv3 --kernel-trace timeline:
   rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace -- python3 tools/trace_one_call.py 20 3"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "go-curdleproofs_amd"))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import curdlemsm as cm
from bench import uniform_scalars

logn = int(sys.argv[1]) if len(sys.argv) > 1 else 20
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
cm.init(0)
n = 1 << logn
d_pts = torch.empty((n, 12), dtype=torch.int64, device="cuda:0")
cm.synth_points_walk_device(12345, 6789, n, d_pts.data_ptr())
d_sc = torch.from_numpy(uniform_scalars(np.random.default_rng(2), n).view(np.int64)).to("cuda:0")
torch.cuda.synchronize()
for _ in range(reps):
    cm.msm_g1_device(d_pts.data_ptr(), d_sc.data_ptr(), n)
    torch.cuda.synchronize()
