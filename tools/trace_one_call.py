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

# first argument: log2 of the pair count, or "n=<pairs>" for any size (the verifier's 1,268)
a1 = sys.argv[1] if len(sys.argv) > 1 else "20"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
cm.init(0)
n = int(a1[2:]) if a1.startswith("n=") else 1 << int(a1)
d_pts = torch.empty((n, 12), dtype=torch.int64, device="cuda:0")
cm.synth_points_walk_device(12345, 6789, n, d_pts.data_ptr())
d_sc = torch.from_numpy(uniform_scalars(np.random.default_rng(2), n).view(np.int64)).to("cuda:0")
torch.cuda.synchronize()
import time
for _ in range(reps):
    t = time.perf_counter()
    cm.msm_g1_device(d_pts.data_ptr(), d_sc.data_ptr(), n)
    dt = time.perf_counter() - t
    torch.cuda.synchronize()
    time.sleep(0.005)          # a gap in the trace between two calls (tools/timeline.py cuts at 2 ms)
print(f"last call: {dt * 1e3:.4f} ms on the host", file=sys.stderr)
