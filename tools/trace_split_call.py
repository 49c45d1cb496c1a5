#!/usr/bin/env python3
"""A few mid-size MSMs as TWO window halves in flight (submit high, submit low, wait, wait, g1_sum), for a rocprofv3 timeline:
   rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace -- python3 tools/trace_split_call.py 16 3 [parts]
(tools/exp_split_call.py is the timing; this is where the time goes)."""
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

logn = int(sys.argv[1]) if len(sys.argv) > 1 else 16
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
parts = int(sys.argv[3]) if len(sys.argv) > 3 else 2
cm.init(0)
n = 1 << logn
d_pts = torch.empty((n, 12), dtype=torch.int64, device="cuda:0")
cm.synth_points_walk_device(12345, 6789, n, d_pts.data_ptr())
d_sc = torch.from_numpy(uniform_scalars(np.random.default_rng(2), n).view(np.int64)).to("cuda:0")
torch.cuda.synchronize()
W = cm.num_windows(n)
flags = cm.MSM_BASES_UNCHANGED
cuts = [W * i // parts for i in range(parts + 1)]
rng = [(cuts[i], cuts[i + 1]) for i in range(parts)][::-1]
for _ in range(reps):
    t = time.perf_counter()
    ts = [cm.msm_g1_device_submit(d_pts.data_ptr(), d_sc.data_ptr(), n, 0, b, e, flags=flags) for b, e in rng]
    t1 = time.perf_counter()
    r = [cm.msm_wait(x) for x in ts]
    t2 = time.perf_counter()
    cm.g1_sum(np.stack(r))
    t3 = time.perf_counter()
    print(f"submit {1e3 * (t1 - t):.3f} ms, waits {1e3 * (t2 - t1):.3f}, sum {1e3 * (t3 - t2):.3f}", flush=True)
    torch.cuda.synchronize()
    time.sleep(0.005)
