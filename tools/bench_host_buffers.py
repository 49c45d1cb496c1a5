"""curdle_msm_g1 from HOST buffers (what a cgo caller hands over: pageable memory), i.e. the
PCIe-inclusive latency of one call, against the device-resident call.
    python tools/bench_host_buffers.py
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "go-curdleproofs_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle", "py"))
import numpy as np
import torch
import curdlemsm as cm
import bls12381_ref as o

cm.init(0)
k, q = o.Rand(1).get_frs(2)
for logn in (16, 18, 20, 22):
    n = 1 << logn
    d_pts = torch.empty((n, 12), dtype=torch.int64, device="cuda:0")
    cm.synth_points_walk_device(k, q, n, d_pts.data_ptr())
    pts = d_pts.cpu().numpy().view(np.uint64)
    rng = np.random.default_rng(2)
    sc = rng.integers(0, 1 << 64, size=(n, 4), dtype=np.uint64)
    sc[:, 3] &= np.uint64((1 << 62) - 1)
    d_sc = torch.from_numpy(sc.view(np.int64)).to("cuda:0")
    ref = cm.msm_g1_device(d_pts.data_ptr(), d_sc.data_ptr(), n)
    got = cm.msm_g1(pts, sc)
    assert (got == ref).all()
    reps = 5
    t0 = time.perf_counter()
    for _ in range(reps):
        cm.msm_g1(pts, sc)
    t_host = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    for _ in range(reps):
        cm.msm_g1_device(d_pts.data_ptr(), d_sc.data_ptr(), n)
    t_dev = (time.perf_counter() - t0) / reps
    print(f"n=2^{logn}: host buffers {t_host*1e3:.2f} ms ({128*n/t_host/1e9:.1f} GB/s of input), device-resident {t_dev*1e3:.2f} ms", flush=True)
