"""Per-phase timing of small MSMs (the size of one verification's accumulator MSM and
smaller), where the call is latency-bound rather than throughput-bound.
    python tools/small_msm_profile.py [n ...]
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "go-curdleproofs_amd"))
import numpy as np
import curdlemsm as cm

cm.init(0)
sizes = [int(a) for a in sys.argv[1:]] or [64, 256, 1200, 4096, 16384]
rand = cm.Rand(5)
for n in sizes:
    pts = rand.get_g1_affines(min(n, 512))
    reps = (n + len(pts) - 1) // len(pts)
    pts = np.concatenate([pts] * reps)[:n]
    sc = np.stack([rand.get_fr() for _ in range(min(n, 512))])
    sc = np.concatenate([sc] * reps)[:n].copy()
    cm.msm_g1(pts, sc)
    t0 = time.perf_counter()
    for _ in range(20):
        cm.msm_g1(pts, sc)
    dt = (time.perf_counter() - t0) / 20
    cm.profile_enable(True)
    cm.msm_g1(pts, sc)
    prof = cm.profile_last()
    cm.profile_enable(False)
    ks = ", ".join(f"{k} {v*1e3:.0f}" for k, v in prof["kernels"].items())
    print(f"n={n}: {dt*1e3:.3f} ms per host call; c={prof['window_bits']} W={prof['num_windows']}; phases (us): {ks}", flush=True)
