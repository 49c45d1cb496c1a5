"""Latency of curdle_g1_scalar_mul_batch by batch size, with one shared scalar (the fold step)
and with per-point scalars.
    python tools/bench_scalar_mul_batch.py
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "go-curdleproofs_amd"))
import numpy as np
import curdlemsm as cm
cm.init(0)
r = cm.Rand(3)
for n in (32, 256, 756, 4096):
    P = r.get_g1_affines(min(n,256)); P = np.concatenate([P]*((n+255)//256))[:n]
    A = P[::-1].copy()
    s1 = r.get_fr()
    sn = np.stack([r.get_fr() for _ in range(min(n,256))]); sn = np.concatenate([sn]*((n+255)//256))[:n].copy()
    for name, sc in (("shared", s1), ("per-lane", sn)):
        cm.g1_scalar_mul_batch(P, sc, A)
        t0 = time.perf_counter()
        for _ in range(10): cm.g1_scalar_mul_batch(P, sc, A)
        print(n, name, round((time.perf_counter()-t0)/10*1e3, 3), "ms", flush=True)
