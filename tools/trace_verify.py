#!/usr/bin/env python3
"""A few curdleproof.Verify calls of a decoded proof at ell = 252, for a rocprofv3 timeline and
the library's own CURDLE_VERIFY_TRACE:
   CURDLE_VERIFY_TRACE=1 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d out -- python3 tools/trace_verify.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "go-curdleproofs_amd"))
import curdlemsm as cm

cm.init(0)
ell = 252
rand = cm.Rand(0)
crs = cm.CRS(ell, rand)
perm = cm.Rand(42).generate_permutation(ell)
k = rand.get_fr()
Rs, Ss = rand.get_g1_affines(ell), rand.get_g1_affines(ell)
Ts, Us, M, rs_m = cm.shuffle_permute_commit(crs, Rs, Ss, perm, k, rand)
proof = cm.Proof(cm.prove(crs, Rs, Ss, Ts, Us, M, perm, k, rs_m, cm.Rand(42)))
for i in range(12):
    t = time.perf_counter()
    assert cm.verify_proof(crs, proof, Rs, Ss, Ts, Us, M, cm.Rand(100 + i))
    print(f"verify {i}: {(time.perf_counter() - t) * 1e3:.3f} ms", flush=True)
    time.sleep(0.005)
