#!/usr/bin/env python3
"""Does a device allocation + free change the time of the verifications after it?  (DESIGN.md section 8.10c:
`bench.py --mode verify` alone reports ~950 /s where the default line reports ~1,170.)  One process:
verify x N, hipMalloc + hipFree through the HIP runtime directly, verify x N again."""
import ctypes as C
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


def loop(n, tag):
    for i in range(20):
        assert cm.verify_proof(crs, proof, Rs, Ss, Ts, Us, M, cm.Rand(100 + i))
    t = time.perf_counter()
    for i in range(n):
        assert cm.verify_proof(crs, proof, Rs, Ss, Ts, Us, M, cm.Rand(100 + i))
    ms = (time.perf_counter() - t) * 1e3 / n
    print("%-46s %.3f ms per verification = %.0f /s" % (tag, ms, 1e3 / ms), flush=True)


hip = C.CDLL("libamdhip64.so")
loop(200, "fresh process")
loop(200, "again")
p = C.c_void_p()
size = int(sys.argv[1]) if len(sys.argv) > 1 else (64 << 20)
assert hip.hipMalloc(C.byref(p), C.c_size_t(size)) == 0
assert hip.hipFree(p) == 0
loop(200, "after hipMalloc + hipFree of %d MiB" % (size >> 20))
loop(200, "again")
