"""Every single-bit change of a serialised proof must be rejected (accept bit False or a
decoding error), in both evaluation modes of the verifier -- the whole proof is bound by the
transcript and the accumulated checks.
    python tools/fuzz_proof_bits.py [n] [stride]
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "go-curdleproofs_amd"))
import numpy as np
import curdlemsm as cm

cm.init(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
stride = int(sys.argv[2]) if len(sys.argv) > 2 else 1
ell = n - 4
rand = cm.Rand(0)
crs = cm.CRS(ell, rand)
perm = cm.Rand(42).generate_permutation(ell)
k = rand.get_fr()
Rs, Ss = rand.get_g1_affines(ell), rand.get_g1_affines(ell)
Ts, Us, M, rs_m = cm.shuffle_permute_commit(crs, Rs, Ss, perm, k, rand)
proof = cm.prove(crs, Rs, Ss, Ts, Us, M, perm, k, rs_m, cm.Rand(1))
assert cm.verify(crs, proof, Rs, Ss, Ts, Us, M, cm.Rand(2))
rng = np.random.default_rng(0)
stats = {"rejected": 0, "error": 0}
t0 = time.time()
for mode in (False, True):
    prev = cm.verify_set_eager(mode)
    for pos in range(0, len(proof), stride):
        b = bytearray(proof)
        b[pos] ^= 1 << int(rng.integers(0, 8))
        try:
            ok = cm.verify(crs, bytes(b), Rs, Ss, Ts, Us, M, cm.Rand(3))
        except cm.CurdleError:
            stats["error"] += 1
            continue
        if ok:
            print(f"ACCEPTED a proof with byte {pos} changed (eager={mode})")
            sys.exit(1)
        stats["rejected"] += 1
    cm.verify_set_eager(prev)
print(f"fuzz_proof_bits: n={n}, {len(proof)} bytes, stride {stride}, both modes: {stats} in {time.time()-t0:.0f} s")
