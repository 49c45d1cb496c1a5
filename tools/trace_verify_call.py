import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "go-curdleproofs_amd"))
import numpy as np, curdlemsm as cm
cm.init(0)
ell = 252
rand = cm.Rand(0)
crs = cm.CRS(ell, rand)
perm = cm.Rand(42).generate_permutation(ell)
k = rand.get_fr()
Rs, Ss = rand.get_g1_affines(ell), rand.get_g1_affines(ell)
Ts, Us, M, rs_m = cm.shuffle_permute_commit(crs, Rs, Ss, perm, k, rand)
proof = cm.Proof(cm.prove(crs, Rs, Ss, Ts, Us, M, perm, k, rs_m, cm.Rand(42)))
for i in range(300):
    assert cm.verify_proof(crs, proof, Rs, Ss, Ts, Us, M, cm.Rand(43 + i))
time.sleep(0.01)
t = time.perf_counter()
assert cm.verify_proof(crs, proof, Rs, Ss, Ts, Us, M, cm.Rand(999))
print(f"{(time.perf_counter()-t)*1e3:.4f} ms", file=sys.stderr)
