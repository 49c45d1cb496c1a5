"""Generates tests/golden/proof_vectors.npz: serialised proofs of THIS implementation's prover
for fixed seeds (ell = 12, 60, 124), with the seeds that rebuild the CRS and the instance, and
the Whisk shuffle proof (4,576 bytes) for seed 7.  They pin the transcript framing, the wire
format and the prover's use of common.Rand from now on: any drift in labels, challenge order,
encodings or draw order changes these bytes (tests/test_proof_fixtures.py).  They are NOT
Go-produced vectors (none can be made here: no Go toolchain) -- interoperability with the Go
implementation stays UNVERIFIED; this catches regressions of what was checked by reading.

Run on a GPU box (the prover's MSMs run on the GPU):
    python tests/golden/gen_proof_fixtures.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "go-curdleproofs_amd"))
import curdlemsm as cm  # noqa: E402


def instance(ell, seed):
    """The reference's test setup (curdleproof_test.go:239-274) with common.Rand throughout."""
    rand = cm.Rand(seed)
    crs = cm.CRS(ell, rand)
    perm = cm.Rand(seed + 42).generate_permutation(ell)
    k = rand.get_fr()
    Rs, Ss = rand.get_g1_affines(ell), rand.get_g1_affines(ell)
    Ts, Us, M, rs_m = cm.shuffle_permute_commit(crs, Rs, Ss, perm, k, rand)
    return crs, Rs, Ss, Ts, Us, M, perm, k, rs_m


def main():
    cm.init(0)
    out = {}
    for ell, seed in ((12, 3), (60, 0), (124, 1)):
        crs, Rs, Ss, Ts, Us, M, perm, k, rs_m = instance(ell, seed)
        proof = cm.prove(crs, Rs, Ss, Ts, Us, M, perm, k, rs_m, cm.Rand(seed + 1000))
        assert cm.verify(crs, proof, Rs, Ss, Ts, Us, M, cm.Rand(5))
        out[f"ell{ell}_seed"] = np.array([seed])
        out[f"ell{ell}_proof"] = np.frombuffer(proof, dtype=np.uint8)
        out[f"ell{ell}_M"] = np.frombuffer(cm.g1_compress(M), dtype=np.uint8)
        out[f"ell{ell}_T0"] = Ts[0]
    # Whisk: 124 trackers from common.Rand(7), proof from the same stream
    rand = cm.Rand(7)
    crs = cm.CRS(124, rand)
    pts = rand.get_g1_affines(124)
    ks = rand.get_frs(124)
    one = np.array([0x760900000002fffd, 0xebf4000bc40c0002, 0x5f48985753c758ba, 0x77ce585370525745, 0x5c071a97a256ec6d,
                    0x15f65ec3fa80e493], dtype=np.uint64)
    krg = cm.g1_scalar_mul_batch(pts, ks)
    pre = [cm.g1_compress(np.concatenate([p, one])) + cm.g1_compress(np.concatenate([q, one])) for p, q in zip(pts, krg)]
    post, proof = cm.whisk_generate_shuffle_proof(crs, pre, rand)
    assert cm.whisk_is_valid_shuffle_proof(crs, pre, post, proof, cm.Rand(8))
    out["whisk_pre"] = np.frombuffer(b"".join(pre), dtype=np.uint8)
    out["whisk_post"] = np.frombuffer(b"".join(post), dtype=np.uint8)
    out["whisk_proof"] = np.frombuffer(proof, dtype=np.uint8)
    path = os.path.join(ROOT, "gpurun_out", "proof_vectors.npz")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    np.savez_compressed(path, **out)
    print("wrote", path, {k_: v.shape for k_, v in out.items()})


if __name__ == "__main__":
    main()
