"""Serialised proofs of this implementation's prover for fixed seeds, committed under
tests/golden/proof_vectors.npz (generator: tests/golden/gen_proof_fixtures.py).  The prover is
deterministic in its seeds, so these bytes pin the transcript framing (labels, challenge
order), the wire format and the draw order of common.Rand: any drift shows up here.  They are
this implementation's own vectors -- Go-produced ones cannot be made in this environment, and
byte compatibility with the Go implementation stays UNVERIFIED."""
import os

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def vectors():
    return np.load(os.path.join(ROOT, "tests", "golden", "proof_vectors.npz"))


def instance(gpu, ell, seed):
    rand = gpu.Rand(seed)
    crs = gpu.CRS(ell, rand)
    perm = gpu.Rand(seed + 42).generate_permutation(ell)
    k = rand.get_fr()
    Rs, Ss = rand.get_g1_affines(ell), rand.get_g1_affines(ell)
    Ts, Us, M, rs_m = gpu.shuffle_permute_commit(crs, Rs, Ss, perm, k, rand)
    return crs, Rs, Ss, Ts, Us, M, perm, k, rs_m


@pytest.mark.parametrize("ell", [12, 60, 124])
def test_prover_reproduces_the_committed_proof_bytes(gpu, vectors, ell):
    seed = int(vectors[f"ell{ell}_seed"][0])
    crs, Rs, Ss, Ts, Us, M, perm, k, rs_m = instance(gpu, ell, seed)
    assert gpu.g1_compress(M) == vectors[f"ell{ell}_M"].tobytes()
    assert (Ts[0] == vectors[f"ell{ell}_T0"]).all()
    want = vectors[f"ell{ell}_proof"].tobytes()
    assert gpu.prove(crs, Rs, Ss, Ts, Us, M, perm, k, rs_m, gpu.Rand(seed + 1000)) == want
    # the committed bytes verify on both accumulators and in eager mode
    for device_acc, eager in ((True, False), (False, False), (False, True)):
        gpu.verify_set_device_acc(device_acc)
        gpu.verify_set_eager(eager)
        try:
            assert gpu.verify(crs, want, Rs, Ss, Ts, Us, M, gpu.Rand(5)) is True
            assert gpu.verify(crs, want, Ss, Rs, Ts, Us, M, gpu.Rand(5)) is False
        finally:
            gpu.verify_set_device_acc(True)
            gpu.verify_set_eager(False)


def test_whisk_shuffle_proof_fixture(gpu, vectors):
    pre = vectors["whisk_pre"].tobytes()
    post = vectors["whisk_post"].tobytes()
    proof = vectors["whisk_proof"].tobytes()
    # the reference's fixed sizes (whisk/types.go:14-21): 96-byte trackers, a 4,576-byte proof = M + 4,488
    # proof bytes + zero padding (4-byte slice prefixes leave 40 bytes of padding)
    assert len(proof) == 4576 and proof[4536:] == b"\x00" * 40 and any(proof[4500:4536])
    pre_l = [pre[96 * i:96 * (i + 1)] for i in range(124)]
    post_l = [post[96 * i:96 * (i + 1)] for i in range(124)]
    crs = gpu.CRS(124, gpu.Rand(7))
    assert gpu.whisk_is_valid_shuffle_proof(crs, pre_l, post_l, proof, gpu.Rand(8)) is True
    assert gpu.whisk_is_valid_shuffle_proof(crs, post_l, pre_l, proof, gpu.Rand(8)) is False
