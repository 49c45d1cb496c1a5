"""BASELINE.json's configurations at their REAL sizes (configs 3 and 5 were only
exercised at reduced size in -m gpu): the full Prove -> Verify at n = 256 (ell = 252, config 3,
curdleproof_test.go:184-237 benches exactly this size) and n = 512 in both check modes with
the soundness flips of curdleproof_test.go:48-182; the 1,024 x 628-pair MSM batch of config 5
against the C oracle; cross-proof batch verification at k = 256 with planted bad proofs."""
import os

import numpy as np
import pytest

from test_protocol_gpu import setup
from test_msm_gpu import rand_scalars

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n", [256, 512])
def test_full_verify_at_baseline_size(gpu, oracle, n):
    crs, Rs, Ss, Ts, Us, M, perm, k, rs_m = setup(gpu, n)
    ell = n - 4
    proof = gpu.prove(crs, Rs, Ss, Ts, Us, M, perm, k, rs_m, gpu.Rand(42))
    m = n.bit_length() - 1
    # wire size: 18 single points, 10 vectors of m points behind a 4-byte prefix, 7 scalars
    assert len(proof) == 48 * 18 + 10 * (4 + 48 * m) + 32 * 7
    other = gpu.Rand(1234).generate_permutation(ell)
    r2 = gpu.Rand(99)
    T2, U2, _, _ = gpu.shuffle_permute_commit(crs, Rs, Ss, perm, r2.get_fr(), r2)
    aff = gpu.g1_decompress(gpu.g1_compress(M), False)
    k_int = oracle.fr_from_mont_limbs([int(v) for v in k])
    Mk = np.array(oracle.jac_to_mont_limbs(oracle.scalar_mul(k_int, oracle.jac_from_mont_limbs([int(v) for v in aff]))),
                  dtype=np.uint64)
    try:
        for eager in (False, True):
            gpu.verify_set_eager(eager)
            assert gpu.verify(crs, proof, Rs, Ss, Ts, Us, M, gpu.Rand(43)) is True                      # completeness
            assert gpu.verify(crs, proof, Ss, Rs, Ts, Us, M, gpu.Rand(43)) is False                     # flips Ss and Rs
            assert gpu.verify(crs, proof, Rs, Ss, Ts[other], Us[other], M, gpu.Rand(43)) is False       # another permutation
            assert gpu.verify(crs, proof, Rs, Ss, Ts, Us, Mk, gpu.Rand(43)) is False                    # wrong commitment
            assert gpu.verify(crs, proof, Rs, Ss, T2, U2, M, gpu.Rand(43)) is False                     # another randomizer
    finally:
        gpu.verify_set_eager(False)
    assert gpu.proof_reencode(proof) == proof


def test_config5_msm_batch_1024_x_628(gpu, oracle, coracle):
    """1,024 independent 628-pair MSMs (the Whisk verifier's final MSM, ell = 124) in ONE call:
    sampled members, the empty one and the all-infinity one against the C oracle, all members
    against a second (device-resident) run, and the batch total against one big MSM."""
    import torch
    k, q = oracle.Rand(21).get_frs(2)
    sizes = [628] * 1024
    sizes[17], sizes[400] = 0, 3
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint64)
    n = int(offs[-1])
    d_pts = torch.empty((n, 12), dtype=torch.int64, device="cuda:0")
    gpu.synth_points_walk_device(k, q, n, d_pts.data_ptr())
    pts = d_pts.cpu().numpy().view(np.uint64).copy()
    pts[offs[33]:offs[34]] = 0                                   # MSM 33: every base is the point at infinity
    sc = rand_scalars(np.random.default_rng(21), n, oracle)
    out = gpu.msm_g1_batch(pts, sc, offs)
    assert out.shape == (1024, 18)
    rng = np.random.default_rng(5)
    sample = sorted(set([0, 1, 17, 33, 400, 511, 1022, 1023]) | set(int(v) for v in rng.integers(0, 1024, 28)))
    assert len(sample) >= 32
    for j in sample:
        lo, hi = int(offs[j]), int(offs[j + 1])
        assert (out[j] == coracle.msm_pippenger(pts[lo:hi], sc[lo:hi], threads=4)).all(), j
    inf = coracle.msm_pippenger(pts[:0], sc[:0])
    assert (out[17] == inf).all() and (out[33] == inf).all()
    d_p = torch.from_numpy(pts.view(np.int64)).to("cuda:0")
    d_s = torch.from_numpy(sc.view(np.int64)).to("cuda:0")
    assert (gpu.msm_g1_batch_device(d_p.data_ptr(), d_s.data_ptr(), offs) == out).all()
    # checksum of checksums: the sum of all 1,024 results is the MSM over all pairs at once
    assert (gpu.g1_sum(out) == gpu.msm_g1_device(d_p.data_ptr(), d_s.data_ptr(), n)).all()


def test_cross_proof_batch_verification_at_k_256(gpu):
    """curdle_verify_batch at k = 256 over one CRS (ell = 60) with planted bad members: a proof
    checked against another proof's instance, a truncated proof, a proof with one flipped bit --
    the accept bits stay exact."""
    n = 64
    ell = n - 4
    rand = gpu.Rand(0)
    crs = gpu.CRS(ell, rand)
    base = []
    for j in range(4):                                           # four distinct honest (proof, instance) pairs
        perm = gpu.Rand(100 + j).generate_permutation(ell)
        k = rand.get_fr()
        Rs, Ss = rand.get_g1_affines(ell), rand.get_g1_affines(ell)
        Ts, Us, M, rs_m = gpu.shuffle_permute_commit(crs, Rs, Ss, perm, k, rand)
        base.append([gpu.prove(crs, Rs, Ss, Ts, Us, M, perm, k, rs_m, gpu.Rand(42 + j)), Rs, Ss, Ts, Us, M])
    items = [list(base[i % 4]) for i in range(256)]
    expect = [True] * 256
    items[5][1], items[5][2] = base[2][1], base[2][2]            # instance of another proof
    expect[5] = False
    items[77][0] = items[77][0][:-9]                             # truncated
    expect[77] = False
    flipped = bytearray(items[200][0])
    flipped[len(flipped) - 20] ^= 0x10                           # inside the last scalar
    items[200][0] = bytes(flipped)
    expect[200] = False
    items[255][3] = base[(255 + 1) % 4][3]                       # Ts of another instance
    expect[255] = False
    cols = [list(c) for c in zip(*items)]
    got = gpu.verify_batch(crs, *cols, gpu.Rand(9), nthreads=8)
    assert got == expect


def test_config5_1024_whisk_shuffle_proofs_in_one_batch(gpu, oracle):
    """BASELINE config 5 as stated: 1,024 IsValidWhiskShuffleProof verifications (whisk.go:20-61)
    in one curdle_whisk_is_valid_shuffle_proof_batch call, over 8 distinct honest shuffles
    (whisk_test.go:13-90 generates one; here eight, repeated) with planted bad members -- post
    trackers of another shuffle, pre and post swapped, one post tracker replaced, a tracker that
    is not a curve point, a tracker point outside the prime-order subgroup, a proof cut off half way, a
    proof with one bit flipped, a proof that does not parse -- and the accept bits exact; then
    the same batch with every member honest."""
    from test_whisk import shuffle_trackers
    crs = gpu.CRS(gpu.WHISK_ELL, gpu.Rand(4))
    sets = []
    for j in range(8):
        pre = shuffle_trackers(gpu, oracle, gpu.Rand(300 + j), gpu.WHISK_ELL)
        post, proof = gpu.whisk_generate_shuffle_proof(crs, pre, gpu.Rand(600 + j))
        assert gpu.whisk_is_valid_shuffle_proof(crs, pre, post, proof, gpu.Rand(j)) is True
        sets.append((pre, post, proof))
    assert len({s[2] for s in sets}) == 8
    k = 1024
    pres = [sets[i % 8][0] for i in range(k)]
    posts = [sets[i % 8][1] for i in range(k)]
    proofs = [sets[i % 8][2] for i in range(k)]
    assert gpu.whisk_is_valid_shuffle_proof_batch(crs, pres, posts, proofs, gpu.Rand(1), nthreads=16) == [True] * k
    expect = [True] * k
    # a point on the curve outside G1 (SetBytes rejects it, types.go:85-95)
    p = oracle.P
    x = 6
    while True:
        rhs = (x * x * x + 4) % p
        y = pow(rhs, (p + 1) // 4, p)
        if y * y % p == rhs and oracle.scalar_mul(oracle.R, (x, y)) is not None:
            break
        x += 1
    rogue = oracle.compress((x, y))

    def plant(i, pre=None, post=None, proof=None):
        if pre is not None:
            pres[i] = pre
        if post is not None:
            posts[i] = post
        if proof is not None:
            proofs[i] = proof
        expect[i] = False

    plant(0, post=sets[1][1])                                         # another shuffle's post trackers
    plant(9, pre=sets[1][1], post=sets[1][0])                         # swapped
    swapped = list(sets[2][1])
    swapped[5] = swapped[6]
    plant(130, post=swapped)
    notcurve = list(sets[3][1])
    notcurve[3] = b"\x01" * 96
    plant(259, post=notcurve)
    outside = list(sets[4][0])
    outside[7] = outside[7][:48] + rogue
    plant(516, pre=outside)
    plant(645, proof=sets[5][2][:2000] + b"\x00" * 2576)               # cut off inside the fixed 4,576-byte array (types.go:67-69)
    flipped = bytearray(sets[6][2])
    flipped[4536 - 20] ^= 0x10                                        # inside the last scalar of the proof
    plant(774, proof=bytes(flipped))
    plant(1023, proof=b"\x00" * 4576)
    plant(1022, proof=rogue + sets[6][2][48:])                        # M outside the subgroup
    for nthreads in (16, 3):
        assert gpu.whisk_is_valid_shuffle_proof_batch(crs, pres, posts, proofs, gpu.Rand(2), nthreads=nthreads) == expect


@pytest.mark.parametrize("k,size", [(1024, 2548), (2100, 628)])
def test_batches_beyond_one_pass_of_bucket_slots(gpu, oracle, coracle, k, size):
    """A batch whose bucket slots exceed what the slot scans hold in one pass (4,194,304: 1,024
    verifier MSMs at ell = 508, or 2,100 Whisk ones) runs in passes inside
    curdle_msm_g1_batch[_device] instead of being refused: sampled members against the C oracle,
    the first and last member of every pass boundary region, and the sum of all results against
    one MSM over all pairs."""
    import torch
    kk, q = oracle.Rand(77).get_frs(2)
    sizes = [size] * k
    sizes[5], sizes[k - 2] = 0, 3
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint64)
    n = int(offs[-1])
    d_pts = torch.empty((n, 12), dtype=torch.int64, device="cuda:0")
    gpu.synth_points_walk_device(kk, q, n, d_pts.data_ptr())
    sc = rand_scalars(np.random.default_rng(k), n, oracle)
    d_sc = torch.from_numpy(sc.view(np.int64)).to("cuda:0")
    out = gpu.msm_g1_batch_device(d_pts.data_ptr(), d_sc.data_ptr(), offs)
    assert out.shape == (k, 18)
    pts = d_pts.cpu().numpy().view(np.uint64)
    rng = np.random.default_rng(size)
    sample = sorted(set([0, 1, 5, k - 2, k - 1]) | set(int(v) for v in rng.integers(0, k, 24)))
    for j in sample:
        lo, hi = int(offs[j]), int(offs[j + 1])
        assert (out[j] == coracle.msm_pippenger(pts[lo:hi], sc[lo:hi], threads=4)).all(), j
    assert (gpu.g1_sum(out) == gpu.msm_g1_device(d_pts.data_ptr(), d_sc.data_ptr(), n)).all()
    # the same batch forced into many small passes, and from host buffers
    try:
        gpu.plan_override("MAX_MSMS_PER_PASS", 300)
        assert (gpu.msm_g1_batch_device(d_pts.data_ptr(), d_sc.data_ptr(), offs) == out).all()
    finally:
        gpu.plan_override("MAX_MSMS_PER_PASS", None)
    if k == 2100:
        assert (gpu.msm_g1_batch(pts, sc, offs) == out).all()


def test_plan_tables_have_not_regressed_by_a_factor(gpu, oracle):
    """The plan rules (window widths, segment lengths, scatter form, chunking) are tables of
    measured steps; the parity tests catch a wrong result, not a table entry that makes a call
    twice as slow.  Loose ceilings on one synchronous call at the BASELINE sizes -- about twice
    what profiles/r03_sweep.json shows on a healthy box -- so that a broken rule fails here
    instead of in the bench: 1,268 pairs (config 3's final MSM, 0.3-0.4 ms), 2^16 (config 2,
    0.7 ms), 2^20 (the headline, 3.4-3.6 ms; 5.0-5.6 ms from host slices)."""
    import time
    import torch
    k, q = oracle.Rand(1).get_frs(2)
    n = 1 << 20
    d_pts = torch.empty((n, 12), dtype=torch.int64, device="cuda:0")
    gpu.synth_points_walk_device(k, q, n, d_pts.data_ptr())
    sc = rand_scalars(np.random.default_rng(3), n, oracle)
    d_sc = torch.from_numpy(sc.view(np.int64)).to("cuda:0")
    pts = d_pts.cpu().numpy().view(np.uint64)

    def best_of(fn, reps=5):
        fn()
        ts = []
        for _ in range(reps):
            torch.cuda.synchronize()
            t = time.perf_counter()
            fn()
            ts.append((time.perf_counter() - t) * 1e3)
        return min(ts)

    for m, ceiling in ((1268, 0.9), (1 << 16, 1.6), (1 << 20, 7.0)):
        ms = best_of(lambda: gpu.msm_g1_device(d_pts.data_ptr(), d_sc.data_ptr(), m))
        assert ms < ceiling, (m, ms)
    ms = best_of(lambda: gpu.msm_g1(pts, sc), reps=3)
    assert ms < 11.0, ms
    # ... and the verifier's worst case (samepermutationargument.go:67): every scalar equal, so that every
    # term of a window lands in one of two buckets.  Until round 6 such a call took 2.4x a uniform one at 2^16 pairs and
    # 2.6x from host slices at 2^20 (an empty-slot walk in k_accumulate, one block per large bucket in k_merge_large:
    # profiles/r06_adversarial_before.json); now it is within a tenth of uniform (profiles/r06_adversarial.json).
    beta = np.tile(sc[7], (n, 1))
    d_beta = torch.from_numpy(beta.view(np.int64)).to("cuda:0")
    for m in (1 << 16, 1 << 20):
        uni = best_of(lambda: gpu.msm_g1_device(d_pts.data_ptr(), d_sc.data_ptr(), m))
        eq = best_of(lambda: gpu.msm_g1_device(d_pts.data_ptr(), d_beta.data_ptr(), m))
        assert eq < 1.5 * uni, (m, eq, uni)
    eq = best_of(lambda: gpu.msm_g1(pts, beta), reps=3)
    assert eq < 1.5 * ms, (eq, ms)
