"""End-to-end Curdleproofs on this stack: the host restatement of the protocol
(go-curdleproofs_amd/host/curdleproofs.cpp) with every MSM and the final batched
msmaccumulator check on the GPU.  Mirrors the reference's own tests
(/root/reference/curdleproof_test.go): completeness, the four soundness cases, the
serialisation round trip.  Inputs follow the reference's `setup` (curdleproof_test.go:
239-274) with common.Rand.GeneratePermutation in place of math/rand (whose stream is
not reproducible outside Go)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def setup(cm, n):
    ell = n - 4
    rand = cm.Rand(0)
    crs = cm.CRS(ell, rand)
    perm = cm.Rand(42).generate_permutation(ell)
    k = rand.get_fr()
    Rs = rand.get_g1_affines(ell)
    Ss = rand.get_g1_affines(ell)
    Ts, Us, M, rs_m = cm.shuffle_permute_commit(crs, Rs, Ss, perm, k, rand)
    return crs, Rs, Ss, Ts, Us, M, perm, k, rs_m


@pytest.fixture(params=["deferred", "eager"])
def check_mode(gpu, request):
    """Both ways of evaluating the verifier's check points must give the same accept bit."""
    prev = gpu.verify_set_eager(request.param == "eager")
    yield request.param
    gpu.verify_set_eager(prev)


def test_completeness(gpu, check_mode):
    # curdleproof_test.go:16-46, n = 64, prover seed 42, verifier seed 43
    crs, Rs, Ss, Ts, Us, M, perm, k, rs_m = setup(gpu, 64)
    proof = gpu.prove(crs, Rs, Ss, Ts, Us, M, perm, k, rs_m, gpu.Rand(42))
    assert gpu.verify(crs, proof, Rs, Ss, Ts, Us, M, gpu.Rand(43)) is True
    # the accept bit does not depend on the verifier's randomness
    assert gpu.verify(crs, proof, Rs, Ss, Ts, Us, M, gpu.Rand(7)) is True
    # the reference's split: Proof.FromReader once, Verify(proof Proof, ...) on the value
    decoded = gpu.Proof(proof)
    assert gpu.verify_proof(crs, decoded, Rs, Ss, Ts, Us, M, gpu.Rand(43)) is True
    assert gpu.verify_proof(crs, decoded, Ss, Rs, Ts, Us, M, gpu.Rand(43)) is False
    with pytest.raises(gpu.CurdleError):
        gpu.Proof(proof[:-1])
    # a proof point on the curve but outside the subgroup (A is the first 48 bytes): the decoder
    # rejects it, in the one-step and in the overlapped (curdle_verify) form alike
    rogue = bytes.fromhex("80" + "00" * 46 + "05")       # x = 5: on the curve, not in G1
    assert gpu.g1_decompress_batch(rogue, True)[1][0] == gpu.DECODE_NOT_IN_SUBGROUP
    with pytest.raises(gpu.CurdleError):
        gpu.Proof(rogue + proof[48:])
    with pytest.raises(gpu.CurdleError) as e:
        gpu.verify(crs, rogue + proof[48:], Rs, Ss, Ts, Us, M, gpu.Rand(43))
    assert "decoding" in e.value.msg


def test_soundness_and_encoding(gpu, oracle, check_mode):
    # curdleproof_test.go:48-182, n = 128
    n = 128
    crs, Rs, Ss, Ts, Us, M, perm, k, rs_m = setup(gpu, n)
    proof = gpu.prove(crs, Rs, Ss, Ts, Us, M, perm, k, rs_m, gpu.Rand(0))
    assert gpu.verify(crs, proof, Rs, Ss, Ts, Us, M, gpu.Rand(0)) is True
    # Whisk: M (48 B) + proof = 4,536 B with 4-byte slice prefixes
    assert len(proof) == 4488

    # "flips Ss and Rs"
    assert gpu.verify(crs, proof, Ss, Rs, Ts, Us, M, gpu.Rand(0)) is False
    # "apply a different permutation than the one proved"
    other = gpu.Rand(1234).generate_permutation(n - 4)
    assert gpu.verify(crs, proof, Rs, Ss, Ts[other], Us[other], M, gpu.Rand(0)) is False
    # "provide wrong perm commitment": M * k
    aff = gpu.g1_decompress(gpu.g1_compress(M), False)
    k_int = oracle.fr_from_mont_limbs([int(v) for v in k])
    Mk = oracle.scalar_mul(k_int, oracle.jac_from_mont_limbs([int(v) for v in aff]))
    touched = np.array(oracle.jac_to_mont_limbs(Mk), dtype=np.uint64)
    assert gpu.verify(crs, proof, Rs, Ss, Ts, Us, touched, gpu.Rand(0)) is False
    # "instance outputs use a different randomizer"
    r2 = gpu.Rand(99)
    k2 = r2.get_fr()
    T2, U2, _, _ = gpu.shuffle_permute_commit(crs, Rs, Ss, perm, k2, r2)
    assert gpu.verify(crs, proof, Rs, Ss, T2, U2, M, gpu.Rand(0)) is False
    # a zero randomizer is a structural error, not a reject (curdleproof.go:213-215)
    Tz = Ts.copy()
    Tz[0] = 0
    with pytest.raises(gpu.CurdleError) as e:
        gpu.verify(crs, proof, Rs, Ss, Tz, Us, M, gpu.Rand(0))
    assert "randomizer is zero" in e.value.msg

    # "encode/decode"
    assert gpu.proof_reencode(proof) == proof
    # a bit flipped inside the proof: either the decoder refuses it or the verifier rejects
    for pos in (10, 700, len(proof) - 5):
        bad = bytearray(proof)
        bad[pos] ^= 0x01
        try:
            assert gpu.verify(crs, bytes(bad), Rs, Ss, Ts, Us, M, gpu.Rand(0)) is False
        except gpu.CurdleError:
            pass
    with pytest.raises(gpu.CurdleError):
        gpu.verify(crs, proof[:-3], Rs, Ss, Ts, Us, M, gpu.Rand(0))


def test_prover_is_deterministic_in_its_seed(gpu):
    crs, Rs, Ss, Ts, Us, M, perm, k, rs_m = setup(gpu, 16)
    a = gpu.prove(crs, Rs, Ss, Ts, Us, M, perm, k, rs_m, gpu.Rand(5))
    b = gpu.prove(crs, Rs, Ss, Ts, Us, M, perm, k, rs_m, gpu.Rand(5))
    c = gpu.prove(crs, Rs, Ss, Ts, Us, M, perm, k, rs_m, gpu.Rand(6))
    assert a == b and a != c
    assert gpu.verify(crs, c, Rs, Ss, Ts, Us, M, gpu.Rand(1)) is True


def test_batch_verification_shares_one_accumulator(gpu, check_mode):
    """curdle_verify_batch: several proofs over one CRS, one shared
    accumulator and one MSM; the per-proof bits stay exact when one proof is bad."""
    n = 32
    ell = n - 4
    rand = gpu.Rand(0)
    crs = gpu.CRS(ell, rand)
    inst = []
    for j in range(5):
        perm = gpu.Rand(100 + j).generate_permutation(ell)
        k = rand.get_fr()
        Rs, Ss = rand.get_g1_affines(ell), rand.get_g1_affines(ell)
        Ts, Us, M, rs_m = gpu.shuffle_permute_commit(crs, Rs, Ss, perm, k, rand)
        proof = gpu.prove(crs, Rs, Ss, Ts, Us, M, perm, k, rs_m, gpu.Rand(42 + j))
        inst.append((proof, Rs, Ss, Ts, Us, M))
    cols = lambda items: [list(c) for c in zip(*items)]
    proofs, Rs, Ss, Ts, Us, Ms = cols(inst)
    assert gpu.verify_batch(crs, proofs, Rs, Ss, Ts, Us, Ms, gpu.Rand(9), nthreads=3) == [True] * 5
    assert gpu.verify_batch(crs, proofs[:1], Rs[:1], Ss[:1], Ts[:1], Us[:1], Ms[:1], gpu.Rand(9), nthreads=4) == [True]
    assert gpu.verify_batch(crs, [], [], [], [], [], [], gpu.Rand(9)) == []
    # proof 2 checked against proof 3's instance, proof 4 truncated: exact bits, the others still accept
    bad = [list(x) for x in (proofs, Rs, Ss, Ts, Us, Ms)]
    bad[1][2], bad[2][2] = Rs[3], Ss[3]
    bad[0][4] = proofs[4][:-7]
    assert gpu.verify_batch(crs, *bad, gpu.Rand(9), nthreads=2) == [True, True, False, True, False]


def test_gpu_decodes_the_published_generator_encoding(gpu, oracle):
    from test_oracle import G1_GENERATOR_COMPRESSED, G1_INFINITY_COMPRESSED
    recs = [G1_GENERATOR_COMPRESSED, bytes([0xB7]) + G1_GENERATOR_COMPRESSED[1:], G1_INFINITY_COMPRESSED] * 20
    pts, st = gpu.g1_decompress_batch(b"".join(recs), True)
    want = [oracle.G1, oracle.neg(oracle.G1), None] * 20
    for g, w, code in zip(pts, want, st):
        if w is None:
            assert code == gpu.DECODE_INFINITY and not g.any()
        else:
            assert code == gpu.DECODE_OK and oracle.affine_from_mont_limbs([int(v) for v in g]) == w


def test_batched_point_decoding_on_the_gpu(gpu, oracle):
    """curdle_g1_decompress_batch: every status class, against the oracle's compress / curve /
    subgroup definitions and the host decoder."""
    import random
    p, r = oracle.P, oracle.R
    rnd = random.Random(11)
    recs, want_pt, want_st = [], [], []

    def put(enc, pt, st):
        recs.append(enc)
        want_pt.append(pt)
        want_st.append(st)

    for k in [1, 2, 3, r - 1] + [rnd.randrange(r) for _ in range(40)]:
        pt = oracle.scalar_mul(k, oracle.G1)
        put(oracle.compress(pt), pt, gpu.DECODE_OK)
        put(oracle.compress(oracle.neg(pt)), oracle.neg(pt), gpu.DECODE_OK)      # the other root / sign flag
    put(oracle.compress(None), None, gpu.DECODE_INFINITY)
    put(b"\xc0" + b"\x00" * 46 + b"\x01", None, gpu.DECODE_BAD_ENCODING)         # infinity flag with stray bits
    put(b"\xe0" + b"\x00" * 47, None, gpu.DECODE_BAD_ENCODING)                   # infinity + sign flag
    put(b"\x00" * 48, None, gpu.DECODE_BAD_ENCODING)                             # uncompressed form
    put(bytes([0x9f]) + b"\xff" * 47, None, gpu.DECODE_BAD_ENCODING)             # x >= p
    pb = bytearray(p.to_bytes(48, "big"))
    pb[0] |= 0x80
    put(bytes(pb), None, gpu.DECODE_BAD_ENCODING)                                # x == p
    on_curve_not_g1 = 0
    x = 5
    while on_curve_not_g1 < 6:
        x += 1
        rhs = (x * x * x + 4) % p
        y = pow(rhs, (p + 1) // 4, p)
        enc = bytearray(x.to_bytes(48, "big"))
        enc[0] |= 0x80
        if y * y % p != rhs:
            put(bytes(enc), None, gpu.DECODE_NOT_ON_CURVE)
            continue
        pt = (x, y)
        assert oracle.scalar_mul(r, pt) is not None
        put(oracle.compress(pt), pt, gpu.DECODE_NOT_IN_SUBGROUP)
        on_curve_not_g1 += 1
    blob = b"".join(recs)
    got_pt, got_st = gpu.g1_decompress_batch(blob, True)
    got_pt_first = got_pt.copy()
    assert list(got_st) == want_st
    for g, w, st in zip(got_pt, want_pt, want_st):
        if st == gpu.DECODE_OK:
            assert oracle.affine_from_mont_limbs([int(v) for v in g]) == w
        else:
            assert not g.any()
    # without the subgroup test, curve points outside G1 decode to themselves
    got_pt, got_st = gpu.g1_decompress_batch(blob, False)
    for g, w, st, st0 in zip(got_pt, want_pt, got_st, want_st):
        if st0 == gpu.DECODE_NOT_IN_SUBGROUP:
            assert st == gpu.DECODE_OK and oracle.affine_from_mont_limbs([int(v) for v in g]) == w
        else:
            assert st == st0
    # agreement with the host decoder on every record
    for enc, st in zip(recs, want_st):
        if st in (gpu.DECODE_OK, gpu.DECODE_INFINITY):
            gpu.g1_decompress(enc, True)
        else:
            with pytest.raises(gpu.CurdleError):
                gpu.g1_decompress(enc, True)
    assert gpu.g1_decompress_batch(b"", True)[0].shape == (0, 12)
    # the two-step form: points and encoding / curve verdicts at once, the subgroup verdict at finish
    pts2, st2, ticket = gpu.g1_decompress_begin(blob)
    for g, w, st, st0 in zip(pts2, want_pt, st2, want_st):
        if st0 in (gpu.DECODE_OK, gpu.DECODE_NOT_IN_SUBGROUP):
            assert st == gpu.DECODE_OK and oracle.affine_from_mont_limbs([int(v) for v in g]) == w
        else:
            assert st == st0
    assert list(gpu.g1_decompress_finish(ticket, len(want_st))) == want_st
    # ... and on a batch too large for four lanes per point (the one-lane build of both
    # kernels): the same records repeated, statuses and points periodic with them
    reps = 33000 // len(recs) + 1
    big_pt, big_st, ticket = gpu.g1_decompress_begin(blob * reps)
    big_final = gpu.g1_decompress_finish(ticket, reps * len(recs))
    k = len(recs)
    assert (big_final.reshape(reps, k) == np.array(want_st, dtype=np.uint8)).all()
    assert (big_st.reshape(reps, k) == st2).all()
    assert (big_pt.reshape(reps, k, 12) == pts2).all()
    # the one-shot form at that size is the fused kernel (square root, then the subgroup test)
    one_pt, one_st = gpu.g1_decompress_batch(blob * reps, True)
    assert (one_st.reshape(reps, k) == np.array(want_st, dtype=np.uint8)).all()
    assert (one_pt.reshape(reps, k, 12) == got_pt_first).all()


def test_same_scalar_argument_is_enforced(gpu, check_mode):
    """The same-scalar argument's responses (Z_k, Z_t, Z_u) and commitments (A, B) are bound by
    two commitment equations (samescalarargument.go:83-100).  In the deferred mode those
    ride in the accumulator's MSM instead of being compared on the spot: touching any of
    them must still reject."""
    n = 64
    crs, Rs, Ss, Ts, Us, M, perm, k, rs_m = setup(gpu, n)
    proof = gpu.prove(crs, Rs, Ss, Ts, Us, M, perm, k, rs_m, gpu.Rand(3))
    assert gpu.verify(crs, proof, Rs, Ss, Ts, Us, M, gpu.Rand(4)) is True
    m = 6                                            # log2(n)
    same_perm = 48 + (48 + 32) + (96 + 4 * (4 + 48 * m) + 64)
    at = 48 + 96 + 96 + 48 + 48 + same_perm          # A, T, U, R, S, then proofSamePermutation
    assert len(proof) == at + 288 + (3 * 48 + 6 * (4 + 48 * m) + 32)
    for name, off in (("Z_k", at + 192 + 31), ("Z_t", at + 224 + 31), ("Z_u", at + 256 + 31)):
        touched = bytearray(proof)
        touched[off] ^= 1
        assert gpu.verify(crs, bytes(touched), Rs, Ss, Ts, Us, M, gpu.Rand(4)) is False, name
    # swap the argument's two commitments A and B (both valid points): rejected as well
    swapped = bytearray(proof)
    swapped[at:at + 96], swapped[at + 96:at + 192] = proof[at + 96:at + 192], proof[at:at + 96]
    assert gpu.verify(crs, bytes(swapped), Rs, Ss, Ts, Us, M, gpu.Rand(4)) is False


def test_single_bit_changes_of_a_proof_are_never_accepted(gpu, check_mode):
    """The transcript and the accumulated checks bind the whole proof: a changed bit anywhere in
    its serialisation gives a reject or a decoding error, never an accept (a stride through
    the proof here; tools/fuzz_proof_bits.py walks every byte)."""
    crs, Rs, Ss, Ts, Us, M, perm, k, rs_m = setup(gpu, 32)
    proof = gpu.prove(crs, Rs, Ss, Ts, Us, M, perm, k, rs_m, gpu.Rand(1))
    assert gpu.verify(crs, proof, Rs, Ss, Ts, Us, M, gpu.Rand(2)) is True
    rng = np.random.default_rng(5)
    for pos in range(3, len(proof), 41):
        touched = bytearray(proof)
        touched[pos] ^= 1 << int(rng.integers(0, 8))
        try:
            assert gpu.verify(crs, bytes(touched), Rs, Ss, Ts, Us, M, gpu.Rand(3)) is False, pos
        except gpu.CurdleError:
            pass
