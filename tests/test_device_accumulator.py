"""the verifier's accumulator on the GPU (CRS bases resident, scalars
built by index in an Fr kernel) against the host mirror of msmaccumulator and the Python
oracle.  Parity = the two accumulators hold the same base -> scalar map (bit-exact Fr
elements) for the same proof, instance and verifier randomness, and decide identically."""
import numpy as np
import pytest

from test_protocol_gpu import setup

pytestmark = pytest.mark.gpu


def as_map(oracle, pts, sc):
    """{base bytes: canonical scalar} with repeated bases merged; zero scalars and the point at
    infinity dropped (the reference's map keeps (0,0) as a key, the device never stores it)."""
    out = {}
    for p, s in zip(pts, sc):
        if not p.any():
            continue
        k = p.tobytes()
        out[k] = (out.get(k, 0) + oracle.fr_from_mont_limbs([int(v) for v in s])) % oracle.R
    return {k: v for k, v in out.items() if v}


@pytest.mark.parametrize("n", [64, 256])
def test_device_accumulator_holds_the_mirrors_map(gpu, oracle, n):
    crs, Rs, Ss, Ts, Us, M, perm, k, rs_m = setup(gpu, n)
    ell = n - 4
    proof = gpu.Proof(gpu.prove(crs, Rs, Ss, Ts, Us, M, perm, k, rs_m, gpu.Rand(42)))
    pm, sm, ok_m = gpu.verify_export_accumulator(crs, proof, Rs, Ss, Ts, Us, M, gpu.Rand(43), device=False)
    pd, sd, ok_d = gpu.verify_export_accumulator(crs, proof, Rs, Ss, Ts, Us, M, gpu.Rand(43), device=True)
    assert ok_m and ok_d
    mirror, device = as_map(oracle, pm, sm), as_map(oracle, pd, sd)
    # 5 ell + 8 bases of the statement (one of them is the point at infinity) + the proof's points
    assert len(pm) >= 5 * ell + 8
    assert mirror == device
    # the device's resident slots come first, in index order: Gs | Hs | H | Gt | Gu | Rs | Ss | Ts | Us
    assert (pd[ell + 7:2 * ell + 7] == Rs).all() and (pd[4 * ell + 7:5 * ell + 7] == Us).all()
    # same randomness, another instance (swapped R / S): both reject, and still hold the same map
    pm, sm, ok_m = gpu.verify_export_accumulator(crs, proof, Ss, Rs, Ts, Us, M, gpu.Rand(43), device=False)
    pd, sd, ok_d = gpu.verify_export_accumulator(crs, proof, Ss, Rs, Ts, Us, M, gpu.Rand(43), device=True)
    assert not ok_m and not ok_d
    assert as_map(oracle, pm, sm) == as_map(oracle, pd, sd)


def test_exported_map_sums_to_infinity_in_the_oracle(gpu, oracle):
    """The accumulated pairs, fed to the Python oracle's textbook MSM, give the point at infinity
    for an honest proof (A_c is the identity when every check point rides on the base side)."""
    crs, Rs, Ss, Ts, Us, M, perm, k, rs_m = setup(gpu, 16)
    proof = gpu.Proof(gpu.prove(crs, Rs, Ss, Ts, Us, M, perm, k, rs_m, gpu.Rand(3)))
    pd, sd, ok = gpu.verify_export_accumulator(crs, proof, Rs, Ss, Ts, Us, M, gpu.Rand(4), device=True)
    assert ok
    acc = oracle.INF
    for p, s in zip(pd, sd):
        if not p.any():
            continue
        pt = oracle.affine_from_mont_limbs([int(v) for v in p])
        acc = oracle.add(acc, oracle.scalar_mul(oracle.fr_from_mont_limbs([int(v) for v in s]), pt))
    assert acc is oracle.INF


def test_verify_decides_the_same_on_either_accumulator(gpu):
    crs, Rs, Ss, Ts, Us, M, perm, k, rs_m = setup(gpu, 128)
    proof = gpu.prove(crs, Rs, Ss, Ts, Us, M, perm, k, rs_m, gpu.Rand(1))
    other = gpu.Rand(77).generate_permutation(124)
    try:
        for on in (True, False):
            gpu.verify_set_device_acc(on)
            assert gpu.verify(crs, proof, Rs, Ss, Ts, Us, M, gpu.Rand(2)) is True
            assert gpu.verify(crs, proof, Ss, Rs, Ts, Us, M, gpu.Rand(2)) is False
            assert gpu.verify(crs, proof, Rs, Ss, Ts[other], Us[other], M, gpu.Rand(2)) is False
            bad = bytearray(proof)
            bad[len(bad) - 40] ^= 0x04            # a scalar of the same-multiscalar argument
            try:
                assert gpu.verify(crs, bytes(bad), Rs, Ss, Ts, Us, M, gpu.Rand(2)) is False
            except gpu.CurdleError:
                pass
    finally:
        gpu.verify_set_device_acc(True)


def test_verifying_while_decoding_classifies_mutated_proofs_like_the_two_pass_route(gpu):
    """curdle_verify's default route runs the whole transcript from the wire bytes while the GPU
    decodes (pending points filled in as bases afterwards); with the device accumulator off it
    takes the two-pass route over the host mirror.  Same accept / reject / error class for the
    honest proof and for byte flips in every part of it: point records (invalid encodings, points
    off the curve, other valid points), slice prefixes and scalars."""
    crs, Rs, Ss, Ts, Us, M, perm, k, rs_m = setup(gpu, 64)
    proof = gpu.prove(crs, Rs, Ss, Ts, Us, M, perm, k, rs_m, gpu.Rand(1))
    rng = np.random.default_rng(11)
    cases = [bytes(proof)]
    for at in rng.choice(len(proof), size=48, replace=False):
        bad = bytearray(proof)
        bad[int(at)] ^= 1 << int(rng.integers(8))
        cases.append(bytes(bad))
    cases.append(bytes(proof[:-7]))

    def outcome(pf):
        try:
            return "accept" if gpu.verify(crs, pf, Rs, Ss, Ts, Us, M, gpu.Rand(3)) else "reject"
        except gpu.CurdleError:
            return "error"

    try:
        gpu.verify_set_device_acc(True)
        lazy = [outcome(c) for c in cases]
        gpu.verify_set_device_acc(False)
        two_pass = [outcome(c) for c in cases]
    finally:
        gpu.verify_set_device_acc(True)
    assert lazy == two_pass
    assert lazy[0] == "accept" and "accept" not in lazy[1:]
    assert "error" in lazy and "reject" in lazy  # both kinds of failure were exercised


def test_device_accumulator_abi_rejects_malformed_descriptions(gpu):
    """curdle_dacc_run validates every offset of the caller's descriptions before a kernel
    reads through them."""
    import ctypes as C
    rand = gpu.Rand(5)
    pts = rand.get_g1_affines(8)
    lib = C.CDLL(gpu.LIB_PATH)
    lib.curdle_dbases_create.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]
    lib.curdle_dacc_begin.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]
    lib.curdle_dacc_run.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p,
                                    C.c_size_t, C.c_void_p, C.c_void_p]
    lib.curdle_dbases_free.argtypes = [C.c_void_p]
    bases = C.c_void_p()
    assert lib.curdle_dbases_create(pts.ctypes.data, 8, C.byref(bases)) == 0
    pool = np.stack([rand.get_fr() for _ in range(4)])
    out = np.zeros(18, dtype=np.uint64)

    def run(words):
        acc = C.c_void_p()
        assert lib.curdle_dacc_begin(bases, pts.ctypes.data, 4, C.byref(acc)) == 0
        chk = np.zeros(35, dtype=np.uint32)
        chk[:len(words)] = words
        return lib.curdle_dacc_run(acc, chk.ctypes.data, 1, pool.ctypes.data, 4, None, None, 0, out.ctypes.data, None)

    # kind, n_struct, m, q_cap, weight_off, alpha_off, gammas_off, q_off, tail_off, n_tail, nseg, seg0(set, first, len, vec_first)
    good = [1, 8, 0, 0, 1, 0, 0, 0, 0, 0, 1, 0, 0, 8, 0]
    assert run(good) == 0
    assert run([1, 8, 0, 0, 9, 0, 0, 0, 0, 0, 1, 0, 0, 8, 0]) == gpu.EINVAL       # weight outside the pool
    assert run([1, 8, 0, 0, 1, 0, 0, 0, 0, 0, 1, 0, 4, 8, 0]) == gpu.EINVAL       # segment past the set
    assert run([1, 4, 0, 0, 1, 0, 0, 0, 0, 0, 1, 0, 0, 8, 0]) == gpu.EINVAL       # segment past the vector
    assert run([2, 8, 2, 0, 1, 0, 0, 0, 0, 0, 1, 0, 0, 8, 0]) == gpu.EINVAL       # 8 structured elements, 2^2 folds
    assert run([7, 8, 0, 0, 1, 0, 0, 0, 0, 0, 1, 0, 0, 8, 0]) == gpu.EINVAL       # unknown kind

    # the two-step form: submit, poll until done, wait -- the same sum; misuse is refused
    lib.curdle_dacc_submit.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p,
                                       C.c_size_t, C.c_void_p]
    lib.curdle_dacc_poll.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
    lib.curdle_dacc_wait.argtypes = [C.c_void_p, C.c_void_p]
    lib.curdle_dacc_abort.argtypes = [C.c_void_p]
    want = out.copy()
    assert run(good) == 0 and (out == want).all() and want.any()
    chk = np.zeros(35, dtype=np.uint32)
    chk[:len(good)] = good
    acc = C.c_void_p()
    done = C.c_int(0)
    assert lib.curdle_dacc_begin(bases, pts.ctypes.data, 4, C.byref(acc)) == 0
    assert lib.curdle_dacc_poll(acc, C.byref(done)) == gpu.EINVAL                  # nothing submitted yet
    assert lib.curdle_dacc_wait(acc, out.ctypes.data) == gpu.EINVAL
    assert lib.curdle_dacc_submit(acc, chk.ctypes.data, 1, pool.ctypes.data, 4, None, None, 0, None) == 0
    assert lib.curdle_dacc_submit(acc, chk.ctypes.data, 1, pool.ctypes.data, 4, None, None, 0, None) == gpu.EINVAL  # twice
    for _ in range(100000):
        assert lib.curdle_dacc_poll(acc, C.byref(done)) == 0
        if done.value:
            break
    assert done.value == 1
    two_step = np.zeros(18, dtype=np.uint64)
    assert lib.curdle_dacc_wait(acc, two_step.ctypes.data) == 0
    assert (two_step == want).all()
    # a submitted accumulation can be dropped; a malformed submission ends it by itself
    assert lib.curdle_dacc_begin(bases, pts.ctypes.data, 4, C.byref(acc)) == 0
    assert lib.curdle_dacc_submit(acc, chk.ctypes.data, 1, pool.ctypes.data, 4, None, None, 0, None) == 0
    lib.curdle_dacc_abort(acc)
    bad = chk.copy()
    bad[4] = 9
    assert lib.curdle_dacc_begin(bases, pts.ctypes.data, 4, C.byref(acc)) == 0
    assert lib.curdle_dacc_submit(acc, bad.ctypes.data, 1, pool.ctypes.data, 4, None, None, 0, None) == gpu.EINVAL
    for _ in range(12):  # ... and no workspace slot leaked on any of those paths (there are eight)
        assert run(good) == 0
    lib.curdle_dbases_free(bases)


def test_resident_crs_survives_shutdown_and_reinit(gpu):
    """curdle_shutdown frees every device buffer, the CRS's resident copy included: the next
    verification notices (context epoch) and makes it resident again instead of reading freed
    memory."""
    crs, Rs, Ss, Ts, Us, M, perm, k, rs_m = setup(gpu, 16)
    proof = gpu.prove(crs, Rs, Ss, Ts, Us, M, perm, k, rs_m, gpu.Rand(1))
    assert gpu.verify(crs, proof, Rs, Ss, Ts, Us, M, gpu.Rand(2)) is True
    gpu.shutdown()
    gpu.init(0)
    assert gpu.verify(crs, proof, Rs, Ss, Ts, Us, M, gpu.Rand(2)) is True
    assert gpu.verify(crs, proof, Ss, Rs, Ts, Us, M, gpu.Rand(2)) is False


def test_fused_accumulator_front_at_every_pool_size(gpu, oracle):
    """Round 5: a small device accumulation runs the loose bases' conversion, the slot scalars and the recoding as
    ONE launch (k_dacc_front), the slot scalars staged through LDS; accumulations beyond 16,384 bases (the batch
    verifiers' groups) keep the four separate operations.  (Round 6: the knob that forced the separate form on small
    ones is gone; the fused form is held against the HOST mirror's map here, the separate one by the batch tests.)
    n = 16 has a pool of a few dozen elements, n = 256 the verifier's ~1,000."""
    for n in (16, 64, 256):
        crs, Rs, Ss, Ts, Us, M, perm, k, rs_m = setup(gpu, n)
        proof = gpu.Proof(gpu.prove(crs, Rs, Ss, Ts, Us, M, perm, k, rs_m, gpu.Rand(42)))
        for inst, want in (((Rs, Ss), True), ((Ss, Rs), False)):
            pm, sm, ok_m = gpu.verify_export_accumulator(crs, proof, inst[0], inst[1], Ts, Us, M, gpu.Rand(43), device=False)
            pd, sd, ok_d = gpu.verify_export_accumulator(crs, proof, inst[0], inst[1], Ts, Us, M, gpu.Rand(43), device=True)
            assert ok_m is want and ok_d is want, (n, want)
            assert as_map(oracle, pm, sm) == as_map(oracle, pd, sd), (n, want)
