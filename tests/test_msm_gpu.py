"""Parity tests proper: the HIP path, called through the C ABI, against the
oracle -- golden vectors, seeded random inputs at sizes the C oracle finishes in
seconds, and size-independent properties at BASELINE.json's full sizes.
Bit-exact throughout (integer arithmetic; results compared as the canonical
Jacobian representative)."""
import os

import numpy as np
import pytest

from conftest import golden_case_names

pytestmark = pytest.mark.gpu


def rand_scalars(rng, n, oracle):
    """n uniform Montgomery-form fr.Elements (any value < r is a valid one)."""
    out = np.zeros((n, 4), dtype=np.uint64)
    have = 0
    top = oracle.R >> 192
    while have < n:
        cand = rng.integers(0, 1 << 64, size=(n - have + 16, 4), dtype=np.uint64)
        cand[:, 3] &= np.uint64((1 << 63) - 1)
        ok = cand[:, 3] < np.uint64(top)          # strictly below r's top limb: < r for sure
        cand = cand[ok][: n - have]
        out[have:have + len(cand)] = cand
        have += len(cand)
    return out


def canonical_ints(sc, oracle):
    return [oracle.fr_from_mont_limbs([int(v) for v in row]) for row in sc]


# ------------------------------------------------------------------ primitives ---
def test_device_primitives_match_host(gpu, oracle):
    """Same limb code on the GPU and on the host (host checked against the oracle in
    tests/test_host_mirror.py), operation by operation, exceptional cases included."""
    from test_host_mirror import fp32, group_cases, group_inputs
    rng = np.random.default_rng(5)
    n = 4096
    A = [int.from_bytes(rng.bytes(48), "big") % oracle.P for _ in range(n)]
    B = [int.from_bytes(rng.bytes(48), "big") % oracle.P for _ in range(n)]
    A[0], B[1], A[2], B[2], A[3], B[3] = 0, 0, oracle.P - 1, oracle.P - 1, 1, oracle.P - 1
    inp = np.array([fp32(a) + fp32(b) for a, b in zip(A, B)], dtype=np.uint32)
    for op in (0, 1, 2, 3):
        assert (gpu.selftest_op(op, inp, True) == gpu.selftest_op(op, inp, False)).all(), op
    # spot-check the device result against big-integer arithmetic directly
    out = gpu.selftest_op(0, inp[:64], True)
    for i in range(64):
        assert sum(int(x) << (32 * k) for k, x in enumerate(out[i])) == A[i] * B[i] * oracle.R_FP_INV % oracle.P
    S = rng.integers(0, 1 << 32, size=(n, 16), dtype=np.uint32)
    S[:, 7] &= 0x3FFFFFFF
    assert (gpu.selftest_op(4, S, True) == gpu.selftest_op(4, S, False)).all()
    cases = group_cases(oracle)
    for op in (5, 6, 7):
        arr = group_inputs(oracle, op, cases)
        assert (gpu.selftest_op(op, arr, True) == gpu.selftest_op(op, arr, False)).all(), op
    # the lane-distributed ("quad") point operations of the latency-bound kernels (quad28.h):
    # same group elements as the single-lane formulas (coordinates may differ by the
    # projective scale only where the exceptional branches differ, so compare affine points)
    from test_host_mirror import xyzz_to_affine
    cases = cases * 7                                   # several quads per wave, odd count
    for op, host_op in ((8, 6), (9, 7), (10, 10)):
        arr = group_inputs(oracle, 6, cases)
        dev, host = gpu.selftest_op(op, arr, True), gpu.selftest_op(host_op, arr, False)
        for i in range(len(cases)):
            assert xyzz_to_affine(oracle, dev[i]) == xyzz_to_affine(oracle, host[i]), (op, i)


def test_bases_enter_the_msm_curve_and_come_back(gpu, oracle):
    """The MSM kernels run on the image of the curve under (x, y) -> (x / 16, y / 64): a base enters by shifts
    of its gnark words and conditional subtractions (fp28.h from_gnark_iso_*), a result leaves through 2^388 /
    2^390 (to_gnark_msm).  Operation 12 runs the two ends on plain field elements: the round trip is the
    identity and the values in between respect madd's bounds (normalised limbs, below 2p) -- at every boundary
    of the conditional subtractions (x = k p / 16, y = k p / 4, either side) and on random elements."""
    P = oracle.P
    vals = [0, 1, 2, P - 1, P - 2, (P - 1) // 2]
    for k in range(1, 16):
        for d in (-2, -1, 0, 1, 2):
            vals.append((k * P // 16 + d) % P)
    rng = np.random.default_rng(44)
    vals += [int.from_bytes(rng.bytes(48), "big") % P for _ in range(4000)]
    from test_host_mirror import fp32
    # the words a caller passes are Montgomery forms: every canonical word pattern below p occurs, so the
    # patterns themselves are what is swept
    inp = np.array([fp32(v) + fp32(vals[-1 - i]) for i, v in enumerate(vals)], dtype=np.uint32)
    out = gpu.selftest_op(12, inp, True)
    assert (out[:, :24] == inp).all()
    assert (out[:, 24:] == 1).all()
    assert (out == gpu.selftest_op(12, inp, False)).all()


def test_the_split_on_the_device(gpu, oracle):
    """The device build of glv_split (what k_digits runs) on the same 300,000 random and boundary
    scalars as the host build (tests/test_abi.py), against big-integer division, and word for
    word equal to the host build."""
    from test_abi import check_glv_split_outputs, glv_split_cases, scalars_as_words
    vals = glv_split_cases(oracle.R, 300_000, 12)
    words = scalars_as_words(vals)
    dev = gpu.selftest_op(11, words, True)
    check_glv_split_outputs(vals, dev, oracle.R)
    assert (dev == gpu.selftest_op(11, words, False)).all()


# -------------------------------------------------------------- golden vectors ---
def test_published_double_of_the_generator_on_the_gpu(gpu, oracle):
    """EIP-2537's published 2 * G1 (tests/test_oracle.py) out of the GPU's MSM: as 2 * G, as G + G
    (two terms in one bucket: the accumulation's doubling branch), as (r - 1) * (-2 G) ... every
    route through the split, the buckets and the host's Horner pass must land on the published
    coordinates."""
    from test_oracle import EIP2537_2G_X, EIP2537_2G_Y
    two_g = (EIP2537_2G_X, EIP2537_2G_Y)
    want = np.array(oracle.jac_to_mont_limbs(two_g), dtype=np.uint64)
    g = np.array([oracle.affine_to_mont_limbs(oracle.G1)], dtype=np.uint64)
    fr = lambda v: oracle.fr_to_mont_limbs(v % oracle.R)
    assert (gpu.msm_g1(g, np.array([fr(2)], dtype=np.uint64)) == want).all()
    assert (gpu.msm_g1(np.concatenate([g, g]), np.array([fr(1), fr(1)], dtype=np.uint64)) == want).all()
    neg_g = np.array([oracle.affine_to_mont_limbs(oracle.neg(oracle.G1))], dtype=np.uint64)
    assert (gpu.msm_g1(neg_g, np.array([fr(oracle.R - 2)], dtype=np.uint64)) == want).all()
    # gnark's own spelling of the scalar one (fr SetOne) is taken as 1, and the result's Z is gnark's fp one
    from test_oracle import GNARK_FP_ONE, GNARK_FR_ONE
    one_g = gpu.msm_g1(g, np.array([GNARK_FR_ONE], dtype=np.uint64))
    assert [int(v) for v in one_g] == oracle.jac_to_mont_limbs(oracle.G1) and [int(v) for v in one_g[12:]] == GNARK_FP_ONE
    # ... and from 1,000 terms whose scalars sum to 2 (mod r)
    rng = np.random.default_rng(2537)
    sc = [int.from_bytes(rng.bytes(32), "big") % oracle.R for _ in range(999)]
    sc.append((2 - sum(sc)) % oracle.R)
    assert (gpu.msm_g1(np.repeat(g, 1000, axis=0), np.array([fr(v) for v in sc], dtype=np.uint64)) == want).all()


def test_golden_vectors_bit_exact(gpu, golden):
    names = golden_case_names(golden)
    assert len(names) >= 17
    for name in names:
        got = gpu.msm_g1(golden[name + "_points"], golden[name + "_scalars"])
        assert (got == golden[name + "_expected"]).all(), name


def test_golden_vectors_every_window_size(gpu, golden):
    """Each supported window width c (4..16), including the widths that need an
    extra top window, on vectors that exercise carries and exceptional additions."""
    try:
        for c in range(4, 17):
            gpu.plan_override("WINDOW_BITS", c)
            for name in ("rand0_n16", "rand0_n257", "edge_window_boundaries", "edge_extreme_scalars",
                         "edge_duplicate_bases", "edge_opposite_points", "edge_infinity_bases"):
                got = gpu.msm_g1(golden[name + "_points"], golden[name + "_scalars"])
                assert (got == golden[name + "_expected"]).all(), (name, c)
    finally:
        gpu.plan_override("WINDOW_BITS", None)


def test_segment_lengths(gpu, golden):
    """The accumulate kernel's lane segment length must not change results (fragment
    bookkeeping at every alignment)."""
    try:
        for L in (4, 5, 8, 9, 13, 32, 128):
            gpu.plan_override("SEG_LEN", L)
            for name in ("rand0_n257", "rand0_n1024", "edge_all_equal_scalars", "edge_small_scalars"):
                got = gpu.msm_g1(golden[name + "_points"], golden[name + "_scalars"])
                assert (got == golden[name + "_expected"]).all(), (name, L)
    finally:
        gpu.plan_override("SEG_LEN", None)


def test_reduce_segment_lengths(gpu, golden, oracle, coracle):
    """The bucket reduction walks segments of 1..64 buckets per quad (the plan picks the length
    from the call's size); every length must give the same bits, including on inputs that reach
    the doubling / infinity branches inside the reduction."""
    k, q = oracle.Rand(15).get_frs(2)
    n = 3000
    pts = coracle.points_walk(k, q, n)
    sc = rand_scalars(np.random.default_rng(15), n, oracle)
    exp = coracle.msm_pippenger(pts, sc, threads=8)
    try:
        for quad in ("1", "4", "32", "64"):
            gpu.plan_override("REDUCE_SEG", int(quad))
            for name in ("rand0_n16", "rand0_n257", "rand0_n1024", "edge_duplicate_bases", "edge_opposite_points",
                         "edge_cancels_to_infinity", "edge_all_equal_scalars", "edge_small_scalars",
                         "edge_window_boundaries", "edge_all_infinity"):
                got = gpu.msm_g1(golden[name + "_points"], golden[name + "_scalars"])
                assert (got == golden[name + "_expected"]).all(), (name, quad)
            for c in (6, 11, 16):
                gpu.plan_override("WINDOW_BITS", c)
                assert (gpu.msm_g1(pts, sc) == exp).all(), (quad, c)
            gpu.plan_override("WINDOW_BITS", None)
    finally:
        gpu.plan_override("REDUCE_SEG", None)
        gpu.plan_override("WINDOW_BITS", None)


# ------------------------------------------------------- seeded random vs C oracle ---
@pytest.mark.parametrize("n", [5, 6, 7, 8, 9, 60, 64, 124, 128, 252, 256, 308, 628, 1268, 2548, 1 << 12, 1 << 14])
def test_random_inputs_match_c_oracle(gpu, oracle, coracle, n):
    # protocol sizes of SURVEY.md 8(d): m = 6..9, ell, n = ell+4, 5*ell+8
    k, q = oracle.Rand(1).get_frs(2)
    pts = coracle.points_walk(k, q + n, n)
    sc = rand_scalars(np.random.default_rng(n), n, oracle)
    got = gpu.msm_g1(pts, sc)
    exp = coracle.msm_pippenger(pts, sc, threads=8)
    assert (got == exp).all()
    assert (gpu.msm_g1(pts, sc) == got).all()   # run-to-run identical (scatter order is not)


def test_skewed_scalar_sets_match_c_oracle(gpu, oracle, coracle):
    """All-equal scalars (a11), <= 9-bit scalars (a13), 1 % infinity bases, one hot bucket."""
    n = 4096
    k, q = oracle.Rand(2).get_frs(2)
    pts = coracle.points_walk(k, q, n)
    rng = np.random.default_rng(77)
    beta = np.array(oracle.fr_to_mont_limbs(oracle.Rand(4).get_fr()), dtype=np.uint64)
    sets = {
        "all_equal": np.tile(beta, (n, 1)),
        "small": np.array([oracle.fr_to_mont_limbs(int(v)) for v in rng.integers(0, 508, n)], dtype=np.uint64),
        "one_hot_window": np.array([oracle.fr_to_mont_limbs((int(v) << 64) | 0x1234) for v in rng.integers(0, 1 << 60, n)], dtype=np.uint64),
    }
    for name, sc in sets.items():
        assert (gpu.msm_g1(pts, sc) == coracle.msm_pippenger(pts, sc, threads=8)).all(), name
    p2 = pts.copy()
    p2[rng.random(n) < 0.01] = 0
    sc = rand_scalars(rng, n, oracle)
    assert (gpu.msm_g1(p2, sc) == coracle.msm_pippenger(p2, sc, threads=8)).all()


def test_randomised_differential_small_cases(gpu, oracle, coracle):
    """150 random small MSMs with random structure -- duplicated bases, negated twins,
    infinity bases, zero / tiny / huge scalars, repeated scalars -- under random window
    widths and segment lengths, against the C oracle.  Small cases make buckets collide,
    which is what reaches the doubling and cancellation branches of the group law inside
    accumulate, merge and reduce."""
    rng = np.random.default_rng(2024)
    k, q = oracle.Rand(16).get_frs(2)
    pool = coracle.points_walk(k, q, 64)
    neg = pool.copy()
    for i in range(len(neg)):                     # -P: y -> p - y on the Montgomery limbs
        y = oracle._from_limbs(neg[i, 6:12])
        neg[i, 6:12] = oracle._limbs((oracle.P - y) % oracle.P, 6)
    special = [0, 1, 2, oracle.R - 1, oracle.R - 2, (1 << 128), (1 << 200) + 5, 77]
    try:
        for case in range(150):
            n = int(rng.integers(1, 48))
            idx = rng.integers(0, 12 if case % 3 == 0 else 64, n)      # few distinct bases -> collisions
            pts = pool[idx].copy()
            flip = rng.random(n) < 0.25
            pts[flip] = neg[idx[flip]]
            pts[rng.random(n) < 0.08] = 0                                # infinity bases
            sc_int = []
            for _ in range(n):
                r = rng.random()
                if r < 0.3:
                    sc_int.append(int(special[int(rng.integers(0, len(special)))]))
                elif r < 0.5:
                    sc_int.append(int(rng.integers(0, 1 << 20)))
                elif r < 0.6 and sc_int:
                    sc_int.append(sc_int[-1])
                else:
                    sc_int.append(int.from_bytes(rng.bytes(32), "big") % oracle.R)
            sc = np.array([oracle.fr_to_mont_limbs(v) for v in sc_int], dtype=np.uint64)
            c_case, L_case = int(rng.integers(4, 17)), int(rng.integers(8, 40))
            gpu.plan_override("WINDOW_BITS", c_case)
            gpu.plan_override("SEG_LEN", L_case)
            gpu.plan_override("REDUCE_SEG", 1 << (case % 6))
            got = gpu.msm_g1(pts, sc)
            exp = coracle.msm_naive(pts, sc)
            assert (got == exp).all(), (case, n, c_case, L_case)
    finally:
        for v in ("WINDOW_BITS", "SEG_LEN", "REDUCE_SEG"):
            gpu.plan_override(v, None)


# ---------------------------------------------------- full sizes via properties ---
def _walk_expected(oracle, coracle, k, q, sc):
    """P_i = (k + i q) G  =>  MSM = (k * sum s_i + q * sum i s_i) G  (SURVEY.md 8d)."""
    s = canonical_ints(sc, oracle)
    s0 = sum(s) % oracle.R
    s1 = sum(i * v for i, v in enumerate(s)) % oracle.R
    e = (k * s0 + q * s1) % oracle.R
    aff = coracle.scalar_mul_gen(e)
    pt = oracle.affine_from_mont_limbs([int(v) for v in aff])
    return np.array(oracle.jac_to_mont_limbs(pt), dtype=np.uint64)


@pytest.mark.parametrize("logn", [10, 16, 20])
def test_full_size_known_discrete_log(gpu, oracle, coracle, logn):
    """BASELINE configs[1] (N = 2^16) and the headline N = 2^20, bit-exact without a
    2^20-point oracle.  Inputs generated on the GPU and resident in HBM."""
    import torch
    n = 1 << logn
    k, q = oracle.Rand(1).get_frs(2)
    d_pts = torch.empty((n, 12), dtype=torch.int64, device="cuda:0")
    gpu.synth_points_walk_device(k, q, n, d_pts.data_ptr())
    head = d_pts[:64].cpu().numpy().view(np.uint64)
    assert (head == coracle.points_walk(k, q, 64)).all()      # generator itself vs the oracle
    sc = rand_scalars(np.random.default_rng(logn), n, oracle)
    d_sc = torch.from_numpy(sc.view(np.int64)).to("cuda:0")
    got = gpu.msm_g1_device(d_pts.data_ptr(), d_sc.data_ptr(), n)
    assert (got == _walk_expected(oracle, coracle, k, q, sc)).all()
    # host-buffer entry point, same inputs (at 2^20 it goes to the GPU in point-range chunks,
    # each copied while the one before it is accumulated)
    assert (gpu.msm_g1(d_pts.cpu().numpy().view(np.uint64), sc) == got).all()


def test_host_buffers_in_uneven_chunks(gpu, oracle, coracle):
    """curdle_msm_g1 sends a large call to the GPU in point-range chunks (run_host_chunked): an odd
    size cut into 3 and into 5 chunks, the last one short, against the device-resident call."""
    import torch
    n = (1 << 19) + 5
    k, q = oracle.Rand(1).get_frs(2)
    d_pts = torch.empty((n, 12), dtype=torch.int64, device="cuda:0")
    gpu.synth_points_walk_device(k, q, n, d_pts.data_ptr())
    sc = rand_scalars(np.random.default_rng(19), n, oracle)
    d_sc = torch.from_numpy(sc.view(np.int64)).to("cuda:0")
    exp = gpu.msm_g1_device(d_pts.data_ptr(), d_sc.data_ptr(), n)
    pts = d_pts.cpu().numpy().view(np.uint64)
    try:
        for chunks in ("3", "5"):
            gpu.plan_override("HOST_CHUNKS", int(chunks))
            assert (gpu.msm_g1(pts, sc) == exp).all(), chunks
    finally:
        gpu.plan_override("HOST_CHUNKS", None)


def test_chunked_host_call_folds_fragments_also_for_skewed_scalars(gpu, oracle, coracle):
    """The chunks of a host-buffer call fold their fragments into one running sum per bucket as they finish
    (k_fold_fragments), and the call's one reduction reads the sums beside the last chunk's fragments.  With
    uniform scalars, with scalars that all fall into a handful of buckets (every chunk takes the k_merge_large
    detour and the fold reads ONE pre-merged fragment for those buckets), with a
    chunk whose scalars are all zero (no fragments at all), with folding switched off (HOST_FOLD=0: the
    reduction walks every chunk's list): the same bits as the device-resident call and the closed form."""
    import torch
    n = (1 << 19) + 77
    k, q = oracle.Rand(3).get_frs(2)
    d_pts = torch.empty((n, 12), dtype=torch.int64, device="cuda:0")
    gpu.synth_points_walk_device(k, q, n, d_pts.data_ptr())
    pts = d_pts.cpu().numpy().view(np.uint64)
    rng = np.random.default_rng(77)
    uniform = rand_scalars(rng, n, oracle)
    few = np.array([oracle.fr_to_mont_limbs(v) for v in (5, 5 + (1 << 40), oracle.R - 3, 12345678901234567890)], dtype=np.uint64)
    skewed = few[rng.integers(0, len(few), size=n)]
    holed = uniform.copy()
    holed[n // 4: n // 2] = 0                                     # the second of four chunks contributes nothing
    try:
        for name, sc in (("uniform", uniform), ("skewed", skewed), ("holed", holed)):
            d_sc = torch.from_numpy(sc.view(np.int64)).to("cuda:0")
            exp = gpu.msm_g1_device(d_pts.data_ptr(), d_sc.data_ptr(), n)
            if name != "holed":
                assert (exp == _walk_expected(oracle, coracle, k, q, sc)).all(), name
            for chunks, fold in ((4, None), (2, None), (6, None), (4, 0)):
                gpu.plan_override("HOST_CHUNKS", chunks)
                gpu.plan_override("HOST_FOLD", fold)
                assert (gpu.msm_g1(pts, sc) == exp).all(), (name, chunks, fold)
    finally:
        gpu.plan_override("HOST_CHUNKS", None)
        gpu.plan_override("HOST_FOLD", None)


def test_scalars_at_the_boundaries_of_the_split(gpu, oracle, coracle):
    """k_digits splits every scalar as k = +-(k1 + k2 lambda) with a Barrett division
    (bls12_381.h glv_split): the scalars where its branches flip -- around (r - 1) / 2,
    multiples of lambda and lambda / 2 either side (remainder 0, rounding boundary, the
    quotient's correction steps), 0, 1, r - 1, powers of two -- each on its own point and all
    together, against the C oracle."""
    from test_abi import GLV_LAMBDA as lam
    R = oracle.R
    half = (R - 1) // 2
    vals = [0, 1, 2, R - 1, R - 2, half, half + 1, half - 1, lam, lam - 1, lam + 1, lam >> 1, (lam >> 1) + 1, (lam >> 1) - 1,
            R - lam, R - lam - 1, R - lam + 1, half - (half % lam), half - (half % lam) + (lam >> 1),
            half - (half % lam) + (lam >> 1) + 1, half - (half % lam) - 1]
    vals += [(j * lam + d) % R for j in (2, 3, lam >> 1, (lam >> 1) - 1, lam - 1) for d in (-1, 0, 1, lam >> 1, (lam >> 1) + 1)]
    vals += [1 << b for b in (31, 32, 63, 64, 126, 127, 128, 191, 192, 253, 254)]
    vals = [v % R for v in vals]
    k, q = oracle.Rand(21).get_frs(2)
    pts = coracle.points_walk(k, q, len(vals))
    sc = np.array([oracle.fr_to_mont_limbs(v) for v in vals], dtype=np.uint64)
    assert (gpu.msm_g1(pts, sc) == coracle.msm_naive(pts, sc)).all()
    for i in range(len(vals)):          # one at a time: a wrong half cannot cancel against another
        assert (gpu.msm_g1(pts[i:i + 1], sc[i:i + 1]) == coracle.msm_naive(pts[i:i + 1], sc[i:i + 1])).all(), hex(vals[i])


def test_bases_outside_the_prime_order_subgroup_are_a_documented_precondition(gpu, oracle):
    """Every scalar goes through the GLV split k P = k1 P + k2 phi(P), phi(x, y) = (beta x, y),
    which equals [lambda] only on the prime-order subgroup.  gnark's MultiExp does not use the
    endomorphism, so for a curve point OUTSIDE G1 this library's result differs from k P --
    include/curdle_msm.h states the precondition (bases in G1, which every decoded proof point and
    CRS point is: gnark's Decoder / SetBytes check it).  This pins what happens instead: the
    result is exactly k1 P + k2 (beta x, y), a point on the curve, never a fault."""
    from test_abi import GLV_LAMBDA, glv_split
    p, R = oracle.P, oracle.R
    # beta: the cube root of unity with (beta x, y) = [lambda] (x, y) on G1
    s3 = pow(p - 3, (p + 1) // 4, p)
    assert s3 * s3 % p == p - 3
    betas = [(-1 + s3) * pow(2, -1, p) % p, (-1 - s3) * pow(2, -1, p) % p]
    gx, gy = oracle.G1
    beta = [b for b in betas if oracle.scalar_mul(GLV_LAMBDA, oracle.G1) == (b * gx % p, gy)]
    assert len(beta) == 1
    beta = beta[0]
    x = 6
    while True:
        rhs = (x * x * x + 4) % p
        y = pow(rhs, (p + 1) // 4, p)
        if y * y % p == rhs and oracle.scalar_mul(R, (x, y)) is not None:
            break
        x += 1
    P = (x, y)                                                   # on the curve, not in G1
    for k in (5, GLV_LAMBDA + 1, 0x1234567890ABCDEF1234567890ABCDEF1234567890ABCDEF % R, R - 2):
        pts = np.array([oracle.affine_to_mont_limbs(P)], dtype=np.uint64)
        sc = np.array([oracle.fr_to_mont_limbs(k)], dtype=np.uint64)
        got = gpu.msm_g1(pts, sc)
        k1, k2 = glv_split(k, R)
        phiP = (beta * x % p, y)
        neg = lambda pt: (pt[0], (p - pt[1]) % p)
        t1 = oracle.scalar_mul(abs(k1), P if k1 >= 0 else neg(P))
        t2 = oracle.scalar_mul(abs(k2), phiP if k2 >= 0 else neg(phiP))
        want = oracle.add(t1, t2)
        assert [int(v) for v in got] == oracle.jac_to_mont_limbs(want), hex(k)
        if k > 5:
            assert want != oracle.scalar_mul(k, P)               # ... which is NOT k P here
    # in G1 the same construction IS k P (the identity the kernels rely on)
    Q = oracle.scalar_mul(12345, oracle.G1)
    k = R - 2
    got = gpu.msm_g1(np.array([oracle.affine_to_mont_limbs(Q)], dtype=np.uint64), np.array([oracle.fr_to_mont_limbs(k)], dtype=np.uint64))
    assert [int(v) for v in got] == oracle.jac_to_mont_limbs(oracle.scalar_mul(k, Q))


def test_sizes_at_the_steps_of_the_plan_tables(gpu, oracle, coracle):
    """The window width, the reduce segments and the positions per accumulate lane are stepwise
    rules of n (make_plan, choose_window_bits): both sides of every step, synchronous and
    pipelined, against the closed form -- prefixes of one walk, so one set of inputs serves."""
    import torch
    sizes = [299, 300, 2000, 2001, 3000, 3001, 45000, 45001, 65536, 100000, 100001, 200001, 450001]
    nmax = max(sizes)
    k, q = oracle.Rand(1).get_frs(2)
    d_pts = torch.empty((nmax, 12), dtype=torch.int64, device="cuda:0")
    gpu.synth_points_walk_device(k, q, nmax, d_pts.data_ptr())
    sc = rand_scalars(np.random.default_rng(77), nmax, oracle)
    d_sc = torch.from_numpy(sc.view(np.int64)).to("cuda:0")
    s = canonical_ints(sc, oracle)
    widths = set()
    s0 = s1 = 0
    done = 0
    for n in sizes:
        for i in range(done, n):
            s0 += s[i]
            s1 += i * s[i]
        done = n
        aff = coracle.scalar_mul_gen((k * s0 + q * s1) % oracle.R)
        exp = np.array(oracle.jac_to_mont_limbs(oracle.affine_from_mont_limbs([int(v) for v in aff])), dtype=np.uint64)
        widths.add(gpu.window_bits(n))
        assert (gpu.msm_g1_device(d_pts.data_ptr(), d_sc.data_ptr(), n) == exp).all(), n
        tickets = [gpu.msm_g1_device_submit(d_pts.data_ptr(), d_sc.data_ptr(), n) for _ in range(3)]
        for t in tickets:
            assert (gpu.msm_wait(t) == exp).all(), n
    assert len(widths) >= 7      # the table's steps were really crossed


def test_linearity_at_full_size(gpu, oracle):
    """MSM(P, a) + MSM(P, b) == MSM(P, a + b) at N = 2^18 (size-independent property)."""
    import torch
    n = 1 << 18
    k, q = oracle.Rand(6).get_frs(2)
    d_pts = torch.empty((n, 12), dtype=torch.int64, device="cuda:0")
    gpu.synth_points_walk_device(k, q, n, d_pts.data_ptr())
    rng = np.random.default_rng(1)
    # canonical small-ish integers in Montgomery form are not needed: linearity holds
    # on the Montgomery values too because x -> x*R^-1 is linear mod r.
    a = rand_scalars(rng, n, oracle)
    b = rand_scalars(rng, n, oracle)
    R = oracle.R
    ab = np.array([oracle._limbs((oracle._from_limbs(x) + oracle._from_limbs(y)) % R, 4) for x, y in zip(a, b)], dtype=np.uint64)
    res = []
    for sc in (a, b, ab):
        d_sc = torch.from_numpy(sc.view(np.int64)).to("cuda:0")
        res.append(gpu.msm_g1_device(d_pts.data_ptr(), d_sc.data_ptr(), n))
    assert (gpu.g1_sum(np.stack(res[:2])) == res[2]).all()


def test_two_pass_scatter_with_skewed_scalars(gpu, oracle, coracle):
    """Single MSMs from ~200,000 pairs on sort their terms in two passes (coarse bins of 128
    buckets, then buckets: msm_sort_kernels.hip k_scatter_coarse / k_scatter_fine).  Uniform scalars
    fill every bin evenly; these do not: all-equal scalars (one bucket per window holds every
    term), scalars below 300 (one window, a few bins), a hot top window, 1 % infinity bases, and
    an odd size -- each against the closed form of the known-discrete-log inputs, for the window
    widths 15 and 16."""
    import torch
    n = (1 << 18) + 77
    k, q = oracle.Rand(1).get_frs(2)
    d_pts = torch.empty((n, 12), dtype=torch.int64, device="cuda:0")
    gpu.synth_points_walk_device(k, q, n, d_pts.data_ptr())
    rng = np.random.default_rng(18)
    uniform = rand_scalars(rng, n, oracle)
    fams = {"uniform": uniform}

    def from_ints(vals):
        v = np.asarray([oracle.fr_to_mont_limbs(int(x) % oracle.R) for x in vals], dtype=np.uint64)
        return v

    fams["all_equal"] = np.tile(from_ints([123456789123456789123456789])[0], (n, 1))
    small = np.tile(from_ints(range(300)), ((n + 299) // 300, 1))[:n].copy()
    fams["small"] = small
    hot = uniform.copy()
    hot[: n // 2] = np.tile(from_ints([(7 << 250) % oracle.R])[0], (n // 2, 1))      # half the terms share every digit
    fams["half_equal"] = hot
    for name, sc in fams.items():
        d_sc = torch.from_numpy(sc.view(np.int64)).to("cuda:0")
        exp = _walk_expected(oracle, coracle, k, q, sc)
        for c in (0, 15, 16):
            got = gpu.msm_g1_device(d_pts.data_ptr(), d_sc.data_ptr(), n, window_bits=c)
            assert (got == exp).all(), (name, c)
    # infinity bases: their terms are sorted like any other and skipped by the accumulation
    pts = d_pts.cpu().numpy().view(np.uint64).copy()
    dead = rng.integers(0, n, n // 100)
    pts[dead] = 0
    sc = uniform.copy()
    sc_dead = sc.copy()
    sc_dead[dead] = 0                                            # same sum: a zero scalar on the original point
    d_p2 = torch.from_numpy(pts.view(np.int64)).to("cuda:0")
    got = gpu.msm_g1_device(d_p2.data_ptr(), torch.from_numpy(sc.view(np.int64)).to("cuda:0").data_ptr(), n)
    assert (got == _walk_expected(oracle, coracle, k, q, sc_dead)).all()


def test_merge_of_large_buckets_in_every_form(gpu, oracle, coracle):
    """k_merge_large (round 6): buckets with more than 16 fragments are summed by waves of 16 quads, a chunk of 32 to
    256 fragments each, and the wave that counts a bucket's last chunk adds the chunk sums.  Every form of it against
    the closed form of the known-discrete-log inputs: a queue far longer than one pass over it (1,024 entries) made of
    one-chunk buckets; buckets of ~1,000, ~4,000, ~8,000 and ~33,000 fragments (the four chunk sizes, the counter, the
    second stage); the same through the chunked host-buffer path; and three base sets sharing one fragment
    bookkeeping.  The bucket-slot walk of k_accumulate over thousands of EMPTY slots (all-equal scalars leave two
    occupied buckets per window) is the other thing these inputs reach."""
    import torch
    k, q = oracle.Rand(1).get_frs(2)
    n = (1 << 19) + 5
    d_pts = torch.empty((n, 12), dtype=torch.int64, device="cuda:0")
    gpu.synth_points_walk_device(k, q, n, d_pts.data_ptr())
    rng = np.random.default_rng(66)
    try:
        # (a) ~12,000 queued buckets of 30-70 fragments each
        m = 1 << 15
        table = rand_scalars(rng, 1024, oracle)
        sc = table[rng.integers(0, 1024, m)]
        d_sc = torch.from_numpy(sc.view(np.int64)).to("cuda:0")
        gpu.plan_override("SEG_LEN", 2)
        assert (gpu.msm_g1_device(d_pts.data_ptr(), d_sc.data_ptr(), m) == _walk_expected(oracle, coracle, k, q, sc)).all()
        gpu.plan_override("SEG_LEN", None)
        # (b) two occupied buckets per window: 2^19 / L fragments each
        beta = np.array(oracle.fr_to_mont_limbs(oracle.Rand(4).get_fr()), dtype=np.uint64)
        sc = np.tile(beta, (n, 1))
        d_sc = torch.from_numpy(sc.view(np.int64)).to("cuda:0")
        exp = _walk_expected(oracle, coracle, k, q, sc)
        for L in (None, 16, 128):
            gpu.plan_override("SEG_LEN", L)
            assert (gpu.msm_g1_device(d_pts.data_ptr(), d_sc.data_ptr(), n) == exp).all(), L
        gpu.plan_override("SEG_LEN", None)
        assert (gpu.msm_g1(d_pts.cpu().numpy().view(np.uint64), sc) == exp).all()          # four chunks, each merged, then folded
        t = gpu.msm_g1_device_submit(d_pts.data_ptr(), d_sc.data_ptr(), n)                   # the pipelined plan
        assert (gpu.msm_wait(t) == exp).all()
        m = 1 << 14
        assert (gpu.msm_g1_device(d_pts.data_ptr(), d_sc.data_ptr(), m) == _walk_expected(oracle, coracle, k, q, sc[:m])).all()
        # (b') a few hundred distinct values: every occupied bucket holds 9..16 fragments -- over the plan's merge limit at this
        # size (8), under the chunks' (16: a chunk of a host-buffer call keeps its mid-size buckets for the fold).  The limit
        # must be the same in a chunk's sort step and in its accumulate step, with two chunks as with four: a bucket merged
        # under one limit and read under the other would count twice.
        table = rand_scalars(rng, 700, oracle)
        sc = table[rng.integers(0, 700, n)]
        d_sc = torch.from_numpy(sc.view(np.int64)).to("cuda:0")
        exp = _walk_expected(oracle, coracle, k, q, sc)
        assert (gpu.msm_g1_device(d_pts.data_ptr(), d_sc.data_ptr(), n) == exp).all()
        pts_h = d_pts.cpu().numpy().view(np.uint64)
        for chunks in (None, 2, 3):
            gpu.plan_override("HOST_CHUNKS", chunks)
            assert (gpu.msm_g1(pts_h, sc) == exp).all(), chunks
        gpu.plan_override("HOST_CHUNKS", None)
        gpu.plan_override("HOST_FOLD", 0)
        assert (gpu.msm_g1(pts_h, sc) == exp).all()
        gpu.plan_override("HOST_FOLD", None)
        sc = np.tile(beta, (n, 1))
        # (c) three base sets, one scalar vector, buckets of several chunks
        m = 40000
        sets = [coracle.points_walk(k + 5 * j, q, m) for j in range(3)]
        out = gpu.msm_g1_multi(sets, sc[:m])
        for j in range(3):
            assert (out[j] == coracle.msm_pippenger(sets[j], np.ascontiguousarray(sc[:m]), threads=8)).all(), j
    finally:
        for knob in ("SEG_LEN", "HOST_CHUNKS", "HOST_FOLD"):
            gpu.plan_override(knob, None)


# ------------------------------------------------------------- window partition ---
def test_window_partials_sum_to_full_msm(gpu, oracle, coracle):
    """The multi-GPU split: partials over a partition of the windows, summed, equal
    the full MSM -- for 1, 2, 4, 8 ranks and for a window count that does not divide."""
    import torch
    from curdlemsm.distributed import window_partition
    n = 1 << 12
    k, q = oracle.Rand(8).get_frs(2)
    pts = coracle.points_walk(k, q, n)
    sc = rand_scalars(np.random.default_rng(8), n, oracle)
    d_pts = torch.from_numpy(pts.view(np.int64)).to("cuda:0")
    d_sc = torch.from_numpy(sc.view(np.int64)).to("cuda:0")
    exp = coracle.msm_pippenger(pts, sc, threads=8)
    for c in (16, 15, 8):
        W = gpu.num_windows(n, c)
        assert (gpu.msm_g1_device(d_pts.data_ptr(), d_sc.data_ptr(), n, window_bits=c) == exp).all()
        for world in (2, 4, 8):
            parts = []
            for rank in range(world):
                b, e = window_partition(W, world, rank)
                parts.append(gpu.msm_g1_device(d_pts.data_ptr(), d_sc.data_ptr(), n, window_bits=c, win_begin=b, win_end=e))
            assert (gpu.g1_sum(np.stack(parts)) == exp).all(), (c, world)


def test_partials_of_the_headline_plan_sum_to_the_full_msm(gpu, oracle, coracle):
    """BASELINE config 4 at its real size: N = 2^20, the plan that size takes (c = 16: eight
    windows of the 127-bit halves, one per rank at world size 8).  The 8 / 4 / 2 window-range
    partials AND the 8 / 4 / 2 point-range partials, each summed with curdle_g1_sum, equal the
    full result, which equals the closed form of the known-discrete-log inputs."""
    import torch
    from curdlemsm.distributed import point_partition, window_partition
    n = 1 << 20
    k, q = oracle.Rand(1).get_frs(2)
    d_pts = torch.empty((n, 12), dtype=torch.int64, device="cuda:0")
    gpu.synth_points_walk_device(k, q, n, d_pts.data_ptr())
    sc = rand_scalars(np.random.default_rng(2020), n, oracle)
    d_sc = torch.from_numpy(sc.view(np.int64)).to("cuda:0")
    exp = _walk_expected(oracle, coracle, k, q, sc)
    assert gpu.window_bits(n) == 16
    W = gpu.num_windows(n, 0)
    assert W == 8
    assert (gpu.msm_g1_device(d_pts.data_ptr(), d_sc.data_ptr(), n) == exp).all()
    for world in (8, 4, 2):
        parts = []
        for rank in range(world):
            b, e = window_partition(W, world, rank)
            assert e - b == W // world
            parts.append(gpu.msm_g1_device(d_pts.data_ptr(), d_sc.data_ptr(), n, win_begin=b, win_end=e))
        assert (gpu.g1_sum(np.stack(parts)) == exp).all(), ("windows", world)
        parts = []
        for rank in range(world):
            b, e = point_partition(n, world, rank)
            parts.append(gpu.msm_g1_device(d_pts.data_ptr() + 96 * b, d_sc.data_ptr() + 32 * b, e - b))
        assert (gpu.g1_sum(np.stack(parts)) == exp).all(), ("points", world)


def test_priority_turns_of_the_accumulate_waves_do_not_change_results(gpu, oracle, coracle):
    """A synchronous call from half a round of accumulate lanes lets the two waves of a SIMD take turns at
    high priority (plan field acc_prio, knob ACC_PRIO: scheduling only).  Same bits with the turns off, at the
    default and at a slice so short that priorities flip inside every mixed addition -- at 2^18 pairs (a
    whole round of lanes) against the closed form."""
    import torch
    n = 1 << 18
    k, q = oracle.Rand(5).get_frs(2)
    d_pts = torch.empty((n, 12), dtype=torch.int64, device="cuda:0")
    gpu.synth_points_walk_device(k, q, n, d_pts.data_ptr())
    sc = rand_scalars(np.random.default_rng(55), n, oracle)
    d_sc = torch.from_numpy(sc.view(np.int64)).to("cuda:0")
    exp = _walk_expected(oracle, coracle, k, q, sc)
    try:
        for v in (None, 0, 1, 6, 15, 24, 40):
            gpu.plan_override("ACC_PRIO", v)
            assert (gpu.msm_g1_device(d_pts.data_ptr(), d_sc.data_ptr(), n) == exp).all(), v
            t = gpu.msm_g1_device_submit(d_pts.data_ptr(), d_sc.data_ptr(), n)
            assert (gpu.msm_wait(t) == exp).all(), v
        gpu.plan_override("ACC_PRIO", None)
        # ... and the priority the reduction's waves run at (REDUCE_PRIO: 3 by default in pipelined calls)
        for v in (0, 1, 2, 3, 9):
            gpu.plan_override("REDUCE_PRIO", v)
            tickets = [gpu.msm_g1_device_submit(d_pts.data_ptr(), d_sc.data_ptr(), n) for _ in range(3)]
            for t in tickets:
                assert (gpu.msm_wait(t) == exp).all(), v
            assert (gpu.msm_g1_device(d_pts.data_ptr(), d_sc.data_ptr(), n) == exp).all(), v
    finally:
        gpu.plan_override("ACC_PRIO", None)
        gpu.plan_override("REDUCE_PRIO", None)


def test_msm_over_a_resident_base_set(gpu, oracle, coracle):
    """curdle_msm_g1_dbases*: the plain MSM over a pre-converted, resident base set (SURVEY.md 8b's
    caller-managed device handle; msmaccumulator.Verify's bases are mostly the CRS).  Against the C
    oracle: the whole set and a prefix of it, scalars on the device and in host memory, infinity
    bases in the set, window-range partials summing to the full result, the asynchronous form
    (several in flight, the set freed while they are: deferred), and at 2^16 + 5 pairs."""
    import torch
    from curdlemsm.distributed import window_partition
    k, q = oracle.Rand(31).get_frs(2)
    for n in (1, 700, (1 << 16) + 5):
        pts = coracle.points_walk(k, q, n)
        if n > 10:
            pts[3] = 0                                          # (0, 0) = infinity stays infinity in the set
            pts[n - 1] = 0
        sc = rand_scalars(np.random.default_rng(31 + n), n, oracle)
        d_sc = torch.from_numpy(sc.view(np.int64)).to("cuda:0")
        exp = coracle.msm_pippenger(pts, sc, threads=8)
        bases = gpu.DBases(pts)
        assert (bases.msm(d_sc.data_ptr()) == exp).all(), n
        assert (bases.msm_host(sc) == exp).all(), n
        m = n // 2
        assert (bases.msm(d_sc.data_ptr(), n=m) == coracle.msm_pippenger(pts[:m], sc[:m], threads=8)).all(), (n, m)
        assert (bases.msm_host(sc[:m]) == coracle.msm_pippenger(pts[:m], sc[:m], threads=8)).all(), (n, m)
        with pytest.raises(gpu.CurdleError):
            bases.msm(d_sc.data_ptr(), n=n + 1)                 # more pairs than resident bases
        c = gpu.window_bits(n)
        W = gpu.num_windows(n, c)
        for world in (2, 8):
            parts = [bases.msm(d_sc.data_ptr(), window_bits=c, win_begin=b, win_end=e)
                     for b, e in (window_partition(W, world, r) for r in range(world))]
            assert (gpu.g1_sum(np.stack(parts)) == exp).all(), (n, world)
        tickets = [bases.submit(d_sc.data_ptr()) for _ in range(4)]
        bases.free()                                            # deferred: four MSMs still read the set
        for t in tickets:
            assert (gpu.msm_wait(t) == exp).all(), n


# ------------------------------------------------------------------ batch / multi ---
def test_batch_and_multi_entry_points(gpu, oracle, coracle):
    k, q = oracle.Rand(9).get_frs(2)
    sizes = [0, 1, 628, 3, 308]
    offs = np.concatenate([[0], np.cumsum(sizes)])
    pts = coracle.points_walk(k, q, int(offs[-1]))
    sc = rand_scalars(np.random.default_rng(9), int(offs[-1]), oracle)
    out = gpu.msm_g1_batch(pts, sc, offs)
    for j, (lo, hi) in enumerate(zip(offs[:-1], offs[1:])):
        assert (out[j] == coracle.msm_pippenger(pts[lo:hi], sc[lo:hi], threads=4)).all(), j
    # one scalar vector against three base sets (samemultiscalarargument.go:64-70)
    n = 256
    sets = [coracle.points_walk(k + j, q, n) for j in range(3)]
    s = rand_scalars(np.random.default_rng(10), n, oracle)
    out = gpu.msm_g1_multi(sets, s)
    for j in range(3):
        assert (out[j] == coracle.msm_pippenger(sets[j], s, threads=4)).all()
    # the scalars are recoded and sorted ONCE for all base sets: sizes of the prover (2,548 = 5 ell + 8 at
    # ell = 508), many sets (one takes the GPU window combine), every window width, and the scalar families that
    # route buckets through the block-per-bucket pre-merge (shared fragment bookkeeping) or leave windows empty
    for n, nsets in ((2548, 3), (5, 2), (700, 40)):
        sets = [coracle.points_walk(k + 11 * j, q + j, n) for j in range(nsets)]
        sets[1][n // 2] = 0                                       # an infinity base in one set only
        s = rand_scalars(np.random.default_rng(n), n, oracle)
        out = gpu.msm_g1_multi(sets, s)
        for j in range(nsets):
            assert (out[j] == coracle.msm_pippenger(sets[j], s, threads=4)).all(), (n, j)
    n = 3000
    sets = [coracle.points_walk(k + 5 * j, q, n) for j in range(3)]
    fams = {"all_equal": [123456789123456789] * n, "small": [i % 300 for i in range(n)],
            "hot_window": [(7 << 130) + i for i in range(n)], "zero": [0] * n}
    try:
        for name, fam in fams.items():
            s = np.array([oracle.fr_to_mont_limbs(v % oracle.R) for v in fam], dtype=np.uint64)
            for c in (0, 6, 13):
                if c:
                    gpu.plan_override("WINDOW_BITS", c)
                out = gpu.msm_g1_multi(sets, s)
                for j in range(3):
                    assert (out[j] == coracle.msm_pippenger(sets[j], s, threads=4)).all(), (name, c, j)
    finally:
        gpu.plan_override("WINDOW_BITS", None)


def test_large_batch_runs_in_one_pass(gpu, oracle, coracle):
    """Config 5 shape: many independent 628-pair MSMs (the Whisk verifier's final MSM),
    combined on the GPU, with empty and all-infinity members in the batch."""
    import torch
    k, q = oracle.Rand(13).get_frs(2)
    sizes = [628] * 60 + [0, 1, 7, 628, 0, 300]
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint64)
    n = int(offs[-1])
    pts = coracle.points_walk(k, q, n)
    pts[offs[3]:offs[4]] = 0                      # MSM 3: every base is infinity
    sc = rand_scalars(np.random.default_rng(13), n, oracle)
    out = gpu.msm_g1_batch(pts, sc, offs)
    d_pts = torch.from_numpy(pts.view(np.int64)).to("cuda:0")
    d_sc = torch.from_numpy(sc.view(np.int64)).to("cuda:0")
    out_dev = gpu.msm_g1_batch_device(d_pts.data_ptr(), d_sc.data_ptr(), offs)
    assert (out == out_dev).all()
    for j, (lo, hi) in enumerate(zip(offs[:-1], offs[1:])):
        lo, hi = int(lo), int(hi)
        assert (out[j] == coracle.msm_pippenger(pts[lo:hi], sc[lo:hi], threads=4)).all(), j
    assert (out[3] == out[60]).all()              # infinity either way
    # a batch of one takes the single-MSM path and must agree
    assert (gpu.msm_g1_batch(pts[:628], sc[:628], [0, 628])[0] == out[0]).all()


def test_async_submit_wait_and_concurrent_callers(gpu, oracle, coracle):
    """Several MSMs in flight on the workspace slots (submit / wait), waits in any
    order, CURDLE_EBUSY when every slot is taken; and concurrent synchronous callers
    from threads (the reference's tests run t.Parallel())."""
    import threading
    import torch
    k, q = oracle.Rand(14).get_frs(2)
    sizes = [1 << 12, 3000, 1 << 13]
    pts = [coracle.points_walk(k + i, q, n) for i, n in enumerate(sizes)]
    scs = [rand_scalars(np.random.default_rng(100 + i), n, oracle) for i, n in enumerate(sizes)]
    exp = [coracle.msm_pippenger(p, s, threads=8) for p, s in zip(pts, scs)]
    d_p = [torch.from_numpy(p.view(np.int64)).to("cuda:0") for p in pts]
    d_s = [torch.from_numpy(s.view(np.int64)).to("cuda:0") for s in scs]
    for _ in range(3):
        ns = gpu.MSM_SLOTS
        tickets = [gpu.msm_g1_device_submit(d_p[i % 3].data_ptr(), d_s[i % 3].data_ptr(), sizes[i % 3]) for i in range(ns)]
        assert sorted(t & 0xFF for t in tickets) == list(range(ns))   # one slot each (slot index in the low byte)
        with pytest.raises(gpu.CurdleError) as e:
            gpu.msm_g1_device_submit(d_p[0].data_ptr(), d_s[0].data_ptr(), sizes[0])
        assert e.value.code == gpu.EBUSY
        for i in [1, 2, 0] + list(range(3, ns)):    # out-of-order waits
            assert (gpu.msm_wait(tickets[i]) == exp[i % 3]).all(), i
        with pytest.raises(gpu.CurdleError):
            gpu.msm_wait(tickets[0])                # already collected
        # a stale ticket stays refused when its slot is busy again for another caller
        fresh = gpu.msm_g1_device_submit(d_p[0].data_ptr(), d_s[0].data_ptr(), sizes[0])
        stale = [t for t in tickets if (t & 0xFF) == (fresh & 0xFF)][0]
        assert stale != fresh
        with pytest.raises(gpu.CurdleError):
            gpu.msm_wait(stale)
        assert (gpu.msm_wait(fresh) == exp[0]).all()
    # a window-range partial submitted asynchronously
    W = gpu.num_windows(sizes[0], 12)
    t1 = gpu.msm_g1_device_submit(d_p[0].data_ptr(), d_s[0].data_ptr(), sizes[0], window_bits=12, win_begin=0, win_end=W // 2)
    t2 = gpu.msm_g1_device_submit(d_p[0].data_ptr(), d_s[0].data_ptr(), sizes[0], window_bits=12, win_begin=W // 2, win_end=W)
    assert (gpu.g1_sum(np.stack([gpu.msm_wait(t1), gpu.msm_wait(t2)])) == exp[0]).all()
    # ... and with the library's own width: a range counts the windows num_windows(n, 0) reports
    W0 = gpu.num_windows(sizes[2], 0)
    t1 = gpu.msm_g1_device_submit(d_p[2].data_ptr(), d_s[2].data_ptr(), sizes[2], win_begin=0, win_end=W0 // 3)
    t2 = gpu.msm_g1_device_submit(d_p[2].data_ptr(), d_s[2].data_ptr(), sizes[2], win_begin=W0 // 3, win_end=W0)
    assert (gpu.g1_sum(np.stack([gpu.msm_wait(t1), gpu.msm_wait(t2)])) == exp[2]).all()
    assert (gpu.msm_wait(gpu.msm_g1_device_submit(d_p[2].data_ptr(), d_s[2].data_ptr(), sizes[2])) == exp[2]).all()
    # threads
    errs = []

    def worker(i):
        try:
            for _ in range(4):
                if not (gpu.msm_g1(pts[i % 3], scs[i % 3]) == exp[i % 3]).all():
                    errs.append(i)
        except Exception as ex:  # noqa: BLE001
            errs.append(repr(ex))
    th = [threading.Thread(target=worker, args=(i,)) for i in range(6)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert errs == []


# ----------------------------------------------------- msmaccumulator (reference tests) ---
def test_msm_accumulator_reference_test(gpu):
    """msmaccumulator/msmaccumulator_test.go:12-50, sizes 0..3 (the Go loop ranges
    over indices), MultiExp and Verify on the GPU."""
    for n in (0, 1, 2, 3):
        rand = gpu.Rand(0)
        A = rand.get_g1_affines(n)
        x = rand.get_frs(n)
        C1 = gpu.msm_g1(A, x)
        B = rand.get_g1_affines(n)
        y = rand.get_frs(n)
        C2 = gpu.msm_g1(B, y)
        ma = gpu.MsmAccumulator()
        ma.accumulate_check(C1, x, A, rand)
        ma.accumulate_check(C2, y, B, rand)
        assert ma.verify() is True


def test_msm_accumulator_matches_golden_and_rejects_wrong_instance(gpu, golden_acc):
    for n in (1, 2, 3):
        rand = gpu.Rand(0)
        A = rand.get_g1_affines(n); x = rand.get_frs(n)
        B = rand.get_g1_affines(n); y = rand.get_frs(n)
        assert (A == golden_acc[f"n{n}_A"]).all() and (y == golden_acc[f"n{n}_y"]).all()
        C1, C2 = gpu.msm_g1(A, x), gpu.msm_g1(B, y)
        assert (C1 == golden_acc[f"n{n}_C1"]).all() and (C2 == golden_acc[f"n{n}_C2"]).all()
        ma = gpu.MsmAccumulator()
        ma.accumulate_check(C1, x, A, rand)
        ma.accumulate_check(C2, y, B, rand)
        assert (ma.A_c == golden_acc[f"n{n}_A_c"]).all()
        assert ma.verify()
        # soundness: a wrong claimed result flips only the final batched check
        # (grandproductargument_test.go:107-111)
        bad = gpu.MsmAccumulator()
        r2 = gpu.Rand(1)
        bad.accumulate_check(C2, x, A, r2)      # C2 is not MSM(A, x)
        bad.accumulate_check(C2, y, B, r2)
        assert bad.verify() is False


def test_verifier_shaped_accumulation(gpu, oracle, coracle):
    """Config 3's final MSM shape: ell = 252 -> 5*ell + 8 = 1,268 distinct bases with
    shared CRS bases merged across 8 checks and one infinity base."""
    ell, nbl = 252, 4
    n = ell + nbl
    k, q = oracle.Rand(11).get_frs(2)
    allp = coracle.points_walk(k, q, 5 * ell + 8)
    Gs, Hs, H = allp[:ell], allp[ell:ell + nbl], allp[ell + nbl:ell + nbl + 1]
    Gt, Gu = allp[n + 1:n + 2], allp[n + 2:n + 3]
    Rs, Ss, Ts, Us = (allp[n + 3 + i * ell:n + 3 + (i + 1) * ell] for i in range(4))
    zero = np.zeros((1, 12), dtype=np.uint64)
    checks = [
        Gs,                                             # sameperm :140 (all-equal scalars beta)
        np.concatenate([Gs, Hs, H]),                    # ipa :269  (n + 1 terms)
        np.concatenate([Gs, Hs]),                       # ipa :292
        np.concatenate([Gs, Hs[:2], Gt, Gu]),           # samemsm :206  G
        np.concatenate([Ts, zero, zero, H, zero]),      # samemsm :218  T' (infinity bases)
        np.concatenate([Us, zero, zero, zero, H]),      # samemsm :231  U'
        Rs, Ss,                                         # curdleproof.go:306, :309
    ]
    rng = np.random.default_rng(12)
    rand = gpu.Rand(43)
    ma = gpu.MsmAccumulator()
    for i, v in enumerate(checks):
        x = rand_scalars(rng, len(v), oracle)
        if i == 0:
            x[:] = x[0]
        C = coracle.msm_pippenger(v, x, threads=8)
        ma.accumulate_check(C, x, v, rand)
    assert ma.num_bases() == 5 * ell + 8
    assert ma.verify() is True
    pts, sc = ma.export()
    assert (gpu.msm_g1(pts, sc) == ma.A_c).all()
    assert (coracle.msm_pippenger(pts, sc, threads=8) == ma.A_c).all()


def test_batched_scalar_multiplication(gpu, oracle):
    """curdle_g1_scalar_mul_batch: out[i] = A[i] + s[i] * P[i] (the prover's fold step and its
    plain scalar multiplications) against the oracle, incl. zero / one / r-1 scalars, infinity
    operands, P + (-P) and doubling through the addend, one shared scalar, and a size that
    takes the one-lane-per-point kernel."""
    r = oracle.Rand(21)
    n = 40
    P = r.get_g1_affines(n)
    A = r.get_g1_affines(n)
    s = r.get_frs(n)
    s[0], s[1], s[2], s[3] = 0, 1, oracle.R - 1, 2
    P[4] = oracle.INF
    A[5] = oracle.INF
    A[6] = oracle.neg(oracle.scalar_mul(s[6], P[6]))   # result is infinity
    A[7] = oracle.scalar_mul(s[7], P[7])                # addition of equal points
    Pl = np.array([oracle.affine_to_mont_limbs(p) for p in P], dtype=np.uint64)
    Al = np.array([oracle.affine_to_mont_limbs(p) for p in A], dtype=np.uint64)
    sl = np.array([oracle.fr_to_mont_limbs(v) for v in s], dtype=np.uint64)
    got = gpu.g1_scalar_mul_batch(Pl, sl, Al)
    for i in range(n):
        want = oracle.add(A[i], oracle.scalar_mul(s[i], P[i]))
        assert oracle.affine_from_mont_limbs([int(v) for v in got[i]]) == want, i
    got = gpu.g1_scalar_mul_batch(Pl, sl)               # no addend
    for i in range(n):
        assert oracle.affine_from_mont_limbs([int(v) for v in got[i]]) == oracle.scalar_mul(s[i], P[i]), i
    got = gpu.g1_scalar_mul_batch(Pl, sl[9], Al)        # one scalar for all (the fold step)
    for i in range(n):
        assert oracle.affine_from_mont_limbs([int(v) for v in got[i]]) == oracle.add(A[i], oracle.scalar_mul(s[9], P[i])), i
    assert gpu.g1_scalar_mul_batch(Pl[:0], sl[:0]).shape == (0, 12)
    # 2^15 + 1 points: past the four-lanes-per-point limit; checked through linearity against the MSM
    big = 32769
    pts = gpu.Rand(4).get_g1_affines(64)
    pts = np.concatenate([pts] * (big // 64 + 1))[:big]
    for K in (0x1234567, oracle.R * 2 // 3 + 0x1234567, oracle.R // 3 - 5):   # one half only; both halves, negative; both, positive
        k = np.array(oracle.fr_to_mont_limbs(K), dtype=np.uint64)
        out = gpu.g1_scalar_mul_batch(pts, k)
        ones = np.array([oracle.fr_to_mont_limbs(1)] * big, dtype=np.uint64)
        lhs = gpu.msm_g1(out, ones)                                  # sum_i k P_i
        rhs = gpu.msm_g1(pts, np.array([oracle.fr_to_mont_limbs(K)] * big, dtype=np.uint64))
        assert (lhs == rhs).all(), hex(K)
        assert oracle.affine_from_mont_limbs([int(v) for v in out[big - 1]]) == oracle.scalar_mul(
            K, oracle.affine_from_mont_limbs([int(v) for v in pts[big - 1]])), hex(K)


def _off_subgroup_point(oracle):
    """A point of y^2 = x^3 + 4 over F_p that is NOT in the prime-order subgroup."""
    p, R = oracle.P, oracle.R
    x = 6
    while True:
        rhs = (x * x * x + 4) % p
        y = pow(rhs, (p + 1) // 4, p)
        if y * y % p == rhs and oracle.scalar_mul(R, (x, y)) is not None:
            return (x, y)
        x += 1


def test_any_curve_point_flag_is_gnarks_contract(gpu, oracle, coracle):
    """CURDLE_MSM_ANY_CURVE_POINT: no endomorphism, the scalar recoded whole over twice
    the windows -- the result is k P for EVERY point of the curve, which is what gnark's MultiExp returns
    (go.mod:6; SURVEY.md a6 states no subgroup condition).  (1) the off-subgroup point of the test above,
    where the default path returns k1 P + k2 (beta x, y): with the flag, oracle.scalar_mul(k, P) exactly;
    (2) mixed MSMs -- bases in G1, outside it, infinity, duplicates -- of every plan family (tiny, the
    verifier's size, one-pass and two-pass scatter) against the textbook sum; (3) on G1 inputs the flag must
    not change a bit, at 2^17 + 77 pairs with skewed scalar sets and over window partials."""
    import torch
    R = oracle.R
    P = _off_subgroup_point(oracle)
    for k in (1, 5, (1 << 127) + 12345, 0x1234567890ABCDEF1234567890ABCDEF1234567890ABCDEF % R, R - 2, R - 1):
        pts = np.array([oracle.affine_to_mont_limbs(P)], dtype=np.uint64)
        sc = np.array([oracle.fr_to_mont_limbs(k)], dtype=np.uint64)
        got = gpu.msm_g1(pts, sc, flags=gpu.MSM_ANY_CURVE_POINT)
        assert [int(v) for v in got] == oracle.jac_to_mont_limbs(oracle.scalar_mul(k, P)), hex(k)
    # mixed bases: multiples of P (outside G1 unless the multiplier kills the cofactor part), G1 points, infinity
    rng = np.random.default_rng(55)
    walk_k, walk_q = oracle.Rand(1).get_frs(2)
    g1 = coracle.points_walk(walk_k, walk_q, 4096)
    offs = [oracle.scalar_mul(m, P) for m in (1, 2, 3, 5, 7, 11)]
    off_limbs = np.array([oracle.affine_to_mont_limbs(t) for t in offs], dtype=np.uint64)
    for n in (2, 9, 70, 700, 2548, 4096):
        pts = g1[:n].copy()
        where = rng.choice(n, size=max(1, n // 7), replace=False)
        pts[where] = off_limbs[rng.integers(0, len(offs), len(where))]
        pts[rng.integers(0, n)] = 0                                   # an infinity base
        sc = rand_scalars(rng, n, oracle)
        sc[rng.integers(0, n)] = np.array(oracle.fr_to_mont_limbs(R - 1), dtype=np.uint64)
        exp = coracle.msm_naive(pts, sc) if n <= 700 else coracle.msm_pippenger(pts, sc, threads=4)
        got = gpu.msm_g1(pts, sc, flags=gpu.MSM_ANY_CURVE_POINT)
        assert (got == exp).all(), n
        d_p = torch.from_numpy(pts.view(np.int64)).to("cuda:0")
        d_s = torch.from_numpy(sc.view(np.int64)).to("cuda:0")
        assert (gpu.msm_g1_device(d_p.data_ptr(), d_s.data_ptr(), n, flags=gpu.MSM_ANY_CURVE_POINT) == exp).all(), n
        if n == 700:
            assert not (gpu.msm_g1(pts, sc) == exp).all()             # the default path is NOT gnark's there
    # on G1 the flag changes nothing: two-pass scatter size, skewed scalars, window partials
    n = (1 << 17) + 77
    d_pts = torch.empty((n, 12), dtype=torch.int64, device="cuda:0")
    gpu.synth_points_walk_device(walk_k, walk_q, n, d_pts.data_ptr())
    uniform = rand_scalars(rng, n, oracle)
    fams = {"uniform": uniform,
            "all_equal": np.tile(np.array(oracle.fr_to_mont_limbs(123456789123456789123456789), dtype=np.uint64), (n, 1)),
            "small": np.array([oracle.fr_to_mont_limbs(i % 300) for i in range(n)], dtype=np.uint64)}
    for name, sc in fams.items():
        d_sc = torch.from_numpy(sc.view(np.int64)).to("cuda:0")
        exp = _walk_expected(oracle, coracle, walk_k, walk_q, sc)
        for c in (0, 13, 16):
            got = gpu.msm_g1_device(d_pts.data_ptr(), d_sc.data_ptr(), n, window_bits=c, flags=gpu.MSM_ANY_CURVE_POINT)
            assert (got == exp).all(), (name, c)
    d_sc = torch.from_numpy(uniform.view(np.int64)).to("cuda:0")
    exp = _walk_expected(oracle, coracle, walk_k, walk_q, uniform)
    W = (255 + 15) // 16
    parts = [gpu.msm_g1_device(d_pts.data_ptr(), d_sc.data_ptr(), n, window_bits=16, win_begin=w, win_end=min(w + 3, W),
                               flags=gpu.MSM_ANY_CURVE_POINT) for w in range(0, W, 3)]
    assert (gpu.g1_sum(np.stack(parts)) == exp).all()
    # host buffers in chunks (2^19 pairs and more) take the flag too
    n = 1 << 19
    d_big = torch.empty((n, 12), dtype=torch.int64, device="cuda:0")
    gpu.synth_points_walk_device(walk_k, walk_q, n, d_big.data_ptr())
    sc = rand_scalars(rng, n, oracle)
    got = gpu.msm_g1(d_big.cpu().numpy().view(np.uint64), sc, flags=gpu.MSM_ANY_CURVE_POINT)
    assert (got == _walk_expected(oracle, coracle, walk_k, walk_q, sc)).all()
    with pytest.raises(gpu.CurdleError):
        gpu.msm_g1(g1[:4], rand_scalars(rng, 4, oracle), flags=gpu.MSM_BASES_UNCHANGED)   # host buffers are never cached
    with pytest.raises(gpu.CurdleError):
        gpu.msm_g1(g1[:4], rand_scalars(rng, 4, oracle), flags=64)


def test_bases_unchanged_flag_keeps_a_converted_copy(gpu, oracle, coracle):
    """CURDLE_MSM_BASES_UNCHANGED: the caller's promise lets the library keep its converted copy of a device
    base array -- keyed by pointer and count, never without the flag (what a rank of the
    window split does with its resident bases).  Results are those of the unflagged call: whole MSM, prefix
    of the array (its own entry), window partials, six pipelined calls, more base arrays than entries (the idle
    one is replaced), a forgotten array whose memory was rewritten."""
    import torch
    F = gpu.MSM_BASES_UNCHANGED
    k, q = oracle.Rand(1).get_frs(2)
    n = (1 << 17) + 5
    d_pts = torch.empty((n, 12), dtype=torch.int64, device="cuda:0")
    gpu.synth_points_walk_device(k, q, n, d_pts.data_ptr())
    rng = np.random.default_rng(77)
    for rep in range(3):                                              # first call converts, the others reuse
        sc = rand_scalars(rng, n, oracle)
        d_sc = torch.from_numpy(sc.view(np.int64)).to("cuda:0")
        exp = _walk_expected(oracle, coracle, k, q, sc)
        assert (gpu.msm_g1_device(d_pts.data_ptr(), d_sc.data_ptr(), n, flags=F) == exp).all(), rep
    m = 3000                                                          # a prefix is another (pointer, count)
    assert (gpu.msm_g1_device(d_pts.data_ptr(), d_sc.data_ptr(), m, flags=F) == _walk_expected(oracle, coracle, k, q, sc[:m])).all()
    W = gpu.num_windows(n, 16)
    parts = [gpu.msm_g1_device(d_pts.data_ptr(), d_sc.data_ptr(), n, window_bits=16, win_begin=w, win_end=w + 1, flags=F)
             for w in range(W)]
    assert (gpu.g1_sum(np.stack(parts)) == exp).all()
    tickets = [gpu.msm_g1_device_submit(d_pts.data_ptr(), d_sc.data_ptr(), n, 16, w % W, w % W + 1, flags=F) for w in range(6)]
    for w, t in enumerate(tickets):
        assert (gpu.msm_wait(t) == parts[w % W]).all(), w
    both = gpu.msm_g1_device(d_pts.data_ptr(), d_sc.data_ptr(), n, flags=F | gpu.MSM_ANY_CURVE_POINT)
    assert (both == exp).all()
    # more arrays than cache entries
    others = []
    for j in range(6):
        kj, qj = oracle.Rand(20 + j).get_frs(2)
        t = torch.empty((2000, 12), dtype=torch.int64, device="cuda:0")
        gpu.synth_points_walk_device(kj, qj, 2000, t.data_ptr())
        others.append((kj, qj, t))
    s2 = rand_scalars(rng, 2000, oracle)
    d_s2 = torch.from_numpy(s2.view(np.int64)).to("cuda:0")
    for rnd in range(2):
        for kj, qj, t in others:
            assert (gpu.msm_g1_device(t.data_ptr(), d_s2.data_ptr(), 2000, flags=F) == _walk_expected(oracle, coracle, kj, qj, s2)).all()
    # rewritten memory: forget first, then the new contents count
    kj, qj, t = others[-1]
    gpu.msm_forget_bases(t.data_ptr())
    k9, q9 = oracle.Rand(99).get_frs(2)
    gpu.synth_points_walk_device(k9, q9, 2000, t.data_ptr())
    torch.cuda.synchronize()
    assert (gpu.msm_g1_device(t.data_ptr(), d_s2.data_ptr(), 2000, flags=F) == _walk_expected(oracle, coracle, k9, q9, s2)).all()
    gpu.msm_forget_bases(d_pts.data_ptr())


def test_bucket_slot_scans_at_every_size(gpu, oracle, coracle):
    """The builds of the bucket-slot scan -- k_scan_one (one block, slots read once, wave-shuffle block scans: synchronous
    calls up to 32,768 slots), k_scan_chain (one launch at any size, tile sums handed down a chain: beyond that, and
    every pipelined or chunked call) and the six-launch multi-block form (L = 1 only) -- each reached the way a caller
    reaches it, against the closed form: sizes on both sides of 32,768 slots (one tile / many tiles of the chain), a
    batch, pipelined calls that reuse a slot's chain words under new epochs, window ranges, one position per lane,
    and a bucket that takes the large-bucket queue.  (Round 6: the knob that forced a build, and k_scan_fused, are gone.)"""
    import torch
    k, q = oracle.Rand(1).get_frs(2)
    nmax = 1 << 17
    d_pts = torch.empty((nmax, 12), dtype=torch.int64, device="cuda:0")
    gpu.synth_points_walk_device(k, q, nmax, d_pts.data_ptr())
    sc = rand_scalars(np.random.default_rng(8), nmax, oracle)
    d_sc = torch.from_numpy(sc.view(np.int64)).to("cuda:0")
    for n, c in ((9, 0), (300, 0), (1268, 0), (4096, 0), (40000, 11), (65536, 0), (nmax, 14), (nmax, 16)):
        exp = _walk_expected(oracle, coracle, k, q, sc[:n])
        assert (gpu.msm_g1_device(d_pts.data_ptr(), d_sc.data_ptr(), n, window_bits=c) == exp).all(), (n, c)   # k_scan_one or the chain, by slot count
        assert (gpu.msm_wait(gpu.msm_g1_device_submit(d_pts.data_ptr(), d_sc.data_ptr(), n, window_bits=c)) == exp).all(), (n, c)   # the chain
    pts = d_pts[:6000].cpu().numpy().view(np.uint64)
    offs = [0, 100, 100, 2600, 6000]
    got = gpu.msm_g1_batch(pts, sc[:6000], offs)
    for j in (0, 2, 3):
        assert (got[j] == coracle.msm_pippenger(pts[offs[j]:offs[j + 1]], sc[offs[j]:offs[j + 1]], threads=4)).all(), j
    # pipelined calls (several in flight, slots and their chain words reused) and window ranges
    n = 1 << 16
    exp = _walk_expected(oracle, coracle, k, q, sc[:n])
    for _ in range(3):
        ts = [gpu.msm_g1_device_submit(d_pts.data_ptr(), d_sc.data_ptr(), n) for _ in range(5)]
        for t in ts:
            assert (gpu.msm_wait(t) == exp).all()
    W = gpu.num_windows(n)
    parts = []
    for w0 in range(0, W, 5):   # at most eight calls can be in flight
        ts = [gpu.msm_g1_device_submit(d_pts.data_ptr(), d_sc.data_ptr(), n, 0, w, w + 1) for w in range(w0, min(W, w0 + 5))]
        parts += [gpu.msm_wait(t) for t in ts]
    assert (gpu.g1_sum(np.stack(parts)) == exp).all()
    # one position per accumulate lane (knob SEG_LEN = 1): the one-launch scans divide by L with a multiply that needs
    # L >= 2, so the plan takes the six launches
    gpu.plan_override("SEG_LEN", 1)
    try:
        exp = _walk_expected(oracle, coracle, k, q, sc[:3000])
        assert (gpu.msm_g1_device(d_pts.data_ptr(), d_sc.data_ptr(), 3000) == exp).all()
        assert (gpu.msm_wait(gpu.msm_g1_device_submit(d_pts.data_ptr(), d_sc.data_ptr(), 3000)) == exp).all()
    finally:
        gpu.plan_override("SEG_LEN", -1)
    # every scalar the same: one bucket per window holds all the terms and goes through the queue of large buckets,
    # which the chain's first tile clears and the others append to
    same = np.repeat(sc[:1], 1 << 16, axis=0)
    d_same = torch.from_numpy(same.view(np.int64)).to("cuda:0")
    exp = _walk_expected(oracle, coracle, k, q, same)
    assert (gpu.msm_g1_device(d_pts.data_ptr(), d_same.data_ptr(), 1 << 16) == exp).all()
    assert (gpu.msm_wait(gpu.msm_g1_device_submit(d_pts.data_ptr(), d_same.data_ptr(), 1 << 16)) == exp).all()


def test_concurrent_host_buffer_calls(gpu, oracle, coracle):
    """Host-buffer MSMs from several threads at once: chunked calls (from 2^19 pairs: three or four chunks, each on its slot's
    own streams, all copies on the context's one copy stream, up to four slots per call out of eight) beside mid-size calls
    whose sort runs under their points' copy -- every result equals the resident call's."""
    import threading
    import torch

    def prep(n, seed):
        k, q = oracle.Rand(seed).get_frs(2)
        d = torch.empty((n, 12), dtype=torch.int64, device="cuda:0")
        gpu.synth_points_walk_device(k, q, n, d.data_ptr())
        pts = d.cpu().numpy().view(np.uint64).copy()
        sc = rand_scalars(np.random.default_rng(seed), n, oracle)
        ref = gpu.msm_g1_device(d.data_ptr(), torch.from_numpy(sc.view(np.int64)).to("cuda:0").data_ptr(), n)
        return pts, sc, ref

    cases = [prep(1 << 19, 31), prep(600001, 32), prep(40000, 33), prep(1 << 17, 34)]
    bad = []

    def work(t):
        for it in range(4):
            pts, sc, ref = cases[(t + it) % len(cases)]
            if not (gpu.msm_g1(pts, sc) == ref).all():
                bad.append((t, it))

    ths = [threading.Thread(target=work, args=(t,)) for t in range(4)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert not bad, bad
    # and one of them against the oracle
    pts, sc, ref = cases[2]
    assert (ref == coracle.msm_pippenger(pts, sc, threads=4)).all()
