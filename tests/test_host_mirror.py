"""Host side of the library (the C++ mirror of common.Rand and msmaccumulator and
the limb arithmetic shared with the kernels) against the oracle, on the CPU."""
import json
import os
import random

import numpy as np
import pytest

from conftest import ROOT


def fp32(v):
    return [(v >> (32 * i)) & 0xFFFFFFFF for i in range(12)]


def from32(l):
    return sum(int(x) << (32 * i) for i, x in enumerate(l))


def test_rand_mirror_matches_reference_construction(cm, oracle):
    with open(os.path.join(ROOT, "tests", "golden", "rand_known_answers.json")) as f:
        ka = json.load(f)
    for seed, key in ((0, "seed0_first_fr"), (42, "seed42_first_fr"), (43, "seed43_first_fr")):
        assert hex(oracle.fr_from_mont_limbs([int(v) for v in cm.Rand(seed).get_fr()])) == ka[key]
    g = cm.Rand(0).get_g1_affine()
    assert oracle.compress(oracle.affine_from_mont_limbs([int(v) for v in g])).hex() == ka["seed0_first_g1_compressed"]
    r = cm.Rand(0)
    assert [hex(oracle.fr_from_mont_limbs([int(v) for v in r.get_fr()])) for _ in range(8)] == ka["seed0_frs_8"]
    assert list(cm.Rand(42).generate_permutation(10)) == ka["seed42_permutation_10"]
    assert list(cm.Rand(0).generate_permutation(124)) == ka["seed0_permutation_124"]
    # interleaved draws stay in lock-step with the oracle stream (rejections included)
    a, b = cm.Rand(3), oracle.Rand(3)
    for _ in range(3):
        assert oracle.fr_from_mont_limbs([int(v) for v in a.get_fr()]) == b.get_fr()
        assert oracle.affine_from_mont_limbs([int(v) for v in a.get_g1_affine()]) == b.get_g1_affine()
        assert list(a.generate_permutation(7)) == b.generate_permutation(7)


def test_field_primitives_host(cm, oracle):
    rnd = random.Random(1)
    N = 300
    A = [rnd.randrange(oracle.P) for _ in range(N)]
    B = [rnd.randrange(oracle.P) for _ in range(N)]
    A[0], B[1], A[2], B[2], A[3], B[3] = 0, 0, oracle.P - 1, oracle.P - 1, 1, oracle.P - 1
    inp = np.array([fp32(a) + fp32(b) for a, b in zip(A, B)], dtype=np.uint32)
    ops = {0: lambda a, b: a * b * oracle.R_FP_INV % oracle.P, 1: lambda a, b: (a + b) % oracle.P,
           2: lambda a, b: (a - b) % oracle.P, 3: lambda a, b: a * a * oracle.R_FP_INV % oracle.P}
    for op, f in ops.items():
        out = cm.selftest_op(op, inp, False)
        assert all(from32(out[i]) == f(A[i], B[i]) for i in range(N)), op
    S = [rnd.randrange(oracle.R) for _ in range(N)]
    S[0], S[1] = 0, oracle.R - 1
    inp = np.array([[(s >> (32 * i)) & 0xFFFFFFFF for i in range(8)] + [0] * 8 for s in S], dtype=np.uint32)
    out = cm.selftest_op(4, inp, False)
    assert all(sum(int(x) << (32 * i) for i, x in enumerate(out[k])) == S[k] * oracle.R_FR_INV % oracle.R for k in range(N))


def xyzz_limbs(o, pt, z):
    if pt is None:
        return fp32(o.R_FP) + fp32(o.R_FP) + [0] * 24
    zz = z * z % o.P
    zzz = zz * z % o.P
    return (fp32(pt[0] * zz % o.P * o.R_FP % o.P) + fp32(pt[1] * zzz % o.P * o.R_FP % o.P)
            + fp32(zz * o.R_FP % o.P) + fp32(zzz * o.R_FP % o.P))


def xyzz_to_affine(o, l):
    X, Y, ZZ, ZZZ = [from32(l[12 * i:12 * i + 12]) * o.R_FP_INV % o.P for i in range(4)]
    if ZZ == 0:
        return None
    return (X * pow(ZZ, -1, o.P) % o.P, Y * pow(ZZZ, -1, o.P) % o.P)


def group_cases(o):
    rnd = random.Random(2)
    pts = [o.scalar_mul(rnd.randrange(1, o.R), o.G1) for _ in range(12)]
    cases = [(pts[i], rnd.randrange(2, o.P), pts[i + 1], rnd.randrange(2, o.P)) for i in range(10)]
    cases += [(pts[0], 5, pts[0], 7),            # equal points -> doubling branch
              (pts[0], 5, o.neg(pts[0]), 7),     # opposite points -> infinity branch
              (None, 1, pts[1], 3), (pts[1], 3, None, 1), (None, 1, None, 1)]
    return cases


def group_inputs(o, op, cases):
    rows = []
    for a, za, b, zb in cases:
        if op == 5:
            bl = (fp32(b[0] * o.R_FP % o.P) + fp32(b[1] * o.R_FP % o.P) + [0] * 24) if b else [0] * 48
        else:
            bl = xyzz_limbs(o, b, zb)
        rows.append(xyzz_limbs(o, a, za) + bl)
    return np.array(rows, dtype=np.uint32)


def test_group_law_host_including_exceptional_cases(cm, oracle):
    cases = group_cases(oracle)
    for op in (5, 6, 7):
        out = cm.selftest_op(op, group_inputs(oracle, op, cases), False)
        exp = [oracle.add(a, b) if op != 7 else oracle.add(a, a) for a, _, b, _ in cases]
        assert all(xyzz_to_affine(oracle, out[i]) == exp[i] for i in range(len(cases))), op


def test_accumulate_check_mirror_matches_oracle(cm, oracle, golden_acc):
    """AccumulateCheck (msmaccumulator.go:23-47) on the host mirror: A_c and the
    flattened base->scalar map equal the oracle's for the reference test's inputs."""
    for n in (0, 1, 2, 3):
        r = cm.Rand(0)
        for _ in range(n):
            r.get_g1_affine()
        for _ in range(n):
            r.get_fr()
        for _ in range(n):
            r.get_g1_affine()
        for _ in range(n):
            r.get_fr()
        acc = cm.MsmAccumulator()
        acc.accumulate_check(golden_acc[f"n{n}_C1"], golden_acc[f"n{n}_x"], golden_acc[f"n{n}_A"], r)
        acc.accumulate_check(golden_acc[f"n{n}_C2"], golden_acc[f"n{n}_y"], golden_acc[f"n{n}_B"], r)
        assert (acc.A_c == golden_acc[f"n{n}_A_c"]).all(), n
        pts, sc = acc.export()
        assert acc.num_bases() == len(golden_acc[f"n{n}_map_points"])
        want = {bytes(p.tobytes()): bytes(s.tobytes()) for p, s in zip(golden_acc[f"n{n}_map_points"], golden_acc[f"n{n}_map_scalars"])}
        got = {bytes(p.tobytes()): bytes(s.tobytes()) for p, s in zip(pts, sc)}
        assert got == want, n


def test_accumulate_check_length_mismatch(cm, oracle):
    # msmaccumulator.go:28-30
    acc = cm.MsmAccumulator()
    P = np.array([oracle.affine_to_mont_limbs(oracle.G1)] * 2, dtype=np.uint64)
    S = np.array([oracle.fr_to_mont_limbs(5)], dtype=np.uint64)
    with pytest.raises(cm.CurdleError) as e:
        acc.accumulate_check(np.array(oracle.jac_to_mont_limbs(oracle.G1), dtype=np.uint64), S, P, cm.Rand(0))
    assert e.value.code == cm.EINVAL and "same length" in e.value.msg


def test_accumulator_merges_shared_bases_and_infinity_key(cm, oracle):
    r = oracle.Rand(5)
    A = r.get_g1_affines(3) + [oracle.INF]
    x, y = r.get_frs(4), r.get_frs(4)
    Al = np.array([oracle.affine_to_mont_limbs(p) for p in A], dtype=np.uint64)
    acc = cm.MsmAccumulator()
    rr = cm.Rand(1)
    for sc in (x, y):
        C = np.array(oracle.jac_to_mont_limbs(oracle.msm(A, sc)), dtype=np.uint64)
        acc.accumulate_check(C, np.array([oracle.fr_to_mont_limbs(s) for s in sc], dtype=np.uint64), Al, rr)
    assert acc.num_bases() == 4
    o_acc = oracle.MsmAccumulator()
    ro = oracle.Rand(1)
    o_acc.accumulate_check(oracle.msm(A, x), x, A, ro)
    o_acc.accumulate_check(oracle.msm(A, y), y, A, ro)
    assert [int(v) for v in acc.A_c] == oracle.jac_to_mont_limbs(o_acc.A_c)
    pts, sc = acc.export()
    for p, s in zip(pts, sc):
        key = oracle.affine_from_mont_limbs([int(v) for v in p])
        assert o_acc.base_scalar_map[key if key is not None else "inf"] == oracle.fr_from_mont_limbs([int(v) for v in s])


def test_accumulate_check_deferred_is_the_same_equation(cm, oracle):
    """AccumulateCheckDeferred hands C over as sum_j c_j P_j: alpha is drawn like
    AccumulateCheck's, A_c stays put and the map gains -alpha c_j P_j, so
    MSM(map) - A_c is the same group element either way (zero iff the check holds)."""
    r = oracle.Rand(11)
    V = r.get_g1_affines(4)
    x = r.get_frs(4)
    P = r.get_g1_affines(3) + [V[1]]           # one check-point base is also a statement base: entries merge
    c = r.get_frs(4)
    C_true = oracle.msm(P, c)
    to_pts = lambda pts: np.array([oracle.affine_to_mont_limbs(p) for p in pts], dtype=np.uint64)
    to_frs = lambda frs: np.array([oracle.fr_to_mont_limbs(s) for s in frs], dtype=np.uint64)

    def residual(acc):
        pts, sc = acc.export()
        bases = [oracle.affine_from_mont_limbs([int(v) for v in p]) for p in pts]
        scal = [oracle.fr_from_mont_limbs([int(v) for v in s]) for s in sc]
        lhs = oracle.msm(bases, scal)
        a_c = oracle.jac_from_mont_limbs([int(v) for v in acc.A_c])
        return oracle.add(lhs, oracle.neg(a_c))

    eager, deferred = cm.MsmAccumulator(), cm.MsmAccumulator()
    eager.accumulate_check(np.array(oracle.jac_to_mont_limbs(C_true), dtype=np.uint64), to_frs(x), to_pts(V), cm.Rand(3))
    deferred.accumulate_check_deferred(to_frs(c), to_pts(P), to_frs(x), to_pts(V), cm.Rand(3))
    assert oracle.jac_from_mont_limbs([int(v) for v in deferred.A_c]) is None      # A_c untouched (infinity)
    assert deferred.num_bases() == 4 + 3                                           # V[1] merged
    assert residual(eager) == residual(deferred)
    # and the residual is zero exactly when C really is MSM(V, x)
    ok_e, ok_d = cm.MsmAccumulator(), cm.MsmAccumulator()
    ok_e.accumulate_check(np.array(oracle.jac_to_mont_limbs(oracle.msm(V, x)), dtype=np.uint64), to_frs(x), to_pts(V), cm.Rand(4))
    ok_d.accumulate_check_deferred(to_frs(x), to_pts(V), to_frs(x), to_pts(V), cm.Rand(4))
    assert residual(ok_e) is None and residual(ok_d) is None
    # mismatched lengths are the same structural error
    with pytest.raises(cm.CurdleError) as e:
        cm.MsmAccumulator().accumulate_check_deferred(to_frs(c), to_pts(P), to_frs(x[:3]), to_pts(V), cm.Rand(0))
    assert e.value.code == cm.EINVAL and "same length" in e.value.msg


def test_subgroup_check_of_the_decoder(cm, oracle):
    """gnark's Decoder / SetBytes reject curve points outside G1.  The host uses the
    endomorphism test [z^2] phi(P) + P == inf; it must agree with the definition [r] P == inf
    on subgroup points, on random curve points (cofactor ~2^126, so essentially never in G1)
    and on points of small order."""
    p, r = oracle.P, oracle.R
    rnd = random.Random(5)
    # G1 points decode with the check on
    for k in (1, 2, r - 1, rnd.randrange(r)):
        pt = oracle.scalar_mul(k, oracle.G1)
        got = cm.g1_decompress(oracle.compress(pt), True)
        assert oracle.jac_from_mont_limbs([int(v) for v in got]) == pt
    # random curve points
    found = 0
    while found < 6:
        x = rnd.randrange(p)
        rhs = (x * x * x + 4) % p
        y = pow(rhs, (p + 1) // 4, p)
        if y * y % p != rhs:
            continue
        found += 1
        pt = (x, y)
        in_g1 = oracle.scalar_mul(r, pt) is None
        assert not in_g1
        enc = oracle.compress(pt)
        assert oracle.jac_from_mont_limbs([int(v) for v in cm.g1_decompress(enc, False)]) == pt   # on the curve: fine unchecked
        with pytest.raises(cm.CurdleError):
            cm.g1_decompress(enc, True)
        # clearing the cofactor lands in G1 and is accepted
        h = 0x396C8C005555E1568C00AAAB0000AAAB
        cleared = oracle.scalar_mul(h, pt)
        assert oracle.scalar_mul(r, cleared) is None
        cm.g1_decompress(oracle.compress(cleared), True)
        # a point of order dividing the cofactor only (r * pt) is rejected unless it is infinity
        small = oracle.scalar_mul(r, pt)
        with pytest.raises(cm.CurdleError):
            cm.g1_decompress(oracle.compress(small), True)


def test_inner_product_known_answer_of_the_reference(cm, oracle):
    """The one hard-coded expected value the reference's tests hold on this path:
    common/util_test.go:10-27, IPA([1,2,3,4], [2,3,4,5]) == 40; plus the length-mismatch
    error of util.go:27-29 and a random vector against Python integers."""
    a = np.array([oracle.fr_to_mont_limbs(v) for v in (1, 2, 3, 4)], dtype=np.uint64)
    b = np.array([oracle.fr_to_mont_limbs(v) for v in (2, 3, 4, 5)], dtype=np.uint64)
    assert [int(v) for v in cm.fr_inner_product(a, b)] == oracle.fr_to_mont_limbs(40)
    with pytest.raises(cm.CurdleError):
        cm.fr_inner_product(a, b[:3])
    r = oracle.Rand(9)
    xs, ys = r.get_frs(17), r.get_frs(17)
    exp = sum(x * y for x, y in zip(xs, ys)) % oracle.R
    A = np.array([oracle.fr_to_mont_limbs(v) for v in xs], dtype=np.uint64)
    B = np.array([oracle.fr_to_mont_limbs(v) for v in ys], dtype=np.uint64)
    assert [int(v) for v in cm.fr_inner_product(A, B)] == oracle.fr_to_mont_limbs(exp)
    assert [int(v) for v in cm.fr_inner_product(A[:0], B[:0])] == oracle.fr_to_mont_limbs(0)
