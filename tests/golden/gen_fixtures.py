#!/usr/bin/env python3
"""Generates the committed golden fixtures from the pure-Python big-integer
oracle (oracle/py/bls12381_ref.py).  No reference code is run: the reference is
Go on top of the un-vendored gnark-crypto and cannot be built in this image
("Oracle").  Inputs follow the reference's own test recipes:

  * msmaccumulator/msmaccumulator_test.go:12-50 -- NewRand(0); A = GetG1Affines(n);
    x = GetFrs(n); C1 = MSM(A, x); B, y, C2 likewise; two AccumulateCheck calls with
    the same rand; Verify() == true.  The Go loop `for n := range []int{1,4,8,16}`
    iterates INDICES, so the sizes are 0, 1, 2, 3.
  * the edge inputs listed in SURVEY.md appendix C (G3-G5): infinity base with a
    non-zero scalar, duplicate bases, P and -P, zero scalar, r-1, all-equal
    scalars (samepermutationargument.go:132-140), small scalars (common/util.go:68-75).

Run:  python tests/golden/gen_fixtures.py      (about a minute)
Outputs: tests/golden/msm_vectors.npz, accumulator_vectors.npz, rand_known_answers.json
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle", "py"))
import bls12381_ref as o  # noqa: E402


def pts_arr(pts):
    return np.array([o.affine_to_mont_limbs(p) for p in pts], dtype=np.uint64).reshape(len(pts), 12)


def frs_arr(frs):
    return np.array([o.fr_to_mont_limbs(s) for s in frs], dtype=np.uint64).reshape(len(frs), 4)


def jac_arr(pt):
    return np.array(o.jac_to_mont_limbs(pt), dtype=np.uint64)


def main():
    o.self_check()
    out = {}

    # --- random MSMs, inputs drawn like the reference's tests draw them ----------
    r = o.Rand(0)
    P = r.get_g1_affines(1024)
    S = r.get_frs(1024)
    for n in (0, 1, 2, 3, 16, 257, 1024):
        out[f"rand0_n{n}_points"] = pts_arr(P[:n])
        out[f"rand0_n{n}_scalars"] = frs_arr(S[:n])
        out[f"rand0_n{n}_expected"] = jac_arr(o.msm(P[:n], S[:n]))
        print("rand0 n", n, flush=True)

    # --- edge cases ---------------------------------------------------------------
    r = o.Rand(7)
    Q = r.get_g1_affines(16)
    T = r.get_frs(16)
    cases = {
        # infinity bases with non-zero scalars (curdleproof.go:281,285 T'/U' padding)
        "edge_infinity_bases": ([Q[0], o.INF, Q[1], o.INF, Q[2]], [T[0], T[1], T[2], 1, T[3]]),
        "edge_all_infinity": ([o.INF, o.INF, o.INF], [T[0], T[1], T[2]]),
        # duplicate bases (same bucket in some window -> doubling inside the bucket)
        "edge_duplicate_bases": ([Q[0], Q[0], Q[0], Q[1], Q[1]], [T[0], T[0], T[1], T[2], T[2]]),
        # P and -P with the same scalar (bucket sums hit infinity)
        "edge_opposite_points": ([Q[0], o.neg(Q[0]), Q[1]], [T[0], T[0], T[1]]),
        "edge_cancels_to_infinity": ([Q[0], o.neg(Q[0])], [T[0], T[0]]),
        # zero scalars, scalar one, scalar r-1
        "edge_zero_scalars": (Q[:4], [0, 0, 0, 0]),
        "edge_extreme_scalars": (Q[:6], [0, 1, o.R - 1, o.R - 2, 2, (o.R - 1) // 2]),
        # all-equal scalars (samepermutationargument.go:67, :140: every scalar is beta)
        "edge_all_equal_scalars": (Q[:16], [T[5]] * 16),
        # tiny scalars perm(i) < ell (common/util.go:68-75)
        "edge_small_scalars": (Q[:16], list(o.Rand(3).generate_permutation(16))),
        # powers of two around window boundaries (signed-digit carries)
        "edge_window_boundaries": (Q[:12], [(1 << 15), (1 << 15) + 1, (1 << 16) - 1, (1 << 16), (1 << 127),
                                            (1 << 128) - 1, (1 << 254), (1 << 254) + (1 << 253), 0x8000800080008000,
                                            (1 << 240) - 1, 0x7FFF, (1 << 255) % o.R]),
    }
    for name, (pts, frs) in cases.items():
        out[name + "_points"] = pts_arr(pts)
        out[name + "_scalars"] = frs_arr(frs)
        out[name + "_expected"] = jac_arr(o.msm(pts, frs))
        print(name, flush=True)
    np.savez_compressed(os.path.join(HERE, "msm_vectors.npz"), **out)

    # --- accumulator flow of msmaccumulator_test.go -------------------------------
    acc = {}
    for n in (0, 1, 2, 3):
        r = o.Rand(0)
        A = r.get_g1_affines(n)
        x = r.get_frs(n)
        C1 = o.msm(A, x)
        B = r.get_g1_affines(n)
        y = r.get_frs(n)
        C2 = o.msm(B, y)
        ma = o.MsmAccumulator()
        ma.accumulate_check(C1, x, A, r)
        ma.accumulate_check(C2, y, B, r)
        assert ma.verify()
        v, sc = ma.flatten()
        acc[f"n{n}_A"] = pts_arr(A)
        acc[f"n{n}_x"] = frs_arr(x)
        acc[f"n{n}_C1"] = jac_arr(C1)
        acc[f"n{n}_B"] = pts_arr(B)
        acc[f"n{n}_y"] = frs_arr(y)
        acc[f"n{n}_C2"] = jac_arr(C2)
        acc[f"n{n}_A_c"] = jac_arr(ma.A_c)
        acc[f"n{n}_map_points"] = pts_arr(v)
        acc[f"n{n}_map_scalars"] = frs_arr(sc)
        print("accumulator n", n, flush=True)
    np.savez_compressed(os.path.join(HERE, "accumulator_vectors.npz"), **acc)

    # --- common.Rand known answers -------------------------
    ka = {
        "seed0_first_fr": hex(o.Rand(0).get_fr()),
        "seed42_first_fr": hex(o.Rand(42).get_fr()),
        "seed43_first_fr": hex(o.Rand(43).get_fr()),
        "seed0_first_g1_compressed": o.compress(o.Rand(0).get_g1_affine()).hex(),
        "seed0_frs_8": [hex(v) for v in o.Rand(0).get_frs(8)],
        "seed42_permutation_10": o.Rand(42).generate_permutation(10),
        "seed0_permutation_124": o.Rand(0).generate_permutation(124),
    }
    with open(os.path.join(HERE, "rand_known_answers.json"), "w") as f:
        json.dump(ka, f, indent=1)
    print("done")


if __name__ == "__main__":
    main()
