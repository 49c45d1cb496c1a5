"""Adversarial scalar / base families of SURVEY.md 8(d) ("Adversarial sets also timed: all-equal scalars
(a11), <= 9-bit scalars (a13), 1 % infinity bases") and three more that aim at the bucket sort
(few distinct values, one hot window, half of all terms equal), with the exact expected result of an
MSM over the walk points P_i = (k + i q) G computed WITHOUT any group arithmetic per term:

    sum s_i P_i = (k * S0 + q * S1) G,   S0 = sum s_i,  S1 = sum i s_i   (mod r)

and Montgomery form is linear (s = m / 2^256 mod r), so S0 and S1 come straight from the limbs the
library is handed -- sixteen 16-bit columns summed in uint64 by numpy (exact up to 2^24 pairs), put
together as Python integers.  Shared by tools/sweep_adversarial.py and bench.py's "adversarial" leg.

Where the reference produces such inputs: samepermutationargument/samepermutationargument.go:67,132-140
(every scalar = beta), common/util.go:68-75 (scalars = perm(i) < ell <= 512), curdleproof.go:281,285
(zero points padding T', U')."""
import numpy as np

R_MOD = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
R_INV = pow(1 << 256, -1, R_MOD)
LAMBDA = 0xAC45A4010001A40200000000FFFFFFFF          # z^2 - 1: phi(P) = lambda P (csrc/bls12_381.h glv_split)
FAMILIES = ("uniform", "all_equal", "small_9bit", "infinity_1pct", "distinct_64", "distinct_600", "distinct_1024", "hot_window", "half_equal")


def to_mont(v: int) -> np.ndarray:
    m = (v << 256) % R_MOD
    return np.array([(m >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)], dtype=np.uint64)


def limb_sums(sc: np.ndarray, index_from: int = 0):
    """(sum m_i, sum (index_from + i) m_i) over the rows of uint64[n, 4], as Python integers."""
    n = len(sc)
    assert n <= (1 << 24)
    idx = np.arange(index_from, index_from + n, dtype=np.uint64)
    s0 = s1 = 0
    for limb in range(4):
        col = sc[:, limb]
        for part in range(4):
            piece = (col >> np.uint64(16 * part)) & np.uint64(0xFFFF)
            sh = 64 * limb + 16 * part
            s0 += int(piece.sum(dtype=np.uint64)) << sh
            s1 += int((piece * idx).sum(dtype=np.uint64)) << sh
    return s0, s1


def walk_exponent(k: int, q: int, sc: np.ndarray, dead=None) -> int:
    """e with sum s_i P_i = e G for P_i = (k + i q) G and Montgomery-form scalars sc; `dead` = indices
    whose base is the point at infinity (their terms drop out)."""
    s0, s1 = limb_sums(sc)
    if dead is not None and len(dead):
        d = np.unique(np.asarray(dead))
        for i in d:
            m = sum(int(v) << (64 * j) for j, v in enumerate(sc[int(i)]))
            s0 -= m
            s1 -= int(i) * m
    return (k * (s0 % R_MOD) + q * (s1 % R_MOD)) % R_MOD * R_INV % R_MOD


def make_family(name: str, n: int, uniform: np.ndarray, window_bits: int = 16, seed: int = 6):
    """(scalars uint64[n, 4] in Montgomery form, dead) -- `dead` = None or the indices whose BASE the
    caller must overwrite with the point at infinity (96 zero bytes).  `uniform` = n uniform scalars."""
    rng = np.random.default_rng(seed)
    if name == "uniform":
        return uniform[:n], None
    if name == "all_equal":            # samepermutationargument.go:67: every scalar = beta
        beta = int.from_bytes(rng.bytes(32), "little") % R_MOD
        return np.tile(to_mont(beta), (n, 1)), None
    if name == "small_9bit":           # common/util.go:75: scalars = perm(i) < ell <= 512
        table = np.stack([to_mont(v) for v in range(512)])
        return table[rng.integers(0, 512, n)], None
    if name == "infinity_1pct":        # curdleproof.go:281,285 pads with zero points; here 1 % of all bases
        return uniform[:n], rng.choice(n, max(1, n // 100), replace=False)
    if name.startswith("distinct_"):   # K distinct scalar values: K..2K occupied buckets per window
        kk = int(name.split("_")[1])
        table = np.stack([to_mont(int.from_bytes(rng.bytes(32), "little") % R_MOD) for _ in range(kk)])
        return table[rng.integers(0, kk, n)], None
    if name == "half_equal":           # every other term shares every digit; the rest uniform
        sc = uniform[:n].copy()
        sc[::2] = to_mont((7 << 250) % R_MOD)
        return sc, None
    if name == "hot_window":
        # k = k1 + lambda k2 with k1, k2 uniform below 2^126 except window 3 of BOTH halves, which holds the
        # same digit for every term: one bucket of one window (two with the recoding's carry) receives 2 n
        # entries, everything else is uniform but for the top window's highest bit.  (0 <= k1 < lambda / 2
        # and k < 2^253.5 <= (r - 1) / 2, so the library's centred split returns exactly these halves.)
        c, w = window_bits, 3
        mask = ((1 << c) - 1) << (c * w)
        hot = 0x2B67 & ((1 << c) - 1) & ~1 | 2
        raw = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64)
        out = np.empty((n, 4), dtype=np.uint64)
        for i in range(n):
            k1 = (int(raw[i, 0]) | (int(raw[i, 1]) << 63)) & ~mask | (hot << (c * w))
            k2 = (int(raw[i, 2]) | (int(raw[i, 3]) << 63)) & ~mask | (hot << (c * w))
            m = (((k1 + LAMBDA * k2) % R_MOD) << 256) % R_MOD
            out[i, 0] = m & 0xFFFFFFFFFFFFFFFF
            out[i, 1] = (m >> 64) & 0xFFFFFFFFFFFFFFFF
            out[i, 2] = (m >> 128) & 0xFFFFFFFFFFFFFFFF
            out[i, 3] = m >> 192
        return out, None
    raise ValueError(name)
