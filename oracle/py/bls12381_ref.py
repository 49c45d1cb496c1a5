"""TEST INFRASTRUCTURE ONLY -- big-integer oracle for the BLS12-381 G1 MSM path.

This file is the pure-Python half of the oracle (see oracle/README.md).  Only
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it;
the product path (go-curdleproofs_amd/) never does.

PARITY UNPINNED (in the known-answer sense): the reference's tests hold no
golden vectors for this path and the reference (Go,
depending on the un-vendored gnark-crypto v0.11.0, go.mod:6) cannot be built
here.  The oracle is anchored instead on
  * the published BLS12-381 parameters (p, r, generator) checked arithmetically
    in `self_check()` (curve equation, [r]G = infinity, Montgomery constants),
  * the reference's own call-site contracts, restated function by function
    below with file:line citations,
  * the behavioural invariants of the reference's tests (tests/test_oracle.py).

What is restated, and from where (all paths relative to /root/reference):
  * gnark-crypto `(*G1Jac).MultiExp(points, scalars, cfg)` -- external,
    go.mod:6; contract as used at msmaccumulator/msmaccumulator.go:59:
    sum_i scalars[i]*points[i]; N=0 -> identity; (0,0) affine == infinity.
    Textbook affine add/double, no Pippenger, so it is obviously correct.
  * `common.Rand`  -- common/rand.go:19-113 (SHAKE256 DRBG, rejection sampling,
    GetG1Affine = GetFr * generator, GeneratePermutation with the 16-byte read).
  * `msmaccumulator.MsmAccumulator` -- msmaccumulator/msmaccumulator.go:11-64.
  * gnark memory layouts -- fr.Element = [4]uint64 Montgomery (R = 2^256),
    fp.Element = [6]uint64 Montgomery (R = 2^384), little-endian limbs
    (SURVEY.md section 8a).
"""
from __future__ import annotations

import hashlib
import struct

# ---------------------------------------------------------------------------
# Published BLS12-381 parameters
# ---------------------------------------------------------------------------
P = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB
R = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
B_COEFF = 4
GX = 0x17F1D3A73197D7942695638C4FA9AC0FC3688C4F9774B905A14E3A3F171BAC586C55E83FF97A1AEFFB3AF00ADB22C6BB
GY = 0x08B3F481E3AAA0F1A09E30ED741D8AE4FCF5E095D5D00AF600DB18CB2C04B3EDD03CC744A2888AE40CAA232946C5E7E1

FP_BITS = 384
FR_BITS = 256
R_FP = (1 << FP_BITS) % P          # Montgomery one in Fp
R_FR = (1 << FR_BITS) % R          # Montgomery one in Fr
R_FP_INV = pow(R_FP, -1, P)
R_FR_INV = pow(R_FR, -1, R)

INF = None                         # affine point at infinity
G1 = (GX, GY)


def self_check() -> None:
    """Arithmetic anchors for the constants above."""
    assert (GY * GY - GX * GX * GX - B_COEFF) % P == 0
    assert scalar_mul(R, G1) is INF
    assert R_FP == int(
        "15f65ec3fa80e4935c071a97a256ec6d77ce5853705257455f48985753c758baebf4000bc40c0002760900000002fffd", 16)
    assert R_FR == int("1824b159acc5056f998c4fefecbc4ff55884b7fa0003480200000001fffffffe", 16)
    assert (-pow(P, -1, 1 << 64)) % (1 << 64) == 0x89F3FFFCFFFCFFFD
    assert (-pow(R, -1, 1 << 64)) % (1 << 64) == 0xFFFFFFFEFFFFFFFF


# ---------------------------------------------------------------------------
# Affine group law (textbook)
# ---------------------------------------------------------------------------
def is_on_curve(pt) -> bool:
    if pt is INF:
        return True
    x, y = pt
    return (y * y - x * x * x - B_COEFF) % P == 0


def neg(pt):
    if pt is INF:
        return INF
    return (pt[0], (-pt[1]) % P)


def add(a, b):
    if a is INF:
        return b
    if b is INF:
        return a
    x1, y1 = a
    x2, y2 = b
    if x1 == x2:
        if (y1 + y2) % P == 0:
            return INF
        lam = (3 * x1 * x1) * pow(2 * y1, -1, P) % P
    else:
        lam = (y2 - y1) * pow(x2 - x1, -1, P) % P
    x3 = (lam * lam - x1 - x2) % P
    y3 = (lam * (x1 - x3) - y1) % P
    return (x3, y3)


def scalar_mul(k: int, pt):
    """Right-to-left double-and-add; k is a canonical non-negative integer."""
    acc = INF
    base = pt
    while k:
        if k & 1:
            acc = add(acc, base)
        base = add(base, base)
        k >>= 1
    return acc


def msm(points, scalars):
    """sum_i scalars[i] * points[i]  (canonical ints, affine points / INF).

    Contract of gnark-crypto G1Jac.MultiExp as used by the reference
    (msmaccumulator/msmaccumulator.go:59): length mismatch is an error, N=0
    gives the identity (msmaccumulator_test.go:14 exercises sizes 0..3).
    """
    if len(points) != len(scalars):
        raise ValueError("len(points) != len(scalars)")
    acc = INF
    for pt, k in zip(points, scalars):
        acc = add(acc, scalar_mul(k % R, pt))
    return acc


# ---------------------------------------------------------------------------
# gnark memory layouts (SURVEY.md section 8a "Data layouts")
# ---------------------------------------------------------------------------
def _limbs(v: int, n: int):
    return [(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(n)]


def _from_limbs(limbs) -> int:
    v = 0
    for i, l in enumerate(limbs):
        v |= int(l) << (64 * i)
    return v


def fr_to_mont_limbs(k: int):
    """Canonical scalar -> fr.Element ([4]uint64, Montgomery, little-endian limbs)."""
    return _limbs((k % R) * R_FR % R, 4)


def fr_from_mont_limbs(limbs) -> int:
    return _from_limbs(limbs) * R_FR_INV % R


def fp_to_mont_limbs(x: int):
    return _limbs((x % P) * R_FP % P, 6)


def fp_from_mont_limbs(limbs) -> int:
    return _from_limbs(limbs) * R_FP_INV % P


def affine_to_mont_limbs(pt):
    """G1Affine{X,Y} = 12 uint64; infinity == (0,0) (curdleproof.go:23 zeroPoint)."""
    if pt is INF:
        return [0] * 12
    return fp_to_mont_limbs(pt[0]) + fp_to_mont_limbs(pt[1])


def affine_from_mont_limbs(limbs):
    x = fp_from_mont_limbs(limbs[0:6])
    y = fp_from_mont_limbs(limbs[6:12])
    if x == 0 and y == 0:
        return INF
    return (x, y)


def jac_to_mont_limbs(pt):
    """Canonical Jacobian representative the C-ABI returns: (x, y, 1) or Z = 0.

    For infinity the library writes X = Y = one, Z = 0 (gnark's own convention
    for G1Jac infinity after `FromAffine`); consumers only test Z.
    """
    if pt is INF:
        one = _limbs(R_FP, 6)
        return one + one + [0] * 6
    return fp_to_mont_limbs(pt[0]) + fp_to_mont_limbs(pt[1]) + _limbs(R_FP, 6)


def jac_from_mont_limbs(limbs):
    """Any Jacobian representative (X, Y, Z Montgomery) -> affine / INF."""
    x = fp_from_mont_limbs(limbs[0:6])
    y = fp_from_mont_limbs(limbs[6:12])
    z = fp_from_mont_limbs(limbs[12:18])
    if z == 0:
        return INF
    zi = pow(z, -1, P)
    zi2 = zi * zi % P
    return (x * zi2 % P, y * zi2 * zi % P)


def compress(pt) -> bytes:
    """ZCash-style 48-byte compressed G1 (what gnark `point.Bytes()` emits,
    transcript/transcript.go:35).  UNVERIFIED against gnark."""
    if pt is INF:
        return bytes([0xC0]) + bytes(47)
    x, y = pt
    b = bytearray(x.to_bytes(48, "big"))
    b[0] |= 0x80
    if y > (P - 1) // 2:
        b[0] |= 0x20
    return bytes(b)


# ---------------------------------------------------------------------------
# common.Rand  (common/rand.go)
# ---------------------------------------------------------------------------
class Rand:
    """common/rand.go:13-33: SHAKE256 absorbing the 8-byte big-endian seed."""

    def __init__(self, seed: int):
        self._seed = struct.pack(">Q", seed)          # rand.go:20-21
        self._off = 0
        self._buf = b""

    def _read(self, n: int) -> bytes:
        # hashlib's shake has no streaming squeeze; re-squeeze a longer prefix.
        need = self._off + n
        if need > len(self._buf):
            size = max(4096, 2 * need)
            self._buf = hashlib.shake_256(self._seed).digest(size)   # rand.go:23
        out = self._buf[self._off:need]
        self._off = need
        return out

    def get_fr(self) -> int:
        """rand.go:35-47: 32 bytes, big-endian, rejected while >= r."""
        while True:
            v = int.from_bytes(self._read(32), "big")
            if v < R:
                return v

    def get_frs(self, n: int):
        return [self.get_fr() for _ in range(n)]               # rand.go:49-59

    def get_g1_affine(self):
        return scalar_mul(self.get_fr(), G1)                     # rand.go:72-83

    def get_g1_affines(self, n: int):
        return [self.get_g1_affine() for _ in range(n)]         # rand.go:85-95

    def generate_permutation(self, n: int):
        """rand.go:97-113: reads 16 bytes per step, uses the first two (BE u16)."""
        perm = list(range(n))
        for i in range(n):
            tmp = self._read(16)
            j = int.from_bytes(tmp[:2], "big") % (i + 1)
            perm[i], perm[j] = perm[j], perm[i]
        return perm


# ---------------------------------------------------------------------------
# msmaccumulator  (msmaccumulator/msmaccumulator.go)
# ---------------------------------------------------------------------------
class MsmAccumulator:
    """msmaccumulator.go:11-21.  Points are affine tuples / INF, scalars ints."""

    def __init__(self):
        self.A_c = INF
        self.base_scalar_map = {}

    def accumulate_check(self, C, x, v, rand: Rand) -> None:
        """msmaccumulator.go:23-47."""
        if len(v) != len(x):
            raise ValueError("x and v must have the same length")     # :28-30
        alpha = rand.get_fr()                                          # :32
        for xi, vi in zip(x, v):                                       # :38-43
            key = vi if vi is not INF else "inf"
            self.base_scalar_map[key] = (self.base_scalar_map.get(key, 0) + alpha * xi) % R
        self.A_c = add(self.A_c, scalar_mul(alpha, C))                 # :44

    def flatten(self):
        v = [k if k != "inf" else INF for k in self.base_scalar_map]
        x = [self.base_scalar_map[k] for k in self.base_scalar_map]
        return v, x                                                    # :50-56

    def verify(self) -> bool:
        v, x = self.flatten()
        return msm(v, x) == self.A_c                                   # :59-63


if __name__ == "__main__":
    self_check()
    print("bls12381_ref self_check OK")
