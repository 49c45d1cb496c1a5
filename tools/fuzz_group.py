"""Randomised differential test of the two other GPU kernels against independent code paths:
  * curdle_g1_decompress_batch vs the host decoder (curdle_g1_decompress), on valid encodings,
    random byte strings, flipped flag bits and x >= p;
  * curdle_g1_scalar_mul_batch (double-and-add) vs size-1 MSMs through the bucket method
    (curdle_msm_g1_batch) plus curdle_g1_sum for the addend.
    python tools/fuzz_group.py [seconds] [seed]
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "go-curdleproofs_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle", "py"))
import numpy as np
import curdlemsm as cm
import bls12381_ref as o
import coracle as co

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
cm.init(0)
k0, q0 = o.Rand(9).get_frs(2)
pool = co.points_walk(k0, q0, 2048)
ONE = np.array([0x760900000002fffd, 0xebf4000bc40c0002, 0x5f48985753c758ba, 0x77ce585370525745, 0x5c071a97a256ec6d,
                0x15f65ec3fa80e493], dtype=np.uint64)
enc_pool = [cm.g1_compress(np.concatenate([p, ONE])) for p in pool[:512]]
t_end = time.time() + budget
dec_cases = smul_cases = 0
while time.time() < t_end:
    # ---- decode ----
    n = int(rng.choice([1, 5, 48, 100, 594, 3000]))
    recs = []
    for _ in range(n):
        kind = int(rng.integers(0, 6))
        e = bytearray(enc_pool[int(rng.integers(0, 512))])
        if kind == 1:
            e = bytearray(rng.bytes(48))                      # random bytes (mostly not on the curve)
        elif kind == 2:
            e[0] ^= 0x20                                      # the other root
        elif kind == 3:
            e[0] ^= 1 << int(rng.integers(5, 8))              # a flag bit
        elif kind == 4:
            e[int(rng.integers(1, 48))] ^= 1 << int(rng.integers(0, 8))   # another x
        recs.append(bytes(e))
    pts, st = cm.g1_decompress_batch(b"".join(recs), True)
    for i, e in enumerate(recs):
        try:
            h = cm.g1_decompress(e, True)
            ok = True
        except cm.CurdleError:
            ok = False
        if ok != (st[i] in (cm.DECODE_OK, cm.DECODE_INFINITY)):
            print("DECODE MISMATCH (accept)", e.hex(), st[i], ok); sys.exit(1)
        if ok and st[i] == cm.DECODE_OK and not (h[:12] == pts[i]).all():
            print("DECODE MISMATCH (point)", e.hex()); sys.exit(1)
    dec_cases += n
    # ---- scalar mul ----
    n = int(rng.choice([1, 3, 24, 100, 700]))
    P = pool[rng.integers(0, 2048, size=n)].copy()
    A = pool[rng.integers(0, 2048, size=n)].copy()
    P[rng.integers(0, 30, size=n) == 0] = 0
    A[rng.integers(0, 30, size=n) == 0] = 0
    sc = rng.integers(0, 1 << 64, size=(n, 4), dtype=np.uint64)
    sc[:, 3] &= np.uint64((1 << 61) - 1)
    small = rng.integers(0, 4, size=n) == 0
    sc[small, 1:] = 0
    shared = bool(rng.integers(0, 2))
    use = sc[0] if shared else sc
    with_add = bool(rng.integers(0, 2))
    got = cm.g1_scalar_mul_batch(P, use, A if with_add else None)
    full = np.repeat(sc[:1], n, axis=0) if shared else sc
    ref = cm.msm_g1_batch(P, full, np.arange(n + 1, dtype=np.uint64))     # n MSMs of one pair each
    for i in range(n):
        want = ref[i]
        if with_add:
            want = cm.g1_sum(np.stack([ref[i], np.concatenate([A[i], ONE if A[i].any() else np.zeros(6, dtype=np.uint64)])]) if A[i].any() else ref[i:i + 1])
        want_aff = want[:12] if want[12:].any() else np.zeros(12, dtype=np.uint64)
        if not (got[i] == want_aff).all():
            print("SMUL MISMATCH", i, n, shared, with_add); sys.exit(1)
    smul_cases += n
print(f"fuzz_group: {dec_cases} decoded records and {smul_cases} scalar multiplications agree (seed {seed}, {budget:.0f} s)")
