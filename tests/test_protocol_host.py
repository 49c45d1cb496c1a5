"""Host-side protocol pieces that need no MSM: the Merlin transcript against its
published test vector, gnark's compressed G1 encoding against the oracle, malformed
encodings.  (The full Prove / Verify flows need the GPU MSM: tests/test_protocol_gpu.py.)"""
import numpy as np
import pytest


def test_merlin_published_test_vector(cm):
    # merlin's own "equivalence_simple" test (quoted in SURVEY.md section 8c): STROBE-128 / Keccak-f[1600]
    out = cm.merlin_test_vector(b"test protocol", b"some label", b"some data", b"challenge", 32)
    assert out.hex() == "d5a21972d0d5fe320c0d263fac7fffb8145aa640af6e9bca177c03c7efcf0615"


def test_compressed_g1_matches_oracle(cm, oracle):
    r = oracle.Rand(0)
    pts = r.get_g1_affines(6) + [oracle.neg(oracle.G1), oracle.INF]
    for p in pts:
        jac = np.array(oracle.jac_to_mont_limbs(p), dtype=np.uint64)
        enc = cm.g1_compress(jac)
        assert enc == oracle.compress(p)
        back = cm.g1_decompress(enc, True)
        assert [int(v) for v in back] == oracle.jac_to_mont_limbs(p)
    assert cm.g1_compress(np.array(oracle.jac_to_mont_limbs(pts[0]), dtype=np.uint64)).hex() == (
        "b058e2c67ce70d724988ddfb90b744d65d03df778ecf68eeb0a0d5713e34fe51a53fa09820b7068cf427c2ae3ba25305")


def test_malformed_encodings_are_rejected(cm, oracle):
    good = bytearray(oracle.compress(oracle.G1))
    cases = []
    b = bytearray(good); b[0] &= 0x7F; cases.append(bytes(b))            # compression flag cleared
    b = bytearray(good); b[0] |= 0x40; cases.append(bytes(b))            # infinity flag on a finite point
    cases.append(bytes([0xC0] + [0] * 46 + [1]))                         # infinity with junk
    cases.append(bytes([0x9F]) + b"\xff" * 47)                           # x >= p
    x = 0
    while True:                                                          # an x with no point on the curve
        x += 1
        y2 = (x ** 3 + 4) % oracle.P
        if pow(y2, (oracle.P - 1) // 2, oracle.P) != 1:
            break
    enc = bytearray(x.to_bytes(48, "big")); enc[0] |= 0x80
    cases.append(bytes(enc))
    for c in cases:
        with pytest.raises(cm.CurdleError):
            cm.g1_decompress(c, False)
    # a curve point outside the r-torsion is caught by the subgroup check only
    x = 0
    while True:
        x += 1
        y2 = (x ** 3 + 4) % oracle.P
        if pow(y2, (oracle.P - 1) // 2, oracle.P) == 1:
            y = pow(y2, (oracle.P + 1) // 4, oracle.P)
            if oracle.scalar_mul(oracle.R, (x, y)) is not oracle.INF:
                break
    enc = bytearray(x.to_bytes(48, "big")); enc[0] |= 0x80
    if y > (oracle.P - 1) // 2:
        enc[0] |= 0x20
    cm.g1_decompress(bytes(enc), False)
    with pytest.raises(cm.CurdleError):
        cm.g1_decompress(bytes(enc), True)
