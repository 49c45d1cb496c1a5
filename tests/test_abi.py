"""C-ABI surface of libcurdlemsm.so without a GPU: the library loads, exports
every symbol include/curdle_msm.h declares, keeps the reference's error
conventions, and FAILS LOUDLY (no CPU fallback) when asked to compute without a
device."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "curdle_msm.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(curdle_[A-Za-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported(cm):
    declared = _declared_symbols()
    assert len(declared) >= 25
    lib = ctypes.CDLL(cm.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/curdle_msm.h but not exported"
    assert sorted(cm.SYMBOLS) == declared


def test_library_does_not_link_the_oracle(cm):
    # the product must not route through oracle/: no oracle symbol, no oracle dependency
    import subprocess
    out = subprocess.run(["nm", "-D", cm.LIB_PATH], capture_output=True, text=True).stdout
    assert "oracle_" not in out
    ldd = subprocess.run(["ldd", cm.LIB_PATH], capture_output=True, text=True).stdout
    assert "curdle_oracle" not in ldd
    for root, _, files in os.walk(os.path.join(ROOT, "go-curdleproofs_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", ".inc", ".go")):
                src = open(os.path.join(root, f), errors="ignore").read()
                assert "coracle" not in src and "bls12381_ref" not in src and "curdle_oracle" not in src, f


def test_empty_msm_is_identity_without_a_device(cm, oracle):
    # MultiExp contract: N = 0 -> identity, nil error (msmaccumulator_test.go:14 size 0)
    out = cm.msm_g1(np.zeros((0, 12), np.uint64), np.zeros((0, 4), np.uint64))
    assert [int(v) for v in out] == oracle.jac_to_mont_limbs(oracle.INF)


def test_length_mismatch_is_an_error(cm):
    with pytest.raises(cm.CurdleError) as e:
        cm.msm_g1(np.zeros((2, 12), np.uint64), np.zeros((1, 4), np.uint64))
    assert e.value.code == cm.EINVAL


def test_window_plan(cm):
    assert cm.num_windows(1 << 20, 16) == 16
    # c = 3, 5, 15 need one extra window: the top digit could otherwise borrow past the end
    assert cm.num_windows(10, 15) == 18 and cm.num_windows(10, 5) == 52 and cm.num_windows(10, 3) == 86
    assert cm.num_windows(10, 8) == 32 and cm.num_windows(10, 13) == 20
    for n in (1, 7, 308, 1268, 1 << 16, 1 << 20):
        c = cm.window_bits(n)
        assert 2 <= c <= 16
        assert cm.num_windows(n, c) * c >= 255
    with pytest.raises(cm.CurdleError):
        cm.num_windows(10, 17)


def test_signed_digit_windows_cover_every_scalar(cm, oracle):
    """Host restatement of the kernels' recoding: for every c the digits of r-1 (the
    largest scalar) and of awkward values reconstruct the scalar within num_windows(c)."""
    vals = [oracle.R - 1, oracle.R - 2, (1 << 254) + (1 << 253), (1 << 255) % oracle.R, 0x8000800080008000, 1, 0]
    for c in range(2, 17):
        W = cm.num_windows(10, c)
        half = 1 << (c - 1)
        for s in vals:
            carry, acc, v = 0, 0, s
            for w in range(W):
                raw = (v & ((1 << c) - 1)) + carry
                v >>= c
                if raw > half:
                    d, carry = raw - (1 << c), 1
                else:
                    d, carry = raw, 0
                assert -half <= d <= half
                acc += d << (c * w)
            assert carry == 0 and v == 0 and acc == s, (c, hex(s))


def test_g1_sum_host(cm, oracle):
    pts = oracle.Rand(9).get_g1_affines(5)
    jac = np.array([oracle.jac_to_mont_limbs(p) for p in pts] + [oracle.jac_to_mont_limbs(oracle.INF)], dtype=np.uint64)
    exp = oracle.INF
    for p in pts:
        exp = oracle.add(exp, p)
    assert [int(v) for v in cm.g1_sum(jac)] == oracle.jac_to_mont_limbs(exp)
    assert [int(v) for v in cm.g1_sum(np.zeros((0, 18), np.uint64))] == oracle.jac_to_mont_limbs(oracle.INF)
    # a non-canonical representative (x*z^2, y*z^3, z) is accepted
    z = 0x1234567
    x, y = pts[0]
    rep = oracle.fp_to_mont_limbs(x * z * z) + oracle.fp_to_mont_limbs(y * z ** 3) + oracle.fp_to_mont_limbs(z)
    assert [int(v) for v in cm.g1_sum(np.array([rep], dtype=np.uint64))] == oracle.jac_to_mont_limbs(pts[0])


@pytest.mark.skipif(os.environ.get("CURDLE_EXPECT_GPU") == "1", reason="GPU box")
def test_no_device_means_loud_failure_not_fallback(cm, oracle):
    if cm.device_available():
        pytest.skip("a device is visible")
    P = np.array([oracle.affine_to_mont_limbs(oracle.G1)], dtype=np.uint64)
    S = np.array([oracle.fr_to_mont_limbs(5)], dtype=np.uint64)
    with pytest.raises(cm.CurdleError) as e:
        cm.msm_g1(P, S)
    assert e.value.code == cm.ENODEV
    acc = cm.MsmAccumulator()
    acc.accumulate_check(np.array(oracle.jac_to_mont_limbs(oracle.G1), dtype=np.uint64), S, P, cm.Rand(0))
    with pytest.raises(cm.CurdleError) as e:
        acc.verify()
    assert e.value.code == cm.ENODEV and "computing msm" in e.value.msg  # msmaccumulator.go:60
