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


GLV_LAMBDA = 0xac45a4010001a40200000000ffffffff      # z^2 - 1, z = 0xd201000000010000: lambda^2 + lambda + 1 = r


def test_device_selection_and_multi_device_entry_points_check_their_arguments(cm, oracle):
    """The multi-device surface without a device: argument errors are CURDLE_EINVAL whatever the
    hardware, the empty MSM is the identity, and configuring devices that do not exist fails
    loudly with CURDLE_ENODEV -- never a silent single-device or CPU fallback."""
    import ctypes as C
    lib = C.CDLL(cm.LIB_PATH)
    lib.curdle_init_devices.argtypes = [C.POINTER(C.c_int), C.c_int]
    lib.curdle_msm_g1_replicated.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
    assert lib.curdle_init_devices(None, 0) == cm.EINVAL
    too_many = (C.c_int * 17)(*([0] * 17))
    assert lib.curdle_init_devices(too_many, 17) == cm.EINVAL
    assert cm.device_count() >= 1 and 0 <= cm.get_device() < cm.device_count()
    with pytest.raises(cm.CurdleError) as e:
        cm.set_device(cm.device_count())
    assert e.value.code == cm.EINVAL
    with pytest.raises(cm.CurdleError):
        cm.set_device(-2)
    cm.set_device(-1)                      # "no selection": context 0, host MSMs may spread over all devices again
    assert cm.get_device() == 0
    out = np.zeros(18, dtype=np.uint64)
    assert lib.curdle_msm_g1_replicated(None, None, 5, 0, None) == cm.EINVAL          # no output
    assert lib.curdle_msm_g1_replicated(None, None, 5, 3, out.ctypes.data) == cm.EINVAL  # no such split
    assert lib.curdle_msm_g1_replicated(None, None, 0, 0, out.ctypes.data) == cm.OK      # empty: the identity
    assert [int(v) for v in out] == oracle.jac_to_mont_limbs(oracle.INF)
    assert lib.curdle_msm_g1_replicated(None, None, 5, 1, out.ctypes.data) == cm.EINVAL  # null inputs
    if not cm.device_available():
        two = (C.c_int * 2)(0, 0)
        assert lib.curdle_init_devices(two, 2) == cm.ENODEV
        assert "no HIP device" in cm.last_error()


def glv_split(k, R):
    """Host restatement of the kernels' split (bls12_381.h glv_split): k = s * (k1 + k2 *
    lambda) mod r with s = -1 for k > (r - 1) / 2, k2 = round(k' / lambda) >= 0 and k1 in
    [-lambda / 2, lambda / 2).  Returns the two signed halves."""
    s = -1 if k > (R - 1) // 2 else 1
    kp = R - k if s < 0 else k
    k2 = (kp + (GLV_LAMBDA >> 1)) // GLV_LAMBDA
    k1 = kp - k2 * GLV_LAMBDA
    return s * k1, s * k2


def test_knobs_are_read_once_and_changed_only_through_the_hook(cm):
    """The tunables live in one table (host/knobs.h), read from the environment ONCE; afterwards only
    curdle_plan_override changes them (a dozen getenv calls on every MSM, next to a
    document that said there were none)."""
    n = 1 << 20
    assert cm.window_bits(n) == 16
    os.environ["CURDLE_WINDOW_BITS"] = "9"              # too late: the table was loaded at the first question
    try:
        assert cm.window_bits(n) == 16
    finally:
        del os.environ["CURDLE_WINDOW_BITS"]
    with cm.knobs(WINDOW_BITS=9):
        assert cm.window_bits(n) == 9 and cm.num_windows(n) == 15
    assert cm.window_bits(n) == 16
    cm.plan_override("CURDLE_WINDOW_BITS", 12)          # the prefixed spelling names the same knob
    assert cm.window_bits(n) == 12
    cm.plan_override("WINDOW_BITS", None)
    assert cm.window_bits(n) == 16
    for bad in ("NO_SUCH_KNOB", "", "CURDLE_TWO_ROUNDS", "CURDLE_SYNC_STREAMS", "SCAN", "FRONT", "HOST_GRADED", "PIPE_LANES"):   # closed experiments are gone
        with pytest.raises(cm.CurdleError):
            cm.plan_override(bad, 1)
    import subprocess, sys
    child = ("import sys; sys.path.insert(0, %r); import curdlemsm as cm; print(cm.window_bits(1 << 20))"
             % os.path.join(ROOT, "go-curdleproofs_amd"))
    out = subprocess.run([sys.executable, "-c", child], env=dict(os.environ, CURDLE_WINDOW_BITS="11"), capture_output=True, text=True)
    assert out.stdout.strip() == "11", (out.stdout, out.stderr[-500:])   # at process start the environment does count


def test_selftest_operations_share_one_table(cm):
    """curdle_selftest_op's buffer sizes, its launcher's grid and its kernel's indexing come from ONE
    table, which the binding asks for too: an unknown operation is refused before anything is
    allocated or launched (round 3's r3a abort was an operation known to one of them only)."""
    shapes = [cm.selftest_shape(op) for op in range(13)]
    assert shapes[:5] == [(24, 12)] * 4 + [(16, 8)]
    assert shapes[5:11] == [(96, 48)] * 6 and shapes[11] == (8, 10) and shapes[12] == (24, 26)
    for bad in (-1, 13, 99):
        with pytest.raises(RuntimeError):
            cm.selftest_shape(bad)
    inp = np.zeros((3, 8), dtype=np.uint32)
    out = np.zeros((3, 10), dtype=np.uint32)
    for on_device in (0, 1):
        assert cm._selftest_op(13, cm._ptr(inp), 3, cm._ptr(out), on_device) == cm.EINVAL
    # the host build of every operation runs on exactly the table's widths
    for op, (iw, ow) in enumerate(shapes):
        if op in (8, 9, 10):
            continue            # need valid points: covered on the GPU
        assert cm.selftest_op(op, np.zeros((2, iw), dtype=np.uint32), False).shape == (2, ow)


def test_window_plan(cm):
    # the windows cover ONE half of the split: 127 bits
    assert cm.num_windows(1 << 20, 16) == 8
    assert cm.window_widths(1 << 20, 16) == [16] * 7 + [15]
    assert cm.window_widths(10, 15) == [15] + [14] * 8
    assert cm.window_widths(10, 14) == [13] * 7 + [12] * 3
    for c in range(4, 17):
        w = cm.window_widths(10, c)
        assert len(w) == cm.num_windows(10, c) == -(-127 // c)
        assert sum(w) == 127 and max(w) <= c and max(w) - min(w) <= 1
        assert w[-1] == min(w) and w[-1] <= 15          # top window: narrowest, unsigned, fits the LDS histogram
    for n in (1, 7, 308, 1268, 1 << 16, 1 << 20):
        assert 4 <= cm.window_bits(n) <= 16
    for bad in (3, 17):
        with pytest.raises(cm.CurdleError):
            cm.num_windows(10, bad)
    # a call that passes CURDLE_MSM_ANY_CURVE_POINT recodes the whole 255-bit scalar: twice the windows, and the window
    # ranges of *_windows_ex / *_submit_ex are over THOSE (review of round 5)
    F = cm.MSM_ANY_CURVE_POINT
    for c in range(4, 17):
        w = cm.window_widths(10, c, flags=F)
        assert len(w) == cm.num_windows(10, c, flags=F) == -(-255 // c)
        assert sum(w) == 255 and max(w) <= c and max(w) - min(w) <= 1
    assert cm.num_windows(1 << 20, 0, flags=cm.MSM_BASES_UNCHANGED) == cm.num_windows(1 << 20)
    with pytest.raises(cm.CurdleError):
        cm.num_windows(10, 8, flags=64)


def test_submit_refuses_a_window_width_outside_the_plan_before_it_touches_anything(cm):
    """Review of round 5: curdle_msm_g1_device_submit_ex computed the window count of (n, window_bits) into a
    64-entry array BEFORE the width was validated -- window_bits = 1 has 127 windows (255 without the split), a
    large negative one none (a division by zero).  Now every width outside [4, 16] is CURDLE_EINVAL before a slot,
    a cache entry or the device is touched (so this runs without a GPU: the pointers are never read)."""
    fake = 0x1000
    for flags in (0, cm.MSM_ANY_CURVE_POINT):
        for bad in (1, 2, 3, -200, -126, 17, 1 << 20):
            with pytest.raises(cm.CurdleError) as e:
                cm.msm_g1_device_submit(fake, fake, 64, window_bits=bad, flags=flags)
            assert e.value.code == cm.EINVAL, (bad, flags)
            assert "window_bits" in str(e.value)


def recode(s, widths):
    """Host restatement of the kernels' recoding of ONE half of the split (msm_sort_kernels.hip
    for_each_digit): sign taken out, signed digits below the top window, unsigned top window.
    Returns [(digit, shift)]."""
    sign = -1 if s < 0 else 1
    out, carry, v, shift = [], 0, abs(s), 0
    for w, c in enumerate(widths):
        raw = (v & ((1 << c) - 1)) + carry
        v >>= c
        carry = 0
        d = raw
        if w != len(widths) - 1 and raw > (1 << (c - 1)):
            d, carry = raw - (1 << c), 1
        out.append((sign * d, shift))
        shift += c
    assert carry == 0 and v == 0
    return out


def test_digit_recoding_covers_every_scalar(cm, oracle):
    """For every window plan the split and the digits of r-1 (the largest scalar), of the
    scalars around (r - 1) / 2 and the multiples of lambda, and of awkward values reconstruct the
    scalar, stay inside the bucket range of their window and leave no carry behind."""
    R, lam = oracle.R, GLV_LAMBDA
    assert (lam * lam + lam + 1) % R == 0
    vals = [R - 1, R - 2, (1 << 254) + (1 << 253), (1 << 255) % R, 0x8000800080008000, 1, 0,
            (1 << 254) - 1, int("5" * 63, 16) % R, (R - 1) // 2, (R - 1) // 2 + 1, lam, lam - 1, lam + 1,
            lam >> 1, (lam >> 1) + 1, (lam // 2) * lam, (lam // 2) * lam + (lam >> 1) - 1]
    rng = np.random.default_rng(5)
    vals += [int.from_bytes(rng.bytes(32), "big") % R for _ in range(200)]
    for c in range(4, 17):
        widths = cm.window_widths(10, c)
        for s in vals:
            k1, k2 = glv_split(s, R)
            assert (k1 + k2 * lam - s) % R == 0 and abs(k1) < 1 << 127 and abs(k2) < 1 << 127, hex(s)
            for half in (k1, k2):
                digs = recode(half, widths)
                assert sum(d << sh for d, sh in digs) == half, (c, hex(s))
                for w, (d, _) in enumerate(digs):
                    if w == len(widths) - 1:
                        assert abs(d) <= (1 << widths[w]), (c, w)       # 2^b slots, index |d| - 1
                    else:
                        assert abs(d) <= 1 << (widths[w] - 1), (c, w)   # 2^(b-1) slots


def glv_split_cases(R, n_random, seed):
    """Scalars for the direct test of the C routine: every branch boundary of the split (the sign
    flip at (r - 1) / 2, the rounding of k' / lambda at every multiple of lambda +- lambda / 2
    near both ends and in the middle, the Barrett quotient's correction steps) plus n_random
    uniform ones."""
    lam = GLV_LAMBDA
    vals = {0, 1, 2, R - 1, R - 2, (R - 1) // 2, (R - 1) // 2 + 1, (R - 1) // 2 - 1, (R + 1) // 2 + 1}
    mults = list(range(0, 40)) + [lam // 2 - 1, lam // 2, lam // 2 + 1, lam - 2, lam - 1, lam, (R // 2) // lam - 1,
                                  (R // 2) // lam, 1 << 126, (1 << 126) - 1, (1 << 127) % lam]
    for q in mults:
        for d in (-2, -1, 0, 1, 2):
            for base in (q * lam, q * lam + (lam >> 1), q * lam + (lam >> 1) + 1):
                v = base + d
                if 0 <= v < R:
                    vals.add(v)
                    vals.add(R - 1 - v)
    for e in range(1, 255):
        for d in (-1, 0, 1):
            vals.add(((1 << e) + d) % R)
    rng = np.random.default_rng(seed)
    raw = rng.integers(0, 1 << 32, size=(n_random, 8), dtype=np.uint64)
    rnd = [sum(int(x) << (32 * j) for j, x in enumerate(row)) % R for row in raw]
    return sorted(vals) + rnd


def check_glv_split_outputs(vals, out, R):
    """out[i] = curdle_selftest_op(11) of vals[i]: compared with big-integer division."""
    lam = GLV_LAMBDA
    worst = 0
    for s, o in zip(vals, out):
        a = sum(int(o[j]) << (32 * j) for j in range(4))
        b = sum(int(o[4 + j]) << (32 * j) for j in range(4))
        assert int(o[8]) in (0, 0x80000000) and int(o[9]) in (0, 0x80000000), hex(s)
        k1 = -a if int(o[8]) else a
        k2 = -b if int(o[9]) else b
        e1, e2 = glv_split(s, R)                  # floor division on Python integers
        assert (k1, k2) == (e1, e2) or (a == 0 and abs(k1) == abs(e1) and k2 == e2), hex(s)
        assert (k1 + k2 * lam - s) % R == 0, hex(s)
        assert a < 1 << 127 and b < 1 << 127, hex(s)      # the top window's bucket index stays in range
        worst = max(worst, a, b)
    return worst


def scalars_as_words(vals):
    return np.array([[(v >> (32 * j)) & 0xFFFFFFFF for j in range(8)] for v in vals], dtype=np.uint32)


def test_the_c_split_routine_against_big_integer_division(cm, oracle):
    """glv_split itself (csrc/bls12_381.h: Barrett division with two correction steps, shared by
    k_digits and the host's scalar multiplication) -- not a Python restatement of it -- against
    big-integer division on 300,000 random scalars and every boundary: an out-of-range half would
    index past the LDS histogram of k_hist, not merely give a wrong sum.  Host build here, the
    device build in tests/test_msm_gpu.py::test_the_split_on_the_device."""
    vals = glv_split_cases(oracle.R, 300_000, 11)
    out = cm.selftest_op(11, scalars_as_words(vals), False)
    worst = check_glv_split_outputs(vals, out, oracle.R)
    assert worst < int(1.35 * (1 << 126))             # the bound DESIGN.md quotes


def test_host_scalar_multiplication_through_the_split(cm, oracle):
    """curdle_host_scalar_mul (host code: every single scalar multiplication of the protocol
    layers) runs on the same GLV split as the kernels: the scalars where the split's branches
    flip, and random ones, on the generator and on a second point, against the oracle."""
    import ctypes as C
    lib = C.CDLL(cm.LIB_PATH)
    f = lib.curdle_host_scalar_mul
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    f.restype = None
    R, lam = oracle.R, GLV_LAMBDA
    half = (R - 1) // 2
    vals = [0, 1, 2, R - 1, half, half + 1, lam, lam - 1, lam + 1, lam >> 1, (lam >> 1) + 1, R - lam, (lam >> 1) * lam,
            (lam >> 1) * lam + (lam >> 1), (lam >> 1) * lam + (lam >> 1) - 1, 3 * lam - 1, 1 << 127, 1 << 128, 1 << 254]
    rng = np.random.default_rng(11)
    vals += [int.from_bytes(rng.bytes(32), "big") % R for _ in range(40)]
    one = oracle.fp_to_mont_limbs(1)
    for base in (oracle.G1, oracle.scalar_mul(0xC0FFEE, oracle.G1)):
        aff = oracle.affine_to_mont_limbs(base)
        p = np.array(list(aff) + list(one) + list(one), dtype=np.uint64)      # X | Y | ZZ | ZZZ, Montgomery
        for k in vals:
            kk = np.array([(k >> (32 * i)) & 0xFFFFFFFF for i in range(8)], dtype=np.uint32)
            out = np.zeros(24, dtype=np.uint64)
            f(out.ctypes.data, p.ctypes.data, kk.ctypes.data)
            X, Y, ZZ, ZZZ = (oracle.fp_from_mont_limbs([int(v) for v in out[6 * i: 6 * i + 6]]) for i in range(4))
            exp = oracle.scalar_mul(k, base)
            if exp is None:
                assert ZZ == 0, hex(k)
            else:
                assert ZZ != 0, hex(k)
                got = (X * pow(ZZ, -1, oracle.P) % oracle.P, Y * pow(ZZZ, -1, oracle.P) % oracle.P)   # x = X / ZZ, y = Y / ZZZ
                assert got == exp, hex(k)


def test_host_compression_of_the_generator(cm, oracle):
    """curdle_g1_compress (host code: what the transcript hashes) on the one published value of
    the encoding, and its negative."""
    from test_oracle import G1_GENERATOR_COMPRESSED, G1_INFINITY_COMPRESSED
    g = np.array(oracle.jac_to_mont_limbs(oracle.G1), dtype=np.uint64)
    assert cm.g1_compress(g) == G1_GENERATOR_COMPRESSED
    assert cm.g1_compress(np.array(oracle.jac_to_mont_limbs(oracle.INF), dtype=np.uint64)) == G1_INFINITY_COMPRESSED
    n = np.array(oracle.jac_to_mont_limbs(oracle.neg(oracle.G1)), dtype=np.uint64)
    assert cm.g1_compress(n) == bytes([0xB7]) + G1_GENERATOR_COMPRESSED[1:]


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


@pytest.mark.skipif(os.environ.get("CURDLE_EXPECT_GPU") == "1", reason="GPU box")
def test_protocol_layer_has_no_cpu_msm_either(cm):
    """The protocol restatement funnels every MultiExp into the GPU entry points: without
    a device ShufflePermuteCommit (common/util.go:75) fails with ENODEV, it does not fall
    back to a host MSM."""
    if cm.device_available():
        pytest.skip("a device is visible")
    ell = 4
    rand = cm.Rand(0)
    crs = cm.CRS(ell, rand)                      # host only: hashing + scalar multiplications
    Rs, Ss = rand.get_g1_affines(ell), rand.get_g1_affines(ell)
    with pytest.raises(cm.CurdleError) as e:
        cm.shuffle_permute_commit(crs, Rs, Ss, rand.generate_permutation(ell), rand.get_fr(), rand)
    assert e.value.code == cm.ENODEV


def _build_c_example(tmp_path, name="msm_from_c"):
    import subprocess
    exe = str(tmp_path / name)
    pkg = os.path.join(ROOT, "go-curdleproofs_amd")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", name + ".c"), "-L" + pkg, "-lcurdlemsm",
                           "-Wl,-rpath," + pkg, "-o", exe])
    return exe


def test_header_is_plain_c_and_a_c_caller_links(cm, tmp_path):
    """cgo parses include/curdle_msm.h as C: it must compile as strict C99 and a C program must
    link against the library; without a device that program reports ENODEV (exit code 2)."""
    import subprocess
    exe = _build_c_example(tmp_path)
    if not cm.device_available():
        r = subprocess.run([exe], capture_output=True, text=True)
        assert r.returncode == 2 and "no HIP device" in r.stderr


def test_multi_device_c_caller_links(cm, tmp_path):
    """The multi-device boundary as a C program sees it (what the cgo shim's InitDevices /
    OnDevice bind): strict C99, links, and without a device fails loudly with ENODEV."""
    import subprocess
    exe = _build_c_example(tmp_path, "multi_device_from_c")
    if not cm.device_available():
        r = subprocess.run([exe], capture_output=True, text=True)
        assert r.returncode == 2 and "no HIP device" in r.stderr


@pytest.mark.gpu
def test_multi_device_c_caller_agrees_across_contexts(gpu, tmp_path):
    """One C process, two contexts on the one GPU: curdle_msm_g1 on one context, split over both
    by point ranges, and a batch on the second context selected with curdle_set_device -- the
    three results are the same bytes (the program compares them and exits non-zero otherwise)."""
    import subprocess
    exe = _build_c_example(tmp_path, "multi_device_from_c")
    r = subprocess.run([exe, "0", "0"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = r.stdout.strip().splitlines()
    assert len(lines) == 3 and lines[0][5:] == lines[1][5:] == lines[2][5:]


@pytest.mark.gpu
def test_c_caller_computes_three_g(gpu, oracle, tmp_path):
    import subprocess
    exe = _build_c_example(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True, check=True)
    want = oracle.jac_to_mont_limbs(oracle.scalar_mul(3, oracle.G1))
    assert r.stdout.strip() == "3G x0=%016x y0=%016x z0=%016x" % (want[0], want[6], want[12])


def _zero_wire_proof(n):
    """A syntactically complete curdleproof for n = ell + 4 (all-zero point and scalar
    records, correct uint32 slice prefixes): it parses, so decoding
    reaches the batched point decoder."""
    m = n.bit_length() - 1
    pt, fr = bytes(48), bytes(32)
    vec = m.to_bytes(4, "big") + pt * m
    ipa = pt * 2 + vec * 4 + fr * 2
    sameperm = pt + (pt + fr + ipa)
    samescalar = pt * 4 + fr * 3
    samemsm = pt * 3 + vec * 6 + fr
    return pt + pt * 2 + pt * 2 + pt + pt + sameperm + samescalar + samemsm


def test_device_failure_in_point_decoding_is_not_reported_as_reject(cm):
    """a device failure inside the batched point decoder ("decoding points: ...")
    must surface with the failing entry point's code (ENODEV here), never as EINVAL -- the code
    of a malformed proof -- so a caller mapping EINVAL to "reject" cannot drop a valid proof
    because the GPU was unavailable."""
    if cm.device_available():
        pytest.skip("a device is visible")
    ell = 60
    rand = cm.Rand(3)
    crs = cm.CRS(ell, rand)
    pts = np.tile(rand.get_g1_affines(1), (ell, 1))
    M = np.zeros(18, dtype=np.uint64)
    with pytest.raises(cm.CurdleError) as e:
        cm.verify(crs, _zero_wire_proof(ell + 4), pts, pts, pts, pts, M, cm.Rand(4))
    assert e.value.code == cm.ENODEV, (e.value.code, e.value.msg)
    assert "decoding points" in e.value.msg
    # a proof that does not parse at all IS a malformed input
    with pytest.raises(cm.CurdleError) as e:
        cm.verify(crs, b"\x00" * 100, pts, pts, pts, pts, M, cm.Rand(4))
    assert e.value.code == cm.EINVAL


def test_stale_and_repeated_tickets_are_refused(cm):
    """tickets carry the slot's generation; a made-up, stale or repeated ticket is
    refused instead of releasing a slot owned by another caller.  (Needs no device: the checks
    come before any HIP call.)"""
    out = np.zeros(18, dtype=np.uint64)
    for t in (0, 3, 7, (5 << 8) | 1):
        with pytest.raises(cm.CurdleError) as e:
            cm.msm_wait(t)
        assert e.value.code == cm.EINVAL
    with pytest.raises(cm.CurdleError):
        cm.g1_decompress_finish(0, 1)
