"""One process driving several GPUs through the C ABI (curdle_init_devices; SURVEY.md sections
8b / 8e).  A one-GPU box cannot hold two devices, so the multi-device code is
run with devices = {0, 0}: two contexts -- each with its own streams, workspace slots, decode
contexts, resident copies of the CRS and host thread -- on the one GPU.  Everything below goes
through the same entry points a Go host would bind (go/curdlemsm/curdlemsm.go)."""
import os
import subprocess
import sys
import threading

import numpy as np
import pytest

from conftest import ROOT
from test_msm_gpu import _walk_expected, rand_scalars

pytestmark = pytest.mark.gpu

# The HIP devices of the two contexts.  {0, 0} on a one-GPU box; "0,1" under the logical-device shim
# (tools/logical_devices_shim.cpp: one GPU posing as two, every cross-device use of a stream, event
# or allocation an error) and on a box that really has two.
DEVS = [int(x) for x in os.environ.get("CURDLE_TEST_DEVICES", "0,0").split(",")]


@pytest.fixture(scope="module")
def two(gpu):
    gpu.init_devices(DEVS)
    assert gpu.device_count() == 2
    yield gpu
    gpu.set_device(-1)
    gpu.shutdown()
    gpu.init(0)
    assert gpu.device_count() == 1


def test_init_devices_is_all_or_nothing(gpu, oracle, coracle):
    """A list with a device that cannot be brought up leaves NOTHING behind: the contexts the call
    had created are closed again, the configuration is what it was, a retry starts clean (review of
    round 3: contexts [0, i) stayed initialised, unreachable for curdle_shutdown, and every later
    list failed with "already initialised on device X")."""
    cm = gpu
    assert cm.device_count() == 1
    for bad in ([DEVS[0], 99], [DEVS[0], DEVS[1], 99], [99]):
        with pytest.raises(cm.CurdleError):
            cm.init_devices(bad)
        assert cm.device_count() == 1
    k, q = oracle.Rand(5).get_frs(2)
    pts = coracle.points_walk(k, q, 300)
    sc = rand_scalars(np.random.default_rng(5), 300, oracle)
    assert (cm.msm_g1(pts, sc) == coracle.msm_pippenger(pts, sc, threads=2)).all()   # context 0 is untouched
    cm.init_devices(DEVS)                                  # ... and the good list comes up
    assert cm.device_count() == 2
    cm.shutdown()
    cm.init(0)
    assert cm.device_count() == 1


def on_device(cm, ordinal, fn):
    """fn() on a thread whose current context is `ordinal`."""
    box = {}

    def run():
        try:
            cm.set_device(ordinal)
            box["v"] = fn()
        except BaseException as e:          # noqa: BLE001 -- handed to the caller's thread
            box["e"] = e

    t = threading.Thread(target=run)
    t.start()
    t.join()
    if "e" in box:
        raise box["e"]
    return box["v"]


def test_contexts_and_tickets(two, oracle, coracle):
    import torch
    cm = two
    with pytest.raises(cm.CurdleError):
        cm.set_device(2)
    with pytest.raises(cm.CurdleError):
        cm.init_devices([0])                                   # a standing configuration is not silently replaced
    cm.init_devices(DEVS)                                      # the same list again is fine
    assert cm.get_device() == 0
    assert on_device(cm, 1, cm.get_device) == 1 and cm.get_device() == 0     # the selection is per thread
    k, q = oracle.Rand(3).get_frs(2)
    n = 3000
    pts = coracle.points_walk(k, q, n)
    sc = rand_scalars(np.random.default_rng(3), n, oracle)
    exp = coracle.msm_pippenger(pts, sc, threads=4)
    assert (on_device(cm, 1, lambda: cm.msm_g1(pts, sc)) == exp).all()
    # a thread that SELECTED a device keeps a large host-buffer MSM there (the calling thread of this
    # test never selected one: its large calls spread over both, test_one_msm_over_both_contexts)
    big_n = 1 << 16
    big_pts = coracle.points_walk(k, q, big_n)
    big_sc = rand_scalars(np.random.default_rng(33), big_n, oracle)
    big_exp = coracle.msm_pippenger(big_pts, big_sc, threads=8)
    assert (on_device(cm, 1, lambda: cm.msm_g1(big_pts, big_sc)) == big_exp).all()
    assert (cm.msm_g1(big_pts, big_sc) == big_exp).all()
    # a ticket names its context: submitted on context 1, waited for from a thread on context 0
    # every context reads inputs resident on ITS device
    d_p = [torch.from_numpy(pts.view(np.int64)).to(f"cuda:{DEVS[d]}") for d in (0, 1)]
    d_s = [torch.from_numpy(sc.view(np.int64)).to(f"cuda:{DEVS[d]}") for d in (0, 1)]
    t1 = on_device(cm, 1, lambda: cm.msm_g1_device_submit(d_p[1].data_ptr(), d_s[1].data_ptr(), n))
    t0 = cm.msm_g1_device_submit(d_p[0].data_ptr(), d_s[0].data_ptr(), n)
    assert (t1 >> 3) & 0x1F == 1 and (t0 >> 3) & 0x1F == 0
    assert (cm.msm_wait(t1) == exp).all() and (cm.msm_wait(t0) == exp).all()
    with pytest.raises(cm.CurdleError):
        cm.msm_wait(t1)                                        # already waited for
    # every slot of BOTH contexts can be in flight at once
    tickets = [on_device(cm, d, lambda d=d: [cm.msm_g1_device_submit(d_p[d].data_ptr(), d_s[d].data_ptr(), n) for _ in range(cm.MSM_SLOTS)])
               for d in (0, 1)]
    for ts in tickets:
        for t in ts:
            assert (cm.msm_wait(t) == exp).all()


def test_a_thread_that_clears_its_selection_spreads_again(two, oracle, coracle):
    """curdle_set_device(ordinal >= 0) pins a thread's host-buffer MSMs to that device; -1 clears the
    selection and large calls spread over all devices again (review of round 4: tools/bench_multi_device.py
    measured "all devices" from a pinned thread, and curdlemsm.OnDevice restored 0 instead of "none", which
    left a recycled OS thread pinned for good).  curdle_get_device_selection is what OnDevice restores."""
    cm = two
    k, q = oracle.Rand(3).get_frs(2)
    n = 1 << 16
    pts = coracle.points_walk(k, q, n)
    sc = rand_scalars(np.random.default_rng(34), n, oracle)
    exp = coracle.msm_pippenger(pts, sc, threads=8)

    def script():
        seen = []
        assert cm.get_device_selection() == -1                      # a fresh thread has none
        c0 = cm.stat_spread_calls()
        assert (cm.msm_g1(pts, sc) == exp).all()
        seen.append(cm.stat_spread_calls() - c0)                    # spread
        cm.set_device(1)
        assert cm.get_device_selection() == 1 and cm.get_device() == 1
        c0 = cm.stat_spread_calls()
        assert (cm.msm_g1(pts, sc) == exp).all()
        seen.append(cm.stat_spread_calls() - c0)                    # pinned: not spread
        prev = -1                                                   # what OnDevice saved before it selected
        cm.set_device(prev)
        assert cm.get_device_selection() == -1 and cm.get_device() == 0
        c0 = cm.stat_spread_calls()
        assert (cm.msm_g1(pts, sc) == exp).all()
        seen.append(cm.stat_spread_calls() - c0)                    # spread again
        return seen

    box = {}
    t = threading.Thread(target=lambda: box.update(v=script()))
    t.start()
    t.join()
    assert box.get("v") == [1, 0, 1], box


def test_flagged_calls_keep_one_converted_copy_per_context(two, oracle, coracle):
    """CURDLE_MSM_BASES_UNCHANGED and CURDLE_MSM_ANY_CURVE_POINT on a context other than 0: the converted copy of
    a base array belongs to the context the call runs on (its own cache, its own device's memory)."""
    import torch
    cm = two
    k, q = oracle.Rand(3).get_frs(2)
    n = 5000
    pts = coracle.points_walk(k, q, n)
    exp = []
    scs = [rand_scalars(np.random.default_rng(40 + j), n, oracle) for j in range(3)]
    for sc in scs:
        exp.append(coracle.msm_pippenger(pts, sc, threads=4))
    for d in (1, 0):
        d_p = torch.from_numpy(pts.view(np.int64)).to(f"cuda:{DEVS[d]}")
        for j, sc in enumerate(scs):
            d_s = torch.from_numpy(sc.view(np.int64)).to(f"cuda:{DEVS[d]}")
            flags = cm.MSM_BASES_UNCHANGED | (cm.MSM_ANY_CURVE_POINT if j == 2 else 0)
            got = on_device(cm, d, lambda: cm.msm_g1_device(d_p.data_ptr(), d_s.data_ptr(), n, flags=flags))
            assert (got == exp[j]).all(), (d, j)
        on_device(cm, d, lambda: cm.msm_forget_bases(d_p.data_ptr()))


def test_one_msm_over_both_contexts(two, oracle, coracle):
    """curdle_msm_g1 (host buffers: point ranges, one host thread per device) and
    curdle_msm_g1_replicated (resident inputs: Pippenger windows or point ranges) against the
    single-context result and the closed form, at a size below the split threshold, at 2^17
    and at the headline 2^20."""
    import torch
    cm = two
    k, q = oracle.Rand(1).get_frs(2)
    for n in (1000, (1 << 17) + 3, 1 << 20):
        d_pts = torch.empty((n, 12), dtype=torch.int64, device="cuda:0")
        cm.synth_points_walk_device(k, q, n, d_pts.data_ptr())
        sc = rand_scalars(np.random.default_rng(n), n, oracle)
        d_sc = torch.from_numpy(sc.view(np.int64)).to("cuda:0")
        exp = _walk_expected(oracle, coracle, k, q, sc)
        assert (cm.msm_g1_device(d_pts.data_ptr(), d_sc.data_ptr(), n) == exp).all()
        assert (cm.msm_g1(d_pts.cpu().numpy().view(np.uint64), sc) == exp).all(), n
        # the replicated inputs: context 1's copy lives on its own device
        # (through the host: a peer copy queues device 1's memory on a stream of device 0, which the
        # logical-device shim counts as a mix-up -- the library itself never does that)
        d_pts1 = d_pts.cpu().to(f"cuda:{DEVS[1]}") if DEVS[1] != DEVS[0] else d_pts
        d_sc1 = torch.from_numpy(sc.view(np.int64)).to(f"cuda:{DEVS[1]}") if DEVS[1] != DEVS[0] else d_sc
        ptrs, sptrs = [d_pts.data_ptr(), d_pts1.data_ptr()], [d_sc.data_ptr(), d_sc1.data_ptr()]
        for split in (cm.SPLIT_AUTO, cm.SPLIT_WINDOWS, cm.SPLIT_POINTS):
            assert (cm.msm_g1_replicated(ptrs, sptrs, n, split) == exp).all(), (n, split)
    with pytest.raises(cm.CurdleError):
        cm.msm_g1_replicated([d_pts.data_ptr(), 0], sptrs, n)


def test_verification_on_the_second_context(two, oracle):
    """A CRS made resident by a thread on context 0 is used by verifications on context 1 (its
    copy there is made on first use), and a batch is sharded over both contexts."""
    cm = two
    ell = 60
    rand = cm.Rand(0)
    crs = cm.CRS(ell, rand)
    base = []
    for j in range(3):
        perm = cm.Rand(100 + j).generate_permutation(ell)
        kk = rand.get_fr()
        Rs, Ss = rand.get_g1_affines(ell), rand.get_g1_affines(ell)
        Ts, Us, M, rs_m = cm.shuffle_permute_commit(crs, Rs, Ss, perm, kk, rand)
        base.append([cm.prove(crs, Rs, Ss, Ts, Us, M, perm, kk, rs_m, cm.Rand(42 + j)), Rs, Ss, Ts, Us, M])
    p0 = base[0]
    assert cm.verify(crs, *p0, cm.Rand(7)) is True
    assert on_device(cm, 1, lambda: cm.verify(crs, *p0, cm.Rand(8))) is True
    assert on_device(cm, 1, lambda: cm.verify(crs, p0[0], p0[2], p0[1], p0[3], p0[4], p0[5], cm.Rand(9))) is False
    items = [list(base[i % 3]) for i in range(96)]
    expect = [True] * 96
    items[5][1], items[5][2] = base[1][1], base[1][2]            # instance of another proof, first shard
    expect[5] = False
    items[70][0] = items[70][0][:-9]                             # truncated, second shard
    expect[70] = False
    items[95][3] = base[(95 + 1) % 3][3]
    expect[95] = False
    cols = [list(c) for c in zip(*items)]
    assert cm.verify_batch(crs, *cols, cm.Rand(9), nthreads=8) == expect


def test_whisk_batch_over_both_contexts(two, oracle):
    from test_whisk import shuffle_trackers
    cm = two
    crs = cm.CRS(cm.WHISK_ELL, cm.Rand(4))
    sets = []
    for j in range(2):
        pre = shuffle_trackers(cm, oracle, cm.Rand(30 + j), cm.WHISK_ELL)
        post, proof = cm.whisk_generate_shuffle_proof(crs, pre, cm.Rand(60 + j))
        sets.append((pre, post, proof))
    k = 64
    pres = [sets[i % 2][0] for i in range(k)]
    posts = [sets[i % 2][1] for i in range(k)]
    proofs = [sets[i % 2][2] for i in range(k)]
    expect = [True] * k
    posts[3] = sets[0][1]                                        # proof 3 is shuffle 1's: wrong post trackers
    expect[3] = False
    proofs[40] = b"\x00" * 4576
    expect[40] = False
    assert cm.whisk_is_valid_shuffle_proof_batch(crs, pres, posts, proofs, cm.Rand(1), nthreads=8) == expect
    assert on_device(cm, 1, lambda: cm.whisk_is_valid_shuffle_proof(crs, sets[0][0], sets[0][1], sets[0][2], cm.Rand(2))) is True


def test_the_suite_above_with_two_logical_devices(gpu, tmp_path):
    """devices = {0, 0} cannot see a missing hipSetDevice: both contexts' streams, events and
    allocations live on the one device whatever thread made them.  So the tests above run once
    more, in a child process, under tools/logical_devices_shim.cpp: the one GPU reports TWO devices,
    every stream / event / allocation is tagged with the logical device current at its creation,
    and a launch, event record or copy that mixes logical devices -- or a big-LDS launch whose
    opt-in was made on the other device -- fails the call.  The child must pass with no violation
    and with launches checked on BOTH logical devices ("what is missing" 5)."""
    if os.environ.get("CURDLE_TEST_DEVICES"):
        pytest.skip("already the child run")
    shim = str(tmp_path / "logical_devices_shim.so")
    subprocess.check_call(["g++", "-O2", "-shared", "-fPIC", "-Wno-deprecated-declarations", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                           os.path.join(ROOT, "tools", "logical_devices_shim.cpp"), "-o", shim, "-ldl"])
    env = dict(os.environ, LD_PRELOAD=shim, CURDLE_TEST_DEVICES="0,1", CURDLE_LOGICAL_DEVICES="2")
    p = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_multi_device.py"), "-x", "-q",
                        "-p", "no:cacheprovider"], env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    tail = p.stdout[-2500:] + p.stderr[-2500:]
    assert p.returncode == 0, tail
    assert "VIOLATION" not in p.stderr, tail
    import re
    m = re.search(r"launches checked per device: (\d+) (\d+); .*violations: (\d+)", p.stderr)
    assert m, tail
    assert int(m.group(1)) > 100 and int(m.group(2)) > 100 and int(m.group(3)) == 0, m.group(0)
