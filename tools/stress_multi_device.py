#!/usr/bin/env python3
"""Concurrency stress of the multi-context library (curdle_init_devices with devices = {0, 0} on a
one-GPU box): several threads, each on a context of its own choosing, mix host-buffer MSMs
(split over both contexts from 2^16 pairs), device-resident MSMs, submit / wait pairs waited for
from another thread's context, verifications over one CRS (made resident per context on first
use) and small batches -- every result checked.
    python tools/stress_multi_device.py [seconds] [threads]"""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "go-curdleproofs_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle", "py"))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import curdlemsm as cm
import coracle as co
from bench import uniform_scalars

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
nthreads = int(sys.argv[2]) if len(sys.argv) > 2 else 6
cm.init_devices([0, 0])
sizes = [300, 5000, 1 << 16, (1 << 17) + 9]
pts = {n: co.points_walk(1234567 + n, 7654321, n) for n in sizes}
scs = {n: uniform_scalars(np.random.default_rng(n), n) for n in sizes}
exp = {n: co.msm_pippenger(pts[n], scs[n], threads=8) for n in sizes}
d_p = {n: torch.from_numpy(pts[n].view(np.int64)).to("cuda:0") for n in sizes}
d_s = {n: torch.from_numpy(scs[n].view(np.int64)).to("cuda:0") for n in sizes}
ell = 60
rand = cm.Rand(0)
crs = cm.CRS(ell, rand)
perm = cm.Rand(42).generate_permutation(ell)
kk = rand.get_fr()
Rs, Ss = rand.get_g1_affines(ell), rand.get_g1_affines(ell)
Ts, Us, M, rs_m = cm.shuffle_permute_commit(crs, Rs, Ss, perm, kk, rand)
proof = cm.prove(crs, Rs, Ss, Ts, Us, M, perm, kk, rs_m, cm.Rand(42))
stop = time.time() + budget
counts = [0] * nthreads
errors = []
tickets = []
tmu = threading.Lock()


def worker(t):
    rng = np.random.default_rng(100 + t)
    try:
        while time.time() < stop and not errors:
            cm.set_device(int(rng.integers(0, 2)))
            op = int(rng.integers(0, 8))
            n = sizes[int(rng.integers(0, len(sizes)))]
            if op == 0:
                assert (cm.msm_g1(pts[n], scs[n]) == exp[n]).all(), ("host", n)
            elif op == 1:
                assert (cm.msm_g1_device(d_p[n].data_ptr(), d_s[n].data_ptr(), n) == exp[n]).all(), ("device", n)
            elif op == 6:                                      # round 5: flagged calls from every thread (kept converted copies, no endomorphism)
                fl = int(rng.integers(1, 4))
                assert (cm.msm_g1_device(d_p[n].data_ptr(), d_s[n].data_ptr(), n, flags=fl) == exp[n]).all(), ("flags", n, fl)
                if rng.integers(0, 8) == 0:
                    cm.msm_forget_bases(d_p[n].data_ptr())     # while other threads may be reading the copy
            elif op == 7:
                assert (cm.msm_g1(pts[n], scs[n], flags=cm.MSM_ANY_CURVE_POINT) == exp[n]).all(), ("host any-curve", n)
            elif op == 2:
                # at most four tickets outstanding in all: a ticket holds a workspace slot until it is
                # waited for, and the synchronous calls of the other threads BLOCK for a slot -- sixteen
                # forgotten tickets and six blocked threads would be this script's deadlock, not the library's
                with tmu:
                    room = len(tickets) < 4
                if not room:
                    continue
                try:
                    tk = cm.msm_g1_device_submit(d_p[n].data_ptr(), d_s[n].data_ptr(), n)
                except cm.CurdleError:
                    continue                                   # every slot of this context in flight
                with tmu:
                    tickets.append((tk, n))
            elif op == 3:
                with tmu:
                    item = tickets.pop() if tickets else None
                if item:
                    assert (cm.msm_wait(item[0]) == exp[item[1]]).all(), ("wait", item[1])
            elif op == 4:
                assert cm.verify(crs, proof, Rs, Ss, Ts, Us, M, cm.Rand(int(rng.integers(1, 1 << 30)))) is True
                assert cm.verify(crs, proof, Ss, Rs, Ts, Us, M, cm.Rand(7)) is False
            else:
                k = 12
                got = cm.verify_batch(crs, [proof] * k, [Rs] * k, [Ss] * k, [Ts] * k, [Us] * k, [M] * k, cm.Rand(9), nthreads=4)
                assert got == [True] * k
            counts[t] += 1
    except BaseException as e:      # noqa: BLE001
        errors.append(repr(e))


th = [threading.Thread(target=worker, args=(t,)) for t in range(nthreads)]
[t.start() for t in th]
[t.join() for t in th]
for tk, n in tickets:                                        # whatever is still in flight
    assert (cm.msm_wait(tk) == exp[n]).all()
if errors:
    print("FAILED:", errors[:3])
    sys.exit(1)
cm.set_device(-1)                                           # no selection left on this thread
cm.shutdown()
print(f"stress_multi_device: {sum(counts)} operations from {nthreads} threads over 2 contexts in {budget:.0f} s, all results exact; per thread {counts}")
