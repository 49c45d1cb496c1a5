"""Randomised differential test of the GPU MSM against the C oracle: random sizes, repeated /
negated / infinity points (doublings and cancellations inside buckets), scalars that are 0, 1,
r-1, short, or share windows; single, batch and window-partial entry points.
    python tools/fuzz_msm.py [seconds] [seed]          (FUZZ_SIZES=65536,262144 for large inputs)
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "go-curdleproofs_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle", "py"))
import numpy as np
import curdlemsm as cm
import bls12381_ref as o
import coracle as co

SIZES = [int(x) for x in os.environ.get("FUZZ_SIZES", "1,2,3,7,33,64,257,1000,1268,4097,20000").split(",")]
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
cm.init(0)
k0, q0 = o.Rand(7).get_frs(2)
pool = co.points_walk(k0, q0, 4096)                      # distinct points with known structure
R = o.R
P_MOD = o.P


def neg_points(pts):
    """-(x, y) = (x, p - y) in Montgomery limbs."""
    out = pts.copy()
    for i in range(len(pts)):
        y = sum(int(v) << (64 * j) for j, v in enumerate(pts[i, 6:]))
        if y:
            y = P_MOD - y          # Montgomery form is linear: -(yR) = (p - y)R
            out[i, 6:] = [(y >> (64 * j)) & 0xFFFFFFFFFFFFFFFF for j in range(6)]
    return out


def scalars(n):
    kind = rng.integers(0, 6)
    if n > 50000:                                                                 # vectorised for large inputs
        v = rng.integers(0, 1 << 64, size=(n, 4), dtype=np.uint64)
        v[:, 3] &= np.uint64((1 << 61) - 1)
        if kind in (1, 3):
            v[:, 1:] = 0
            v[:, 0] &= np.uint64(511)
        elif kind == 2:
            v[:] = v[0]
        elif kind == 4:                                                           # K distinct values: buckets around the merge limit (round 6)
            kk = int(rng.integers(16, 5000))
            v = v[:kk][rng.integers(0, kk, size=n)]
        return v                                                                  # any 4 limbs < r are a valid Montgomery element
    if kind == 0:
        v = [int.from_bytes(rng.bytes(32), "little") % R for _ in range(n)]
    elif kind == 1:
        v = [int(rng.integers(0, 1 << 9)) for _ in range(n)]                      # short (util.go:75)
    elif kind == 2:
        c = int.from_bytes(rng.bytes(32), "little") % R
        v = [c] * n                                                               # all equal
    elif kind == 3:
        v = [[0, 1, R - 1, 2, R - 2][int(rng.integers(0, 5))] for _ in range(n)]
    elif kind == 4:
        w = int(rng.integers(0, 16))
        v = [(int(rng.integers(1, 1 << 16)) << (16 * w)) % R for _ in range(n)]   # one hot window
    else:
        v = [(int.from_bytes(rng.bytes(32), "little") % R) & ~((1 << int(rng.integers(0, 200))) - 1) for _ in range(n)]
    return np.array([o.fr_to_mont_limbs(x) for x in v], dtype=np.uint64)


t_end = time.time() + budget
cases = 0
while time.time() < t_end:
    n = int(rng.choice(SIZES))
    idx = rng.integers(0, 4096 if rng.integers(0, 2) else 8, size=n)          # many repeats half of the time
    pts = pool[idx].copy()
    flip = rng.integers(0, 4, size=n) == 0
    if flip.any():
        pts[flip] = neg_points(pts[flip])
    inf = rng.integers(0, 50, size=n) == 0
    pts[inf] = 0
    sc = scalars(n)
    want = co.msm_pippenger(pts, sc, threads=16)
    mode = int(rng.integers(0, 5))
    if mode == 0:
        got = cm.msm_g1(pts, sc, flags=cm.MSM_ANY_CURVE_POINT if rng.integers(0, 3) == 0 else 0)   # round 5: the opt-out from the endomorphism too
    elif mode == 3:                                                             # round 5: flagged device calls (kept converted bases, no endomorphism)
        import torch
        d_p = torch.from_numpy(pts.view(np.int64)).to("cuda:0")
        d_s = torch.from_numpy(sc.view(np.int64)).to("cuda:0")
        flags = int(rng.integers(1, 4))
        got = cm.msm_g1_device(d_p.data_ptr(), d_s.data_ptr(), n, flags=flags)
        again = cm.msm_g1_device(d_p.data_ptr(), d_s.data_ptr(), n, flags=flags)   # the second call reads the kept copy
        if flags & cm.MSM_BASES_UNCHANGED:
            cm.msm_forget_bases(d_p.data_ptr())
        if not (again == got).all():
            got = again
    elif mode == 4:                                                             # round 5: the pipelined form, three in flight, flags mixed
        import torch
        d_p = torch.from_numpy(pts.view(np.int64)).to("cuda:0")
        d_s = torch.from_numpy(sc.view(np.int64)).to("cuda:0")
        tk = [cm.msm_g1_device_submit(d_p.data_ptr(), d_s.data_ptr(), n, flags=int(f)) for f in rng.integers(0, 4, size=3)]
        res = [cm.msm_wait(t) for t in tk]
        cm.msm_forget_bases(d_p.data_ptr())
        got = res[0] if all((r == res[0]).all() for r in res) else res[int(np.argmax([not (r == want).all() for r in res]))]
    elif mode == 1:                                                             # split into a batch of pieces
        cuts = sorted(set([0, n] + [int(c) for c in rng.integers(0, n + 1, size=3)]))
        parts = cm.msm_g1_batch(pts, sc, np.array(cuts, dtype=np.uint64))
        got = cm.g1_sum(parts)
    else:                                                                       # window partials
        c = cm.window_bits(n)
        W = cm.num_windows(n, c)
        cut = int(rng.integers(0, W + 1))
        import torch
        d_p = torch.from_numpy(pts.view(np.int64)).to("cuda:0")
        d_s = torch.from_numpy(sc.view(np.int64)).to("cuda:0")
        a = cm.msm_g1_device(d_p.data_ptr(), d_s.data_ptr(), n, window_bits=c, win_begin=0, win_end=cut)
        b = cm.msm_g1_device(d_p.data_ptr(), d_s.data_ptr(), n, window_bits=c, win_begin=cut, win_end=W)
        got = cm.g1_sum(np.stack([a, b]))
    if not (got == want).all():
        np.savez("gpurun_out/fuzz_failure.npz", pts=pts, sc=sc, got=got, want=want, mode=mode)
        print(f"MISMATCH after {cases} cases: n={n} mode={mode} seed={seed}; inputs saved to gpurun_out/fuzz_failure.npz", flush=True)
        sys.exit(1)
    cases += 1
    if cases % 50 == 0:
        print(f"{cases} cases ok", flush=True)
print(f"fuzz: {cases} cases ok in {budget:.0f} s (seed {seed})")
