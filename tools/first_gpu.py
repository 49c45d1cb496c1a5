"""First GPU bring-up: device primitives vs host, small MSM parity vs the Python
oracle, large-N consistency and per-kernel timing.  Run on the GPU box:
    python tools/first_gpu.py > gpurun_out/first_gpu.log
"""
import os, sys, time, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle", "py"))
sys.path.insert(0, os.path.join(ROOT, "go-curdleproofs_amd"))
import numpy as np
import bls12381_ref as o
import curdlemsm as cm

def log(*a):
    print(*a, flush=True)

log("device available:", cm.device_available())
cm.init(0)
rnd = random.Random(7)

# 1. primitives: device vs host (host was checked against the oracle on CPU)
def fp32(v): return [(v >> (32 * i)) & 0xffffffff for i in range(12)]
N = 4096
A = [rnd.randrange(o.P) for _ in range(N)]; B = [rnd.randrange(o.P) for _ in range(N)]
A[0] = 0; B[1] = 0; A[2] = o.P - 1; B[2] = o.P - 1; A[3] = 1; B[3] = o.P - 1
inp = np.array([fp32(a) + fp32(b) for a, b in zip(A, B)], dtype=np.uint32)
for op in (0, 1, 2, 3):
    h = cm.selftest_op(op, inp, False); d = cm.selftest_op(op, inp, True)
    log("selftest op", op, "device==host:", bool((h == d).all()))
S = [rnd.randrange(o.R) for _ in range(N)]
inp = np.array([[(s >> (32 * i)) & 0xffffffff for i in range(8)] + [0] * 8 for s in S], dtype=np.uint32)
log("selftest op 4 device==host:", bool((cm.selftest_op(4, inp, False) == cm.selftest_op(4, inp, True)).all()))

def xyzz_limbs(pt, z):
    if pt is None: return fp32(o.R_FP) + fp32(o.R_FP) + [0] * 24
    zz = z * z % o.P; zzz = zz * z % o.P
    return fp32(pt[0] * zz % o.P * o.R_FP % o.P) + fp32(pt[1] * zzz % o.P * o.R_FP % o.P) + fp32(zz * o.R_FP % o.P) + fp32(zzz * o.R_FP % o.P)
pts = [o.scalar_mul(rnd.randrange(1, o.R), o.G1) for _ in range(40)]
cases = [(pts[i], rnd.randrange(2, o.P), pts[i + 1], rnd.randrange(2, o.P)) for i in range(39)]
cases += [(pts[0], 5, pts[0], 7), (pts[0], 5, o.neg(pts[0]), 7), (None, 1, pts[1], 3), (pts[1], 3, None, 1), (None, 1, None, 1)]
for op in (5, 6, 7):
    rows = []
    for a, za, b, zb in cases:
        if op == 5:
            bl = (fp32(b[0] * o.R_FP % o.P) + fp32(b[1] * o.R_FP % o.P) + [0] * 24) if b else [0] * 48
        else:
            bl = xyzz_limbs(b, zb)
        rows.append(xyzz_limbs(a, za) + bl)
    arr = np.array(rows, dtype=np.uint32)
    h = cm.selftest_op(op, arr, False); d = cm.selftest_op(op, arr, True)
    log("selftest op", op, "device==host:", bool((h == d).all()))

# 2. small MSM parity vs the Python oracle
r = o.Rand(0)
base_pts = r.get_g1_affines(300)
base_sc = r.get_frs(300)
P_l = np.array([o.affine_to_mont_limbs(p) for p in base_pts], dtype=np.uint64)
S_l = np.array([o.fr_to_mont_limbs(s) for s in base_sc], dtype=np.uint64)
for n in (0, 1, 2, 3, 16, 64, 257, 300):
    got = o.jac_from_mont_limbs([int(x) for x in cm.msm_g1(P_l[:n], S_l[:n])])
    exp = o.msm(base_pts[:n], base_sc[:n])
    log(f"msm n={n}: parity", got == exp, cm.profile_last() if False else "")
# every window size
exp64 = o.msm(base_pts[:64], base_sc[:64])
for c in range(4, 17):
    cm.plan_override("WINDOW_BITS", c)
    got = o.jac_from_mont_limbs([int(x) for x in cm.msm_g1(P_l[:64], S_l[:64])])
    log(f"msm n=64 c={c}: parity", got == exp64)
cm.plan_override("WINDOW_BITS", None)
# edge: infinity base, zero scalar, scalar r-1, duplicates, all-equal scalars
pts_e = list(base_pts[:8]) + [None, base_pts[0], base_pts[0], o.neg(base_pts[1])]
sc_e = list(base_sc[:8]) + [12345, o.R - 1, 0, base_sc[1]]
got = o.jac_from_mont_limbs([int(x) for x in cm.msm_g1(np.array([o.affine_to_mont_limbs(p) for p in pts_e], dtype=np.uint64),
                                                       np.array([o.fr_to_mont_limbs(s) for s in sc_e], dtype=np.uint64))])
log("msm edge set: parity", got == o.msm(pts_e, sc_e))
beta = base_sc[5]
got = o.jac_from_mont_limbs([int(x) for x in cm.msm_g1(P_l[:60], np.array([o.fr_to_mont_limbs(beta)] * 60, dtype=np.uint64))])
log("msm all-equal scalars: parity", got == o.msm(base_pts[:60], [beta] * 60))

# 3. large N: tile 256 oracle points; expected = MSM_256 of per-residue scalar sums
cm.profile_enable(True)
T = 256
rng = np.random.default_rng(3)
for lg in (10, 12, 14, 16, 18, 20):
    n = 1 << lg
    sc = [int.from_bytes(rng.bytes(32), "big") % o.R for _ in range(n)]
    sums = [0] * T
    for i, s in enumerate(sc): sums[i % T] = (sums[i % T] + s) % o.R
    pts_arr = np.tile(P_l[:T], (n // T, 1))
    sc_arr = np.array([o.fr_to_mont_limbs(s) for s in sc], dtype=np.uint64)
    t0 = time.time(); out = cm.msm_g1(pts_arr, sc_arr); t1 = time.time()
    out2 = cm.msm_g1(pts_arr, sc_arr); t2 = time.time()
    got = o.jac_from_mont_limbs([int(x) for x in out])
    exp = o.msm(base_pts[:T], sums)
    prof = cm.profile_last()
    log(f"msm n=2^{lg}: parity {got == exp}, repeat-identical {bool((out == out2).all())}, wall(1st) {(t1-t0)*1e3:.1f} ms wall(2nd) {(t2-t1)*1e3:.1f} ms, c={prof['window_bits']} W={prof['num_windows']}")
    log("   kernels ms:", {k: round(v, 3) for k, v in prof["kernels"].items()})
log("done")
