"""Where the time of one curdleproof.Prove (ell = 252) goes, under tools/trace_shim.cpp.
    CURDLE_TRACE_LIB=$PWD/go-curdleproofs_amd/libcurdlemsm.so LD_PRELOAD=gpurun_out/trace_shim.so python tools/prove_trace.py
"""
import os, sys, time, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "go-curdleproofs_amd"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import curdlemsm as cm


def mark(what):
    try:
        f = ctypes.CDLL(None).curdle_trace_mark
    except AttributeError:
        return
    f(what.encode())


cm.init(0)
ell = int(sys.argv[1]) if len(sys.argv) > 1 else 252
rand = cm.Rand(0)
crs = cm.CRS(ell, rand)
perm = cm.Rand(42).generate_permutation(ell)
k = rand.get_fr()
Rs, Ss = rand.get_g1_affines(ell), rand.get_g1_affines(ell)
Ts, Us, M, rs_m = cm.shuffle_permute_commit(crs, Rs, Ss, perm, k, rand)
cm.prove(crs, Rs, Ss, Ts, Us, M, perm, k, rs_m, cm.Rand(42))
mark("setup + 1 prove")
reps = 5
t0 = time.perf_counter()
for i in range(reps):
    cm.prove(crs, Rs, Ss, Ts, Us, M, perm, k, rs_m, cm.Rand(50 + i))
dt = (time.perf_counter() - t0) / reps
print(f"ell={ell}: prove {dt*1e3:.1f} ms over {reps} reps", flush=True)
mark(f"{reps} proves")
