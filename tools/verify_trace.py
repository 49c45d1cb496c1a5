"""Where the time of one curdleproof.Verify (ell = 252) goes: run under the LD_PRELOAD
shim tools/trace_shim.cpp, which counts and times the MSM entry points and the serial
host group operations the protocol layer calls.
    g++ -O2 -shared -fPIC tools/trace_shim.cpp -o gpurun_out/trace_shim.so -ldl
    LD_PRELOAD=gpurun_out/trace_shim.so python tools/verify_trace.py
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "go-curdleproofs_amd"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import ctypes
import curdlemsm as cm


def mark(what):
    """Print + zero the shim's counters (no-op when the shim is not preloaded)."""
    try:
        f = ctypes.CDLL(None).curdle_trace_mark
    except AttributeError:
        return
    f(what.encode())


cm.init(0)
ell = int(sys.argv[1]) if len(sys.argv) > 1 else 252
reps = 20
rand = cm.Rand(0)
crs = cm.CRS(ell, rand)
perm = cm.Rand(42).generate_permutation(ell)
k = rand.get_fr()
Rs, Ss = rand.get_g1_affines(ell), rand.get_g1_affines(ell)
Ts, Us, M, rs_m = cm.shuffle_permute_commit(crs, Rs, Ss, perm, k, rand)
proof = cm.prove(crs, Rs, Ss, Ts, Us, M, perm, k, rs_m, cm.Rand(42))
assert cm.verify(crs, proof, Rs, Ss, Ts, Us, M, cm.Rand(43))
mark("setup + prove + 1 verify")
rands = [cm.Rand(100 + i) for i in range(reps)]
t0 = time.perf_counter()
for i in range(reps):
    assert cm.verify(crs, proof, Rs, Ss, Ts, Us, M, rands[i])
dt = (time.perf_counter() - t0) / reps
print(f"ell={ell}: verify {dt*1e3:.2f} ms over {reps} reps", flush=True)
mark(f"{reps} verifies")
