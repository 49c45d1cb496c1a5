"""Batch verification only (curdle_verify_batch), for tuning the chunked decode-ahead:
    python tools/bench_batch.py [ell] [k] [reps] [threads]
Environment: CURDLE_BATCH_CHUNK, CURDLE_BATCH_PRODUCERS, CURDLE_BATCH_GROUP, GPU_MAX_HW_QUEUES."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "go-curdleproofs_amd"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import curdlemsm as cm

cm.init(0)
ell = int(sys.argv[1]) if len(sys.argv) > 1 else 252
kb = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
nt = int(sys.argv[4]) if len(sys.argv) > 4 else 16
rand = cm.Rand(0)
crs = cm.CRS(ell, rand)
insts = []
for j in range(4):
    r2 = cm.Rand(77 + j)
    p2 = r2.generate_permutation(ell)
    k2 = r2.get_fr()
    R2, S2 = r2.get_g1_affines(ell), r2.get_g1_affines(ell)
    T2, U2, M2, rsm2 = cm.shuffle_permute_commit(crs, R2, S2, p2, k2, r2)
    insts.append((cm.prove(crs, R2, S2, T2, U2, M2, p2, k2, rsm2, cm.Rand(5 + j)), R2, S2, T2, U2, M2))
args = [list(c) for c in zip(*[insts[i % 4] for i in range(kb)])]
batch = cm.PreparedVerifyBatch(*args)  # marshalled once: the timed region is the C call
assert all(batch.run(crs, cm.Rand(5), nthreads=nt))
ts = []
for r in range(reps):
    t0 = time.perf_counter()
    assert all(batch.run(crs, cm.Rand(6 + r), nthreads=nt))
    ts.append(time.perf_counter() - t0)
print(f"ell={ell} k={kb} threads={nt} chunk={os.environ.get('CURDLE_BATCH_CHUNK','auto')} producers={os.environ.get('CURDLE_BATCH_PRODUCERS','2')} "
      f"queues={os.environ['GPU_MAX_HW_QUEUES']}: " + ", ".join(f"{t*1e3:.1f} ms" for t in ts) + f" -> best {kb/min(ts):.0f}/s", flush=True)
