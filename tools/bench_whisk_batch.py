"""IsValidWhiskShuffleProofBatch only (BASELINE config 5 end to end from bytes), for tuning:
    python tools/bench_whisk_batch.py [k] [reps] [threads]
Environment: CURDLE_BATCH_CHUNK, CURDLE_BATCH_PRODUCERS, CURDLE_BATCH_GROUP, GPU_MAX_HW_QUEUES."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "go-curdleproofs_amd"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np
import curdlemsm as cm

cm.init(0)
ONE = np.array([0x760900000002fffd, 0xebf4000bc40c0002, 0x5f48985753c758ba, 0x77ce585370525745, 0x5c071a97a256ec6d,
                0x15f65ec3fa80e493], dtype=np.uint64)
compress = lambda aff: cm.g1_compress(np.concatenate([aff, ONE]))
kb = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
nt = int(sys.argv[3]) if len(sys.argv) > 3 else 16
rand = cm.Rand(0)
crs = cm.CRS(cm.WHISK_ELL, rand)
sets = []
for j in range(4):
    r = cm.Rand(10 + j)
    pts = r.get_g1_affines(2 * cm.WHISK_ELL)
    pre = [compress(pts[2 * i]) + compress(pts[2 * i + 1]) for i in range(cm.WHISK_ELL)]
    post, proof = cm.whisk_generate_shuffle_proof(crs, pre, r)
    sets.append((pre, post, proof))
args = tuple([sets[i % 4][c] for i in range(kb)] for c in range(3))
batch = cm.PreparedWhiskBatch(*args)  # marshalled once: the timed region is the C call
assert all(batch.run(crs, cm.Rand(3), nthreads=nt))
ts = []
for r in range(reps):
    t0 = time.perf_counter()
    assert all(batch.run(crs, cm.Rand(4 + r), nthreads=nt))
    ts.append(time.perf_counter() - t0)
print(f"whisk k={kb} threads={nt} chunk={os.environ.get('CURDLE_BATCH_CHUNK','auto')} producers={os.environ.get('CURDLE_BATCH_PRODUCERS','2')} "
      f"queues={os.environ['GPU_MAX_HW_QUEUES']}: " + ", ".join(f"{t*1e3:.1f} ms" for t in ts) + f" -> best {kb/min(ts):.0f}/s", flush=True)
