"""Two host threads decoding batches of compressed points at the same time (what the batch
verifier's two producers do):  python tools/bench_decode_concurrent.py [n] [threads]"""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "go-curdleproofs_amd"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np
import curdlemsm as cm

cm.init(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 37440
T = int(sys.argv[2]) if len(sys.argv) > 2 else 2
ONE = np.array([0x760900000002fffd, 0xebf4000bc40c0002, 0x5f48985753c758ba, 0x77ce585370525745, 0x5c071a97a256ec6d,
                0x15f65ec3fa80e493], dtype=np.uint64)
rand = cm.Rand(3)
pts = rand.get_g1_affines(256)
recs = b"".join(cm.g1_compress(np.concatenate([p, ONE])) for p in pts)
data = (recs * ((n + 255) // 256))[:48 * n]
cm.g1_decompress_batch(data)
for label, threads in (("alone", 1), ("together", T), ("alone again", 1)):
    times = [[] for _ in range(threads)]
    def worker(t):
        for _ in range(5):
            t0 = time.perf_counter()
            cm.g1_decompress_batch(data)
            times[t].append((time.perf_counter() - t0) * 1e3)
    th = [threading.Thread(target=worker, args=(t,)) for t in range(threads)]
    [t.start() for t in th]
    [t.join() for t in th]
    for t in range(threads):
        print(f"n={n} {label} thread {t}: " + " ".join(f"{x:.2f}" for x in times[t]) + " ms", flush=True)
