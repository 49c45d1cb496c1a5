"""How independent small MSMs scale when issued from T host threads at once (each call is
synchronous on its own workspace slot / stream):  python tools/bench_concurrency.py [n]"""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "go-curdleproofs_amd"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np
import curdlemsm as cm

cm.init(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1268
rand = cm.Rand(5)
pts = rand.get_g1_affines(n)
sc = np.stack([rand.get_fr() for _ in range(n)])
cm.msm_g1(pts, sc)
per = 40
for T in (1, 2, 4, 8, 12):
    def worker():
        for _ in range(per):
            cm.msm_g1(pts, sc)
    th = [threading.Thread(target=worker) for _ in range(T)]
    t0 = time.perf_counter()
    [t.start() for t in th]
    [t.join() for t in th]
    dt = time.perf_counter() - t0
    print(f"n={n} threads={T}: {T*per/dt:.0f} calls/s, {dt/per*1e3:.3f} ms per call per thread (queues={os.environ['GPU_MAX_HW_QUEUES']})", flush=True)
