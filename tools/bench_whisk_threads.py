"""IsValidWhiskShuffleProof from T host threads at once (each call synchronous):
    python tools/bench_whisk_threads.py [calls per thread]"""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "go-curdleproofs_amd"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np
import curdlemsm as cm

cm.init(0)
per = int(sys.argv[1]) if len(sys.argv) > 1 else 40
ONE = np.array([0x760900000002fffd, 0xebf4000bc40c0002, 0x5f48985753c758ba, 0x77ce585370525745, 0x5c071a97a256ec6d,
                0x15f65ec3fa80e493], dtype=np.uint64)
compress = lambda aff: cm.g1_compress(np.concatenate([aff, ONE]))
rand = cm.Rand(0)
crs = cm.CRS(cm.WHISK_ELL, rand)
pts = rand.get_g1_affines(2 * cm.WHISK_ELL)
pre = [compress(pts[2 * i]) + compress(pts[2 * i + 1]) for i in range(cm.WHISK_ELL)]
post, proof = cm.whisk_generate_shuffle_proof(crs, pre, rand)
for _ in range(3):
    assert cm.whisk_is_valid_shuffle_proof(crs, pre, post, proof, cm.Rand(1))
out = []
for T in (1, 2, 3, 4, 8, 16):
    rands = [[cm.Rand(1000 + 50 * t + i) for i in range(per)] for t in range(T)]  # made outside the timed region
    def worker(tid):
        for i in range(per):
            assert cm.whisk_is_valid_shuffle_proof(crs, pre, post, proof, rands[tid][i])
    # a first round untimed: workspace slots and their buffers are made on first use, and T
    # threads reach slots no smaller thread count touched
    warm = [threading.Thread(target=lambda tid=t: [cm.whisk_is_valid_shuffle_proof(crs, pre, post, proof, cm.Rand(7 + tid)) for _ in range(4)])
            for t in range(T)]
    [t.start() for t in warm]
    [t.join() for t in warm]
    th = [threading.Thread(target=worker, args=(t,)) for t in range(T)]
    t0 = time.perf_counter()
    [t.start() for t in th]
    [t.join() for t in th]
    dt = time.perf_counter() - t0
    out.append(f"{T}: {T*per/dt:.0f}/s")
print("whisk verifies/s by threads (queues=%s, acc=%s): " % (os.environ["GPU_MAX_HW_QUEUES"], os.environ.get("CURDLE_DEVICE_ACC", "1")) + ", ".join(out), flush=True)
