"""Batched compressed-point decoding on the GPU (curdle_g1_decompress_batch) vs the host
decoder, at the sizes the protocol meets: one proof (98 points at ell = 252), one Whisk
shuffle (594), batches.
    python tools/bench_decode.py
"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "go-curdleproofs_amd"))
import numpy as np
import curdlemsm as cm

cm.init(0)
rand = cm.Rand(1)
base = rand.get_g1_affines(256)
enc = [cm.g1_compress(np.concatenate([p, np.array(cm.MONT_ONE_FP if hasattr(cm, "MONT_ONE_FP") else [0x760900000002fffd, 0xebf4000bc40c0002, 0x5f48985753c758ba, 0x77ce585370525745, 0x5c071a97a256ec6d, 0x15f65ec3fa80e493], dtype=np.uint64)])) for p in base]
out = {}
for n in (98, 594, 4096, 65536, 1 << 20):
    blob = b"".join(enc[i % 256] for i in range(n))
    for sub in (True, False):
        pts, st = cm.g1_decompress_batch(blob, sub)
        assert not st.any()
        reps = 5 if n <= 65536 else 2
        t0 = time.perf_counter()
        for _ in range(reps):
            cm.g1_decompress_batch(blob, sub)
        dt = (time.perf_counter() - t0) / reps
        out[f"n={n},subgroup={int(sub)}"] = {"ms": dt * 1e3, "points_per_s": n / dt}
        print(f"n={n} subgroup={int(sub)}: {dt*1e3:.3f} ms  ({n/dt/1e6:.3f} M points/s)", flush=True)
# host decoder, one thread
t0 = time.perf_counter()
for i in range(98):
    cm.g1_decompress(enc[i], True)
dt = (time.perf_counter() - t0) / 98
out["host_per_point_us_subgroup=1"] = dt * 1e6
print(f"host decoder: {dt*1e6:.1f} us per point (sqrt + subgroup test, through the Python binding)")
print(json.dumps(out))
