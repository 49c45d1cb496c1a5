#!/usr/bin/env python3
"""Large batches of the latency-bound kernels (point decoding with the subgroup test, batched
scalar multiplication): four lanes per point (quad28.h, no spills) against one lane per point
(256 VGPRs, 20-131 of them spilled), per batch size.  One child process per setting.
(Since the end of round 3 only the decoding kernels still have a one-lane build; the scalar
multiplication's was removed on this tool's numbers -- profiles/r03_quad_vs_lane.txt -- and its
columns are the same kernel in both settings now.)
    python tools/bench_quad_vs_lane.py"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ONE = [0x760900000002fffd, 0xebf4000bc40c0002, 0x5f48985753c758ba, 0x77ce585370525745, 0x5c071a97a256ec6d, 0x15f65ec3fa80e493]


def child():
    sys.path.insert(0, os.path.join(ROOT, "go-curdleproofs_amd"))
    import numpy as np
    import curdlemsm as cm
    cm.init(0)
    rand = cm.Rand(1)
    base = rand.get_g1_affines(256)
    enc = [cm.g1_compress(np.concatenate([p, np.array(ONE, dtype=np.uint64)])) for p in base]
    sn256 = np.stack([rand.get_fr() for _ in range(256)])
    out = {"setting": os.environ.get("CURDLE_QUAD_MAX_LANES", "default")}
    for n in (16384, 40000, 65536, 262144, 1 << 20):
        blob = b"".join(enc[i % 256] for i in range(n))
        os.environ["CURDLE_TWO_KERNEL_MAX"] = "0"
        pts, st = cm.g1_decompress_batch(blob, True)
        assert not st.any()
        t0 = time.perf_counter()
        for _ in range(3):
            cm.g1_decompress_batch(blob, True)
        out[f"decode_n={n}_ms"] = round((time.perf_counter() - t0) / 3 * 1e3, 3)
        if n <= 262144:
            P = np.concatenate([base] * ((n + 255) // 256))[:n].copy()
            sn = np.concatenate([sn256] * ((n + 255) // 256))[:n].copy()
            r0 = cm.g1_scalar_mul_batch(P, sn, None)
            t0 = time.perf_counter()
            for _ in range(3):
                cm.g1_scalar_mul_batch(P, sn, None)
            out[f"scalar_mul_n={n}_ms"] = round((time.perf_counter() - t0) / 3 * 1e3, 3)
            out[f"check_{n}"] = int(r0[:: max(1, n // 64)].astype(np.uint64).sum() % (1 << 61))
    print(json.dumps(out))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child()
    else:
        ref = None
        for setting in ("131072", "100000000"):
            env = dict(os.environ, CURDLE_QUAD_MAX_LANES=setting, CURDLE_TWO_KERNEL_MAX="0")
            p = subprocess.run([sys.executable, __file__, "--child"], env=env, capture_output=True, text=True)
            if p.returncode:
                print("FAILED", setting, p.stderr[-1500:])
                sys.exit(1)
            line = json.loads(p.stdout.strip().splitlines()[-1])
            checks = {k: v for k, v in line.items() if k.startswith("check_")}
            if ref is None:
                ref = checks
            line = {k: v for k, v in line.items() if not k.startswith("check_")}
            line["same_results"] = checks == ref
            print(json.dumps(line), flush=True)
