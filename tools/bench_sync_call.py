#!/usr/bin/env python3
"""Latency of ONE synchronous MSM call (curdle_msm_g1_device on resident inputs and curdle_msm_g1
on host buffers: what the Go drop-in makes) under alternative settings of the library's plan
knobs, one child process per setting (the knobs are read once per process); every child's
result must equal the first setting's.
Usage: python tools/bench_sync_call.py [--variants "A=1;B=2,C=3;..."] [logn | n=<pairs> ...]
(default variants: the library's defaults, the one-pass scatter, host buffers in one copy)"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(arg):
    sys.path.insert(0, os.path.join(ROOT, "go-curdleproofs_amd"))
    sys.path.insert(0, ROOT)
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
    import numpy as np
    import torch
    import curdlemsm as cm
    from bench import uniform_scalars
    cm.init(0)
    n = int(arg[2:]) if arg.startswith("n=") else 1 << int(arg)
    logn = n.bit_length() - 1
    d_pts = torch.empty((n, 12), dtype=torch.int64, device="cuda:0")
    cm.synth_points_walk_device(12345, 6789, n, d_pts.data_ptr())
    sc = uniform_scalars(np.random.default_rng(2), n)
    d_sc = torch.from_numpy(sc.view(np.int64)).to("cuda:0")
    torch.cuda.synchronize()
    res = None
    for _ in range(3):
        res = cm.msm_g1_device(d_pts.data_ptr(), d_sc.data_ptr(), n)
    lat = []
    for _ in range(15 if n >= (1 << 15) else 60):
        torch.cuda.synchronize()
        t = time.perf_counter()
        cm.msm_g1_device(d_pts.data_ptr(), d_sc.data_ptr(), n)
        lat.append((time.perf_counter() - t) * 1e3)
    host = None
    if logn <= 20:
        pts = d_pts.cpu().numpy().view(np.uint64)
        hl = []
        for _ in range(int(os.environ.get("CURDLE_BENCH_HOST_REPS", "6"))):
            t = time.perf_counter()
            r2 = cm.msm_g1(pts, sc)
            hl.append((time.perf_counter() - t) * 1e3)
        assert (r2 == res).all()
        host = float(np.median(hl[1:]))
    print(json.dumps({"n": n, "logn": logn, "variant": os.environ.get("CURDLE_BENCH_VARIANT", ""),
                      "median_ms": round(float(np.median(lat)), 4), "min_ms": round(min(lat), 4),
                      "host_buffers_ms": host and round(host, 4),
                      "result": [int(v) for v in res[:2]]}))


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        return child(sys.argv[2])
    args = sys.argv[1:]
    variants = "DEFAULTS=1;CURDLE_SCATTER=1;CURDLE_HOST_CHUNKS=1"
    if args and args[0] == "--variants":
        variants, args = args[1], args[2:]
    logns = args or ["20", "18", "16"]
    for logn in logns:
        ref = None
        for v in variants.split(";"):
            env = dict(os.environ, CURDLE_BENCH_VARIANT=v)
            for kv in v.split(","):
                if kv:
                    name, val = kv.split("=", 1)
                    env[name] = val
            g = v
            out = subprocess.run([sys.executable, __file__, "--child", str(logn)], env=env, capture_output=True, text=True)
            if out.returncode:
                print("FAILED", logn, g, out.stderr[-2000:])
                return 1
            line = json.loads(out.stdout.strip().splitlines()[-1])
            if ref is None:
                ref = line["result"]
            line["matches_one_stream"] = line["result"] == ref
            del line["result"]
            print(json.dumps(line), flush=True)
            if not line["matches_one_stream"]:
                return 1
    return 0


if __name__ == "__main__":
    sys.exit(main())
