#!/usr/bin/env python3
"""Pipelined step time (submit / wait, F MSMs in flight) at N = 2^logn, inputs resident -- bench.py's
timed loop without its checks, for A/B runs of plan knobs: one child process per variant.
Usage: python tools/bench_pipeline.py [--variants "A=1;B=2,C=3"] [--logn 20] [--in-flight 4] [--steps 40]
       [--windows wb:we]"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(logn, depth, steps, wb, we):
    sys.path.insert(0, os.path.join(ROOT, "go-curdleproofs_amd"))
    sys.path.insert(0, ROOT)
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
    import numpy as np
    import torch
    import curdlemsm as cm
    from bench import uniform_scalars
    cm.init(0)
    if os.environ.get("CURDLE_DEBUG_SKIP"):      # experiment build only (tools/exp/build_alt.sh -DCURDLE_EXP_SKIP msm_enqueue)
        import ctypes
        ctypes.CDLL(cm.LIB_PATH).curdle_debug_skip(ctypes.c_uint(int(os.environ["CURDLE_DEBUG_SKIP"])))
    n = 1 << logn
    d_pts = torch.empty((n, 12), dtype=torch.int64, device="cuda:0")
    cm.synth_points_walk_device(12345, 6789, n, d_pts.data_ptr())
    sc = uniform_scalars(np.random.default_rng(2), n)
    d_sc = torch.from_numpy(sc.view(np.int64)).to("cuda:0")
    torch.cuda.synchronize()
    c = cm.window_bits(n) if we >= 0 else 0

    def run(count):
        pending, res = [], None
        for _ in range(count):
            if len(pending) == depth:
                res = cm.msm_wait(pending.pop(0))
            pending.append(cm.msm_g1_device_submit(d_pts.data_ptr(), d_sc.data_ptr(), n, window_bits=c, win_begin=wb, win_end=we))
        while pending:
            res = cm.msm_wait(pending.pop(0))
        return res
    run(depth + 1)
    run(10)
    best = None
    for _ in range(3):
        torch.cuda.synchronize()
        t = time.perf_counter()
        res = run(steps)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) * 1e3 / steps
        best = dt if best is None else min(best, dt)
    print(json.dumps({"logn": logn, "variant": os.environ.get("CURDLE_BENCH_VARIANT", ""), "in_flight": depth,
                      "windows": [wb, we], "ms_per_step": round(best, 4), "mpairs_s": round(n / best / 1e3, 1),
                      "result": [int(v) for v in res[:2]]}))


def main():
    a = sys.argv[1:]
    opt = {"--variants": "DEFAULTS=1", "--logn": "20", "--in-flight": "4", "--steps": "40", "--windows": "0:-1"}
    while a and a[0] in opt:
        opt[a[0]] = a[1]
        a = a[2:]
    wb, we = (int(x) for x in opt["--windows"].split(":"))
    if a and a[0] == "--child":
        return child(int(opt["--logn"]), int(opt["--in-flight"]), int(opt["--steps"]), wb, we)
    for v in opt["--variants"].split(";"):
        env = dict(os.environ, CURDLE_BENCH_VARIANT=v)
        for kv in v.split(","):
            if kv:
                name, val = kv.split("=", 1)
                env[name] = val
        out = subprocess.run([sys.executable, __file__, "--logn", opt["--logn"], "--in-flight", opt["--in-flight"], "--steps",
                              opt["--steps"], "--windows", opt["--windows"], "--child"], env=env, capture_output=True, text=True)
        if out.returncode:
            print("FAILED", v, out.stderr[-1500:])
            return 1
        print(out.stdout.strip().splitlines()[-1], flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
