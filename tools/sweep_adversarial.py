#!/usr/bin/env python3
""""Adversarial sets also timed" -- every family of tools/adversarial_inputs.py at
N in {2^12, 2^16, 2^20} through (a) one synchronous curdle_msm_g1_device call on resident inputs,
(b) curdle_msm_g1 from pageable host slices and, at 2^20, (c) the pipelined submit / wait path
bench.py's headline runs (4 in flight); every result compared with the closed form (k S0 + q S1) G
of the walk points, every family with the kernels' own durations (HIP events, nothing else in
flight) so that the slow phase has a name.  `ratio` = wall time over the uniform control's at the
same N and entry point (the bar: <= 1.25).

    python tools/sweep_adversarial.py [--logn 12,16,20] [--families a,b,...] [--out profiles/r06_adversarial.json]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "go-curdleproofs_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle", "py"))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--logn", default="12,16,20")
    ap.add_argument("--families", default="")
    ap.add_argument("--out", default="")
    ap.add_argument("--no-host", action="store_true")
    args = ap.parse_args()
    import torch
    import curdlemsm as cm
    import coracle as co
    import adversarial_inputs as adv
    from bench import uniform_scalars, limbs_to_int, cm_one_limbs

    cm.init(0)
    r1 = cm.Rand(1)
    k = limbs_to_int(r1.get_fr()) * adv.R_INV % adv.R_MOD
    q = limbs_to_int(r1.get_fr()) * adv.R_INV % adv.R_MOD
    one = np.array(cm_one_limbs(), dtype=np.uint64)

    def expected(e):
        return co.jac_normalise(np.concatenate([co.scalar_mul_gen(e), one]))

    fams = [f for f in (args.families.split(",") if args.families else adv.FAMILIES)]
    rows, worst = [], 0.0
    for logn in [int(x) for x in args.logn.split(",")]:
        n = 1 << logn
        d_pts = torch.empty((n, 12), dtype=torch.int64, device="cuda:0")
        cm.synth_points_walk_device(k, q, n, d_pts.data_ptr())
        torch.cuda.synchronize()
        uniform = uniform_scalars(np.random.default_rng(2), n)
        c = cm.window_bits(n)
        base = {}
        for fam in fams:
            sc, dead = adv.make_family(fam, n, uniform, window_bits=c)
            sc = np.ascontiguousarray(sc)
            pts_d = d_pts
            if dead is not None:
                pts_d = d_pts.clone()
                pts_d[torch.from_numpy(np.asarray(dead, dtype=np.int64)).to("cuda:0")] = 0
            exp = expected(adv.walk_exponent(k, q, sc, dead))
            d_sc = torch.from_numpy(sc.view(np.int64)).to("cuda:0")
            pp, sp = pts_d.data_ptr(), d_sc.data_ptr()
            torch.cuda.synchronize()
            cm.profile_enable(0)
            for _ in range(3):
                res = cm.msm_g1_device(pp, sp, n)
            ok = bool((res == exp).all())
            lat = []
            for _ in range(15 if n >= (1 << 16) else 40):
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                cm.msm_g1_device(pp, sp, n)
                lat.append((time.perf_counter() - t1) * 1e3)
            cm.profile_enable(1)
            ks, counts = {}, {}
            for _ in range(5):
                cm.msm_g1_device(pp, sp, n)
                pr = cm.profile_last()
                for name, ms in pr["kernels"].items():
                    if not name.startswith("("):
                        ks.setdefault(name, []).append(ms)
                counts = {"entries": pr["entries"], "fragments": pr["fragments"]}
            cm.profile_enable(0)
            row = {"family": fam, "logn": logn, "window_bits": c, "sync_ms": round(float(np.median(lat)), 4),
                   "sync_min_ms": round(min(lat), 4), "matches_closed_form": ok,
                   "kernel_ms_alone": {a: round(float(np.mean(b)), 4) for a, b in ks.items()}, **counts}
            if not args.no_host:
                pts_h = pts_d.cpu().numpy().view(np.uint64)
                hb = []
                for _ in range(6 if n >= (1 << 16) else 20):
                    t1 = time.perf_counter()
                    r_h = cm.msm_g1(pts_h, sc)
                    hb.append((time.perf_counter() - t1) * 1e3)
                row["host_slices_ms"] = round(float(np.median(hb[1:])), 4)
                row["matches_closed_form"] = ok = ok and bool((r_h == exp).all())
            if logn >= 20:
                depth, steps = 4, 24

                def run(count):
                    pend, last = [], None
                    for _ in range(count):
                        if len(pend) == depth:
                            last = cm.msm_wait(pend.pop(0))
                        pend.append(cm.msm_g1_device_submit(pp, sp, n))
                    while pend:
                        last = cm.msm_wait(pend.pop(0))
                    return last
                run(depth + 2)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                r_p = run(steps)
                torch.cuda.synchronize()
                row["pipelined_ms_per_step"] = round((time.perf_counter() - t1) * 1e3 / steps, 4)
                row["matches_closed_form"] = ok = ok and bool((r_p == exp).all())
            if fam == "uniform":
                base = dict(row)
            if base:
                row["ratio"] = {key.replace("_ms", "").replace("_per_step", ""): round(row[key] / base[key], 3)
                                for key in ("sync_ms", "host_slices_ms", "pipelined_ms_per_step") if key in row and key in base}
                worst = max([worst] + list(row["ratio"].values()))
            rows.append(row)
            print(json.dumps(row), file=sys.stderr, flush=True)
            del d_sc
    out = {"what": "adversarial scalar / base families (SURVEY.md 8d) against the uniform control: wall time of one synchronous "
                   "resident call, of curdle_msm_g1 from pageable host slices, and (2^20) of the pipelined path, 4 in flight",
           "bar": "every family within 1.25x of uniform at the same N and entry point", "worst_ratio": worst,
           "all_match_closed_form": all(r["matches_closed_form"] for r in rows), "rows": rows}
    print(json.dumps(out), flush=True)
    if args.out:
        with open(os.path.join(ROOT, args.out), "w") as f:
            json.dump(out, f, indent=1)
    if not out["all_match_closed_form"]:
        raise SystemExit("sweep_adversarial: a result differs from the closed form")


if __name__ == "__main__":
    main()
