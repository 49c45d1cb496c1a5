#!/usr/bin/env python3
"""bench.py -- BLS12-381 G1 MSM throughput on MI355X (BASELINE.json's metric).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A step is one MSM of N = 2^20 scalar-point pairs (the size the metric is quoted
on; it fits one GPU) with points and scalars already resident in HBM.  With
N > 1 ranks the SAME 2^20-pair MSM is split by Pippenger windows across the
ranks (strong scaling: total work fixed) and the 144-byte partials are
all-gathered over RCCL.  Rank 0 prints ONE JSON line.

Up to --in-flight (default 3) steps are in flight at once through the library's
asynchronous submit / wait pair: every step is still a complete MSM (all GPU
phases, D2H of the window sums, host combine and -- with N > 1 -- the all-gather
and sum), but the latency-bound tail of step i overlaps the accumulation of step
i+1, as it does for a host that verifies many proofs concurrently.  The latency of
one isolated call is reported next to the throughput (config.single_call_ms).

Inputs are synthetic: P_i = (k + i q) G generated on the GPU
(curdle_synth_points_walk_device), scalars uniform in [0, r) from a seeded
generator.  Nothing is cached between steps and nothing is skipped inside the
timed region: every step runs all six phases plus the host window combine.

The oracle (oracle/) is used only by the cpu_baseline leg, as the checker of the
GPU result and as the timed CPU port.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "go-curdleproofs_amd"))
# The library keeps up to 10 HIP streams busy (sort, accumulate, one tail per MSM in flight);
# ROCm's default of 4 hardware queues per process makes some of them share a queue and
# serialise.  Must be set before the HIP runtime initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

R_MOD = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
R_INV = pow(1 << 256, -1, R_MOD)
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
BYTES_PER_PAIR = 128           # 96 B affine point + 32 B scalar, read once
MADS_PER_MADD = 3878           # v_mad_u64_u32 per mixed XYZZ addition in k_accumulate


def uniform_scalars(rng, n):
    """n uniform values in [0, r) as uint64[n, 4] (taken as Montgomery-form fr.Elements)."""
    out = np.zeros((n, 4), dtype=np.uint64)
    have = 0
    while have < n:
        cand = rng.integers(0, 1 << 64, size=(n - have + 64, 4), dtype=np.uint64)
        cand[:, 3] >>= np.uint64(1)
        # exact comparison with r on the top limb, conservative on ties
        ok = cand[:, 3] < np.uint64(R_MOD >> 192)
        cand = cand[ok][: n - have]
        out[have:have + len(cand)] = cand
        have += len(cand)
    return out


def limbs_to_int(row):
    return sum(int(v) << (64 * i) for i, v in enumerate(row))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--logn", type=int, default=20, help="log2 of the MSM size (default: the headline 2^20)")
    ap.add_argument("--in-flight", type=int, default=0,
                    help="MSMs in flight (curdle_msm_g1_device_submit/wait); 1 = strictly one after the other; "
                         "default: 4 for a whole MSM per GPU, 5 for a window-range partial (measured best)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--emulate-world", type=int, default=0,
                    help="diagnostic: on ONE GPU, run only the window range rank 0 of an N-rank job would "
                         "run (no collective); prints the per-rank step time, not a bench line")
    args = ap.parse_args()

    import torch
    import curdlemsm as cm

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("CURDLE_DIST_BACKEND", "nccl")   # "gloo": rehearsal of the N > 1 path on one GPU
        if backend != "nccl":
            local_rank = local_rank % max(1, torch.cuda.device_count())
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(f"cuda:{local_rank}"))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    if not cm.device_available():
        raise SystemExit("bench.py: no HIP device visible; the MSM has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device(f"cuda:{local_rank}")
    cm.init(local_rank)

    n = 1 << args.logn
    # k, q: first two draws of common.Rand(1) (host mirror), as canonical integers
    r1 = cm.Rand(1)
    k = limbs_to_int(r1.get_fr()) * R_INV % R_MOD
    q = limbs_to_int(r1.get_fr()) * R_INV % R_MOD
    d_pts = torch.empty((n, 12), dtype=torch.int64, device=dev)
    cm.synth_points_walk_device(k, q, n, d_pts.data_ptr())
    sc = uniform_scalars(np.random.default_rng(2), n)
    d_sc = torch.from_numpy(sc.view(np.int64)).to(dev)
    torch.cuda.synchronize()

    c = cm.window_bits(n)
    W = cm.num_windows(n, c)
    in_flight = args.in_flight or (4 if world == 1 and args.emulate_world <= 1 else 5)
    depth = max(1, min(in_flight, cm.MSM_SLOTS - 1))
    if world > 1:
        from curdlemsm.distributed import gather_partials, window_partition
        wb, we = window_partition(W, world, rank)
    elif args.emulate_world > 1:
        from curdlemsm.distributed import window_partition
        wb, we = window_partition(W, args.emulate_world, 0)
    else:
        wb, we = 0, W

    host_t = {"submit": 0.0, "wait": 0.0}

    def submit():
        t_ = time.perf_counter()
        tk = cm.msm_g1_device_submit(d_pts.data_ptr(), d_sc.data_ptr(), n, window_bits=c, win_begin=wb, win_end=we)
        host_t["submit"] += time.perf_counter() - t_
        return tk

    def collect(ticket):
        """Result of one step on every rank: wait for this rank's window range, then (N > 1)
        all-gather the 144-byte partials over RCCL and add them."""
        t_ = time.perf_counter()
        part = cm.msm_wait(ticket)
        host_t["wait"] += time.perf_counter() - t_
        if world == 1:
            return part
        return cm.g1_sum(gather_partials(part, device=dev if dist.get_backend() == "nccl" else None))

    def run_steps(count, on_step=None):
        """`count` steps with up to `depth` MSMs in flight; every step is submitted,
        completed and (N > 1) exchanged inside the call."""
        pending, res = [], None
        for _ in range(count):
            if len(pending) == depth:
                res = collect(pending.pop(0))
                if on_step:
                    on_step()
            pending.append(submit())
        while pending:
            res = collect(pending.pop(0))
            if on_step:
                on_step()
        return res

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # timed region: HIP events around the dominant kernel only, on the stream it is launched
    # on (bracketing all ten phases puts ten barrier packets per MSM into the hardware
    # queues and costs the pipeline ~0.1 ms per step); the other phases are timed below
    cm.profile_enable(2)
    result = run_steps(args.warmup)
    kernel_ms = {}

    def record():
        for name, ms in cm.profile_last()["kernels"].items():   # HIP events on the slot's stream
            kernel_ms.setdefault(name, []).append(ms)

    barrier()
    host_t["submit"] = host_t["wait"] = 0.0
    t0 = time.perf_counter()
    result = run_steps(args.steps, record)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = elapsed * 1e3 / args.steps
    value = n * args.steps / elapsed
    host_ms = {k_: round(v * 1e3 / args.steps, 4) for k_, v in host_t.items()}   # host time per step in submit / wait
    # after the timed region: latency of one call with nothing else in flight, and the
    # kernels' durations when they run alone (with several MSMs in flight the HIP-event
    # spans of the timed region include the time a kernel shares the chip with the
    # previous MSM's tail; those overlapped spans are what `roofline` uses)
    lat = []
    solo_ms = {}
    cm.profile_enable(1)
    for _ in range(5):
        barrier()
        t1 = time.perf_counter()
        collect(submit())
        lat.append((time.perf_counter() - t1) * 1e3)
        for name, ms in cm.profile_last()["kernels"].items():
            solo_ms.setdefault(name, []).append(ms)
    single_call_ms = float(np.median(lat))

    if args.emulate_world > 1:
        print(json.dumps({"emulated_world": args.emulate_world, "windows": [wb, we], "ms_per_step_rank0": ms_per_step,
                          "single_call_ms": single_call_ms, "in_flight": depth, "host_ms_per_step": host_ms,
                          "kernel_ms": {k_: round(float(np.mean(v)), 4) for k_, v in kernel_ms.items()}}))
        return
    if rank == 0:
        # dominant kernel: bucket accumulation.  One launch covers this rank's windows
        # over all n pairs; algorithmic bytes per launch = 128 B x n (inputs read once).
        # "(queue)" is not a kernel: it is the time an MSM waited for the shared accumulate stream
        avg = {kname: float(np.mean(v)) for kname, v in kernel_ms.items() if not kname.startswith("(")}
        solo = {kname: round(float(np.mean(v)), 4) for kname, v in solo_ms.items() if not kname.startswith("(")}
        dom = max(avg, key=avg.get) if avg else None
        roofline = None
        if dom:
            ach = BYTES_PER_PAIR * n / (avg[dom] * 1e-3) / 1e9
            traffic = None
            pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
            if os.path.exists(pmc) and world == 1 and args.logn == 20:
                try:
                    traffic = json.load(open(pmc)).get(dom, {}).get("hbm_bytes_per_launch")
                except Exception:
                    traffic = None
            # The kernel is bound by 32-bit integer multiply issue, not by HBM or MFMA, so the
            # explanatory fraction is reported next to the contract's HBM one: lane-level
            # v_mad_u64_u32 per launch (one mixed addition per pair and window, MADS_PER_MADD
            # multiply-adds each) over the measured duration, against 1024 SIMDs x 64 lanes x
            # 2.4 GHz / 4.9 cycles per wave instruction (profiles/r01_ubench_valu.txt).
            valu = None
            if dom == "accumulate":
                mads = float(we - wb) * n * MADS_PER_MADD
                peak = 1024 * 64 * 2.4e9 / 4.9
                ach_v = mads / (avg[dom] * 1e-3)
                valu = {"unit": "lane v_mad_u64_u32 /s", "achieved": ach_v, "peak": peak, "frac": round(ach_v / peak, 4)}
                if solo.get(dom):
                    valu["frac_kernel_alone"] = round(mads / (solo[dom] * 1e-3) / peak, 4)
            roofline = {"bound": "hbm", "kernel": dom, "achieved": round(ach, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(ach / HBM_PEAK_GBS, 6), "traffic": traffic,
                        "kernel_ms": {kname: round(v, 4) for kname, v in avg.items()},
                        "kernel_ms_alone": solo, "valu": valu,
                        "note": "integer-VALU bound (381-bit Montgomery arithmetic), see DESIGN.md"}
        out = {
            "metric": "BLS12-381 G1 MSM scalar-point pairs/sec at N=2^20",
            "value": value, "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "u32", "data": "synthetic",
            "config": {"workload": f"single G1 MSM, N=2^{args.logn} random Fr scalars x walk points, inputs resident in HBM",
                       "n_pairs": n, "window_bits": c, "num_windows": W, "in_flight": depth,
                       "single_call_ms": round(single_call_ms, 4),
                       "parallelism": "single GPU" if world == 1 else f"Pippenger windows split x{world}, all_gather of 144 B partials"},
            "roofline": roofline,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cm, k, q, n, sc, result, d_pts)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(cm, k, q, n, sc, gpu_result, d_pts):
    """The oracle's C Pippenger (a port of the bucket method the reference gets from
    gnark-crypto -- which cannot run here: no Go toolchain) timed on this box's host
    cores on a bounded sample of the same workload, and the GPU result checked against
    the closed form (k sum s_i + q sum i s_i) G."""
    sys.path.insert(0, os.path.join(ROOT, "oracle", "py"))
    import coracle as co
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    sample_log = min(18, int(np.log2(n)))
    m = 1 << sample_log
    pts = d_pts[:m].cpu().numpy().view(np.uint64)
    threads = min(cores, 32)
    t0 = time.perf_counter()
    ref = co.msm_pippenger(pts, sc[:m], threads=threads)
    dt = time.perf_counter() - t0
    # same sample on the GPU must agree bit for bit
    import torch
    d_s = torch.from_numpy(sc[:m].view(np.int64)).to(d_pts.device)
    same = bool((cm.msm_g1_device(d_pts.data_ptr(), d_s.data_ptr(), m) == ref).all())
    # full-size result vs closed form
    s_int = [limbs_to_int(row) * R_INV % R_MOD for row in sc]
    e = (k * (sum(s_int) % R_MOD) + q * (sum(i * v for i, v in enumerate(s_int)) % R_MOD)) % R_MOD
    aff = co.scalar_mul_gen(e)
    exp = co.jac_normalise(np.concatenate([aff, np.array(cm_one_limbs(), dtype=np.uint64)]))
    full_ok = bool((gpu_result == exp).all())
    return {"value": m / dt, "unit": "pairs/s", "cores": threads, "kind": "port",
            "sample": f"one MSM of 2^{sample_log} pairs (prefix of the same inputs), {dt:.2f} s, oracle C Pippenger, {threads} threads",
            "gpu_matches_cpu_on_sample": same, "gpu_full_size_verified": full_ok}


def cm_one_limbs():
    # Montgomery one of Fp (Z = 1) as six uint64 limbs
    one = (1 << 384) % 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB
    return [(one >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(6)]


if __name__ == "__main__":
    main()
