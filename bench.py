#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on MI355X: BLS12-381 G1 MSM scalar-point pairs/sec at
N = 2^20, and Curdleproofs verifies/sec at N = 252 (carried in the same JSON line under
"verify"; `--mode verify` prints it as a line of its own).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A step is one MSM of N = 2^20 scalar-point pairs (the size the metric is quoted on; it fits
one GPU) with points and scalars already resident in HBM.  With N > 1 ranks the SAME
2^20-pair MSM is split by Pippenger windows across the ranks (strong scaling: total work
fixed) and the 144-byte partials are all-gathered over RCCL.  Rank 0 prints ONE JSON line.

Up to --in-flight steps are in flight at once through the library's asynchronous submit /
wait pair: every step is still a complete MSM (all GPU phases, D2H of the window sums, host
combine and -- with N > 1 -- the all-gather and sum), but the latency-bound tail of step i
overlaps the accumulation of step i+1, as it does for a host that verifies many proofs
concurrently.  The latency of ONE isolated synchronous call is reported next to the
throughput (config.single_call_ms).

Inputs are synthetic: P_i = (k + i q) G generated on the GPU, scalars uniform in [0, r) from
a seeded generator.  Nothing is cached between steps and nothing is skipped inside the timed
region.  Roofline: `achieved` = algorithmic bytes (128 B per pair) / the
dominant kernel's duration measured with HIP events on its own stream while NOTHING else
runs (so it is a kernel duration, always <= ms_per_step; the rocprofv3 average of the same
kernel under profiles/ must agree); the span of the same kernel inside the pipelined timed
region, where it shares the chip with the previous MSM's tail, is reported separately.

The oracle (oracle/) is used only by the cpu_baseline leg: as the checker of the GPU result
and as the timed CPU port (oracle/cpu_msm_fast.c on every host core the process may use).
bench.py exits non-zero if either check fails.
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "go-curdleproofs_amd"))
# The library keeps up to 10 HIP streams busy (sort, accumulate, one tail per MSM in flight);
# ROCm's default of 4 hardware queues per process makes some of them share a queue and
# serialise.  Must be set before the HIP runtime initialises -- under rocprofv3 the profiler's
# preload initialises HIP before Python runs, so export it in the shell there
# (tools/refresh_profiles.sh does).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

R_MOD = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
R_INV = pow(1 << 256, -1, R_MOD)
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
BYTES_PER_PAIR = 128           # 96 B affine point + 32 B scalar, read once
# v_mad_u64_u32 per mixed XYZZ addition in k_accumulate (csrc/fp28.h madd): 6 mul (406 each)
# + 2 sqr (315) + one fused two-product mul2 (602)
MADS_PER_MADD = 6 * 406 + 2 * 315 + 602
VERIFY_ELL = 252
VERIFY_PAIRS = 10 * 8 + (5 * VERIFY_ELL + 8)      # 1,348 pairs = 172,544 algorithmic bytes per verify
REFERENCE_README_VERIFIES_PER_S = 65.5            # /root/reference README.md:25, Ryzen 3800XT, 16 threads (published context)


def uniform_scalars(rng, n):
    """n uniform values in [0, r) as uint64[n, 4] (taken as Montgomery-form fr.Elements)."""
    out = np.zeros((n, 4), dtype=np.uint64)
    have = 0
    while have < n:
        cand = rng.integers(0, 1 << 64, size=(n - have + 64, 4), dtype=np.uint64)
        cand[:, 3] >>= np.uint64(1)
        ok = cand[:, 3] < np.uint64(R_MOD >> 192)   # strictly below r's top limb
        cand = cand[ok][: n - have]
        out[have:have + len(cand)] = cand
        have += len(cand)
    return out


def limbs_to_int(row):
    return sum(int(v) << (64 * i) for i, v in enumerate(row))


def host_cores():
    """Cores this process may actually use: the affinity mask capped by the cgroup CPU quota
    (the GPU box shows 256 CPUs but grants 16 to a one-GPU lease: cpu.max = 1600000 100000)."""
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        cores = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            cores = max(1, min(cores, int(int(quota) / int(period))))
    except Exception:
        pass
    return cores


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--mode", choices=["msm", "verify", "whisk-batch"], default="msm",
                    help="msm: the pairs/s line (with the verify and config-5 figures attached); verify: the verifies/s "
                         "line only; whisk-batch: BASELINE config 5 -- 1,024 IsValidWhiskShuffleProof verifications, "
                         "replicas over the ranks, every rank working")
    ap.add_argument("--sweep", action="store_true",
                    help="north_star's size sweep as one artifact: N = 2^10..2^20 (one GPU), per size the synchronous "
                         "call's wall time, pairs/s, the dominant kernel with its HBM and multiply-issue fractions, "
                         "and the CPU port on the same inputs")
    ap.add_argument("--proofs", type=int, default=1024, help="whisk-batch: proofs per step (all ranks together)")
    ap.add_argument("--logn", type=int, default=20, help="log2 of the MSM size (default: the headline 2^20)")
    ap.add_argument("--in-flight", type=int, default=0,
                    help="MSMs in flight (curdle_msm_g1_device_submit/wait); 1 = strictly one after the other; "
                         "default: 4 for a whole MSM per GPU, 6 for a window-range partial (measured best)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true", help="skip the verifies/s leg")
    ap.add_argument("--split", choices=["windows", "points"], default="windows",
                    help="multi-GPU partition: Pippenger windows (north_star) or point ranges (diagnostic)")
    ap.add_argument("--force-dist", action="store_true",
                    help="take the N > 1 code path at ANY world size, 1 included: torch.distributed process group on the "
                         "chosen backend (nccl = RCCL unless CURDLE_DIST_BACKEND says otherwise), partials exchanged by "
                         "all_gather, accept bits of the config-5 leg too -- so that everything a multi-GPU run executes "
                         "for the first time can be executed on a one-GPU lease (tests/test_bench_cli.py)")
    ap.add_argument("--sweep-sizes", default="",
                    help="--sweep: comma-separated pair counts instead of the default list (tests use a short one)")
    ap.add_argument("--resident-bases", action="store_true",
                    help="diagnostic: the timed steps run curdle_msm_g1_dbases_submit over a resident, pre-converted base "
                         "set instead of gnark-layout points (never the headline `value`: the default run reports this "
                         "figure beside it as config.resident_bases)")
    ap.add_argument("--bases-unchanged", action="store_true",
                    help="diagnostic: the timed steps pass CURDLE_MSM_BASES_UNCHANGED (curdle_msm_g1_device_submit_ex): "
                         "the library converts the resident gnark-layout bases once and keeps its copy -- what a rank of the "
                         "window split may do with inputs that stay put between calls (never the headline `value`)")
    ap.add_argument("--convert-per-call", action="store_true",
                    help="diagnostic: a window-split rank (N > 1 or --emulate-world) converts its gnark-layout bases on EVERY "
                         "call instead of passing CURDLE_MSM_BASES_UNCHANGED (the default there since round 6: the inputs "
                         "are resident and unchanged between steps, which is the flag's contract)")
    ap.add_argument("--exchange-batch", type=int, default=8,
                    help="N > 1: partials of this many steps travel in one all_gather (144 x k bytes per rank); 1 = one collective "
                         "per step.  Every step's result is gathered and summed inside the timed region either way")
    ap.add_argument("--emulate-world", type=int, default=0,
                    help="diagnostic: on ONE GPU, run only the share rank 0 of an N-rank job would run "
                         "(no collective); prints the per-rank step time, not a bench line")
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
    if world > 1 or args.force_dist or os.environ.get("CURDLE_FORCE_DIST") == "1":
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1 and "MASTER_PORT" not in os.environ:     # a one-rank group needs a rendezvous too
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
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

    if args.sweep:
        if dist is not None:
            sweep_distributed(cm, torch, dist, dev, rank, world, args)
        elif rank == 0:
            sweep(cm, torch, dev, args)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    if args.mode == "whisk-batch":
        line = whisk_batch_leg(cm, torch, dist, dev, rank, world, args.proofs, args.steps, args.warmup)
        if rank == 0:
            print(json.dumps(line), flush=True)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    if args.mode == "verify":
        if rank == 0:
            v = verify_leg(cm, args.steps, args.warmup)
            line = {"metric": "Curdleproofs verifies/sec N=252", "value": v["value"], "unit": "verifies/s", "n_gpus": 1,
                    "steps": v["verifies_timed"], "warmup": args.warmup, "ms_per_step": v["ms_per_verify"],
                    "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
                    "config": {"workload": v["workload"]}, "roofline": v["roofline"], "detail": v}
            print(json.dumps(line), flush=True)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

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
    parts = world if world > 1 else max(1, args.emulate_world)
    my_part = rank if world > 1 else 0
    in_flight = args.in_flight or (4 if parts == 1 else 6)
    depth = max(1, min(in_flight, cm.MSM_SLOTS - 1))
    from curdlemsm.distributed import PartialExchange, point_partition, window_partition
    wb, we, p_lo, p_hi = 0, W, 0, n
    if parts > 1 and args.split == "windows":
        wb, we = window_partition(W, parts, my_part)
    elif parts > 1:
        p_lo, p_hi = point_partition(n, parts, my_part)
    n_mine = p_hi - p_lo
    c_mine = c if args.split == "windows" else 0      # a point range picks its own window width
    pts_ptr = d_pts.data_ptr() + p_lo * 96
    sc_ptr = d_sc.data_ptr() + p_lo * 32

    # A rank of the window split walks ALL n points for its few windows: converting them on every call (0.06 ms of a 0.4 ms
    # step at 8 ranks) is what the flag's contract saves when the resident inputs stay put -- which they do here, step after
    # step, exactly as msmaccumulator.Verify's mostly-CRS bases do (msmaccumulator.go:59).  N = 1 keeps per-call conversion as
    # the headline `value` and reports the kept-bases figure beside it (config.bases_unchanged): the scaling baseline of the
    # SAME contract.
    keep_bases = bool(args.bases_unchanged or (parts > 1 and args.split == "windows" and not args.convert_per_call))
    msm_flags = cm.MSM_BASES_UNCHANGED if keep_bases else 0
    host_t = {"submit": 0.0, "wait": 0.0}
    # the same pairs as a resident, pre-converted base set (curdle_dbases): made lazily, used by the
    # resident-bases figure beside the headline and by --resident-bases
    bases_box = {"set": None, "on": bool(args.resident_bases)}

    def bases():
        if bases_box["set"] is None:
            bases_box["set"] = cm.DBases(d_pts[p_lo:p_hi].cpu().numpy().view(np.uint64))
        return bases_box["set"]

    def submit():
        t_ = time.perf_counter()
        if bases_box["on"]:
            tk = bases().submit(sc_ptr, n_mine, window_bits=c_mine, win_begin=wb, win_end=we if args.split == "windows" else -1)
        else:
            tk = cm.msm_g1_device_submit(pts_ptr, sc_ptr, n_mine, window_bits=c_mine, win_begin=wb,
                                         win_end=we if args.split == "windows" else -1,
                                         flags=msm_flags)
        host_t["submit"] += time.perf_counter() - t_
        return tk

    exchange = PartialExchange(device=dev if dist.get_backend() == "nccl" else None) if dist is not None else None
    exchanging = []     # (N > 1) the all_gather of the step before, finished one step later

    batch_k = max(1, args.exchange_batch)
    gathered = []       # this rank's partials waiting for the next all_gather

    def sum_batch(allp, k_):
        """uint64[world, k_ * 18] -> the k_ full results, one host sum each"""
        per = allp.reshape(allp.shape[0], k_, 18)
        return [cm.g1_sum(np.ascontiguousarray(per[:, j, :])) for j in range(k_)]

    def collect(ticket):
        """Result of one step on every rank: wait for this rank's share, then (N > 1)
        all-gather the 144-byte partials over RCCL and add them.  Partials of --exchange-batch
        steps travel in ONE all_gather (144 x k bytes per rank: fewer, larger collectives --
        the collective's kernel and its two small copies cost a step 0.1 ms of GPU time beside
        an accumulation that fills the chip, whatever they carry), and the exchange started
        here is finished when the next one is started (or by flush()), so that its wait for a
        free wave slot is not in every step.  Every step's result is still gathered to every
        rank and summed inside the timed region."""
        t_ = time.perf_counter()
        part = cm.msm_wait(ticket)
        host_t["wait"] += time.perf_counter() - t_
        if dist is None:
            return part
        t_ = time.perf_counter()
        res = None
        gathered.append(part)
        if len(gathered) == batch_k:
            exchanging.append((exchange.start(np.concatenate(gathered)), len(gathered)))
            gathered.clear()
            if len(exchanging) > 1:
                h, k_ = exchanging.pop(0)
                res = sum_batch(exchange.finish(h), k_)[-1]
        host_t["exchange"] = host_t.get("exchange", 0.0) + time.perf_counter() - t_
        return res

    def flush():
        res = None
        if gathered:
            exchanging.append((exchange.start(np.concatenate(gathered)), len(gathered)))
            gathered.clear()
        while exchanging:
            h, k_ = exchanging.pop(0)
            res = sum_batch(exchange.finish(h), k_)[-1]
        return res

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
        if dist is not None:
            res = flush()
        return res

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # timed region: HIP events around the dominant kernel only, on the stream it is launched
    # on (bracketing all ten phases puts ten barrier packets per MSM into the hardware
    # queues and costs the pipeline ~0.1 ms per step); the other phases are timed below
    cm.profile_enable(2)
    # set-up, like generating the inputs: every workspace slot the pipeline will rotate over
    # allocates its buffers the first time it is used (hipMalloc of hundreds of megabytes at
    # 2^22 pairs and beyond); with fewer warm-up steps than slots in flight those allocations
    # would land in the timed region
    run_steps(depth + 1)
    result = run_steps(args.warmup)
    span_ms = {}

    def record():
        for name, ms in cm.profile_last()["kernels"].items():   # HIP events on the stream the kernel runs on
            span_ms.setdefault(name, []).append(ms)

    barrier()
    host_t["submit"] = host_t["wait"] = 0.0
    host_t.pop("exchange", None)
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

    # After the timed region, with nothing else in flight: every kernel's own duration (HIP
    # events, same streams), the sorted-entry and fragment counts, and the latency of one
    # isolated synchronous call (the entry point a single caller uses).
    solo_ms, counts = {}, {}
    cm.profile_enable(1)
    for _ in range(5):
        barrier()
        collect(submit())
        if dist is not None:
            flush()
        pr = cm.profile_last()
        for name, ms in pr["kernels"].items():
            solo_ms.setdefault(name, []).append(ms)
        counts = {"entries": pr["entries"], "fragments": pr["fragments"]}
    lat = []
    cm.profile_enable(0)
    for _ in range(5):
        barrier()
        t1 = time.perf_counter()
        if args.split == "windows":
            cm.msm_g1_device(pts_ptr, sc_ptr, n_mine, window_bits=c_mine, win_begin=wb, win_end=we, flags=msm_flags)
        else:
            cm.msm_g1_device(pts_ptr, sc_ptr, n_mine)
        lat.append((time.perf_counter() - t1) * 1e3)
    single_call_ms = float(np.median(lat))

    # What an N > 1 line must say beside `value`: the pipelined figure is a THROUGHPUT (6 window-range
    # calls in flight per rank, the exchange of step i finished during step i + 1); one synchronous call -- what
    # msmaccumulator.Verify issues -- scales by latency, which is another number.  So: one whole distributed call
    # (barrier, every rank's synchronous window-range call, all_gather, sum; MAX over ranks), the exchange alone, and the
    # SAME contract at one rank (the whole MSM with the flag on this rank's own GPU: the baseline a speed-up is a ratio to).
    multi = None
    if dist is not None and not args.emulate_world:   # (--emulate-world with --force-dist: one rank's share WITH the exchange, at world size 1)
        from curdlemsm.distributed import msm_g1_distributed
        on_gpu = dist.get_backend() == "nccl"

        def max_over_ranks(x):
            t = torch.tensor([x], dtype=torch.float64, device=dev if on_gpu else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item())

        whole, exch = [], []
        for _ in range(7):
            barrier()
            t1 = time.perf_counter()
            r_d = msm_g1_distributed(d_pts.data_ptr(), d_sc.data_ptr(), n, device=dev if on_gpu else None, c=c,
                                     split=args.split, flags=msm_flags)
            whole.append(max_over_ranks((time.perf_counter() - t1) * 1e3))
        for _ in range(7):
            barrier()
            t1 = time.perf_counter()
            exchange.finish(exchange.start(result if result is not None else np.zeros(18, dtype=np.uint64)))
            exch.append(max_over_ranks((time.perf_counter() - t1) * 1e3))
        base_ms = None
        if args.split == "windows":
            def run_whole(count):
                pend = []
                for _ in range(count):
                    if len(pend) == 4:
                        cm.msm_wait(pend.pop(0))
                    pend.append(cm.msm_g1_device_submit(d_pts.data_ptr(), d_sc.data_ptr(), n, flags=msm_flags))
                last = None
                while pend:
                    last = cm.msm_wait(pend.pop(0))
                return last
            run_whole(6)
            barrier()
            t1 = time.perf_counter()
            r_w = run_whole(20)
            torch.cuda.synchronize()
            base_ms = max_over_ranks((time.perf_counter() - t1) * 1e3 / 20)
            if result is not None and not (r_w == result).all():
                raise SystemExit("bench.py: the one-rank baseline and the distributed result differ")
        if result is not None and not (r_d == result).all():
            raise SystemExit("bench.py: a synchronous distributed call and the pipelined steps differ")
        multi = {"rccl_ranks": dist.get_world_size(), "backend": dist.get_backend(), "exchange_batch": batch_k,
                 "bases_unchanged": keep_bases,
                 "rank_step_ms": round(ms_per_step, 4),
                 "rank_single_call_ms": round(single_call_ms, 4),
                 "whole_single_call_ms": round(float(np.median(whole)), 4),
                 "exchange_ms": round(float(np.median(exch)), 4),
                 "scaling_baseline": None if base_ms is None else {
                     "what": "the SAME contract at one rank: the whole MSM, 4 in flight, "
                             + ("CURDLE_MSM_BASES_UNCHANGED" if keep_bases else "bases converted per call")
                             + ", run by every rank on its own GPU right after the timed region (MAX over ranks)",
                     "ms_per_step": round(base_ms, 4), "pairs_per_s": round(n / base_ms * 1e3, 1)},
                 "note": "`value` is pipelined throughput (in_flight window-range calls per rank; the partials of exchange_batch "
                         "steps travel in one all_gather, finished while the next steps run); ONE synchronous call scales as whole_single_call_ms against "
                         "config.single_call.ms_per_call of the N = 1 line.  No scaling curve was measured by the builder "
                         "(one-GPU leases): the driver computes efficiency from its own runs."}
        assert multi["rccl_ranks"] == args.gpus == world, (multi["rccl_ranks"], args.gpus, world)

    # The same MSM over a RESIDENT, pre-converted base set (curdle_msm_g1_dbases*): what a caller whose bases
    # do not change between calls gets (msmaccumulator.Verify's are mostly the CRS; every rank of a window
    # split otherwise converts ALL points again).  Beside the headline at every N, never `value`, which stays
    # on gnark-layout inputs.  All ranks run it (the steps exchange their partials like the timed ones).
    resident = None
    if args.logn <= 22 and not args.no_verify and not bases_box["on"] and not args.emulate_world:
        bases_box["on"] = True
        try:
            run_steps(depth + 1)
            barrier()
            t1 = time.perf_counter()
            r_b = run_steps(20)
            barrier()
            rb_s = time.perf_counter() - t1
            if dist is not None:
                t = torch.tensor([rb_s], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                rb_s = float(t.item())
            rb_ms = rb_s * 1e3 / 20
            resident = {"entry_points": "curdle_msm_g1_dbases_submit / _dbases / _dbases_host (bases converted once, resident)",
                        "pipelined_ms_per_step": round(rb_ms, 4), "pipelined_pairs_per_s": round(n / rb_ms * 1e3, 1),
                        "ok": bool((r_b == result).all())}
            if dist is None:
                lat_b, lat_h = [], []
                for _ in range(5):
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    r_s = bases().msm(sc_ptr, n_mine)
                    lat_b.append((time.perf_counter() - t1) * 1e3)
                for _ in range(5):
                    t1 = time.perf_counter()
                    r_h2 = bases().msm_host(sc)
                    lat_h.append((time.perf_counter() - t1) * 1e3)
                resident["ok"] = resident["ok"] and bool((r_s == result).all() and (r_h2 == result).all())
                resident.update({"single_call_ms": round(float(np.median(lat_b)), 4),
                                 "host_scalars_ms": round(float(np.median(lat_h[1:])), 4),
                                 "host_scalars_pairs_per_s": round(n / float(np.median(lat_h[1:])) * 1e3, 1)})
        finally:
            bases_box["on"] = False

    # ... and with CURDLE_MSM_BASES_UNCHANGED on the SAME gnark-layout device array (the library keeps its converted copy):
    # the one-rank figure of the contract every rank of an N > 1 window split runs under -- the scaling baseline.
    kept = None
    if dist is None and args.logn <= 22 and not args.no_verify and not keep_bases and not args.emulate_world and not args.resident_bases:
        msm_flags = cm.MSM_BASES_UNCHANGED
        try:
            run_steps(depth + 1)
            barrier()
            t1 = time.perf_counter()
            r_k = run_steps(20)
            barrier()
            k_ms = (time.perf_counter() - t1) * 1e3 / 20
            kept = {"entry_points": "curdle_msm_g1_device_submit_ex / _device_ex with CURDLE_MSM_BASES_UNCHANGED (gnark-layout "
                                    "device bases converted once, the library keeps its copy): the contract of a window-split rank",
                    "pipelined_ms_per_step": round(k_ms, 4), "pipelined_pairs_per_s": round(n / k_ms * 1e3, 1),
                    "ok": bool((r_k == result).all())}
        finally:
            msm_flags = 0
            cm.msm_forget_bases(pts_ptr)

    if args.emulate_world > 1:
        print(json.dumps({"emulated_world": args.emulate_world, "split": args.split, "windows": [wb, we],
                          "bases_unchanged_flag": keep_bases,
                          "resident_bases": bases_box["on"],
                          "points": [p_lo, p_hi], "ms_per_step_rank0": ms_per_step,
                          "single_call_ms": single_call_ms, "in_flight": depth, "host_ms_per_step": host_ms,
                          "kernel_ms_alone": {k_: round(float(np.mean(v)), 4) for k_, v in solo_ms.items()}}))
        return
    # north_star's size table on an N > 1 line too (so that the driver's 2 / 4 / 8-GPU runs fill the "N in {2^10 .. 2^20} at
    # 1 / 2 / 4 / 8 GPUs" table by themselves): every size through ONE synchronous distributed call on ALL ranks (window
    # ranges under the line's contract; barrier, call, MAX over the ranks; median of 7); rank 0 adds the CPU port below.
    dsweep = None
    if dist is not None and not args.emulate_world and args.logn == 20 and not args.no_cpu_baseline and args.split == "windows":
        from curdlemsm.distributed import msm_g1_distributed
        on_gpu = dist.get_backend() == "nccl"
        dsweep = []
        for lg in range(10, 21, 2):
            m = 1 << lg
            call = lambda: msm_g1_distributed(d_pts.data_ptr(), d_sc.data_ptr(), m, device=dev if on_gpu else None, split="windows",
                                              flags=msm_flags)
            for _ in range(3):
                r_m = call()
            lat = []
            for _ in range(7):
                barrier()
                t1 = time.perf_counter()
                call()
                t = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device=dev if on_gpu else "cpu")
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                lat.append(float(t.item()) * 1e3)
            dsweep.append((lg, float(np.median(lat)), r_m))
        cm.msm_forget_bases(d_pts.data_ptr())      # (the prefixes' kept copies; the whole array's is made again on its next use)
    ok = True
    if rank == 0:
        # dominant kernel: bucket accumulation.  One launch covers this rank's windows over
        # all n pairs; algorithmic bytes per launch = 128 B x n (inputs read once).
        # "(queue)" is not a kernel: it is the time an MSM waited for the shared accumulate stream
        solo = {kname: round(float(np.mean(v)), 4) for kname, v in solo_ms.items() if not kname.startswith("(")}
        span = {kname: round(float(np.mean(v)), 4) for kname, v in span_ms.items() if not kname.startswith("(")}
        # the bucket accumulation: named, not picked by duration -- in a gloo
        # rehearsal two ranks share one GPU and a short kernel's events span the other rank's work
        dom = "accumulate" if "accumulate" in solo else (max(solo, key=solo.get) if solo else None)
        roofline = None
        if dom:
            ach = BYTES_PER_PAIR * n_mine / (solo[dom] * 1e-3) / 1e9   # the pairs THIS rank's launch reads
            traffic, traffic_src = None, None
            pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
            if os.path.exists(pmc) and world == 1 and args.logn == 20:
                try:
                    pj = json.load(open(pmc))
                    traffic = pj.get(dom, {}).get("hbm_bytes_per_launch")
                    traffic_src = pj.get("_measured", "profiles/pmc_traffic.json (separate rocprofv3 --pmc passes)")
                except Exception:
                    traffic = None
            # The kernel is bound by 32-bit integer multiply issue, not by HBM or MFMA, so the
            # explanatory fraction is reported next to the contract's HBM one: lane-level
            # v_mad_u64_u32 per launch -- one mixed addition per sorted entry EXCEPT the first
            # of every fragment (a copy), MADS_PER_MADD multiply-adds each -- over the kernel's
            # own duration, against 1024 SIMDs x 64 lanes x 2.4 GHz / 4.9 cycles per wave
            # instruction (profiles/r01_ubench_valu.txt).
            valu = None
            if dom == "accumulate" and counts.get("entries"):
                madds = counts["entries"] - counts["fragments"]
                peak = 1024 * 64 * 2.4e9 / 4.9
                valu = {"unit": "lane v_mad_u64_u32 /s", "mads_per_mixed_addition": MADS_PER_MADD,
                        "mixed_additions_per_launch": madds, "sorted_entries": counts["entries"],
                        "fragments": counts["fragments"], "peak": peak,
                        "achieved": madds * MADS_PER_MADD / (solo[dom] * 1e-3),
                        "frac": round(madds * MADS_PER_MADD / (solo[dom] * 1e-3) / peak, 4),
                        "peak_note": "nominal: 1024 SIMDs x 64 lanes x 2.4 GHz / 4.9 cycles; under this kernel the chip "
                                     "sustains ~2.05 GHz (s_memtime stamps, profiles/r04_wave_trace.txt), where a "
                                     "v_mad_u64_u32 is a 4-cycle instruction: the same achieved rate is ~0.90 of that"}
            roofline = {"bound": "hbm", "kernel": dom, "achieved": round(ach, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(ach / HBM_PEAK_GBS, 6), "traffic": traffic, "traffic_source": traffic_src,
                        "kernel_ms": solo[dom], "kernel_ms_alone": solo,
                        "kernel_span_ms_in_pipeline": span, "valu": valu,
                        "note": "achieved = 128 B x n / the kernel's own duration (HIP events, nothing else in flight); "
                                "integer-VALU bound (381-bit Montgomery arithmetic), see DESIGN.md.  kernel_ms_alone is "
                                "the PIPELINED plan's kernels run alone: its bucket reduce (k_reduce_segments) takes "
                                "32-bucket segments (a quarter round of lanes, a longer chain) because it runs beside "
                                "the next MSM's accumulation; the synchronous plan behind single_call_ms takes 8 "
                                "(0.26-0.30 ms; window_sum = k_reduce_level)"}
        out = {
            "metric": f"BLS12-381 G1 MSM scalar-point pairs/sec at N=2^{args.logn}",
            "value": value, "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "u32", "data": "synthetic",
            "config": {"workload": f"single G1 MSM, N=2^{args.logn} random Fr scalars x walk points, inputs resident in HBM",
                       "n_pairs": n, "window_bits": c, "num_windows": W, "in_flight": depth,
                       "single_call_ms": round(single_call_ms, 4), "host_ms_per_step": host_ms,
                       "parallelism": "single GPU" if dist is None else
                       (f"Pippenger windows split x{world}" if args.split == "windows" else f"point ranges x{world}")
                       + f", all_gather of 144 B partials over {dist.get_backend()}"},
            "roofline": roofline,
        }
        # the verification leg BEFORE the CPU baseline: sixteen saturated host threads right in
        # front of a latency measurement run into the box's CPU quota (observed: 12 % fewer
        # verifies/s)
        if world == 1 and args.logn <= 22 and not args.no_verify:
            # PCIe-inclusive: the entry point a cgo caller binds takes HOST slices (pageable memory);
            # reported beside the headline, never as `value`
            pts_h = d_pts.cpu().numpy().view(np.uint64)
            hb = []
            for _ in range(6):
                t1 = time.perf_counter()
                r_h = cm.msm_g1(pts_h, sc)
                hb.append((time.perf_counter() - t1) * 1e3)
            if not (r_h == result).all():
                ok = False
            hb_ms = float(np.median(hb[1:]))
            out["config"]["host_buffers"] = {"entry_point": "curdle_msm_g1 (pageable host slices, H2D inside the call)",
                                             "ms_per_call": round(hb_ms, 4), "pairs_per_s": round(n / hb_ms * 1e3, 1)}
            out["config"]["single_call"] = {"entry_point": "curdle_msm_g1_device (synchronous, inputs resident)",
                                            "ms_per_call": round(single_call_ms, 4),
                                            "pairs_per_s": round(n / single_call_ms * 1e3, 1)}
            del pts_h
        if resident is not None:
            if not resident.pop("ok"):
                ok = False
            out["config"]["resident_bases"] = resident
        if kept is not None:
            if not kept.pop("ok"):
                ok = False
            out["config"]["bases_unchanged"] = kept
        if multi is not None:
            out["config"]["multi_gpu"] = multi
            out["config"]["bases_unchanged_flag"] = keep_bases
        if world == 1 and not args.no_verify and args.logn == 20:
            out["verify"] = verify_leg(cm, 200, 20)
        if not args.no_cpu_baseline:
            # on rank 0 at every N (north_star: the CPU "in the same run"); the other ranks wait at the
            # barrier below -- seconds, far inside the collective's timeout
            out["cpu_baseline"] = cpu_baseline(cm, k, q, n, sc, result, d_pts)
            ok = out["cpu_baseline"]["gpu_matches_cpu"] and out["cpu_baseline"]["gpu_full_size_verified"]
            if dsweep is not None:
                sys.path.insert(0, os.path.join(ROOT, "oracle", "py"))
                import coracle as co
                pts_h = d_pts.cpu().numpy().view(np.uint64)
                native = os.path.exists(os.path.join(ROOT, "oracle", "libcurdle_cpufast_native.so"))
                threads = min(host_cores(), 256)
                rows = []
                for lg, wall, r_m in dsweep:
                    m = 1 << lg
                    t1 = time.perf_counter()
                    ref = co.msm_fast(pts_h[:m], sc[:m], threads=threads if m >= 4096 else 1, native=native)
                    dt = time.perf_counter() - t1
                    rows.append({"logn": lg, "wall_ms": round(wall, 4), "pairs_per_s": round(m / wall * 1e3, 1),
                                 "hbm_frac_whole_call": round(BYTES_PER_PAIR * m / (wall * 1e-3) / 1e9 / (HBM_PEAK_GBS * world), 6),
                                 "cpu_port_pairs_per_s": round(m / dt, 1), "gpu_matches_cpu": bool((r_m == ref).all())})
                out["sweep"] = rows
                out["sweep_what"] = (f"one synchronous msm_g1_distributed call per size on all {world} ranks (window ranges, all_gather of the "
                                     "144 B partials, sum), MAX over ranks, median of 7; hbm_frac_whole_call against world x 8 TB/s; CPU port on rank 0")
                ok = ok and all(r_["gpu_matches_cpu"] for r_ in rows)
            if world == 1 and dist is None and args.logn == 20 and not args.emulate_world:
                out["sweep"] = compact_sweep(cm, torch, d_pts, sc, d_sc)
                ok = ok and all(r_["gpu_matches_cpu"] for r_ in out["sweep"])
                out["adversarial"] = compact_adversarial(cm, torch, d_pts, k, q, sc)
                ok = ok and all(r_["ok"] for r_ in out["adversarial"]["rows"])
        if not ok:
            out["value"] = None   # a wrong result has no throughput
    # BASELINE config 5 beside the headline at every N (replicas: every rank verifies its share of
    # 1,024 Whisk shuffle proofs; collective calls, so all ranks take part)
    c5 = None
    if not args.no_verify and args.logn == 20 and not args.emulate_world:
        c5 = whisk_batch_leg(cm, torch, dist, dev, rank, world, 1024, 3, 1)
    if rank == 0:
        # One process, several GPUs through the C ABI (curdle_init_devices) -- in a CHILD process with
        # a timeout, after everything above is measured, so that nothing in it can take this line
        # down: only when a one-rank run sees more than one GPU (never on a one-GPU lease).
        if world == 1 and not args.no_verify and args.logn == 20 and torch.cuda.device_count() > 1 \
                and os.environ.get("CURDLE_BENCH_MULTI_DEVICE", "1") != "0":
            try:
                cp = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_multi_device.py")], capture_output=True,
                                    text=True, timeout=240)
                lines = [ln for ln in cp.stdout.strip().splitlines() if ln.startswith("{")]
                out["multi_device_c_abi"] = json.loads(lines[-1]) if lines else {"error": (cp.stderr or "no output")[-400:]}
            except Exception as e:      # noqa: BLE001 -- a diagnostic leg: its failure is reported, not raised
                out["multi_device_c_abi"] = {"error": repr(e)[:400]}
        if c5 is not None:
            out["config5"] = {k_: c5[k_] for k_ in ("metric", "value", "unit", "ms_per_step", "config", "accept_bits_exact")}
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if not ok:
        raise SystemExit("bench.py: the GPU result failed verification (see cpu_baseline in the line above)")


def gpu_share_of_batch(which):
    """What the host-bound batch lines say about the GPU, REPLAYED from profiles/r06_gpu_busy.json (a
    rocprofv3 kernel trace cannot be taken from inside this process; tools/exp/gpu_busy.sh made it): the share of a step
    during which at least one kernel was running, how much those kernels overlap, and how the same step's throughput
    follows the host thread count.  HIP events do not give this: around the 10 us kernels of calls whose launches the
    host issues one by one a bracket includes the host's gaps (summed brackets came to 1.5x the step)."""
    path = os.path.join(ROOT, "profiles", "r06_gpu_busy.json")
    try:
        d = json.load(open(path))[which]
    except Exception:
        return None
    return {"source": "replayed from profiles/r06_gpu_busy.json (rocprofv3 --kernel-trace of tools/bench_batch_busy.py, 5 steps of 1,024 proofs)",
            "gpu_timeline_coverage": d["gpu_timeline_coverage"], "kernel_overlap_factor": d["overlap_factor"],
            "proofs_per_s_by_host_threads": {str(r["host_threads"]): r["proofs_per_s"] for r in d["thread_scaling"]},
            "reading": "a kernel is resident on the GPU for most of the step, but they are latency-bound chains at low occupancy "
                       "(subgroup tests, slot scalars, bucket reductions of 32-proof groups) that overlap each other several "
                       "times over; the figure follows the HOST thread count up to the lease's 16 cores -- transcript hashing "
                       "(Keccak) and Fr algebra, out of scope (SURVEY.md section 2) -- so it is a host-thread figure, not a GPU one"}


def whisk_batch_leg(cm, torch, dist, dev, rank, world, k, steps, warmup):
    """BASELINE config 5: k IsValidWhiskShuffleProof verifications (whisk.go:20-61) per step, as
    replicas over the ranks -- rank r verifies proofs r, r + world, ... with ONE
    curdle_whisk_is_valid_shuffle_proof_batch call on its GPU (everything from bytes: tracker and
    proof points decoded on the GPU, host threads hashing, one device accumulation per group of
    proofs) and the accept bits are exchanged with one all_gather of ceil(k / world) bytes
    (curdlemsm.distributed.verify_replicas).  Eight distinct honest shuffles; the timed steps
    verify honest proofs (a rejected group of 32 is re-verified member by member, which is the
    price of exact bits, not the steady state), and one more untimed step with every 61st member
    a planted reject must return the exact bits on every rank.  The k proofs are fixed as the
    ranks grow: STRONG scaling of one batch."""
    from curdlemsm.distributed import replica_shard, verify_replicas
    ONE = np.array(cm_one_limbs(), dtype=np.uint64)
    compress = lambda aff: cm.g1_compress(np.concatenate([aff, ONE]))
    crs = cm.CRS(cm.WHISK_ELL, cm.Rand(0))
    sets = []
    for j in range(8):                                  # the same eight shuffles on every rank (seeded)
        r = cm.Rand(10 + j)
        pts = r.get_g1_affines(2 * cm.WHISK_ELL)
        pre = [compress(pts[2 * i]) + compress(pts[2 * i + 1]) for i in range(cm.WHISK_ELL)]
        post, proof = cm.whisk_generate_shuffle_proof(crs, pre, r)
        sets.append((pre, post, proof))
    mine = replica_shard(k, world, rank)

    def prepare(planted):
        expect = np.ones(k, dtype=np.uint8)
        pres, posts, proofs = [], [], []
        for i in mine:
            pre, post, proof = sets[i % 8]
            if planted and i % 61 == 60:                # another shuffle's post trackers
                post = sets[(i + 1) % 8][1]
            pres.append(pre), posts.append(post), proofs.append(proof)
        if planted:
            expect[60::61] = 0
        return cm.PreparedWhiskBatch(pres, posts, proofs), expect

    honest, all_ones = prepare(False)
    with_rejects, expect_rejects = prepare(True)
    # the ranks of one node share its cores (and one cgroup quota): each takes its share
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    threads = max(1, min(16, host_cores() // max(1, local_world)))
    seed = [100]

    def step(batch=None):
        batch = batch or honest

        def shard_bits(idx):
            assert len(idx) == len(mine)
            seed[0] += 1
            return np.array(batch.run(crs, cm.Rand(seed[0] * 1000 + rank), nthreads=threads), dtype=np.uint8)

        if dist is None:
            return shard_bits(mine)
        return verify_replicas(k, shard_bits, device=dev if dist.get_backend() == "nccl" else None)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    exact = True
    for _ in range(warmup):
        exact = exact and bool((step() == all_ones).all())
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        exact = exact and bool((step() == all_ones).all())
    barrier()
    elapsed = time.perf_counter() - t0
    t1 = time.perf_counter()
    exact = exact and bool((step(with_rejects) == expect_rejects).all())     # untimed: exact bits with bad members
    barrier()
    rejects_ms = (time.perf_counter() - t1) * 1e3
    if dist is not None:
        t = torch.tensor([elapsed, 0.0 if exact else 1.0], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, exact = float(t[0].item()), float(t[1].item()) == 0.0
    return {"metric": "Whisk shuffle proofs verified/sec (BASELINE config 5)", "value": k * steps / elapsed if exact else None,
            "unit": "proofs/s", "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": elapsed * 1e3 / steps,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {"workload": f"{k} IsValidWhiskShuffleProof verifications per step (ell = {cm.WHISK_ELL}, 4,576-byte proofs, "
                                   f"496 tracker points each), replicas round-robin over {world} rank(s), "
                                   f"{threads} host threads per rank, 8 distinct shuffles, all honest in the timed steps",
                       "parallelism": "single GPU" if dist is None else f"replicas x{world}, all_gather of accept bits over {dist.get_backend()}",
                       "host_threads": threads, "gpu_share": gpu_share_of_batch("whisk"),
                       "step_with_planted_rejects": {"rejects": int(k - expect_rejects.sum()), "ms": round(rejects_ms, 2)}},
            "accept_bits_exact": exact}


def sweep(cm, torch, dev, args):
    """north_star: "MSM throughput on synthetic random scalars/points at N in {2^10 .. 2^20} ... as
    absolute numbers and as fraction of HBM roofline, next to the ... CPU MultiExp timed on the
    GPU box's own host cores (core count stated) in the same run" -- one JSON object, one entry
    per size: the synchronous call's wall time on resident inputs (median of 9) and from host
    slices, every kernel's own duration (HIP events), the dominant kernel with its HBM fraction
    (128 B x N / its duration / 8 TB/s) and, for the accumulation, the multiply-issue fraction,
    and the CPU port (oracle/cpu_msm_fast.c, all granted cores) on the SAME inputs with its
    result compared bit for bit."""
    sys.path.insert(0, os.path.join(ROOT, "oracle", "py"))
    import coracle as co
    cores = host_cores()
    native = False
    try:
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "native"], check=True, capture_output=True, timeout=120)
        native = True
    except Exception:
        native = False
    r1 = cm.Rand(1)
    k = limbs_to_int(r1.get_fr()) * R_INV % R_MOD
    q = limbs_to_int(r1.get_fr()) * R_INV % R_MOD
    nmax = 1 << 20
    d_all = torch.empty((nmax, 12), dtype=torch.int64, device=dev)
    cm.synth_points_walk_device(k, q, nmax, d_all.data_ptr())
    pts_all = d_all.cpu().numpy().view(np.uint64)
    sc_all = uniform_scalars(np.random.default_rng(2), nmax)
    d_sc_all = torch.from_numpy(sc_all.view(np.int64)).to(dev)
    torch.cuda.synchronize()
    peak_mads = 1024 * 64 * 2.4e9 / 4.9
    rows, all_ok = [], True
    sizes = [8, 32, 64, 128, 256, 512] + [1 << e for e in range(10, 21)]   # the protocol's small MSMs, then north_star's sweep
    if args.sweep_sizes:
        sizes = [int(x) for x in args.sweep_sizes.split(",")]
    for n in sizes:
        logn = int(np.log2(n))
        pp, sp = d_all.data_ptr(), d_sc_all.data_ptr()
        cm.profile_enable(0)
        for _ in range(3):
            res = cm.msm_g1_device(pp, sp, n)
        lat = []
        for _ in range(9):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            cm.msm_g1_device(pp, sp, n)
            lat.append((time.perf_counter() - t1) * 1e3)
        hb = []
        for _ in range(5):
            t1 = time.perf_counter()
            r_h = cm.msm_g1(pts_all[:n], sc_all[:n])
            hb.append((time.perf_counter() - t1) * 1e3)
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
        ks = {a: round(float(np.mean(b)), 4) for a, b in ks.items()}
        dom = max(ks, key=ks.get)
        threads = min(cores, 256)
        co.msm_fast(pts_all[:min(n, 4096)], sc_all[:min(n, 4096)], threads=threads, native=native)
        best, best_threads = None, threads
        for th in ([threads] if n >= 4096 else [1, threads]):   # a small MSM is faster on one core than fanned out
            for _ in range(3):
                t1 = time.perf_counter()
                ref = co.msm_fast(pts_all[:n], sc_all[:n], threads=th, native=native)
                dt = time.perf_counter() - t1
                if best is None or dt < best:
                    best, best_threads = dt, th
        same = bool((res == ref).all()) and bool((r_h == ref).all())
        all_ok = all_ok and same
        wall = float(np.median(lat))
        row = {"logn": logn if n >= 1024 else None, "n_pairs": n, "window_bits": cm.window_bits(n), "num_windows": cm.num_windows(n, 0),
               "wall_ms": round(wall, 4), "pairs_per_s": round(n / wall * 1e3, 1),
               "host_buffers_ms": round(float(np.median(hb[1:])), 4),
               "kernel_ms_alone": ks, "dominant_kernel": dom,
               "hbm_frac_dominant": round(BYTES_PER_PAIR * n / (ks[dom] * 1e-3) / 1e9 / HBM_PEAK_GBS, 6),
               "hbm_frac_whole_call": round(BYTES_PER_PAIR * n / (wall * 1e-3) / 1e9 / HBM_PEAK_GBS, 6),
               "valu_frac_accumulate": (round((counts["entries"] - counts["fragments"]) * MADS_PER_MADD
                                              / (ks["accumulate"] * 1e-3) / peak_mads, 4)
                                        if counts.get("entries") and "accumulate" in ks else None),
               "cpu_port_ms": round(best * 1e3, 4), "cpu_port_pairs_per_s": round(n / best, 1), "cpu_cores": best_threads,
               "gpu_over_cpu": round(best * 1e3 / wall, 2), "gpu_matches_cpu": same}
        rows.append(row)
        print(json.dumps(row), file=sys.stderr, flush=True)
    # where a Go caller should route a MultiExp to gnark's CPU path instead: the smallest size from
    # which the GPU's synchronous call from HOST slices beats the CPU port for good
    cross = None
    for r in reversed(rows):
        if r["host_buffers_ms"] < r["cpu_port_ms"]:
            cross = r["n_pairs"]
        else:
            break
    print(json.dumps({"metric": "BLS12-381 G1 MSM size sweep N=2^10..2^20 (north_star), with the protocol's small sizes in front", "n_gpus": 1, "unit": "pairs/s",
                      "dtype": "u32", "data": "synthetic", "hbm_peak_GBs": HBM_PEAK_GBS, "bytes_per_pair": BYTES_PER_PAIR,
                      "cpu_baseline": {"kind": "port", "cores": min(cores, 256),
                                       "what": f"oracle/cpu_msm_fast.c ({'-march=native' if native else 'portable'}), best of 3 per size"},
                      "gpu_beats_cpu_from_host_slices_from_n": cross, "all_results_match_cpu": all_ok, "sweep": rows}), flush=True)
    if not all_ok:
        raise SystemExit("bench.py --sweep: a GPU result differs from the CPU port's")


def sweep_distributed(cm, torch, dist, dev, rank, world, args):
    """The size sweep on N ranks (north_star: "at 1/2/4/8 GPUs"): every size through
    curdlemsm.distributed.msm_g1_distributed(split="auto") on ALL ranks -- window ranges up to 2^21
    pairs, point ranges beyond -- the wall time of one synchronous distributed call (barrier, call,
    MAX over the ranks; median of 9), and on rank 0 the CPU port on the same inputs with the result
    compared bit for bit.  Rank 0 prints one JSON object with n_gpus = world."""
    from curdlemsm.distributed import choose_split, msm_g1_distributed
    sys.path.insert(0, os.path.join(ROOT, "oracle", "py"))
    import coracle as co
    cores = host_cores()
    native = False
    if rank == 0:
        try:
            subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "native"], check=True, capture_output=True, timeout=120)
            native = True
        except Exception:
            native = False
    on_gpu = dist.get_backend() == "nccl"
    r1 = cm.Rand(1)
    k = limbs_to_int(r1.get_fr()) * R_INV % R_MOD
    q = limbs_to_int(r1.get_fr()) * R_INV % R_MOD
    sizes = [1 << e for e in range(10, 21)]
    if args.sweep_sizes:
        sizes = [int(x) for x in args.sweep_sizes.split(",")]
    nmax = max(sizes)
    d_all = torch.empty((nmax, 12), dtype=torch.int64, device=dev)
    cm.synth_points_walk_device(k, q, nmax, d_all.data_ptr())
    sc_all = uniform_scalars(np.random.default_rng(2), nmax)     # the same seed on every rank: replicated inputs
    d_sc_all = torch.from_numpy(sc_all.view(np.int64)).to(dev)
    pts_all = d_all.cpu().numpy().view(np.uint64) if rank == 0 else None
    torch.cuda.synchronize()
    rows, all_ok = [], True
    for n in sizes:
        pp, sp = d_all.data_ptr(), d_sc_all.data_ptr()
        call = lambda: msm_g1_distributed(pp, sp, n, device=dev if on_gpu else None, split="auto")
        for _ in range(3):
            res = call()
        lat = []
        for _ in range(9):
            dist.barrier()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            res = call()
            t = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device=dev if on_gpu else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            lat.append(float(t.item()) * 1e3)
        wall = float(np.median(lat))
        row = {"n_pairs": n, "logn": int(np.log2(n)), "split": choose_split(n, world), "window_bits": cm.window_bits(n),
               "num_windows": cm.num_windows(n, 0), "wall_ms": round(wall, 4), "pairs_per_s": round(n / wall * 1e3, 1),
               "hbm_frac_whole_call": round(BYTES_PER_PAIR * n / (wall * 1e-3) / 1e9 / (HBM_PEAK_GBS * world), 6)}
        if rank == 0:
            threads = min(cores, 256)
            best, best_threads = None, threads
            for th in ([threads] if n >= 4096 else [1, threads]):
                for _ in range(3):
                    t1 = time.perf_counter()
                    ref = co.msm_fast(pts_all[:n], sc_all[:n], threads=th, native=native)
                    dt = time.perf_counter() - t1
                    if best is None or dt < best:
                        best, best_threads = dt, th
            same = bool((res == ref).all())
            all_ok = all_ok and same
            row.update({"cpu_port_ms": round(best * 1e3, 4), "cpu_port_pairs_per_s": round(n / best, 1), "cpu_cores": best_threads,
                        "gpu_over_cpu": round(best * 1e3 / wall, 2), "gpu_matches_cpu": same})
            print(json.dumps(row), file=sys.stderr, flush=True)
        rows.append(row)
    flag = torch.tensor([0.0 if all_ok else 1.0], dtype=torch.float64, device=dev if on_gpu else "cpu")
    dist.all_reduce(flag, op=dist.ReduceOp.MAX)
    all_ok = float(flag.item()) == 0.0
    if rank == 0:
        print(json.dumps({"metric": "BLS12-381 G1 MSM size sweep N=2^10..2^20 (north_star), distributed", "n_gpus": world, "unit": "pairs/s",
                          "dtype": "u32", "data": "synthetic", "hbm_peak_GBs": HBM_PEAK_GBS, "bytes_per_pair": BYTES_PER_PAIR,
                          "backend": dist.get_backend(),
                          "what": "one synchronous msm_g1_distributed(split='auto') per size on every rank: window ranges "
                                  "(all_gather of 144 B partials) up to 2^21 pairs, point ranges beyond; barrier, call, MAX over ranks",
                          "cpu_baseline": {"kind": "port", "cores": min(cores, 256),
                                           "what": f"oracle/cpu_msm_fast.c ({'-march=native' if native else 'portable'}), best of 3 per size, rank 0"},
                          "all_results_match_cpu": all_ok, "sweep": rows}), flush=True)
    if not all_ok:
        raise SystemExit("bench.py --sweep: a distributed GPU result differs from the CPU port's")


def compact_sweep(cm, torch, d_pts, sc, d_sc):
    """north_star's size table inside the DEFAULT line (the table existed only as a
    builder-run artifact): N = 2^10, 2^12, .., 2^20 on prefixes of the headline's resident inputs -- the wall
    time of one synchronous curdle_msm_g1_device call (median of 7, no phase events), pairs/s, the dominant
    kernel's HBM fraction (128 B x N / the accumulation's own duration / 8 TB/s) and multiply-issue fraction
    from a separate profiled pass, and the CPU port on the same inputs (all granted cores; one run below 2^18,
    best of two from there), its result compared bit for bit.  About two seconds in all."""
    sys.path.insert(0, os.path.join(ROOT, "oracle", "py"))
    import coracle as co
    cores = host_cores()
    threads = min(cores, 256)
    native = os.path.exists(os.path.join(ROOT, "oracle", "libcurdle_cpufast_native.so"))   # cpu_baseline() built it, if it can be built here
    pts = d_pts.cpu().numpy().view(np.uint64)
    peak_mads = 1024 * 64 * 2.4e9 / 4.9
    pp, sp = d_pts.data_ptr(), d_sc.data_ptr()
    # the GPU columns of all sizes first, behind a quarter second of warm-up (the CPU port between two sizes would leave the
    # GPU idle for up to half a second, and a GPU that has idled runs its next calls at lower clocks), then the CPU column
    cm.profile_enable(0)
    tw = time.perf_counter()
    while time.perf_counter() - tw < 0.25:
        cm.msm_g1_device(pp, sp, 1 << 18)
    gpu = {}
    for logn in range(20, 9, -2):
        n = 1 << logn
        cm.profile_enable(0)
        for _ in range(3):
            res = cm.msm_g1_device(pp, sp, n)
        lat = []
        for _ in range(7):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            cm.msm_g1_device(pp, sp, n)
            lat.append((time.perf_counter() - t1) * 1e3)
        cm.profile_enable(1)
        acc, counts = [], {}
        for _ in range(3):
            cm.msm_g1_device(pp, sp, n)
            pr = cm.profile_last()
            acc.append(pr["kernels"].get("accumulate", 0.0))
            counts = {"entries": pr["entries"], "fragments": pr["fragments"]}
        cm.profile_enable(0)
        gpu[logn] = (res, float(np.median(lat)), float(np.mean(acc)), counts)
    rows = []
    for logn in range(10, 21, 2):
        n = 1 << logn
        res, wall, acc_ms, counts = gpu[logn]
        best = None
        for _ in range(1 if logn < 18 else 2):
            t1 = time.perf_counter()
            ref = co.msm_fast(pts[:n], sc[:n], threads=threads if n >= 4096 else 1, native=native)
            dt = time.perf_counter() - t1
            best = dt if best is None else min(best, dt)
        rows.append({"logn": logn, "wall_ms": round(wall, 4), "pairs_per_s": round(n / wall * 1e3, 1),
                     "hbm_frac": round(BYTES_PER_PAIR * n / (acc_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 6) if acc_ms else None,
                     "valu_frac": (round((counts["entries"] - counts["fragments"]) * MADS_PER_MADD / (acc_ms * 1e-3) / peak_mads, 4)
                                   if acc_ms and counts.get("entries") else None),
                     "cpu_port_pairs_per_s": round(n / best, 1), "gpu_matches_cpu": bool((res == ref).all())})
    return rows


def compact_adversarial(cm, torch, d_pts, k, q, uniform):
    """"Adversarial sets also timed: all-equal scalars (a11), <= 9-bit scalars (a13), 1 % infinity
    bases" -- and two that aim at the bucket sort (64 distinct values; one hot window: every term in ONE bucket of one
    window, built through the GLV halves, below 2^20 only: its construction is a big-integer product per term) -- at
    N = 2^12, 2^16, 2^20 on prefixes of the headline's points, through the synchronous resident call and through
    curdle_msm_g1 from pageable host slices; every result against the closed form (k S0 + q S1) G, every time as a ratio
    to the uniform control at the same N and entry point, and the kernel whose own duration grew most beside uniform's.
    The reference produces such inputs at samepermutationargument.go:67,132-140 (all scalars = beta), common/util.go:68-75
    (scalars = perm(i) < ell) and curdleproof.go:281,285 (zero points).  tools/sweep_adversarial.py is the full table
    (profiles/r06_adversarial.json); the bar is 1.25x of uniform."""
    sys.path.insert(0, os.path.join(ROOT, "oracle", "py"))
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import coracle as co
    import adversarial_inputs as adv
    one = np.array(cm_one_limbs(), dtype=np.uint64)
    rows, worst = [], 0.0
    for logn in (12, 16, 20):
        n = 1 << logn
        base, base_k = {}, {}
        for fam in ("uniform", "all_equal", "small_9bit", "infinity_1pct", "distinct_64", "hot_window"):
            if fam == "hot_window" and logn > 16:
                continue
            fsc, dead = adv.make_family(fam, n, uniform, window_bits=cm.window_bits(n))
            fsc = np.ascontiguousarray(fsc)
            pts_d = d_pts[:n]
            if dead is not None:
                pts_d = pts_d.clone()
                pts_d[torch.from_numpy(np.asarray(dead, dtype=np.int64)).to(pts_d.device)] = 0
            exp = co.jac_normalise(np.concatenate([co.scalar_mul_gen(adv.walk_exponent(k, q, fsc, dead)), one]))
            d_f = torch.from_numpy(fsc.view(np.int64)).to(d_pts.device)
            pp, sp = pts_d.data_ptr(), d_f.data_ptr()
            cm.profile_enable(0)
            for _ in range(2):
                res = cm.msm_g1_device(pp, sp, n)
            lat = []
            for _ in range(7):
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                cm.msm_g1_device(pp, sp, n)
                lat.append((time.perf_counter() - t1) * 1e3)
            pts_h = pts_d.cpu().numpy().view(np.uint64)
            hb = []
            for _ in range(4):
                t1 = time.perf_counter()
                r_h = cm.msm_g1(pts_h, fsc)
                hb.append((time.perf_counter() - t1) * 1e3)
            cm.profile_enable(1)
            ks = {}
            for _ in range(3):
                cm.msm_g1_device(pp, sp, n)
                for name, ms in cm.profile_last()["kernels"].items():
                    if not name.startswith("("):
                        ks.setdefault(name, []).append(ms)
            cm.profile_enable(0)
            ks = {a: float(np.mean(b)) for a, b in ks.items()}
            row = {"family": fam, "logn": logn, "sync_ms": round(float(np.median(lat)), 4),
                   "host_slices_ms": round(float(np.median(hb[1:])), 4),
                   "ok": bool((res == exp).all() and (r_h == exp).all())}
            if fam == "uniform":
                base, base_k = dict(row), ks
            else:
                row["ratio_sync"] = round(row["sync_ms"] / base["sync_ms"], 3)
                row["ratio_host_slices"] = round(row["host_slices_ms"] / base["host_slices_ms"], 3)
                worst = max(worst, row["ratio_sync"], row["ratio_host_slices"])
                grown = max(ks, key=lambda a: ks[a] - base_k.get(a, 0.0))
                row["grew_most"] = {"kernel": grown, "ms": round(ks[grown], 4), "uniform_ms": round(base_k.get(grown, 0.0), 4)}
            rows.append(row)
    return {"bar": "every family within 1.25x of uniform at the same N and entry point", "worst_ratio": worst,
            "note": "2^12: a bucket that holds a whole window's terms is ~2,000 fragments at 4 positions per lane, and summing "
                    "them (k_merge_large: a chain of ~14 dependent additions) shows on a 0.3 ms call; from 2^16 on these families "
                    "are inside the bar (a few hundred distinct values at 2^14..2^16 are not: 1.3-1.38x, "
                    "profiles/r06_adversarial.json)", "rows": rows}


def cpu_baseline(cm, k, q, n, sc, gpu_result, d_pts):
    """The multi-threaded CPU port (oracle/cpu_msm_fast.c: mulx/adx field products, signed
    digits, XYZZ buckets, windows split over all host cores -- the bucket method the reference
    gets from gnark-crypto, which cannot run here: no Go toolchain) timed on this box's host
    cores on the FULL workload, its result compared bit for bit with the GPU's, and the GPU
    result also checked against the closed form (k sum s_i + q sum i s_i) G."""
    sys.path.insert(0, os.path.join(ROOT, "oracle", "py"))
    import coracle as co
    cores = host_cores()
    threads = min(cores, 256)
    native = False
    try:   # tuned for THIS machine (the prebuilt portable library is the fallback)
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "native"], check=True, capture_output=True, timeout=120)
        native = True
    except Exception:
        native = False
    pts = d_pts.cpu().numpy().view(np.uint64)
    co.msm_fast(pts[:4096], sc[:4096], threads=threads, native=native)   # page the library in
    best, ref = None, None
    for _ in range(3):
        t0 = time.perf_counter()
        ref = co.msm_fast(pts, sc, threads=threads, native=native)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    same = bool((gpu_result == ref).all())
    # full-size result vs closed form
    s_int = [limbs_to_int(row) * R_INV % R_MOD for row in sc]
    e = (k * (sum(s_int) % R_MOD) + q * (sum(i * v for i, v in enumerate(s_int)) % R_MOD)) % R_MOD
    aff = co.scalar_mul_gen(e)
    exp = co.jac_normalise(np.concatenate([aff, np.array(cm_one_limbs(), dtype=np.uint64)]))
    full_ok = bool((gpu_result == exp).all())
    return {"value": n / best, "unit": "pairs/s", "cores": threads, "kind": "port",
            "sample": f"the full MSM of 2^{int(np.log2(n))} pairs (same inputs), best of 3: {best:.3f} s, "
                      f"oracle/cpu_msm_fast.c ({'-march=native' if native else 'portable BMI2+ADX build'}), {threads} threads",
            "note": "gnark-crypto (the reference's MultiExp) cannot be run here: no Go toolchain on this box; this port "
                    "has gnark's field multiplier, signed digits, extended-Jacobian buckets and window splitting, "
                    "not its batch-affine additions",
            "gpu_matches_cpu": same, "gpu_full_size_verified": full_ok}


def verify_leg(cm, reps, warmup):
    """Second half of BASELINE.json's metric: curdleproof.Verify of a DECODED proof at
    ell = 252 (what the reference's BenchmarkVerifier times, curdleproof_test.go:210-237),
    one call after the other from one thread, every MSM on the GPU."""
    ell = VERIFY_ELL
    rand = cm.Rand(0)
    crs = cm.CRS(ell, rand)
    perm = cm.Rand(42).generate_permutation(ell)
    kk = rand.get_fr()
    Rs, Ss = rand.get_g1_affines(ell), rand.get_g1_affines(ell)
    Ts, Us, M, rs_m = cm.shuffle_permute_commit(crs, Rs, Ss, perm, kk, rand)
    proof_bytes = cm.prove(crs, Rs, Ss, Ts, Us, M, perm, kk, rs_m, cm.Rand(42))
    proof = cm.Proof(proof_bytes)
    rands = [cm.Rand(1000 + i) for i in range(reps + warmup)]
    # the instance marshalled once: the timed call is curdle_verify_proof and nothing else (the reference's benchmark
    # times Verify on values already in memory; numpy -> ctypes conversions were ~15 us of every call until round 5)
    call = cm.PreparedVerify(crs, proof, Rs, Ss, Ts, Us, M)
    for i in range(warmup):
        if not call.run(rands[i]):
            raise SystemExit("bench.py: an honest proof was rejected")
    # A process' first ~250 verifications run 20 % slower than the ones after them (1.04 against 0.85 ms:
    # tools/verify_settle_probe.py -- a GPU that has only seen 0.25 ms bursts has not left its idle clocks), which
    # is what `--mode verify` alone used to time: keep warming until half a second of them has gone by.
    tw = time.perf_counter()
    extra = 0
    while time.perf_counter() - tw < 0.5 and extra < 2000:
        if not call.run(cm.Rand(5000 + extra)):
            raise SystemExit("bench.py: an honest proof was rejected")
        extra += 1
    t0 = time.perf_counter()
    for i in range(reps):
        if not call.run(rands[warmup + i]):
            raise SystemExit("bench.py: an honest proof was rejected")
    dt = (time.perf_counter() - t0) / reps
    rejects = not cm.verify_proof(crs, proof, Ss, Rs, Ts, Us, M, cm.Rand(5))
    # throughput form of the same work: 1,024 proofs from bytes in one curdle_verify_batch call
    # (points decoded ahead in chunks, host threads verifying, one device accumulation per 32
    # proofs).  Reported beside the headline, which stays the one-after-the-other figure.
    kb, threads = 1024, min(16, host_cores())
    distinct = [(proof_bytes, Rs, Ss, Ts, Us, M)]
    for j in range(7):                                  # eight distinct (proof, instance) pairs, not one repeated
        pj = cm.Rand(50 + j).generate_permutation(ell)
        kj = rand.get_fr()
        Rj, Sj = rand.get_g1_affines(ell), rand.get_g1_affines(ell)
        Tj, Uj, Mj, rsj = cm.shuffle_permute_commit(crs, Rj, Sj, pj, kj, rand)
        distinct.append((cm.prove(crs, Rj, Sj, Tj, Uj, Mj, pj, kj, rsj, cm.Rand(60 + j)), Rj, Sj, Tj, Uj, Mj))
    cols = [[distinct[i % 8][c] for i in range(kb)] for c in range(6)]
    batch = cm.PreparedVerifyBatch(*cols)
    if not all(batch.run(crs, cm.Rand(7), nthreads=threads)):
        raise SystemExit("bench.py: an honest proof was rejected by the batch verifier")
    tb = []
    for rep in range(3):
        t0 = time.perf_counter()
        ok_bits = batch.run(crs, cm.Rand(8 + rep), nthreads=threads)
        tb.append(time.perf_counter() - t0)
        if not all(ok_bits):
            raise SystemExit("bench.py: an honest proof was rejected by the batch verifier")
    # the one MSM behind a verification (5 ell + 8 CRS / instance bases + the proof's points), alone
    import torch
    nb = 5 * ell + 8 + 100
    d_p = torch.empty((nb, 12), dtype=torch.int64, device="cuda")
    cm.synth_points_walk_device(12345, 6789, nb, d_p.data_ptr())
    d_s = torch.from_numpy(uniform_scalars(np.random.default_rng(7), nb).view(np.int64)).to("cuda")
    cm.profile_enable(1)
    ks = {}
    for _ in range(5):
        cm.msm_g1_device(d_p.data_ptr(), d_s.data_ptr(), nb)
        for name, ms in cm.profile_last()["kernels"].items():
            if not name.startswith("("):
                ks.setdefault(name, []).append(ms)
    cm.profile_enable(0)
    ks = {a: round(float(np.mean(b)), 4) for a, b in ks.items()}
    dom = max(ks, key=ks.get)
    abytes = VERIFY_PAIRS * BYTES_PER_PAIR
    ach = abytes / (ks[dom] * 1e-3) / 1e9
    return {"metric": "Curdleproofs verifies/sec N=252", "value": 1.0 / dt, "unit": "verifies/s",
            "ms_per_verify": dt * 1e3, "verifies_timed": reps,
            "workload": f"curdleproof.Verify of a decoded proof, ell={ell} (n=256), one thread, sequential; "
                        "accumulator on the device, one MSM per verification",
            "proof_bytes": len(proof_bytes), "rejects_swapped_instance": bool(rejects),
            "batch": {"value": kb / min(tb), "unit": "verifies/s", "proofs": kb, "host_threads": threads,
                      "ms_per_batch": [round(t * 1e3, 2) for t in tb], "gpu_share": gpu_share_of_batch("verify"),
                      "workload": "curdle_verify_batch: 1,024 proofs (8 distinct) from bytes in one call (best of 3)"},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(ach, 4), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(ach / HBM_PEAK_GBS, 8), "traffic": None,
                         "algorithmic_bytes_per_verify": abytes, "kernel_ms_alone": ks,
                         "note": "latency-bound: a 1,368-pair MSM is a chain of dependent point additions, not a stream"},
            "published_reference": {"value": REFERENCE_README_VERIFIES_PER_S, "unit": "verifies/s",
                                    "where": "reference README.md:25 (BenchmarkVerifier shuffled_elements=252, "
                                             "Ryzen 3800XT, 16 threads) -- other hardware, context only"}}


def cm_one_limbs():
    # Montgomery one of Fp (Z = 1) as six uint64 limbs
    one = (1 << 384) % 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB
    return [(one >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(6)]


if __name__ == "__main__":
    main()
