"""Multi-GPU MSM: one process per GPU, Pippenger windows split across ranks.

north_star: "partitioned across the 8 GPUs of one node by splitting Pippenger
windows with an RCCL reduce of 8 partial G1 points over xGMI".  Every rank
holds all n pairs and computes the partial sum over its window range
(curdle_msm_g1_device_windows); the 144-byte partials are exchanged with one
all_gather (RCCL has no user-defined reduction and G1 addition is not an
element-wise sum, so "reduce" = gather + 7 additions) and
summed on every rank by curdle_g1_sum.  The exchange is latency-only
(world_size x 144 B); xGMI bandwidth is irrelevant to it.

torch.distributed is plumbing here: the process group and the all_gather.
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import numpy as np

from . import g1_sum, msm_g1_device, num_windows, window_bits


def window_partition(n_windows: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous, as-even-as-possible split of [0, n_windows) over the ranks.
    Ranks beyond n_windows get an empty range."""
    base, extra = divmod(n_windows, world_size)
    begin = rank * base + min(rank, extra)
    end = begin + base + (1 if rank < extra else 0)
    return begin, end


def point_partition(n: int, world_size: int, rank: int) -> Tuple[int, int]:
    """The reserve partition of SURVEY.md section 8e: contiguous point ranges [begin, end) --
    every rank runs ALL windows over n / world_size pairs (its own window width), so the
    per-rank recoding, sort and conversion shrink with the rank count too."""
    return window_partition(n, world_size, rank)


class PartialExchange:
    """The all_gather of one uint64[18] Jacobian point per rank, in two halves: start() queues
    it and returns at once, finish() hands back uint64[world, 18].  A pipelined caller starts
    the exchange of step i and finishes it while step i + 1 is on the GPU -- the RCCL kernel has
    to find a free wave slot beside a bucket accumulation that fills the chip, which can take as
    long as a whole 8-way rank step; started and finished in one go it would put that wait into
    every step.  One [world, 18] result tensor and one copy back (not one per rank)."""

    def __init__(self, group=None, device=None):
        import torch
        import torch.distributed as dist
        self.group, self.device = group, device
        self.world = dist.get_world_size(group)
        self._torch, self._dist = torch, dist

    def start(self, partial: np.ndarray):
        torch, dist = self._torch, self._dist
        # torch has no uint64 collectives on every backend; int64 carries the same bits
        t = torch.from_numpy(np.ascontiguousarray(partial, dtype=np.uint64).view(np.int64).copy())
        if self.device is not None:
            t = t.pin_memory().to(self.device, non_blocking=True)
        out = torch.empty((self.world, t.numel()), dtype=t.dtype, device=t.device)
        work = dist.all_gather(list(out.unbind(0)), t, group=self.group, async_op=True)
        return work, out, t        # t kept alive until the collective has run

    def finish(self, handle) -> np.ndarray:
        work, out, _ = handle
        work.wait()
        return out.cpu().numpy().view(np.uint64)


def gather_partials(partial: np.ndarray, group=None, device=None) -> np.ndarray:
    """all_gather of one uint64[18] Jacobian point per rank -> uint64[world, 18]."""
    ex = PartialExchange(group=group, device=device)
    return ex.finish(ex.start(partial))


def choose_split(n: int, world_size: int) -> str:
    """Which partition pays at which size (per-rank step times measured on one MI355X with
    bench.py --emulate-world, profiles/r02_multi_gpu_emulation.jsonl): with
    a window range every rank still converts and recodes all n pairs (two records per point since
    the GLV split), with a point range it runs its own, narrower Pippenger (more windows per
    pair); the window range is ahead up to N = 2^21 (8 ranks: 0.56 against 0.67 ms at 2^20, 0.94
    against 0.95 at 2^21), the point range from 2^22 on (1.53 against 1.86 ms; N = 2^24: 5.2 ms)."""
    return "points" if n >= (1 << 22) and world_size > 1 else "windows"


def msm_g1_distributed(d_points: int, d_scalars: int, n: int, group=None, device=None, stream: int = 0,
                       c: int = 0, split: str = "windows",
                       partial_fn: Optional[Callable[[int, int, int], np.ndarray]] = None, flags: int = 0) -> np.ndarray:
    """Full MSM result (uint64[18]) on every rank.

    flags: those of curdle_msm_g1_device_windows_ex -- MSM_BASES_UNCHANGED when the ranks' resident bases stay put
    between calls (every rank otherwise converts ALL n points again for its few windows), MSM_ANY_CURVE_POINT for
    gnark's any-point contract (the window range is then taken over the 255-bit plan's windows).

    split = "windows": every rank runs its window range over all n pairs (north_star).
    split = "points":  every rank runs all windows over its point range.
    split = "auto":    choose_split(n, world).

    partial_fn(c, begin, end) -> uint64[18] replaces the GPU partial (begin / end are the
    window range or the point range); it exists so the collective plumbing can be exercised
    by the world_size-2 gloo tests on a machine without a GPU.  The product default is the
    HIP path.
    """
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if split == "auto":
        split = choose_split(n, world)
    if split == "points":
        begin, end = point_partition(n, world, rank)
        if partial_fn is not None:
            partial = partial_fn(0, begin, end)
        else:
            partial = msm_g1_device(d_points + 96 * begin, d_scalars + 32 * begin, end - begin, stream=stream,
                                    flags=flags & ~2)      # a point range is another base array on every rank: nothing to keep
    else:
        c = c or window_bits(n)
        W = num_windows(n, c, flags)           # ceil(255 / c) windows with MSM_ANY_CURVE_POINT (review of round 5)
        begin, end = window_partition(W, world, rank)
        if partial_fn is not None:
            partial = partial_fn(c, begin, end)
        else:
            partial = msm_g1_device(d_points, d_scalars, n, stream=stream, window_bits=c, win_begin=begin, win_end=end,
                                    flags=flags)
    allp = gather_partials(np.ascontiguousarray(partial, dtype=np.uint64), group=group, device=device)
    return g1_sum(allp)


def replica_shard(k: int, world_size: int, rank: int) -> np.ndarray:
    """BASELINE config 5 (many independent verifications) is replicas, not a partitioned
    kernel: rank r takes items r, r + world, r + 2 world, ...."""
    return np.arange(rank, k, world_size)


def verify_replicas(k: int, verify_shard: Callable[[np.ndarray], np.ndarray], group=None, device=None) -> np.ndarray:
    """Accept bits of k independent verifications, computed round-robin by the ranks of the
    group -- rank r calls verify_shard(indices) for its shard (e.g. a
    curdle_whisk_is_valid_shuffle_proof_batch over those proofs on its GPU) -- and exchanged
    with ONE all_gather of ceil(k / world) bytes per rank; every rank returns all k bits.
    There is no data-path collective: the verifications do not depend on each other."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    mine = replica_shard(k, world, rank)
    bits = np.asarray(verify_shard(mine), dtype=np.uint8)
    if bits.shape != mine.shape:
        raise ValueError("verify_shard must return one bit per index of its shard")
    per = -(-k // world)
    buf = np.full(per, 2, dtype=np.uint8)          # 2 = padding
    buf[:len(bits)] = bits
    t = torch.from_numpy(buf)
    if device is not None:
        t = t.to(device)
    out = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(out, t, group=group)
    res = np.zeros(k, dtype=np.uint8)
    for r, o in enumerate(out):
        idx = replica_shard(k, world, r)
        res[idx] = o.cpu().numpy()[:len(idx)]
    return res
