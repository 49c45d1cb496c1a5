"""The N > 1 path on CPU: two processes, gloo backend, the same window split,
all_gather and g1_sum the GPU ranks use (curdlemsm.distributed).  There is no GPU
in this container, so each rank's partial MSM comes from the oracle (the partial
over windows [a, b) is the MSM with every scalar replaced by the value of its
signed digits in that range); on the GPU box the partials come from the HIP path
(tests/test_msm_gpu.py::test_window_partials_sum_to_full_msm)."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import PKG, ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def partial_scalars(scalars, widths, begin, end, R):
    """Value of digits [begin, end) of the library's recoding (what a rank covers): both halves
    of the scalar's split, the second one times lambda."""
    from test_abi import GLV_LAMBDA, glv_split, recode
    out = []
    for s in scalars:
        k1, k2 = glv_split(s, R)
        v1 = sum(d << sh for d, sh in recode(k1, widths)[begin:end])
        v2 = sum(d << sh for d, sh in recode(k2, widths)[begin:end])
        out.append((v1 + v2 * GLV_LAMBDA) % R)
    return out


def _worker(rank, world, port, n, result_dir):
    sys.path[:0] = [os.path.join(ROOT, "oracle", "py"), PKG, os.path.join(ROOT, "tests")]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    import bls12381_ref as o
    import coracle as co
    import curdlemsm as cm
    from curdlemsm.distributed import msm_g1_distributed
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        k, q = o.Rand(1).get_frs(2)
        pts = co.points_walk(k, q, n)
        sc_int = o.Rand(2).get_frs(n)
        sc = np.array([o.fr_to_mont_limbs(s) for s in sc_int], dtype=np.uint64)

        def partial_fn(c, begin, end):
            ps = partial_scalars(sc_int, cm.window_widths(n, c), begin, end, o.R)
            return co.msm_pippenger(pts, np.array([o.fr_to_mont_limbs(s) for s in ps], dtype=np.uint64), threads=2)

        res = {}
        for c in (16, 15, 7):
            res[c] = msm_g1_distributed(0, 0, n, c=c, partial_fn=partial_fn)
        # the reserve partition: point ranges, all windows per rank
        res["points"] = msm_g1_distributed(
            0, 0, n, split="points", partial_fn=lambda c, lo, hi: co.msm_pippenger(pts[lo:hi], sc[lo:hi], threads=2))
        # gnark's any-point contract: the plan recodes the whole 255-bit scalar, so the ranks partition ceil(255 / c)
        # windows, not ceil(127 / c) (review of round 5: the low half of every scalar only, with no error)
        from test_abi import recode

        def partial_any(c, begin, end):
            widths = cm.window_widths(n, c, flags=cm.MSM_ANY_CURVE_POINT)
            assert sum(widths) == 255
            ps = [sum(d << sh for d, sh in recode(s, widths)[begin:end]) % o.R for s in sc_int]
            return co.msm_pippenger(pts, np.array([o.fr_to_mont_limbs(s) for s in ps], dtype=np.uint64), threads=2)
        res["any"] = msm_g1_distributed(0, 0, n, c=9, partial_fn=partial_any, flags=cm.MSM_ANY_CURVE_POINT)
        full = co.msm_pippenger(pts, sc, threads=2)
        ok = all((res[c] == full).all() for c in res)
        # config 5 replicas: 11 independent "verifications", round-robin, one all_gather of bits
        from curdlemsm.distributed import replica_shard, verify_replicas
        truth = np.array([(i * 7 + 3) % 5 != 0 for i in range(11)], dtype=np.uint8)
        seen = []

        def verify_shard(idx):
            seen.extend(int(i) for i in idx)
            return truth[idx]
        bits = verify_replicas(11, verify_shard)
        ok = ok and (bits == truth).all() and seen == [int(i) for i in replica_shard(11, world, rank)]
        ok = ok and len(verify_replicas(0, lambda idx: np.zeros(0, dtype=np.uint8))) == 0
        # the pipelined form of the exchange (bench.py): three all_gathers in flight, finished in
        # order, each handing every rank's partial of ITS step to every rank
        from curdlemsm.distributed import PartialExchange
        ex = PartialExchange()
        mine = [np.full(18, 1000 * step + rank, dtype=np.uint64) for step in range(3)]
        handles = [ex.start(m) for m in mine]
        for step, h in enumerate(handles):
            got = ex.finish(h)
            ok = ok and got.shape == (world, 18) and all((got[r] == 1000 * step + r).all() for r in range(world))
        np.save(os.path.join(result_dir, f"rank{rank}.npy"), np.array([int(ok)]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_window_split_allgather_sum_gloo(tmp_path, world):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(world, port, 48, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert np.load(tmp_path / f"rank{r}.npy")[0] == 1


def test_replica_shards_cover_every_item_once():
    sys.path.insert(0, PKG)
    from curdlemsm.distributed import replica_shard
    for k in (0, 1, 7, 8, 1024, 1025):
        for world in (1, 2, 3, 8):
            got = np.sort(np.concatenate([replica_shard(k, world, r) for r in range(world)]))
            assert (got == np.arange(k)).all()
            assert max(len(replica_shard(k, world, r)) for r in range(world)) == -(-k // world)


def test_window_partition_properties():
    sys.path.insert(0, PKG)
    from curdlemsm.distributed import window_partition
    for W in (1, 16, 18, 32, 52, 64):
        for world in (1, 2, 3, 4, 8, 64, 100):
            ranges = [window_partition(W, world, r) for r in range(world)]
            assert ranges[0][0] == 0 and ranges[-1][1] == W
            assert all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
            sizes = [e - b for b, e in ranges]
            assert max(sizes) - min(sizes) <= 1
