"""bench.py is the driver's contract; without a GPU all that can be checked is that it parses,
documents its modes, and fails loudly (non-zero, no JSON line claiming a value) when there is no
device -- the MSM has no CPU fallback to time instead."""
import json
import os
import subprocess
import sys

from conftest import ROOT


def test_bench_help_lists_the_modes():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0
    for word in ("--gpus", "--steps", "--warmup", "--sweep", "whisk-batch", "verify", "--emulate-world", "--bases-unchanged",
                 "--convert-per-call"):
        assert word in p.stdout, word


def test_bench_refuses_to_run_without_a_device(cm):
    if cm.device_available():
        return          # on a GPU box the real run is the test (the driver runs it)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"], capture_output=True, text=True,
                       timeout=300)
    assert p.returncode != 0
    assert "no HIP device" in (p.stderr + p.stdout)
    for line in p.stdout.splitlines():            # and no line that could be read as a measurement
        if line.startswith("{"):
            assert json.loads(line).get("value") is None


# ------------------------------------------------------------------------------------------------
# On the GPU box: everything a multi-GPU run would execute for the first time -- RCCL
# initialisation with device_id, the pinned PartialExchange, the lagged all_gather of the
# partials, the accept-bit all_gather of the config-5 leg, the distributed sweep, the CPU baseline
# on an N > 1 line -- executed ONCE on the one GPU there is: bench.py in a FRESH child process
# (the launcher env is set before anything touches the GPU) with the nccl backend forced at world
# size 1 ("the nccl backend has never run, not even at world size 1").
# ------------------------------------------------------------------------------------------------
import pytest


def _run_forced_dist(extra, timeout=600):
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", LOCAL_WORLD_SIZE="1", MASTER_ADDR="127.0.0.1",
               CURDLE_DIST_BACKEND="nccl")
    env.pop("MASTER_PORT", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-dist"] + extra, env=env,
                       capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-1500:]
    return json.loads(lines[0])


@pytest.mark.gpu
def test_the_nccl_path_of_the_bench_runs_at_world_size_one(gpu):
    # (--exchange-batch 3 with 6 steps: two full batches through the lagged finish, and the flush at the end)
    line = _run_forced_dist(["--steps", "6", "--warmup", "2", "--logn", "16", "--no-verify", "--exchange-batch", "3"])
    assert line["n_gpus"] == 1 and line["value"] and line["value"] > 0
    assert "over nccl" in line["config"]["parallelism"]            # the partials went through RCCL's all_gather
    assert "exchange" in line["config"]["host_ms_per_step"]
    cb = line["cpu_baseline"]                                       # attached on the distributed path too
    assert cb["gpu_matches_cpu"] and cb["gpu_full_size_verified"]
    _check_multi_gpu_keys(line, 1, "nccl")


def _check_multi_gpu_keys(line, world, backend):
    """What an N > 1 line says beside `value`: the ranks RCCL saw, the contract (kept bases), the
    per-rank step, one rank's and the whole call's synchronous latency, the exchange alone, and the same contract at
    one rank as the baseline a speed-up is a ratio to."""
    mg = line["config"]["multi_gpu"]
    assert mg["rccl_ranks"] == world == line["n_gpus"] and mg["backend"] == backend
    assert mg["bases_unchanged"] is (world > 1) and line["config"]["bases_unchanged_flag"] is (world > 1)
    assert mg["exchange_batch"] >= 1
    for key in ("rank_step_ms", "rank_single_call_ms", "whole_single_call_ms", "exchange_ms"):
        assert mg[key] > 0, key
    assert mg["whole_single_call_ms"] >= 0.9 * mg["rank_single_call_ms"]
    assert mg["scaling_baseline"]["ms_per_step"] > 0 and mg["scaling_baseline"]["pairs_per_s"] > 0
    assert "unmeasured" in mg["note"] or "No scaling curve" in mg["note"]


@pytest.mark.gpu
def test_two_ranks_over_gloo_print_the_multi_gpu_keys(gpu):
    """The N = 2 line, rehearsed with two ranks sharing the one GPU of the lease (gloo: RCCL refuses two ranks on one
    device; the figures mean nothing, the branches do): the window split passes CURDLE_MSM_BASES_UNCHANGED by default,
    every rank runs the one-rank baseline of the same contract, rank 0 prints the new keys and the result still
    matches the CPU port and the closed form."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, CURDLE_DIST_BACKEND="gloo")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6",
                        "--warmup", "2", "--logn", "16", "--no-verify", "--exchange-batch", "4"], env=env, capture_output=True, text=True,
                       timeout=600)      # (4 of 6 steps in one all_gather, the other two by the flush)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-1500:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["value"] > 0 and "over gloo" in line["config"]["parallelism"]
    assert line["cpu_baseline"]["gpu_matches_cpu"] and line["cpu_baseline"]["gpu_full_size_verified"]
    _check_multi_gpu_keys(line, 2, "gloo")


@pytest.mark.gpu
def test_a_distributed_line_carries_the_size_table_too(gpu):
    """north_star wants N in {2^10 .. 2^20} at 1 / 2 / 4 / 8 GPUs: an N > 1 line (here: the N > 1 path at world size 1
    over nccl) carries the table itself -- one synchronous distributed call per size on all ranks, the CPU port beside
    it on rank 0, bit-compared -- so that the driver's multi-GPU runs fill it without another command."""
    line = _run_forced_dist(["--steps", "3", "--warmup", "1", "--no-verify"])
    rows = line["sweep"]
    assert [r["logn"] for r in rows] == [10, 12, 14, 16, 18, 20] and "msm_g1_distributed" in line["sweep_what"]
    assert all(r["gpu_matches_cpu"] and r["wall_ms"] > 0 and r["cpu_port_pairs_per_s"] > 0 for r in rows)
    assert rows[-1]["pairs_per_s"] > rows[0]["pairs_per_s"]
    _check_multi_gpu_keys(line, 1, "nccl")


@pytest.mark.gpu
def test_the_accept_bit_exchange_of_config_5_runs_over_nccl(gpu):
    line = _run_forced_dist(["--mode", "whisk-batch", "--proofs", "96", "--steps", "1", "--warmup", "1"])
    assert line["accept_bits_exact"] is True and line["value"] and line["value"] > 0
    assert "over nccl" in line["config"]["parallelism"]
    # the line says how many host threads bound the step and what share of it the GPU had work
    assert line["config"]["host_threads"] >= 1
    gs = line["config"]["gpu_share"]
    assert 0 < gs["gpu_timeline_coverage"] <= 1 and gs["kernel_overlap_factor"] >= 1 and "16" in gs["proofs_per_s_by_host_threads"]


@pytest.mark.gpu
def test_the_distributed_sweep_runs_on_every_rank(gpu):
    line = _run_forced_dist(["--sweep", "--sweep-sizes", "1024,32768"])
    assert line["n_gpus"] == 1 and line["backend"] == "nccl" and line["all_results_match_cpu"] is True
    assert [r["n_pairs"] for r in line["sweep"]] == [1024, 32768]
    assert all(r["gpu_matches_cpu"] and r["wall_ms"] > 0 for r in line["sweep"])


@pytest.mark.gpu
def test_the_default_line_carries_the_size_table(gpu):
    """north_star's N = 2^10 .. 2^20 table is part of the line the driver records -- wall time,
    pairs/s, the dominant kernel's HBM and multiply-issue fractions, the CPU port on the same inputs, bit-compared."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--no-verify"],
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    line = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["value"] and line["roofline"]["frac"] > 0 and line["cpu_baseline"]["gpu_matches_cpu"]
    rows = line["sweep"]
    assert [r["logn"] for r in rows] == [10, 12, 14, 16, 18, 20]
    for r in rows:
        assert r["gpu_matches_cpu"] is True and r["wall_ms"] > 0 and r["cpu_port_pairs_per_s"] > 0
        assert 0 < r["hbm_frac"] < 0.02 and 0 < r["valu_frac"] < 1.0
    assert rows[-1]["pairs_per_s"] > rows[0]["pairs_per_s"]
    # the adversarial families are timed in the same line, each result against the closed form
    adv = line["adversarial"]
    fams = {(r["family"], r["logn"]) for r in adv["rows"]}
    for fam in ("uniform", "all_equal", "small_9bit", "infinity_1pct", "distinct_64"):
        assert {(fam, 12), (fam, 16), (fam, 20)} <= fams, fam
    assert ("hot_window", 16) in fams
    assert all(r["ok"] for r in adv["rows"])
    for r in adv["rows"]:
        if r["family"] != "uniform" and r["logn"] >= 16:
            assert r["ratio_sync"] < 1.5 and r["ratio_host_slices"] < 1.5, r      # the bar is 1.25; boxes jitter
