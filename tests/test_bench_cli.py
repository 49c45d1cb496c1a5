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
    for word in ("--gpus", "--steps", "--warmup", "--sweep", "whisk-batch", "verify", "--emulate-world", "--bases-unchanged"):
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
    line = _run_forced_dist(["--steps", "6", "--warmup", "2", "--logn", "16", "--no-verify"])
    assert line["n_gpus"] == 1 and line["value"] and line["value"] > 0
    assert "over nccl" in line["config"]["parallelism"]            # the partials went through RCCL's all_gather
    assert "exchange" in line["config"]["host_ms_per_step"]
    cb = line["cpu_baseline"]                                       # attached on the distributed path too
    assert cb["gpu_matches_cpu"] and cb["gpu_full_size_verified"]


@pytest.mark.gpu
def test_the_accept_bit_exchange_of_config_5_runs_over_nccl(gpu):
    line = _run_forced_dist(["--mode", "whisk-batch", "--proofs", "96", "--steps", "1", "--warmup", "1"])
    assert line["accept_bits_exact"] is True and line["value"] and line["value"] > 0
    assert "over nccl" in line["config"]["parallelism"]


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
