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
    for word in ("--gpus", "--steps", "--warmup", "--sweep", "whisk-batch", "verify", "--emulate-world"):
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
