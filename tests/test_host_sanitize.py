"""The host layer of libcurdlemsm.so (wire-format readers of attacker-controlled proof bytes,
PointDecoder, the MsmAccumulator table, transcript, the five arguments, curdleproof Prove /
Verify, the device accumulator's description path) built with AddressSanitizer + UBSan over a
test-only host backend (tests/hostbuild/) and driven through completeness, the soundness
flips of curdleproof_test.go:48-182, the mirror-vs-device accumulator comparison and a
mutation fuzz of every parser -- the counterpart of the reference CI's `go test -race`
(.github/workflows/buildlintcheck.yml:21).  CPU only (GPU AddressSanitizer is not available
on this pool); any sanitizer report aborts the binary and fails the test."""
import os
import subprocess

import pytest

from conftest import ROOT

HB = os.path.join(ROOT, "tests", "hostbuild")


@pytest.fixture(scope="module")
def harness():
    subprocess.run(["make", "-C", HB], check=True, capture_output=True, timeout=600)
    return os.path.join(HB, "host_flow_asan")


def _run(exe, *args):
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    p = subprocess.run([exe, *args], capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "ERROR: AddressSanitizer" not in p.stderr and "runtime error" not in p.stderr, p.stderr
    return p.stdout


@pytest.fixture(scope="module")
def tsan_harness():
    subprocess.run(["make", "-C", HB, "tsan"], check=True, capture_output=True, timeout=600)
    return os.path.join(HB, "host_flow_tsan")


def test_batch_threads_under_tsan(tsan_harness):
    """The counterpart of the reference CI's `go test -race`: batch verification's producer and
    worker threads (decode-ahead, queued group MSMs, the decoder buffer pool) under
    ThreadSanitizer, over the test-only backend."""
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1")
    for args in (("flow", "12"), ("whisk", "0")):
        p = subprocess.run([tsan_harness, *args], capture_output=True, text=True, timeout=900, env=env)
        assert p.returncode == 0, p.stdout + p.stderr
        assert "ThreadSanitizer" not in p.stderr, p.stderr
        assert "ok" in p.stdout


def test_batch_workers_never_wait_for_a_decoding_while_holding_a_slot(tsan_harness):
    """a batch worker kept its queued group's workspace slot while it blocked on a
    chunk the producers had not decoded yet, and a producer whose decoding needs a slot of its
    own (chunk beyond the two-kernel size, or all decode contexts taken) could then find all
    eight held by waiting workers.  Run with 16 threads over the stub's eight modelled slots,
    groups and chunks of one proof, slow slot-taking decodings: it must finish (the harness is
    killed after the timeout otherwise) with exact accept bits and every slot returned."""
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1", CURDLE_STUB_SLOTS="1", CURDLE_TWO_KERNEL_MAX="0",
               CURDLE_BATCH_GROUP="1", CURDLE_BATCH_CHUNK="1", CURDLE_STUB_DECODE_DELAY_MS="20")
    p = subprocess.run([tsan_harness, "slots", "12"], capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "ThreadSanitizer" not in p.stderr, p.stderr
    assert "over 8 slots: ok" in p.stdout


def test_batch_is_sharded_over_the_configured_devices(tsan_harness):
    """curdle_verify_batch through the C ABI with the stub backend posing as three devices: the
    proofs are sharded over the contexts (config 5: replicas), every thread of a shard selects
    its device first, each posed device sees work, the accept bits are exact -- under
    ThreadSanitizer."""
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1", CURDLE_STUB_DEVICES="3", CURDLE_BATCH_CHUNK="2", CURDLE_BATCH_GROUP="3")
    p = subprocess.run([tsan_harness, "devices", "12"], capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "ThreadSanitizer" not in p.stderr, p.stderr
    assert "sharded over 3 devices: ok" in p.stdout


def test_protocol_flow_under_asan_ubsan(harness):
    out = _run(harness, "flow", "12")
    assert "mirror == device accumulator, batch: ok" in out


def test_baseline_config_1_shape_on_the_cpu_only_build(harness):
    """BASELINE config 1 is "BenchmarkVerifier shuffled_elements=60 on the pure-Go CPU path
    (plumbing, no GPU)".  There is no Go here; its counterpart is the host restatement at
    ell = 60 over the test-only CPU backend: Prove, Verify (completeness, the soundness flips of
    curdleproof_test.go:48-182, both accumulators, batch verification) with no GPU in the
    process -- under AddressSanitizer + UBSan."""
    out = _run(harness, "flow", "60")
    assert "flow ell=60" in out and "batch: ok" in out


def test_vectorised_point_compression_matches_scalar(harness):
    out = _run(harness, "compress", "0")
    assert "batch == scalar" in out


def test_whisk_shuffle_flow_under_asan_ubsan(harness):
    out = _run(harness, "whisk", "0")
    assert "both routes: ok" in out


def test_parsers_survive_mutated_proofs_under_asan_ubsan(harness):
    out = _run(harness, "fuzz", "12", "200")
    assert "0 accepted" in out
