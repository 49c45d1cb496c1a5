"""BASELINE config 3 end to end: full curdleproof.Verify at ell = 252 (n = 256) through
the host restatement, every MSM on the GPU.  Also ell = 60 / 124 / 508 like the
reference's BenchmarkVerifier (curdleproof_test.go:210-237).
    python tools/bench_verify.py [ell ...] > gpurun_out/verify.log
"""
import os, sys, time, json, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "go-curdleproofs_amd"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np
import curdlemsm as cm

cm.init(0)
out = {}
for n in ([int(a) + 4 for a in sys.argv[1:]] or (64, 128, 256, 512)):  # optional: ell values
    ell = n - 4
    rand = cm.Rand(0)
    crs = cm.CRS(ell, rand)
    perm = cm.Rand(42).generate_permutation(ell)
    k = rand.get_fr()
    Rs, Ss = rand.get_g1_affines(ell), rand.get_g1_affines(ell)
    Ts, Us, M, rs_m = cm.shuffle_permute_commit(crs, Rs, Ss, perm, k, rand)
    t_prove = 1e9
    for _ in range(3):      # best of three: the first call at a new size also sizes the workspaces
        t0 = time.perf_counter()
        proof = cm.prove(crs, Rs, Ss, Ts, Us, M, perm, k, rs_m, cm.Rand(42))
        t_prove = min(t_prove, time.perf_counter() - t0)
    assert cm.verify(crs, proof, Rs, Ss, Ts, Us, M, cm.Rand(43))
    reps = 10
    t0 = time.perf_counter()
    for i in range(reps):
        assert cm.verify(crs, proof, Rs, Ss, Ts, Us, M, cm.Rand(43 + i))
    t_seq = (time.perf_counter() - t0) / reps
    # the reference's BenchmarkVerifier times Verify on an already-decoded Proof value
    t0 = time.perf_counter()
    for i in range(reps):
        decoded = cm.Proof(proof)
    t_decode = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    for i in range(reps):
        assert cm.verify_proof(crs, decoded, Rs, Ss, Ts, Us, M, cm.Rand(43 + i))
    t_mem = (time.perf_counter() - t0) / reps
    # concurrent verifiers (the Whisk tracker batch of config 5 is many independent verifies)
    res = {}
    for nthreads in (4, 8):
        per = 6
        def worker(tid):
            for i in range(per):
                assert cm.verify(crs, proof, Rs, Ss, Ts, Us, M, cm.Rand(1000 + tid * 100 + i))
        th = [threading.Thread(target=worker, args=(t,)) for t in range(nthreads)]
        t0 = time.perf_counter()
        [t.start() for t in th]
        [t.join() for t in th]
        res[nthreads] = nthreads * per / (time.perf_counter() - t0)
    # cross-proof batch: one shared accumulator, one MSM for the whole batch (curdle_verify_batch).
    # Four distinct instances cycled; the batch does not merge bases across proofs, so the MSM
    # has the honest k * (4 ell + ~100) pairs.
    insts = [(proof, Rs, Ss, Ts, Us, M)]
    for j in range(3):
        r2 = cm.Rand(77 + j)
        p2 = r2.generate_permutation(ell)
        k2 = r2.get_fr()
        R2, S2 = r2.get_g1_affines(ell), r2.get_g1_affines(ell)
        T2, U2, M2, rsm2 = cm.shuffle_permute_commit(crs, R2, S2, p2, k2, r2)
        insts.append((cm.prove(crs, R2, S2, T2, U2, M2, p2, k2, rsm2, cm.Rand(5 + j)), R2, S2, T2, U2, M2))
    batch, batch_runs = {}, {}
    for kb, nt in ((16, 8), (64, 16), (256, 16), (1024, 16)):
        args = [list(c) for c in zip(*[insts[i % 4] for i in range(kb)])]
        prepared = cm.PreparedVerifyBatch(*args)  # marshalled once: the timed region is the C call
        assert all(prepared.run(crs, cm.Rand(5), nthreads=nt))
        ts = []
        for rep in range(3):  # best of three: one run is a few ms and thread start-up noise is of that order
            t0 = time.perf_counter()
            assert all(prepared.run(crs, cm.Rand(6 + rep), nthreads=nt))
            ts.append(time.perf_counter() - t0)
        batch[f"k={kb},threads={nt}"] = kb / min(ts)
        batch_runs[f"k={kb},threads={nt}"] = [round(t * 1e3, 2) for t in ts]
    out[f"shuffled_elements={ell}"] = {"proof_bytes": len(proof), "prove_ms": t_prove * 1e3, "verify_ms": t_seq * 1e3, "decode_ms": t_decode * 1e3, "verify_decoded_ms": t_mem * 1e3,
                                       "verifies_per_s_sequential": 1 / t_seq,
                                       "verifies_per_s_threads": res, "verifies_per_s_batch": batch, "batch_runs_ms": batch_runs}
    print(f"ell={ell}: prove {t_prove*1e3:.1f} ms, verify from bytes {t_seq*1e3:.2f} ms ({1/t_seq:.1f}/s) = decode {t_decode*1e3:.2f} + verify {t_mem*1e3:.2f} ms ({1/t_mem:.1f}/s), threads {res}, batch {batch}", flush=True)
print(json.dumps(out))
