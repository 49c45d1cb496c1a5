"""BASELINE config 3 end to end: full curdleproof.Verify at ell = 252 (n = 256) through
the host restatement, every MSM on the GPU.  Also ell = 60 / 124 / 508 like the
reference's BenchmarkVerifier (curdleproof_test.go:210-237).
    python tools/bench_verify.py > gpurun_out/verify.log
"""
import os, sys, time, json, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "go-curdleproofs_amd"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np
import curdlemsm as cm

cm.init(0)
out = {}
for n in (64, 128, 256, 512):
    ell = n - 4
    rand = cm.Rand(0)
    crs = cm.CRS(ell, rand)
    perm = cm.Rand(42).generate_permutation(ell)
    k = rand.get_fr()
    Rs, Ss = rand.get_g1_affines(ell), rand.get_g1_affines(ell)
    Ts, Us, M, rs_m = cm.shuffle_permute_commit(crs, Rs, Ss, perm, k, rand)
    t0 = time.perf_counter()
    proof = cm.prove(crs, Rs, Ss, Ts, Us, M, perm, k, rs_m, cm.Rand(42))
    t_prove = time.perf_counter() - t0
    assert cm.verify(crs, proof, Rs, Ss, Ts, Us, M, cm.Rand(43))
    reps = 10
    t0 = time.perf_counter()
    for i in range(reps):
        assert cm.verify(crs, proof, Rs, Ss, Ts, Us, M, cm.Rand(43 + i))
    t_seq = (time.perf_counter() - t0) / reps
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
    out[f"shuffled_elements={ell}"] = {"proof_bytes": len(proof), "prove_ms": t_prove * 1e3, "verify_ms": t_seq * 1e3,
                                       "verifies_per_s_sequential": 1 / t_seq,
                                       "verifies_per_s_threads": res}
    print(f"ell={ell}: prove {t_prove*1e3:.1f} ms, verify {t_seq*1e3:.2f} ms ({1/t_seq:.1f}/s), threads {res}", flush=True)
print(json.dumps(out))
