"""BASELINE.json configs through the C ABI on one GPU (inputs resident in HBM):
  C2  one msmaccumulator.Verify-sized MSM, N = 2^16
  C3  the MSM work of one full Verify at ell = 252: ten MSMs of m = 8 pairs and the
      final batched MSM of 5*252+8 = 1,268 pairs, as ONE batched call
  C5  1024 independent 628-pair MSMs (Whisk tracker batch, ell = 124) in one call
Only the MSM part of Verify runs here (the transcript / Fr glue is the Go host's).
    python tools/bench_configs.py > gpurun_out/configs.log
"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "go-curdleproofs_amd"))
import numpy as np
import torch
import curdlemsm as cm

R_MOD = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
cm.init(0)
cm.profile_enable(True)
nmax = 1024 * 628
d_pts = torch.empty((nmax, 12), dtype=torch.int64, device="cuda:0")
cm.synth_points_walk_device(1234567 % R_MOD, 7654321 % R_MOD, nmax, d_pts.data_ptr())
rng = np.random.default_rng(5)
sc = rng.integers(0, 1 << 64, size=(nmax, 4), dtype=np.uint64)
sc[:, 3] &= np.uint64((1 << 62) - 1)
d_sc = torch.from_numpy(sc.view(np.int64)).to("cuda:0")

def timeit(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps

out = {}
n = 1 << 16
dt = timeit(lambda: cm.msm_g1_device(d_pts.data_ptr(), d_sc.data_ptr(), n), 20)
out["C2_single_msm_2^16"] = {"ms": dt * 1e3, "pairs_per_s": n / dt, "kernels_ms": cm.profile_last()["kernels"]}

offs = np.cumsum([0] + [8] * 10 + [1268]).astype(np.uint64)
dt = timeit(lambda: cm.msm_g1_batch_device(d_pts.data_ptr(), d_sc.data_ptr(), offs), 20)
out["C3_verify_shaped_msms_ell252"] = {"ms": dt * 1e3, "verify_msm_sets_per_s": 1 / dt, "pairs_per_s": int(offs[-1]) / dt,
                                       "kernels_ms": cm.profile_last()["kernels"]}
dt1 = timeit(lambda: cm.msm_g1_device(d_pts.data_ptr(), d_sc.data_ptr(), 1268), 20)
out["C3_final_msm_1268_alone"] = {"ms": dt1 * 1e3}

for k in (64, 256, 1024):
    offs = (np.arange(k + 1) * 628).astype(np.uint64)
    dt = timeit(lambda: cm.msm_g1_batch_device(d_pts.data_ptr(), d_sc.data_ptr(), offs), 5)
    out[f"C5_batch_{k}x628"] = {"ms": dt * 1e3, "msms_per_s": k / dt, "pairs_per_s": k * 628 / dt,
                                "kernels_ms": cm.profile_last()["kernels"]}
# verify-shaped sets, many at once (what a verifier farm would submit)
for k in (64, 512):
    sizes = ([8] * 10 + [1268]) * k
    offs = np.cumsum([0] + sizes).astype(np.uint64)
    if offs[-1] > nmax:
        continue
    dt = timeit(lambda: cm.msm_g1_batch_device(d_pts.data_ptr(), d_sc.data_ptr(), offs), 5)
    out[f"C3x{k}_verify_shaped_sets"] = {"ms": dt * 1e3, "verify_msm_sets_per_s": k / dt, "pairs_per_s": int(offs[-1]) / dt}
print(json.dumps(out, indent=1))
