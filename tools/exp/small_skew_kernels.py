"""Round 6: every kernel's own duration for uniform scalars and for 16 / 64 distinct values at the verifier's sizes -- where a
small skewed call's extra time goes (k_merge_large and the reduction's chain over buckets just under the merge limit)."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(R, "go-curdleproofs_amd"), R, os.path.join(R, "tools")]
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np, torch, curdlemsm as cm, adversarial_inputs as adv
from bench import uniform_scalars
cm.init(0)
for n in (1268, 2548):
    d_pts = torch.empty((n, 12), dtype=torch.int64, device="cuda:0")
    cm.synth_points_walk_device(12345, 6789, n, d_pts.data_ptr())
    uni = uniform_scalars(np.random.default_rng(2), n)
    for fam in ("uniform", "distinct_64", "distinct_16"):
        sc, _ = adv.make_family(fam, n, uni, window_bits=cm.window_bits(n))
        d_sc = torch.from_numpy(np.ascontiguousarray(sc).view(np.int64)).to("cuda:0")
        cm.profile_enable(1)
        ks = {}
        for _ in range(6):
            cm.msm_g1_device(d_pts.data_ptr(), d_sc.data_ptr(), n)
            pr = cm.profile_last()
            for a, b in pr["kernels"].items(): ks.setdefault(a, []).append(b)
        cm.profile_enable(0)
        print(n, fam, {a: round(float(np.mean(b[1:])), 4) for a, b in ks.items()}, pr["entries"], pr["fragments"], flush=True)
