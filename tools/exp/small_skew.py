"""Round 6: the reference's own adversarial inputs (all-equal scalars, <= 9-bit scalars) and 64 distinct values at the
reference's own sizes (ell = 252 / 512 pairs; 1,268 / 2,548 = the verifier's merged MSM), one synchronous resident call each,
against the C oracle (profiles/r06_adversarial_protocol_sizes.txt)."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(R, "go-curdleproofs_amd"), os.path.join(R, "oracle", "py"), R, os.path.join(R, "tools")]
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np, torch, curdlemsm as cm, adversarial_inputs as adv, coracle as co
from bench import uniform_scalars
cm.init(0)
for n in (252, 512, 1268, 2548):
    d_pts = torch.empty((n, 12), dtype=torch.int64, device="cuda:0")
    cm.synth_points_walk_device(12345, 6789, n, d_pts.data_ptr())
    uni = uniform_scalars(np.random.default_rng(2), n)
    pts = d_pts.cpu().numpy().view(np.uint64)
    base = None
    for fam in ("uniform", "all_equal", "small_9bit", "distinct_64"):
        sc, _ = adv.make_family(fam, n, uni, window_bits=cm.window_bits(n))
        sc = np.ascontiguousarray(sc)
        d_sc = torch.from_numpy(sc.view(np.int64)).to("cuda:0")
        exp = co.msm_pippenger(pts, sc, threads=4)
        for _ in range(5): r = cm.msm_g1_device(d_pts.data_ptr(), d_sc.data_ptr(), n)
        lat = []
        for _ in range(60):
            torch.cuda.synchronize(); t = time.perf_counter(); cm.msm_g1_device(d_pts.data_ptr(), d_sc.data_ptr(), n); lat.append((time.perf_counter() - t) * 1e3)
        m = float(np.median(lat)); base = base or m
        print(n, fam.ljust(12), round(m, 4), round(m / base, 3), bool((r == exp).all()), flush=True)
