#!/usr/bin/env python3
"""Where the accumulation's time goes wave by wave (experiment build only).

Needs the experiment build (-DCURDLE_TRACE_WAVES: k_accumulate stamps the 100 MHz wall clock and the shader clock
at each wave's first and last instruction, and its hardware id); build it and point CURDLE_MSM_LIB at it:
    make -C go-curdleproofs_amd trace
    CURDLE_MSM_LIB=$PWD/build_trace/libcurdlemsm_trace.so python tools/trace_waves.py [logn]
Runs ONE synchronous MSM (nothing else in flight), reads the stamps of its accumulate launch and prints the
spread of starts, ends and durations, per XCD and per SIMD slot."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "go-curdleproofs_amd"))
sys.path.insert(0, ROOT)


def main():
    logn = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    import numpy as np
    import torch
    import curdlemsm as cm
    from bench import uniform_scalars
    cm.init(0)
    n = 1 << logn
    d_pts = torch.empty((n, 12), dtype=torch.int64, device="cuda:0")
    cm.synth_points_walk_device(12345, 6789, n, d_pts.data_ptr())
    sc = uniform_scalars(np.random.default_rng(2), n)
    d_sc = torch.from_numpy(sc.view(np.int64)).to("cuda:0")
    torch.cuda.synchronize()
    for _ in range(5):
        cm.msm_g1_device(d_pts.data_ptr(), d_sc.data_ptr(), n)
    torch.cuda.synchronize()
    fn = cm._lib.curdle_debug_wave_trace
    fn.argtypes = [C.c_void_p, C.c_size_t]
    fn.restype = C.c_int
    W = 8192
    buf = np.zeros((W, 4), dtype=np.uint64)
    assert fn(buf.ctypes.data, W * 4) == 0
    live = buf[:, 1] > 0
    clk = np.zeros((W, 2), dtype=np.uint64)
    fc = cm._lib.curdle_debug_wave_clk
    fc.argtypes = [C.c_void_p, C.c_size_t]
    fc.restype = C.c_int
    assert fc(clk.ctypes.data, W * 2) == 0
    t0 = buf[live, 0].astype(np.int64)
    t1 = buf[live, 1].astype(np.int64)
    hw = buf[live, 2].astype(np.int64)
    xcc = buf[live, 3].astype(np.int64) & 0xF
    base = t0.min()
    s = (t0 - base) / 100.0   # microseconds
    e = (t1 - base) / 100.0
    d = e - s
    q = lambda a, p: float(np.percentile(a, p))
    print("waves traced: %d   kernel span %.1f us" % (live.sum(), e.max()))
    print("start  us: min %.1f  p50 %.1f  p90 %.1f  p99 %.1f  max %.1f" % (s.min(), q(s, 50), q(s, 90), q(s, 99), s.max()))
    print("end    us: min %.1f  p1 %.1f  p10 %.1f  p50 %.1f  p90 %.1f  p99 %.1f  max %.1f" % (e.min(), q(e, 1), q(e, 10), q(e, 50), q(e, 90), q(e, 99), e.max()))
    print("length us: min %.1f  p1 %.1f  p10 %.1f  p50 %.1f  p90 %.1f  p99 %.1f  max %.1f" % (d.min(), q(d, 1), q(d, 10), q(d, 50), q(d, 90), q(d, 99), d.max()))
    dc = (clk[live, 1].astype(np.int64) - clk[live, 0].astype(np.int64)) / np.maximum(t1 - t0, 1) * 100.0  # s_memtime ticks per microsecond
    print("s_memtime ticks per us over a wave's life: min %.1f  p50 %.1f  max %.1f  (the counter s_memtime reads; 100 = it is the same 100 MHz clock)" % (dc.min(), q(dc, 50), dc.max()))
    print("mean length %.1f us = %.3f of the span: the rest of the span is slots standing empty" % (d.mean(), d.mean() / e.max()))
    for x in sorted(set(xcc.tolist())):
        m = xcc == x
        print("  XCD %d: %4d waves  start p50 %.1f  end p50 %.1f  end max %.1f  length p50 %.1f" % (x, m.sum(), q(s[m], 50), q(e[m], 50), e[m].max(), q(d[m], 50)))
    # gfx9 HW_ID: wave 3:0, simd 5:4, pipe 7:6, cu 11:8, sh 12, se 15:13
    cu = (hw >> 8) & 0xF
    se = (hw >> 13) & 0x7
    simd = (hw >> 4) & 0x3
    key = xcc * 4096 + se * 256 + ((hw >> 12) & 1) * 64 + cu * 4 + simd
    uniq, cnt = np.unique(key, return_counts=True)
    print("distinct (XCD, SE, SH, CU, SIMD): %d; waves per SIMD: min %d max %d; histogram %s" % (len(uniq), cnt.min(), cnt.max(), np.bincount(cnt).tolist()))
    # waves that started late: did they wait for a slot?
    late = s > 50.0
    print("waves starting later than 50 us: %d (their mean start %.1f us, mean length %.1f us)" % (late.sum(), s[late].mean() if late.any() else 0.0, d[late].mean() if late.any() else 0.0))


if __name__ == "__main__":
    main()
