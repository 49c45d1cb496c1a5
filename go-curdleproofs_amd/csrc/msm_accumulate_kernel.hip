// gfx950 kernels of the BLS12-381 G1 MSM: the bucket accumulation, the dominant kernel (see msm_sort_kernels.hip for the
// phases of one MSM).  A translation unit of its own since round 6: its ~45 KB inlined loop is most of the build time.
#include "msm_kernels_common.h"

namespace curdle {

// Balanced bucket accumulation.  Lane t owns L consecutive positions of the
// bucket-sorted point list, whatever buckets they belong to, so every lane of
// every wave does the same number of mixed additions however skewed the scalars
// are (all-equal scalars, short top window).  When the list moves on to the
// next bucket the lane stores its running sum as a fragment of the finished
// bucket and starts again from infinity.  Fragments of one bucket are
// contiguous: slot = foff[bucket] + (t - start[bucket] / L).
#ifdef CURDLE_TRACE_WAVES
// Experiment build only: start / end (100 MHz wall clock) and hardware id of every wave of the LAST accumulate launch.
__device__ unsigned long long g_wave_trace[4 * 8192];
__device__ unsigned long long g_wave_clk[2 * 8192];  // s_memtime (shader clock) at the same two points
hipError_t debug_read_wave_trace(unsigned long long* out, size_t words) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wave_trace), words * 8 < sizeof(g_wave_trace) ? words * 8 : sizeof(g_wave_trace));
}
hipError_t debug_read_wave_clk(unsigned long long* out, size_t words) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wave_clk), words * 8 < sizeof(g_wave_clk) ? words * 8 : sizeof(g_wave_clk));
}
#endif
template <int WAVES>
__global__ void __launch_bounds__(kBlock, WAVES)
    k_accumulate(const A28* __restrict__ points, const u32* __restrict__ sorted, const u32* __restrict__ starts,
                 const u32* __restrict__ foff, X28* __restrict__ frags, u32 nb, u32 L, u32 set_points, u32 frag_stride,
                 u32 prio_shift) {
  const u32 t = blockIdx.x * kBlock + threadIdx.x;
  // base set blockIdx.y of a shared-scalar call: its own points and fragments, the one sorted list
  points = a28_at(points, (size_t)blockIdx.y * set_points);
  frags += (size_t)blockIdx.y * frag_stride;
  const u32 total = starts[nb];
  u32 pos = t * L;
#ifdef CURDLE_TRACE_WAVES
  const u32 wv = t >> 6;
  if ((t & 63u) == 0 && wv < 8192u) {
    g_wave_trace[4 * wv] = wall_clock64();
    g_wave_clk[2 * wv] = __builtin_amdgcn_s_memtime();
    g_wave_trace[4 * wv + 2] = __builtin_amdgcn_s_getreg(kGetregHwId);
    g_wave_trace[4 * wv + 3] = __builtin_amdgcn_s_getreg(kGetregXccId);
  }
#endif
  if (pos >= total) return;
  const u32 end = min(pos + L, total);
  // bucket containing `pos`: first index with starts[idx] > pos, minus one
  u32 lo = 0, hi = nb;
  while (lo < hi) {
    u32 mid = (lo + hi) >> 1;
    if (starts[mid] > pos) hi = mid;
    else lo = mid + 1;
  }
  u32 g = lo - 1;
  u32 gend = starts[lo];
  X28 acc;
  d28::set_inf(acc);
  // The lane's indices are read eight at a time into a register queue: between two of its
  // iterations the XCD's other lanes gather megabytes of points through the 4 MiB L2, so a
  // 4-byte read per iteration fetched a whole 128-byte line of `sorted` from memory every time
  // (2.1 GB of the launch's 4.3 GB, profiles/r02_fetch_calibration.txt); eight reads issued
  // back to back share one line fetch.  The gather of position pos + 1 is issued before the
  // addition of position pos.
  u32 q[8];
  auto refill = [&](u32 from) {
#pragma unroll
    for (int j = 0; j < 8; j++) q[j] = from + j < end ? sorted[from + j] : 0u;
  };
  refill(pos);
  u32 queued = 0;
  u32 e_next = q[0];
  A28 pt_next;
  d28::load(pt_next, a28_at(points, e_next & 0x7fffffffu));
  const u32 slot = __builtin_amdgcn_s_getreg(kGetregHwId) & 1u;  // this wave's slot on its SIMD, low bit
  for (; pos < end; pos++) {
    if (prio_shift) {
      if ((((u32)wall_clock64() >> prio_shift) ^ slot) & 1u)
        __builtin_amdgcn_s_setprio(3);
      else
        __builtin_amdgcn_s_setprio(0);
    }
    const u32 e = e_next;
    A28 pt = pt_next;
    if (++queued == 8) {
      refill(pos + 1);
      queued = 0;
    } else {
#pragma unroll
      for (int j = 0; j < 7; j++) q[j] = q[j + 1];
    }
    if (pos + 1 < end) {
      e_next = q[0];
      d28::load(pt_next, a28_at(points, e_next & 0x7fffffffu));
    }
    if (pos == gend) {
      d28::store(&frags[foff[g] + (t - starts[g] / L)], acc);
      d28::set_inf(acc);
      g++;
      gend = starts[g + 1];
      if (gend == pos) {
        // An EMPTY bucket.  Uniform scalars leave next to none, skewed ones leave runs of thousands (all-equal scalars: two
        // occupied buckets per window; a hot window: one) and a lane that walked such a run one dependent load at a time held
        // its whole wave back -- 32,768 loads at N = 2^20: the launch took 4.0 ms against 2.25 (profiles/r06_adversarial.json).
        // Bisect for the first slot that starts beyond pos, as at the lane's start.
        u32 l2 = g + 2, h2 = nb;
        while (l2 < h2) {
          const u32 mid = (l2 + h2) >> 1;
          if (starts[mid] > pos) h2 = mid;
          else l2 = mid + 1;
        }
        g = l2 - 1;
        gend = starts[l2];
      }
    }
    if (d28::affine_is_inf(pt)) continue;  // (0,0) = infinity (curdleproof.go:23)
    if (e >> 31) {
      F28 z;
      d28::set_zero(z);
      d28::sub_raw<4>(pt.y, z, pt.y);  // 4p - y
    }
    d28::madd<true>(acc, pt.x, pt.y);
  }
  d28::store(&frags[foff[g] + (t - starts[g] / L)], acc);
#ifdef CURDLE_TRACE_WAVES
  if ((t & 63u) == 0 && wv < 8192u) {
    g_wave_trace[4 * wv + 1] = wall_clock64();
    g_wave_clk[2 * wv + 1] = __builtin_amdgcn_s_memtime();
  }
#endif
}

hipError_t launch_accumulate(const MsmPlan& p, const MsmWorkspace& ws, hipStream_t stream) {
  const u32 nw = p.win_end - p.win_begin;
  const u32 nb = p.k * p.NB;
  const u32 nlanes = cdiv((u64)nw * p.n, p.L);
  // two waves per SIMD (203 VGPRs, no spills): a three-wave build (168 VGPRs) spilled 26
  // registers and was slower
  hipLaunchKernelGGL(k_accumulate<2>, dim3(cdiv(nlanes, kBlock), p.sets), dim3(kBlock), 0, stream,
                     reinterpret_cast<const A28*>(ws.points28), ws.sorted, ws.starts, ws.foff,
                     reinterpret_cast<X28*>(ws.frags), nb, p.L, p.n, p.frag_stride, p.acc_prio);
  return hipGetLastError();
}

}  // namespace curdle
