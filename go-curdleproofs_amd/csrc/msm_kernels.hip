// gfx950 kernels of the BLS12-381 G1 MSM (Pippenger bucket method).
//
// Replaces the body of gnark-crypto's (*G1Jac).MultiExp as the reference uses it
// (/root/reference/msmaccumulator/msmaccumulator.go:59 and the call sites in
// SURVEY.md section 8a).  Phases, one kernel each:
//   hist        scalar Montgomery->canonical, signed c-bit digits, bucket sizes
//   scan        per-window exclusive prefix of the bucket sizes
//   scatter     point indices grouped by (window, bucket)
//   accumulate  one lane per bucket: gathers its affine points (96 B each, AoS as
//               gnark stores them) and sums them with XYZZ mixed additions
//   reduce      sum_b (b+1)*bucket[b] per window, as running sums over short
//               segments (the serial chain is what matters: one G1 addition is
//               ~10 us of dependent 32-bit multiply-adds on one lane)
//   window_sum  per-window tree sum of the segment results
// The last 255 doublings (combining the <= 64 window sums) are O(1) work with a
// serial dependency chain and are done by the host side of the library.
//
// This is 381-bit integer arithmetic: no MFMA, no floating point.  Wave size 64.
#include <hip/hip_runtime.h>

#include "msm_kernels.h"

namespace curdle {

static constexpr int kBlock = 256;

// ---------------------------------------------------------------------------
// Signed-digit recoding shared by hist and scatter.
// f(w, d): d in [-2^(c-1), 2^(c-1)], called for w = 0..W-1 in order.
// The scalar is shifted down c bits per window so every limb index is static
// (runtime-indexed register arrays would go to scratch).
// ---------------------------------------------------------------------------
template <class F>
__device__ __forceinline__ void for_each_digit(Fr s, int c, int W, F&& f) {
  const u32 mask = (1u << c) - 1u;
  const u32 half = 1u << (c - 1);
  u32 carry = 0;
  for (int w = 0; w < W; w++) {
    u32 raw = (s.l[0] & mask) + carry;
#pragma unroll
    for (int i = 0; i < 7; i++) s.l[i] = (s.l[i] >> c) | (s.l[i + 1] << (32 - c));
    s.l[7] >>= c;
    int d;
    if (raw > half) {
      d = (int)raw - (int)(1u << c);
      carry = 1;
    } else {
      d = (int)raw;
      carry = 0;
    }
    f(w, d);
  }
}

__device__ __forceinline__ Fr load_scalar_canonical(const uint4* scalars, u32 i) {
  uint4 lo = scalars[2 * (size_t)i], hi = scalars[2 * (size_t)i + 1];
  Fr m, s;
  m.l[0] = lo.x; m.l[1] = lo.y; m.l[2] = lo.z; m.l[3] = lo.w;
  m.l[4] = hi.x; m.l[5] = hi.y; m.l[6] = hi.z; m.l[7] = hi.w;
  f_from_mont<FrParams>(s, m);  // gnark fr.Element is Montgomery; digits need the integer
  return s;
}

__global__ void __launch_bounds__(kBlock) k_hist(const uint4* __restrict__ scalars, MsmPlan p,
                                                 u32* __restrict__ counts) {
  u32 i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= p.n) return;
  Fr s = load_scalar_canonical(scalars, i);
  for_each_digit(s, p.c, p.W, [&](int w, int d) {
    if (d != 0 && w >= p.win_begin && w < p.win_end) {
      u32 mag = d < 0 ? (u32)(-d) : (u32)d;
      atomicAdd(&counts[(size_t)(w - p.win_begin) * p.B + (mag - 1)], 1u);
    }
  });
}

// One block per window: starts[b] = window base + sum_{b' < b} counts[b'].
__global__ void __launch_bounds__(1024) k_scan(const u32* __restrict__ counts, u32* __restrict__ starts,
                                               u32* __restrict__ cursor, MsmPlan p) {
  __shared__ u32 part[1024];
  const u32 lw = blockIdx.x;
  const u32 tid = threadIdx.x;
  const u32 per = (p.B + 1023u) / 1024u;
  const u32 lo = tid * per;
  const u32* cw = counts + (size_t)lw * p.B;
  u32 sum = 0;
  for (u32 k = 0; k < per; k++)
    if (lo + k < p.B) sum += cw[lo + k];
  part[tid] = sum;
  __syncthreads();
  for (u32 off = 1; off < 1024; off <<= 1) {
    u32 v = tid >= off ? part[tid - off] : 0;
    __syncthreads();
    part[tid] += v;
    __syncthreads();
  }
  u32 run = part[tid] - sum + lw * p.n;  // exclusive prefix + base of this window's slice of `sorted`
  for (u32 k = 0; k < per; k++) {
    if (lo + k < p.B) {
      size_t o = (size_t)lw * p.B + lo + k;
      starts[o] = run;
      cursor[o] = run;
      run += cw[lo + k];
    }
  }
}

__global__ void __launch_bounds__(kBlock) k_scatter(const uint4* __restrict__ scalars, MsmPlan p,
                                                    u32* __restrict__ cursor, u32* __restrict__ sorted) {
  u32 i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= p.n) return;
  Fr s = load_scalar_canonical(scalars, i);
  for_each_digit(s, p.c, p.W, [&](int w, int d) {
    if (d != 0 && w >= p.win_begin && w < p.win_end) {
      u32 mag = d < 0 ? (u32)(-d) : (u32)d;
      u32 pos = atomicAdd(&cursor[(size_t)(w - p.win_begin) * p.B + (mag - 1)], 1u);
      sorted[pos] = i | (d < 0 ? 0x80000000u : 0u);
    }
  });
}

__device__ __forceinline__ void load_fp(Fp& r, const uint4* src) {
  uint4 a = src[0], b = src[1], c = src[2];
  r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w;
  r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w;
  r.l[8] = c.x; r.l[9] = c.y; r.l[10] = c.z; r.l[11] = c.w;
}
__device__ __forceinline__ void store_fp(uint4* dst, const Fp& r) {
  dst[0] = make_uint4(r.l[0], r.l[1], r.l[2], r.l[3]);
  dst[1] = make_uint4(r.l[4], r.l[5], r.l[6], r.l[7]);
  dst[2] = make_uint4(r.l[8], r.l[9], r.l[10], r.l[11]);
}
__device__ __forceinline__ void load_xyzz(G1XYZZ& r, const G1XYZZ* src) {
  const uint4* s = reinterpret_cast<const uint4*>(src);
  load_fp(r.x, s);
  load_fp(r.y, s + 3);
  load_fp(r.zz, s + 6);
  load_fp(r.zzz, s + 9);
}
__device__ __forceinline__ void store_xyzz(G1XYZZ* dst, const G1XYZZ& r) {
  uint4* d = reinterpret_cast<uint4*>(dst);
  store_fp(d, r.x);
  store_fp(d + 3, r.y);
  store_fp(d + 6, r.zz);
  store_fp(d + 9, r.zzz);
}

// One lane per (window, bucket): sum of the bucket's points.
__global__ void __launch_bounds__(kBlock, 2)
    k_accumulate(const uint4* __restrict__ points, const u32* __restrict__ sorted, const u32* __restrict__ starts,
                 const u32* __restrict__ counts, G1XYZZ* __restrict__ buckets, u32 ntasks) {
  u32 t = blockIdx.x * kBlock + threadIdx.x;
  if (t >= ntasks) return;
  const u32 start = starts[t];
  const u32 cnt = counts[t];
  G1XYZZ acc;
  g1_set_inf(acc);
  for (u32 k = 0; k < cnt; k++) {
    const u32 e = sorted[start + k];
    const u32 idx = e & 0x7fffffffu;
    const uint4* src = points + (size_t)idx * 6;
    Fp x, y;
    load_fp(x, src);
    load_fp(y, src + 3);
    if (f_is_zero(x) & f_is_zero(y)) continue;  // (0,0) = infinity (curdleproof.go:23)
    if (e >> 31) fp_neg(y, y);
    g1_madd(acc, x, y);
  }
  store_xyzz(&buckets[t], acc);
}

// r = k * p for a small k (k < 2^16): left-to-right double-and-add.
__device__ __forceinline__ void g1_mul_small(G1XYZZ& r, const G1XYZZ& p, u32 k) {
  g1_set_inf(r);
  if (k == 0) return;
  int top = 31 - __clz(k);
  for (int bit = top; bit >= 0; bit--) {
    g1_dbl(r);
    if ((k >> bit) & 1u) g1_add(r, p);
  }
}

// One lane per segment of `seg` consecutive buckets of one window:
//   out = sum_{u < seg} (lo + u + 1) * bucket[lo + u]
// as the classic running sum over the segment plus lo * (segment total).
__global__ void __launch_bounds__(kBlock, 2)
    k_bucket_reduce(const G1XYZZ* __restrict__ buckets, G1XYZZ* __restrict__ partials, MsmPlan p, u32 nw) {
  u32 t = blockIdx.x * kBlock + threadIdx.x;
  if (t >= nw * p.nseg) return;
  const u32 lw = t / p.nseg;
  const u32 j = t - lw * p.nseg;
  const u32 lo = j * p.seg;
  const G1XYZZ* bw = buckets + (size_t)lw * p.B + lo;
  G1XYZZ run, acc, b;
  g1_set_inf(run);
  g1_set_inf(acc);
  for (int u = (int)p.seg - 1; u >= 0; u--) {
    load_xyzz(b, &bw[u]);
    g1_add(run, b);
    g1_add(acc, run);
  }
  if (lo != 0) {
    G1XYZZ s;
    g1_mul_small(s, run, lo);
    g1_add(acc, s);
  }
  store_xyzz(&partials[t], acc);
}

// One block per window: winsums[w] = sum of the window's segment results.
__global__ void __launch_bounds__(kBlock, 2)
    k_window_sum(const G1XYZZ* __restrict__ partials, G1XYZZ* __restrict__ winsums, MsmPlan p) {
  __shared__ G1XYZZ sh[kBlock];
  const u32 lw = blockIdx.x;
  const u32 tid = threadIdx.x;
  const G1XYZZ* pw = partials + (size_t)lw * p.nseg;
  G1XYZZ acc, b;
  g1_set_inf(acc);
  for (u32 k = tid; k < p.nseg; k += kBlock) {
    load_xyzz(b, &pw[k]);
    g1_add(acc, b);
  }
  sh[tid] = acc;
  __syncthreads();
  for (u32 off = kBlock / 2; off > 0; off >>= 1) {
    if (tid < off) {
      b = sh[tid + off];
      g1_add(acc, b);
      sh[tid] = acc;
    }
    __syncthreads();
  }
  if (tid == 0) store_xyzz(&winsums[lw], acc);
}

// ---------------------------------------------------------------------------
// Launchers
// ---------------------------------------------------------------------------
static inline u32 cdiv(u64 a, u32 b) { return (u32)((a + b - 1) / b); }

hipError_t launch_hist(const MsmPlan& p, const MsmWorkspace& ws, const void* d_scalars, hipStream_t stream) {
  hipLaunchKernelGGL(k_hist, dim3(cdiv(p.n, kBlock)), dim3(kBlock), 0, stream,
                     reinterpret_cast<const uint4*>(d_scalars), p, ws.counts);
  return hipGetLastError();
}

hipError_t launch_scan(const MsmPlan& p, const MsmWorkspace& ws, hipStream_t stream) {
  hipLaunchKernelGGL(k_scan, dim3(p.win_end - p.win_begin), dim3(1024), 0, stream, ws.counts, ws.starts,
                     ws.cursor, p);
  return hipGetLastError();
}

hipError_t launch_scatter(const MsmPlan& p, const MsmWorkspace& ws, const void* d_scalars, hipStream_t stream) {
  hipLaunchKernelGGL(k_scatter, dim3(cdiv(p.n, kBlock)), dim3(kBlock), 0, stream,
                     reinterpret_cast<const uint4*>(d_scalars), p, ws.cursor, ws.sorted);
  return hipGetLastError();
}

hipError_t launch_accumulate(const MsmPlan& p, const MsmWorkspace& ws, const void* d_points, hipStream_t stream) {
  const u32 nw = p.win_end - p.win_begin;
  const u32 ntasks = nw * p.B;
  hipLaunchKernelGGL(k_accumulate, dim3(cdiv(ntasks, kBlock)), dim3(kBlock), 0, stream,
                     reinterpret_cast<const uint4*>(d_points), ws.sorted, ws.starts, ws.counts, ws.buckets, ntasks);
  return hipGetLastError();
}

hipError_t launch_bucket_reduce(const MsmPlan& p, const MsmWorkspace& ws, hipStream_t stream) {
  const u32 nw = p.win_end - p.win_begin;
  hipLaunchKernelGGL(k_bucket_reduce, dim3(cdiv((u64)nw * p.nseg, kBlock)), dim3(kBlock), 0, stream, ws.buckets,
                     ws.partials, p, nw);
  return hipGetLastError();
}

hipError_t launch_window_sum(const MsmPlan& p, const MsmWorkspace& ws, hipStream_t stream) {
  const u32 nw = p.win_end - p.win_begin;
  hipLaunchKernelGGL(k_window_sum, dim3(nw), dim3(kBlock), 0, stream, ws.partials, ws.winsums, p);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// Primitive self-test (curdle_selftest_op)
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock, 2) k_selftest(int op, const u32* __restrict__ in, size_t n, u32* __restrict__ out) {
  size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  if (op <= 3) {
    Fp a, b, r;
    for (int k = 0; k < 12; k++) {
      a.l[k] = in[i * 24 + k];
      b.l[k] = in[i * 24 + 12 + k];
    }
    if (op == 0) fp_mul(r, a, b);
    else if (op == 1) fp_add(r, a, b);
    else if (op == 2) fp_sub(r, a, b);
    else fp_sqr(r, a);
    for (int k = 0; k < 12; k++) out[i * 12 + k] = r.l[k];
  } else if (op == 4) {
    Fr a, r;
    for (int k = 0; k < 8; k++) a.l[k] = in[i * 16 + k];
    f_from_mont<FrParams>(r, a);
    for (int k = 0; k < 8; k++) out[i * 8 + k] = r.l[k];
  } else {
    G1XYZZ acc, b;
    const u32* src = in + i * 96;
    u32* a32 = reinterpret_cast<u32*>(&acc);
    u32* b32 = reinterpret_cast<u32*>(&b);
    for (int k = 0; k < 48; k++) {
      a32[k] = src[k];
      b32[k] = src[48 + k];
    }
    if (op == 5) {
      if (!(f_is_zero(b.x) & f_is_zero(b.y))) g1_madd(acc, b.x, b.y);
    } else if (op == 6) {
      g1_add(acc, b);
    } else {
      g1_dbl(acc);
    }
    for (int k = 0; k < 48; k++) out[i * 48 + k] = a32[k];
  }
}

hipError_t launch_selftest(int op, const uint32_t* d_in, size_t n, uint32_t* d_out, hipStream_t stream) {
  hipLaunchKernelGGL(k_selftest, dim3(cdiv(n, kBlock)), dim3(kBlock), 0, stream, op, d_in, n, d_out);
  return hipGetLastError();
}

}  // namespace curdle
