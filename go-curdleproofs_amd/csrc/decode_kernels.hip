// Batched decoding of gnark's compressed G1 encoding on the GPU: the input format on
// either side of the MSM path (proof points, Whisk trackers arrive as 48-byte strings and
// become MSM bases).  One lane per point:
//   parse big-endian x and the three flag bits -> x < p?  ->  y = (x^3 + 4)^((p+1)/4)
//   -> y^2 == x^3 + 4?  (on the curve)  -> pick the root the sign flag asks for
//   -> [z^2] phi(P) + P == inf?  (in the prime-order subgroup; gnark's Decoder / SetBytes
//      check this for every point, /root/reference/curdleproof.go:322, whisk/types.go:85-95)
//   -> affine (x, y) in gnark's in-memory layout, ready to be an MSM base.
// About 570 field products for the square root and 1,300 for the subgroup test per point;
// the same lazily-reduced 14 x 28-bit arithmetic as the MSM kernels (fp28.h).
#include <hip/hip_runtime.h>
#include <stdlib.h>

#include "../../include/curdle_msm.h"
#include "fp28.h"
#include "quad28.h"
#include "msm_kernels.h"
#include "../host/knobs.h"

namespace curdle {
// Up to this many lanes the latency-bound kernels of this file run on quads (four lanes per
// point, quad28.h); beyond it on one lane per point.  knob QUAD_MAX_LANES overrides (tuning; host/knobs.h).
static inline uint64_t quad_max_lanes() {
  return knobs::is_set(knobs::QUAD_MAX_LANES) ? (uint64_t)knobs::get(knobs::QUAD_MAX_LANES) : (uint64_t)131072;
}


using d28::F28;
using d28::X28;

static constexpr int kBlock = 256;

namespace {

#define CURDLE_DEC_TABLE28(name, ...)                    \
  __device__ __forceinline__ u32 name(int i) {           \
    constexpr u32 t[d28::N] = {__VA_ARGS__};             \
    return t[i];                                         \
  }
// 2^784 mod p: a Montgomery product with it maps a canonical residue to internal form
CURDLE_DEC_TABLE28(kCanonToInt, 0x10370edu, 0x6d1c345u, 0xe243d62u, 0xec45c53u, 0x3b1d65au, 0x093317du, 0xb4f36a0u,
                   0x5d74088u, 0xc10ea72u, 0x865d118u, 0x7320a75u, 0xfd5cd50u, 0xcc8a759u, 0x000c8d4u)
// 4 in internal form (the curve constant b)
CURDLE_DEC_TABLE28(kFour, 0xd1ff2e0u, 0x6000000u, 0x00ac467u, 0x3379b48u, 0x1c84b80u, 0x0e88243u, 0x0dd9a7eu,
                   0x683dcf8u, 0x6c26d0bu, 0x4a5eec2u, 0x457663cu, 0x04b29f1u, 0x967f3e8u, 0x0015de9u)
// beta in internal form: the cube root of unity with phi(x, y) = (beta x, y) = [z^2 - 1](x, y) on G1
CURDLE_DEC_TABLE28(kBeta, 0x2421b59u, 0xbee4867u, 0x1d31002u, 0x4760184u, 0x4cc5086u, 0xc76dc00u, 0xaae891bu,
                   0xac70ad2u, 0xfe377c4u, 0xe4686b8u, 0x5ed1568u, 0x8f5a180u, 0x02b5c1fu, 0x000d1a4u)
#undef CURDLE_DEC_TABLE28

__device__ __forceinline__ u32 kSqrtExp(int i) {  // (p + 1) / 4, 379 bits
  constexpr u32 t[12] = {0xffffeaabu, 0xee7fbfffu, 0xac54ffffu, 0x07aaffffu, 0x3dac3d89u, 0xd9cc34a8u,
                         0x3ce144afu, 0xd91dd2e1u, 0x90d2eb35u, 0x92c6e9edu, 0x8e5ff9a6u, 0x0680447au};
  return t[i];
}
__device__ __forceinline__ u32 kHalfP(int i) {  // (p - 1) / 2
  constexpr u32 t[12] = {0xffffd555u, 0xdcff7fffu, 0x58a9ffffu, 0x0f55ffffu, 0x7b587b12u, 0xb3986950u,
                         0x79c2895fu, 0xb23ba5c2u, 0x21a5d66bu, 0x258dd3dbu, 0x1cbff34du, 0x0d0088f5u};
  return t[i];
}
__device__ __forceinline__ u32 kP32(int i) {
  constexpr u32 t[12] = {0xffffaaabu, 0xb9feffffu, 0xb153ffffu, 0x1eabfffeu, 0xf6b0f624u, 0x6730d2a0u,
                         0xf38512bfu, 0x64774b84u, 0x434bacd7u, 0x4b1ba7b6u, 0x397fe69au, 0x1a0111eau};
  return t[i];
}

// a > b for 12-limb little-endian integers
template <class FA, class FB>
__device__ __forceinline__ int cmp12(FA a, FB b) {
  int r = 0;
#pragma unroll
  for (int i = 0; i < 12; i++) {
    const u32 x = a(i), y = b(i);
    r = x > y ? 1 : (x < y ? -1 : r);  // higher limbs decide last
  }
  return r;
}

// internal (< 32p) -> canonical residue as 12 saturated limbs
__device__ __forceinline__ void to_canonical(u32* w, const F28& a) {
  F28 one, t;
  d28::set_zero(one);
  one.l[0] = 1;
  d28::mul(t, a, one);
  d28::canonical_lt2p(t);
  d28::pack(w, t);
}

}  // namespace

// [z^2] phi(P) + P == inf for the point whose internal-form coordinates are parked in
// sh_x / sh_y [.][tid] (and passed in x, y); |z| = 0xd201000000010000, the sign cancels in
// z^2.  QUAD: the four lanes of a quad hold one coordinate each of every point (quad28.h).
template <bool QUAD>
__device__ __forceinline__ bool in_subgroup(F28& x, F28& y, u32 (*sh_x)[kBlock], u32 (*sh_y)[kBlock], u32 tid) {
  F28 c;
  F28 bx;
#pragma unroll
  for (int k = 0; k < d28::N; k++) c.l[k] = kBeta(k);
  d28::mul(bx, x, c);
  const unsigned long long zabs = 0xd201000000010000ull;
  if constexpr (QUAD) {
    // the point lives spread over the quad's four lanes (quad28.h)
    F28 q, acc;
    q28::from_affine(q, bx, y);
    acc = q;
    for (int bit = 62; bit >= 0; bit--) {
      q28::dbl(acc);
      if ((zabs >> bit) & 1ull) q28::add(acc, q);
    }
    q = acc;
    for (int bit = 62; bit >= 0; bit--) {
      q28::dbl(acc);
      if ((zabs >> bit) & 1ull) q28::add(acc, q);
    }
#pragma unroll
    for (int k = 0; k < d28::N; k++) {
      x.l[k] = sh_x[k][tid];
      y.l[k] = sh_y[k][tid];
    }
    q28::from_affine(q, x, y);
    q28::add(acc, q);
    return q28::is_inf(acc);
  } else {
    X28 q;
    q.x = bx;
    q.y = y;
    d28::set_one(q.zz);
    d28::set_one(q.zzz);
    X28 acc = q;
    // first multiplication: the addend is affine (mixed additions)
    for (int bit = 62; bit >= 0; bit--) {
      d28::dbl(acc);
      if ((zabs >> bit) & 1ull) d28::madd(acc, bx, y);
    }
    // second: the addend is the first result
    q = acc;
    for (int bit = 62; bit >= 0; bit--) {
      d28::dbl(acc);
      if ((zabs >> bit) & 1ull) d28::add(acc, q);
    }
#pragma unroll
    for (int k = 0; k < d28::N; k++) {
      x.l[k] = sh_x[k][tid];
      y.l[k] = sh_y[k][tid];
    }
    d28::madd(acc, x, y);
    return d28::is_inf(acc);
  }
}

// QUAD: four adjacent lanes per point.  They run the square root redundantly and share the
// point operations of the subgroup test (quad28.h: 3 and 4 product steps per doubling /
// addition instead of 9 and 14), which shortens the per-point chain from ~1,900 to ~1,050 products:
// the launch is latency-bound until tens of thousands of points, so small batches use it.
// SUB: with the subgroup test behind the square root (one-shot decodings too large for the
// two-kernel form); without it the kernel is the square-root half of that form and carries none
// of the point arithmetic's registers.
// The one-lane builds (batches beyond 32,768 points) at TWO waves per SIMD have 256 registers and spill
// 20-22 of them to scratch (96 bytes per lane).  Compiled for ONE wave per SIMD (-DCURDLE_LANE_WAVES=1)
// the allocator takes the accumulation registers too and nothing spills -- measured, round 4
// (profiles/r04_decode_one_lane.txt): 65,536 points 2.94 ms without spills against
// 2.97 with, 2^20 points 54.0 against 49.0 ms (subgroup test on), 30.4 against 29.6 (off): the second
// wave is worth more than the 22 scratch words cost, so the spilling build stays.
#ifndef CURDLE_LANE_WAVES
#define CURDLE_LANE_WAVES 2
#endif
template <bool QUAD, bool SUB>
__global__ void __launch_bounds__(kBlock, QUAD ? 2 : CURDLE_LANE_WAVES)
    k_g1_decompress(const uint8_t* __restrict__ in, u32 n, u32* __restrict__ out, uint8_t* __restrict__ status) {
  // x and y wait in LDS (limb-major: conflict-free) while the subgroup test runs: with the
  // two 56-register points of the scalar multiplication live there is no room for them.
  __shared__ u32 sh_x[d28::N][kBlock];
  __shared__ u32 sh_y[d28::N][kBlock];
  const u32 tid = threadIdx.x;
  const u32 lane = blockIdx.x * kBlock + tid;
  const u32 i = QUAD ? lane >> 2 : lane;
  const bool writer = !QUAD || (tid & 3u) == 0;
  if (i >= n) return;  // whole quads leave together
  const uint8_t* b = in + (size_t)i * 48;
  u32* o = out + (size_t)i * 24;
  u32 xw[12];
  const u32* b32 = reinterpret_cast<const u32*>(b);  // 48-byte records keep 4-byte alignment
#pragma unroll
  for (int k = 0; k < 12; k++) xw[k] = __builtin_bswap32(b32[11 - k]);
  const u32 flags = xw[11] >> 29;
  xw[11] &= 0x1fffffffu;
  auto fail = [&](uint8_t code) {
    if (!writer) return;
#pragma unroll
    for (int k = 0; k < 24; k++) o[k] = 0;
    status[i] = code;
  };
  if (!(flags & 4u)) return fail(CURDLE_DECODE_BAD_ENCODING);  // uncompressed form is not used on this wire
  if (flags & 2u) {                                            // infinity: every other bit must be clear
    u32 any = flags & 1u;
#pragma unroll
    for (int k = 0; k < 12; k++) any |= xw[k];
    return fail(any ? CURDLE_DECODE_BAD_ENCODING : CURDLE_DECODE_INFINITY);
  }
  if (cmp12([&](int k) { return xw[k]; }, [](int k) { return kP32(k); }) >= 0) return fail(CURDLE_DECODE_BAD_ENCODING);

  F28 x, t, c, rhs, y;
  d28::unpack(t, xw);
#pragma unroll
  for (int k = 0; k < d28::N; k++) c.l[k] = kCanonToInt(k);
  d28::mul(x, t, c);
  d28::sqr(t, x);
  d28::mul(t, t, x);
#pragma unroll
  for (int k = 0; k < d28::N; k++) c.l[k] = kFour(k);
  d28::add(rhs, t, c);  // x^3 + 4 < 4p

  // y = rhs^((p+1)/4), left to right; the top set bit is bit 378
  // 3-bit fixed windows over the 379-bit exponent (the same for every point, so the window
  // digit is wave-uniform and picks one of seven call sites): 378 squarings + 6 + ~110 products
  // instead of the 378 + ~190 of the bit-by-bit ladder
  {
    F28 t2, t3, t4, t5, t6, t7;
    d28::sqr(t2, rhs);
    d28::mul(t3, t2, rhs);
    d28::sqr(t4, t2);
    d28::mul(t5, t4, rhs);
    d28::sqr(t6, t3);
    d28::mul(t7, t6, rhs);
    y = rhs;  // bit 378
    for (int w = 125; w >= 0; w--) {
      // inlined: a call marshals 28 registers, ~4 % of a squaring when the wave is alone on its SIMD
      d28::sqr_inl(y, y);
      d28::sqr_inl(y, y);
      d28::sqr_inl(y, y);
      const int bit = 3 * w;
      u32 d = kSqrtExp(bit >> 5) >> (bit & 31);
      if ((bit & 31) > 29) d |= kSqrtExp((bit >> 5) + 1) << (32 - (bit & 31));
      switch (d & 7u) {
        case 1: d28::mul(y, y, rhs); break;
        case 2: d28::mul(y, y, t2); break;
        case 3: d28::mul(y, y, t3); break;
        case 4: d28::mul(y, y, t4); break;
        case 5: d28::mul(y, y, t5); break;
        case 6: d28::mul(y, y, t6); break;
        case 7: d28::mul(y, y, t7); break;
        default: break;
      }
    }
  }
  u32 yc[12], want[12], got[12];
  d28::sqr(t, y);
  to_canonical(got, t);
  to_canonical(want, rhs);
  u32 diff = 0;
#pragma unroll
  for (int k = 0; k < 12; k++) diff |= got[k] ^ want[k];
  if (diff) return fail(CURDLE_DECODE_NOT_ON_CURVE);

  to_canonical(yc, y);
  const bool larger = cmp12([&](int k) { return yc[k]; }, [](int k) { return kHalfP(k); }) > 0;
  if (larger != ((flags & 1u) != 0)) {
    F28 z;
    d28::set_zero(z);
    d28::sub<4>(y, z, y);  // 4p - y
  }
#pragma unroll
  for (int k = 0; k < d28::N; k++) {
    sh_x[k][tid] = x.l[k];
    sh_y[k][tid] = y.l[k];
  }

  if constexpr (SUB) {
    if (!in_subgroup<QUAD>(x, y, sh_x, sh_y, tid)) return fail(CURDLE_DECODE_NOT_IN_SUBGROUP);
  }
#pragma unroll
  for (int k = 0; k < d28::N; k++) {
    x.l[k] = sh_x[k][tid];
    y.l[k] = sh_y[k][tid];
  }
  if (!writer) return;
  d28::to_gnark(o, x);
  d28::to_gnark(o + 12, y);
  status[i] = CURDLE_DECODE_OK;
}

// The subgroup test WITHOUT the square root, so that it can run beside the decoding kernel
// instead of behind it (the two ~0.5 ms chains per point were the whole latency of a small
// decoding).  For a record's x let w = x^3 + 4.  If w = s^2 is a non-zero square, the map
//     psi(X, Y) = (w X, w s Y)            (u = s in the usual (u^2 X, u^3 Y))
// is an isomorphism, over Fp, from E: Y^2 = X^3 + 4 onto E': Y^2 = X^3 + 4 w^3; it sends the
// decoded point (x, s) to P' = (w x, w^2), whose coordinates need no square root, and it
// commutes with phi(X, Y) = (beta X, Y).  The group law of a curve with a = 0 never uses b,
// so the SAME doubling / addition code decides [z^2] phi(P') + P' == inf on E', which holds iff
// [z^2] phi(P) + P == inf on E -- for (x, s) and equally for its negative (x, -s), whichever
// root the sign flag asks for.  If w is not a square the record is not on the curve: the
// decoding kernel says so and this kernel's verdict for it is never read.  (w = 0 cannot
// happen for x < p: the curve has odd order, so no point has y = 0.)
// Writes sub[i] = 1 (in the subgroup, or not a candidate: malformed / infinity records) or 0.
template <bool QUAD>
__global__ void __launch_bounds__(kBlock, QUAD ? 2 : CURDLE_LANE_WAVES)
    k_g1_subgroup_from_x(const uint8_t* __restrict__ in, u32 n, uint8_t* __restrict__ sub) {
  __shared__ u32 sh_x[d28::N][kBlock];
  __shared__ u32 sh_y[d28::N][kBlock];
  const u32 tid = threadIdx.x;
  const u32 lane = blockIdx.x * kBlock + tid;
  const u32 i = QUAD ? lane >> 2 : lane;
  const bool writer = !QUAD || (tid & 3u) == 0;
  if (i >= n) return;  // whole quads leave together
  u32 xw[12];
  const u32* b32 = reinterpret_cast<const u32*>(in + (size_t)i * 48);
#pragma unroll
  for (int k = 0; k < 12; k++) xw[k] = __builtin_bswap32(b32[11 - k]);
  const u32 flags = xw[11] >> 29;
  xw[11] &= 0x1fffffffu;
  // not a candidate (the decoding kernel reports these): uniform over a quad
  if (!(flags & 4u) || (flags & 2u) || cmp12([&](int k) { return xw[k]; }, [](int k) { return kP32(k); }) >= 0) {
    if (writer) sub[i] = 1;
    return;
  }
  F28 x, t, c, w, xp, yp;
  d28::unpack(t, xw);
#pragma unroll
  for (int k = 0; k < d28::N; k++) c.l[k] = kCanonToInt(k);
  d28::mul(x, t, c);
  d28::sqr(t, x);
  d28::mul(t, t, x);
#pragma unroll
  for (int k = 0; k < d28::N; k++) c.l[k] = kFour(k);
  d28::add(w, t, c);    // w = x^3 + 4 < 4p
  d28::mul(xp, w, x);   // P' = (w x, w^2)
  d28::sqr(yp, w);
#pragma unroll
  for (int k = 0; k < d28::N; k++) {
    sh_x[k][tid] = xp.l[k];
    sh_y[k][tid] = yp.l[k];
  }
  const bool ok = in_subgroup<QUAD>(xp, yp, sh_x, sh_y, tid);
  if (writer) sub[i] = ok ? 1 : 0;
}

hipError_t launch_g1_subgroup_from_bytes(const uint8_t* in, uint32_t n, uint8_t* sub, hipStream_t stream) {
  if (n == 0) return hipSuccess;
  if ((uint64_t)n * 4 <= quad_max_lanes())
    hipLaunchKernelGGL(k_g1_subgroup_from_x<true>, dim3((4 * n + kBlock - 1) / kBlock), dim3(kBlock), 0, stream, in, n, sub);
  else
    hipLaunchKernelGGL(k_g1_subgroup_from_x<false>, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, stream, in, n, sub);
  return hipGetLastError();
}

hipError_t launch_g1_decompress(const uint8_t* in, uint32_t n, int subgroup_check, uint32_t* out, uint8_t* status,
                                hipStream_t stream) {
  if (n == 0) return hipSuccess;
  // four lanes per point while even that is at most one round of the chip (2 waves per SIMD)
  if (subgroup_check && (uint64_t)n * 4 <= quad_max_lanes())
    hipLaunchKernelGGL((k_g1_decompress<true, true>), dim3((4 * n + kBlock - 1) / kBlock), dim3(kBlock), 0, stream, in, n, out,
                       status);
  else if (subgroup_check)
    hipLaunchKernelGGL((k_g1_decompress<false, true>), dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, stream, in, n, out,
                       status);
  else
    hipLaunchKernelGGL((k_g1_decompress<false, false>), dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, stream, in, n, out,
                       status);
  return hipGetLastError();
}

}  // namespace curdle
