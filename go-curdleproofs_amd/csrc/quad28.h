// Lane-distributed G1 points for the latency-bound phases of the MSM on gfx950.
//
// Why.  One XYZZ addition is 14 dependent field products (fp28.h); a wave that is alone
// on its SIMD issues a multiply-add every ~10 cycles, so on one lane that is ~25 us, and
// the bucket reduction / window combine / subgroup test / scalar multiplication are chains
// of 40..500 of them.  Whenever such a launch has far fewer lanes than the chip, FOUR
// adjacent lanes (a DPP quad) own ONE point between them:
//
//     lane 4q+0 holds X,  4q+1 holds Y,  4q+2 holds ZZ,  4q+3 holds ZZZ
//
// (one F28 = 14 VGPRs per point and lane instead of 56), and every step of the group law
// is ONE field product per lane -- the four independent products of the step -- with the
// operands moved between the lanes by quad_perm DPP moves (14 per field element, no LDS).
// An addition is 4 product steps instead of 14, a doubling 3 instead of 9.
//
// This replaces the first version of the quad kernels (fp28.h quad_add / quad_dbl in round
// 1), which kept four identical copies of every point and selected each lane's operands out
// of them: 56 VGPRs per point, ~800 B of scratch per lane and a loop body beyond the
// instruction cache made its product step 3.7x slower than the single-lane one
// (profiles/r02_quad_experiment.txt).
//
// Value bounds are those of fp28.h: stored X < 10p, Y < 6p, ZZ, ZZZ < 2p, limbs normalised;
// infinity <=> ZZ == 0 (all limbs; lane 2).
//
// Formulas: EFD "xyzz" add-2008-s and dbl-2008-s-1 (a = 0), as in fp28.h add() / dbl().
// Replaces, on the GPU, the G1Jac additions the reference gets from gnark-crypto inside
// (*G1Jac).MultiExp (go.mod:6; call sites in SURVEY.md section 8a).
#pragma once
#include "fp28.h"

namespace curdle {
namespace q28 {
// The products of the quad formulas are INLINED (d28::mul_inl / sqr_inl), not the out-of-line
// calls the one-lane formulas use: a call costs 28 register moves to marshal its operands, at
// ~5.5 cycles each when the wave is alone on its SIMD (the regime these kernels live in) -- 3-4 %
// of every product step.  Measured: bucket reduce of the verifier's MSM 0.158 -> 0.142 ms,
// k_combine 1.15 -> 0.93 ms, a 98-point decoding 0.65 -> 0.61 ms, and no kernel grew past the
// instruction cache (the largest, k_bucket_reduce_quad, has four addition sites and one doubling).
#define Q28_MUL(r, a, b) d28::mul_inl(r, a, b)
#define Q28_SQR(r, a) d28::sqr_inl(r, a)

using d28::F28;
using d28::N;
using d28::X28;

// Which coordinate this lane holds: 0 X, 1 Y, 2 ZZ, 3 ZZZ.  Blocks are multiples of four
// lanes and quads are aligned, so the low bits of threadIdx.x are the lane's rank.
__device__ __forceinline__ u32 role() { return threadIdx.x & 3u; }

// dst on lane i of every quad = src on lane S_i of the same quad.
template <int S0, int S1, int S2, int S3>
__device__ __forceinline__ void perm(F28& dst, const F28& src) {
  constexpr int ctrl = S0 | (S1 << 2) | (S2 << 4) | (S3 << 6);
#pragma unroll
  for (int i = 0; i < N; i++) {
    u32 v = (u32)__builtin_amdgcn_update_dpp((int)src.l[i], (int)src.l[i], ctrl, 0xf, 0xf, true);
    // Keep the move a move.  Left alone, LLVM's DPP combine folds it into a following
    // subtraction as `v_subrev_u32_dpp d, v, v` (both sources the same register), and on
    // gfx950 that returned dpp(v) - v where own - dpp(v) was meant: every Y3 of the addition
    // came out negated (found with selftest op 8, tests/test_msm_gpu.py; ROCm 7.2.0).
    asm volatile("" : "+v"(v));
    dst.l[i] = v;
  }
}
template <int S>
__device__ __forceinline__ void bcast(F28& dst, const F28& src) {
  perm<S, S, S, S>(dst, src);
}
// The predicate of lane S, for every lane of the quad.
template <int S>
__device__ __forceinline__ bool flag(bool mine) {
  const int v = mine ? 1 : 0;
  return __builtin_amdgcn_update_dpp(v, v, S * 0x55, 0xf, 0xf, true) != 0;
}
__device__ __forceinline__ void sel(F28& dst, bool c, const F28& a, const F28& b) {
#pragma unroll
  for (int i = 0; i < N; i++) dst.l[i] = c ? a.l[i] : b.l[i];
}

__device__ __forceinline__ void set_inf(F28& a) {
  if (role() < 2)
    d28::set_one(a);
  else
    d28::set_zero(a);
}
__device__ __forceinline__ bool is_inf(const F28& a) { return flag<2>(d28::all_zero(a)); }

// a = 2a.  Three product steps.  Infinity stays infinity (ZZ3 = V * 0, ZZZ3 = W * 0).
__device__ __forceinline__ void dbl(F28& a) {
  const u32 r = role();
  F28 u, A, B, m1, v, xx, m, m2, s, w, mm, x3, t, z;
  d28::dbl_raw(u, a);                   // lane 1: U = 2Y < 12p, limbs < 2^29
  sel(A, r == 1, u, a);
  Q28_SQR(m1, A);                      // XX | V = U^2 | - | -
  bcast<1>(v, m1);
  bcast<0>(xx, m1);
  d28::triple_raw(m, xx);               // M = 3 XX < 6p
  sel(A, r == 1, u, a);
  sel(A, r == 3, m, A);
  sel(B, r == 3, m, v);
  Q28_MUL(m2, A, B);                   // S = X V | W = U V | ZZ3 = ZZ V | MM = M^2
  bcast<0>(s, m2);
  bcast<1>(w, m2);
  bcast<3>(mm, m2);
  d28::set_zero(z);
  d28::x3_fused(x3, mm, z, s);          // X3 = M^2 - 2S < 10p
  d28::sub_raw<16>(t, s, x3);           // S - X3 < 18p
  sel(A, r == 0, m, w);
  sel(B, r == 0, t, a);
  Q28_MUL(m1, A, B);                   // M (S - X3) | W Y | - | ZZZ3 = W ZZZ
  bcast<0>(t, m1);
  d28::sub<4>(t, t, m1);                // lane 1: Y3 = M (S - X3) - W Y + 4p < 6p
  sel(a, r == 0, x3, t);
  sel(a, r == 2, m2, a);
  sel(a, r == 3, m1, a);
}

// The exceptional branch of an addition (equal points) is rare and must not be inlined
// into every addition site: one out-of-line copy per code object, operand in VGPRs.
__device__ __noinline__ inline d28::u32x14 dbl_call(d28::u32x14 a) {
  F28 x;
#pragma unroll
  for (int i = 0; i < N; i++) x.l[i] = a[i];
  dbl(x);
  d28::u32x14 o;
#pragma unroll
  for (int i = 0; i < N; i++) o[i] = x.l[i];
  return o;
}
__device__ __forceinline__ void dbl_outofline(F28& a) {
  d28::u32x14 x;
#pragma unroll
  for (int i = 0; i < N; i++) x[i] = a.l[i];
  d28::u32x14 o = dbl_call(x);
#pragma unroll
  for (int i = 0; i < N; i++) a.l[i] = o[i];
}

// a += b.  Four product steps; b may be any stored point (an affine one has ZZ = ZZZ = one).
__device__ __forceinline__ void add(F28& a, const F28& b) {
  const u32 r = role();
  if (is_inf(b)) return;                // uniform over the quad
  if (is_inf(a)) {
    a = b;
    return;
  }
  F28 o, m1, d, A, B, m2, pp, t, m3, ppp, q, rr, x3, m4;
  perm<2, 3, 0, 1>(o, b);               // ZZ2 | ZZZ2 | X2 | Y2
  Q28_MUL(m1, a, o);                   // U1 = X1 ZZ2 | S1 = Y1 ZZZ2 | U2 = ZZ1 X2 | S2 = ZZZ1 Y2
  perm<2, 3, 2, 3>(t, m1);
  d28::sub_raw<4>(d, t, m1);            // P = U2 - U1 | R = S2 - S1 | - | -      (< 6p)
  sel(A, r < 2, d, a);
  sel(B, r < 2, d, b);
  Q28_MUL(m2, A, B);                   // PP | RR | ZZ1 ZZ2 | ZZZ1 ZZZ2
  if (flag<0>(d28::is_zero_lt2p(m2))) { // P == 0 mod p: the same x
    if (flag<1>(d28::is_zero_lt2p(m2)))
      dbl_outofline(a);                 // the same point
    else
      set_inf(a);                       // opposite points
    return;
  }
  bcast<0>(pp, m2);
  bcast<0>(t, m1);                      // U1
  sel(A, r == 0, d, m2);
  sel(A, r == 1, t, A);
  Q28_MUL(m3, A, pp);                  // PPP = P PP | Q = U1 PP | ZZ3 = ZZ1 ZZ2 PP | -
  bcast<0>(ppp, m3);
  bcast<1>(q, m3);
  bcast<1>(rr, m2);
  d28::x3_fused(x3, rr, ppp, q);        // X3 = R^2 - PPP - 2Q < 10p
  bcast<1>(t, m1);                      // S1
  sel(A, r == 0, t, m2);
  sel(A, r == 1, d, A);                 // S1 | R | - | ZZZ1 ZZZ2
  d28::sub_raw<16>(t, q, x3);           // Q - X3 < 18p
  sel(B, r == 1, t, ppp);
  Q28_MUL(m4, A, B);                   // S1 PPP | R (Q - X3) | - | ZZZ3
  bcast<0>(t, m4);
  d28::sub<4>(t, m4, t);                // lane 1: Y3 = R (Q - X3) - S1 PPP + 4p < 6p
  sel(a, r == 0, x3, t);
  sel(a, r == 2, m3, a);
  sel(a, r == 3, m4, a);
}

// r = k * p, small k: left-to-right double-and-add from bit `top` down (top >= the index of
// k's highest set bit; the doublings above it double infinity).  Every quad of a wave runs
// the same number of steps when `top` is wave-uniform.
__device__ __forceinline__ void mul_small(F28& r, const F28& p, u32 k, int top) {
  set_inf(r);
  for (int bit = top; bit >= 0; bit--) {
    dbl(r);
    if ((k >> bit) & 1u) add(r, p);
  }
}

// Memory: a stored X28 is x | y | zz | zzz, 56 bytes each (8-byte aligned): every lane of
// the quad moves its own coordinate, 7 x 8 bytes.
__device__ __forceinline__ void load(F28& c, const X28* src) {
  const uint2* s = reinterpret_cast<const uint2*>(reinterpret_cast<const char*>(src) + 56u * role());
#pragma unroll
  for (int i = 0; i < N / 2; i++) {
    uint2 v = s[i];
    c.l[2 * i] = v.x;
    c.l[2 * i + 1] = v.y;
  }
}
__device__ __forceinline__ void store(X28* dst, const F28& c) {
  uint2* d = reinterpret_cast<uint2*>(reinterpret_cast<char*>(dst) + 56u * role());
#pragma unroll
  for (int i = 0; i < N / 2; i++) d[i] = make_uint2(c.l[2 * i], c.l[2 * i + 1]);
}
// The quad's point `off` quads further up the wave (off < 16).
__device__ __forceinline__ void shfl_down(F28& dst, const F28& src, u32 off) {
#pragma unroll
  for (int i = 0; i < N; i++) dst.l[i] = __shfl_down(src.l[i], off * 4, 64);
}

// ... and the one `off` quads further down (off < 16).
__device__ __forceinline__ void shfl_up(F28& dst, const F28& src, u32 off) {
#pragma unroll
  for (int i = 0; i < N; i++) dst.l[i] = __shfl_up(src.l[i], off * 4, 64);
}

// Replicated X28 (every lane holds the whole point) <-> distributed.
__device__ __forceinline__ void from_x28(F28& c, const X28& p) {
  const u32 r = role();
  sel(c, r == 0, p.x, p.y);
  sel(c, r == 2, p.zz, c);
  sel(c, r == 3, p.zzz, c);
}
__device__ __forceinline__ void to_x28(X28& p, const F28& c) {
  bcast<0>(p.x, c);
  bcast<1>(p.y, c);
  bcast<2>(p.zz, c);
  bcast<3>(p.zzz, c);
}
// Affine point (x, y), not infinity -> distributed (ZZ = ZZZ = one).
__device__ __forceinline__ void from_affine(F28& c, const F28& x, const F28& y) {
  const u32 r = role();
  F28 one;
  d28::set_one(one);
  sel(c, r == 0, x, y);
  sel(c, r >= 2, one, c);
}

#undef Q28_MUL
#undef Q28_SQR
}  // namespace q28
}  // namespace curdle
