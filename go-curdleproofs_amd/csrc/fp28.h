// Device-internal BLS12-381 base field and G1 arithmetic for gfx950:
// 14 limbs of 28 bits, Montgomery radix R' = 2^392, lazily reduced.
//
// Why not the 12 x 32-bit limbs gnark stores (bls12_381.h)?  On gfx950 the only
// wide multiplier is v_mad_u64_u32 (32x32+64).  With saturated 32-bit limbs every
// product needs its carry captured (v_addc_co_u32), and the pair issues every
// 9.25-10.4 cycles per SIMD; the multiply alone issues every 4.9
// (profiles/r01_ubench_valu.txt).  With 28-bit limbs a 64-bit column accumulator
// holds all <= 29 products of a column without overflow, so a field product is
// 406 bare multiply-adds (315 for a square) plus a shift and a mask per column:
// measured 1.33x the throughput of the 32-bit form.  R' = 2^392 leaves 11 spare
// bits over p (381 bits), which buys lazy reduction: products come out below 2p
// with NO conditional subtraction as long as the operands are below 32p, and
// additions / subtractions only propagate carries.
//
// Value bounds (multiples of p) are part of every function's contract below.
// Limbs 0..12 are always < 2^28 after `norm`; limb 13 carries the excess.
//
// External (gnark, R = 2^384, 12 x u32 saturated, canonical) <-> internal
// conversions are one Montgomery product by a constant each way; the MSM converts
// every input point once (k_convert_points) and every window sum once.
//
// Device-only header.  Replaces, on the GPU, the fp.Element arithmetic the
// reference gets from gnark-crypto (go.mod:6) inside (*G1Jac).MultiExp.
#pragma once
#include <hip/hip_runtime.h>

#include "bls12_381.h"

namespace curdle {
namespace d28 {

static constexpr int N = 14;
static constexpr u32 MASK = 0x0fffffffu;
static constexpr u32 N0 = 0x0ffcfffdu;  // -p^-1 mod 2^28

struct F28 {
  u32 l[N];
};
// XYZZ point, x = X/ZZ, y = Y/ZZZ; infinity <=> ZZ == 0 (all limbs zero).
// Invariant for stored points: X, Y < 10p; ZZ, ZZZ < 2p; limbs normalised.
struct X28 {
  F28 x, y, zz, zzz;
};
// Affine point in internal form, x, y < 2p; (0,0) = infinity.
struct A28 {
  F28 x, y;
};

#define CURDLE_D28_TABLE(name, ...)                      \
  __device__ __forceinline__ u32 name(int i) {           \
    constexpr u32 t[N] = {__VA_ARGS__};                  \
    return t[i];                                         \
  }
// p
CURDLE_D28_TABLE(kP, 0xfffaaabu, 0xfefffffu, 0x3ffffb9u, 0xfffeb15u, 0x6241eabu, 0xa0f6b0fu, 0xf6730d2u, 0xf38512bu,
                 0x4774b84u, 0x4bacd76u, 0xba7b643u, 0xe69a4b1u, 0x1ea397fu, 0x001a011u)
// 2^392 mod p  (one)
CURDLE_D28_TABLE(kOne, 0x347fcb8u, 0xd800000u, 0x002b119u, 0x0cde6d2u, 0xc7212e0u, 0x83a2090u, 0x037669fu, 0xda0f73eu,
                 0x9b09b42u, 0x1297bb0u, 0x515d98fu, 0x012ca7cu, 0x659fcfau, 0x000577au)
// 2^400 mod p: mul by it maps gnark form (x*2^384) to internal form (x*2^392)
CURDLE_D28_TABLE(kToInt, 0x80e6299u, 0x3500034u, 0xeb12856u, 0xdeb2699u, 0xc988670u, 0x4ef6697u, 0x70983e8u, 0xa4e6fe9u,
                 0x3e8a053u, 0xecf271eu, 0xc20d323u, 0x6eb6385u, 0x47f1286u, 0x00156dau)
// beta * 2^392 mod p, beta the cube root of unity with (beta x, y) = [lambda](x, y), lambda = z^2 - 1
CURDLE_D28_TABLE(kBeta, 0x2421b59u, 0xbee4867u, 0x1d31002u, 0x4760184u, 0x4cc5086u, 0xc76dc00u, 0xaae891bu, 0xac70ad2u,
                 0xfe377c4u, 0xe4686b8u, 0x5ed1568u, 0x8f5a180u, 0x02b5c1fu, 0x000d1a4u)
// 2^384 mod p: mul by it maps internal form back to gnark form
CURDLE_D28_TABLE(kToExt, 0x002fffdu, 0x0900000u, 0xc000276u, 0x000bc40u, 0x8baebf4u, 0x5753c75u, 0x55f4898u, 0x7052574u,
                 0x7ce5853u, 0x56ec6d7u, 0x71a97a2u, 0xe4935c0u, 0xec3fa80u, 0x0015f65u)
// 2^388 mod p and 2^390 mod p: the X and Y coordinates of an MSM result leave through these instead of
// kToExt (the bucket sums live on an isomorphic curve, see from_gnark_iso)
CURDLE_D28_TABLE(kToExtX16, 0x0345521u, 0x9d00000u, 0xc002aeeu, 0x00cd3f7u, 0xbd93084u, 0x48b5790u, 0xdb70ed3u, 0xa763809u,
                 0x2d6af76u, 0x96ffe76u, 0xa2538bau, 0x935ff00u, 0x35abc8fu, 0x000d580u)
CURDLE_D28_TABLE(kToExtY64, 0x0d1ff2eu, 0x7600000u, 0x800ac46u, 0x03379b4u, 0x31c84b8u, 0xe0e8824u, 0x80dd9a7u, 0xb683dcfu,
                 0x26c26d0u, 0xc4a5eecu, 0x1457663u, 0x804b29fu, 0x9967f3eu, 0x00015deu)
// K*p in borrow-proof limb form: limbs 0..12 >= 2^28 - 1, limb 13 = top(K*p) - 1,
// so (a + K - b) is non-negative limb by limb for normalised b < (K-1)p.
CURDLE_D28_TABLE(kK4, 0x1ffeaaacu, 0x1fbffffeu, 0x1ffffee6u, 0x1fffac53u, 0x18907aaeu, 0x183dac3cu, 0x1d9cc349u,
                 0x1ce144aeu, 0x11dd2e12u, 0x12eb35d8u, 0x1e9ed90cu, 0x19a692c5u, 0x17a8e5feu, 0x0068043u)
CURDLE_D28_TABLE(kK8, 0x1ffd5558u, 0x1f7ffffeu, 0x1ffffdceu, 0x1fff58a8u, 0x1120f55eu, 0x107b587au, 0x1b398694u,
                 0x19c2895eu, 0x13ba5c26u, 0x15d66bb1u, 0x1d3db219u, 0x134d258cu, 0x1f51cbfeu, 0x00d0087u)
CURDLE_D28_TABLE(kK16, 0x1ffaaab0u, 0x1efffffeu, 0x1ffffb9eu, 0x1ffeb152u, 0x1241eabeu, 0x10f6b0f5u, 0x16730d29u,
                 0x138512beu, 0x1774b84eu, 0x1bacd763u, 0x1a7b6433u, 0x169a4b1au, 0x1ea397fdu, 0x01a0110u)
// 8p with limbs 0..12 >= 2^30 - 4: absorbs an unnormalised subtrahend with limbs < 2^30.
CURDLE_D28_TABLE(kK8B, 0x4ffd5558u, 0x4f7ffffbu, 0x4ffffdcbu, 0x4fff58a5u, 0x4120f55bu, 0x407b5877u, 0x4b398691u,
                 0x49c2895bu, 0x43ba5c23u, 0x45d66baeu, 0x4d3db216u, 0x434d2589u, 0x4f51cbfbu, 0x000d0084u)
#undef CURDLE_D28_TABLE

#ifdef CURDLE_MAC_PLAIN
// Experiment switch (round 4): the column blocks in plain C instead of the generated inline asm --
// the same v_mad_u64_u32 count, none of the wait states hipcc pads after every asm statement, but
// ~210 more v_lshl_add_u64 per mixed addition.  profiles/r04_mac_plain_vs_asm.txt has the timing.
template <int K>
__device__ __forceinline__ void m28v(u64& acc, const u32* a, const u32* b) {
#pragma unroll
  for (int i = 0; i < K; i++) acc += (u64)a[i] * (u64)b[-i];
}
template <int K>
__device__ __forceinline__ void m28s(u64& acc, const u32* a, const u32* b) {
#pragma unroll
  for (int i = 0; i < K; i++) acc += (u64)a[i] * (u64)b[-i];
}
template <int K>
__device__ __forceinline__ void m28vset(u64& acc, const u32* a, const u32* b) {
  acc = 0;
  m28v<K>(acc, a, b);
}
#else
#include "mac28_gfx950.inc"
#ifdef CURDLE_MAC_NO_SET  // A/B: the accumulator zeroed by a move in front of the chain, as before round 4
#define m28vset m28vset_by_move
template <int K>
__device__ __forceinline__ void m28vset_by_move(u64& acc, const u32* a, const u32* b) {
  acc = 0;
  m28v<K>(acc, a, b);
}
#endif
#endif

struct PTable {
  u32 v[N];
  __device__ __forceinline__ PTable() {
#pragma unroll
    for (int i = 0; i < N; i++) v[i] = kP(i);
  }
};

// ---------------------------------------------------------------------------
// Montgomery product, r = a*b / 2^392 mod p (not canonical).
// Requires limbs(a), limbs(b) < 2^30 and a*b < 2^392 * p (e.g. a, b < 32p).
// Ensures r < 2p, limbs normalised.
// ---------------------------------------------------------------------------
__device__ __forceinline__ void mul_inl(F28& r, const F28& a, const F28& b) {
  const PTable P;
  u32 m[N];
  u32 t[N];
  u64 acc;
  static_for<0, N>([&](auto kc) {
    constexpr int k = decltype(kc)::value;
    if constexpr (k == 0)
      m28vset<1>(acc, &a.l[0], &b.l[0]);  // the chain's first product sets the accumulator (no v_mov_b64 acc, 0)
    else
      m28v<k + 1>(acc, &a.l[0], &b.l[k]);
    if constexpr (k > 0) m28s<k>(acc, &m[0], &P.v[k]);
    m[k] = ((u32)acc * N0) & MASK;
    m28s<1>(acc, &m[k], &P.v[0]);
    acc >>= 28;
  });
  static_for<N, 2 * N - 1>([&](auto kc) {
    constexpr int k = decltype(kc)::value;
    constexpr int i0 = k - N + 1;
    m28v<N - i0>(acc, &a.l[i0], &b.l[N - 1]);
    m28s<N - i0>(acc, &m[i0], &P.v[N - 1]);
    t[k - N] = (u32)acc & MASK;
    acc >>= 28;
  });
  t[N - 1] = (u32)acc;
#pragma unroll
  for (int i = 0; i < N; i++) r.l[i] = t[i];
}

// Square: cross products once, doubled by a 64-bit shift-add (v_lshl_add_u64).
// Same contract as mul_inl(r, a, a).
__device__ __forceinline__ void sqr_inl(F28& r, const F28& a) {
  const PTable P;
  u32 m[N];
  u32 t[N];
  u64 acc;
  static_for<0, 2 * N - 1>([&](auto kc) {
    constexpr int k = decltype(kc)::value;
    constexpr int lo = k < N ? 0 : k - N + 1;   // smallest i with k - i <= N-1
    constexpr int pairs = (k + 1) / 2 - lo;     // i in [lo, k/2) with i < k - i
    if constexpr (pairs > 0) {
      u64 cross;
      m28vset<pairs>(cross, &a.l[lo], &a.l[k - lo]);
      acc += cross << 1;
    }
    if constexpr (k == 0)
      m28vset<1>(acc, &a.l[0], &a.l[0]);
    else if constexpr (k % 2 == 0)
      m28v<1>(acc, &a.l[k / 2], &a.l[k / 2]);
    if constexpr (k < N) {
      if constexpr (k > 0) m28s<k>(acc, &m[0], &P.v[k]);
      m[k] = ((u32)acc * N0) & MASK;
      m28s<1>(acc, &m[k], &P.v[0]);
    } else {
      constexpr int i0 = k - N + 1;
      m28s<N - i0>(acc, &m[i0], &P.v[N - 1]);
      t[k - N] = (u32)acc & MASK;
    }
    acc >>= 28;
  });
  t[N - 1] = (u32)acc;
#pragma unroll
  for (int i = 0; i < N; i++) r.l[i] = t[i];
}

// Sum of two products with ONE Montgomery reduction: r = (a*b + c*d) / 2^392 mod p.
// Saves the 210 multiply-adds of a second reduction wherever a formula subtracts
// two products (the Y3 of every XYZZ formula: the subtrahend's first factor is
// negated with a borrow-proof multiple of p beforehand).
// Requires limbs < 2^30 with (a, b) not both above 2^29.6 limbs, c limbs < 2^29.2,
// d normalised, and a*b + c*d < 2^392 * p.  Ensures r < 2p, limbs normalised.
// Inlined (four operands do not fit the register calling convention).
__device__ __forceinline__ void mul2_inl(F28& r, const F28& a, const F28& b, const F28& c, const F28& d) {
  const PTable P;
  u32 m[N];
  u32 t[N];
  u64 acc;
  static_for<0, N>([&](auto kc) {
    constexpr int k = decltype(kc)::value;
    if constexpr (k == 0)
      m28vset<1>(acc, &a.l[0], &b.l[0]);
    else
      m28v<k + 1>(acc, &a.l[0], &b.l[k]);
    m28v<k + 1>(acc, &c.l[0], &d.l[k]);
    if constexpr (k > 0) m28s<k>(acc, &m[0], &P.v[k]);
    m[k] = ((u32)acc * N0) & MASK;
    m28s<1>(acc, &m[k], &P.v[0]);
    acc >>= 28;
  });
  static_for<N, 2 * N - 1>([&](auto kc) {
    constexpr int k = decltype(kc)::value;
    constexpr int i0 = k - N + 1;
    m28v<N - i0>(acc, &a.l[i0], &b.l[N - 1]);
    m28v<N - i0>(acc, &c.l[i0], &d.l[N - 1]);
    m28s<N - i0>(acc, &m[i0], &P.v[N - 1]);
    t[k - N] = (u32)acc & MASK;
    acc >>= 28;
  });
  t[N - 1] = (u32)acc;
#pragma unroll
  for (int i = 0; i < N; i++) r.l[i] = t[i];
}

// Out-of-line entry points (one copy per code object, operands in VGPRs) for kernels with many
// product sites (the one-lane decoding / subgroup kernels, the exceptional branches): ~4 KB of
// code per inlined product adds up there.  A call costs 28 register moves to marshal its
// operands, so the kernels with one or few sites -- k_accumulate's mixed addition, the quad
// formulas, the square-root chain -- inline the products instead (mul_inl / sqr_inl).
typedef u32 u32x14 __attribute__((ext_vector_type(14)));
__device__ __noinline__ inline u32x14 mul_call(u32x14 a, u32x14 b) {
  F28 x, y, r;
#pragma unroll
  for (int i = 0; i < N; i++) {
    x.l[i] = a[i];
    y.l[i] = b[i];
  }
  mul_inl(r, x, y);
  u32x14 o;
#pragma unroll
  for (int i = 0; i < N; i++) o[i] = r.l[i];
  return o;
}
__device__ __noinline__ inline u32x14 sqr_call(u32x14 a) {
  F28 x, r;
#pragma unroll
  for (int i = 0; i < N; i++) x.l[i] = a[i];
  sqr_inl(r, x);
  u32x14 o;
#pragma unroll
  for (int i = 0; i < N; i++) o[i] = r.l[i];
  return o;
}
__device__ __forceinline__ void mul(F28& r, const F28& a, const F28& b) {
  u32x14 x, y;
#pragma unroll
  for (int i = 0; i < N; i++) {
    x[i] = a.l[i];
    y[i] = b.l[i];
  }
  u32x14 o = mul_call(x, y);
#pragma unroll
  for (int i = 0; i < N; i++) r.l[i] = o[i];
}
__device__ __forceinline__ void sqr(F28& r, const F28& a) {
  u32x14 x;
#pragma unroll
  for (int i = 0; i < N; i++) x[i] = a.l[i];
  u32x14 o = sqr_call(x);
#pragma unroll
  for (int i = 0; i < N; i++) r.l[i] = o[i];
}

// ---------------------------------------------------------------------------
// Linear operations: limb-wise, then one carry pass.  No reduction mod p.
// ---------------------------------------------------------------------------
__device__ __forceinline__ void norm(F28& a) {
  u32 c = 0;
#pragma unroll
  for (int i = 0; i < N - 1; i++) {
    u32 t = a.l[i] + c;
    a.l[i] = t & MASK;
    c = t >> 28;
  }
  a.l[N - 1] += c;
}
__device__ __forceinline__ void set_zero(F28& a) {
#pragma unroll
  for (int i = 0; i < N; i++) a.l[i] = 0;
}
__device__ __forceinline__ void set_one(F28& a) {
#pragma unroll
  for (int i = 0; i < N; i++) a.l[i] = kOne(i);
}
__device__ __forceinline__ bool all_zero(const F28& a) {
  u32 o = 0;
#pragma unroll
  for (int i = 0; i < N; i++) o |= a.l[i];
  return o == 0;
}
// a == 0 mod p for a normalised a < 2p (a product): a is 0 or p.
__device__ __forceinline__ bool is_zero_lt2p(const F28& a) {
  u32 o0 = 0, op = 0;
#pragma unroll
  for (int i = 0; i < N; i++) {
    o0 |= a.l[i];
    op |= a.l[i] ^ kP(i);
  }
  return (o0 == 0) | (op == 0);
}
// r = a + b.  value(r) = a + b.
__device__ __forceinline__ void add(F28& r, const F28& a, const F28& b) {
#pragma unroll
  for (int i = 0; i < N; i++) r.l[i] = a.l[i] + b.l[i];
  norm(r);
}
// r = a - b + K*p, K in {4, 8, 16}.  Requires b normalised and b < (K-1)p.
template <int K>
__device__ __forceinline__ void sub(F28& r, const F28& a, const F28& b) {
#pragma unroll
  for (int i = 0; i < N; i++) {
    const u32 k = K == 4 ? kK4(i) : (K == 8 ? kK8(i) : kK16(i));
    r.l[i] = a.l[i] + k - b.l[i];
  }
  norm(r);
}

// The same without the carry pass, for results that only feed products: limbs stay
// < 2^28 + 2^29 < 2^30, which the 64-bit column accumulators of mul / sqr absorb
// (14 * 2^59.2 + 14 * 2^56 < 2^64 even with both factors unnormalised).
template <int K>
__device__ __forceinline__ void sub_raw(F28& r, const F28& a, const F28& b) {
#pragma unroll
  for (int i = 0; i < N; i++) {
    const u32 k = K == 4 ? kK4(i) : (K == 8 ? kK8(i) : kK16(i));
    r.l[i] = a.l[i] + k - b.l[i];
  }
}
// r = a + a, no carry pass (limbs < 2^29).
__device__ __forceinline__ void dbl_raw(F28& r, const F28& a) {
#pragma unroll
  for (int i = 0; i < N; i++) r.l[i] = a.l[i] << 1;
}
// r = 3a, no carry pass (limbs < 3 * 2^28 < 2^30).
__device__ __forceinline__ void triple_raw(F28& r, const F28& a) {
#pragma unroll
  for (int i = 0; i < N; i++) r.l[i] = a.l[i] * 3u;
}
// r = rr - (c + 2q) + 8p in one pass and one carry pass (the X3 of every XYZZ
// formula): rr, c, q normalised and < 2p each  =>  r < 10p.
__device__ __forceinline__ void x3_fused(F28& r, const F28& rr, const F28& c, const F28& q) {
#pragma unroll
  for (int i = 0; i < N; i++) r.l[i] = rr.l[i] + kK8B(i) - c.l[i] - 2u * q.l[i];
  norm(r);
}

// ---------------------------------------------------------------------------
// External <-> internal
// ---------------------------------------------------------------------------
// 12 saturated 32-bit limbs -> 14 limbs of 28 bits (same integer).
__device__ __forceinline__ void unpack(F28& r, const u32* w) {
#pragma unroll
  for (int j = 0; j < N; j++) {
    const int bit = 28 * j;
    const int lo = bit >> 5, sh = bit & 31;
    u32 v = w[lo] >> sh;
    if (sh > 4 && lo + 1 < 12) v |= w[lo + 1] << (32 - sh);
    r.l[j] = v & MASK;
  }
}
// 14 x 28-bit limbs (value < 2^384) -> 12 saturated limbs.
__device__ __forceinline__ void pack(u32* w, const F28& a) {
#pragma unroll
  for (int k = 0; k < 12; k++) {
    const int bit = 32 * k;
    const int j = bit / 28, sh = bit % 28;
    u32 v = a.l[j] >> sh;
    if (j + 1 < N) v |= a.l[j + 1] << (28 - sh);
    if (28 - sh + 28 < 32 && j + 2 < N) v |= a.l[j + 2] << (56 - sh);
    w[k] = v;
  }
}
// Canonical representative of a normalised a < 2p.
__device__ __forceinline__ void canonical_lt2p(F28& a) {
  u32 d[N];
  int borrow = 0;
#pragma unroll
  for (int i = 0; i < N; i++) {
    int t = (int)a.l[i] - (int)kP(i) - borrow;
    borrow = t < 0;
    d[i] = (u32)t & (i < N - 1 ? MASK : 0xffffffffu);
  }
  if (!borrow) {
#pragma unroll
    for (int i = 0; i < N; i++) a.l[i] = d[i];
  }
}
// Limb i of p << S, normalised (S < 28).
template <int S>
__device__ __forceinline__ u32 kPshl(int i) {
  const u64 lo = i > 0 ? (u64)kP(i - 1) >> (28 - S) : 0;
  const u64 v = ((u64)kP(i) << S) | lo;
  return (u32)(i < N - 1 ? v & MASK : v);
}
// a -= 2^S p if a >= 2^S p (a normalised).
template <int S>
__device__ __forceinline__ void cond_sub_pshl(F28& a) {
  u32 d[N];
  int borrow = 0;
#pragma unroll
  for (int i = 0; i < N; i++) {
    int t = (int)a.l[i] - (int)kPshl<S>(i) - borrow;
    borrow = t < 0;
    d[i] = (u32)t & (i < N - 1 ? MASK : 0xffffffffu);
  }
  if (!borrow) {
#pragma unroll
    for (int i = 0; i < N; i++) a.l[i] = d[i];
  }
}
// 12 saturated limbs -> 14 limbs of 28 bits of (the integer << S), S <= 4.
template <int S>
__device__ __forceinline__ void unpack_shl(F28& r, const u32* w) {
#pragma unroll
  for (int j = 0; j < N; j++) {
    const int bit = 28 * j - S;
    u32 v;
    if (bit < 0) {
      v = w[0] << (-bit);
    } else {
      const int lo = bit >> 5, sh = bit & 31;
      v = w[lo] >> sh;
      if (sh > 4 && lo + 1 < 12) v |= w[lo + 1] << (32 - sh);
    }
    r.l[j] = v & MASK;
  }
}
// The bases of an MSM WITHOUT a product.  A gnark element x*2^384 read as an internal one is x/2^8 in
// internal form.  (x, y) -> (x / u^2, y / u^3) maps y^2 = x^3 + 4 onto y^2 = x^3 + 4 / u^6, and the XYZZ
// formulas for a = 0 (madd, add, dbl) never use the constant term, nor does phi (beta x, y): with u = 4
// the image of a base is (16 x / 2^8, 4 y / 2^8) = the gnark words shifted left by 4 and 2 bits.  Every
// bucket, segment and window sum of the MSM kernels lives on that curve; the three places a result leaves
// (write_point_quad, write_window_sum_quad, k_combine) map it back by the constants kToExtX16 / kToExtY64
// in the product they spend on the 2^392 -> 2^384 change anyway.  Conditional subtractions bring the
// shifted values under the bounds madd asks of an affine operand (x, y < 2p).
__device__ __forceinline__ void from_gnark_iso_x(F28& r, const u32* w) {
  unpack_shl<4>(r, w);  // < 16p
  cond_sub_pshl<3>(r);
  cond_sub_pshl<2>(r);
  cond_sub_pshl<1>(r);  // < 2p
}
__device__ __forceinline__ void from_gnark_iso_y(F28& r, const u32* w) {
  unpack_shl<2>(r, w);  // < 4p
  cond_sub_pshl<1>(r);  // < 2p
}
// gnark fp.Element (12 x u32, Montgomery 2^384, canonical) -> internal, < 2p.
__device__ __forceinline__ void from_gnark(F28& r, const u32* w) {
  F28 t, c;
  unpack(t, w);
#pragma unroll
  for (int i = 0; i < N; i++) c.l[i] = kToInt(i);
  mul(r, t, c);
}
// internal (< 32p) -> gnark fp.Element, canonical.
__device__ __forceinline__ void to_gnark(u32* w, const F28& a) {
  F28 t, c;
#pragma unroll
  for (int i = 0; i < N; i++) c.l[i] = kToExt(i);
  mul(t, a, c);
  canonical_lt2p(t);
  pack(w, t);
}

// Coordinate `role` (0 X, 1 Y, 2 ZZ, 3 ZZZ) of an MSM result, from the curve the bases were mapped to
// (from_gnark_iso) back to gnark form.
__device__ __forceinline__ void to_gnark_msm(u32* w, const F28& a, u32 role) {
  F28 t, c;
#pragma unroll
  for (int i = 0; i < N; i++) c.l[i] = role == 0 ? kToExtX16(i) : (role == 1 ? kToExtY64(i) : kToExt(i));
  mul(t, a, c);
  canonical_lt2p(t);
  pack(w, t);
}

// ---------------------------------------------------------------------------
// G1 on XYZZ coordinates (EFD "xyzz" formulas for a = 0), lazily reduced.
// ---------------------------------------------------------------------------
__device__ __forceinline__ bool is_inf(const X28& p) { return all_zero(p.zz); }
__device__ __forceinline__ bool affine_is_inf(const A28& p) { return all_zero(p.x) && all_zero(p.y); }
__device__ __forceinline__ void set_inf(X28& p) {
  set_one(p.x);
  set_one(p.y);
  set_zero(p.zz);
  set_zero(p.zzz);
}

// r = 2 * (x1, y1), affine input with x1 < 2p, y1 < 4p + (normalised), not
// infinity.  mdbl-2008-s-1.
__device__ __forceinline__ void dbl_affine(X28& r, const F28& x1, const F28& y1) {
  F28 u, v, w, s, m, t, s2;
  dbl_raw(u, y1);         // 2 y1 < 10p, limbs < 2^29
  sqr(v, u);              // V
  mul(w, u, v);           // W
  mul(s, x1, v);          // S
  sqr(t, x1);
  triple_raw(m, t);       // M = 3 x1^2 < 6p
  sqr(t, m);
  set_zero(s2);
  x3_fused(r.x, t, s2, s);  // X3 = M^2 - 2S < 10p
  sub_raw<16>(t, s, r.x);   // S - X3 < 18p
  mul(t, m, t);
  mul(u, w, y1);
  sub<4>(r.y, t, u);      // Y3 < 6p
  r.zz = v;
  r.zzz = w;
}

// p = 2p.  dbl-2008-s-1.  Infinity stays infinity (ZZ = 0 => ZZ3 = 0).
__device__ __forceinline__ void dbl(X28& p) {
  F28 u, v, w, s, m, t, x3, z;
  dbl_raw(u, p.y);        // < 12p
  sqr(v, u);
  mul(w, u, v);
  mul(s, p.x, v);
  sqr(t, p.x);
  triple_raw(m, t);       // < 6p
  sqr(t, m);
  set_zero(z);
  x3_fused(x3, t, z, s);  // < 10p
  sub_raw<16>(t, s, x3);  // < 18p
  mul(t, m, t);
  mul(u, w, p.y);
  sub<4>(p.y, t, u);      // < 6p
  p.x = x3;
  mul(p.zz, v, p.zz);
  mul(p.zzz, w, p.zzz);
}

// acc += (x2, y2), affine with x2 < 2p, y2 < 4p + (a negated y is 4p - y), limbs
// < 2^30, not infinity.  madd-2008-s with the exceptional cases.
// INLINE: the eight products of the main path inlined instead of called.  The out-of-line
// products cost 28 register moves per call to put the operands where the callee wants them --
// 624 v_mov per mixed addition, 7 % of its issue slots -- so the ONE addition site of
// k_accumulate inlines them (2.50 -> 2.27-2.35 ms at N = 2^20, 342 -> 360 M pairs/s; the ~45 KB
// loop body still fits the instruction cache, which round 1 had assumed it would not); every
// other kernel has several addition sites and keeps the calls.
template <bool INLINE = false>
__device__ __forceinline__ void madd(X28& acc, const F28& x2, const F28& y2) {
  auto pmul = [](F28& r, const F28& a, const F28& b) {
    if constexpr (INLINE)
      mul_inl(r, a, b);
    else
      mul(r, a, b);
  };
  auto psqr = [](F28& r, const F28& a) {
    if constexpr (INLINE)
      sqr_inl(r, a);
    else
      sqr(r, a);
  };
  if (is_inf(acc)) {
    acc.x = x2;
    acc.y = y2;
    norm(acc.y);
    set_one(acc.zz);
    set_one(acc.zzz);
    return;
  }
  F28 pp, r, t, q, ppp, p;
  pmul(p, x2, acc.zz);
  sub_raw<16>(p, p, acc.x);  // P = U2 - X1 < 18p
  pmul(r, y2, acc.zzz);
  sub_raw<16>(r, r, acc.y);  // R = S2 - Y1 < 18p
  psqr(pp, p);    // PP < 2p
  if (is_zero_lt2p(pp)) {    // P == 0 mod p: same x
    sqr(t, r);
    if (is_zero_lt2p(t)) {
      F28 yn = y2;
      norm(yn);
      dbl_affine(acc, x2, yn);
    } else {
      set_inf(acc);
    }
    return;
  }
  pmul(ppp, p, pp);  // PPP
  pmul(q, acc.x, pp);  // Q
  pmul(acc.zz, acc.zz, pp);
  pmul(acc.zzz, acc.zzz, ppp);
  psqr(t, r);
  x3_fused(t, t, ppp, q);    // X3 = R^2 - PPP - 2Q < 10p
  sub_raw<16>(q, q, t);      // Q - X3 < 18p
  F28 ny, z;
  set_zero(z);
  sub_raw<16>(ny, z, acc.y); // 16p - Y1 (Y1 < 6p)
  mul2_inl(acc.y, r, q, ny, ppp);  // Y3 = R (Q - X3) - Y1 PPP  (mod p), < 2p
  acc.x = t;
}

// acc += b.  add-2008-s with the exceptional cases.
__device__ __forceinline__ void add(X28& acc, const X28& b) {
  if (is_inf(b)) return;
  if (is_inf(acc)) {
    acc = b;
    return;
  }
  F28 u1, u2, s1, s2, p, r, pp, ppp, q, t;
  mul(u1, acc.x, b.zz);
  mul(u2, b.x, acc.zz);
  mul(s1, acc.y, b.zzz);
  mul(s2, b.y, acc.zzz);
  sub_raw<4>(p, u2, u1);     // < 6p
  sub_raw<4>(r, s2, s1);     // < 6p
  sqr(pp, p);
  if (is_zero_lt2p(pp)) {
    sqr(t, r);
    if (is_zero_lt2p(t))
      dbl(acc);
    else
      set_inf(acc);
    return;
  }
  mul(ppp, p, pp);
  mul(q, u1, pp);
  mul(t, acc.zz, b.zz);
  mul(acc.zz, t, pp);
  mul(t, acc.zzz, b.zzz);
  mul(acc.zzz, t, ppp);
  sqr(t, r);
  x3_fused(t, t, ppp, q);    // X3 < 10p
  sub_raw<16>(q, q, t);      // < 18p
  mul(q, r, q);
  mul(s1, s1, ppp);
  sub<4>(acc.y, q, s1);      // < 6p
  acc.x = t;
}

// r = k * p, small k: left-to-right double-and-add.
__device__ __forceinline__ void mul_small(X28& r, const X28& p, u32 k) {
  set_inf(r);
  if (k == 0) return;
  const int top = 31 - __clz(k);
  for (int bit = top; bit >= 0; bit--) {
    dbl(r);
    if ((k >> bit) & 1u) add(r, p);
  }
}

// ---------------------------------------------------------------------------
// Memory: 16-byte vector accesses (an F28 is 56 B, X28 224 B, A28 112 B; arrays
// of X28 / A28 are 16-B aligned).
// ---------------------------------------------------------------------------
template <int WORDS>
__device__ __forceinline__ void load_words(u32* dst, const void* src) {
  static_assert(WORDS % 4 == 0, "");
  const uint4* s = reinterpret_cast<const uint4*>(src);
#pragma unroll
  for (int i = 0; i < WORDS / 4; i++) {
    uint4 v = s[i];
    dst[4 * i] = v.x;
    dst[4 * i + 1] = v.y;
    dst[4 * i + 2] = v.z;
    dst[4 * i + 3] = v.w;
  }
}
template <int WORDS>
__device__ __forceinline__ void store_words(void* dst, const u32* src) {
  static_assert(WORDS % 4 == 0, "");
  uint4* d = reinterpret_cast<uint4*>(dst);
#pragma unroll
  for (int i = 0; i < WORDS / 4; i++) d[i] = make_uint4(src[4 * i], src[4 * i + 1], src[4 * i + 2], src[4 * i + 3]);
}
__device__ __forceinline__ void load(X28& r, const X28* src) { load_words<56>(reinterpret_cast<u32*>(&r), src); }
__device__ __forceinline__ void store(X28* dst, const X28& r) { store_words<56>(dst, reinterpret_cast<const u32*>(&r)); }
__device__ __forceinline__ void load(A28& r, const A28* src) { load_words<28>(reinterpret_cast<u32*>(&r), src); }
__device__ __forceinline__ void store(A28* dst, const A28& r) { store_words<28>(dst, reinterpret_cast<const u32*>(&r)); }

// X28 -> gnark-form XYZZ (G1XYZZ of bls12_381.h), canonical coordinates.
__device__ __forceinline__ void to_gnark(G1XYZZ& o, const X28& p) {
  to_gnark(o.x.l, p.x);
  to_gnark(o.y.l, p.y);
  to_gnark(o.zz.l, p.zz);
  to_gnark(o.zzz.l, p.zzz);
}
__device__ __forceinline__ void from_gnark(X28& o, const G1XYZZ& p) {
  from_gnark(o.x, p.x.l);
  from_gnark(o.y, p.y.l);
  from_gnark(o.zz, p.zz.l);
  from_gnark(o.zzz, p.zzz.l);
}

}  // namespace d28
}  // namespace curdle
