// BLS12-381 base-field / scalar-field / G1 arithmetic on 32-bit limbs.
//
// One header, two consumers: the gfx950 kernels (msm_*_kernels.hip) and the host
// side of libcurdlemsm.so (window combine, Jacobian normalisation, the
// msmaccumulator mirror).  It replaces what the reference gets from the
// un-vendored gnark-crypto v0.11.0 (/root/reference/go.mod:6): fp.Element,
// fr.Element, G1Affine, G1Jac and the bucket arithmetic inside
// (*G1Jac).MultiExp (called at msmaccumulator/msmaccumulator.go:59).
//
// Memory layout is gnark's: fp.Element = [6]uint64 and fr.Element = [4]uint64,
// little-endian limbs, Montgomery form (R = 2^384 resp. 2^256).  On a
// little-endian machine that is bit-identical to 12 resp. 8 uint32 limbs, which
// is what the CDNA4 integer multiplier (v_mad_u64_u32: 32x32+64) wants.
//
// This is product code.  It never includes or links anything under oracle/.
#pragma once
#include <stdint.h>
#include <type_traits>

#if defined(__HIPCC__)
#define CURDLE_HD __host__ __device__ __forceinline__
#define CURDLE_HD_NOINLINE __host__ __device__ __noinline__ inline
#else
#define CURDLE_HD inline
#define CURDLE_HD_NOINLINE inline
#endif

namespace curdle {

typedef uint32_t u32;
typedef uint64_t u64;

// ---------------------------------------------------------------------------
// Parameters (32-bit little-endian limbs)
// ---------------------------------------------------------------------------
struct FpParams {
  static constexpr int N = 12;
  // p = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab
  static CURDLE_HD u32 mod(int i) {
    constexpr u32 m[12] = {0xffffaaabu, 0xb9feffffu, 0xb153ffffu, 0x1eabfffeu, 0xf6b0f624u, 0x6730d2a0u,
                           0xf38512bfu, 0x64774b84u, 0x434bacd7u, 0x4b1ba7b6u, 0x397fe69au, 0x1a0111eau};
    return m[i];
  }
  // R = 2^384 mod p (Montgomery one)
  static CURDLE_HD u32 one(int i) {
    constexpr u32 m[12] = {0x0002fffdu, 0x76090000u, 0xc40c0002u, 0xebf4000bu, 0x53c758bau, 0x5f489857u,
                           0x70525745u, 0x77ce5853u, 0xa256ec6du, 0x5c071a97u, 0xfa80e493u, 0x15f65ec3u};
    return m[i];
  }
  static constexpr u32 n0inv = 0xfffcfffdu;  // -p^-1 mod 2^32
};

struct FrParams {
  static constexpr int N = 8;
  // r = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
  static CURDLE_HD u32 mod(int i) {
    constexpr u32 m[8] = {0x00000001u, 0xffffffffu, 0xfffe5bfeu, 0x53bda402u,
                          0x09a1d805u, 0x3339d808u, 0x299d7d48u, 0x73eda753u};
    return m[i];
  }
  // 2^256 mod r
  static CURDLE_HD u32 one(int i) {
    constexpr u32 m[8] = {0xfffffffeu, 0x00000001u, 0x00034802u, 0x5884b7fau,
                          0xecbc4ff5u, 0x998c4fefu, 0xacc5056fu, 0x1824b159u};
    return m[i];
  }
  static constexpr u32 n0inv = 0xffffffffu;  // -r^-1 mod 2^32
};

// ---------------------------------------------------------------------------
// Generic Montgomery field element on N 32-bit limbs, always fully reduced.
// ---------------------------------------------------------------------------
template <class PR>
struct Mont {
  static constexpr int N = PR::N;
  u32 l[N];
};
typedef Mont<FpParams> Fp;
typedef Mont<FrParams> Fr;

template <class PR>
CURDLE_HD void f_zero(Mont<PR>& r) {
#pragma unroll
  for (int i = 0; i < PR::N; i++) r.l[i] = 0;
}
template <class PR>
CURDLE_HD void f_one(Mont<PR>& r) {
#pragma unroll
  for (int i = 0; i < PR::N; i++) r.l[i] = PR::one(i);
}
template <class PR>
CURDLE_HD bool f_is_zero(const Mont<PR>& a) {
  u32 acc = 0;
#pragma unroll
  for (int i = 0; i < PR::N; i++) acc |= a.l[i];
  return acc == 0;
}
template <class PR>
CURDLE_HD bool f_eq(const Mont<PR>& a, const Mont<PR>& b) {
  u32 acc = 0;
#pragma unroll
  for (int i = 0; i < PR::N; i++) acc |= a.l[i] ^ b.l[i];
  return acc == 0;
}

// r = a - p if a >= p else a   (a < 2p, carry = extra top bit of a)
template <class PR>
CURDLE_HD void f_cond_sub(Mont<PR>& r, const u32* a, u32 top) {
  u32 d[PR::N];
  u64 borrow = 0;
#pragma unroll
  for (int i = 0; i < PR::N; i++) {
    u64 t = (u64)a[i] - PR::mod(i) - borrow;
    d[i] = (u32)t;
    borrow = (t >> 32) & 1;
  }
  // a >= p  <=>  top set or no final borrow
  bool ge = (top != 0) | (borrow == 0);
#pragma unroll
  for (int i = 0; i < PR::N; i++) r.l[i] = ge ? d[i] : a[i];
}

template <class PR>
CURDLE_HD void f_add(Mont<PR>& r, const Mont<PR>& a, const Mont<PR>& b) {
  u32 s[PR::N];
  u64 c = 0;
#pragma unroll
  for (int i = 0; i < PR::N; i++) {
    u64 t = (u64)a.l[i] + b.l[i] + c;
    s[i] = (u32)t;
    c = t >> 32;
  }
  f_cond_sub<PR>(r, s, (u32)c);
}

template <class PR>
CURDLE_HD void f_sub(Mont<PR>& r, const Mont<PR>& a, const Mont<PR>& b) {
  u32 d[PR::N];
  u64 borrow = 0;
#pragma unroll
  for (int i = 0; i < PR::N; i++) {
    u64 t = (u64)a.l[i] - b.l[i] - borrow;
    d[i] = (u32)t;
    borrow = (t >> 32) & 1;
  }
  u32 mask = (u32)0 - (u32)borrow;  // all ones if a < b
  u64 c = 0;
#pragma unroll
  for (int i = 0; i < PR::N; i++) {
    u64 t = (u64)d[i] + (PR::mod(i) & mask) + c;
    r.l[i] = (u32)t;
    c = t >> 32;
  }
}

template <class PR>
CURDLE_HD void f_neg(Mont<PR>& r, const Mont<PR>& a) {
  // r = (a == 0) ? 0 : p - a
  u32 nz = 0;
#pragma unroll
  for (int i = 0; i < PR::N; i++) nz |= a.l[i];
  u32 mask = nz ? 0xffffffffu : 0u;
  u64 borrow = 0;
#pragma unroll
  for (int i = 0; i < PR::N; i++) {
    u64 t = (u64)PR::mod(i) - a.l[i] - borrow;
    r.l[i] = (u32)t & mask;
    borrow = (t >> 32) & 1;
  }
}

template <class PR>
CURDLE_HD void f_dbl(Mont<PR>& r, const Mont<PR>& a) {
  f_add<PR>(r, a, a);
}

// ---------------------------------------------------------------------------
// Column accumulator for product-scanning multiplication: a 96-bit value
// {hi:lo}.  On gfx950 one product costs one v_mad_u64_u32 (quarter-rate
// 32x32+64 multiply-add, carry-out in VCC) plus one v_addc_co_u32, with no
// zero-extension moves; hipcc cannot be coaxed into using the multiply-add's
// carry-out from C, hence the generated asm blocks in mac_gfx950.inc.
// ---------------------------------------------------------------------------
#if defined(__HIP_DEVICE_COMPILE__)
#include "mac_gfx950.inc"
#endif

template <int I, int E, class F>
CURDLE_HD void static_for(F&& f) {
  if constexpr (I < E) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, E>(f);
  }
}

// acc += sum_{i<K} a[i] * b[-i]   (b walks downwards)
template <int K>
CURDLE_HD void col_mac_vv(u64& lo, u32& hi, const u32* a, const u32* b) {
#if defined(__HIP_DEVICE_COMPILE__)
  macv<K>(lo, hi, a, b);
#else
  for (int i = 0; i < K; i++) {
    u64 p = (u64)a[i] * b[-i];
    lo += p;
    hi += (lo < p);
  }
#endif
}
// Same, second factor known at compile time (modulus limbs; SGPRs on gfx950).
template <int K>
CURDLE_HD void col_mac_vs(u64& lo, u32& hi, const u32* a, const u32* b) {
#if defined(__HIP_DEVICE_COMPILE__)
  macs<K>(lo, hi, a, b);
#else
  col_mac_vv<K>(lo, hi, a, b);
#endif
}
CURDLE_HD void col_shift(u64& lo, u32& hi) {
  lo = (lo >> 32) | ((u64)hi << 32);
  hi = 0;
}

template <class PR>
struct ModTable {
  u32 v[PR::N];
  CURDLE_HD ModTable() {
#pragma unroll
    for (int i = 0; i < PR::N; i++) v[i] = PR::mod(i);
  }
};

// Product-scanning (FIPS) Montgomery product, r = a*b*R^-1 mod p, fully reduced.
// 2*N^2 + N multiplies.  p < 2^(32N-2) for both fields, so the result before
// the conditional subtraction is < 2p and fits N limbs.
template <class PR>
CURDLE_HD void f_mul_inl(Mont<PR>& r, const Mont<PR>& a, const Mont<PR>& b) {
  constexpr int N = PR::N;
  const ModTable<PR> P;
  u32 m[N];
  u32 t[N];
  u64 lo = 0;
  u32 hi = 0;
  static_for<0, N>([&](auto kc) {
    constexpr int k = decltype(kc)::value;
    col_mac_vv<k + 1>(lo, hi, &a.l[0], &b.l[k]);
    if constexpr (k > 0) col_mac_vs<k>(lo, hi, &m[0], &P.v[k]);
    m[k] = (u32)lo * PR::n0inv;
    col_mac_vs<1>(lo, hi, &m[k], &P.v[0]);
    col_shift(lo, hi);
  });
  static_for<N, 2 * N - 1>([&](auto kc) {
    constexpr int k = decltype(kc)::value;
    constexpr int i0 = k - N + 1;
    col_mac_vv<N - i0>(lo, hi, &a.l[i0], &b.l[N - 1]);
    col_mac_vs<N - i0>(lo, hi, &m[i0], &P.v[N - 1]);
    t[k - N] = (u32)lo;
    col_shift(lo, hi);
  });
  t[N - 1] = (u32)lo;
  f_cond_sub<PR>(r, t, 0);
}

// Montgomery reduction of a single element: r = a * R^-1 (Montgomery -> canonical)
template <class PR>
CURDLE_HD void f_from_mont(Mont<PR>& r, const Mont<PR>& a) {
  constexpr int N = PR::N;
  const ModTable<PR> P;
  u32 m[N];
  u32 t[N];
  u64 lo = 0;
  u32 hi = 0;
  static_for<0, N>([&](auto kc) {
    constexpr int k = decltype(kc)::value;
    u64 s = lo + a.l[k];
    hi += (s < lo);
    lo = s;
    if constexpr (k > 0) col_mac_vs<k>(lo, hi, &m[0], &P.v[k]);
    m[k] = (u32)lo * PR::n0inv;
    col_mac_vs<1>(lo, hi, &m[k], &P.v[0]);
    col_shift(lo, hi);
  });
  static_for<N, 2 * N - 1>([&](auto kc) {
    constexpr int k = decltype(kc)::value;
    constexpr int i0 = k - N + 1;
    col_mac_vs<N - i0>(lo, hi, &m[i0], &P.v[N - 1]);
    t[k - N] = (u32)lo;
    col_shift(lo, hi);
  });
  t[N - 1] = (u32)lo;
  f_cond_sub<PR>(r, t, 0);
}

// Fp multiply entry points.  In device code the 700-instruction body is kept
// out of line (one copy per kernel image, operands and result passed in VGPRs
// as 12-lane vectors so nothing touches scratch): a G1 mixed add holds ten of
// them and would otherwise overflow the instruction cache.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(CURDLE_INLINE_MUL)
typedef u32 u32x12 __attribute__((ext_vector_type(12)));
__device__ __noinline__ inline u32x12 fp_mul_call(u32x12 a, u32x12 b) {
  Fp x, y, r;
#pragma unroll
  for (int i = 0; i < 12; i++) {
    x.l[i] = a[i];
    y.l[i] = b[i];
  }
  f_mul_inl<FpParams>(r, x, y);
  u32x12 o;
#pragma unroll
  for (int i = 0; i < 12; i++) o[i] = r.l[i];
  return o;
}
CURDLE_HD void fp_mul(Fp& r, const Fp& a, const Fp& b) {
  u32x12 x, y;
#pragma unroll
  for (int i = 0; i < 12; i++) {
    x[i] = a.l[i];
    y[i] = b.l[i];
  }
  u32x12 o = fp_mul_call(x, y);
#pragma unroll
  for (int i = 0; i < 12; i++) r.l[i] = o[i];
}
#elif defined(__HIP_DEVICE_COMPILE__)
CURDLE_HD void fp_mul(Fp& r, const Fp& a, const Fp& b) { f_mul_inl<FpParams>(r, a, b); }
#elif defined(__x86_64__) && defined(__BMI2__) && defined(__ADX__)
// Host, BMI2 + ADX translation units (host/host_ops_bmi2.cpp): 6 x 64-bit limbs, one row
// of the operand-scanning Montgomery product per multiplier word with the two carry chains
// of adcx / adox running side by side, and the reduction row folded in right behind it
// (p < 2^383, so the running value never needs a seventh limb).  The asm body is generated
// by gen_mont_x86.py into mont_x86_64.inc.
CURDLE_HD void fp_mul(Fp& r, const Fp& a, const Fp& b) {
  static const u64 P64[6] = {0xb9feffffffffaaabull, 0x1eabfffeb153ffffull, 0x6730d2a0f6b0f624ull,
                             0x64774b84f38512bfull, 0x4b1ba7b6434bacd7ull, 0x1a0111ea397fe69aull};
  static const u64 N0 = 0x89f3fffcfffcfffdull;
  u64 x[6], y[6];
  __builtin_memcpy(x, a.l, 48);
  __builtin_memcpy(y, b.l, 48);
  u64 t0, t1, t2, t3, t4, t5, A, ax, bx, dx;
  asm(
#include "mont_x86_64.inc"
      : [t0] "=&r"(t0), [t1] "=&r"(t1), [t2] "=&r"(t2), [t3] "=&r"(t3), [t4] "=&r"(t4), [t5] "=&r"(t5), [A] "=&r"(A),
        [ax] "=&r"(ax), [bx] "=&r"(bx), "=&d"(dx)
      : [x] "r"(x), [y] "r"(y), [p] "r"(P64), [ninv] "m"(N0), "m"(x), "m"(y), "m"(P64)
      : "cc");
  // t < 2p: one conditional subtraction
  u64 d0, d1, d2, d3, d4, d5;
  unsigned long long s;
  unsigned char bw = 0;
  bw = __builtin_ia32_sbb_u64(bw, t0, P64[0], &s), d0 = s;
  bw = __builtin_ia32_sbb_u64(bw, t1, P64[1], &s), d1 = s;
  bw = __builtin_ia32_sbb_u64(bw, t2, P64[2], &s), d2 = s;
  bw = __builtin_ia32_sbb_u64(bw, t3, P64[3], &s), d3 = s;
  bw = __builtin_ia32_sbb_u64(bw, t4, P64[4], &s), d4 = s;
  bw = __builtin_ia32_sbb_u64(bw, t5, P64[5], &s), d5 = s;
  u64 o[6] = {bw ? t0 : d0, bw ? t1 : d1, bw ? t2 : d2, bw ? t3 : d3, bw ? t4 : d4, bw ? t5 : d5};
  __builtin_memcpy(r.l, o, 48);
}
#else
// Host, generic build: the window combine after the GPU phases is a serial chain of ~2,400
// field multiplications, so the host path uses 64-bit limbs instead of the kernels' 32-bit
// columns.  Same Montgomery form, same results.
CURDLE_HD void fp_mul(Fp& r, const Fp& a, const Fp& b) {
  typedef unsigned __int128 u128;
  static const u64 P64[6] = {0xb9feffffffffaaabull, 0x1eabfffeb153ffffull, 0x6730d2a0f6b0f624ull,
                             0x64774b84f38512bfull, 0x4b1ba7b6434bacd7ull, 0x1a0111ea397fe69aull};
  static const u64 N0 = 0x89f3fffcfffcfffdull;
  u64 x[6], y[6], t[7] = {0, 0, 0, 0, 0, 0, 0};
  for (int i = 0; i < 6; i++) {
    x[i] = (u64)a.l[2 * i] | ((u64)a.l[2 * i + 1] << 32);
    y[i] = (u64)b.l[2 * i] | ((u64)b.l[2 * i + 1] << 32);
  }
  for (int i = 0; i < 6; i++) {
    u64 c = 0;
    for (int j = 0; j < 6; j++) {
      u128 s = (u128)x[j] * y[i] + t[j] + c;
      t[j] = (u64)s;
      c = (u64)(s >> 64);
    }
    u64 top = t[6] + c;  // value < 2p: no overflow past limb 6
    const u64 m = t[0] * N0;
    u128 s = (u128)m * P64[0] + t[0];
    c = (u64)(s >> 64);
    for (int j = 1; j < 6; j++) {
      s = (u128)m * P64[j] + t[j] + c;
      t[j - 1] = (u64)s;
      c = (u64)(s >> 64);
    }
    s = (u128)top + c;
    t[5] = (u64)s;
    t[6] = (u64)(s >> 64);
  }
  // conditional subtraction
  u64 d[6], borrow = 0;
  for (int i = 0; i < 6; i++) {
    u128 s = (u128)t[i] - P64[i] - borrow;
    d[i] = (u64)s;
    borrow = (u64)(s >> 64) & 1;
  }
  const bool ge = t[6] != 0 || borrow == 0;
  for (int i = 0; i < 6; i++) {
    u64 v = ge ? d[i] : t[i];
    r.l[2 * i] = (u32)v;
    r.l[2 * i + 1] = (u32)(v >> 32);
  }
}
#endif
CURDLE_HD void fp_sqr(Fp& r, const Fp& a) { fp_mul(r, a, a); }
#if !defined(__HIP_DEVICE_COMPILE__) && defined(__x86_64__) && defined(__BMI2__) && defined(__ADX__)
// Host, BMI2 + ADX translation units: additions and subtractions on six 64-bit limbs
// (adc / sbb chains) instead of the generic twelve 32-bit ones.
namespace x86 {
static const u64 kP64[6] = {0xb9feffffffffaaabull, 0x1eabfffeb153ffffull, 0x6730d2a0f6b0f624ull,
                            0x64774b84f38512bfull, 0x4b1ba7b6434bacd7ull, 0x1a0111ea397fe69aull};
inline unsigned char add6(u64* r, const u64* a, const u64* b) {
  unsigned char c = 0;
  unsigned long long s;
  for (int i = 0; i < 6; i++) c = __builtin_ia32_addcarryx_u64(c, a[i], b[i], &s), r[i] = s;
  return c;
}
inline unsigned char sub6(u64* r, const u64* a, const u64* b) {
  unsigned char c = 0;
  unsigned long long s;
  for (int i = 0; i < 6; i++) c = __builtin_ia32_sbb_u64(c, a[i], b[i], &s), r[i] = s;
  return c;
}
}  // namespace x86
CURDLE_HD void fp_add(Fp& r, const Fp& a, const Fp& b) {
  u64 x[6], y[6], s[6], d[6];
  __builtin_memcpy(x, a.l, 48);
  __builtin_memcpy(y, b.l, 48);
  x86::add6(s, x, y);  // a, b < p < 2^383: no carry out
  const bool lt = x86::sub6(d, s, x86::kP64);
  __builtin_memcpy(r.l, lt ? s : d, 48);
}
CURDLE_HD void fp_sub(Fp& r, const Fp& a, const Fp& b) {
  u64 x[6], y[6], d[6], e[6];
  __builtin_memcpy(x, a.l, 48);
  __builtin_memcpy(y, b.l, 48);
  const bool lt = x86::sub6(d, x, y);
  x86::add6(e, d, x86::kP64);
  __builtin_memcpy(r.l, lt ? e : d, 48);
}
CURDLE_HD void fp_neg(Fp& r, const Fp& a) {
  u64 x[6], d[6];
  __builtin_memcpy(x, a.l, 48);
  const bool nz = (x[0] | x[1] | x[2] | x[3] | x[4] | x[5]) != 0;
  x86::sub6(d, x86::kP64, x);
  for (int i = 0; i < 6; i++) d[i] = nz ? d[i] : 0;
  __builtin_memcpy(r.l, d, 48);
}
CURDLE_HD void fp_dbl(Fp& r, const Fp& a) { fp_add(r, a, a); }
#else
CURDLE_HD void fp_add(Fp& r, const Fp& a, const Fp& b) { f_add<FpParams>(r, a, b); }
CURDLE_HD void fp_sub(Fp& r, const Fp& a, const Fp& b) { f_sub<FpParams>(r, a, b); }
CURDLE_HD void fp_neg(Fp& r, const Fp& a) { f_neg<FpParams>(r, a); }
CURDLE_HD void fp_dbl(Fp& r, const Fp& a) { f_dbl<FpParams>(r, a); }
#endif
#if defined(__HIP_DEVICE_COMPILE__)
CURDLE_HD void fr_mul(Fr& r, const Fr& a, const Fr& b) { f_mul_inl<FrParams>(r, a, b); }
#else
// Host: four 64-bit limbs, one operand-scanning row + one reduction row per multiplier
// word (the verifier's challenge algebra is a few thousand of these per proof).
CURDLE_HD void fr_mul(Fr& r, const Fr& a, const Fr& b) {
  typedef unsigned __int128 u128;
  static const u64 R64[4] = {0xffffffff00000001ull, 0x53bda402fffe5bfeull, 0x3339d80809a1d805ull, 0x73eda753299d7d48ull};
  static const u64 N0 = 0xfffffffeffffffffull;  // -r^-1 mod 2^64
  u64 x[4], y[4], t[5] = {0, 0, 0, 0, 0};
  __builtin_memcpy(x, a.l, 32);
  __builtin_memcpy(y, b.l, 32);
  for (int i = 0; i < 4; i++) {
    u64 c = 0;
    for (int j = 0; j < 4; j++) {
      u128 s = (u128)x[j] * y[i] + t[j] + c;
      t[j] = (u64)s;
      c = (u64)(s >> 64);
    }
    const u64 top = t[4] + c;  // running value < 2r < 2^256: fits
    const u64 m = t[0] * N0;
    u128 s = (u128)m * R64[0] + t[0];
    c = (u64)(s >> 64);
    for (int j = 1; j < 4; j++) {
      s = (u128)m * R64[j] + t[j] + c;
      t[j - 1] = (u64)s;
      c = (u64)(s >> 64);
    }
    s = (u128)top + c;
    t[3] = (u64)s;
    t[4] = (u64)(s >> 64);
  }
  u64 d[4], borrow = 0;
  for (int i = 0; i < 4; i++) {
    u128 s = (u128)t[i] - R64[i] - borrow;
    d[i] = (u64)s;
    borrow = (u64)(s >> 64) & 1;
  }
  const bool ge = t[4] != 0 || borrow == 0;
  __builtin_memcpy(r.l, ge ? d : t, 32);
}
#endif
CURDLE_HD void fr_add(Fr& r, const Fr& a, const Fr& b) { f_add<FrParams>(r, a, b); }
CURDLE_HD void fr_sub(Fr& r, const Fr& a, const Fr& b) { f_sub<FrParams>(r, a, b); }

// ---------------------------------------------------------------------------
// G1: y^2 = x^3 + 4
// ---------------------------------------------------------------------------
// gnark G1Affine: (0,0) is the point at infinity (curdleproof.go:23 zeroPoint).
struct G1Affine {
  Fp x, y;
};
// Extended Jacobian ("XYZZ"): x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2; infinity <=> ZZ = 0.
struct G1XYZZ {
  Fp x, y, zz, zzz;
};
// gnark G1Jac: x = X/Z^2, y = Y/Z^3; infinity <=> Z = 0.
struct G1Jac {
  Fp x, y, z;
};

CURDLE_HD bool g1_affine_is_inf(const G1Affine& p) { return f_is_zero(p.x) && f_is_zero(p.y); }
CURDLE_HD bool g1_is_inf(const G1XYZZ& p) { return f_is_zero(p.zz); }
CURDLE_HD void g1_set_inf(G1XYZZ& p) {
  f_one(p.x);
  f_one(p.y);
  f_zero(p.zz);
  f_zero(p.zzz);
}
CURDLE_HD void g1_from_affine(G1XYZZ& r, const G1Affine& a) {
  if (g1_affine_is_inf(a)) {
    g1_set_inf(r);
    return;
  }
  r.x = a.x;
  r.y = a.y;
  f_one(r.zz);
  f_one(r.zzz);
}

// r = 2*(x,y) for an affine, non-infinity input.  EFD mdbl-2008-s-1 (a = 0).
CURDLE_HD void g1_dbl_affine(G1XYZZ& r, const Fp& x1, const Fp& y1) {
  Fp u, v, w, s, m, t;
  fp_dbl(u, y1);
  fp_sqr(v, u);
  fp_mul(w, u, v);
  fp_mul(s, x1, v);
  fp_sqr(t, x1);
  fp_dbl(m, t);
  fp_add(m, m, t);  // 3*x1^2
  fp_sqr(r.x, m);
  fp_sub(r.x, r.x, s);
  fp_sub(r.x, r.x, s);
  fp_sub(t, s, r.x);
  fp_mul(t, m, t);
  fp_mul(u, w, y1);
  fp_sub(r.y, t, u);
  r.zz = v;
  r.zzz = w;
}

// p = 2*p.  EFD dbl-2008-s-1 (a = 0).  y = 0 cannot occur on a prime-order
// curve, so the only special case is infinity (ZZ = 0 stays 0).
CURDLE_HD void g1_dbl(G1XYZZ& p) {
  Fp u, v, w, s, m, t;
  fp_dbl(u, p.y);
  fp_sqr(v, u);
  fp_mul(w, u, v);
  fp_mul(s, p.x, v);
  fp_sqr(t, p.x);
  fp_dbl(m, t);
  fp_add(m, m, t);
  Fp x3;
  fp_sqr(x3, m);
  fp_sub(x3, x3, s);
  fp_sub(x3, x3, s);
  fp_sub(t, s, x3);
  fp_mul(t, m, t);
  fp_mul(u, w, p.y);
  fp_sub(p.y, t, u);
  p.x = x3;
  fp_mul(p.zz, v, p.zz);
  fp_mul(p.zzz, w, p.zzz);
}

// acc += (x2, +-y2) with (x2,y2) affine.  EFD madd-2008-s plus the exceptional
// cases (acc infinity, equal points -> doubling, opposite points -> infinity).
// The caller filters an infinity (0,0) base.
CURDLE_HD void g1_madd(G1XYZZ& acc, const Fp& x2, const Fp& y2) {
  if (g1_is_inf(acc)) {
    acc.x = x2;
    acc.y = y2;
    f_one(acc.zz);
    f_one(acc.zzz);
    return;
  }
  Fp pp, r, t, q, ppp;
  fp_mul(pp, x2, acc.zz);
  fp_sub(pp, pp, acc.x);  // P = U2 - X1
  fp_mul(r, y2, acc.zzz);
  fp_sub(r, r, acc.y);  // R = S2 - Y1
  if (f_is_zero(pp)) {
    if (f_is_zero(r))
      g1_dbl_affine(acc, x2, y2);
    else
      g1_set_inf(acc);
    return;
  }
  fp_sqr(t, pp);         // PP
  fp_mul(ppp, pp, t);    // PPP
  fp_mul(q, acc.x, t);   // Q = X1*PP
  fp_mul(acc.zz, acc.zz, t);
  fp_mul(acc.zzz, acc.zzz, ppp);
  fp_sqr(t, r);
  fp_sub(t, t, ppp);
  fp_sub(t, t, q);
  fp_sub(t, t, q);  // X3
  fp_sub(q, q, t);
  fp_mul(q, r, q);  // R*(Q - X3)
  fp_mul(ppp, acc.y, ppp);
  fp_sub(acc.y, q, ppp);
  acc.x = t;
}

// acc += b.  EFD add-2008-s plus exceptional cases.
CURDLE_HD void g1_add(G1XYZZ& acc, const G1XYZZ& b) {
  if (g1_is_inf(b)) return;
  if (g1_is_inf(acc)) {
    acc = b;
    return;
  }
  Fp u1, u2, s1, s2, pp, r, t, q, ppp;
  fp_mul(u1, acc.x, b.zz);
  fp_mul(u2, b.x, acc.zz);
  fp_mul(s1, acc.y, b.zzz);
  fp_mul(s2, b.y, acc.zzz);
  fp_sub(pp, u2, u1);
  fp_sub(r, s2, s1);
  if (f_is_zero(pp)) {
    if (f_is_zero(r))
      g1_dbl(acc);
    else
      g1_set_inf(acc);
    return;
  }
  fp_sqr(t, pp);
  fp_mul(ppp, pp, t);
  fp_mul(q, u1, t);
  fp_mul(acc.zz, acc.zz, b.zz);
  fp_mul(acc.zz, acc.zz, t);
  fp_mul(acc.zzz, acc.zzz, b.zzz);
  fp_mul(acc.zzz, acc.zzz, ppp);
  fp_sqr(t, r);
  fp_sub(t, t, ppp);
  fp_sub(t, t, q);
  fp_sub(t, t, q);
  fp_sub(q, q, t);
  fp_mul(q, r, q);
  fp_mul(s1, s1, ppp);
  fp_sub(acc.y, q, s1);
  acc.x = t;
}

CURDLE_HD void g1_neg(G1XYZZ& p) { fp_neg(p.y, p.y); }

// ---------------------------------------------------------------------------
// GLV split of a scalar (kernels: k_digits; host: g1_scalar_mul_glv).  phi(x, y) = (beta x, y)
// is multiplication by lambda = z^2 - 1 (lambda^2 + lambda + 1 = r), so k P = k1 P + k2 phi(P).
// The split: k' = min(k, r - k) (sign s), k2 = round(k' / lambda), k1 = k' - k2 lambda in
// [-lambda/2, lambda/2); both magnitudes are below 1.35 * 2^126, so 127 bits suffice and the top
// window's digit plus carry stays below 2^(width of the top window).  k2 by Barrett division
// (HAC 14.42: mu = floor(2^256 / lambda), at most two corrections; checked against big-integer
// division on 300,000 random and the boundary scalars).  In: k canonical (< r).  Out: |k1|, k2 as
// four words each and the signs of the two terms (0x80000000 = negative).
// ---------------------------------------------------------------------------
CURDLE_HD void glv_split(const Fr& k, u32 a[4], u32 b[4], u32& neg_a, u32& neg_b) {
  constexpr u32 R_[8] = {0x00000001u, 0xffffffffu, 0xfffe5bfeu, 0x53bda402u, 0x09a1d805u, 0x3339d808u, 0x299d7d48u, 0x73eda753u};
  constexpr u32 HALF[8] = {0x80000000u, 0x7fffffffu, 0x7fff2dffu, 0xa9ded201u, 0x04d0ec02u, 0x199cec04u, 0x94cebea4u, 0x39f6d3a9u};
  constexpr u32 LAM[5] = {0xffffffffu, 0x00000000u, 0x0001a402u, 0xac45a401u, 0u};
  constexpr u32 LH[5] = {0x7fffffffu, 0x00000000u, 0x8000d201u, 0x5622d200u, 0u};  // lambda >> 1
  constexpr u32 MU[5] = {0xf6cfee30u, 0x63f6e522u, 0xe01faaddu, 0x7c6becf1u, 0x00000001u};
  // s = k > (r - 1) / 2;  m = (s ? r - k : k) + (lambda >> 1)
  bool gt = false, decided = false;
#pragma unroll
  for (int i = 7; i >= 0; i--) {
    if (!decided && k.l[i] != HALF[i]) {
      gt = k.l[i] > HALF[i];
      decided = true;
    }
  }
  u32 m[9];
  {
    u64 borrow = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const u64 d = (u64)R_[i] - k.l[i] - borrow;
      m[i] = gt ? (u32)d : k.l[i];
      borrow = (d >> 32) & 1u;
    }
    u64 carry = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
      carry += (u64)m[i] + (i < 4 ? LH[i] : 0u);
      m[i] = (u32)carry;
      carry >>= 32;
    }
    m[8] = 0;
  }
  // q1 = m >> 127 (< 2^128);  q3 = (q1 * mu) >> 129
  u32 q1[4];
#pragma unroll
  for (int j = 0; j < 4; j++) q1[j] = (m[j + 3] >> 31) | (m[j + 4] << 1);
  u32 q2[10];
  {
    u64 acc = 0, hi = 0;  // column sums of up to four 64-bit products: carry the overflow separately
#pragma unroll
    for (int col = 0; col < 9; col++) {
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const int j = col - i;
        if (j >= 0 && j < 5) {
          const u64 pr = (u64)q1[i] * MU[j];
          acc += pr;
          hi += acc < pr ? 1u : 0u;
        }
      }
      q2[col] = (u32)acc;
      acc = (acc >> 32) | (hi << 32);
      hi = 0;
    }
    q2[9] = (u32)acc;
  }
  u32 q3[5];
#pragma unroll
  for (int j = 0; j < 5; j++) q3[j] = (q2[j + 4] >> 1) | ((j + 5 < 10 ? q2[j + 5] : 0u) << 31);
  // rem = m - q3 * lambda  (mod 2^160: the true remainder is below 3 lambda)
  u32 rem[5];
  {
    u32 t[5];
    u64 acc = 0, hi = 0;
#pragma unroll
    for (int col = 0; col < 5; col++) {
#pragma unroll
      for (int i = 0; i < 5; i++) {
        const int j = col - i;
        if (j >= 0 && j < 4) {
          const u64 pr = (u64)q3[i] * LAM[j];
          acc += pr;
          hi += acc < pr ? 1u : 0u;
        }
      }
      t[col] = (u32)acc;
      acc = (acc >> 32) | (hi << 32);
      hi = 0;
    }
    u64 borrow = 0;
#pragma unroll
    for (int i = 0; i < 5; i++) {
      const u64 d = (u64)m[i] - t[i] - borrow;
      rem[i] = (u32)d;
      borrow = (d >> 32) & 1u;
    }
  }
  // at most two corrections: while rem >= lambda { rem -= lambda; q3 += 1 }
#pragma unroll
  for (int it = 0; it < 2; it++) {
    u32 d[5];
    u64 borrow = 0;
#pragma unroll
    for (int i = 0; i < 5; i++) {
      const u64 x = (u64)rem[i] - LAM[i] - borrow;
      d[i] = (u32)x;
      borrow = (x >> 32) & 1u;
    }
    if (!borrow) {  // rem >= lambda
#pragma unroll
      for (int i = 0; i < 5; i++) rem[i] = d[i];
      u64 carry = 1;
#pragma unroll
      for (int i = 0; i < 5; i++) {
        carry += q3[i];
        q3[i] = (u32)carry;
        carry >>= 32;
      }
    }
  }
  // k2 = q3;  k1 = rem - (lambda >> 1), as sign and magnitude
  u32 d[4], e[4];
  u64 b1 = 0, b2 = 0;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const u64 x = (u64)rem[i] - LH[i] - b1;
    d[i] = (u32)x;
    b1 = (x >> 32) & 1u;
    const u64 y = (u64)LH[i] - rem[i] - b2;
    e[i] = (u32)y;
    b2 = (y >> 32) & 1u;
  }
  const bool k1_neg = b1 != 0;  // rem < lambda >> 1  (rem < lambda < 2^128: limb 4 is zero)
#pragma unroll
  for (int i = 0; i < 4; i++) {
    a[i] = k1_neg ? e[i] : d[i];
    b[i] = q3[i];
  }
  neg_b = gt ? 0x80000000u : 0u;
  neg_a = neg_b ^ (k1_neg ? 0x80000000u : 0u);
}



}  // namespace curdle
