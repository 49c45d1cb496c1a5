// gnark G1Affine.Bytes() for eight points at a time with AVX-512 IFMA: the verifier hashes the
// 4 ell instance points into its transcript (curdleproof.go:217-224), and turning their
// Montgomery coordinates into canonical big-endian bytes -- two Montgomery reductions per point
// -- was 0.1 ms of a 1.02 ms verification at ell = 252.  One field element per 64-bit lane,
// radix 2^52, v_pmadd52{lo,hi}.  Built with -mavx512f -mavx512ifma -mavx512vl -mavx512bw -mavx512dq; only
// called when the CPU has them (alg::CompressAffineBatch dispatches).
#include <immintrin.h>
#include <stdint.h>
#include <string.h>

#include "../csrc/bls12_381.h"

namespace curdle {
namespace alg {

namespace {
constexpr uint64_t kMask52 = (1ull << 52) - 1;
// p in radix 2^52 (8 limbs), -p^-1 mod 2^52, and (p + 1) / 2 in radix 2^52
struct Consts {
  uint64_t p[8], half[8], n0;
  Consts() {
    // p = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab
    const uint64_t p64[6] = {0xb9feffffffffaaabull, 0x1eabfffeb153ffffull, 0x6730d2a0f6b0f624ull,
                             0x64774b84f38512bfull, 0x4b1ba7b6434bacd7ull, 0x1a0111ea397fe69aull};
    unsigned __int128 acc = 0;
    int bits = 0, src = 0;
    for (int j = 0; j < 8; j++) {
      while (bits < 52 && src < 6) {
        acc |= (unsigned __int128)p64[src++] << bits;
        bits += 64;
      }
      p[j] = (uint64_t)acc & kMask52;
      acc >>= 52;
      bits -= 52;
    }
    // (p + 1) / 2
    uint64_t c = 1;
    uint64_t t[8];
    for (int j = 0; j < 8; j++) {
      t[j] = p[j] + c;
      c = t[j] >> 52;
      t[j] &= kMask52;
    }
    for (int j = 0; j < 8; j++) half[j] = ((t[j] >> 1) | ((j < 7 ? t[j + 1] : 0) << 51)) & kMask52;
    // n0 = -p^-1 mod 2^52 by Newton iteration
    uint64_t inv = 1;
    for (int i = 0; i < 6; i++) inv *= 2 - p[0] * inv;
    n0 = (0 - inv) & kMask52;
  }
};
const Consts& K() {
  static const Consts k;
  return k;
}

// One coordinate of eight points (limbs as six vectors of 64-bit lanes, Montgomery form with
// R = 2^384) -> canonical value in radix 2^52, eight vectors.
inline void from_mont_x8(const __m512i a[6], __m512i r[8]) {
  const Consts& k = K();
  const __m512i mask = _mm512_set1_epi64((long long)kMask52);
  const __m512i zero = _mm512_setzero_si512();
  // t = a << 32 in radix 2^52: REDC with R' = 2^416 then gives a * 2^32 / 2^416 = a / 2^384
  __m512i t[9];
  for (int j = 0; j < 8; j++) {
    const int s = 52 * j - 32;  // first bit of `a` in limb j
    __m512i w;
    if (s < 0) {
      w = _mm512_slli_epi64(a[0], -s);
    } else {
      const int q = s / 64, rr = s % 64;
      w = q < 6 ? _mm512_srli_epi64(a[q], rr) : zero;
      if (rr && q + 1 < 6) w = _mm512_or_si512(w, _mm512_slli_epi64(a[q + 1], 64 - rr));
    }
    t[j] = _mm512_and_si512(w, mask);
  }
  t[8] = zero;
  __m512i P[8];
  for (int j = 0; j < 8; j++) P[j] = _mm512_set1_epi64((long long)k.p[j]);
  const __m512i n0 = _mm512_set1_epi64((long long)k.n0);
  for (int i = 0; i < 8; i++) {
    const __m512i m = _mm512_and_si512(_mm512_madd52lo_epu64(zero, t[0], n0), mask);
    for (int j = 0; j < 8; j++) {
      t[j] = _mm512_madd52lo_epu64(t[j], m, P[j]);
      t[j + 1] = _mm512_madd52hi_epu64(t[j + 1], m, P[j]);
    }
    // t[0] is now a multiple of 2^52 (below 2^52 only in its carry part): shift one limb down
    const __m512i carry = _mm512_srli_epi64(t[0], 52);
    for (int j = 0; j < 8; j++) t[j] = t[j + 1];
    t[0] = _mm512_add_epi64(t[0], carry);
    t[8] = zero;
  }
  // carries, then one conditional subtraction of p (the value is below 2p)
  __m512i c = zero;
  for (int j = 0; j < 8; j++) {
    t[j] = _mm512_add_epi64(t[j], c);
    c = _mm512_srli_epi64(t[j], 52);
    t[j] = _mm512_and_si512(t[j], mask);
  }
  __m512i d[8];
  __m512i borrow = zero;
  for (int j = 0; j < 8; j++) {
    const __m512i x = _mm512_sub_epi64(_mm512_sub_epi64(t[j], P[j]), borrow);
    borrow = _mm512_srli_epi64(x, 63);
    d[j] = _mm512_and_si512(x, mask);
  }
  const __mmask8 ge = _mm512_cmpeq_epi64_mask(borrow, zero);  // no final borrow: t >= p
  for (int j = 0; j < 8; j++) r[j] = _mm512_mask_blend_epi64(ge, t[j], d[j]);
}

// lanes whose canonical value is > (p - 1) / 2, i.e. >= (p + 1) / 2
inline __mmask8 is_larger_x8(const __m512i r[8]) {
  const Consts& k = K();
  const __m512i zero = _mm512_setzero_si512();
  __m512i borrow = zero;
  for (int j = 0; j < 8; j++) {
    const __m512i x = _mm512_sub_epi64(_mm512_sub_epi64(r[j], _mm512_set1_epi64((long long)k.half[j])), borrow);
    borrow = _mm512_srli_epi64(x, 63);
  }
  return _mm512_cmpeq_epi64_mask(borrow, zero);
}
}  // namespace

// n a multiple of 8 is not required: the tail is padded with the last point.  Points at
// infinity (0, 0) are left to the caller (CompressAffineBatch screens them).
void CompressAffineAvx512(const G1Affine* pts, size_t n, uint8_t* out) {
  const __m512i idx = _mm512_setr_epi64(0, 12, 24, 36, 48, 60, 72, 84);  // G1Affine = 12 u64
  for (size_t base = 0; base < n; base += 8) {
    const size_t live = n - base < 8 ? n - base : 8;
    __m512i ix = idx;
    if (live < 8) {  // lanes past the end re-read the last point
      alignas(64) long long v[8];
      for (int l = 0; l < 8; l++) v[l] = 12 * (long long)((size_t)l < live ? l : live - 1);
      ix = _mm512_load_si512(v);
    }
    const long long* src = reinterpret_cast<const long long*>(pts + base);
    __m512i x[6], y[6], xr[8], yr[8];
    for (int i = 0; i < 6; i++) {
      x[i] = _mm512_i64gather_epi64(ix, src + i, 8);
      y[i] = _mm512_i64gather_epi64(ix, src + 6 + i, 8);
    }
    from_mont_x8(x, xr);
    from_mont_x8(y, yr);
    const __mmask8 larger = is_larger_x8(yr);
    // radix 2^52 -> six 64-bit limbs -> big-endian bytes
    alignas(64) uint64_t limb[6][8];
    for (int i = 0; i < 6; i++) {
      const int s = 64 * i;  // first bit of limb i
      const int q = s / 52, rr = s % 52;
      __m512i w = _mm512_srli_epi64(xr[q], rr);
      if (q + 1 < 8) w = _mm512_or_si512(w, _mm512_slli_epi64(xr[q + 1], 52 - rr));
      if (52 - rr + 52 < 64 && q + 2 < 8) w = _mm512_or_si512(w, _mm512_slli_epi64(xr[q + 2], 104 - rr));
      _mm512_store_si512(limb[i], w);
    }
    for (size_t l = 0; l < live; l++) {
      uint8_t* o = out + 48 * (base + l);
      for (int i = 0; i < 6; i++) {
        const uint64_t be = __builtin_bswap64(limb[5 - i][l]);
        memcpy(o + 8 * i, &be, 8);
      }
      o[0] |= 0x80;
      if ((larger >> l) & 1) o[0] |= 0x20;
    }
  }
}

}  // namespace alg
}  // namespace curdle
