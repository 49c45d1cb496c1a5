// Host-side G1 / Fr helpers of libcurdlemsm.so, built on the same limb
// arithmetic as the kernels (bls12_381.h).  Used for the O(256)-doubling window
// combine after the GPU phases, the canonical Jacobian output, and the
// msmaccumulator mirror (one 255-bit scalar multiplication per
// AccumulateCheck, msmaccumulator/msmaccumulator.go:44).
//
// Product code: nothing here touches oracle/.
#pragma once
#include <string.h>

#include "bls12_381.h"

namespace curdle {

// a^e for a multi-limb exponent e (little-endian 32-bit limbs), Montgomery in/out.
// Fixed 4-bit windows, most significant first (square roots and inversions over Fp are
// 381-bit exponentiations: 4 squarings + at most one product per nibble).
inline void fp_pow(Fp& r, const Fp& a, const u32* e, int nlimbs) {
  Fp tab[16];
  f_one(tab[0]);
  tab[1] = a;
  for (int i = 2; i < 16; i++) fp_mul(tab[i], tab[i - 1], a);
  Fp acc;
  f_one(acc);
  bool started = false;
  for (int i = nlimbs * 8 - 1; i >= 0; i--) {
    if (started)
      for (int k = 0; k < 4; k++) fp_sqr(acc, acc);
    const u32 w = (e[i / 8] >> (4 * (i % 8))) & 15u;
    if (w) {
      fp_mul(acc, acc, tab[w]);
      started = true;
    }
  }
  r = acc;
}

// r = a^-1 (a != 0) by Fermat: a^(p-2).
inline void fp_inv(Fp& r, const Fp& a) {
  u32 e[12];
  for (int i = 0; i < 12; i++) e[i] = FpParams::mod(i);
  e[0] -= 2;  // p ends in ...aaab, no borrow
  fp_pow(r, a, e, 12);
}

// XYZZ -> affine; returns false for infinity (out set to (0,0), gnark's encoding).
inline bool g1_to_affine(G1Affine& out, const G1XYZZ& p) {
  if (g1_is_inf(p)) {
    f_zero(out.x);
    f_zero(out.y);
    return false;
  }
  Fp one;
  f_one(one);
  if (f_eq(p.zz, one) && f_eq(p.zzz, one)) {  // already normalised (affine inputs, decoded points)
    out.x = p.x;
    out.y = p.y;
    return true;
  }
  // invert ZZ*ZZZ once: 1/ZZ = ZZZ / (ZZ ZZZ), 1/ZZZ = ZZ / (ZZ ZZZ)
  Fp t, ti, izz, izzz;
  fp_mul(t, p.zz, p.zzz);
  fp_inv(ti, t);
  fp_mul(izz, ti, p.zzz);
  fp_mul(izzz, ti, p.zz);
  fp_mul(out.x, p.x, izz);
  fp_mul(out.y, p.y, izzz);
  return true;
}

// n XYZZ points -> affine with ONE inversion (Montgomery's trick over t_i = ZZ_i ZZZ_i:
// 1/ZZ = ZZZ / t, 1/ZZZ = ZZ / t); infinities (ZZ = 0) become (0, 0).  ~8 products per point.
inline void g1_batch_to_affine(G1Affine* out, const G1XYZZ* in, size_t n) {
  if (n == 0) return;
  Fp* prefix = new Fp[n];  // prefix[i] = prod of t_j over finite points j <= i
  Fp run;
  f_one(run);
  for (size_t i = 0; i < n; i++) {
    if (!g1_is_inf(in[i])) {
      Fp t;
      fp_mul(t, in[i].zz, in[i].zzz);
      fp_mul(run, run, t);
    }
    prefix[i] = run;
  }
  Fp inv;
  fp_inv(inv, run);  // run != 0: a product of non-zero field elements (or one)
  for (size_t i = n; i-- > 0;) {
    if (g1_is_inf(in[i])) {
      f_zero(out[i].x);
      f_zero(out[i].y);
      continue;
    }
    Fp t, ti, izz, izzz, before;
    if (i > 0) before = prefix[i - 1]; else f_one(before);
    fp_mul(ti, inv, before);            // 1 / t_i
    fp_mul(t, in[i].zz, in[i].zzz);
    fp_mul(inv, inv, t);                // drop t_i from the running inverse
    fp_mul(izz, ti, in[i].zzz);
    fp_mul(izzz, ti, in[i].zz);
    fp_mul(out[i].x, in[i].x, izz);
    fp_mul(out[i].y, in[i].y, izzz);
  }
  delete[] prefix;
}

// gnark G1Jac (X, Y, Z), any representative -> XYZZ.
inline void g1_from_jac(G1XYZZ& r, const G1Jac& j) {
  if (f_is_zero(j.z)) {
    g1_set_inf(r);
    return;
  }
  r.x = j.x;
  r.y = j.y;
  fp_sqr(r.zz, j.z);
  fp_mul(r.zzz, r.zz, j.z);
}

// Canonical Jacobian the C ABI returns: (x, y, 1), or (1, 1, 0) for infinity.
inline void g1_to_canonical_jac(u64 out[18], const G1XYZZ& p) {
  G1Affine a;
  G1Jac j;
  if (g1_to_affine(a, p)) {
    j.x = a.x;
    j.y = a.y;
    f_one(j.z);
  } else {
    f_one(j.x);
    f_one(j.y);
    f_zero(j.z);
  }
  memcpy(out, &j, sizeof(j));
}

// r = k * p, k canonical (not Montgomery) little-endian limbs; fixed 4-bit windows,
// most significant first: 4 doublings + at most one addition per window, after 14
// additions to tabulate 1p .. 15p.  Host only.
inline void g1_scalar_mul(G1XYZZ& r, const G1XYZZ& p, const u32* k, int nlimbs) {
  G1XYZZ tab[16];
  g1_set_inf(tab[0]);
  tab[1] = p;
  tab[2] = p;
  g1_dbl(tab[2]);
  for (int i = 3; i < 16; i++) {
    tab[i] = tab[i - 1];
    g1_add(tab[i], p);
  }
  G1XYZZ acc;
  g1_set_inf(acc);
  for (int i = nlimbs * 8 - 1; i >= 0; i--) {
    if (!g1_is_inf(acc))
      for (int d = 0; d < 4; d++) g1_dbl(acc);
    u32 w = (k[i / 8] >> (4 * (i % 8))) & 15u;
    if (w) g1_add(acc, tab[w]);
  }
  r = acc;
}

// r = k * p for a canonical k (< r), through the GLV split the kernels use (bls12_381.h
// glv_split): k p = +-k1 p +- k2 phi(p) with 127-bit halves, one chain of 127 doublings with
// 4-bit windows of both halves instead of 255 -- 58 against 88 us on a host core.  A scalar
// that is not below r takes the plain routine.
inline void g1_scalar_mul_glv(G1XYZZ& r, const G1XYZZ& p, const u32* k8) {
  static const u32 kR[8] = {0x00000001u, 0xffffffffu, 0xfffe5bfeu, 0x53bda402u, 0x09a1d805u, 0x3339d808u, 0x299d7d48u, 0x73eda753u};
  bool below = false;
  for (int i = 7; i >= 0; i--) {
    if (k8[i] != kR[i]) {
      below = k8[i] < kR[i];
      break;
    }
  }
  if (!below || g1_is_inf(p)) {
    g1_scalar_mul(r, p, k8, 8);
    return;
  }
  static const u32 kBeta[12] = {0x8671f071u, 0xcd03c9e4u, 0x1fcda5d2u, 0x5dab2246u, 0xd3851b95u, 0x587042afu,
                                0x01bacb9eu, 0x8eb60ebeu, 0x83d050d2u, 0x03f97d6eu, 0x54638741u, 0x18f02065u};
  Fr k;
  for (int i = 0; i < 8; i++) k.l[i] = k8[i];
  u32 a[4], b[4], neg_a, neg_b;
  glv_split(k, a, b, neg_a, neg_b);
  // tables of 1..15 times +-p and +-phi(p); phi is linear, so the second table is the first
  // with x scaled by beta (and the sign adjusted)
  G1XYZZ ta[16], tb[16];
  g1_set_inf(ta[0]);
  ta[1] = p;
  if (neg_a) g1_neg(ta[1]);
  ta[2] = ta[1];
  g1_dbl(ta[2]);
  for (int i = 3; i < 16; i++) {
    ta[i] = ta[i - 1];
    g1_add(ta[i], ta[1]);
  }
  Fp beta;
  for (int i = 0; i < 12; i++) beta.l[i] = kBeta[i];
  g1_set_inf(tb[0]);
  for (int i = 1; i < 16; i++) {
    tb[i] = ta[i];
    fp_mul(tb[i].x, tb[i].x, beta);
    if ((neg_a != 0) != (neg_b != 0)) g1_neg(tb[i]);
  }
  G1XYZZ acc;
  g1_set_inf(acc);
  for (int i = 31; i >= 0; i--) {  // 128 bits, 4 at a time
    if (!g1_is_inf(acc))
      for (int d = 0; d < 4; d++) g1_dbl(acc);
    const u32 wa = (a[i / 8] >> (4 * (i % 8))) & 15u, wb = (b[i / 8] >> (4 * (i % 8))) & 15u;
    if (wa) g1_add(acc, ta[wa]);
    if (wb) g1_add(acc, tb[wb]);
  }
  r = acc;
}

// Membership in the prime-order subgroup G1 for a point already known to be on the curve
// (what gnark's Decoder / SetBytes check after decompression).  With beta the cube root of
// unity for which phi(x, y) = (beta x, y) acts on G1 as [z^2 - 1] (z = -0xd201000000010000
// the curve parameter, r = z^4 - z^2 + 1), P is in G1 iff [z^2] phi(P) + P is the point at
// infinity: two 64-bit scalar multiplications (weight 6) instead of one by the 255-bit r.
inline bool g1_in_subgroup(const G1XYZZ& p) {
  if (g1_is_inf(p)) return true;
  static const u32 kBeta[12] = {0x8671f071u, 0xcd03c9e4u, 0x1fcda5d2u, 0x5dab2246u, 0xd3851b95u, 0x587042afu,
                                0x01bacb9eu, 0x8eb60ebeu, 0x83d050d2u, 0x03f97d6eu, 0x54638741u, 0x18f02065u};
  Fp beta;
  for (int i = 0; i < 12; i++) beta.l[i] = kBeta[i];
  G1XYZZ q = p;
  fp_mul(q.x, q.x, beta);
  const u64 z = 0xd201000000010000ull;  // |z|; the sign cancels in z^2
  for (int rep = 0; rep < 2; rep++) {
    G1XYZZ acc = q;  // top bit (63) is set
    for (int i = 62; i >= 0; i--) {
      g1_dbl(acc);
      if ((z >> i) & 1) g1_add(acc, q);
    }
    q = acc;
  }
  g1_add(q, p);
  return g1_is_inf(q);
}

// Projective equality of two XYZZ points (gnark G1Jac.Equal, msmaccumulator.go:63).
inline bool g1_equal(const G1XYZZ& a, const G1XYZZ& b) {
  bool ia = g1_is_inf(a), ib = g1_is_inf(b);
  if (ia || ib) return ia && ib;
  Fp l, r;
  fp_mul(l, a.x, b.zz);
  fp_mul(r, b.x, a.zz);
  if (!f_eq(l, r)) return false;
  fp_mul(l, a.y, b.zzz);
  fp_mul(r, b.y, a.zzz);
  return f_eq(l, r);
}

// The G1 generator (bls12381.Generators(), used by common/rand.go:27).
inline void g1_generator(G1Affine& g) {
  static const u64 gx[6] = {0x5cb38790fd530c16ull, 0x7817fc679976fff5ull, 0x154f95c7143ba1c1ull,
                            0xf0ae6acdf3d0e747ull, 0xedce6ecc21dbf440ull, 0x120177419e0bfb75ull};
  static const u64 gy[6] = {0xbaac93d50ce72271ull, 0x8c22631a7918fd8eull, 0xdd595f13570725ceull,
                            0x51ac582950405194ull, 0x0e1c8c3fad0059c0ull, 0x0bbc3efc5008a26aull};
  memcpy(&g.x, gx, 48);
  memcpy(&g.y, gy, 48);
}

// canonical integer (little-endian limbs, < r) -> Montgomery fr.Element
inline void fr_to_mont(Fr& r, const Fr& canonical) {
  // multiply by R^2 mod r
  static const u32 r2[8] = {0xf3f29c6du, 0xc999e990u, 0x87925c23u, 0x2b6cedcbu,
                            0x7254398fu, 0x05d31496u, 0x9f59ff11u, 0x0748d9d9u};
  Fr rr;
  for (int i = 0; i < 8; i++) rr.l[i] = r2[i];
  fr_mul(r, canonical, rr);
}

}  // namespace curdle
