// Fr / G1 value types, wire encodings and the MSM funnel -- see algebra.h.
#include "algebra.h"

#include <stdlib.h>

#include "host_ops.h"

#include <stdexcept>

#include "../../include/curdle_msm.h"

namespace curdle {
namespace alg {

// single-point group operations go through the ISA-dispatched builds (host_ops.cpp)
static void ScalarMulImpl(G1XYZZ& r, const G1XYZZ& p, const u32* k) { curdle_host_scalar_mul(&r, &p, k); }
static void AddImpl(G1XYZZ& acc, const G1XYZZ& b) { curdle_host_add(&acc, &b); }
static bool ToAffineImpl(G1Affine& out, const G1XYZZ& p) { return curdle_host_to_affine(&out, &p) != 0; }
static void FpPowImpl(Fp& r, const Fp& a, const u32* e) { curdle_host_fp_pow(&r, &a, e); }
static void FpFromMontImpl(Fp& r, const Fp& a) { curdle_host_fp_from_mont(&r, &a); }

// ------------------------------------------------------------------ Scalar ---
Scalar Scalar::FromU64(uint64_t x) {
  Fr c;
  f_zero(c);
  c.l[0] = (u32)x;
  c.l[1] = (u32)(x >> 32);
  Scalar s;
  fr_to_mont(s.v, c);
  return s;
}

void Scalar::Canonical(u32 out[8]) const {
  Fr c;
  f_from_mont<FrParams>(c, v);
  memcpy(out, c.l, 32);
}

void Scalar::Bytes(uint8_t out[32]) const {
  u32 c[8];
  Canonical(c);
  for (int i = 0; i < 8; i++) {
    u32 w = c[7 - i];
    out[4 * i] = (uint8_t)(w >> 24);
    out[4 * i + 1] = (uint8_t)(w >> 16);
    out[4 * i + 2] = (uint8_t)(w >> 8);
    out[4 * i + 3] = (uint8_t)w;
  }
}

bool Scalar::SetBytesCanonical(const uint8_t in[32], Scalar* out) {
  Fr c;
  for (int i = 0; i < 8; i++) {
    const uint8_t* q = in + 28 - 4 * i;
    c.l[i] = ((u32)q[0] << 24) | ((u32)q[1] << 16) | ((u32)q[2] << 8) | (u32)q[3];
  }
  for (int i = 7; i >= 0; i--) {
    u32 m = FrParams::mod(i);
    if (c.l[i] < m) break;
    if (c.l[i] > m || i == 0) return false;  // >= r
  }
  fr_to_mont(out->v, c);
  return true;
}

Scalar Scalar::Inverse() const {
  if (IsZero()) return Zero();
  static const u32 e[8] = {0xffffffffu, 0xfffffffeu, 0xfffe5bfeu, 0x53bda402u,
                           0x09a1d805u, 0x3339d808u, 0x299d7d48u, 0x73eda753u};  // r - 2
  Scalar acc = One();
  for (int i = 255; i >= 0; i--) {
    acc = acc * acc;
    if ((e[i / 32] >> (i % 32)) & 1) acc = acc * *this;
  }
  return acc;
}

Scalar Scalar::Pow(uint64_t e) const {
  Scalar acc = One(), base = *this;
  while (e) {
    if (e & 1) acc = acc * base;
    base = base * base;
    e >>= 1;
  }
  return acc;
}

std::vector<Scalar> BatchInvert(const std::vector<Scalar>& xs) {
  // Montgomery's trick; zero entries are skipped and stay zero (fr.BatchInvert).
  std::vector<Scalar> out(xs.size());
  std::vector<Scalar> pref(xs.size());
  Scalar acc = Scalar::One();
  for (size_t i = 0; i < xs.size(); i++) {
    pref[i] = acc;
    if (!xs[i].IsZero()) acc = acc * xs[i];
  }
  Scalar inv = acc.Inverse();
  for (size_t i = xs.size(); i-- > 0;) {
    if (xs[i].IsZero()) {
      out[i] = Scalar::Zero();
      continue;
    }
    out[i] = inv * pref[i];
    inv = inv * xs[i];
  }
  return out;
}

Scalar InnerProduct(const std::vector<Scalar>& a, const std::vector<Scalar>& b) {
  if (a.size() != b.size()) throw std::runtime_error("IPA: len(a) != len(b)");  // util.go:27-29
  Scalar acc = Scalar::Zero();
  for (size_t i = 0; i < a.size(); i++) acc = acc + a[i] * b[i];
  return acc;
}

// ------------------------------------------------------------------- Point ---
static void fp_from_be48(Fp& canonical, const uint8_t in[48], uint8_t top_mask) {
  for (int i = 0; i < 12; i++) {
    const uint8_t* q = in + 44 - 4 * i;
    u32 b0 = q[0];
    if (i == 11) b0 &= top_mask;
    canonical.l[i] = (b0 << 24) | ((u32)q[1] << 16) | ((u32)q[2] << 8) | (u32)q[3];
  }
}
static bool fp_lt(const Fp& a, const u32* m) {
  for (int i = 11; i >= 0; i--) {
    if (a.l[i] != m[i]) return a.l[i] < m[i];
  }
  return false;
}
static const u32 kFpR2[12] = {0x1c341746u, 0xf4df1f34u, 0x09d104f1u, 0x0a76e6a6u, 0x4c95b6d5u, 0x8de5476cu,
                              0x939d83c0u, 0x67eb88a9u, 0xb519952du, 0x9a793e85u, 0x92cae3aau, 0x11988fe5u};
static const u32 kFpHalf[12] = {0xffffd555u, 0xdcff7fffu, 0x58a9ffffu, 0x0f55ffffu, 0x7b587b12u, 0xb3986950u,
                                0x79c2895fu, 0xb23ba5c2u, 0x21a5d66bu, 0x258dd3dbu, 0x1cbff34du, 0x0d0088f5u};  // (p-1)/2
static const u32 kFpSqrtExp[12] = {0xffffeaabu, 0xee7fbfffu, 0xac54ffffu, 0x07aaffffu, 0x3dac3d89u, 0xd9cc34a8u,
                                   0x3ce144afu, 0xd91dd2e1u, 0x90d2eb35u, 0x92c6e9edu, 0x8e5ff9a6u, 0x0680447au};  // (p+1)/4

static bool y_is_larger(const Fp& y_mont) {
  Fp c;
  FpFromMontImpl(c, y_mont);
  return !fp_lt(c, kFpHalf) && !f_eq(c, *reinterpret_cast<const Fp*>(kFpHalf));  // y > (p-1)/2
}

Point Point::operator+(const Point& o) const {
  NeedValue();
  o.NeedValue();
  Point r;
  r.p = p;
  AddImpl(r.p, o.p);
  return r;
}

G1Affine Point::Affine() const {
  NeedValue();
  G1Affine a;
  ToAffineImpl(a, p);
  return a;
}

bool Point::operator==(const Point& o) const {
  NeedValue();
  o.NeedValue();
  return curdle_host_equal(&p, &o.p) != 0;
}

Point Point::FromJac(const uint64_t jac[18]) {
  G1Jac j;
  memcpy(&j, jac, sizeof(j));
  Point r;
  g1_from_jac(r.p, j);
  return r;
}

Point Point::Generator() {
  G1Affine g;
  g1_generator(g);
  return FromAffine(g);
}

Point Point::Mul(const Scalar& k) const {
  NeedValue();
  u32 c[8];
  k.Canonical(c);
  Point r;
  ScalarMulImpl(r.p, p, c);
  return r;
}

void Point::Compressed(uint8_t out[48]) const {
  if (wire) {
    memcpy(out, wire, 48);
    return;
  }
  G1Affine a;
  if (!ToAffineImpl(a, p)) {
    memset(out, 0, 48);
    out[0] = 0xc0;
    return;
  }
  Fp xc;
  FpFromMontImpl(xc, a.x);
  for (int i = 0; i < 12; i++) {
    u32 w = xc.l[11 - i];
    out[4 * i] = (uint8_t)(w >> 24);
    out[4 * i + 1] = (uint8_t)(w >> 16);
    out[4 * i + 2] = (uint8_t)(w >> 8);
    out[4 * i + 3] = (uint8_t)w;
  }
  out[0] |= 0x80;
  if (y_is_larger(a.y)) out[0] |= 0x20;
}

void CompressAffine(const G1Affine& a, uint8_t out[48]) {
  if (g1_affine_is_inf(a)) {
    memset(out, 0, 48);
    out[0] = 0xc0;
    return;
  }
  Fp xc;
  FpFromMontImpl(xc, a.x);
  for (int i = 0; i < 12; i++) {
    u32 w = xc.l[11 - i];
    out[4 * i] = (uint8_t)(w >> 24);
    out[4 * i + 1] = (uint8_t)(w >> 16);
    out[4 * i + 2] = (uint8_t)(w >> 8);
    out[4 * i + 3] = (uint8_t)w;
  }
  out[0] |= 0x80;
  if (y_is_larger(a.y)) out[0] |= 0x20;
}

void CompressAffineAvx512(const G1Affine* pts, size_t n, uint8_t* out);  // compress_avx512.cpp

void CompressAffineBatch(const G1Affine* pts, size_t n, uint8_t* out) {
  static const bool ifma = [] {
    __builtin_cpu_init();
    return __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512ifma") && __builtin_cpu_supports("avx512vl") &&
           __builtin_cpu_supports("avx512bw") && __builtin_cpu_supports("avx512dq");
  }();
  if (!ifma || n < 8) {
    for (size_t i = 0; i < n; i++) CompressAffine(pts[i], out + 48 * i);
    return;
  }
  CompressAffineAvx512(pts, n, out);
  for (size_t i = 0; i < n; i++)  // the vector routine knows no point at infinity
    if (g1_affine_is_inf(pts[i])) CompressAffine(pts[i], out + 48 * i);
}

FixedBase::FixedBase(const G1Affine& p) : table_(32 * 255) { curdle_host_fixed_base_table(table_.data(), &p); }
Point FixedBase::Mul(const Scalar& k) const {
  u32 kc[8];
  k.Canonical(kc);
  Point r;
  curdle_host_fixed_base_mul(&r.p, table_.data(), kc);
  return r;
}

bool Point::FromCompressed(const uint8_t in[48], Point* out, bool subgroup_check) {
  const uint8_t flags = in[0] & 0xe0;
  if (!(flags & 0x80)) return false;  // only the compressed form is used on this wire
  if (flags & 0x40) {
    if (flags & 0x20) return false;
    if (in[0] & 0x1f) return false;
    for (int i = 1; i < 48; i++)
      if (in[i]) return false;
    *out = Infinity();
    return true;
  }
  Fp xc;
  fp_from_be48(xc, in, 0x1f);
  u32 pm[12];
  for (int i = 0; i < 12; i++) pm[i] = FpParams::mod(i);
  if (!fp_lt(xc, pm)) return false;
  Fp x, r2, x3, rhs, four, y, y2;
  memcpy(&r2, kFpR2, 48);
  fp_mul(x, xc, r2);  // to Montgomery
  fp_sqr(x3, x);
  fp_mul(x3, x3, x);
  f_one(four);
  fp_dbl(four, four);
  fp_dbl(four, four);
  fp_add(rhs, x3, four);  // x^3 + 4
  FpPowImpl(y, rhs, kFpSqrtExp);
  fp_sqr(y2, y);
  if (!f_eq(y2, rhs)) return false;  // not on the curve
  if (y_is_larger(y) != ((flags & 0x20) != 0)) fp_neg(y, y);
  G1Affine a;
  a.x = x;
  a.y = y;
  *out = FromAffine(a);
  if (subgroup_check) {
    if (!curdle_host_in_subgroup(&out->p)) return false;  // endomorphism test, host_math.h
  }
  return true;
}

std::vector<G1Affine> BatchToAffine(const std::vector<Point>& pts) {
  std::vector<G1Affine> out(pts.size());
  for (size_t i = 0; i < pts.size(); i++) ToAffineImpl(out[i], pts[i].p);
  return out;
}

// --------------------------------------------------------------------- MSM ---
static MsmError msm_error(int rc) {
  char buf[256];
  curdle_last_error(buf, sizeof(buf));
  return MsmError(std::string("computing msm: ") + buf + " (rc " + std::to_string(rc) + ")", rc);
}

Point MultiExp(const std::vector<G1Affine>& points, const std::vector<Scalar>& scalars) {
  if (points.size() != scalars.size()) throw std::runtime_error("computing msm: len(points) != len(scalars)");
  uint64_t out[18];
  int rc = curdle_msm_g1(reinterpret_cast<const uint64_t*>(points.data()),
                         reinterpret_cast<const uint64_t*>(scalars.data()), points.size(), out);
  if (rc != CURDLE_OK) throw msm_error(rc);
  return Point::FromJac(out);
}

std::vector<Point> MultiExpBatch(const std::vector<const std::vector<G1Affine>*>& points,
                                 const std::vector<const std::vector<Scalar>*>& scalars) {
  if (points.size() != scalars.size()) throw std::runtime_error("computing msm: batch shape");
  const size_t k = points.size();
  std::vector<size_t> off(k + 1, 0);
  for (size_t j = 0; j < k; j++) {
    if (points[j]->size() != scalars[j]->size()) throw std::runtime_error("computing msm: len(points) != len(scalars)");
    off[j + 1] = off[j] + points[j]->size();
  }
  std::vector<G1Affine> P(off[k]);
  std::vector<Scalar> S(off[k]);
  for (size_t j = 0; j < k; j++) {
    std::copy(points[j]->begin(), points[j]->end(), P.begin() + off[j]);
    std::copy(scalars[j]->begin(), scalars[j]->end(), S.begin() + off[j]);
  }
  std::vector<uint64_t> out(18 * k);
  int rc = curdle_msm_g1_batch(reinterpret_cast<const uint64_t*>(P.data()), reinterpret_cast<const uint64_t*>(S.data()),
                               off.data(), k, out.data());
  if (rc != CURDLE_OK) throw msm_error(rc);
  std::vector<Point> res(k);
  for (size_t j = 0; j < k; j++) res[j] = Point::FromJac(out.data() + 18 * j);
  return res;
}

std::vector<Point> MultiExpShared(const std::vector<const std::vector<G1Affine>*>& sets,
                                  const std::vector<Scalar>& scalars) {
  const size_t k = sets.size();
  std::vector<const uint64_t*> ptrs(k);
  for (size_t j = 0; j < k; j++) {
    if (sets[j]->size() != scalars.size()) throw std::runtime_error("computing msm: len(points) != len(scalars)");
    ptrs[j] = reinterpret_cast<const uint64_t*>(sets[j]->data());
  }
  std::vector<uint64_t> out(18 * k);
  if (scalars.empty()) {
    std::vector<Point> res(k, Point::Infinity());
    return res;
  }
  int rc = curdle_msm_g1_multi(ptrs.data(), k, reinterpret_cast<const uint64_t*>(scalars.data()), scalars.size(), out.data());
  if (rc != CURDLE_OK) throw msm_error(rc);
  std::vector<Point> res(k);
  for (size_t j = 0; j < k; j++) res[j] = Point::FromJac(out.data() + 18 * j);
  return res;
}

std::vector<G1Affine> ScalarMulBatch(const std::vector<G1Affine>& points, const std::vector<Scalar>& scalars,
                                     const std::vector<G1Affine>* addends) {
  const size_t n = points.size();
  if (scalars.size() != n && scalars.size() != 1) throw std::runtime_error("scalar mul batch: len(scalars) must be n or 1");
  if (addends && addends->size() != n) throw std::runtime_error("scalar mul batch: len(addends) != len(points)");
  std::vector<G1Affine> out(n);
  if (n == 0) return out;
  if (n >= kScalarMulBatchMin) {
    int rc = curdle_g1_scalar_mul_batch(reinterpret_cast<const uint64_t*>(points.data()),
                                        reinterpret_cast<const uint64_t*>(scalars.data()), scalars.size(),
                                        addends ? reinterpret_cast<const uint64_t*>(addends->data()) : nullptr, n,
                                        reinterpret_cast<uint64_t*>(out.data()));
    if (rc != CURDLE_OK) throw msm_error(rc);
    return out;
  }
  for (size_t i = 0; i < n; i++) {  // a handful of points: a kernel launch would cost more than it saves
    Point r = Point::FromAffine(points[i]).Mul(scalars.size() == 1 ? scalars[0] : scalars[i]);
    if (addends) r = r + Point::FromAffine((*addends)[i]);
    out[i] = r.Affine();
  }
  return out;
}

}  // namespace alg
}  // namespace curdle
