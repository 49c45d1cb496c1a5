// Small algebra layer for the host-side protocol code: Fr scalars and G1 points as
// value types with operators, the byte encodings gnark uses on the wire, and the one
// function every MSM of the protocol goes through (alg::MultiExp -> the GPU, via the
// C ABI).  Built on bls12_381.h / host_math.h.
//
// Replaces what the reference's protocol packages get from gnark-crypto
// (fr.Element, bls12381.G1Jac / G1Affine and their Bytes / SetBytes, go.mod:6).
#pragma once
#include <stdexcept>
#include <string>
#include <stdint.h>
#include <string.h>

#include <string>
#include <vector>

#include "../csrc/host_math.h"

namespace curdle {
namespace alg {

// A failed call into the MSM entry points (no device, HIP error, ...): distinct from the
// protocol's structural errors so callers can tell "reject" from "could not compute".
struct MsmError : std::runtime_error {
  int rc;
  MsmError(const std::string& what, int code) : std::runtime_error(what), rc(code) {}
};

struct Scalar {
  Fr v;  // Montgomery form, as fr.Element

  static Scalar Zero() {
    Scalar s;
    f_zero(s.v);
    return s;
  }
  static Scalar One() {
    Scalar s;
    f_one(s.v);
    return s;
  }
  static Scalar FromU64(uint64_t x);  // fr.NewElement
  static Scalar FromMont(const uint64_t limbs[4]) {
    Scalar s;
    memcpy(&s.v, limbs, 32);
    return s;
  }
  Scalar operator+(const Scalar& o) const {
    Scalar r;
    fr_add(r.v, v, o.v);
    return r;
  }
  Scalar operator-(const Scalar& o) const {
    Scalar r;
    fr_sub(r.v, v, o.v);
    return r;
  }
  Scalar operator-() const {
    Scalar r;
    f_neg<FrParams>(r.v, v);
    return r;
  }
  Scalar operator*(const Scalar& o) const {
    Scalar r;
    fr_mul(r.v, v, o.v);
    return r;
  }
  Scalar Neg() const { return Zero() - *this; }
  Scalar Inverse() const;          // 0 -> 0, as fr.Element.Inverse
  Scalar Pow(uint64_t e) const;    // fr.Element.Exp with a small exponent
  bool IsZero() const { return f_is_zero(v); }
  bool operator==(const Scalar& o) const { return f_eq(v, o.v); }
  bool operator!=(const Scalar& o) const { return !f_eq(v, o.v); }
  void Canonical(u32 out[8]) const;             // the integer, little-endian limbs (fr.Element.BigInt)
  void Bytes(uint8_t out[32]) const;            // fr.Element.Bytes: big-endian canonical
  static bool SetBytesCanonical(const uint8_t in[32], Scalar* out);  // false if >= r
};

std::vector<Scalar> BatchInvert(const std::vector<Scalar>& xs);      // fr.BatchInvert (zeros stay zero)
Scalar InnerProduct(const std::vector<Scalar>& a, const std::vector<Scalar>& b);  // common.IPA, util.go:26

struct Point {
  G1XYZZ p;
  // Set only by the verifier's lazy proof reader (proto::Reader with `lazy`), for points whose
  // decoding is still running on the GPU: `wire` is the 48-byte record the point comes from --
  // what the transcript absorbs, so it needs no coordinates -- and `pending` >= 0 says the
  // coordinates are NOT known yet (the record's index in the proto::PointDecoder).  A pending
  // point can be hashed and listed as a base; arithmetic on it is a logic error.  Both die with
  // the verification call: results of operators never carry them.
  const uint8_t* wire = nullptr;
  int32_t pending = -1;
  void NeedValue() const {
    if (pending >= 0) throw std::logic_error("arithmetic on a point that is still being decoded");
  }

  static Point Infinity() {
    Point r;
    g1_set_inf(r.p);
    return r;
  }
  static Point FromAffine(const G1Affine& a) {
    Point r;
    g1_from_affine(r.p, a);
    return r;
  }
  static Point FromJac(const uint64_t jac[18]);
  static Point Generator();
  Point operator+(const Point& o) const;
  Point Neg() const {
    NeedValue();
    Point r;
    r.p = p;
    if (!g1_is_inf(r.p)) fp_neg(r.p.y, r.p.y);
    return r;
  }
  Point operator-(const Point& o) const { return *this + o.Neg(); }
  Point Mul(const Scalar& k) const;   // ScalarMultiplication with FrToBigInt(k)
  bool IsInfinity() const {
    NeedValue();
    return g1_is_inf(p);
  }
  bool operator==(const Point& o) const;   // G1Jac.Equal
  G1Affine Affine() const;
  void Jac(uint64_t out[18]) const {
    NeedValue();
    g1_to_canonical_jac(out, p);
  }
  // gnark G1Affine.Bytes(): 48 bytes, big-endian x, flags in the top three bits
  // (0x80 compressed, 0x40 infinity, 0x20 y is the lexicographically larger root).  A point
  // that carries its wire record hands that back (a valid record is its point's only encoding).
  void Compressed(uint8_t out[48]) const;
  // G1Affine.SetBytes: false on a malformed encoding or a point not on the curve;
  // the (slow, [r]P) subgroup check is optional.
  static bool FromCompressed(const uint8_t in[48], Point* out, bool subgroup_check);
};

std::vector<G1Affine> BatchToAffine(const std::vector<Point>& pts);   // BatchJacobianToAffineG1

// gnark G1Affine.Bytes() of an affine point without the round trip through Point: two
// Montgomery reductions (x to canonical bytes, y for the sign bit).
void CompressAffine(const G1Affine& a, uint8_t out[48]);
// n points -> 48 n bytes.  Eight at a time with AVX-512 IFMA where the CPU has it (the verifier
// hashes 4 ell instance points per verification), the scalar routine otherwise and for the tail
// of fewer than eight.
void CompressAffineBatch(const G1Affine* pts, size_t n, uint8_t* out);

// k * P for a base that never changes (CRS points): 8-bit windows precomputed once, then at
// most 32 mixed additions per multiplication instead of ~255 doublings + 64 additions.
class FixedBase {
 public:
  explicit FixedBase(const G1Affine& p);
  Point Mul(const Scalar& k) const;
 private:
  std::vector<G1Affine> table_;  // 32 x 255
};

// common.MultiExp: every MSM of the protocol code funnels through
// here and runs on the GPU (curdle_msm_g1).  Throws std::runtime_error on a length
// mismatch or a device error, with the Go error text.
Point MultiExp(const std::vector<G1Affine>& points, const std::vector<Scalar>& scalars);
// Several MSMs in one GPU pass (curdle_msm_g1_batch): results[i] = MultiExp(points[i], scalars[i]).
std::vector<Point> MultiExpBatch(const std::vector<const std::vector<G1Affine>*>& points,
                                 const std::vector<const std::vector<Scalar>*>& scalars);

// One scalar vector against several base sets (curdle_msm_g1_multi): results[i] = MultiExp(*sets[i],
// scalars), with the scalars uploaded, recoded and bucket-sorted once for all sets
// (samemultiscalarargument.go:64-70; curdleproof.go:110,:114).
std::vector<Point> MultiExpShared(const std::vector<const std::vector<G1Affine>*>& sets,
                                  const std::vector<Scalar>& scalars);

// out[i] = addends[i] + scalars[i] * points[i]; scalars.size() is points.size(), or 1 for one
// scalar shared by all; addends may be null.  The independent scalar multiplications that
// dominate the prover: a batch of at least kScalarMulBatchMin goes to
// the GPU (curdle_g1_scalar_mul_batch), smaller ones run on the host one by one.
static constexpr size_t kScalarMulBatchMin = 24;
std::vector<G1Affine> ScalarMulBatch(const std::vector<G1Affine>& points, const std::vector<Scalar>& scalars,
                                     const std::vector<G1Affine>* addends = nullptr);

}  // namespace alg
}  // namespace curdle
