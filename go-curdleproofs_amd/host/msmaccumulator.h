// Host mirror of the reference's `msmaccumulator` package
// (/root/reference/msmaccumulator/msmaccumulator.go:11-64): same names,
// argument order and error behaviour, with the final MultiExp (:59) running on
// the GPU through the C ABI (curdle_msm_g1).  Go's (value, error) pairs become
// a Status return + out-params.
#pragma once
#include <stdint.h>

#include <string>
#include <vector>

#include "../csrc/bls12_381.h"
#include "common_rand.h"

namespace curdle {
namespace msmaccumulator {

struct Status {
  bool ok;
  std::string err;  // text of the Go error ("x and v must have the same length", "computing msm: ...")
  int rc;           // CURDLE_* code behind a failed MSM (0 for the structural errors of the Go API)
  static Status OK() { return Status{true, "", 0}; }
  static Status Error(const std::string& e, int code = 0) { return Status{false, e, code}; }
};

class MsmAccumulator {
 public:
  // New(), msmaccumulator.go:16-21
  MsmAccumulator();

  // AccumulateCheck(C, x, v, rand), msmaccumulator.go:23-47.
  // C is a gnark G1Jac (any representative), x Montgomery scalars, v affine bases.
  Status AccumulateCheck(const G1Jac& C, const std::vector<Fr>& x, const std::vector<G1Affine>& v,
                         common::Rand* rand);

  // Same with C already in the library's XYZZ form (the protocol layer's points): skips
  // the normalisation a round trip through G1Jac would cost.
  Status AccumulateCheckXYZZ(const G1XYZZ& C, const std::vector<Fr>& x, const std::vector<G1Affine>& v,
                             common::Rand* rand);

  // The same check with C handed over as the linear combination it was going to be
  // computed from, C = sum_j c_scalars[j] * c_points[j]: instead of evaluating C (one MSM)
  // and alpha * C (one scalar multiplication) now, the terms -alpha * c_scalars[j] join
  // the base / scalar map, so Verify()'s single MSM checks
  //     sum_i alpha x_i v_i - sum_j alpha c_j P_j == A_c
  // which is the reference's equation with alpha * C moved to the other side.  Draws
  // alpha exactly like AccumulateCheck, so the accept bit is identical for every input.
  Status AccumulateCheckDeferred(const std::vector<Fr>& c_scalars, const std::vector<G1Affine>& c_points,
                                 const std::vector<Fr>& x, const std::vector<G1Affine>& v, common::Rand* rand);

  // Fold another accumulator's pending checks into this one (bases shared by both merge,
  // A_c adds): the union is one random linear combination of all their checks, so one
  // Verify() answers "do all of them hold" with one MSM (cross-proof batch verification).
  void Merge(const MsmAccumulator& other);

  // Verify(), msmaccumulator.go:49-64: flatten the map, one MultiExp, Equal(A_c).
  Status Verify(bool* ok);

  // Exported field A_c (msmaccumulator.go:12), kept as XYZZ internally.
  G1XYZZ A_c;

  size_t NumBases() const { return bases_.size(); }
  const std::vector<G1Affine>& Bases() const { return bases_; }
  const std::vector<Fr>& Scalars() const { return scalars_; }

 private:
  // baseScalarMap map[G1Affine]fr.Element (:13): keyed by the 96 key bytes, so
  // a base shared by several checks merges; kept in insertion order (Go's map
  // order is random per run, :53-56, so only the group element is defined).
  void AddTerm(const G1Affine& base, const Fr& scalar);
  size_t Find(const G1Affine& base, bool* found);
  void Grow();
  // open-addressing table over bases_ (slot = index + 1, 0 = empty), keyed by the 96 key
  // bytes; a verification inserts ~1,400 keys, so no per-key allocation
  std::vector<uint32_t> table_;
  std::vector<G1Affine> bases_;
  std::vector<Fr> scalars_;
};

}  // namespace msmaccumulator
}  // namespace curdle
