// msmaccumulator mirror -- see msmaccumulator.h.
// Reference: /root/reference/msmaccumulator/msmaccumulator.go.
#include "msmaccumulator.h"

#include <string.h>

#include "../../include/curdle_msm.h"
#include "../csrc/host_math.h"
#include "host_ops.h"

namespace curdle {
namespace msmaccumulator {

// alpha * C and the addition into A_c, through the ISA-dispatched builds (host_ops.cpp)
static void MulAdd(G1XYZZ& acc, const G1XYZZ& c, const u32* k) {
  G1XYZZ t;
  curdle_host_scalar_mul(&t, &c, k);
  curdle_host_add(&acc, &t);
}

MsmAccumulator::MsmAccumulator() { g1_set_inf(A_c); }

Status MsmAccumulator::AccumulateCheck(const G1Jac& C, const std::vector<Fr>& x, const std::vector<G1Affine>& v,
                                       common::Rand* rand) {
  G1XYZZ Cx;
  g1_from_jac(Cx, C);
  return AccumulateCheckXYZZ(Cx, x, v, rand);
}

Status MsmAccumulator::AccumulateCheckXYZZ(const G1XYZZ& Cx, const std::vector<Fr>& x, const std::vector<G1Affine>& v,
                                           common::Rand* rand) {
  if (v.size() != x.size()) return Status::Error("x and v must have the same length");  // :28-30

  Fr alpha;
  rand->GetFr(alpha);  // :32

  Fr tmp;
  for (size_t i = 0; i < v.size(); i++) {  // :38-43
    fr_mul(tmp, alpha, x[i]);
    AddTerm(v[i], tmp);
  }

  // A_c += alpha * C  (:44, ScalarMultiplication with the canonical big.Int of alpha)
  Fr alpha_c;
  f_from_mont<FrParams>(alpha_c, alpha);
  MulAdd(A_c, Cx, alpha_c.l);
  return Status::OK();
}

void MsmAccumulator::AddTerm(const G1Affine& base, const Fr& scalar) {
  std::string key(reinterpret_cast<const char*>(&base), sizeof(G1Affine));
  auto it = index_.find(key);
  if (it == index_.end()) {
    index_.emplace(std::move(key), bases_.size());
    bases_.push_back(base);
    scalars_.push_back(scalar);
  } else {
    fr_add(scalars_[it->second], scalars_[it->second], scalar);
  }
}

Status MsmAccumulator::AccumulateCheckDeferred(const std::vector<Fr>& c_scalars, const std::vector<G1Affine>& c_points,
                                               const std::vector<Fr>& x, const std::vector<G1Affine>& v,
                                               common::Rand* rand) {
  if (v.size() != x.size()) return Status::Error("x and v must have the same length");
  if (c_scalars.size() != c_points.size()) return Status::Error("c_scalars and c_points must have the same length");
  Fr alpha, tmp;
  rand->GetFr(alpha);
  for (size_t i = 0; i < v.size(); i++) {
    fr_mul(tmp, alpha, x[i]);
    AddTerm(v[i], tmp);
  }
  for (size_t j = 0; j < c_points.size(); j++) {
    if (g1_affine_is_inf(c_points[j])) continue;  // contributes nothing; (0,0) is not a valid MSM base key
    fr_mul(tmp, alpha, c_scalars[j]);
    f_neg<FrParams>(tmp, tmp);
    AddTerm(c_points[j], tmp);
  }
  return Status::OK();
}

void MsmAccumulator::Merge(const MsmAccumulator& other) {
  for (size_t i = 0; i < other.bases_.size(); i++) AddTerm(other.bases_[i], other.scalars_[i]);
  curdle_host_add(&A_c, &other.A_c);
}

Status MsmAccumulator::Verify(bool* ok) {
  *ok = false;
  uint64_t out[CURDLE_G1_JAC_U64];
  int rc = curdle_msm_g1(reinterpret_cast<const uint64_t*>(bases_.data()),
                         reinterpret_cast<const uint64_t*>(scalars_.data()), bases_.size(), out);  // :59
  if (rc != CURDLE_OK) {
    char buf[256];
    curdle_last_error(buf, sizeof(buf));
    return Status::Error(std::string("computing msm: ") + buf);  // :60
  }
  G1Jac j;
  memcpy(&j, out, sizeof(j));
  G1XYZZ res;
  g1_from_jac(res, j);
  *ok = curdle_host_equal(&res, &A_c) != 0;  // :63
  return Status::OK();
}

}  // namespace msmaccumulator
}  // namespace curdle
