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

static inline uint64_t KeyHash(const G1Affine& b) {
  // the limbs of a point are uniformly distributed field elements already: mix two of them
  uint64_t x0, y0;
  memcpy(&x0, &b.x, 8);
  memcpy(&y0, &b.y, 8);
  uint64_t h = x0 ^ (y0 * 0x9e3779b97f4a7c15ull);
  return h ^ (h >> 29);
}

void MsmAccumulator::Grow() {
  size_t cap = table_.empty() ? 1024 : table_.size() * 2;
  table_.assign(cap, 0);
  for (size_t i = 0; i < bases_.size(); i++) {
    size_t s = KeyHash(bases_[i]) & (cap - 1);
    while (table_[s]) s = (s + 1) & (cap - 1);
    table_[s] = (uint32_t)i + 1;
  }
}

// slot of `base` in table_: *found tells whether it holds the base already
size_t MsmAccumulator::Find(const G1Affine& base, bool* found) {
  if (table_.empty() || (bases_.size() + 1) * 2 > table_.size()) Grow();
  const size_t mask = table_.size() - 1;
  size_t s = KeyHash(base) & mask;
  while (table_[s]) {
    if (memcmp(&bases_[table_[s] - 1], &base, sizeof(G1Affine)) == 0) {
      *found = true;
      return s;
    }
    s = (s + 1) & mask;
  }
  *found = false;
  return s;
}

void MsmAccumulator::AddTerm(const G1Affine& base, const Fr& scalar) {
  bool found;
  const size_t s = Find(base, &found);
  if (found) {
    Fr& acc = scalars_[table_[s] - 1];
    fr_add(acc, acc, scalar);
    return;
  }
  bases_.push_back(base);
  scalars_.push_back(scalar);
  table_[s] = (uint32_t)bases_.size();
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
    return Status::Error(std::string("computing msm: ") + buf, rc);  // :60
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
