// Host-side restatement of the Curdleproofs protocol layers that sit around the MSM
// hot path: the five arguments, the top-level shuffle proof,
// the CRS, ShufflePermuteCommit and the proof wire format.  Same package / function
// names, argument meaning and accept / error behaviour as the reference
// (/root/reference, cited per function in curdleproofs.cpp); every MultiExp goes to
// the GPU through alg::MultiExp, every deferred check through the msmaccumulator
// mirror.
//
// This exists so configs 1 / 3 / 5 can be run end to end without a Go toolchain.  It
// is NOT the product's hot path (that is the MSM); it is the caller either side of it.
// UNVERIFIED against Go-produced proofs (none exist in this environment): pinned by the
// Merlin test vector, the common.Rand known answers and the reference's own
// completeness / soundness / serialisation tests, restated in tests/.
//
// Go's (value, error) becomes: bool for the accept bit, std::runtime_error for the
// structural errors the reference returns as a non-nil error.
#pragma once
#include <stdint.h>

#include <string>
#include <vector>

#include "algebra.h"
#include "common_rand.h"
#include "msmaccumulator.h"
#include "transcript.h"

namespace curdle {
namespace proto {

using alg::Point;
using alg::Scalar;

static constexpr int N_BLINDERS = 4;  // common/constants.go:3

// ---- wire format helpers (gnark Encoder / Decoder as the reference uses them) ----
struct Writer {
  std::vector<uint8_t> buf;
  void PutPoint(const Point& p);
  void PutScalar(const Scalar& s);
  void PutPoints(const std::vector<Point>& v);  // uint32 big-endian length, then the points
};
// Batched decoding of compressed points on the GPU (curdle_g1_decompress_batch): callers
// register the 48-byte records they will need, Run() decodes them all in one kernel, Get()
// hands them out.  Fewer than kMinDeviceBatch records are decoded on the host, point by
// point (a kernel launch would cost more); larger batches need the GPU -- there is no
// silent host fallback, a missing device is an alg::MsmError.  CURDLE_HOST_DECODE=1 forces
// the host decoder for A/B measurements.
class PointDecoder {
 public:
  explicit PointDecoder(bool subgroup_check) : subgroup_(subgroup_check) {}
  ~PointDecoder();
  PointDecoder(const PointDecoder&) = delete;
  PointDecoder& operator=(const PointDecoder&) = delete;
  size_t Add(const uint8_t rec[48]);          // -> index for Get
  // Decodes every record.  With defer_subgroup the GPU's subgroup test keeps running after
  // Run() returns (Get() then only reflects encoding / curve errors) and Finish() collects
  // its verdict: the caller overlaps its own work with it and must not publish a result
  // before Finish() has returned true.
  void Run(bool defer_subgroup = false);
  bool Finish();                              // true: no record failed the (deferred) subgroup test
  bool Get(size_t index, Point* out) const;   // false: not a valid encoding / not on the curve / not in G1
  size_t size() const { return n_; }
  static bool OnDevice();                     // false only under CURDLE_HOST_DECODE=1
  static constexpr size_t kMinDeviceBatch = 48;
 private:
  bool subgroup_;
  size_t n_ = 0;
  std::vector<uint8_t> blob_;
  std::vector<G1Affine> pts_;
  std::vector<uint8_t> status_;
  int ticket_ = -1;                           // >= 0: a deferred subgroup test is in flight
};

struct Reader {
  const uint8_t* p;
  size_t left;
  bool subgroup_check;
  // two-pass decoding: with `collect` set GetPoint only registers the record and returns
  // infinity; with `decoded` set it hands out the records in the same order
  PointDecoder* collect = nullptr;
  const PointDecoder* decoded = nullptr;
  size_t decoded_pos = 0;
  Reader(const uint8_t* data, size_t len, bool subgroup = false) : p(data), left(len), subgroup_check(subgroup) {}
  Point GetPoint(const char* what);
  Scalar GetScalar(const char* what);
  std::vector<Point> GetPoints(const char* what);
};

// ---- groupcommitment (groupcommitment/groupcommitment.go) ----
struct GroupCommitment {
  Point T_1, T_2;
  static GroupCommitment New(const Point& crsG, const Point& crsH, const Point& T, const Scalar& r);  // :17
  GroupCommitment Add(const GroupCommitment& cm) const;   // :33
  GroupCommitment Mul(const Scalar& s) const;             // :41
  bool Eq(const GroupCommitment& cm) const;               // :50
  void Serialize(Writer& w) const;
  void FromReader(Reader& r);
};

// ---- crs.go ----
struct CRS {
  std::vector<G1Affine> Gs, Hs;
  Point H, Gt, Gu;
  G1Affine Gsum, Hsum;
};
CRS GenerateCRS(size_t size, common::Rand& rand);  // crs.go:20

// common.ShufflePermuteCommit (common/util.go:45)
struct ShuffleCommit {
  std::vector<G1Affine> Ts, Us;
  Point M;
  std::vector<Scalar> rs_m;
};
ShuffleCommit ShufflePermuteCommit(const std::vector<G1Affine>& crsGs, const std::vector<G1Affine>& crsHs,
                                   const std::vector<G1Affine>& Rs, const std::vector<G1Affine>& Ss,
                                   const std::vector<uint32_t>& perm, const Scalar& k, common::Rand& rand);

// ---- samescalarargument ----
namespace samescalar {
struct Proof {
  GroupCommitment A, B;
  Scalar Z_k, Z_t, Z_u;
  void Serialize(Writer& w) const;
  void FromReader(Reader& r);
};
Proof Prove(const Point& Gt, const Point& Gu, const Point& H, const Point& R, const Point& S,
            const GroupCommitment& T, const GroupCommitment& U, const Scalar& k, const Scalar& r_t, const Scalar& r_u,
            transcript::Transcript& tr, common::Rand& rand);
// With an accumulator (and not in eager mode) the two commitment equations join the batched
// check instead of being evaluated on the spot; without one this is the reference's Verify.
bool Verify(const Proof& proof, const Point& Gt, const Point& Gu, const Point& H, const Point& R, const Point& S,
            const GroupCommitment& T, const GroupCommitment& U, transcript::Transcript& tr,
            msmaccumulator::MsmAccumulator* acc = nullptr, common::Rand* rand = nullptr);
}  // namespace samescalar

// ---- innerproductargument ----
namespace ipa {
struct Proof {
  Point B_c, B_d;
  std::vector<Point> L_Cs, R_Cs, L_Ds, R_Ds;
  Scalar c0, d0;
  void Serialize(Writer& w) const;
  void FromReader(Reader& r);
};
// Gs / Gs_prime / cs / ds are taken by value: the prover folds them in place.
Proof Prove(std::vector<G1Affine> Gs, std::vector<G1Affine> Gs_prime, const Point& H, const Point& C, const Point& D,
            const Scalar& z, std::vector<Scalar> cs, std::vector<Scalar> ds, transcript::Transcript& tr,
            common::Rand& rand);
bool Verify(const Proof& proof, const std::vector<G1Affine>& Gs, const Point& H, const Point& C, const Point& D,
            const Scalar& z, const std::vector<Scalar>& us, transcript::Transcript& tr,
            msmaccumulator::MsmAccumulator& acc, common::Rand& rand);
}  // namespace ipa

// ---- grandproductargument ----
namespace gprod {
struct Proof {
  Point C;
  Scalar Rp;
  ipa::Proof IPAProof;
  void Serialize(Writer& w) const;
  void FromReader(Reader& r);
};
Proof Prove(const std::vector<G1Affine>& Gs, const std::vector<G1Affine>& Hs, const Point& H, const Point& B,
            const Scalar& result, const std::vector<Scalar>& bs, const std::vector<Scalar>& r_bs,
            transcript::Transcript& tr, common::Rand& rand);
bool Verify(const Proof& proof, const std::vector<G1Affine>& Gs, const std::vector<G1Affine>& Hs, const Point& H,
            const G1Affine& Gsum, const G1Affine& Hsum, const Point& B, const Scalar& result, int numBlinders,
            transcript::Transcript& tr, msmaccumulator::MsmAccumulator& acc, common::Rand& rand);
}  // namespace gprod

// ---- samepermutationargument ----
namespace sameperm {
struct Proof {
  Point B;
  gprod::Proof gpaProof;
  void Serialize(Writer& w) const;
  void FromReader(Reader& r);
};
Proof Prove(const std::vector<G1Affine>& Gs, const std::vector<G1Affine>& Hs, const Point& H, const Point& A,
            const Point& M, const std::vector<Scalar>& as, const std::vector<uint32_t>& permutation,
            const std::vector<Scalar>& rs_a, const std::vector<Scalar>& rs_m, transcript::Transcript& tr,
            common::Rand& rand);
bool Verify(const Proof& proof, const std::vector<G1Affine>& Gs, const std::vector<G1Affine>& Hs, const Point& H,
            const G1Affine& Gsum, const G1Affine& Hsum, const Point& A, const Point& M, const std::vector<Scalar>& as,
            int numBlinders, transcript::Transcript& tr, msmaccumulator::MsmAccumulator& acc, common::Rand& rand);
}  // namespace sameperm

// ---- samemultiscalarargument ----
namespace samemsm {
struct Proof {
  Point B_a, B_t, B_u;
  std::vector<Point> L_A, L_T, L_U, R_A, R_T, R_U;
  Scalar x;
  void Serialize(Writer& w) const;
  void FromReader(Reader& r);
};
// G, T, U, x by value: folded in place by the prover.
Proof Prove(std::vector<G1Affine> G, const Point& A, const Point& Z_t, const Point& Z_u, std::vector<G1Affine> T,
            std::vector<G1Affine> U, std::vector<Scalar> x, transcript::Transcript& tr, common::Rand& rand);
bool Verify(const Proof& proof, const std::vector<G1Affine>& G, const Point& A, const Point& Z_t, const Point& Z_u,
            const std::vector<G1Affine>& T, const std::vector<G1Affine>& U, transcript::Transcript& tr,
            msmaccumulator::MsmAccumulator& acc, common::Rand& rand);
}  // namespace samemsm

// ---- curdleproof.go ----
struct Proof {
  Point A;
  GroupCommitment T, U;
  Point R, S;
  sameperm::Proof proofSamePermutation;
  samescalar::Proof proofSameScalar;
  samemsm::Proof proofSameMultiscalar;
  std::vector<uint8_t> Serialize() const;                                    // curdleproof.go:358
  static Proof FromBytes(const uint8_t* data, size_t len, bool subgroup_check = false);  // :320
  static Proof FromReader(Reader& r);  // the same from a stream position (trailing bytes are left unread)
  // Decoding with the GPU's subgroup test left running: the caller must call dec.Finish()
  // and treat `false` as the decoding error it is before using any result derived from
  // the proof.
  static Proof FromBytesDeferred(const uint8_t* data, size_t len, PointDecoder& dec);
};
Proof Prove(const CRS& crs, const std::vector<G1Affine>& Rs, const std::vector<G1Affine>& Ss,
            const std::vector<G1Affine>& Ts, const std::vector<G1Affine>& Us, const Point& M,
            const std::vector<uint32_t>& perm, const Scalar& k, const std::vector<Scalar>& rs_m,
            common::Rand& rand);                                            // :38
bool Verify(const Proof& proof, const CRS& crs, const std::vector<G1Affine>& Rs, const std::vector<G1Affine>& Ss,
            const std::vector<G1Affine>& Ts, const std::vector<G1Affine>& Us, const Point& M,
            common::Rand& rand);                                            // :199

// The body of Verify up to, but not including, the accumulator's final MSM (false: a direct,
// non-accumulated check already failed); lets several proofs share one accumulator.
bool VerifyInto(const Proof& proof, const CRS& crs, const std::vector<G1Affine>& Rs, const std::vector<G1Affine>& Ss,
                const std::vector<G1Affine>& Ts, const std::vector<G1Affine>& Us, const Point& M, common::Rand& rand,
                msmaccumulator::MsmAccumulator& acc);

// Cross-proof batch verification over one CRS: one shared accumulator, one MSM (see the
// definition).  Returns the per-proof accept bits.
struct BatchItem {  // borrowed buffers: ell affine points each, M as 18 Jacobian limbs
  const uint8_t* proof;
  size_t proof_len;
  const G1Affine *Rs, *Ss, *Ts, *Us;
  size_t ell;
  const uint64_t* M;
};
std::vector<int> VerifyBatch(const CRS& crs, const std::vector<BatchItem>& items, common::Rand& rand, int nthreads);

// Deferred (default) or eager evaluation of the verifier's check points; see
// curdle_verify_set_eager in include/curdle_msm.h.  Returns the previous setting.
int SetEagerChecks(int eager);

}  // namespace proto
}  // namespace curdle
