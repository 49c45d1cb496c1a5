// Fiat-Shamir transcript of the reference (/root/reference/transcript/transcript.go):
// a Merlin transcript (STROBE-128 over Keccak-f[1600]; the reference uses
// github.com/jsign/merlin, go.mod:7, un-vendored) plus the reference's wrapper that
// appends points as 48-byte compressed G1 and scalars as 32-byte big-endian, and
// squeezes Fr challenges with rejection and re-append (transcript.go:48-58).
//
// Host code; part of the restatement of curdleproof.Verify / Prove.
// UNVERIFIED against the Go fork: anchored on the published Merlin test vector
// (tests/test_protocol_host.py).
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "algebra.h"

namespace curdle {
namespace transcript {

// STROBE-128, the subset Merlin uses (meta-AD, AD, PRF).
class Strobe128 {
 public:
  explicit Strobe128(const std::string& protocol_label);
  void MetaAd(const uint8_t* data, size_t len, bool more);
  void Ad(const uint8_t* data, size_t len, bool more);
  void Prf(uint8_t* out, size_t len, bool more);

 private:
  void RunF();
  void Absorb(const uint8_t* data, size_t len);
  void Squeeze(uint8_t* out, size_t len);
  void BeginOp(uint8_t flags, bool more);
  alignas(8) uint8_t st_[200];
  uint8_t pos_, pos_begin_, cur_flags_;
};

// merlin.Transcript
class Merlin {
 public:
  explicit Merlin(const std::string& label);
  void AppendMessage(const std::string& label, const uint8_t* msg, size_t len);
  void ChallengeBytes(const std::string& label, uint8_t* out, size_t len);

 private:
  Strobe128 strobe_;
};

// transcript.Transcript (transcript.go:11-66)
class Transcript {
 public:
  explicit Transcript(const std::string& label) : inner_(label) {}               // New, :15
  void AppendPoints(const std::string& label, const std::vector<alg::Point>& points);  // :25
  void AppendPoint(const std::string& label, const alg::Point& p);
  void AppendPointsAffine(const std::string& label, const std::vector<G1Affine>& points);  // :32
  // the same from the points' 48-byte compressed encodings (count records, back to back): what
  // AppendPointsAffine would produce for the decoded points, without compressing them again
  void AppendCompressed(const std::string& label, const uint8_t* records, size_t count);
  void AppendScalars(const std::string& label, const std::vector<alg::Scalar>& scalars);   // :41
  void AppendScalar(const std::string& label, const alg::Scalar& s);
  alg::Scalar GetAndAppendChallenge(const std::string& label);                              // :48
  std::vector<alg::Scalar> GetAndAppendChallenges(const std::string& label, size_t count);   // :60
  Merlin& inner() { return inner_; }

 private:
  Merlin inner_;
};

}  // namespace transcript
}  // namespace curdle
