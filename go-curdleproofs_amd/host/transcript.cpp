// Merlin / STROBE-128 transcript and the reference's wrapper -- see transcript.h.
#include "transcript.h"

#include "keccak.h"

#include <string.h>

namespace curdle {
namespace transcript {

// Keccak-f[1600] on STROBE's byte state (little-endian lanes; keccak.h has the round; the
// state is 8-byte aligned in the class).
static void keccak_f1600(uint8_t st8[200]) { curdle::keccak_f1600_dispatch(reinterpret_cast<uint64_t*>(st8)); }

// ---------------------------------------------------------------- STROBE ---
static constexpr uint8_t kStrobeR = 166;
static constexpr uint8_t FLAG_I = 1, FLAG_A = 1 << 1, FLAG_C = 1 << 2, FLAG_M = 1 << 4, FLAG_K = 1 << 5;

Strobe128::Strobe128(const std::string& protocol_label) : pos_(0), pos_begin_(0), cur_flags_(0) {
  memset(st_, 0, sizeof(st_));
  const uint8_t init[6] = {1, (uint8_t)(kStrobeR + 2), 1, 0, 1, 96};
  memcpy(st_, init, 6);
  memcpy(st_ + 6, "STROBEv1.0.2", 12);
  keccak_f1600(st_);
  MetaAd(reinterpret_cast<const uint8_t*>(protocol_label.data()), protocol_label.size(), false);
}

void Strobe128::RunF() {
  st_[pos_] ^= pos_begin_;
  st_[pos_ + 1] ^= 0x04;
  st_[kStrobeR + 1] ^= 0x80;
  keccak_f1600(st_);
  pos_ = 0;
  pos_begin_ = 0;
}

// One verification absorbs ~115 KB in ~1,500 messages: whole runs up to the rate boundary at a
// time, not byte by byte.
void Strobe128::Absorb(const uint8_t* data, size_t len) {
  while (len) {
    size_t run = (size_t)(kStrobeR - pos_);
    if (run > len) run = len;
    for (size_t i = 0; i < run; i++) st_[pos_ + i] ^= data[i];  // vectorised by the compiler
    pos_ = (uint8_t)(pos_ + run);
    data += run;
    len -= run;
    if (pos_ == kStrobeR) RunF();
  }
}

void Strobe128::Squeeze(uint8_t* out, size_t len) {
  while (len) {
    size_t run = (size_t)(kStrobeR - pos_);
    if (run > len) run = len;
    memcpy(out, st_ + pos_, run);
    memset(st_ + pos_, 0, run);
    pos_ = (uint8_t)(pos_ + run);
    out += run;
    len -= run;
    if (pos_ == kStrobeR) RunF();
  }
}

void Strobe128::BeginOp(uint8_t flags, bool more) {
  if (more) return;  // continuation of the current operation (same flags)
  const uint8_t old_begin = pos_begin_;
  pos_begin_ = pos_ + 1;
  cur_flags_ = flags;
  const uint8_t hdr[2] = {old_begin, flags};
  Absorb(hdr, 2);
  const bool force_f = (flags & (FLAG_C | FLAG_K)) != 0;
  if (force_f && pos_ != 0) RunF();
}

void Strobe128::MetaAd(const uint8_t* data, size_t len, bool more) {
  BeginOp(FLAG_M | FLAG_A, more);
  Absorb(data, len);
}
void Strobe128::Ad(const uint8_t* data, size_t len, bool more) {
  BeginOp(FLAG_A, more);
  Absorb(data, len);
}
void Strobe128::Prf(uint8_t* out, size_t len, bool more) {
  BeginOp(FLAG_I | FLAG_A | FLAG_C, more);
  Squeeze(out, len);
}

// ---------------------------------------------------------------- Merlin ---
static void le32(uint8_t out[4], size_t v) {
  out[0] = (uint8_t)v;
  out[1] = (uint8_t)(v >> 8);
  out[2] = (uint8_t)(v >> 16);
  out[3] = (uint8_t)(v >> 24);
}

Merlin::Merlin(const std::string& label) : strobe_("Merlin v1.0") {
  AppendMessage("dom-sep", reinterpret_cast<const uint8_t*>(label.data()), label.size());
}

void Merlin::AppendMessage(const std::string& label, const uint8_t* msg, size_t len) {
  uint8_t n[4];
  le32(n, len);
  strobe_.MetaAd(reinterpret_cast<const uint8_t*>(label.data()), label.size(), false);
  strobe_.MetaAd(n, 4, true);
  strobe_.Ad(msg, len, false);
}

void Merlin::ChallengeBytes(const std::string& label, uint8_t* out, size_t len) {
  uint8_t n[4];
  le32(n, len);
  strobe_.MetaAd(reinterpret_cast<const uint8_t*>(label.data()), label.size(), false);
  strobe_.MetaAd(n, 4, true);
  strobe_.Prf(out, len, false);
}

// ------------------------------------------------------------- Transcript ---
void Transcript::AppendPoint(const std::string& label, const alg::Point& p) {
  uint8_t b[48];
  p.Compressed(b);
  inner_.AppendMessage(label, b, 48);  // transcript.go:34-38
}
void Transcript::AppendPoints(const std::string& label, const std::vector<alg::Point>& points) {
  for (const auto& p : points) AppendPoint(label, p);  // :25-30 (normalise, then one message per point)
}
void Transcript::AppendPointsAffine(const std::string& label, const std::vector<G1Affine>& points) {
  std::vector<uint8_t> b(48 * points.size());
  alg::CompressAffineBatch(points.data(), points.size(), b.data());
  AppendCompressed(label, b.data(), points.size());
}
void Transcript::AppendCompressed(const std::string& label, const uint8_t* records, size_t count) {
  for (size_t i = 0; i < count; i++) inner_.AppendMessage(label, records + 48 * i, 48);
}
void Transcript::AppendScalar(const std::string& label, const alg::Scalar& s) {
  uint8_t b[32];
  s.Bytes(b);
  inner_.AppendMessage(label, b, 32);  // :41-46
}
void Transcript::AppendScalars(const std::string& label, const std::vector<alg::Scalar>& scalars) {
  for (const auto& s : scalars) AppendScalar(label, s);
}
alg::Scalar Transcript::GetAndAppendChallenge(const std::string& label) {
  for (;;) {  // :48-58: 32 bytes, canonical or retry, then re-append under the same label
    uint8_t dest[32];
    inner_.ChallengeBytes(label, dest, 32);
    alg::Scalar c;
    if (alg::Scalar::SetBytesCanonical(dest, &c)) {
      AppendScalar(label, c);
      return c;
    }
  }
}
std::vector<alg::Scalar> Transcript::GetAndAppendChallenges(const std::string& label, size_t count) {
  std::vector<alg::Scalar> out;
  out.reserve(count);
  for (size_t i = 0; i < count; i++) out.push_back(GetAndAppendChallenge(label));
  return out;
}

}  // namespace transcript
}  // namespace curdle
