// Merlin / STROBE-128 transcript and the reference's wrapper -- see transcript.h.
#include "transcript.h"

#include <string.h>

namespace curdle {
namespace transcript {

// Keccak-f[1600] on a byte state (little-endian lanes).
static void keccak_f1600(uint8_t st8[200]) {
  static const uint64_t RC[24] = {
      0x0000000000000001ull, 0x0000000000008082ull, 0x800000000000808aull, 0x8000000080008000ull,
      0x000000000000808bull, 0x0000000080000001ull, 0x8000000080008081ull, 0x8000000000008009ull,
      0x000000000000008aull, 0x0000000000000088ull, 0x0000000080008009ull, 0x000000008000000aull,
      0x000000008000808bull, 0x800000000000008bull, 0x8000000000008089ull, 0x8000000000008003ull,
      0x8000000000008002ull, 0x8000000000000080ull, 0x000000000000800aull, 0x800000008000000aull,
      0x8000000080008081ull, 0x8000000000008080ull, 0x0000000080000001ull, 0x8000000080008008ull};
  static const int ROT[24] = {1, 3, 6, 10, 15, 21, 28, 36, 45, 55, 2, 14, 27, 41, 56, 8, 25, 43, 62, 18, 39, 61, 20, 44};
  static const int PI[24] = {10, 7, 11, 17, 18, 3, 5, 16, 8, 21, 24, 4, 15, 23, 19, 13, 12, 2, 20, 14, 22, 9, 6, 1};
  uint64_t st[25];
  memcpy(st, st8, 200);
  for (int round = 0; round < 24; round++) {
    uint64_t bc[5];
    for (int i = 0; i < 5; i++) bc[i] = st[i] ^ st[i + 5] ^ st[i + 10] ^ st[i + 15] ^ st[i + 20];
    for (int i = 0; i < 5; i++) {
      uint64_t t = bc[(i + 4) % 5] ^ ((bc[(i + 1) % 5] << 1) | (bc[(i + 1) % 5] >> 63));
      for (int j = 0; j < 25; j += 5) st[j + i] ^= t;
    }
    uint64_t t = st[1];
    for (int i = 0; i < 24; i++) {
      int j = PI[i];
      uint64_t b = st[j];
      st[j] = (t << ROT[i]) | (t >> (64 - ROT[i]));
      t = b;
    }
    for (int j = 0; j < 25; j += 5) {
      for (int i = 0; i < 5; i++) bc[i] = st[j + i];
      for (int i = 0; i < 5; i++) st[j + i] ^= (~bc[(i + 1) % 5]) & bc[(i + 2) % 5];
    }
    st[0] ^= RC[round];
  }
  memcpy(st8, st, 200);
}

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

void Strobe128::Absorb(const uint8_t* data, size_t len) {
  for (size_t i = 0; i < len; i++) {
    st_[pos_] ^= data[i];
    pos_++;
    if (pos_ == kStrobeR) RunF();
  }
}

void Strobe128::Squeeze(uint8_t* out, size_t len) {
  for (size_t i = 0; i < len; i++) {
    out[i] = st_[pos_];
    st_[pos_] = 0;
    pos_++;
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
  for (const auto& a : points) AppendPoint(label, alg::Point::FromAffine(a));
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
