// common.Rand mirror -- see common_rand.h.  Reference: /root/reference/common/rand.go.
#include "common_rand.h"

#include "keccak.h"

#include <string.h>

#include "../csrc/host_math.h"
#include "host_ops.h"

namespace curdle {
namespace common {

// ---------------------------------------------------------------------------
// SHAKE256 (FIPS 202)
// ---------------------------------------------------------------------------

static constexpr size_t kRate = 136;  // SHAKE256: 1600 - 2*256 bits

Shake256::Shake256() : pos_(0), squeezing_(false) { memset(st_, 0, sizeof(st_)); }

void Shake256::Permute() { keccak_f1600(st_); }

void Shake256::Write(const uint8_t* data, size_t len) {
  uint8_t* sb = reinterpret_cast<uint8_t*>(st_);  // little-endian host
  for (size_t i = 0; i < len; i++) {
    sb[pos_++] ^= data[i];
    if (pos_ == kRate) {
      Permute();
      pos_ = 0;
    }
  }
}

void Shake256::Read(uint8_t* out, size_t len) {
  uint8_t* sb = reinterpret_cast<uint8_t*>(st_);
  if (!squeezing_) {
    sb[pos_] ^= 0x1f;
    sb[kRate - 1] ^= 0x80;
    Permute();
    pos_ = 0;
    squeezing_ = true;
  }
  for (size_t i = 0; i < len; i++) {
    if (pos_ == kRate) {
      Permute();
      pos_ = 0;
    }
    out[i] = sb[pos_++];
  }
}

// ---------------------------------------------------------------------------
// Rand
// ---------------------------------------------------------------------------
Rand::Rand(uint64_t seed) {
  uint8_t b[8];
  for (int i = 0; i < 8; i++) b[i] = (uint8_t)(seed >> (56 - 8 * i));  // binary.BigEndian.PutUint64, rand.go:20-21
  shake_.Write(b, 8);
}

// 32 bytes big-endian; retried while >= r (fr.SetBytesCanonical fails), rand.go:35-47.
void Rand::GetFrCanonical(Fr& out) {
  for (;;) {
    uint8_t b[32];
    shake_.Read(b, 32);
    for (int i = 0; i < 8; i++) {
      const uint8_t* q = b + 28 - 4 * i;
      out.l[i] = ((u32)q[0] << 24) | ((u32)q[1] << 16) | ((u32)q[2] << 8) | (u32)q[3];
    }
    bool lt = false;
    for (int i = 7; i >= 0; i--) {
      u32 m = FrParams::mod(i);
      if (out.l[i] != m) {
        lt = out.l[i] < m;
        break;
      }
    }
    if (lt) return;
  }
}

void Rand::GetFr(Fr& out) {
  Fr c;
  GetFrCanonical(c);
  fr_to_mont(out, c);
}

void Rand::GetFrs(size_t n, std::vector<Fr>& out) {
  out.resize(n);
  for (size_t i = 0; i < n; i++) GetFr(out[i]);
}

// scalar <- GetFr; res = scalar * generator (rand.go:72-83).
void Rand::GetG1Affine(G1Affine& out) {
  Fr c;
  GetFrCanonical(c);
  G1Affine g;
  g1_generator(g);
  G1XYZZ gp, r;
  g1_from_affine(gp, g);
  curdle_host_scalar_mul(&r, &gp, c.l);  // ISA-dispatched build, host_ops.cpp
  curdle_host_to_affine(&out, &r);
}

void Rand::GetG1Affines(size_t n, std::vector<G1Affine>& out) {
  out.resize(n);
  for (size_t i = 0; i < n; i++) GetG1Affine(out[i]);
}

// rand.go:97-113: 16 bytes are read per step and only the first two are used.
void Rand::GeneratePermutation(size_t n, std::vector<uint32_t>& out) {
  out.resize(n);
  for (size_t i = 0; i < n; i++) out[i] = (uint32_t)i;
  for (size_t i = 0; i < n; i++) {
    uint8_t tmp[16];
    shake_.Read(tmp, 16);
    uint32_t v = ((uint32_t)tmp[0] << 8) | tmp[1];
    size_t j = v % (i + 1);
    uint32_t t = out[i];
    out[i] = out[j];
    out[j] = t;
  }
}

}  // namespace common
}  // namespace curdle
