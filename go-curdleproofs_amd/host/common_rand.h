// Host mirror of the reference's `common.Rand` (/root/reference/common/rand.go):
// the deterministic SHAKE256 DRBG every reference test draws its inputs and the
// verifier's alpha randomisers from.  Same method names, argument meaning and
// error behaviour; Go (value, error) pairs become bool returns + out-params.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <vector>

#include "../csrc/bls12_381.h"

namespace curdle {
namespace common {

// golang.org/x/crypto/sha3 NewShake256 (rand.go:23): Keccak-f[1600], rate 136,
// domain-separation suffix 0x1f, arbitrary-length squeeze.
class Shake256 {
 public:
  Shake256();
  void Write(const uint8_t* data, size_t len);  // absorb (only before the first Read)
  void Read(uint8_t* out, size_t len);          // squeeze
 private:
  void Permute();
  uint64_t st_[25];
  size_t pos_;
  bool squeezing_;
};

class Rand {
 public:
  explicit Rand(uint64_t seed);                       // NewRand, rand.go:19-33
  void GetFr(Fr& out);                                // rand.go:35-47 (Montgomery form, as fr.Element)
  void GetFrs(size_t n, std::vector<Fr>& out);        // rand.go:49-59
  void GetG1Affine(G1Affine& out);                    // rand.go:72-83
  void GetG1Affines(size_t n, std::vector<G1Affine>& out);  // rand.go:85-95
  void GeneratePermutation(size_t n, std::vector<uint32_t>& out);  // rand.go:97-113
 private:
  void GetFrCanonical(Fr& out);
  Shake256 shake_;
};

}  // namespace common
}  // namespace curdle
