// Host-side restatement of the reference's `whisk` package: the
// byte-level API the Ethereum Whisk SSLE spec consumes -- shuffle proofs over 124 trackers
// (curdleproof.Prove / Verify behind a fixed-size encoding) and the tracker opening proofs
// (a discrete-log-equality proof).  Same names, argument order and error behaviour as
// /root/reference/whisk/{whisk.go,types.go}; Go's (value, error) becomes a return value plus
// std::runtime_error for the error leg.  Every MSM underneath goes to the GPU.
#pragma once
#include <stdint.h>

#include <vector>

#include "curdleproofs.h"

namespace curdle {
namespace whisk {

static constexpr size_t G1POINT_SIZE = 48;                       // types.go:14
static constexpr size_t N = 128;                                 // :16
static constexpr size_t ELL = N - proto::N_BLINDERS;             // :17
static constexpr size_t TRACKER_PROOF_SIZE = 128;                // :19
static constexpr size_t WHISK_SHUFFLE_PROOF_SIZE = 4576;         // :20

struct WhiskTracker {  // types.go:73-76: the two points in gnark's compressed form
  uint8_t rG[G1POINT_SIZE];
  uint8_t krG[G1POINT_SIZE];
};
WhiskTracker NewWhiskTracker(const G1Affine& rG, const G1Affine& krG);  // :78

struct TrackerProof {  // types.go:99-103
  G1Affine A, B;
  alg::Scalar S;
  static TrackerProof FromBytes(const uint8_t buf[TRACKER_PROOF_SIZE]);  // :105
  void Serialize(uint8_t out[TRACKER_PROOF_SIZE]) const;                 // :119
};

// whisk.go:20 -- (true|false, nil) is the return value, (false, err) throws.
bool IsValidWhiskShuffleProof(const proto::CRS& crs, const std::vector<WhiskTracker>& preST,
                              const std::vector<WhiskTracker>& postST, const uint8_t proof[WHISK_SHUFFLE_PROOF_SIZE],
                              common::Rand& rand);
// Many shuffle proofs over one CRS at once (no reference counterpart; BASELINE config 5 is
// "1024 Whisk proofs"): every point of every proof and tracker set decoded by one GPU
// kernel, the proofs verified by proto::VerifyBatchCore (worker threads, one MSM per group
// of 32 proofs).  Returns the accept bits; a proof or tracker set that does not decode is
// rejected (bit 0) rather than raised.
struct ShuffleBatchItem {
  const WhiskTracker* preST;
  const WhiskTracker* postST;
  size_t n;
  const uint8_t* proof;  // WHISK_SHUFFLE_PROOF_SIZE bytes
};
std::vector<int> IsValidWhiskShuffleProofBatch(const proto::CRS& crs, const std::vector<ShuffleBatchItem>& items,
                                               common::Rand& rand, int nthreads);
// whisk.go:63 -- returns the post-shuffle trackers, writes the 4,576-byte proof.
std::vector<WhiskTracker> GenerateWhiskShuffleProof(const proto::CRS& crs, const std::vector<WhiskTracker>& preTrackers,
                                                    common::Rand& rand, uint8_t proof_out[WHISK_SHUFFLE_PROOF_SIZE]);
// whisk.go:116
bool IsValidWhiskTrackerProof(const WhiskTracker& tracker, const uint8_t kComm[G1POINT_SIZE],
                              const uint8_t trackerProof[TRACKER_PROOF_SIZE]);
// whisk.go:149
void GenerateWhiskTrackerProof(const WhiskTracker& tracker, const alg::Scalar& k, common::Rand& rand,
                               uint8_t out[TRACKER_PROOF_SIZE]);

}  // namespace whisk
}  // namespace curdle
