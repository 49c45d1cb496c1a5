// whisk package restatement -- see whisk.h.  Reference: /root/reference/whisk/whisk.go, types.go.
#include "knobs.h"
#include "whisk.h"

#include <chrono>
#include <memory>

#include "verify_batch_impl.h"

#include <string.h>

#include <stdexcept>
#include <string>

namespace curdle {
namespace whisk {

using alg::Point;
using alg::Scalar;

namespace {
const char* kWhiskOpeningProof = "whisk_opening_proof";                          // whisk.go:15
const char* kTrackerOpeningProof = "tracker_opening_proof";                      // :16
const char* kTrackerOpeningProofChallenge = "tracker_opening_proof_challenge";   // :17

std::runtime_error err(const std::string& m) { return std::runtime_error(m); }

// G1Affine.SetBytes on a 48-byte compressed point: curve + subgroup checked (gnark default)
Point SetBytes(const uint8_t in[G1POINT_SIZE], const char* what) {
  Point p;
  if (!Point::FromCompressed(in, &p, /*subgroup_check=*/true)) throw err(std::string("failed to set ") + what);
  return p;
}

// WhiskTracker.getPoints, types.go:85-95
void GetPoints(const WhiskTracker& wt, G1Affine* rG, G1Affine* krG) {
  *rG = SetBytes(wt.rG, "rG").Affine();
  *krG = SetBytes(wt.krG, "krG").Affine();
}
}  // namespace

WhiskTracker NewWhiskTracker(const G1Affine& rG, const G1Affine& krG) {
  WhiskTracker t;
  Point::FromAffine(rG).Compressed(t.rG);
  Point::FromAffine(krG).Compressed(t.krG);
  return t;
}

TrackerProof TrackerProof::FromBytes(const uint8_t buf[TRACKER_PROOF_SIZE]) {
  TrackerProof tp;
  Point a, b;
  if (!Point::FromCompressed(buf, &a, true)) throw err("failed to decode A");
  if (!Point::FromCompressed(buf + 48, &b, true)) throw err("failed to decode B");
  if (!Scalar::SetBytesCanonical(buf + 96, &tp.S)) throw err("failed to decode s");
  tp.A = a.Affine();
  tp.B = b.Affine();
  return tp;
}

void TrackerProof::Serialize(uint8_t out[TRACKER_PROOF_SIZE]) const {
  Point::FromAffine(A).Compressed(out);
  Point::FromAffine(B).Compressed(out + 48);
  S.Bytes(out + 96);
}

bool IsValidWhiskShuffleProof(const proto::CRS& crs, const std::vector<WhiskTracker>& preST,
                              const std::vector<WhiskTracker>& postST, const uint8_t proof[WHISK_SHUFFLE_PROOF_SIZE],
                              common::Rand& rand) {
  if (preST.size() != postST.size()) throw err("pre and post shuffle trackers must be the same length");  // :21-23

  // Every point of the call -- M, the proof's ~90 points and the 4 n tracker points -- is
  // decoded (square root, curve and subgroup tests) by ONE batched kernel on the GPU
  // (proto::PointDecoder); the reference decodes them one by one (types.go:39-51, :85-95).
  // Pass 1 registers the records, pass 2 reads the results in the reference's order, so
  // the first error reported is the one the reference would report.
  const size_t n = preST.size();
  proto::PointDecoder dec(/*subgroup_check=*/true);
  const bool lazy = proto::CanVerifyWhileDecoding();
  Point M;
  proto::Proof p;
  try {
    // WhiskShuffleProof.FromReader: M, then the curdleproof; the fixed-size array's zero
    // padding after the proof is never read.  In the lazy form this one pass also yields the
    // proof, its points pending.
    proto::Reader scan(proof, WHISK_SHUFFLE_PROOF_SIZE, true);
    scan.collect = &dec;
    scan.lazy = lazy;
    M = scan.GetPoint("M");
    p = proto::Proof::FromReader(scan);
  } catch (const std::runtime_error& e) {
    throw err(std::string("decoding proof: ") + e.what());
  }
  const size_t first_tracker = dec.size();
  for (size_t i = 0; i < n; i++) {
    dec.Add(preST[i].rG);
    dec.Add(preST[i].krG);
    dec.Add(postST[i].rG);
    dec.Add(postST[i].krG);
  }
  dec.Start();  // the GPU takes the 4 n + ~100 square roots ...
  // ... while the host runs Verify from the RAW encodings: the transcript absorbs a point as
  // its 48-byte compressed form, which for a valid record is the record itself (an invalid one
  // fails the call below whatever was hashed).  curdleproof.go:217-224.
  proto::VerifyPrelude pre;
  {
    std::vector<uint8_t> b(4 * n * G1POINT_SIZE);
    for (size_t i = 0; i < n; i++) {
      memcpy(&b[48 * i], preST[i].rG, 48);
      memcpy(&b[48 * (n + i)], preST[i].krG, 48);
      memcpy(&b[48 * (2 * n + i)], postST[i].rG, 48);
      memcpy(&b[48 * (3 * n + i)], postST[i].krG, 48);
    }
    proto::StartVerify(pre, n, &b[0], &b[48 * n], &b[96 * n], &b[144 * n], proof);  // M is the proof's first record
  }
  // the trackers' points out of the decoder, in the reference's order and with its errors (:35-44)
  auto instance = [&](std::vector<G1Affine>& Rs, std::vector<G1Affine>& Ss, std::vector<G1Affine>& Ts,
                      std::vector<G1Affine>& Us) {
    Rs.resize(n);
    Ss.resize(n);
    Ts.resize(n);
    Us.resize(n);
    for (size_t i = 0; i < n; i++) {
      const size_t at = first_tracker + 4 * i;
      if (!dec.GetAffine(at, &Rs[i])) throw err("getting pre shuffle points: failed to set rG");
      if (!dec.GetAffine(at + 1, &Ss[i])) throw err("getting pre shuffle points: failed to set krG");
      if (!dec.GetAffine(at + 2, &Ts[i])) throw err("getting post shuffle points: failed to set rG");
      if (!dec.GetAffine(at + 3, &Us[i])) throw err("getting post shuffle points: failed to set krG");
    }
  };
  bool accept = false;
  std::string verify_error;
  if (lazy) {
    // the whole of Verify's host part overlaps the decoding; only the accumulator's MSM waits for it
    bool decoded = false;
    try {
      accept = proto::VerifyWhileDecoding(
          pre, p, crs, M, dec,
          [&](proto::DecodedInstance& inst) {
            G1Affine a;
            for (size_t i = 0; i < first_tracker; i++)
              if (!dec.GetAffine(i, &a)) throw err("decoding proof: invalid point");
            instance(inst.Rs, inst.Ss, inst.Ts, inst.Us);
            decoded = true;
          },
          rand);  // :46-58
    } catch (const alg::MsmError&) {
      throw;
    } catch (const std::runtime_error& e) {
      if (!decoded) throw;  // a decoding error: reported as it is
      verify_error = e.what();
    }
  } else {
    dec.Run(/*defer_subgroup=*/true);  // the points; the subgroup test keeps running on the GPU, collected below
    try {
      proto::Reader r(proof, WHISK_SHUFFLE_PROOF_SIZE, true);
      r.decoded = &dec;
      M = r.GetPoint("M");
      p = proto::Proof::FromReader(r);
    } catch (const std::runtime_error& e) {
      throw err(std::string("decoding proof: ") + e.what());
    }
    std::vector<G1Affine> Rs, Ss, Ts, Us;
    instance(Rs, Ss, Ts, Us);
    try {
      accept = proto::VerifyStarted(pre, p, crs, Rs, Ss, Ts, Us, M, rand);  // :46-58
    } catch (const alg::MsmError&) {
      throw;
    } catch (const std::runtime_error& e) {
      verify_error = e.what();
    }
  }
  // the decoding verdict comes first, as in the reference (SetBytes / Decode fail before Verify runs)
  if (!dec.Finish()) {
    Point pt;
    for (size_t i = 0; i < first_tracker; i++)
      if (!dec.Get(i, &pt)) throw err("decoding proof: invalid point (not in the prime-order subgroup)");
    for (size_t i = 0; i < n; i++) {
      const size_t at = first_tracker + 4 * i;
      if (!dec.Get(at, &pt)) throw err("getting pre shuffle points: failed to set rG");
      if (!dec.Get(at + 1, &pt)) throw err("getting pre shuffle points: failed to set krG");
      if (!dec.Get(at + 2, &pt)) throw err("getting post shuffle points: failed to set rG");
      if (!dec.Get(at + 3, &pt)) throw err("getting post shuffle points: failed to set krG");
    }
  }
  if (!verify_error.empty()) throw err("verifying proof: " + verify_error);
  return accept;
}

std::vector<int> IsValidWhiskShuffleProofBatch(const proto::CRS& crs, const std::vector<ShuffleBatchItem>& items,
                                               common::Rand& rand, int nthreads) {
  const size_t k = items.size();
  // Pass 1 (producers, chunk by chunk, ahead of the workers): walk each proof's wire format,
  // register its records and the trackers' with the chunk's decoder, one GPU decoding per chunk.
  // Pass 2 (workers): read the decoded points back in the same order.
  struct Source {
    const std::vector<ShuffleBatchItem>& items;
    std::vector<size_t> first_point, first_tracker;
    std::vector<char> parses;
    std::unique_ptr<proto::DecodeAhead> ahead;
    explicit Source(const std::vector<ShuffleBatchItem>& it)
        : items(it), first_point(it.size(), 0), first_tracker(it.size(), 0), parses(it.size(), 0) {}
    void Scan(size_t i, proto::PointDecoder& dec) {
      first_point[i] = dec.size();
      try {
        proto::Reader scan(items[i].proof, WHISK_SHUFFLE_PROOF_SIZE, true);
        scan.collect = &dec;
        scan.GetPoint("M");
        proto::Proof::FromReader(scan);
        parses[i] = 1;
      } catch (const std::runtime_error&) {
      }
      first_tracker[i] = dec.size();
      for (size_t t = 0; t < items[i].n; t++) {
        dec.Add(items[i].preST[t].rG);
        dec.Add(items[i].preST[t].krG);
        dec.Add(items[i].postST[t].rG);
        dec.Add(items[i].postST[t].krG);
      }
    }
    bool Ready(size_t i) { return ahead->Ready(i); }
    bool Usable(size_t i) {
      ahead->Wait(i);
      return parses[i] != 0;
    }
    proto::Proof DecodeProof(size_t i) {
      proto::Reader r(items[i].proof, WHISK_SHUFFLE_PROOF_SIZE, true);
      r.decoded = &ahead->Wait(i);
      r.decoded_pos = first_point[i];
      r.keep_wire = true;  // the proof value lives inside this call, like the caller's bytes
      r.GetPoint("M");
      return proto::Proof::FromReader(r);
    }
    // curdleproof.go:217-224 from the trackers' own bytes (a valid record is its point's encoding)
    bool Prelude(size_t i, proto::VerifyPrelude& pre) const {
      const size_t n = items[i].n;
      std::vector<uint8_t> b(4 * n * G1POINT_SIZE);
      for (size_t t = 0; t < n; t++) {
        memcpy(&b[48 * t], items[i].preST[t].rG, 48);
        memcpy(&b[48 * (n + t)], items[i].preST[t].krG, 48);
        memcpy(&b[48 * (2 * n + t)], items[i].postST[t].rG, 48);
        memcpy(&b[48 * (3 * n + t)], items[i].postST[t].krG, 48);
      }
      proto::StartVerify(pre, n, &b[0], &b[48 * n], &b[96 * n], &b[144 * n], items[i].proof);  // M is the proof's first record
      return true;
    }
    void Instance(size_t i, std::vector<G1Affine>& Rs, std::vector<G1Affine>& Ss, std::vector<G1Affine>& Ts,
                  std::vector<G1Affine>& Us, Point& M) {
      const proto::PointDecoder& dec = ahead->Wait(i);
      const size_t n = items[i].n;
      Rs.resize(n);
      Ss.resize(n);
      Ts.resize(n);
      Us.resize(n);
      if (!dec.Get(first_point[i], &M)) throw err("decoding proof: invalid point");
      for (size_t t = 0; t < n; t++) {
        const size_t at = first_tracker[i] + 4 * t;
        std::vector<G1Affine>* dst[4] = {&Rs, &Ss, &Ts, &Us};
        for (int c = 0; c < 4; c++)
          if (!dec.GetAffine(at + c, &(*dst[c])[t])) throw err("getting shuffle points: invalid tracker point");
      }
    }
  } src(items);
  const bool trace = knobs::get(knobs::VERIFY_TRACE) > 0;  // the batch's wall time (stderr)
  const auto t0 = std::chrono::steady_clock::now();
  const size_t points_per_proof = k ? WHISK_SHUFFLE_PROOF_SIZE / 48 + 4 * items[0].n : 0;  // an upper bound
  const size_t chunk = proto::DecodeAheadChunk(k, points_per_proof);
  src.ahead = std::make_unique<proto::DecodeAhead>(k, chunk, proto::DecodeAheadProducers(),
                                                   [&src](size_t i, proto::PointDecoder& dec) { src.Scan(i, dec); });
  std::vector<int> oks;
  try {
    oks = proto::VerifyBatchCore(crs, k, src, rand, proto::BatchWorkers(nthreads));
  } catch (...) {
    src.ahead->Abandon();
    throw;
  }
  if (trace) {
    const auto t1 = std::chrono::steady_clock::now();
    fprintf(stderr, "[whisk batch] k=%zu, chunks of %zu, %d threads: %.2f ms\n", k, chunk, nthreads,
            std::chrono::duration<double, std::milli>(t1 - t0).count());
  }
  return oks;
}

std::vector<WhiskTracker> GenerateWhiskShuffleProof(const proto::CRS& crs, const std::vector<WhiskTracker>& preTrackers,
                                                    common::Rand& rand, uint8_t proof_out[WHISK_SHUFFLE_PROOF_SIZE]) {
  std::vector<uint32_t> permutation;
  rand.GeneratePermutation(ELL, permutation);  // :64
  Scalar k;
  rand.GetFr(k.v);                             // :68
  const size_t n = preTrackers.size();
  if (n != ELL) throw err("shuffling and permuting: the whisk shuffle works on ELL trackers");  // permutation length, util.go:49
  // The 2n tracker points in ONE batched decoding (square roots + subgroup tests on the GPU,
  // 0.6 ms) -- the reference decodes them one by one (types.go:39-51), 49 us each on a host core
  std::vector<G1Affine> Rs(n), Ss(n);
  {
    proto::PointDecoder dec(/*subgroup_check=*/true);
    for (size_t i = 0; i < n; i++) {
      dec.Add(preTrackers[i].rG);
      dec.Add(preTrackers[i].krG);
    }
    dec.Run();
    for (size_t i = 0; i < n; i++) {  // the same errors as WhiskTracker.getPoints, types.go:85-95
      if (!dec.GetAffine(2 * i, &Rs[i])) throw err("getting points: failed to set rG");
      if (!dec.GetAffine(2 * i + 1, &Ss[i])) throw err("getting points: failed to set krG");
    }
  }
  proto::ShuffleCommit sc = proto::ShufflePermuteCommit(crs.Gs, crs.Hs, Rs, Ss, permutation, k, rand);  // :83
  proto::Proof proof = proto::Prove(crs, Rs, Ss, sc.Ts, sc.Us, sc.M, permutation, k, sc.rs_m, rand);      // :88

  // WhiskShuffleProof.Serialize, types.go:53-71: M, the proof, zero padding up to the array size
  proto::Writer w;
  w.PutPoint(sc.M);
  const std::vector<uint8_t> body = proof.Serialize();
  if (w.buf.size() + body.size() > WHISK_SHUFFLE_PROOF_SIZE) throw err("serializing proof: larger than WHISK_SHUFFLE_PROOF_SIZE");
  memset(proof_out, 0, WHISK_SHUFFLE_PROOF_SIZE);
  memcpy(proof_out, w.buf.data(), w.buf.size());
  memcpy(proof_out + w.buf.size(), body.data(), body.size());

  std::vector<WhiskTracker> post(n);
  for (size_t i = 0; i < n; i++) post[i] = NewWhiskTracker(sc.Ts[i], sc.Us[i]);  // :109-112
  return post;
}

bool IsValidWhiskTrackerProof(const WhiskTracker& tracker, const uint8_t kComm[G1POINT_SIZE],
                              const uint8_t trackerProofBytes[TRACKER_PROOF_SIZE]) {
  TrackerProof tp;
  try {
    tp = TrackerProof::FromBytes(trackerProofBytes);
  } catch (const std::runtime_error& e) {
    throw err(std::string("decoding proof: ") + e.what());
  }
  G1Affine rG, krG;
  try {
    GetPoints(tracker, &rG, &krG);
  } catch (const std::runtime_error& e) {
    throw err(std::string("deserializing rG and krG: ") + e.what());
  }
  Point kG;
  try {
    kG = SetBytes(kComm, "kG");
  } catch (const std::runtime_error&) {
    throw err("deserializing kG: invalid point");
  }
  const Point g = Point::Generator();
  transcript::Transcript tr(kWhiskOpeningProof);  // :131-134
  tr.AppendPointsAffine(kTrackerOpeningProof, {kG.Affine(), g.Affine(), krG, rG, tp.A, tp.B});
  const Scalar challenge = tr.GetAndAppendChallenge(kTrackerOpeningProofChallenge);

  const Point A_prime = g.Mul(tp.S) + kG.Mul(challenge);                                        // :136-139
  const Point B_prime = Point::FromAffine(rG).Mul(tp.S) + Point::FromAffine(krG).Mul(challenge); // :141-144
  return A_prime == Point::FromAffine(tp.A) && B_prime == Point::FromAffine(tp.B);
}

void GenerateWhiskTrackerProof(const WhiskTracker& tracker, const Scalar& k, common::Rand& rand,
                               uint8_t out[TRACKER_PROOF_SIZE]) {
  G1Affine rG, krG;
  try {
    GetPoints(tracker, &rG, &krG);
  } catch (const std::runtime_error& e) {
    throw err(std::string("deserializing rG and krG: ") + e.what());
  }
  const Point g = Point::Generator();
  const Point kG = g.Mul(k);  // :156
  Scalar blinder;
  rand.GetFr(blinder.v);      // :157
  const Point A = g.Mul(blinder);
  const Point B = Point::FromAffine(rG).Mul(blinder);

  transcript::Transcript tr(kWhiskOpeningProof);
  tr.AppendPointsAffine(kTrackerOpeningProof, {kG.Affine(), g.Affine(), krG, rG, A.Affine(), B.Affine()});
  const Scalar challenge = tr.GetAndAppendChallenge(kTrackerOpeningProofChallenge);

  TrackerProof tp;
  tp.A = A.Affine();
  tp.B = B.Affine();
  tp.S = blinder - challenge * k;  // :171-172
  tp.Serialize(out);
}

}  // namespace whisk
}  // namespace curdle
