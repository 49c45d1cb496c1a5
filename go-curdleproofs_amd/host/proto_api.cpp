// C ABI of the host-side protocol restatement (curdleproofs.h): CRS, the shuffle
// instance, curdleproof.Prove / Verify on serialised proofs.  Declared in
// include/curdle_msm.h under "Protocol layers".
#include <string.h>

#include <exception>
#include <new>
#include <stdexcept>
#include <thread>
#include <vector>

#include "../../include/curdle_msm.h"
#include "curdleproofs.h"
#include "device_accumulator.h"
#include "whisk.h"

using namespace curdle;
using alg::Point;
using alg::Scalar;

// defined in csrc/msm_context.hip
extern "C" int curdle_set_last_error(int code, const char* msg);

struct curdle_rand {  // same layout as in csrc/misc_api.hip
  common::Rand r;
  explicit curdle_rand(uint64_t seed) : r(seed) {}
};
struct curdle_crs {
  proto::CRS crs;
};
struct curdle_proof {
  proto::Proof p;
};

static std::vector<G1Affine> Affines(const uint64_t* p, size_t n) {
  std::vector<G1Affine> v(n);
  if (n) memcpy(v.data(), p, n * 96);
  return v;
}

template <class F>
static int Guard(F&& f) {
  try {
    return f();
  } catch (const std::bad_alloc&) {
    return curdle_set_last_error(CURDLE_ENOMEM, "out of memory");
  } catch (const alg::MsmError& e) {
    // a device-side failure (no device, HIP error, out of device memory) carries the code of
    // the entry point that failed: "could not compute", never to be read as "reject"
    return curdle_set_last_error(e.rc == CURDLE_OK || e.rc == CURDLE_EINVAL ? CURDLE_EHIP : e.rc, e.what());
  } catch (const std::exception& e) {
    return curdle_set_last_error(CURDLE_EINVAL, e.what());  // malformed input / structural error of the protocol
  }
}

extern "C" curdle_crs* curdle_crs_generate(size_t ell, curdle_rand* rand) {
  if (!rand) return nullptr;
  try {
    curdle_crs* c = new curdle_crs();
    c->crs = proto::GenerateCRS(ell, rand->r);
    return c;
  } catch (...) {
    return nullptr;
  }
}
extern "C" void curdle_crs_free(curdle_crs* c) { delete c; }
extern "C" size_t curdle_crs_size(const curdle_crs* c) { return c ? c->crs.Gs.size() : 0; }

extern "C" int curdle_shuffle_permute_commit(const curdle_crs* crs, const uint64_t* Rs, const uint64_t* Ss, size_t ell,
                                             const uint32_t* perm, const uint64_t k[4], curdle_rand* rand,
                                             uint64_t* Ts_out, uint64_t* Us_out, uint64_t M_out[18],
                                             uint64_t rs_m_out[16]) {
  if (!crs || !Rs || !Ss || !perm || !k || !rand || !Ts_out || !Us_out || !M_out || !rs_m_out)
    return curdle_set_last_error(CURDLE_EINVAL, "null argument");
  if (ell != crs->crs.Gs.size()) return curdle_set_last_error(CURDLE_EINVAL, "ell does not match the CRS");
  return Guard([&]() {
    std::vector<uint32_t> pv(perm, perm + ell);
    for (uint32_t v : pv)
      if (v >= ell) throw std::runtime_error("permutation entry out of range");
    proto::ShuffleCommit sc = proto::ShufflePermuteCommit(crs->crs.Gs, crs->crs.Hs, Affines(Rs, ell), Affines(Ss, ell),
                                                          pv, Scalar::FromMont(k), rand->r);
    memcpy(Ts_out, sc.Ts.data(), ell * 96);
    memcpy(Us_out, sc.Us.data(), ell * 96);
    sc.M.Jac(M_out);
    for (int i = 0; i < proto::N_BLINDERS; i++) memcpy(rs_m_out + 4 * i, &sc.rs_m[i].v, 32);
    return CURDLE_OK;
  });
}

extern "C" int curdle_prove(const curdle_crs* crs, const uint64_t* Rs, const uint64_t* Ss, const uint64_t* Ts,
                            const uint64_t* Us, size_t ell, const uint64_t M[18], const uint32_t* perm,
                            const uint64_t k[4], const uint64_t rs_m[16], curdle_rand* rand, uint8_t* proof_out,
                            size_t cap, size_t* proof_len) {
  if (!crs || !Rs || !Ss || !Ts || !Us || !M || !perm || !k || !rs_m || !rand || !proof_len)
    return curdle_set_last_error(CURDLE_EINVAL, "null argument");
  if (ell != crs->crs.Gs.size()) return curdle_set_last_error(CURDLE_EINVAL, "ell does not match the CRS");
  return Guard([&]() {
    std::vector<Scalar> rsm(proto::N_BLINDERS);
    for (int i = 0; i < proto::N_BLINDERS; i++) rsm[i] = Scalar::FromMont(rs_m + 4 * i);
    proto::Proof p = proto::Prove(crs->crs, Affines(Rs, ell), Affines(Ss, ell), Affines(Ts, ell), Affines(Us, ell),
                                  Point::FromJac(M), std::vector<uint32_t>(perm, perm + ell), Scalar::FromMont(k), rsm,
                                  rand->r);
    std::vector<uint8_t> bytes = p.Serialize();
    *proof_len = bytes.size();
    if (!proof_out || cap < bytes.size()) return curdle_set_last_error(CURDLE_EINVAL, "proof buffer too small");
    memcpy(proof_out, bytes.data(), bytes.size());
    return CURDLE_OK;
  });
}

extern "C" int curdle_verify(const curdle_crs* crs, const uint8_t* proof, size_t proof_len, const uint64_t* Rs,
                             const uint64_t* Ss, const uint64_t* Ts, const uint64_t* Us, size_t ell,
                             const uint64_t M[18], curdle_rand* rand, int* ok) {
  if (!crs || !proof || !Rs || !Ss || !Ts || !Us || !M || !rand || !ok)
    return curdle_set_last_error(CURDLE_EINVAL, "null argument");
  *ok = 0;
  if (ell != crs->crs.Gs.size()) return curdle_set_last_error(CURDLE_EINVAL, "ell does not match the CRS");
  return Guard([&]() {
    // gnark's Decoder checks curve and subgroup membership of every point (curdleproof.go:322).
    // The subgroup half of that runs on the GPU while the host verifies; its verdict is
    // collected before anything is reported.
    proto::PointDecoder dec(/*subgroup_check=*/true);
    const Point Mp = Point::FromJac(M);
    bool accept;
    if (proto::CanVerifyWhileDecoding()) {
      // one pass over the wire format: the records go to the GPU, the proof comes back with its
      // points pending, and the WHOLE host part (transcript, challenge algebra) runs from the
      // bytes while the decoding kernel takes the square roots
      proto::Reader scan(proof, proof_len, true);
      scan.collect = &dec;
      const proto::Proof p = proto::Proof::ScanLazy(scan);
      dec.Start();
      proto::DecodedInstance given{Affines(Rs, ell), Affines(Ss, ell), Affines(Ts, ell), Affines(Us, ell)};
      proto::VerifyPrelude pre;
      proto::StartVerify(pre, given.Rs, given.Ss, given.Ts, given.Us, Mp);
      accept = proto::VerifyWhileDecoding(
          pre, p, crs->crs, Mp, dec,
          [&](proto::DecodedInstance& inst) {
            G1Affine a;
            for (size_t i = 0; i < dec.size(); i++)
              if (!dec.GetAffine(i, &a)) throw std::runtime_error("decoding proof: invalid point");
            inst = std::move(given);
          },
          rand->r);
    } else {
      proto::Proof::ScanAndStart(proof, proof_len, dec);  // the GPU takes the square roots ...
      const std::vector<G1Affine> R = Affines(Rs, ell), S = Affines(Ss, ell), T = Affines(Ts, ell), U = Affines(Us, ell);
      proto::VerifyPrelude pre;
      proto::StartVerify(pre, R, S, T, U, Mp);            // ... while the host absorbs the instance and draws `as`
      proto::Proof p = proto::Proof::FromStarted(proof, proof_len, dec);
      accept = proto::VerifyStarted(pre, p, crs->crs, R, S, T, U, Mp, rand->r);
    }
    if (!dec.Finish()) throw std::runtime_error("decoding proof: invalid point (not in the prime-order subgroup)");
    *ok = accept ? 1 : 0;
    return CURDLE_OK;
  });
}

extern "C" int curdle_proof_from_bytes(const uint8_t* proof, size_t proof_len, curdle_proof** out) {
  if (!proof || !out) return curdle_set_last_error(CURDLE_EINVAL, "null argument");
  *out = nullptr;
  return Guard([&]() {
    curdle_proof* h = new curdle_proof();
    try {
      h->p = proto::Proof::FromBytes(proof, proof_len, /*subgroup_check=*/true);
    } catch (...) {
      delete h;
      throw;
    }
    *out = h;
    return CURDLE_OK;
  });
}
extern "C" void curdle_proof_free(curdle_proof* p) { delete p; }

extern "C" int curdle_verify_proof(const curdle_crs* crs, const curdle_proof* proof, const uint64_t* Rs, const uint64_t* Ss,
                                   const uint64_t* Ts, const uint64_t* Us, size_t ell, const uint64_t M[18],
                                   curdle_rand* rand, int* ok) {
  if (!crs || !proof || !Rs || !Ss || !Ts || !Us || !M || !rand || !ok)
    return curdle_set_last_error(CURDLE_EINVAL, "null argument");
  *ok = 0;
  if (ell != crs->crs.Gs.size()) return curdle_set_last_error(CURDLE_EINVAL, "ell does not match the CRS");
  return Guard([&]() {
    bool accept = proto::Verify(proof->p, crs->crs, Affines(Rs, ell), Affines(Ss, ell), Affines(Ts, ell), Affines(Us, ell),
                                Point::FromJac(M), rand->r);
    *ok = accept ? 1 : 0;
    return CURDLE_OK;
  });
}

// Where the verifier's accumulator lives: 1 (default) on the device, 0 on the host mirror.
extern "C" int curdle_verify_set_device_acc(int on) { return proto::SetDeviceAccumulator(on); }

// The accumulated (base, scalar) list of one verification, before the final MSM, from the host
// mirror (device = 0) or from the device accumulator (device = 1), and the accept bit: what
// the parity test of the two accumulators compares (tests/test_device_accumulator.py).
extern "C" int curdle_verify_export_accumulator(const curdle_crs* crs, const curdle_proof* proof, const uint64_t* Rs,
                                                const uint64_t* Ss, const uint64_t* Ts, const uint64_t* Us, size_t ell,
                                                const uint64_t M[18], curdle_rand* rand, int device, uint64_t* points,
                                                uint64_t* scalars, size_t cap, size_t* n_out, int* ok) {
  if (!crs || !proof || !Rs || !Ss || !Ts || !Us || !M || !rand || !n_out || !ok)
    return curdle_set_last_error(CURDLE_EINVAL, "null argument");
  if (ell != crs->crs.Gs.size()) return curdle_set_last_error(CURDLE_EINVAL, "ell does not match the CRS");
  return Guard([&]() {
    const std::vector<G1Affine> R = Affines(Rs, ell), S = Affines(Ss, ell), T = Affines(Ts, ell), U = Affines(Us, ell);
    std::vector<G1Affine> bases;
    std::vector<Scalar> sc;
    bool accept = false;
    if (device) {
      proto::DeviceSink sink(crs->crs, R, S, T, U);
      if (proto::VerifyWithSink(proof->p, crs->crs, R, S, T, U, Point::FromJac(M), rand->r, sink))
        accept = sink.VerifyAndExport(&bases, &sc);
    } else {
      msmaccumulator::MsmAccumulator acc;
      if (proto::VerifyInto(proof->p, crs->crs, R, S, T, U, Point::FromJac(M), rand->r, acc)) {
        bases = acc.Bases();
        sc.resize(acc.Scalars().size());
        for (size_t i = 0; i < sc.size(); i++) sc[i].v = acc.Scalars()[i];
        msmaccumulator::Status st = acc.Verify(&accept);
        if (!st.ok) throw alg::MsmError("verifying msm accumulator: " + st.err, st.rc ? st.rc : CURDLE_EHIP);
      }
    }
    *n_out = bases.size();
    *ok = accept ? 1 : 0;
    if (bases.size() > cap || (bases.size() && (!points || !scalars)))
      return curdle_set_last_error(CURDLE_EINVAL, "export buffers too small");
    if (!bases.empty()) {
      memcpy(points, bases.data(), bases.size() * 96);
      memcpy(scalars, sc.data(), sc.size() * 32);
    }
    return CURDLE_OK;
  });
}

// A batch over SEVERAL devices (curdle_init_devices): contiguous shards of the k proofs, one
// per context, each verified by its own group of host threads that select the shard's device
// first -- BASELINE config 5 (many independent verifications) is replicas: no data moves
// between devices, and the accept bits are exactly those of the single-device call.  run(lo,
// hi, rand, threads) verifies proofs [lo, hi) on the calling thread's device.
template <class Run>
static std::vector<int> ShardOverDevices(size_t k, common::Rand& rand, int nthreads, Run run) {
  const int D = curdle_device_count();
  if (D <= 1 || k < (size_t)(2 * D)) return run((size_t)0, k, rand, nthreads);
  if (nthreads < 1) nthreads = 1;
  std::vector<uint64_t> seeds((size_t)D);
  for (auto& sd : seeds) {  // each shard its own verifier randomness, drawn from the caller's
    Fr f;
    rand.GetFr(f);
    sd = (uint64_t)f.l[0] | ((uint64_t)f.l[1] << 32);
  }
  const int per = nthreads / D > 2 ? nthreads / D : 2;
  std::vector<std::vector<int>> res((size_t)D);
  std::vector<std::exception_ptr> errs((size_t)D);
  std::vector<std::thread> th;
  for (int d = 0; d < D; d++)
    th.emplace_back([&, d] {
      try {
        if (curdle_set_device(d) != CURDLE_OK) throw std::runtime_error("selecting a device for a batch shard");
        const size_t lo = k * (size_t)d / (size_t)D, hi = k * (size_t)(d + 1) / (size_t)D;
        common::Rand r(seeds[(size_t)d]);
        res[(size_t)d] = run(lo, hi, r, per);
      } catch (...) {
        errs[(size_t)d] = std::current_exception();
      }
    });
  for (auto& t : th) t.join();
  for (auto& e : errs)
    if (e) std::rethrow_exception(e);
  std::vector<int> all;
  all.reserve(k);
  for (auto& r : res) all.insert(all.end(), r.begin(), r.end());
  return all;
}

extern "C" int curdle_verify_batch(const curdle_crs* crs, size_t k, const uint8_t* const* proofs, const size_t* proof_lens,
                                   const uint64_t* const* Rs, const uint64_t* const* Ss, const uint64_t* const* Ts,
                                   const uint64_t* const* Us, size_t ell, const uint64_t* Ms, curdle_rand* rand,
                                   int nthreads, int* oks) {
  if (!crs || !rand || !oks || (k && (!proofs || !proof_lens || !Rs || !Ss || !Ts || !Us || !Ms)))
    return curdle_set_last_error(CURDLE_EINVAL, "null argument");
  if (ell != crs->crs.Gs.size()) return curdle_set_last_error(CURDLE_EINVAL, "ell does not match the CRS");
  return Guard([&]() {
    std::vector<proto::BatchItem> items(k);
    for (size_t i = 0; i < k; i++) {
      if (!proofs[i] || !Rs[i] || !Ss[i] || !Ts[i] || !Us[i]) throw std::runtime_error("null argument in batch item");
      items[i] = proto::BatchItem{proofs[i],
                                  proof_lens[i],
                                  reinterpret_cast<const G1Affine*>(Rs[i]),
                                  reinterpret_cast<const G1Affine*>(Ss[i]),
                                  reinterpret_cast<const G1Affine*>(Ts[i]),
                                  reinterpret_cast<const G1Affine*>(Us[i]),
                                  ell,
                                  Ms + 18 * i};
    }
    std::vector<int> res = ShardOverDevices(k, rand->r, nthreads, [&](size_t lo, size_t hi, common::Rand& r, int threads) {
      if (lo == 0 && hi == k) return proto::VerifyBatch(crs->crs, items, r, threads);
      std::vector<proto::BatchItem> shard(items.begin() + lo, items.begin() + hi);
      return proto::VerifyBatch(crs->crs, shard, r, threads);
    });
    for (size_t i = 0; i < k; i++) oks[i] = res[i];
    return CURDLE_OK;
  });
}

// ---- whisk package (whisk/whisk.go) ----
static std::vector<whisk::WhiskTracker> Trackers(const uint8_t* p, size_t n) {
  std::vector<whisk::WhiskTracker> v(n);
  static_assert(sizeof(whisk::WhiskTracker) == 96, "tracker layout");
  if (n) memcpy(v.data(), p, n * 96);
  return v;
}

extern "C" int curdle_whisk_is_valid_shuffle_proof(const curdle_crs* crs, const uint8_t* pre_trackers,
                                                   const uint8_t* post_trackers, size_t n_pre, size_t n_post,
                                                   const uint8_t* proof, curdle_rand* rand, int* ok) {
  if (!crs || !proof || !rand || !ok || (n_pre && !pre_trackers) || (n_post && !post_trackers))
    return curdle_set_last_error(CURDLE_EINVAL, "null argument");
  *ok = 0;
  return Guard([&]() {
    *ok = whisk::IsValidWhiskShuffleProof(crs->crs, Trackers(pre_trackers, n_pre), Trackers(post_trackers, n_post), proof,
                                          rand->r)
              ? 1
              : 0;
    return CURDLE_OK;
  });
}

extern "C" int curdle_whisk_is_valid_shuffle_proof_batch(const curdle_crs* crs, size_t k, const uint8_t* const* pre_trackers,
                                                         const uint8_t* const* post_trackers, size_t n,
                                                         const uint8_t* const* proofs, curdle_rand* rand, int nthreads,
                                                         int* oks) {
  if (!crs || !rand || !oks || (k && (!pre_trackers || !post_trackers || !proofs)))
    return curdle_set_last_error(CURDLE_EINVAL, "null argument");
  return Guard([&]() {
    std::vector<whisk::ShuffleBatchItem> items(k);
    for (size_t i = 0; i < k; i++) {
      if (!pre_trackers[i] || !post_trackers[i] || !proofs[i]) throw std::runtime_error("null argument in batch item");
      items[i] = whisk::ShuffleBatchItem{reinterpret_cast<const whisk::WhiskTracker*>(pre_trackers[i]),
                                         reinterpret_cast<const whisk::WhiskTracker*>(post_trackers[i]), n, proofs[i]};
    }
    std::vector<int> res = ShardOverDevices(k, rand->r, nthreads, [&](size_t lo, size_t hi, common::Rand& r, int threads) {
      if (lo == 0 && hi == k) return whisk::IsValidWhiskShuffleProofBatch(crs->crs, items, r, threads);
      std::vector<whisk::ShuffleBatchItem> shard(items.begin() + lo, items.begin() + hi);
      return whisk::IsValidWhiskShuffleProofBatch(crs->crs, shard, r, threads);
    });
    for (size_t i = 0; i < k; i++) oks[i] = res[i];
    return CURDLE_OK;
  });
}

extern "C" int curdle_whisk_generate_shuffle_proof(const curdle_crs* crs, const uint8_t* pre_trackers, size_t n,
                                                   curdle_rand* rand, uint8_t* post_trackers_out, uint8_t* proof_out) {
  if (!crs || !pre_trackers || !rand || !post_trackers_out || !proof_out)
    return curdle_set_last_error(CURDLE_EINVAL, "null argument");
  return Guard([&]() {
    std::vector<whisk::WhiskTracker> post = whisk::GenerateWhiskShuffleProof(crs->crs, Trackers(pre_trackers, n), rand->r, proof_out);
    memcpy(post_trackers_out, post.data(), post.size() * 96);
    return CURDLE_OK;
  });
}

extern "C" int curdle_whisk_is_valid_tracker_proof(const uint8_t* tracker, const uint8_t* k_commitment, const uint8_t* proof,
                                                   int* ok) {
  if (!tracker || !k_commitment || !proof || !ok) return curdle_set_last_error(CURDLE_EINVAL, "null argument");
  *ok = 0;
  return Guard([&]() {
    whisk::WhiskTracker t;
    memcpy(&t, tracker, 96);
    *ok = whisk::IsValidWhiskTrackerProof(t, k_commitment, proof) ? 1 : 0;
    return CURDLE_OK;
  });
}

extern "C" int curdle_whisk_generate_tracker_proof(const uint8_t* tracker, const uint64_t k[4], curdle_rand* rand,
                                                   uint8_t* proof_out) {
  if (!tracker || !k || !rand || !proof_out) return curdle_set_last_error(CURDLE_EINVAL, "null argument");
  return Guard([&]() {
    whisk::WhiskTracker t;
    memcpy(&t, tracker, 96);
    whisk::GenerateWhiskTrackerProof(t, Scalar::FromMont(k), rand->r, proof_out);
    return CURDLE_OK;
  });
}

extern "C" int curdle_verify_set_eager(int eager) { return proto::SetEagerChecks(eager); }

// Round trip of the wire format: decode, re-encode (curdleproof_test.go "encode/decode").
extern "C" int curdle_proof_reencode(const uint8_t* proof, size_t proof_len, uint8_t* out, size_t cap, size_t* out_len) {
  if (!proof || !out_len) return curdle_set_last_error(CURDLE_EINVAL, "null argument");
  return Guard([&]() {
    proto::Proof p = proto::Proof::FromBytes(proof, proof_len, /*subgroup_check=*/true);
    std::vector<uint8_t> bytes = p.Serialize();
    *out_len = bytes.size();
    if (!out || cap < bytes.size()) return curdle_set_last_error(CURDLE_EINVAL, "buffer too small");
    memcpy(out, bytes.data(), bytes.size());
    return CURDLE_OK;
  });
}

// common.IPA (/root/reference/common/util.go:26-35): inner product of two Fr vectors, Montgomery
// limbs in and out; the reference's own known answer (common/util_test.go:10-27) is asserted
// against this entry point.
extern "C" int curdle_fr_inner_product(const uint64_t* a, size_t a_len, const uint64_t* b, size_t b_len,
                                       uint64_t out_fr[4]) {
  if (!out_fr || (a_len && !a) || (b_len && !b)) return curdle_set_last_error(CURDLE_EINVAL, "null argument");
  return Guard([&]() {
    std::vector<Scalar> va(a_len), vb(b_len);
    for (size_t i = 0; i < a_len; i++) va[i] = Scalar::FromMont(a + 4 * i);
    for (size_t i = 0; i < b_len; i++) vb[i] = Scalar::FromMont(b + 4 * i);
    const Scalar r = alg::InnerProduct(va, vb);  // throws on a length mismatch, as util.go:27-29 errors
    memcpy(out_fr, &r.v, 32);
    return CURDLE_OK;
  });
}

// transcript pieces exposed for the known-answer test of the Merlin construction
extern "C" int curdle_merlin_test_vector(const char* protocol, const char* label, const uint8_t* msg, size_t msg_len,
                                         const char* challenge_label, uint8_t* out, size_t out_len) {
  if (!protocol || !label || !challenge_label || !out) return curdle_set_last_error(CURDLE_EINVAL, "null argument");
  transcript::Merlin m(protocol);
  m.AppendMessage(label, msg, msg_len);
  m.ChallengeBytes(challenge_label, out, out_len);
  return CURDLE_OK;
}

// G1 compressed encoding helpers (gnark G1Affine.Bytes / SetBytes)
extern "C" int curdle_g1_compress(const uint64_t jac[18], uint8_t out[48]) {
  if (!jac || !out) return curdle_set_last_error(CURDLE_EINVAL, "null argument");
  Point::FromJac(jac).Compressed(out);
  return CURDLE_OK;
}
extern "C" int curdle_g1_decompress(const uint8_t in[48], int subgroup_check, uint64_t out_jac[18]) {
  if (!in || !out_jac) return curdle_set_last_error(CURDLE_EINVAL, "null argument");
  Point p;
  if (!Point::FromCompressed(in, &p, subgroup_check != 0)) return curdle_set_last_error(CURDLE_EINVAL, "invalid point");
  p.Jac(out_jac);
  return CURDLE_OK;
}
