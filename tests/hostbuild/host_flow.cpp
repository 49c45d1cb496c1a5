// TEST INFRASTRUCTURE ONLY.  Drives the host layer of libcurdlemsm.so (everything under
// go-curdleproofs_amd/host/: wire-format readers, PointDecoder, MsmAccumulator table,
// transcript, the five arguments, curdleproof Prove / Verify) over the naive host backend of
// stub_backend.cpp, so it can run under AddressSanitizer + UBSan on a machine without a GPU
// -- the counterpart of the reference CI's `go test -race`
// (.github/workflows/buildlintcheck.yml:21).  Modes:
//   host_flow flow <ell>          Prove -> serialise -> Verify (deferred and eager), soundness
//                                 flips of curdleproof_test.go:48-182, encode/decode round trip
//   host_flow whisk 0             Whisk shuffle proof: generate, verify (both routes), tampered trackers
//   host_flow compress 0          the vectorised point compressor against the scalar one
//   host_flow fuzz <ell> <iters>  attacker-controlled bytes into every parser: bit flips,
//                                 truncations, random slice prefixes, random blobs
//   host_flow time <ell> <reps>   host share of Verify (everything but the final MSM), for gprof
//   host_flow emit <ell> <file>   the serialised proof + instance as a fixture file
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <stdexcept>
#include <string>
#include <vector>

#include <map>

#include "../../go-curdleproofs_amd/host/knobs.h"
#include "../../go-curdleproofs_amd/host/curdleproofs.h"
#include "../../go-curdleproofs_amd/host/device_accumulator.h"
#include "../../go-curdleproofs_amd/host/whisk.h"

using namespace curdle;
using alg::Point;
using alg::Scalar;

struct Instance {
  proto::CRS crs;
  std::vector<G1Affine> Rs, Ss, Ts, Us;
  Point M;
  std::vector<uint32_t> perm;
  Scalar k;
  std::vector<Scalar> rs_m;
  std::vector<uint8_t> proof;
};

static Instance Make(size_t ell, uint64_t seed, const proto::CRS* crs = nullptr) {
  Instance in;
  common::Rand rand(seed);
  if (crs)
    in.crs = *crs;  // another instance over the same CRS
  else
    in.crs = proto::GenerateCRS(ell, rand);
  common::Rand prand(seed + 42);
  prand.GeneratePermutation(ell, in.perm);
  rand.GetFr(in.k.v);
  rand.GetG1Affines(ell, in.Rs);
  rand.GetG1Affines(ell, in.Ss);
  proto::ShuffleCommit sc = proto::ShufflePermuteCommit(in.crs.Gs, in.crs.Hs, in.Rs, in.Ss, in.perm, in.k, rand);
  in.Ts = sc.Ts;
  in.Us = sc.Us;
  in.M = sc.M;
  in.rs_m = sc.rs_m;
  common::Rand pr(seed + 1000);
  in.proof = proto::Prove(in.crs, in.Rs, in.Ss, in.Ts, in.Us, in.M, in.perm, in.k, in.rs_m, pr).Serialize();
  return in;
}

static bool VerifyBytes(const Instance& in, const std::vector<uint8_t>& bytes, uint64_t seed) {
  proto::Proof p = proto::Proof::FromBytes(bytes.data(), bytes.size(), true);
  common::Rand vr(seed);
  return proto::Verify(p, in.crs, in.Rs, in.Ss, in.Ts, in.Us, in.M, vr);
}

// curdle_verify's default route (proto_api.cpp): one lazy pass over the bytes, the whole host part
// run while the points are "still being decoded", pending bases filled in at the end
static bool VerifyLazyBytes(const Instance& in, const std::vector<uint8_t>& bytes, uint64_t seed) {
  proto::PointDecoder dec(true);
  proto::Reader scan(bytes.data(), bytes.size(), true);
  scan.collect = &dec;
  const proto::Proof p = proto::Proof::ScanLazy(scan);
  dec.Start();
  proto::VerifyPrelude pre;
  proto::StartVerify(pre, in.Rs, in.Ss, in.Ts, in.Us, in.M);
  common::Rand vr(seed);
  const bool accept = proto::VerifyWhileDecoding(
      pre, p, in.crs, in.M, dec,
      [&](proto::DecodedInstance& inst) {
        G1Affine a;
        for (size_t i = 0; i < dec.size(); i++)
          if (!dec.GetAffine(i, &a)) throw std::runtime_error("decoding proof: invalid point");
        inst.Rs = in.Rs;
        inst.Ss = in.Ss;
        inst.Ts = in.Ts;
        inst.Us = in.Us;
      },
      vr);
  if (!dec.Finish()) throw std::runtime_error("decoding proof: invalid point (not in the prime-order subgroup)");
  return accept;
}

#define CHECK(cond)                                                          \
  do {                                                                       \
    if (!(cond)) {                                                           \
      fprintf(stderr, "host_flow: check failed at line %d: %s\n", __LINE__, #cond); \
      exit(1);                                                               \
    }                                                                        \
  } while (0)

// alg::CompressAffineBatch (AVX-512 IFMA, eight points per step, where the CPU has it) against
// the scalar CompressAffine: random points, every tail length, infinity, coordinates at the
// edges of the field
static int CompressCheck() {
  common::Rand r(99);
  std::vector<G1Affine> pts;
  r.GetG1Affines(67, pts);
  pts[3] = G1Affine{};   // infinity
  pts[40] = G1Affine{};
  // (x, y) pairs that are not curve points but exercise the reductions: Montgomery images of 0, 1, p - 1, (p - 1) / 2, (p + 1) / 2
  {
    alg::Point g = alg::Point::Generator();
    G1Affine ga = g.Affine();
    G1Affine t = ga;
    memset(&t.x, 0xff, sizeof(t.x));
    t.x.l[11] = 0x0a0111eau;  // a large value below p in Montgomery form
    pts[7] = t;
    t = ga;
    memset(&t.y, 0, sizeof(t.y));
    t.y.l[0] = 1;
    pts[8] = t;
  }
  for (size_t n = 0; n <= pts.size(); n++) {
    std::vector<uint8_t> a(48 * n + 1, 0xAB), b(48 * n + 1, 0xAB);
    alg::CompressAffineBatch(pts.data(), n, a.data());
    for (size_t i = 0; i < n; i++) alg::CompressAffine(pts[i], &b[48 * i]);
    CHECK(a == b);
  }
  // y and p - y: exactly one of them is "larger"
  for (size_t i = 0; i < 16; i++) {
    if (g1_affine_is_inf(pts[i])) continue;
    alg::Point neg = alg::Point::FromAffine(pts[i]).Neg();
    G1Affine both[8];
    for (int k = 0; k < 8; k++) both[k] = (k & 1) ? neg.Affine() : pts[i];
    uint8_t out[8 * 48];
    alg::CompressAffineBatch(both, 8, out);
    CHECK(((out[0] ^ out[48]) & 0x20) == 0x20);
  }
  printf("compress: batch == scalar for every length up to %zu\n", pts.size());
  return 0;
}

static int Flow(size_t ell) {
  Instance in = Make(ell, 7);
  {  // the prover that never folds its bases (default) against the reference's order of operations
    curdle::knobs::set("PROVER_FOLD_BASES", 1);
    Instance ref = Make(ell, 7);
    curdle::knobs::set("PROVER_FOLD_BASES", -1);
    CHECK(ref.proof == in.proof);
  }
  for (int eager = 0; eager < 2; eager++) {
    proto::SetEagerChecks(eager);
    CHECK(VerifyBytes(in, in.proof, 43));  // TestCompleteness, curdleproof_test.go:14-46
    // encode / decode round trip
    proto::Proof p = proto::Proof::FromBytes(in.proof.data(), in.proof.size(), true);
    CHECK(p.Serialize() == in.proof);
    // TestSoundness (:48-182): swapped R/S, another permutation, wrong commitment, other randomiser
    {
      Instance bad = in;
      std::swap(bad.Rs, bad.Ss);
      CHECK(!VerifyBytes(bad, in.proof, 44));
    }
    {
      Instance bad = in;
      std::swap(bad.Ts[0], bad.Ts[1]);
      CHECK(!VerifyBytes(bad, in.proof, 45));
    }
    {
      Instance bad = in;
      bad.M = bad.M + Point::Generator();
      CHECK(!VerifyBytes(bad, in.proof, 46));
    }
    {
      Instance bad = in;
      for (auto& t : bad.Ts) t = (Point::FromAffine(t) + Point::FromAffine(t)).Affine();
      CHECK(!VerifyBytes(bad, in.proof, 47));
    }
    {  // zero randomiser is a structural error (curdleproof.go:213-215)
      Instance bad = in;
      memset(&bad.Ts[0], 0, sizeof(G1Affine));
      bool threw = false;
      try {
        VerifyBytes(bad, in.proof, 48);
      } catch (const std::runtime_error&) {
        threw = true;
      }
      CHECK(threw);
    }
  }
  proto::SetEagerChecks(0);
  // msmaccumulator table: merge of shared bases, growth past the initial capacity
  {
    msmaccumulator::MsmAccumulator acc;
    common::Rand r(5);
    std::vector<G1Affine> v;
    r.GetG1Affines(40, v);
    for (int round = 0; round < 60; round++) {
      std::vector<Fr> x(v.size());
      for (auto& f : x) r.GetFr(f);
      std::vector<alg::Scalar> xs(x.size());
      for (size_t i = 0; i < x.size(); i++) xs[i].v = x[i];
      Point C = alg::MultiExp(v, xs);
      uint64_t j[18];
      C.Jac(j);
      G1Jac cj;
      memcpy(&cj, j, sizeof(cj));
      CHECK(acc.AccumulateCheck(cj, x, v, &r).ok);
      if (round % 7 == 0) {  // a fresh base now and then: the table grows
        G1Affine extra;
        r.GetG1Affine(extra);
        v.push_back(extra);
      }
    }
    CHECK(acc.NumBases() == v.size());
    bool ok = false;
    CHECK(acc.Verify(&ok).ok && ok);
  }
  // The two accumulators -- host mirror and the device accumulator's description path (here over
  // the stub's host evaluation of the same descriptions) -- hold the same base -> scalar map.
  {
    proto::Proof p = proto::Proof::FromBytes(in.proof.data(), in.proof.size(), true);
    auto key = [](const G1Affine& a) { return std::string(reinterpret_cast<const char*>(&a), sizeof(a)); };
    std::map<std::string, Scalar> mirror, device;
    {
      msmaccumulator::MsmAccumulator acc;
      common::Rand vr(99);
      CHECK(proto::VerifyInto(p, in.crs, in.Rs, in.Ss, in.Ts, in.Us, in.M, vr, acc));
      for (size_t i = 0; i < acc.NumBases(); i++) {
        Scalar sc;
        sc.v = acc.Scalars()[i];
        if (!g1_affine_is_inf(acc.Bases()[i])) mirror[key(acc.Bases()[i])] = sc;
      }
    }
    {
      proto::DeviceSink sink(in.crs, in.Rs, in.Ss, in.Ts, in.Us);
      common::Rand vr(99);
      CHECK(proto::VerifyWithSink(p, in.crs, in.Rs, in.Ss, in.Ts, in.Us, in.M, vr, sink));
      std::vector<G1Affine> bases;
      std::vector<Scalar> sc;
      CHECK(sink.VerifyAndExport(&bases, &sc));
      for (size_t i = 0; i < bases.size(); i++) {
        auto it = device.find(key(bases[i]));
        if (it == device.end()) device[key(bases[i])] = sc[i];
        else it->second = it->second + sc[i];
      }
    }
    for (auto it = mirror.begin(); it != mirror.end();) it = it->second.IsZero() ? mirror.erase(it) : std::next(it);
    for (auto it = device.begin(); it != device.end();) it = it->second.IsZero() ? device.erase(it) : std::next(it);
    CHECK(mirror.size() == device.size());
    for (const auto& kv : mirror) {
      auto it = device.find(kv.first);
      CHECK(it != device.end() && it->second == kv.second);
    }
    // and the default Verify (device accumulator) agrees with the mirror on accept and reject
    CHECK(proto::SetDeviceAccumulator(1) == 1);
    CHECK(VerifyBytes(in, in.proof, 50));
    Instance bad = in;
    std::swap(bad.Us[0], bad.Us[1]);
    CHECK(!VerifyBytes(bad, in.proof, 51));
    proto::SetDeviceAccumulator(0);
    CHECK(VerifyBytes(in, in.proof, 50));
    CHECK(!VerifyBytes(bad, in.proof, 51));
    proto::SetDeviceAccumulator(1);
    // ... and so does the verification that runs while the proof is still being decoded
    CHECK(proto::CanVerifyWhileDecoding());
    CHECK(VerifyLazyBytes(in, in.proof, 50));
    CHECK(!VerifyLazyBytes(bad, in.proof, 51));
    {
      Instance zero = in;  // curdleproof.go:213-215
      zero.Ts[0] = G1Affine{};
      bool threw = false;
      try {
        VerifyLazyBytes(zero, in.proof, 52);
      } catch (const std::runtime_error& e) {
        threw = std::string(e.what()).find("randomizer is zero") != std::string::npos;
      }
      CHECK(threw);
    }
  }
  // cross-proof batch verification, on the device accumulator's description path and on the
  // host mirror: exact per-proof bits with a bad instance, a truncated and a bit-flipped proof;
  // decoded two proofs per chunk, ahead of the workers
  {
    curdle::knobs::set("BATCH_CHUNK", 2);
    Instance other = Make(ell, 21, &in.crs);
    std::vector<uint8_t> truncated(in.proof.begin(), in.proof.end() - 5), flipped(in.proof);
    flipped[flipped.size() - 9] ^= 0x20;
    std::vector<proto::BatchItem> items;
    uint64_t Mj[2][18];
    in.M.Jac(Mj[0]);
    other.M.Jac(Mj[1]);
    auto item = [&](const std::vector<uint8_t>& pf, const Instance& inst, const uint64_t* M) {
      return proto::BatchItem{pf.data(), pf.size(), inst.Rs.data(), inst.Ss.data(), inst.Ts.data(), inst.Us.data(), ell, M};
    };
    items.push_back(item(in.proof, in, Mj[0]));
    items.push_back(item(other.proof, other, Mj[1]));
    items.push_back(item(in.proof, other, Mj[1]));      // another proof's instance
    items.push_back(item(truncated, in, Mj[0]));
    items.push_back(item(flipped, in, Mj[0]));
    items.push_back(item(other.proof, other, Mj[1]));
    const std::vector<int> want = {1, 1, 0, 0, 0, 1};
    for (int dev = 1; dev >= 0; dev--) {
      proto::SetDeviceAccumulator(dev);
      common::Rand br(5);
      const std::vector<int> got = proto::VerifyBatch(in.crs, items, br, 2);
      if (got != want) {
        fprintf(stderr, "batch (device accumulator %d):", dev);
        for (int v : got) fprintf(stderr, " %d", v);
        fprintf(stderr, "\n");
      }
      CHECK(got == want);
      common::Rand br2(6);
      std::vector<proto::BatchItem> good = {items[0], items[1], items[5]};
      CHECK(proto::VerifyBatch(in.crs, good, br2, 3) == (std::vector<int>{1, 1, 1}));
    }
    curdle::knobs::set("BATCH_CHUNK", -1);
    proto::SetDeviceAccumulator(1);
  }
  printf("flow ell=%zu: completeness, round trip, soundness flips, accumulator table, mirror == device accumulator, batch: ok\n", ell);
  return 0;
}

// Batch verification against a finite slot pool (stub_backend.cpp, CURDLE_STUB_SLOTS): more
// workers than workspace slots, every group a single proof (so every worker holds a queued
// group's slot when it asks for its next proof), chunks of one proof decoded by producers that
// need a slot themselves (CURDLE_TWO_KERNEL_MAX=0) and are slow.  A worker that waited for a
// decoding while holding its slot would leave the producers without one: this run must finish.
static int BatchSlots(size_t ell) {
  Instance a = Make(ell, 7), b = Make(ell, 21, &a.crs);
  uint64_t Mj[2][18];
  a.M.Jac(Mj[0]);
  b.M.Jac(Mj[1]);
  std::vector<proto::BatchItem> items;
  std::vector<int> want;
  for (int i = 0; i < 48; i++) {
    const Instance& in = i % 2 ? b : a;
    const bool bad = i % 11 == 5;
    const Instance& inst = bad ? (i % 2 ? a : b) : in;  // another proof's instance
    items.push_back(proto::BatchItem{in.proof.data(), in.proof.size(), inst.Rs.data(), inst.Ss.data(), inst.Ts.data(),
                                     inst.Us.data(), ell, Mj[(i % 2) ^ (bad ? 1 : 0)]});
    want.push_back(bad ? 0 : 1);
  }
  common::Rand br(5);
  const std::vector<int> got = proto::VerifyBatch(a.crs, items, br, 16);
  CHECK(got == want);
  CHECK(curdle_msm_free_slots() == CURDLE_MSM_SLOTS);  // every slot handed back
  printf("batch with 16 threads over %d slots: ok\n", CURDLE_MSM_SLOTS);
  return 0;
}

// The batch entry point of the C ABI over SEVERAL devices: the stub poses as CURDLE_STUB_DEVICES
// contexts and counts the device-entry-point calls each one sees.  The k proofs are sharded over
// the contexts (host/proto_api.cpp ShardOverDevices), every thread of a shard selects its device
// first, and the accept bits are those of the single-device run.
struct curdle_rand {  // same layout as in misc_api.hip / proto_api.cpp
  common::Rand r;
  explicit curdle_rand(uint64_t seed) : r(seed) {}
};
struct curdle_crs {
  proto::CRS crs;
};
extern "C" unsigned long long curdle_stub_calls_on(int ordinal);
static int BatchDevices(size_t ell) {
  const int D = curdle_device_count();
  CHECK(D >= 2);
  CHECK(curdle_set_device(D) == CURDLE_EINVAL && curdle_set_device(-1) == CURDLE_EINVAL);
  CHECK(curdle_set_device(D - 1) == CURDLE_OK && curdle_get_device() == D - 1);
  CHECK(curdle_set_device(0) == CURDLE_OK);
  Instance a = Make(ell, 7), b = Make(ell, 21, &a.crs);
  curdle_crs crs{a.crs};
  uint64_t Mj[2][18];
  a.M.Jac(Mj[0]);
  b.M.Jac(Mj[1]);
  const size_t k = 26;
  std::vector<const uint8_t*> proofs(k);
  std::vector<size_t> lens(k);
  std::vector<const uint64_t*> Rs(k), Ss(k), Ts(k), Us(k);
  std::vector<uint64_t> Ms(18 * k);
  std::vector<int> want(k), got(k, -1);
  for (size_t i = 0; i < k; i++) {
    const Instance& in = i % 2 ? b : a;
    const bool bad = i % 7 == 3;
    const Instance& inst = bad ? (i % 2 ? a : b) : in;  // another proof's instance
    proofs[i] = in.proof.data();
    lens[i] = in.proof.size();
    Rs[i] = reinterpret_cast<const uint64_t*>(inst.Rs.data());
    Ss[i] = reinterpret_cast<const uint64_t*>(inst.Ss.data());
    Ts[i] = reinterpret_cast<const uint64_t*>(inst.Ts.data());
    Us[i] = reinterpret_cast<const uint64_t*>(inst.Us.data());
    memcpy(&Ms[18 * i], Mj[(i % 2) ^ (bad ? 1 : 0)], sizeof(Mj[0]));
    want[i] = bad ? 0 : 1;
  }
  curdle_rand r(5);
  CHECK(curdle_verify_batch(&crs, k, proofs.data(), lens.data(), Rs.data(), Ss.data(), Ts.data(), Us.data(), ell, Ms.data(), &r,
                            2 * D, got.data()) == CURDLE_OK);
  CHECK(got == want);
  for (int d = 0; d < D; d++) CHECK(curdle_stub_calls_on(d) > 0);  // every posed device took part
  // fewer proofs than two per device: the batch stays on the caller's device
  const unsigned long long before = curdle_stub_calls_on(D - 1);
  std::vector<int> few(3, -1);
  CHECK(curdle_verify_batch(&crs, 3, proofs.data(), lens.data(), Rs.data(), Ss.data(), Ts.data(), Us.data(), ell, Ms.data(), &r, 2,
                            few.data()) == CURDLE_OK);
  CHECK(few == (std::vector<int>{1, 1, 1}));
  CHECK(curdle_stub_calls_on(D - 1) == before);
  printf("batch sharded over %d devices: ok\n", D);
  return 0;
}

// the Whisk byte API end to end (whisk.go:20, :63) over Whisk's own CRS size: the route that
// verifies while the points are being decoded (device accumulator) and the two-pass one
static int WhiskFlow() {
  const size_t ell = whisk::ELL;
  common::Rand cr(3);
  Instance in;
  in.crs = proto::GenerateCRS(ell, cr);
    common::Rand wr(77);
    std::vector<G1Affine> rg, krg;
    wr.GetG1Affines(ell, rg);
    wr.GetG1Affines(ell, krg);
    std::vector<whisk::WhiskTracker> pre(ell);
    for (size_t i = 0; i < ell; i++) pre[i] = whisk::NewWhiskTracker(rg[i], krg[i]);
    std::vector<uint8_t> wproof(whisk::WHISK_SHUFFLE_PROOF_SIZE);
    std::vector<whisk::WhiskTracker> post;
    bool generated = true;
    try {
      post = whisk::GenerateWhiskShuffleProof(in.crs, pre, wr, wproof.data());
    } catch (const std::runtime_error&) {
      generated = false;  // the fixed-size proof array only fits Whisk's own ell
    }
    if (generated) {
      for (int dev = 1; dev >= 0; dev--) {
        proto::SetDeviceAccumulator(dev);
        common::Rand v1(78), v2(79), v3(80);
        CHECK(whisk::IsValidWhiskShuffleProof(in.crs, pre, post, wproof.data(), v1));
        std::vector<whisk::WhiskTracker> swapped(post);
        std::swap(swapped[0], swapped[1]);
        CHECK(!whisk::IsValidWhiskShuffleProof(in.crs, pre, swapped, wproof.data(), v2));
        std::vector<whisk::WhiskTracker> broken(post);
        memset(broken[2].krG, 0xff, 48);  // not an encoding
        bool threw = false;
        try {
          whisk::IsValidWhiskShuffleProof(in.crs, pre, broken, wproof.data(), v3);
        } catch (const std::runtime_error& e) {
          threw = std::string(e.what()).find("getting post shuffle points") != std::string::npos;
        }
        CHECK(threw);
      }
      proto::SetDeviceAccumulator(1);
      // the batch form, decoded chunk by chunk ahead of the workers (one proof per chunk here)
      curdle::knobs::set("BATCH_CHUNK", 1);
      std::vector<whisk::WhiskTracker> swapped(post);
      std::swap(swapped[0], swapped[1]);
      std::vector<uint8_t> garbled(wproof);
      garbled[100] ^= 0x10;
      std::vector<whisk::ShuffleBatchItem> items = {
          {pre.data(), post.data(), ell, wproof.data()},    {pre.data(), swapped.data(), ell, wproof.data()},
          {pre.data(), post.data(), ell, garbled.data()},   {pre.data(), post.data(), ell, wproof.data()},
          {pre.data(), post.data(), ell, wproof.data()}};
      for (int dev = 1; dev >= 0; dev--) {
        proto::SetDeviceAccumulator(dev);
        common::Rand br(81);
        CHECK(whisk::IsValidWhiskShuffleProofBatch(in.crs, items, br, 3) == (std::vector<int>{1, 0, 0, 1, 1}));
      }
      proto::SetDeviceAccumulator(1);
      curdle::knobs::set("BATCH_CHUNK", -1);
    }
  CHECK(generated);
  printf("whisk flow: shuffle proof accepted, swapped trackers rejected, broken tracker reported, both routes: ok\n");
  return 0;
}

static uint64_t g_rng = 0x9e3779b97f4a7c15ull;
static uint64_t Rng() {
  g_rng ^= g_rng << 13;
  g_rng ^= g_rng >> 7;
  g_rng ^= g_rng << 17;
  return g_rng;
}

static int Fuzz(size_t ell, int iters) {
  Instance in = Make(ell, 9);
  long accepted = 0, rejected = 0, errors = 0;
  auto attempt = [&](const std::vector<uint8_t>& bytes) {
    int eager = 2, lazy = 2;  // 1 accept, 0 reject, 2 error
    try {
      eager = VerifyBytes(in, bytes, 77) ? 1 : 0;
    } catch (const std::runtime_error&) {
    }
    try {  // the verification that overlaps the decoding must classify every input the same way
      lazy = VerifyLazyBytes(in, bytes, 77) ? 1 : 0;
    } catch (const std::runtime_error&) {
    }
    CHECK(eager == lazy);
    if (eager == 1)
      accepted++;
    else if (eager == 0)
      rejected++;
    else
      errors++;
  };
  for (int it = 0; it < iters; it++) {
    std::vector<uint8_t> b = in.proof;
    switch (it % 5) {
      case 0:  // one flipped bit
        b[Rng() % b.size()] ^= (uint8_t)(1u << (Rng() % 8));
        break;
      case 1:  // truncation
        b.resize(Rng() % b.size());
        break;
      case 2: {  // a slice prefix replaced by an arbitrary 32-bit value
        size_t at = Rng() % (b.size() - 4);
        uint32_t v = (uint32_t)Rng();
        if (it % 2) v &= 0xff;
        memcpy(&b[at], &v, 4);
        break;
      }
      case 3:  // random tail
        for (size_t i = Rng() % b.size(); i < b.size(); i++) b[i] = (uint8_t)Rng();
        break;
      default:  // random blob of random length
        b.resize(Rng() % (2 * b.size()));
        for (auto& c : b) c = (uint8_t)Rng();
    }
    attempt(b);
  }
  // whisk byte API: tracker proofs and shuffle proofs from arbitrary bytes
  for (int it = 0; it < iters / 4 + 1; it++) {
    whisk::WhiskTracker t;
    uint8_t kc[48], tp[128];
    for (auto& c : t.rG) c = (uint8_t)Rng();
    for (auto& c : t.krG) c = (uint8_t)Rng();
    for (auto& c : kc) c = (uint8_t)Rng();
    for (auto& c : tp) c = (uint8_t)Rng();
    try {
      if (whisk::IsValidWhiskTrackerProof(t, kc, tp)) accepted++;
    } catch (const std::runtime_error&) {
      errors++;
    }
  }
  CHECK(accepted == 0);
  printf("fuzz ell=%zu: %d mutated proofs + tracker proofs: %ld rejected, %ld decode errors, 0 accepted\n", ell, iters,
         rejected, errors);
  return 0;
}

static int Time(size_t ell, int reps) {
  Instance in = Make(ell, 11);
  proto::Proof p = proto::Proof::FromBytes(in.proof.data(), in.proof.size(), true);
  auto t0 = std::chrono::steady_clock::now();
  size_t bases = 0;
  for (int r = 0; r < reps; r++) {
    msmaccumulator::MsmAccumulator acc;
    common::Rand vr(100 + r);
    CHECK(proto::VerifyInto(p, in.crs, in.Rs, in.Ss, in.Ts, in.Us, in.M, vr, acc));
    bases = acc.NumBases();
  }
  double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / reps;
  printf("time ell=%zu: host share of Verify (all but the final MSM of %zu bases): %.3f ms\n", ell, bases, ms);
  return 0;
}

// host share of the default Verify: the device accumulator's description path (no MSM)
static int TimeDevicePath(size_t ell, int reps) {
  Instance in = Make(ell, 11);
  proto::Proof p = proto::Proof::FromBytes(in.proof.data(), in.proof.size(), true);
  auto t0 = std::chrono::steady_clock::now();
  for (int r = 0; r < reps; r++) {
    proto::DeviceSink sink(in.crs, in.Rs, in.Ss, in.Ts, in.Us);
    common::Rand vr(100 + r);
    CHECK(proto::VerifyWithSink(p, in.crs, in.Rs, in.Ss, in.Ts, in.Us, in.M, vr, sink));
  }
  double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / reps;
  printf("time ell=%zu: host share of Verify on the device-accumulator path: %.3f ms\n", ell, ms);
  return 0;
}

static int Emit(size_t ell, const char* path) {
  Instance in = Make(ell, 7);
  FILE* f = fopen(path, "wb");
  if (!f) return 1;
  fwrite(in.proof.data(), 1, in.proof.size(), f);
  fclose(f);
  printf("emit ell=%zu: %zu proof bytes\n", ell, in.proof.size());
  return 0;
}

int main(int argc, char** argv) {
  if (argc < 3) {
    fprintf(stderr, "usage: host_flow flow|fuzz|time|emit <ell> [arg]\n");
    return 2;
  }
  const std::string mode = argv[1];
  const size_t ell = (size_t)atoi(argv[2]);
  try {
    if (mode == "flow") return Flow(ell);
    if (mode == "whisk") return WhiskFlow();
    if (mode == "compress") return CompressCheck();
    if (mode == "slots") return BatchSlots(ell);
    if (mode == "devices") return BatchDevices(ell);
    if (mode == "fuzz") return Fuzz(ell, argc > 3 ? atoi(argv[3]) : 200);
    if (mode == "time") return Time(ell, argc > 3 ? atoi(argv[3]) : 20);
    if (mode == "timedev") return TimeDevicePath(ell, argc > 3 ? atoi(argv[3]) : 20);
    if (mode == "emit") return Emit(ell, argc > 3 ? argv[3] : "proof.bin");
  } catch (const std::exception& e) {
    fprintf(stderr, "host_flow: unexpected exception: %s\n", e.what());
    return 1;
  }
  return 2;
}
