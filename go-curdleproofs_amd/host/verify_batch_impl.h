// Core of the cross-proof batch verification, shared by proto::VerifyBatch (proofs + decoded
// instances from the caller) and whisk::IsValidWhiskShuffleProofBatch (everything from bytes).
// `Source` supplies, per proof i:  bool Usable(i)  (false: rejected before any verification),
// Proof DecodeProof(i)  and  void Instance(i, Rs, Ss, Ts, Us, M)  -- all callable from the
// worker threads.  See the comment on proto::VerifyBatch for the scheme.
#pragma once
#include <stdlib.h>

#include <atomic>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/curdle_msm.h"
#include "curdleproofs.h"
#include "device_accumulator.h"

namespace curdle {
namespace proto {

template <class Source>
std::vector<int> VerifyBatchCore(const CRS& crs, size_t k, Source& src, common::Rand& rand, int nthreads) {
  using msmaccumulator::MsmAccumulator;
  std::vector<int> oks(k, 0);
  if (k == 0) return oks;
  std::vector<uint64_t> seeds(k);
  for (size_t i = 0; i < k; i++) {
    Fr f;
    rand.GetFr(f);
    seeds[i] = (uint64_t)f.l[0] | ((uint64_t)f.l[1] << 32);
  }
  if (nthreads < 1) nthreads = 1;
  if ((size_t)nthreads > k) nthreads = (int)k;
  size_t flush = 32;
  if (const char* e = getenv("CURDLE_BATCH_GROUP")) flush = (size_t)atoi(e);
  if (flush < 1) flush = 1;

  std::atomic<size_t> next(0);
  std::atomic<bool> failed(false);
  std::string first_error;
  int first_rc = CURDLE_EHIP;
  std::mutex err_mu;
  // By default a group's checks go to the device accumulator as descriptions: the CRS resident,
  // the members' instances back to back, the slot scalars built by index on the GPU, one MSM
  // per group -- no host-side vectors, no 96-byte-key map.  Eager
  // mode and CURDLE_DEVICE_ACC=0 keep the host mirror.
  const bool on_device = !EagerChecksEnabled() && DeviceAccumulatorEnabled() && crs.device != nullptr;
  auto device_worker = [&]() {
    std::vector<G1Affine> inst;
    std::vector<curdle_dacc_check> checks;
    std::vector<Scalar> pool, extra_scalars;
    std::vector<G1Affine> extra_points;
    std::vector<size_t> members;
    auto settle = [&]() {
      if (members.empty()) return;
      const bool all = RunRecordedChecks(crs, inst, checks, pool, extra_points, extra_scalars);
      for (size_t i : members) {
        if (all) {
          oks[i] = 1;
          continue;
        }
        try {  // some check of the group failed: find out whose
          Proof p = src.DecodeProof(i);
          common::Rand r(seeds[i]);
          std::vector<G1Affine> Rs, Ss, Ts, Us;
          Point M;
          src.Instance(i, Rs, Ss, Ts, Us, M);
          oks[i] = Verify(p, crs, Rs, Ss, Ts, Us, M, r) ? 1 : 0;
        } catch (const alg::MsmError&) {
          throw;  // device failure, not a verdict
        } catch (const std::runtime_error&) {
          oks[i] = 0;
        }
      }
      inst.clear();
      checks.clear();
      pool.clear();
      extra_points.clear();
      extra_scalars.clear();
      members.clear();
    };
    try {
      for (size_t i = next.fetch_add(1); i < k && !failed.load(); i = next.fetch_add(1)) {
        CheckRecorder rec;  // joins the group only if the proof's direct checks pass
        std::vector<G1Affine> Rs, Ss, Ts, Us;
        bool pre = false;
        try {
          if (!src.Usable(i)) throw std::runtime_error("malformed proof or instance");
          Proof p = src.DecodeProof(i);
          common::Rand r(seeds[i]);
          Point M;
          src.Instance(i, Rs, Ss, Ts, Us, M);
          pre = VerifyWithSink(p, crs, Rs, Ss, Ts, Us, M, r, rec);
        } catch (const alg::MsmError&) {
          throw;
        } catch (const std::runtime_error&) {
          pre = false;  // malformed proof / zero randomizer: rejected in a batch
        }
        if (!pre) continue;
        rec.AppendTo(inst.size(), &checks, &pool, &extra_points, &extra_scalars);
        inst.insert(inst.end(), Rs.begin(), Rs.end());  // InstIndex: Rs | Ss | Ts | Us
        inst.insert(inst.end(), Ss.begin(), Ss.end());
        inst.insert(inst.end(), Ts.begin(), Ts.end());
        inst.insert(inst.end(), Us.begin(), Us.end());
        members.push_back(i);
        if (members.size() >= flush || extra_points.size() + 512 > CURDLE_DACC_MAX_EXTRA) settle();
      }
      settle();
    } catch (const alg::MsmError& e) {
      std::lock_guard<std::mutex> g(err_mu);
      if (!failed.exchange(true)) {
        first_error = e.what();
        first_rc = e.rc;
      }
    } catch (const std::exception& e) {
      std::lock_guard<std::mutex> g(err_mu);
      if (!failed.exchange(true)) first_error = e.what();
    }
  };
  auto worker = [&]() {
    if (on_device) return device_worker();
    std::vector<G1Affine> bases;
    std::vector<Scalar> scalars;
    std::vector<size_t> members;
    Point a_c = Point::Infinity();
    auto settle = [&]() {
      if (members.empty()) return;
      const bool all = alg::MultiExp(bases, scalars) == a_c;  // the group's one MSM, on the GPU
      for (size_t i : members) {
        if (all) {
          oks[i] = 1;
          continue;
        }
        try {  // some accumulated check of the group failed: find out whose
          Proof p = src.DecodeProof(i);
          common::Rand r(seeds[i]);
          std::vector<G1Affine> Rs, Ss, Ts, Us;
          Point M;
          src.Instance(i, Rs, Ss, Ts, Us, M);
          oks[i] = Verify(p, crs, Rs, Ss, Ts, Us, M, r) ? 1 : 0;
        } catch (const alg::MsmError&) {
          throw;  // device failure, not a verdict
        } catch (const std::runtime_error&) {
          oks[i] = 0;
        }
      }
      bases.clear();
      scalars.clear();
      members.clear();
      a_c = Point::Infinity();
    };
    try {
      for (size_t i = next.fetch_add(1); i < k && !failed.load(); i = next.fetch_add(1)) {
        MsmAccumulator mine;  // joins the group only if the proof's direct checks pass
        bool pre = false;
        try {
          if (!src.Usable(i)) throw std::runtime_error("malformed proof or instance");
          Proof p = src.DecodeProof(i);
          common::Rand r(seeds[i]);
          // the instance copies happen here, on the worker, not serially before the batch starts
          std::vector<G1Affine> Rs, Ss, Ts, Us;
          Point M;
          src.Instance(i, Rs, Ss, Ts, Us, M);
          pre = VerifyInto(p, crs, Rs, Ss, Ts, Us, M, r, mine);
        } catch (const alg::MsmError&) {
          throw;  // device failure (eager mode computes MSMs here), not a verdict
        } catch (const std::runtime_error&) {
          pre = false;  // malformed proof / zero randomizer: rejected in a batch
        }
        if (!pre) continue;
        bases.insert(bases.end(), mine.Bases().begin(), mine.Bases().end());
        for (const Fr& f : mine.Scalars()) {
          Scalar sc;
          sc.v = f;
          scalars.push_back(sc);
        }
        Point ac;
        ac.p = mine.A_c;
        a_c = a_c + ac;
        members.push_back(i);
        if (members.size() >= flush) settle();
      }
      settle();
    } catch (const alg::MsmError& e) {  // device failure inside an MSM: the whole call fails
      std::lock_guard<std::mutex> g(err_mu);
      if (!failed.exchange(true)) {
        first_error = e.what();
        first_rc = e.rc;
      }
    } catch (const std::exception& e) {
      std::lock_guard<std::mutex> g(err_mu);
      if (!failed.exchange(true)) first_error = e.what();
    }
  };
  std::vector<std::thread> th;
  for (int t = 1; t < nthreads; t++) th.emplace_back(worker);
  worker();
  for (auto& x : th) x.join();
  if (failed.load()) throw alg::MsmError("batch verification: " + first_error, first_rc);
  return oks;
}

}  // namespace proto
}  // namespace curdle
