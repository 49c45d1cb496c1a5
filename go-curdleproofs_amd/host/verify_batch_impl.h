// Core of the cross-proof batch verification, shared by proto::VerifyBatch (proofs + decoded
// instances from the caller) and whisk::IsValidWhiskShuffleProofBatch (everything from bytes).
// `Source` supplies, per proof i:  bool Usable(i)  (false: rejected before any verification),
// Proof DecodeProof(i)  and  void Instance(i, Rs, Ss, Ts, Us, M)  -- all callable from the
// worker threads.  See the comment on proto::VerifyBatch for the scheme.
#pragma once
#include <stdlib.h>

#include <stdio.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <exception>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "knobs.h"
#include "../../include/curdle_msm.h"
#include "curdleproofs.h"
#include "device_accumulator.h"

namespace curdle {
namespace proto {

// The batch's point decoding, in chunks, AHEAD of the verifying workers: producer threads walk
// the wire format of the proofs of chunk c (`scan(i, dec)` registers proof i's records with the
// chunk's decoder; a proof that does not parse throws and is simply not usable) and run the
// chunk's ONE GPU decoding, while the workers are busy with the chunks before it.  One decoding
// over the whole batch first -- what this replaces -- left the GPU idle during the serial scan
// and every host core idle during the decoding (k = 1,024 Whisk proofs: 17 + 32 ms of 100).
// Workers call Wait(i) before they touch proof i; proofs are handed out in index order, so
// chunks are consumed in the order they are produced.
inline bool BatchTrace() { return knobs::get(knobs::VERIFY_TRACE) > 0; }  // the batch's chunk / group timeline (stderr)

class DecodeAhead {
 public:
  using ScanFn = std::function<void(size_t i, PointDecoder& dec)>;
  DecodeAhead(size_t k, size_t chunk, int producers, ScanFn scan)
      : k_(k), chunk_(chunk < 1 ? 1 : chunk), scan_(std::move(scan)) {
    const size_t nchunks = (k_ + chunk_ - 1) / chunk_;
    decs_.resize(nchunks);
    ready_.assign(nchunks, 0);
    if (producers < 1) producers = 1;
    if ((size_t)producers > nchunks) producers = (int)nchunks;
    // the producers decode on the device of the thread that started the batch (a shard of a
    // multi-device batch lives on one device: every thread of it selects that device first)
    const int device = curdle_get_device();
    for (int t = 0; t < producers; t++)
      threads_.emplace_back([this, device] {
        (void)curdle_set_device(device);
        Produce();
      });
  }
  ~DecodeAhead() {
    stop_.store(true);
    for (auto& t : threads_) t.join();
  }
  DecodeAhead(const DecodeAhead&) = delete;
  DecodeAhead& operator=(const DecodeAhead&) = delete;
  // The decoder holding proof i's records, once its chunk is decoded.  Rethrows a producer's
  // failure (a device error: the whole batch fails, as before).
  const PointDecoder& Wait(size_t i) {
    const size_t c = i / chunk_;
    std::unique_lock<std::mutex> g(mu_);
    cv_.wait(g, [&] { return ready_[c] != 0 || error_ != nullptr; });
    if (!ready_[c]) std::rethrow_exception(error_);
    return *decs_[c];
  }
  // Is proof i's chunk decoded already (or has a producer failed, so that Wait would not block)?
  bool Ready(size_t i) {
    std::lock_guard<std::mutex> g(mu_);
    return ready_[i / chunk_] != 0 || error_ != nullptr;
  }
  void Abandon() { stop_.store(true); }  // the workers gave up: producers stop after their current chunk

 private:
  void Produce() {
    try {
      for (size_t c = next_.fetch_add(1); c < decs_.size() && !stop_.load(); c = next_.fetch_add(1)) {
        auto dec = std::make_unique<PointDecoder>(/*subgroup_check=*/true);
        const size_t end = (c + 1) * chunk_ < k_ ? (c + 1) * chunk_ : k_;
        const auto t0 = std::chrono::steady_clock::now();
        for (size_t i = c * chunk_; i < end; i++) scan_(i, *dec);
        const auto t1 = std::chrono::steady_clock::now();
        dec->Run();
        if (BatchTrace()) {
          const auto t2 = std::chrono::steady_clock::now();
          auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
          fprintf(stderr, "[batch] chunk %zu (%zu points): at %.2f ms, scan %.2f ms, decode %.2f ms\n", c, dec->size(), ms(start_, t0),
                  ms(t0, t1), ms(t1, t2));
        }
        std::lock_guard<std::mutex> g(mu_);
        decs_[c] = std::move(dec);
        ready_[c] = 1;
        cv_.notify_all();
      }
    } catch (...) {
      std::lock_guard<std::mutex> g(mu_);
      if (!error_) error_ = std::current_exception();
      cv_.notify_all();
    }
  }
  size_t k_, chunk_;
  ScanFn scan_;
  std::chrono::steady_clock::time_point start_ = std::chrono::steady_clock::now();
  std::vector<std::unique_ptr<PointDecoder>> decs_;
  std::vector<char> ready_;
  std::mutex mu_;
  std::condition_variable cv_;
  std::exception_ptr error_;
  std::atomic<size_t> next_{0};
  std::atomic<bool> stop_{false};
  std::vector<std::thread> threads_;
};

// Chunk size and producer count for a batch of k proofs.  64 proofs per chunk (a quarter of
// the batch below 256 proofs, never under 16): a decoding kernel is a ~1.1 ms dependent chain
// however few points it has, so small chunks cost producer time, but its waves sit on their
// SIMDs at a fraction of the issue rate for that long, so LARGE chunks starve the groups' MSMs
// of SIMD slots -- measured at k = 1,024 (profiles/r02_batch_chunk_sweep.txt): ell = 252
// verify_batch 58-66 ms at 64 proofs per chunk, 140-180 ms at 128, 68-82 ms undivided; Whisk
// 63 ms at 64, 97-111 ms undivided.  Two producers (a third takes a core from the workers).
inline size_t DecodeAheadChunk(size_t k, size_t points_per_proof = 0) {
  if (knobs::get(knobs::BATCH_CHUNK) > 0) return (size_t)knobs::get(knobs::BATCH_CHUNK);  // tests, tuning
  size_t c = k / 4;
  c = c < 16 ? 16 : c > 64 ? 64 : c;
  // ... and at most 32,768 points: up to there a decoding is the pair of concurrent four-lane
  // kernels (0.7 ms) instead of the fused one-lane kernel (2.2 ms for the 37,440 points of 64
  // Whisk proofs) -- 1,024 Whisk proofs: 39-41 ms at 64 proofs per chunk, 33 ms at 56
  if (points_per_proof && c * points_per_proof > 32768) c = 32768 / points_per_proof;
  return c < 16 ? 16 : c;
}
inline int DecodeAheadProducers() {
  if (knobs::get(knobs::BATCH_PRODUCERS) > 0) return (int)knobs::get(knobs::BATCH_PRODUCERS);
  return 2;
}
// The caller's thread budget covers the producers: `nthreads` threads in all while they run.
inline int BatchWorkers(int nthreads) {
  const int w = nthreads - DecodeAheadProducers();
  return nthreads >= 4 ? (w < 2 ? 2 : w) : nthreads;
}

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
  if (knobs::get(knobs::BATCH_GROUP) > 0) flush = (size_t)knobs::get(knobs::BATCH_GROUP);
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
  // Workers start together and take equal time per proof, so with one threshold they would all
  // settle their groups at the same moment -- 16 MSMs queueing for the GPU, then none for a
  // while.  Each worker's FIRST group is cut at its own fraction of the threshold.
  std::atomic<int> worker_ids(0);
  auto device_worker = [&]() {
    const int wid = worker_ids.fetch_add(1);
    size_t threshold = flush * (size_t)(wid + 1) / (size_t)nthreads;
    if (threshold < 4) threshold = flush < 4 ? flush : 4;
    std::vector<G1Affine> inst;
    std::vector<curdle_dacc_check> checks;
    std::vector<Scalar> pool, extra_scalars;
    std::vector<G1Affine> extra_points;
    std::vector<size_t> members;
    // A group's MSM runs while this worker verifies the proofs of its NEXT group: settle() only
    // queues it (RecordedChecksRun::Start copies every argument), collect() takes the verdict --
    // polled after every proof, because the group holds a workspace slot until then -- and only a
    // group that failed is gone through member by member.
    RecordedChecksRun run;
    std::vector<size_t> in_flight;  // the members of the group `run` is computing
    std::chrono::steady_clock::time_point run_t0;
    auto collect = [&]() {
      if (!run.Active()) return;
      const bool all = run.Finish();
      if (BatchTrace())
        fprintf(stderr, "[batch] group of %zu proofs: verdict %.2f ms after its submission\n", in_flight.size(),
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - run_t0).count());
      for (size_t i : in_flight) {
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
      in_flight.clear();
    };
    auto settle = [&]() {
      if (members.empty()) return;
      collect();  // at most one group in flight per worker
      run_t0 = std::chrono::steady_clock::now();
      run.Start(crs, inst, checks, pool, extra_points, extra_scalars);
      in_flight.swap(members);
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
        // Never wait for a decoding while holding a workspace slot: the queued group keeps its slot
        // until collect(), and the producer that decodes proof i's chunk may itself need a slot
        // (chunks beyond the two-kernel size, or every decode context taken) -- with all eight
        // slots held by waiting workers nobody would move.  The group's verdict is taken first.
        if (run.Active() && !src.Ready(i)) collect();
        try {
          if (!src.Usable(i)) throw std::runtime_error("malformed proof or instance");
          Proof p = src.DecodeProof(i);
          common::Rand r(seeds[i]);
          Point M;
          src.Instance(i, Rs, Ss, Ts, Us, M);
          VerifyPrelude from_bytes;  // a source that has the instance's encodings starts the transcript from them
          const bool have = src.Prelude(i, from_bytes);
          pre = VerifyWithSink(p, crs, Rs, Ss, Ts, Us, M, r, rec, have ? &from_bytes : nullptr);
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
        if (members.size() >= threshold || extra_points.size() + 512 > CURDLE_DACC_MAX_EXTRA) {
          settle();
          threshold = flush;
        } else if (run.Active() && run.Done()) {
          collect();
        }
      }
      settle();
      collect();
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
  const int device = curdle_get_device();  // the batch runs where its caller is
  for (int t = 1; t < nthreads; t++)
    th.emplace_back([&worker, device] {
      (void)curdle_set_device(device);
      worker();
    });
  worker();
  for (auto& x : th) x.join();
  if (failed.load()) throw alg::MsmError("batch verification: " + first_error, first_rc);
  return oks;
}

}  // namespace proto
}  // namespace curdle
