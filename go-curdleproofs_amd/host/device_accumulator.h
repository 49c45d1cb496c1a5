// The verifier's accumulator on the GPU: a CheckSink that ships the
// sub-arguments' checks, as descriptions, to the device accumulator of include/curdle_msm.h
// (curdle_dacc_*).  Replaces, for curdleproof.Verify, the host map of
// /root/reference/msmaccumulator/msmaccumulator.go:11-64:
//   * the CRS bases Gs | Hs | H | Gt | Gu (crs.go:10-18) are converted once and stay resident
//     (proto::CRS carries the handle); the 4 ell instance points are uploaded when the
//     verification STARTS and are converted while the host hashes the transcript;
//   * AccumulateCheck's `map[v_i] += alpha x_i` (:38-43) becomes an index-addressed add in an
//     Fr kernel, which also evaluates the verifier's s_i / s'_i / x s_i vectors
//     (innerproductargument.go:223-234, samemultiscalarargument.go:267-277) from the log n
//     challenges -- the host only draws alpha (same order, same values as the mirror);
//   * the check points C and the loose bases (proof points) travel as ~100 (point, scalar)
//     pairs; Verify()'s MultiExp (:59) runs over the slots in place.
// The accept bit equals the host mirror's for every input (tests/test_device_accumulator.py
// compares the exported base / scalar lists of both).
#pragma once
#include <memory>
#include <mutex>
#include <vector>

#include "../../include/curdle_msm.h"
#include "curdleproofs.h"

namespace curdle {
namespace proto {

// The CRS's device-resident bases; created on first use, shared by copies of the CRS.
class DeviceCrs {
 public:
  DeviceCrs() = default;
  ~DeviceCrs();
  DeviceCrs(const DeviceCrs&) = delete;
  DeviceCrs& operator=(const DeviceCrs&) = delete;
  const curdle_dbases* Get(const CRS& crs);  // throws alg::MsmError without a device

 private:
  std::mutex mu_;
  curdle_dbases* h_ = nullptr;
};

// CURDLE_DEVICE_ACC=0 (or SetDeviceAccumulator(0)) keeps curdleproof.Verify on the host mirror.
bool DeviceAccumulatorEnabled();
int SetDeviceAccumulator(int on);  // returns the previous setting

// The checks of ONE verification as descriptions for the device accumulator: what DeviceSink
// ships, without owning a device accumulation, so that several recordings can be concatenated
// into one group (cross-proof batch verification: one resident CRS, the members' instances one
// after the other, one MSM per group).
class CheckRecorder : public CheckSink {
 public:
  void Check(const Terms& C, const VecExpr& x, const std::vector<BaseSeg>& segs, const std::vector<LooseBase>& loose,
             common::Rand& rand, const char* what) override;
  // Appends this recording to a group whose instance set already holds `inst_base` slots and
  // whose pool holds group_pool->size() elements: offsets are rebased.
  void AppendTo(size_t inst_base, std::vector<curdle_dacc_check>* group_checks, std::vector<Scalar>* group_pool,
                std::vector<G1Affine>* group_extra_points, std::vector<Scalar>* group_extra_scalars) const;
  std::vector<curdle_dacc_check> checks;
  std::vector<Scalar> pool;
  std::vector<G1Affine> extra_points;
  std::vector<Scalar> extra_scalars;
  // Loose pairs whose point was still being decoded when the check was recorded
  // (VerifyWhileDecoding): (index into extra_points, PointDecoder record).  Resolve() fills them
  // in -- a pair whose point turns out to be infinity is dropped, like every other -- and throws
  // the decoding error if a record is not a valid point.
  std::vector<std::pair<uint32_t, uint32_t>> pending_extras;
  void Resolve(const PointDecoder& dec);

 private:
  uint32_t Put(const Scalar& s);
};

// One MSM over a resident CRS, `inst` (the members' instance points, back to back) and the
// loose pairs, with the slot scalars built on the device from `checks`: true iff the sum is the
// point at infinity (every recorded check holds, up to the soundness error of the random
// weights).  Throws alg::MsmError on a device failure.
bool RunRecordedChecks(const CRS& crs, const std::vector<G1Affine>& inst, const std::vector<curdle_dacc_check>& checks,
                       const std::vector<Scalar>& pool, const std::vector<G1Affine>& extra_points,
                       const std::vector<Scalar>& extra_scalars);

// The same in two steps: Start queues the group's MSM and returns; Done() polls; Finish() waits
// and gives the verdict.  The group holds one of the library's eight workspace slots from Start
// to Finish, so a caller polls between its other work and finishes as soon as Done() says so.
class RecordedChecksRun {
 public:
  RecordedChecksRun() = default;
  ~RecordedChecksRun();
  RecordedChecksRun(const RecordedChecksRun&) = delete;
  RecordedChecksRun& operator=(const RecordedChecksRun&) = delete;
  void Start(const CRS& crs, const std::vector<G1Affine>& inst, const std::vector<curdle_dacc_check>& checks,
             const std::vector<Scalar>& pool, const std::vector<G1Affine>& extra_points,
             const std::vector<Scalar>& extra_scalars);  // every argument is copied before it returns
  bool Active() const { return acc_ != nullptr; }
  bool Done();
  bool Finish();

 private:
  curdle_dacc* acc_ = nullptr;
};

class DeviceSink : public CheckSink {
 public:
  DeviceSink(const CRS& crs, const std::vector<G1Affine>& Rs, const std::vector<G1Affine>& Ss,
             const std::vector<G1Affine>& Ts, const std::vector<G1Affine>& Us);
  // Recording only: for a verification whose instance is still being decoded.  Begin() uploads
  // it (the vectors must outlive the sink's Verify) and must come before Verify().
  explicit DeviceSink(const CRS& crs);
  void Begin(const std::vector<G1Affine>& Rs, const std::vector<G1Affine>& Ss, const std::vector<G1Affine>& Ts,
             const std::vector<G1Affine>& Us);
  void Resolve(const PointDecoder& dec) { rec_.Resolve(dec); }
  ~DeviceSink() override;
  void Check(const Terms& C, const VecExpr& x, const std::vector<BaseSeg>& segs, const std::vector<LooseBase>& loose,
             common::Rand& rand, const char* what) override;
  // msmAccumulator.Verify(): the MSM over every slot, compared with A_c (the point at infinity:
  // every C was moved to the base side).  Ends the accumulation.
  bool Verify();
  // The same, also handing back what the device accumulated: bases (CRS | instance | loose) and
  // their scalars (Montgomery), for the parity tests.
  bool VerifyAndExport(std::vector<G1Affine>* bases, std::vector<Scalar>* scalars);

 private:
  bool Run(std::vector<Scalar>* slot_scalars);
  curdle_dacc* acc_ = nullptr;
  size_t ell_, n_crs_, n_inst_;
  const CRS& crs_;
  const std::vector<G1Affine>*inst_[4];
  bool consumed_ = false;
  CheckRecorder rec_;
};

}  // namespace proto
}  // namespace curdle
