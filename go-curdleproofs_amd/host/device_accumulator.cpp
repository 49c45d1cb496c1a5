// DeviceSink -- see device_accumulator.h.
#include "knobs.h"
#include "device_accumulator.h"

#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <mutex>
#include <stdexcept>

namespace curdle {
namespace proto {

namespace {
alg::MsmError device_error(const char* what, int rc) {
  char buf[256];
  curdle_last_error(buf, sizeof(buf));
  return alg::MsmError(std::string(what) + ": " + buf + " (rc " + std::to_string(rc) + ")", rc);
}
std::atomic<int>& Flag() {
  static std::atomic<int> on(knobs::get(knobs::DEVICE_ACC) == 0 ? 0 : 1);
  return on;
}
}  // namespace

bool DeviceAccumulatorEnabled() { return Flag().load(std::memory_order_relaxed) != 0; }
int SetDeviceAccumulator(int on) { return Flag().exchange(on ? 1 : 0); }

DeviceCrs::~DeviceCrs() { curdle_dbases_free(h_); }

const curdle_dbases* DeviceCrs::Get(const CRS& crs) {
  std::lock_guard<std::mutex> g(mu_);
  if (h_ && !curdle_dbases_valid(h_)) {  // the library was shut down and re-initialised since
    curdle_dbases_free(h_);
    h_ = nullptr;
  }
  if (!h_) {
    std::vector<G1Affine> pts(crs.Gs);  // Gs | Hs | H | Gt | Gu  (CrsIndex)
    pts.insert(pts.end(), crs.Hs.begin(), crs.Hs.end());
    pts.push_back(crs.H.Affine());
    pts.push_back(crs.Gt.Affine());
    pts.push_back(crs.Gu.Affine());
    int rc = curdle_dbases_create(reinterpret_cast<const uint64_t*>(pts.data()), pts.size(), &h_);
    if (rc != CURDLE_OK) throw device_error("making the CRS resident", rc);
  }
  return h_;
}

DeviceSink::DeviceSink(const CRS& crs)
    : ell_(crs.Gs.size()), n_crs_(CrsIndex{crs.Gs.size()}.size()), n_inst_(4 * crs.Gs.size()), crs_(crs),
      inst_{nullptr, nullptr, nullptr, nullptr} {
  if (!crs.device) throw std::runtime_error("CRS without a device holder");
}

DeviceSink::DeviceSink(const CRS& crs, const std::vector<G1Affine>& Rs, const std::vector<G1Affine>& Ss,
                       const std::vector<G1Affine>& Ts, const std::vector<G1Affine>& Us)
    : DeviceSink(crs) {
  Begin(Rs, Ss, Ts, Us);
}

void DeviceSink::Begin(const std::vector<G1Affine>& Rs, const std::vector<G1Affine>& Ss, const std::vector<G1Affine>& Ts,
                       const std::vector<G1Affine>& Us) {
  if (acc_ || consumed_) throw std::runtime_error("device accumulator already started");
  if (Rs.size() != ell_ || Ss.size() != ell_ || Ts.size() != ell_ || Us.size() != ell_)
    throw std::runtime_error("instance vectors do not match the CRS");
  inst_[0] = &Rs;
  inst_[1] = &Ss;
  inst_[2] = &Ts;
  inst_[3] = &Us;
  const curdle_dbases* bases = crs_.device->Get(crs_);
  // Rs | Ss | Ts | Us (InstIndex), uploaded now: converted on the GPU while the host hashes
  std::vector<G1Affine> inst;
  inst.reserve(n_inst_);
  for (const auto* v : inst_) inst.insert(inst.end(), v->begin(), v->end());
  int rc = curdle_dacc_begin(bases, reinterpret_cast<const uint64_t*>(inst.data()), inst.size(), &acc_);
  if (rc != CURDLE_OK) throw device_error("starting the device accumulator", rc);
}

DeviceSink::~DeviceSink() {
  if (acc_) curdle_dacc_abort(acc_);
}

uint32_t CheckRecorder::Put(const Scalar& s) {
  pool.push_back(s);
  return (uint32_t)(pool.size() - 1);
}

void CheckRecorder::Check(const Terms& C, const VecExpr& x, const std::vector<BaseSeg>& segs,
                          const std::vector<LooseBase>& loose, common::Rand& rand, const char* what) {
  if (segs.size() > CURDLE_DACC_MAX_SEGS) throw std::runtime_error(std::string(what) + ": too many base segments");
  Scalar alpha;
  rand.GetFr(alpha.v);  // msmaccumulator.go:32 -- the same draw, in the same order, as the host mirror
  curdle_dacc_check ck;
  memset(&ck, 0, sizeof(ck));
  ck.kind = (uint32_t)x.kind;
  ck.n_struct = (uint32_t)x.n_struct;
  ck.m = (uint32_t)x.gammas.size();
  ck.q_cap = (uint32_t)x.q_cap;
  ck.alpha_off = Put(alpha);
  ck.weight_off = Put(alpha * x.scale);
  ck.gammas_off = (uint32_t)pool.size();
  for (const Scalar& g : x.gammas) Put(g);
  ck.q_off = Put(x.q);
  ck.tail_off = (uint32_t)pool.size();
  ck.n_tail = (uint32_t)x.tail.size();
  for (const Scalar& t : x.tail) Put(t);
  ck.nseg = (uint32_t)segs.size();
  for (size_t s = 0; s < segs.size(); s++) {
    ck.seg[s].set = segs[s].set;
    ck.seg[s].first = segs[s].first;
    ck.seg[s].len = segs[s].len;
    ck.seg[s].vec_first = segs[s].vec_first;
  }
  checks.push_back(ck);
  // bases outside the resident sets: the proof points of the same-scalar argument
  for (const LooseBase& lb : loose) {
    if (lb.pending >= 0)
      pending_extras.emplace_back((uint32_t)extra_points.size(), (uint32_t)lb.pending);
    else if (g1_affine_is_inf(lb.point))
      continue;
    extra_points.push_back(lb.point);
    extra_scalars.push_back(alpha * x.At(lb.index));
  }
  // C moves to the base side: - alpha c_j P_j  (MsmAccumulator::AccumulateCheckDeferred)
  for (size_t j = 0; j < C.p.size(); j++) {
    if (C.pending[j] >= 0)
      pending_extras.emplace_back((uint32_t)extra_points.size(), (uint32_t)C.pending[j]);
    else if (g1_affine_is_inf(C.p[j]))
      continue;
    extra_points.push_back(C.p[j]);
    extra_scalars.push_back(-(alpha * C.s[j]));
  }
}

void CheckRecorder::Resolve(const PointDecoder& dec) {
  std::vector<uint32_t> drop;
  for (const auto& pe : pending_extras) {
    G1Affine a;
    if (!dec.GetAffine(pe.second, &a)) throw std::runtime_error("decoding proof: invalid point");
    if (g1_affine_is_inf(a))
      drop.push_back(pe.first);
    else
      extra_points[pe.first] = a;
  }
  pending_extras.clear();
  for (size_t k = drop.size(); k-- > 0;) {  // ascending indices: erase from the back
    extra_points.erase(extra_points.begin() + drop[k]);
    extra_scalars.erase(extra_scalars.begin() + drop[k]);
  }
}

void CheckRecorder::AppendTo(size_t inst_base, std::vector<curdle_dacc_check>* group_checks, std::vector<Scalar>* group_pool,
                             std::vector<G1Affine>* group_extra_points, std::vector<Scalar>* group_extra_scalars) const {
  const uint32_t pb = (uint32_t)group_pool->size();
  for (curdle_dacc_check ck : checks) {
    ck.weight_off += pb;
    ck.alpha_off += pb;
    ck.gammas_off += pb;
    ck.q_off += pb;
    ck.tail_off += pb;
    for (uint32_t s = 0; s < ck.nseg; s++)
      if (ck.seg[s].set == kSetInst) ck.seg[s].first += (uint32_t)inst_base;
    group_checks->push_back(ck);
  }
  group_pool->insert(group_pool->end(), pool.begin(), pool.end());
  group_extra_points->insert(group_extra_points->end(), extra_points.begin(), extra_points.end());
  group_extra_scalars->insert(group_extra_scalars->end(), extra_scalars.begin(), extra_scalars.end());
}

bool RunRecordedChecks(const CRS& crs, const std::vector<G1Affine>& inst, const std::vector<curdle_dacc_check>& checks,
                       const std::vector<Scalar>& pool, const std::vector<G1Affine>& extra_points,
                       const std::vector<Scalar>& extra_scalars) {
  if (!crs.device) throw std::runtime_error("CRS without a device holder");
  const curdle_dbases* bases = crs.device->Get(crs);
  curdle_dacc* acc = nullptr;
  int rc = curdle_dacc_begin(bases, reinterpret_cast<const uint64_t*>(inst.data()), inst.size(), &acc);
  if (rc != CURDLE_OK) throw device_error("starting the device accumulator", rc);
  uint64_t out[18];
  rc = curdle_dacc_run(acc, checks.data(), checks.size(), reinterpret_cast<const uint64_t*>(pool.data()), pool.size(),
                       reinterpret_cast<const uint64_t*>(extra_points.data()),
                       reinterpret_cast<const uint64_t*>(extra_scalars.data()), extra_points.size(), out, nullptr);
  if (rc != CURDLE_OK) throw device_error("verifying msm accumulator: computing msm", rc);
  return Point::FromJac(out).IsInfinity();
}

RecordedChecksRun::~RecordedChecksRun() {
  if (acc_) curdle_dacc_abort(acc_);
}

void RecordedChecksRun::Start(const CRS& crs, const std::vector<G1Affine>& inst, const std::vector<curdle_dacc_check>& checks,
                              const std::vector<Scalar>& pool, const std::vector<G1Affine>& extra_points,
                              const std::vector<Scalar>& extra_scalars) {
  if (acc_) throw std::logic_error("a recorded-checks run is already in flight");
  if (!crs.device) throw std::runtime_error("CRS without a device holder");
  const curdle_dbases* bases = crs.device->Get(crs);
  curdle_dacc* acc = nullptr;
  int rc = curdle_dacc_begin(bases, reinterpret_cast<const uint64_t*>(inst.data()), inst.size(), &acc);
  if (rc != CURDLE_OK) throw device_error("starting the device accumulator", rc);
  rc = curdle_dacc_submit(acc, checks.data(), checks.size(), reinterpret_cast<const uint64_t*>(pool.data()), pool.size(),
                          reinterpret_cast<const uint64_t*>(extra_points.data()),
                          reinterpret_cast<const uint64_t*>(extra_scalars.data()), extra_points.size(), nullptr);
  if (rc != CURDLE_OK) throw device_error("verifying msm accumulator: computing msm", rc);  // the submission ended it
  acc_ = acc;
}

bool RecordedChecksRun::Done() {
  if (!acc_) return true;
  int done = 0;
  const int rc = curdle_dacc_poll(acc_, &done);
  if (rc != CURDLE_OK) throw device_error("verifying msm accumulator: computing msm", rc);
  return done != 0;
}

bool RecordedChecksRun::Finish() {
  if (!acc_) throw std::logic_error("no recorded-checks run in flight");
  curdle_dacc* a = acc_;
  acc_ = nullptr;  // wait() ends the accumulation whatever happens
  uint64_t out[18];
  const int rc = curdle_dacc_wait(a, out);
  if (rc != CURDLE_OK) throw device_error("verifying msm accumulator: computing msm", rc);
  return Point::FromJac(out).IsInfinity();
}

void DeviceSink::Check(const Terms& C, const VecExpr& x, const std::vector<BaseSeg>& segs,
                       const std::vector<LooseBase>& loose, common::Rand& rand, const char* what) {
  rec_.Check(C, x, segs, loose, rand, what);
}

bool DeviceSink::Run(std::vector<Scalar>* slot_scalars) {
  if (!acc_) throw std::runtime_error(consumed_ ? "device accumulator already consumed" : "device accumulator not started");
  if (!rec_.pending_extras.empty()) throw std::logic_error("device accumulator run with undecoded points");
  consumed_ = true;
  uint64_t out[18];
  if (slot_scalars) slot_scalars->assign(n_crs_ + n_inst_, Scalar::Zero());
  curdle_dacc* a = acc_;
  acc_ = nullptr;  // run() ends the accumulation whatever happens
  int rc = curdle_dacc_run(a, rec_.checks.data(), rec_.checks.size(), reinterpret_cast<const uint64_t*>(rec_.pool.data()), rec_.pool.size(),
                           reinterpret_cast<const uint64_t*>(rec_.extra_points.data()),
                           reinterpret_cast<const uint64_t*>(rec_.extra_scalars.data()), rec_.extra_points.size(), out,
                           slot_scalars ? reinterpret_cast<uint64_t*>(slot_scalars->data()) : nullptr);
  if (rc != CURDLE_OK) throw device_error("verifying msm accumulator: computing msm", rc);  // msmaccumulator.go:60
  return Point::FromJac(out).IsInfinity();  // :63, A_c is the point at infinity here
}

bool DeviceSink::Verify() { return Run(nullptr); }

bool DeviceSink::VerifyAndExport(std::vector<G1Affine>* bases, std::vector<Scalar>* scalars) {
  std::vector<Scalar> slots;
  const bool ok = Run(&slots);
  bases->assign(crs_.Gs.begin(), crs_.Gs.end());
  bases->insert(bases->end(), crs_.Hs.begin(), crs_.Hs.end());
  bases->push_back(crs_.H.Affine());
  bases->push_back(crs_.Gt.Affine());
  bases->push_back(crs_.Gu.Affine());
  for (const auto* v : inst_) bases->insert(bases->end(), v->begin(), v->end());
  bases->insert(bases->end(), rec_.extra_points.begin(), rec_.extra_points.end());
  *scalars = slots;
  scalars->insert(scalars->end(), rec_.extra_scalars.begin(), rec_.extra_scalars.end());
  return ok;
}

}  // namespace proto
}  // namespace curdle
