// Curdleproofs protocol layers around the MSM hot path -- see curdleproofs.h.
// Each function cites the reference lines it restates (paths relative to
// /root/reference).  Transcript labels are those of SURVEY.md appendix A.
#include "knobs.h"
#include "curdleproofs.h"
#include "device_accumulator.h"
#include "verify_batch_impl.h"

#include <stdlib.h>

#include "../../include/curdle_msm.h"

#include <stdio.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <mutex>
#include <thread>

#include <stdexcept>

namespace curdle {
namespace proto {

using transcript::Transcript;
using msmaccumulator::MsmAccumulator;

namespace {

std::runtime_error err(const std::string& s) { return std::runtime_error(s); }

std::vector<Scalar> GetFrs(common::Rand& rand, size_t n) {
  std::vector<Fr> raw;
  rand.GetFrs(n, raw);
  std::vector<Scalar> out(n);
  for (size_t i = 0; i < n; i++) out[i].v = raw[i];
  return out;
}
Scalar GetFr(common::Rand& rand) {
  Scalar s;
  rand.GetFr(s.v);
  return s;
}

// msmAccumulator.AccumulateCheck(C, x, v, rand) on the host mirror.
void Accumulate(MsmAccumulator& acc, const Point& C, const std::vector<Scalar>& x, const std::vector<G1Affine>& v,
                common::Rand& rand, const char* what) {
  std::vector<Fr> xs(x.size());
  for (size_t i = 0; i < x.size(); i++) xs[i] = x[i].v;
  msmaccumulator::Status st = acc.AccumulateCheckXYZZ(C.p, xs, v, &rand);
  if (!st.ok) throw err(std::string(what) + ": " + st.err);
}

// The verifier's check points (the C of each AccumulateCheck) are linear combinations of
// proof and statement points.  By default they are handed to the accumulator as such
// (MsmAccumulator::AccumulateCheckDeferred), so one verification is ONE MSM on the GPU
// and no alpha * C scalar multiplications on the host.  CURDLE_VERIFY_EAGER=1 evaluates
// every C where the reference does (MultiExp per argument, then AccumulateCheck) --
// same accept bit, kept for differential testing.
std::atomic<int>& EagerFlag() {
  static std::atomic<int> eager(knobs::get(knobs::VERIFY_EAGER) > 0 ? 1 : 0);
  return eager;
}
bool EagerChecks() { return EagerFlag().load(std::memory_order_relaxed) != 0; }

// The eager value of a check point: one MSM on the GPU.
Point EvalTerms(const Terms& t) { return alg::MultiExp(t.p, t.s); }

template <class T>
std::vector<T> Concat(const std::vector<T>& a, const std::vector<T>& b) {
  std::vector<T> out(a);
  out.insert(out.end(), b.begin(), b.end());
  return out;
}

// common.Permute (util.go:37): ret[i] = vs[perm[i]]
template <class T>
std::vector<T> Permute(const std::vector<T>& vs, const std::vector<uint32_t>& perm) {
  std::vector<T> out(vs.size());
  for (size_t i = 0; i < perm.size(); i++) out[i] = vs[perm[i]];
  return out;
}

int Log2Exact(size_t n, const char* what) {
  if (n == 0 || (n & (n - 1))) throw err(std::string(what) + " is not a power of two");
  int m = 0;
  while (((size_t)1 << m) < n) m++;
  return m;
}

G1Affine AffineOf(const Point& p) { return p.Affine(); }

// The value of a point the verifier must compute with before the GPU has decoded it: square root
// on the host (~15 us), no subgroup test -- the GPU's verdict on the same record still gates the
// result (VerifyWhileDecoding).
Point Decoded(const Point& p) {
  if (p.pending < 0) return p;
  Point out;
  if (!Point::FromCompressed(p.wire, &out, /*subgroup_check=*/false)) throw err("decoding proof: invalid point");
  out.wire = p.wire;
  return out;
}

const G1Affine kZeroPoint = [] {
  G1Affine z;
  f_zero(z.x);
  f_zero(z.y);
  return z;
}();

// Knob PROVER_FOLD_BASES = 1: the recursive provers fold their bases every round, in the
// reference's order of operations (A/B and tests, which flip it through curdle_plan_override).
bool ProverFoldsBases() { return knobs::get(knobs::PROVER_FOLD_BASES) > 0; }

}  // namespace

int SetEagerChecks(int eager) { return EagerFlag().exchange(eager ? 1 : 0); }
bool EagerChecksEnabled() { return EagerChecks(); }

// ================================================= deferred-check descriptions =====
Scalar VecExpr::At(size_t i) const {
  if (i >= n_struct) return tail.at(i - n_struct);
  if (kind == kConst) return scale;
  Scalar v = scale;
  const size_t m = gammas.size();
  for (size_t j = 0; j < m; j++)
    if ((i >> j) & 1u) v = v * gammas[m - 1 - j];
  if (kind == kFoldPow) v = v * q.Pow((i < q_cap ? i : q_cap) + 1);
  return v;
}

std::vector<Scalar> VecExpr::Materialise() const {
  std::vector<Scalar> out(size());
  if (kind == kConst) {
    for (size_t i = 0; i < n_struct; i++) out[i] = scale;
  } else if (kind == kFold || kind == kFoldPow) {
    // index i with top bit j extends index i - 2^j: n products instead of n m / 2
    const size_t m = gammas.size();
    if (n_struct) out[0] = scale;
    for (size_t j = 0; j < m; j++)
      for (size_t i = (size_t)1 << j; i < ((size_t)2 << j) && i < n_struct; i++)
        out[i] = out[i - ((size_t)1 << j)] * gammas[m - j - 1];
    if (kind == kFoldPow) {
      Scalar qi = q;  // q^(i+1)
      for (size_t i = 0; i < n_struct; i++) {
        out[i] = out[i] * qi;
        if (i < q_cap) qi = qi * q;
      }
    }
  }
  for (size_t i = 0; i < tail.size(); i++) out[n_struct + i] = tail[i];
  return out;
}

MirrorSink::MirrorSink(MsmAccumulator& acc, const CRS& crs, const std::vector<G1Affine>& Rs,
                       const std::vector<G1Affine>& Ss, const std::vector<G1Affine>& Ts, const std::vector<G1Affine>& Us)
    : acc_(acc), crs_(crs.Gs), inst_{&Rs, &Ss, &Ts, &Us}, ell_(crs.Gs.size()) {
  crs_.insert(crs_.end(), crs.Hs.begin(), crs.Hs.end());
  crs_.push_back(AffineOf(crs.H));
  crs_.push_back(AffineOf(crs.Gt));
  crs_.push_back(AffineOf(crs.Gu));
}

void MirrorSink::Check(const Terms& C, const VecExpr& x, const std::vector<BaseSeg>& segs,
                       const std::vector<LooseBase>& loose, common::Rand& rand, const char* what) {
  // rebuild the reference's (x, v) pair: v[i] is the base x[i] multiplies, the point at
  // infinity where the description names none (curdleproof.go:281,285)
  const std::vector<Scalar> xv = x.Materialise();
  std::vector<G1Affine> v(xv.size(), kZeroPoint);
  for (const BaseSeg& sg : segs)
    for (uint32_t j = 0; j < sg.len; j++) {
      const uint32_t slot = sg.first + j;
      if (sg.vec_first + j >= v.size()) throw err(std::string(what) + ": base segment beyond the scalar vector");
      if (sg.set == kSetCrs) {
        v[sg.vec_first + j] = crs_.at(slot);
      } else {
        if (slot >= 4 * ell_) throw err(std::string(what) + ": instance slot out of range");
        v[sg.vec_first + j] = inst_[slot / ell_]->at(slot % ell_);
      }
    }
  // the host mirror needs coordinates: pending points belong to VerifyWhileDecoding's DeviceSink
  for (int32_t pd : C.pending)
    if (pd >= 0) throw std::logic_error(std::string(what) + ": pending point handed to the host accumulator");
  for (const LooseBase& lb : loose) {
    if (lb.pending >= 0) throw std::logic_error(std::string(what) + ": pending point handed to the host accumulator");
    v.at(lb.index) = lb.point;
  }
  std::vector<Fr> xs(xv.size());
  for (size_t i = 0; i < xv.size(); i++) xs[i] = xv[i].v;
  msmaccumulator::Status st;
  if (EagerChecks()) {
    st = acc_.AccumulateCheckXYZZ(EvalTerms(C).p, xs, v, &rand);
  } else {
    std::vector<Fr> cs(C.s.size());
    for (size_t i = 0; i < cs.size(); i++) cs[i] = C.s[i].v;
    st = acc_.AccumulateCheckDeferred(cs, C.p, xs, v, &rand);
  }
  if (!st.ok) throw err(std::string(what) + ": " + st.err);
}

// =========================================================== wire format =====
void Writer::PutPoint(const Point& p) {
  uint8_t b[48];
  p.Compressed(b);
  buf.insert(buf.end(), b, b + 48);
}
void Writer::PutScalar(const Scalar& s) {
  uint8_t b[32];
  s.Bytes(b);
  buf.insert(buf.end(), b, b + 32);
}
void Writer::PutPoints(const std::vector<Point>& v) {
  const uint32_t n = (uint32_t)v.size();  // gnark Encoder: uint32 big-endian slice length
  const uint8_t len[4] = {(uint8_t)(n >> 24), (uint8_t)(n >> 16), (uint8_t)(n >> 8), (uint8_t)n};
  buf.insert(buf.end(), len, len + 4);
  for (const auto& p : v) PutPoint(p);
}
namespace {
// The decoders' three arrays come from (and go back to) a small process-wide pool: a batch
// verification makes and drops one decoder per chunk -- 87 MB of records, points and statuses for
// 1,024 Whisk proofs -- and handing that to the allocator each time cost ~11 ms per batch in page
// faults and munmap (of 55).  At most kPoolBytes are kept.
struct DecoderBuffers {
  std::vector<uint8_t> blob, status;
  std::vector<G1Affine> pts;
  size_t bytes() const { return blob.capacity() + status.capacity() + pts.capacity() * sizeof(G1Affine); }
};
constexpr size_t kPoolBytes = (size_t)256 << 20;
std::mutex g_pool_mu;
std::vector<DecoderBuffers> g_pool;
size_t g_pool_bytes = 0;
}  // namespace

PointDecoder::PointDecoder(bool subgroup_check) : subgroup_(subgroup_check) {
  std::lock_guard<std::mutex> g(g_pool_mu);
  if (g_pool.empty()) return;
  DecoderBuffers b = std::move(g_pool.back());
  g_pool.pop_back();
  g_pool_bytes -= b.bytes();
  blob_ = std::move(b.blob);
  status_ = std::move(b.status);
  pts_ = std::move(b.pts);
  blob_.clear();
  status_.clear();
  pts_.clear();
}

size_t PointDecoder::Add(const uint8_t rec[48]) {
  blob_.insert(blob_.end(), rec, rec + 48);
  return n_++;
}
bool PointDecoder::OnDevice() {
  // CURDLE_HOST_DECODE=1 is an explicit request (A/B measurements of the host decoder), not a
  // fallback: without it a batch of kMinDeviceBatch or more records goes to the GPU and the
  // call fails loudly if there is none.
  return !(knobs::get(knobs::HOST_DECODE) > 0);
}
void PointDecoder::Start() {
  if (started_ || n_ == 0 || !subgroup_ || !OnDevice() || n_ < kMinDeviceBatch) return;
  int rc = curdle_g1_decompress_start(blob_.data(), n_, &ticket_);
  if (rc == CURDLE_EBUSY) {  // enough deferred decodings in flight: Run() takes the one-shot form
    ticket_ = -1;
    return;
  }
  if (rc != CURDLE_OK) {
    ticket_ = -1;
    char buf[256];
    curdle_last_error(buf, sizeof(buf));
    throw alg::MsmError(std::string("decoding points: ") + buf, rc);
  }
  started_ = true;
}
void PointDecoder::Run(bool defer_subgroup) {
  pts_.assign(n_, G1Affine{});
  status_.assign(n_, CURDLE_DECODE_BAD_ENCODING);
  if (n_ == 0) return;
  if (started_) {  // Start() launched it: collect the points; the subgroup test keeps running
    started_ = false;
    int rc = curdle_g1_decompress_points(ticket_, reinterpret_cast<uint64_t*>(pts_.data()), status_.data());
    if (rc != CURDLE_OK) {
      char buf[256];
      curdle_last_error(buf, sizeof(buf));
      (void)curdle_g1_decompress_finish(ticket_, nullptr);
      ticket_ = -1;
      throw alg::MsmError(std::string("decoding points: ") + buf, rc);
    }
    return;
  }
  // below a few dozen points one kernel launch (~1.5 ms: it is a serial chain of a thousand
  // products per point) is slower than the host's ~45 us per point
  if (OnDevice() && n_ >= kMinDeviceBatch) {
    int rc = CURDLE_EBUSY;
    if (subgroup_ && defer_subgroup)
      rc = curdle_g1_decompress_begin(blob_.data(), n_, reinterpret_cast<uint64_t*>(pts_.data()), status_.data(), &ticket_);
    if (rc == CURDLE_EBUSY)  // not deferring, or enough deferred decodings in flight already
      rc = curdle_g1_decompress_batch(blob_.data(), n_, subgroup_ ? 1 : 0, reinterpret_cast<uint64_t*>(pts_.data()),
                                      status_.data());
    if (rc != CURDLE_OK) {
      ticket_ = -1;
      char buf[256];
      curdle_last_error(buf, sizeof(buf));
      throw alg::MsmError(std::string("decoding points: ") + buf, rc);
    }
    return;
  }
  for (size_t i = 0; i < n_; i++) {
    Point pt;
    if (!Point::FromCompressed(&blob_[48 * i], &pt, subgroup_)) continue;
    pts_[i] = pt.Affine();
    status_[i] = g1_affine_is_inf(pts_[i]) ? CURDLE_DECODE_INFINITY : CURDLE_DECODE_OK;
  }
}
bool PointDecoder::Finish() {
  if (ticket_ < 0) return true;  // nothing deferred: Get() already told everything
  const int t = ticket_;
  ticket_ = -1;
  int rc = curdle_g1_decompress_finish(t, status_.data());
  if (rc != CURDLE_OK) {
    char buf[256];
    curdle_last_error(buf, sizeof(buf));
    throw alg::MsmError(std::string("decoding points: ") + buf, rc);
  }
  for (uint8_t st : status_)
    if (st == CURDLE_DECODE_NOT_IN_SUBGROUP) return false;
  return true;
}
PointDecoder::~PointDecoder() {
  if (ticket_ >= 0) (void)curdle_g1_decompress_finish(ticket_, nullptr);  // never leak the workspace slot
  DecoderBuffers b;
  b.blob = std::move(blob_);
  b.status = std::move(status_);
  b.pts = std::move(pts_);
  const size_t sz = b.bytes();
  if (sz < ((size_t)64 << 10)) return;  // one proof's worth: not worth a lock
  std::lock_guard<std::mutex> g(g_pool_mu);
  if (g_pool_bytes + sz <= kPoolBytes) {
    g_pool_bytes += sz;
    g_pool.push_back(std::move(b));
  }
}
bool PointDecoder::Get(size_t index, Point* out) const {
  if (index >= n_ || status_[index] > CURDLE_DECODE_INFINITY) return false;
  *out = status_[index] == CURDLE_DECODE_INFINITY ? Point::Infinity() : Point::FromAffine(pts_[index]);
  return true;
}

bool PointDecoder::GetAffine(size_t index, G1Affine* out) const {
  if (index >= n_ || status_[index] > CURDLE_DECODE_INFINITY) return false;
  *out = status_[index] == CURDLE_DECODE_INFINITY ? kZeroPoint : pts_[index];
  return true;
}

Point Reader::GetPoint(const char* what) {
  if (left < 48) throw err(std::string("decoding ") + what + ": unexpected end of input");
  Point out;
  if (collect) {
    const size_t at = collect->Add(p);
    out = Point::Infinity();
    if (lazy) {
      out.wire = p;
      out.pending = (int32_t)at;
    }
  } else if (decoded) {
    if (!decoded->Get(decoded_pos++, &out)) throw err(std::string("decoding ") + what + ": invalid point");
    if (keep_wire) out.wire = p;
  } else if (!Point::FromCompressed(p, &out, subgroup_check)) {
    throw err(std::string("decoding ") + what + ": invalid point");
  }
  p += 48;
  left -= 48;
  return out;
}
Scalar Reader::GetScalar(const char* what) {
  if (left < 32) throw err(std::string("decoding ") + what + ": unexpected end of input");
  Scalar s;
  if (!Scalar::SetBytesCanonical(p, &s)) throw err(std::string("decoding ") + what + ": scalar not canonical");
  p += 32;
  left -= 32;
  return s;
}
std::vector<Point> Reader::GetPoints(const char* what) {
  if (left < 4) throw err(std::string("decoding ") + what + ": unexpected end of input");
  const uint32_t n = ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3];
  p += 4;
  left -= 4;
  if ((size_t)n * 48 > left) throw err(std::string("decoding ") + what + ": slice longer than the input");
  std::vector<Point> out;
  out.reserve(n);
  for (uint32_t i = 0; i < n; i++) out.push_back(GetPoint(what));
  return out;
}

// ======================================================== groupcommitment =====
GroupCommitment GroupCommitment::New(const Point& crsG, const Point& crsH, const Point& T, const Scalar& r) {
  GroupCommitment g;  // groupcommitment.go:17-31
  g.T_1 = crsG.Mul(r);
  g.T_2 = T + crsH.Mul(r);
  return g;
}
GroupCommitment GroupCommitment::Add(const GroupCommitment& cm) const { return GroupCommitment{T_1 + cm.T_1, T_2 + cm.T_2}; }
GroupCommitment GroupCommitment::Mul(const Scalar& s) const { return GroupCommitment{T_1.Mul(s), T_2.Mul(s)}; }
bool GroupCommitment::Eq(const GroupCommitment& cm) const { return T_1 == cm.T_1 && T_2 == cm.T_2; }
void GroupCommitment::Serialize(Writer& w) const {
  w.PutPoint(T_1);
  w.PutPoint(T_2);
}
void GroupCommitment::FromReader(Reader& r) {
  T_1 = r.GetPoint("T_1");
  T_2 = r.GetPoint("T_2");
}

// ==================================================================== CRS =====
CRS GenerateCRS(size_t size, common::Rand& rand) {
  CRS crs;  // crs.go:20-59; draw order: Gs, Hs, H, Gt, Gu
  rand.GetG1Affines(size, crs.Gs);
  rand.GetG1Affines(N_BLINDERS, crs.Hs);
  G1Affine t;
  rand.GetG1Affine(t);
  crs.H = Point::FromAffine(t);
  rand.GetG1Affine(t);
  crs.Gt = Point::FromAffine(t);
  rand.GetG1Affine(t);
  crs.Gu = Point::FromAffine(t);
  Point gs = Point::Infinity(), hs = Point::Infinity();
  for (const auto& g : crs.Gs) gs = gs + Point::FromAffine(g);  // :41-48, sequential adds
  for (const auto& h : crs.Hs) hs = hs + Point::FromAffine(h);
  crs.Gsum = gs.Affine();
  crs.Hsum = hs.Affine();
  crs.GsumTable = std::make_shared<const alg::FixedBase>(crs.Gsum);
  crs.HsumTable = std::make_shared<const alg::FixedBase>(crs.Hsum);
  crs.device = std::make_shared<DeviceCrs>();
  return crs;
}

ShuffleCommit ShufflePermuteCommit(const std::vector<G1Affine>& crsGs, const std::vector<G1Affine>& crsHs,
                                   const std::vector<G1Affine>& Rs, const std::vector<G1Affine>& Ss,
                                   const std::vector<uint32_t>& perm, const Scalar& k, common::Rand& rand) {
  ShuffleCommit out;  // common/util.go:45-88
  std::vector<G1Affine> Ts(Rs.size()), Us(Ss.size());
  {
    const std::vector<G1Affine> scaled = alg::ScalarMulBatch(Concat(Rs, Ss), {k});  // :55-63, 2 ell by one scalar
    std::copy(scaled.begin(), scaled.begin() + Rs.size(), Ts.begin());
    std::copy(scaled.begin() + Rs.size(), scaled.end(), Us.begin());
  }
  out.Ts = Permute(Ts, perm);
  out.Us = Permute(Us, perm);
  std::vector<Scalar> range(crsGs.size(), Scalar::Zero());
  for (size_t i = 0; i < perm.size(); i++) range[i] = Scalar::FromU64(i);  // :68-71
  std::vector<Scalar> permRange = Permute(range, perm);
  Point M1 = alg::MultiExp(crsGs, permRange);                             // :75
  out.rs_m = GetFrs(rand, N_BLINDERS);                                    // :78
  Point M2 = alg::MultiExp(crsHs, out.rs_m);                              // :82
  out.M = M1 + M2;
  return out;
}

// ====================================================== samescalarargument =====
namespace samescalar {
static const std::string kPoints = "sameexp_points";
static const std::string kAlpha = "sameexp_alpha";

static void AppendStatement(Transcript& tr, const Point& R, const Point& S, const GroupCommitment& T,
                            const GroupCommitment& U, const GroupCommitment& A, const GroupCommitment& B) {
  tr.AppendPoints(kPoints, {R, S, T.T_1, T.T_2, U.T_1, U.T_2, A.T_1, A.T_2, B.T_1, B.T_2});
}

Proof Prove(const Point& Gt, const Point& Gu, const Point& H, const Point& R, const Point& S,
            const GroupCommitment& T, const GroupCommitment& U, const Scalar& k, const Scalar& r_t, const Scalar& r_u,
            Transcript& tr, common::Rand& rand) {
  // samescalarargument.go:34-81
  const Scalar r_a = GetFr(rand), r_b = GetFr(rand), r_k = GetFr(rand);
  Proof p;
  p.A = GroupCommitment::New(Gt, H, R.Mul(r_k), r_a);
  p.B = GroupCommitment::New(Gu, H, S.Mul(r_k), r_b);
  AppendStatement(tr, R, S, T, U, p.A, p.B);
  const Scalar alpha = tr.GetAndAppendChallenge(kAlpha);
  p.Z_k = r_k + k * alpha;
  p.Z_t = r_a + r_t * alpha;
  p.Z_u = r_b + r_u * alpha;
  return p;
}

bool Verify(const Proof& proof, const Point& Gt, const Point& Gu, const Point& H, const Point& R, const Point& S,
            const GroupCommitment& T, const GroupCommitment& U, Transcript& tr, CheckSink* sink, common::Rand* rand,
            size_t ell) {
  // samescalarargument.go:83-100
  AppendStatement(tr, R, S, T, U, proof.A, proof.B);
  const Scalar alpha = tr.GetAndAppendChallenge(kAlpha);
  if (sink && rand && !EagerChecks()) {
    // The reference evaluates both sides of  A + alpha T == Com(Z_k R; Z_t)  and
    // B + alpha U == Com(Z_k S; Z_u)  (ten scalar multiplications) and compares.  Each of
    // the four coordinate equations is "a combination of known points equals an MSM", which
    // is what the accumulator batches: hand them over like every other check, so they ride
    // in the verification's one MSM.  Sound for the same reason the other checks are
    // (a fresh random weight each); rejects surface at the final MSM instead of here.
    // Gt, Gu and H are the CRS's (resident bases, addressed by index); R and S are proof points.
    const CrsIndex ix{ell};
    auto check = [&](const Point& lhs0, const Point& lhs1, std::vector<Scalar> x, std::vector<BaseSeg> segs,
                     std::vector<LooseBase> loose, const char* what) {
      Terms c;
      c.Add(Scalar::One(), lhs0);
      c.Add(alpha, lhs1);
      sink->Check(c, VecExpr::Explicit(std::move(x)), segs, loose, *rand, what);
    };
    check(proof.A.T_1, T.T_1, {proof.Z_t}, {{kSetCrs, ix.Gt(), 1, 0}}, {}, "same scalar check A.T_1");
    check(proof.A.T_2, T.T_2, {proof.Z_k, proof.Z_t}, {{kSetCrs, ix.H(), 1, 1}}, {{0, R}},
          "same scalar check A.T_2");
    check(proof.B.T_1, U.T_1, {proof.Z_u}, {{kSetCrs, ix.Gu(), 1, 0}}, {}, "same scalar check B.T_1");
    check(proof.B.T_2, U.T_2, {proof.Z_k, proof.Z_u}, {{kSetCrs, ix.H(), 1, 1}}, {{0, S}},
          "same scalar check B.T_2");
    return true;
  }
  const GroupCommitment e1 = GroupCommitment::New(Gt, H, R.Mul(proof.Z_k), proof.Z_t);
  const GroupCommitment e2 = GroupCommitment::New(Gu, H, S.Mul(proof.Z_k), proof.Z_u);
  return proof.A.Add(T.Mul(alpha)).Eq(e1) && proof.B.Add(U.Mul(alpha)).Eq(e2);
}

void Proof::Serialize(Writer& w) const {  // :122-140
  A.Serialize(w);
  B.Serialize(w);
  w.PutScalar(Z_k);
  w.PutScalar(Z_t);
  w.PutScalar(Z_u);
}
void Proof::FromReader(Reader& r) {  // :102-120
  A.FromReader(r);
  B.FromReader(r);
  Z_k = r.GetScalar("Z_k");
  Z_t = r.GetScalar("Z_t");
  Z_u = r.GetScalar("Z_u");
}
}  // namespace samescalar

// ==================================================== innerproductargument =====
namespace ipa {
static const std::string kStep1 = "ipa_step1";
static const std::string kAlpha = "ipa_alpha";
static const std::string kBeta = "ipa_beta";
static const std::string kLoop = "ipa_loop";
static const std::string kGamma = "ipa_gamma";

// generateIPABlinders (innerproductargument.go:299-391): r, z with <r,d> + <z,c> = 0 and
// <r,z> = 0; all but the last two z are random, the last two solve the 2x2 system.
static void GenerateBlinders(common::Rand& rand, const std::vector<Scalar>& cs, const std::vector<Scalar>& ds,
                             std::vector<Scalar>* rs_out, std::vector<Scalar>* zs_out) {
  const size_t n = cs.size();
  std::vector<Scalar> rs = GetFrs(rand, n);
  std::vector<Scalar> zs = GetFrs(rand, n - 2);
  const std::vector<Scalar> cs_head(cs.begin(), cs.begin() + (n - 2));
  const std::vector<Scalar> rs_head(rs.begin(), rs.begin() + (n - 2));
  const Scalar omega = alg::InnerProduct(rs, ds) + alg::InnerProduct(zs, cs_head);
  const Scalar delta = alg::InnerProduct(rs_head, zs);
  const Scalar inv_c = cs[n - 2].Inverse();
  const Scalar num = rs[n - 2] * inv_c * omega - delta;
  const Scalar den = rs[n - 2].Neg() * inv_c * cs[n - 1] + rs[n - 1];
  if (den.IsZero()) throw err("last_z_term2 is zero");
  const Scalar last_z = num * den.Inverse();
  const Scalar penultimate_z = inv_c.Neg() * (last_z * cs[n - 1] + omega);
  zs.push_back(penultimate_z);
  zs.push_back(last_z);
  if (!(alg::InnerProduct(rs, ds) + alg::InnerProduct(zs, cs)).IsZero() || !alg::InnerProduct(rs, zs).IsZero())
    throw err("failed to generate IPA blinders: constraints not satisfied");
  *rs_out = std::move(rs);
  *zs_out = std::move(zs);
}

Proof Prove(std::vector<G1Affine> Gs, std::vector<G1Affine> Gs_prime, const Point& Hcrs, const Point& C,
            const Point& D, const Scalar& z, std::vector<Scalar> cs, std::vector<Scalar> ds, Transcript& tr,
            common::Rand& rand, const std::vector<Scalar>* Gs_prime_scale) {
  // innerproductargument.go:42-188
  if (cs.size() != ds.size()) throw err("cs and ds are not the same length");
  if (cs.empty() || (cs.size() & (cs.size() - 1))) throw err("cs and ds are not a power of two");
  if (Gs_prime_scale && (Gs_prime_scale->size() != cs.size() || ProverFoldsBases()))
    throw std::logic_error("ipa prover: scaled second bases need one scale per base and the unfolded form");

  std::vector<Scalar> rs_c, rs_d;
  GenerateBlinders(rand, cs, ds, &rs_c, &rs_d);
  Proof proof;
  {
    std::vector<Scalar> rs_d_scaled;
    if (Gs_prime_scale) {
      rs_d_scaled.resize(rs_d.size());
      for (size_t i = 0; i < rs_d.size(); i++) rs_d_scaled[i] = rs_d[i] * (*Gs_prime_scale)[i];
    }
    std::vector<Point> b = alg::MultiExpBatch({&Gs, &Gs_prime}, {&rs_c, Gs_prime_scale ? &rs_d_scaled : &rs_d});  // :66, :70
    proof.B_c = b[0];
    proof.B_d = b[1];
  }
  tr.AppendPoints(kStep1, {C, D});
  tr.AppendScalar(kStep1, z);
  tr.AppendPoints(kStep1, {proof.B_c, proof.B_d});
  const Scalar alpha = tr.GetAndAppendChallenge(kAlpha);
  const Scalar beta = tr.GetAndAppendChallenge(kBeta);

  size_t n = cs.size();
  for (size_t i = 0; i < n; i++) {  // :83-89
    cs[i] = rs_c[i] + alpha * cs[i];
    ds[i] = rs_d[i] + alpha * ds[i];
  }
  const Point H = Hcrs.Mul(beta);  // :91-92

  // The reference folds the bases in place every round (:155-166): n scalar multiplications,
  // each a 255-step chain -- on this GPU 1.7-2.4 ms per round against 0.45 ms for the round's
  // MSMs.  A folded base is a fixed combination of the ORIGINAL ones,
  //   G^(j)[i] = sum over k = i mod len_j of coef[k] G[k],   coef[k] = product of the gammas of
  //   the rounds in which k sat in the right half,
  // and it is only ever used as an MSM base, so the rounds' MSMs are taken over the original
  // bases with the coefficients multiplied into the scalars (n/2 Fr products per MSM) and no
  // base is ever folded: same L / R points, same proof bytes
  // (test_prover_reproduces_the_committed_proof_bytes).  CURDLE_PROVER_FOLD_BASES=1 keeps the
  // reference's order of operations for A/B.
  const bool fold_bases = ProverFoldsBases();
  const size_t N = n;
  const std::vector<G1Affine> G0 = fold_bases ? std::vector<G1Affine>() : Gs;
  const std::vector<G1Affine> Gp0 = fold_bases ? std::vector<G1Affine>() : Gs_prime;
  std::vector<Scalar> coefG, coefGp;
  if (!fold_bases) {
    coefG.assign(N, Scalar::One());
    if (Gs_prime_scale)
      coefGp = *Gs_prime_scale;
    else
      coefGp.assign(N, Scalar::One());
  }
  const G1Affine H_affine = H.Affine();
  while (n > 1) {  // :101-173
    n /= 2;
    const std::vector<Scalar> c_L(cs.begin(), cs.begin() + n), c_R(cs.begin() + n, cs.begin() + 2 * n);
    const std::vector<Scalar> d_L(ds.begin(), ds.begin() + n), d_R(ds.begin() + n, ds.begin() + 2 * n);
    std::vector<Point> ms;
    std::vector<G1Affine> G_L, G_R, Gp_L, Gp_R;
    if (fold_bases) {
      G_L.assign(Gs.begin(), Gs.begin() + n);
      G_R.assign(Gs.begin() + n, Gs.begin() + 2 * n);
      Gp_L.assign(Gs_prime.begin(), Gs_prime.begin() + n);
      Gp_R.assign(Gs_prime.begin() + n, Gs_prime.begin() + 2 * n);
      // the four MSMs of a round (:109, :121, :126, :138) are independent: one GPU pass
      ms = alg::MultiExpBatch({&G_R, &Gp_L, &G_L, &Gp_R}, {&c_L, &d_R, &c_R, &d_L});
    } else {
      // the original bases whose index falls into the left / right half of the current length
      std::vector<Scalar> s_cL, s_dR, s_cR, s_dL;
      for (size_t k = 0; k < N; k++) {
        const size_t r = k % (2 * n);
        if (r >= n) {
          G_R.push_back(G0[k]);
          Gp_R.push_back(Gp0[k]);
          s_cL.push_back(c_L[r - n] * coefG[k]);
          s_dL.push_back(d_L[r - n] * coefGp[k]);
        } else {
          G_L.push_back(G0[k]);
          Gp_L.push_back(Gp0[k]);
          s_cR.push_back(c_R[r] * coefG[k]);
          s_dR.push_back(d_R[r] * coefGp[k]);
        }
      }
      // ... and H's multiples <c_L, d_R> H and <c_R, d_L> H (:113, :130) ride along as one more
      // pair of their MSM instead of a 255-step scalar multiplication on the host each
      G_R.push_back(H_affine);
      s_cL.push_back(alg::InnerProduct(c_L, d_R));
      G_L.push_back(H_affine);
      s_cR.push_back(alg::InnerProduct(c_R, d_L));
      ms = alg::MultiExpBatch({&G_R, &Gp_L, &G_L, &Gp_R}, {&s_cL, &s_dR, &s_cR, &s_dL});
    }
    const Point L_C = fold_bases ? ms[0] + H.Mul(alg::InnerProduct(c_L, d_R)) : ms[0];
    const Point L_D = ms[1];
    const Point R_C = fold_bases ? ms[2] + H.Mul(alg::InnerProduct(c_R, d_L)) : ms[2];
    const Point R_D = ms[3];
    proof.L_Cs.push_back(L_C);
    proof.L_Ds.push_back(L_D);
    proof.R_Cs.push_back(R_C);
    proof.R_Ds.push_back(R_D);

    tr.AppendPoints(kLoop, {L_C, L_D, R_C, R_D});
    const Scalar gamma = tr.GetAndAppendChallenge(kGamma);
    if (gamma.IsZero()) throw err("ipa gamma challenge is zero");
    const Scalar gamma_inv = gamma.Inverse();

    for (size_t i = 0; i < n; i++) {  // fold, :155-166
      cs[i] = c_L[i] + gamma_inv * c_R[i];
      ds[i] = d_L[i] + gamma * d_R[i];
    }
    if (fold_bases) {
      // G_L + gamma G_R and G'_L + gamma^-1 G'_R: 2n independent scalar multiplications, one batch
      std::vector<G1Affine> pts = Concat(G_R, Gp_R), adds = Concat(G_L, Gp_L);
      std::vector<Scalar> ks(2 * n, gamma);
      std::fill(ks.begin() + n, ks.end(), gamma_inv);
      const std::vector<G1Affine> folded = alg::ScalarMulBatch(pts, ks, &adds);
      std::copy(folded.begin(), folded.begin() + n, Gs.begin());
      std::copy(folded.begin() + n, folded.end(), Gs_prime.begin());
      Gs.resize(n);
      Gs_prime.resize(n);
    } else {
      for (size_t k = 0; k < N; k++)
        if (k % (2 * n) >= n) {
          coefG[k] = coefG[k] * gamma;
          coefGp[k] = coefGp[k] * gamma_inv;
        }
    }
    cs.resize(n);
    ds.resize(n);
  }
  proof.c0 = cs[0];
  proof.d0 = ds[0];
  return proof;
}

bool Verify(const Proof& proof, size_t ell, const Point& Hcrs, const Point& C, const Point& D, const Scalar& z,
            const Scalar& u_q, Transcript& tr, CheckSink& sink, common::Rand& rand) {
  // innerproductargument.go:190-297.  The bases are the CRS's Gs | Hs (n = ell + 4 of them,
  // resident, addressed by index) and H; the vector us of the reference is u_i = u_q^(min(i,
  // ell) + 1) (grandproductargument.go:234-242), handed over as that description.
  tr.AppendPoints(kStep1, {C, D});
  tr.AppendScalar(kStep1, z);
  tr.AppendPoints(kStep1, {proof.B_c, proof.B_d});
  const Scalar alpha = tr.GetAndAppendChallenge(kAlpha);
  Scalar beta = tr.GetAndAppendChallenge(kBeta);

  const size_t n = ell + N_BLINDERS;
  const int m = Log2Exact(n, "ipa n");
  if ((int)proof.L_Cs.size() != m || (int)proof.R_Cs.size() != m || (int)proof.L_Ds.size() != m ||
      (int)proof.R_Ds.size() != m)
    throw err("ipa proof has the wrong number of rounds");

  std::vector<Scalar> gamma;
  for (int i = 0; i < m; i++) {  // :215-219
    tr.AppendPoints(kLoop, {proof.L_Cs[i], proof.L_Ds[i], proof.R_Cs[i], proof.R_Ds[i]});
    gamma.push_back(tr.GetAndAppendChallenge(kGamma));
  }
  const std::vector<Scalar> gamma_inv = alg::BatchInvert(gamma);
  const CrsIndex ix{ell};

  // s_i = prod_{j : bit j of i set} gamma_{m-j-1}, s'_i likewise over the inverses (:223-234):
  // VecExpr::Fold -- n products by doubling on the host mirror, one lane per i on the device.

  // accumulate check 1 (:237-271): AC1 = <gamma, L_C> + B_c + alpha C + (beta alpha^2 z) H + <gamma^-1, R_C>
  Terms AC1;
  AC1.Add(gamma, proof.L_Cs);
  AC1.Add(Scalar::One(), proof.B_c);
  AC1.Add(alpha, C);
  AC1.Add(beta * alpha * alpha * z, Hcrs);
  AC1.Add(gamma_inv, proof.R_Cs);
  beta = beta * proof.d0 * proof.c0;
  {
    // x = (s_0 c0, ..., s_{n-1} c0, beta d0 c0) against Gs | Hs | H
    VecExpr x = VecExpr::Fold(gamma, proof.c0, n + 1);
    x.tail.push_back(beta);
    sink.Check(AC1, x, {{kSetCrs, ix.G(0), (uint32_t)n, 0}, {kSetCrs, ix.H(), 1, (uint32_t)n}}, {}, rand,
               "accumulate check 1");
  }

  // accumulate check 2 (:273-294)
  Terms AC2;  // <gamma, L_D> + B_d + alpha D + <gamma^-1, R_D>
  AC2.Add(gamma, proof.L_Ds);
  AC2.Add(Scalar::One(), proof.B_d);
  AC2.Add(alpha, D);
  AC2.Add(gamma_inv, proof.R_Ds);
  // x_i = s'_i u_i d0 against Gs | Hs
  sink.Check(AC2, VecExpr::FoldPow(gamma_inv, proof.d0, u_q, ell, n), {{kSetCrs, ix.G(0), (uint32_t)n, 0}}, {}, rand,
             "accumulate check 2");
  return true;
}

void Proof::Serialize(Writer& w) const {  // :428-460
  w.PutPoint(B_c);
  w.PutPoint(B_d);
  w.PutPoints(L_Cs);
  w.PutPoints(R_Cs);
  w.PutPoints(L_Ds);
  w.PutPoints(R_Ds);
  w.PutScalar(c0);
  w.PutScalar(d0);
}
void Proof::FromReader(Reader& r) {  // :393-426
  B_c = r.GetPoint("B_c");
  B_d = r.GetPoint("B_d");
  L_Cs = r.GetPoints("L_Cs");
  R_Cs = r.GetPoints("R_Cs");
  L_Ds = r.GetPoints("L_Ds");
  R_Ds = r.GetPoints("R_Ds");
  c0 = r.GetScalar("c0");
  d0 = r.GetScalar("d0");
}
}  // namespace ipa

// ==================================================== grandproductargument =====
namespace gprod {
static const std::string kStep1 = "gprod_step1";
static const std::string kStep2 = "gprod_step2";
static const std::string kAlpha = "gprod_alpha";
static const std::string kBeta = "gprod_beta";

Proof Prove(const std::vector<G1Affine>& Gs, const std::vector<G1Affine>& Hs, const Point& H, const Point& B,
            const Scalar& result, const std::vector<Scalar>& bs, const std::vector<Scalar>& r_bs, Transcript& tr,
            common::Rand& rand) {
  // grandproductargument.go:42-204
  const size_t ell = Gs.size(), nb = r_bs.size();
  tr.AppendPoint(kStep1, B);
  tr.AppendScalar(kStep1, result);
  const Scalar alpha = tr.GetAndAppendChallenge(kAlpha);

  // step 2: partial products c_i = b_0 ... b_{i-1}
  std::vector<Scalar> cs(ell);
  cs[0] = Scalar::One();
  for (size_t i = 1; i < ell; i++) cs[i] = cs[i - 1] * bs[i - 1];
  const std::vector<Scalar> r_cs = GetFrs(rand, nb);
  Proof proof;
  {
    std::vector<Point> cc = alg::MultiExpBatch({&Gs, &Hs}, {&cs, &r_cs});  // :67, :70
    proof.C = cc[0] + cc[1];
  }
  std::vector<Scalar> r_b_plus_alpha(nb);
  for (size_t i = 0; i < nb; i++) r_b_plus_alpha[i] = r_bs[i] + alpha;
  proof.Rp = alg::InnerProduct(r_b_plus_alpha, r_cs);

  tr.AppendPoint(kStep2, proof.C);
  tr.AppendScalar(kStep2, proof.Rp);
  const Scalar beta = tr.GetAndAppendChallenge(kBeta);
  if (beta.IsZero()) throw err("beta is zero");

  // step 3: rescaled bases G'_i = beta^-(i+1) G_i, H'_i = beta^-(ell+1) H_i (:94-103) -- ell + 4
  // scalar multiplications in the reference.  They are only ever MSM bases (D, the self-check,
  // the inner product argument), so the scale goes into those MSMs' scalars instead and the
  // points are never computed (CURDLE_PROVER_FOLD_BASES=1: computed, in one GPU batch).
  const Scalar betaInv = beta.Inverse();
  const bool scale_bases = ProverFoldsBases();
  std::vector<Scalar> ks(ell + Hs.size());
  {
    Scalar bi = betaInv;
    for (size_t i = 0; i < ell; i++) {
      ks[i] = bi;
      bi = bi * betaInv;
    }
    for (size_t i = 0; i < Hs.size(); i++) ks[ell + i] = bi;
  }
  std::vector<G1Affine> Gs_prime = Gs, Hs_prime = Hs;
  if (scale_bases) {
    const std::vector<G1Affine> scaled = alg::ScalarMulBatch(Concat(Gs, Hs), ks);  // ell + 4 in one batch
    std::copy(scaled.begin(), scaled.begin() + ell, Gs_prime.begin());
    std::copy(scaled.begin() + ell, scaled.end(), Hs_prime.begin());
  }

  std::vector<Scalar> betaPowers(ell), ds(ell);
  Scalar bp = Scalar::One();  // beta^i
  for (size_t i = 0; i < ell; i++) {
    betaPowers[i] = bp;
    ds[i] = bs[i] * bp * beta - bp;  // b_i beta^(i+1) - beta^i  (:104-117)
    bp = bp * beta;
  }
  const Scalar betaExpL = bp;                 // beta^ell
  const Scalar betaExpLPlus1 = bp * beta;     // beta^(ell+1)  (:120-121)
  std::vector<Scalar> r_ds(nb), alphaBeta(nb, alpha * betaExpLPlus1);
  for (size_t i = 0; i < nb; i++) r_ds[i] = betaExpLPlus1 * r_b_plus_alpha[i];
  Point D;
  {
    std::vector<Scalar> bp_s = betaPowers, ab_s = alphaBeta;
    if (!scale_bases) {
      for (size_t i = 0; i < ell; i++) bp_s[i] = bp_s[i] * ks[i];
      for (size_t i = 0; i < nb; i++) ab_s[i] = ab_s[i] * ks[ell + i];
    }
    std::vector<Point> dd = alg::MultiExpBatch({&Gs_prime, &Hs_prime}, {&bp_s, &ab_s});  // :132, :135
    D = B - dd[0] + dd[1];
  }

  // step 4: the inner-product instance (:141-176)
  const std::vector<G1Affine> G_full = Concat(Gs, Hs), Gp_full = Concat(Gs_prime, Hs_prime);
  const Scalar z = proof.Rp * betaExpLPlus1 + result * betaExpL - Scalar::One();
  const std::vector<Scalar> cs_full = Concat(cs, r_cs), ds_full = Concat(ds, r_ds);
  if (alg::InnerProduct(cs_full, ds_full) != z) throw err("IPA(C, D) != z");
  {
    std::vector<Scalar> ds_s = ds_full;
    if (!scale_bases)
      for (size_t i = 0; i < ds_s.size(); i++) ds_s[i] = ds_s[i] * ks[i];
    std::vector<Point> chk = alg::MultiExpBatch({&G_full, &Gp_full}, {&cs_full, &ds_s});  // :165, :172 self-checks
    if (!(chk[0] == proof.C)) throw err("msm(G, c) != C");
    if (!(chk[1] == D)) throw err("msm(G', d) != D");
  }
  proof.IPAProof = ipa::Prove(G_full, Gp_full, H, proof.C, D, z, cs_full, ds_full, tr, rand, scale_bases ? nullptr : &ks);
  return proof;
}

bool Verify(const Proof& proof, const CRS& crs, const Point& B, const Scalar& result, Transcript& tr, CheckSink& sink,
            common::Rand& rand) {
  // grandproductargument.go:206-286
  const size_t ell = crs.Gs.size();
  tr.AppendPoint(kStep1, B);
  tr.AppendScalar(kStep1, result);
  const Scalar alpha = tr.GetAndAppendChallenge(kAlpha);
  tr.AppendPoint(kStep2, proof.C);
  tr.AppendScalar(kStep2, proof.Rp);
  const Scalar beta = tr.GetAndAppendChallenge(kBeta);
  if (beta.IsZero()) throw err("beta is zero");

  // us_i = beta^-(i+1) for i < ell, beta^-(ell+1) for the blinder positions (:234-242): the
  // inner-product verifier takes the description (u_q = beta^-1), not the vector.
  const Scalar betaInv = beta.Inverse();
  // D = B - beta^-1 Gsum + alpha Hsum (:243-246): Gsum and Hsum never change, so the two scalar
  // multiplications go through the CRS's fixed-base tables (32 mixed additions each)
  const Point gs = crs.GsumTable ? crs.GsumTable->Mul(betaInv) : Point::FromAffine(crs.Gsum).Mul(betaInv);
  const Point hs = crs.HsumTable ? crs.HsumTable->Mul(alpha) : Point::FromAffine(crs.Hsum).Mul(alpha);
  // normalised once: D is hashed into the transcript AND rides in the accumulator as a base
  const Point D = Point::FromAffine((Decoded(B) - gs + hs).Affine());

  const Scalar betaExpL = beta.Pow(ell);
  const Scalar z = result * betaExpL + proof.Rp * (betaExpL * beta) - Scalar::One();  // :253-260
  return ipa::Verify(proof.IPAProof, ell, crs.H, proof.C, D, z, betaInv, tr, sink, rand);
}

void Proof::Serialize(Writer& w) const {  // :304-318
  w.PutPoint(C);
  w.PutScalar(Rp);
  IPAProof.Serialize(w);
}
void Proof::FromReader(Reader& r) {  // :288-302
  C = r.GetPoint("C");
  Rp = r.GetScalar("Rp");
  IPAProof.FromReader(r);
}
}  // namespace gprod

// ================================================= samepermutationargument =====
namespace sameperm {
static const std::string kStep1 = "same_perm_step1";
static const std::string kAlpha = "same_perm_alpha";
static const std::string kBeta = "same_perm_beta";

Proof Prove(const std::vector<G1Affine>& Gs, const std::vector<G1Affine>& Hs, const Point& H, const Point& A,
            const Point& M, const std::vector<Scalar>& as, const std::vector<uint32_t>& permutation,
            const std::vector<Scalar>& rs_a, const std::vector<Scalar>& rs_m, Transcript& tr, common::Rand& rand) {
  // samepermutationargument.go:32-101
  tr.AppendPoints(kStep1, {A, M});
  tr.AppendScalars(kStep1, as);
  const Scalar alpha = tr.GetAndAppendChallenge(kAlpha);
  const Scalar beta = tr.GetAndAppendChallenge(kBeta);

  const std::vector<Scalar> permutedAs = Permute(as, permutation);
  std::vector<Scalar> bs(as.size());
  Scalar p = Scalar::One();
  for (size_t i = 0; i < as.size(); i++) {
    bs[i] = alpha * Scalar::FromU64(permutation[i]) + permutedAs[i] + beta;
    p = p * bs[i];
  }
  const std::vector<Scalar> betas(Gs.size(), beta);
  Proof proof;
  proof.B = A + M.Mul(alpha) + alg::MultiExp(Gs, betas);  // :67, all-equal scalars
  std::vector<Scalar> rs_b(rs_a.size());
  for (size_t i = 0; i < rs_a.size(); i++) rs_b[i] = alpha * rs_m[i] + rs_a[i];
  proof.gpaProof = gprod::Prove(Gs, Hs, H, proof.B, p, bs, rs_b, tr, rand);
  return proof;
}

bool Verify(const Proof& proof, const CRS& crs, const Point& A, const Point& M, const std::vector<Scalar>& as,
            Transcript& tr, CheckSink& sink, common::Rand& rand) {
  // samepermutationargument.go:103-164
  const size_t ell = crs.Gs.size();
  tr.AppendPoints(kStep1, {A, M});
  tr.AppendScalars(kStep1, as);
  const Scalar alpha = tr.GetAndAppendChallenge(kAlpha);
  const Scalar beta = tr.GetAndAppendChallenge(kBeta);

  Scalar p = Scalar::One();
  for (size_t i = 0; i < as.size(); i++) p = p * (Scalar::FromU64(i) * alpha + beta + as[i]);  // :125-130
  Terms C;  // proof.B - A - alpha M, :136-139
  C.Add(Scalar::One(), proof.B);
  C.Add(-Scalar::One(), A);
  C.Add(-alpha, M);
  const CrsIndex ix{ell};
  sink.Check(C, VecExpr::Const(beta, ell), {{kSetCrs, ix.G(0), (uint32_t)ell, 0}}, {}, rand,
             "failed to accumulate check");  // :140, betas against Gs
  return gprod::Verify(proof.gpaProof, crs, proof.B, p, tr, sink, rand);
}

void Proof::Serialize(Writer& w) const {  // :181-192
  w.PutPoint(B);
  gpaProof.Serialize(w);
}
void Proof::FromReader(Reader& r) {  // :166-179
  B = r.GetPoint("B");
  gpaProof.FromReader(r);
}
}  // namespace sameperm

// ================================================= samemultiscalarargument =====
namespace samemsm {
static const std::string kStep1 = "same_msm_step1";
static const std::string kAlpha = "same_msm_alpha";
static const std::string kLoop = "same_msm_loop";
static const std::string kGamma = "same_msm_gamma";

static void AppendStatement(Transcript& tr, const Point& A, const Point& Z_t, const Point& Z_u,
                            const std::vector<G1Affine>& T, const std::vector<G1Affine>& U, const Point& B_a,
                            const Point& B_t, const Point& B_u, const std::vector<uint8_t>* Tb = nullptr,
                            const std::vector<uint8_t>* Ub = nullptr) {
  tr.AppendPoints(kStep1, {A, Z_t, Z_u});
  if (Tb && Ub && Tb->size() == 48 * T.size() && Ub->size() == 48 * U.size()) {  // encodings kept by the caller
    tr.AppendCompressed(kStep1, Tb->data(), T.size());
    tr.AppendCompressed(kStep1, Ub->data(), U.size());
  } else {
    tr.AppendPointsAffine(kStep1, T);
    tr.AppendPointsAffine(kStep1, U);
  }
  tr.AppendPoints(kStep1, {B_a, B_t, B_u});
}

Proof Prove(std::vector<G1Affine> G, const Point& A, const Point& Z_t, const Point& Z_u, std::vector<G1Affine> T,
            std::vector<G1Affine> U, std::vector<Scalar> x, Transcript& tr, common::Rand& rand) {
  // samemultiscalarargument.go:37-157
  size_t n = x.size();
  Log2Exact(n, "same msm n");
  const std::vector<Scalar> r = GetFrs(rand, n);
  Proof proof;
  {
    std::vector<Point> b = alg::MultiExpShared({&G, &T, &U}, r);  // :64-70, one scalar vector, three base sets: recoded once
    proof.B_a = b[0];
    proof.B_t = b[1];
    proof.B_u = b[2];
  }
  AppendStatement(tr, A, Z_t, Z_u, T, U, proof.B_a, proof.B_t, proof.B_u);
  const Scalar alpha = tr.GetAndAppendChallenge(kAlpha);
  for (size_t i = 0; i < n; i++) x[i] = r[i] + x[i] * alpha;  // :78-81

  // As in the inner product argument's prover: the bases are never folded (:128-135 of the
  // reference); a round's MSMs run over the original T, U, G with the fold coefficients
  // multiplied into the scalars.  One coefficient vector: the three sets fold by the same gamma.
  const bool fold_bases = ProverFoldsBases();
  const size_t N = n;
  const std::vector<G1Affine> G0 = fold_bases ? std::vector<G1Affine>() : G;
  const std::vector<G1Affine> T0 = fold_bases ? std::vector<G1Affine>() : T;
  const std::vector<G1Affine> U0 = fold_bases ? std::vector<G1Affine>() : U;
  std::vector<Scalar> coef;
  if (!fold_bases) coef.assign(N, Scalar::One());
  while (n > 1) {  // :83-141
    n /= 2;
    const std::vector<Scalar> x_L(x.begin(), x.begin() + n), x_R(x.begin() + n, x.begin() + 2 * n);
    std::vector<G1Affine> T_L, T_R, U_L, U_R, G_L, G_R;
    std::vector<Point> ms;
    if (fold_bases) {
      T_L.assign(T.begin(), T.begin() + n);
      T_R.assign(T.begin() + n, T.begin() + 2 * n);
      U_L.assign(U.begin(), U.begin() + n);
      U_R.assign(U.begin() + n, U.begin() + 2 * n);
      G_L.assign(G.begin(), G.begin() + n);
      G_R.assign(G.begin() + n, G.begin() + 2 * n);
      // six MSMs per round (:94-109): x_L against the right halves, x_R against the left halves
      ms = alg::MultiExpBatch({&G_R, &T_R, &U_R, &G_L, &T_L, &U_L}, {&x_L, &x_L, &x_L, &x_R, &x_R, &x_R});
    } else {
      std::vector<Scalar> s_L, s_R;
      for (size_t k = 0; k < N; k++) {
        const size_t r = k % (2 * n);
        if (r >= n) {
          G_R.push_back(G0[k]);
          T_R.push_back(T0[k]);
          U_R.push_back(U0[k]);
          s_L.push_back(x_L[r - n] * coef[k]);
        } else {
          G_L.push_back(G0[k]);
          T_L.push_back(T0[k]);
          U_L.push_back(U0[k]);
          s_R.push_back(x_R[r] * coef[k]);
        }
      }
      ms = alg::MultiExpBatch({&G_R, &T_R, &U_R, &G_L, &T_L, &U_L}, {&s_L, &s_L, &s_L, &s_R, &s_R, &s_R});
    }
    proof.L_A.push_back(ms[0]);
    proof.L_T.push_back(ms[1]);
    proof.L_U.push_back(ms[2]);
    proof.R_A.push_back(ms[3]);
    proof.R_T.push_back(ms[4]);
    proof.R_U.push_back(ms[5]);
    tr.AppendPoints(kLoop, {ms[0], ms[1], ms[2], ms[3], ms[4], ms[5]});
    const Scalar gamma = tr.GetAndAppendChallenge(kGamma);
    if (gamma.IsZero()) throw err("gamma is zero");
    const Scalar gamma_inv = gamma.Inverse();
    for (size_t i = 0; i < n; i++) x[i] = x_L[i] + gamma_inv * x_R[i];  // fold vectors and bases, :128-135
    if (fold_bases) {
      // T_L + gamma T_R, U_L + gamma U_R, G_L + gamma G_R: 3n scalar multiplications by one scalar
      const std::vector<G1Affine> pts = Concat(Concat(T_R, U_R), G_R), adds = Concat(Concat(T_L, U_L), G_L);
      const std::vector<G1Affine> folded = alg::ScalarMulBatch(pts, {gamma}, &adds);
      std::copy(folded.begin(), folded.begin() + n, T.begin());
      std::copy(folded.begin() + n, folded.begin() + 2 * n, U.begin());
      std::copy(folded.begin() + 2 * n, folded.end(), G.begin());
      T.resize(n);
      U.resize(n);
      G.resize(n);
    } else {
      for (size_t k = 0; k < N; k++)
        if (k % (2 * n) >= n) coef[k] = coef[k] * gamma;
    }
    x.resize(n);
  }
  proof.x = x[0];
  return proof;
}

bool Verify(const Proof& proof, size_t ell, const Point& A, const Point& Z_t, const Point& Z_u,
            const std::vector<G1Affine>& T, const std::vector<G1Affine>& U, Transcript& tr, CheckSink& sink,
            common::Rand& rand, const std::vector<uint8_t>* Tbytes, const std::vector<uint8_t>* Ubytes) {
  // samemultiscalarargument.go:159-236 and unfoldedScalars :239-280.  T and U (the padded
  // vectors T', U' of curdleproof.go:271-285) are passed for the transcript; as bases they are
  // the instance's Ts / Us (resident, by index) followed by 0 | 0 | H | 0 and 0 | 0 | 0 | H, and
  // G is the CRS's Gs | Hs[:2] | Gt | Gu.
  const size_t n = T.size();
  AppendStatement(tr, A, Z_t, Z_u, T, U, proof.B_a, proof.B_t, proof.B_u, Tbytes, Ubytes);
  const Scalar alpha = tr.GetAndAppendChallenge(kAlpha);

  const size_t lg_n = proof.L_A.size();
  if (lg_n >= 32) throw err("recursive steps greater than expected");
  if (n != ((size_t)1 << lg_n)) throw err("must by log2(L_a)");
  if (n != ell + N_BLINDERS) throw err("same msm vectors do not match the CRS");
  if (proof.L_T.size() != lg_n || proof.L_U.size() != lg_n || proof.R_A.size() != lg_n || proof.R_T.size() != lg_n ||
      proof.R_U.size() != lg_n)
    throw err("same msm proof vectors differ in length");
  std::vector<Scalar> gamma;
  for (size_t i = 0; i < lg_n; i++) {
    tr.AppendPoints(kLoop, {proof.L_A[i], proof.L_T[i], proof.L_U[i], proof.R_A[i], proof.R_T[i], proof.R_U[i]});
    gamma.push_back(tr.GetAndAppendChallenge(kGamma));
  }
  // unfoldedScalars (:267-277): x * s_i with s_i = prod_{b : bit b of i set} gamma_{lg_n-b-1}
  const VecExpr xs = VecExpr::Fold(gamma, proof.x, n);  // :184-187
  const std::vector<Scalar> gamma_inv = alg::BatchInvert(gamma);

  // the three check points (:196-231): B + alpha Z + <gamma, L> + <gamma^-1, R>
  auto check_point = [&](const Point& B, const Point& Z, const std::vector<Point>& L, const std::vector<Point>& R) {
    Terms t;
    t.Add(Scalar::One(), B);
    t.Add(alpha, Z);
    t.Add(gamma, L);
    t.Add(gamma_inv, R);
    return t;
  };
  const CrsIndex ix{ell};
  const InstIndex in{ell};
  const uint32_t e = (uint32_t)ell;
  const Terms pA = check_point(proof.B_a, A, proof.L_A, proof.R_A);
  sink.Check(pA, xs, {{kSetCrs, ix.G(0), e + 2, 0}, {kSetCrs, ix.Gt(), 1, e + 2}, {kSetCrs, ix.Gu(), 1, e + 3}}, {}, rand,
             "accumulating msm 1");  // :206
  const Terms pT = check_point(proof.B_t, Z_t, proof.L_T, proof.R_T);
  sink.Check(pT, xs, {{kSetInst, in.T(0), e, 0}, {kSetCrs, ix.H(), 1, e + 2}}, {}, rand, "accumulating msm 2");  // :218
  const Terms pU = check_point(proof.B_u, Z_u, proof.L_U, proof.R_U);
  sink.Check(pU, xs, {{kSetInst, in.U(0), e, 0}, {kSetCrs, ix.H(), 1, e + 3}}, {}, rand, "accumulating msm 3");  // :231
  return true;
}

void Proof::Serialize(Writer& w) const {  // :325-365
  w.PutPoint(B_a);
  w.PutPoint(B_t);
  w.PutPoint(B_u);
  w.PutPoints(L_A);
  w.PutPoints(L_T);
  w.PutPoints(L_U);
  w.PutPoints(R_A);
  w.PutPoints(R_T);
  w.PutPoints(R_U);
  w.PutScalar(x);
}
void Proof::FromReader(Reader& r) {  // :282-323
  B_a = r.GetPoint("B_a");
  B_t = r.GetPoint("B_t");
  B_u = r.GetPoint("B_u");
  L_A = r.GetPoints("L_A");
  L_T = r.GetPoints("L_T");
  L_U = r.GetPoints("L_U");
  R_A = r.GetPoints("R_A");
  R_T = r.GetPoints("R_T");
  R_U = r.GetPoints("R_U");
  x = r.GetScalar("x");
}
}  // namespace samemsm

// ============================================================= curdleproof =====
static const std::string kTranscript = "curdleproofs";
static const std::string kStep1 = "curdleproofs_step1";
static const std::string kVecA = "curdleproofs_vec_a";

static void AppendInstance(Transcript& tr, const std::vector<G1Affine>& Rs, const std::vector<G1Affine>& Ss,
                           const std::vector<G1Affine>& Ts, const std::vector<G1Affine>& Us, const Point& M) {
  tr.AppendPointsAffine(kStep1, Rs);
  tr.AppendPointsAffine(kStep1, Ss);
  tr.AppendPointsAffine(kStep1, Ts);
  tr.AppendPointsAffine(kStep1, Us);
  tr.AppendPoint(kStep1, M);
}

// The bases of the same-multiscalar argument (curdleproof.go:150-164 / :271-285):
// G = Gs | Hs[:2] | Gt | Gu,  T' = Ts | 0 | 0 | H | 0,  U' = Us | 0 | 0 | 0 | H.
static void MultiscalarBases(const CRS& crs, const std::vector<G1Affine>& Ts, const std::vector<G1Affine>& Us,
                             std::vector<G1Affine>* G, std::vector<G1Affine>* Tp, std::vector<G1Affine>* Up) {
  *G = crs.Gs;
  G->insert(G->end(), crs.Hs.begin(), crs.Hs.begin() + (N_BLINDERS - 2));
  G->push_back(AffineOf(crs.Gt));
  G->push_back(AffineOf(crs.Gu));
  const G1Affine Haff = AffineOf(crs.H);
  *Tp = Ts;
  Tp->insert(Tp->end(), {kZeroPoint, kZeroPoint, Haff, kZeroPoint});
  *Up = Us;
  Up->insert(Up->end(), {kZeroPoint, kZeroPoint, kZeroPoint, Haff});
}

Proof Prove(const CRS& crs, const std::vector<G1Affine>& Rs, const std::vector<G1Affine>& Ss,
            const std::vector<G1Affine>& Ts, const std::vector<G1Affine>& Us, const Point& M,
            const std::vector<uint32_t>& perm, const Scalar& k, const std::vector<Scalar>& rs_m,
            common::Rand& rand) {
  // curdleproof.go:38-197
  Transcript tr(kTranscript);
  AppendInstance(tr, Rs, Ss, Ts, Us, M);
  const std::vector<Scalar> as = tr.GetAndAppendChallenges(kVecA, Rs.size());

  // step 2 (:66-103)
  const std::vector<Scalar> rs_a = GetFrs(rand, N_BLINDERS - 2);
  std::vector<Scalar> rs_a_prime(rs_a);
  rs_a_prime.push_back(Scalar::Zero());
  rs_a_prime.push_back(Scalar::Zero());
  const std::vector<Scalar> perm_as = Permute(as, perm);
  Proof proof;
  {
    std::vector<Point> aa = alg::MultiExpBatch({&crs.Gs, &crs.Hs}, {&perm_as, &rs_a_prime});  // :73, :76
    proof.A = aa[0] + aa[1];
  }
  proof.proofSamePermutation =
      sameperm::Prove(crs.Gs, crs.Hs, crs.H, proof.A, M, as, perm, rs_a_prime, rs_m, tr, rand);

  // step 3 (:105-146)
  const Scalar r_t = GetFr(rand), r_u = GetFr(rand);
  {
    std::vector<Point> rs = alg::MultiExpShared({&Rs, &Ss}, as);  // :110, :114, shared scalars: recoded once
    proof.R = rs[0];
    proof.S = rs[1];
  }
  proof.T = GroupCommitment::New(crs.Gt, crs.H, proof.R.Mul(k), r_t);
  proof.U = GroupCommitment::New(crs.Gu, crs.H, proof.S.Mul(k), r_u);
  proof.proofSameScalar =
      samescalar::Prove(crs.Gt, crs.Gu, crs.H, proof.R, proof.S, proof.T, proof.U, k, r_t, r_u, tr, rand);

  // step 4 (:148-185)
  const Point A_prime = proof.A + proof.T.T_1 + proof.U.T_1;
  std::vector<G1Affine> G, Tp, Up;
  MultiscalarBases(crs, Ts, Us, &G, &Tp, &Up);
  std::vector<Scalar> x(perm_as);
  x.insert(x.end(), rs_a.begin(), rs_a.end());
  x.push_back(r_t);
  x.push_back(r_u);
  proof.proofSameMultiscalar = samemsm::Prove(G, A_prime, proof.T.T_2, proof.U.T_2, Tp, Up, x, tr, rand);
  return proof;
}

// The body of curdleproof.Verify up to, but not including, the accumulator's final MSM:
// false = a direct (non-accumulated) check already failed.  Every AccumulateCheck of the
// sub-arguments goes to `sink`.
VerifyPrelude::VerifyPrelude() : tr(kTranscript) {}

void StartVerify(VerifyPrelude& pre, size_t ell, const uint8_t* Rb, const uint8_t* Sb, const uint8_t* Tb, const uint8_t* Ub,
                 const uint8_t Mb[48]) {
  // curdleproof.go:217-224: Rs, Ss, Ts, Us, M under "curdleproofs_step1", then the vector a
  pre.tr.AppendCompressed(kStep1, Rb, ell);
  pre.tr.AppendCompressed(kStep1, Sb, ell);
  pre.tr.AppendCompressed(kStep1, Tb, ell);
  pre.tr.AppendCompressed(kStep1, Ub, ell);
  pre.tr.AppendCompressed(kStep1, Mb, 1);
  pre.as = pre.tr.GetAndAppendChallenges(kVecA, ell);
  pre.Tb.assign(Tb, Tb + 48 * ell);
  pre.Ub.assign(Ub, Ub + 48 * ell);
}

void StartVerify(VerifyPrelude& pre, const std::vector<G1Affine>& Rs, const std::vector<G1Affine>& Ss,
                 const std::vector<G1Affine>& Ts, const std::vector<G1Affine>& Us, const Point& M) {
  const size_t ell = Rs.size();
  if (Ss.size() != ell || Ts.size() != ell || Us.size() != ell) throw err("instance vectors differ in length");
  std::vector<uint8_t> b(48 * (4 * ell + 1));
  const std::vector<G1Affine>* v[4] = {&Rs, &Ss, &Ts, &Us};
  for (int k = 0; k < 4; k++) alg::CompressAffineBatch(v[k]->data(), ell, &b[48 * k * ell]);
  M.Compressed(&b[48 * 4 * ell]);
  StartVerify(pre, ell, &b[0], &b[48 * ell], &b[96 * ell], &b[144 * ell], &b[192 * ell]);
}

// curdleproof.go:225-311 -- everything after the prelude.  Ts / Us are only needed when the
// prelude did not keep their encodings (it always does); the instance as BASES is addressed by
// index, so a caller that does not have the decoded instance yet passes null.
static bool VerifyBody(const Proof& proof, const CRS& crs, const std::vector<G1Affine>* Ts, const std::vector<G1Affine>* Us,
                       const Point& M, common::Rand& rand, CheckSink& sink, VerifyPrelude& started) {
  const size_t ell = crs.Gs.size();
  if (started.as.size() != ell) throw err("verification prelude does not match the CRS");
  Transcript& tr = started.tr;
  const std::vector<Scalar>& as = started.as;

  if (!sameperm::Verify(proof.proofSamePermutation, crs, proof.A, M, as, tr, sink, rand)) return false;
  if (!samescalar::Verify(proof.proofSameScalar, crs.Gt, crs.Gu, crs.H, proof.R, proof.S, proof.T, proof.U, tr, &sink,
                          &rand, ell))
    return false;

  const Point Aprime =
      Point::FromAffine((Decoded(proof.A) + Decoded(proof.T.T_1) + Decoded(proof.U.T_1)).Affine());  // hashed and accumulated
  // T' = Ts | 0 | 0 | H | 0 and U' = Us | 0 | 0 | 0 | H for the transcript (curdleproof.go:271-285)
  const G1Affine Haff = AffineOf(crs.H);
  std::vector<G1Affine> Tp, Up;
  if (Ts && Us) {
    Tp = *Ts;
    Up = *Us;
  } else {  // absorbed from the encodings below; only the length matters
    if (started.Tb.size() != 48 * ell || started.Ub.size() != 48 * ell) throw err("verification prelude without the instance's encodings");
    Tp.assign(ell, kZeroPoint);
    Up.assign(ell, kZeroPoint);
  }
  Tp.insert(Tp.end(), {kZeroPoint, kZeroPoint, Haff, kZeroPoint});
  Up.insert(Up.end(), {kZeroPoint, kZeroPoint, kZeroPoint, Haff});
  // ... and their encodings: the instance's, kept by the prelude, plus the four padding points
  std::vector<uint8_t> Tpb(started.Tb), Upb(started.Ub);
  {
    uint8_t zero[48], h[48];
    alg::CompressAffine(kZeroPoint, zero);
    alg::CompressAffine(Haff, h);
    const uint8_t* tpad[4] = {zero, zero, h, zero};
    const uint8_t* upad[4] = {zero, zero, zero, h};
    for (int k = 0; k < 4; k++) {
      Tpb.insert(Tpb.end(), tpad[k], tpad[k] + 48);
      Upb.insert(Upb.end(), upad[k], upad[k] + 48);
    }
  }
  if (!samemsm::Verify(proof.proofSameMultiscalar, ell, Aprime, proof.T.T_2, proof.U.T_2, Tp, Up, tr, sink, rand, &Tpb,
                       &Upb))
    return false;

  Terms R, S;
  R.Add(Scalar::One(), proof.R);
  S.Add(Scalar::One(), proof.S);
  const InstIndex in{ell};
  VecExpr xa = VecExpr::Explicit(as);
  sink.Check(R, xa, {{kSetInst, in.R(0), (uint32_t)ell, 0}}, {}, rand, "msm accumulator check R, as, Rs");  // :306
  sink.Check(S, xa, {{kSetInst, in.S(0), (uint32_t)ell, 0}}, {}, rand, "msm accumulator check S, as, Ss");  // :309
  return true;
}

bool VerifyWithSink(const Proof& proof, const CRS& crs, const std::vector<G1Affine>& Rs, const std::vector<G1Affine>& Ss,
                    const std::vector<G1Affine>& Ts, const std::vector<G1Affine>& Us, const Point& M, common::Rand& rand,
                    CheckSink& sink, VerifyPrelude* started) {
  // curdleproof.go:199-311
  const size_t ell = crs.Gs.size();
  if (Rs.size() != ell || Ss.size() != ell || Ts.size() != ell || Us.size() != ell)
    throw err("instance vectors do not match the CRS");
  if (Ts.empty() || g1_affine_is_inf(Ts[0])) throw err("randomizer is zero");  // :213-215
  VerifyPrelude own;
  if (!started) {
    StartVerify(own, Rs, Ss, Ts, Us, M);
    started = &own;
  }
  return VerifyBody(proof, crs, &Ts, &Us, M, rand, sink, *started);
}

bool CanVerifyWhileDecoding() { return !EagerChecks() && DeviceAccumulatorEnabled(); }

bool VerifyWhileDecoding(VerifyPrelude& pre, const Proof& proof, const CRS& crs, const Point& M, PointDecoder& dec,
                         const std::function<void(DecodedInstance&)>& after_decode, common::Rand& rand) {
  if (!CanVerifyWhileDecoding()) throw std::logic_error("VerifyWhileDecoding needs the device accumulator and deferred checks");
  const size_t ell = crs.Gs.size();
  const bool trace = knobs::get(knobs::VERIFY_TRACE) > 0;  // where the time goes (stderr)
  const auto t0 = std::chrono::steady_clock::now();
  DeviceSink sink(crs);  // records; the accumulation itself starts once the instance is there
  // the whole host part, from the wire bytes, while the GPU takes the square roots
  bool body_ok = false;
  std::string body_error;
  try {
    body_ok = VerifyBody(proof, crs, nullptr, nullptr, M, rand, sink, pre);
  } catch (const alg::MsmError&) {
    throw;
  } catch (const std::runtime_error& e) {
    body_error = e.what();
  }
  const auto t1 = std::chrono::steady_clock::now();
  dec.Run(/*defer_subgroup=*/true);  // the points (the subgroup test may still be running)
  const auto t2 = std::chrono::steady_clock::now();
  DecodedInstance inst;
  after_decode(inst);                // decoding errors first, in the caller's (the reference's) order
  if (inst.Rs.size() != ell || inst.Ss.size() != ell || inst.Ts.size() != ell || inst.Us.size() != ell)
    throw err("instance vectors do not match the CRS");
  if (inst.Ts.empty() || g1_affine_is_inf(inst.Ts[0])) throw err("randomizer is zero");  // :213-215
  if (!body_error.empty()) throw err(body_error);
  if (!body_ok) return false;
  sink.Resolve(dec);                 // the proof points, now that they are known, as the check points' bases
  sink.Begin(inst.Rs, inst.Ss, inst.Ts, inst.Us);
  const auto t3 = std::chrono::steady_clock::now();
  const bool ok = sink.Verify();     // :313, the accumulator's one MSM
  if (trace) {
    const auto t4 = std::chrono::steady_clock::now();
    auto us = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
    fprintf(stderr, "[verify while decoding] host transcript+algebra %.0f us, wait for the points %.0f us, instance+begin %.0f us, device run %.0f us\n",
            us(t0, t1), us(t1, t2), us(t2, t3), us(t3, t4));
  }
  return ok;
}

bool VerifyInto(const Proof& proof, const CRS& crs, const std::vector<G1Affine>& Rs, const std::vector<G1Affine>& Ss,
                const std::vector<G1Affine>& Ts, const std::vector<G1Affine>& Us, const Point& M, common::Rand& rand,
                MsmAccumulator& acc) {
  MirrorSink sink(acc, crs, Rs, Ss, Ts, Us);
  return VerifyWithSink(proof, crs, Rs, Ss, Ts, Us, M, rand, sink);
}

bool Verify(const Proof& proof, const CRS& crs, const std::vector<G1Affine>& Rs, const std::vector<G1Affine>& Ss,
            const std::vector<G1Affine>& Ts, const std::vector<G1Affine>& Us, const Point& M, common::Rand& rand) {
  VerifyPrelude pre;
  StartVerify(pre, Rs, Ss, Ts, Us, M);
  return VerifyStarted(pre, proof, crs, Rs, Ss, Ts, Us, M, rand);
}

bool VerifyStarted(VerifyPrelude& pre, const Proof& proof, const CRS& crs, const std::vector<G1Affine>& Rs,
                   const std::vector<G1Affine>& Ss, const std::vector<G1Affine>& Ts, const std::vector<G1Affine>& Us,
                   const Point& M, common::Rand& rand) {
  // curdleproof.go:199-318: every sub-argument folds its checks into one accumulator, whose
  // single MSM (:313) decides.  By default that accumulator lives on the GPU
  // (device_accumulator.h: CRS bases resident, scalars built by index in an Fr kernel, MSM
  // fed directly); the host mirror does the same job in eager mode and under
  // CURDLE_DEVICE_ACC=0.
  if (!EagerChecks() && DeviceAccumulatorEnabled()) {
    // CURDLE_VERIFY_TRACE=1: where one verification's time goes (stderr), for the host-share
    // figure of DESIGN.md: starting the accumulation (instance upload), the host's transcript
    // and challenge algebra, the device part (scalars kernel + MSM + wait)
    const bool trace = knobs::get(knobs::VERIFY_TRACE) > 0;
    const auto t0 = std::chrono::steady_clock::now();
    // The accumulation takes its workspace slot NOW -- the instance points are uploaded and
    // converted while the host hashes -- unless slots are scarce: with more verifying threads
    // than slots, a slot held idle through the host's transcript phase (most of a verification)
    // caps the throughput and stalls plain MSM callers, so the accumulation then starts after the
    // host algebra, as VerifyWhileDecoding does.
    DeviceSink sink(crs);
    const bool early = curdle_msm_free_slots() > CURDLE_MSM_SLOTS / 2;
    if (early) sink.Begin(Rs, Ss, Ts, Us);
    const auto t1 = std::chrono::steady_clock::now();
    if (!VerifyWithSink(proof, crs, Rs, Ss, Ts, Us, M, rand, sink, &pre)) return false;
    if (!early) sink.Begin(Rs, Ss, Ts, Us);
    const auto t2 = std::chrono::steady_clock::now();
    const bool ok = sink.Verify();  // the batched MSM on the GPU, == A_c
    if (trace) {
      const auto t3 = std::chrono::steady_clock::now();
      auto us = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
      fprintf(stderr, "[verify] begin %.0f us, host transcript+algebra %.0f us, device run %.0f us\n", us(t0, t1), us(t1, t2),
              us(t2, t3));
    }
    return ok;
  }
  MsmAccumulator acc;
  {
    MirrorSink sink(acc, crs, Rs, Ss, Ts, Us);
    if (!VerifyWithSink(proof, crs, Rs, Ss, Ts, Us, M, rand, sink, &pre)) return false;
  }
  bool ok = false;
  msmaccumulator::Status st = acc.Verify(&ok);                                // :313, the batched MSM on the GPU
  if (!st.ok) throw alg::MsmError("verifying msm accumulator: " + st.err, st.rc ? st.rc : CURDLE_EHIP);  // device failure
  return ok;
}

// Cross-proof batch verification (no reference counterpart): the
// host part of every proof (decoding, transcript, challenge algebra, the direct
// same-scalar check) runs on `nthreads` worker threads.  Each worker folds the checks of
// the proofs it handles into one pending group -- concatenated base / scalar lists, no
// cross-proof merging: the GPU does not care about repeated bases, and hashing ~1,400 keys
// per proof into one shared map would serialise the batch -- and every `flush` proofs (and
// at the end) settles the group with ONE MSM on the GPU, from its own thread, so the
// groups' MSMs overlap the other workers' host work.  Per-proof randomness is derived from
// `rand`.  If a group's MSM fails, its proofs are verified one by one, so oks[] is exact
// either way; a proof that does not decode or fails a direct check is rejected without
// joining a group.
std::vector<int> VerifyBatch(const CRS& crs, const std::vector<BatchItem>& items, common::Rand& rand, int nthreads) {
  const size_t k = items.size();
  // pass 1 (DecodeAhead's producers, chunk by chunk, ahead of the workers) walks each proof's
  // wire format and registers its records -- a proof that does not parse is rejected here --,
  // one GPU decoding per chunk; pass 2 (workers) builds the values
  struct BytesSource {
    const std::vector<BatchItem>& items;
    std::vector<size_t> first_point;
    std::vector<char> parses;
    std::unique_ptr<DecodeAhead> ahead;
    explicit BytesSource(const std::vector<BatchItem>& it) : items(it), first_point(it.size(), 0), parses(it.size(), 0) {}
    void Scan(size_t i, PointDecoder& dec) {
      first_point[i] = dec.size();
      try {
        Reader scan(items[i].proof, items[i].proof_len, true);
        scan.collect = &dec;
        Proof::FromReader(scan);
        parses[i] = 1;
      } catch (const std::runtime_error&) {
      }
    }
    bool Ready(size_t i) { return ahead->Ready(i); }
    bool Usable(size_t i) {
      ahead->Wait(i);
      return parses[i] != 0;
    }
    Proof DecodeProof(size_t i) {
      Reader r(items[i].proof, items[i].proof_len, true);
      r.decoded = &ahead->Wait(i);
      r.decoded_pos = first_point[i];
      r.keep_wire = true;  // the proof value lives inside this call, like the caller's bytes
      return Proof::FromReader(r);
    }
    bool Prelude(size_t, VerifyPrelude&) const { return false; }  // the instance arrives decoded
    void Instance(size_t i, std::vector<G1Affine>& Rs, std::vector<G1Affine>& Ss, std::vector<G1Affine>& Ts,
                  std::vector<G1Affine>& Us, Point& M) const {
      const BatchItem& it = items[i];
      Rs.assign(it.Rs, it.Rs + it.ell);
      Ss.assign(it.Ss, it.Ss + it.ell);
      Ts.assign(it.Ts, it.Ts + it.ell);
      Us.assign(it.Us, it.Us + it.ell);
      M = Point::FromJac(it.M);
    }
  } src(items);
  src.ahead = std::make_unique<DecodeAhead>(k, DecodeAheadChunk(k, k ? items[0].proof_len / 48 : 0), DecodeAheadProducers(),
                                            [&src](size_t i, PointDecoder& dec) { src.Scan(i, dec); });
  try {
    return VerifyBatchCore(crs, k, src, rand, BatchWorkers(nthreads));
  } catch (...) {
    src.ahead->Abandon();
    throw;
  }
}

std::vector<uint8_t> Proof::Serialize() const {  // :358-387
  Writer w;
  w.PutPoint(A);
  T.Serialize(w);
  U.Serialize(w);
  w.PutPoint(R);
  w.PutPoint(S);
  proofSamePermutation.Serialize(w);
  proofSameScalar.Serialize(w);
  proofSameMultiscalar.Serialize(w);
  return w.buf;
}
Proof Proof::FromBytes(const uint8_t* data, size_t len, bool subgroup_check) {  // :320-356
  // pass 1 walks the wire format and registers every point record, one GPU kernel decodes
  // them (square roots, curve and subgroup tests), pass 2 builds the value
  PointDecoder dec(subgroup_check);
  Reader scan(data, len, subgroup_check);
  scan.collect = &dec;
  FromReader(scan);
  dec.Run();
  Reader r(data, len, subgroup_check);
  r.decoded = &dec;
  return FromReader(r);
}
Proof Proof::FromBytesDeferred(const uint8_t* data, size_t len, PointDecoder& dec) {
  ScanAndStart(data, len, dec);
  return FromStarted(data, len, dec);
}
void Proof::ScanAndStart(const uint8_t* data, size_t len, PointDecoder& dec) {
  Reader scan(data, len, true);
  scan.collect = &dec;
  FromReader(scan);
  dec.Start();
}
Proof Proof::FromStarted(const uint8_t* data, size_t len, PointDecoder& dec) {
  dec.Run(/*defer_subgroup=*/true);
  Reader r(data, len, true);
  r.decoded = &dec;
  return FromReader(r);
}
Proof Proof::ScanLazy(Reader& r) {
  if (!r.collect) throw std::logic_error("ScanLazy needs a collecting reader");
  r.lazy = true;
  return FromReader(r);
}
Proof Proof::FromReader(Reader& r) {
  Proof p;
  p.A = r.GetPoint("A");
  p.T.FromReader(r);
  p.U.FromReader(r);
  p.R = r.GetPoint("R");
  p.S = r.GetPoint("S");
  p.proofSamePermutation.FromReader(r);
  p.proofSameScalar.FromReader(r);
  p.proofSameMultiscalar.FromReader(r);
  return p;
}

}  // namespace proto
}  // namespace curdle
