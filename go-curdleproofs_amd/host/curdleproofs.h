// Host-side restatement of the Curdleproofs protocol layers that sit around the MSM
// hot path: the five arguments, the top-level shuffle proof,
// the CRS, ShufflePermuteCommit and the proof wire format.  Same package / function
// names, argument meaning and accept / error behaviour as the reference
// (/root/reference, cited per function in curdleproofs.cpp); every MultiExp goes to
// the GPU through alg::MultiExp, every deferred check through the msmaccumulator
// mirror.
//
// This exists so configs 1 / 3 / 5 can be run end to end without a Go toolchain.  It
// is NOT the product's hot path (that is the MSM); it is the caller either side of it.
// UNVERIFIED against Go-produced proofs (none exist in this environment): pinned by the
// Merlin test vector, the common.Rand known answers and the reference's own
// completeness / soundness / serialisation tests, restated in tests/.
//
// Go's (value, error) becomes: bool for the accept bit, std::runtime_error for the
// structural errors the reference returns as a non-nil error.
#pragma once
#include <stdint.h>

#include <functional>
#include <memory>
#include <string>
#include <vector>

#include "algebra.h"
#include "common_rand.h"
#include "msmaccumulator.h"
#include "transcript.h"

namespace curdle {
namespace proto {

using alg::Point;
using alg::Scalar;

static constexpr int N_BLINDERS = 4;  // common/constants.go:3

// ---- wire format helpers (gnark Encoder / Decoder as the reference uses them) ----
struct Writer {
  std::vector<uint8_t> buf;
  void PutPoint(const Point& p);
  void PutScalar(const Scalar& s);
  void PutPoints(const std::vector<Point>& v);  // uint32 big-endian length, then the points
};
// Batched decoding of compressed points on the GPU (curdle_g1_decompress_batch): callers
// register the 48-byte records they will need, Run() decodes them all in one kernel, Get()
// hands them out.  Fewer than kMinDeviceBatch records are decoded on the host, point by
// point (a kernel launch would cost more); larger batches need the GPU -- there is no
// silent host fallback, a missing device is an alg::MsmError.  CURDLE_HOST_DECODE=1 forces
// the host decoder for A/B measurements.
class PointDecoder {
 public:
  explicit PointDecoder(bool subgroup_check);
  ~PointDecoder();
  PointDecoder(const PointDecoder&) = delete;
  PointDecoder& operator=(const PointDecoder&) = delete;
  size_t Add(const uint8_t rec[48]);          // -> index for Get
  // Decodes every record.  With defer_subgroup the GPU's subgroup test keeps running after
  // Run() returns (Get() then only reflects encoding / curve errors) and Finish() collects
  // its verdict: the caller overlaps its own work with it and must not publish a result
  // before Finish() has returned true.
  void Run(bool defer_subgroup = false);
  // Optional first step of a deferred Run(true): launches the GPU decoding and returns, so the
  // caller can hash its transcript from the raw bytes until it calls Run(true), which then only
  // waits for the points.  A no-op when the batch would not go to the GPU's deferred form.
  void Start();
  bool Finish();                              // true: no record failed the (deferred) subgroup test
  bool Get(size_t index, Point* out) const;   // false: not a valid encoding / not on the curve / not in G1
  bool GetAffine(size_t index, G1Affine* out) const;  // the same without the detour; infinity is (0, 0)
  size_t size() const { return n_; }
  static bool OnDevice();                     // false only under CURDLE_HOST_DECODE=1
  static constexpr size_t kMinDeviceBatch = 48;
 private:
  bool subgroup_;
  size_t n_ = 0;
  std::vector<uint8_t> blob_;
  std::vector<G1Affine> pts_;
  std::vector<uint8_t> status_;
  int ticket_ = -1;                           // >= 0: a deferred subgroup test is in flight
  bool started_ = false;                      // Start() launched the decoding; Run() collects the points
};

struct Reader {
  const uint8_t* p;
  size_t left;
  bool subgroup_check;
  // two-pass decoding: with `collect` set GetPoint only registers the record and returns
  // infinity; with `decoded` set it hands out the records in the same order
  // ... and with `lazy` on top of `collect` it returns the point as a pending one (alg::Point:
  // wire record + decoder index, no coordinates), so that ONE pass yields a proof value the
  // verifier can hash and list as bases while the GPU still decodes (VerifyWhileDecoding)
  PointDecoder* collect = nullptr;
  bool lazy = false;
  const PointDecoder* decoded = nullptr;
  size_t decoded_pos = 0;
  // with `decoded`: the points also keep a pointer to their wire record, which the transcript
  // then absorbs as it is instead of compressing the point again.  Only for callers whose proof
  // value does not outlive the bytes (the batch verifiers).
  bool keep_wire = false;
  Reader(const uint8_t* data, size_t len, bool subgroup = false) : p(data), left(len), subgroup_check(subgroup) {}
  Point GetPoint(const char* what);
  Scalar GetScalar(const char* what);
  std::vector<Point> GetPoints(const char* what);
};

// ---- groupcommitment (groupcommitment/groupcommitment.go) ----
struct GroupCommitment {
  Point T_1, T_2;
  static GroupCommitment New(const Point& crsG, const Point& crsH, const Point& T, const Scalar& r);  // :17
  GroupCommitment Add(const GroupCommitment& cm) const;   // :33
  GroupCommitment Mul(const Scalar& s) const;             // :41
  bool Eq(const GroupCommitment& cm) const;               // :50
  void Serialize(Writer& w) const;
  void FromReader(Reader& r);
};

// ---- crs.go ----
struct CRS {
  std::vector<G1Affine> Gs, Hs;
  Point H, Gt, Gu;
  G1Affine Gsum, Hsum;
  // Derived, never serialised: fixed-base tables of Gsum and Hsum (the two points every
  // verification rescales, grandproductargument.go:243-246), built by GenerateCRS.
  std::shared_ptr<const alg::FixedBase> GsumTable, HsumTable;
  // ... and the holder of the device-resident copy of Gs | Hs | H | Gt | Gu
  // (device_accumulator.h), filled on the first verification that uses it.
  std::shared_ptr<class DeviceCrs> device;
};
CRS GenerateCRS(size_t size, common::Rand& rand);  // crs.go:20

// common.ShufflePermuteCommit (common/util.go:45)
struct ShuffleCommit {
  std::vector<G1Affine> Ts, Us;
  Point M;
  std::vector<Scalar> rs_m;
};
ShuffleCommit ShufflePermuteCommit(const std::vector<G1Affine>& crsGs, const std::vector<G1Affine>& crsHs,
                                   const std::vector<G1Affine>& Rs, const std::vector<G1Affine>& Ss,
                                   const std::vector<uint32_t>& perm, const Scalar& k, common::Rand& rand);

// ---- the verifier's deferred checks, described rather than materialised ----
// Every sub-argument ends in msmAccumulator.AccumulateCheck(C, x, v, rand)
// (msmaccumulator.go:23): "sum_i x_i v_i == C".  Here C is handed over as the linear
// combination it is (Terms), x as a description (VecExpr) and v as index ranges of the two base
// sets that are resident on the GPU (the CRS and the instance) plus the odd loose point, so
// that a sink can either rebuild the reference's call on the host mirror (MirrorSink) or
// ship the description to the device accumulator (DeviceSink, device_accumulator.h), which
// evaluates x, multiplies by the check's random weight and adds into the base's scalar slot
// by index -- SURVEY.md section 8f-3.
struct Terms {
  std::vector<Scalar> s;
  std::vector<G1Affine> p;
  std::vector<int32_t> pending;  // per term: >= 0 -> p[j] is not known yet, PointDecoder record index
  void Add(const Scalar& k, const Point& pt) {
    s.push_back(k);
    pending.push_back(pt.pending);
    p.push_back(pt.pending >= 0 ? G1Affine{} : pt.Affine());
  }
  void Add(const std::vector<Scalar>& ks, const std::vector<Point>& pts) {
    for (size_t i = 0; i < pts.size(); i++) Add(ks[i], pts[i]);
  }
};

struct VecExpr {
  enum Kind : uint32_t {
    kExplicit = 0,  // the tail only
    kConst = 1,     // x_i = scale
    kFold = 2,      // x_i = scale * prod_{j : bit j of i set} gammas[m-1-j]   (innerproductargument.go:223-234,
                    //                                                         samemultiscalarargument.go:267-277)
    kFoldPow = 3,   // ... * q^(min(i, q_cap) + 1)                             (grandproductargument.go:234-242)
  };
  Kind kind = kExplicit;
  size_t n_struct = 0;          // elements given by the rule above
  Scalar scale = Scalar::One();
  std::vector<Scalar> gammas;
  Scalar q = Scalar::One();
  size_t q_cap = 0;
  std::vector<Scalar> tail;     // explicit elements after the structured ones
  size_t size() const { return n_struct + tail.size(); }
  Scalar At(size_t i) const;
  std::vector<Scalar> Materialise() const;  // n products for the folds (doubling), not n log n
  static VecExpr Explicit(std::vector<Scalar> v) {
    VecExpr e;
    e.tail = std::move(v);
    return e;
  }
  static VecExpr Const(const Scalar& c, size_t n) {
    VecExpr e;
    e.kind = kConst;
    e.n_struct = n;
    e.scale = c;
    return e;
  }
  // n may exceed 2^m only through the tail: n_struct = min(n, 2^m)
  static VecExpr Fold(const std::vector<Scalar>& gammas, const Scalar& scale, size_t n) {
    VecExpr e;
    e.kind = kFold;
    e.gammas = gammas;
    e.scale = scale;
    const size_t full = (size_t)1 << gammas.size();
    e.n_struct = n < full ? n : full;
    return e;
  }
  static VecExpr FoldPow(const std::vector<Scalar>& gammas, const Scalar& scale, const Scalar& q, size_t q_cap, size_t n) {
    VecExpr e = Fold(gammas, scale, n);
    e.kind = kFoldPow;
    e.q = q;
    e.q_cap = q_cap;
    return e;
  }
};

// Resident base sets and their index layouts.
enum : uint32_t { kSetCrs = 0, kSetInst = 1 };
struct CrsIndex {  // Gs | Hs | H | Gt | Gu
  size_t ell;
  uint32_t G(size_t i) const { return (uint32_t)i; }  // Gs[i] for i < ell, Hs[i - ell] after
  uint32_t H() const { return (uint32_t)(ell + N_BLINDERS); }
  uint32_t Gt() const { return (uint32_t)(ell + N_BLINDERS + 1); }
  uint32_t Gu() const { return (uint32_t)(ell + N_BLINDERS + 2); }
  size_t size() const { return ell + N_BLINDERS + 3; }
};
struct InstIndex {  // Rs | Ss | Ts | Us
  size_t ell;
  uint32_t R(size_t i) const { return (uint32_t)i; }
  uint32_t S(size_t i) const { return (uint32_t)(ell + i); }
  uint32_t T(size_t i) const { return (uint32_t)(2 * ell + i); }
  uint32_t U(size_t i) const { return (uint32_t)(3 * ell + i); }
  size_t size() const { return 4 * ell; }
};
struct BaseSeg {  // slots [first, first + len) of resident set `set` take x[vec_first + j]
  uint32_t set, first, len, vec_first;
};
struct LooseBase {  // a base outside the resident sets takes x[index]
  uint32_t index;
  G1Affine point;
  int32_t pending = -1;  // >= 0: `point` is still being decoded (PointDecoder record index)
  LooseBase(uint32_t i, const Point& pt) : index(i), point(pt.pending >= 0 ? G1Affine{} : pt.Affine()), pending(pt.pending) {}
};

class CheckSink {
 public:
  virtual ~CheckSink() {}
  // msmAccumulator.AccumulateCheck(C, x, v, rand); indices of x covered by neither a segment
  // nor a loose base pair with the point at infinity (the zero entries of T' and U',
  // curdleproof.go:271-285).
  virtual void Check(const Terms& C, const VecExpr& x, const std::vector<BaseSeg>& segs,
                     const std::vector<LooseBase>& loose, common::Rand& rand, const char* what) = 0;
};

// The reference's behaviour: every check goes to the msmaccumulator mirror (deferred or,
// under curdle_verify_set_eager, with C evaluated on the spot).
class MirrorSink : public CheckSink {
 public:
  MirrorSink(msmaccumulator::MsmAccumulator& acc, const struct CRS& crs, const std::vector<G1Affine>& Rs,
             const std::vector<G1Affine>& Ss, const std::vector<G1Affine>& Ts, const std::vector<G1Affine>& Us);
  void Check(const Terms& C, const VecExpr& x, const std::vector<BaseSeg>& segs, const std::vector<LooseBase>& loose,
             common::Rand& rand, const char* what) override;

 private:
  msmaccumulator::MsmAccumulator& acc_;
  std::vector<G1Affine> crs_;  // Gs | Hs | H | Gt | Gu
  const std::vector<G1Affine>*inst_[4];
  size_t ell_;
};

// ---- samescalarargument ----
namespace samescalar {
struct Proof {
  GroupCommitment A, B;
  Scalar Z_k, Z_t, Z_u;
  void Serialize(Writer& w) const;
  void FromReader(Reader& r);
};
Proof Prove(const Point& Gt, const Point& Gu, const Point& H, const Point& R, const Point& S,
            const GroupCommitment& T, const GroupCommitment& U, const Scalar& k, const Scalar& r_t, const Scalar& r_u,
            transcript::Transcript& tr, common::Rand& rand);
// With an accumulator (and not in eager mode) the two commitment equations join the batched
// check instead of being evaluated on the spot; without one this is the reference's Verify.
bool Verify(const Proof& proof, const Point& Gt, const Point& Gu, const Point& H, const Point& R, const Point& S,
            const GroupCommitment& T, const GroupCommitment& U, transcript::Transcript& tr,
            CheckSink* sink = nullptr, common::Rand* rand = nullptr, size_t ell = 0);
}  // namespace samescalar

// ---- innerproductargument ----
namespace ipa {
struct Proof {
  Point B_c, B_d;
  std::vector<Point> L_Cs, R_Cs, L_Ds, R_Ds;
  Scalar c0, d0;
  void Serialize(Writer& w) const;
  void FromReader(Reader& r);
};
// Gs / Gs_prime / cs / ds are taken by value: the prover folds them in place.  With
// `Gs_prime_scale` the second base vector is scale[i] * Gs_prime[i] without anybody having
// computed those points: the prover only ever uses them as MSM bases and multiplies the scale
// into its scalars (the grand product argument's rescaled bases, grandproductargument.go:94-103).
Proof Prove(std::vector<G1Affine> Gs, std::vector<G1Affine> Gs_prime, const Point& H, const Point& C, const Point& D,
            const Scalar& z, std::vector<Scalar> cs, std::vector<Scalar> ds, transcript::Transcript& tr,
            common::Rand& rand, const std::vector<Scalar>* Gs_prime_scale = nullptr);
// Bases Gs | Hs (n = ell + 4, the CRS's, by index) and H; us_i = u_q^(min(i, ell) + 1).
bool Verify(const Proof& proof, size_t ell, const Point& H, const Point& C, const Point& D, const Scalar& z,
            const Scalar& u_q, transcript::Transcript& tr, CheckSink& sink, common::Rand& rand);
}  // namespace ipa

// ---- grandproductargument ----
namespace gprod {
struct Proof {
  Point C;
  Scalar Rp;
  ipa::Proof IPAProof;
  void Serialize(Writer& w) const;
  void FromReader(Reader& r);
};
Proof Prove(const std::vector<G1Affine>& Gs, const std::vector<G1Affine>& Hs, const Point& H, const Point& B,
            const Scalar& result, const std::vector<Scalar>& bs, const std::vector<Scalar>& r_bs,
            transcript::Transcript& tr, common::Rand& rand);
bool Verify(const Proof& proof, const struct CRS& crs, const Point& B, const Scalar& result, transcript::Transcript& tr,
            CheckSink& sink, common::Rand& rand);
}  // namespace gprod

// ---- samepermutationargument ----
namespace sameperm {
struct Proof {
  Point B;
  gprod::Proof gpaProof;
  void Serialize(Writer& w) const;
  void FromReader(Reader& r);
};
Proof Prove(const std::vector<G1Affine>& Gs, const std::vector<G1Affine>& Hs, const Point& H, const Point& A,
            const Point& M, const std::vector<Scalar>& as, const std::vector<uint32_t>& permutation,
            const std::vector<Scalar>& rs_a, const std::vector<Scalar>& rs_m, transcript::Transcript& tr,
            common::Rand& rand);
bool Verify(const Proof& proof, const struct CRS& crs, const Point& A, const Point& M, const std::vector<Scalar>& as,
            transcript::Transcript& tr, CheckSink& sink, common::Rand& rand);
}  // namespace sameperm

// ---- samemultiscalarargument ----
namespace samemsm {
struct Proof {
  Point B_a, B_t, B_u;
  std::vector<Point> L_A, L_T, L_U, R_A, R_T, R_U;
  Scalar x;
  void Serialize(Writer& w) const;
  void FromReader(Reader& r);
};
// G, T, U, x by value: folded in place by the prover.
Proof Prove(std::vector<G1Affine> G, const Point& A, const Point& Z_t, const Point& Z_u, std::vector<G1Affine> T,
            std::vector<G1Affine> U, std::vector<Scalar> x, transcript::Transcript& tr, common::Rand& rand);
// T, U: the padded vectors T', U' (for the transcript); as bases they are addressed by index.
bool Verify(const Proof& proof, size_t ell, const Point& A, const Point& Z_t, const Point& Z_u,
            const std::vector<G1Affine>& T, const std::vector<G1Affine>& U, transcript::Transcript& tr, CheckSink& sink,
            common::Rand& rand, const std::vector<uint8_t>* Tbytes = nullptr, const std::vector<uint8_t>* Ubytes = nullptr);
}  // namespace samemsm

// ---- curdleproof.go ----
struct Proof {
  Point A;
  GroupCommitment T, U;
  Point R, S;
  sameperm::Proof proofSamePermutation;
  samescalar::Proof proofSameScalar;
  samemsm::Proof proofSameMultiscalar;
  std::vector<uint8_t> Serialize() const;                                    // curdleproof.go:358
  static Proof FromBytes(const uint8_t* data, size_t len, bool subgroup_check = false);  // :320
  static Proof FromReader(Reader& r);  // the same from a stream position (trailing bytes are left unread)
  // Decoding with the GPU's subgroup test left running: the caller must call dec.Finish()
  // and treat `false` as the decoding error it is before using any result derived from
  // the proof.
  static Proof FromBytesDeferred(const uint8_t* data, size_t len, PointDecoder& dec);
  // The same in two steps, for callers with work to do while the GPU decodes: ScanAndStart
  // registers the proof's point records with `dec` and launches the decoding; FromStarted waits
  // for the points and builds the value (the subgroup verdict still comes from dec.Finish()).
  static void ScanAndStart(const uint8_t* data, size_t len, PointDecoder& dec);
  static Proof FromStarted(const uint8_t* data, size_t len, PointDecoder& dec);
  // One pass for VerifyWhileDecoding: registers the records with `dec` and returns the proof with
  // every point pending (valid only while `data` and `dec` live).  The caller adds what else it
  // wants decoded and calls dec.Start().
  static Proof ScanLazy(Reader& r);
};
Proof Prove(const CRS& crs, const std::vector<G1Affine>& Rs, const std::vector<G1Affine>& Ss,
            const std::vector<G1Affine>& Ts, const std::vector<G1Affine>& Us, const Point& M,
            const std::vector<uint32_t>& perm, const Scalar& k, const std::vector<Scalar>& rs_m,
            common::Rand& rand);                                            // :38
bool Verify(const Proof& proof, const CRS& crs, const std::vector<G1Affine>& Rs, const std::vector<G1Affine>& Ss,
            const std::vector<G1Affine>& Ts, const std::vector<G1Affine>& Us, const Point& M,
            common::Rand& rand);                                            // :199

// The first steps of Verify (curdleproof.go:199-224) -- absorbing the instance and drawing the
// challenge vector `as` -- need only the instance's compressed encodings, not the proof: a
// caller that still waits for the GPU to decode the proof runs them meanwhile
// (curdle_verify, IsValidWhiskShuffleProof).  The encodings are kept: the same-multiscalar
// argument absorbs Ts and Us a second time.
struct VerifyPrelude {
  transcript::Transcript tr;
  std::vector<Scalar> as;
  std::vector<uint8_t> Tb, Ub;  // 48 bytes per point
  VerifyPrelude();
};
// from compressed encodings (ell records each, M one record) ...
void StartVerify(VerifyPrelude& pre, size_t ell, const uint8_t* Rb, const uint8_t* Sb, const uint8_t* Tb, const uint8_t* Ub,
                 const uint8_t Mb[48]);
// ... or from the decoded instance
void StartVerify(VerifyPrelude& pre, const std::vector<G1Affine>& Rs, const std::vector<G1Affine>& Ss,
                 const std::vector<G1Affine>& Ts, const std::vector<G1Affine>& Us, const Point& M);

// The body of Verify up to, but not including, the accumulator's final MSM (false: a direct,
// non-accumulated check already failed); lets several proofs share one accumulator.
bool VerifyInto(const Proof& proof, const CRS& crs, const std::vector<G1Affine>& Rs, const std::vector<G1Affine>& Ss,
                const std::vector<G1Affine>& Ts, const std::vector<G1Affine>& Us, const Point& M, common::Rand& rand,
                msmaccumulator::MsmAccumulator& acc);
// The same into any sink.
bool VerifyWithSink(const Proof& proof, const CRS& crs, const std::vector<G1Affine>& Rs, const std::vector<G1Affine>& Ss,
                    const std::vector<G1Affine>& Ts, const std::vector<G1Affine>& Us, const Point& M, common::Rand& rand,
                    CheckSink& sink, VerifyPrelude* started = nullptr);
// curdleproof.Verify continued from a prelude the caller started earlier
bool VerifyStarted(VerifyPrelude& pre, const Proof& proof, const CRS& crs, const std::vector<G1Affine>& Rs,
                   const std::vector<G1Affine>& Ss, const std::vector<G1Affine>& Ts, const std::vector<G1Affine>& Us,
                   const Point& M, common::Rand& rand);

// curdleproof.Verify for a proof whose points the GPU is still decoding: `proof` and M come from
// a lazy Reader over `dec` (Proof::ScanLazy), so the WHOLE transcript and challenge algebra run
// from the wire bytes while the decoding kernel works; the four points the verifier does
// arithmetic on (B, and A, T.T_1, U.T_1 for A') are decoded on the host meanwhile -- the GPU's
// verdict on the same records still decides.  Then: wait for the points, `after_decode` (the
// caller checks the records' statuses in the reference's order, throwing its decoding errors,
// and fills the instance), the pending bases are filled in and the accumulator's one MSM runs.
// Structural errors of the proof surface after the decoding errors, as in the reference (where
// decoding comes first).  Needs the device accumulator and deferred checks
// (CanVerifyWhileDecoding); the caller still owes dec.Finish() for the subgroup verdict.
bool CanVerifyWhileDecoding();
struct DecodedInstance {
  std::vector<G1Affine> Rs, Ss, Ts, Us;
};
bool VerifyWhileDecoding(VerifyPrelude& pre, const Proof& proof, const CRS& crs, const Point& M, PointDecoder& dec,
                         const std::function<void(DecodedInstance&)>& after_decode, common::Rand& rand);

// Cross-proof batch verification over one CRS: one shared accumulator, one MSM (see the
// definition).  Returns the per-proof accept bits.
struct BatchItem {  // borrowed buffers: ell affine points each, M as 18 Jacobian limbs
  const uint8_t* proof;
  size_t proof_len;
  const G1Affine *Rs, *Ss, *Ts, *Us;
  size_t ell;
  const uint64_t* M;
};
std::vector<int> VerifyBatch(const CRS& crs, const std::vector<BatchItem>& items, common::Rand& rand, int nthreads);

// Deferred (default) or eager evaluation of the verifier's check points; see
// curdle_verify_set_eager in include/curdle_msm.h.  Returns the previous setting.
int SetEagerChecks(int eager);
bool EagerChecksEnabled();

}  // namespace proto
}  // namespace curdle
