// TEST INFRASTRUCTURE ONLY -- never linked into libcurdlemsm.so.
//
// A naive host stand-in for the device entry points the host-side protocol code calls
// (curdle_msm_g1, curdle_msm_g1_batch, curdle_g1_scalar_mul_batch, curdle_g1_decompress_*),
// so that the host layer (wire-format readers, PointDecoder, MsmAccumulator table, transcript,
// the five arguments, Whisk) can be built with -fsanitize=address,undefined and run on a
// machine without a GPU (tests/test_host_sanitize.py), and profiled with gprof.  The product
// has no CPU MSM: libcurdlemsm.so's entry points fail with CURDLE_ENODEV without a device
// (tests/test_abi.py).  The sums below are textbook double-and-add over host_math.h.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/curdle_msm.h"
#include "../../go-curdleproofs_amd/csrc/host_math.h"
#include "../../go-curdleproofs_amd/host/algebra.h"

using namespace curdle;

static thread_local char g_err[256] = "";

extern "C" int curdle_set_last_error(int code, const char* msg) {
  snprintf(g_err, sizeof(g_err), "%s", msg);
  return code;
}
extern "C" int curdle_last_error(char* buf, size_t len) {
  if (!buf || !len) return CURDLE_EINVAL;
  snprintf(buf, len, "%s", g_err);
  return CURDLE_OK;
}

// The library's eight workspace slots, modelled (CURDLE_STUB_SLOTS=1) so that the host layer's
// slot discipline can be tested without a GPU: an accumulation holds a slot from begin to wait /
// abort, an MSM for the duration of the call, and a one-shot decoding takes one when the device
// would (CURDLE_TWO_KERNEL_MAX=0: every chunk goes to the fused kernel).  A caller that waits
// for a decoding while holding a slot deadlocks here exactly as it would on the device.
namespace {
struct SlotPool {
  std::mutex mu;
  std::condition_variable cv;
  int free_slots = CURDLE_MSM_SLOTS;
  bool on() const {
    static const bool v = getenv("CURDLE_STUB_SLOTS") != nullptr;
    return v;
  }
  void acquire() {
    if (!on()) return;
    std::unique_lock<std::mutex> g(mu);
    cv.wait(g, [&] { return free_slots > 0; });
    free_slots--;
  }
  void release() {
    if (!on()) return;
    {
      std::lock_guard<std::mutex> g(mu);
      free_slots++;
    }
    cv.notify_one();
  }
};
SlotPool g_slots;
struct SlotHold {
  SlotHold() { g_slots.acquire(); }
  ~SlotHold() { g_slots.release(); }
};
}  // namespace

// Contexts: the stub poses as CURDLE_STUB_DEVICES devices (default 1) so that the sharding of
// the batch entry points and the per-thread device selection can be exercised without a GPU;
// it records which "device" every accumulation and decoding ran on.
static int stub_devices() {
  static const int v = [] {
    const char* e = getenv("CURDLE_STUB_DEVICES");
    const int n = e ? atoi(e) : 1;
    return n < 1 ? 1 : n > CURDLE_MAX_DEVICES ? CURDLE_MAX_DEVICES : n;
  }();
  return v;
}
static thread_local int tl_stub_dev = 0;
static std::mutex g_stub_count_mu;
static unsigned long long g_stub_calls[CURDLE_MAX_DEVICES] = {0};
static void stub_count_call() {
  std::lock_guard<std::mutex> g(g_stub_count_mu);
  g_stub_calls[tl_stub_dev]++;
}
extern "C" int curdle_device_count(void) { return stub_devices(); }
extern "C" int curdle_set_device(int ordinal) {
  if (ordinal < 0 || ordinal >= stub_devices()) return curdle_set_last_error(CURDLE_EINVAL, "device ordinal out of range");
  tl_stub_dev = ordinal;
  return CURDLE_OK;
}
extern "C" int curdle_get_device(void) { return tl_stub_dev; }
// test hook of the stub only: device-entry-point calls seen per posed device
extern "C" unsigned long long curdle_stub_calls_on(int ordinal) {
  std::lock_guard<std::mutex> g(g_stub_count_mu);
  return ordinal >= 0 && ordinal < CURDLE_MAX_DEVICES ? g_stub_calls[ordinal] : 0;
}

extern "C" int curdle_msm_free_slots(void) {
  std::lock_guard<std::mutex> g(g_slots.mu);
  return g_slots.free_slots;
}

static void msm_naive(const uint64_t* points, const uint64_t* scalars, size_t n, uint64_t out[18]) {
  G1XYZZ acc;
  g1_set_inf(acc);
  for (size_t i = 0; i < n; i++) {
    G1Affine p;
    memcpy(&p, points + 12 * i, 96);
    if (g1_affine_is_inf(p)) continue;
    Fr m, k;
    memcpy(&m, scalars + 4 * i, 32);
    f_from_mont<FrParams>(k, m);
    G1XYZZ px, t;
    g1_from_affine(px, p);
    g1_scalar_mul(t, px, k.l, 8);
    g1_add(acc, t);
  }
  g1_to_canonical_jac(out, acc);
}

extern "C" int curdle_msm_g1(const uint64_t* points, const uint64_t* scalars, size_t n, uint64_t out_jac[18]) {
  if (!out_jac || (n && (!points || !scalars))) return curdle_set_last_error(CURDLE_EINVAL, "null argument");
  SlotHold hold;
  msm_naive(points, scalars, n, out_jac);
  return CURDLE_OK;
}

extern "C" int curdle_msm_g1_batch(const uint64_t* points, const uint64_t* scalars, const size_t* offsets, size_t k,
                                   uint64_t* out_jac) {
  if (!offsets || (k && !out_jac)) return curdle_set_last_error(CURDLE_EINVAL, "null argument");
  SlotHold hold;
  for (size_t j = 0; j < k; j++)
    msm_naive(points + 12 * offsets[j], scalars + 4 * offsets[j], offsets[j + 1] - offsets[j], out_jac + 18 * j);
  return CURDLE_OK;
}

extern "C" int curdle_msm_g1_multi(const uint64_t* const* points_sets, size_t k, const uint64_t* scalars, size_t n,
                                   uint64_t* out_jac) {
  for (size_t j = 0; j < k; j++) msm_naive(points_sets[j], scalars, n, out_jac + 18 * j);
  return CURDLE_OK;
}

extern "C" int curdle_g1_scalar_mul_batch(const uint64_t* points, const uint64_t* scalars, size_t n_scalars,
                                          const uint64_t* addends, size_t n, uint64_t* out_affine) {
  if (n_scalars != n && n_scalars != 1) return curdle_set_last_error(CURDLE_EINVAL, "n_scalars must be n or 1");
  for (size_t i = 0; i < n; i++) {
    uint64_t jac[18];
    msm_naive(points + 12 * i, scalars + 4 * (n_scalars == 1 ? 0 : i), 1, jac);
    alg::Point r = alg::Point::FromJac(jac);
    if (addends) {
      G1Affine a;
      memcpy(&a, addends + 12 * i, 96);
      r = r + alg::Point::FromAffine(a);
    }
    G1Affine o = r.Affine();
    memcpy(out_affine + 12 * i, &o, 96);
  }
  return CURDLE_OK;
}

static void decode_all(const uint8_t* in, size_t n, int subgroup_check, uint64_t* out_affine, uint8_t* status) {
  for (size_t i = 0; i < n; i++) {
    alg::Point p;
    G1Affine a;
    memset(&a, 0, sizeof(a));
    status[i] = CURDLE_DECODE_BAD_ENCODING;
    if (alg::Point::FromCompressed(in + 48 * i, &p, subgroup_check != 0)) {
      a = p.Affine();
      status[i] = g1_affine_is_inf(a) ? CURDLE_DECODE_INFINITY : CURDLE_DECODE_OK;
    } else if (subgroup_check && alg::Point::FromCompressed(in + 48 * i, &p, false)) {
      status[i] = CURDLE_DECODE_NOT_IN_SUBGROUP;
    }
    memcpy(out_affine + 12 * i, &a, 96);
  }
}

extern "C" int curdle_g1_decompress_batch(const uint8_t* in, size_t n, int subgroup_check, uint64_t* out_affine,
                                          uint8_t* status) {
  // the device's one-shot decoding takes a workspace slot whenever the batch is beyond the
  // two-kernel size (decode_api.hip: two_kernel_max) or every decode context is taken
  stub_count_call();
  const char* tk = getenv("CURDLE_TWO_KERNEL_MAX");
  const bool needs_slot = tk && (size_t)atoll(tk) < n;
  if (const char* d = getenv("CURDLE_STUB_DECODE_DELAY_MS")) std::this_thread::sleep_for(std::chrono::milliseconds(atoi(d)));
  if (needs_slot) g_slots.acquire();
  decode_all(in, n, subgroup_check, out_affine, status);
  if (needs_slot) g_slots.release();
  return CURDLE_OK;
}

// two-step form: everything happens in begin; finish hands the final status bytes back
static std::vector<uint8_t> g_deferred[2];
static bool g_busy[2] = {false, false};
extern "C" int curdle_g1_decompress_begin(const uint8_t* in, size_t n, uint64_t* out_affine, uint8_t* status, int* ticket) {
  for (int t = 0; t < 2; t++) {
    if (g_busy[t]) continue;
    g_busy[t] = true;
    g_deferred[t].assign(n, 0);
    decode_all(in, n, 1, out_affine, g_deferred[t].data());
    // like the device: the first step reports curve membership only
    for (size_t i = 0; i < n; i++) status[i] = g_deferred[t][i] == CURDLE_DECODE_NOT_IN_SUBGROUP ? CURDLE_DECODE_OK : g_deferred[t][i];
    for (size_t i = 0; i < n; i++)
      if (g_deferred[t][i] == CURDLE_DECODE_NOT_IN_SUBGROUP) {
        alg::Point p;
        alg::Point::FromCompressed(in + 48 * i, &p, false);
        G1Affine a = p.Affine();
        memcpy(out_affine + 12 * i, &a, 96);
      }
    *ticket = t;
    return CURDLE_OK;
  }
  return curdle_set_last_error(CURDLE_EBUSY, "deferred decodings in flight");
}
// three-step form: start decodes into a parked copy, points hands it out
static std::vector<uint64_t> g_parked_pts[2];
static std::vector<uint8_t> g_parked_st[2];
extern "C" int curdle_g1_decompress_start(const uint8_t* in, size_t n, int* ticket) {
  std::vector<uint64_t> pts(12 * (n ? n : 1));
  std::vector<uint8_t> st(n ? n : 1);
  int rc = curdle_g1_decompress_begin(in, n, pts.data(), st.data(), ticket);
  if (rc) return rc;
  g_parked_pts[*ticket] = pts;
  g_parked_st[*ticket] = st;
  return CURDLE_OK;
}
extern "C" int curdle_g1_decompress_points(int ticket, uint64_t* out_affine, uint8_t* status) {
  if (ticket < 0 || ticket > 1 || !g_busy[ticket]) return curdle_set_last_error(CURDLE_EINVAL, "bad ticket");
  const size_t n = g_deferred[ticket].size();
  memcpy(out_affine, g_parked_pts[ticket].data(), n * 96);
  memcpy(status, g_parked_st[ticket].data(), n);
  return CURDLE_OK;
}
extern "C" int curdle_g1_decompress_finish(int ticket, uint8_t* status) {
  if (ticket < 0 || ticket > 1 || !g_busy[ticket]) return curdle_set_last_error(CURDLE_EINVAL, "bad ticket");
  if (status) memcpy(status, g_deferred[ticket].data(), g_deferred[ticket].size());
  g_busy[ticket] = false;
  return CURDLE_OK;
}

// ---- device accumulator (curdle_dbases_* / curdle_dacc_*), evaluated naively on the host ----
struct curdle_dbases {
  std::vector<G1Affine> pts;
};
struct curdle_dacc {
  const curdle_dbases* crs;
  std::vector<G1Affine> inst;
  bool submitted = false;
  uint64_t result[18] = {0};
};
extern "C" int curdle_dbases_create(const uint64_t* points, size_t n, curdle_dbases** out) {
  curdle_dbases* b = new curdle_dbases();
  b->pts.resize(n);
  if (n) memcpy(b->pts.data(), points, n * 96);
  *out = b;
  return CURDLE_OK;
}
extern "C" void curdle_dbases_free(curdle_dbases* b) { delete b; }
extern "C" size_t curdle_dbases_size(const curdle_dbases* b) { return b ? b->pts.size() : 0; }
extern "C" int curdle_dbases_valid(const curdle_dbases* b) { return b ? 1 : 0; }
extern "C" int curdle_dacc_begin(const curdle_dbases* crs, const uint64_t* inst_points, size_t n_inst, curdle_dacc** out) {
  stub_count_call();
  g_slots.acquire();  // held until wait / abort / a failed submit
  curdle_dacc* a = new curdle_dacc();
  a->crs = crs;
  a->inst.resize(n_inst);
  if (n_inst) memcpy(a->inst.data(), inst_points, n_inst * 96);
  *out = a;
  return CURDLE_OK;
}
extern "C" void curdle_dacc_abort(curdle_dacc* acc) {
  if (!acc) return;
  g_slots.release();
  delete acc;
}
// submit computes at once (there is nothing to overlap with on the host backend); poll is always
// done; wait hands the stored result out
extern "C" int curdle_dacc_submit(curdle_dacc* acc, const curdle_dacc_check* checks, size_t n_checks, const uint64_t* pool,
                                  size_t pool_len, const uint64_t* extra_points, const uint64_t* extra_scalars,
                                  size_t n_extra, uint64_t* export_scalars) {
  uint64_t* out_jac = acc->result;
  const size_t n_crs = acc->crs->pts.size(), n_inst = acc->inst.size(), n_res = n_crs + n_inst;
  auto P = [&](uint32_t off) {
    alg::Scalar s;
    if (off >= pool_len) throw 1;
    memcpy(&s.v, pool + 4 * (size_t)off, 32);
    return s;
  };
  std::vector<alg::Scalar> slots(n_res, alg::Scalar::Zero());
  int rc = CURDLE_OK;
  try {
    for (size_t slot = 0; slot < n_res; slot++) {
      const uint32_t set = slot < n_crs ? CURDLE_SET_CRS : CURDLE_SET_INST;
      const uint32_t idx = (uint32_t)(slot < n_crs ? slot : slot - n_crs);
      for (size_t c = 0; c < n_checks; c++) {
        const curdle_dacc_check& ck = checks[c];
        for (uint32_t s = 0; s < ck.nseg; s++) {
          if (ck.seg[s].set != set || idx < ck.seg[s].first || idx - ck.seg[s].first >= ck.seg[s].len) continue;
          const uint32_t i = ck.seg[s].vec_first + (idx - ck.seg[s].first);
          alg::Scalar v;
          if (i >= ck.n_struct) {
            if (i - ck.n_struct >= ck.n_tail) continue;
            v = P(ck.alpha_off) * P(ck.tail_off + (i - ck.n_struct));
          } else {
            v = P(ck.weight_off);
            if (ck.kind >= CURDLE_VEC_FOLD)
              for (uint32_t j = 0; j < ck.m; j++)
                if ((i >> j) & 1u) v = v * P(ck.gammas_off + (ck.m - 1 - j));
            if (ck.kind == CURDLE_VEC_FOLD_POW) v = v * P(ck.q_off).Pow((i < ck.q_cap ? i : ck.q_cap) + 1);
          }
          slots[slot] = slots[slot] + v;
        }
      }
    }
  } catch (int) {
    rc = curdle_set_last_error(CURDLE_EINVAL, "offset outside the pool");
  }
  if (rc == CURDLE_OK) {
    std::vector<G1Affine> pts(acc->crs->pts);
    pts.insert(pts.end(), acc->inst.begin(), acc->inst.end());
    std::vector<uint64_t> sc(4 * (n_res + n_extra));
    if (n_res) memcpy(sc.data(), slots.data(), n_res * 32);
    pts.resize(n_res + n_extra);
    if (n_extra) {
      memcpy(pts.data() + n_res, extra_points, n_extra * 96);
      memcpy(sc.data() + 4 * n_res, extra_scalars, n_extra * 32);
    }
    msm_naive(reinterpret_cast<const uint64_t*>(pts.data()), sc.data(), pts.size(), out_jac);
    if (export_scalars && n_res) memcpy(export_scalars, slots.data(), n_res * 32);
    acc->submitted = true;
  } else {
    g_slots.release();
    delete acc;  // a failed submission ends the accumulation
  }
  return rc;
}
extern "C" int curdle_dacc_poll(curdle_dacc* acc, int* done) {
  if (!acc || !done || !acc->submitted) return curdle_set_last_error(CURDLE_EINVAL, "accumulation not submitted");
  *done = 1;
  return CURDLE_OK;
}
extern "C" int curdle_dacc_wait(curdle_dacc* acc, uint64_t out_jac[18]) {
  if (!acc || !out_jac || !acc->submitted) return curdle_set_last_error(CURDLE_EINVAL, "accumulation not submitted");
  memcpy(out_jac, acc->result, sizeof(acc->result));
  g_slots.release();
  delete acc;
  return CURDLE_OK;
}
extern "C" int curdle_dacc_run(curdle_dacc* acc, const curdle_dacc_check* checks, size_t n_checks, const uint64_t* pool,
                               size_t pool_len, const uint64_t* extra_points, const uint64_t* extra_scalars,
                               size_t n_extra, uint64_t out_jac[18], uint64_t* export_scalars) {
  int rc = curdle_dacc_submit(acc, checks, n_checks, pool, pool_len, extra_points, extra_scalars, n_extra, export_scalars);
  if (rc) return rc;
  return curdle_dacc_wait(acc, out_jac);
}
