// Resident, pre-converted base sets (curdle_dbases) and the verifier's accumulator on the device (curdle_dacc): C ABI and
// the host side of both.  (Part of msm_api.hip until round 6.)
#include "msm_internal.h"

// ---------------------------------------------------------------------------
// Accumulator on the device
// ---------------------------------------------------------------------------
struct curdle_dbases {
  size_t n = 0;
  std::vector<uint64_t> host;  // the n gnark points: a context that has not used the set yet converts its own copy from here
  std::mutex mu;
  void* d28[kMaxDevices] = {};       // per context: n internal-form points (P and phi(P) each), kA28Bytes apart
  unsigned epoch[kMaxDevices] = {};  // the context generation each copy belongs to
  int hipdev[kMaxDevices] = {};      // ... and the HIP device it was allocated on (a copy outlives its context: no device reset)
  int users = 0;                     // accumulations in flight that copy from this set
  bool dead = false;                 // curdle_dbases_free came while users > 0: the last user deletes
};
struct curdle_dacc {
  Ctx* ctx = nullptr;  // the context the accumulation runs on (the beginning thread's current one)
  int slot = -1;
  curdle_dbases* crs = nullptr;
  size_t n_crs = 0;
  size_t n_inst = 0;
  // after curdle_dacc_submit
  bool submitted = false;
  size_t n_total = 0;                  // resident + loose bases of the MSM
  uint64_t* export_scalars = nullptr;  // caller's buffer for the slot scalars (tests), filled by wait
  size_t export_off = 0;               // where they wait in the slot's pinned staging
};

namespace curdle_api {
void dbases_destroy(curdle_dbases* b) {  // nobody else holds b any more
  for (int i = 0; i < kMaxDevices; i++) {
    if (!b->d28[i]) continue;
    // curdle_shutdown destroys a context's streams and workspaces, not the device: a copy made under a
    // closed context is still an allocation of its device and is freed here (review of round 3: it leaked)
    if (hipSetDevice(b->hipdev[i]) == hipSuccess) (void)hipFree(b->d28[i]);
  }
  delete b;
}

// The set's copy on context cx, converted on first use (and again after a shutdown); takes a
// user reference that dbases_release gives back.
int dbases_acquire(Ctx& cx, curdle_dbases* b, void** d28) {
  std::lock_guard<std::mutex> gb(b->mu);
  if (b->dead) return fail(CURDLE_EINVAL, "resident bases were freed");
  const int o = cx.ordinal;
  unsigned epoch;
  int device;
  hipStream_t st = nullptr;
  {  // the context's mutex only to bring it up and read what the copy is tied to: the upload and the
     // conversion below run under the set's own mutex, so that they do not hold up every slot acquire /
     // release of the context (review of round 3).  On the context's utility stream: a stream created and
     // destroyed here cost every later verification 0.25 ms until the next hipFree
     // (tools/dbg_verify_slowdown.py: 0.85 -> 1.09 ms at ell = 252 after ANY base set had been made).
    std::lock_guard<std::mutex> g(cx.mu);
    int rc = init_default_locked(cx);
    if (rc) return rc;
    epoch = cx.epoch;
    device = cx.device;
    st = cx.util_stream;
  }
  if (b->d28[o] && (b->epoch[o] != epoch || b->hipdev[o] != device)) {  // a copy made under a context that was closed since
    if (hipSetDevice(b->hipdev[o]) == hipSuccess) (void)hipFree(b->d28[o]);
    b->d28[o] = nullptr;
  }
  if (!b->d28[o] && b->n) {
    // the upload runs outside cx.mu on the context's utility stream: counted as in flight, so that a
    // concurrent curdle_shutdown answers CURDLE_EBUSY instead of destroying the stream under the copy
    // (review of round 4), and published only if the context is still the one the copy was made under
    {
      std::lock_guard<std::mutex> g(cx.mu);
      if (!cx.inited || cx.epoch != epoch) return fail(CURDLE_EBUSY, "the context was shut down under a resident-bases upload");
      cx.pending_uploads++;
    }
    struct Pending {
      Ctx& c;
      ~Pending() {
        std::lock_guard<std::mutex> g(c.mu);
        c.pending_uploads--;
      }
    } pending{cx};
    hipError_t e = hipSetDevice(device);
    void *dst = nullptr, *tmp = nullptr;
    if (e == hipSuccess) e = hipMalloc(&dst, 2 * b->n * kA28Bytes);  // P and phi(P) per base (launch_convert_points_raw)
    if (e == hipSuccess) e = hipMalloc(&tmp, b->n * 96);
    if (e == hipSuccess) e = hipMemcpyAsync(tmp, b->host.data(), b->n * 96, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = launch_convert_points_raw(tmp, (uint32_t)b->n, dst, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (tmp) (void)hipFree(tmp);
    if (e != hipSuccess) {
      if (dst) (void)hipFree(dst);
      return fail(e == hipErrorOutOfMemory ? CURDLE_ENOMEM : CURDLE_EHIP, "resident bases: %s", hipGetErrorString(e));
    }
    b->d28[o] = dst;
    b->epoch[o] = epoch;
    b->hipdev[o] = device;
  }
  b->users++;
  *d28 = b->d28[o];
  return CURDLE_OK;
}

void dbases_release(curdle_dbases* b) {
  bool last = false;
  {
    std::lock_guard<std::mutex> g(b->mu);
    last = --b->users == 0 && b->dead;
  }
  if (last) dbases_destroy(b);
}
}  // namespace curdle_api

extern "C" int curdle_dbases_create(const uint64_t* points, size_t n, curdle_dbases** out) {
  if (!out || (n && !points)) return fail(CURDLE_EINVAL, "null argument");
  *out = nullptr;
  if (n > ((size_t)1 << 24)) return fail(CURDLE_EINVAL, "n = %zu exceeds the supported 2^24 resident bases", n);
  curdle_dbases* b = new (std::nothrow) curdle_dbases();
  if (!b) return fail(CURDLE_ENOMEM, "out of memory");
  b->n = n;
  try {
    b->host.assign(points, points + 12 * n);
  } catch (const std::bad_alloc&) {
    delete b;
    return fail(CURDLE_ENOMEM, "out of memory");
  }
  // resident on the creating thread's context now (a failure here is the caller's to see); on
  // every other context when an accumulation there first names the set
  void* d = nullptr;
  int rc = dbases_acquire(cur(), b, &d);
  if (rc) {
    delete b;
    return rc;
  }
  dbases_release(b);
  *out = b;
  return CURDLE_OK;
}

extern "C" void curdle_dbases_free(curdle_dbases* b) {
  if (!b) return;
  {
    std::lock_guard<std::mutex> g(b->mu);
    if (b->users > 0) {  // an accumulation is still copying from the set: its end deletes it
      b->dead = true;
      return;
    }
  }
  dbases_destroy(b);
}

extern "C" size_t curdle_dbases_size(const curdle_dbases* b) { return b ? b->n : 0; }

// A handle keeps its points and re-creates its device copies as needed (another context, a
// context re-initialised after curdle_shutdown): it stays usable until it is freed.
extern "C" int curdle_dbases_valid(const curdle_dbases* b) { return b ? 1 : 0; }

void curdle_api::dbases_release_handle(curdle_dbases* b) { dbases_release(b); }

// ---------------------------------------------------------------------------
// The plain MSM over a resident, pre-converted base set: msmaccumulator.Verify's bases are mostly
// the CRS (/root/reference/crs.go:10-18; msmaccumulator.go:59), which never changes -- and
// k_convert_points is 0.10 ms and 360 MB of every 2^20 call (0.16 ms of a pipelined step), which every
// rank of a window split repeats for ALL points.  The set's internal-form records are read in place.
// ---------------------------------------------------------------------------
namespace curdle_api {
int dbases_msm_args(const curdle_dbases* bases, const void* scalars, size_t n, uint64_t* out_jac) {
  if (!bases || !out_jac) return fail(CURDLE_EINVAL, "null argument");
  if (n > bases->n) return fail(CURDLE_EINVAL, "n = %zu exceeds the %zu resident bases", n, bases->n);
  if (n && !scalars) return fail(CURDLE_EINVAL, "scalars null with n = %zu", n);
  return CURDLE_OK;
}
}  // namespace curdle_api

extern "C" int curdle_msm_g1_dbases_windows(const curdle_dbases* bases, const void* d_scalars, size_t n, int window_bits,
                                            int win_begin, int win_end, uint64_t out_jac[18]) {
  int rc = dbases_msm_args(bases, d_scalars, n, out_jac);
  if (rc) return rc;
  if (n == 0) {
    set_out_infinity(out_jac);
    return CURDLE_OK;
  }
  curdle_dbases* set = const_cast<curdle_dbases*>(bases);
  void* d28 = nullptr;
  if ((rc = dbases_acquire(cur(), set, &d28))) return rc;  // this context's copy, made on first use
  const uint32_t off[2] = {0, (uint32_t)n};
  rc = run_device(nullptr, d_scalars, off, 1, window_bits, win_begin, win_end, out_jac, nullptr, d28);
  dbases_release(set);
  return rc;
}

extern "C" int curdle_msm_g1_dbases(const curdle_dbases* bases, const void* d_scalars, size_t n, uint64_t out_jac[18]) {
  return curdle_msm_g1_dbases_windows(bases, d_scalars, n, 0, 0, -1, out_jac);
}

// ... with the scalars in HOST memory (what msmaccumulator.Verify holds): 32 bytes per pair cross
// PCIe instead of 128.
extern "C" int curdle_msm_g1_dbases_host(const curdle_dbases* bases, const uint64_t* scalars, size_t n, uint64_t out_jac[18]) {
  int rc = dbases_msm_args(bases, scalars, n, out_jac);
  if (rc) return rc;
  if (n == 0) {
    set_out_infinity(out_jac);
    return CURDLE_OK;
  }
  Ctx& cx = cur();
  curdle_dbases* set = const_cast<curdle_dbases*>(bases);
  void* d28 = nullptr;
  if ((rc = dbases_acquire(cx, set, &d28))) return rc;
  int idx;
  rc = acquire_slot(cx, true, &idx);
  if (rc) {
    dbases_release(set);
    return rc;
  }
  Slot& S = cx.slots[idx];
  auto body = [&]() -> int {
    HIP_TRY(hipSetDevice(cx.device));
    int r;
    if ((r = ensure(S.scalars, n * 32))) return r;
    const SyncStreams st = sync_streams(cx, S);
    HIP_TRY(hipMemcpyAsync(S.scalars.p, scalars, n * 32, hipMemcpyHostToDevice, st.pre));
    const uint32_t off[2] = {0, (uint32_t)n};
    return run_passes(cx, S, nullptr, S.scalars.p, off, 1, 0, 0, -1, st.pre, st.main, st.tail, out_jac, d28);
  };
  rc = body();
  if (rc) drain_slot(cx, S);
  release_slot(cx, idx);
  dbases_release(set);
  return rc;
}

// The pipelined form (curdle_msm_wait finishes it); the set stays referenced until then.
extern "C" int curdle_msm_g1_dbases_submit(const curdle_dbases* bases, const void* d_scalars, size_t n, int window_bits,
                                           int win_begin, int win_end, int* ticket) {
  uint64_t dummy[18];
  int rc = dbases_msm_args(bases, d_scalars, n, dummy);
  if (rc) return rc;
  if (!ticket) return fail(CURDLE_EINVAL, "ticket is null");
  Ctx& cx = cur();
  curdle_dbases* set = const_cast<curdle_dbases*>(bases);
  void* d28 = nullptr;
  if ((rc = dbases_acquire(cx, set, &d28))) return rc;
  int idx;
  rc = acquire_slot(cx, false, &idx);
  if (rc) {
    dbases_release(set);
    return rc;
  }
  Slot& S = cx.slots[idx];
  hipError_t he = hipSetDevice(cx.device);
  if (he != hipSuccess) {
    release_slot(cx, idx);
    dbases_release(set);
    return fail(CURDLE_EHIP, "hipSetDevice: %s", hipGetErrorString(he));
  }
  const uint32_t off[2] = {0, (uint32_t)n};
  const unsigned seq = cx.submit_count.fetch_add(1, std::memory_order_relaxed);
  const unsigned turn = seq % (unsigned)cx.main_streams;
  hipStream_t main = turn == 0 ? cx.main_stream : cx.main_extra[turn - 1];
  const bool partial = win_begin > 0 || (win_end >= 0 && win_end < curdle_msm_num_windows(n, window_bits));
  hipStream_t pre = partial && cx.pre_streams == 2 && (seq & 1u) ? cx.pre_stream2 : cx.pre_stream;
  rc = enqueue_slot(cx, S, nullptr, d_scalars, off, 1, window_bits, win_begin, win_end, pre, main, S.stream,
                    /*latency_mode=*/false, false, 1, false, nullptr, d28);
  if (rc) {
    drain_slot(cx, S);
    release_slot(cx, idx);
    dbases_release(set);
    return rc;
  }
  S.held_bases = set;
  *ticket = make_ticket(cx, idx, S.gen);
  return CURDLE_OK;
}

extern "C" int curdle_dacc_begin(const curdle_dbases* crs, const uint64_t* inst_points, size_t n_inst, curdle_dacc** out) {
  Ctx& cx = cur();
  if (!crs || !out || (n_inst && !inst_points)) return fail(CURDLE_EINVAL, "null argument");
  *out = nullptr;
  if (crs->n + n_inst + CURDLE_DACC_MAX_EXTRA > ((size_t)1 << 27)) return fail(CURDLE_EINVAL, "too many bases");
  // this context's copy of the set (made on first use), held until the accumulation ends
  curdle_dbases* set = const_cast<curdle_dbases*>(crs);
  void* crs28 = nullptr;
  int rc = dbases_acquire(cx, set, &crs28);
  if (rc) return rc;
  int idx;
  rc = acquire_slot(cx, true, &idx);
  if (rc) {
    dbases_release(set);
    return rc;
  }
  Slot& S = cx.slots[idx];
  auto body = [&]() -> int {
    HIP_TRY(hipSetDevice(cx.device));
    const size_t cap = crs->n + n_inst + CURDLE_DACC_MAX_EXTRA;
    int r;
    // sized for the whole accumulation now: the MSM pipeline's own ensure() must not move them later
    if ((r = ensure(S.points28, 2 * cap * kA28Bytes))) return r;  // two records per base: P, phi(P)
    if ((r = ensure(S.scalars, cap * 32))) return r;
    if ((r = ensure(S.points, (n_inst + CURDLE_DACC_MAX_EXTRA) * 96))) return r;
    if ((r = ensure_pinned(S, 0, n_inst * 96))) return r;
    if (crs->n)
      HIP_TRY(hipMemcpyAsync(S.points28.p, crs28, 2 * crs->n * kA28Bytes, hipMemcpyDeviceToDevice, S.stream));
    if (n_inst) {
      memcpy(S.h_stage[0], inst_points, n_inst * 96);  // pinned: the copy below is truly asynchronous
      HIP_TRY(hipMemcpyAsync(S.points.p, S.h_stage[0], n_inst * 96, hipMemcpyHostToDevice, S.stream));
      HIP_TRY(launch_convert_points_raw(S.points.p, (uint32_t)n_inst, (char*)S.points28.p + 2 * crs->n * kA28Bytes, S.stream));
    }
    return CURDLE_OK;
  };
  rc = body();
  curdle_dacc* a = rc ? nullptr : new (std::nothrow) curdle_dacc();
  if (!a) {
    (void)hipStreamSynchronize(S.stream);
    release_slot(cx, idx);
    dbases_release(set);
    return rc ? rc : fail(CURDLE_ENOMEM, "out of memory");
  }
  a->ctx = &cx;
  a->slot = idx;
  a->crs = set;
  a->n_crs = crs->n;
  a->n_inst = n_inst;
  *out = a;
  return CURDLE_OK;
}

namespace curdle_api {
// the end of an accumulation, however it ends: the slot and the base set go back
void dacc_end(curdle_dacc* acc) {
  release_slot(*acc->ctx, acc->slot);
  dbases_release(acc->crs);
  delete acc;
}
}  // namespace curdle_api

extern "C" void curdle_dacc_abort(curdle_dacc* acc) {
  if (!acc) return;
  Ctx& cx = *acc->ctx;
  (void)hipSetDevice(cx.device);
  (void)hipStreamSynchronize(cx.slots[acc->slot].stream);
  dacc_end(acc);
}

namespace curdle_api {
int dacc_submit_impl(curdle_dacc* acc, const curdle_dacc_check* checks, size_t n_checks, const uint64_t* pool, size_t pool_len,
                     const uint64_t* extra_points, const uint64_t* extra_scalars, size_t n_extra, uint64_t* export_scalars,
                     bool queued);
}
// The asynchronous form: what a batch worker queues before it goes on verifying (light on the host, see make_plan).
extern "C" int curdle_dacc_submit(curdle_dacc* acc, const curdle_dacc_check* checks, size_t n_checks, const uint64_t* pool,
                                  size_t pool_len, const uint64_t* extra_points, const uint64_t* extra_scalars,
                                  size_t n_extra, uint64_t* export_scalars) {
  return dacc_submit_impl(acc, checks, n_checks, pool, pool_len, extra_points, extra_scalars, n_extra, export_scalars, true);
}
namespace curdle_api {
int dacc_submit_impl(curdle_dacc* acc, const curdle_dacc_check* checks, size_t n_checks, const uint64_t* pool, size_t pool_len,
                     const uint64_t* extra_points, const uint64_t* extra_scalars, size_t n_extra, uint64_t* export_scalars,
                     bool queued) {
  if (!acc) return fail(CURDLE_EINVAL, "null accumulator");
  if (acc->submitted) return fail(CURDLE_EINVAL, "accumulation already submitted");
  Ctx& cx = *acc->ctx;
  const int idx = acc->slot;
  Slot& S = cx.slots[idx];
  const size_t n_crs = acc->n_crs, n_inst = acc->n_inst, n_res = n_crs + n_inst, n = n_res + n_extra;
  auto body = [&]() -> int {
    if ((n_checks && !checks) || (pool_len && !pool) || (n_extra && (!extra_points || !extra_scalars)))
      return fail(CURDLE_EINVAL, "null argument");
    if (n_extra > CURDLE_DACC_MAX_EXTRA) return fail(CURDLE_EINVAL, "%zu loose bases exceed CURDLE_DACC_MAX_EXTRA", n_extra);
    // the descriptions come from the caller: every offset is checked before a kernel reads through it
    for (size_t c = 0; c < n_checks; c++) {
      const curdle_dacc_check& k = checks[c];
      if (k.kind > CURDLE_VEC_FOLD_POW || k.nseg > CURDLE_DACC_MAX_SEGS || k.m > 31)
        return fail(CURDLE_EINVAL, "check %zu: malformed description", c);
      if (k.weight_off >= pool_len || k.alpha_off >= pool_len || (size_t)k.tail_off + k.n_tail > pool_len ||
          (k.kind >= CURDLE_VEC_FOLD && (size_t)k.gammas_off + k.m > pool_len) ||
          (k.kind == CURDLE_VEC_FOLD_POW && k.q_off >= pool_len))
        return fail(CURDLE_EINVAL, "check %zu: offset outside the pool", c);
      if (k.kind >= CURDLE_VEC_FOLD && k.n_struct > ((uint64_t)1 << k.m))
        return fail(CURDLE_EINVAL, "check %zu: more structured elements than 2^m", c);
      for (uint32_t s = 0; s < k.nseg; s++) {
        const size_t set_n = k.seg[s].set == CURDLE_SET_CRS ? n_crs : n_inst;
        if (k.seg[s].set > CURDLE_SET_INST || (size_t)k.seg[s].first + k.seg[s].len > set_n ||
            (size_t)k.seg[s].vec_first + k.seg[s].len > (size_t)k.n_struct + k.n_tail)
          return fail(CURDLE_EINVAL, "check %zu: segment %u out of range", c, s);
      }
    }
    HIP_TRY(hipSetDevice(cx.device));
    acc->export_scalars = export_scalars;
    acc->n_total = n;
    if (n == 0) {
      acc->submitted = true;
      return CURDLE_OK;
    }
    // one pinned block: checks | pool | extra points | extra scalars
    const size_t o_pool = (n_checks * sizeof(curdle_dacc_check) + 31) & ~(size_t)31;
    const size_t o_xp = o_pool + pool_len * 32;
    const size_t o_xs = o_xp + n_extra * 96;
    const size_t bytes = o_xs + n_extra * 32;
    int r;
    if ((r = ensure_pinned(S, 1, bytes + n_res * 32))) return r;
    if ((r = ensure(S.job, bytes))) return r;
    char* h = (char*)S.h_stage[1];
    if (n_checks) memcpy(h, checks, n_checks * sizeof(curdle_dacc_check));
    if (pool_len) memcpy(h + o_pool, pool, pool_len * 32);
    if (n_extra) {
      memcpy(h + o_xp, extra_points, n_extra * 96);
      memcpy(h + o_xs, extra_scalars, n_extra * 32);
    }
    hipStream_t st = S.stream;
    HIP_TRY(hipMemcpyAsync(S.job.p, h, bytes, hipMemcpyHostToDevice, st));
    char* dj = (char*)S.job.p;
    // Small jobs (a verification's 1,268 + loose pairs; always below the two-level plans): the loose bases' conversion,
    // the slot scalars and the recoding in ONE launch (k_dacc_front) instead of four operations on the stream -- the
    // front of a verification is bound by the host's launches.  Knob FRONT=0: the separate launches.
    const bool fused = n <= 16384;
    if (!fused) {
      if (n_extra) {
        HIP_TRY(launch_convert_points_raw(dj + o_xp, (uint32_t)n_extra, (char*)S.points28.p + 2 * n_res * kA28Bytes, st));
        HIP_TRY(hipMemcpyAsync((char*)S.scalars.p + n_res * 32, dj + o_xs, n_extra * 32, hipMemcpyDeviceToDevice, st));
      }
      HIP_TRY(launch_dacc_scalars(dj, (uint32_t)n_checks, dj + o_pool, (uint32_t)pool_len, (uint32_t)n_crs, (uint32_t)n_inst, S.scalars.p, st));
      if (export_scalars && n_res)
        HIP_TRY(hipMemcpyAsync(h + bytes, S.scalars.p, n_res * 32, hipMemcpyDeviceToHost, st));
    }
    DaccFront df;
    df.d_checks = dj;
    df.d_pool = dj + o_pool;
    df.d_extra_points = dj + o_xp;
    df.d_extra_scalars = dj + o_xs;
    df.d_scalars_out = export_scalars ? S.scalars.p : nullptr;
    df.n_checks = (uint32_t)n_checks;
    df.pool_len = (uint32_t)pool_len;
    df.n_crs = (uint32_t)n_crs;
    df.n_inst = (uint32_t)n_inst;
    df.n_extra = (uint32_t)n_extra;
    const uint32_t off[2] = {0, (uint32_t)n};
    if ((r = enqueue_slot(cx, S, nullptr, S.scalars.p, off, 1, 0, 0, -1, st, st, st, /*latency_mode=*/true,
                          /*points28_ready=*/true, 1, false, nullptr, nullptr, /*light_host=*/queued, true, fused ? &df : nullptr)))
      return r;
    if (fused && export_scalars && n_res)  // (tests: behind the whole call on the stream, read at the wait)
      HIP_TRY(hipMemcpyAsync(h + bytes, S.scalars.p, n_res * 32, hipMemcpyDeviceToHost, st));
    acc->export_off = bytes;
    acc->submitted = true;
    return CURDLE_OK;
  };
  int rc = body();
  if (rc) {  // a failed submission ends the accumulation, like a failed run
    (void)hipStreamSynchronize(S.stream);
    dacc_end(acc);
  }
  return rc;
}
}  // namespace curdle_api

extern "C" int curdle_dacc_poll(curdle_dacc* acc, int* done) {
  if (!acc || !done) return fail(CURDLE_EINVAL, "null argument");
  if (!acc->submitted) return fail(CURDLE_EINVAL, "accumulation not submitted");
  Ctx& cx = *acc->ctx;
  *done = 1;
  if (acc->n_total == 0) return CURDLE_OK;
  Slot& S = cx.slots[acc->slot];
  (void)hipSetDevice(cx.device);
  const hipError_t e = hipStreamQuery(S.run_stream);
  if (e == hipErrorNotReady) {
    *done = 0;
    return CURDLE_OK;
  }
  if (e != hipSuccess) return fail(CURDLE_EHIP, "dacc poll: %s", hipGetErrorString(e));
  return CURDLE_OK;
}

extern "C" int curdle_dacc_wait(curdle_dacc* acc, uint64_t out_jac[18]) {
  if (!acc) return fail(CURDLE_EINVAL, "null accumulator");
  if (!acc->submitted) return fail(CURDLE_EINVAL, "accumulation not submitted");
  Ctx& cx = *acc->ctx;
  const int idx = acc->slot;
  Slot& S = cx.slots[idx];
  auto body = [&]() -> int {
    if (!out_jac) return fail(CURDLE_EINVAL, "null argument");
    if (acc->n_total == 0) {
      set_out_infinity(out_jac);
      return CURDLE_OK;
    }
    HIP_TRY(hipSetDevice(cx.device));
    int r;
    if ((r = finish_slot(cx, S, out_jac))) return r;
    const size_t n_res = acc->n_crs + acc->n_inst;
    if (acc->export_scalars && n_res) memcpy(acc->export_scalars, (char*)S.h_stage[1] + acc->export_off, n_res * 32);
    return CURDLE_OK;
  };
  int rc = body();
  if (rc) (void)hipStreamSynchronize(S.stream);
  dacc_end(acc);
  return rc;
}

extern "C" int curdle_dacc_run(curdle_dacc* acc, const curdle_dacc_check* checks, size_t n_checks, const uint64_t* pool,
                               size_t pool_len, const uint64_t* extra_points, const uint64_t* extra_scalars,
                               size_t n_extra, uint64_t out_jac[18], uint64_t* export_scalars) {
  if (!acc) return fail(CURDLE_EINVAL, "null accumulator");
  if (!out_jac) {
    curdle_dacc_abort(acc);
    return fail(CURDLE_EINVAL, "null argument");
  }
  int rc = dacc_submit_impl(acc, checks, n_checks, pool, pool_len, extra_points, extra_scalars, n_extra, export_scalars, false);
  if (rc) return rc;  // the submission already ended the accumulation
  return curdle_dacc_wait(acc, out_jac);
}
