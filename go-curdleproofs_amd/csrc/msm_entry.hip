// The MSM entry points of the C ABI (include/curdle_msm.h): host slices, device-resident inputs, window ranges, the
// submit / wait pair, the kept-bases cache behind CURDLE_MSM_BASES_UNCHANGED, several devices behind one call, batches
// and shared-scalar calls.  (Part of msm_api.hip until round 6.)
#include "msm_internal.h"

namespace curdle_api {
// One MSM from host buffers on the calling thread's context.
int msm_host_one_device(const uint64_t* points, const uint64_t* scalars, size_t n, uint64_t out_jac[18], bool glv = true) {
  if (n == 0) {
    set_out_infinity(out_jac);
    return CURDLE_OK;
  }
  if (n >= kHostChunkMin && knobs::get(knobs::HOST_CHUNKS) != 1) return run_host_chunked(points, scalars, n, out_jac, glv);
  const uint32_t off[2] = {0, (uint32_t)n};
  return run_host(points, scalars, off, 1, out_jac, glv);
}

// share(d, out18) runs on the host thread of context d (whose current context is d) for every
// d < D, all at once; the D partial sums are added on the host (what curdle_g1_sum does).  The
// exchange of north_star's "RCCL reduce of 8 partial points" inside ONE process: 144 bytes per
// device through host memory, no collective.
int run_on_devices(int D, const std::function<int(int, uint64_t*)>& share, uint64_t out_jac[18]) {
  struct Part {
    uint64_t jac[18];
    int rc = CURDLE_OK;
    char err[256] = "";
  };
  std::vector<Part> parts((size_t)D);
  std::mutex mu;
  std::condition_variable cv;
  int left = D;
  // the devices' host threads are read under the configuration mutex and pinned for the duration of the
  // call: curdle_shutdown meanwhile returns CURDLE_EBUSY instead of deleting them underneath (review of round 3)
  struct InFlight {
    bool on = false;
    ~InFlight() {
      if (on) g_multi_calls.fetch_sub(1, std::memory_order_acq_rel);
    }
  } in_flight;
  {
    std::lock_guard<std::mutex> cfg(g_cfg_mu);
    if (D != g_ndev.load(std::memory_order_acquire)) return fail(CURDLE_EBUSY, "the device configuration changed under the call");
    for (int d = 0; d < D; d++)  // before anything is posted: the jobs below refer to this frame
      if (!g_ctxs[d].worker) return fail(CURDLE_EINVAL, "context %d has no host thread (curdle_init_devices was not called)", d);
    g_multi_calls.fetch_add(1, std::memory_order_acq_rel);
    in_flight.on = true;
  }
  for (int d = 0; d < D; d++) {
    DevWorker* w = g_ctxs[d].worker;
    w->post([&, d] {
      Part& p = parts[(size_t)d];
      try {
        p.rc = share(d, p.jac);
        if (p.rc) snprintf(p.err, sizeof(p.err), "%s", g_err);  // the worker's thread-local text
      } catch (const std::bad_alloc&) {  // the std::vector allocations of finish_slot / run_passes / run_host_chunked
        p.rc = CURDLE_ENOMEM;
        snprintf(p.err, sizeof(p.err), "out of host memory");
      } catch (const std::exception& e) {
        p.rc = CURDLE_EHIP;
        snprintf(p.err, sizeof(p.err), "%s", e.what());
      } catch (...) {  // anything else: the decrement below must run, or the caller waits for ever (review of round 4)
        p.rc = CURDLE_EHIP;
        snprintf(p.err, sizeof(p.err), "unknown exception in a device's host thread");
      }
      std::lock_guard<std::mutex> g(mu);
      if (--left == 0) cv.notify_one();
    });
  }
  {
    std::unique_lock<std::mutex> g(mu);
    cv.wait(g, [&] { return left == 0; });
  }
  G1XYZZ total;
  g1_set_inf(total);
  for (int d = 0; d < D; d++) {
    if (parts[(size_t)d].rc) return fail(parts[(size_t)d].rc, "device %d: %s", d, parts[(size_t)d].err);
    G1Jac j;
    memcpy(&j, parts[(size_t)d].jac, sizeof(j));
    G1XYZZ t;
    g1_from_jac(t, j);
    g1_add(total, t);
  }
  g1_to_canonical_jac(out_jac, total);
  return CURDLE_OK;
}

// contiguous, as-even-as-possible split of [0, n) over D parts (curdlemsm/distributed.py window_partition)
inline void even_range(size_t n, int D, int d, size_t* lo, size_t* hi) {
  const size_t base = n / (size_t)D, extra = n % (size_t)D;
  *lo = (size_t)d * base + ((size_t)d < extra ? (size_t)d : extra);
  *hi = *lo + base + ((size_t)d < extra ? 1 : 0);
}

// Below this many pairs a host-buffer MSM stays on the calling thread's device: the hand-off to
// D host threads and D separate small MSMs (each a fixed ~0.3 ms chain) cost more than they save.
size_t multi_device_min() {
  return knobs::is_set(knobs::MULTI_DEVICE_MIN) ? (size_t)knobs::get(knobs::MULTI_DEVICE_MIN) : (size_t)1 << 16;
}
}  // namespace curdle_api

extern "C" int curdle_msm_g1_ex(const uint64_t* points, const uint64_t* scalars, size_t n, unsigned flags, uint64_t out_jac[18]) {
  if (!out_jac) return fail(CURDLE_EINVAL, "out_jac is null");
  if (flags & ~(unsigned)CURDLE_MSM_ANY_CURVE_POINT)
    return fail(CURDLE_EINVAL, "flags 0x%x: a host-buffer MSM takes CURDLE_MSM_ANY_CURVE_POINT only", flags);
  const bool glv = !(flags & CURDLE_MSM_ANY_CURVE_POINT);
  if (n == 0) {
    set_out_infinity(out_jac);
    return CURDLE_OK;
  }
  if (!points || !scalars) return fail(CURDLE_EINVAL, "points/scalars null with n = %zu", n);
  if (n > ((size_t)1 << 27)) return fail(CURDLE_EINVAL, "n = %zu exceeds the supported 2^27 pairs", n);
  const int D = g_ndev.load(std::memory_order_acquire);
  if (D > 1 && n >= multi_device_min() && !tl_selected) {  // a thread that selected a device (a batch shard, OnDevice) keeps its MSM there
    // Several GPUs behind this one call (curdle_init_devices): by POINT RANGES -- from host buffers
    // the copy is most of the call (128 MiB at N = 2^20 over one GPU's PCIe link), and only a point
    // range divides it: every device copies its own n / D pairs over its own link and runs all
    // windows over them (a window range would send all n pairs to every device).
    g_spread_calls.fetch_add(1, std::memory_order_relaxed);
    return run_on_devices(D, [&](int d, uint64_t* part) {
      size_t lo, hi;
      even_range(n, D, d, &lo, &hi);
      return msm_host_one_device(points + 12 * lo, scalars + 4 * lo, hi - lo, part, glv);
    }, out_jac);
  }
  return msm_host_one_device(points, scalars, n, out_jac, glv);
}

extern "C" int curdle_msm_g1(const uint64_t* points, const uint64_t* scalars, size_t n, uint64_t out_jac[18]) {
  return curdle_msm_g1_ex(points, scalars, n, 0, out_jac);
}

extern "C" int curdle_msm_g1_replicated(const void* const* d_points, const void* const* d_scalars, size_t n, int split,
                                        uint64_t out_jac[18]) {
  if (!out_jac) return fail(CURDLE_EINVAL, "out_jac is null");
  if (split < 0 || split > 2) return fail(CURDLE_EINVAL, "split must be 0 (library's choice), 1 (windows) or 2 (points)");
  if (n == 0) {
    set_out_infinity(out_jac);
    return CURDLE_OK;
  }
  if (n > ((size_t)1 << 27)) return fail(CURDLE_EINVAL, "n = %zu exceeds the supported 2^27 pairs", n);
  const int D = g_ndev.load(std::memory_order_acquire);
  if (!d_points || !d_scalars) return fail(CURDLE_EINVAL, "points/scalars null with n = %zu", n);
  for (int d = 0; d < D; d++)
    if (!d_points[d] || !d_scalars[d]) return fail(CURDLE_EINVAL, "device %d: null input pointer", d);
  if (D == 1) {
    const uint32_t off[2] = {0, (uint32_t)n};
    return run_device(d_points[0], d_scalars[0], off, 1, 0, 0, -1, out_jac, nullptr);
  }
  // which partition pays at which size: DESIGN.md section 5 (per-rank step times on one MI355X)
  if (split == 0) split = n >= ((size_t)1 << 22) ? 2 : 1;
  const int c = choose_window_bits(n);
  uint8_t bits[kMaxWindows];
  const int W = window_widths(c, bits);
  return run_on_devices(D, [&](int d, uint64_t* part) {
    if (split == 1) {  // windows [wb, we) of the plan for all n pairs; a device beyond the last window adds infinity
      size_t wb, we;
      even_range((size_t)W, D, d, &wb, &we);
      const uint32_t off[2] = {0, (uint32_t)n};
      return run_device(d_points[d], d_scalars[d], off, 1, c, (int)wb, (int)we, part, nullptr);
    }
    size_t lo, hi;
    even_range(n, D, d, &lo, &hi);
    const uint32_t off[2] = {0, (uint32_t)(hi - lo)};
    return run_device((const char*)d_points[d] + lo * 96, (const char*)d_scalars[d] + lo * 32, off, 1, 0, 0, -1, part,
                      nullptr);
  }, out_jac);
}

namespace curdle_api {
// The converted copy of (d_points, n) on this context, made now if there is none.  On success *entry >= 0 names
// the cache entry (one user reference taken: bases_cache_release gives it back), *d28 its records and *ready the
// event behind its conversion; *entry = -1 with CURDLE_OK means every entry is in use by calls in flight: the
// caller converts per call, as without the flag.
int bases_cache_acquire(Ctx& cx, const void* d_points, size_t n, int* entry, const void** d28, hipEvent_t* ready) {
  *entry = -1;
  std::lock_guard<std::mutex> g(cx.mu);
  int rc = init_default_locked(cx);
  if (rc) return rc;
  HIP_TRY(hipSetDevice(cx.device));
  int victim = -1;
  for (int i = 0; i < Ctx::kBaseCache; i++) {
    Ctx::BaseCache& bc = cx.bcache[i];
    if (bc.key == d_points && bc.n == n) {
      bc.users++;
      bc.stamp = ++cx.bstamp;
      *entry = i;
      *d28 = bc.buf.p;
      *ready = bc.ready;
      return CURDLE_OK;
    }
    if (bc.users == 0 && (victim < 0 || bc.stamp < cx.bcache[victim].stamp)) victim = i;
  }
  if (victim < 0) return CURDLE_OK;
  Ctx::BaseCache& bc = cx.bcache[victim];
  bc.key = nullptr;
  // (under the context's mutex: a second caller with the same key must find the entry complete, its event recorded;
  // the allocation below happens once per base array)
  if ((rc = ensure(bc.buf, 2 * n * kA28Bytes))) return rc;
  HIP_TRY(launch_convert_points_raw(d_points, (uint32_t)n, bc.buf.p, cx.util_stream));
  HIP_TRY(hipEventRecord(bc.ready, cx.util_stream));
  bc.key = d_points;
  bc.n = n;
  bc.users = 1;
  bc.stamp = ++cx.bstamp;
  *entry = victim;
  *d28 = bc.buf.p;
  *ready = bc.ready;
  return CURDLE_OK;
}
void bases_cache_release(Ctx& cx, int entry) {
  if (entry < 0) return;
  std::lock_guard<std::mutex> g(cx.mu);
  if (cx.bcache[entry].users > 0) cx.bcache[entry].users--;
}
int check_flags(unsigned flags) {
  if (flags & ~(unsigned)(CURDLE_MSM_ANY_CURVE_POINT | CURDLE_MSM_BASES_UNCHANGED)) return fail(CURDLE_EINVAL, "unknown flags 0x%x", flags);
  return CURDLE_OK;
}
}  // namespace curdle_api

extern "C" int curdle_msm_forget_bases(const void* d_points) {
  Ctx& cx = cur();
  std::lock_guard<std::mutex> g(cx.mu);
  for (auto& bc : cx.bcache)
    if (bc.key == d_points) bc.key = nullptr;  // calls in flight keep reading the copy; nobody finds it any more
  return CURDLE_OK;
}

extern "C" int curdle_msm_g1_device_windows_ex(const void* d_points, const void* d_scalars, size_t n, int window_bits,
                                               int win_begin, int win_end, unsigned flags, uint64_t out_jac[18], void* stream) {
  if (!out_jac) return fail(CURDLE_EINVAL, "out_jac is null");
  int rc = check_flags(flags);
  if (rc) return rc;
  if (n && (!d_points || !d_scalars)) return fail(CURDLE_EINVAL, "points/scalars null with n = %zu", n);
  if (n > ((size_t)1 << 27)) return fail(CURDLE_EINVAL, "n = %zu exceeds the supported 2^27 pairs", n);
  const bool glv = !(flags & CURDLE_MSM_ANY_CURVE_POINT);
  const uint32_t off[2] = {0, (uint32_t)n};
  Ctx& cx = cur();
  int entry = -1;
  const void* d28 = nullptr;
  hipEvent_t ready = nullptr;
  if ((flags & CURDLE_MSM_BASES_UNCHANGED) && n && (rc = bases_cache_acquire(cx, d_points, n, &entry, &d28, &ready))) return rc;
  rc = run_device(d_points, d_scalars, off, 1, window_bits, win_begin, win_end, out_jac, stream, entry >= 0 ? d28 : nullptr, glv,
                  entry >= 0 ? ready : nullptr);
  bases_cache_release(cx, entry);
  return rc;
}

extern "C" int curdle_msm_g1_device_ex(const void* d_points, const void* d_scalars, size_t n, unsigned flags, uint64_t out_jac[18],
                                       void* stream) {
  return curdle_msm_g1_device_windows_ex(d_points, d_scalars, n, 0, 0, -1, flags, out_jac, stream);
}

extern "C" int curdle_msm_g1_device_windows(const void* d_points, const void* d_scalars, size_t n, int window_bits,
                                            int win_begin, int win_end, uint64_t out_jac[18], void* stream) {
  return curdle_msm_g1_device_windows_ex(d_points, d_scalars, n, window_bits, win_begin, win_end, 0, out_jac, stream);
}

extern "C" int curdle_msm_g1_device(const void* d_points, const void* d_scalars, size_t n, uint64_t out_jac[18],
                                    void* stream) {
  return curdle_msm_g1_device_windows(d_points, d_scalars, n, 0, 0, -1, out_jac, stream);
}

// Asynchronous pair: submit enqueues all GPU phases of one MSM (or one window range)
// on a free workspace slot and returns at once; wait blocks for it and finishes on
// the host.  Up to CURDLE_MSM_SLOTS calls can be in flight.
extern "C" int curdle_msm_g1_device_submit_ex(const void* d_points, const void* d_scalars, size_t n, int window_bits,
                                              int win_begin, int win_end, unsigned flags, int* ticket) {
  Ctx& cx = cur();
  if (!ticket) return fail(CURDLE_EINVAL, "ticket is null");
  int rc = check_flags(flags);
  if (rc) return rc;
  if (n && (!d_points || !d_scalars)) return fail(CURDLE_EINVAL, "points/scalars null with n = %zu", n);
  if (n > ((size_t)1 << 27)) return fail(CURDLE_EINVAL, "n = %zu exceeds the supported 2^27 pairs", n);
  const bool glv = !(flags & CURDLE_MSM_ANY_CURVE_POINT);
  int c_checked;  // before a slot or a cache reference is held, and before window_widths() below
  if ((rc = checked_window_bits(n, window_bits, &c_checked))) return rc;
  int idx;
  rc = acquire_slot(cx, false, &idx);
  if (rc) return rc;
  Slot& S = cx.slots[idx];
  hipError_t he = hipSetDevice(cx.device);
  if (he != hipSuccess) {
    release_slot(cx, idx);
    return fail(CURDLE_EHIP, "hipSetDevice: %s", hipGetErrorString(he));
  }
  int entry = -1;
  const void* d28 = nullptr;
  hipEvent_t ready = nullptr;
  if ((flags & CURDLE_MSM_BASES_UNCHANGED) && n && (rc = bases_cache_acquire(cx, d_points, n, &entry, &d28, &ready))) {
    release_slot(cx, idx);
    return rc;
  }
  const uint32_t off[2] = {0, (uint32_t)n};
  const unsigned seq = cx.submit_count.fetch_add(1, std::memory_order_relaxed);
  const unsigned turn = seq % (unsigned)cx.main_streams;
  hipStream_t main = turn == 0 ? cx.main_stream : cx.main_extra[turn - 1];
  uint8_t wb[kMaxWindows];
  const int W = window_widths(c_checked, wb, glv ? kScalarBits : kScalarBitsNoGlv);
  const bool partial = win_begin > 0 || (win_end >= 0 && win_end < W);
  hipStream_t pre = partial && cx.pre_streams == 2 && (seq & 1u) ? cx.pre_stream2 : cx.pre_stream;
  if (entry >= 0 && (he = hipStreamWaitEvent(main, ready, 0)) != hipSuccess)  // the accumulation is what reads the copy
    rc = fail(CURDLE_EHIP, "hipStreamWaitEvent: %s", hipGetErrorString(he));
  if (!rc)
    rc = enqueue_slot(cx, S, d_points, d_scalars, off, 1, window_bits, win_begin, win_end, pre, main, S.stream,
                      /*latency_mode=*/false, false, 1, false, nullptr, entry >= 0 ? d28 : nullptr, false, glv);
  if (rc) {
    drain_slot(cx, S);
    release_slot(cx, idx);
    bases_cache_release(cx, entry);
    return rc;
  }
  S.held_cache = entry;
  *ticket = make_ticket(cx, idx, S.gen);
  return CURDLE_OK;
}

extern "C" int curdle_msm_g1_device_submit(const void* d_points, const void* d_scalars, size_t n, int window_bits,
                                           int win_begin, int win_end, int* ticket) {
  return curdle_msm_g1_device_submit_ex(d_points, d_scalars, n, window_bits, win_begin, win_end, 0, ticket);
}


extern "C" int curdle_msm_wait(int ticket, uint64_t out_jac[18]) {
  Ctx* cp = ticket_ctx(ticket);
  if (!cp || ticket_index(ticket) >= kSlots || !out_jac) return fail(CURDLE_EINVAL, "bad ticket or null output");
  Ctx& cx = *cp;
  const int idx = ticket_index(ticket);
  {
    std::lock_guard<std::mutex> g(cx.mu);
    Slot& S = cx.slots[idx];
    if (!cx.inited || !S.busy || S.claimed || (S.gen & 0x7fffffu) != ticket_gen(ticket))
      return fail(CURDLE_EINVAL, "ticket %d is not in flight (stale or already waited for)", ticket);
    S.claimed = true;
  }
  hipError_t he = hipSetDevice(cx.device);
  if (he != hipSuccess) {
    std::lock_guard<std::mutex> g(cx.mu);
    cx.slots[idx].claimed = false;  // the call is still in flight: the caller may wait again
    return fail(CURDLE_EHIP, "hipSetDevice: %s", hipGetErrorString(he));
  }
  int rc = finish_slot(cx, cx.slots[idx], out_jac);
  if (rc) drain_slot(cx, cx.slots[idx]);
  curdle_dbases* held = cx.slots[idx].held_bases;  // a resident base set the call read from
  cx.slots[idx].held_bases = nullptr;
  const int cached = cx.slots[idx].held_cache;     // ... or a cached converted copy
  cx.slots[idx].held_cache = -1;
  release_slot(cx, idx);
  if (held) dbases_release_handle(held);
  bases_cache_release(cx, cached);
  return rc;
}

extern "C" int curdle_msm_free_slots(void) {
  Ctx& cx = cur();
  std::lock_guard<std::mutex> g(cx.mu);
  int n = 0;
  for (const Slot& S : cx.slots) n += S.busy ? 0 : 1;
  return n;
}

// k MSMs in one pass of the pipeline; inputs resident on the device.
extern "C" int curdle_msm_g1_batch_device(const void* d_points, const void* d_scalars, const size_t* offsets, size_t k,
                                          uint64_t* out_jac, void* stream) {
  if (!offsets || (k && !out_jac)) return fail(CURDLE_EINVAL, "null argument");
  if (k == 0) return CURDLE_OK;
  if (offsets[k] - offsets[0] > ((size_t)1 << 27)) return fail(CURDLE_EINVAL, "more than the supported 2^27 pairs");
  if (offsets[k] != offsets[0] && (!d_points || !d_scalars)) return fail(CURDLE_EINVAL, "points/scalars null");
  std::vector<uint32_t> off(k + 1);
  for (size_t j = 0; j <= k; j++) {
    if (j && offsets[j] < offsets[j - 1]) return fail(CURDLE_EINVAL, "offsets not monotone at %zu", j - 1);
    off[j] = (uint32_t)(offsets[j] - offsets[0]);
  }
  const char* dp = (const char*)d_points + offsets[0] * 96;
  const char* ds = (const char*)d_scalars + offsets[0] * 32;
  return run_device(dp, ds, off.data(), k, 0, 0, -1, out_jac, stream);
}

extern "C" int curdle_g1_sum(const uint64_t* jac_points, size_t k, uint64_t out_jac[18]) {
  if (!out_jac || (k && !jac_points)) return fail(CURDLE_EINVAL, "null argument");
  G1XYZZ acc;
  g1_set_inf(acc);
  for (size_t i = 0; i < k; i++) {
    G1Jac j;
    memcpy(&j, jac_points + 18 * i, sizeof(j));
    G1XYZZ t;
    g1_from_jac(t, j);
    g1_add(acc, t);
  }
  g1_to_canonical_jac(out_jac, acc);
  return CURDLE_OK;
}

extern "C" int curdle_msm_g1_batch(const uint64_t* points, const uint64_t* scalars, const size_t* offsets, size_t k,
                                   uint64_t* out_jac) {
  if (!offsets || (k && !out_jac)) return fail(CURDLE_EINVAL, "null argument");
  if (k == 0) return CURDLE_OK;
  for (size_t j = 0; j < k; j++)
    if (offsets[j + 1] < offsets[j]) return fail(CURDLE_EINVAL, "offsets not monotone at %zu", j);
  const size_t lo = offsets[0], n = offsets[k] - offsets[0];
  if (n == 0) {
    for (size_t j = 0; j < k; j++) set_out_infinity(out_jac + 18 * j);
    return CURDLE_OK;
  }
  if (!points || !scalars) return fail(CURDLE_EINVAL, "points/scalars null with n = %zu", n);
  if (n > ((size_t)1 << 27)) return fail(CURDLE_EINVAL, "n = %zu exceeds the supported 2^27 pairs", n);
  std::vector<uint32_t> off(k + 1);
  for (size_t j = 0; j <= k; j++) off[j] = (uint32_t)(offsets[j] - lo);
  return run_host(points + 12 * lo, scalars + 4 * lo, off.data(), k, out_jac);
}

// k base sets against ONE scalar vector (samemultiscalarargument.go:64-70: the same r against
// G, T, U; curdleproof.go:110,:114): the scalars are uploaded, recoded and bucket-sorted once;
// the accumulate kernel walks the one sorted index list once per base set (grid.y), and the
// reduce kernels read the shared fragment bookkeeping with a per-set fragment offset.
extern "C" int curdle_msm_g1_multi(const uint64_t* const* points_sets, size_t k, const uint64_t* scalars, size_t n,
                                   uint64_t* out_jac) {
  Ctx& cx = cur();
  if ((k && !out_jac) || (k && !points_sets)) return fail(CURDLE_EINVAL, "null argument");
  if (k == 0) return CURDLE_OK;
  if (n == 0) {
    for (size_t j = 0; j < k; j++) set_out_infinity(out_jac + 18 * j);
    return CURDLE_OK;
  }
  if (!scalars) return fail(CURDLE_EINVAL, "scalars null with n = %zu", n);
  if (k * n > ((size_t)1 << 27)) return fail(CURDLE_EINVAL, "k*n = %zu exceeds the supported 2^27 pairs", k * n);
  for (size_t j = 0; j < k; j++)
    if (!points_sets[j]) return fail(CURDLE_EINVAL, "points_sets[%zu] is null", j);
  int idx;
  int rc = acquire_slot(cx, true, &idx);
  if (rc) return rc;
  Slot& S = cx.slots[idx];
  auto body = [&]() -> int {
    HIP_TRY(hipSetDevice(cx.device));
    int r;
    if ((r = ensure(S.points, k * n * 96))) return r;
    if ((r = ensure(S.scalars, n * 32))) return r;
    const SyncStreams st = sync_streams(cx, S);
    for (size_t j = 0; j < k; j++)
      HIP_TRY(hipMemcpyAsync((char*)S.points.p + j * n * 96, points_sets[j], n * 96, hipMemcpyHostToDevice, st.pre));
    HIP_TRY(hipMemcpyAsync(S.scalars.p, scalars, n * 32, hipMemcpyHostToDevice, st.pre));
    const uint32_t off[2] = {0, (uint32_t)n};
    if ((r = enqueue_slot(cx, S, S.points.p, S.scalars.p, off, 1, 0, 0, -1, st.pre, st.main, st.tail,
                          /*latency_mode=*/true, /*points28_ready=*/false, /*sets=*/k)))
      return r;
    return finish_slot(cx, S, out_jac);
  };
  rc = body();
  if (rc) drain_slot(cx, S);
  release_slot(cx, idx);
  return rc;
}
