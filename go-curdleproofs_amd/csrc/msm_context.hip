// C ABI of libcurdlemsm.so (include/curdle_msm.h): context, workspace slots,
// phase sequencing, the host-side window combine and the accumulator / rand
// handles.
//
// There is deliberately no CPU implementation of the MSM behind these entry
// points: if the HIP runtime has no device, they fail with CURDLE_ENODEV.
//
// THIS FILE: the contexts (one per configured device), their workspace slots and buffers, the error text, device
// selection, initialisation and shutdown.  The plan is msm_plan.hip, the phase sequencing msm_enqueue.hip (+ msm_host_chunks.hip),
// the MSM entry points msm_entry.hip, resident bases and the device accumulator dbases_api.hip, point decoding
// decode_api.hip, the rest misc_api.hip.
#include "msm_internal.h"

// ---------------------------------------------------------------------------
// Errors
// ---------------------------------------------------------------------------
namespace curdle_api {
thread_local char g_err[256] = "";

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}
}  // namespace curdle_api

// for the other translation units of the library (host/proto_api.cpp)
extern "C" int curdle_set_last_error(int code, const char* msg) { return fail(code, "%s", msg); }



namespace curdle_api {
Ctx g_ctxs[kMaxDevices];
std::atomic<int> g_ndev{1};  // configured contexts: [0, g_ndev)
std::mutex g_cfg_mu;         // configuration (curdle_init_devices / curdle_shutdown)
std::atomic<int> g_multi_calls{0};  // calls that span the devices' host threads right now
std::atomic<unsigned long long> g_spread_calls{0};  // host-buffer MSMs that were spread over several devices, ever
thread_local int tl_dev = 0;
thread_local bool tl_selected = false;  // the thread called curdle_set_device: its host-buffer MSMs stay on that device
const bool g_ordinals_set = [] {
  for (int i = 0; i < kMaxDevices; i++) g_ctxs[i].ordinal = i;
  return true;
}();


// Capacity for a request of `bytes`: the next power of two up to 256 MiB (at least 64 KiB), an
// eighth of slack above.  hipFree waits for the whole device, so a buffer that creeps up with
// the request size (batch verification: groups of 2..32 proofs landing on eight slots in any
// order) stalls every MSM in flight each time it moves -- 8-11 ms per group where the MSM
// itself takes 1.2 ms.  With 288 GB of HBM the rounding costs nothing that matters.
size_t grow_size(size_t bytes) {
  if (bytes > ((size_t)256 << 20)) return bytes + bytes / 8 + 256;
  size_t c = (size_t)64 << 10;
  while (c < bytes) c <<= 1;
  return c;
}

int ensure(Buf& b, size_t bytes) {
  if (bytes <= b.cap) return CURDLE_OK;
  if (b.p) {
    HIP_TRY(hipFree(b.p));
    b.p = nullptr;
    b.cap = 0;
  }
  size_t want = grow_size(bytes);
  HIP_TRY(hipMalloc(&b.p, want));
  b.cap = want;
  return CURDLE_OK;
}

// The slot's pinned staging buffers (device accumulator, one-shot point decoding).
int ensure_pinned(Slot& S, int which, size_t bytes) {
  if (S.h_stage_cap[which] >= bytes) return CURDLE_OK;
  if (S.h_stage[which]) HIP_TRY(hipHostFree(S.h_stage[which]));
  S.h_stage[which] = nullptr;
  S.h_stage_cap[which] = 0;
  const size_t want = grow_size(bytes);
  HIP_TRY(hipHostMalloc(&S.h_stage[which], want, hipHostMallocDefault));
  S.h_stage_cap[which] = want;
  return CURDLE_OK;
}

int init_locked(Ctx& cx, int device) {
  if (cx.inited) {
    if (device != cx.device) return fail(CURDLE_EINVAL, "already initialised on device %d", cx.device);
    return CURDLE_OK;
  }
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0)
    return fail(CURDLE_ENODEV, "no HIP device visible (%s)", e == hipSuccess ? "count is 0" : hipGetErrorString(e));
  if (device < 0 || device >= ndev) return fail(CURDLE_EINVAL, "device %d out of range (%d visible)", device, ndev);
  HIP_TRY(hipSetDevice(device));
  HIP_TRY(hipStreamCreateWithFlags(&cx.util_stream, hipStreamNonBlocking));
  // only now: a failure before the first handle exists leaves the context exactly as it was (review of round 4: with
  // the id set first, a failed util_stream left cx.device on the failed id and nothing to tear down by)
  cx.device = device;  // from here on a failure leaves handles behind: curdle_init_devices tears them down
  HIP_TRY(hipStreamCreateWithFlags(&cx.h2d_stream, hipStreamNonBlocking));
  int prio_least = 0, prio_greatest = 0;
  HIP_TRY(hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest));
  HIP_TRY(hipStreamCreateWithPriority(&cx.main_stream, hipStreamNonBlocking, prio_least));
  if (knobs::is_set(knobs::MAIN_STREAMS)) {
    cx.main_streams = (int)knobs::get(knobs::MAIN_STREAMS);
    if (cx.main_streams < 1 || cx.main_streams > 4) cx.main_streams = 1;
  }
  // Only the streams that will be used: with more streams than hardware queues
  // (GPU_MAX_HW_QUEUES: 4 by default, 16 under bench.py) streams share queues, and which ones
  // do depends on the creation order -- a stream nobody launches on must not cost a queue.
  for (int i = 0; i + 1 < cx.main_streams; i++)
    HIP_TRY(hipStreamCreateWithPriority(&cx.main_extra[i], hipStreamNonBlocking, prio_least));
  HIP_TRY(hipStreamCreateWithPriority(&cx.pre_stream, hipStreamNonBlocking, prio_least));
  HIP_TRY(hipStreamCreateWithPriority(&cx.pre_stream2, hipStreamNonBlocking, prio_least));
  // Tail streams at normal priority: on ROCm 7.2 all high-priority streams of a process appear to
  // share one hardware queue, which serialises the tails of consecutive MSMs (measured in round 2:
  // 0.93 vs 0.84 ms per 2-window partial).
  const int tail_prio = prio_least;
  // (Confining the tail streams to 32 / 64 / 128 compute units with hipExtStreamCreateWithCUMask --
  // so that the latency-bound bucket reductions, whose waves sit on their SIMDs for 0.3-0.7 ms
  // and leave room for only one accumulate wave beside them, stop taking a wave slot on half the
  // chip -- was measured in round 3: the pipelined step went from 2.69 to 3.87 / 3.18 / 2.80 ms and
  // an 8-way rank's from 0.60 to 1.04 / 0.69 / 0.63: the confined reductions take 0.86 / 0.47 /
  // 0.29 ms and the pipeline waits for them.  profiles/r03_pipeline_experiments.txt.)
  for (Slot& s : cx.slots) {
    HIP_TRY(hipStreamCreateWithPriority(&s.stream, hipStreamNonBlocking, tail_prio));
    HIP_TRY(hipEventCreateWithFlags(&s.acc_done, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&s.pre_done, hipEventDisableTiming));
  }
  for (auto& bc : cx.bcache) HIP_TRY(hipEventCreateWithFlags(&bc.ready, hipEventDisableTiming));
  cx.prio_greatest = prio_greatest;
  cx.prio_least = prio_least;
  cx.device = device;
  cx.inited = true;
  return CURDLE_OK;
}

int init_default_locked(Ctx& cx) { return init_locked(cx, cx.device); }


void set_out_infinity(uint64_t out[18]) {
  G1XYZZ inf;
  g1_set_inf(inf);
  g1_to_canonical_jac(out, inf);
}

// --- slot management -----------------------------------------------------------
// A slot belongs to the caller from acquire to release; the context mutex only
// guards the busy flags, so several threads can run MSMs concurrently.
int acquire_slot(Ctx& cx, bool block, int* idx) {
  std::unique_lock<std::mutex> g(cx.mu);
  int rc = init_default_locked(cx);
  if (rc) return rc;
  for (;;) {
    for (int i = 0; i < kSlots; i++) {
      if (!cx.slots[i].busy) {
        cx.slots[i].busy = true;
        cx.slots[i].claimed = false;
        cx.slots[i].gen++;
        *idx = i;
        return CURDLE_OK;
      }
    }
    if (!block) return fail(CURDLE_EBUSY, "all %d MSM slots are in flight; call curdle_msm_wait first", kSlots);
    cx.cv.wait(g);
  }
}

void release_slot(Ctx& cx, int idx) {
  {
    std::lock_guard<std::mutex> g(cx.mu);
    cx.slots[idx].busy = false;
  }
  cx.cv.notify_one();
}

}  // namespace curdle_api

// ---------------------------------------------------------------------------
// Life cycle
// ---------------------------------------------------------------------------
namespace curdle_api {
// Everything a context owns; the caller holds cx.mu and has checked that nothing is in flight.
// Every handle is checked for null: a context whose initialisation failed half-way is torn down by
// the same code (teardown_partial_locked).
void teardown_locked(Ctx& C) {
  (void)hipSetDevice(C.device);
  for (DSlot& d : C.dslots) {
    if (d.stream) (void)hipStreamSynchronize(d.stream);
    if (d.sub_stream) {
      (void)hipStreamSynchronize(d.sub_stream);
      (void)hipStreamDestroy(d.sub_stream);
    }
    d.sub_stream = nullptr;
    if (d.uploaded) (void)hipEventDestroy(d.uploaded);
    d.uploaded = nullptr;
    for (Buf* b : {&d.in, &d.out, &d.status, &d.sub}) {
      if (b->p) (void)hipFree(b->p);
      b->p = nullptr;
      b->cap = 0;
    }
    if (d.h_in) (void)hipHostFree(d.h_in);
    d.h_in = nullptr;
    d.h_in_cap = 0;
    if (d.h_out) (void)hipHostFree(d.h_out);
    d.h_out = nullptr;
    d.h_out_cap = 0;
    if (d.stream) (void)hipStreamDestroy(d.stream);
    d.stream = nullptr;
    if (d.copy_stream) {
      (void)hipStreamSynchronize(d.copy_stream);
      (void)hipStreamDestroy(d.copy_stream);
    }
    d.copy_stream = nullptr;
    if (d.decoded) (void)hipEventDestroy(d.decoded);
    d.decoded = nullptr;
  }
  for (Slot& S : C.slots) {
    if (S.stream) (void)hipStreamSynchronize(S.stream);
    for (int i = 0; Buf* b = S.all_bufs(i); i++) {
      if (b->p) (void)hipFree(b->p);
      b->p = nullptr;
      b->cap = 0;
    }
    if (S.h_err) (void)hipHostFree(S.h_err);
    S.h_err = nullptr;
    S.scan_epoch = S.scan_base = 0;
    if (S.h_buf) (void)hipHostFree(S.h_buf);
    S.h_buf = nullptr;
    S.h_buf_cap = 0;
    for (int q = 0; q < 2; q++) {
      if (S.h_stage[q]) (void)hipHostFree(S.h_stage[q]);
      S.h_stage[q] = nullptr;
      S.h_stage_cap[q] = 0;
    }
    if (S.ev_made)
      for (auto& e : S.ev) (void)hipEventDestroy(e);
    S.ev_made = false;
    if (S.stream) (void)hipStreamDestroy(S.stream);
    S.stream = nullptr;
    if (S.acc_done) (void)hipEventDestroy(S.acc_done);
    S.acc_done = nullptr;
    if (S.pre_done) (void)hipEventDestroy(S.pre_done);
    S.pre_done = nullptr;
  }
  for (auto& bc : C.bcache) {
    if (bc.buf.p) (void)hipFree(bc.buf.p);
    bc.buf.p = nullptr;
    bc.buf.cap = 0;
    if (bc.ready) (void)hipEventDestroy(bc.ready);
    bc.ready = nullptr;
    bc.key = nullptr;
    bc.n = 0;
    bc.users = 0;
  }
  if (C.main_stream) {
    (void)hipStreamSynchronize(C.main_stream);
    (void)hipStreamDestroy(C.main_stream);
  }
  C.main_stream = nullptr;
  for (auto& st : C.main_extra) {
    if (!st) continue;
    (void)hipStreamSynchronize(st);
    (void)hipStreamDestroy(st);
    st = nullptr;
  }
  for (hipStream_t* st : {&C.pre_stream, &C.pre_stream2}) {
    if (*st) {
      (void)hipStreamSynchronize(*st);
      (void)hipStreamDestroy(*st);
    }
    *st = nullptr;
  }
  if (C.util_stream) (void)hipStreamDestroy(C.util_stream);
  C.util_stream = nullptr;
  if (C.h2d_stream) (void)hipStreamDestroy(C.h2d_stream);
  C.h2d_stream = nullptr;
  C.inited = false;
  C.dstreams_ready.store(false, std::memory_order_release);
  C.epoch++;  // curdle_dbases copies made under this context are re-made (and the old ones freed) on their next use
}
void teardown_partial_locked(Ctx& C) {
  teardown_locked(C);
  C.device = 0;
}
}  // namespace curdle_api

extern "C" int curdle_init(int device) {
  std::lock_guard<std::mutex> cfg(g_cfg_mu);
  Ctx& cx = g_ctxs[0];
  std::lock_guard<std::mutex> g(cx.mu);
  return init_locked(cx, device);
}

extern "C" int curdle_init_devices(const int* devices, int n) {
  if (!devices || n < 1 || n > kMaxDevices) return fail(CURDLE_EINVAL, "devices[] of 1..%d entries expected", kMaxDevices);
  std::lock_guard<std::mutex> cfg(g_cfg_mu);
  const int have = g_ndev.load(std::memory_order_acquire);
  bool any = false;
  for (int i = 0; i < have; i++) {
    std::lock_guard<std::mutex> g(g_ctxs[i].mu);
    any = any || g_ctxs[i].inited;
  }
  if (any && have > 1) {  // a multi-device configuration stands until curdle_shutdown
    bool same = have == n;
    for (int i = 0; same && i < n; i++) same = g_ctxs[i].device == devices[i];
    if (!same) return fail(CURDLE_EINVAL, "already initialised on %d device(s); curdle_shutdown first", have);
    return CURDLE_OK;
  }
  // all or nothing: a context this call brought up is torn down again when a later one fails (bad id,
  // stream creation, out of memory) -- left standing it made every retry fail with "already initialised
  // on device X" until the process ended, and curdle_shutdown, which walks [0, g_ndev), never reached
  // it (review of round 3).  A context that was up before the call (curdle_init) stays.
  bool brought_up[kMaxDevices] = {};
  for (int i = 0; i < n; i++) {
    Ctx& cx = g_ctxs[i];
    int rc;
    {
      std::lock_guard<std::mutex> g(cx.mu);
      const bool was = cx.inited;
      rc = init_locked(cx, devices[i]);  // context 0 may be up already (curdle_init): same device or CURDLE_EINVAL
      brought_up[i] = !was && cx.inited;
      if (rc && !was && !cx.inited) brought_up[i] = cx.util_stream != nullptr;  // failed half-way: its streams exist
    }
    if (rc) {
      char keep[256];
      snprintf(keep, sizeof(keep), "%s", g_err);
      for (int j = 0; j <= i; j++) {
        if (!brought_up[j]) continue;
        Ctx& cj = g_ctxs[j];
        std::lock_guard<std::mutex> g(cj.mu);
        teardown_partial_locked(cj);
      }
      return fail(rc, "%s", keep);
    }
  }
  if (n > 1)
    for (int i = 0; i < n; i++)
      if (!g_ctxs[i].worker) g_ctxs[i].worker = new DevWorker(i);
  g_ndev.store(n, std::memory_order_release);
  return CURDLE_OK;
}

extern "C" int curdle_device_count(void) { return g_ndev.load(std::memory_order_acquire); }

extern "C" int curdle_set_device(int ordinal) {
  if (ordinal == -1) {  // no selection: context 0, and large host-buffer MSMs may spread over all devices again
    tl_dev = 0;
    tl_selected = false;
    return CURDLE_OK;
  }
  if (ordinal < 0 || ordinal >= g_ndev.load(std::memory_order_acquire))
    return fail(CURDLE_EINVAL, "device ordinal %d outside [0, %d)", ordinal, g_ndev.load());
  tl_dev = ordinal;
  tl_selected = true;
  return CURDLE_OK;
}

extern "C" int curdle_get_device(void) { return cur().ordinal; }

// how many host-buffer MSMs were spread over several devices so far (tests: a thread that cleared its selection spreads again)
extern "C" unsigned long long curdle_stat_spread_calls(void) { return g_spread_calls.load(std::memory_order_relaxed); }

// -1 when the calling thread has made no selection (its large host-buffer MSMs spread over all devices)
extern "C" int curdle_get_device_selection(void) { return tl_selected ? cur().ordinal : -1; }

extern "C" int curdle_shutdown(void) {
  std::lock_guard<std::mutex> cfg(g_cfg_mu);
  const int have = g_ndev.load(std::memory_order_acquire);
  if (g_multi_calls.load(std::memory_order_acquire) > 0) return fail(CURDLE_EBUSY, "a multi-device call is still in flight");
  // all or nothing, and nothing may start in between: every context's mutex is held from the check to the end of
  // the teardown (in ordinal order; no other path holds two of them), so a slot acquire or a resident-bases upload
  // either is seen here (CURDLE_EBUSY) or finds the context closed afterwards
  std::vector<std::unique_lock<std::mutex>> held;
  held.reserve((size_t)have);
  for (int i = 0; i < have; i++) held.emplace_back(g_ctxs[i].mu);
  for (int i = 0; i < have; i++) {
    Ctx& C = g_ctxs[i];
    if (!C.inited) continue;
    for (Slot& S : C.slots)
      if (S.busy) return fail(CURDLE_EBUSY, "an MSM is still in flight");
    for (DSlot& d : C.dslots)
      if (d.busy) return fail(CURDLE_EBUSY, "a point decoding is still in flight");
    if (C.pending_uploads > 0) return fail(CURDLE_EBUSY, "a resident base set is still being uploaded");
  }
  for (int i = 0; i < have; i++) {
    Ctx& C = g_ctxs[i];
    delete C.worker;  // joins the device's host thread (its queue is empty: nothing is in flight, and posting takes g_cfg_mu)
    C.worker = nullptr;
    if (C.inited) teardown_locked(C);
    C.device = 0;
  }
  g_ndev.store(1, std::memory_order_release);
  return CURDLE_OK;
}

extern "C" int curdle_plan_override(const char* name, long long value) {
  if (knobs::set(name, value)) return fail(CURDLE_EINVAL, "no knob named %s", name ? name : "(null)");
  return CURDLE_OK;
}

extern "C" int curdle_last_error(char* buf, size_t len) {
  if (!buf || len == 0) return CURDLE_EINVAL;
  snprintf(buf, len, "%s", g_err);
  return CURDLE_OK;
}

extern "C" int curdle_device_available(void) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess) return 0;
  return ndev > 0 ? 1 : 0;
}
