// C ABI of libcurdlemsm.so (include/curdle_msm.h): context, workspace slots,
// phase sequencing, the host-side window combine and the accumulator / rand
// handles.
//
// There is deliberately no CPU implementation of the MSM behind these entry
// points: if the HIP runtime has no device, they fail with CURDLE_ENODEV.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <condition_variable>
#include <chrono>
#include <deque>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/curdle_msm.h"
#include "../host/common_rand.h"
#include "../host/knobs.h"
#include "../host/msmaccumulator.h"
#include "host_math.h"
#include "msm_kernels.h"

using namespace curdle;

extern "C" void curdle_window_combine(const void* winsums_xyzz, int nw, const int* dbls, uint64_t out[18]);
extern "C" void curdle_host_batch_to_affine(void* out_affine, const void* in_xyzz, size_t n);

// ---------------------------------------------------------------------------
// Errors
// ---------------------------------------------------------------------------
static thread_local char g_err[256] = "";

static int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

// for the other translation units of the library (host/proto_api.cpp)
extern "C" int curdle_set_last_error(int code, const char* msg) { return fail(code, "%s", msg); }

#define HIP_TRY(expr)                                                                      \
  do {                                                                                     \
    hipError_t e_ = (expr);                                                                \
    if (e_ != hipSuccess)                                                                  \
      return fail(e_ == hipErrorOutOfMemory ? CURDLE_ENOMEM : CURDLE_EHIP, "%s: %s", #expr, \
                  hipGetErrorString(e_));                                                  \
  } while (0)

// ---------------------------------------------------------------------------
// Contexts: one per configured device (curdle_init: one; curdle_init_devices: up to
// CURDLE_MAX_DEVICES, one process driving several GPUs).  A context has kSlots independent
// workspaces, each with its own HIP stream, so several MSMs can be in flight
// (curdle_msm_g1_device_submit / curdle_msm_wait): the latency-bound tail of one MSM (bucket
// reduce at one wave per SIMD, D2H, host combine) overlaps the throughput-bound accumulation
// of the next.  Workspaces only grow; nothing is allocated in steady state.  Every entry
// point works on the CALLING THREAD's current context (curdle_set_device, default 0), like
// hipSetDevice; tickets and handles remember the context they were made on.
// ---------------------------------------------------------------------------
namespace {

static constexpr int kSlots = 8;
// one-shot decodings up to this size take the two-kernel form
static inline size_t two_kernel_max() { return knobs::is_set(knobs::TWO_KERNEL_MAX) ? (size_t)knobs::get(knobs::TWO_KERNEL_MAX) : (size_t)32768; }
static constexpr int kMaxDeferred = 4;          // two-step point decodings in flight (see curdle_g1_decompress_begin); with 2, eight threads verifying from bytes ran at 1,500-2,200 /s, with 4 at 2,400-2,500
// Batches at least this large combine their window sums on the GPU (k_combine: one quad per
// MSM, 127 doublings, ~0.5 ms however many) instead of one Horner pass per MSM on the host
// (0.05 ms each for small MSMs).  Measured after the GLV split, k x 128 / 628 pairs: k = 8 0.67
// (host) against 0.81 ms, k = 12 0.89 against 0.76, k = 16 1.09 against 0.77, k = 24 1.51
// against 0.77.
static inline size_t gpu_combine_min() { return 12; }

struct Buf {
  void* p = nullptr;
  size_t cap = 0;
};

struct Slot {
  hipStream_t stream = nullptr;  // high priority: the tail phases
  hipEvent_t acc_done = nullptr;
  hipEvent_t pre_done = nullptr;
  Buf points, scalars, offsets, points28, counts, starts, cursor, fragcnt, foff, small, digits, sorted, frags, partials,
      winsums, winsums28, results, job, tmp, ccur, fold_sums, fold_meta;
  void* h_stage[2] = {nullptr, nullptr};  // pinned staging of the device accumulator (instance points; job)
  size_t h_stage_cap[2] = {0, 0};
  void* h_buf = nullptr;  // pinned: window sums (host combine) or results (GPU combine)
  size_t h_buf_cap = 0;
  hipEvent_t ev[CURDLE_PROF_MAX_KERNELS + 1];
  bool ev_made = false;
  // the call in flight
  struct curdle_dbases* held_bases = nullptr;  // a pipelined MSM over a resident base set keeps its reference until the wait
  int held_cache = -1;                         // ... or over a cached converted copy (Ctx::bcache)
  bool busy = false;
  bool claimed = false;   // a curdle_msm_wait is finishing this call (a second wait on the ticket is refused)
  uint32_t gen = 0;       // bumped at every acquire: tickets carry it, stale ones are refused
  hipStream_t run_stream = nullptr;
  MsmPlan plan;
  bool profiled = false;
  Buf chain;                  // k_scan_chain's words and ticket counter (zero when made, never cleared again)
  bool chain_dirty = false;   // a call failed between taking the chain and its last launch: the host's ticket count may lag the device's -- clear both
  Buf mdone;                  // k_merge_large's chunk counters, one per queue entry and base set (zero when made; the kernel leaves them zero)
  uint32_t* h_err = nullptr;  // pinned word a kernel raises when a wait inside it gave up (finish_slot reads it)
  uint32_t scan_epoch = 0;    // epoch of the slot's last k_scan_chain launch (30 bits, never 0)
  uint32_t scan_base = 0;     // tickets the slot's launches have taken so far
  uint32_t coarse_nw = 0;     // window count the zeroed tail of `ccur` was laid out for
  bool coarse_dirty = false;  // a call was abandoned between its ensure and its last launch: clear `ccur` again
  int prof_n = 0;
  const char* prof_name[CURDLE_PROF_MAX_KERNELS];

  Buf* all_bufs(int i) {
    Buf* b[] = {&points, &scalars, &offsets, &points28, &counts, &starts, &cursor, &fragcnt, &foff, &small,
                &digits, &sorted,  &frags,   &partials, &winsums, &winsums28, &results, &job,      &tmp,    &ccur,
                &fold_sums, &fold_meta, &chain,   &mdone};
    return i < (int)(sizeof(b) / sizeof(b[0])) ? b[i] : nullptr;
  }
};

// A decode context: what a two-step point decoding (curdle_g1_decompress_begin / _finish)
// holds between the two calls.  A pool of its own, NOT the MSM slots: the holder goes on to
// call MSM entry points while the subgroup test runs, and eight such callers holding the
// eight MSM slots would wait for each other forever.
struct DSlot {
  hipStream_t stream = nullptr;       // upload, decoding kernel (square roots)
  hipStream_t sub_stream = nullptr;   // the subgroup test, from the records, beside the decoding kernel
  hipStream_t copy_stream = nullptr;  // hands the points back while the subgroup test runs
  hipEvent_t uploaded = nullptr;      // the records are in device memory
  hipEvent_t decoded = nullptr;       // the decoding kernel is done
  Buf in, out, status, sub;
  void* h_in = nullptr;  // pinned staging of the compressed records
  size_t h_in_cap = 0;
  void* h_out = nullptr;  // pinned staging of what comes back: n x 96 B of points, 2 x n status bytes
  size_t h_out_cap = 0;
  bool busy = false;
  bool claimed = false;
  uint32_t gen = 0;
  uint32_t n = 0;
};

struct DevWorker;
struct Ctx {
  std::mutex mu;
  std::condition_variable cv;
  DSlot dslots[kMaxDeferred];
  std::atomic<bool> dstreams_ready{false};  // the decode contexts' streams exist (published with release / acquire)
  bool inited = false;
  int device = 0;   // HIP device id
  int ordinal = 0;  // index of this context (what curdle_set_device takes, what tickets carry)
  DevWorker* worker = nullptr;  // the host thread of this device for multi-device calls (made by curdle_init_devices)
  hipStream_t util_stream = nullptr;  // synthetic inputs, self-test
  hipStream_t h2d_stream = nullptr;   // the chunk copies of large host-buffer MSMs, one behind the other
  // Sort + accumulate of every MSM run in order on this normal-priority stream; each
  // slot's latency-bound tail (merge / bucket reduce / window sum / D2H) runs on the
  // slot's own high-priority stream, so it fills the chip's idle issue slots beside the
  // next MSM's accumulation instead of two accumulations time-slicing each other.
  hipStream_t main_stream = nullptr;
  // Pipelined (submit / wait) calls rotate their accumulate launches over main_streams
  // streams (default 2, CURDLE_MAIN_STREAMS=1..4): the next MSM's accumulation fills the
  // chip while the previous one drains its last blocks, instead of waiting behind it in
  // one in-order queue (measured at N = 2^20: 3.44 -> 3.30 ms per MSM with 4 in flight).
  hipStream_t main_extra[3] = {nullptr, nullptr, nullptr};
  int prio_least = 0, prio_greatest = 0;  // stream priority range of the device
  int main_streams = 2;
  std::atomic<unsigned> submit_count{0};  // submits come from any thread
  // Recoding + bucket sort of every MSM, in order; light, memory/LDS-bound phases that
  // overlap the previous MSM's accumulation.
  hipStream_t pre_stream = nullptr;
  // Window-range partials (the multi-GPU split) alternate their sort phases over two streams:
  // per partial the sort is no smaller than for a whole MSM (every rank converts and recodes
  // all n pairs) while the accumulation is 1/world of it, so one in-order sort stream was the
  // bottleneck of the pipeline (8 ranks: 1.06 -> 0.80 ms per step with 5 in flight).
  hipStream_t pre_stream2 = nullptr;
  int pre_streams = 2;
  Slot slots[kSlots];
  // Converted copies of base arrays whose callers promised they do not change (CURDLE_MSM_BASES_UNCHANGED): keyed
  // by the device pointer and the count, made on first use on util_stream, reused by every later flagged call.
  struct BaseCache {
    const void* key = nullptr;
    size_t n = 0;
    Buf buf;
    hipEvent_t ready = nullptr;  // recorded behind the conversion; every user's first stream waits for it
    int users = 0;               // calls in flight that read the copy
    uint64_t stamp = 0;          // last use (the least recently used idle entry is replaced)
  };
  static constexpr int kBaseCache = 4;
  BaseCache bcache[kBaseCache];
  uint64_t bstamp = 0;
  unsigned epoch = 0;  // bumped by curdle_shutdown: resident base sets of a closed context are refused
  int pending_uploads = 0;  // resident base sets being copied + converted on util_stream right now (outside cx.mu): curdle_shutdown waits for none
  int profile = 0;  // 0 off, 1 every phase, 2 the dominant kernel only
  curdle_profile last = {};
};

static constexpr int kMaxDevices = CURDLE_MAX_DEVICES;
Ctx g_ctxs[kMaxDevices];
std::atomic<int> g_ndev{1};  // configured contexts: [0, g_ndev)
std::mutex g_cfg_mu;         // configuration (curdle_init_devices / curdle_shutdown)
std::atomic<int> g_multi_calls{0};  // calls that span the devices' host threads right now
std::atomic<unsigned long long> g_spread_calls{0};  // host-buffer MSMs that were spread over several devices, ever
thread_local int tl_dev = 0;
thread_local bool tl_selected = false;  // the thread called curdle_set_device: its host-buffer MSMs stay on that device
// the calling thread's context; a thread whose selection no longer exists (curdle_shutdown since) is on 0
inline Ctx& cur() { return g_ctxs[tl_dev < g_ndev.load(std::memory_order_acquire) ? tl_dev : 0]; }
const bool g_ordinals_set = [] {
  for (int i = 0; i < kMaxDevices; i++) g_ctxs[i].ordinal = i;
  return true;
}();

// The host thread of one context: calls that span devices (curdle_msm_g1 over host buffers,
// curdle_msm_g1_replicated) hand each device's share to that device's thread, which lives on the
// context (tl_dev) for good -- SURVEY.md section 7 step 6: one host thread per device.
struct DevWorker {
  std::mutex mu;
  std::condition_variable cv;
  std::deque<std::function<void()>> q;
  bool stop = false;
  std::thread th;
  explicit DevWorker(int ordinal) {
    th = std::thread([this, ordinal] {
      tl_dev = ordinal;
      tl_selected = true;
      for (;;) {
        std::function<void()> job;
        {
          std::unique_lock<std::mutex> g(mu);
          cv.wait(g, [&] { return stop || !q.empty(); });
          if (q.empty()) return;  // stop, and nothing left to run
          job = std::move(q.front());
          q.pop_front();
        }
        try {
          job();
        } catch (...) {
          // a job reports through its own record (run_on_devices converts exceptions to a status); nothing
          // may leave this thread function: an escaped exception is std::terminate for the whole host process
        }
      }
    });
  }
  void post(std::function<void()> f) {
    {
      std::lock_guard<std::mutex> g(mu);
      q.push_back(std::move(f));
    }
    cv.notify_one();
  }
  ~DevWorker() {
    {
      std::lock_guard<std::mutex> g(mu);
      stop = true;
    }
    cv.notify_one();
    th.join();
  }
};

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

int choose_window_bits(size_t n, bool many = false) {
  {
    const long long c = knobs::get(knobs::WINDOW_BITS);
    if (c >= 4 && c <= 16) return (int)c;
  }
  // in TERMS of the GLV split, two per pair: that is what a window's buckets hold
  n *= 2;
  int lg = 0;
  while (((size_t)1 << (lg + 1)) <= n) lg++;
  int c = lg - 2;
  // Below 2^20 pairs the bucket reduction's chain weighs more than the additions a wider window
  // saves, and widths that cut the 127 bits unevenly have fewer bucket slots per window.
  // Measured with tools/sweep.py n,c (profiles/r02_window_bits_sweep.txt, last section: with the
  // split; pairs -> ms): 8 up to 1,500 pairs (1,268: 0.390; 10: 0.393), 10 up to 6,000 (2,000:
  // 0.400 against 0.505 at 8), 11 up to 80,000 (40,000: 0.636; 14: 0.661), 14 up to 200,000
  // (131,072: 0.936; 11: 1.047), 15 up to 450,000 (262,144: 1.336; 14: 1.453; 16: 1.368), 16
  // beyond (524,288: 2.258; 15: 2.278).  Each width has a cliff below it (buckets of hundreds of
  // terms go through merge_large), so the steps sit well before them.  A large batch of small
  // MSMs is throughput-bound instead (wider windows double its bucket-reduce work) and keeps
  // lg - 2.
  // Round 4, with the bucket reduction that no longer multiplies (its chain is a handful of additions per
  // level, so wider windows cost less than they did; profiles/r04_window_bits_sweep.txt, pairs -> ms): 10
  // from 300 pairs (628: 0.346 against 0.354 at 8; 1,268 -- the verifier's MSM -- 0.359 against 0.387;
  // 2,548: 0.383), 11 from 3,000 (4,096: 0.398 against 0.418 at 10) up to 45,000, 13 up to 100,000
  // (65,536: 0.666 against 0.711 at 11), 14 up to 200,000, 15 up to 450,000, 16 beyond.
  // Round 5, on the unprofiled call (profiles/r05_window_bits_sweep.txt): the table holds -- 2^14 / 2^15: 11; 2^16: 13 or 14; 2^17: 14; 2^18: 15 --
  // except that 11 starts paying from ~2,000 pairs (2,548: 0.290 against 0.296 ms at 10; 1,268: 0.279 against 0.270, so the verifier's stays at 10).
  if (!many && n >= 600) c = n <= 4000 ? 10 : n <= 90000 ? 11 : n <= 200000 ? 13 : n <= 400000 ? 14 : n <= 900000 ? 15 : 16;
  if (c < 4) c = 4;
  if (c > 16) c = 16;
  return c;
}

// Window widths for a maximum width c.  The kernels never see a 255-bit scalar: k_digits splits
// every scalar into two 127-bit halves (GLV, msm_kernels.hip: k P = k1 P + k2 phi(P)), so an MSM
// of n pairs is 2 n terms over W = ceil(127 / c) windows, the 127 bits spread as evenly as
// possible (the wider windows lowest), the top window unsigned.  For c = 16 that is 7 windows
// of 16 bits and a 15-bit top window.
// (Without the split -- CURDLE_MSM_ANY_CURVE_POINT -- the same rule over the 255 bits of the whole scalar.)
constexpr int kScalarBits = 127;
constexpr int kScalarBitsNoGlv = 255;
int window_widths(int c, uint8_t bits[kMaxWindows], int scalar_bits = kScalarBits) {
  const int W = (scalar_bits + c - 1) / c;
  const int base = scalar_bits / W, extra = scalar_bits % W;
  for (int w = 0; w < W; w++) bits[w] = (uint8_t)(base + (w < extra ? 1 : 0));
  return W;
}

// Plan for k MSMs of n_total pairs in all, the largest having n_max pairs.
int make_plan(MsmPlan& p, size_t n_total, size_t k, size_t n_max, int c, int win_begin, int win_end,
              bool latency_mode, size_t sets = 1, bool many = false, uint32_t seg_override = 0, bool light_host = false,
              bool glv = true) {
  if (n_total > ((size_t)1 << 27)) return fail(CURDLE_EINVAL, "n = %zu exceeds the supported 2^27 pairs", n_total);
  many = many || k * sets >= gpu_combine_min();  // a pass of a larger batch keeps the batch's rules
  if (c == 0) c = choose_window_bits(n_max, many);
  if (c < 4 || c > 16) return fail(CURDLE_EINVAL, "window_bits %d outside [4, 16]", c);
  memset(&p, 0, sizeof(p));
  // from here on the counts are the split's terms: two per pair (k1 P and k2 phi(P), adjacent)
  n_total *= 2;
  n_max *= 2;
  p.n = (uint32_t)n_total;
  p.k = (uint32_t)k;
  p.sets = (uint32_t)sets;
  p.kr = (uint32_t)(k * sets);
  p.n_max = (uint32_t)n_max;
  p.c = c;
  p.glv = glv ? 1u : 0u;
  // (without the split the terms keep their numbering -- 2 i is k P_i, 2 i + 1 never contributes -- so every
  // kernel behind k_digits is the same code; the opt-out pays for it with a digit array twice the needed size)
  p.W = window_widths(c, p.bits, glv ? kScalarBits : kScalarBitsNoGlv);
  if (win_end < 0) win_end = p.W;
  if (win_begin < 0 || win_begin > win_end || win_end > p.W)
    return fail(CURDLE_EINVAL, "window range [%d, %d) outside [0, %d)", win_begin, win_end, p.W);
  p.win_begin = win_begin;
  p.win_end = win_end;
  uint32_t sh = 0, min_nbkt = 0xffffffffu;
  for (int w = 0; w < p.W; w++) {
    p.shift[w] = (uint16_t)sh;
    sh += p.bits[w];
    p.nbkt[w] = w == p.W - 1 ? (1u << p.bits[w]) : (1u << (p.bits[w] - 1));
  }
  for (int w = win_begin; w < win_end; w++) {
    p.base[w] = p.NB;
    p.NB += p.nbkt[w];
    if (p.nbkt[w] > p.max_nbkt) p.max_nbkt = p.nbkt[w];
    if (p.nbkt[w] < min_nbkt) min_nbkt = p.nbkt[w];
  }
  if (p.max_nbkt > 32768) return fail(CURDLE_EINVAL, "window of %u buckets exceeds the LDS histogram", p.max_nbkt);
  if (win_begin == win_end) return CURDLE_OK;
  const uint64_t nbk = (uint64_t)k * sets * p.NB;  // bucket slots the reduce kernels walk
  // Buckets per running-sum segment: long segments amortise the per-segment scalar
  // multiple, short ones keep the serial chain short when there are few buckets.
  // This is the starting point of the pipelined rule below.
  p.seg = nbk >= (1u << 19) ? 16 : (nbk >= (1u << 14) ? 4 : 1);
  // The latency-bound kernels work on quads (four lanes per point, quad28.h).  When the caller
  // waits for this very call (synchronous entry points) a segment is ONE bucket (the verifier's
  // 1,370-pair MSM, 4,096 slots: fragments + multiple + tree, 0.158 ms against 0.199 with two)
  // and is lengthened, up to 32 buckets, until the four-fold lane count is at most HALF a round
  // of the chip at two waves per SIMD (65,536 lanes): a quad's addition is 4 product steps
  // against 14, the chain (2.5 seg + log2(buckets) point operations) is what the caller waits
  // for, and beyond one wave per SIMD the waves share the multiplier.  Measured against the rule
  // before it (4 buckets from 2^14 slots, a whole round of lanes;
  // profiles/r02_sync_reduce_segments.txt): 8,192 pairs 0.566 -> 0.511 ms, 32,768 pairs 0.762 ->
  // 0.722, 65,536 pairs 0.836 -> 0.778, 2^18 pairs 1.59 -> 1.56, 2^20 pairs 3.63 -> 3.58.  A call
  // too large for half a round even so (big batches) takes 16-bucket segments over several rounds (1,024 x 628
  // pairs: 7.67 ms against 7.69 / 7.70 with 32 / 64).  Pipelined (submit / wait) calls hide
  // their tails behind other MSMs' accumulation, whose waves leave room for ONE more wave of
  // quads per SIMD at best: their segments are lengthened, up to 64 buckets, until the quads are
  // a quarter of a round (32,768 lanes).  Measured per MSM with 4-6 in flight
  // (profiles/r02_multi_gpu_emulation.jsonl): all 16 windows of N = 2^20 2.89 -> 2.84 ms
  // against half a round; the 8 / 4 / 2 windows of a rank of the multi-GPU split -- which the
  // rule before this one left at 4-bucket segments, two rounds of quads for 8 windows --
  // 1.83 -> 1.52, 0.98 -> 0.88 and 0.61 -> 0.59 ms.
  if (latency_mode) {
    // (round 4: a whole round, 131,072 lanes, since the reduction lost its per-segment scalar multiple: shorter
    // segments now cost a quad almost nothing extra -- 131,072 pairs 0.93 -> 0.89 ms, 2^20 3.45 -> 3.41)
    const uint64_t lanes = 131072;
    uint32_t seg = 1;
    while (nbk / seg * 4 > lanes && seg < 32) seg *= 2;
    p.seg = nbk / seg * 4 <= lanes ? seg : 16;
  } else {
    const uint64_t lanes = 32768;  // (16,384 / 65,536 / 131,072 measured equal: profiles/r04_pipeline_phase_costs.txt)
    uint32_t seg = p.seg;
    while (nbk / seg * 4 > lanes && seg < 64) seg *= 2;
    p.seg = seg;
  }
  if (seg_override) p.seg = seg_override;
  if (knobs::get(knobs::REDUCE_SEG) > 0) p.seg = (uint32_t)knobs::get(knobs::REDUCE_SEG);
  if (p.seg < 1) p.seg = 1;
  while (p.seg > min_nbkt || (p.seg & (p.seg - 1))) p.seg >>= 1;
  p.NS = p.NB / p.seg;
  // Single MSMs reduce their buckets without a scalar multiple (msm_kernels.hip, k_reduce_segments):
  // the window leaves the GPU as bit-positioned points for the host's Horner pass.  Batches and
  // shared-scalar calls keep k_bucket_reduce_quad: their host pass runs once per RESULT, and ~8 more
  // additions per window and result cost the host more than the GPU saves.
  // ... and tiny MSMs too (below 300 pairs the plan has 19-32 windows of 8-64 buckets: five to seven points
  // per window for the host against a 3-to-6-bit multiple on the GPU -- 8..299 pairs measured 0.01-0.03 ms
  // slower, and the batch verifiers, which are host-bound, lost a quarter of their throughput to the
  // longer host passes of their per-proof MSMs: profiles/r04_reduce_bits_small.txt).
  {
    // (light_host: a queued MSM of a batch verifier -- a host-bound caller with dozens in flight, to whom the
    // longer host pass costs throughput and the shorter GPU chain buys nothing: 1,024 Whisk proofs 31-36 ms
    // per batch with k_bucket_reduce_quad, 41-46 with this form)
    const bool shapes = k * sets == 1 && !many && !light_host;
    p.reduce_bits = shapes && n_total >= 600 ? 1u : 0u;  // terms: two per pair
  }
  const uint32_t gmax = p.reduce_bits ? 16u : 64u;  // quads per group: one wave's, or one block's
  p.G = min_nbkt / p.seg < gmax ? min_nbkt / p.seg : gmax;
  if (p.reduce_bits) {
    while ((1u << p.lg_seg) < p.seg) p.lg_seg++;
    while ((1u << p.lgG) < p.G) p.lgG++;
    p.NG = p.NS / p.G;
    uint32_t lgn = 0;
    while ((p.seg << (lgn + 1)) <= p.max_nbkt) lgn++;
    p.nout = 2 + lgn;
  }
  // Sorted positions per accumulate lane: about two full-chip rounds of lanes
  // (256 CUs x 4 SIMDs x 2 waves x 64 lanes) for large inputs, never below 8.
  const uint64_t entries = (uint64_t)(win_end - win_begin) * n_total;
  uint64_t L = (entries + 2 * 131072 - 1) / (2 * 131072);
  // ... but ONE round up to 128 positions per lane: every lane ends with a fragment the bucket
  // reduce has to add, and with 2^21..2^22 entries (a window range of the multi-GPU split) half
  // as many lanes take 0.03-0.05 ms off its chain at no cost to the accumulation; at 2^24
  // entries (N = 2^20) it is 1 % of the pipelined step (2.71 -> 2.68 ms)
  {
    const uint64_t round_lanes = 131072;  // (0.8 of a round measured no better: profiles/r05_rank_step_knobs.txt)
    const uint64_t one = (entries + round_lanes - 1) / round_lanes;
    if (one <= 128 && one > L) L = one;
  }
  const bool L_forced = knobs::get(knobs::SEG_LEN) > 0;
  if (L_forced) L = (uint64_t)knobs::get(knobs::SEG_LEN);
  // small MSMs are latency-bound on the lane's chain of L mixed additions: halve it while the
  // launch stays far below one round of the chip (every lane emits at least one fragment,
  // which the bucket reduce has to add, so not below 4)
  const uint64_t Lmin = entries <= 8 * 65536 ? 4 : 8;
  if (L < Lmin) L = Lmin;
  // ... and long enough that an evenly loaded bucket of the narrowest window is cut into about
  // eight fragments at most: beyond max_small (16) a bucket takes the merge_large detour, which
  // is there for skewed scalars, not for uniform ones (16,384 pairs: 0.81 -> 0.5 ms)
  if (!L_forced && min_nbkt) {
    const uint64_t load = (n_max + min_nbkt - 1) / min_nbkt;
    if (L < (load + 7) / 8) L = (load + 7) / 8;
  }
  if (L > 128) L = 128;
  // beyond ~2^25 pairs even 128 positions per lane leave more than 4M lanes and cut a bucket
  // into more than max_small fragments (they would all take the merge_large detour): grow L
  if (entries / L > ((uint64_t)1 << 22)) L = entries >> 22;
  p.L = (uint32_t)L;
  // The two waves a SIMD holds of a synchronous call's accumulation take turns at high priority (k_accumulate):
  // left to the hardware's oldest-first rule one of them ran ahead, finished after 57 % of the kernel and
  // left the other alone at 0.77 of the pair's rate (wave stamps: profiles/r04_wave_trace.txt; 2^20: kernel
  // 2.50 -> 2.38 ms, the call 3.18 -> 3.05).  Not for pipelined calls: the next MSM's waves take the freed
  // slots there, and raised priorities starve the sort and reduce kernels beside them (2.58 -> 2.69 ms per step).
  // A pipelined call's reduction runs beside the next MSMs' accumulations, whose waves are older and win the
  // arbiter: raised to priority 3 its waves give their slots back sooner (2.59-2.61 -> 2.54-2.55 ms per step, four
  // A/B pairs on two boxes, profiles/r04_wave_priorities.txt; the sort kernels raised as well: half the gain lost)
  {
    const long long v = knobs::get(knobs::REDUCE_PRIO);
    p.reduce_prio = v >= 0 ? (uint32_t)(v > 3 ? 3 : v) : (latency_mode ? 0u : 3u);
  }
  {
    // The sort kernels (and the conversion) of a PIPELINED call at priority 3 (round 5): beside the two accumulate
    // waves of a neighbouring call -- older, and never short of an instruction -- a young wave gets what is left, and
    // k_digits took 0.2-0.3 ms there against 0.03 alone; raised, they are through before they have cost the accumulation
    // anything that shows.  One rank of the 8-way window split 0.466-0.470 -> 0.424-0.426 ms with kept bases and 0.54 ->
    // 0.485 on gnark-layout inputs, a 4-way rank 0.74 -> 0.69 / 0.84 -> 0.75, the whole MSM 2.598 -> 2.555 ms per step
    // (two A/B rounds on one box, gpurun_out/r5r; round 4 had measured the raised sort TOGETHER with the raised
    // reduction slower than the reduction alone -- with the lighter sort of this round it is the other way round).
    // Synchronous calls have the chip to themselves; chunked host-buffer calls do not move (4.45-4.57 ms either way).
    const long long v = knobs::get(knobs::AUX_PRIO);
    p.aux_prio = v >= 0 ? (uint32_t)(v > 3 ? 3 : v) : (!latency_mode ? 3u : 0u);
  }
  {
    const long long v = knobs::get(knobs::ACC_PRIO);
    p.acc_prio = v >= 0 ? (uint32_t)(v > 24 ? 24 : v) : (latency_mode && !light_host && entries / L >= 65536 ? 15u : 0u);
  }
  // Buckets with more fragments than this go through k_merge_large first.  Measured as a knob in round 6
  // (profiles/r06_max_small.txt): 12 / 8 / 6 / 4 cost uniform inputs of 8,192 .. 65,536 pairs up to 0.03 / 0.07 / 0.10 /
  // 0.29 ms (their narrow windows hold 9-13 fragments per bucket by design, and a wave per such bucket is a poor trade)
  // and buy back 0.15 ms only where a few dozen buckets hold exactly that many (64 distinct scalar values at 2^12 pairs).
  // Up to 4,096 pairs 8 costs uniform inputs nothing (same file: 1,268 and 4,096 pairs equal to the microsecond) and takes
  // 0.15 ms off a call whose occupied buckets hold exactly 9..16 (0.54 -> 0.39 ms at 4,096 pairs).
  p.max_small = n_total <= 8192 && k * sets == 1 ? 8 : 16;
  // a bucket with more than max_small fragments holds more than (max_small - 1) * L entries
  uint64_t ml = entries / ((uint64_t)(p.max_small - 1) * p.L) + 1;
  p.max_large = (uint32_t)(ml < nbk ? ml : nbk);
  if (p.max_large == 0) p.max_large = 1;
  // Pairs per sort block: about 512 blocks over all windows, at least 4096 pairs each
  // (an MSM of a batch is never split below that).
  uint64_t ch = (entries + 511) / 512;
  if (ch < 4096) ch = 4096;
  if (ch > n_max) ch = n_max ? n_max : 1;
  p.chunk = (uint32_t)ch;
  p.gpu_combine = many ? 1u : 0u;
  // The scatter in two passes (msm_kernels.hip): one MSM whose windows are whole numbers of
  // 128-bucket bins (c >= 13) and whose term indices fit the intermediate entries' 24 bits; below
  // 2^17 terms the extra launches cost more than the stores save.  CURDLE_SCATTER=1 / 2 forces
  // the one-pass / two-pass form where the shapes allow.
  {
    const long long forced = knobs::get(knobs::SCATTER);
    const bool shapes = k == 1 && min_nbkt >= 4096 && (min_nbkt & 127u) == 0 && p.max_nbkt <= 32768 && p.n <= (1u << 24) &&
                        win_end - win_begin <= 20;
    p.two_level = shapes && forced != 1 && (forced == 2 || p.n >= (1u << 17)) ? 1u : 0u;
  }
  // The bucket-slot scans.  k_scan_one -- one block, the slots read once, coalesced, the block scans by wave shuffles --
  // up to 32,768 slots; k_scan_chain (round 5) -- one launch at ANY size, tile sums handed down a chain -- beyond that
  // and (enqueue_slot) for every pipelined or chunked call, beside whose neighbours k_scan_one's 16 x 121-register block
  // cannot start: a rank of the 8-way window split 0.428 -> 0.396 ms per step, synchronous 2^16 / 2^17 / 2^18 pairs 0.610 /
  // 0.808 / 1.149 -> 0.589 / 0.788 / 1.130 ms (profiles/r05_scan_chain.txt).  The six-launch multi-block form is left for
  // ONE case: L = 1 (knob SEG_LEN), since both one-launch forms divide by L with a multiply that needs L >= 2.
  // (Round 6: the knob SCAN and k_scan_fused, the round-2 single-block form, are gone: every comparison is under profiles/.)
  {
    const uint64_t nbs = (uint64_t)k * p.NB;
    p.fuse_scan = nbs <= 32768 ? 2u : 3u;
    if (p.L < 2) p.fuse_scan = 0;
  }
  return CURDLE_OK;
}

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

// A ticket names a context, a slot AND the acquisition it was handed out for (slot index in
// the low three bits, the context's ordinal in the five above them, the slot's generation
// above the low byte), so a ticket that was already waited for, or one kept across a later
// submit, is refused instead of touching another caller's workspace, and a wait may come from
// any thread whatever its current device.
inline int make_ticket(const Ctx& cx, int idx, uint32_t gen) {
  return (int)(((gen & 0x7fffffu) << 8) | ((uint32_t)cx.ordinal << 3) | (uint32_t)idx);
}
inline int ticket_index(int ticket) { return ticket & 0x7; }
inline int ticket_dev(int ticket) { return (ticket >> 3) & 0x1f; }
inline uint32_t ticket_gen(int ticket) { return ((uint32_t)ticket >> 8) & 0x7fffffu; }
// the context a ticket was made on; nullptr for a ticket that names none
inline Ctx* ticket_ctx(int ticket) {
  if (ticket < 0 || ticket_dev(ticket) >= g_ndev.load(std::memory_order_acquire)) return nullptr;
  return &g_ctxs[ticket_dev(ticket)];
}

// HIP-event bracketing of the phases of one call.  mode 1: an event after every phase (ten
// timed events per MSM, on three streams).  mode 2: only the dominant kernel (accumulate) is
// bracketed -- two events on its own stream -- because every timed event is a barrier packet
// in the hardware queue and the full set costs a pipelined caller ~0.1 ms per MSM.
struct Prof {
  Slot& s;
  hipStream_t st;
  int mode;
  Prof(Slot& slot, hipStream_t stream, int m) : s(slot), st(stream), mode(m) {
    s.profiled = mode != 0;
    s.prof_n = 0;
    if (mode && !s.ev_made) {
      for (auto& e : s.ev) (void)hipEventCreate(&e);
      s.ev_made = true;
    }
    if (mode == 1) (void)hipEventRecord(s.ev[0], st);
  }
  void mark(const char* name) {
    if (!mode || s.prof_n >= CURDLE_PROF_MAX_KERNELS) return;
    if (mode == 2 && strcmp(name, "accumulate")) return;
    s.prof_name[s.prof_n++] = name;
    (void)hipEventRecord(s.ev[s.prof_n], st);
  }
  // right before the dominant kernel is launched: opens its bracket in mode 2
  void before_dominant() {
    if (mode == 2) (void)hipEventRecord(s.ev[0], st);
  }
};

// One MSM accumulated in CHUNKS over one plan (run_host_chunked): a chunk that is not the last
// stops after its accumulation (its fragments stay in its slot; `frags_done` on its tail stream
// says when), the last one waits for the earlier chunks' events and folds their fragment lists
// into its own bucket reduction (FragSources, msm_kernels.h).
struct ChunkJoin {
  bool accumulate_only = false;       // an earlier chunk: no reduce, no window sums, no D2H
  std::vector<Slot*> earlier;         // the last chunk: the slots of the chunks before it
  uint32_t seg = 0;                   // buckets per reduce segment, the same for every chunk (0: the plan's rule)
  // A chunk is enqueued in two steps: its sort needs only its scalars, which cross PCIe first; the
  // conversion and everything behind it wait for its points.  0: all at once.
  int phase = 0;                      // 1: recoding + sort only; 2: conversion, accumulation, tail (same slot, same plan)
  // Progressive folding (round 4): an earlier chunk's fragments are added into one running sum per bucket as soon as
  // its accumulation is done (k_fold_fragments); the last chunk's reduction reads the sums as ONE fragment source.
  Slot* fold_home = nullptr;          // the slot that owns the sums (the first chunk's); null: the reduction walks every chunk's fragments
  Slot* fold_prev = nullptr;          // the chunk before this one: its acc_done then means "accumulated AND folded"
};

#ifdef CURDLE_EXP_SKIP
// Experiment build only (tools/exp/phase_costs.sh -> build_alt/, never the product library): phases of a PIPELINED call left
// out after the slot's first uses, so that what each phase costs the pipeline can be priced (profiles/r06_pipeline_phase_costs.txt).
// CURDLE_DEBUG_SKIP bits: 1 conversion, 2 sort, 4 merge_large, 8 reduce_segments, 16 reduce_level.  Results are garbage by construction.
static unsigned exp_skip_mask() {
  static const unsigned m = [] {
    const char* e = getenv("CURDLE_DEBUG_SKIP");
    return e ? (unsigned)atoi(e) : 0u;
  }();
  return m;
}
#define EXP_SKIP(bit) (exp_skip_mask() & (bit) && S.gen > 12 && !latency_mode)
#else
#define EXP_SKIP(bit) false
#endif

// Enqueue every GPU phase of k MSMs on the slot's stream (no host synchronisation).
// d_points / d_scalars are device pointers holding the pairs of all MSMs back to back
// and must stay valid until the matching finish_slot(); h_off has k + 1 entries.
int enqueue_slot_impl(Ctx& cx, Slot& S, const void* d_points, const void* d_scalars, const uint32_t* h_off, size_t k, int c,
                      int win_begin, int win_end, hipStream_t pre, hipStream_t stream, hipStream_t tail, bool latency_mode,
                      bool points28_ready, size_t sets, bool many, const ChunkJoin* join, const void* ext_points28,
                      bool light_host, bool glv, const DaccFront* dfront);
int enqueue_slot(Ctx& cx, Slot& S, const void* d_points, const void* d_scalars, const uint32_t* h_off, size_t k, int c,
                 int win_begin, int win_end, hipStream_t pre, hipStream_t stream, hipStream_t tail,
                 bool latency_mode = true, bool points28_ready = false, size_t sets = 1, bool many = false,
                 const ChunkJoin* join = nullptr, const void* ext_points28 = nullptr, bool light_host = false,
                 bool glv = true, const DaccFront* dfront = nullptr) {
  const int rc = enqueue_slot_impl(cx, S, d_points, d_scalars, h_off, k, c, win_begin, win_end, pre, stream, tail, latency_mode,
                                   points28_ready, sets, many, join, ext_points28, light_host, glv, dfront);
  if (rc == CURDLE_OK) {
    S.coarse_dirty = false;  // every launch of the call is in its queue: k_digits leaves its counters zero
    S.chain_dirty = false;   // ... and the host's count of the scan chain's tickets is the device's
  }
  return rc;
}
int enqueue_slot_impl(Ctx& cx, Slot& S, const void* d_points, const void* d_scalars, const uint32_t* h_off, size_t k, int c,
                      int win_begin, int win_end, hipStream_t pre, hipStream_t stream, hipStream_t tail, bool latency_mode,
                      bool points28_ready, size_t sets, bool many, const ChunkJoin* join, const void* ext_points28,
                      bool light_host, bool glv, const DaccFront* dfront) {
  // ext_points28: the bases are a resident, pre-converted set (curdle_dbases: two records per base in the internal
  // form, the first h_off[k] of them) -- d_points is not read, nothing is converted or copied
  if (ext_points28 && (k != 1 || sets != 1)) return fail(CURDLE_EINVAL, "resident bases take one MSM per call");
  // sets > 1 (curdle_msm_g1_multi): d_points holds `sets` base sets of h_off[k] points each, all
  // multiplied by the SAME scalars: recoded and sorted once, accumulated per set
  const size_t n_pairs = h_off[k];
  size_t n_max = 0;
  for (size_t j = 0; j < k; j++) {
    if (h_off[j + 1] < h_off[j]) return fail(CURDLE_EINVAL, "offsets not monotone at %zu", j);
    if (h_off[j + 1] - h_off[j] > n_max) n_max = h_off[j + 1] - h_off[j];
  }
  MsmPlan& p = S.plan;
  int rc = make_plan(p, n_pairs, k, n_max, c, win_begin, win_end, latency_mode, sets, many, join ? join->seg : 0, light_host, glv);
  if (rc) return rc;
  if (dfront && (p.two_level || k != 1 || sets != 1)) return fail(CURDLE_EINVAL, "internal: the fused accumulator front takes one small MSM");
  // k_scan_one is 16 waves of 121 registers: a block of it needs four SIMDs of one compute unit EMPTY, so beside another
  // call's accumulation it waits for accumulate waves to end.  Only calls that have the chip to themselves take it.
  if (p.fuse_scan == 2 && (join || !latency_mode)) p.fuse_scan = 3;  // k_scan_chain: four 60-register waves (L >= 2 holds: make_plan)
  // a chunk of a host-buffer call sorts (and folds) beside the chunks before it: raised like a pipelined call's sort
  if (join && knobs::get(knobs::AUX_PRIO) < 0) p.aux_prio = 3;
  // the kernels work on the GLV split's terms, two per pair (records and digits 2 i, 2 i + 1)
  const size_t n = 2 * n_pairs;
  const size_t kr = k * sets;
  S.run_stream = tail;
  S.profiled = false;
  const uint32_t nw = p.win_end - p.win_begin;
  if (n == 0 || nw == 0) return CURDLE_OK;  // finish_slot writes infinities
  const size_t nb = k * (size_t)p.NB;
  if (nb > (size_t)1024 * 4096)  // run_passes cuts larger batches; a caller that gets here skipped it
    return fail(CURDLE_EINVAL, "%zu bucket slots exceed the scan capacity of one pass", nb);
  const size_t nlanes = ((size_t)nw * n + p.L - 1) / p.L;
  if ((rc = ensure(S.offsets, (k + 1) * 4))) return rc;
  if ((rc = ensure(S.counts, nb * 4))) return rc;
  if ((rc = ensure(S.starts, (nb + 1) * 4))) return rc;
  if ((rc = ensure(S.cursor, nb * 4))) return rc;
  if ((rc = ensure(S.fragcnt, nb * 4))) return rc;
  if ((rc = ensure(S.foff, (nb + 1) * 4))) return rc;
  if ((rc = ensure(S.small, (1024 + 1 + (size_t)p.max_large) * 4))) return rc;
  if ((rc = ensure(S.digits, (size_t)nw * n * 4))) return rc;
  if ((rc = ensure(S.sorted, (size_t)nw * n * 4))) return rc;
  if (p.two_level) {
    if ((rc = ensure(S.tmp, (size_t)nw * n * 4))) return rc;
    // the bins' cursors, their packed starts + sentinel, and the coarse counts + ticket of k_digits, which must be
    // zero before the call's first launch: the kernel leaves them zero, so they are cleared only when the buffer is
    // made (or moved), when the window count changes their place, and after a call that failed half-way
    const void* before = S.ccur.p;
    if ((rc = ensure(S.ccur, coarse_words(nw) * 4))) return rc;
    if (S.ccur.p != before || S.coarse_nw != nw || S.coarse_dirty) {
      HIP_TRY(hipMemsetAsync(S.ccur.p, 0, coarse_words(nw) * 4, pre));
      S.coarse_nw = nw;
    }
    S.coarse_dirty = true;  // until this call's kernels are all enqueued
  }
  p.frag_stride = (uint32_t)(nb + nlanes + 1);
  if (!ext_points28 && (rc = ensure(S.points28, sets * n * kA28Bytes))) return rc;
  if ((rc = ensure(S.frags, sets * (size_t)p.frag_stride * kX28Bytes))) return rc;
  // what leaves the GPU per window: one sum, or the reduce_bits form's nout bit-positioned points
  const size_t wpts = p.reduce_bits ? p.nout : 1;
  if ((rc = ensure(S.partials, (kr * (size_t)p.NS / p.G * (p.reduce_bits ? 2 * (2 + p.lgG) : 1) + 2) * kX28Bytes))) return rc;
  if ((rc = ensure(S.winsums, kr * (size_t)nw * wpts * sizeof(G1XYZZ)))) return rc;
  if (p.gpu_combine) {
    if ((rc = ensure(S.winsums28, kr * (size_t)nw * kX28Bytes))) return rc;
    if ((rc = ensure(S.results, kr * sizeof(G1XYZZ)))) return rc;
  }
  const size_t host_need = (p.gpu_combine ? kr * sizeof(G1XYZZ) : kr * (size_t)nw * wpts * sizeof(G1XYZZ)) + (k + 1) * 4;
  if (S.h_buf_cap < host_need) {
    if (S.h_buf) HIP_TRY(hipHostFree(S.h_buf));
    S.h_buf = nullptr;
    S.h_buf_cap = 0;
    HIP_TRY(hipHostMalloc(&S.h_buf, grow_size(host_need), hipHostMallocDefault));
    S.h_buf_cap = grow_size(host_need);
  }
  MsmWorkspace ws;
  ws.offsets = (const uint32_t*)S.offsets.p;
  ws.counts = (uint32_t*)S.counts.p;
  ws.starts = (uint32_t*)S.starts.p;
  ws.cursor = (uint32_t*)S.cursor.p;
  ws.fragcnt = (uint32_t*)S.fragcnt.p;
  ws.foff = (uint32_t*)S.foff.p;
  ws.blocksum = (uint32_t*)S.small.p;
  ws.nlarge = ws.blocksum + 1024;
  ws.large = ws.blocksum + 1025;
  ws.digits = (uint32_t*)S.digits.p;
  ws.sorted = (uint32_t*)S.sorted.p;
  ws.tmp = p.two_level ? (uint32_t*)S.tmp.p : nullptr;
  ws.ccur = p.two_level ? (uint32_t*)S.ccur.p : nullptr;
  ws.points28 = ext_points28 ? const_cast<void*>(ext_points28) : S.points28.p;
  ws.frags = S.frags.p;
  ws.partials = S.partials.p;
  ws.winsums28 = S.winsums28.p;
  // Round 5: what the host's Horner pass reads leaves the GPU by the kernels' own stores into the slot's pinned buffer
  // (device-visible like all pinned memory here) -- a hundred points or so, 16 bytes per store -- instead of through a
  // device array and a copy command behind the last kernel (~10 us of every synchronous call; knob DIRECT_RESULTS=0: the copy).
  const size_t win_bytes = kr * (size_t)nw * wpts * sizeof(G1XYZZ);
  const bool direct = !p.gpu_combine && win_bytes <= ((size_t)256 << 10);
  ws.winsums = direct ? (G1XYZZ*)S.h_buf : (G1XYZZ*)S.winsums.p;
  ws.results = (G1XYZZ*)S.results.p;
  ws.chain = nullptr;
  ws.chain_ticket = nullptr;
  ws.host_err = nullptr;
  ws.chain_base = ws.chain_epoch = 0;
  if (p.fuse_scan == 3) {
    const void* before = S.chain.p;
    if ((rc = ensure(S.chain, scan_chain_bytes()))) return rc;
    if (!S.h_err) {
      HIP_TRY(hipHostMalloc((void**)&S.h_err, 64, hipHostMallocDefault));
      *S.h_err = 0;
    }
    S.scan_epoch = (S.scan_epoch + 1) & 0x3fffffffu;
    // (chain_dirty, review of round 5: a call that failed after its k_scan_chain was enqueued -- launch_scan returns
    // hipGetLastError(), which may be an EARLIER launch's error -- left scan_base behind the device's counter, and every
    // later launch of the slot would have taken tickets beyond its tile count)
    if (S.chain.p != before || S.scan_epoch == 0 || S.chain_dirty) {  // a new buffer, or the epochs have gone round: no word may look current
      HIP_TRY(hipMemsetAsync(S.chain.p, 0, scan_chain_bytes(), pre));
      S.scan_base = 0;
      if (S.scan_epoch == 0) S.scan_epoch = 1;
    }
    ws.chain = (unsigned long long*)S.chain.p;
    ws.chain_ticket = (uint32_t*)((char*)S.chain.p + scan_chain_bytes() - 64);
    ws.host_err = S.h_err;
    ws.chain_base = S.scan_base;
    ws.chain_epoch = S.scan_epoch;
    S.chain_dirty = true;  // until this call's kernels are all enqueued (enqueue_slot)
  }
  {
    const void* before = S.mdone.p;
    if ((rc = ensure(S.mdone, sets * (size_t)p.max_large * 4))) return rc;
    if (S.mdone.p != before) HIP_TRY(hipMemsetAsync(S.mdone.p, 0, S.mdone.cap, pre));
    ws.mdone = (uint32_t*)S.mdone.p;
  }

  // the offsets are staged in pinned memory (tail of h_buf) so the copy is truly asynchronous
  uint32_t* h_off_pinned = (uint32_t*)((char*)S.h_buf + host_need - (k + 1) * 4);
  if (k > 1) {  // a single MSM's kernels take [0, n) from the plan
    for (size_t j = 0; j <= k; j++) h_off_pinned[j] = 2 * h_off[j];  // in terms, like everything the kernels index
    HIP_TRY(hipMemcpyAsync(S.offsets.p, h_off_pinned, (k + 1) * 4, hipMemcpyHostToDevice, pre));
  }
  // counts are cleared by k_digits, the large-bucket counter by the scan
  Prof prof(S, pre, cx.profile);
  // Two ways of overlapping the phases INSIDE one synchronous call were built and measured in
  // round 3, and removed again (profiles/r03_sync_groups_experiment.txt).  (1) The windows of one
  // call in G groups with their own sort -> accumulate -> reduce chains on three streams, so that
  // group g + 1 is sorted and group g - 1 reduced while group g accumulates: 3.64 ms as one
  // chain, 3.88 / 4.55 / 5.18 ms in 2 / 4 / 8 groups -- the latency-bound kernels crawl beside a
  // full-chip accumulation (a single-block scan 0.26-0.6 ms instead of 0.04, window sums 0.3-0.4
  // instead of 0.05, every group's reduce 0.45-0.67), the accumulations stretch from 2.65 to
  // 3.37 ms in all and the tails queue up behind each other.  (2) The point conversion on a
  // second stream beside the sort, which never reads a point: 3.53 -> 3.67 ms at 2^20, nothing
  // at 2^17..2^19 -- conversion and sort are both HBM-bound, so side by side they take as long
  // as one after the other, plus two event hops.
  const int phase = join ? join->phase : 0;
  const bool convert_here = !points28_ready && !ext_points28;  // the device accumulator fills S.points28 itself; a resident base set is converted already
  // small calls: conversion and recoding in one launch (the host's launches bound the call until the accumulation
  // starts)
  const size_t front_max = 16384;  // (larger limits measured equal: profiles/r05_small_sort_one_block.txt)
  const bool front = convert_here && phase == 0 && !p.two_level && sets * n_pairs <= front_max;
  if (convert_here && phase == 0 && !front && !EXP_SKIP(1)) {
    HIP_TRY(launch_convert_points_raw(d_points, (uint32_t)(sets * n_pairs), ws.points28, pre, p.aux_prio));
    prof.mark("convert_points");
  }
  if (phase != 2 && !EXP_SKIP(2)) {
    if (front)
      HIP_TRY(launch_front(p, ws, d_points, (uint32_t)(sets * n_pairs), d_scalars, pre));
    else if (dfront)  // the device accumulator's job: loose bases, slot scalars and recoding in one launch
      HIP_TRY(launch_dacc_front(p, ws, *dfront, pre));
    else
      HIP_TRY(launch_digits(p, ws, d_scalars, pre));
    prof.mark("digits");
    HIP_TRY(launch_hist(p, ws, pre));
    prof.mark("hist");
    HIP_TRY(launch_scan(p, ws, pre));
    if (ws.chain) S.scan_base += scan_chain_tiles((uint32_t)nb);  // the tickets that launch takes
    prof.mark("scan");
    HIP_TRY(launch_scatter(p, ws, pre));
    prof.mark("scatter");
    if (phase == 1) {  // the sorted list waits in the slot; the second step comes when the points are there
      HIP_TRY(hipEventRecord(S.pre_done, pre));
      return CURDLE_OK;
    }
  }
  if (phase == 2) {  // the caller has made `pre` (the chunk's sort stream) and `stream` wait for the chunk's points
    if (convert_here) {
      // on the sort stream, beside the accumulation of the chunk before -- on `stream` the four conversions of a 2^20-pair
      // call sat BETWEEN the accumulations, 0.18 ms of the call's critical path (timeline gpurun_out/r5_hosttrace2)
      HIP_TRY(launch_convert_points_raw(d_points, (uint32_t)(sets * n_pairs), ws.points28, pre, p.aux_prio));
      prof.mark("convert_points");
      HIP_TRY(hipEventRecord(S.pre_done, pre));  // behind the sort's record on the same stream: covers both
    }
    HIP_TRY(hipStreamWaitEvent(stream, S.pre_done, 0));
    prof.st = stream;
  } else if (stream != pre) {
    HIP_TRY(hipEventRecord(S.pre_done, pre));
    HIP_TRY(hipStreamWaitEvent(stream, S.pre_done, 0));
    prof.st = stream;
    prof.mark("(queue)");  // not a kernel: time this MSM waited for the accumulate stream
  }
  prof.before_dominant();
  HIP_TRY(launch_accumulate(p, ws, stream));
  prof.mark("accumulate");
  if (tail != stream) {
    HIP_TRY(hipEventRecord(S.acc_done, stream));
    HIP_TRY(hipStreamWaitEvent(tail, S.acc_done, 0));
    stream = tail;
    prof.st = tail;
  }
  if (!EXP_SKIP(4)) HIP_TRY(launch_merge_large(p, ws, stream, latency_mode && !join));
  prof.mark("merge_large");
  if (join && join->accumulate_only) {
    if (join->fold_home) {
      Slot& H = *join->fold_home;
      if (p.k != 1 || sets != 1) return fail(CURDLE_EINVAL, "folding takes one MSM per chunk");
      if (!join->fold_prev) {
        int rf;
        if ((rf = ensure(H.fold_sums, (size_t)p.NB * kX28Bytes))) return rf;
        if ((rf = ensure(H.fold_meta, 2 * (size_t)p.NB * 4))) return rf;
      } else {
        HIP_TRY(hipStreamWaitEvent(stream, join->fold_prev->acc_done, 0));  // the sums so far
      }
      HIP_TRY(launch_fold_fragments(p, ws, H.fold_sums.p, H.fold_meta.p, !join->fold_prev, stream));
      prof.mark("fold");
    }
    HIP_TRY(hipEventRecord(S.acc_done, stream));  // the fragments are complete (and folded): the next chunk / the last one waits for this
    return CURDLE_OK;
  }
  FragSources extra;
  memset(&extra, 0, sizeof(extra));
  if (join && join->fold_home && join->fold_prev) {
    const Slot& H = *join->fold_home;
    const MsmPlan& q = join->fold_prev->plan;
    if (q.c != p.c || q.NB != p.NB || q.seg != p.seg || q.k != 1 || p.k != 1 || sets != 1 || q.win_begin != p.win_begin ||
        q.win_end != p.win_end || H.fold_sums.cap < (size_t)p.NB * kX28Bytes)
      return fail(CURDLE_EINVAL, "chunks of one MSM must share the plan");
    extra.frags[0] = H.fold_sums.p;
    extra.foff[0] = (const uint32_t*)H.fold_meta.p;
    extra.fragcnt[0] = (const uint32_t*)H.fold_meta.p + p.NB;
    extra.n = 1;
    HIP_TRY(hipStreamWaitEvent(stream, join->fold_prev->acc_done, 0));
  } else if (join) {
    for (Slot* E : join->earlier) {
      if (extra.n >= (uint32_t)kMaxFragSources - 1) return fail(CURDLE_EINVAL, "too many chunks for one reduction");
      const MsmPlan& q = E->plan;
      if (q.c != p.c || q.NB != p.NB || q.seg != p.seg || q.k != 1 || p.k != 1 || sets != 1 || q.win_begin != p.win_begin ||
          q.win_end != p.win_end)
        return fail(CURDLE_EINVAL, "chunks of one MSM must share the plan");
      extra.frags[extra.n] = E->frags.p;
      extra.foff[extra.n] = (const uint32_t*)E->foff.p;
      extra.fragcnt[extra.n] = (const uint32_t*)E->fragcnt.p;
      extra.n++;
      HIP_TRY(hipStreamWaitEvent(stream, E->acc_done, 0));
    }
  }
  if (p.reduce_bits) {
    if (!EXP_SKIP(8)) HIP_TRY(launch_reduce_segments(p, ws, stream, extra.n ? &extra : nullptr));
    prof.mark("bucket_reduce");
    if (!EXP_SKIP(16)) HIP_TRY(launch_reduce_groups(p, ws, stream));
    prof.mark("window_sum");
  } else {
    HIP_TRY(launch_bucket_reduce(p, ws, stream, extra.n ? &extra : nullptr));
    prof.mark("bucket_reduce");
    HIP_TRY(launch_window_sum(p, ws, stream));
    prof.mark("window_sum");
  }
  if (p.gpu_combine) {
    HIP_TRY(launch_combine(p, ws, stream));
    prof.mark("combine");
    HIP_TRY(hipMemcpyAsync(S.h_buf, ws.results, kr * sizeof(G1XYZZ), hipMemcpyDeviceToHost, stream));
  } else if (!direct) {
    HIP_TRY(hipMemcpyAsync(S.h_buf, ws.winsums, win_bytes, hipMemcpyDeviceToHost, stream));
  }
  return CURDLE_OK;
}

// Wait for the slot's GPU work and produce the k results (host combine unless the
// batch combined on the GPU).
int finish_slot(Ctx& cx, Slot& S, uint64_t* out) {
  const MsmPlan& p = S.plan;
  const size_t k = p.kr;  // results
  const uint32_t nw = p.win_end - p.win_begin;
  if (p.n == 0 || nw == 0) {
    for (size_t j = 0; j < k; j++) set_out_infinity(out + 18 * j);
    return CURDLE_OK;
  }
  HIP_TRY(hipStreamSynchronize(S.run_stream));
  if (S.h_err && *S.h_err) {
    *S.h_err = 0;
    return fail(CURDLE_EHIP, "internal: a wait inside the bucket-slot scan gave up");
  }
  if (S.profiled) {
    std::lock_guard<std::mutex> g(cx.mu);
    curdle_profile& L = cx.last;
    L.n_kernels = S.prof_n;
    for (int i = 0; i < S.prof_n; i++) {
      L.name[i] = S.prof_name[i];
      (void)hipEventElapsedTime(&L.ms[i], S.ev[i], S.ev[i + 1]);
    }
    L.window_bits = p.c;
    L.num_windows = p.W;
    L.entries = L.fragments = 0;
    if (cx.profile == 1) {  // two 4-byte reads after the call has drained: diagnostics only
      const size_t nb = (size_t)p.k * p.NB;
      uint32_t v[2] = {0, 0};
      (void)hipMemcpy(&v[0], (const char*)S.starts.p + nb * 4, 4, hipMemcpyDeviceToHost);
      (void)hipMemcpy(&v[1], (const char*)S.foff.p + nb * 4, 4, hipMemcpyDeviceToHost);
      L.entries = v[0];
      L.fragments = v[1];
    }
  }
  if (p.gpu_combine) {
    // the GPU ran the Horner passes; one shared inversion normalises the whole batch
    std::vector<G1Affine> aff(k);
    curdle_host_batch_to_affine(aff.data(), S.h_buf, k);
    for (size_t j = 0; j < k; j++) {
      G1Jac r;
      if (g1_affine_is_inf(aff[j])) {
        f_one(r.x);
        f_one(r.y);
        f_zero(r.z);
      } else {
        r.x = aff[j].x;
        r.y = aff[j].y;
        f_one(r.z);
      }
      memcpy(out + 18 * j, &r, sizeof(r));
    }
    return CURDLE_OK;
  }
  // Window combine on the host: Horner from the top window down, each step shifting
  // by the width of the window below, then the 2^shift scaling of a partial
  // (host/host_ops.cpp).
  if (p.reduce_bits) {
    // every window arrived as nout points with bit positions (reduce_bits_position; -1 = unused slot,
    // which holds infinity): one Horner pass over all of them, top bit first, like over window sums
    int dbls[kMaxWindows * 16];
    const uint32_t np = nw * p.nout;
    int prev = 0;
    for (uint32_t i = 0; i < np; i++) {
      const int w = p.win_begin + (int)(i / p.nout);
      const int rel = reduce_bits_position(p, w, i % p.nout);
      const int pos = rel < 0 ? prev : (int)p.shift[w] + rel;
      if (pos < prev) return fail(CURDLE_EHIP, "internal: bit positions of the reduction are not monotone");
      dbls[i] = pos - prev;
      prev = pos;
    }
    for (size_t j = 0; j < k; j++)
      curdle_window_combine((const G1XYZZ*)S.h_buf + j * np, (int)np, dbls, out + 18 * j);
    return CURDLE_OK;
  }
  int dbls[kMaxWindows];
  for (uint32_t lw = 0; lw < nw; lw++) dbls[lw] = lw > 0 ? p.bits[p.win_begin + lw - 1] : p.shift[p.win_begin];
  for (size_t j = 0; j < k; j++)
    curdle_window_combine((const G1XYZZ*)S.h_buf + j * nw, (int)nw, dbls, out + 18 * j);
  return CURDLE_OK;
}

void drain_slot(Ctx& cx, Slot& S) {
  (void)hipStreamSynchronize(cx.h2d_stream);
  (void)hipStreamSynchronize(cx.pre_stream);
  (void)hipStreamSynchronize(cx.pre_stream2);
  (void)hipStreamSynchronize(cx.main_stream);
  for (auto& st : cx.main_extra)
    if (st) (void)hipStreamSynchronize(st);
  (void)hipStreamSynchronize(S.stream);
}

// Streams of a synchronous call.  The caller waits for this very call, so every phase goes to
// the slot's own stream: no event hops between streams (each costs a cross-queue dependency,
// ~10 us with 16 hardware queues: 0.58 -> 0.50 ms for a 1,268-pair MSM), and concurrent callers
// still overlap, each on its slot's stream.  (The three-stream layout of the pipelined entry points was
// measured slower for synchronous calls in round 2; its knob is gone.)
struct SyncStreams {
  hipStream_t pre, main, tail;
};
SyncStreams sync_streams(Ctx&, Slot& S) { return {S.stream, S.stream, S.stream}; }

// The scans of the bucket slots hold 1,024 blocks of 4,096 slots: a batch with more slots than
// that (1,024 MSMs of 2,548 pairs; 2,100 of 628) runs in passes of as many whole MSMs as fit,
// one after the other on the caller's slot -- every pass is milliseconds of GPU work, so the
// gap between two passes is noise, and the workspaces stay bounded.
constexpr size_t kMaxSlotsPerPass = (size_t)1024 * 4096;

// enqueue + finish of k MSMs on slot S, in passes if the batch is too large for one.
int run_passes(Ctx& cx, Slot& S, const void* d_points, const void* d_scalars, const uint32_t* h_off, size_t k, int c,
               int win_begin, int win_end, hipStream_t pre, hipStream_t main, hipStream_t tail, uint64_t* out,
               const void* ext_points28 = nullptr, bool glv = true) {
  if (k > 1) {
    size_t n_max = 0;
    for (size_t j = 0; j < k; j++) {
      if (h_off[j + 1] < h_off[j]) return fail(CURDLE_EINVAL, "offsets not monotone at %zu", j);
      if (h_off[j + 1] - h_off[j] > n_max) n_max = h_off[j + 1] - h_off[j];
    }
    MsmPlan probe;
    int rc = make_plan(probe, h_off[k] - h_off[0], k, n_max, c, win_begin, win_end, true);
    if (rc) return rc;
    size_t per_pass = probe.NB ? kMaxSlotsPerPass / probe.NB : k;
    if (knobs::get(knobs::MAX_MSMS_PER_PASS) > 0) per_pass = (size_t)knobs::get(knobs::MAX_MSMS_PER_PASS);
    if (k > per_pass) {
      std::vector<uint32_t> off;
      for (size_t j0 = 0; j0 < k; j0 += per_pass) {
        const size_t kg = k - j0 < per_pass ? k - j0 : per_pass;
        off.resize(kg + 1);
        for (size_t j = 0; j <= kg; j++) off[j] = h_off[j0 + j] - h_off[j0];
        // the whole batch's window width for every pass (a pass's own n_max must not change it)
        rc = enqueue_slot(cx, S, (const char*)d_points + (size_t)h_off[j0] * 96, (const char*)d_scalars + (size_t)h_off[j0] * 32,
                          off.data(), kg, probe.c, win_begin, win_end, pre, main, tail, true, false, 1,
                          /*many=*/true);
        if (!rc) rc = finish_slot(cx, S, out + 18 * j0);
        if (rc) return rc;
      }
      return CURDLE_OK;
    }
  }
  int rc = enqueue_slot(cx, S, d_points, d_scalars, h_off, k, c, win_begin, win_end, pre, main, tail, true, false, 1, false,
                        nullptr, ext_points28, false, glv);
  if (!rc) rc = finish_slot(cx, S, out);
  return rc;
}

// Synchronous run of k MSMs with inputs on the device.
int run_device(const void* d_points, const void* d_scalars, const uint32_t* h_off, size_t k, int c, int win_begin,
               int win_end, uint64_t* out, void* user_stream, const void* ext_points28 = nullptr, bool glv = true,
               hipEvent_t wait_for = nullptr) {
  Ctx& cx = cur();
  int idx;
  int rc = acquire_slot(cx, true, &idx);
  if (rc) return rc;
  Slot& S = cx.slots[idx];
  hipError_t he = hipSetDevice(cx.device);
  if (he != hipSuccess) {
    release_slot(cx, idx);
    return fail(CURDLE_EHIP, "hipSetDevice: %s", hipGetErrorString(he));
  }
  if (user_stream) {
    if (wait_for) he = hipStreamWaitEvent((hipStream_t)user_stream, wait_for, 0);
    rc = he != hipSuccess ? fail(CURDLE_EHIP, "hipStreamWaitEvent: %s", hipGetErrorString(he))
                          : run_passes(cx, S, d_points, d_scalars, h_off, k, c, win_begin, win_end, (hipStream_t)user_stream,
                                       (hipStream_t)user_stream, (hipStream_t)user_stream, out, ext_points28, glv);
  } else {
    const SyncStreams st = sync_streams(cx, S);
    if (wait_for) he = hipStreamWaitEvent(st.pre, wait_for, 0);
    rc = he != hipSuccess ? fail(CURDLE_EHIP, "hipStreamWaitEvent: %s", hipGetErrorString(he))
                          : run_passes(cx, S, d_points, d_scalars, h_off, k, c, win_begin, win_end, st.pre, st.main, st.tail, out,
                                       ext_points28, glv);
  }
  if (rc) drain_slot(cx, S);
  release_slot(cx, idx);
  return rc;
}

// Synchronous run with inputs in host memory: staged through the slot's own buffers.
int run_host(const uint64_t* points, const uint64_t* scalars, const uint32_t* h_off, size_t k, uint64_t* out, bool glv = true) {
  Ctx& cx = cur();
  int idx;
  int rc = acquire_slot(cx, true, &idx);
  if (rc) return rc;
  Slot& S = cx.slots[idx];
  const size_t n = h_off[k];
  auto body = [&]() -> int {
    HIP_TRY(hipSetDevice(cx.device));
    int r;
    if ((r = ensure(S.points, n * 96))) return r;
    if ((r = ensure(S.scalars, n * 32))) return r;
    const SyncStreams st = sync_streams(cx, S);
    // One mid-size MSM: the scalars cross first and the recoding + sort run while the points are still crossing (on the
    // context's copy stream; a pageable copy occupies this thread, not the GPU) -- the sort, 0.07-0.15 ms of such a call,
    // is off the call's critical path for one event hop.  From 16,384 pairs.
    const size_t overlap_min = 16384;
    if (k == 1 && n >= overlap_min) {
      const uint32_t off[2] = {0, (uint32_t)n};
      ChunkJoin join;
      HIP_TRY(hipMemcpyAsync(S.scalars.p, scalars, n * 32, hipMemcpyHostToDevice, st.pre));
      join.phase = 1;
      if ((r = enqueue_slot(cx, S, S.points.p, S.scalars.p, off, 1, 0, 0, -1, st.pre, st.main, st.tail, /*latency_mode=*/true, false, 1,
                            false, &join, nullptr, false, glv)))
        return r;
      HIP_TRY(hipMemcpyAsync(S.points.p, points, n * 96, hipMemcpyHostToDevice, cx.h2d_stream));
      HIP_TRY(hipEventRecord(S.acc_done, cx.h2d_stream));  // (a scratch event until the accumulation re-records it)
      HIP_TRY(hipStreamWaitEvent(st.pre, S.acc_done, 0));
      join.phase = 2;
      if ((r = enqueue_slot(cx, S, S.points.p, S.scalars.p, off, 1, 0, 0, -1, st.pre, st.main, st.tail, /*latency_mode=*/true, false, 1,
                            false, &join, nullptr, false, glv)))
        return r;
      return finish_slot(cx, S, out);
    }
    HIP_TRY(hipMemcpyAsync(S.points.p, points, n * 96, hipMemcpyHostToDevice, st.pre));
    HIP_TRY(hipMemcpyAsync(S.scalars.p, scalars, n * 32, hipMemcpyHostToDevice, st.pre));
    return run_passes(cx, S, S.points.p, S.scalars.p, h_off, k, 0, 0, -1, st.pre, st.main, st.tail, out, nullptr, glv);
  };
  rc = body();
  if (rc) drain_slot(cx, S);
  release_slot(cx, idx);
  return rc;
}

// One large MSM from HOST buffers (what a cgo caller hands over): the pairs go to the GPU in
// point-range chunks, each an MSM of its own on the submit / wait pipeline, so that chunk i + 1
// crosses PCIe while chunk i is being accumulated -- a copy from pageable memory occupies the
// calling thread, not the GPU -- and the partial sums are added on the host.  At N = 2^20 the
// copy (128 MiB) costs more than the arithmetic.
constexpr size_t kHostChunkMin = (size_t)1 << 19;  // below this a call is one chunk
int run_host_chunked(const uint64_t* points, const uint64_t* scalars, size_t n, uint64_t* out, bool glv = true) {
  Ctx& cx = cur();
  // Round 2 ran every chunk as an MSM of its own and added the results: 6.4 ms in one copy, 5.8
  // in two chunks, 6.5 in four at N = 2^20 on a 29 GB/s link -- two half-size MSMs cost more than
  // one whole (each has the full set of buckets to reduce).  Now the chunks share ONE plan (the
  // whole call's window width) and ONE bucket reduction: a chunk is sorted and accumulated into
  // fragments while the next one is copied, and the last chunk's reduction folds in the fragment
  // lists of all of them.  tools/bench_sync_call.py --variants CURDLE_HOST_CHUNKS=...:
  // see profiles/r03_host_buffer_chunks.txt.
  // Round 5: GRADED chunks, pairs and points interleaved.  The GPU cannot start before the first chunk has landed
  // (0.69 ms of a 4.5 ms call with four equal chunks) and, once it runs, it is the slower side of the pipeline (four
  // quarter accumulations 2.9 ms against 2.4 ms of copies): so the first two chunks are half-size -- the GPU starts
  // after an eighth of the bytes -- and every chunk's scalars cross right before its points, so that no accumulation
  // waits for points queued behind other chunks' scalars (the gap 1.43-1.78 ms in profiles/r04_host_fold.txt).
  // Round 5, late: graded chunks by DEFAULT, together with two things that were missing when they were first measured (and lost
  // to round 4's order): the chunks' sorts raised to wave priority 3 (they run beside the accumulation of the chunk before;
  // enqueue_slot) and every sort on its chunk's own stream instead of all of them in a row on the context's sort stream.
  // N = 2^20: 4.24-4.26 -> 4.03-4.18 ms on one box (four chunks of 1/6, 1/6, 1/3, 1/3), 2^19: 2.63-2.69 -> 2.51-2.53
  // (profiles/r05_host_buffer_call.txt, the last section).
  // (Round 6: the knobs HOST_GRADED, HOST_PATTERN and HOST_SORT_STREAMS that walked these choices are gone; equal chunks, round 4's copy
  // order, other size patterns and the one sort stream are all in profiles/r05_host_buffer_call.txt.)
  size_t nchunks = n >= ((size_t)1 << 20) ? 4 : 3;
  if (knobs::get(knobs::HOST_CHUNKS) > 0) nchunks = (size_t)knobs::get(knobs::HOST_CHUNKS);
  // (without folding the reduction takes one fragment list per chunk: at most kMaxFragSources)
  const bool fold_on = knobs::get(knobs::HOST_FOLD) != 0;
  if (nchunks > (fold_on ? (size_t)kSlots : (size_t)kMaxFragSources)) nchunks = fold_on ? kSlots : kMaxFragSources;
  // every chunk needs a slot until the reduction has read its fragments: take what is free now
  // (never wait for a slot while holding one), at least one
  std::vector<int> slots;
  {
    int idx = -1;
    int rc = acquire_slot(cx, true, &idx);
    if (rc) return rc;
    slots.push_back(idx);
    while (slots.size() < nchunks && acquire_slot(cx, false, &idx) == CURDLE_OK) slots.push_back(idx);
    // ... and leave other callers some: with fewer than three slots free afterwards, two chunks do
    size_t busy = 0;
    {
      std::lock_guard<std::mutex> g(cx.mu);
      for (const Slot& x : cx.slots) busy += x.busy ? 1 : 0;
    }
    while (slots.size() > 2 && kSlots - busy < 3) {
      release_slot(cx, slots.back());
      slots.pop_back();
      busy--;
    }
    nchunks = slots.size();
  }
  const int c = choose_window_bits(n);
  // chunk sizes: (three chunks and more) the first two one unit and the others two units each; otherwise equal
  std::vector<size_t> bounds(nchunks + 1, n);
  bounds[0] = 0;
  if (nchunks >= 3) {
    const size_t unit = (n + 2 * (nchunks - 1) - 1) / (2 * (nchunks - 1));
    size_t at = 0;
    for (size_t i = 0; i < nchunks; i++) {
      at += i < 2 ? unit : 2 * unit;
      bounds[i + 1] = at < n ? at : n;
    }
  } else {
    const size_t per = (n + nchunks - 1) / nchunks;
    for (size_t i = 1; i < nchunks; i++) bounds[i] = i * per < n ? i * per : n;
  }
  bounds[nchunks] = n;
  auto body = [&]() -> int {
    HIP_TRY(hipSetDevice(cx.device));
    // A pageable copy occupies the thread that issues it, so the order of the copies below is the timeline.
    struct Part {
      Slot* S;
      size_t lo, m;
      ChunkJoin join;
      hipStream_t main;
    };
    std::vector<Part> parts;
    for (size_t i = 0; i < nchunks; i++) {
      if (bounds[i + 1] <= bounds[i]) continue;  // (tiny n under a forced chunk count)
      Part pt;
      pt.S = &cx.slots[slots[parts.size()]];
      pt.lo = bounds[i];
      pt.m = bounds[i + 1] - bounds[i];
      const unsigned seq = cx.submit_count.fetch_add(1, std::memory_order_relaxed);
      const unsigned turn = seq % (unsigned)cx.main_streams;
      pt.main = turn == 0 ? cx.main_stream : cx.main_extra[turn - 1];
      // with three or four fragment lists per bucket the reduction's chain is fragments, not
      // running sums: half as many buckets per quad (N = 2^20, four chunks: 5.30 -> 5.02 ms)
      pt.join.seg = nchunks >= 3 ? 8 : 0;
      parts.push_back(pt);
    }
    int r;
    for (Part& pt : parts) {
      if ((r = ensure(pt.S->points, pt.m * 96))) return r;
      if ((r = ensure(pt.S->scalars, pt.m * 32))) return r;
    }
    // The copies are issued by a thread of their own, back to back (a pageable copy occupies the thread
    // that issues it for its whole duration): first chunk 0 whole -- the GPU starts after a quarter of
    // the bytes --, then the scalars of all other chunks, whose sorts run beside the first accumulation
    // and are done long before their points land, then the point chunks.  This thread queues each
    // step's kernels as soon as the copy it needs has been issued and its event recorded.  With the
    // copies and the launches on ONE thread the ~0.1 ms of launches per step sat between the copies, and
    // the GPU idled 0.6 ms between the first and the second accumulation (profiles/r04_host_buffer_call.txt).
    struct CopyJob {
      void* dst;
      const void* src;
      size_t bytes;
      hipEvent_t ev;
    };
    std::vector<CopyJob> jobs;
    std::vector<size_t> job_s(parts.size()), job_p(parts.size());  // which job carries chunk i's scalars / points
    auto add_s = [&](size_t i) {
      job_s[i] = jobs.size();
      jobs.push_back({parts[i].S->scalars.p, scalars + 4 * parts[i].lo, parts[i].m * 32, parts[i].S->pre_done});
    };
    auto add_p = [&](size_t i) {  // (acc_done is re-recorded by the chunk's accumulation: a scratch event until then)
      job_p[i] = jobs.size();
      jobs.push_back({parts[i].S->points.p, points + 12 * parts[i].lo, parts[i].m * 96, parts[i].S->acc_done});
    };
    for (size_t i = 0; i < parts.size(); i++) {  // every chunk's scalars right before its points
      add_s(i);
      add_p(i);
    }
    std::mutex cmu;
    std::condition_variable ccv;
    size_t issued = 0;
    hipError_t copy_err = hipSuccess;
    std::thread copier([&] {
      hipError_t e = hipSetDevice(cx.device);
      for (size_t j = 0; j < jobs.size(); j++) {
        if (e == hipSuccess) e = hipMemcpyAsync(jobs[j].dst, jobs[j].src, jobs[j].bytes, hipMemcpyHostToDevice, cx.h2d_stream);
        if (e == hipSuccess) e = hipEventRecord(jobs[j].ev, cx.h2d_stream);
        {
          std::lock_guard<std::mutex> g(cmu);
          issued = j + 1;
          if (e != hipSuccess && copy_err == hipSuccess) copy_err = e;
        }
        ccv.notify_all();
      }
    });
    struct Joiner {  // on every way out
      std::thread& t;
      ~Joiner() { t.join(); }
    } joiner{copier};
    auto wait_copy = [&](size_t j) -> int {  // copy j has been issued and its event recorded
      std::unique_lock<std::mutex> g(cmu);
      ccv.wait(g, [&] { return issued > j; });
      if (copy_err != hipSuccess) return fail(CURDLE_EHIP, "host-buffer chunk copy: %s", hipGetErrorString(copy_err));
      return CURDLE_OK;
    };
    ChunkJoin last;
    // (knob HOST_FOLD=0: the reduction walks every chunk's fragment list, as until round 4)
    const bool fold = knobs::get(knobs::HOST_FOLD) != 0 && parts.size() >= 2;
    // every chunk's sort on its slot's own stream, not all of them one after the other on the context's sort stream: three
    // sorts in a row beside the accumulations are late
    auto sort_stream = [&](Part& pt) { return &pt != &parts[0] ? pt.S->stream : cx.pre_stream; };
    auto enqueue_sort = [&](Part& pt) -> int {  // behind the chunk's scalars
      hipStream_t ps = sort_stream(pt);
      HIP_TRY(hipStreamWaitEvent(ps, pt.S->pre_done, 0));
      const uint32_t off[2] = {0, (uint32_t)pt.m};
      pt.join.phase = 1;
      // every chunk takes the synchronous rule for its segments (the reduction walks all of them with one plan)
      return enqueue_slot(cx, *pt.S, pt.S->points.p, pt.S->scalars.p, off, 1, c, 0, -1, ps, pt.main, pt.S->stream,
                          /*latency_mode=*/true, false, 1, false, &pt.join, nullptr, false, glv);
    };
    auto enqueue_accumulate = [&](size_t i) -> int {  // behind the chunk's points
      Part& pt = parts[i];
      HIP_TRY(hipStreamWaitEvent(pt.main, pt.S->acc_done, 0));
      HIP_TRY(hipStreamWaitEvent(sort_stream(pt), pt.S->acc_done, 0));  // the conversion runs there
      const uint32_t off[2] = {0, (uint32_t)pt.m};
      const bool is_last = i + 1 == parts.size();
      pt.join.phase = 2;
      pt.join.accumulate_only = !is_last;
      if (fold) {
        pt.join.fold_home = parts[0].S;
        pt.join.fold_prev = i ? parts[i - 1].S : nullptr;
      }
      if (is_last) pt.join.earlier = last.earlier;
      int rr = enqueue_slot(cx, *pt.S, pt.S->points.p, pt.S->scalars.p, off, 1, c, 0, -1, sort_stream(pt), pt.main, pt.S->stream,
                            /*latency_mode=*/true, false, 1, false, &pt.join, nullptr, false, glv);
      last.earlier.push_back(pt.S);
      return rr;
    };
    // every step's kernels are queued as soon as the copy they need has been issued: in the order of the jobs
    const size_t K = parts.size();
    for (size_t j = 0; j < jobs.size(); j++)
      for (size_t i = 0; i < K; i++) {
        if (job_s[i] == j) {
          if ((r = wait_copy(j))) return r;
          if ((r = enqueue_sort(parts[i]))) return r;
        } else if (job_p[i] == j) {
          if ((r = wait_copy(j))) return r;
          if ((r = enqueue_accumulate(i))) return r;
        }
      }
    r = finish_slot(cx, *parts.back().S, out);  // the last chunk's slot holds the window sums
    // every chunk ran a bucket-slot scan of its own on its own slot: a scan that gave up in an EARLIER chunk raised that
    // slot's word, which finish_slot above does not look at (review of round 5: the call returned a wrong sum with
    // CURDLE_OK and the stale flag failed the next, unrelated call on that slot).  The last chunk's reduction is behind
    // all of them, so every word is final here.
    for (size_t i = 0; i + 1 < K; i++) {
      Slot& Sp = *parts[i].S;
      if (Sp.h_err && *Sp.h_err) {
        *Sp.h_err = 0;
        if (!r) r = fail(CURDLE_EHIP, "internal: a wait inside the bucket-slot scan of chunk %zu gave up", i);
      }
    }
    return r;
  };
  int rc = body();
  for (int idx : slots) {
    if (rc) drain_slot(cx, cx.slots[idx]);
    release_slot(cx, idx);
  }
  return rc;
}

}  // namespace

// ---------------------------------------------------------------------------
// Batched point decoding (decode_kernels.hip)
// ---------------------------------------------------------------------------
extern "C" int curdle_g1_decompress_batch(const uint8_t* in, size_t n, int subgroup_check, uint64_t* out_affine,
                                          uint8_t* status) {
  Ctx& cx = cur();
  if (n && (!in || !out_affine || !status)) return fail(CURDLE_EINVAL, "null argument");
  if (n == 0) return CURDLE_OK;
  if (n > ((size_t)1 << 27)) return fail(CURDLE_EINVAL, "n = %zu exceeds the supported 2^27 points", n);
  // A batch small enough to be one latency-bound chain per point (four lanes per point still fit
  // one round of the chip) takes the two-kernel form of the three-step entry points: the square
  // roots and, beside them, the subgroup test on the twisted model -- two ~0.5 ms chains that
  // overlap (98 points: 1.10 -> 0.63 ms) where the fused kernel below runs them one after the
  // other.  With every decode context taken it falls through to the fused kernel.
  if (subgroup_check && n <= two_kernel_max()) {
    int ticket = -1;
    int rc2 = curdle_g1_decompress_begin(in, n, out_affine, status, &ticket);
    if (rc2 == CURDLE_OK) {
      rc2 = curdle_g1_decompress_finish(ticket, status);
      if (rc2 == CURDLE_OK)  // this entry point hands back zeros for every record that is not a usable point
        for (size_t i = 0; i < n; i++)
          if (status[i] == CURDLE_DECODE_NOT_IN_SUBGROUP) memset(out_affine + 12 * i, 0, 96);
      return rc2;
    }
    if (rc2 != CURDLE_EBUSY) return rc2;
  }
  int idx;
  int rc = acquire_slot(cx, true, &idx);
  if (rc) return rc;
  Slot& S = cx.slots[idx];
  auto body = [&]() -> int {
    HIP_TRY(hipSetDevice(cx.device));
    int r;
    // the slot's generic buffers: compressed input, decoded output, status bytes
    if ((r = ensure(S.scalars, n * 48))) return r;
    if ((r = ensure(S.points, n * 96))) return r;
    if ((r = ensure(S.counts, n))) return r;
    // Through the slot's pinned staging, not straight from / to the caller's pageable memory:
    // a pageable hipMemcpyAsync is synchronous, and the first one a host thread issues was
    // measured at 15-20 ms for 1.8 MB (the batch verifier's decoding producers are new threads
    // on every call) against 0.1 ms afterwards.
    if ((r = ensure_pinned(S, 0, n * 48))) return r;
    if ((r = ensure_pinned(S, 1, n * 97))) return r;
    memcpy(S.h_stage[0], in, n * 48);
    uint8_t* h_out = static_cast<uint8_t*>(S.h_stage[1]);
    HIP_TRY(hipMemcpyAsync(S.scalars.p, S.h_stage[0], n * 48, hipMemcpyHostToDevice, S.stream));
    HIP_TRY(launch_g1_decompress((const uint8_t*)S.scalars.p, (uint32_t)n, subgroup_check, (uint32_t*)S.points.p,
                                 (uint8_t*)S.counts.p, S.stream));
    HIP_TRY(hipMemcpyAsync(h_out, S.points.p, n * 96, hipMemcpyDeviceToHost, S.stream));
    HIP_TRY(hipMemcpyAsync(h_out + n * 96, S.counts.p, n, hipMemcpyDeviceToHost, S.stream));
    HIP_TRY(hipStreamSynchronize(S.stream));
    memcpy(out_affine, h_out, n * 96);
    memcpy(status, h_out + n * 96, n);
    return CURDLE_OK;
  };
  rc = body();
  if (rc) (void)hipStreamSynchronize(S.stream);  // nothing queued may outlive the slot's hold
  release_slot(cx, idx);
  return rc;
}

// Two-step form: begin decodes (square root, curve check, sign) and returns the points, and
// leaves the subgroup test running on the slot's stream; finish waits for it and returns the
// final status bytes.  The caller can work with the points in between.
extern "C" int curdle_g1_decompress_finish(int ticket, uint8_t* status);

// Three-step form: start launches the decoding (square root, curve check, sign) and returns at
// once; points waits for that half and hands the points back, leaving the subgroup test running
// on the context's stream; finish waits for it and returns the final status bytes.  The caller
// can hash its transcript between start and points, and verify between points and finish.
// begin = start + points.
namespace {
// A decode context's streams, made on its first use (see init_locked: unused streams must not
// take hardware queues), at the same (lowest) priority as every other stream of the library.
// While the subgroup test still ran BEHIND the square roots, for 0.65 ms beside the caller's MSM,
// the highest priority kept that MSM from queueing behind it (0.46 against 0.88 ms,
// profiles/r02_verify_from_bytes_queues.txt); since the two chains overlap the test is over when
// the MSM starts, priorities no longer change one verification's latency (1.14-1.17 ms at ell =
// 252 in all six combinations of 4 / 16 hardware queues and the three priorities), and with
// eight threads verifying at once the lowest one measured best (the knob that selected the others is gone).
int ensure_dslot_streams(Ctx& cx) {
  // the flag is set (release) after every stream and event of every decode context was stored, and
  // read (acquire) before any of them is used: no thread sees a half-made context
  if (cx.dstreams_ready.load(std::memory_order_acquire)) return CURDLE_OK;
  // every decode context at once, under the lock: creating a stream (its hardware queue) takes
  // tens of milliseconds, and a second context first used under load would put that into some
  // caller's verification (seen as 354 instead of 1,300 Whisk verifications/s from four threads)
  std::lock_guard<std::mutex> g(cx.mu);
  if (cx.dstreams_ready.load(std::memory_order_relaxed)) return CURDLE_OK;
  const int dprio = cx.prio_least;
  for (DSlot& x : cx.dslots) {
    if (x.stream) continue;
    hipStream_t a = nullptr, b = nullptr, c = nullptr;
    hipEvent_t e1 = nullptr, e2 = nullptr;
    HIP_TRY(hipStreamCreateWithPriority(&a, hipStreamNonBlocking, dprio));
    HIP_TRY(hipStreamCreateWithPriority(&b, hipStreamNonBlocking, dprio));
    HIP_TRY(hipStreamCreateWithFlags(&c, hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags(&e1, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&e2, hipEventDisableTiming));
    x.sub_stream = b;
    x.copy_stream = c;
    x.uploaded = e1;
    x.decoded = e2;
    x.stream = a;
  }
  cx.dstreams_ready.store(true, std::memory_order_release);
  return CURDLE_OK;
}
}  // namespace

extern "C" int curdle_g1_decompress_start(const uint8_t* in, size_t n, int* ticket) {
  Ctx& cx = cur();
  if (!ticket || (n && !in)) return fail(CURDLE_EINVAL, "null argument");
  if (n > ((size_t)1 << 27)) return fail(CURDLE_EINVAL, "n = %zu exceeds the supported 2^27 points", n);
  int idx = -1;
  {
    std::unique_lock<std::mutex> g(cx.mu);
    int rc = init_default_locked(cx);
    if (rc) return rc;
    // Deferring the subgroup test only pays for a caller that would otherwise wait for it;
    // with several verifications in flight the GPU is busy anyway and every extra stream
    // costs hardware-queue sharing: beyond kMaxDeferred the caller is told to use the
    // one-shot form.
    int busy = 0;
    for (int i = 0; i < kMaxDeferred; i++) {
      if (cx.dslots[i].busy)
        busy++;
      else if (idx < 0)
        idx = i;
    }
    if (idx < 0 || busy >= kMaxDeferred)
      return fail(CURDLE_EBUSY, "%d deferred point decodings in flight; use curdle_g1_decompress_batch", busy);
    cx.dslots[idx].busy = true;
    cx.dslots[idx].claimed = false;
    cx.dslots[idx].gen++;
  }
  DSlot& D = cx.dslots[idx];
  auto body = [&]() -> int {
    HIP_TRY(hipSetDevice(cx.device));
    D.n = (uint32_t)n;
    int r;
    if ((r = ensure_dslot_streams(cx))) return r;
    if (n == 0) return CURDLE_OK;
    if ((r = ensure(D.in, n * 48))) return r;
    if ((r = ensure(D.out, n * 96))) return r;
    if ((r = ensure(D.status, n))) return r;
    if ((r = ensure(D.sub, n))) return r;
    // pinned staging: the copy must not block the caller, who wants to hash meanwhile
    if (D.h_in_cap < n * 48) {
      if (D.h_in) HIP_TRY(hipHostFree(D.h_in));
      D.h_in = nullptr;
      D.h_in_cap = 0;
      HIP_TRY(hipHostMalloc(&D.h_in, grow_size(n * 48), hipHostMallocDefault));
      D.h_in_cap = grow_size(n * 48);
    }
    if (D.h_out_cap < n * 98) {
      if (D.h_out) HIP_TRY(hipHostFree(D.h_out));
      D.h_out = nullptr;
      D.h_out_cap = 0;
      HIP_TRY(hipHostMalloc(&D.h_out, grow_size(n * 98), hipHostMallocDefault));
      D.h_out_cap = grow_size(n * 98);
    }
    memcpy(D.h_in, in, n * 48);
    HIP_TRY(hipMemcpyAsync(D.in.p, D.h_in, n * 48, hipMemcpyHostToDevice, D.stream));
    HIP_TRY(hipEventRecord(D.uploaded, D.stream));
    HIP_TRY(launch_g1_decompress((const uint8_t*)D.in.p, (uint32_t)n, 0, (uint32_t*)D.out.p, (uint8_t*)D.status.p,
                                 D.stream));
    HIP_TRY(hipEventRecord(D.decoded, D.stream));
    // The subgroup test does not wait for the square roots: it works on a twisted model of the
    // curve that needs only x (decode_kernels.hip), on its own stream, beside the decoding
    // kernel -- the two ~0.5 ms chains per point overlap instead of adding up.
    HIP_TRY(hipStreamWaitEvent(D.sub_stream, D.uploaded, 0));
    HIP_TRY(launch_g1_subgroup_from_bytes((const uint8_t*)D.in.p, (uint32_t)n, (uint8_t*)D.sub.p, D.sub_stream));
    return CURDLE_OK;
  };
  int rc = body();
  if (rc) {
    if (D.stream) (void)hipStreamSynchronize(D.stream);
    if (D.sub_stream) (void)hipStreamSynchronize(D.sub_stream);
    {
      std::lock_guard<std::mutex> g(cx.mu);
      D.busy = false;
    }
    return rc;
  }
  *ticket = make_ticket(cx, idx, D.gen);
  return CURDLE_OK;
}

namespace {
// the decode context behind a ticket that is in flight and not being finished; nullptr otherwise
DSlot* dslot_of(int ticket) {
  Ctx* cp = ticket_ctx(ticket);
  if (!cp || ticket_index(ticket) >= kMaxDeferred) return nullptr;
  Ctx& cx = *cp;
  DSlot& D = cx.dslots[ticket_index(ticket)];
  std::lock_guard<std::mutex> g(cx.mu);
  if (!D.busy || D.claimed || (D.gen & 0x7fffffu) != ticket_gen(ticket)) return nullptr;
  return &D;
}
}  // namespace

extern "C" int curdle_g1_decompress_points(int ticket, uint64_t* out_affine, uint8_t* status) {
  DSlot* Dp = dslot_of(ticket);
  if (!Dp) return fail(CURDLE_EINVAL, "ticket %d is not in flight (stale or already finished)", ticket);
  Ctx& cx = *ticket_ctx(ticket);
  DSlot& D = *Dp;
  const size_t n = D.n;
  if (n && (!out_affine || !status)) return fail(CURDLE_EINVAL, "null argument");
  auto body = [&]() -> int {
    HIP_TRY(hipSetDevice(cx.device));
    if (n == 0) return CURDLE_OK;
    // encoding / curve verdicts only: the subgroup test's arrive with curdle_g1_decompress_finish
    // through pinned staging: a copy into the caller's pageable memory goes through the runtime's
    // shared bounce buffers, which concurrent verifications then queue for
    uint8_t* h = static_cast<uint8_t*>(D.h_out);
    HIP_TRY(hipStreamWaitEvent(D.copy_stream, D.decoded, 0));
    HIP_TRY(hipMemcpyAsync(h, D.out.p, n * 96, hipMemcpyDeviceToHost, D.copy_stream));
    HIP_TRY(hipMemcpyAsync(h + n * 96, D.status.p, n, hipMemcpyDeviceToHost, D.copy_stream));
    HIP_TRY(hipStreamSynchronize(D.copy_stream));
    memcpy(out_affine, h, n * 96);
    memcpy(status, h + n * 96, n);
    return CURDLE_OK;
  };
  int rc = body();
  if (rc) (void)hipStreamSynchronize(D.copy_stream);  // the ticket stays valid: the caller still has to finish it
  return rc;
}

extern "C" int curdle_g1_decompress_begin(const uint8_t* in, size_t n, uint64_t* out_affine, uint8_t* status,
                                          int* ticket) {
  if (!ticket || (n && (!in || !out_affine || !status))) return fail(CURDLE_EINVAL, "null argument");
  int rc = curdle_g1_decompress_start(in, n, ticket);
  if (rc) return rc;
  rc = curdle_g1_decompress_points(*ticket, out_affine, status);
  if (rc) {
    char saved[256];
    snprintf(saved, sizeof(saved), "%s", g_err);
    (void)curdle_g1_decompress_finish(*ticket, nullptr);
    *ticket = -1;
    return fail(rc, "%s", saved);
  }
  return CURDLE_OK;
}

extern "C" int curdle_g1_decompress_finish(int ticket, uint8_t* status) {
  Ctx* cp = ticket_ctx(ticket);
  if (!cp || ticket_index(ticket) >= kMaxDeferred) return fail(CURDLE_EINVAL, "bad ticket");
  Ctx& cx = *cp;
  DSlot& D = cx.dslots[ticket_index(ticket)];
  {
    std::lock_guard<std::mutex> g(cx.mu);
    if (!D.busy || D.claimed || (D.gen & 0x7fffffu) != ticket_gen(ticket))
      return fail(CURDLE_EINVAL, "ticket %d is not in flight (stale or already finished)", ticket);
    D.claimed = true;
  }
  const size_t n = D.n;
  int rc = CURDLE_OK;
  hipError_t he = hipSetDevice(cx.device);
  uint8_t* h = static_cast<uint8_t*>(D.h_out);  // [0, n): statuses, [n, 2n): subgroup verdicts (the points' block is free again)
  if (he == hipSuccess) he = hipStreamSynchronize(D.copy_stream);
  if (he == hipSuccess && n && status) {
    he = hipMemcpyAsync(h, D.status.p, n, hipMemcpyDeviceToHost, D.stream);
    if (he == hipSuccess) he = hipMemcpyAsync(h + n, D.sub.p, n, hipMemcpyDeviceToHost, D.sub_stream);
  }
  if (he == hipSuccess) he = hipStreamSynchronize(D.stream);
  if (he == hipSuccess) he = hipStreamSynchronize(D.sub_stream);
  if (he != hipSuccess) rc = fail(CURDLE_EHIP, "decompress finish: %s", hipGetErrorString(he));
  if (rc == CURDLE_OK && n && status)  // a decoded point outside the subgroup: the one verdict the points came without
    for (size_t i = 0; i < n; i++)
      status[i] = (h[i] == CURDLE_DECODE_OK && !h[n + i]) ? (uint8_t)CURDLE_DECODE_NOT_IN_SUBGROUP : h[i];
  {
    std::lock_guard<std::mutex> g(cx.mu);
    D.busy = false;
  }
  return rc;
}

// ---------------------------------------------------------------------------
// Batched independent scalar multiplications (group_kernels.hip)
// ---------------------------------------------------------------------------
extern "C" int curdle_g1_scalar_mul_batch(const uint64_t* points, const uint64_t* scalars, size_t n_scalars,
                                          const uint64_t* addends, size_t n, uint64_t* out_affine) {
  Ctx& cx = cur();
  if (n && (!points || !scalars || !out_affine)) return fail(CURDLE_EINVAL, "null argument");
  if (n == 0) return CURDLE_OK;
  if (n_scalars != n && n_scalars != 1) return fail(CURDLE_EINVAL, "n_scalars must be n or 1");
  if (n > ((size_t)1 << 24)) return fail(CURDLE_EINVAL, "n = %zu exceeds the supported 2^24 points", n);
  int idx;
  int rc = acquire_slot(cx, true, &idx);
  if (rc) return rc;
  Slot& S = cx.slots[idx];
  auto body = [&]() -> int {
    HIP_TRY(hipSetDevice(cx.device));
    int r;
    if ((r = ensure(S.points, n * 96))) return r;
    if ((r = ensure(S.scalars, n_scalars * 32))) return r;
    if (addends && (r = ensure(S.digits, n * 96))) return r;
    if ((r = ensure(S.sorted, n * sizeof(G1XYZZ)))) return r;
    HIP_TRY(hipMemcpyAsync(S.points.p, points, n * 96, hipMemcpyHostToDevice, S.stream));
    HIP_TRY(hipMemcpyAsync(S.scalars.p, scalars, n_scalars * 32, hipMemcpyHostToDevice, S.stream));
    if (addends) HIP_TRY(hipMemcpyAsync(S.digits.p, addends, n * 96, hipMemcpyHostToDevice, S.stream));
    HIP_TRY(launch_scalar_mul_batch(S.points.p, S.scalars.p, n_scalars == 1 ? 1 : 0, addends ? S.digits.p : nullptr,
                                    (uint32_t)n, S.sorted.p, S.stream));
    std::vector<G1XYZZ> res(n);
    HIP_TRY(hipMemcpyAsync(res.data(), S.sorted.p, n * sizeof(G1XYZZ), hipMemcpyDeviceToHost, S.stream));
    HIP_TRY(hipStreamSynchronize(S.stream));
    curdle_host_batch_to_affine(out_affine, res.data(), n);  // one shared inversion
    return CURDLE_OK;
  };
  rc = body();
  if (rc) (void)hipStreamSynchronize(S.stream);  // `res` is a local: nothing may still be copying into it
  release_slot(cx, idx);
  return rc;
}

// ---------------------------------------------------------------------------
// Life cycle
// ---------------------------------------------------------------------------
namespace {
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
}  // namespace

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

// ---------------------------------------------------------------------------
// MSM
// ---------------------------------------------------------------------------
extern "C" int curdle_msm_window_bits(size_t n) { return choose_window_bits(n); }

// The window width a call will run with: the caller's, or the library's choice for n pairs; anything outside
// [4, 16] is refused HERE, before window_widths() writes kMaxWindows bytes for it (review of round 5: a width of
// 1..3 has more than 64 windows, a large negative one none at all).
static int checked_window_bits(size_t n, int window_bits, int* c) {
  *c = window_bits ? window_bits : choose_window_bits(n);
  if (*c < 4 || *c > 16) return fail(CURDLE_EINVAL, "window_bits %d outside [4, 16]", *c);
  return CURDLE_OK;
}

extern "C" int curdle_msm_window_widths_ex(size_t n, int window_bits, unsigned flags, int widths[64]) {
  if (flags & ~(unsigned)(CURDLE_MSM_ANY_CURVE_POINT | CURDLE_MSM_BASES_UNCHANGED)) return fail(CURDLE_EINVAL, "unknown flags 0x%x", flags);
  int c;
  if (checked_window_bits(n, window_bits, &c)) return CURDLE_EINVAL;
  uint8_t bits[kMaxWindows];
  int W = window_widths(c, bits, (flags & CURDLE_MSM_ANY_CURVE_POINT) ? kScalarBitsNoGlv : kScalarBits);
  if (widths)
    for (int w = 0; w < W; w++) widths[w] = bits[w];
  return W;
}

extern "C" int curdle_msm_window_widths(size_t n, int window_bits, int widths[64]) {
  return curdle_msm_window_widths_ex(n, window_bits, 0, widths);
}

extern "C" int curdle_msm_num_windows(size_t n, int window_bits) { return curdle_msm_window_widths_ex(n, window_bits, 0, nullptr); }
extern "C" int curdle_msm_num_windows_ex(size_t n, int window_bits, unsigned flags) {
  return curdle_msm_window_widths_ex(n, window_bits, flags, nullptr);
}

namespace {
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
}  // namespace

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

namespace {
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
}  // namespace

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

void dbases_release_handle(struct curdle_dbases* b);  // defined with the resident base sets below

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

namespace {
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
}  // namespace

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

void dbases_release_handle(curdle_dbases* b) { dbases_release(b); }

// ---------------------------------------------------------------------------
// The plain MSM over a resident, pre-converted base set: msmaccumulator.Verify's bases are mostly
// the CRS (/root/reference/crs.go:10-18; msmaccumulator.go:59), which never changes -- and
// k_convert_points is 0.10 ms and 360 MB of every 2^20 call (0.16 ms of a pipelined step), which every
// rank of a window split repeats for ALL points.  The set's internal-form records are read in place.
// ---------------------------------------------------------------------------
namespace {
int dbases_msm_args(const curdle_dbases* bases, const void* scalars, size_t n, uint64_t* out_jac) {
  if (!bases || !out_jac) return fail(CURDLE_EINVAL, "null argument");
  if (n > bases->n) return fail(CURDLE_EINVAL, "n = %zu exceeds the %zu resident bases", n, bases->n);
  if (n && !scalars) return fail(CURDLE_EINVAL, "scalars null with n = %zu", n);
  return CURDLE_OK;
}
}  // namespace

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

namespace {
// the end of an accumulation, however it ends: the slot and the base set go back
void dacc_end(curdle_dacc* acc) {
  release_slot(*acc->ctx, acc->slot);
  dbases_release(acc->crs);
  delete acc;
}
}  // namespace

extern "C" void curdle_dacc_abort(curdle_dacc* acc) {
  if (!acc) return;
  Ctx& cx = *acc->ctx;
  (void)hipSetDevice(cx.device);
  (void)hipStreamSynchronize(cx.slots[acc->slot].stream);
  dacc_end(acc);
}

namespace {
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
namespace {
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
}  // namespace

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

// ---------------------------------------------------------------------------
// Synthetic inputs
// ---------------------------------------------------------------------------
extern "C" int curdle_synth_points_walk_device(const uint64_t k[4], const uint64_t q[4], size_t n, void* d_out) {
  Ctx& cx = cur();
  if (!k || !q || (n && !d_out)) return fail(CURDLE_EINVAL, "null argument");
  if (n > ((size_t)1 << 27)) return fail(CURDLE_EINVAL, "n = %zu exceeds the supported 2^27 points", n);
  if (n == 0) return CURDLE_OK;
  std::lock_guard<std::mutex> g(cx.mu);
  int rc = init_default_locked(cx);
  if (rc) return rc;
  HIP_TRY(hipSetDevice(cx.device));
  G1Affine gen;
  g1_generator(gen);
  G1XYZZ gx, t;
  g1_from_affine(gx, gen);
  G1Affine p0, table[27];
  g1_scalar_mul(t, gx, reinterpret_cast<const u32*>(k), 8);
  g1_to_affine(p0, t);
  g1_scalar_mul(t, gx, reinterpret_cast<const u32*>(q), 8);
  for (int j = 0; j < 27; j++) {
    g1_to_affine(table[j], t);
    g1_dbl(t);
  }
  void* d_table = nullptr;
  HIP_TRY(hipMalloc(&d_table, sizeof(table)));
  HIP_TRY(hipMemcpyAsync(d_table, table, sizeof(table), hipMemcpyHostToDevice, cx.util_stream));
  HIP_TRY(launch_synth_walk((const G1Affine*)d_table, p0, (uint32_t)n, d_out, cx.util_stream));
  HIP_TRY(hipStreamSynchronize(cx.util_stream));
  HIP_TRY(hipFree(d_table));
  return CURDLE_OK;
}

// ---------------------------------------------------------------------------
// Profiling / self-test
// ---------------------------------------------------------------------------
extern "C" int curdle_profile_enable(int on) {
  Ctx& cx = cur();
  std::lock_guard<std::mutex> g(cx.mu);
  cx.profile = on < 0 || on > 2 ? 1 : on;
  return CURDLE_OK;
}

extern "C" int curdle_profile_last(curdle_profile* out) {
  Ctx& cx = cur();
  if (!out) return fail(CURDLE_EINVAL, "null argument");
  std::lock_guard<std::mutex> g(cx.mu);
  *out = cx.last;
  return CURDLE_OK;
}

extern "C" int curdle_selftest_op(int op, const uint64_t* in64, size_t n, uint64_t* out64, int on_device) {
  Ctx& cx = cur();
  if (op < 0 || op >= kSelftestOps || !in64 || !out64) return fail(CURDLE_EINVAL, "bad selftest arguments");
  const uint32_t* in = reinterpret_cast<const uint32_t*>(in64);
  uint32_t* out = reinterpret_cast<uint32_t*>(out64);
  // the widths of every operation live in ONE table (msm_kernels.h), which the launcher and the kernel index too
  const size_t in_w = kSelftestTable[op].in_words, out_w = kSelftestTable[op].out_words;
  if (n > ((size_t)1 << 26)) return fail(CURDLE_EINVAL, "selftest of %zu elements", n);
  if (!on_device) {
    for (size_t i = 0; i < n; i++) {
      const uint32_t* s = in + i * in_w;
      uint32_t* d = out + i * out_w;
      if (op == 11) {  // the GLV split of a canonical scalar: |k1| (4 words), k2 (4 words), the two signs
        Fr k;
        memcpy(&k, s, 32);
        u32 sa = 0, sb = 0;
        glv_split(k, d, d + 4, sa, sb);
        d[8] = sa;
        d[9] = sb;
      } else if (op == 12) {  // into the MSM's curve and back: the identity, both bounds kept
        memcpy(d, s, 96);
        d[24] = d[25] = 1;
      } else if (op <= 3) {
        Fp a, b, r;
        memcpy(&a, s, 48);
        memcpy(&b, s + 12, 48);
        if (op == 0) fp_mul(r, a, b);
        else if (op == 1) fp_add(r, a, b);
        else if (op == 2) fp_sub(r, a, b);
        else fp_sqr(r, a);
        memcpy(d, &r, 48);
      } else if (op == 4) {
        Fr a, r;
        memcpy(&a, s, 32);
        f_from_mont<FrParams>(r, a);
        memcpy(d, &r, 32);
      } else {
        G1XYZZ acc, b;
        memcpy(&acc, s, 192);
        memcpy(&b, s + 48, 192);
        if (op == 5) {
          if (!(f_is_zero(b.x) && f_is_zero(b.y))) g1_madd(acc, b.x, b.y);
        } else if (op == 6 || op == 8) {
          g1_add(acc, b);
        } else if (op == 7 || op == 9) {
          g1_dbl(acc);
        } else {  // op 10: k * b with the kernel's 20-bit k (msm_kernels.hip k_selftest)
          const u32 k = (u32)((u32)i * 2654435761u) >> 12;
          g1_scalar_mul(acc, b, &k, 1);
        }
        memcpy(d, &acc, 192);
      }
    }
    return CURDLE_OK;
  }
  std::lock_guard<std::mutex> g(cx.mu);
  int rc = init_default_locked(cx);
  if (rc) return rc;
  HIP_TRY(hipSetDevice(cx.device));
  if (n == 0) return CURDLE_OK;
  struct DevBuf {  // freed on every way out, after the stream has drained
    void* p = nullptr;
    hipStream_t st;
    explicit DevBuf(hipStream_t s) : st(s) {}
    ~DevBuf() {
      if (p) {
        (void)hipStreamSynchronize(st);
        (void)hipFree(p);
      }
    }
  } d_in(cx.util_stream), d_out(cx.util_stream);
  HIP_TRY(hipMalloc(&d_in.p, n * in_w * 4));
  HIP_TRY(hipMalloc(&d_out.p, n * out_w * 4));
  HIP_TRY(hipMemcpyAsync(d_in.p, in, n * in_w * 4, hipMemcpyHostToDevice, cx.util_stream));
  HIP_TRY(launch_selftest(op, (const uint32_t*)d_in.p, n, (uint32_t*)d_out.p, cx.util_stream));
  HIP_TRY(hipMemcpyAsync(out, d_out.p, n * out_w * 4, hipMemcpyDeviceToHost, cx.util_stream));
  HIP_TRY(hipStreamSynchronize(cx.util_stream));
  return CURDLE_OK;
}

#ifdef CURDLE_TRACE_WAVES
namespace curdle { hipError_t debug_read_wave_trace(unsigned long long* out, size_t words); hipError_t debug_read_wave_clk(unsigned long long* out, size_t words); }
extern "C" int curdle_debug_wave_clk(uint64_t* out, size_t words) {
  return curdle::debug_read_wave_clk((unsigned long long*)out, words) == hipSuccess ? 0 : -1;
}
extern "C" int curdle_debug_wave_trace(uint64_t* out, size_t words) {
  return curdle::debug_read_wave_trace((unsigned long long*)out, words) == hipSuccess ? 0 : -1;
}
#endif
extern "C" int curdle_selftest_shape(int op, uint32_t* in_words, uint32_t* out_words) {
  if (op < 0 || op >= kSelftestOps || !in_words || !out_words) return fail(CURDLE_EINVAL, "selftest op %d outside [0, %d)", op, kSelftestOps);
  *in_words = kSelftestTable[op].in_words;
  *out_words = kSelftestTable[op].out_words;
  return CURDLE_OK;
}

// ---------------------------------------------------------------------------
// common.Rand and msmaccumulator handles
// ---------------------------------------------------------------------------
struct curdle_rand {
  common::Rand r;
  explicit curdle_rand(uint64_t seed) : r(seed) {}
};
struct curdle_acc {
  msmaccumulator::MsmAccumulator a;
};

extern "C" curdle_rand* curdle_rand_new(uint64_t seed) { return new (std::nothrow) curdle_rand(seed); }
extern "C" void curdle_rand_free(curdle_rand* r) { delete r; }

extern "C" int curdle_rand_get_fr(curdle_rand* r, uint64_t out_fr[4]) {
  if (!r || !out_fr) return fail(CURDLE_EINVAL, "null argument");
  Fr f;
  r->r.GetFr(f);
  memcpy(out_fr, &f, 32);
  return CURDLE_OK;
}

extern "C" int curdle_rand_get_g1_affine(curdle_rand* r, uint64_t out_aff[12]) {
  if (!r || !out_aff) return fail(CURDLE_EINVAL, "null argument");
  G1Affine p;
  r->r.GetG1Affine(p);
  memcpy(out_aff, &p, 96);
  return CURDLE_OK;
}

extern "C" int curdle_rand_permutation(curdle_rand* r, size_t n, uint32_t* out) {
  if (!r || (n && !out)) return fail(CURDLE_EINVAL, "null argument");
  std::vector<uint32_t> perm;
  r->r.GeneratePermutation(n, perm);
  if (n) memcpy(out, perm.data(), n * 4);
  return CURDLE_OK;
}

extern "C" curdle_acc* curdle_acc_new(void) { return new (std::nothrow) curdle_acc(); }
extern "C" void curdle_acc_free(curdle_acc* a) { delete a; }

extern "C" int curdle_acc_accumulate_check(curdle_acc* a, const uint64_t C_jac[18], const uint64_t* x, size_t x_len,
                                           const uint64_t* v, size_t v_len, curdle_rand* rand) {
  if (!a || !C_jac || !rand || (x_len && !x) || (v_len && !v)) return fail(CURDLE_EINVAL, "null argument");
  G1Jac C;
  memcpy(&C, C_jac, sizeof(C));
  std::vector<Fr> xs(x_len);
  std::vector<G1Affine> vs(v_len);
  if (x_len) memcpy(xs.data(), x, x_len * 32);
  if (v_len) memcpy(vs.data(), v, v_len * 96);
  msmaccumulator::Status st = a->a.AccumulateCheck(C, xs, vs, &rand->r);
  if (!st.ok) return fail(CURDLE_EINVAL, "%s", st.err.c_str());
  return CURDLE_OK;
}

extern "C" int curdle_acc_accumulate_check_deferred(curdle_acc* a, const uint64_t* c_scalars, const uint64_t* c_points,
                                                    size_t c_len, const uint64_t* x, size_t x_len, const uint64_t* v,
                                                    size_t v_len, curdle_rand* rand) {
  if (!a || !rand || (c_len && (!c_scalars || !c_points)) || (x_len && !x) || (v_len && !v))
    return fail(CURDLE_EINVAL, "null argument");
  std::vector<Fr> cs(c_len), xs(x_len);
  std::vector<G1Affine> cp(c_len), vs(v_len);
  if (c_len) memcpy(cs.data(), c_scalars, c_len * 32);
  if (c_len) memcpy(cp.data(), c_points, c_len * 96);
  if (x_len) memcpy(xs.data(), x, x_len * 32);
  if (v_len) memcpy(vs.data(), v, v_len * 96);
  msmaccumulator::Status st = a->a.AccumulateCheckDeferred(cs, cp, xs, vs, &rand->r);
  if (!st.ok) return fail(CURDLE_EINVAL, "%s", st.err.c_str());
  return CURDLE_OK;
}

extern "C" int curdle_acc_verify(curdle_acc* a, int* ok) {
  if (!a || !ok) return fail(CURDLE_EINVAL, "null argument");
  bool b = false;
  char saved[256];
  msmaccumulator::Status st = a->a.Verify(&b);
  *ok = b ? 1 : 0;
  if (!st.ok) {
    snprintf(saved, sizeof(saved), "%s", st.err.c_str());
    // keep the class of the underlying failure (no device vs HIP error) visible to the caller
    return fail(st.rc ? st.rc : CURDLE_EHIP, "%s", saved);
  }
  return CURDLE_OK;
}

extern "C" int curdle_acc_get_A_c(const curdle_acc* a, uint64_t out_jac[18]) {
  if (!a || !out_jac) return fail(CURDLE_EINVAL, "null argument");
  g1_to_canonical_jac(out_jac, a->a.A_c);
  return CURDLE_OK;
}

extern "C" size_t curdle_acc_num_bases(const curdle_acc* a) { return a ? a->a.NumBases() : 0; }

extern "C" int curdle_acc_export(const curdle_acc* a, uint64_t* points, uint64_t* scalars) {
  if (!a || !points || !scalars) return fail(CURDLE_EINVAL, "null argument");
  size_t n = a->a.NumBases();
  if (n) {
    memcpy(points, a->a.Bases().data(), n * 96);
    memcpy(scalars, a->a.Scalars().data(), n * 32);
  }
  return CURDLE_OK;
}
