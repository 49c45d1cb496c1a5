// Internals shared by the translation units of the C ABI (msm_context.hip, msm_plan.hip, msm_enqueue.hip,
// msm_host_chunks.hip, msm_entry.hip, dbases_api.hip, decode_api.hip, misc_api.hip -- ONE file, msm_api.hip, of 3,200 lines
// until round 6): the context and its workspace slots, the error text, tickets, the phase profiler, and the prototypes
// of what one unit calls in another.  Everything here is in namespace curdle_api; nothing is exported.
#pragma once
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


namespace curdle_api {
// ---------------------------------------------------------------------------
// Errors
// ---------------------------------------------------------------------------
extern thread_local char g_err[256];
int fail(int code, const char* fmt, ...);
}  // namespace curdle_api

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
namespace curdle_api {

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
extern Ctx g_ctxs[kMaxDevices];
extern std::atomic<int> g_ndev;  // configured contexts: [0, g_ndev)
extern std::mutex g_cfg_mu;      // configuration (curdle_init_devices / curdle_shutdown)
extern std::atomic<int> g_multi_calls;  // calls that span the devices' host threads right now
extern std::atomic<unsigned long long> g_spread_calls;  // host-buffer MSMs that were spread over several devices, ever
extern thread_local int tl_dev;
extern thread_local bool tl_selected;  // the thread called curdle_set_device: its host-buffer MSMs stay on that device
// the calling thread's context; a thread whose selection no longer exists (curdle_shutdown since) is on 0
inline Ctx& cur() { return g_ctxs[tl_dev < g_ndev.load(std::memory_order_acquire) ? tl_dev : 0]; }

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


// --- msm_context.hip ---------------------------------------------------------
size_t grow_size(size_t bytes);
int ensure(Buf& b, size_t bytes);
int ensure_pinned(Slot& S, int which, size_t bytes);
int init_locked(Ctx& cx, int device);
int init_default_locked(Ctx& cx);
void set_out_infinity(uint64_t out[18]);
int acquire_slot(Ctx& cx, bool block, int* idx);
void release_slot(Ctx& cx, int idx);

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


// --- msm_plan.hip ------------------------------------------------------------
constexpr int kScalarBits = 127;
constexpr int kScalarBitsNoGlv = 255;
int choose_window_bits(size_t n, bool many = false);
int window_widths(int c, uint8_t bits[kMaxWindows], int scalar_bits = kScalarBits);
int make_plan(MsmPlan& p, size_t n_total, size_t k, size_t n_max, int c, int win_begin, int win_end,
              bool latency_mode, size_t sets = 1, bool many = false, uint32_t seg_override = 0, bool light_host = false,
              bool glv = true);
int checked_window_bits(size_t n, int window_bits, int* c);

// --- msm_enqueue.hip ---------------------------------------------------------
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
  bool chunked = false;               // one of SEVERAL chunks of a host-buffer MSM (set for every chunk, in both of its enqueue steps): the plan rules that differ
                                      // for chunks -- bucket-merge limit, merge grid -- must be the same in the sort step and in the accumulate step
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



int enqueue_slot(Ctx& cx, Slot& S, const void* d_points, const void* d_scalars, const uint32_t* h_off, size_t k, int c,
                 int win_begin, int win_end, hipStream_t pre, hipStream_t stream, hipStream_t tail,
                 bool latency_mode = true, bool points28_ready = false, size_t sets = 1, bool many = false,
                 const ChunkJoin* join = nullptr, const void* ext_points28 = nullptr, bool light_host = false,
                 bool glv = true, const DaccFront* dfront = nullptr);
int finish_slot(Ctx& cx, Slot& S, uint64_t* out);
void drain_slot(Ctx& cx, Slot& S);
struct SyncStreams {
  hipStream_t pre, main, tail;
};
SyncStreams sync_streams(Ctx&, Slot& S);
constexpr size_t kMaxSlotsPerPass = (size_t)1024 * 4096;
int run_passes(Ctx& cx, Slot& S, const void* d_points, const void* d_scalars, const uint32_t* h_off, size_t k, int c,
               int win_begin, int win_end, hipStream_t pre, hipStream_t main, hipStream_t tail, uint64_t* out,
               const void* ext_points28 = nullptr, bool glv = true);
int run_device(const void* d_points, const void* d_scalars, const uint32_t* h_off, size_t k, int c, int win_begin,
               int win_end, uint64_t* out, void* user_stream, const void* ext_points28 = nullptr, bool glv = true,
               hipEvent_t wait_for = nullptr);
int run_host(const uint64_t* points, const uint64_t* scalars, const uint32_t* h_off, size_t k, uint64_t* out, bool glv = true);

// --- msm_host_chunks.hip -----------------------------------------------------
constexpr size_t kHostChunkMin = (size_t)1 << 19;  // below this a call is one chunk
int run_host_chunked(const uint64_t* points, const uint64_t* scalars, size_t n, uint64_t* out, bool glv = true);

// --- dbases_api.hip ----------------------------------------------------------
void dbases_release_handle(struct ::curdle_dbases* b);
}  // namespace curdle_api

using namespace curdle_api;
