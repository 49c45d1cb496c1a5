// Host-callable launchers of the gfx950 MSM kernels (msm_sort_kernels.hip, msm_accumulate_kernel.hip, msm_reduce_kernels.hip, msm_misc_kernels.hip).
// Internal to libcurdlemsm.so; the public surface is include/curdle_msm.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "bls12_381.h"

namespace curdle {

static constexpr int kMaxWindows = 64;

// Pippenger decomposition shared by the k MSMs of one call (k = 1 for a single
// MSM).  The 255 scalar bits are cut into W windows of bits[w] bits each, as
// even as possible.  Windows below the top one are recoded into signed digits
// d in [-2^(b-1), 2^(b-1)] (2^(b-1) buckets: negating an affine point is free);
// the top window keeps its unsigned digit (2^b buckets), so no carry can leave
// it and no window is left with a handful of heavily loaded buckets.  Bucket
// slot `base[w] + m - 1` of an MSM holds the points whose |digit| in window w
// is m; the MSMs' slot ranges follow each other (NB slots each).
struct MsmPlan {
  uint32_t n;          // pairs of all MSMs of the call
  uint32_t k;          // scalar vectors (sorted MSMs) in the call
  uint32_t sets;       // base sets sharing every scalar vector's recoding and sort (1 but for curdle_msm_g1_multi)
  uint32_t kr;         // results = k * sets; result r = set * k + j
  uint32_t frag_stride;  // fragments reserved per base set
  uint32_t reduce_prio;  // s_setprio of the reduction kernels' waves (knob REDUCE_PRIO; 3 for pipelined calls)
  uint32_t aux_prio;     // s_setprio of the sort kernels' and the fold's waves (knob AUX_PRIO; experiment, 0 by default)
  uint32_t acc_prio;     // k_accumulate: log2 of the priority time slice in 10 ns ticks, 0 = no turns (knob ACC_PRIO)
  uint32_t n_max;      // pairs of the largest MSM
  int c;               // requested maximum window width
  int W;               // windows of the full decomposition
  int win_begin;       // windows [win_begin, win_end) are computed by this call
  int win_end;
  uint32_t NB;         // bucket slots per MSM over [win_begin, win_end)
  uint32_t NS;         // bucket-reduce segments per MSM = NB / seg
  uint32_t seg;        // buckets per running-sum segment
  uint32_t G;          // segment results tree-summed per partial = min(256, smallest nbkt / seg)
  uint32_t max_nbkt;   // largest nbkt[] in range
  uint32_t L;          // sorted positions per accumulate lane
  uint32_t max_small;  // buckets with more fragments than this are pre-merged by a block
  uint32_t max_large;  // capacity of the large-bucket queue
  uint32_t chunk;      // pairs per sort block
  uint32_t gpu_combine;  // 1: window sums combined on the GPU (large batches), 0: on the host
  uint32_t fuse_scan;    // the bucket-slot scans: 0 six launches (multi-block; L = 1 only), 2 k_scan_one (one block, up to 32,768 slots), 3 k_scan_chain (one launch, any size)
  uint32_t glv;          // 1: scalars split with the endomorphism (bases must be in G1); 0: 255-bit scalars whole, any curve point
  uint32_t two_level;    // 1: the scatter runs in two passes (coarse bins, then buckets): single large MSMs
  // The bucket reduction without a scalar multiple (single MSMs; k_reduce_segments / k_reduce_groups):
  // a window's sum_b (b + 1) B_b leaves the GPU as `nout` points with bit positions, which the host's
  // Horner pass over the windows takes in like window sums.  Then G <= 16 (one wave's quads).
  uint32_t reduce_bits;  // 1: that form; 0: k_bucket_reduce_quad + window sums (batches, shared-scalar calls)
  uint32_t lg_seg;       // log2(seg)
  uint32_t lgG;          // log2(G)
  uint32_t NG;           // segment groups per MSM = NS / G
  uint32_t nout;         // points per window handed to the host = 2 + log2(max_nbkt / seg): sum S | - | X_0 ..
  uint8_t bits[kMaxWindows];    // width of window w
  uint16_t shift[kMaxWindows];  // bit offset of window w
  uint32_t nbkt[kMaxWindows];   // bucket slots of window w
  uint32_t base[kMaxWindows];   // first slot of window w inside an MSM's NB slots
};

// Sizes of the internal (fp28.h) point formats, for workspace allocation.
static constexpr size_t kX28Bytes = 224;
static constexpr size_t kA28Bytes = 128;  // stride of the internal point array: a 112-byte d28::A28 padded to one 128-byte line

// Device workspace, laid out by msm_enqueue.hip.  nb = k * NB bucket slots.
struct MsmWorkspace {
  const uint32_t* offsets;  // [k + 1]   first pair of each MSM (device)
  uint32_t* counts;   // [nb]      points per bucket
  uint32_t* starts;   // [nb + 1]  exclusive prefix of counts; [nb] = number of sorted entries
  uint32_t* cursor;   // [nb]      scatter cursors (copy of starts)
  uint32_t* fragcnt;  // [nb]      accumulation fragments per bucket
  uint32_t* foff;     // [nb + 1]  exclusive prefix of fragcnt
  uint32_t* blocksum; // [1024]    scan scratch
  uint32_t* large;    // [max_large] buckets queued for merge_large
  uint32_t* nlarge;   // [1]
  uint32_t* mdone;    // [sets][max_large] k_merge_large: finished chunks per queue entry (zero between calls)
  uint32_t* digits;   // [nw][n]   |digit| | sign<<31, window-major
  uint32_t* sorted;   // [nw * n]  pair index | sign<<31, grouped by bucket
  uint32_t* tmp;      // [nw * n]  two-level scatter: entries grouped by coarse bin (null otherwise)
  uint32_t* ccur;     // [coarse_words(nw)] two-level sort: the bins' cursors, their packed starts, and the copies of the coarse counts
  void* points28;     // [n]       input points in internal form (d28::A28, 112 B)
  void* frags;        // [nb + lanes + 1]  d28::X28 (224 B)
  void* partials;     // [k * NS / G]      d28::X28, one per group of G bucket-reduce lanes
  void* winsums28;    // [k][nw]           d28::X28 (large batches: combined on the GPU)
  G1XYZZ* winsums;    // [k][nw]   gnark-form XYZZ, canonical coordinates (host combine)
  G1XYZZ* results;    // [k]       XYZZ results of a batched call, gnark form (normalised by the host)
  // k_scan_chain (MsmPlan::fuse_scan == 3): the chain words [2][1024] (zero when allocated, never cleared again), the
  // ticket counter behind them, the count it stood at before this launch, this launch's epoch (never 0), and a word
  // of pinned host memory the kernel raises if a wait inside it gave up
  unsigned long long* chain;
  uint32_t* chain_ticket;
  uint32_t* host_err;
  uint32_t chain_base, chain_epoch;
};
uint32_t scan_chain_tiles(uint32_t nb);  // blocks (= tickets) one k_scan_chain launch over nb slots takes
size_t scan_chain_bytes();               // bytes behind MsmWorkspace::chain (the ticket counter at the end)

// The fragment lists the bucket reduction folds in: one per CHUNK of an MSM whose pairs were
// accumulated in several pieces over the same plan (host-buffer calls: a piece is accumulated
// while the next one crosses PCIe; msm_host_chunks.hip run_host_chunked) -- the pieces share the bucket
// slots, so their fragments meet in ONE reduction instead of one reduction per piece.
static constexpr int kMaxFragSources = 4;
struct FragSources {
  const void* frags[kMaxFragSources];        // d28::X28
  const uint32_t* foff[kMaxFragSources];     // [nb + 1]
  const uint32_t* fragcnt[kMaxFragSources];  // [nb]
  uint32_t n;
};

// Words of MsmWorkspace::ccur for nw windows.  Its tail (the coarse counts) must be ZERO before a call's first launch;
// the kernels leave it zero again (msm_enqueue.hip clears the buffer when it is made and after a failed call).
size_t coarse_words(uint32_t nw);

// The device accumulator's job as the fused front of a small call sees it (launch_dacc_front): device pointers into the
// uploaded job (checks | pool | loose points | loose scalars) and the counts.
struct DaccFront {
  const void* d_checks;
  const void* d_pool;
  const void* d_extra_points;
  const void* d_extra_scalars;
  void* d_scalars_out;  // [n_crs + n_inst + n_extra] fr.Elements, or null
  uint32_t n_checks, pool_len, n_crs, n_inst, n_extra;
};

// Every launcher enqueues on `stream` and returns the launch status.
// n gnark affine points -> internal form at d_out28 (kA28Bytes apart), outside a plan: the device accumulator
// converts its resident base sets once and per-verification points as they arrive.
hipError_t launch_convert_points_raw(const void* d_points, uint32_t n, void* d_out28, hipStream_t stream, uint32_t prio = 0);
// chunked host-buffer calls: this chunk's fragments into the per-bucket running sums (sums: NB points, meta: 2 * NB words)
hipError_t launch_fold_fragments(const MsmPlan& p, const MsmWorkspace& ws, void* sums, void* meta, bool first, hipStream_t stream);
hipError_t launch_digits(const MsmPlan& p, const MsmWorkspace& ws, const void* d_scalars, hipStream_t stream);
// conversion of npts points into ws.points28 and the recoding in ONE launch (small calls; not for two-level plans)
hipError_t launch_front(const MsmPlan& p, const MsmWorkspace& ws, const void* d_points, uint32_t npts, const void* d_scalars,
                        hipStream_t stream);
// loose-base conversion + slot scalars + recoding in ONE launch (small device-accumulator calls; not two-level plans)
hipError_t launch_dacc_front(const MsmPlan& p, const MsmWorkspace& ws, const DaccFront& f, hipStream_t stream);
hipError_t launch_hist(const MsmPlan& p, const MsmWorkspace& ws, hipStream_t stream);
hipError_t launch_scan(const MsmPlan& p, const MsmWorkspace& ws, hipStream_t stream);
hipError_t launch_scatter(const MsmPlan& p, const MsmWorkspace& ws, hipStream_t stream);
hipError_t launch_accumulate(const MsmPlan& p, const MsmWorkspace& ws, hipStream_t stream);
hipError_t launch_merge_large(const MsmPlan& p, const MsmWorkspace& ws, hipStream_t stream, uint32_t max_blocks);
// extra: fragment lists of earlier chunks (same plan) to fold in besides ws's own; may be null
hipError_t launch_bucket_reduce(const MsmPlan& p, const MsmWorkspace& ws, hipStream_t stream, const FragSources* extra = nullptr);
hipError_t launch_window_sum(const MsmPlan& p, const MsmWorkspace& ws, hipStream_t stream);
// The reduce_bits form: segments -> groups (ws.partials, room for 2 x kr x NG x (2 + lgG) + 2 points), groups ->
// nout points per window (ws.winsums), in one or two levels.
hipError_t launch_reduce_segments(const MsmPlan& p, const MsmWorkspace& ws, hipStream_t stream, const FragSources* extra = nullptr);
hipError_t launch_reduce_groups(const MsmPlan& p, const MsmWorkspace& ws, hipStream_t stream);
// Bit position (relative to the window's shift) of output point `slot` of window w in the reduce_bits
// form, or -1 for a slot the window does not use (it holds infinity).  Host and device agree on this.
int reduce_bits_position(const MsmPlan& p, int w, uint32_t slot);
// Batched calls only: Horner over each MSM's window sums, one quad per MSM.
hipError_t launch_combine(const MsmPlan& p, const MsmWorkspace& ws, hipStream_t stream);

// P_i = p0 + i*Q for i < n (n <= 2^27); d_table holds 27 affine points 2^j * Q.
hipError_t launch_synth_walk(const G1Affine* d_table, const G1Affine& p0, uint32_t n, void* d_out, hipStream_t stream);

// Element-wise primitive test (curdle_selftest_op); all pointers are device memory.
// ONE table says what an operation reads and writes per element and how many lanes it takes: the host
// wrapper sizes its buffers from it, the launcher its grid, the kernel its indexing, and the Python
// binding asks for it (curdle_selftest_shape) -- an operation missing from any of the four used to
// fall through to another operation's widths (the r3a abort).
struct SelftestOp {
  uint32_t in_words;   // 32-bit words read per element
  uint32_t out_words;  // ... written per element
  uint32_t lanes;      // lanes per element: 4 for the lane-distributed (quad) operations
};
static constexpr int kSelftestOps = 13;
static constexpr SelftestOp kSelftestTable[kSelftestOps] = {
    {24, 12, 1}, {24, 12, 1}, {24, 12, 1}, {24, 12, 1},  // 0..3  Fp mul / add / sub / sqr
    {16, 8, 1},                                           // 4     Fr Montgomery -> canonical
    {96, 48, 1}, {96, 48, 1}, {96, 48, 1},                // 5..7  XYZZ madd / add / dbl, one lane
    {96, 48, 4}, {96, 48, 4}, {96, 48, 4},                // 8..10 add / dbl / small multiple on quads
    {8, 10, 1},                                           // 11    the GLV split
    {24, 26, 1},                                          // 12    a base into the MSM's curve and back (+ the two bound flags)
};
// hipErrorInvalidValue for an op outside the table (nothing is launched).
hipError_t launch_selftest(int op, const uint32_t* d_in, size_t n, uint32_t* d_out, hipStream_t stream);

// decode_kernels.hip: n compressed G1 points (48 B each) -> n gnark affine points (24 u32
// each) + one CURDLE_DECODE_* status byte per point.
hipError_t launch_g1_decompress(const uint8_t* in, uint32_t n, int subgroup_check, uint32_t* out, uint8_t* status,
                                hipStream_t stream);

// The subgroup test straight from the n compressed records, without their square roots (so it
// can run beside launch_g1_decompress(..., subgroup_check = 0, ...)): sub[i] = 0 iff record i,
// if it decodes to a point at all, is not in the prime-order subgroup.
hipError_t launch_g1_subgroup_from_bytes(const uint8_t* in, uint32_t n, uint8_t* sub, hipStream_t stream);

// group_kernels.hip: out[i] = addends[i] + scalars[i or 0] * points[i] as gnark-format XYZZ
// (ZZ = 0 for infinity); points / addends gnark affine (addends may be null), scalars
// Montgomery fr.Elements.
// dacc_kernels.hip: the slot scalars of the device accumulator (checks: curdle_dacc_check[], pool: fr.Elements).
hipError_t launch_dacc_scalars(const void* d_checks, uint32_t n_checks, const void* d_pool, uint32_t pool_len, uint32_t n_crs,
                               uint32_t n_inst, void* d_out, hipStream_t stream);

hipError_t launch_scalar_mul_batch(const void* points, const void* scalars, int shared_scalar, const void* addends,
                                   uint32_t n, void* out_xyzz, hipStream_t stream);

}  // namespace curdle
