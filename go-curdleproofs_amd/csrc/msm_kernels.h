// Host-callable launchers of the gfx950 MSM kernels (msm_kernels.hip).
// Internal to libcurdlemsm.so; the public surface is include/curdle_msm.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "bls12_381.h"

namespace curdle {

// Pippenger decomposition of one MSM.  Scalars are recoded into W signed c-bit
// digits d_w in [-2^(c-1), 2^(c-1)], so a window has B = 2^(c-1) buckets
// (bucket b holds the points whose |digit| is b+1).
struct MsmPlan {
  uint32_t n;       // pairs
  int c;            // window bits
  int W;            // windows in the full decomposition
  uint32_t B;       // buckets per window = 2^(c-1)
  int win_begin;    // windows [win_begin, win_end) are computed by this call
  int win_end;
  uint32_t seg;     // buckets per running-sum segment in the bucket reduce
  uint32_t nseg;    // segments per window = B / seg
  uint32_t L;       // sorted positions per accumulate lane
  uint32_t max_small;  // buckets with more fragments than this are pre-merged by a block
  uint32_t max_large;  // capacity of the large-bucket queue (= grid of merge_large)
  uint32_t chunk;      // scalars per sort block
};

// Sizes of the internal (fp28.h) point formats, for workspace allocation.
static constexpr size_t kX28Bytes = 224;
static constexpr size_t kA28Bytes = 112;

// Device workspace, laid out by msm_api.hip.
struct MsmWorkspace {
  uint32_t* counts;   // [nb]      points per bucket, nb = nw * B slots (window-major)
  uint32_t* starts;   // [nb + 1]  exclusive prefix of counts; [nb] = number of sorted entries
  uint32_t* cursor;   // [nb]      scatter cursors (copy of starts)
  uint32_t* fragcnt;  // [nb]      accumulation fragments per bucket
  uint32_t* foff;     // [nb + 1]  exclusive prefix of fragcnt
  uint32_t* blocksum; // [1024]    scan scratch
  uint32_t* large;    // [max_large] buckets queued for merge_large
  uint32_t* nlarge;   // [1]
  uint32_t* digits;   // [nw][n]   |digit| | sign<<31, window-major
  uint32_t* sorted;   // [nw * n]  point index | sign<<31, grouped by bucket
  void* points28;     // [n]       input points in internal form (d28::A28, 112 B)
  void* frags;        // [nb + lanes + 1]  d28::X28 (224 B)
  void* partials;     // [nw][nseg]        d28::X28
  G1XYZZ* winsums;    // [nw]      gnark-form XYZZ, canonical coordinates
};

// Every launcher enqueues on `stream` and returns the launch status.
hipError_t launch_digits(const MsmPlan& p, const MsmWorkspace& ws, const void* d_scalars, hipStream_t stream);
hipError_t launch_hist(const MsmPlan& p, const MsmWorkspace& ws, hipStream_t stream);
hipError_t launch_scan(const MsmPlan& p, const MsmWorkspace& ws, hipStream_t stream);
hipError_t launch_scatter(const MsmPlan& p, const MsmWorkspace& ws, hipStream_t stream);
hipError_t launch_convert_points(const MsmPlan& p, const MsmWorkspace& ws, const void* d_points, hipStream_t stream);
hipError_t launch_accumulate(const MsmPlan& p, const MsmWorkspace& ws, hipStream_t stream);
hipError_t launch_merge_large(const MsmPlan& p, const MsmWorkspace& ws, hipStream_t stream);
hipError_t launch_bucket_reduce(const MsmPlan& p, const MsmWorkspace& ws, hipStream_t stream);
hipError_t launch_window_sum(const MsmPlan& p, const MsmWorkspace& ws, hipStream_t stream);

// P_i = p0 + i*Q for i < n (n <= 2^27); d_table holds 27 affine points 2^j * Q.
hipError_t launch_synth_walk(const G1Affine* d_table, const G1Affine& p0, uint32_t n, void* d_out, hipStream_t stream);

// Element-wise primitive test (curdle_selftest_op); all pointers are device memory.
hipError_t launch_selftest(int op, const uint32_t* d_in, size_t n, uint32_t* d_out, hipStream_t stream);

}  // namespace curdle
