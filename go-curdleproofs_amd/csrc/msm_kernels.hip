// gfx950 kernels of the BLS12-381 G1 MSM (Pippenger bucket method).
//
// Replaces the body of gnark-crypto's (*G1Jac).MultiExp as the reference uses it
// (/root/reference/msmaccumulator/msmaccumulator.go:59 and the call sites in
// SURVEY.md section 8a).  Phases, one kernel each:
//   convert     gnark affine points (96 B, R = 2^384) -> internal form (fp28.h: 14 limbs of 28
//               bits, R = 2^392), TWO 128-byte records per point: P and phi(P) = (beta x, y)
//   digits      scalar Montgomery->canonical, the GLV split k = k1 + k2 lambda into two
//               127-bit halves, signed c-bit digits of both (window-major): 2n terms, W =
//               ceil(127 / c) windows
//   hist        bucket sizes, per-window histogram staged in LDS
//   scan        exclusive prefix of the bucket sizes over all (window, bucket)
//               slots, then of the per-bucket fragment counts -- ONE launch: k_scan_one (one block, synchronous
//               calls up to 32,768 slots) or k_scan_chain (any size: tiles hand their sums down a chain)
//   scatter     term indices grouped by (window, bucket).  One pass (LDS histogram again, one
//               returning atomic per (block, bucket) reserves the range) for batches and small
//               MSMs.  Single large MSMs sort COARSE-FIRST (round 5): the recoding also counts the
//               terms per coarse bin of 128 buckets, the terms are written grouped by bin as
//               contiguous runs, the buckets are counted on that array (a tile of 2,048 entries
//               touches <= 128 counters), scanned, and the entries placed inside their bins
//   accumulate  one lane per L consecutive sorted positions: gathers the internal affine
//               points (112 B in one 128-byte line each), sums them with XYZZ mixed additions
//               and emits one fragment per bucket it touches, so the work per lane is the same
//               however skewed the scalars are
//   merge_large block-per-bucket tree sum for buckets with many fragments
//   reduce      sum_b (b+1)*bucket[b] per window, as running sums over short
//               segments on quads (four lanes per point, quad28.h: the serial chain is what
//               matters); folds in the fragment lists of up to four chunks of one MSM
//   window_sum  per-window tree sum of the segment results
// The last 127 doublings (combining the <= 32 window sums of the 127-bit halves) are O(1) work
// with a serial dependency chain and are done by the host side of the library for single MSMs,
// by k_combine (one quad per MSM) for batches.
//
// This is 381-bit integer arithmetic: no MFMA, no floating point.  Wave size 64.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>

#include "msm_kernels.h"

#include "fp28.h"
#include "quad28.h"
#include "dacc_eval.h"

namespace curdle {

using d28::A28;
using d28::F28;
using d28::X28;

static constexpr int kBlock = 256;
// s_getreg_b32 operands: (size - 1) << 11 | offset << 6 | register id.  HW_ID (4): wave slot 3:0, SIMD 5:4, CU 11:8,
// SH 12, SE 15:13; XCC_ID (20): the XCD in 3:0.
static constexpr int kGetregHwId = ((32 - 1) << 11) | 4;
static constexpr int kGetregXccId = ((32 - 1) << 11) | 20;
// s_setprio takes an immediate
__device__ __forceinline__ void set_wave_prio(u32 v) {
  if (v == 1) __builtin_amdgcn_s_setprio(1);
  else if (v == 2) __builtin_amdgcn_s_setprio(2);
  else if (v == 3) __builtin_amdgcn_s_setprio(3);
}

// p - 2 (Fermat inversion exponent), 32-bit words
__constant__ u32 kPminus2[12] = {0xffffaaa9u, 0xb9feffffu, 0xb153ffffu, 0x1eabfffeu, 0xf6b0f624u, 0x6730d2a0u,
                                 0xf38512bfu, 0x64774b84u, 0x434bacd7u, 0x4b1ba7b6u, 0x397fe69au, 0x1a0111eau};

// the two-level sort of single large MSMs (further down): bins of 2^kFineBits buckets
static constexpr int kFineBits = 7;
static constexpr u32 kFineMask = (1u << kFineBits) - 1u;
static constexpr int kCoarseMax = 256;          // bins per window: 32,768 buckets / 128
static constexpr int kCoarseThreads = 512;
static constexpr int kCoarseTile = 8192;        // terms per block of the first pass: 32-entry (one line) runs per bin on average
static constexpr int kFineTile = 2048;          // positions per block of the second pass
static constexpr int kFineCap = 1024;           // LDS counters of the second pass (bucket slots per sweep)

// ---------------------------------------------------------------------------
// Digit recoding, for w = 0..W-1 in order: windows
// below the top are signed (a raw digit above half the window range becomes its
// negative complement and carries one into the next window), the top window is
// unsigned.  The scalar is shifted down window by window so every limb index is
// static (runtime-indexed register arrays would go to scratch).
// ---------------------------------------------------------------------------
// The scalar here is one HALF of a GLV split (below): 127 bits in four words (the loop is in k_digits).

// GLV: phi(x, y) = (beta x, y) is multiplication by lambda = z^2 - 1 (lambda^2 + lambda + 1 =
// r), so k P = k1 P + k2 phi(P) with half-length k1, k2: the MSM runs over 2n points with half
// the windows -- the same bucket additions, half the buckets and half the Horner pass.  The
// split itself is glv_split (bls12_381.h, shared with the host's scalar multiplication).

__device__ __forceinline__ Fr load_scalar_canonical(const uint4* scalars, u32 i) {
  uint4 lo = scalars[2 * (size_t)i], hi = scalars[2 * (size_t)i + 1];
  Fr m, s;
  m.l[0] = lo.x; m.l[1] = lo.y; m.l[2] = lo.z; m.l[3] = lo.w;
  m.l[4] = hi.x; m.l[5] = hi.y; m.l[6] = hi.z; m.l[7] = hi.w;
  f_from_mont<FrParams>(s, m);  // gnark fr.Element is Montgomery; digits need the integer
  return s;
}

// Phase 1: recode every scalar once.  digits[lw][i] = |d| | sign << 31 (0 = no
// contribution), window-major so the sort passes below read them coalesced.
//
// GLV = false (MsmPlan::glv == 0, CURDLE_MSM_ANY_CURVE_POINT): no endomorphism -- the 255-bit scalar is
// recoded whole over W = ceil(255 / c) windows as term 2 i, term 2 i + 1 (the phi(P) record) never
// contributes.  The result is then k P for EVERY point of the curve, in the subgroup or not.
//
// COARSE = true (MsmPlan::two_level: single large MSMs): the same launch also counts the terms per COARSE
// bin (128 buckets) of every window in LDS; k_coarse_scan (one block) turns the counts into the bins' first
// positions (ccur: the coarse scatter's cursors; cstart: the same packed over the windows in slot order,
// with the total behind them, for the fine passes).  Round 5: the sort of such an MSM used
// to begin with a 32,768-counter LDS histogram per block and ~1.9 M global atomics per window (k_hist:
// 0.052 ms for ONE window of N = 2^20) and a scan of all bucket slots BEFORE anything could be placed; now the
// coarse partition needs only these <= 256 counters per window, and the buckets are counted afterwards on
// the bin-grouped array, where a tile of 2,048 entries touches <= 128 counters (k_scatter_fine<true>).
static constexpr int kCoarseWinMax = 20;  // windows of a two-level plan: c >= 13, so <= 10 with the split, <= 20 without
// Blocks publish their coarse counts into one of kCoarseReps copies of the counters (blockIdx mod kCoarseReps): 4,096
// blocks adding into ONE copy took 0.35 ms at N = 2^20 -- 4,096 atomics on each of 256 addresses, one after the other
// (gpurun_out/r5d) -- and the launch is capped at kDigitsCoarseMaxBlocks blocks, each walking several tiles of scalars.
static constexpr int kCoarseReps = 16;
struct CoarseOut {
  u32* ccount;  // [kCoarseReps][nw * 256] zero when k_digits starts; k_coarse_scan leaves them zero again
  u32* ccur;    // [nw * 256]
  u32* cstart;  // [bins + 1]
};
// The coarse build runs few, large blocks: what a block publishes at its end is nw x 256 global atomics however few
// scalars it walked, and the chip completes ~50 G of them per second (1,024 blocks of 256 threads: +0.03-0.04 ms on
// the launch at N = 2^20, gpurun_out/r5f).  512 threads, not 1,024: at 40 registers two waves per SIMD fit beside
// the two 180-register waves of a neighbouring call's accumulation (512 - 360 = 152), four do not -- a block that
// cannot be placed waits for accumulate waves to END, and the launch took 0.2-0.4 ms in a chunked host-buffer
// call (rocprofv3 timeline, gpurun_out/r5m).
static constexpr int kDigitsCoarseBlock = 512;
static constexpr int kDigitsCoarseMaxBlocks = 512;
// One scalar (canonical integer, below r) -> the digits of pair i in every window of the call's range.
template <bool GLV, bool COARSE>
__device__ __forceinline__ void recode_scalar(const Fr& s, const u32 i, const MsmPlan& p, u32* __restrict__ digits, u32* cc) {
  constexpr int NA = GLV ? 4 : 8;
  u32 a[NA], b[4], neg_a = 0, neg_b = 0;
  if constexpr (GLV) {
    glv_split(s, a, b, neg_a, neg_b);
  } else {
#pragma unroll
    for (int j = 0; j < 8; j++) a[j] = s.l[j];
#pragma unroll
    for (int j = 0; j < 4; j++) b[j] = 0;
  }
  // the two halves' digits of a window leave as ONE 8-byte store: written one half after the other, every
  // line of `digits` went to memory twice (128 MB for a 64 MB array at N = 2^20, profiles/r04_pmc_summary.txt)
  u32 ca = 0, cb = 0;
  for (int w = 0; w < p.W; w++) {
    const u32 c = p.bits[w];
    const u32 ra = (a[0] & ((1u << c) - 1u)) + ca, rb = (b[0] & ((1u << c) - 1u)) + cb;
#pragma unroll
    for (int j = 0; j + 1 < NA; j++) a[j] = (a[j] >> c) | (a[j + 1] << (32 - c));
    a[NA - 1] >>= c;
    if constexpr (GLV) {
      b[0] = (b[0] >> c) | (b[1] << (32 - c));
      b[1] = (b[1] >> c) | (b[2] << (32 - c));
      b[2] = (b[2] >> c) | (b[3] << (32 - c));
      b[3] >>= c;
    }
    u32 ma = ra, na = 0, mb = rb, nb2 = 0;
    ca = cb = 0;
    if (w != p.W - 1) {  // windows below the top are signed (header comment above)
      if (ra > (1u << (c - 1))) {
        ma = (1u << c) - ra;
        na = 0x80000000u;
        ca = 1;
      }
      if (rb > (1u << (c - 1))) {
        mb = (1u << c) - rb;
        nb2 = 0x80000000u;
        cb = 1;
      }
    }
    if (w >= p.win_begin && w < p.win_end) {
      const u32 lw = (u32)(w - p.win_begin);
      *reinterpret_cast<uint2*>(&digits[(size_t)lw * p.n + 2 * (size_t)i]) =
          make_uint2(ma ? (ma | (na ^ neg_a)) : 0u, mb ? (mb | (nb2 ^ neg_b)) : 0u);
      if constexpr (COARSE) {
        if (ma) atomicAdd(&cc[lw * 256u + ((ma - 1u) >> kFineBits)], 1u);
        if (mb) atomicAdd(&cc[lw * 256u + ((mb - 1u) >> kFineBits)], 1u);
      }
    }
  }
}

// (block `bid` of `nblocks`: the body also runs as the second role of k_front below)
template <bool GLV, bool COARSE>
__device__ __forceinline__ void digits_body(const uint4* __restrict__ scalars, const MsmPlan& p, u32* __restrict__ digits,
                                            u32* __restrict__ counts, u32 nb, const CoarseOut& co, const u32 bid, const u32 nblocks) {
  constexpr u32 kBlock = COARSE ? kDigitsCoarseBlock : curdle::kBlock;  // the block size of the kernels that run this body
  __shared__ u32 cc[COARSE ? kCoarseWinMax * 256 : 1];
  set_wave_prio(p.aux_prio);
  const u32 tid = threadIdx.x;
  const u32 nw = (u32)(p.win_end - p.win_begin);
  // the histogram's counters start from zero: cleared here, one launch before the first kernel adds into
  // them, instead of by a memset node of their own
  for (u32 b = bid * kBlock + tid; b < nb; b += nblocks * kBlock) counts[b] = 0;
  if constexpr (COARSE) {
    for (u32 x = tid; x < nw * 256u; x += kBlock) cc[x] = 0;
    __syncthreads();
  }
  // p.n counts the split's terms: entry 2 i is k1 P_i, entry 2 i + 1 is k2 phi(P_i)
  for (u32 i = bid * kBlock + tid; i < p.n / 2; i += nblocks * kBlock) {
    const Fr s = load_scalar_canonical(scalars, i);
    recode_scalar<GLV, COARSE>(s, i, p, digits, cc);
  }
  if constexpr (COARSE) {
    __syncthreads();
    u32* mine = co.ccount + (size_t)(bid % kCoarseReps) * nw * 256u;
    for (u32 x = tid; x < nw * 256u; x += kBlock) {
      const u32 v = cc[x];
      if (v) atomicAdd(&mine[x], v);
    }
  }
}
template <bool GLV, bool COARSE>
__global__ void __launch_bounds__(COARSE ? kDigitsCoarseBlock : kBlock) k_digits(const uint4* __restrict__ scalars, MsmPlan p,
                                                   u32* __restrict__ digits, u32* __restrict__ counts, u32 nb, CoarseOut co) {
  digits_body<GLV, COARSE>(scalars, p, digits, counts, nb, co, blockIdx.x, gridDim.x);
}

// The coarse counts -> the bins' first positions: one block.  (Round 5's first build let the LAST block of k_digits do
// this, found by a ticket every block incremented: 1,024 atomics on one address took 0.08 ms -- same-address device
// atomics complete one after the other, ~80 ns each on this chip (gpurun_out/r5e) -- a launch of its own is ~7 us.)
// (late in round 5: at the call's wave priority like the other sort kernels -- beside an accumulation this one block took
// 0.18 ms of a chunk's sort against 0.02 alone, rocprofv3 timeline gpurun_out/r5_hosttrace2 -- and the copies read and
// cleared 16 bytes at a time: 32 loads and stores per thread at eight windows instead of 128)
__global__ void __launch_bounds__(kBlock) k_coarse_scan(MsmPlan p, CoarseOut co) {
  __shared__ u32 cc[kCoarseWinMax * 256];
  __shared__ u32 sh_scan[kBlock];
  set_wave_prio(p.aux_prio);
  const u32 tid = threadIdx.x;
  const u32 nw = (u32)(p.win_end - p.win_begin);
  const u32 tot = nw * 256u;
  // the counts are summed over the copies and left zero for the next call
  for (u32 x4 = tid; x4 < tot / 4u; x4 += kBlock) {
    uint4 v = make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < kCoarseReps; r++) {
      uint4* at = reinterpret_cast<uint4*>(&co.ccount[(size_t)r * tot]) + x4;
      const uint4 q = *at;
      v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w;
      *at = make_uint4(0, 0, 0, 0);
    }
    cc[4 * x4] = v.x; cc[4 * x4 + 1] = v.y; cc[4 * x4 + 2] = v.z; cc[4 * x4 + 3] = v.w;
  }
  __syncthreads();
  // exclusive prefix over the windows' bins in slot order (a window's unused bins hold zero): thread t
  // takes per = ceil(tot / 256) consecutive entries
  const u32 per = (tot + kBlock - 1) / kBlock;
  const u32 lo = min(tid * per, tot), hi = min(lo + per, tot);
  u32 sum = 0;
  for (u32 x = lo; x < hi; x++) sum += cc[x];
  sh_scan[tid] = sum;
  __syncthreads();
  for (u32 off = 1; off < (u32)kBlock; off <<= 1) {
    const u32 t = tid >= off ? sh_scan[tid - off] : 0u;
    __syncthreads();
    sh_scan[tid] += t;
    __syncthreads();
  }
  u32 run = sh_scan[tid] - sum;
  const u32 total = sh_scan[kBlock - 1];
  for (u32 x = lo; x < hi; x++) {
    const u32 lw = x >> 8, bin = x & 255u;
    const u32 w = (u32)p.win_begin + lw;
    const u32 nbins = p.nbkt[w] >> kFineBits;
    if (bin < nbins) {
      co.ccur[x] = run;
      u32 first = 0;
      for (u32 y = (u32)p.win_begin; y < w; y++) first += p.nbkt[y] >> kFineBits;
      co.cstart[first + bin] = run;
      if (lw == nw - 1 && bin == nbins - 1) co.cstart[first + nbins] = total;  // the sentinel: every entry lies below it
    }
    run += cc[x];
  }
}

// Bucket sort of the pair indices, per-window histogram staged in LDS (at most
// 32768 counters = 128 KiB of the CU's 160 KiB).  Block (x, lw, j) owns pairs
// [off[j] + x*chunk, ...) of MSM j for local window lw.
static constexpr int kSortThreads = 1024;

// Phase 2: bucket sizes.  LDS atomics absorb the increments; one coalesced
// global atomic per (block, non-empty bucket) publishes them.
__global__ void __launch_bounds__(kSortThreads) k_hist(const u32* __restrict__ digits, MsmPlan p,
                                                      const u32* __restrict__ offsets, u32* __restrict__ counts) {
  extern __shared__ u32 lds_cnt[];
  set_wave_prio(p.aux_prio);
  const u32 lw = blockIdx.y, j = blockIdx.z;
  // a single MSM covers pairs [0, n): no offsets array to wait for
  const u32 i0 = (p.k == 1 ? 0u : offsets[j]) + blockIdx.x * p.chunk;
  const u32 i1 = min(i0 + p.chunk, p.k == 1 ? p.n : offsets[j + 1]);
  if (i0 >= i1) return;  // block-uniform
  const u32 nb = p.nbkt[p.win_begin + lw];
  for (u32 b = threadIdx.x; b < nb; b += kSortThreads) lds_cnt[b] = 0;
  __syncthreads();
  const u32* dw = digits + (size_t)lw * p.n;
  for (u32 i = i0 + threadIdx.x; i < i1; i += kSortThreads) {
    u32 mag = dw[i] & 0x7fffffffu;
    if (mag) atomicAdd(&lds_cnt[mag - 1], 1u);
  }
  __syncthreads();
  u32* cw = counts + (size_t)j * p.NB + p.base[p.win_begin + lw];
  for (u32 b = threadIdx.x; b < nb; b += kSortThreads) {
    u32 v = lds_cnt[b];
    if (v) atomicAdd(&cw[b], v);
  }
}

// Phase 4: scatter.  The block rebuilds its local histogram, reserves a range per
// bucket with one returning global atomic (wavefront-coalesced), then hands out
// positions inside the ranges with LDS atomics.
__global__ void __launch_bounds__(kSortThreads) k_scatter(const u32* __restrict__ digits, MsmPlan p,
                                                         const u32* __restrict__ offsets, u32* __restrict__ cursor,
                                                         u32* __restrict__ sorted) {
  extern __shared__ u32 lds_cnt[];
  set_wave_prio(p.aux_prio);
  const u32 lw = blockIdx.y, j = blockIdx.z;
  const u32 i0 = (p.k == 1 ? 0u : offsets[j]) + blockIdx.x * p.chunk;
  const u32 i1 = min(i0 + p.chunk, p.k == 1 ? p.n : offsets[j + 1]);
  if (i0 >= i1) return;  // block-uniform
  const u32 nb = p.nbkt[p.win_begin + lw];
  for (u32 b = threadIdx.x; b < nb; b += kSortThreads) lds_cnt[b] = 0;
  __syncthreads();
  const u32* dw = digits + (size_t)lw * p.n;
  for (u32 i = i0 + threadIdx.x; i < i1; i += kSortThreads) {
    u32 mag = dw[i] & 0x7fffffffu;
    if (mag) atomicAdd(&lds_cnt[mag - 1], 1u);
  }
  __syncthreads();
  u32* cw = cursor + (size_t)j * p.NB + p.base[p.win_begin + lw];
  for (u32 b = threadIdx.x; b < nb; b += kSortThreads) {
    u32 v = lds_cnt[b];
    if (v) lds_cnt[b] = atomicAdd(&cw[b], v);
  }
  __syncthreads();
  for (u32 i = i0 + threadIdx.x; i < i1; i += kSortThreads) {
    u32 d = dw[i];
    u32 mag = d & 0x7fffffffu;
    if (mag) {
      u32 pos = atomicAdd(&lds_cnt[mag - 1], 1u);
      sorted[pos] = i | (d & 0x80000000u);
    }
  }
}

// ---------------------------------------------------------------------------
// Two-level scatter (single large MSMs).  k_scatter above writes 4 bytes at a time to positions
// spread over the whole window's range -- at N = 2^20 a block's 32,768 terms go to ~32,768
// different buckets of an 8 MB range, every store dirties its own line and the launch writes
// 587 MB for a 64 MB payload (profiles/pmc_traffic.json, round 2).  Here the terms are first
// partitioned by the COARSE part of their bucket (bins of 2^kFineBits buckets; a bin's range in
// the output is known from the fine scan: starts[] at the bin's first bucket), sorted by bin in
// LDS and written out as contiguous runs; the second pass walks the bin-grouped array in tiles of
// consecutive positions and hands out the final positions with the old scheme (LDS counters, one
// returning global atomic per tile and bucket) -- but a tile's stores now land in the few dozen
// kilobytes its own bins cover, which the L2 merges into whole lines.
// An entry of the intermediate array: term index (24 bits) | fine bucket (7 bits) << 24 | sign << 31.
// ---------------------------------------------------------------------------

// coarse cursors (written by the last block of k_digits<.., true>): ccur[lw * kCoarseMax + bin] = global position
// of the bin's first entry (advanced by the first pass), and the same positions packed over all windows
// in slot order, with the total behind them, for the fine passes: cstart[binbase(lw) + bin]
__global__ void __launch_bounds__(kCoarseThreads)
    k_scatter_coarse(const u32* __restrict__ digits, MsmPlan p, u32* __restrict__ ccur, u32* __restrict__ tmp) {
  __shared__ u32 cnt[kCoarseMax], off[kCoarseMax + 1], gbase[kCoarseMax];
  set_wave_prio(p.aux_prio);
  extern __shared__ u32 lds_cnt[];  // 40 KiB (kCoarseTile * 5 bytes): the tile's entries sorted by bin, and each entry's bin -- below the 64 KiB default, no opt-in needed
  u32* buf = lds_cnt;
  unsigned char* binof = reinterpret_cast<unsigned char*>(lds_cnt + kCoarseTile);
  const u32 tid = threadIdx.x, lw = blockIdx.y;
  const u32 i0 = blockIdx.x * kCoarseTile;
  if (i0 >= p.n) return;  // block-uniform
  constexpr int PER = kCoarseTile / kCoarseThreads;
  if (tid < kCoarseMax) cnt[tid] = 0;
  __syncthreads();
  const u32* dw = digits + (size_t)lw * p.n;
  u32 word[PER], rank[PER];
#pragma unroll
  for (int j = 0; j < PER; j++) {
    const u32 i = i0 + j * kCoarseThreads + tid;
    const u32 d = i < p.n ? dw[i] : 0u;
    const u32 mag = d & 0x7fffffffu;
    word[j] = 0xffffffffu;  // no entry
    rank[j] = 0;
    if (mag) {
      const u32 bkt = mag - 1;
      rank[j] = atomicAdd(&cnt[bkt >> kFineBits], 1u) | ((bkt >> kFineBits) << 16);  // rank inside the bin (< 16,384) | bin
      word[j] = i | ((bkt & kFineMask) << 24) | (d & 0x80000000u);
    }
  }
  __syncthreads();
  // exclusive scan of the bin sizes (256 values: Hillis-Steele in LDS), and the bins' global ranges
  if (tid < kCoarseMax) off[tid + 1] = cnt[tid];
  if (tid == 0) off[0] = 0;
  __syncthreads();
  for (u32 step = 1; step < kCoarseMax; step <<= 1) {
    u32 v = 0;
    if (tid < kCoarseMax && tid + 1 > step) v = off[tid + 1 - step];
    __syncthreads();
    if (tid < kCoarseMax) off[tid + 1] += v;
    __syncthreads();
  }
  if (tid < kCoarseMax && cnt[tid]) gbase[tid] = atomicAdd(&ccur[lw * kCoarseMax + tid], cnt[tid]);
  __syncthreads();
#pragma unroll
  for (int j = 0; j < PER; j++) {
    if (word[j] != 0xffffffffu) {
      const u32 bin = rank[j] >> 16;
      const u32 pos = off[bin] + (rank[j] & 0xffffu);
      buf[pos] = word[j];
      binof[pos] = (unsigned char)bin;
    }
  }
  __syncthreads();
  const u32 total = off[kCoarseMax];
  for (u32 pos = tid; pos < total; pos += kCoarseThreads) {
    const u32 bin = binof[pos];
    tmp[gbase[bin] + (pos - off[bin])] = buf[pos];  // consecutive lanes, consecutive addresses inside a bin's run
  }
}

// Second pass: a tile of consecutive positions of the bin-grouped array -> final positions.
// cstart: the first position of every bin, bins of all windows in slot order, plus the total as a
// sentinel (k_coarse_init).  Small blocks (256 threads, 2,048 positions, 4 KiB of counters), many
// per compute unit: a block is a chain of dependent steps (find the tile's bins, count, reserve
// with returning global atomics, hand out), so it is the number of blocks in flight that hides
// their latency -- with 1,024 threads and 8,192 positions per block the launch took 0.10 ms at
// N = 2^20, however the inside was arranged (entries kept in registers, bin tables in LDS, output
// staged in LDS for contiguous stores: 0.099-0.152 ms).
// COUNT = true is the same walk as a counting pass (round 5): the tile's entries per bucket slot, published with
// one global atomic per touched slot -- at most 128 per bin the tile covers, where the LDS histogram over the
// unsorted digits (k_hist) needed one per (block, non-empty bucket): 1.9 M per window at N = 2^20, now 0.13 M.
// `cursor` is then the array of bucket sizes (zeroed by k_digits), `sorted` is not touched.
static constexpr int kFineThreads = 256;
static constexpr int kFineBinsCached = 64;
template <bool COUNT>
__global__ void __launch_bounds__(kFineThreads)
    k_scatter_fine(const u32* __restrict__ tmp, MsmPlan p, const u32* __restrict__ cstart, u32 nbins, u32* __restrict__ cursor,
                   u32* __restrict__ sorted) {
  __shared__ u32 cnt[kFineCap], loff[kFineCap], gb[kFineCap];      // per slot of the sweep: entries, offset in the tile, reserved position
  __shared__ u32 buf[kFineTile];                                   // the tile's entries ordered by slot
  __shared__ unsigned short sl[kFineTile];                         // ... and each one's slot - s0
  __shared__ u32 scan_sh[kFineThreads];
  __shared__ u32 lb[kFineBinsCached + 1], ls[kFineBinsCached + 1];  // the tile's bins: first position, first slot
  __shared__ u32 sh_bin[2];
  set_wave_prio(p.aux_prio);
  const u32 tid = threadIdx.x;
  const u32 total = cstart[nbins];
  const u32 a = blockIdx.x * kFineTile;
  if (a >= total) return;  // block-uniform
  const u32 b = min(a + kFineTile, total);
  auto bin_slot = [&](u32 bin) {  // bin index -> first bucket slot of the bin
    u32 first = 0;
    int w = p.win_begin;
    for (;;) {
      const u32 nbw = p.nbkt[w] >> kFineBits;
      if (bin < first + nbw) return p.base[w] + ((bin - first) << kFineBits);
      first += nbw;
      w++;
    }
  };
  auto bin_of = [&](u32 pos) {  // last bin with cstart[bin] <= pos (empty bins share their start with the next one)
    u32 lo = 0, hi = nbins;
    while (lo < hi) {
      const u32 mid = (lo + hi) >> 1;
      if (cstart[mid] > pos) hi = mid;
      else lo = mid + 1;
    }
    return lo - 1;
  };
  if (tid < 2) sh_bin[tid] = bin_of(tid == 0 ? a : b - 1);
  __syncthreads();
  const u32 bin_lo = sh_bin[0], bin_hi = sh_bin[1];  // two or three bins as a rule
  const u32 nbt = bin_hi - bin_lo + 1;
  const bool cached = nbt <= (u32)kFineBinsCached;
  if (cached && tid <= nbt) {
    lb[tid] = cstart[bin_lo + tid];
    ls[tid] = tid < nbt ? bin_slot(bin_lo + tid) : 0u;
  }
  __syncthreads();
  const u32 g_lo = bin_slot(bin_lo);
  const u32 g_end = bin_slot(bin_hi) + (1u << kFineBits);
  constexpr int PER = kFineTile / kFineThreads;
  u32 ent[PER], slot[PER];
#pragma unroll
  for (int j = 0; j < PER; j++) {
    const u32 pos = a + j * kFineThreads + tid;
    slot[j] = 0xffffffffu;
    ent[j] = 0;
    if (pos < b) {
      ent[j] = tmp[pos];
      u32 first_slot;
      if (cached) {
        u32 i = 0;
        while (i + 1 < nbt && lb[i + 1] <= pos) i++;
        first_slot = ls[i];
      } else {
        first_slot = bin_slot(bin_of(pos));
      }
      slot[j] = first_slot + ((ent[j] >> 24) & kFineMask);
    }
  }
  // sweeps over the covered slots, kFineCap at a time (one sweep unless the tile's bins are tiny):
  // count (the atomic's return value is the entry's rank in its bucket), prefix of the counts inside
  // the tile, one returning global atomic per touched bucket, the entries laid out by bucket in LDS
  // and written out in that order -- consecutive lanes, consecutive addresses inside a bucket's run
  u32 rank[PER];
  for (u32 s0 = g_lo; s0 < g_end; s0 += kFineCap) {
    const u32 s1 = min(s0 + (u32)kFineCap, g_end);
    const u32 S = s1 - s0;
    for (u32 k = tid; k < kFineCap; k += kFineThreads) cnt[k] = 0;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < PER; j++)
      if (slot[j] >= s0 && slot[j] < s1) rank[j] = atomicAdd(&cnt[slot[j] - s0], 1u);
    __syncthreads();
    if constexpr (COUNT) {
      for (u32 k = tid; k < S; k += kFineThreads) {
        const u32 v = cnt[k];
        if (v) atomicAdd(&cursor[s0 + k], v);
      }
      __syncthreads();
      continue;
    }
    constexpr int CPT = kFineCap / kFineThreads;  // consecutive counters per thread
    u32 c[CPT], sum = 0;
#pragma unroll
    for (int x = 0; x < CPT; x++) {
      c[x] = cnt[tid * CPT + x];
      sum += c[x];
    }
    // exclusive scan of the 256 partial sums
    scan_sh[tid] = sum;
    __syncthreads();
    for (u32 step = 1; step < kFineThreads; step <<= 1) {
      const u32 t = tid >= step ? scan_sh[tid - step] : 0;
      __syncthreads();
      scan_sh[tid] += t;
      __syncthreads();
    }
    const u32 tile_total = scan_sh[kFineThreads - 1];
    u32 run = scan_sh[tid] - sum;
#pragma unroll
    for (int x = 0; x < CPT; x++) {
      const u32 k = tid * CPT + x;
      loff[k] = run;
      run += c[x];
      if (c[x] && k < S) gb[k] = atomicAdd(&cursor[s0 + k], c[x]);
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < PER; j++)
      if (slot[j] >= s0 && slot[j] < s1) {
        const u32 k = slot[j] - s0;
        const u32 at = loff[k] + rank[j];
        buf[at] = ent[j] & 0x80ffffffu;
        sl[at] = (unsigned short)k;
      }
    __syncthreads();
    for (u32 i = tid; i < tile_total; i += kFineThreads) {
      const u32 k = sl[i];
      sorted[gb[k] + (i - loff[k])] = buf[i];
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------
// Exclusive prefix sums over all (window, bucket) slots, three small kernels:
// block-local scan (4096 slots per block), scan of the block totals, fix-up.
// out has len + 1 entries; out[len] = grand total.
// ---------------------------------------------------------------------------
static constexpr int kScanThreads = 1024;
static constexpr int kScanItems = 4;
static constexpr int kScanTile = kScanThreads * kScanItems;

__device__ __forceinline__ u32 block_exclusive_scan_1024(u32 v, u32* sh, u32& total) {
  const u32 tid = threadIdx.x;
  sh[tid] = v;
  __syncthreads();
  for (u32 off = 1; off < kScanThreads; off <<= 1) {
    u32 t = tid >= off ? sh[tid - off] : 0;
    __syncthreads();
    sh[tid] += t;
    __syncthreads();
  }
  total = sh[kScanThreads - 1];
  return sh[tid] - v;
}

__global__ void __launch_bounds__(kScanThreads) k_scan_local(const u32* __restrict__ in, u32 len, u32* __restrict__ out,
                                                            u32* __restrict__ blocksum) {
  __shared__ u32 sh[kScanThreads];
  const u32 base = blockIdx.x * kScanTile + threadIdx.x * kScanItems;
  u32 v[kScanItems], sum = 0;
#pragma unroll
  for (int k = 0; k < kScanItems; k++) {
    v[k] = base + k < len ? in[base + k] : 0;
    sum += v[k];
  }
  u32 total;
  u32 run = block_exclusive_scan_1024(sum, sh, total);
#pragma unroll
  for (int k = 0; k < kScanItems; k++) {
    if (base + k < len) out[base + k] = run;
    run += v[k];
  }
  if (threadIdx.x == 0) blocksum[blockIdx.x] = total;
}

// Single block: exclusive scan of the (<= 1024) block totals in place; grand total to out[len].
__global__ void __launch_bounds__(kScanThreads) k_scan_top(u32* __restrict__ blocksum, u32 nblocks, u32* __restrict__ out,
                                                          u32 len) {
  __shared__ u32 sh[kScanThreads];
  u32 v = threadIdx.x < nblocks ? blocksum[threadIdx.x] : 0;
  u32 total;
  u32 ex = block_exclusive_scan_1024(v, sh, total);
  if (threadIdx.x < nblocks) blocksum[threadIdx.x] = ex;
  if (threadIdx.x == 0) out[len] = total;
}

// Fix-up after the scan of the bucket sizes: global start of every bucket, the
// scatter cursor, and the number of accumulation fragments the bucket will have.
// Lane t of the accumulate kernel owns sorted positions [t*L, (t+1)*L); a bucket
// spanning [s, s+cnt) is touched by lanes s/L .. (s+cnt-1)/L, one fragment each.
__global__ void __launch_bounds__(kBlock) k_fix_starts(u32* __restrict__ starts, const u32* __restrict__ blocksum,
                                                       const u32* __restrict__ counts, u32* __restrict__ cursor,
                                                       u32* __restrict__ fragcnt, u32 len, u32 L) {
  u32 i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= len) return;
  u32 s = starts[i] + blocksum[i / kScanTile];
  starts[i] = s;
  cursor[i] = s;
  u32 cnt = counts[i];
  fragcnt[i] = cnt ? ((s + cnt - 1) / L - s / L + 1) : 0;
}

// Fix-up after the scan of the fragment counts; buckets with more than
// `max_small` fragments are queued for the block-per-bucket pre-reduction.
__global__ void __launch_bounds__(kBlock) k_fix_foff(u32* __restrict__ foff, const u32* __restrict__ blocksum,
                                                     const u32* __restrict__ fragcnt, u32* __restrict__ large,
                                                     u32* __restrict__ nlarge, u32 len, u32 max_small, u32 max_large) {
  u32 i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= len) return;
  foff[i] += blocksum[i / kScanTile];
  if (fragcnt[i] > max_small) {
    u32 k = atomicAdd(nlarge, 1u);
    if (k < max_large) large[k] = i;
  }
}

// All of the above in ONE single-block launch for up to 32,768 slots, with the slots read ONCE, coalesced (16 bytes per
// lane), and the two block scans by wave shuffles (round 5): a tile of 4,096 slots per step, two steps' loads in flight.
// (k_scan_fused, the round-2 single-block form that gave every thread 8..64 consecutive slots and read them twice, four
// bytes at a time -- 0.14 ms for 32,768 slots against 0.036 for the six launches -- was removed in round 6.)
static constexpr u32 kScanOneMax = 32768;
static constexpr int kScanOnePer = kScanOneMax / kScanThreads;  // 32 consecutive slots per thread at most
// floor(x / L) for x < 2^32 by one 64 x 64 -> high-64 product with M = floor((2^64 - 1) / L) + 1 (exact: the error term
// x / 2^64 is far below 1 / L).  A hardware-less 32-bit division is ~40 instructions, and one block walks every slot.
__device__ __forceinline__ u32 div_by(u32 x, u32 m_lo, u32 m_hi) {
  return (u32)(((u64)__umulhi(x, m_lo) + (u64)x * m_hi) >> 32);
}
__device__ __forceinline__ u32 block_exclusive_scan_shfl(u32 v, u32* sh /* [16] */, u32& total) {
  const u32 lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
  u32 inc = v;
#pragma unroll
  for (u32 off = 1; off < 64; off <<= 1) {
    const u32 t = __shfl_up(inc, off, 64);
    if (lane >= off) inc += t;
  }
  __syncthreads();  // sh may still be read from the previous call
  if (lane == 63) sh[wv] = inc;
  __syncthreads();
  u32 before = 0, all = 0;
#pragma unroll
  for (u32 k = 0; k < kScanThreads / 64; k++) {
    const u32 t = sh[k];
    if (k < wv) before += t;
    all += t;
  }
  total = all;
  return before + inc - v;
}
// Thread t owns slots [t per, (t + 1) per), per a multiple of 4: every load is issued before anything is
// added (k_scan_fused walks its slots one dependent 4-byte load after the other, twice), two block scans in all.
__global__ void __launch_bounds__(kScanThreads)
    k_scan_one(const u32* __restrict__ counts, u32 nb, u32* __restrict__ starts, u32* __restrict__ cursor,
               u32* __restrict__ fragcnt, u32* __restrict__ foff, u32* __restrict__ large, u32* __restrict__ nlarge, u32 L,
               u32 max_small, u32 max_large, u32 m_lo, u32 m_hi) {
  __shared__ u32 sh[kScanThreads / 64];
  __shared__ u32 sh_nl;
  const u32 tid = threadIdx.x;
  if (tid == 0) sh_nl = 0;
  const u32 per = (((nb + kScanThreads - 1) / kScanThreads) + 3u) & ~3u;  // <= kScanOnePer (checked by the launcher)
  const u32 lo = tid * per;
  const bool vec = (nb & 3u) == 0;  // then every group of four slots is whole and 16-byte aligned
  u32 v[kScanOnePer], f[kScanOnePer];
#pragma unroll
  for (int g = 0; g < kScanOnePer / 4; g++) {
    const u32 at = lo + 4 * g;
    if (4u * g < per && at < nb) {
      if (vec) {
        const uint4 q = *reinterpret_cast<const uint4*>(counts + at);
        v[4 * g] = q.x; v[4 * g + 1] = q.y; v[4 * g + 2] = q.z; v[4 * g + 3] = q.w;
      } else {
#pragma unroll
        for (int k = 0; k < 4; k++) v[4 * g + k] = at + k < nb ? counts[at + k] : 0u;
      }
    } else {
      v[4 * g] = v[4 * g + 1] = v[4 * g + 2] = v[4 * g + 3] = 0;
    }
  }
  u32 sum = 0;
#pragma unroll
  for (int k = 0; k < kScanOnePer; k++) sum += v[k];
  u32 total, ftotal;
  const u32 first = block_exclusive_scan_shfl(sum, sh, total);
  // fragments per slot from the slot's start: lanes (start + cnt - 1) / L - start / L + 1; the end of one slot is
  // the start of the next, so one division per slot and one for the inclusive end
  u32 run = first, fsum = 0;
#pragma unroll
  for (int k = 0; k < kScanOnePer; k++) {
    const u32 cnt = v[k];
    f[k] = cnt ? (div_by(run + cnt - 1, m_lo, m_hi) - div_by(run, m_lo, m_hi) + 1u) : 0u;
    fsum += f[k];
    run += cnt;
  }
  u32 frun = block_exclusive_scan_shfl(fsum, sh, ftotal);
  run = first;
#pragma unroll
  for (int g = 0; g < kScanOnePer / 4; g++) {
    const u32 at = lo + 4 * g;
    u32 st[4], fo[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      st[k] = run;
      fo[k] = frun;
      run += v[4 * g + k];
      frun += f[4 * g + k];
      if (f[4 * g + k] > max_small) {
        const u32 q = atomicAdd(&sh_nl, 1u);
        if (q < max_large) large[q] = at + k;
      }
    }
    if (4u * g < per && at < nb) {
      if (vec) {
        *reinterpret_cast<uint4*>(starts + at) = make_uint4(st[0], st[1], st[2], st[3]);
        *reinterpret_cast<uint4*>(cursor + at) = make_uint4(st[0], st[1], st[2], st[3]);
        *reinterpret_cast<uint4*>(fragcnt + at) = make_uint4(f[4 * g], f[4 * g + 1], f[4 * g + 2], f[4 * g + 3]);
        *reinterpret_cast<uint4*>(foff + at) = make_uint4(fo[0], fo[1], fo[2], fo[3]);
      } else {
#pragma unroll
        for (int k = 0; k < 4; k++)
          if (at + k < nb) {
            starts[at + k] = st[k];
            cursor[at + k] = st[k];
            fragcnt[at + k] = f[4 * g + k];
            foff[at + k] = fo[k];
          }
      }
    }
  }
  __syncthreads();
  if (tid == 0) {
    starts[nb] = total;
    foff[nb] = ftotal;
    *nlarge = sh_nl;
  }
}

// The same in ONE launch for ANY number of slots (round 5): blocks of 256 threads take tiles of 4,096 slots and hand their
// sums down a chain -- the single-pass scan with decoupled look-back, twice in one kernel, because a slot's fragment count
// needs its global start: (1) tile sums of the counts -> starts, (2) tile sums of the fragment counts -> offsets.  What the
// six launches of the multi-block form cost a synchronous mid-size call is not their GPU time (5 us each) but the host's
// ~8 us per launch: at 2^16 pairs (40,960 slots, beyond k_scan_one's one block) the scan section of the timeline was
// launch-bound, 0.10 ms from k_hist's end to k_scatter's start (rocprofv3, gpurun_out/r5_t16).  64 registers, four waves:
// unlike k_scan_one a block of this fits beside two accumulate waves of a SIMD, so pipelined calls take it too.
//   chain[c][t] (c = 0: counts, c = 1: fragments; t = tile): epoch << 34 | state << 32 | value, state 1 = the tile's own sum,
//   2 = the sum of all tiles up to and including t.  The EPOCH (a per-slot launch counter from the host, never 0) makes
//   words of earlier launches read as "not there yet", so nothing is cleared between calls.  Tiles are taken in TICKET order
//   (one atomic per block on a counter that only ever grows; the host passes the count it stood at before this launch), so
//   a block only ever waits for blocks that have started: every wait ends, whatever order the hardware dispatches in.
//   A wait that does not end after ~2^20 polls all the same (it cannot, by the argument above) gives up: it raises *host_err
//   (pinned host memory, read by finish_slot) and goes on with a prefix of 0 -- positions stay inside the arrays.
static constexpr u32 kChainThreads = 256;
static constexpr int kChainPer = 16;
static constexpr u32 kChainTile = kChainThreads * kChainPer;
static constexpr u32 kChainMaxTiles = 1024;  // 4,194,304 slots: what one pass of a batch may hold (msm_api.hip kMaxSlotsPerPass)
static constexpr u32 kChainSpinLimit = 1u << 20;
__device__ __forceinline__ u64 chain_word(u32 epoch, u32 state, u32 value) {
  return ((u64)epoch << 34) | ((u64)state << 32) | (u64)value;
}
__device__ __forceinline__ u32 wave_sum_u32(u32 v) {
#pragma unroll
  for (u32 off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
// Called by the 64 lanes of ONE wave: publishes tile `my`'s sum `agg` and returns the sum of all tiles before it.
__device__ __forceinline__ u32 chain_lookback(u64* __restrict__ words, u32 my, u32 epoch, u32 agg, u32* __restrict__ host_err) {
  const u32 lane = threadIdx.x & 63u;
  if (my == 0) {
    if (lane == 0) __hip_atomic_store(&words[0], chain_word(epoch, 2u, agg), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    return 0;
  }
  if (lane == 0) __hip_atomic_store(&words[my], chain_word(epoch, 1u, agg), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  u32 prefix = 0, spins = 0;
  int j = (int)my - 1;  // the nearest tile not summed yet
  while (true) {
    const int idx = j - (int)lane;
    // (tiles before the first count as one with prefix 0)
    const u64 w = idx >= 0 ? __hip_atomic_load(&words[idx], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) : chain_word(epoch, 2u, 0u);
    const u32 state = (u32)(w >> 34) == epoch ? (u32)(w >> 32) & 3u : 0u;
    const u64 missing = __ballot(state == 0u);
    const u32 usable = missing ? (u32)__builtin_ctzll(missing) : 64u;  // lanes [0, usable) hold sums
    const u64 below = usable >= 64u ? ~(u64)0 : (((u64)1 << usable) - 1u);
    const u64 full = __ballot(state == 2u) & below;
    if (full) {
      const u32 stop = (u32)__builtin_ctzll(full);  // the nearest tile that knows its whole prefix
      prefix += wave_sum_u32(lane <= stop ? (u32)w : 0u);
      break;
    }
    prefix += wave_sum_u32(lane < usable ? (u32)w : 0u);
    j -= (int)usable;
    if (usable == 0u) {
      if (++spins > kChainSpinLimit) {
        if (lane == 0) *host_err = 1u;
        prefix = 0;
        break;
      }
      __builtin_amdgcn_s_sleep(4);
    }
  }
  if (lane == 0) __hip_atomic_store(&words[my], chain_word(epoch, 2u, prefix + agg), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  return prefix;
}
__device__ __forceinline__ u32 block_exclusive_scan_256(u32 v, u32* sh /* [4] */, u32& total) {
  const u32 lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
  u32 inc = v;
#pragma unroll
  for (u32 off = 1; off < 64; off <<= 1) {
    const u32 t = __shfl_up(inc, off, 64);
    if (lane >= off) inc += t;
  }
  __syncthreads();  // sh may still be read from the previous call
  if (lane == 63) sh[wv] = inc;
  __syncthreads();
  u32 before = 0, all = 0;
#pragma unroll
  for (u32 k = 0; k < kChainThreads / 64; k++) {
    const u32 t = sh[k];
    if (k < wv) before += t;
    all += t;
  }
  total = all;
  return before + inc - v;
}
struct ScanChain {
  unsigned long long* words;  // [2][kChainMaxTiles]
  u32* ticket;
  u32* host_err;
  u32 ticket_base, epoch, ntiles;
};
__global__ void __launch_bounds__(kChainThreads)
    k_scan_chain(const u32* __restrict__ counts, u32 nb, u32* __restrict__ starts, u32* __restrict__ cursor,
                 u32* __restrict__ fragcnt, u32* __restrict__ foff, u32* __restrict__ large, u32* __restrict__ nlarge, u32 L,
                 u32 max_small, u32 max_large, u32 m_lo, u32 m_hi, ScanChain ch, u32 prio) {
  set_wave_prio(prio);
  __shared__ u32 sh[kChainThreads / 64];
  __shared__ u32 sh_my, sh_pre[2];
  const u32 tid = threadIdx.x;
  if (tid == 0) sh_my = atomicAdd(ch.ticket, 1u) - ch.ticket_base;
  __syncthreads();
  const u32 my = sh_my;
  if (my >= ch.ntiles) {  // block-uniform; a launch has exactly ntiles blocks, so this is a host count that lags the device's:
    if (tid == 0) *ch.host_err = 1u;  // the call must fail, not return sums over unwritten starts (review of round 5)
    return;
  }
  u64* words0 = reinterpret_cast<u64*>(ch.words);
  u64* words1 = words0 + kChainMaxTiles;
  const u32 lo = my * kChainTile + tid * kChainPer;
  const bool vec = (nb & 3u) == 0;
  u32 v[kChainPer], f[kChainPer];
#pragma unroll
  for (int g = 0; g < kChainPer / 4; g++) {
    const u32 at = lo + 4 * g;
    if (at < nb) {
      if (vec) {
        const uint4 q = *reinterpret_cast<const uint4*>(counts + at);
        v[4 * g] = q.x; v[4 * g + 1] = q.y; v[4 * g + 2] = q.z; v[4 * g + 3] = q.w;
      } else {
#pragma unroll
        for (int k = 0; k < 4; k++) v[4 * g + k] = at + k < nb ? counts[at + k] : 0u;
      }
    } else {
      v[4 * g] = v[4 * g + 1] = v[4 * g + 2] = v[4 * g + 3] = 0;
    }
  }
  u32 sum = 0;
#pragma unroll
  for (int k = 0; k < kChainPer; k++) sum += v[k];
  u32 agg, fagg;
  const u32 first = block_exclusive_scan_256(sum, sh, agg);
  if (tid < 64) {
    // the first tile clears the queue of large buckets BEFORE it publishes: every other tile appends after its look-backs,
    // which (transitively) follow that publication
    if (my == 0 && tid == 0) __hip_atomic_store(nlarge, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const u32 pre = chain_lookback(words0, my, ch.epoch, agg, ch.host_err);
    if (tid == 0) sh_pre[0] = pre;
  }
  __syncthreads();
  const u32 pre0 = sh_pre[0];
  u32 run = pre0 + first, fsum = 0;
#pragma unroll
  for (int k = 0; k < kChainPer; k++) {
    const u32 cnt = v[k];
    f[k] = cnt ? (div_by(run + cnt - 1, m_lo, m_hi) - div_by(run, m_lo, m_hi) + 1u) : 0u;
    fsum += f[k];
    run += cnt;
  }
  const u32 ffirst = block_exclusive_scan_256(fsum, sh, fagg);
  if (tid < 64) {
    const u32 pre = chain_lookback(words1, my, ch.epoch, fagg, ch.host_err);
    if (tid == 0) sh_pre[1] = pre;
  }
  __syncthreads();
  const u32 pre1 = sh_pre[1];
  run = pre0 + first;
  u32 frun = pre1 + ffirst;
#pragma unroll
  for (int g = 0; g < kChainPer / 4; g++) {
    const u32 at = lo + 4 * g;
    u32 st[4], fo[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      st[k] = run;
      fo[k] = frun;
      run += v[4 * g + k];
      frun += f[4 * g + k];
      if (f[4 * g + k] > max_small) {
        const u32 q = atomicAdd(nlarge, 1u);
        if (q < max_large) large[q] = at + k;
      }
    }
    if (at < nb) {
      if (vec) {
        *reinterpret_cast<uint4*>(starts + at) = make_uint4(st[0], st[1], st[2], st[3]);
        *reinterpret_cast<uint4*>(cursor + at) = make_uint4(st[0], st[1], st[2], st[3]);
        *reinterpret_cast<uint4*>(fragcnt + at) = make_uint4(f[4 * g], f[4 * g + 1], f[4 * g + 2], f[4 * g + 3]);
        *reinterpret_cast<uint4*>(foff + at) = make_uint4(fo[0], fo[1], fo[2], fo[3]);
      } else {
#pragma unroll
        for (int k = 0; k < 4; k++)
          if (at + k < nb) {
            starts[at + k] = st[k];
            cursor[at + k] = st[k];
            fragcnt[at + k] = f[4 * g + k];
            foff[at + k] = fo[k];
          }
      }
    }
  }
  if (my == ch.ntiles - 1 && tid == 0) {
    starts[nb] = pre0 + agg;
    foff[nb] = pre1 + fagg;
  }
}

// Phase 0: gnark points (R = 2^384, saturated limbs) -> internal form (fp28.h),
// once per MSM: one Montgomery product per coordinate.  (0,0) stays (0,0).
// element i of the internal point array (kA28Bytes apart: one 128-byte line per point, so a
// gathered point never straddles two lines)
__device__ __forceinline__ A28* a28_at(A28* base, size_t i) {
  return reinterpret_cast<A28*>(reinterpret_cast<char*>(base) + i * kA28Bytes);
}
__device__ __forceinline__ const A28* a28_at(const A28* base, size_t i) {
  return reinterpret_cast<const A28*>(reinterpret_cast<const char*>(base) + i * kA28Bytes);
}

// 128 points per block.  Every lane converts one base; the two 128-byte records it produces go through
// LDS so that the block writes its 32 KiB of output as whole lines, 1 KiB per wave instruction (written
// straight from the lanes, a store instruction touched 64 lines, 16 bytes of each).  Chunk k of lane t
// sits at chunk k ^ (t & 15) of the lane's 256 bytes: 4-way bank conflicts on the way in, none on the
// way out.
static constexpr int kCvtBlock = 128;
// (block `bid`; threads beyond kCvtBlock of a larger block only keep the barrier company: k_front)
__device__ __forceinline__ void convert_body(const uint4* __restrict__ points, u32 n, A28* __restrict__ out, const u32 bid) {
  __shared__ uint4 stage[kCvtBlock * 16];
  const u32 tid = threadIdx.x;
  const u32 base = bid * kCvtBlock;
  const u32 i = base + tid;
  if (tid < (u32)kCvtBlock && i < n) {
    u32 w[24];
    d28::load_words<24>(w, points + (size_t)i * 6);
    A28 a;
    d28::from_gnark_iso_x(a.x, w);  // no product: the MSM runs on an isomorphic curve (fp28.h)
    d28::from_gnark_iso_y(a.y, w + 12);
    uint4* mine = stage + tid * 16;
    const u32 sw = tid & 15u;
    const u32* aw = reinterpret_cast<const u32*>(&a);
#pragma unroll
    for (u32 k = 0; k < 7; k++) mine[k ^ sw] = make_uint4(aw[4 * k], aw[4 * k + 1], aw[4 * k + 2], aw[4 * k + 3]);
    mine[7u ^ sw] = make_uint4(0, 0, 0, 0);
    // ... and phi(P) = (beta x, y) as the record behind it (GLV, k_digits); infinity (0, 0) stays (0, 0)
    F28 beta;
#pragma unroll
    for (int j = 0; j < d28::N; j++) beta.l[j] = d28::kBeta(j);
    d28::mul(a.x, a.x, beta);
#pragma unroll
    for (u32 k = 0; k < 7; k++) mine[(8u + k) ^ sw] = make_uint4(aw[4 * k], aw[4 * k + 1], aw[4 * k + 2], aw[4 * k + 3]);
    mine[15u ^ sw] = make_uint4(0, 0, 0, 0);
  }
  __syncthreads();
  const u32 cnt = min((u32)kCvtBlock, n - base) * 16u;  // 16-byte chunks this block owns
  uint4* dst = reinterpret_cast<uint4*>(a28_at(out, 2 * (size_t)base));
  for (u32 c = tid; c < cnt; c += blockDim.x) {
    const u32 t = c >> 4, k = c & 15u;
    dst[c] = stage[t * 16 + (k ^ (t & 15u))];
  }
}
__global__ void __launch_bounds__(kCvtBlock)
    k_convert_points(const uint4* __restrict__ points, u32 n, A28* __restrict__ out, u32 prio) {
  set_wave_prio(prio);
  convert_body(points, n, out, blockIdx.x);
}

// Conversion and recoding of a SMALL call in one launch: they read different inputs and nothing of each other,
// and a small synchronous call is bound by the host's launches until the accumulation starts (nine launches at
// ~8 us each against ~5 us kernels: rocprofv3 timeline of a 1,268-pair call, gpurun_out/r5i).  Blocks [0, nconv)
// convert 128 points each, the rest recode.
template <bool GLV>
__global__ void __launch_bounds__(kBlock)
    k_front(const uint4* __restrict__ points, u32 npts, A28* __restrict__ out28, u32 nconv, const uint4* __restrict__ scalars,
            MsmPlan p, u32* __restrict__ digits, u32* __restrict__ counts, u32 nb) {
  if (blockIdx.x < nconv) {  // block-uniform
    convert_body(points, npts, out28, blockIdx.x);
    return;
  }
  const CoarseOut none = {nullptr, nullptr, nullptr};
  digits_body<GLV, false>(scalars, p, digits, counts, nb, none, blockIdx.x - nconv, gridDim.x - nconv);
}

// The device accumulator's front of a SMALL call in one launch (round 5): the conversion of the loose bases, the slot
// scalars (dacc_eval.h) and their recoding -- four operations on the stream until now (k_convert_points, a device-to-
// device copy of the loose scalars, k_dacc_scalars, k_digits), each a launch the host pays ~8 us for while the GPU
// waits (the front of a verification is bound by the host's launches: rocprofv3 timeline, gpurun_out/r5w).  Blocks
// [0, nconv) convert 128 loose points each into the records behind the resident ones; the others take 256 pairs each:
// pair i < n_res is a resident slot, whose scalar is evaluated here; the pairs behind them are the loose bases with their
// scalars from the job.  The scalars are written out as well when the caller wants them (tests; scalars_out).
struct DaccFrontArgs {
  const curdle_dacc_check* checks;
  const uint4* pool;
  const uint4* extra_points;   // n_extra gnark points
  const uint4* extra_scalars;  // n_extra fr.Elements (Montgomery)
  uint4* scalars_out;          // [n_res + n_extra] or null
  A28* extra_out28;            // where the loose bases' records go (2 per point)
  u32 n_checks, pool_len, n_crs, n_inst, n_extra, nconv, staged;
};
template <bool GLV>
__global__ void __launch_bounds__(kBlock)
    k_dacc_front(DaccFrontArgs a, MsmPlan p, u32* __restrict__ digits, u32* __restrict__ counts, u32 nb) {
  extern __shared__ uint4 lds_stage[];
  if (blockIdx.x < a.nconv) {  // block-uniform
    convert_body(a.extra_points, a.n_extra, a.extra_out28, blockIdx.x);
    return;
  }
  const u32 bid = blockIdx.x - a.nconv, nblocks = gridDim.x - a.nconv;
  const u32 tid = threadIdx.x;
  for (u32 b = bid * kBlock + tid; b < nb; b += nblocks * kBlock) counts[b] = 0;  // as k_digits does
  const dacc::View vw = dacc::setup(lds_stage, a.checks, a.n_checks, a.pool, a.pool_len, a.staged != 0, tid, kBlock);
  const u32 n_res = a.n_crs + a.n_inst;
  const u32 i = bid * kBlock + tid;
  if (i >= p.n / 2) return;
  Fr m;  // the pair's scalar, Montgomery form
  if (i < n_res)
    m = dacc::eval_slot(vw, i, a.n_crs);
  else
    m = dacc::load_fr(a.extra_scalars, i - n_res);
  if (a.scalars_out) dacc::store_fr(a.scalars_out, i, m);
  Fr s;
  f_from_mont<FrParams>(s, m);
  u32 none = 0;
  recode_scalar<GLV, false>(s, i, p, digits, &none);
}

// Sum of `acc` over aligned groups of G lanes (G a power of two <= 256) of a
// 256-thread block, by wave shuffles; the result is valid in the first lane of each
// group.  No LDS tile (a 56 KiB one would keep the next MSM's LDS-histogram blocks
// off the CU while this latency-bound kernel runs): only 4 x 224 B when G = 256.
__device__ __forceinline__ void shfl_down_x28(X28& dst, const X28& src, u32 off) {
  const u32* s = reinterpret_cast<const u32*>(&src);
  u32* d = reinterpret_cast<u32*>(&dst);
#pragma unroll
  for (int i = 0; i < 56; i++) d[i] = __shfl_down(s[i], off, 64);
}
__device__ __forceinline__ void group_sum(X28& acc, u32 G, X28* wave_partials /* LDS, 4 entries */) {
  const u32 tid = threadIdx.x;
  const u32 lane = tid & 63u;
  const u32 gw = G < 64u ? G : 64u;
  X28 b;
  for (u32 off = gw / 2; off > 0; off >>= 1) {
    shfl_down_x28(b, acc, off);
    if ((lane & (gw - 1)) >= off) d28::set_inf(b);  // lanes outside the live half contribute nothing
    d28::add(acc, b);
  }
  if (G > 64u) {  // G = 128 or 256: combine the waves' results
    if (lane == 0) wave_partials[tid >> 6] = acc;
    __syncthreads();
    if ((tid & (G - 1)) == 0) {
      for (u32 k = 1; k < G / 64u; k++) {
        b = wave_partials[(tid >> 6) + k];
        d28::add(acc, b);
      }
    }
  }
}

// Balanced bucket accumulation.  Lane t owns L consecutive positions of the
// bucket-sorted point list, whatever buckets they belong to, so every lane of
// every wave does the same number of mixed additions however skewed the scalars
// are (all-equal scalars, short top window).  When the list moves on to the
// next bucket the lane stores its running sum as a fragment of the finished
// bucket and starts again from infinity.  Fragments of one bucket are
// contiguous: slot = foff[bucket] + (t - start[bucket] / L).
#ifdef CURDLE_TRACE_WAVES
// Experiment build only: start / end (100 MHz wall clock) and hardware id of every wave of the LAST accumulate launch.
__device__ unsigned long long g_wave_trace[4 * 8192];
__device__ unsigned long long g_wave_clk[2 * 8192];  // s_memtime (shader clock) at the same two points
hipError_t debug_read_wave_trace(unsigned long long* out, size_t words) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wave_trace), words * 8 < sizeof(g_wave_trace) ? words * 8 : sizeof(g_wave_trace));
}
hipError_t debug_read_wave_clk(unsigned long long* out, size_t words) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wave_clk), words * 8 < sizeof(g_wave_clk) ? words * 8 : sizeof(g_wave_clk));
}
#endif
template <int WAVES>
__global__ void __launch_bounds__(kBlock, WAVES)
    k_accumulate(const A28* __restrict__ points, const u32* __restrict__ sorted, const u32* __restrict__ starts,
                 const u32* __restrict__ foff, X28* __restrict__ frags, u32 nb, u32 L, u32 set_points, u32 frag_stride,
                 u32 prio_shift) {
  const u32 t = blockIdx.x * kBlock + threadIdx.x;
  // base set blockIdx.y of a shared-scalar call: its own points and fragments, the one sorted list
  points = a28_at(points, (size_t)blockIdx.y * set_points);
  frags += (size_t)blockIdx.y * frag_stride;
  const u32 total = starts[nb];
  u32 pos = t * L;
#ifdef CURDLE_TRACE_WAVES
  const u32 wv = t >> 6;
  if ((t & 63u) == 0 && wv < 8192u) {
    g_wave_trace[4 * wv] = wall_clock64();
    g_wave_clk[2 * wv] = __builtin_amdgcn_s_memtime();
    g_wave_trace[4 * wv + 2] = __builtin_amdgcn_s_getreg(kGetregHwId);
    g_wave_trace[4 * wv + 3] = __builtin_amdgcn_s_getreg(kGetregXccId);
  }
#endif
  if (pos >= total) return;
  const u32 end = min(pos + L, total);
  // bucket containing `pos`: first index with starts[idx] > pos, minus one
  u32 lo = 0, hi = nb;
  while (lo < hi) {
    u32 mid = (lo + hi) >> 1;
    if (starts[mid] > pos) hi = mid;
    else lo = mid + 1;
  }
  u32 g = lo - 1;
  u32 gend = starts[lo];
  X28 acc;
  d28::set_inf(acc);
  // The lane's indices are read eight at a time into a register queue: between two of its
  // iterations the XCD's other lanes gather megabytes of points through the 4 MiB L2, so a
  // 4-byte read per iteration fetched a whole 128-byte line of `sorted` from memory every time
  // (2.1 GB of the launch's 4.3 GB, profiles/r02_fetch_calibration.txt); eight reads issued
  // back to back share one line fetch.  The gather of position pos + 1 is issued before the
  // addition of position pos.
  u32 q[8];
  auto refill = [&](u32 from) {
#pragma unroll
    for (int j = 0; j < 8; j++) q[j] = from + j < end ? sorted[from + j] : 0u;
  };
  refill(pos);
  u32 queued = 0;
  u32 e_next = q[0];
  A28 pt_next;
  d28::load(pt_next, a28_at(points, e_next & 0x7fffffffu));
  const u32 slot = __builtin_amdgcn_s_getreg(kGetregHwId) & 1u;  // this wave's slot on its SIMD, low bit
  for (; pos < end; pos++) {
    if (prio_shift) {
      if ((((u32)wall_clock64() >> prio_shift) ^ slot) & 1u)
        __builtin_amdgcn_s_setprio(3);
      else
        __builtin_amdgcn_s_setprio(0);
    }
    const u32 e = e_next;
    A28 pt = pt_next;
    if (++queued == 8) {
      refill(pos + 1);
      queued = 0;
    } else {
#pragma unroll
      for (int j = 0; j < 7; j++) q[j] = q[j + 1];
    }
    if (pos + 1 < end) {
      e_next = q[0];
      d28::load(pt_next, a28_at(points, e_next & 0x7fffffffu));
    }
    if (pos == gend) {
      d28::store(&frags[foff[g] + (t - starts[g] / L)], acc);
      d28::set_inf(acc);
      g++;
      gend = starts[g + 1];
      if (gend == pos) {
        // An EMPTY bucket.  Uniform scalars leave next to none, skewed ones leave runs of thousands (all-equal scalars: two
        // occupied buckets per window; a hot window: one) and a lane that walked such a run one dependent load at a time held
        // its whole wave back -- 32,768 loads at N = 2^20: the launch took 4.0 ms against 2.25 (profiles/r06_adversarial.json).
        // Bisect for the first slot that starts beyond pos, as at the lane's start.
        u32 l2 = g + 2, h2 = nb;
        while (l2 < h2) {
          const u32 mid = (l2 + h2) >> 1;
          if (starts[mid] > pos) h2 = mid;
          else l2 = mid + 1;
        }
        g = l2 - 1;
        gend = starts[l2];
      }
    }
    if (d28::affine_is_inf(pt)) continue;  // (0,0) = infinity (curdleproof.go:23)
    if (e >> 31) {
      F28 z;
      d28::set_zero(z);
      d28::sub_raw<4>(pt.y, z, pt.y);  // 4p - y
    }
    d28::madd<true>(acc, pt.x, pt.y);
  }
  d28::store(&frags[foff[g] + (t - starts[g] / L)], acc);
#ifdef CURDLE_TRACE_WAVES
  if ((t & 63u) == 0 && wv < 8192u) {
    g_wave_trace[4 * wv + 1] = wall_clock64();
    g_wave_clk[2 * wv + 1] = __builtin_amdgcn_s_memtime();
  }
#endif
}

// (k_merge_large: behind group_sum_quad, below)

// ---------------------------------------------------------------------------
// The latency-bound kernels (quad28.h): four adjacent lanes own ONE point between them
// (X | Y | ZZ | ZZZ), so a 256-thread block carries 64 logical lanes ("quads") and a point
// addition is 4 product steps instead of 14.  (A one-lane-per-segment build of the bucket
// reduction existed until round 2: 256 VGPRs with 76 spilled, and slower wherever it was
// measured against quads with the right segment length -- N = 2^20 pipelined 3.09 vs 3.04
// ms, 1,024 x 628-pair batch 8.09 vs 7.67 ms, profiles/r02_quad_everywhere.txt.)
// ---------------------------------------------------------------------------
// Sum over aligned groups of G quads (G a power of two <= 64); valid in the first quad of
// each group.  wave_partials: LDS, [4 waves][4 coordinates].
__device__ __forceinline__ void group_sum_quad(F28& acc, u32 G, F28 (*wave_partials)[4]) {
  const u32 tid = threadIdx.x;
  const u32 ll = (tid & 63u) >> 2;       // quad inside the wave, 0..15
  const u32 gw = G < 16u ? G : 16u;
  F28 b;
  for (u32 off = gw / 2; off > 0; off >>= 1) {
    q28::shfl_down(b, acc, off);
    if ((ll & (gw - 1)) >= off) q28::set_inf(b);  // quads outside the live half contribute nothing
    q28::add(acc, b);
  }
  if (G > 16u) {  // G = 32 or 64: combine the waves' results
    if ((tid & 63u) < 4u) wave_partials[tid >> 6][tid & 3u] = acc;
    __syncthreads();
    const u32 lt = tid >> 2;             // quad inside the block, 0..63
    if ((lt & (G - 1)) == 0) {           // whole quads take this branch
      for (u32 k = 1; k < G / 16u; k++) {
        b = wave_partials[(tid >> 6) + k][tid & 3u];
        q28::add(acc, b);
      }
    }
  }
}

// Buckets with more than max_small fragments (queued by the scan kernels: skewed scalars -- all-equal ones, a few
// distinct values, one hot window; uniform scalars queue nothing and every block leaves at once): their fragments
// are summed into the bucket's FIRST fragment slot, and the reduce kernels read ONE fragment for such a bucket
// (fragcnt itself stays as scanned: base sets of a shared-scalar call share it).
// Round 6.  Until now: one block per queued bucket, every thread a chain of m / 256 whole-point additions (14 product
// steps each) and a 9-level tree behind it -- all-equal scalars at N = 2^20 queue 16 buckets of 8,192 fragments: 64
// waves on a chip of 1,024 SIMDs for 0.9 ms, a third of the accumulation (profiles/r06_adversarial.json, "before").
// Now the unit of work is a CHUNK of 16 S consecutive fragments of one bucket, taken by one wave as 16 quads
// (quad28.h: an addition is 4 product steps): a quad adds S fragments, the wave's 16 sums meet by shuffles, and the
// chunk's sum replaces the chunk's first fragment.  Chunks of all queued buckets are numbered through (a prefix sum
// over the queue, redone by every block in LDS: the queue is short) and dealt to the launch's waves round robin, so
// 16 buckets of 8,192 fragments are 1,024 waves at once.  The wave that finishes a bucket's LAST chunk (a counter per
// queue entry, left at zero for the next call) adds the chunk sums the same way into slot 0.  S grows with the
// bucket (2 up to 1,024 fragments, 4 up to 4,096, 8 up to 16,384, 16 beyond) so that the two serial parts stay
// balanced: a chain of 4 + 4 + 2 + 4 quad additions for 2,048 fragments, 8 + 4 + 4 + 4 for 8,192, instead of
// 32 + 9 whole ones.
static constexpr u32 kMergeTile = 1024;  // queue entries per pass over the queue
__host__ __device__ inline u32 merge_chunk_shift(u32 m) {  // log2 of the chunk a bucket of m fragments is cut into
  return m <= 1024u ? 5u : m <= 4096u ? 6u : m <= 16384u ? 7u : 8u;
}
__global__ void __launch_bounds__(kBlock, 2)
    k_merge_large(const u32* __restrict__ large, const u32* __restrict__ nlarge, const u32* __restrict__ foff,
                  const u32* __restrict__ fragcnt, X28* __restrict__ frags, u32* __restrict__ done, u32 max_large,
                  u32 frag_stride, u32 prio) {
  __shared__ u32 pre[kMergeTile], gq[kMergeTile], mq[kMergeTile];
  __shared__ u32 sh_scan[kBlock / 64];
  const u32 nl = min(*nlarge, max_large);
  if (nl == 0) return;  // the usual case
  set_wave_prio(prio);
  const u32 tid = threadIdx.x;
  const u32 lane = tid & 63u, qd = lane >> 2;  // quad inside the wave
  frags += (size_t)blockIdx.y * frag_stride;
  done += (size_t)blockIdx.y * max_large;
  const u32 nwaves = gridDim.x * (kBlock / 64), mywave = blockIdx.x * (kBlock / 64) + (tid >> 6);
  // More chunks than the launch has waves (hundreds of buckets of a hundred fragments each: 64 distinct scalar values)
  // would go round several times at the small buckets' chunk size: every bucket's chunks are doubled (up to 256
  // fragments) until one round takes them all -- 5,120 chunks of 32 on 1,024 waves took 0.30 ms, 2,560 of 64 on 3,072 take 0.1.
  u32 bump = 0;
  {
    u32 mine = 0;
    for (u32 x = tid; x < nl; x += kBlock) {
      const u32 m = fragcnt[large[x]], sh = merge_chunk_shift(m);
      mine += (m + (1u << sh) - 1u) >> sh;
    }
    u32 total0;
    (void)block_exclusive_scan_256(mine, sh_scan, total0);
    while (bump < 3u && (total0 >> bump) > nwaves) bump++;
  }
  auto chunk_shift = [&](u32 m) { return min(merge_chunk_shift(m) + bump, 8u); };
  for (u32 t0 = 0; t0 < nl; t0 += kMergeTile) {  // block-uniform trip count
    const u32 cnt = min(kMergeTile, nl - t0);
    __syncthreads();  // the pass before is done with the tables
    u32 nch[kMergeTile / kBlock], sum = 0;
#pragma unroll
    for (u32 k = 0; k < kMergeTile / kBlock; k++) {
      const u32 idx = tid * (kMergeTile / kBlock) + k;
      nch[k] = 0;
      if (idx < cnt) {
        const u32 g = large[t0 + idx], m = fragcnt[g];
        gq[idx] = g;
        mq[idx] = m;
        const u32 sh = chunk_shift(m);
        nch[k] = (m + (1u << sh) - 1u) >> sh;
      }
      sum += nch[k];
    }
    u32 total;
    u32 ex = block_exclusive_scan_256(sum, sh_scan, total);
#pragma unroll
    for (u32 k = 0; k < kMergeTile / kBlock; k++) {
      pre[tid * (kMergeTile / kBlock) + k] = ex;
      ex += nch[k];
    }
    __syncthreads();
    for (u32 c = mywave; c < total; c += nwaves) {  // wave-uniform from here on
      u32 lo = 0, hi = cnt;  // the queue entry of chunk c: the last one whose first chunk is <= c
      while (lo < hi) {
        const u32 mid = (lo + hi) >> 1;
        if (pre[mid] > c) hi = mid;
        else lo = mid + 1;
      }
      const u32 e = lo - 1;
      const u32 m = mq[e], sh = chunk_shift(m), base = (c - pre[e]) << sh;
      const u32 nchunks = (m + (1u << sh) - 1u) >> sh;
      X28* f = frags + foff[gq[e]];
      const u32 lim = min(1u << sh, m - base);
      F28 acc, b, nxt;
      q28::set_inf(acc);
      u32 i = qd;  // this quad's fragments: base + qd, + 16, ...; the next one is loaded under the addition before it
      q28::load(nxt, &f[base + (i < lim ? i : 0u)]);
      while (i < lim) {  // quad-uniform
        b = nxt;
        i += 16;
        q28::load(nxt, &f[base + (i < lim ? i : 0u)]);
        q28::add(acc, b);
      }
      group_sum_quad(acc, 16u, nullptr);  // shuffles only
      if (qd == 0) q28::store(&f[base], acc);
      if (nchunks == 1) continue;
      // the chunk sums of one bucket are written by waves anywhere on the chip: released here, acquired by the wave
      // that counts the last one (agent scope: the XCDs' L2s are written back / invalidated by the fences)
      __threadfence();
      u32 t = 0;
      if (lane == 0) t = atomicAdd(&done[t0 + e], 1u);
      t = __shfl(t, 0, 64);
      if (t != nchunks - 1) continue;
      __threadfence();
      q28::set_inf(acc);
      i = qd;
      q28::load(nxt, &f[(size_t)(i < nchunks ? i : 0u) << sh]);
      while (i < nchunks) {
        b = nxt;
        i += 16;
        q28::load(nxt, &f[(size_t)(i < nchunks ? i : 0u) << sh]);
        q28::add(acc, b);
      }
      group_sum_quad(acc, 16u, nullptr);
      if (qd == 0) q28::store(&f[0], acc);
      if (lane == 0) done[t0 + e] = 0;  // as the next call expects it
    }
  }
}

// One quad per segment of `seg` consecutive buckets.  The running-sum recurrence is run as
// ONE addition per step -- the next fragment of the current bucket into the running sum,
// or, when the bucket is exhausted, the running sum into the segment sum -- so quads whose
// buckets have different fragment counts do not wait for each other bucket by bucket, and
// the kernel has a single copy of the addition in its main loop.
__global__ void __launch_bounds__(kBlock, 2)
    k_bucket_reduce_quad(FragSources src, X28* __restrict__ partials, MsmPlan p) {
  __shared__ F28 sh[4][4];
  const u32 tid = threadIdx.x;
  const u32 q = blockIdx.x * (kBlock / 4) + (tid >> 2);  // logical lane
  const bool live = q < p.kr * p.NS;
  F28 acc, run, b;
  q28::set_inf(acc);
  q28::set_inf(run);
  if (live) {
    const u32 jr = q / p.NS;            // result index = set * k + j
    const u32 r = q - jr * p.NS;
    const u32 set = jr / p.k, j = jr - set * p.k;
    const size_t set_off = (size_t)set * p.frag_stride;  // base sets of a shared-scalar call: one source only
    int w = p.win_begin;
    while (r >= (p.base[w] + p.nbkt[w]) / p.seg) w++;
    const u32 lo = (r - p.base[w] / p.seg) * p.seg;
    const u32 g0 = j * p.NB + p.base[w] + lo;
    int u = (int)p.seg - 1;
    // the bucket's fragments: those of source 0, then source 1, ... (one source but for chunked calls)
    u32 s = 0, m = 0, k = 0;
    const X28* f = nullptr;
    auto open = [&]() {
      m = src.fragcnt[s][g0 + u];
      if (m > p.max_small) m = 1;  // pre-merged by k_merge_large into its first slot
      f = reinterpret_cast<const X28*>(src.frags[s]) + set_off + src.foff[s][g0 + u];
      k = 0;
    };
    open();
    while (u >= 0) {
      while (k >= m && s + 1 < src.n) {  // this source has nothing (more) for the bucket: the next one
        s++;
        open();
      }
      const bool take = k < m;  // uniform over the quad
      if (take) {
        q28::load(b, &f[k]);
        k++;
      }
      F28 dst, from;
      q28::sel(dst, take, run, acc);
      q28::sel(from, take, b, run);
      q28::add(dst, from);
      q28::sel(run, take, dst, run);
      q28::sel(acc, take, acc, dst);
      if (!take) {
        u--;
        s = 0;
        if (u >= 0) open();
      }
    }
    // lo * (segment total): every quad of the window runs the same number of steps
    const int top = 31 - __clz((int)(p.nbkt[w] | 1u));
    q28::mul_small(b, run, lo, top);
    q28::add(acc, b);
  }
  if (p.G > 1) group_sum_quad(acc, p.G, sh);
  if (((tid >> 2) & (p.G - 1)) == 0 && live) q28::store(&partials[q / p.G], acc);
}

// ---------------------------------------------------------------------------
// The bucket reduction WITHOUT a scalar multiple (single MSMs; MsmPlan::reduce_bits).
//
// k_bucket_reduce_quad above gives every quad its segment's lo * (segment total) by a 15-bit
// double-and-add: 105 product steps of the ~290 in a quad's chain at N = 2^20, as much work again
// as the running sums, and a 6-level tree plus a window-sum launch behind it (0.42 ms
// for 5 % of the accumulation's additions).  Here nothing is multiplied on the GPU.  With segment
// totals T_j and segment running sums S_j (j the segment's index inside its window),
//
//     sum_b (b + 1) B_b  =  sum_j S_j  +  seg * sum_j j T_j ,      sum_j j T_j = sum_i 2^i X_i ,
//     X_i = sum of T_j over the j whose bit i is set,
//
// and ALL the X_i fall out of ONE butterfly over the T_j at the cost of a plain tree sum: at the level
// with offset o every quad whose index has bit o clear adds the value of the quad o further up; at
// the end the quad with index 0 holds the total and the quad with index 2^i holds X_i.  The quads
// whose bit o is SET are idle in that level, so they carry the plain sum of the S_j towards the
// group's last quad in the same instruction stream: one addition per level for both trees.
// A window leaves the GPU as <= 14 points with bit positions (sum S at 0, X_i at log2(seg) + i) and
// the host's Horner pass over the windows -- 127 doublings it runs anyway -- takes them in like
// window sums: ~12 additions per window at ~0.3 us each, where the GPU pays ~5 us per dependent
// addition.  Two launches: segments -> groups of <= 16 quads (one wave, shuffles only), groups ->
// the window's points (k_reduce_groups).
// ---------------------------------------------------------------------------
// S: plain sum over aligned groups of G quads (G a power of two <= 16), result in the group's LAST
// quad.  T: the butterfly, total in the group's first quad, X_i in the quad with index 2^i.
__device__ __forceinline__ void group_bits_and_sum(F28& S, F28& T, u32 G) {
  const u32 idx = ((threadIdx.x & 63u) >> 2) & (G - 1);
  for (u32 o = 1; o < G; o <<= 1) {
    F28 up, dn, a, b;
    q28::shfl_down(up, T, o);
    q28::shfl_up(dn, S, o);
    const bool is_t = (idx & o) == 0;
    const bool is_s = (idx & (2 * o - 1)) == 2 * o - 1;
    q28::sel(a, is_t, T, S);
    q28::sel(b, is_t, up, dn);
    if (!is_t && !is_s) q28::set_inf(b);  // quad-uniform
    q28::add(a, b);
    q28::sel(T, is_t, a, T);
    q28::sel(S, is_t, S, a);
  }
}

// Host-buffer MSMs accumulated in chunks (msm_api.hip run_host_chunked): a chunk's fragments are folded into
// ONE running sum per bucket as soon as its accumulation is done -- while later chunks are still crossing
// PCIe or being accumulated -- so that the call's single reduction, which is what remains after the last copy
// has landed, walks one point per bucket for all the earlier chunks instead of their 1.5 fragments each.
// One quad per bucket slot; meta[b] = b (the sums' "fragment offset"), meta[nb + b] = 1 once any chunk had a
// fragment there: the sums are a fragment source like any other (FragSources).
__global__ void __launch_bounds__(kBlock, 2)
    k_fold_fragments(const X28* __restrict__ frags, const u32* __restrict__ foff, const u32* __restrict__ fragcnt,
                     X28* __restrict__ sums, u32* __restrict__ meta, u32 nb, u32 max_small, u32 first, u32 prio) {
  set_wave_prio(prio);
  // (a 128-register build at priority 3, so that a wave fits beside the next chunk's two accumulate waves, and the same
  // for the usually empty k_merge_large launch in front of it: measured, no better -- profiles/r04_host_fold.txt)
  const u32 b = blockIdx.x * (kBlock / 4) + (threadIdx.x >> 2);
  if (b >= nb) return;  // whole quads leave together
  u32 m = fragcnt[b];
  if (m > max_small) m = 1;  // pre-merged by k_merge_large into its first slot
  const X28* f = frags + foff[b];
  F28 acc, x;
  u32 any = m ? 1u : 0u;
  if (first) {
    q28::set_inf(acc);
  } else {
    q28::load(acc, &sums[b]);
    any |= meta[nb + b];
  }
  for (u32 k = 0; k < m; k++) {
    q28::load(x, &f[k]);
    q28::add(acc, x);
  }
  q28::store(&sums[b], acc);
  if (q28::role() == 0) {
    meta[b] = b;
    meta[nb + b] = any;
  }
}

// One quad per segment of `seg` consecutive buckets, as in k_bucket_reduce_quad: ONE addition per
// step, the next fragment into the running sum or the running sum into the segment sum.  The
// fragment a step adds was loaded during the step before it, and a bucket's bookkeeping one bucket
// ahead (source 0): the loads are off the chain.  Output per group of G quads: 2 + log2(G) points,
// [sum S | total T | X_0 .. X_(lgG-1)].
__global__ void __launch_bounds__(kBlock, 2)
    k_reduce_segments(FragSources src, X28* __restrict__ groups, MsmPlan p) {
  set_wave_prio(p.reduce_prio);
  const u32 tid = threadIdx.x;
  const u32 q = blockIdx.x * (kBlock / 4) + (tid >> 2);  // logical lane
  const bool live = q < p.kr * p.NS;
  F28 acc, run;
  q28::set_inf(acc);
  q28::set_inf(run);
  if (live) {
    const u32 jr = q / p.NS;            // result index = set * k + j
    const u32 r = q - jr * p.NS;
    const u32 set = jr / p.k, j = jr - set * p.k;
    const size_t set_off = (size_t)set * p.frag_stride;
    int w = p.win_begin;
    while (r >= (p.base[w] + p.nbkt[w]) / p.seg) w++;
    const u32 lo = (r - p.base[w] / p.seg) * p.seg;
    const u32 g0 = j * p.NB + p.base[w] + lo;
    int u = (int)p.seg - 1;
    u32 s = 0, m = 0, k = 0;
    const X28* f = nullptr;
    // Round 5 (late): NO load of this loop sits under a branch.  The loop used to fetch the next fragment, and the next
    // bucket's bookkeeping, inside its "take a fragment" / "close the bucket" branches; a load under a branch lands in
    // registers of its own, and the copy into the loop-carried registers -- placed where the branch ends -- waited for it on
    // the spot (s_waitcnt vmcnt(0) a dozen instructions behind the global_load: the ISA of round 4's kernel), so every step of
    // the chain paid a memory round trip on top of its addition: ~9 us per step where the addition is ~5.7.  Now every lane
    // issues the same loads at the same place in every step -- what the NEXT step needs, or any valid record if it needs
    // nothing -- and they are copied into the loop's registers after the addition.
    const X28* const safe = reinterpret_cast<const X28*>(src.frags[0]);  // a valid record for steps with nothing to fetch
    auto set_bucket = [&](u32 cnt, u32 fo, u32 source) {  // the fragments of bucket u in `source`
      m = cnt > p.max_small ? 1u : cnt;                   // (beyond max_small: pre-merged by k_merge_large into its first slot)
      f = reinterpret_cast<const X28*>(src.frags[source]) + set_off + fo;
      k = 0;
    };
    auto more_sources = [&]() {  // bucket u's fragments in the next source that has any (chunked host-buffer calls only)
      while (k >= m && s + 1 < src.n) {
        s++;
        set_bucket(src.fragcnt[s][g0 + u], src.foff[s][g0 + u], s);
      }
    };
    set_bucket(src.fragcnt[0][g0 + u], src.foff[0][g0 + u], 0);
    more_sources();
    // source 0's bookkeeping of the bucket BELOW the current one, fetched in every step for the step after it
    u32 cm, cf;
    {
      const u32 un = u > 0 ? (u32)u - 1u : 0u;
      cm = src.fragcnt[0][g0 + un];
      cf = src.foff[0][g0 + un];
    }
    F28 nxt;
    q28::load(nxt, k < m ? &f[k] : safe);
    while (u >= 0) {
      const bool take = k < m;  // uniform over the quad
      const F28 b = nxt;
      if (take) {
        k++;
        more_sources();
      } else {
        u--;
        s = 0;
        if (u >= 0) {
          set_bucket(cm, cf, 0);
          more_sources();
        }
      }
      // the next step's fragment and the bookkeeping of the bucket below the (possibly new) current one: issued here by
      // every lane, consumed one addition later
      const u32 un = u > 0 ? (u32)u - 1u : 0u;
      const u32 lm = src.fragcnt[0][g0 + un], lf = src.foff[0][g0 + un];
      F28 ld;
      q28::load(ld, (u >= 0 && k < m) ? &f[k] : safe);
      F28 dst, from;
      q28::sel(dst, take, run, acc);
      q28::sel(from, take, b, run);
      q28::add(dst, from);
      q28::sel(run, take, dst, run);
      q28::sel(acc, take, acc, dst);
      nxt = ld;
      cm = lm;
      cf = lf;
    }
  }
  if (p.G > 1) group_bits_and_sum(acc, run, p.G);
  if (!live) return;  // groups are live or dead as a whole
  const u32 idx = q & (p.G - 1);
  X28* out = groups + (size_t)(q / p.G) * (2 + p.lgG);
  if (idx == p.G - 1) q28::store(&out[0], acc);
  if (idx == 0)
    q28::store(&out[1], run);
  else if ((idx & (idx - 1)) == 0)
    q28::store(&out[2 + (31 - __clz((int)idx))], run);
}

// A window's output point `slot`: which bit position it carries, relative to the window's shift;
// -1 if the window has no such point (msm_kernels.h).  Slot 0 is the sum of the segment sums, slot 1
// the total of all buckets (bookkeeping of the levels, of no use to the host), slot 2 + i is X_i.
__host__ __device__ inline int reduce_slot_position(const MsmPlan& p, int w, u32 slot) {
  if (slot == 0) return 0;
  if (slot == 1) return -1;
  u32 lgn = 0;  // log2 of the window's segments
  while ((p.seg << (lgn + 1)) <= p.nbkt[w]) lgn++;
  return slot - 2 < lgn ? (int)(p.lg_seg + slot - 2) : -1;
}
int reduce_bits_position(const MsmPlan& p, int w, uint32_t slot) { return reduce_slot_position(p, w, slot); }

// This lane's coordinate of a point, gnark form, to the host's array.
__device__ __forceinline__ void write_point_quad(const F28& c, G1XYZZ* dst) {
  u32 w12[12];
  d28::to_gnark_msm(w12, c, q28::role());
  // three 16-byte stores: the array may be the host's pinned buffer, where every store instruction is a write over PCIe
  uint4* d = reinterpret_cast<uint4*>(reinterpret_cast<u32*>(dst) + 12u * q28::role());
#pragma unroll
  for (int i = 0; i < 3; i++) d[i] = make_uint4(w12[4 * i], w12[4 * i + 1], w12[4 * i + 2], w12[4 * i + 3]);
}

// One LEVEL above k_reduce_segments: up to 128 consecutive groups of a window -> one group, the same
// record with log2(128) more bits: [sum S | total A | X_0 .. ].  Block (window lw, result jr and
// block index, quantity z) of 64 quads; a quad takes TWO consecutive input groups.  z = 0 and z >= 2
// are plain sums of that slot.  z = 1 weighs the totals A_e by the group index e, as bit sums again:
// the lowest new bit is the plain sum of the odd groups' totals (it rides on the S side of the dual
// tree), the pairs' totals go through the butterfly.  One wave reduces its 16 quads by shuffles;
// the waves' six results (S in the last quad, T in quads 0 1 2 4 8) meet in LDS, where 6 x nwv <= 24
// quads of waves 0 and 1 reduce them in groups of nwv quads the same way.  Every wave of a block
// sits on a SIMD of its own (one block of 512 threads per window and quantity took 0.087 ms at
// N = 2^20 where this takes half: two waves on a SIMD share its multiplier).
// A window with more than 128 groups takes a second pass (ng_shift = 7: the groups that are left);
// the last pass writes the host's array (gnark form) instead of records.
struct ReduceLevel {
  u32 ng_shift;   // input groups of window w: nbkt[w] / (seg * G) >> ng_shift (at least 1)
  u32 P_in;       // points per input record
  u32 P_out;      // points per output record (last pass: nout)
  u32 nblk_max;   // blocks per window and result in blockIdx.y
  u32 in_stride;  // input records per result
  u32 out_stride; // output records per result (not used by the last pass)
  u32 last;       // 1: write the host's array
};
__global__ void __launch_bounds__(kBlock, 2)
    k_reduce_level(const X28* __restrict__ in, X28* __restrict__ out, G1XYZZ* __restrict__ host_out, MsmPlan p, ReduceLevel lv) {
  set_wave_prio(p.reduce_prio);
  __shared__ F28 sh[4][6][4];
  const u32 lw = blockIdx.x, jr = blockIdx.y / lv.nblk_max, blk = blockIdx.y - jr * lv.nblk_max, z = blockIdx.z;
  const int w = p.win_begin + (int)lw;
  const u32 nw = p.win_end - p.win_begin;
  const u32 per = p.seg * p.G;
  u32 ng = (p.nbkt[w] / per) >> lv.ng_shift;  // this window's input groups (a power of two)
  if (ng == 0) ng = 1;
  const u32 B = ng < 128u ? ng : 128u;         // input groups per block
  if (blk * B >= ng) return;                   // block-uniform
  const u32 Qa = B >= 2u ? B / 2u : 1u;        // quads at work
  const u32 tid = threadIdx.x, q = tid >> 2, wave = tid >> 6, ql = (tid & 63u) >> 2;
  // records of window w start where the windows before it end: every window's count is shifted alike
  u32 first_in = 0, first_out = 0;
  for (int x = p.win_begin; x < w; x++) {
    u32 g = (p.nbkt[x] / per) >> lv.ng_shift;
    if (g == 0) g = 1;
    first_in += g;
    first_out += g <= 128u ? 1u : g / 128u;
  }
  const X28* base = in + ((size_t)jr * lv.in_stride + first_in + (size_t)blk * B) * lv.P_in;
  X28* orec = out + ((size_t)jr * lv.out_stride + first_out + blk) * lv.P_out;
  G1XYZZ* hrec = host_out + ((size_t)jr * nw + lw) * lv.P_out;
  auto emit = [&](const F28& v, u32 slot) {  // the calling quad's point -> slot of the output record
    if (lv.last)
      write_point_quad(v, &hrec[slot]);
    else
      q28::store(&orec[slot], v);
  };
  F28 S, T, b;
  q28::set_inf(S);
  q28::set_inf(T);
  if (B == 1) {  // nothing to reduce: the record moves on (the host's array gets infinity for the total)
    if (z >= lv.P_in) return;
    if (q == 0) {
      q28::load(S, base + z);
      if (lv.last && z == 1) q28::set_inf(S);
      emit(S, z);
    }
    if (z == 1 && q >= lv.P_in && q < lv.P_out) emit(S, q);  // S is infinity here: q != 0
    return;
  }
  if (q < Qa) {
    const X28* e = base + (size_t)(2 * q) * lv.P_in + z;
    if (z != 1) {
      q28::load(S, e);
      q28::load(b, e + lv.P_in);
      q28::add(S, b);
    } else {
      q28::load(T, e);
      q28::load(S, e + lv.P_in);  // the odd group: bit 0 of the group index
      q28::add(T, S);
    }
  }
  const u32 G1 = Qa < 16u ? Qa : 16u;
  u32 lg1 = 0;
  while ((1u << lg1) < G1) lg1++;
  if (G1 > 1) group_bits_and_sum(S, T, G1);
  const u32 nwv = Qa > 16u ? Qa / 16u : 1u;  // 1, 2 or 4 waves hold groups
  u32 lgw = 0;
  while ((1u << lgw) < nwv) lgw++;
  // z = 1: the new bits in order: odd groups | quad bits inside a wave | wave bits
  const u32 bit0 = lv.P_in;
  if (nwv == 1) {
    if (wave == 0) {
      if (ql == G1 - 1) emit(S, z == 1 ? bit0 : z);
      if (z == 1) {
        if (ql == 0) {
          if (lv.last) q28::set_inf(T);
          emit(T, 1);
        } else if (ql < G1 && (ql & (ql - 1)) == 0) {
          emit(T, bit0 + 1 + (31 - __clz((int)ql)));
        }
      }
    }
  } else {
    if (wave < nwv) {
      if (ql == 15) sh[wave][0][tid & 3u] = S;
      if (ql == 0) sh[wave][1][tid & 3u] = T;
      if (ql == 1) sh[wave][2][tid & 3u] = T;
      if (ql == 2) sh[wave][3][tid & 3u] = T;
      if (ql == 4) sh[wave][4][tid & 3u] = T;
      if (ql == 8) sh[wave][5][tid & 3u] = T;
    }
    __syncthreads();
    const u32 slot2 = wave * 16u + ql;    // quad slot of the second stage: quantity c, wave v
    const u32 c = slot2 / nwv, v = slot2 - c * nwv;
    if (wave < 2) {
      q28::set_inf(S);
      q28::set_inf(T);
      if (c < 6) {
        if (c == 1)
          T = sh[v][1][tid & 3u];
        else
          S = sh[v][c][tid & 3u];
      }
      group_bits_and_sum(S, T, nwv);
      if (c == 0) {
        if (v == nwv - 1) emit(S, z == 1 ? bit0 : z);
      } else if (z == 1 && c < 6) {
        if (c == 1) {
          if (v == 0) {
            if (lv.last) q28::set_inf(T);
            emit(T, 1);
          } else if ((v & (v - 1)) == 0) {
            emit(T, bit0 + 1 + 4 + (31 - __clz((int)v)));
          }
        } else if (v == nwv - 1) {
          emit(S, bit0 + 1 + (c - 2));
        }
      }
    }
  }
  // the bit slots this block's window does not fill hold infinity (z = 1's block owns them)
  const u32 made = 1 + lg1 + lgw;  // new bits of this level
  if (z == 1 && q >= bit0 + made && q < lv.P_out) {
    q28::set_inf(S);
    emit(S, q);
  }
}

// Window sums from the group partials.  A window owns nseg / G consecutive
// partials.  Wide windows (many partials): one 64-quad block per (window, MSM)
// with an LDS tree.  A call whose windows all have <= 4 partials (batches of small
// MSMs): one lane per window, so tens of thousands of windows fill the chip.
// With the host combine (single MSMs, small batches) the window sums are written in
// gnark form (canonical XYZZ coordinates); a large batch keeps them in internal form
// for k_combine.
// The quad's own coordinate of a window sum: gnark form for the host combine, internal form
// for k_combine.
__device__ __forceinline__ void write_window_sum_quad(const F28& c, G1XYZZ* winsums, X28* winsums28, const MsmPlan& p,
                                                      u32 j, u32 lw) {
  const u32 nw = p.win_end - p.win_begin;
  if (!p.gpu_combine) {
    u32 w12[12];
    d28::to_gnark_msm(w12, c, q28::role());
    uint4* dst = reinterpret_cast<uint4*>(reinterpret_cast<u32*>(&winsums[(size_t)j * nw + lw]) + 12u * q28::role());
#pragma unroll
    for (int i = 0; i < 3; i++) dst[i] = make_uint4(w12[4 * i], w12[4 * i + 1], w12[4 * i + 2], w12[4 * i + 3]);
  } else {
    q28::store(&winsums28[(size_t)j * nw + lw], c);
  }
}

__global__ void __launch_bounds__(kBlock, 2)
    k_window_sum_wide_quad(const X28* __restrict__ partials, G1XYZZ* __restrict__ winsums, X28* __restrict__ winsums28,
                           MsmPlan p) {
  __shared__ F28 sh[4][4];
  const u32 lw = blockIdx.x, j = blockIdx.y;
  const u32 w = p.win_begin + lw;
  const u32 tid = threadIdx.x;
  const u32 lt = tid >> 2;  // logical lane 0..63
  const u32 np = p.nbkt[w] / p.seg / p.G;
  const X28* pw = partials + ((size_t)j * p.NS + p.base[w] / p.seg) / p.G;
  F28 acc, b;
  q28::set_inf(acc);
  for (u32 k = lt; k < np; k += 64) {
    q28::load(b, &pw[k]);
    q28::add(acc, b);
  }
  group_sum_quad(acc, 64, sh);
  if (tid < 4) write_window_sum_quad(acc, winsums, winsums28, p, j, lw);
}

__global__ void __launch_bounds__(kBlock, 2)
    k_window_sum_flat(const X28* __restrict__ partials, G1XYZZ* __restrict__ winsums, X28* __restrict__ winsums28,
                      MsmPlan p) {
  // one quad per window (an addition is 4 product steps against a single lane's 14)
  const u32 nw = p.win_end - p.win_begin;
  const u32 gw = (blockIdx.x * kBlock + threadIdx.x) >> 2;
  if (gw >= p.kr * nw) return;  // whole quads leave together
  const u32 j = gw / nw, lw = gw - j * nw;
  const u32 w = p.win_begin + lw;
  const u32 np = p.nbkt[w] / p.seg / p.G;
  const X28* pw = partials + ((size_t)j * p.NS + p.base[w] / p.seg) / p.G;
  F28 acc, b;
  q28::load(acc, &pw[0]);
  for (u32 k = 1; k < np; k++) {
    q28::load(b, &pw[k]);
    q28::add(acc, b);
  }
  write_window_sum_quad(acc, winsums, winsums28, p, j, lw);
}

// Batched calls: one quad per MSM runs the Horner pass over its window sums (what the host
// does for a single MSM) and the 2^shift scaling of a partial: ~255 doublings + one
// addition per window, 3 and 4 product steps each (quad28.h); with hundreds of MSMs in
// flight the serial chain is amortised over the batch.  The results leave as XYZZ in gnark
// form; the host normalises the whole batch with ONE shared inversion
// (curdle_host_batch_to_affine) -- a Fermat inversion here would be another 570-product
// serial chain per MSM (round 1: 3.7 ms for this kernel, 1.2 ms of it the inversion).
__global__ void __launch_bounds__(kBlock, 2)
    k_combine(const X28* __restrict__ winsums28, G1XYZZ* __restrict__ results, MsmPlan p) {
  const u32 j = blockIdx.x * (kBlock / 4) + (threadIdx.x >> 2);
  if (j >= p.kr) return;  // whole quads leave together
  const u32 nw = p.win_end - p.win_begin;
  F28 acc, b;
  q28::set_inf(acc);
  for (int lw = (int)nw - 1; lw >= 0; lw--) {
    q28::load(b, &winsums28[(size_t)j * nw + lw]);
    q28::add(acc, b);
    const int dbls = lw > 0 ? p.bits[p.win_begin + lw - 1] : p.shift[p.win_begin];
    if (!q28::is_inf(acc))
      for (int q = 0; q < dbls; q++) q28::dbl(acc);
  }
  u32 w12[12];
  d28::to_gnark_msm(w12, acc, q28::role());  // this lane's coordinate; ZZ = 0 (infinity) stays 0
  u32* dst = reinterpret_cast<u32*>(&results[j]) + 12u * q28::role();
#pragma unroll
  for (int i = 0; i < 12; i++) dst[i] = w12[i];
}

// ---------------------------------------------------------------------------
// Launchers
// ---------------------------------------------------------------------------
static inline u32 cdiv(u64 a, u32 b) { return (u32)((a + b - 1) / b); }

// LDS beyond the default 64 KiB needs an opt-in per kernel -- and per DEVICE (a process may drive
// several: curdle_init_devices), so it is made once for every device a launch comes from.
static hipError_t sort_lds_optin() {
  static std::atomic<uint32_t> done{0};
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  const uint32_t bit = 1u << (dev & 31);
  if (done.load(std::memory_order_acquire) & bit) return hipSuccess;
  e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_hist), hipFuncAttributeMaxDynamicSharedMemorySize, 32768 * 4);
  if (e == hipSuccess)
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_scatter), hipFuncAttributeMaxDynamicSharedMemorySize, 32768 * 4);
  if (e == hipSuccess)
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_scatter_coarse), hipFuncAttributeMaxDynamicSharedMemorySize,
                            kCoarseTile * 5);
  if (e == hipSuccess) done.fetch_or(bit, std::memory_order_release);
  return e;
}

// Layout of ws.ccur (two-level plans): [nw * 256] coarse cursors | [kCoarseReps][nw * 256] coarse counts (16-byte
// aligned: k_coarse_scan reads them four at a time) | [nw * 256 + 1] packed bin starts + sentinel.  The counts must be
// zero when k_digits starts and are zero again when it ends (msm_api.hip clears them when the buffer is made and after a
// failed call).
static inline CoarseOut coarse_out(const MsmPlan& p, const MsmWorkspace& ws) {
  const size_t nw = (size_t)(p.win_end - p.win_begin);
  CoarseOut co;
  co.ccur = ws.ccur;
  co.ccount = ws.ccur + nw * kCoarseMax;
  co.cstart = ws.ccur + (1 + (size_t)kCoarseReps) * nw * kCoarseMax;
  return co;
}
size_t coarse_words(uint32_t nw) { return (2 + (size_t)kCoarseReps) * nw * kCoarseMax + 1; }

// operand shapes the two passes assume (checked on the host): one MSM, every window a whole number of
// bins and at most kCoarseMax of them, term indices that fit 24 bits
static hipError_t two_level_shapes(const MsmPlan& p, const MsmWorkspace& ws) {
  const u32 nw = p.win_end - p.win_begin;
  if (p.k != 1 || p.n > (1u << 24) || !ws.tmp || !ws.ccur || nw > (u32)kCoarseWinMax) return hipErrorInvalidValue;
  for (int w = p.win_begin; w < p.win_end; w++)
    if ((p.nbkt[w] & kFineMask) || (p.nbkt[w] >> kFineBits) > (u32)kCoarseMax) return hipErrorInvalidValue;
  return hipSuccess;
}

hipError_t launch_digits(const MsmPlan& p, const MsmWorkspace& ws, const void* d_scalars, hipStream_t stream) {
  const u32 bs = p.two_level ? (u32)kDigitsCoarseBlock : (u32)kBlock;
  const u32 blocks = cdiv(p.n / 2, bs);
  const dim3 grid(p.two_level && blocks > (u32)kDigitsCoarseMaxBlocks ? (u32)kDigitsCoarseMaxBlocks : blocks), block(bs);
  const uint4* sc = reinterpret_cast<const uint4*>(d_scalars);
  CoarseOut co = {nullptr, nullptr, nullptr};
  if (p.two_level) {
    hipError_t e = two_level_shapes(p, ws);
    if (e != hipSuccess) return e;
    co = coarse_out(p, ws);
    if (p.glv)
      hipLaunchKernelGGL((k_digits<true, true>), grid, block, 0, stream, sc, p, ws.digits, ws.counts, p.k * p.NB, co);
    else
      hipLaunchKernelGGL((k_digits<false, true>), grid, block, 0, stream, sc, p, ws.digits, ws.counts, p.k * p.NB, co);
    hipLaunchKernelGGL(k_coarse_scan, dim3(1), dim3(kBlock), 0, stream, p, co);
  } else if (p.glv) {
    hipLaunchKernelGGL((k_digits<true, false>), grid, block, 0, stream, sc, p, ws.digits, ws.counts, p.k * p.NB, co);
  } else {
    hipLaunchKernelGGL((k_digits<false, false>), grid, block, 0, stream, sc, p, ws.digits, ws.counts, p.k * p.NB, co);
  }
  return hipGetLastError();
}

// Bucket sizes.  Two-level plans: the coarse partition (k_scatter_coarse, from the bins' positions k_digits left)
// and a counting walk over the bin-grouped array; everything else: the LDS histogram over the digits.
hipError_t launch_hist(const MsmPlan& p, const MsmWorkspace& ws, hipStream_t stream) {
  hipError_t e = sort_lds_optin();
  if (e != hipSuccess) return e;
  const u32 nw = p.win_end - p.win_begin;
  if (p.two_level) {
    if ((e = two_level_shapes(p, ws)) != hipSuccess) return e;
    u32 nbins = 0;
    for (int w = p.win_begin; w < p.win_end; w++) nbins += p.nbkt[w] >> kFineBits;
    const CoarseOut co = coarse_out(p, ws);
    hipLaunchKernelGGL(k_scatter_coarse, dim3(cdiv(p.n, kCoarseTile), nw), dim3(kCoarseThreads), kCoarseTile * 5, stream,
                       ws.digits, p, co.ccur, ws.tmp);
    hipLaunchKernelGGL(k_scatter_fine<true>, dim3(cdiv((u64)nw * p.n, kFineTile)), dim3(kFineThreads), 0, stream, ws.tmp, p,
                       co.cstart, nbins, ws.counts, ws.sorted);
    return hipGetLastError();
  }
  hipLaunchKernelGGL(k_hist, dim3(cdiv(p.n_max, p.chunk), nw, p.k), dim3(kSortThreads), p.max_nbkt * 4, stream,
                     ws.digits, p, ws.offsets, ws.counts);
  return hipGetLastError();
}

static hipError_t scan_u32(const u32* in, u32 len, u32* out, u32* blocksum, hipStream_t stream) {
  const u32 nblocks = cdiv(len, kScanTile);
  if (nblocks > (u32)kScanThreads) return hipErrorInvalidValue;
  hipLaunchKernelGGL(k_scan_local, dim3(nblocks), dim3(kScanThreads), 0, stream, in, len, out, blocksum);
  hipLaunchKernelGGL(k_scan_top, dim3(1), dim3(kScanThreads), 0, stream, blocksum, nblocks, out, len);
  return hipGetLastError();
}

uint32_t scan_chain_tiles(uint32_t nb) { return cdiv(nb, kChainTile); }
size_t scan_chain_bytes() { return (size_t)2 * kChainMaxTiles * 8 + 64; }  // the words, then the ticket counter

hipError_t launch_scan(const MsmPlan& p, const MsmWorkspace& ws, hipStream_t stream) {
  const u32 nb = p.k * p.NB;
  if (p.fuse_scan == 2 && nb <= kScanOneMax && p.L >= 2) {
    const u64 magic = ~(u64)0 / p.L + 1;  // div_by: floor(x / L) for 32-bit x
    hipLaunchKernelGGL(k_scan_one, dim3(1), dim3(kScanThreads), 0, stream, ws.counts, nb, ws.starts, ws.cursor,
                       ws.fragcnt, ws.foff, ws.large, ws.nlarge, p.L, p.max_small, p.max_large, (u32)magic, (u32)(magic >> 32));
    return hipGetLastError();
  }
  if (p.fuse_scan == 3) {
    // (the caller has counted this launch's tickets already: anything that cannot take the chain is an error, not another path)
    if (!ws.chain || !ws.chain_ticket || !ws.host_err || p.L < 2 || nb > kChainMaxTiles * kChainTile || ws.chain_epoch == 0) return hipErrorInvalidValue;
    const u64 magic = ~(u64)0 / p.L + 1;
    ScanChain ch;
    ch.words = ws.chain;
    ch.ticket = ws.chain_ticket;
    ch.host_err = ws.host_err;
    ch.ticket_base = ws.chain_base;
    ch.epoch = ws.chain_epoch;
    ch.ntiles = scan_chain_tiles(nb);
    hipLaunchKernelGGL(k_scan_chain, dim3(ch.ntiles), dim3(kChainThreads), 0, stream, ws.counts, nb, ws.starts, ws.cursor,
                       ws.fragcnt, ws.foff, ws.large, ws.nlarge, p.L, p.max_small, p.max_large, (u32)magic, (u32)(magic >> 32), ch,
                       p.aux_prio);
    return hipGetLastError();
  }
  hipError_t e0 = hipMemsetAsync(ws.nlarge, 0, 4, stream);
  if (e0 != hipSuccess) return e0;
  hipError_t e = scan_u32(ws.counts, nb, ws.starts, ws.blocksum, stream);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k_fix_starts, dim3(cdiv(nb, kBlock)), dim3(kBlock), 0, stream, ws.starts, ws.blocksum, ws.counts,
                     ws.cursor, ws.fragcnt, nb, p.L);
  e = scan_u32(ws.fragcnt, nb, ws.foff, ws.blocksum, stream);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k_fix_foff, dim3(cdiv(nb, kBlock)), dim3(kBlock), 0, stream, ws.foff, ws.blocksum, ws.fragcnt,
                     ws.large, ws.nlarge, nb, p.max_small, p.max_large);
  return hipGetLastError();
}

hipError_t launch_scatter(const MsmPlan& p, const MsmWorkspace& ws, hipStream_t stream) {
  hipError_t e = sort_lds_optin();
  if (e != hipSuccess) return e;
  const u32 nw = p.win_end - p.win_begin;
  if (p.two_level) {  // the entries are grouped by bin already (launch_hist): bins -> buckets
    if ((e = two_level_shapes(p, ws)) != hipSuccess) return e;
    u32 nbins = 0;
    for (int w = p.win_begin; w < p.win_end; w++) nbins += p.nbkt[w] >> kFineBits;
    hipLaunchKernelGGL(k_scatter_fine<false>, dim3(cdiv((u64)nw * p.n, kFineTile)), dim3(kFineThreads), 0, stream, ws.tmp, p,
                       coarse_out(p, ws).cstart, nbins, ws.cursor, ws.sorted);
    return hipGetLastError();
  }
  hipLaunchKernelGGL(k_scatter, dim3(cdiv(p.n_max, p.chunk), nw, p.k), dim3(kSortThreads), p.max_nbkt * 4, stream,
                     ws.digits, p, ws.offsets, ws.cursor, ws.sorted);
  return hipGetLastError();
}

hipError_t launch_front(const MsmPlan& p, const MsmWorkspace& ws, const void* d_points, uint32_t npts, const void* d_scalars,
                        hipStream_t stream) {
  if (p.two_level || npts == 0) return hipErrorInvalidValue;
  const u32 nconv = cdiv(npts, kCvtBlock), ndig = cdiv(p.n / 2, kBlock);
  const uint4* pts = reinterpret_cast<const uint4*>(d_points);
  const uint4* sc = reinterpret_cast<const uint4*>(d_scalars);
  if (p.glv)
    hipLaunchKernelGGL(k_front<true>, dim3(nconv + ndig), dim3(kBlock), 0, stream, pts, npts, reinterpret_cast<A28*>(ws.points28),
                       nconv, sc, p, ws.digits, ws.counts, p.k * p.NB);
  else
    hipLaunchKernelGGL(k_front<false>, dim3(nconv + ndig), dim3(kBlock), 0, stream, pts, npts, reinterpret_cast<A28*>(ws.points28),
                       nconv, sc, p, ws.digits, ws.counts, p.k * p.NB);
  return hipGetLastError();
}

static constexpr u32 kDaccFrontLds = 120 * 1024;
static hipError_t dacc_front_optin() {
  static std::atomic<uint32_t> done{0};
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  const uint32_t bit = 1u << (dev & 31);
  if (done.load(std::memory_order_acquire) & bit) return hipSuccess;
  e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_dacc_front<true>), hipFuncAttributeMaxDynamicSharedMemorySize, kDaccFrontLds);
  if (e == hipSuccess)
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_dacc_front<false>), hipFuncAttributeMaxDynamicSharedMemorySize, kDaccFrontLds);
  if (e == hipSuccess) done.fetch_or(bit, std::memory_order_release);
  return e;
}

hipError_t launch_dacc_front(const MsmPlan& p, const MsmWorkspace& ws, const DaccFront& f, hipStream_t stream) {
  // operand shapes: one MSM of n_crs + n_inst + n_extra pairs, not a two-level plan (its recoding also counts coarse bins)
  if (p.two_level || p.k != 1 || p.sets != 1 || p.n / 2 != f.n_crs + f.n_inst + f.n_extra) return hipErrorInvalidValue;
  hipError_t e = dacc_front_optin();
  if (e != hipSuccess) return e;
  DaccFrontArgs a;
  a.checks = reinterpret_cast<const curdle_dacc_check*>(f.d_checks);
  a.pool = reinterpret_cast<const uint4*>(f.d_pool);
  a.extra_points = reinterpret_cast<const uint4*>(f.d_extra_points);
  a.extra_scalars = reinterpret_cast<const uint4*>(f.d_extra_scalars);
  a.scalars_out = reinterpret_cast<uint4*>(f.d_scalars_out);
  a.extra_out28 = reinterpret_cast<A28*>(reinterpret_cast<char*>(ws.points28) + 2 * (size_t)(f.n_crs + f.n_inst) * kA28Bytes);
  a.n_checks = f.n_checks;
  a.pool_len = f.pool_len;
  a.n_crs = f.n_crs;
  a.n_inst = f.n_inst;
  a.n_extra = f.n_extra;
  a.nconv = cdiv(f.n_extra, kCvtBlock);
  const size_t need = dacc::lds_bytes(f.pool_len, f.n_checks, kDaccFrontLds - 36 * 1024);  // beside convert_body's 32 KiB stage
  a.staged = need ? 1u : 0u;
  const u32 ndig = cdiv(p.n / 2, kBlock);
  if (p.glv)
    hipLaunchKernelGGL(k_dacc_front<true>, dim3(a.nconv + ndig), dim3(kBlock), need, stream, a, p, ws.digits, ws.counts, p.k * p.NB);
  else
    hipLaunchKernelGGL(k_dacc_front<false>, dim3(a.nconv + ndig), dim3(kBlock), need, stream, a, p, ws.digits, ws.counts, p.k * p.NB);
  return hipGetLastError();
}

hipError_t launch_convert_points_raw(const void* d_points, uint32_t n, void* d_out28, hipStream_t stream, uint32_t prio) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(k_convert_points, dim3(cdiv(n, kCvtBlock)), dim3(kCvtBlock), 0, stream,
                     reinterpret_cast<const uint4*>(d_points), n, reinterpret_cast<A28*>(d_out28), prio);
  return hipGetLastError();
}

hipError_t launch_accumulate(const MsmPlan& p, const MsmWorkspace& ws, hipStream_t stream) {
  const u32 nw = p.win_end - p.win_begin;
  const u32 nb = p.k * p.NB;
  const u32 nlanes = cdiv((u64)nw * p.n, p.L);
  // two waves per SIMD (203 VGPRs, no spills): a three-wave build (168 VGPRs) spilled 26
  // registers and was slower
  hipLaunchKernelGGL(k_accumulate<2>, dim3(cdiv(nlanes, kBlock), p.sets), dim3(kBlock), 0, stream,
                     reinterpret_cast<const A28*>(ws.points28), ws.sorted, ws.starts, ws.foff,
                     reinterpret_cast<X28*>(ws.frags), nb, p.L, p.n, p.frag_stride, p.acc_prio);
  return hipGetLastError();
}

hipError_t launch_merge_large(const MsmPlan& p, const MsmWorkspace& ws, hipStream_t stream, bool wide) {
  // One wave per chunk of 32 to 256 fragments; more chunks than waves go round.  The launch is almost always empty
  // (uniform scalars queue nothing) and every block of it has to find room beside the next accumulation before it can
  // read the empty queue and leave: a synchronous call, which has the chip to itself, takes up to 768 blocks = three
  // 165-register waves on every SIMD; a pipelined one 256 (768 empty blocks cost its step 0.02 ms of 2.53:
  // profiles/r06_pipeline_phase_costs.txt).
  const u32 nw = p.win_end - p.win_begin;
  const u64 nlanes = ((u64)nw * p.n + p.L - 1) / p.L;  // fragments <= bucket slots + lanes
  const u64 chunks = (u64)p.max_large + ((u64)p.k * p.NB + nlanes) / 32u;
  const u32 cap = wide ? 768u : 256u;
  const u32 blocks = (u32)(chunks / 4u + 1u < cap ? chunks / 4u + 1u : cap);
  hipLaunchKernelGGL(k_merge_large, dim3(blocks, p.sets), dim3(kBlock), 0, stream, ws.large, ws.nlarge, ws.foff, ws.fragcnt,
                     reinterpret_cast<X28*>(ws.frags), ws.mdone, p.max_large, p.frag_stride, p.reduce_prio);
  return hipGetLastError();
}

hipError_t launch_bucket_reduce(const MsmPlan& p, const MsmWorkspace& ws, hipStream_t stream, const FragSources* extra) {
  const u64 lanes = (u64)p.kr * p.NS;  // quads
  FragSources src;
  memset(&src, 0, sizeof(src));
  if (extra) src = *extra;  // the earlier chunks first (any order gives the same bucket sums)
  if (src.n >= (u32)kMaxFragSources) return hipErrorInvalidValue;
  src.frags[src.n] = ws.frags;
  src.foff[src.n] = ws.foff;
  src.fragcnt[src.n] = ws.fragcnt;
  src.n++;
  hipLaunchKernelGGL(k_bucket_reduce_quad, dim3(cdiv(lanes, kBlock / 4)), dim3(kBlock), 0, stream, src,
                     reinterpret_cast<X28*>(ws.partials), p);
  return hipGetLastError();
}

hipError_t launch_reduce_segments(const MsmPlan& p, const MsmWorkspace& ws, hipStream_t stream, const FragSources* extra) {
  // operand shapes the kernels assume: whole groups of <= 16 quads, every window a whole number of groups
  if (!p.reduce_bits || p.G < 1 || p.G > 16 || (p.G & (p.G - 1)) || (1u << p.lgG) != p.G || (1u << p.lg_seg) != p.seg ||
      p.NS != p.NB / p.seg || p.NG * p.G != p.NS)
    return hipErrorInvalidValue;
  for (int w = p.win_begin; w < p.win_end; w++)
    if (p.nbkt[w] % (p.seg * p.G) || p.base[w] % (p.seg * p.G)) return hipErrorInvalidValue;
  const u64 lanes = (u64)p.kr * p.NS;  // quads
  FragSources src;
  memset(&src, 0, sizeof(src));
  if (extra) src = *extra;
  if (src.n >= (u32)kMaxFragSources) return hipErrorInvalidValue;
  src.frags[src.n] = ws.frags;
  src.foff[src.n] = ws.foff;
  src.fragcnt[src.n] = ws.fragcnt;
  src.n++;
  hipLaunchKernelGGL(k_reduce_segments, dim3(cdiv(lanes, kBlock / 4)), dim3(kBlock), 0, stream, src,
                     reinterpret_cast<X28*>(ws.partials), p);
  return hipGetLastError();
}

hipError_t launch_fold_fragments(const MsmPlan& p, const MsmWorkspace& ws, void* sums, void* meta, bool first, hipStream_t stream) {
  if (p.k != 1 || p.sets != 1) return hipErrorInvalidValue;  // one MSM, one base set: the chunks of a host-buffer call
  const u32 nb = p.NB;
  hipLaunchKernelGGL(k_fold_fragments, dim3(cdiv(nb, kBlock / 4)), dim3(kBlock), 0, stream,
                     reinterpret_cast<const X28*>(ws.frags), ws.foff, ws.fragcnt, reinterpret_cast<X28*>(sums),
                     reinterpret_cast<u32*>(meta), nb, p.max_small, first ? 1u : 0u, p.aux_prio);
  return hipGetLastError();
}

hipError_t launch_reduce_groups(const MsmPlan& p, const MsmWorkspace& ws, hipStream_t stream) {
  const u32 nw = p.win_end - p.win_begin;
  if (!p.reduce_bits || !p.NG) return hipErrorInvalidValue;
  // levels of up to 128 groups per block until every window is one record; records ping-pong between
  // the two halves of ws.partials (the first half holds k_reduce_segments' output)
  const u32 per = p.seg * p.G;
  X28* bufs[2] = {reinterpret_cast<X28*>(ws.partials), reinterpret_cast<X28*>(ws.partials) + (size_t)p.kr * p.NG * (2 + p.lgG) + 1};
  ReduceLevel lv;
  lv.ng_shift = 0;
  lv.P_in = 2 + p.lgG;
  lv.in_stride = p.NG;
  for (int pass = 0;; pass++) {
    u32 ng_max = 1, in_recs = 0, out_recs = 0;
    for (int w = p.win_begin; w < p.win_end; w++) {
      u32 g = (p.nbkt[w] / per) >> lv.ng_shift;
      if (g == 0) g = 1;
      if (g > ng_max) ng_max = g;
      in_recs += g;
      out_recs += g <= 128u ? 1u : g / 128u;
    }
    if (in_recs != lv.in_stride) return hipErrorInvalidValue;
    const u32 B = ng_max < 128u ? ng_max : 128u;
    u32 lgB = 0;
    while ((1u << lgB) < B) lgB++;
    lv.last = ng_max <= 128u ? 1u : 0u;
    lv.P_out = lv.P_in + lgB;
    lv.nblk_max = ng_max <= 128u ? 1u : ng_max / 128u;
    lv.out_stride = out_recs;
    if (lv.last && lv.P_out != p.nout) return hipErrorInvalidValue;
    if (lv.P_out > 64u) return hipErrorInvalidValue;  // the filler takes one quad per slot
    hipLaunchKernelGGL(k_reduce_level, dim3(nw, p.kr * lv.nblk_max, lv.P_in), dim3(kBlock), 0, stream, bufs[pass & 1],
                       bufs[(pass + 1) & 1], ws.winsums, p, lv);
    if (lv.last) break;
    lv.ng_shift += 7;
    lv.P_in = lv.P_out;
    lv.in_stride = out_recs;
    if (pass > 4) return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

hipError_t launch_window_sum(const MsmPlan& p, const MsmWorkspace& ws, hipStream_t stream) {
  const u32 nw = p.win_end - p.win_begin;
  if (p.max_nbkt / p.seg / p.G > 4)
    hipLaunchKernelGGL(k_window_sum_wide_quad, dim3(nw, p.kr), dim3(kBlock), 0, stream,
                       reinterpret_cast<const X28*>(ws.partials), ws.winsums, reinterpret_cast<X28*>(ws.winsums28), p);
  else
    hipLaunchKernelGGL(k_window_sum_flat, dim3(cdiv((u64)p.kr * nw, kBlock / 4)), dim3(kBlock), 0, stream,
                       reinterpret_cast<const X28*>(ws.partials), ws.winsums, reinterpret_cast<X28*>(ws.winsums28), p);
  return hipGetLastError();
}

hipError_t launch_combine(const MsmPlan& p, const MsmWorkspace& ws, hipStream_t stream) {
  hipLaunchKernelGGL(k_combine, dim3(cdiv(p.kr, kBlock / 4)), dim3(kBlock), 0, stream,
                     reinterpret_cast<const X28*>(ws.winsums28), ws.results, p);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// Synthetic bases of SURVEY.md section 8(d): P_i = P_0 + i*Q, affine, gnark
// layout.  table[j] = 2^j * Q (affine).  One lane per point: <= 27 mixed adds,
// then one Fermat inversion of ZZ*ZZZ to normalise.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock, 2)
    k_synth_walk(const G1Affine* __restrict__ table, G1Affine p0, u32 n, uint4* __restrict__ out) {
  u32 i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  X28 acc;
  if (g1_affine_is_inf(p0)) {
    d28::set_inf(acc);
  } else {
    d28::from_gnark(acc.x, p0.x.l);
    d28::from_gnark(acc.y, p0.y.l);
    d28::set_one(acc.zz);
    d28::set_one(acc.zzz);
  }
  for (int j = 0; j < 27; j++) {
    if ((i >> j) & 1u) {
      G1Affine t = table[j];
      F28 x, y;
      d28::from_gnark(x, t.x.l);
      d28::from_gnark(y, t.y.l);
      d28::madd(acc, x, y);
    }
  }
  u32 w[24];
  if (d28::is_inf(acc)) {
    for (int k = 0; k < 24; k++) w[k] = 0;
  } else {
    F28 t, inv, izz, izzz, x, y;
    d28::mul(t, acc.zz, acc.zzz);
    d28::set_one(inv);
    for (int b = 383; b >= 0; b--) {
      d28::sqr(inv, inv);
      if ((kPminus2[b >> 5] >> (b & 31)) & 1u) d28::mul(inv, inv, t);
    }
    d28::mul(izz, inv, acc.zzz);
    d28::mul(izzz, inv, acc.zz);
    d28::mul(x, acc.x, izz);
    d28::mul(y, acc.y, izzz);
    d28::to_gnark(w, x);
    d28::to_gnark(w + 12, y);
  }
  d28::store_words<24>(out + (size_t)i * 6, w);
}

hipError_t launch_synth_walk(const G1Affine* d_table, const G1Affine& p0, uint32_t n, void* d_out,
                             hipStream_t stream) {
  hipLaunchKernelGGL(k_synth_walk, dim3(cdiv(n, kBlock)), dim3(kBlock), 0, stream, d_table, p0, n,
                     reinterpret_cast<uint4*>(d_out));
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// Primitive self-test (curdle_selftest_op)
// ---------------------------------------------------------------------------
// All operands and results cross this kernel in gnark form; the operation itself
// runs in the internal radix-2^28 form the MSM kernels use.
// The widths come from kSelftestTable (msm_kernels.h) as arguments; a branch whose own layout does
// not match them returns without touching memory, and so does an unknown op.
__global__ void __launch_bounds__(kBlock, 2)
    k_selftest(int op, const u32* __restrict__ in, size_t n, u32* __restrict__ out, u32 in_w, u32 out_w) {
  size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (op >= 8 && op <= 10) {  // lane-distributed point operations (quad28.h): four lanes per element
    if (in_w != 96 || out_w != 48) return;
    i >>= 2;
    if (i >= n) return;  // whole quads leave together
    G1XYZZ ga, gb;
    const u32* src = in + i * in_w;
    u32* a32 = reinterpret_cast<u32*>(&ga);
    u32* b32 = reinterpret_cast<u32*>(&gb);
    for (int k = 0; k < 48; k++) {
      a32[k] = src[k];
      b32[k] = src[48 + k];
    }
    X28 pa, pb;
    d28::from_gnark(pa, ga);
    d28::from_gnark(pb, gb);
    F28 ca, cb;
    q28::from_x28(ca, pa);
    q28::from_x28(cb, pb);
    if (op == 8) q28::add(ca, cb);
    else if (op == 9) q28::dbl(ca);
    else q28::mul_small(ca, cb, (u32)(i * 2654435761u) >> 12, 19);  // op 10: k * b, k = 20 bits of a hash of i
    q28::to_x28(pa, ca);
    if (threadIdx.x & 3u) return;
    G1XYZZ o;
    d28::to_gnark(o, pa);
    const u32* o32 = reinterpret_cast<const u32*>(&o);
    for (int k = 0; k < 48; k++) out[i * out_w + k] = o32[k];
    return;
  }
  if (i >= n) return;
  if (op == 11) {  // the GLV split exactly as k_digits runs it
    if (in_w != 8 || out_w != 10) return;
    Fr k;
    for (int j = 0; j < 8; j++) k.l[j] = in[i * in_w + j];
    u32 a[4], b[4], sa, sb;
    glv_split(k, a, b, sa, sb);
    for (int j = 0; j < 4; j++) {
      out[i * out_w + j] = a[j];
      out[i * out_w + 4 + j] = b[j];
    }
    out[i * out_w + 8] = sa;
    out[i * out_w + 9] = sb;
  } else if (op == 12) {  // the conversion k_convert_points runs and the exit product of the MSM kernels
    if (in_w != 24 || out_w != 26) return;
    u32 w[24];
    for (int k = 0; k < 24; k++) w[k] = in[i * in_w + k];
    F28 x, y;
    d28::from_gnark_iso_x(x, w);
    d28::from_gnark_iso_y(y, w + 12);
    u32 o[24];
    d28::to_gnark_msm(o, x, 0);
    d28::to_gnark_msm(o + 12, y, 1);
    for (int k = 0; k < 24; k++) out[i * out_w + k] = o[k];
    // what madd asks of an affine operand: normalised limbs, value below 2p
    F28 x2 = x, y2 = y;
    d28::cond_sub_pshl<1>(x2);
    d28::cond_sub_pshl<1>(y2);
    bool same_x = true, same_y = true;
    for (int k = 0; k < d28::N; k++) {
      same_x = same_x && x2.l[k] == x.l[k] && x.l[k] <= d28::MASK;
      same_y = same_y && y2.l[k] == y.l[k] && y.l[k] <= d28::MASK;
    }
    out[i * out_w + 24] = same_x;
    out[i * out_w + 25] = same_y;
  } else if (op >= 0 && op <= 3) {
    if (in_w != 24 || out_w != 12) return;
    u32 w[24];
    for (int k = 0; k < 24; k++) w[k] = in[i * in_w + k];
    F28 a, b, r;
    d28::from_gnark(a, w);
    d28::from_gnark(b, w + 12);
    if (op == 0) d28::mul(r, a, b);
    else if (op == 1) d28::add(r, a, b);
    else if (op == 2) d28::sub<4>(r, a, b);
    else d28::sqr(r, a);
    u32 o[12];
    d28::to_gnark(o, r);
    for (int k = 0; k < 12; k++) out[i * out_w + k] = o[k];
  } else if (op == 4) {
    if (in_w != 16 || out_w != 8) return;
    Fr a, r;
    for (int k = 0; k < 8; k++) a.l[k] = in[i * in_w + k];
    f_from_mont<FrParams>(r, a);
    for (int k = 0; k < 8; k++) out[i * out_w + k] = r.l[k];
  } else if (op >= 5 && op <= 7) {
    if (in_w != 96 || out_w != 48) return;
    G1XYZZ ga, gb;
    const u32* src = in + i * in_w;
    u32* a32 = reinterpret_cast<u32*>(&ga);
    u32* b32 = reinterpret_cast<u32*>(&gb);
    for (int k = 0; k < 48; k++) {
      a32[k] = src[k];
      b32[k] = src[48 + k];
    }
    X28 acc, b;
    d28::from_gnark(acc, ga);
    d28::from_gnark(b, gb);
    // gnark-form infinity is ZZ = 0 (X = Y = one): from_gnark maps 0 -> 0
    if (op == 5) {
      if (!(f_is_zero(gb.x) && f_is_zero(gb.y))) d28::madd(acc, b.x, b.y);
    } else if (op == 6) {
      d28::add(acc, b);
    } else {
      d28::dbl(acc);
    }
    G1XYZZ o;
    d28::to_gnark(o, acc);
    const u32* o32 = reinterpret_cast<const u32*>(&o);
    for (int k = 0; k < 48; k++) out[i * out_w + k] = o32[k];
  }
}

hipError_t launch_selftest(int op, const uint32_t* d_in, size_t n, uint32_t* d_out, hipStream_t stream) {
  if (op < 0 || op >= kSelftestOps || !d_in || !d_out) return hipErrorInvalidValue;
  if (n == 0) return hipSuccess;
  const SelftestOp& t = kSelftestTable[op];
  const size_t lanes = (size_t)t.lanes * n;
  hipLaunchKernelGGL(k_selftest, dim3(cdiv(lanes, kBlock)), dim3(kBlock), 0, stream, op, d_in, n, d_out, t.in_words,
                     t.out_words);
  return hipGetLastError();
}

}  // namespace curdle
