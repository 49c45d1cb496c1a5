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
//
// THIS FILE: the sort -- conversion, recoding, bucket counts, the bucket-slot scans, the scatter -- and its launchers.
// The accumulation is msm_accumulate_kernel.hip, everything behind it msm_reduce_kernels.hip, the synthetic bases and
// the self-test msm_misc_kernels.hip (one translation unit until round 6).
#include "msm_kernels_common.h"
#include "dacc_eval.h"

namespace curdle {

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
// added (the round-2 single-block form walked its slots one dependent 4-byte load after the other, twice), two block scans in all.
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
static constexpr u32 kChainMaxTiles = 1024;  // 4,194,304 slots: what one pass of a batch may hold (msm_internal.h kMaxSlotsPerPass)
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
  // The queue of large buckets: ONE atomic per wave reserves the places of all its lanes' large buckets (round 6: one per
  // bucket was 8,000 same-address atomics at ~80 ns each when a few hundred distinct scalar values make every occupied
  // bucket a large one -- the scan took 0.051 ms against 0.018).
  u32 qat;
  {
    u32 nlg = 0;
#pragma unroll
    for (int k = 0; k < kChainPer; k++) nlg += f[k] > max_small ? 1u : 0u;
    const u32 lane = tid & 63u;
    u32 inc = nlg;
#pragma unroll
    for (u32 off = 1; off < 64; off <<= 1) {
      const u32 t = __shfl_up(inc, off, 64);
      if (lane >= off) inc += t;
    }
    const u32 wtot = __shfl(inc, 63, 64);
    u32 wbase = 0;
    if (lane == 63 && wtot) wbase = atomicAdd(nlarge, wtot);
    wbase = __shfl(wbase, 63, 64);
    qat = wbase + inc - nlg;
  }
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
        if (qat < max_large) large[qat] = at + k;
        qat++;
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

// ---------------------------------------------------------------------------
// Launchers
// ---------------------------------------------------------------------------
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
// zero when k_digits starts and are zero again when it ends (msm_enqueue.hip clears them when the buffer is made and after a
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

}  // namespace curdle
