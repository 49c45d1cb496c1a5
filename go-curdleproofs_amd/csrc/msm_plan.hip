// The plan of one MSM call (MsmPlan, msm_kernels.h): window width by size, the windows' widths, segment lengths of the
// bucket reduction, positions per accumulate lane, which sort / scan / reduction form -- every rule with the measurement
// that set it -- and the window-count entry points of the C ABI.  (Part of msm_api.hip until round 6.)
#include "msm_internal.h"

namespace curdle_api {
int choose_window_bits(size_t n, bool many) {
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
// every scalar into two 127-bit halves (GLV, msm_sort_kernels.hip: k P = k1 P + k2 phi(P)), so an MSM
// of n pairs is 2 n terms over W = ceil(127 / c) windows, the 127 bits spread as evenly as
// possible (the wider windows lowest), the top window unsigned.  For c = 16 that is 7 windows
// of 16 bits and a 15-bit top window.
// (Without the split -- CURDLE_MSM_ANY_CURVE_POINT -- the same rule over the 255 bits of the whole scalar.)
int window_widths(int c, uint8_t bits[kMaxWindows], int scalar_bits) {
  const int W = (scalar_bits + c - 1) / c;
  const int base = scalar_bits / W, extra = scalar_bits % W;
  for (int w = 0; w < W; w++) bits[w] = (uint8_t)(base + (w < extra ? 1 : 0));
  return W;
}

// Plan for k MSMs of n_total pairs in all, the largest having n_max pairs.
int make_plan(MsmPlan& p, size_t n_total, size_t k, size_t n_max, int c, int win_begin, int win_end,
              bool latency_mode, size_t sets, bool many, uint32_t seg_override, bool light_host, bool glv) {
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
    const uint64_t lanes = 196608;  // (round 6: k_reduce_segments runs three waves per SIMD; 131,072 = two until then)
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
  // Single MSMs reduce their buckets without a scalar multiple (msm_reduce_kernels.hip, k_reduce_segments):
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
  // ... and beyond that: what an evenly loaded bucket of the narrowest window comes to plus four standard deviations of its
  // (Poisson) load, between 8 and 16 -- 16 at 8,192 pairs (8 + 1 fragments, sigma 1.4), 12 at 2^16 (6.4 + 1, sigma 0.8), 8 from
  // 2^18 on (1.8 + 1 at L = 36; 1.5 at N = 2^20, L = 128): uniform inputs still queue nothing, and a bucket just UNDER the limit,
  // which the reduction walks fragment by fragment on its chain, is shorter.  Eight such buckets in one segment made the reduction
  // of a synchronous 2^20 call 0.74 ms where it takes 0.25 (600 distinct scalar values: 13-16 fragments in every occupied bucket;
  // profiles/r06_adversarial_distinct_k.txt).
  p.max_small = 16;
  if (k * sets == 1) {
    if (n_total <= 8192) {
      p.max_small = 8;
    } else if (min_nbkt) {
      const uint64_t load = (n_max + min_nbkt - 1) / min_nbkt;
      uint64_t sd = 0;
      while ((sd + 1) * (sd + 1) <= load) sd++;  // floor(sqrt(load))
      const uint64_t lim = (load + 4 * (sd + 1) + p.L - 1) / p.L + 2;  // ceil((load + 4 sigma) / L) + 1 for the straddled lane + 1
      p.max_small = (uint32_t)(lim < 8 ? 8 : lim > 16 ? 16 : lim);
    }
  }
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
  // The scatter in two passes (msm_sort_kernels.hip): one MSM whose windows are whole numbers of
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

}  // namespace curdle_api

// ---------------------------------------------------------------------------
// MSM
// ---------------------------------------------------------------------------
extern "C" int curdle_msm_window_bits(size_t n) { return choose_window_bits(n); }

// The window width a call will run with: the caller's, or the library's choice for n pairs; anything outside
// [4, 16] is refused HERE, before window_widths() writes kMaxWindows bytes for it (review of round 5: a width of
// 1..3 has more than 64 windows, a large negative one none at all).
int curdle_api::checked_window_bits(size_t n, int window_bits, int* c) {
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
