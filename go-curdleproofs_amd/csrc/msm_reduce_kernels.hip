// gfx950 kernels of the BLS12-381 G1 MSM: everything behind the accumulation -- large buckets merged, the bucket reduction
// (running sums over segments on quads, quad28.h), the window sums / bit-positioned points, the batch combine -- and the
// launchers (see msm_sort_kernels.hip for the phases of one MSM).
#include "msm_kernels_common.h"

namespace curdle {

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
// waves on a chip of 1,024 SIMDs for 0.9 ms, a third of the accumulation (profiles/r06_adversarial_before.json).
// Now everything queued is worked on at once, on quads (quad28.h: an addition is 4 product steps):
//   * a bucket of more than 16 fragments is cut into CHUNKS of 16 S consecutive fragments, one wave each: a quad adds S
//     fragments, the wave's 16 sums meet by shuffles, the chunk's sum replaces the chunk's first fragment; the wave that
//     finishes a bucket's LAST chunk (a counter per queue entry, left at zero for the next call) adds the chunk sums the
//     same way into slot 0.  S grows with the bucket (2 up to 1,024 fragments, 4 up to 4,096, 8 up to 16,384, 16 beyond)
//     so that the two serial parts stay balanced -- 4 + 4 + 2 + 4 quad additions for 2,048 fragments, 8 + 4 + 4 + 4 for
//     8,192, instead of 32 + 9 whole ones -- and doubles while there are more chunks than waves;
//   * buckets of at most 16 fragments go FOUR to a wave, a quarter wave each (kMergeSmall below);
//   * the queue is numbered through by prefix sums in LDS, kMergeTile entries per pass, and the launch's blocks are dealt
//     to the passes round robin, so all passes run at once.
static constexpr u32 kMergeTile = 1024;  // queue entries per pass over the queue
// Buckets of at most this many fragments do not take a wave each: FOUR of them share one, a quarter wave (4 quads) per bucket --
// a quad adds <= 4 fragments, two shuffle levels, no counter.  Thousands of 9..16-fragment buckets (a few hundred distinct scalar
// values) are the case: a wave each is 9 of 16 quads at work for five steps.
static constexpr u32 kMergeSmall = 16;
__host__ __device__ inline u32 merge_chunk_shift(u32 m) {  // log2 of the chunk a bucket of m fragments is cut into
  return m <= 1024u ? 5u : m <= 4096u ? 6u : m <= 16384u ? 7u : 8u;
}
__global__ void __launch_bounds__(kBlock, 2)
    k_merge_large(const u32* __restrict__ large, const u32* __restrict__ nlarge, const u32* __restrict__ foff,
                  const u32* __restrict__ fragcnt, X28* __restrict__ frags, u32* __restrict__ done, u32 max_large,
                  u32 frag_stride, u32 prio) {
  __shared__ u32 pre[kMergeTile], spre[kMergeTile], gq[kMergeTile], mq[kMergeTile];
  __shared__ u32 sh_scan[kBlock / 64];
  const u32 nl = min(*nlarge, max_large);
  if (nl == 0) return;  // the usual case
  set_wave_prio(prio);
  const u32 tid = threadIdx.x;
  const u32 lane = tid & 63u, qd = lane >> 2;  // quad inside the wave
  frags += (size_t)blockIdx.y * frag_stride;
  done += (size_t)blockIdx.y * max_large;
  // The queue is worked on in passes of kMergeTile entries (the tables are LDS), and the launch's blocks are dealt to the passes
  // round robin, so that all passes run at once: pass t belongs to blocks t, t + T, t + 2 T, ... (one pass: all blocks; more
  // passes than blocks: a block takes several, one after the other).  With every block walking every pass in turn, 16 passes of
  // 1,024 nine-fragment buckets kept a quarter of the waves busy sixteen times over: 1.07 ms where this takes 0.1.
  const u32 T = (nl + kMergeTile - 1) / kMergeTile;
  const u32 tile = (nl + T - 1) / T;  // passes of equal length (1,280 entries: two of 640, not 1,024 + 256 behind equal teams)
  const u32 team = gridDim.x >= T ? (gridDim.x - (blockIdx.x % T) + T - 1) / T : 1u;  // blocks on my pass
  const u32 nwaves = team * (kBlock / 64);
  const u32 mywave = (gridDim.x >= T ? blockIdx.x / T : 0u) * (kBlock / 64) + (tid >> 6);
  // More chunks than the launch has waves (hundreds of buckets of a hundred fragments each: 64 distinct scalar values)
  // would go round several times at the small buckets' chunk size: every bucket's chunks are doubled (up to 256
  // fragments) until one round takes them all -- 5,120 chunks of 32 on 1,024 waves took 0.30 ms, 2,560 of 64 on 3,072 take 0.1.
  u32 bump = 0;
  {
    u32 mine = 0;
    for (u32 x = tid; x < nl; x += kBlock) {
      const u32 m = fragcnt[large[x]], sh = merge_chunk_shift(m);
      if (m > kMergeSmall) mine += (m + (1u << sh) - 1u) >> sh;
    }
    u32 total0;
    (void)block_exclusive_scan_256(mine, sh_scan, total0);
    while (bump < 3u && (total0 >> bump) > gridDim.x * (kBlock / 64)) bump++;
  }
  auto chunk_shift = [&](u32 m) { return min(merge_chunk_shift(m) + bump, 8u); };
  for (u32 t0 = (blockIdx.x % T) * tile; t0 < nl; t0 += gridDim.x * tile) {  // block-uniform trip count
    const u32 cnt = min(tile, nl - t0);
    __syncthreads();  // the pass before is done with the tables
    u32 nch[kMergeTile / kBlock], nsm[kMergeTile / kBlock], sum = 0, ssum = 0;
#pragma unroll
    for (u32 k = 0; k < kMergeTile / kBlock; k++) {
      const u32 idx = tid * (kMergeTile / kBlock) + k;
      nch[k] = nsm[k] = 0;
      if (idx < cnt) {
        const u32 g = large[t0 + idx], m = fragcnt[g];
        gq[idx] = g;
        mq[idx] = m;
        if (m <= kMergeSmall) {
          nsm[k] = 1;
        } else {
          const u32 sh = chunk_shift(m);
          nch[k] = (m + (1u << sh) - 1u) >> sh;
        }
      }
      sum += nch[k];
      ssum += nsm[k];
    }
    u32 total, stotal;
    u32 ex = block_exclusive_scan_256(sum, sh_scan, total);
    u32 sex = block_exclusive_scan_256(ssum, sh_scan, stotal);
#pragma unroll
    for (u32 k = 0; k < kMergeTile / kBlock; k++) {
      pre[tid * (kMergeTile / kBlock) + k] = ex;
      spre[tid * (kMergeTile / kBlock) + k] = sex;
      ex += nch[k];
      sex += nsm[k];
    }
    __syncthreads();
    // the small buckets of this pass, four to a wave: quarter `lane >> 4` takes the (4 c + quarter)-th of them
    for (u32 c = mywave; 4u * c < stotal; c += nwaves) {  // wave-uniform trip count
      const u32 si = 4u * c + (lane >> 4);
      const bool have = si < stotal;
      u32 lo = 0, hi = cnt;  // the last entry whose count of small ones before it is <= si: the si-th small one
      while (lo < hi) {
        const u32 mid = (lo + hi) >> 1;
        if (spre[mid] > si) hi = mid;
        else lo = mid + 1;
      }
      const u32 e = have ? lo - 1 : 0u;
      const u32 m = have ? mq[e] : 0u;
      X28* f = frags + foff[gq[e]];
      F28 acc, b;
      q28::set_inf(acc);
      for (u32 i = qd & 3u; i < m; i += 4) {  // quad-uniform; at most four
        q28::load(b, &f[i]);
        q28::add(acc, b);
      }
      group_sum_quad(acc, 4u, nullptr);  // every lane of the wave shuffles; sums meet in each quarter's first quad
      if (have && (qd & 3u) == 0) q28::store(&f[0], acc);
    }
    for (u32 c = mywave; c < total; c += nwaves) {  // wave-uniform from here on
      u32 lo = 0, hi = cnt;  // the queue entry of chunk c: the last one whose first chunk is <= c
      while (lo < hi) {
        const u32 mid = (lo + hi) >> 1;
        if (pre[mid] > c) hi = mid;
        else lo = mid + 1;
      }
      u32 e = lo - 1;
      while (mq[e] <= kMergeSmall) e--;  // small entries own no chunk (they share the prefix of the big one before them)
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

// Host-buffer MSMs accumulated in chunks (msm_host_chunks.hip run_host_chunked): a chunk's fragments are folded into
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
// (round 6: three waves per SIMD -- 168 registers, 8 of them spilled -- so that a synchronous mid-size call fits one bucket per quad
// into ONE round of the chip, 196,608 lanes: 2^16 / 2^17 / 2^18 pairs 0.588 / 0.795 / 1.124 -> 0.578 / 0.775 / 1.110 ms, pipelined calls
// equal: profiles/r06_reduce_three_waves_ab.txt)
__global__ void __launch_bounds__(kBlock, 3)
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
hipError_t launch_merge_large(const MsmPlan& p, const MsmWorkspace& ws, hipStream_t stream, uint32_t max_blocks) {
  // One wave per chunk of 32 to 256 fragments; more chunks than waves go round.  The launch is almost always empty
  // (uniform scalars queue nothing) and every block of it has to find room beside the next accumulation before it can
  // read the empty queue and leave: a synchronous call, which has the chip to itself, takes up to 768 blocks = three
  // 165-register waves on every SIMD; a pipelined one 64 (768 empty blocks cost its step 0.02 ms of 2.53, 256 still 0.01
  // of an 8-way rank's 0.43: profiles/r06_pipeline_phase_costs.txt; 64 against 256 on one lease: 2.452 against 2.478 ms
  // per whole MSM, profiles/r06_merge_grid_ab.txt -- a skewed pipelined call goes round a few times, behind other work).
  const u32 nw = p.win_end - p.win_begin;
  const u64 nlanes = ((u64)nw * p.n + p.L - 1) / p.L;  // fragments <= bucket slots + lanes
  const u64 chunks = (u64)p.max_large + ((u64)p.k * p.NB + nlanes) / 32u;
  const u32 cap = max_blocks < 1u ? 1u : max_blocks > 768u ? 768u : max_blocks;
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

}  // namespace curdle
