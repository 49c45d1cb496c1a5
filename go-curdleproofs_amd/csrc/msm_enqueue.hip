// Phase sequencing of one MSM call on a workspace slot: every GPU phase enqueued on the slot's streams (enqueue_slot), the
// wait and the host's Horner pass (finish_slot), batches in passes, and the synchronous device / host-buffer paths built
// from them.  (Part of msm_api.hip until round 6.)
#include "msm_internal.h"

namespace curdle_api {
#ifdef CURDLE_EXP_SKIP
// Experiment build only (tools/exp/phase_costs.sh -> build_alt/, never the product library): phases of a PIPELINED call left
// out after the slot's first uses, so that what each phase costs the pipeline can be priced (profiles/r06_pipeline_phase_costs.txt).
// curdle_debug_skip(mask) bits: 1 conversion, 2 sort, 4 merge_large, 8 reduce_segments, 16 reduce_level.  Results are garbage by construction.
static std::atomic<unsigned> g_exp_skip{0};
static unsigned exp_skip_mask() { return g_exp_skip.load(std::memory_order_relaxed); }
}  // namespace curdle_api
extern "C" int curdle_debug_skip(unsigned mask) {  // (tools/bench_pipeline.py sets it from CURDLE_DEBUG_SKIP: no getenv in the library)
  curdle_api::g_exp_skip.store(mask, std::memory_order_relaxed);
  return 0;
}
namespace curdle_api {
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
                 bool latency_mode, bool points28_ready, size_t sets, bool many, const ChunkJoin* join, const void* ext_points28,
                 bool light_host, bool glv, const DaccFront* dfront) {
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
  // ... and ONE merge limit for every chunk of an MSM, in both of its enqueue steps, whatever the sizes of the chunks say: the
  // reduction reads all their fragment lists under the last chunk's plan, and a bucket merged under one limit and read under
  // another would count twice.  8, what the size rule gives every input large enough to be chunked: a bucket just under the limit
  // is walked fragment by fragment by the fold and by the reduction (limits of 16 and 32 made 256..1,024 distinct scalar values
  // from host slices 1.3-1.6x a uniform call: profiles/r06_adversarial_distinct_k.txt).
  if (join && join->chunked) p.max_small = 8;
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
  // device array and a copy command behind the last kernel (~10 us of every synchronous call).
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
  // The merge launch's grid: a synchronous call has the chip to itself (768 blocks; also the one-chunk host-buffer call, whose join
  // only splits its own enqueue in two); a chunk of a chunked call runs beside the next chunk's accumulation (256); a pipelined call
  // pays for every empty block (64: msm_reduce_kernels.hip launch_merge_large).
  const bool alone = !join || !join->chunked;
  const uint32_t merge_blocks = latency_mode && alone ? 768u : (join && join->chunked ? 256u : 64u);
  if (!EXP_SKIP(4)) HIP_TRY(launch_merge_large(p, ws, stream, merge_blocks));
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
SyncStreams sync_streams(Ctx&, Slot& S) { return {S.stream, S.stream, S.stream}; }

// The scans of the bucket slots hold 1,024 blocks of 4,096 slots: a batch with more slots than
// that (1,024 MSMs of 2,548 pairs; 2,100 of 628) runs in passes of as many whole MSMs as fit,
// one after the other on the caller's slot -- every pass is milliseconds of GPU work, so the
// gap between two passes is noise, and the workspaces stay bounded.

// enqueue + finish of k MSMs on slot S, in passes if the batch is too large for one.
int run_passes(Ctx& cx, Slot& S, const void* d_points, const void* d_scalars, const uint32_t* h_off, size_t k, int c,
               int win_begin, int win_end, hipStream_t pre, hipStream_t main, hipStream_t tail, uint64_t* out,
               const void* ext_points28, bool glv) {
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
               int win_end, uint64_t* out, void* user_stream, const void* ext_points28, bool glv, hipEvent_t wait_for) {
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
int run_host(const uint64_t* points, const uint64_t* scalars, const uint32_t* h_off, size_t k, uint64_t* out, bool glv) {
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
}  // namespace curdle_api
