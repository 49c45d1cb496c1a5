// One large MSM from pageable host slices (curdle_msm_g1 from 2^19 pairs): graded chunks over ONE plan, every chunk copied,
// sorted and accumulated while the next one crosses PCIe, finished chunks folded into per-bucket sums, one reduction.
// (Part of msm_api.hip until round 6.)
#include "msm_internal.h"

namespace curdle_api {
int run_host_chunked(const uint64_t* points, const uint64_t* scalars, size_t n, uint64_t* out, bool glv) {
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
      pt.join.chunked = true;
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

}  // namespace curdle_api
