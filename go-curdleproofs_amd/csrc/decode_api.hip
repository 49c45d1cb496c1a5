// C ABI of the batched point decoding (decode_kernels.hip) and the batched scalar multiplications (group_kernels.hip).
// (Part of msm_api.hip until round 6.)
#include "msm_internal.h"


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
namespace curdle_api {
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
}  // namespace curdle_api

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

namespace curdle_api {
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
}  // namespace curdle_api

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
