/*
 * curdle_msm.h -- C ABI of libcurdlemsm.so: BLS12-381 G1 multi-scalar
 * multiplication on MI355X (gfx950), the drop-in for the MSM path of
 * jsign/go-curdleproofs -- plus, further down, the entry points of the layers either side of
 * that path (batched point decoding, batched scalar multiplications, the protocol and Whisk
 * restatements that funnel every MSM into the GPU).
 *
 * Every entry point replaces one reference interface (paths relative to
 * /root/reference); INTEGRATION.md shows the cgo binding for each.
 *
 * Data layouts are gnark-crypto v0.11.0's in-memory layouts (go.mod:6), so Go
 * slices can be handed over with unsafe.Pointer(&s[0]) and no copy:
 *   fr.Element            = 4 x uint64, little-endian limbs, Montgomery (R = 2^256)
 *   fp.Element            = 6 x uint64, little-endian limbs, Montgomery (R = 2^384)
 *   bls12381.G1Affine     = {X, Y fp.Element}   = 12 x uint64; (0,0) is infinity
 *   bls12381.G1Jac        = {X, Y, Z fp.Element} = 18 x uint64; Z = 0 is infinity
 *
 * Results are returned as the canonical Jacobian representative: (x, y, 1) of
 * the affine result (Z = Montgomery one), or (1, 1, 0) for infinity.  The
 * reference never inspects Jacobian limbs (only Equal / AddAssign / to-affine,
 * e.g. msmaccumulator/msmaccumulator.go:63), so any representative is valid;
 * a canonical one makes results byte-comparable.
 *
 * All functions return CURDLE_OK (0) or a negative CURDLE_E* code; the text of
 * the last error on the calling thread is available from curdle_last_error().
 * No C++ exception crosses this boundary.  Calls are synchronous (no pointer
 * is retained after return -- the cgo rule) and thread-safe.  There is no CPU
 * fallback: if no gfx950 device is usable the MSM entry points fail with
 * CURDLE_ENODEV.
 */
#ifndef CURDLE_MSM_H
#define CURDLE_MSM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CURDLE_OK 0
#define CURDLE_EINVAL (-1) /* bad argument (null pointer, length mismatch)      */
#define CURDLE_ENODEV (-2) /* no usable HIP device / library not initialised    */
#define CURDLE_EHIP (-3)   /* a HIP runtime call failed                         */
#define CURDLE_ENOMEM (-4) /* device or host allocation failed                  */
#define CURDLE_EBUSY (-5)  /* every MSM slot is in flight (async API)              */
#define CURDLE_MSM_SLOTS 8 /* MSMs that can be in flight at once                   */

#define CURDLE_G1_AFFINE_U64 12
#define CURDLE_G1_JAC_U64 18
#define CURDLE_FR_U64 4

/* ------------------------------------------------------------------------- *
 * Library life cycle
 * ------------------------------------------------------------------------- */

/* Selects the HIP device this process computes on (one process per GPU) and
 * creates the stream + workspace.  Called implicitly with device 0 by the
 * first MSM call.  Replaces: common.MultiExpConf (common/util.go:14), the
 * reference's only configuration knob. */
int curdle_init(int device);
/* One process driving SEVERAL GPUs (the reference's caller is one
 * process making synchronous MultiExp calls, msmaccumulator/msmaccumulator.go:59): configures
 * one context per entry of devices[] (HIP device ids; an id may repeat, which puts two
 * contexts on one GPU -- how the multi-device code is tested on a one-GPU machine), each with
 * its own streams, workspace slots and one host worker thread.  After it
 *   - every entry point runs on the calling thread's CURRENT context: curdle_set_device(i)
 *     selects context i for this thread (default 0), like hipSetDevice; tickets and handles
 *     (curdle_dacc, decode tickets) remember their context; a curdle_dbases / curdle_crs is
 *     made resident lazily on every context that uses it;
 *   - curdle_msm_g1 splits a large host-buffer MSM by POINT RANGES over all contexts (each GPU
 *     copies and runs its own n / D pairs, the host thread of each device drives it, the D
 *     partial sums are added on the host with the code of curdle_g1_sum: no collective) --
 *     unless the calling thread has SELECTED a device with curdle_set_device: a thread pinned
 *     to a device (a batch shard, curdlemsm.OnDevice) keeps its MSM there;
 *   - curdle_msm_g1_replicated does the same for inputs the caller keeps resident on every
 *     GPU, by Pippenger windows (north_star's partition) or point ranges;
 *   - curdle_verify_batch and curdle_whisk_is_valid_shuffle_proof_batch shard their proofs
 *     over the contexts (BASELINE config 5: replicas, no data-path exchange).
 * n = 1 is curdle_init(devices[0]).  Fails with CURDLE_EINVAL if the library is already
 * initialised on a different list.  All or nothing: if one entry cannot be brought up (bad id,
 * out of memory), every context this call created is closed again and a retry starts clean.
 * The multi-device paths were exercised as two contexts on one GPU and under a logical-device
 * interposer (tools/logical_devices_shim.cpp), never on two physical GPUs: unmeasured there. */
#define CURDLE_MAX_DEVICES 16
int curdle_init_devices(const int* devices, int n);
int curdle_device_count(void);      /* contexts configured (1 unless curdle_init_devices said more) */
/* This thread's current context, 0 <= ordinal < curdle_device_count(); -1 = no selection (context
 * 0, and large host-buffer MSMs may spread over all devices again). */
int curdle_set_device(int ordinal);
int curdle_get_device(void);
/* The calling thread's SELECTION: the ordinal it passed to curdle_set_device, or -1 if it has made none (or
 * cleared it with -1).  What a binding that selects a device temporarily restores afterwards
 * (curdlemsm.OnDevice): restoring curdle_get_device()'s 0 would leave the thread pinned to device 0. */
int curdle_get_device_selection(void);
/* Diagnostics: host-buffer MSMs (curdle_msm_g1) that were spread over several devices since the library was loaded. */
unsigned long long curdle_stat_spread_calls(void);
/* Closes every context (streams, workspaces, host threads).  CURDLE_EBUSY while an MSM or a point
 * decoding is in flight on a slot, or a call that spans the devices' host threads is running.  Like every teardown of a library, it must not run
 * concurrently with other entry points: a call that has not taken its slot yet is not seen. */
int curdle_shutdown(void);
/* Copies the calling thread's last error text (NUL-terminated) into buf. */
int curdle_last_error(char* buf, size_t len);

/* Test / measurement hook: the library's tunables ("knobs", go-curdleproofs_amd/host/knobs.h: window
 * width, lane and segment lengths, chunk counts, ...) are read ONCE from the environment
 * (CURDLE_<NAME>), when the library is first used, and never again.  This call changes one
 * afterwards -- name with or without the CURDLE_ prefix, value < 0 = back to "not set" (the
 * library's own rule) -- so that a test can walk a knob through its range inside one process.
 * Not for production callers; CURDLE_EINVAL for a name the table does not have. */
int curdle_plan_override(const char* name, long long value);
/* 1 if a HIP device is visible to the library, else 0 (never fails). */
int curdle_device_available(void);

/* ------------------------------------------------------------------------- *
 * The hot path: (*G1Jac).MultiExp(points []G1Affine, scalars []fr.Element, cfg)
 *   gnark-crypto v0.11.0, called at msmaccumulator/msmaccumulator.go:59 and
 *   the 38 other sites listed in SURVEY.md section 8(a).
 * out_jac = sum_i scalars[i] * points[i].  n = 0 gives infinity and CURDLE_OK
 * (msmaccumulator_test.go:14 runs sizes 0..3).  Inputs are host memory and
 * are copied to the device on every call: the prover mutates bases in place
 * between calls (innerproductargument.go:155-166), so nothing is cached by
 * pointer.
 * PRECONDITION (differs from gnark): every base must lie in the prime-order subgroup G1 (or be
 * the point at infinity).  The kernels split every scalar as k P = k1 P + k2 phi(P) with the
 * curve endomorphism phi(x, y) = (beta x, y), which is multiplication by lambda only on G1;
 * gnark's MultiExp does not use the endomorphism and is defined for any curve point.  Every
 * base the reference feeds this path is in G1 (CRS points are multiples of the generator,
 * proof and tracker points pass gnark's Decoder / SetBytes subgroup check).  For an on-curve
 * point outside G1 -- e.g. one taken from curdle_g1_decompress with subgroup_check = 0, or from
 * the begin / points decoding steps before the subgroup verdict -- the result is the well-defined
 * but different point k1 P + k2 (beta x, y)
 * (tests/test_msm_gpu.py::test_bases_outside_the_prime_order_subgroup_...).  The same holds for
 * curdle_g1_scalar_mul_batch and the device accumulator.
 * ------------------------------------------------------------------------- */
int curdle_msm_g1(const uint64_t* points, const uint64_t* scalars, size_t n,
                  uint64_t out_jac[CURDLE_G1_JAC_U64]);

/* The same call with options (flags = 0 is curdle_msm_g1).
 *
 * CURDLE_MSM_ANY_CURVE_POINT -- the opt-out from the precondition above: gnark's contract.  The scalars are
 *   NOT split with the endomorphism; each is recoded whole (255 bits, twice the windows), so the result is
 *   k P for every point P of the curve y^2 = x^3 + 4, in the prime-order subgroup or not -- what
 *   (*G1Jac).MultiExp (go.mod:6) returns for such input.  Costs about twice the bucket reduction and a digit
 *   array of twice the needed size; the additions are the same.  For callers that take bases from outside
 *   without a subgroup check (curdleproof.Verify takes []G1Affine from its caller, curdleproof.go:199-207).
 * CURDLE_MSM_BASES_UNCHANGED (device-input entry points only) -- the caller promises that the n points at
 *   d_points have not changed since the previous call that named the same (d_points, n) with this flag on this
 *   context: the library keeps its converted copy of them (one per pointer and count, at most four per context,
 *   the least recently used idle one is replaced; 256 bytes per point) and converts nothing on later calls.
 *   This is the "explicit, caller-managed" form of caching SURVEY.md section 8b allows: without the flag nothing
 *   is ever cached by pointer (the prover mutates its bases in place between calls).  A rank of the multi-GPU
 *   window split passes the same resident bases on every call; msmaccumulator.Verify's bases are mostly the CRS.
 *   curdle_msm_forget_bases(d_points) drops the copy (before the memory is freed or rewritten). */
#define CURDLE_MSM_ANY_CURVE_POINT 1u
#define CURDLE_MSM_BASES_UNCHANGED 2u
int curdle_msm_g1_ex(const uint64_t* points, const uint64_t* scalars, size_t n, unsigned flags,
                     uint64_t out_jac[CURDLE_G1_JAC_U64]);

/* Same MSM with inputs already resident in device memory (HIP device
 * pointers, same layouts).  `stream` is a hipStream_t or NULL for the
 * library's own stream.  This is what bench.py times. */
int curdle_msm_g1_device(const void* d_points, const void* d_scalars, size_t n,
                         uint64_t out_jac[CURDLE_G1_JAC_U64], void* stream);
/* ... with the flags of curdle_msm_g1_ex (both apply to device inputs). */
int curdle_msm_g1_device_ex(const void* d_points, const void* d_scalars, size_t n, unsigned flags,
                            uint64_t out_jac[CURDLE_G1_JAC_U64], void* stream);
int curdle_msm_forget_bases(const void* d_points);

/* The same MSM over inputs the caller keeps resident on EVERY configured context:
 * d_points[i] / d_scalars[i] are device pointers on context i's GPU, each holding all n pairs
 * (curdle_device_count() entries).  split: 1 = Pippenger windows [w_i, w_i+1) per context
 * (north_star's partition: every GPU walks all n pairs for its windows), 2 = point ranges
 * (every GPU runs all windows over its n / D pairs), 0 = the library's choice (windows up to
 * 2^21 pairs, point ranges beyond: the per-rank step times of DESIGN.md section 5).  One host
 * thread per device; the D 144-byte partials are summed on the host. */
int curdle_msm_g1_replicated(const void* const* d_points, const void* const* d_scalars, size_t n, int split,
                             uint64_t out_jac[CURDLE_G1_JAC_U64]);

/* Asynchronous form of the device-resident MSM.  submit() enqueues every GPU phase
 * of one MSM (window range as below; window_bits = 0, win_begin = 0, win_end = -1 for
 * the whole MSM) on a free workspace slot and returns immediately with a ticket;
 * wait() blocks until that MSM is done, finishes it on the host and writes the
 * result.  Up to CURDLE_MSM_SLOTS MSMs can be in flight: the latency-bound tail of
 * one (bucket reduce, D2H, host combine) then overlaps the accumulation of the next.
 * The device inputs must stay valid and unchanged until wait() returns.  submit()
 * fails with CURDLE_EBUSY when every slot is in flight. */
int curdle_msm_g1_device_submit(const void* d_points, const void* d_scalars, size_t n,
                                int window_bits, int win_begin, int win_end, int* ticket);
int curdle_msm_g1_device_submit_ex(const void* d_points, const void* d_scalars, size_t n,
                                   int window_bits, int win_begin, int win_end, unsigned flags, int* ticket);
int curdle_msm_wait(int ticket, uint64_t out_jac[CURDLE_G1_JAC_U64]);
/* How many of the CURDLE_MSM_SLOTS workspace slots are free right now (a hint: other threads
 * take and release slots concurrently).  A caller that would hold a slot across host work of
 * its own (the verifier starts its accumulation before it hashes) uses it to decide whether
 * to take the slot early or late. */
int curdle_msm_free_slots(void);

/* Partial MSM over Pippenger windows [win_begin, win_end) of the
* decomposition the library would use for (n, window_bits); the partial is
 * already scaled by 2^(bit offset of window win_begin).  Summing the partials of a
 * partition of [0, curdle_msm_num_windows()) with curdle_g1_sum() gives the
 * full MSM.  This is the multi-GPU split of north_star: one rank per GPU, each
 * taking a window range, 144-byte partials all-gathered over RCCL.
 * window_bits = 0 lets the library choose. */
int curdle_msm_g1_device_windows(const void* d_points, const void* d_scalars, size_t n,
                                 int window_bits, int win_begin, int win_end,
                                 uint64_t out_jac[CURDLE_G1_JAC_U64], void* stream);
int curdle_msm_g1_device_windows_ex(const void* d_points, const void* d_scalars, size_t n,
                                    int window_bits, int win_begin, int win_end, unsigned flags,
                                    uint64_t out_jac[CURDLE_G1_JAC_U64], void* stream);
int curdle_msm_window_bits(size_t n);              /* the library's choice of c for n   */
int curdle_msm_num_windows(size_t n, int window_bits); /* W = ceil(127 / c) for that choice */
/* Widths of the W windows for (n, window_bits), lowest window first (sum = 127: the
 * library splits every scalar into two 127-bit halves, k P = k1 P + k2 phi(P), and a
 * window covers the same bits of both): the bits are spread as evenly as possible; all
 * windows but the top one are recoded into signed digits, the top one is unsigned.
 * Returns W. */
int curdle_msm_window_widths(size_t n, int window_bits, int widths[64]);
/* The same two for a call that passes `flags` (review of round 5): with CURDLE_MSM_ANY_CURVE_POINT the
 * plan recodes the whole 255-bit scalar, so it has W = ceil(255 / c) windows whose widths sum to 255,
 * and the window ranges of *_windows_ex / *_submit_ex are ranges over THOSE windows -- a caller that
 * partitions [0, curdle_msm_num_windows()) and passes the flag would cover the low half of every
 * scalar only, with no error.  flags = 0 (or CURDLE_MSM_BASES_UNCHANGED alone) gives the values above. */
int curdle_msm_num_windows_ex(size_t n, int window_bits, unsigned flags);
int curdle_msm_window_widths_ex(size_t n, int window_bits, unsigned flags, int widths[64]);

/* out = sum of k Jacobian points (host memory, any representatives).
 * Replaces the chain of G1Jac.AddAssign a caller would do on partials. */
int curdle_g1_sum(const uint64_t* jac_points, size_t k, uint64_t out_jac[CURDLE_G1_JAC_U64]);

/* k independent MSMs in one call (many
 * concurrent msmaccumulator.Verify calls).  MSM j covers pairs
 * [offsets[j], offsets[j+1]) of the concatenated points/scalars arrays;
 * offsets has k+1 entries.  out_jac holds k results. */
int curdle_msm_g1_batch(const uint64_t* points, const uint64_t* scalars,
                        const size_t* offsets, size_t k, uint64_t* out_jac);

/* The same with inputs resident in device memory (offsets stays a host array). */
int curdle_msm_g1_batch_device(const void* d_points, const void* d_scalars, const size_t* offsets,
                               size_t k, uint64_t* out_jac, void* stream);

/* k MSMs that share ONE scalar vector against k base sets of n points each
 * (samemultiscalarargument.go:64-70 and :206,218,231; curdleproof.go:110,114).
 * points_sets[j] points at n affine points. */
int curdle_msm_g1_multi(const uint64_t* const* points_sets, size_t k,
                        const uint64_t* scalars, size_t n, uint64_t* out_jac);

/* ------------------------------------------------------------------------- *
 * msmaccumulator (msmaccumulator/msmaccumulator.go:11-64), host-side mirror.
 * The Go package keeps its own map; these entry points exist so that the
 * C++/Python side of this repository can run the reference's accumulator
 * tests through the same library.  `rand` is a curdle_rand handle
 * (common/rand.go).
 * ------------------------------------------------------------------------- */
typedef struct curdle_acc curdle_acc;
typedef struct curdle_rand curdle_rand;

curdle_rand* curdle_rand_new(uint64_t seed);                  /* common.NewRand, rand.go:19        */
void curdle_rand_free(curdle_rand* r);
int curdle_rand_get_fr(curdle_rand* r, uint64_t out_fr[4]);   /* GetFr, rand.go:35 (Montgomery)    */
int curdle_rand_get_g1_affine(curdle_rand* r, uint64_t out_aff[12]); /* GetG1Affine, rand.go:72    */
int curdle_rand_permutation(curdle_rand* r, size_t n, uint32_t* out); /* GeneratePermutation :97   */

curdle_acc* curdle_acc_new(void);                             /* New, msmaccumulator.go:16         */
void curdle_acc_free(curdle_acc* a);
/* AccumulateCheck(C, x, v, rand), msmaccumulator.go:23.  len(x) != len(v) is
 * CURDLE_EINVAL ("x and v must have the same length"). */
int curdle_acc_accumulate_check(curdle_acc* a, const uint64_t C_jac[18],
                                const uint64_t* x, size_t x_len,
                                const uint64_t* v, size_t v_len, curdle_rand* rand);
/* The same check with C given as the linear combination it would have been computed
 * from, C = sum_j c_scalars[j] * c_points[j] (affine points, Montgomery scalars): alpha is
 * drawn exactly as in AccumulateCheck and the terms -alpha * c_scalars[j] join the base /
 * scalar map, so Verify()'s one MSM checks the reference's equation with alpha * C moved
 * to the other side -- no MSM for C and no alpha * C scalar multiplication now.  A_c is
 * left unchanged.  Same accept bit as accumulate_check(C) for every input. */
int curdle_acc_accumulate_check_deferred(curdle_acc* a, const uint64_t* c_scalars, const uint64_t* c_points,
                                         size_t c_len, const uint64_t* x, size_t x_len,
                                         const uint64_t* v, size_t v_len, curdle_rand* rand);
/* Verify(), msmaccumulator.go:49: *ok = 1 iff MSM(bases, scalars) == A_c.  The
 * MSM runs on the GPU through curdle_msm_g1. */
int curdle_acc_verify(curdle_acc* a, int* ok);
int curdle_acc_get_A_c(const curdle_acc* a, uint64_t out_jac[18]); /* exported field A_c, :12     */
size_t curdle_acc_num_bases(const curdle_acc* a);
/* Flattened map (bases then scalars), in insertion order. */
int curdle_acc_export(const curdle_acc* a, uint64_t* points, uint64_t* scalars);

/* ------------------------------------------------------------------------- *
 * Protocol layers around the hot path: a host-side
 * restatement of the reference's CRS, ShufflePermuteCommit, curdleproof.Prove and
 * curdleproof.Verify, so that a whole Verify (all sub-argument MSMs + the final
 * batched msmaccumulator MSM on the GPU) can be run without a Go toolchain.  The Go
 * package keeps its own protocol code; this is the caller either side of the MSM,
 * for end-to-end tests and the verifies/s measurement.  UNVERIFIED against
 * Go-produced proofs.
 * Points are gnark G1Affine arrays (ell x 12 u64), M a G1Jac, scalars fr.Elements.
 * ------------------------------------------------------------------------- */
/* BYTE COMPATIBILITY WITH THE GO IMPLEMENTATION IS UNVERIFIED for every entry point of this
 * section that produces or consumes serialised proofs (curdle_prove, curdle_verify,
 * curdle_proof_from_bytes, curdle_verify_batch, curdle_whisk_*): no Go toolchain and no
 * Go-produced proof exist in the build environment.  The transcript labels, challenge order,
 * gnark Encoder framing and verifier equations were checked against the reference source by
 * reading; what the tests pin is Merlin's published vector, the common.Rand known answers, the
 * reference's size constants (48 / 128 / 4576 bytes) and this implementation's own proof bytes
 * for fixed seeds (tests/golden/proof_vectors.npz).  The MSM entry points above do not depend
 * on any of this. */
typedef struct curdle_crs curdle_crs;
curdle_crs* curdle_crs_generate(size_t ell, curdle_rand* rand);   /* GenerateCRS, crs.go:20           */
void curdle_crs_free(curdle_crs* crs);
size_t curdle_crs_size(const curdle_crs* crs);
/* common.ShufflePermuteCommit, common/util.go:45 */
int curdle_shuffle_permute_commit(const curdle_crs* crs, const uint64_t* Rs, const uint64_t* Ss, size_t ell,
                                  const uint32_t* perm, const uint64_t k[4], curdle_rand* rand,
                                  uint64_t* Ts_out, uint64_t* Us_out, uint64_t M_out[18], uint64_t rs_m_out[16]);
/* curdleproof.Prove, curdleproof.go:38; the proof is returned serialised (Proof.Serialize, :358).
 * If cap is too small *proof_len still receives the needed size. */
int curdle_prove(const curdle_crs* crs, const uint64_t* Rs, const uint64_t* Ss, const uint64_t* Ts,
                 const uint64_t* Us, size_t ell, const uint64_t M[18], const uint32_t* perm,
                 const uint64_t k[4], const uint64_t rs_m[16], curdle_rand* rand,
                 uint8_t* proof_out, size_t cap, size_t* proof_len);
/* curdleproof.Verify, curdleproof.go:199, on a serialised proof (Proof.FromReader, :320):
 * returns CURDLE_OK with *ok = accept bit for (true|false, nil); a negative code for
 * (false, err) -- malformed proof, zero randomizer, device failure. */
int curdle_verify(const curdle_crs* crs, const uint8_t* proof, size_t proof_len, const uint64_t* Rs,
                  const uint64_t* Ss, const uint64_t* Ts, const uint64_t* Us, size_t ell,
                  const uint64_t M[18], curdle_rand* rand, int* ok);
/* The same split as the reference's API, where Verify takes a decoded `Proof` value
 * (curdleproof.go:199) and decoding is Proof.FromReader (:320): decode once -- curve and
 * subgroup checks on every point -- into a handle, verify the handle (this is what the
 * reference's BenchmarkVerifier times, curdleproof_test.go:210-237), free it. */
typedef struct curdle_proof curdle_proof;
int curdle_proof_from_bytes(const uint8_t* proof, size_t proof_len, curdle_proof** out);
void curdle_proof_free(curdle_proof* p);
int curdle_verify_proof(const curdle_crs* crs, const curdle_proof* proof, const uint64_t* Rs,
                        const uint64_t* Ss, const uint64_t* Ts, const uint64_t* Us, size_t ell,
                        const uint64_t M[18], curdle_rand* rand, int* ok);
/* Cross-proof batch verification (the reference verifies one proof
 * at a time): k proofs over the same CRS and the same ell.  The host part of each proof runs
 * on `nthreads` worker threads; all proofs' checks are folded into ONE accumulator (per-proof
 * randomness derived from `rand`; the CRS bases merge) and ONE MSM of ~k * (4 ell + 100) pairs
 * decides.  oks[i] receives each proof's accept bit: if the batch MSM fails, the proofs are
 * settled one by one, so the bits are exact either way.  A malformed proof or a zero
 * randomizer counts as rejected here (curdle_verify reports those as errors).
 * proofs / Rs / Ss / Ts / Us: arrays of k pointers; Ms: k x 18 limbs. */
int curdle_verify_batch(const curdle_crs* crs, size_t k, const uint8_t* const* proofs, const size_t* proof_lens,
                        const uint64_t* const* Rs, const uint64_t* const* Ss, const uint64_t* const* Ts,
                        const uint64_t* const* Us, size_t ell, const uint64_t* Ms, curdle_rand* rand,
                        int nthreads, int* oks);
/* How curdle_verify evaluates the check points it hands to the accumulator.  0 (default):
 * deferred -- each is passed as the linear combination of proof / statement points it is,
 * so a verification is ONE MSM on the GPU.  1: eager -- evaluated where the reference
 * evaluates them (a MultiExp per argument, then AccumulateCheck's alpha * C on the host).
 * Identical accept bit; the eager mode exists for differential tests.  Initial value from
 * the environment variable CURDLE_VERIFY_EAGER.  Returns the previous setting. */
int curdle_verify_set_eager(int eager);
int curdle_proof_reencode(const uint8_t* proof, size_t proof_len, uint8_t* out, size_t cap, size_t* out_len);
/* The reference's `whisk` package (whisk/whisk.go, whisk/types.go): the byte-level API of the
 * Whisk SSLE spec.  A tracker is 96 bytes (rG || krG, gnark compressed points, types.go:73-76);
 * a shuffle proof is CURDLE_WHISK_SHUFFLE_PROOF_SIZE bytes (M, the curdleproof, zero padding,
 * types.go:53-71) over CURDLE_WHISK_ELL trackers; a tracker (opening) proof is 128 bytes (A, B, s).
 * Every decoded point is curve- and subgroup-checked, as gnark's Decoder / SetBytes do.
 * Return CURDLE_OK with *ok = the accept bit for (true|false, nil); negative for (false, err). */
#define CURDLE_WHISK_ELL 124
#define CURDLE_WHISK_TRACKER_SIZE 96
#define CURDLE_WHISK_TRACKER_PROOF_SIZE 128
#define CURDLE_WHISK_SHUFFLE_PROOF_SIZE 4576
/* IsValidWhiskShuffleProof, whisk.go:20 */
int curdle_whisk_is_valid_shuffle_proof(const curdle_crs* crs, const uint8_t* pre_trackers, const uint8_t* post_trackers,
                                        size_t n_pre, size_t n_post,
                                        const uint8_t proof[CURDLE_WHISK_SHUFFLE_PROOF_SIZE], curdle_rand* rand, int* ok);
/* k shuffle proofs over one CRS at once (no reference counterpart; BASELINE config 5): all
 * points of all proofs and tracker sets decoded by one GPU kernel, the proofs verified like
 * curdle_verify_batch.  pre / post: k pointers to n trackers each; proofs: k pointers to
 * CURDLE_WHISK_SHUFFLE_PROOF_SIZE bytes.  oks[i] = accept bit; what does not decode is rejected. */
int curdle_whisk_is_valid_shuffle_proof_batch(const curdle_crs* crs, size_t k, const uint8_t* const* pre_trackers,
                                              const uint8_t* const* post_trackers, size_t n,
                                              const uint8_t* const* proofs, curdle_rand* rand, int nthreads, int* oks);
/* GenerateWhiskShuffleProof, whisk.go:63: n must be CURDLE_WHISK_ELL (the permutation length is fixed there) */
int curdle_whisk_generate_shuffle_proof(const curdle_crs* crs, const uint8_t* pre_trackers, size_t n, curdle_rand* rand,
                                        uint8_t* post_trackers_out, uint8_t proof_out[CURDLE_WHISK_SHUFFLE_PROOF_SIZE]);
/* IsValidWhiskTrackerProof, whisk.go:116 (host only: four scalar multiplications) */
int curdle_whisk_is_valid_tracker_proof(const uint8_t tracker[CURDLE_WHISK_TRACKER_SIZE], const uint8_t k_commitment[48],
                                        const uint8_t proof[CURDLE_WHISK_TRACKER_PROOF_SIZE], int* ok);
/* GenerateWhiskTrackerProof, whisk.go:149 */
int curdle_whisk_generate_tracker_proof(const uint8_t tracker[CURDLE_WHISK_TRACKER_SIZE], const uint64_t k[4],
                                        curdle_rand* rand, uint8_t proof_out[CURDLE_WHISK_TRACKER_PROOF_SIZE]);
/* ------------------------------------------------------------------------- *
 * Accumulator on the device
 *   msmaccumulator.AccumulateCheck / Verify (msmaccumulator/msmaccumulator.go:23-64) with
 *   the base -> scalar map kept on the GPU as an array of scalar slots indexed by base, for
 *   callers whose bases come from sets that stay resident: the CRS (crs.go:10-18: Gs | Hs |
 *   H | Gt | Gu never change) and the per-verification instance points.  Instead of hashing
 *   96-byte keys on the host (msmaccumulator.go:38-43) and re-uploading and re-converting every
 *   base for the final MultiExp (:59), a check names slot ranges of the resident sets and
 *   describes its scalar vector; an Fr kernel evaluates the vectors -- including the
 *   verifier's O(n) products s_i, s'_i (innerproductargument.go:223-234,
 *   samemultiscalarargument.go:267-277) -- weights them with the check's random alpha and
 *   adds them into the slots; the result feeds the MSM's digit kernel directly.
 * ------------------------------------------------------------------------- */
typedef struct curdle_dbases curdle_dbases; /* a base set on the device, in the kernels' internal form */
int curdle_dbases_create(const uint64_t* points /* n x 12, gnark G1Affine */, size_t n, curdle_dbases** out);
void curdle_dbases_free(curdle_dbases* b);
size_t curdle_dbases_size(const curdle_dbases* b);
/* 1 for a live handle: a set keeps its points and re-creates its device copy as needed (another
 * context, a context re-initialised after curdle_shutdown), so it stays usable until it is freed. */
int curdle_dbases_valid(const curdle_dbases* b);

/* The plain MSM over a resident base set: out = sum_{i < n} scalars[i] * bases[i] over the FIRST n
 * points of the set (n <= curdle_dbases_size).  Replaces (*G1Jac).MultiExp where the bases do not
 * change between calls -- msmaccumulator.Verify's (/root/reference/msmaccumulator/msmaccumulator.go:59)
 * are mostly the CRS (crs.go:10-18): SURVEY.md section 8(b)'s "explicit, caller-managed device
 * handle".  Nothing is uploaded or converted per call (the conversion is 0.10 ms and 360 MB of every
 * 2^20-pair call with gnark-layout inputs); the set's copy on the calling thread's context is made on
 * first use.  Same preconditions as curdle_msm_g1 (bases in G1).
 *   _dbases          scalars in DEVICE memory (n x 32 B Montgomery fr.Elements), synchronous
 *   _dbases_host     scalars in host memory: 32 bytes per pair cross PCIe instead of 128
 *   _dbases_windows  the partial over windows [win_begin, win_end) (win_end = -1: all), already
 *                    scaled -- one device's share of a multi-GPU window split whose every device
 *                    holds the set (curdle_msm_g1_device_windows' counterpart)
 *   _dbases_submit   the asynchronous form; curdle_msm_wait(ticket) finishes it.  The set stays
 *                    referenced until then (curdle_dbases_free in between is deferred). */
int curdle_msm_g1_dbases(const curdle_dbases* bases, const void* d_scalars, size_t n, uint64_t out_jac[18]);
int curdle_msm_g1_dbases_host(const curdle_dbases* bases, const uint64_t* scalars, size_t n, uint64_t out_jac[18]);
int curdle_msm_g1_dbases_windows(const curdle_dbases* bases, const void* d_scalars, size_t n, int window_bits,
                                 int win_begin, int win_end, uint64_t out_jac[18]);
int curdle_msm_g1_dbases_submit(const curdle_dbases* bases, const void* d_scalars, size_t n, int window_bits,
                                int win_begin, int win_end, int* ticket);

#define CURDLE_VEC_EXPLICIT 0 /* x_i = tail[i]                                                        */
#define CURDLE_VEC_CONST 1    /* x_i = scale                                                          */
#define CURDLE_VEC_FOLD 2     /* x_i = scale * prod_{j : bit j of i set} gammas[m-1-j]                */
#define CURDLE_VEC_FOLD_POW 3 /* ... * q^(min(i, q_cap) + 1)                                          */
#define CURDLE_DACC_MAX_SEGS 6
#define CURDLE_DACC_MAX_EXTRA 16384
#define CURDLE_SET_CRS 0
#define CURDLE_SET_INST 1
/* One AccumulateCheck: sum_i x_i v_i (== C, which the caller moves to the other side as extra
 * terms).  Offsets index `pool`, an array of Montgomery fr.Elements; the first n_struct
 * elements of x follow the rule of `kind` with weight = alpha * scale folded in by the caller,
 * the n_tail elements after them are explicit and are multiplied by alpha on the device.
 * Segment s says: slots [first, first + len) of resident set `set` take x[vec_first + j]. */
typedef struct {
  uint32_t kind, n_struct, m, q_cap;
  uint32_t weight_off, alpha_off, gammas_off, q_off, tail_off, n_tail;
  uint32_t nseg;
  struct {
    uint32_t set, first, len, vec_first;
  } seg[CURDLE_DACC_MAX_SEGS];
} curdle_dacc_check;

typedef struct curdle_dacc curdle_dacc;
/* Starts an accumulation over `crs` and n_inst instance points (host memory, gnark affine):
 * takes a workspace slot and begins copying / converting the bases, then returns; the caller
 * computes its challenges meanwhile. */
int curdle_dacc_begin(const curdle_dbases* crs, const uint64_t* inst_points, size_t n_inst, curdle_dacc** out);
/* Applies the checks, appends the n_extra loose (point, scalar) pairs and runs the MSM over
 * all slots: out_jac = sum over CRS, instance and extra bases (canonical Jacobian), i.e. what
 * msmaccumulator.Verify compares with A_c.  export_scalars (optional, (n_crs + n_inst) x 4)
 * receives the slot scalars the device built, for parity tests.  Ends the accumulation. */
int curdle_dacc_run(curdle_dacc* acc, const curdle_dacc_check* checks, size_t n_checks, const uint64_t* pool,
                    size_t pool_len, const uint64_t* extra_points, const uint64_t* extra_scalars, size_t n_extra,
                    uint64_t out_jac[CURDLE_G1_JAC_U64], uint64_t* export_scalars);
/* The same in two steps, for a caller with work to do while the MSM runs (batch verification:
 * a worker submits its group of proofs and goes on verifying the next ones): submit copies
 * every argument and queues the kernels, poll says without blocking whether they are done
 * (the accumulation holds one of the eight workspace slots until it is waited for: collect it
 * soon after), wait hands out the result and ends the accumulation.  A failed submit ends it
 * too.  export_scalars, if given to submit, must stay valid until wait. */
int curdle_dacc_submit(curdle_dacc* acc, const curdle_dacc_check* checks, size_t n_checks, const uint64_t* pool,
                       size_t pool_len, const uint64_t* extra_points, const uint64_t* extra_scalars, size_t n_extra,
                       uint64_t* export_scalars);
int curdle_dacc_poll(curdle_dacc* acc, int* done);
int curdle_dacc_wait(curdle_dacc* acc, uint64_t out_jac[CURDLE_G1_JAC_U64]);
void curdle_dacc_abort(curdle_dacc* acc); /* ends an accumulation without its result (submitted or not) */

/* curdleproof.Verify keeps its accumulator on the device by default (the section above);
 * 0 moves it back to the host mirror of msmaccumulator (same accept bit).  Returns the
 * previous setting.  Environment: CURDLE_DEVICE_ACC=0. */
int curdle_verify_set_device_acc(int on);
/* Test hook: the accumulated (base, scalar) list of one verification of a decoded proof,
 * taken before the final MSM from the host mirror (device = 0) or the device accumulator
 * (device = 1), plus the accept bit.  points: cap x 12, scalars: cap x 4 (Montgomery);
 * *n_out entries are written.  Bases may repeat (the device keeps loose points apart). */
int curdle_verify_export_accumulator(const curdle_crs* crs, const curdle_proof* proof, const uint64_t* Rs,
                                     const uint64_t* Ss, const uint64_t* Ts, const uint64_t* Us, size_t ell,
                                     const uint64_t M[CURDLE_G1_JAC_U64], curdle_rand* rand, int device,
                                     uint64_t* points, uint64_t* scalars, size_t cap, size_t* n_out, int* ok);
/* common.IPA (reference common/util.go:26-35): out = sum_i a[i] * b[i] over Fr, Montgomery limbs
 * in and out; CURDLE_EINVAL if the lengths differ (the reference returns an error). */
int curdle_fr_inner_product(const uint64_t* a, size_t a_len, const uint64_t* b, size_t b_len, uint64_t out_fr[4]);
/* pieces exposed for known-answer tests */
int curdle_merlin_test_vector(const char* protocol, const char* label, const uint8_t* msg, size_t msg_len,
                              const char* challenge_label, uint8_t* out, size_t out_len);
int curdle_g1_compress(const uint64_t jac[18], uint8_t out[48]);
int curdle_g1_decompress(const uint8_t in[48], int subgroup_check, uint64_t out_jac[18]);
/* Batched independent scalar multiplications ON THE GPU:
 *   out[i] = addends[i] + scalars[i] * points[i],   i < n
 * points / addends / out gnark affine (addends may be NULL; (0,0) = infinity), scalars Montgomery
 * fr.Elements; n_scalars = n, or 1 for one scalar shared by all points.  Replaces the loops of
 * single ScalarMultiplication calls that dominate the reference's prover: the fold steps
 * G_L[i] += gamma * G_R[i] (innerproductargument.go:155-166, samemultiscalarargument.go:129-135),
 * grandproductargument.go:94-103 and common/util.go:55-63. */
int curdle_g1_scalar_mul_batch(const uint64_t* points, const uint64_t* scalars, size_t n_scalars,
                               const uint64_t* addends, size_t n, uint64_t* out_affine);
/* Batched decoding of gnark's compressed G1 encoding ON THE GPU (square root, curve check,
 * sign selection, and with subgroup_check != 0 the endomorphism subgroup test gnark's Decoder /
 * SetBytes apply -- run BESIDE the square roots, from the records' x alone, on a twisted model
 * of the curve, for batches of up to 32,768 points; behind them in one kernel for larger
 * ones): n x 48 bytes in, n x 12 limbs of gnark affine points out (directly usable as MSM
 * bases), one status byte per point.  Invalid points do not fail the call: they get a
 * non-zero status and (0, 0).  Replaces the per-point
 * G1Affine.SetBytes loops of whisk/types.go:85-95 (4 * ell tracker points per shuffle) and
 * of Proof.FromReader (curdleproof.go:320-356). */
#define CURDLE_DECODE_OK 0
#define CURDLE_DECODE_INFINITY 1        /* valid encoding of the point at infinity; out = (0, 0) */
#define CURDLE_DECODE_BAD_ENCODING 2    /* not the compressed form, x >= p, or stray bits with the infinity flag */
#define CURDLE_DECODE_NOT_ON_CURVE 3
#define CURDLE_DECODE_NOT_IN_SUBGROUP 4
int curdle_g1_decompress_batch(const uint8_t* in, size_t n, int subgroup_check, uint64_t* out_affine, uint8_t* status);
/* The same in two steps, so the caller can work with the points while the subgroup test (the
 * longer of the two chains, on its own stream) is still running: begin decodes -- square root,
 * curve check, sign -- and returns the points with a preliminary status (0, 1, 2 or 3; here
 * points outside the subgroup are still handed out); finish waits for the subgroup test and
 * writes the final status bytes (now possibly 4).  Every begin must be matched by a finish.
 * At most four may be in flight: a fifth begin returns CURDLE_EBUSY (use the one-shot form
 * then). */
int curdle_g1_decompress_begin(const uint8_t* in, size_t n, uint64_t* out_affine, uint8_t* status, int* ticket);
/* begin, split once more: start only launches the two kernels and returns (the caller hashes its
 * transcript from the raw bytes meanwhile); points waits for the square roots and hands back the
 * points; finish as above.  begin = start + points. */
int curdle_g1_decompress_start(const uint8_t* in, size_t n, int* ticket);
int curdle_g1_decompress_points(int ticket, uint64_t* out_affine, uint8_t* status);
int curdle_g1_decompress_finish(int ticket, uint8_t* status);
int curdle_set_last_error(int code, const char* msg);  /* internal: shared by the library's translation units */

/* ------------------------------------------------------------------------- *
 * Profiling and diagnostics (used by bench.py and tests; not part of the
 * reference's surface).
 * ------------------------------------------------------------------------- */
#define CURDLE_PROF_MAX_KERNELS 16 /* phases incl. the "(queue)" pseudo-phase */
typedef struct {
  int n_kernels;
  const char* name[CURDLE_PROF_MAX_KERNELS];
  float ms[CURDLE_PROF_MAX_KERNELS]; /* HIP-event time of each kernel, last MSM call */
  int window_bits;
  int num_windows;
  /* mode 1 only: sorted (pair, window) entries of the last call (zero digits dropped) and the
   * fragments the accumulate kernel emitted -- every fragment's first addition is a copy, so
   * the launch did entries - fragments real mixed additions */
  unsigned long long entries;
  unsigned long long fragments;
} curdle_profile;
/* on = 1: every MSM call brackets each kernel with hipEvents on the stream it launches on
 * and keeps the durations of the last call.  on = 2: only the dominant kernel (the bucket
 * accumulation) is bracketed -- two events instead of ten, which a pipelined caller does
 * not notice.  on = 0: off. */
int curdle_profile_enable(int on);
int curdle_profile_last(curdle_profile* out);

/* Synthetic MSM bases of SURVEY.md section 8(d), generated on the GPU into
 * device memory: P_i = (k + i*q) * G for i < n, gnark G1Affine layout.  k and
 * q are canonical (non-Montgomery) 256-bit integers, little-endian limbs.
 * Because every P_i has a known discrete log, the exact MSM result at any n is
 * (k*sum(s_i) + q*sum(i*s_i)) * G, which is how full-size runs are checked. */
int curdle_synth_points_walk_device(const uint64_t k[4], const uint64_t q[4], size_t n, void* d_out);

/* Element-wise device self-test of the field / curve primitives, so tests can
 * compare the gfx950 arithmetic with the oracle operation by operation.
 *   op 0: fp_mul(a,b)      in: n x (12+12) u32-limbed Fp pairs, out: n x Fp
 *   op 1: fp_add  2: fp_sub  3: fp_sqr(a)
 *   op 4: fr_from_mont(a)  in: n x Fr pairs (b ignored), out: n x Fr
 *   op 5: xyzz madd: in: n x (XYZZ acc | XYZZ whose X,Y hold the affine addend), out: n x XYZZ
 *   op 6: xyzz add : in: n x (XYZZ | XYZZ),            out: n x XYZZ
 *   op 7: xyzz dbl : in: n x (XYZZ | XYZZ ignored),    out: n x XYZZ
 *   op 8 / 9: the same add / dbl through the lane-distributed ("quad") point operations the
 *             latency-bound kernels use (csrc/quad28.h); op 10: k * (second point) with a
 *             20-bit k derived from the element index, by the quads' double-and-add
 *   op 11: the GLV split every scalar goes through (csrc/bls12_381.h glv_split, run by k_digits and by
 *          the host's scalar multiplication): in: n x 8 words of a CANONICAL scalar k < r,
 *          out: n x (|k1| 4 words | k2 4 words | sign of the k1 term | sign of the k2 term,
 *          0x80000000 = negative), with k = +-|k1| +- k2 * lambda (mod r), both halves < 2^127
 *   op 12: the two ends of the curve change the MSM kernels make (csrc/fp28.h from_gnark_iso_x / _y:
 *          a base enters as (x / 16, y / 64) by shifts of its gnark words; to_gnark_msm: a result
 *          leaves through 2^388 / 2^390) on plain field elements: in: n x (Fp | Fp), out: n x
 *          (Fp | Fp | 1 if the x image is normalised and below 2p | the same for y); the round trip
 *          is the identity (the host build copies)
 * (limbs are passed as uint32 little-endian; 24/16/96/8/24 in and 12/8/48/10/26 out per item)
 * on_device = 0 runs the same header code on the host CPU. */
int curdle_selftest_op(int op, const uint64_t* in, size_t n, uint64_t* out, int on_device);
/* Words per item operation `op` reads and writes: the ONE table the entry point above, its launcher
 * and its kernel index (a binding sizes its arrays from this instead of keeping a copy).
 * CURDLE_EINVAL for an op the library does not have. */
int curdle_selftest_shape(int op, uint32_t* in_words, uint32_t* out_words);

#ifdef __cplusplus
}
#endif
#endif /* CURDLE_MSM_H */
