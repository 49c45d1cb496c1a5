// The rest of the C ABI: synthetic bases, the phase profile, the self-test operations, and the handles of the host
// mirrors of common.Rand and msmaccumulator.  (Part of msm_api.hip until round 6.)
#include "msm_internal.h"

// ---------------------------------------------------------------------------
// Synthetic inputs
// ---------------------------------------------------------------------------
extern "C" int curdle_synth_points_walk_device(const uint64_t k[4], const uint64_t q[4], size_t n, void* d_out) {
  Ctx& cx = cur();
  if (!k || !q || (n && !d_out)) return fail(CURDLE_EINVAL, "null argument");
  if (n > ((size_t)1 << 27)) return fail(CURDLE_EINVAL, "n = %zu exceeds the supported 2^27 points", n);
  if (n == 0) return CURDLE_OK;
  std::lock_guard<std::mutex> g(cx.mu);
  int rc = init_default_locked(cx);
  if (rc) return rc;
  HIP_TRY(hipSetDevice(cx.device));
  G1Affine gen;
  g1_generator(gen);
  G1XYZZ gx, t;
  g1_from_affine(gx, gen);
  G1Affine p0, table[27];
  g1_scalar_mul(t, gx, reinterpret_cast<const u32*>(k), 8);
  g1_to_affine(p0, t);
  g1_scalar_mul(t, gx, reinterpret_cast<const u32*>(q), 8);
  for (int j = 0; j < 27; j++) {
    g1_to_affine(table[j], t);
    g1_dbl(t);
  }
  void* d_table = nullptr;
  HIP_TRY(hipMalloc(&d_table, sizeof(table)));
  HIP_TRY(hipMemcpyAsync(d_table, table, sizeof(table), hipMemcpyHostToDevice, cx.util_stream));
  HIP_TRY(launch_synth_walk((const G1Affine*)d_table, p0, (uint32_t)n, d_out, cx.util_stream));
  HIP_TRY(hipStreamSynchronize(cx.util_stream));
  HIP_TRY(hipFree(d_table));
  return CURDLE_OK;
}

// ---------------------------------------------------------------------------
// Profiling / self-test
// ---------------------------------------------------------------------------
extern "C" int curdle_profile_enable(int on) {
  Ctx& cx = cur();
  std::lock_guard<std::mutex> g(cx.mu);
  cx.profile = on < 0 || on > 2 ? 1 : on;
  return CURDLE_OK;
}

extern "C" int curdle_profile_last(curdle_profile* out) {
  Ctx& cx = cur();
  if (!out) return fail(CURDLE_EINVAL, "null argument");
  std::lock_guard<std::mutex> g(cx.mu);
  *out = cx.last;
  return CURDLE_OK;
}

extern "C" int curdle_selftest_op(int op, const uint64_t* in64, size_t n, uint64_t* out64, int on_device) {
  Ctx& cx = cur();
  if (op < 0 || op >= kSelftestOps || !in64 || !out64) return fail(CURDLE_EINVAL, "bad selftest arguments");
  const uint32_t* in = reinterpret_cast<const uint32_t*>(in64);
  uint32_t* out = reinterpret_cast<uint32_t*>(out64);
  // the widths of every operation live in ONE table (msm_kernels.h), which the launcher and the kernel index too
  const size_t in_w = kSelftestTable[op].in_words, out_w = kSelftestTable[op].out_words;
  if (n > ((size_t)1 << 26)) return fail(CURDLE_EINVAL, "selftest of %zu elements", n);
  if (!on_device) {
    for (size_t i = 0; i < n; i++) {
      const uint32_t* s = in + i * in_w;
      uint32_t* d = out + i * out_w;
      if (op == 11) {  // the GLV split of a canonical scalar: |k1| (4 words), k2 (4 words), the two signs
        Fr k;
        memcpy(&k, s, 32);
        u32 sa = 0, sb = 0;
        glv_split(k, d, d + 4, sa, sb);
        d[8] = sa;
        d[9] = sb;
      } else if (op == 12) {  // into the MSM's curve and back: the identity, both bounds kept
        memcpy(d, s, 96);
        d[24] = d[25] = 1;
      } else if (op <= 3) {
        Fp a, b, r;
        memcpy(&a, s, 48);
        memcpy(&b, s + 12, 48);
        if (op == 0) fp_mul(r, a, b);
        else if (op == 1) fp_add(r, a, b);
        else if (op == 2) fp_sub(r, a, b);
        else fp_sqr(r, a);
        memcpy(d, &r, 48);
      } else if (op == 4) {
        Fr a, r;
        memcpy(&a, s, 32);
        f_from_mont<FrParams>(r, a);
        memcpy(d, &r, 32);
      } else {
        G1XYZZ acc, b;
        memcpy(&acc, s, 192);
        memcpy(&b, s + 48, 192);
        if (op == 5) {
          if (!(f_is_zero(b.x) && f_is_zero(b.y))) g1_madd(acc, b.x, b.y);
        } else if (op == 6 || op == 8) {
          g1_add(acc, b);
        } else if (op == 7 || op == 9) {
          g1_dbl(acc);
        } else {  // op 10: k * b with the kernel's 20-bit k (msm_misc_kernels.hip k_selftest)
          const u32 k = (u32)((u32)i * 2654435761u) >> 12;
          g1_scalar_mul(acc, b, &k, 1);
        }
        memcpy(d, &acc, 192);
      }
    }
    return CURDLE_OK;
  }
  std::lock_guard<std::mutex> g(cx.mu);
  int rc = init_default_locked(cx);
  if (rc) return rc;
  HIP_TRY(hipSetDevice(cx.device));
  if (n == 0) return CURDLE_OK;
  struct DevBuf {  // freed on every way out, after the stream has drained
    void* p = nullptr;
    hipStream_t st;
    explicit DevBuf(hipStream_t s) : st(s) {}
    ~DevBuf() {
      if (p) {
        (void)hipStreamSynchronize(st);
        (void)hipFree(p);
      }
    }
  } d_in(cx.util_stream), d_out(cx.util_stream);
  HIP_TRY(hipMalloc(&d_in.p, n * in_w * 4));
  HIP_TRY(hipMalloc(&d_out.p, n * out_w * 4));
  HIP_TRY(hipMemcpyAsync(d_in.p, in, n * in_w * 4, hipMemcpyHostToDevice, cx.util_stream));
  HIP_TRY(launch_selftest(op, (const uint32_t*)d_in.p, n, (uint32_t*)d_out.p, cx.util_stream));
  HIP_TRY(hipMemcpyAsync(out, d_out.p, n * out_w * 4, hipMemcpyDeviceToHost, cx.util_stream));
  HIP_TRY(hipStreamSynchronize(cx.util_stream));
  return CURDLE_OK;
}

#ifdef CURDLE_TRACE_WAVES
namespace curdle { hipError_t debug_read_wave_trace(unsigned long long* out, size_t words); hipError_t debug_read_wave_clk(unsigned long long* out, size_t words); }
extern "C" int curdle_debug_wave_clk(uint64_t* out, size_t words) {
  return curdle::debug_read_wave_clk((unsigned long long*)out, words) == hipSuccess ? 0 : -1;
}
extern "C" int curdle_debug_wave_trace(uint64_t* out, size_t words) {
  return curdle::debug_read_wave_trace((unsigned long long*)out, words) == hipSuccess ? 0 : -1;
}
#endif
extern "C" int curdle_selftest_shape(int op, uint32_t* in_words, uint32_t* out_words) {
  if (op < 0 || op >= kSelftestOps || !in_words || !out_words) return fail(CURDLE_EINVAL, "selftest op %d outside [0, %d)", op, kSelftestOps);
  *in_words = kSelftestTable[op].in_words;
  *out_words = kSelftestTable[op].out_words;
  return CURDLE_OK;
}

// ---------------------------------------------------------------------------
// common.Rand and msmaccumulator handles
// ---------------------------------------------------------------------------
struct curdle_rand {
  common::Rand r;
  explicit curdle_rand(uint64_t seed) : r(seed) {}
};
struct curdle_acc {
  msmaccumulator::MsmAccumulator a;
};

extern "C" curdle_rand* curdle_rand_new(uint64_t seed) { return new (std::nothrow) curdle_rand(seed); }
extern "C" void curdle_rand_free(curdle_rand* r) { delete r; }

extern "C" int curdle_rand_get_fr(curdle_rand* r, uint64_t out_fr[4]) {
  if (!r || !out_fr) return fail(CURDLE_EINVAL, "null argument");
  Fr f;
  r->r.GetFr(f);
  memcpy(out_fr, &f, 32);
  return CURDLE_OK;
}

extern "C" int curdle_rand_get_g1_affine(curdle_rand* r, uint64_t out_aff[12]) {
  if (!r || !out_aff) return fail(CURDLE_EINVAL, "null argument");
  G1Affine p;
  r->r.GetG1Affine(p);
  memcpy(out_aff, &p, 96);
  return CURDLE_OK;
}

extern "C" int curdle_rand_permutation(curdle_rand* r, size_t n, uint32_t* out) {
  if (!r || (n && !out)) return fail(CURDLE_EINVAL, "null argument");
  std::vector<uint32_t> perm;
  r->r.GeneratePermutation(n, perm);
  if (n) memcpy(out, perm.data(), n * 4);
  return CURDLE_OK;
}

extern "C" curdle_acc* curdle_acc_new(void) { return new (std::nothrow) curdle_acc(); }
extern "C" void curdle_acc_free(curdle_acc* a) { delete a; }

extern "C" int curdle_acc_accumulate_check(curdle_acc* a, const uint64_t C_jac[18], const uint64_t* x, size_t x_len,
                                           const uint64_t* v, size_t v_len, curdle_rand* rand) {
  if (!a || !C_jac || !rand || (x_len && !x) || (v_len && !v)) return fail(CURDLE_EINVAL, "null argument");
  G1Jac C;
  memcpy(&C, C_jac, sizeof(C));
  std::vector<Fr> xs(x_len);
  std::vector<G1Affine> vs(v_len);
  if (x_len) memcpy(xs.data(), x, x_len * 32);
  if (v_len) memcpy(vs.data(), v, v_len * 96);
  msmaccumulator::Status st = a->a.AccumulateCheck(C, xs, vs, &rand->r);
  if (!st.ok) return fail(CURDLE_EINVAL, "%s", st.err.c_str());
  return CURDLE_OK;
}

extern "C" int curdle_acc_accumulate_check_deferred(curdle_acc* a, const uint64_t* c_scalars, const uint64_t* c_points,
                                                    size_t c_len, const uint64_t* x, size_t x_len, const uint64_t* v,
                                                    size_t v_len, curdle_rand* rand) {
  if (!a || !rand || (c_len && (!c_scalars || !c_points)) || (x_len && !x) || (v_len && !v))
    return fail(CURDLE_EINVAL, "null argument");
  std::vector<Fr> cs(c_len), xs(x_len);
  std::vector<G1Affine> cp(c_len), vs(v_len);
  if (c_len) memcpy(cs.data(), c_scalars, c_len * 32);
  if (c_len) memcpy(cp.data(), c_points, c_len * 96);
  if (x_len) memcpy(xs.data(), x, x_len * 32);
  if (v_len) memcpy(vs.data(), v, v_len * 96);
  msmaccumulator::Status st = a->a.AccumulateCheckDeferred(cs, cp, xs, vs, &rand->r);
  if (!st.ok) return fail(CURDLE_EINVAL, "%s", st.err.c_str());
  return CURDLE_OK;
}

extern "C" int curdle_acc_verify(curdle_acc* a, int* ok) {
  if (!a || !ok) return fail(CURDLE_EINVAL, "null argument");
  bool b = false;
  char saved[256];
  msmaccumulator::Status st = a->a.Verify(&b);
  *ok = b ? 1 : 0;
  if (!st.ok) {
    snprintf(saved, sizeof(saved), "%s", st.err.c_str());
    // keep the class of the underlying failure (no device vs HIP error) visible to the caller
    return fail(st.rc ? st.rc : CURDLE_EHIP, "%s", saved);
  }
  return CURDLE_OK;
}

extern "C" int curdle_acc_get_A_c(const curdle_acc* a, uint64_t out_jac[18]) {
  if (!a || !out_jac) return fail(CURDLE_EINVAL, "null argument");
  g1_to_canonical_jac(out_jac, a->a.A_c);
  return CURDLE_OK;
}

extern "C" size_t curdle_acc_num_bases(const curdle_acc* a) { return a ? a->a.NumBases() : 0; }

extern "C" int curdle_acc_export(const curdle_acc* a, uint64_t* points, uint64_t* scalars) {
  if (!a || !points || !scalars) return fail(CURDLE_EINVAL, "null argument");
  size_t n = a->a.NumBases();
  if (n) {
    memcpy(points, a->a.Bases().data(), n * 96);
    memcpy(scalars, a->a.Scalars().data(), n * 32);
  }
  return CURDLE_OK;
}
