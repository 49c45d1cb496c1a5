// LD_PRELOAD shim: time and count the calls the protocol layer makes into the MSM entry
// points and the serial host group operations (they go through the PLT of libcurdlemsm.so).
//   g++ -O2 -shared -fPIC tools/trace_shim.cpp -o gpurun_out/trace_shim.so -ldl
//   CURDLE_TRACE_LIB=$PWD/go-curdleproofs_amd/libcurdlemsm.so LD_PRELOAD=gpurun_out/trace_shim.so python tools/verify_trace.py
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <stddef.h>

#include <atomic>
#include <chrono>
#include <initializer_list>

namespace {
struct Stat {
  const char* name;
  std::atomic<long> calls{0};
  std::atomic<long> ns{0};
  explicit Stat(const char* n) : name(n) {}
};
Stat s_msm("curdle_msm_g1"), s_batch("curdle_msm_g1_batch"), s_smb("curdle_g1_scalar_mul_batch"),
    s_dec("curdle_g1_decompress_batch"), s_smul("curdle_host_scalar_mul"),
    s_add("curdle_host_add"), s_aff("curdle_host_to_affine"), s_pow("curdle_host_fp_pow");
struct Timer {
  Stat& s;
  std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  explicit Timer(Stat& st) : s(st) {}
  ~Timer() {
    s.calls++;
    s.ns += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
  }
};
void report_and_reset(const char* what) {
  fprintf(stderr, "[trace] ---- %s\n", what);
  for (Stat* s : {&s_msm, &s_batch, &s_smb, &s_dec, &s_smul, &s_add, &s_aff, &s_pow}) {
    fprintf(stderr, "[trace] %-24s calls %8ld  total %10.2f ms  avg %9.2f us\n", s->name, s->calls.load(), s->ns / 1e6,
            s->calls ? s->ns / 1e3 / s->calls : 0.0);
    s->calls = 0;
    s->ns = 0;
  }
}
struct Report {
  ~Report() { report_and_reset("at exit"); }
} report;
// the binding dlopens libcurdlemsm.so RTLD_LOCAL, so RTLD_NEXT cannot see it: look the
// real entry points up in the already-loaded library named by CURDLE_TRACE_LIB
template <class F>
F next(const char* name) {
  static void* lib = dlopen(getenv("CURDLE_TRACE_LIB"), RTLD_NOW | RTLD_NOLOAD);
  void* f = lib ? dlsym(lib, name) : nullptr;
  if (!f) {
    fprintf(stderr, "[trace] cannot resolve %s (set CURDLE_TRACE_LIB to the path of libcurdlemsm.so)\n", name);
    abort();
  }
  return reinterpret_cast<F>(f);
}
}  // namespace

extern "C" {
// print the counters accumulated so far under a heading and zero them
void curdle_trace_mark(const char* what) { report_and_reset(what); }

int curdle_msm_g1(const uint64_t* p, const uint64_t* s, size_t n, uint64_t* out) {
  static auto f = next<int (*)(const uint64_t*, const uint64_t*, size_t, uint64_t*)>("curdle_msm_g1");
  Timer t(s_msm);
  return f(p, s, n, out);
}
int curdle_msm_g1_batch(const uint64_t* p, const uint64_t* s, const size_t* off, size_t k, uint64_t* out) {
  static auto f = next<int (*)(const uint64_t*, const uint64_t*, const size_t*, size_t, uint64_t*)>("curdle_msm_g1_batch");
  Timer t(s_batch);
  return f(p, s, off, k, out);
}
int curdle_g1_scalar_mul_batch(const uint64_t* p, const uint64_t* s, size_t ns, const uint64_t* a, size_t n, uint64_t* out) {
  static auto f = next<int (*)(const uint64_t*, const uint64_t*, size_t, const uint64_t*, size_t, uint64_t*)>("curdle_g1_scalar_mul_batch");
  Timer t(s_smb);
  return f(p, s, ns, a, n, out);
}
int curdle_g1_decompress_batch(const uint8_t* in, size_t n, int sub, uint64_t* out, uint8_t* st) {
  static auto f = next<int (*)(const uint8_t*, size_t, int, uint64_t*, uint8_t*)>("curdle_g1_decompress_batch");
  Timer t(s_dec);
  return f(in, n, sub, out, st);
}
void curdle_host_scalar_mul(void* r, const void* p, const uint32_t* k) {
  static auto f = next<void (*)(void*, const void*, const uint32_t*)>("curdle_host_scalar_mul");
  Timer t(s_smul);
  f(r, p, k);
}
void curdle_host_add(void* a, const void* b) {
  static auto f = next<void (*)(void*, const void*)>("curdle_host_add");
  Timer t(s_add);
  f(a, b);
}
int curdle_host_to_affine(void* o, const void* p) {
  static auto f = next<int (*)(void*, const void*)>("curdle_host_to_affine");
  Timer t(s_aff);
  return f(o, p);
}
void curdle_host_fp_pow(void* r, const void* a, const uint32_t* e) {
  static auto f = next<void (*)(void*, const void*, const uint32_t*)>("curdle_host_fp_pow");
  Timer t(s_pow);
  f(r, a, e);
}
}
