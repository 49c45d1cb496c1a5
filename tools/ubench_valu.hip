// Instruction-throughput microbenchmark for the integer/FP64 VALU ops a 381-bit
// Montgomery multiply can be built from on gfx950.  Prints wave-instructions per
// cycle per SIMD (derived from wall time and the measured shader clock).
//   hipcc --offload-arch=gfx950 -O3 -o ubench_valu ubench_valu.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#include <string>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

constexpr int ITERS = 2000;
constexpr int UNROLL = 16;   // independent chains per lane

#define BODY16(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7) S(8) S(9) S(10) S(11) S(12) S(13) S(14) S(15)

__global__ void k_mad_u64_u32(uint64_t* out, uint32_t a, uint32_t b, unsigned long long* clk) {
  uint64_t acc[UNROLL];
  uint32_t x = a + threadIdx.x, y = b + threadIdx.x;
  for (int i = 0; i < UNROLL; i++) acc[i] = i + threadIdx.x;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < ITERS; it++) {
#define S(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(x), "v"(y) : "vcc");
    BODY16(S)
#undef S
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  uint64_t s = 0;
  for (int i = 0; i < UNROLL; i++) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *clk = t1 - t0;
}

__global__ void k_mad_addc(uint64_t* out, uint32_t a, uint32_t b, unsigned long long* clk) {
  uint64_t acc[UNROLL]; uint32_t hi[UNROLL];
  uint32_t x = a + threadIdx.x, y = b + threadIdx.x;
  for (int i = 0; i < UNROLL; i++) { acc[i] = i + threadIdx.x; hi[i] = 0; }
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < ITERS; it++) {
#define S(i) asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32_e32 %1, vcc, 0, %1, vcc" : "+v"(acc[i]), "+v"(hi[i]) : "v"(x), "v"(y) : "vcc");
    BODY16(S)
#undef S
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  uint64_t s = 0;
  for (int i = 0; i < UNROLL; i++) s += acc[i] + hi[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *clk = t1 - t0;
}

// dependent chain of mad+addc (one accumulator) -> latency
__global__ void k_mad_addc_dep(uint64_t* out, uint32_t a, uint32_t b, unsigned long long* clk) {
  uint64_t acc = threadIdx.x; uint32_t hi = 0;
  uint32_t x = a + threadIdx.x, y = b + threadIdx.x;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < ITERS; it++) {
#define S(i) asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32_e32 %1, vcc, 0, %1, vcc" : "+v"(acc), "+v"(hi) : "v"(x), "v"(y) : "vcc");
    BODY16(S)
#undef S
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc + hi;
  if (threadIdx.x == 0 && blockIdx.x == 0) *clk = t1 - t0;
}

#define SIMPLE_KERNEL(NAME, TYPE, ASMSTR)                                                         \
  __global__ void NAME(uint64_t* out, uint32_t a, uint32_t b, unsigned long long* clk) {          \
    TYPE acc[UNROLL];                                                                             \
    TYPE x = (TYPE)(a + threadIdx.x), y = (TYPE)(b + threadIdx.x);                                \
    for (int i = 0; i < UNROLL; i++) acc[i] = (TYPE)(i + threadIdx.x);                            \
    unsigned long long t0 = __builtin_amdgcn_s_memtime();                                         \
    for (int it = 0; it < ITERS; it++) {                                                          \
      _Pragma("unroll") for (int i = 0; i < UNROLL; i++)                                          \
          asm volatile(ASMSTR : "+v"(acc[i]) : "v"(x), "v"(y));                                   \
    }                                                                                             \
    unsigned long long t1 = __builtin_amdgcn_s_memtime();                                         \
    TYPE s = 0;                                                                                   \
    for (int i = 0; i < UNROLL; i++) s += acc[i];                                                 \
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint64_t)s;                                     \
    if (threadIdx.x == 0 && blockIdx.x == 0) *clk = t1 - t0;                                      \
  }

SIMPLE_KERNEL(k_mul_lo_u32, uint32_t, "v_mul_lo_u32 %0, %1, %0")
SIMPLE_KERNEL(k_mul_hi_u32, uint32_t, "v_mul_hi_u32 %0, %1, %0")
SIMPLE_KERNEL(k_mad_u32_u24, uint32_t, "v_mad_u32_u24 %0, %1, %2, %0")
SIMPLE_KERNEL(k_mul_hi_u32_u24, uint32_t, "v_mul_hi_u32_u24_e32 %0, %1, %0")
SIMPLE_KERNEL(k_add_u32, uint32_t, "v_add_u32_e32 %0, %1, %0")
SIMPLE_KERNEL(k_fma_f64, double, "v_fma_f64 %0, %1, %2, %0")
SIMPLE_KERNEL(k_fma_f32, float, "v_fma_f32 %0, %1, %2, %0")
SIMPLE_KERNEL(k_lshl_add_u64, uint64_t, "v_lshl_add_u64 %0, %1, 0, %0")
SIMPLE_KERNEL(k_mad_i32_i24, uint32_t, "v_mad_i32_i24 %0, %1, %2, %0")
SIMPLE_KERNEL(k_mul_f64, double, "v_mul_f64 %0, %1, %0")
SIMPLE_KERNEL(k_add_f64, double, "v_add_f64 %0, %1, %0")

SIMPLE_KERNEL(k_lshr_b64, uint64_t, "v_lshrrev_b64 %0, 3, %0")
SIMPLE_KERNEL(k_alignbit, uint32_t, "v_alignbit_b32 %0, %1, %0, 28")
SIMPLE_KERNEL(k_and_b32, uint32_t, "v_and_b32_e32 %0, %1, %0")

// a multiply-add and ONE other instruction on an independent register, alternating: what the other
// instruction costs NEXT TO the multiplier (round 4: the accumulation's 880 shifts / masks / q-products)
#define PAIR_KERNEL(NAME, TYPE2, ASM2)                                                            \
  __global__ void NAME(uint64_t* out, uint32_t a, uint32_t b, unsigned long long* clk) {          \
    uint64_t acc[UNROLL];                                                                         \
    TYPE2 oth[UNROLL];                                                                            \
    uint32_t x = a + threadIdx.x, y = b + threadIdx.x;                                            \
    for (int i = 0; i < UNROLL; i++) { acc[i] = i + threadIdx.x; oth[i] = (TYPE2)(i + 3 * threadIdx.x); } \
    unsigned long long t0 = __builtin_amdgcn_s_memtime();                                         \
    for (int it = 0; it < ITERS; it++) {                                                          \
      _Pragma("unroll") for (int i = 0; i < UNROLL; i++)                                          \
          asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\t" ASM2 : "+v"(acc[i]), "+v"(oth[i]) : "v"(x), "v"(y) : "vcc"); \
    }                                                                                             \
    unsigned long long t1 = __builtin_amdgcn_s_memtime();                                         \
    uint64_t s = 0;                                                                               \
    for (int i = 0; i < UNROLL; i++) s += acc[i] + (uint64_t)oth[i];                              \
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                               \
    if (threadIdx.x == 0 && blockIdx.x == 0) *clk = t1 - t0;                                      \
  }
PAIR_KERNEL(k_mac_and, uint32_t, "v_and_b32_e32 %1, %2, %1")
PAIR_KERNEL(k_mac_lshr64, uint64_t, "v_lshrrev_b64 %1, 3, %1")
PAIR_KERNEL(k_mac_mullo, uint32_t, "v_mul_lo_u32 %1, %2, %1")
PAIR_KERNEL(k_mac_alignbit, uint32_t, "v_alignbit_b32 %1, %2, %1, 28")

typedef void (*kern_t)(uint64_t*, uint32_t, uint32_t, unsigned long long*);

int main() {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  printf("device: %s, CUs %d, clockRate %d kHz\n", prop.name, prop.multiProcessorCount, prop.clockRate);
  const int cus = prop.multiProcessorCount;
  uint64_t* d_out; unsigned long long* d_clk;
  CHECK(hipMalloc(&d_out, (size_t)cus * 8 * 1024 * 8));
  CHECK(hipMalloc(&d_clk, 8));
  struct K { const char* name; kern_t k; int ops_per_iter; };
  std::vector<K> ks = {
      {"v_mad_u64_u32", k_mad_u64_u32, UNROLL}, {"mad_u64_u32+addc (pair)", k_mad_addc, UNROLL},
      {"mad+addc dependent chain", k_mad_addc_dep, UNROLL},
      {"v_mul_lo_u32", k_mul_lo_u32, UNROLL}, {"v_mul_hi_u32", k_mul_hi_u32, UNROLL},
      {"v_mad_u32_u24", k_mad_u32_u24, UNROLL}, {"v_mad_i32_i24", k_mad_i32_i24, UNROLL},
      {"v_mul_hi_u32_u24", k_mul_hi_u32_u24, UNROLL},
      {"v_add_u32", k_add_u32, UNROLL}, {"v_lshl_add_u64", k_lshl_add_u64, UNROLL},
      {"v_fma_f32", k_fma_f32, UNROLL}, {"v_fma_f64", k_fma_f64, UNROLL}, {"v_mul_f64", k_mul_f64, UNROLL},
      {"v_add_f64", k_add_f64, UNROLL},
      {"v_lshrrev_b64", k_lshr_b64, UNROLL}, {"v_alignbit_b32", k_alignbit, UNROLL}, {"v_and_b32", k_and_b32, UNROLL},
      {"mad + v_and_b32 (per pair)", k_mac_and, UNROLL}, {"mad + v_lshrrev_b64 (per pair)", k_mac_lshr64, UNROLL},
      {"mad + v_mul_lo_u32 (per pair)", k_mac_mullo, UNROLL}, {"mad + v_alignbit_b32 (per pair)", k_mac_alignbit, UNROLL}};
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int waves_per_simd : {1, 2, 4}) {
    printf("--- %d wave(s) per SIMD, all %d CUs ---\n", waves_per_simd, cus);
    // blocks of 256 threads = 4 waves = 1 wave per SIMD; waves_per_simd blocks per CU
    dim3 grid(cus * waves_per_simd), block(256);
    for (auto& k : ks) {
      hipLaunchKernelGGL(k.k, grid, block, 0, 0, d_out, 3u, 5u, d_clk);  // warm
      CHECK(hipDeviceSynchronize());
      CHECK(hipEventRecord(e0));
      hipLaunchKernelGGL(k.k, grid, block, 0, 0, d_out, 3u, 5u, d_clk);
      CHECK(hipEventRecord(e1));
      CHECK(hipEventSynchronize(e1));
      float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
      unsigned long long clk; CHECK(hipMemcpy(&clk, d_clk, 8, hipMemcpyDeviceToHost));
      double ops_per_wave = (double)ITERS * k.ops_per_iter;
      // s_memtime ticks at 100 MHz on gfx9; derive cycles from wall time instead
      double cyc_per_op_wall = (ms * 1e-3 * 2.4e9) / (ops_per_wave * waves_per_simd);
      double lane_ops_per_s = ops_per_wave * 64.0 * grid.x * 4 / (ms * 1e-3);
      printf("%-28s %8.3f ms  %6.2f cyc/wave-instr/SIMD @2.4GHz  %8.2f Tlane-op/s  (memtime ticks %llu)\n",
             k.name, ms, cyc_per_op_wall, lane_ops_per_s * 1e-12, clk);
    }
  }
  return 0;
}
