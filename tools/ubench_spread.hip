// ONE Montgomery product of the device field (fp28.h: 14 limbs of 28 bits, R = 2^392) spread over the FOUR lanes of a DPP
// quad, against the same product in one lane -- measured as chains of dependent products with one wave per SIMD, which
// is the regime of the latency-bound kernels (bucket reduction, level kernel, Horner kernel).  VERDICT r4 item 2 asked
// for this lever; DESIGN.md section 0.3 says why it is measured here and not built into the kernels.
//
// The spread product.  Both operands and the result are REPLICATED: every lane of the quad holds all 14 limbs.  Lane L
// computes the columns c = 4 s + L of a x b (59 multiply-adds instead of 196: the operand b is first aligned per lane,
// B_L[x] = b[x + L], so that one instruction stream serves four different columns), then the Montgomery reduction runs
// column by column as in fp28.h: the owner of column k (lane k & 3) forms m_k, a DPP move broadcasts it, every lane adds
// m_k p[c - k] into its own columns (49 multiply-adds) and the owner's carry moves one lane on.  The high columns are
// brought back to 14 limbs by one carry-save step (limbs stay below 2^29.1, which the products accept) and replicated
// by DPP adds.  Per product and lane: 122 multiply-adds and ~270 other instructions, against 406 + ~120.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I go-curdleproofs_amd/csrc -o tools/ubench_spread tools/ubench_spread.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <chrono>
#include <vector>

#include "fp28.h"

using namespace curdle;
using d28::F28;
using d28::MASK;
using d28::N;
using d28::N0;

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

template <int S0, int S1, int S2, int S3>
__device__ __forceinline__ u32 qperm(u32 v) {
  constexpr int ctrl = S0 | (S1 << 2) | (S2 << 4) | (S3 << 6);
  u32 r = (u32)__builtin_amdgcn_update_dpp((int)v, (int)v, ctrl, 0xf, 0xf, true);
  asm volatile("" : "+v"(r));  // keep the move a move (quad28.h::perm has the story)
  return r;
}
template <int S>
__device__ __forceinline__ u32 qbcast(u32 v) {
  return qperm<S, S, S, S>(v);
}
template <int S>
__device__ __forceinline__ u32 qbcast_dyn(u32 v) { return qbcast<S>(v); }

// value of lane `o` (0..3, compile time) for all lanes
template <int O>
__device__ __forceinline__ u64 qbcast64(u64 v) {
  const u32 lo = qbcast<O>((u32)v), hi = qbcast<O>((u32)(v >> 32));
  return ((u64)hi << 32) | lo;
}
// every lane takes the value of the lane before it in the quad (lane 0 from lane 3)
__device__ __forceinline__ u64 qprev64(u64 v) {
  const u32 lo = qperm<3, 0, 1, 2>((u32)v), hi = qperm<3, 0, 1, 2>((u32)(v >> 32));
  return ((u64)hi << 32) | lo;
}

// t[x + 3] = src[x + L] for x in [-3, 13], zero outside src's 14 limbs: the lane-aligned copy of a replicated element
__device__ __forceinline__ void align_lane(u32 t[17], const u32* src, u32 L) {
#pragma unroll
  for (int x = -3; x <= 13; x++) {
    u32 c[4];
#pragma unroll
    for (int j = 0; j < 4; j++) c[j] = (x + j >= 0 && x + j < N) ? src[x + j] : 0u;
    t[x + 3] = L == 0 ? c[0] : (L == 1 ? c[1] : (L == 2 ? c[2] : c[3]));
  }
}

// r = a b / 2^392 mod p over the four lanes of a quad; a, b, r replicated; limbs of r below 2^29.1, r < 2p for a, b < 2p.
__device__ __forceinline__ void mul_spread(F28& r, const F28& a, const F28& b, const u32 L, const u32 PL[17]) {
  u32 BL[17];
  align_lane(BL, b.l, L);
  u64 col[7];
  // phase A: this lane's columns of a x b
  static_for<0, 7>([&](auto sc) {
    constexpr int s = decltype(sc)::value;
    u64 acc = 0;
    static_for<0, N>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      constexpr int x = 4 * s - i;
      if constexpr (x >= -3 && x <= 13) acc += (u64)a.l[i] * BL[x + 3];
    });
    col[s] = acc;
  });
  // phase B: Montgomery, column by column
  u64 carry = 0;
  static_for<0, N>([&](auto kc) {
    constexpr int k = decltype(kc)::value;
    constexpr int o = k & 3, so = k >> 2;
    if constexpr (k > 0) {
      const u64 cin = qprev64(carry);  // the owner of column k - 1 sits one lane back
      if (L == (u32)o) col[so] += cin;
    }
    const u32 mine = ((u32)col[so] * N0) & MASK;
    const u32 m = qbcast<o>(mine);
    static_for<0, 7>([&](auto sc) {
      constexpr int s = decltype(sc)::value;
      constexpr int x = 4 * s - k;  // this lane's column 4 s + L takes m p[x + L]
      if constexpr (x >= -3 && x <= 13) col[s] += (u64)m * PL[x + 3];
    });
    carry = col[so] >> 28;  // meaningful in the owner lane only: its low 28 bits are zero now
  });
  {
    const u64 cin = qprev64(carry);  // column 13 (lane 1) -> column 14 (lane 2, s = 3)
    if (L == 2u) col[3] += cin;
  }
  // phase C: columns 14 .. 27 -> limbs 0 .. 13 by one carry-save step, replicated
  u32 lo[7], mid[7], hi[7];
#pragma unroll
  for (int s = 0; s < 7; s++) {
    lo[s] = (u32)col[s] & MASK;
    mid[s] = (u32)(col[s] >> 28) & MASK;
    hi[s] = (u32)(col[s] >> 56);
  }
  static_for<0, N>([&](auto jc) {
    constexpr int j = decltype(jc)::value;
    constexpr int c0 = 14 + j, c1 = 13 + j, c2 = 12 + j;  // columns of the low, middle and high part of limb j
    u32 v = qbcast<(c0 & 3)>(lo[c0 >> 2]);
    if constexpr (c1 >= 14) v += qbcast<(c1 & 3)>(mid[c1 >> 2]);
    if constexpr (c2 >= 14) v += qbcast<(c2 & 3)>(hi[c2 >> 2]);
    r.l[j] = v;
  });
}

// ---------------------------------------------------------------------------------------------------------------------
// The same product on ONE lane with TWO column accumulators (round 5, late): fp28.h's product is one dependency chain -- every
// multiply-add of a column adds into the same 64-bit accumulator, and the columns follow each other through the carry -- so a
// wave that is alone on its SIMD waits out every multiply-add's latency.  Here the a x b part of column k + 1 is summed into a
// second accumulator while column k finishes (its m x p part, m_k, the carry): two independent chains the scheduler can
// interleave.  Plain C (the compiler picks v_mad_u64_u32 and places the instructions itself).  Same result bits.
__device__ __forceinline__ void mul_ilp(F28& r, const F28& a, const F28& b) {
  u32 p[N], m[N], t[N];
#pragma unroll
  for (int i = 0; i < N; i++) p[i] = d28::kP(i);
  u64 A = 0, B = (u64)a.l[0] * b.l[0];
  static_for<0, 2 * N - 1>([&](auto kc) {
    constexpr int k = decltype(kc)::value;
    A += B;  // the a x b sum of column k
    // column k + 1's a x b sum: nothing of it depends on column k
    u64 Bn = 0;
    if constexpr (k + 1 <= 2 * N - 2) {
      constexpr int lo = k + 1 < N ? 0 : k + 1 - N + 1;
      constexpr int hi = k + 1 < N ? k + 1 : N - 1;
      static_for<lo, hi + 1>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        Bn += (u64)a.l[i] * b.l[k + 1 - i];
      });
    }
    if constexpr (k < N) {
      static_for<0, k>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        A += (u64)m[i] * p[k - i];
      });
      m[k] = ((u32)A * N0) & MASK;
      A += (u64)m[k] * p[0];
    } else {
      constexpr int i0 = k - N + 1;
      static_for<i0, N>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        A += (u64)m[i] * p[k - i];
      });
      t[k - N] = (u32)A & MASK;
    }
    A >>= 28;
    B = Bn;
  });
  t[N - 1] = (u32)A;
#pragma unroll
  for (int i = 0; i < N; i++) r.l[i] = t[i];
}

static constexpr int kChain = 96;

__device__ __forceinline__ void seed(F28& a, F28& b, u32 id) {
  // two field elements below 2p with normalised limbs, different per chain
#pragma unroll
  for (int i = 0; i < N; i++) {
    a.l[i] = (id * 2654435761u + i * 40503u + 12345u) & MASK;
    b.l[i] = (id * 2246822519u + i * 3266489917u + 777u) & MASK;
  }
  a.l[N - 1] &= 0xfffu;  // below 2^376 < p
  b.l[N - 1] &= 0xfffu;
}
__device__ __forceinline__ void finish(u32* out, F28 r) {
  d28::norm(r);
  d28::canonical_lt2p(r);
#pragma unroll
  for (int i = 0; i < N; i++) out[i] = r.l[i];
}

// one chain per LANE (what the quad kernels' product steps do today, one product per lane)
__global__ void __launch_bounds__(256) k_chain_lane(u32* out, int chains, unsigned long long* clk) {
  const u32 t = blockIdx.x * 256 + threadIdx.x;
  const u32 id = t >> 2;  // the same chain in the four lanes of a quad, so that both kernels compute the same set
  F28 a, b, r;
  seed(a, b, id);
  r = a;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < kChain; it++) d28::mul_inl(r, r, b);
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if ((t & 3u) == 0 && id < (u32)chains) finish(out + (size_t)id * N, r);
  if (t == 0) *clk = t1 - t0;
}

// one chain per lane, the two-accumulator product
__global__ void __launch_bounds__(256) k_chain_ilp(u32* out, int chains, unsigned long long* clk) {
  const u32 t = blockIdx.x * 256 + threadIdx.x;
  const u32 id = t >> 2;
  F28 a, b, r;
  seed(a, b, id);
  r = a;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < kChain; it++) mul_ilp(r, r, b);
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if ((t & 3u) == 0 && id < (u32)chains) finish(out + (size_t)id * N, r);
  if (t == 0) *clk = t1 - t0;
}

// one chain per QUAD, every product spread over its four lanes
__global__ void __launch_bounds__(256) k_chain_spread(u32* out, int chains, unsigned long long* clk) {
  const u32 t = blockIdx.x * 256 + threadIdx.x;
  const u32 id = t >> 2, L = t & 3u;
  F28 a, b, r;
  seed(a, b, id);
  r = a;
  u32 p[N], PL[17];
#pragma unroll
  for (int i = 0; i < N; i++) p[i] = d28::kP(i);
  align_lane(PL, p, L);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < kChain; it++) mul_spread(r, r, b, L, PL);
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (L == 0 && id < (u32)chains) finish(out + (size_t)id * N, r);
  if (t == 0) *clk = t1 - t0;
}

int main() {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  for (int waves_per_simd = 1; waves_per_simd <= 2; waves_per_simd++) {
    const int blocks = cus * waves_per_simd;  // 256 threads = one wave on each of a CU's four SIMDs
    const int chains = blocks * 64;
    u32 *d_a, *d_b;
    unsigned long long* d_clk;
    CHECK(hipMalloc(&d_a, (size_t)chains * N * 4));
    CHECK(hipMalloc(&d_b, (size_t)chains * N * 4));
    CHECK(hipMalloc(&d_clk, 8));
    double ms[3] = {0, 0, 0};
    unsigned long long ticks[3] = {0, 0, 0};
    u32* d_c;
    CHECK(hipMalloc(&d_c, (size_t)chains * N * 4));
    for (int which = 0; which < 3; which++) {
      for (int rep = 0; rep < 4; rep++) {
        CHECK(hipDeviceSynchronize());
        auto t0 = std::chrono::steady_clock::now();
        if (which == 0)
          hipLaunchKernelGGL(k_chain_lane, dim3(blocks), dim3(256), 0, 0, d_a, chains, d_clk);
        else if (which == 1)
          hipLaunchKernelGGL(k_chain_spread, dim3(blocks), dim3(256), 0, 0, d_b, chains, d_clk);
        else
          hipLaunchKernelGGL(k_chain_ilp, dim3(blocks), dim3(256), 0, 0, d_c, chains, d_clk);
        CHECK(hipDeviceSynchronize());
        auto t1 = std::chrono::steady_clock::now();
        ms[which] = std::chrono::duration<double, std::milli>(t1 - t0).count();
      }
      CHECK(hipMemcpy(&ticks[which], d_clk, 8, hipMemcpyDeviceToHost));
    }
    std::vector<u32> ha((size_t)chains * N), hb((size_t)chains * N);
    CHECK(hipMemcpy(ha.data(), d_a, ha.size() * 4, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(hb.data(), d_b, hb.size() * 4, hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (size_t i = 0; i < ha.size(); i++) bad += ha[i] != hb[i];
    std::vector<u32> hc((size_t)chains * N);
    CHECK(hipMemcpy(hc.data(), d_c, hc.size() * 4, hipMemcpyDeviceToHost));
    size_t bad_c = 0;
    for (size_t i = 0; i < ha.size(); i++) bad_c += ha[i] != hc[i];
    printf("%d wave(s) per SIMD: two-accumulator product on one lane %.3f ms = %.2f us per product (ratio to the one-chain product %.2f); results %s (%zu limbs differ)\n",
           waves_per_simd, ms[2], ms[2] * 1e3 / kChain, ms[0] / ms[2], bad_c ? "DIFFER" : "equal", bad_c);
    CHECK(hipFree(d_c));
    // s_memtime counts at 100 MHz on this chip's constant clock?  report wall time per product and the shader-clock estimate
    printf("%d wave(s) per SIMD, %d chains of %d dependent products: one lane per product %.3f ms = %.2f us per product; four lanes per product %.3f ms = %.2f us per product; ratio %.2f; results %s (%zu limbs differ)\n",
           waves_per_simd, chains, kChain, ms[0], ms[0] * 1e3 / kChain, ms[1], ms[1] * 1e3 / kChain, ms[0] / ms[1], bad ? "DIFFER" : "equal", bad);
    printf("   s_memtime ticks of wave 0: %llu / %llu\n", ticks[0], ticks[1]);
    CHECK(hipFree(d_a));
    CHECK(hipFree(d_b));
    CHECK(hipFree(d_clk));
  }
  return 0;
}
