// Micro-benchmark behind DESIGN.md section 8.1: what would BATCHED-AFFINE
// bucket accumulation cost on this chip, against the XYZZ mixed addition k_accumulate uses?
//
// An affine addition behind a shared inversion (Montgomery's trick) is 5 products + 1 square
// (prefix product; inverse of the difference; running inverse; lambda; lambda^2; y3) = 2,345
// multiply-adds against the mixed addition's 3,668 -- IF the inversion (a chain of 381 squarings +
// ~190 products that cannot be spread over lanes) is amortised over enough additions, and those
// additions' operands and prefix products have somewhere to live between the two passes.
//
//   k_xyzz     every lane runs `adds` mixed additions d28::madd<true> on registers (the
//              arithmetic of k_accumulate without its gathers): the bar to beat.
//   k_affine   every lane runs `adds` affine additions in batches of B: forward pass (d = x2 -
//              x1, prefix product, stored), ONE inversion per lane and batch (all 64 lanes of a
//              wave invert at once: the same issue slots as one shared inversion per wave),
//              backward pass (inverse of d, lambda, x3, y3, stored).  The 2 B operand points
//              and the B prefix products live in global memory, laid out [i][lane] so that every
//              access is coalesced -- the friendliest possible layout (k_accumulate GATHERS).
//
// Build:  hipcc --offload-arch=gfx950 -O3 -I go-curdleproofs_amd/csrc -o tools/ubench_affine tools/ubench_affine.hip
// Run:    tools/ubench_affine            (prints a table; profiles/r03_batched_affine.txt)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "bls12_381.h"
#include "fp28.h"

using namespace curdle;
using d28::F28;
using d28::X28;

#define CHECK(x)                                                                            \
  do {                                                                                      \
    hipError_t e_ = (x);                                                                    \
    if (e_ != hipSuccess) {                                                                 \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__);         \
      return 1;                                                                             \
    }                                                                                       \
  } while (0)

static constexpr int kBlock = 256;
static constexpr int kLanes = 131072;  // one round of the chip at two waves per SIMD

__constant__ u32 kPm2[12] = {0xffffaaa9u, 0xb9feffffu, 0xb153ffffu, 0x1eabfffeu, 0xf6b0f624u, 0x6730d2a0u,
                             0xf38512bfu, 0x64774b84u, 0x434bacd7u, 0x4b1ba7b6u, 0x397fe69au, 0x1a0111eau};

__device__ __forceinline__ void seed(F28& f, u32 s) {
  for (int i = 0; i < d28::N; i++) {
    s = s * 1664525u + 1013904223u;
    f.l[i] = s & d28::MASK;
  }
  f.l[d28::N - 1] &= 0xfffu;  // < p
}

// [i][lane] layout: element i of every lane side by side
__device__ __forceinline__ void ld(F28& f, const u32* base, size_t i, u32 lane) {
#pragma unroll
  for (int k = 0; k < d28::N; k++) f.l[k] = base[(i * d28::N + k) * (size_t)kLanes + lane];
}
__device__ __forceinline__ void st(u32* base, size_t i, u32 lane, const F28& f) {
#pragma unroll
  for (int k = 0; k < d28::N; k++) base[(i * d28::N + k) * (size_t)kLanes + lane] = f.l[k];
}

__global__ void __launch_bounds__(kBlock, 2) k_xyzz(u32* sink, int adds) {
  const u32 lane = blockIdx.x * kBlock + threadIdx.x;
  X28 acc;
  seed(acc.x, lane * 4 + 1);
  seed(acc.y, lane * 4 + 2);
  d28::set_one(acc.zz);
  d28::set_one(acc.zzz);
  F28 x2, y2, one;
  seed(x2, lane * 4 + 3);
  seed(y2, lane * 4 + 4);
  d28::set_one(one);
  for (int i = 0; i < adds; i++) {
    d28::madd<true>(acc, x2, y2);
    d28::add(x2, x2, one);  // another operand every time
  }
  u32 s = 0;
  for (int k = 0; k < d28::N; k++) s ^= acc.x.l[k] ^ acc.y.l[k] ^ acc.zz.l[k] ^ acc.zzz.l[k];
  sink[lane] = s;
}

// pts: [2 * B][2 coords] operands (x1, y1, x2, y2 per addition i: elements 4 i .. 4 i + 3), pref: [B]
__global__ void __launch_bounds__(kBlock, 2) k_affine(u32* pts, u32* pref, u32* out, u32* sink, int adds, int B) {
  const u32 lane = blockIdx.x * kBlock + threadIdx.x;
  u32 s = 0;
  for (int done = 0; done < adds; done += B) {
    // forward: prefix products of the differences
    F28 run, x1, x2, d;
    d28::set_one(run);
    for (int i = 0; i < B; i++) {
      ld(x1, pts, 4 * (size_t)i, lane);
      ld(x2, pts, 4 * (size_t)i + 2, lane);
      d28::sub_raw<4>(d, x2, x1);
      d28::mul_inl(run, run, d);
      st(pref, (size_t)i, lane, run);
    }
    // one inversion per lane and batch: run^(p - 2)
    F28 inv;
    d28::set_one(inv);
    for (int b = 380; b >= 0; b--) {
      d28::sqr_inl(inv, inv);
      if ((kPm2[b >> 5] >> (b & 31)) & 1u) d28::mul_inl(inv, inv, run);
    }
    // backward: the inverse of every difference, the sum
    for (int i = B - 1; i >= 0; i--) {
      F28 y1, y2, prev, invd, lam, t, x3, y3;
      ld(x1, pts, 4 * (size_t)i, lane);
      ld(y1, pts, 4 * (size_t)i + 1, lane);
      ld(x2, pts, 4 * (size_t)i + 2, lane);
      ld(y2, pts, 4 * (size_t)i + 3, lane);
      if (i > 0) ld(prev, pref, (size_t)i - 1, lane);
      else d28::set_one(prev);
      d28::sub_raw<4>(d, x2, x1);
      d28::mul_inl(invd, inv, prev);   // 1 / d_i
      d28::mul_inl(inv, inv, d);       // inverse of the prefix before it
      d28::sub_raw<4>(t, y2, y1);
      d28::mul_inl(lam, t, invd);
      d28::sqr_inl(t, lam);
      d28::sub<4>(t, t, x1);
      d28::sub<4>(x3, t, x2);
      d28::sub<8>(t, x1, x3);
      d28::mul_inl(y3, lam, t);
      d28::sub<4>(y3, y3, y1);
      st(out, 2 * (size_t)i, lane, x3);
      st(out, 2 * (size_t)i + 1, lane, y3);
      s ^= x3.l[0] ^ y3.l[3];
    }
  }
  sink[lane] = s;
}

__global__ void k_fill(u32* p, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    p[i] = ((u32)i * 2654435761u) & 0x0ffffffu;  // limbs below 2^24: any value below p
}

int main() {
  const int adds = 512;  // additions per lane and kernel
  u32 *sink, *pts, *pref, *out;
  const int maxB = 512;
  CHECK(hipMalloc(&sink, kLanes * 4));
  CHECK(hipMalloc(&pts, (size_t)4 * maxB * d28::N * kLanes * 4));
  CHECK(hipMalloc(&pref, (size_t)maxB * d28::N * kLanes * 4));
  CHECK(hipMalloc(&out, (size_t)2 * maxB * d28::N * kLanes * 4));
  hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, pts, (size_t)4 * maxB * d28::N * kLanes);
  CHECK(hipDeviceSynchronize());
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  auto timed = [&](auto launch) -> float {
    float best = 1e30f;
    for (int rep = 0; rep < 3; rep++) {
      (void)hipEventRecord(e0, 0);
      launch();
      (void)hipEventRecord(e1, 0);
      (void)hipEventSynchronize(e1);
      float ms = 0;
      (void)hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
    }
    return best;
  };
  const float xyzz = timed([&] { hipLaunchKernelGGL(k_xyzz, dim3(kLanes / kBlock), dim3(kBlock), 0, 0, sink, adds); });
  CHECK(hipGetLastError());
  const double madds = 6 * 406 + 2 * 315 + 602, aff = 5 * 406 + 315;
  printf("# %d lanes (one round of the chip, two waves per SIMD), %d additions per lane\n", kLanes, adds);
  printf("# XYZZ mixed addition on registers (d28::madd, %d multiply-adds): %.3f ms = %.2f ns per addition and lane\n", (int)madds,
         xyzz, xyzz * 1e6 / adds);
  printf("# affine addition behind a shared inversion (%d multiply-adds + inversion / B):\n", (int)aff);
  printf("# %6s %10s %14s %12s %16s %18s\n", "B", "ms", "ns/add/lane", "vs XYZZ", "bytes/addition", "scratch KiB/lane");
  for (int B : {8, 16, 32, 64, 128, 256, 512}) {
    const float ms = timed([&] { hipLaunchKernelGGL(k_affine, dim3(kLanes / kBlock), dim3(kBlock), 0, 0, pts, pref, out, sink, adds, B); });
    CHECK(hipGetLastError());
    // forward 2 x 56 read + 56 written; backward 4 x 56 + 56 read, 2 x 56 written
    const int bytes = (2 + 1 + 4 + 1 + 2) * 56;
    printf("  %6d %10.3f %14.2f %11.2fx %16d %18.1f\n", B, ms, ms * 1e6 / adds, xyzz / ms, bytes, B * 7 * 56 / 1024.0);
  }
  printf("# vs XYZZ > 1 means the batched-affine form is faster per addition.  k_accumulate has L = 128 sorted\n"
         "# positions per lane in all, reduced pairwise in 7 rounds of 64, 32, ... 1 additions per lane: B is the\n"
         "# round's size, and every round needs its own inversion.\n");
  return 0;
}
