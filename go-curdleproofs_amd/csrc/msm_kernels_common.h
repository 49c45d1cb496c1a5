// Shared by the MSM kernel translation units (msm_sort_kernels.hip, msm_accumulate_kernel.hip, msm_reduce_kernels.hip,
// msm_misc_kernels.hip -- one file, msm_kernels.hip, until round 6: the build's longest pole at 35 s): the block size, wave
// priorities, the record addressing of the converted points and the 256-thread block scan.  Internal; the launchers'
// prototypes are in msm_kernels.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>

#include "msm_kernels.h"

#include "fp28.h"
#include "quad28.h"

namespace curdle {


using d28::A28;
using d28::F28;
using d28::X28;

static constexpr int kBlock = 256;
// s_getreg_b32 operands: (size - 1) << 11 | offset << 6 | register id.  HW_ID (4): wave slot 3:0, SIMD 5:4, CU 11:8,
// SH 12, SE 15:13; XCC_ID (20): the XCD in 3:0.
static constexpr int kGetregHwId = ((32 - 1) << 11) | 4;
[[maybe_unused]] static constexpr int kGetregXccId = ((32 - 1) << 11) | 20;  // (the wave-trace experiment build only)
// s_setprio takes an immediate
__device__ __forceinline__ void set_wave_prio(u32 v) {
  if (v == 1) __builtin_amdgcn_s_setprio(1);
  else if (v == 2) __builtin_amdgcn_s_setprio(2);
  else if (v == 3) __builtin_amdgcn_s_setprio(3);
}

static inline u32 cdiv(u64 a, u32 b) { return (u32)((a + b - 1) / b); }

// element i of the internal point array (kA28Bytes apart: one 128-byte line per point, so a
// gathered point never straddles two lines)
__device__ __forceinline__ A28* a28_at(A28* base, size_t i) {
  return reinterpret_cast<A28*>(reinterpret_cast<char*>(base) + i * kA28Bytes);
}
__device__ __forceinline__ const A28* a28_at(const A28* base, size_t i) {
  return reinterpret_cast<const A28*>(reinterpret_cast<const char*>(base) + i * kA28Bytes);
}

// exclusive scan over the 256 threads of a block (sh: 4 words of LDS); total = the block's sum
__device__ __forceinline__ u32 block_exclusive_scan_256(u32 v, u32* sh /* [4] */, u32& total) {
  const u32 lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
  u32 inc = v;
#pragma unroll
  for (u32 off = 1; off < 64; off <<= 1) {
    const u32 t = __shfl_up(inc, off, 64);
    if (lane >= off) inc += t;
  }
  __syncthreads();  // sh may still be read from the previous call
  if (lane == 63) sh[wv] = inc;
  __syncthreads();
  u32 before = 0, all = 0;
#pragma unroll
  for (u32 k = 0; k < kBlock / 64; k++) {
    const u32 t = sh[k];
    if (k < wv) before += t;
    all += t;
  }
  total = all;
  return before + inc - v;
}

}  // namespace curdle
