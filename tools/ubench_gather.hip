// Calibration of rocprofv3's FETCH_SIZE for the access pattern of k_accumulate
// (MI355X_MICROARCH.md, HBM: "other access widths are uncalibrated: calibrate on a known byte
// count in your own access pattern"): every lane reads ONE random 128-byte-aligned record of a
// 128 MiB table as 7 x 16 bytes (112 B, the internal affine point), `passes` times over.
// Known byte counts per launch: records x 112 B requested, records x 128 B in whole lines.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench_gather tools/ubench_gather.hip
//   rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -o gather -- tools/ubench_gather
// A second kernel streams the same table coalesced (16 B per lane), the pattern the guide's
// x2 correction is stated for, as the control.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void __launch_bounds__(256) k_gather(const uint4* __restrict__ table, const uint32_t* __restrict__ idx, uint32_t n,
                                               uint32_t* __restrict__ sink) {
  const uint32_t t = blockIdx.x * 256 + threadIdx.x;
  if (t >= n) return;
  const uint4* rec = table + (size_t)idx[t] * 8;  // 128-byte records
  uint32_t acc = 0;
#pragma unroll
  for (int i = 0; i < 7; i++) {
    const uint4 v = rec[i];
    acc ^= v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == 0x12345678u) sink[0] = acc;  // keeps the loads
}

__global__ void __launch_bounds__(256) k_stream(const uint4* __restrict__ table, size_t n16, uint32_t* __restrict__ sink) {
  size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
  uint32_t acc = 0;
  for (; t < n16; t += (size_t)gridDim.x * 256) {
    const uint4 v = table[t];
    acc ^= v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

int main() {
  const uint32_t records = 1u << 20;          // 128 MiB table
  const uint32_t n = 1u << 24;                // gathers per launch (16 per record on average)
  uint4* table;
  uint32_t *idx, *sink;
  CHECK(hipMalloc(&table, (size_t)records * 128));
  CHECK(hipMalloc(&idx, (size_t)n * 4));
  CHECK(hipMalloc(&sink, 4));
  CHECK(hipMemset(table, 1, (size_t)records * 128));
  uint32_t* h = (uint32_t*)malloc((size_t)n * 4);
  uint64_t s = 88172645463325252ull;
  for (uint32_t i = 0; i < n; i++) {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    h[i] = (uint32_t)(s >> 20) & (records - 1);
  }
  CHECK(hipMemcpy(idx, h, (size_t)n * 4, hipMemcpyHostToDevice));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  for (int rep = 0; rep < 3; rep++) {
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_gather, dim3(n / 256), dim3(256), 0, 0, table, idx, n, sink);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("k_gather: %u gathers, requested %.1f MiB (112 B each), whole lines %.1f MiB, index reads %.1f MiB, %.3f ms\n", n,
           n * 112.0 / 1048576, n * 128.0 / 1048576, n * 4.0 / 1048576, ms);
  }
  for (int rep = 0; rep < 3; rep++) {
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_stream, dim3(2048), dim3(256), 0, 0, table, (size_t)records * 8, sink);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("k_stream: %.1f MiB coalesced 16 B/lane, %.3f ms\n", records * 128.0 / 1048576, ms);
  }
  return 0;
}
