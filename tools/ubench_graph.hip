// Does a HIP graph shorten the launch-bound front of a small synchronous call?  Nine dependent kernels of ~5 us each
// (the shape of a 1,268-pair MSM: profiles/r05_small_call.txt), launched one by one on a stream against one
// hipGraphLaunch of the captured sequence; wall time from the first launch to the end of hipStreamSynchronize.
//   hipcc --offload-arch=gfx950 -O3 -o ubench_graph ubench_graph.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
#include <vector>
#include <algorithm>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

struct Big { unsigned v[120]; };  // a by-value argument the size of MsmPlan

__global__ void k_step(unsigned* p, Big b, int spin) {
  unsigned x = p[threadIdx.x & 63] + b.v[threadIdx.x % 120];
  for (int i = 0; i < spin; i++) x = x * 1664525u + 1013904223u;
  if (x == 42u) p[0] = x;
}

int main() {
  unsigned* d;
  CHECK(hipMalloc(&d, 4096));
  CHECK(hipMemset(d, 0, 4096));
  hipStream_t st;
  CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  Big b = {};
  const int K = 9, spin = 600;  // ~5 us per kernel
  auto run_stream = [&]() {
    for (int k = 0; k < K; k++) hipLaunchKernelGGL(k_step, dim3(64), dim3(256), 0, st, d, b, spin);
    return hipStreamSynchronize(st);
  };
  hipGraph_t g;
  hipGraphExec_t ge;
  CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
  for (int k = 0; k < K; k++) hipLaunchKernelGGL(k_step, dim3(64), dim3(256), 0, st, d, b, spin);
  CHECK(hipStreamEndCapture(st, &g));
  CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  auto run_graph = [&]() {
    hipError_t e = hipGraphLaunch(ge, st);
    if (e != hipSuccess) return e;
    return hipStreamSynchronize(st);
  };
  for (int rep = 0; rep < 3; rep++) {
    for (int which = 0; which < 2; which++) {
      std::vector<double> t;
      for (int i = 0; i < 220; i++) {
        auto t0 = std::chrono::steady_clock::now();
        CHECK(which ? run_graph() : run_stream());
        auto t1 = std::chrono::steady_clock::now();
        if (i >= 20) t.push_back(std::chrono::duration<double, std::micro>(t1 - t0).count());
      }
      std::sort(t.begin(), t.end());
      printf("%s: median %.1f us, min %.1f us  (%d dependent kernels)\n", which ? "one hipGraphLaunch " : "nine stream launches", t[t.size() / 2], t[0], K);
    }
  }
  return 0;
}
