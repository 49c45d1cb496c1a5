// Scalar side of the device-resident accumulator (include/curdle_msm.h "Accumulator on the
// device").
//
// The reference's msmaccumulator keeps `map[G1Affine]fr.Element` and does
// `map[v_i] += alpha * x_i` per check (msmaccumulator/msmaccumulator.go:38-43), with the
// verifier's x vectors computed by O(n log n) host loops
// (innerproductargument/innerproductargument.go:223-234,
// samemultiscalarargument/samemultiscalarargument.go:267-277).  Here the map is an array of
// scalar slots indexed by base (CRS slots, then instance slots); ONE lane per slot walks the
// checks, finds the segments that cover its slot, evaluates the element of x it needs from
// the check's description and accumulates -- no hashing, no atomics, and the output array is
// what k_digits reads.  256-bit modular integer arithmetic, latency-trivial (a few hundred
// Fr products per lane).
#include <hip/hip_runtime.h>

#include "../../include/curdle_msm.h"
#include "msm_kernels.h"
#include "dacc_eval.h"

#include <atomic>

namespace curdle {

// The evaluation itself lives in dacc_eval.h (shared with k_dacc_front, msm_sort_kernels.hip).
static constexpr u32 kDaccLdsBudget = 120 * 1024;  // of the CU's 160 KiB (opt-in beyond 64 KiB, per device)

template <bool LDS>
__global__ void __launch_bounds__(dacc::kBlock)
    k_dacc_scalars(const curdle_dacc_check* __restrict__ checks_g, u32 n_checks, const uint4* __restrict__ pool_g, u32 pool_len,
                   u32 n_crs, u32 n_inst, uint4* __restrict__ out) {
  extern __shared__ uint4 lds_stage[];
  const dacc::View vw = dacc::setup(lds_stage, checks_g, n_checks, pool_g, pool_len, LDS, threadIdx.x, dacc::kBlock);
  const u32 slot = blockIdx.x * dacc::kBlock + threadIdx.x;
  if (slot >= n_crs + n_inst) return;
  const Fr acc = dacc::eval_slot(vw, slot, n_crs);
  dacc::store_fr(out, slot, acc);
}

static hipError_t dacc_lds_optin() {
  static std::atomic<uint32_t> done{0};
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  const uint32_t bit = 1u << (dev & 31);
  if (done.load(std::memory_order_acquire) & bit) return hipSuccess;
  e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_dacc_scalars<true>), hipFuncAttributeMaxDynamicSharedMemorySize, kDaccLdsBudget);
  if (e == hipSuccess) done.fetch_or(bit, std::memory_order_release);
  return e;
}

hipError_t launch_dacc_scalars(const void* d_checks, uint32_t n_checks, const void* d_pool, uint32_t pool_len, uint32_t n_crs,
                               uint32_t n_inst, void* d_out, hipStream_t stream) {
  const uint32_t n = n_crs + n_inst;
  if (n == 0) return hipSuccess;
  const size_t need = dacc::lds_bytes(pool_len, n_checks, kDaccLdsBudget);
  const dim3 grid((n + dacc::kBlock - 1) / dacc::kBlock), block(dacc::kBlock);
  if (need) {
    hipError_t e = dacc_lds_optin();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_dacc_scalars<true>, grid, block, need, stream, reinterpret_cast<const curdle_dacc_check*>(d_checks), n_checks,
                       reinterpret_cast<const uint4*>(d_pool), pool_len, n_crs, n_inst, reinterpret_cast<uint4*>(d_out));
  } else {
    hipLaunchKernelGGL(k_dacc_scalars<false>, grid, block, 0, stream, reinterpret_cast<const curdle_dacc_check*>(d_checks), n_checks,
                       reinterpret_cast<const uint4*>(d_pool), pool_len, n_crs, n_inst, reinterpret_cast<uint4*>(d_out));
  }
  return hipGetLastError();
}

}  // namespace curdle
