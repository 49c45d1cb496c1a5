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

namespace curdle {

static constexpr int kBlock = 256;

__device__ __forceinline__ Fr load_fr(const uint4* pool, u32 off) {
  const uint4 lo = pool[2 * (size_t)off], hi = pool[2 * (size_t)off + 1];
  Fr r;
  r.l[0] = lo.x; r.l[1] = lo.y; r.l[2] = lo.z; r.l[3] = lo.w;
  r.l[4] = hi.x; r.l[5] = hi.y; r.l[6] = hi.z; r.l[7] = hi.w;
  return r;
}

// Round 5: the pool (the checks' Fr constants: weights, alphas, gammas, q, explicit tails) and the check descriptions are
// staged in LDS once per block when they fit (a verification at ell = 252: ~1,000 elements).  Every lane used to read them
// from global memory inside its loops -- a dozen dependent loads per check and slot, the same addresses in every lane --
// and the kernel took 0.050 ms of a 0.83 ms verification (rocprofv3 timeline, gpurun_out/r5w).
static constexpr u32 kDaccLdsBytes = 60 * 1024;
template <bool LDS>
__global__ void __launch_bounds__(kBlock)
    k_dacc_scalars(const curdle_dacc_check* __restrict__ checks_g, u32 n_checks, const uint4* __restrict__ pool_g, u32 pool_len,
                   u32 n_crs, u32 n_inst, uint4* __restrict__ out) {
  extern __shared__ uint4 lds_stage[];
  const uint4* pool = pool_g;
  const curdle_dacc_check* checks = checks_g;
  if constexpr (LDS) {
    const u32 pool_q = 2 * pool_len;                                             // uint4 per pool
    const u32 chk_q = (n_checks * (u32)sizeof(curdle_dacc_check) + 15u) / 16u;   // the host pads the array to 32 bytes
    for (u32 i = threadIdx.x; i < pool_q; i += kBlock) lds_stage[i] = pool_g[i];
    const uint4* cg = reinterpret_cast<const uint4*>(checks_g);
    for (u32 i = threadIdx.x; i < chk_q; i += kBlock) lds_stage[pool_q + i] = cg[i];
    __syncthreads();
    pool = lds_stage;
    checks = reinterpret_cast<const curdle_dacc_check*>(lds_stage + pool_q);
  }
  const u32 slot = blockIdx.x * kBlock + threadIdx.x;
  if (slot >= n_crs + n_inst) return;
  const u32 set = slot < n_crs ? CURDLE_SET_CRS : CURDLE_SET_INST;
  const u32 idx = slot < n_crs ? slot : slot - n_crs;
  Fr acc;
  f_zero(acc);
  for (u32 c = 0; c < n_checks; c++) {
    const curdle_dacc_check& ck = checks[c];
    for (u32 s = 0; s < ck.nseg; s++) {
      if (ck.seg[s].set != set || idx < ck.seg[s].first || idx - ck.seg[s].first >= ck.seg[s].len) continue;
      const u32 i = ck.seg[s].vec_first + (idx - ck.seg[s].first);
      Fr v;
      if (i >= ck.n_struct) {  // explicit element, weighted here
        if (i - ck.n_struct >= ck.n_tail) continue;
        const Fr a = load_fr(pool, ck.alpha_off), t = load_fr(pool, ck.tail_off + (i - ck.n_struct));
        fr_mul(v, a, t);
      } else {
        v = load_fr(pool, ck.weight_off);  // alpha * scale
        if (ck.kind >= CURDLE_VEC_FOLD) {
          for (u32 j = 0; j < ck.m; j++)
            if ((i >> j) & 1u) {
              const Fr g = load_fr(pool, ck.gammas_off + (ck.m - 1 - j));
              fr_mul(v, v, g);
            }
        }
        if (ck.kind == CURDLE_VEC_FOLD_POW) {  // q^(min(i, q_cap) + 1), square and multiply from the top bit
          const u32 e = (i < ck.q_cap ? i : ck.q_cap) + 1u;
          const Fr q = load_fr(pool, ck.q_off);
          Fr p = q;
          for (int bit = 30 - __clz((int)e); bit >= 0; bit--) {
            fr_mul(p, p, p);
            if ((e >> bit) & 1u) fr_mul(p, p, q);
          }
          fr_mul(v, v, p);
        }
      }
      fr_add(acc, acc, v);
    }
  }
  out[2 * (size_t)slot] = make_uint4(acc.l[0], acc.l[1], acc.l[2], acc.l[3]);
  out[2 * (size_t)slot + 1] = make_uint4(acc.l[4], acc.l[5], acc.l[6], acc.l[7]);
}

hipError_t launch_dacc_scalars(const void* d_checks, uint32_t n_checks, const void* d_pool, uint32_t pool_len, uint32_t n_crs,
                               uint32_t n_inst, void* d_out, hipStream_t stream) {
  const uint32_t n = n_crs + n_inst;
  if (n == 0) return hipSuccess;
  const size_t need = (size_t)pool_len * 32 + ((size_t)n_checks * sizeof(curdle_dacc_check) + 15) / 16 * 16;
  const dim3 grid((n + kBlock - 1) / kBlock), block(kBlock);
  if (need <= kDaccLdsBytes)
    hipLaunchKernelGGL(k_dacc_scalars<true>, grid, block, need, stream, reinterpret_cast<const curdle_dacc_check*>(d_checks), n_checks,
                       reinterpret_cast<const uint4*>(d_pool), pool_len, n_crs, n_inst, reinterpret_cast<uint4*>(d_out));
  else
    hipLaunchKernelGGL(k_dacc_scalars<false>, grid, block, 0, stream, reinterpret_cast<const curdle_dacc_check*>(d_checks), n_checks,
                       reinterpret_cast<const uint4*>(d_pool), pool_len, n_crs, n_inst, reinterpret_cast<uint4*>(d_out));
  return hipGetLastError();
}

}  // namespace curdle
