// Evaluation of the device accumulator's scalar slots (include/curdle_msm.h "Accumulator on the device";
// msmaccumulator/msmaccumulator.go:38-43 with the x vectors of innerproductargument.go:223-234 and
// samemultiscalarargument.go:267-277 described instead of computed on the host).  Shared by k_dacc_scalars
// (dacc_kernels.hip) and k_dacc_front (msm_sort_kernels.hip: the same evaluation feeding the digit recoding directly).
//
// Round 5: a block first stages the pool (the checks' Fr constants) and the check descriptions in LDS when they fit; every
// lane used to read them from global memory inside its loops -- a dozen dependent loads per check and slot, the same
// addresses in every lane: 0.050 -> 0.036-0.038 ms of a 0.83 ms verification (rocprofv3 timelines, gpurun_out/r5w, r5x).
// (Nibble tables of the subset products of the gammas and of the powers of q, built once per block, took the per-lane
// chain from ~16 to ~5 products per check and the kernel nowhere: 0.036 ms, gpurun_out/r5y -- building the tables costs
// what they save at five blocks.  Not kept.)
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/curdle_msm.h"
#include "bls12_381.h"

namespace curdle {
namespace dacc {

static constexpr u32 kBlock = 256;
__device__ __forceinline__ Fr load_fr(const uint4* pool, u32 off) {
  const uint4 lo = pool[2 * (size_t)off], hi = pool[2 * (size_t)off + 1];
  Fr r;
  r.l[0] = lo.x; r.l[1] = lo.y; r.l[2] = lo.z; r.l[3] = lo.w;
  r.l[4] = hi.x; r.l[5] = hi.y; r.l[6] = hi.z; r.l[7] = hi.w;
  return r;
}
__device__ __forceinline__ void store_fr(uint4* pool, u32 off, const Fr& v) {
  pool[2 * (size_t)off] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
  pool[2 * (size_t)off + 1] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
}

// LDS needed to stage (pool_len, n_checks); 0 = does not fit the budget (the walk over global memory).
__host__ __device__ inline size_t lds_bytes(u32 pool_len, u32 n_checks, size_t budget) {
  const size_t chk = ((size_t)n_checks * sizeof(curdle_dacc_check) + 15) / 16 * 16;
  const size_t need = (size_t)pool_len * 32 + chk + 16;
  return need <= budget ? need : 0;
}

// What a block holds after setup(): pointers into LDS (or into global memory when nothing was staged).
struct View {
  const uint4* pool;
  const curdle_dacc_check* checks;
  u32 n_checks;
};

// Block-wide: stage the pool and the checks in `lds` (lds_bytes(..) > 0 bytes of it).  Every thread of the block must
// call it (it synchronises).  tid / nthreads: the calling block's thread index and size.
__device__ __forceinline__ View setup(uint4* lds, const curdle_dacc_check* checks_g, u32 n_checks, const uint4* pool_g, u32 pool_len,
                                      bool staged, u32 tid, u32 nthreads) {
  View v;
  v.pool = pool_g;
  v.checks = checks_g;
  v.n_checks = n_checks;
  if (!staged) return v;
  const u32 pool_q = 2 * pool_len;
  const u32 chk_q = (n_checks * (u32)sizeof(curdle_dacc_check) + 15u) / 16u;  // the job pads the array to 32 bytes
  for (u32 i = tid; i < pool_q; i += nthreads) lds[i] = pool_g[i];
  const uint4* cg = reinterpret_cast<const uint4*>(checks_g);
  for (u32 i = tid; i < chk_q; i += nthreads) lds[pool_q + i] = cg[i];
  __syncthreads();
  v.pool = lds;
  v.checks = reinterpret_cast<const curdle_dacc_check*>(lds + pool_q);
  return v;
}

// The scalar of resident slot `slot` (CRS slots first, then the instance's): sum over the checks that cover it.
__device__ __forceinline__ Fr eval_slot(const View& vw, u32 slot, u32 n_crs) {
  const u32 set = slot < n_crs ? CURDLE_SET_CRS : CURDLE_SET_INST;
  const u32 idx = slot < n_crs ? slot : slot - n_crs;
  Fr acc;
  f_zero(acc);
  for (u32 c = 0; c < vw.n_checks; c++) {
    const curdle_dacc_check& ck = vw.checks[c];
    for (u32 s = 0; s < ck.nseg; s++) {
      if (ck.seg[s].set != set || idx < ck.seg[s].first || idx - ck.seg[s].first >= ck.seg[s].len) continue;
      const u32 i = ck.seg[s].vec_first + (idx - ck.seg[s].first);
      Fr v;
      if (i >= ck.n_struct) {  // explicit element, weighted here
        if (i - ck.n_struct >= ck.n_tail) continue;
        const Fr a = load_fr(vw.pool, ck.alpha_off), t = load_fr(vw.pool, ck.tail_off + (i - ck.n_struct));
        fr_mul(v, a, t);
      } else {
        v = load_fr(vw.pool, ck.weight_off);  // alpha * scale
        if (ck.kind >= CURDLE_VEC_FOLD) {
          for (u32 j = 0; j < ck.m; j++)
            if ((i >> j) & 1u) {
              const Fr g = load_fr(vw.pool, ck.gammas_off + (ck.m - 1 - j));
              fr_mul(v, v, g);
            }
        }
        if (ck.kind == CURDLE_VEC_FOLD_POW) {  // q^(min(i, q_cap) + 1), square and multiply from the top bit
          const u32 e = (i < ck.q_cap ? i : ck.q_cap) + 1u;
          const Fr q = load_fr(vw.pool, ck.q_off);
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
  return acc;
}

}  // namespace dacc
}  // namespace curdle
