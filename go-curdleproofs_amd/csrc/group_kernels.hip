// Batched independent group operations: out_i = A_i + s_i * P_i
// for i < n, every i on its own lanes.  This -- not the MSM -- is what dominates the
// reference's prover: the fold steps G_L[i] += gamma * G_R[i] of the inner-product and
// same-multiscalar arguments (innerproductargument.go:155-166,
// samemultiscalarargument.go:129-135), the (ell + 4) base rescalings of the grand-product
// argument (grandproductargument.go:94-103) and ShufflePermuteCommit's 2 ell scalar
// multiplications (common/util.go:55-63): 1,531 scalar multiplications at ell = 252.
//
// One 255-bit double-and-add per point, most significant bit first, on the MSM's XYZZ /
// 14 x 28-bit arithmetic (fp28.h).  The launches are small (n = 4 .. 512 points), so the
// chain length of ONE point is what a caller waits for: four lanes share every point
// operation (quad28.h: 3 and 4 product steps for a doubling / an addition instead of 9 and 14) while the
// launch is at most one round of the chip.  Results leave as XYZZ in gnark's limb format;
// the host normalises the batch with one shared inversion.
#include <hip/hip_runtime.h>
#include <stdlib.h>

#include "fp28.h"
#include "quad28.h"
#include "msm_kernels.h"

namespace curdle {

using d28::F28;
using d28::X28;

static constexpr int kBlock = 256;

// gnark affine point -> internal affine; false for (0, 0) = infinity
__device__ __forceinline__ bool load_affine(F28& x, F28& y, const uint4* __restrict__ points, size_t i) {
  u32 w[24];
  d28::load_words<24>(w, points + i * 6);
  u32 any = 0;
#pragma unroll
  for (int j = 0; j < 24; j++) any |= w[j];
  if (!any) return false;
  d28::from_gnark(x, w);
  d28::from_gnark(y, w + 12);
  return true;
}

// One quad per element (quad28.h): the point lives spread over four lanes.
__global__ void __launch_bounds__(kBlock, 2)
    k_scalar_mul_batch_quad(const uint4* __restrict__ points, const uint4* __restrict__ scalars, u32 shared_scalar,
                            const uint4* __restrict__ addends, u32 n, G1XYZZ* __restrict__ out) {
  const u32 i = (blockIdx.x * kBlock + threadIdx.x) >> 2;
  if (i >= n) return;  // whole quads leave together
  // The scalar through the GLV split (bls12_381.h glv_split): s P = +-k1 P +- k2 phi(P) with
  // 127-bit halves, ONE chain of 127 doublings with at most one addition each -- of +-P, +-phi(P)
  // or their sum -- instead of 255 doublings and additions (1.5 -> 0.9 ms for a batch).
  Fr k;
  {
    const size_t si = shared_scalar ? 0 : i;
    uint4 lo = scalars[2 * si], hi = scalars[2 * si + 1];
    Fr m;
    m.l[0] = lo.x; m.l[1] = lo.y; m.l[2] = lo.z; m.l[3] = lo.w;
    m.l[4] = hi.x; m.l[5] = hi.y; m.l[6] = hi.z; m.l[7] = hi.w;
    f_from_mont<FrParams>(k, m);
  }
  u32 a[4], b[4], neg_a, neg_b;
  glv_split(k, a, b, neg_a, neg_b);
  F28 x, y, p, acc;
  q28::set_inf(acc);
  if (load_affine(x, y, points, i)) {  // s * inf = inf
    F28 yn, z, beta, bx, p1, p2, p3;
    d28::set_zero(z);
    d28::sub<4>(yn, z, y);  // 4p - y
#pragma unroll
    for (int j = 0; j < d28::N; j++) beta.l[j] = d28::kBeta(j);
    d28::mul(bx, x, beta);
    q28::from_affine(p1, x, neg_a ? yn : y);
    q28::from_affine(p2, bx, neg_b ? yn : y);
    p3 = p1;
    q28::add(p3, p2);
    // bit 126 of each half in the top bit of its top word
#pragma unroll
    for (int j = 3; j > 0; j--) {
      a[j] = (a[j] << 1) | (a[j - 1] >> 31);
      b[j] = (b[j] << 1) | (b[j - 1] >> 31);
    }
    a[0] <<= 1;
    b[0] <<= 1;
    for (int bit = 126; bit >= 0; bit--) {
      q28::dbl(acc);
      const bool ba = a[3] >> 31, bb = b[3] >> 31;
      if (ba || bb) {
        q28::sel(p, ba, p1, p2);
        q28::sel(p, ba && bb, p3, p);
        q28::add(acc, p);
      }
#pragma unroll
      for (int j = 3; j > 0; j--) {
        a[j] = (a[j] << 1) | (a[j - 1] >> 31);
        b[j] = (b[j] << 1) | (b[j - 1] >> 31);
      }
      a[0] <<= 1;
      b[0] <<= 1;
    }
  }
  if (addends && load_affine(x, y, addends, i)) {
    q28::from_affine(p, x, y);
    q28::add(acc, p);
  }
  u32 w12[12];
  d28::to_gnark(w12, acc);  // this lane's coordinate; ZZ = 0 (infinity) stays 0
  u32* dst = reinterpret_cast<u32*>(&out[i]) + 12u * q28::role();
#pragma unroll
  for (int j = 0; j < 12; j++) dst[j] = w12[j];
}

hipError_t launch_scalar_mul_batch(const void* points, const void* scalars, int shared_scalar, const void* addends,
                                   uint32_t n, void* out_xyzz, hipStream_t stream) {
  if (n == 0) return hipSuccess;
  // Quads (four lanes per point) at every size.  Until round 3 batches beyond one round of lanes
  // went to a one-lane-per-point build of the same chain: 256 registers, 131 of them spilled, 528
  // bytes of scratch per lane -- measured against the quads at those sizes
  // (tools/bench_quad_vs_lane.py, profiles/r03_quad_vs_lane.txt): 40,000 points 12.7 ms against
  // 11.6 on quads, 65,536 points 27.3 against 18.3, 262,144 points 75.0 against 75.7.  Removed.
  hipLaunchKernelGGL(k_scalar_mul_batch_quad, dim3((unsigned)(((uint64_t)4 * n + kBlock - 1) / kBlock)), dim3(kBlock), 0, stream,
                     (const uint4*)points, (const uint4*)scalars, (u32)shared_scalar, (const uint4*)addends, n,
                     (G1XYZZ*)out_xyzz);
  return hipGetLastError();
}

}  // namespace curdle
