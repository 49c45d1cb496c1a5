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
// operation (quad_dbl / quad_add: 3 and 4 product steps instead of 9 and 14) while the
// launch is at most one round of the chip.  Results leave as XYZZ in gnark's limb format;
// the host normalises the batch with one shared inversion.
#include <hip/hip_runtime.h>

#include "fp28.h"
#include "msm_kernels.h"

namespace curdle {

using d28::F28;
using d28::X28;

static constexpr int kBlock = 256;

template <bool QUAD>
__global__ void __launch_bounds__(kBlock, 2)
    k_scalar_mul_batch(const uint4* __restrict__ points, const uint4* __restrict__ scalars, u32 shared_scalar,
                       const uint4* __restrict__ addends, u32 n, G1XYZZ* __restrict__ out) {
  const u32 lane = blockIdx.x * kBlock + threadIdx.x;
  const u32 i = QUAD ? lane >> 2 : lane;
  if (i >= n) return;  // whole quads leave together

  // the scalar as an integer, consumed from the top by shifting the eight words left
  const size_t si = shared_scalar ? 0 : i;
  uint4 lo = scalars[2 * si], hi = scalars[2 * si + 1];
  Fr m, k;
  m.l[0] = lo.x; m.l[1] = lo.y; m.l[2] = lo.z; m.l[3] = lo.w;
  m.l[4] = hi.x; m.l[5] = hi.y; m.l[6] = hi.z; m.l[7] = hi.w;
  f_from_mont<FrParams>(k, m);

  u32 w[24];
  d28::load_words<24>(w, points + (size_t)i * 6);
  u32 any = 0;
#pragma unroll
  for (int j = 0; j < 24; j++) any |= w[j];
  X28 p, acc;
  d28::set_inf(acc);
  if (any) {  // (0, 0) is gnark's point at infinity: s * inf = inf
    d28::from_gnark(p.x, w);
    d28::from_gnark(p.y, w + 12);
    d28::set_one(p.zz);
    d28::set_one(p.zzz);
    // r < 2^255: bit 255 is never set, start at 254
#pragma unroll
    for (int j = 7; j > 0; j--) k.l[j] = (k.l[j] << 1) | (k.l[j - 1] >> 31);
    k.l[0] <<= 1;
    for (int bit = 254; bit >= 0; bit--) {
      if constexpr (QUAD) d28::quad_dbl(acc); else d28::dbl(acc);
      if (k.l[7] >> 31) {
        if constexpr (QUAD) d28::quad_add(acc, p); else d28::madd(acc, p.x, p.y);
      }
#pragma unroll
      for (int j = 7; j > 0; j--) k.l[j] = (k.l[j] << 1) | (k.l[j - 1] >> 31);
      k.l[0] <<= 1;
    }
  }
  if (addends) {
    d28::load_words<24>(w, addends + (size_t)i * 6);
    any = 0;
#pragma unroll
    for (int j = 0; j < 24; j++) any |= w[j];
    if (any) {
      d28::from_gnark(p.x, w);
      d28::from_gnark(p.y, w + 12);
      d28::set_one(p.zz);
      d28::set_one(p.zzz);
      if constexpr (QUAD) d28::quad_add(acc, p); else d28::madd(acc, p.x, p.y);
    }
  }
  if (QUAD && (threadIdx.x & 3u)) return;
  G1XYZZ o;
  if (d28::is_inf(acc)) {
    u32* z = reinterpret_cast<u32*>(&o);
#pragma unroll
    for (int j = 0; j < 48; j++) z[j] = 0;  // ZZ = 0
  } else {
    d28::to_gnark(o, acc);
  }
  u32* dst = reinterpret_cast<u32*>(&out[i]);
  const u32* src = reinterpret_cast<const u32*>(&o);
#pragma unroll
  for (int j = 0; j < 48; j++) dst[j] = src[j];
}

hipError_t launch_scalar_mul_batch(const void* points, const void* scalars, int shared_scalar, const void* addends,
                                   uint32_t n, void* out_xyzz, hipStream_t stream) {
  if (n == 0) return hipSuccess;
  if ((uint64_t)n * 4 <= 131072)
    hipLaunchKernelGGL(k_scalar_mul_batch<true>, dim3((4 * n + kBlock - 1) / kBlock), dim3(kBlock), 0, stream,
                       (const uint4*)points, (const uint4*)scalars, (u32)shared_scalar, (const uint4*)addends, n,
                       (G1XYZZ*)out_xyzz);
  else
    hipLaunchKernelGGL(k_scalar_mul_batch<false>, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, stream,
                       (const uint4*)points, (const uint4*)scalars, (u32)shared_scalar, (const uint4*)addends, n,
                       (G1XYZZ*)out_xyzz);
  return hipGetLastError();
}

}  // namespace curdle
