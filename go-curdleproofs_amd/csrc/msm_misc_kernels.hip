// gfx950 kernels beside the MSM: the synthetic bases of SURVEY.md section 8(d) and the self-test operations the
// tests drive the device arithmetic with (curdle_selftest_op).
#include "msm_kernels_common.h"

namespace curdle {

// p - 2 (Fermat inversion exponent), 32-bit words
__constant__ u32 kPminus2[12] = {0xffffaaa9u, 0xb9feffffu, 0xb153ffffu, 0x1eabfffeu, 0xf6b0f624u, 0x6730d2a0u,
                                 0xf38512bfu, 0x64774b84u, 0x434bacd7u, 0x4b1ba7b6u, 0x397fe69au, 0x1a0111eau};

// ---------------------------------------------------------------------------
// Synthetic bases of SURVEY.md section 8(d): P_i = P_0 + i*Q, affine, gnark
// layout.  table[j] = 2^j * Q (affine).  One lane per point: <= 27 mixed adds,
// then one Fermat inversion of ZZ*ZZZ to normalise.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock, 2)
    k_synth_walk(const G1Affine* __restrict__ table, G1Affine p0, u32 n, uint4* __restrict__ out) {
  u32 i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  X28 acc;
  if (g1_affine_is_inf(p0)) {
    d28::set_inf(acc);
  } else {
    d28::from_gnark(acc.x, p0.x.l);
    d28::from_gnark(acc.y, p0.y.l);
    d28::set_one(acc.zz);
    d28::set_one(acc.zzz);
  }
  for (int j = 0; j < 27; j++) {
    if ((i >> j) & 1u) {
      G1Affine t = table[j];
      F28 x, y;
      d28::from_gnark(x, t.x.l);
      d28::from_gnark(y, t.y.l);
      d28::madd(acc, x, y);
    }
  }
  u32 w[24];
  if (d28::is_inf(acc)) {
    for (int k = 0; k < 24; k++) w[k] = 0;
  } else {
    F28 t, inv, izz, izzz, x, y;
    d28::mul(t, acc.zz, acc.zzz);
    d28::set_one(inv);
    for (int b = 383; b >= 0; b--) {
      d28::sqr(inv, inv);
      if ((kPminus2[b >> 5] >> (b & 31)) & 1u) d28::mul(inv, inv, t);
    }
    d28::mul(izz, inv, acc.zzz);
    d28::mul(izzz, inv, acc.zz);
    d28::mul(x, acc.x, izz);
    d28::mul(y, acc.y, izzz);
    d28::to_gnark(w, x);
    d28::to_gnark(w + 12, y);
  }
  d28::store_words<24>(out + (size_t)i * 6, w);
}

hipError_t launch_synth_walk(const G1Affine* d_table, const G1Affine& p0, uint32_t n, void* d_out,
                             hipStream_t stream) {
  hipLaunchKernelGGL(k_synth_walk, dim3(cdiv(n, kBlock)), dim3(kBlock), 0, stream, d_table, p0, n,
                     reinterpret_cast<uint4*>(d_out));
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// Primitive self-test (curdle_selftest_op)
// ---------------------------------------------------------------------------
// All operands and results cross this kernel in gnark form; the operation itself
// runs in the internal radix-2^28 form the MSM kernels use.
// The widths come from kSelftestTable (msm_kernels.h) as arguments; a branch whose own layout does
// not match them returns without touching memory, and so does an unknown op.
__global__ void __launch_bounds__(kBlock, 2)
    k_selftest(int op, const u32* __restrict__ in, size_t n, u32* __restrict__ out, u32 in_w, u32 out_w) {
  size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (op >= 8 && op <= 10) {  // lane-distributed point operations (quad28.h): four lanes per element
    if (in_w != 96 || out_w != 48) return;
    i >>= 2;
    if (i >= n) return;  // whole quads leave together
    G1XYZZ ga, gb;
    const u32* src = in + i * in_w;
    u32* a32 = reinterpret_cast<u32*>(&ga);
    u32* b32 = reinterpret_cast<u32*>(&gb);
    for (int k = 0; k < 48; k++) {
      a32[k] = src[k];
      b32[k] = src[48 + k];
    }
    X28 pa, pb;
    d28::from_gnark(pa, ga);
    d28::from_gnark(pb, gb);
    F28 ca, cb;
    q28::from_x28(ca, pa);
    q28::from_x28(cb, pb);
    if (op == 8) q28::add(ca, cb);
    else if (op == 9) q28::dbl(ca);
    else q28::mul_small(ca, cb, (u32)(i * 2654435761u) >> 12, 19);  // op 10: k * b, k = 20 bits of a hash of i
    q28::to_x28(pa, ca);
    if (threadIdx.x & 3u) return;
    G1XYZZ o;
    d28::to_gnark(o, pa);
    const u32* o32 = reinterpret_cast<const u32*>(&o);
    for (int k = 0; k < 48; k++) out[i * out_w + k] = o32[k];
    return;
  }
  if (i >= n) return;
  if (op == 11) {  // the GLV split exactly as k_digits runs it
    if (in_w != 8 || out_w != 10) return;
    Fr k;
    for (int j = 0; j < 8; j++) k.l[j] = in[i * in_w + j];
    u32 a[4], b[4], sa, sb;
    glv_split(k, a, b, sa, sb);
    for (int j = 0; j < 4; j++) {
      out[i * out_w + j] = a[j];
      out[i * out_w + 4 + j] = b[j];
    }
    out[i * out_w + 8] = sa;
    out[i * out_w + 9] = sb;
  } else if (op == 12) {  // the conversion k_convert_points runs and the exit product of the MSM kernels
    if (in_w != 24 || out_w != 26) return;
    u32 w[24];
    for (int k = 0; k < 24; k++) w[k] = in[i * in_w + k];
    F28 x, y;
    d28::from_gnark_iso_x(x, w);
    d28::from_gnark_iso_y(y, w + 12);
    u32 o[24];
    d28::to_gnark_msm(o, x, 0);
    d28::to_gnark_msm(o + 12, y, 1);
    for (int k = 0; k < 24; k++) out[i * out_w + k] = o[k];
    // what madd asks of an affine operand: normalised limbs, value below 2p
    F28 x2 = x, y2 = y;
    d28::cond_sub_pshl<1>(x2);
    d28::cond_sub_pshl<1>(y2);
    bool same_x = true, same_y = true;
    for (int k = 0; k < d28::N; k++) {
      same_x = same_x && x2.l[k] == x.l[k] && x.l[k] <= d28::MASK;
      same_y = same_y && y2.l[k] == y.l[k] && y.l[k] <= d28::MASK;
    }
    out[i * out_w + 24] = same_x;
    out[i * out_w + 25] = same_y;
  } else if (op >= 0 && op <= 3) {
    if (in_w != 24 || out_w != 12) return;
    u32 w[24];
    for (int k = 0; k < 24; k++) w[k] = in[i * in_w + k];
    F28 a, b, r;
    d28::from_gnark(a, w);
    d28::from_gnark(b, w + 12);
    if (op == 0) d28::mul(r, a, b);
    else if (op == 1) d28::add(r, a, b);
    else if (op == 2) d28::sub<4>(r, a, b);
    else d28::sqr(r, a);
    u32 o[12];
    d28::to_gnark(o, r);
    for (int k = 0; k < 12; k++) out[i * out_w + k] = o[k];
  } else if (op == 4) {
    if (in_w != 16 || out_w != 8) return;
    Fr a, r;
    for (int k = 0; k < 8; k++) a.l[k] = in[i * in_w + k];
    f_from_mont<FrParams>(r, a);
    for (int k = 0; k < 8; k++) out[i * out_w + k] = r.l[k];
  } else if (op >= 5 && op <= 7) {
    if (in_w != 96 || out_w != 48) return;
    G1XYZZ ga, gb;
    const u32* src = in + i * in_w;
    u32* a32 = reinterpret_cast<u32*>(&ga);
    u32* b32 = reinterpret_cast<u32*>(&gb);
    for (int k = 0; k < 48; k++) {
      a32[k] = src[k];
      b32[k] = src[48 + k];
    }
    X28 acc, b;
    d28::from_gnark(acc, ga);
    d28::from_gnark(b, gb);
    // gnark-form infinity is ZZ = 0 (X = Y = one): from_gnark maps 0 -> 0
    if (op == 5) {
      if (!(f_is_zero(gb.x) && f_is_zero(gb.y))) d28::madd(acc, b.x, b.y);
    } else if (op == 6) {
      d28::add(acc, b);
    } else {
      d28::dbl(acc);
    }
    G1XYZZ o;
    d28::to_gnark(o, acc);
    const u32* o32 = reinterpret_cast<const u32*>(&o);
    for (int k = 0; k < 48; k++) out[i * out_w + k] = o32[k];
  }
}

hipError_t launch_selftest(int op, const uint32_t* d_in, size_t n, uint32_t* d_out, hipStream_t stream) {
  if (op < 0 || op >= kSelftestOps || !d_in || !d_out) return hipErrorInvalidValue;
  if (n == 0) return hipSuccess;
  const SelftestOp& t = kSelftestTable[op];
  const size_t lanes = (size_t)t.lanes * n;
  hipLaunchKernelGGL(k_selftest, dim3(cdiv(lanes, kBlock)), dim3(kBlock), 0, stream, op, d_in, n, d_out, t.in_words,
                     t.out_words);
  return hipGetLastError();
}

}  // namespace curdle
