// Bodies of the serial host-side group operations, compiled twice (generic x86-64 and
// BMI2+ADX) -- see host_ops.cpp.  The including file defines CURDLE_ISA_SUFFIX and, for
// the BMI2 build, renames namespace `curdle` so the two builds' inline functions do not
// collide.
//
//  * window combine: out = canonical Jacobian of sum_lw 2^(shift of window lw) * winsums[lw],
//    the last step of the MSM (the Horner pass over the window sums the GPU produced):
//    ~127 doublings (the kernels split every scalar into two 127-bit halves) with a strictly
//    serial dependency, which a CPU core does in ~0.03 ms and a GPU lane would take a
//    millisecond over;
//  * the single-point operations of the protocol layers (one 255-bit scalar multiplication
//    per AccumulateCheck / commitment opening / fold step, point addition, normalisation,
//    the square root of point decompression).
#include "../csrc/host_math.h"

#define CURDLE_CAT_(a, b) a##b
#define CURDLE_CAT(a, b) CURDLE_CAT_(a, b)
#define CURDLE_FN(name) CURDLE_CAT(name, CURDLE_ISA_SUFFIX)

extern "C" void CURDLE_FN(curdle_window_combine)(const void* winsums_xyzz, int nw, const int* dbls, uint64_t out[18]) {
  using namespace curdle;
  const G1XYZZ* ws = static_cast<const G1XYZZ*>(winsums_xyzz);
  G1XYZZ acc;
  g1_set_inf(acc);
  // dbls[lw] = width of the window below lw (lw > 0) or the bit offset of the lowest
  // window computed (lw = 0): acc = 2^dbls[lw] * (acc + ws[lw]), top window first.
  for (int lw = nw - 1; lw >= 0; lw--) {
    g1_add(acc, ws[lw]);
    if (!g1_is_inf(acc))
      for (int k = 0; k < dbls[lw]; k++) g1_dbl(acc);
  }
  g1_to_canonical_jac(out, acc);
}

// r = k * p, k = 8 canonical little-endian 32-bit limbs
extern "C" void CURDLE_FN(curdle_host_scalar_mul)(void* r_xyzz, const void* p_xyzz, const uint32_t* k) {
  using namespace curdle;
  g1_scalar_mul_glv(*static_cast<G1XYZZ*>(r_xyzz), *static_cast<const G1XYZZ*>(p_xyzz), k);
}

// acc += b
extern "C" void CURDLE_FN(curdle_host_add)(void* acc_xyzz, const void* b_xyzz) {
  using namespace curdle;
  g1_add(*static_cast<G1XYZZ*>(acc_xyzz), *static_cast<const G1XYZZ*>(b_xyzz));
}

// XYZZ -> affine, 0 for infinity
extern "C" int CURDLE_FN(curdle_host_to_affine)(void* out_affine, const void* p_xyzz) {
  using namespace curdle;
  return g1_to_affine(*static_cast<G1Affine*>(out_affine), *static_cast<const G1XYZZ*>(p_xyzz)) ? 1 : 0;
}

// r = a^e over Fp, e = 12 little-endian 32-bit limbs
extern "C" void CURDLE_FN(curdle_host_fp_pow)(void* r, const void* a, const uint32_t* e) {
  using namespace curdle;
  fp_pow(*static_cast<Fp*>(r), *static_cast<const Fp*>(a), e, 12);
}

// Montgomery -> canonical residue over Fp (one product with the integer 1)
extern "C" void CURDLE_FN(curdle_host_fp_from_mont)(void* r, const void* a) {
  using namespace curdle;
  Fp one;
  f_zero(one);
  one.l[0] = 1;
  fp_mul(*static_cast<Fp*>(r), *static_cast<const Fp*>(a), one);
}

// projective equality of two XYZZ points
extern "C" int CURDLE_FN(curdle_host_equal)(const void* a_xyzz, const void* b_xyzz) {
  using namespace curdle;
  return g1_equal(*static_cast<const G1XYZZ*>(a_xyzz), *static_cast<const G1XYZZ*>(b_xyzz)) ? 1 : 0;
}

// on-curve point -> is it in the prime-order subgroup (g1_in_subgroup)
extern "C" int CURDLE_FN(curdle_host_in_subgroup)(const void* p_xyzz) {
  using namespace curdle;
  return g1_in_subgroup(*static_cast<const G1XYZZ*>(p_xyzz)) ? 1 : 0;
}

// n XYZZ points -> n affine points with one shared inversion
extern "C" void CURDLE_FN(curdle_host_batch_to_affine)(void* out_affine, const void* in_xyzz, size_t n) {
  using namespace curdle;
  g1_batch_to_affine(static_cast<G1Affine*>(out_affine), static_cast<const G1XYZZ*>(in_xyzz), n);
}

// Fixed-base scalar multiplication (8-bit windows, no doublings) for bases that never change
// -- the CRS points Gsum and Hsum the grand-product verifier rescales in every verification
// (grandproductargument.go:243-246): table[w * 255 + (d - 1)] = d * 2^(8 w) * P, affine, for
// w < 32, d in 1..255; k * P is then at most 32 mixed additions.
extern "C" void CURDLE_FN(curdle_host_fixed_base_table)(void* table_affine, const void* p_affine) {
  using namespace curdle;
  const G1Affine& p = *static_cast<const G1Affine*>(p_affine);
  G1XYZZ* tmp = new G1XYZZ[32 * 255];
  G1XYZZ base;
  g1_from_affine(base, p);
  for (int w = 0; w < 32; w++) {
    G1XYZZ acc = base;
    for (int d = 1; d <= 255; d++) {
      tmp[w * 255 + d - 1] = acc;
      g1_add(acc, base);      // (d + 1) * base; base + base takes the doubling branch
    }
    base = acc;               // 256 * base = 2^(8 (w + 1)) * P
  }
  g1_batch_to_affine(static_cast<G1Affine*>(table_affine), tmp, 32 * 255);
  delete[] tmp;
}

// r = k * P from the table above; k = 8 canonical little-endian 32-bit limbs.
extern "C" void CURDLE_FN(curdle_host_fixed_base_mul)(void* r_xyzz, const void* table_affine, const uint32_t* k) {
  using namespace curdle;
  const G1Affine* tab = static_cast<const G1Affine*>(table_affine);
  G1XYZZ acc;
  g1_set_inf(acc);
  for (int w = 0; w < 32; w++) {
    const uint32_t d = (k[w >> 2] >> (8 * (w & 3))) & 0xffu;
    if (!d) continue;
    const G1Affine& t = tab[w * 255 + d - 1];
    if (!g1_affine_is_inf(t)) g1_madd(acc, t.x, t.y);
  }
  *static_cast<G1XYZZ*>(r_xyzz) = acc;
}
