// Body of the host-side window combine, compiled twice (generic x86-64 and
// BMI2+ADX) -- see window_combine.cpp.  The including file defines
// CURDLE_COMBINE_NAME and, for the BMI2 build, renames namespace `curdle` so the
// two builds' inline functions do not collide.
//
// out = canonical Jacobian of  sum_lw 2^(shift of window lw) * winsums[lw].
// This is the last step of the MSM (the Horner pass over the window sums the
// GPU produced): ~255 doublings with a strictly serial dependency, which a CPU
// core does in ~0.1 ms and a GPU lane would take milliseconds over.
#include "../csrc/host_math.h"

extern "C" void CURDLE_COMBINE_NAME(const void* winsums_xyzz, int nw, const int* dbls, uint64_t out[18]) {
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
