/* Minimal C caller of the MSM boundary (the calls cgo generates): compiles as strict C99
 * against include/curdle_msm.h and links with -lcurdlemsm.  Computes the 3-pair MSM
 * G + G + G from the generator in gnark's Montgomery layout and prints the first limbs of
 * the canonical Jacobian result (3*G, Z = Montgomery one).
 *
 *   gcc -std=c99 -Iinclude examples/msm_from_c.c -Lgo-curdleproofs_amd -lcurdlemsm \
 *       -Wl,-rpath,$PWD/go-curdleproofs_amd -o /tmp/msm_from_c && /tmp/msm_from_c
 *
 * Exit code 0 = computed, 2 = no GPU (the library has no CPU fallback), 1 = other error.
 */
#include <stdio.h>
#include <string.h>

#include "curdle_msm.h"

int main(void) {
  /* fr.Element "one" = 2^256 mod r (Montgomery form), little-endian limbs */
  static const uint64_t fr_one[CURDLE_FR_U64] = {0x00000001fffffffeull, 0x5884b7fa00034802ull, 0x998c4fefecbc4ff5ull,
                                                 0x1824b159acc5056full};
  /* the G1 generator as fp.Element limbs in Montgomery form (bls12381.Generators()) */
  static const uint64_t gen[CURDLE_G1_AFFINE_U64] = {
      0x5cb38790fd530c16ull, 0x7817fc679976fff5ull, 0x154f95c7143ba1c1ull, 0xf0ae6acdf3d0e747ull,
      0xedce6ecc21dbf440ull, 0x120177419e0bfb75ull, 0xbaac93d50ce72271ull, 0x8c22631a7918fd8eull,
      0xdd595f13570725ceull, 0x51ac582950405194ull, 0x0e1c8c3fad0059c0ull, 0x0bbc3efc5008a26aull};
  uint64_t points[3 * CURDLE_G1_AFFINE_U64], scalars[3 * CURDLE_FR_U64], out[CURDLE_G1_JAC_U64];
  char err[256];
  int i, rc;

  for (i = 0; i < 3; i++) {
    memcpy(points + CURDLE_G1_AFFINE_U64 * i, gen, sizeof(gen));
    memcpy(scalars + CURDLE_FR_U64 * i, fr_one, sizeof(fr_one));
  }
  rc = curdle_msm_g1(points, scalars, 3, out);
  if (rc != CURDLE_OK) {
    curdle_last_error(err, sizeof(err));
    fprintf(stderr, "curdle_msm_g1 failed (%d): %s\n", rc, err);
    return rc == CURDLE_ENODEV ? 2 : 1;
  }
  printf("3G x0=%016llx y0=%016llx z0=%016llx\n", (unsigned long long)out[0], (unsigned long long)out[6],
         (unsigned long long)out[12]);
  return 0;
}
