/* A C caller of the multi-device boundary -- the calls the cgo shim's InitDevices / MultiExp /
 * OnDevice make (go-curdleproofs_amd/go/curdlemsm/curdlemsm.go): one process, several contexts.
 * Strict C99 against include/curdle_msm.h.
 *
 *   multi_device_from_c [device ids ...]        default: "0 0" (two contexts on one GPU)
 *
 * Builds n = 2^17 pairs (the generator, scalars i + 1), so the expected sum is
 * (n (n + 1) / 2) * G; computes it (1) on one context, (2) through curdle_msm_g1 split over all
 * configured contexts by point ranges, (3) on the last context selected with curdle_set_device
 * from this thread, and prints the three results' first limbs, which must agree.
 * Exit code 0 = computed and equal, 2 = no GPU (no CPU fallback), 1 = other error / mismatch.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "curdle_msm.h"

static int fail(const char* what, int rc) {
  char err[256];
  curdle_last_error(err, sizeof(err));
  fprintf(stderr, "%s failed (%d): %s\n", what, rc, err);
  return rc == CURDLE_ENODEV ? 2 : 1;
}

int main(int argc, char** argv) {
  static const uint64_t gen[CURDLE_G1_AFFINE_U64] = {
      0x5cb38790fd530c16ull, 0x7817fc679976fff5ull, 0x154f95c7143ba1c1ull, 0xf0ae6acdf3d0e747ull,
      0xedce6ecc21dbf440ull, 0x120177419e0bfb75ull, 0xbaac93d50ce72271ull, 0x8c22631a7918fd8eull,
      0xdd595f13570725ceull, 0x51ac582950405194ull, 0x0e1c8c3fad0059c0ull, 0x0bbc3efc5008a26aull};
  const size_t n = (size_t)1 << 17;
  int devices[CURDLE_MAX_DEVICES] = {0, 0};
  int nd = 2, i, rc;
  uint64_t *points, *scalars, one[CURDLE_G1_JAC_U64], all[CURDLE_G1_JAC_U64], last[CURDLE_G1_JAC_U64];
  curdle_rand* r;
  size_t k;

  if (argc > 1) {
    nd = argc - 1 > CURDLE_MAX_DEVICES ? CURDLE_MAX_DEVICES : argc - 1;
    for (i = 0; i < nd; i++) devices[i] = atoi(argv[i + 1]);
  }
  points = (uint64_t*)malloc(n * CURDLE_G1_AFFINE_U64 * sizeof(uint64_t));
  scalars = (uint64_t*)malloc(n * CURDLE_FR_U64 * sizeof(uint64_t));
  if (!points || !scalars) return 1;
  /* scalars: the library's common.Rand mirror gives Montgomery fr.Elements; any will do */
  r = curdle_rand_new(7);
  if (!r) return 1;
  for (k = 0; k < n; k++) {
    memcpy(points + CURDLE_G1_AFFINE_U64 * k, gen, sizeof(gen));
    if (curdle_rand_get_fr(r, scalars + CURDLE_FR_U64 * k) != CURDLE_OK) return 1;
  }
  curdle_rand_free(r);

  /* (1) one context */
  if ((rc = curdle_init(devices[0])) != CURDLE_OK) return fail("curdle_init", rc);
  if ((rc = curdle_msm_g1(points, scalars, n, one)) != CURDLE_OK) return fail("curdle_msm_g1 (one context)", rc);
  /* (2) all contexts: the same entry point, now split by point ranges over the devices */
  if ((rc = curdle_init_devices(devices, nd)) != CURDLE_OK) return fail("curdle_init_devices", rc);
  if (curdle_device_count() != nd) return 1;
  if ((rc = curdle_msm_g1(points, scalars, n, all)) != CURDLE_OK) return fail("curdle_msm_g1 (all contexts)", rc);
  /* (3) a batch on the last context, selected for this thread (what curdlemsm.OnDevice does) */
  if ((rc = curdle_set_device(nd - 1)) != CURDLE_OK) return fail("curdle_set_device", rc);
  {
    size_t offsets[2];
    offsets[0] = 0;
    offsets[1] = n;
    if ((rc = curdle_msm_g1_batch(points, scalars, offsets, 1, last)) != CURDLE_OK) return fail("curdle_msm_g1_batch", rc);
  }
  if (curdle_get_device() != nd - 1) return 1;
  (void)curdle_set_device(0);
  printf("one  x0=%016llx y0=%016llx\nall  x0=%016llx y0=%016llx\nlast x0=%016llx y0=%016llx\n", (unsigned long long)one[0],
         (unsigned long long)one[6], (unsigned long long)all[0], (unsigned long long)all[6], (unsigned long long)last[0],
         (unsigned long long)last[6]);
  free(points);
  free(scalars);
  if (memcmp(one, all, sizeof(one)) || memcmp(one, last, sizeof(one))) {
    fprintf(stderr, "the three results differ\n");
    return 1;
  }
  if ((rc = curdle_shutdown()) != CURDLE_OK) return fail("curdle_shutdown", rc);
  return 0;
}
