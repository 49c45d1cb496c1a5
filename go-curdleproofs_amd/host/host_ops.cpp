// Generic build of the serial host-side group operations + run-time dispatch to the
// BMI2 + ADX build (host_ops_bmi2.cpp) when the CPU has both.
#define CURDLE_ISA_SUFFIX _generic
#include "host_ops_impl.h"

extern "C" {
void curdle_window_combine_bmi2(const void*, int, const int*, uint64_t[18]);
void curdle_host_scalar_mul_bmi2(void*, const void*, const uint32_t*);
void curdle_host_add_bmi2(void*, const void*);
int curdle_host_to_affine_bmi2(void*, const void*);
void curdle_host_fp_pow_bmi2(void*, const void*, const uint32_t*);
void curdle_host_fp_from_mont_bmi2(void*, const void*);
int curdle_host_equal_bmi2(const void*, const void*);
int curdle_host_in_subgroup_bmi2(const void*);
void curdle_host_batch_to_affine_bmi2(void*, const void*, size_t);
void curdle_host_fixed_base_table_bmi2(void*, const void*);
void curdle_host_fixed_base_mul_bmi2(void*, const void*, const uint32_t*);
}

static bool fast_isa() {
  static const bool fast = __builtin_cpu_supports("bmi2") && __builtin_cpu_supports("adx");
  return fast;
}

extern "C" void curdle_window_combine(const void* winsums_xyzz, int nw, const int* dbls, uint64_t out[18]) {
  if (fast_isa())
    curdle_window_combine_bmi2(winsums_xyzz, nw, dbls, out);
  else
    curdle_window_combine_generic(winsums_xyzz, nw, dbls, out);
}

extern "C" void curdle_host_scalar_mul(void* r_xyzz, const void* p_xyzz, const uint32_t* k) {
  if (fast_isa())
    curdle_host_scalar_mul_bmi2(r_xyzz, p_xyzz, k);
  else
    curdle_host_scalar_mul_generic(r_xyzz, p_xyzz, k);
}

extern "C" void curdle_host_add(void* acc_xyzz, const void* b_xyzz) {
  if (fast_isa())
    curdle_host_add_bmi2(acc_xyzz, b_xyzz);
  else
    curdle_host_add_generic(acc_xyzz, b_xyzz);
}

extern "C" int curdle_host_to_affine(void* out_affine, const void* p_xyzz) {
  return fast_isa() ? curdle_host_to_affine_bmi2(out_affine, p_xyzz) : curdle_host_to_affine_generic(out_affine, p_xyzz);
}

extern "C" void curdle_host_fp_pow(void* r, const void* a, const uint32_t* e) {
  if (fast_isa())
    curdle_host_fp_pow_bmi2(r, a, e);
  else
    curdle_host_fp_pow_generic(r, a, e);
}

extern "C" void curdle_host_fp_from_mont(void* r, const void* a) {
  if (fast_isa())
    curdle_host_fp_from_mont_bmi2(r, a);
  else
    curdle_host_fp_from_mont_generic(r, a);
}

extern "C" int curdle_host_equal(const void* a_xyzz, const void* b_xyzz) {
  return fast_isa() ? curdle_host_equal_bmi2(a_xyzz, b_xyzz) : curdle_host_equal_generic(a_xyzz, b_xyzz);
}

extern "C" int curdle_host_in_subgroup(const void* p_xyzz) {
  return fast_isa() ? curdle_host_in_subgroup_bmi2(p_xyzz) : curdle_host_in_subgroup_generic(p_xyzz);
}

extern "C" void curdle_host_batch_to_affine(void* out_affine, const void* in_xyzz, size_t n) {
  if (fast_isa())
    curdle_host_batch_to_affine_bmi2(out_affine, in_xyzz, n);
  else
    curdle_host_batch_to_affine_generic(out_affine, in_xyzz, n);
}

extern "C" void curdle_host_fixed_base_table(void* table_affine, const void* p_affine) {
  if (fast_isa())
    curdle_host_fixed_base_table_bmi2(table_affine, p_affine);
  else
    curdle_host_fixed_base_table_generic(table_affine, p_affine);
}

extern "C" void curdle_host_fixed_base_mul(void* r_xyzz, const void* table_affine, const uint32_t* k) {
  if (fast_isa())
    curdle_host_fixed_base_mul_bmi2(r_xyzz, table_affine, k);
  else
    curdle_host_fixed_base_mul_generic(r_xyzz, table_affine, k);
}
