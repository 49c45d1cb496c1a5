// Serial host-side group operations with run-time ISA dispatch (host_ops.cpp).
#pragma once
#include <stddef.h>
#include <stdint.h>

extern "C" {
void curdle_window_combine(const void* winsums_xyzz, int nw, const int* dbls, uint64_t out[18]);
void curdle_host_scalar_mul(void* r_xyzz, const void* p_xyzz, const uint32_t* k8);
void curdle_host_add(void* acc_xyzz, const void* b_xyzz);
int curdle_host_to_affine(void* out_affine, const void* p_xyzz);
void curdle_host_fp_pow(void* r, const void* a, const uint32_t* e12);
void curdle_host_fp_from_mont(void* r, const void* a);
int curdle_host_equal(const void* a_xyzz, const void* b_xyzz);
int curdle_host_in_subgroup(const void* p_xyzz);
void curdle_host_batch_to_affine(void* out_affine, const void* in_xyzz, size_t n);
// 32 x 255 affine points (783,360 bytes): d * 2^(8w) * P; then k * P in <= 32 mixed additions
void curdle_host_fixed_base_table(void* table_affine, const void* p_affine);
void curdle_host_fixed_base_mul(void* r_xyzz, const void* table_affine, const uint32_t* k8);
}
