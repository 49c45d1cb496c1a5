// Keccak-f[1600] (FIPS 202) for the two sponge users on the host: SHAKE256 behind
// common.Rand (common_rand.cpp) and STROBE-128 behind the Merlin transcript
// (transcript.cpp).  One verification absorbs ~100 KB of framed transcript data, i.e.
// more than a thousand permutations, so the round is written out over named lanes
// (theta / rho+pi / chi / iota with every index a compile-time constant) rather than as
// the table-driven loop.
#pragma once
#include <stdint.h>
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
#include <immintrin.h>
#endif

namespace curdle {

static inline uint64_t keccak_rotl(uint64_t x, int n) { return (x << n) | (x >> (64 - n)); }

// st: 25 lanes, lane (x, y) at st[x + 5 y], little-endian lanes.
static inline __attribute__((always_inline)) void keccak_f1600(uint64_t st[25]) {
  static const uint64_t RC[24] = {
      0x0000000000000001ull, 0x0000000000008082ull, 0x800000000000808aull, 0x8000000080008000ull,
      0x000000000000808bull, 0x0000000080000001ull, 0x8000000080008081ull, 0x8000000000008009ull,
      0x000000000000008aull, 0x0000000000000088ull, 0x0000000080008009ull, 0x000000008000000aull,
      0x000000008000808bull, 0x800000000000008bull, 0x8000000000008089ull, 0x8000000000008003ull,
      0x8000000000008002ull, 0x8000000000000080ull, 0x000000000000800aull, 0x800000008000000aull,
      0x8000000080008081ull, 0x8000000000008080ull, 0x0000000080000001ull, 0x8000000080008008ull};
  uint64_t a00 = st[0], a10 = st[1], a20 = st[2], a30 = st[3], a40 = st[4];
  uint64_t a01 = st[5], a11 = st[6], a21 = st[7], a31 = st[8], a41 = st[9];
  uint64_t a02 = st[10], a12 = st[11], a22 = st[12], a32 = st[13], a42 = st[14];
  uint64_t a03 = st[15], a13 = st[16], a23 = st[17], a33 = st[18], a43 = st[19];
  uint64_t a04 = st[20], a14 = st[21], a24 = st[22], a34 = st[23], a44 = st[24];
  for (int round = 0; round < 24; round++) {
    // theta
    const uint64_t c0 = a00 ^ a01 ^ a02 ^ a03 ^ a04;
    const uint64_t c1 = a10 ^ a11 ^ a12 ^ a13 ^ a14;
    const uint64_t c2 = a20 ^ a21 ^ a22 ^ a23 ^ a24;
    const uint64_t c3 = a30 ^ a31 ^ a32 ^ a33 ^ a34;
    const uint64_t c4 = a40 ^ a41 ^ a42 ^ a43 ^ a44;
    const uint64_t d0 = c4 ^ keccak_rotl(c1, 1);
    const uint64_t d1 = c0 ^ keccak_rotl(c2, 1);
    const uint64_t d2 = c1 ^ keccak_rotl(c3, 1);
    const uint64_t d3 = c2 ^ keccak_rotl(c4, 1);
    const uint64_t d4 = c3 ^ keccak_rotl(c0, 1);
    // rho + pi: b[y][2x+3y] = rotl(a[x][y] ^ d[x], r[x][y])
    const uint64_t b00 = a00 ^ d0;
    const uint64_t b13 = keccak_rotl(a01 ^ d0, 36);
    const uint64_t b21 = keccak_rotl(a02 ^ d0, 3);
    const uint64_t b34 = keccak_rotl(a03 ^ d0, 41);
    const uint64_t b42 = keccak_rotl(a04 ^ d0, 18);
    const uint64_t b02 = keccak_rotl(a10 ^ d1, 1);
    const uint64_t b10 = keccak_rotl(a11 ^ d1, 44);
    const uint64_t b23 = keccak_rotl(a12 ^ d1, 10);
    const uint64_t b31 = keccak_rotl(a13 ^ d1, 45);
    const uint64_t b44 = keccak_rotl(a14 ^ d1, 2);
    const uint64_t b04 = keccak_rotl(a20 ^ d2, 62);
    const uint64_t b12 = keccak_rotl(a21 ^ d2, 6);
    const uint64_t b20 = keccak_rotl(a22 ^ d2, 43);
    const uint64_t b33 = keccak_rotl(a23 ^ d2, 15);
    const uint64_t b41 = keccak_rotl(a24 ^ d2, 61);
    const uint64_t b01 = keccak_rotl(a30 ^ d3, 28);
    const uint64_t b14 = keccak_rotl(a31 ^ d3, 55);
    const uint64_t b22 = keccak_rotl(a32 ^ d3, 25);
    const uint64_t b30 = keccak_rotl(a33 ^ d3, 21);
    const uint64_t b43 = keccak_rotl(a34 ^ d3, 56);
    const uint64_t b03 = keccak_rotl(a40 ^ d4, 27);
    const uint64_t b11 = keccak_rotl(a41 ^ d4, 20);
    const uint64_t b24 = keccak_rotl(a42 ^ d4, 39);
    const uint64_t b32 = keccak_rotl(a43 ^ d4, 8);
    const uint64_t b40 = keccak_rotl(a44 ^ d4, 14);
    // chi (+ iota on lane (0,0))
    a00 = b00 ^ (~b10 & b20) ^ RC[round];
    a10 = b10 ^ (~b20 & b30);
    a20 = b20 ^ (~b30 & b40);
    a30 = b30 ^ (~b40 & b00);
    a40 = b40 ^ (~b00 & b10);
    a01 = b01 ^ (~b11 & b21);
    a11 = b11 ^ (~b21 & b31);
    a21 = b21 ^ (~b31 & b41);
    a31 = b31 ^ (~b41 & b01);
    a41 = b41 ^ (~b01 & b11);
    a02 = b02 ^ (~b12 & b22);
    a12 = b12 ^ (~b22 & b32);
    a22 = b22 ^ (~b32 & b42);
    a32 = b32 ^ (~b42 & b02);
    a42 = b42 ^ (~b02 & b12);
    a03 = b03 ^ (~b13 & b23);
    a13 = b13 ^ (~b23 & b33);
    a23 = b23 ^ (~b33 & b43);
    a33 = b33 ^ (~b43 & b03);
    a43 = b43 ^ (~b03 & b13);
    a04 = b04 ^ (~b14 & b24);
    a14 = b14 ^ (~b24 & b34);
    a24 = b24 ^ (~b34 & b44);
    a34 = b34 ^ (~b44 & b04);
    a44 = b44 ^ (~b04 & b14);
  }
  st[0] = a00, st[1] = a10, st[2] = a20, st[3] = a30, st[4] = a40;
  st[5] = a01, st[6] = a11, st[7] = a21, st[8] = a31, st[9] = a41;
  st[10] = a02, st[11] = a12, st[12] = a22, st[13] = a32, st[14] = a42;
  st[15] = a03, st[16] = a13, st[17] = a23, st[18] = a33, st[19] = a43;
  st[20] = a04, st[21] = a14, st[22] = a24, st[23] = a34, st[24] = a44;
}

// The same round function compiled for BMI1 + BMI2 (andn for chi, rorx for rho) and picked at
// run time; the generic build is the fallback.
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
__attribute__((target("bmi,bmi2"))) static inline void keccak_f1600_bmi(uint64_t st[25]) { keccak_f1600(st); }
static inline void keccak_f1600_generic(uint64_t st[25]) { keccak_f1600(st); }

// AVX-512 build: the five planes y = 0..4 live in one 512-bit register each (lane x in element
// x).  theta is lane-wise xors plus two lane rotations; rho is one variable rotate per plane;
// pi is done in two halves around chi: first every plane y is permuted so that its element j
// holds the lane destined for plane j (T[y][j] = A[(y + 3j) mod 5, y]), which makes T the
// transpose of the post-pi state and chi a REGISTER-wise operation (three-input logic, no lane
// moves); then the 5 x 5 transpose brings the state back to planes.  ~40 instructions per round
// against ~200 scalar ones; one verification runs ~1,400 permutations.
__attribute__((target("avx512f,avx512vl"))) static inline void keccak_f1600_avx512(uint64_t st[25]) {
  static const uint64_t RC[24] = {
      0x0000000000000001ull, 0x0000000000008082ull, 0x800000000000808aull, 0x8000000080008000ull,
      0x000000000000808bull, 0x0000000080000001ull, 0x8000000080008081ull, 0x8000000000008009ull,
      0x000000000000008aull, 0x0000000000000088ull, 0x0000000080008009ull, 0x000000008000000aull,
      0x000000008000808bull, 0x800000000000008bull, 0x8000000000008089ull, 0x8000000000008003ull,
      0x8000000000008002ull, 0x8000000000000080ull, 0x000000000000800aull, 0x800000008000000aull,
      0x8000000080008081ull, 0x8000000000008080ull, 0x0000000080000001ull, 0x8000000080008008ull};
  const __m512i prev = _mm512_setr_epi64(4, 0, 1, 2, 3, 5, 6, 7);   // element x <- x - 1
  const __m512i next = _mm512_setr_epi64(1, 2, 3, 4, 0, 5, 6, 7);   // element x <- x + 1
  const __m512i rho0 = _mm512_setr_epi64(0, 1, 62, 28, 27, 0, 0, 0);
  const __m512i rho1 = _mm512_setr_epi64(36, 44, 6, 55, 20, 0, 0, 0);
  const __m512i rho2 = _mm512_setr_epi64(3, 10, 43, 25, 39, 0, 0, 0);
  const __m512i rho3 = _mm512_setr_epi64(41, 45, 15, 21, 8, 0, 0, 0);
  const __m512i rho4 = _mm512_setr_epi64(18, 2, 61, 56, 14, 0, 0, 0);
  const __m512i pi0 = _mm512_setr_epi64(0, 3, 1, 4, 2, 5, 6, 7);    // element j <- (y + 3 j) mod 5
  const __m512i pi1 = _mm512_setr_epi64(1, 4, 2, 0, 3, 5, 6, 7);
  const __m512i pi2 = _mm512_setr_epi64(2, 0, 3, 1, 4, 5, 6, 7);
  const __m512i pi3 = _mm512_setr_epi64(3, 1, 4, 2, 0, 5, 6, 7);
  const __m512i pi4 = _mm512_setr_epi64(4, 2, 0, 3, 1, 5, 6, 7);
  // transpose: out = [a[i], a[i+1], b[i], b[i+1], c[.], ...]
  const __m512i tr01 = _mm512_setr_epi64(0, 1, 8, 9, 4, 5, 6, 7);   // elements 0,1 of a then 0,1 of b
  const __m512i tr23 = _mm512_setr_epi64(2, 3, 10, 11, 4, 5, 6, 7);
  const __m512i tr45 = _mm512_setr_epi64(4, 5, 12, 13, 4, 5, 6, 7);
  const __m512i bc0 = _mm512_set1_epi64(0), bc1 = _mm512_set1_epi64(1), bc2 = _mm512_set1_epi64(2),
                bc3 = _mm512_set1_epi64(3), bc4 = _mm512_set1_epi64(4);
  __m512i P0 = _mm512_maskz_loadu_epi64(0x1f, st), P1 = _mm512_maskz_loadu_epi64(0x1f, st + 5),
          P2 = _mm512_maskz_loadu_epi64(0x1f, st + 10), P3 = _mm512_maskz_loadu_epi64(0x1f, st + 15),
          P4 = _mm512_maskz_loadu_epi64(0x1f, st + 20);
  for (int round = 0; round < 24; round++) {
    // theta
    __m512i C = _mm512_ternarylogic_epi64(_mm512_ternarylogic_epi64(P0, P1, P2, 0x96), P3, P4, 0x96);
    const __m512i D0 = _mm512_permutexvar_epi64(prev, C);
    const __m512i D1 = _mm512_rol_epi64(_mm512_permutexvar_epi64(next, C), 1);
    P0 = _mm512_ternarylogic_epi64(P0, D0, D1, 0x96);
    P1 = _mm512_ternarylogic_epi64(P1, D0, D1, 0x96);
    P2 = _mm512_ternarylogic_epi64(P2, D0, D1, 0x96);
    P3 = _mm512_ternarylogic_epi64(P3, D0, D1, 0x96);
    P4 = _mm512_ternarylogic_epi64(P4, D0, D1, 0x96);
    // rho, first half of pi
    const __m512i T0 = _mm512_permutexvar_epi64(pi0, _mm512_rolv_epi64(P0, rho0));
    const __m512i T1 = _mm512_permutexvar_epi64(pi1, _mm512_rolv_epi64(P1, rho1));
    const __m512i T2 = _mm512_permutexvar_epi64(pi2, _mm512_rolv_epi64(P2, rho2));
    const __m512i T3 = _mm512_permutexvar_epi64(pi3, _mm512_rolv_epi64(P3, rho3));
    const __m512i T4 = _mm512_permutexvar_epi64(pi4, _mm512_rolv_epi64(P4, rho4));
    // chi across registers (U[X] = T[X] ^ (~T[X+1] & T[X+2])), iota on element (X = 0, Y = 0)
    __m512i U0 = _mm512_ternarylogic_epi64(T0, T1, T2, 0xD2);
    const __m512i U1 = _mm512_ternarylogic_epi64(T1, T2, T3, 0xD2);
    const __m512i U2 = _mm512_ternarylogic_epi64(T2, T3, T4, 0xD2);
    const __m512i U3 = _mm512_ternarylogic_epi64(T3, T4, T0, 0xD2);
    const __m512i U4 = _mm512_ternarylogic_epi64(T4, T0, T1, 0xD2);
    U0 = _mm512_xor_si512(U0, _mm512_maskz_set1_epi64(0x01, (long long)RC[round]));
    // second half of pi: transpose U[X][Y] -> P[Y][X]
    const __m512i l01 = _mm512_unpacklo_epi64(U0, U1), h01 = _mm512_unpackhi_epi64(U0, U1);
    const __m512i l23 = _mm512_unpacklo_epi64(U2, U3), h23 = _mm512_unpackhi_epi64(U2, U3);
    P0 = _mm512_mask_permutexvar_epi64(_mm512_permutex2var_epi64(l01, tr01, l23), 0x10, bc0, U4);
    P1 = _mm512_mask_permutexvar_epi64(_mm512_permutex2var_epi64(h01, tr01, h23), 0x10, bc1, U4);
    P2 = _mm512_mask_permutexvar_epi64(_mm512_permutex2var_epi64(l01, tr23, l23), 0x10, bc2, U4);
    P3 = _mm512_mask_permutexvar_epi64(_mm512_permutex2var_epi64(h01, tr23, h23), 0x10, bc3, U4);
    P4 = _mm512_mask_permutexvar_epi64(_mm512_permutex2var_epi64(l01, tr45, l23), 0x10, bc4, U4);
  }
  _mm512_mask_storeu_epi64(st, 0x1f, P0);
  _mm512_mask_storeu_epi64(st + 5, 0x1f, P1);
  _mm512_mask_storeu_epi64(st + 10, 0x1f, P2);
  _mm512_mask_storeu_epi64(st + 15, 0x1f, P3);
  _mm512_mask_storeu_epi64(st + 20, 0x1f, P4);
}

// Which build is fastest depends on the core (the AVX-512 one is shuffle-port-bound on Intel
// server cores and no faster than the BMI one there): timed once, at first use.
static inline int keccak_pick_isa() {
  const bool bmi = __builtin_cpu_supports("bmi") && __builtin_cpu_supports("bmi2");
  const bool avx = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512vl");
  if (!avx) return bmi ? 1 : 0;
  if (!bmi) return 2;
  uint64_t st[25];
  for (int i = 0; i < 25; i++) st[i] = 0x9e3779b97f4a7c15ull * (uint64_t)(i + 1);
  unsigned long long best[2] = {~0ull, ~0ull};
  for (int rep = 0; rep < 8; rep++)
    for (int which = 0; which < 2; which++) {
      const unsigned long long t0 = __builtin_ia32_rdtsc();
      for (int i = 0; i < 16; i++) {
        if (which)
          keccak_f1600_avx512(st);
        else
          keccak_f1600_bmi(st);
      }
      const unsigned long long dt = __builtin_ia32_rdtsc() - t0;
      if (dt < best[which]) best[which] = dt;
    }
  return best[1] < best[0] ? 2 : 1;
}

static inline void keccak_f1600_dispatch(uint64_t st[25]) {
  static const int isa = keccak_pick_isa();
  if (isa == 2)
    keccak_f1600_avx512(st);
  else if (isa == 1)
    keccak_f1600_bmi(st);
  else
    keccak_f1600_generic(st);
}
#else
static inline void keccak_f1600_dispatch(uint64_t st[25]) { keccak_f1600(st); }
#endif

}  // namespace curdle
