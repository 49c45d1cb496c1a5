/*
 * curdle_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement, in plain C, of the arithmetic behind the reference's MSM
 * hot path.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load this library; nothing under go-curdleproofs_amd/ does.  It is
 * written independently of the product code on purpose: 64-bit limbs with
 * unsigned __int128 (the product uses 32-bit limbs), Jacobian coordinates (the
 * product uses XYZZ), unsigned Pippenger windows (the product uses signed
 * digits).
 *
 * PARITY UNPINNED in the known-answer sense: the reference's tests hold no
 * golden vectors for this path and the reference
 * cannot be built here (Go, un-vendored gnark-crypto v0.11.0 -- go.mod:6).
 * This file is pinned against oracle/py/bls12381_ref.py (textbook affine
 * big-integer arithmetic, itself anchored on the published curve parameters)
 * by tests/test_oracle.py.
 *
 * What is restated (paths relative to /root/reference):
 *   oracle_msm_naive      the contract of gnark (*G1Jac).MultiExp as called at
 *                         msmaccumulator/msmaccumulator.go:59 -- sum_i s_i*P_i by
 *                         double-and-add; n = 0 -> infinity; (0,0) = infinity
 *                         (curdleproof.go:23).
 *   oracle_msm_pippenger  the same function computed the way gnark computes
 *                         it (bucket method over windows, one task per window,
 *                         NbTasks threads -- common/util.go:14 MultiExpConf);
 *                         this is the "port" CPU baseline of bench.py.
 *   oracle_points_walk    synthetic inputs of SURVEY.md section 8(d):
 *                         P_i = (k + i*q) * G.
 *   oracle_scalar_mul_gen k * G  (common/rand.go:72-83 GetG1Affine).
 *
 * Layouts are gnark's: fp.Element = 6 x u64 Montgomery (R = 2^384),
 * fr.Element = 4 x u64 Montgomery (R = 2^256), G1Affine = X|Y, G1Jac = X|Y|Z.
 *
 * Build: make -C oracle   (gcc -O2 -shared -fPIC -pthread)
 */
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef uint64_t u64;
typedef unsigned __int128 u128;

/* ------------------------------------------------------------------ Fp --- */
static const u64 P[6] = {0xb9feffffffffaaabull, 0x1eabfffeb153ffffull, 0x6730d2a0f6b0f624ull,
                         0x64774b84f38512bfull, 0x4b1ba7b6434bacd7ull, 0x1a0111ea397fe69aull};
static const u64 P_INV = 0x89f3fffcfffcfffdull; /* -p^-1 mod 2^64 */
static const u64 FP_ONE[6] = {0x760900000002fffdull, 0xebf4000bc40c0002ull, 0x5f48985753c758baull,
                              0x77ce585370525745ull, 0x5c071a97a256ec6dull, 0x15f65ec3fa80e493ull};
/* generator, Montgomery form */
static const u64 GEN_X[6] = {0x5cb38790fd530c16ull, 0x7817fc679976fff5ull, 0x154f95c7143ba1c1ull,
                             0xf0ae6acdf3d0e747ull, 0xedce6ecc21dbf440ull, 0x120177419e0bfb75ull};
static const u64 GEN_Y[6] = {0xbaac93d50ce72271ull, 0x8c22631a7918fd8eull, 0xdd595f13570725ceull,
                             0x51ac582950405194ull, 0x0e1c8c3fad0059c0ull, 0x0bbc3efc5008a26aull};

typedef struct { u64 l[6]; } fp;

static int fp_is_zero(const fp* a) { return (a->l[0] | a->l[1] | a->l[2] | a->l[3] | a->l[4] | a->l[5]) == 0; }
static int fp_eq(const fp* a, const fp* b) { return memcmp(a, b, sizeof(fp)) == 0; }

static int ge_p(const u64* a) {
  for (int i = 5; i >= 0; i--) {
    if (a[i] > P[i]) return 1;
    if (a[i] < P[i]) return 0;
  }
  return 1;
}
static void sub_p(u64* a) {
  u64 borrow = 0;
  for (int i = 0; i < 6; i++) {
    u128 t = (u128)a[i] - P[i] - borrow;
    a[i] = (u64)t;
    borrow = (u64)(t >> 64) & 1;
  }
}
static void fp_add(fp* r, const fp* a, const fp* b) {
  u64 c = 0, t[6];
  for (int i = 0; i < 6; i++) {
    u128 s = (u128)a->l[i] + b->l[i] + c;
    t[i] = (u64)s;
    c = (u64)(s >> 64);
  }
  if (c || ge_p(t)) sub_p(t);
  memcpy(r->l, t, 48);
}
static void fp_sub(fp* r, const fp* a, const fp* b) {
  u64 borrow = 0, t[6];
  for (int i = 0; i < 6; i++) {
    u128 s = (u128)a->l[i] - b->l[i] - borrow;
    t[i] = (u64)s;
    borrow = (u64)(s >> 64) & 1;
  }
  if (borrow) {
    u64 c = 0;
    for (int i = 0; i < 6; i++) {
      u128 s = (u128)t[i] + P[i] + c;
      t[i] = (u64)s;
      c = (u64)(s >> 64);
    }
  }
  memcpy(r->l, t, 48);
}
static void fp_neg(fp* r, const fp* a) {
  fp z;
  memset(&z, 0, sizeof z);
  fp_sub(r, &z, a);
}
/* Montgomery product (CIOS, 64-bit limbs) */
static void fp_mul(fp* r, const fp* a, const fp* b) {
  u64 t[8] = {0};
  for (int i = 0; i < 6; i++) {
    u64 c = 0;
    for (int j = 0; j < 6; j++) {
      u128 s = (u128)a->l[j] * b->l[i] + t[j] + c;
      t[j] = (u64)s;
      c = (u64)(s >> 64);
    }
    u128 s = (u128)t[6] + c;
    t[6] = (u64)s;
    t[7] = (u64)(s >> 64);
    u64 m = t[0] * P_INV;
    s = (u128)m * P[0] + t[0];
    c = (u64)(s >> 64);
    for (int j = 1; j < 6; j++) {
      s = (u128)m * P[j] + t[j] + c;
      t[j - 1] = (u64)s;
      c = (u64)(s >> 64);
    }
    s = (u128)t[6] + c;
    t[5] = (u64)s;
    t[6] = t[7] + (u64)(s >> 64);
  }
  if (t[6] || ge_p(t)) sub_p(t);
  memcpy(r->l, t, 48);
}
static void fp_sqr(fp* r, const fp* a) { fp_mul(r, a, a); }
static void fp_inv(fp* r, const fp* a) { /* a^(p-2) */
  u64 e[6];
  memcpy(e, P, 48);
  e[0] -= 2;
  fp acc, base = *a;
  memcpy(acc.l, FP_ONE, 48);
  for (int i = 0; i < 384; i++) {
    if ((e[i / 64] >> (i % 64)) & 1) fp_mul(&acc, &acc, &base);
    fp_sqr(&base, &base);
  }
  *r = acc;
}

/* ------------------------------------------------------------------ Fr --- */
static const u64 RMOD[4] = {0xffffffff00000001ull, 0x53bda402fffe5bfeull, 0x3339d80809a1d805ull, 0x73eda753299d7d48ull};
static const u64 R_INV = 0xfffffffeffffffffull; /* -r^-1 mod 2^64 */

/* Montgomery -> canonical integer (fr.Element.BigInt): multiply by 1 */
static void fr_from_mont(u64 out[4], const u64 in[4]) {
  u64 t[5];
  memcpy(t, in, 32);
  t[4] = 0;
  for (int i = 0; i < 4; i++) {
    u64 m = t[0] * R_INV;
    u128 s = (u128)m * RMOD[0] + t[0];
    u64 c = (u64)(s >> 64);
    for (int j = 1; j < 4; j++) {
      s = (u128)m * RMOD[j] + t[j] + c;
      t[j - 1] = (u64)s;
      c = (u64)(s >> 64);
    }
    s = (u128)t[4] + c;
    t[3] = (u64)s;
    t[4] = (u64)(s >> 64);
  }
  /* result < 2r; reduce once */
  int ge = t[4] != 0;
  if (!ge) {
    ge = 1;
    for (int i = 3; i >= 0; i--) {
      if (t[i] > RMOD[i]) break;
      if (t[i] < RMOD[i]) { ge = 0; break; }
    }
  }
  if (ge) {
    u64 borrow = 0;
    for (int i = 0; i < 4; i++) {
      u128 s = (u128)t[i] - RMOD[i] - borrow;
      t[i] = (u64)s;
      borrow = (u64)(s >> 64) & 1;
    }
  }
  memcpy(out, t, 32);
}

/* ------------------------------------------------------------------ G1 --- */
typedef struct { fp x, y; } g1a;       /* (0,0) = infinity, curdleproof.go:23 */
typedef struct { fp x, y, z; } g1j;    /* Z = 0 = infinity */

static int g1a_is_inf(const g1a* p) { return fp_is_zero(&p->x) && fp_is_zero(&p->y); }
static void g1j_set_inf(g1j* p) {
  memcpy(p->x.l, FP_ONE, 48);
  memcpy(p->y.l, FP_ONE, 48);
  memset(p->z.l, 0, 48);
}
static void g1j_from_affine(g1j* r, const g1a* a) {
  if (g1a_is_inf(a)) { g1j_set_inf(r); return; }
  r->x = a->x;
  r->y = a->y;
  memcpy(r->z.l, FP_ONE, 48);
}
/* dbl-2009-l (a = 0) */
static void g1j_dbl(g1j* r, const g1j* p) {
  if (fp_is_zero(&p->z)) { *r = *p; return; }
  fp A, B, C, D, E, F, t;
  fp_sqr(&A, &p->x);
  fp_sqr(&B, &p->y);
  fp_sqr(&C, &B);
  fp_add(&t, &p->x, &B);
  fp_sqr(&t, &t);
  fp_sub(&t, &t, &A);
  fp_sub(&t, &t, &C);
  fp_add(&D, &t, &t);
  fp_add(&E, &A, &A);
  fp_add(&E, &E, &A);
  fp_sqr(&F, &E);
  fp z3;
  fp_mul(&z3, &p->y, &p->z);
  fp_add(&z3, &z3, &z3);
  fp x3;
  fp_sub(&x3, &F, &D);
  fp_sub(&x3, &x3, &D);
  fp_sub(&t, &D, &x3);
  fp_mul(&t, &E, &t);
  fp c8;
  fp_add(&c8, &C, &C);
  fp_add(&c8, &c8, &c8);
  fp_add(&c8, &c8, &c8);
  fp_sub(&r->y, &t, &c8);
  r->x = x3;
  r->z = z3;
}
/* add-2007-bl with the exceptional cases */
static void g1j_add(g1j* r, const g1j* p, const g1j* q) {
  if (fp_is_zero(&p->z)) { *r = *q; return; }
  if (fp_is_zero(&q->z)) { *r = *p; return; }
  fp z1z1, z2z2, u1, u2, s1, s2, h, i, j, rr, v, t;
  fp_sqr(&z1z1, &p->z);
  fp_sqr(&z2z2, &q->z);
  fp_mul(&u1, &p->x, &z2z2);
  fp_mul(&u2, &q->x, &z1z1);
  fp_mul(&s1, &p->y, &q->z);
  fp_mul(&s1, &s1, &z2z2);
  fp_mul(&s2, &q->y, &p->z);
  fp_mul(&s2, &s2, &z1z1);
  if (fp_eq(&u1, &u2)) {
    if (fp_eq(&s1, &s2)) { g1j_dbl(r, p); return; }
    g1j_set_inf(r);
    return;
  }
  fp_sub(&h, &u2, &u1);
  fp_add(&i, &h, &h);
  fp_sqr(&i, &i);
  fp_mul(&j, &h, &i);
  fp_sub(&rr, &s2, &s1);
  fp_add(&rr, &rr, &rr);
  fp_mul(&v, &u1, &i);
  fp x3, y3, z3;
  fp_sqr(&x3, &rr);
  fp_sub(&x3, &x3, &j);
  fp_sub(&x3, &x3, &v);
  fp_sub(&x3, &x3, &v);
  fp_sub(&t, &v, &x3);
  fp_mul(&y3, &rr, &t);
  fp_mul(&t, &s1, &j);
  fp_add(&t, &t, &t);
  fp_sub(&y3, &y3, &t);
  fp_add(&z3, &p->z, &q->z);
  fp_sqr(&z3, &z3);
  fp_sub(&z3, &z3, &z1z1);
  fp_sub(&z3, &z3, &z2z2);
  fp_mul(&z3, &z3, &h);
  r->x = x3;
  r->y = y3;
  r->z = z3;
}
static void g1j_add_affine(g1j* r, const g1j* p, const g1a* q) {
  g1j qj;
  g1j_from_affine(&qj, q);
  g1j_add(r, p, &qj);
}
static void g1j_to_affine(g1a* r, const g1j* p) {
  if (fp_is_zero(&p->z)) { memset(r, 0, sizeof *r); return; }
  fp zi, zi2, zi3;
  fp_inv(&zi, &p->z);
  fp_sqr(&zi2, &zi);
  fp_mul(&zi3, &zi2, &zi);
  fp_mul(&r->x, &p->x, &zi2);
  fp_mul(&r->y, &p->y, &zi3);
}
/* canonical Jacobian the C ABI returns: (x, y, 1) or (1, 1, 0) */
static void g1j_write_canonical(u64 out[18], const g1j* p) {
  g1a a;
  g1j c;
  if (fp_is_zero(&p->z)) {
    g1j_set_inf(&c);
  } else {
    g1j_to_affine(&a, p);
    c.x = a.x;
    c.y = a.y;
    memcpy(c.z.l, FP_ONE, 48);
  }
  memcpy(out, &c, 144);
}
/* k * p, k a canonical 256-bit integer */
static void g1j_scalar_mul(g1j* r, const g1j* p, const u64 k[4]) {
  g1j acc;
  g1j_set_inf(&acc);
  for (int i = 255; i >= 0; i--) {
    g1j_dbl(&acc, &acc);
    if ((k[i / 64] >> (i % 64)) & 1) g1j_add(&acc, &acc, p);
  }
  *r = acc;
}

/* ----------------------------------------------------------- exported --- */

/* sum_i scalars[i] * points[i] by plain double-and-add (the MultiExp contract,
 * msmaccumulator.go:59).  scalars are Montgomery fr.Elements. */
int oracle_msm_naive(const u64* points, const u64* scalars, size_t n, u64 out_jac[18]) {
  g1j acc;
  g1j_set_inf(&acc);
  for (size_t i = 0; i < n; i++) {
    g1a a;
    memcpy(&a, points + 12 * i, 96);
    if (g1a_is_inf(&a)) continue;
    u64 k[4];
    fr_from_mont(k, scalars + 4 * i);
    g1j pj, t;
    g1j_from_affine(&pj, &a);
    g1j_scalar_mul(&t, &pj, k);
    g1j_add(&acc, &acc, &t);
  }
  g1j_write_canonical(out_jac, &acc);
  return 0;
}

/* Bucket-method MSM the way gnark's MultiExp is organised: the scalar is cut
 * into ceil(255/c) unsigned c-bit windows, each window is an independent task
 * (buckets 1..2^c-1, running-sum reduction), tasks are spread over `threads`
 * workers (MultiExpConf.NbTasks, common/util.go:14) and the window results are
 * combined by Horner.  This is the CPU baseline ("port") bench.py times. */
typedef struct {
  const u64* points;
  const u64* canon; /* n x 4 canonical scalars */
  size_t n;
  int c, nwin;
  g1j* winsum;
  int next; /* next window to take */
  pthread_mutex_t mu;
} pip_job;

static void pip_window(pip_job* jb, int w) {
  const int c = jb->c;
  const size_t nb = ((size_t)1 << c) - 1;
  g1j* buckets = (g1j*)malloc(nb * sizeof(g1j));
  for (size_t b = 0; b < nb; b++) g1j_set_inf(&buckets[b]);
  const int bit = w * c;
  for (size_t i = 0; i < jb->n; i++) {
    const u64* k = jb->canon + 4 * i;
    int limb = bit / 64, off = bit % 64;
    u64 d = k[limb] >> off;
    if (off + c > 64 && limb + 1 < 4) d |= k[limb + 1] << (64 - off);
    d &= ((u64)1 << c) - 1;
    if (!d) continue;
    g1a a;
    memcpy(&a, jb->points + 12 * i, 96);
    if (g1a_is_inf(&a)) continue;
    g1j_add_affine(&buckets[d - 1], &buckets[d - 1], &a);
  }
  g1j run, sum;
  g1j_set_inf(&run);
  g1j_set_inf(&sum);
  for (size_t b = nb; b-- > 0;) {
    g1j_add(&run, &run, &buckets[b]);
    g1j_add(&sum, &sum, &run);
  }
  jb->winsum[w] = sum;
  free(buckets);
}
static void* pip_worker(void* arg) {
  pip_job* jb = (pip_job*)arg;
  for (;;) {
    pthread_mutex_lock(&jb->mu);
    int w = jb->next < jb->nwin ? jb->next++ : -1;
    pthread_mutex_unlock(&jb->mu);
    if (w < 0) return NULL;
    pip_window(jb, w);
  }
}
int oracle_msm_pippenger(const u64* points, const u64* scalars, size_t n, int threads, int c, u64 out_jac[18]) {
  if (c <= 0) { /* gnark-like choice: c ~ log2(n) - 2, clamped */
    c = 4;
    while (c < 16 && ((size_t)1 << (c + 3)) < n) c++;
  }
  if (threads < 1) threads = 1;
  if (threads > 256) threads = 256;
  pip_job jb;
  jb.points = points;
  jb.n = n;
  jb.c = c;
  jb.nwin = (255 + c - 1) / c;
  jb.next = 0;
  u64* canon = (u64*)malloc(n ? n * 32 : 32);
  for (size_t i = 0; i < n; i++) fr_from_mont(canon + 4 * i, scalars + 4 * i);
  jb.canon = canon;
  jb.winsum = (g1j*)malloc(jb.nwin * sizeof(g1j));
  pthread_mutex_init(&jb.mu, NULL);
  pthread_t th[256];
  for (int t = 0; t < threads; t++) pthread_create(&th[t], NULL, pip_worker, &jb);
  for (int t = 0; t < threads; t++) pthread_join(th[t], NULL);
  g1j acc;
  g1j_set_inf(&acc);
  for (int w = jb.nwin - 1; w >= 0; w--) {
    for (int k = 0; k < c; k++) g1j_dbl(&acc, &acc);
    g1j_add(&acc, &acc, &jb.winsum[w]);
  }
  g1j_write_canonical(out_jac, &acc);
  pthread_mutex_destroy(&jb.mu);
  free(jb.winsum);
  free(canon);
  return 0;
}

/* out = k * G for a canonical 256-bit k (little-endian u64 limbs); affine,
 * Montgomery.  common/rand.go:72-83. */
int oracle_scalar_mul_gen(const u64 k[4], u64 out_affine[12]) {
  g1a g, r;
  memcpy(g.x.l, GEN_X, 48);
  memcpy(g.y.l, GEN_Y, 48);
  g1j gj, t;
  g1j_from_affine(&gj, &g);
  g1j_scalar_mul(&t, &gj, k);
  g1j_to_affine(&r, &t);
  memcpy(out_affine, &r, 96);
  return 0;
}

/* Synthetic bases of SURVEY.md 8(d): P_i = (k + i*q) * G for i < n, affine,
 * Montgomery.  The walk is done in Jacobian coordinates and normalised with one
 * batched inversion per block. */
int oracle_points_walk(const u64 k[4], const u64 q[4], size_t n, u64* out_points) {
  if (!n) return 0;
  g1a g, qa;
  memcpy(g.x.l, GEN_X, 48);
  memcpy(g.y.l, GEN_Y, 48);
  g1j gj, cur, qj;
  g1j_from_affine(&gj, &g);
  g1j_scalar_mul(&cur, &gj, k);
  g1j_scalar_mul(&qj, &gj, q);
  g1j_to_affine(&qa, &qj);
  enum { BLK = 4096 };
  g1j* blk = (g1j*)malloc(BLK * sizeof(g1j));
  fp* pref = (fp*)malloc(BLK * sizeof(fp));
  for (size_t base = 0; base < n; base += BLK) {
    size_t m = n - base < BLK ? n - base : BLK;
    for (size_t i = 0; i < m; i++) {
      blk[i] = cur;
      g1j_add_affine(&cur, &cur, &qa);
    }
    /* batch inversion of the non-zero Z (Montgomery's trick) */
    fp acc;
    memcpy(acc.l, FP_ONE, 48);
    for (size_t i = 0; i < m; i++) {
      pref[i] = acc;
      if (!fp_is_zero(&blk[i].z)) fp_mul(&acc, &acc, &blk[i].z);
    }
    fp inv;
    fp_inv(&inv, &acc);
    for (size_t i = m; i-- > 0;) {
      g1a a;
      if (fp_is_zero(&blk[i].z)) {
        memset(&a, 0, sizeof a);
      } else {
        fp zi, zi2, zi3;
        fp_mul(&zi, &inv, &pref[i]);
        fp_mul(&inv, &inv, &blk[i].z);
        fp_sqr(&zi2, &zi);
        fp_mul(&zi3, &zi2, &zi);
        fp_mul(&a.x, &blk[i].x, &zi2);
        fp_mul(&a.y, &blk[i].y, &zi3);
      }
      memcpy(out_points + 12 * (base + i), &a, 96);
    }
  }
  free(blk);
  free(pref);
  return 0;
}

/* Any Jacobian representative -> canonical (x, y, 1) / (1, 1, 0). */
int oracle_jac_normalise(const u64 in_jac[18], u64 out_jac[18]) {
  g1j p;
  memcpy(&p, in_jac, 144);
  g1j_write_canonical(out_jac, &p);
  return 0;
}

/* Field primitive checks used by tests/test_oracle.py. */
int oracle_fp_mul(const u64 a[6], const u64 b[6], u64 out[6]) {
  fp x, y, r;
  memcpy(&x, a, 48);
  memcpy(&y, b, 48);
  fp_mul(&r, &x, &y);
  memcpy(out, &r, 48);
  return 0;
}
int oracle_fr_from_mont(const u64 a[4], u64 out[4]) {
  fr_from_mont(out, a);
  return 0;
}
