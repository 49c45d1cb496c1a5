/*
 * cpu_msm_fast.c -- TEST / MEASUREMENT INFRASTRUCTURE ONLY (bench.py's cpu_baseline leg and
 * tests/test_oracle.py; never linked or loaded by the product).
 *
 * The CPU baseline next to the GPU number: a multi-threaded bucket-method MSM written the way
 * the reference's dependency computes it -- gnark-crypto v0.11.0 (*G1Jac).MultiExp, go.mod:6,
 * called with NbTasks = runtime.NumCPU() (/root/reference/common/util.go:14) at
 * msmaccumulator/msmaccumulator.go:59 -- as far as that can be restated without its source:
 *   * 6 x 64-bit limbs, Montgomery products on mulx / adcx / adox (inline assembly, the
 *     "no-carry" product gnark's amd64 code implements; see fp_mul below);
 *   * signed c-bit digits (c = 16 at N = 2^20), so 2^(c-1) buckets per window;
 *   * extended-Jacobian (XYZZ) buckets with mixed additions (8M + 2S), gnark's g1JacExtended;
 *   * every window's points split over several tasks (gnark splits a window over
 *     NbTasks / windows goroutines when there are more cores than windows), each task with
 *     its own bucket array, merged bucket-wise before the window's running sum.
 * gnark-crypto itself CANNOT be run in this pipeline (no Go toolchain on either box); this
 * port lacks gnark's batch-affine bucket additions, so read it as "the same algorithm with
 * extended-Jacobian buckets", not as gnark's number.
 *
 * Its own group law, recoding and task split; the one thing borrowed from the product tree is
 * the generated mulx / adcx / adox instruction sequence of the field product (data, included
 * below).  It is the TIMED baseline, not the checker: its results are themselves checked
 * against curdle_oracle.c (independent arithmetic) in tests/test_oracle.py.
 *
 * Build: make -C oracle   (gcc -O3 -march=native -mbmi2 -madx -shared -fPIC -pthread)
 */
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <x86intrin.h>

typedef uint64_t u64;
typedef unsigned __int128 u128;

static const u64 P[6] = {0xb9feffffffffaaabull, 0x1eabfffeb153ffffull, 0x6730d2a0f6b0f624ull,
                         0x64774b84f38512bfull, 0x4b1ba7b6434bacd7ull, 0x1a0111ea397fe69aull};
#if !(defined(__BMI2__) && defined(__ADX__))
static const u64 P_INV = 0x89f3fffcfffcfffdull; /* -p^-1 mod 2^64 */
#endif
static const u64 ONE[6] = {0x760900000002fffdull, 0xebf4000bc40c0002ull, 0x5f48985753c758baull,
                           0x77ce585370525745ull, 0x5c071a97a256ec6dull, 0x15f65ec3fa80e493ull};
static const u64 R_MOD[4] = {0xffffffff00000001ull, 0x53bda402fffe5bfeull, 0x3339d80809a1d805ull, 0x73eda753299d7d48ull};
static const u64 R_INV64 = 0xfffffffeffffffffull; /* -r^-1 mod 2^64 */

typedef struct { u64 l[6]; } fp;
typedef struct { fp x, y; } aff;          /* (0,0) = infinity */
typedef struct { fp x, y, zz, zzz; } xyzz; /* zz = 0: infinity */

static inline int fp_is_zero(const fp* a) { return (a->l[0] | a->l[1] | a->l[2] | a->l[3] | a->l[4] | a->l[5]) == 0; }

/* r = a - p if a >= p (a < 2p) */
static inline void fp_reduce(fp* r, const u64 t[6]) {
  u64 d[6];
  unsigned char b = 0;
  for (int i = 0; i < 6; i++) b = _subborrow_u64(b, t[i], P[i], (unsigned long long*)&d[i]);
  for (int i = 0; i < 6; i++) r->l[i] = b ? t[i] : d[i];
}
static inline void fp_add(fp* r, const fp* a, const fp* b) {
  u64 t[6];
  unsigned char c = 0;
  for (int i = 0; i < 6; i++) c = _addcarry_u64(c, a->l[i], b->l[i], (unsigned long long*)&t[i]);
  fp_reduce(r, t); /* p < 2^381: no carry out of the top limb */
}
static inline void fp_sub(fp* r, const fp* a, const fp* b) {
  u64 t[6];
  unsigned char bo = 0;
  for (int i = 0; i < 6; i++) bo = _subborrow_u64(bo, a->l[i], b->l[i], (unsigned long long*)&t[i]);
  if (bo) {
    unsigned char c = 0;
    for (int i = 0; i < 6; i++) c = _addcarry_u64(c, t[i], P[i], (unsigned long long*)&t[i]);
  }
  memcpy(r->l, t, 48);
}
static inline void fp_neg(fp* r, const fp* a) {
  if (fp_is_zero(a)) {
    *r = *a;
    return;
  }
  unsigned char bo = 0;
  for (int i = 0; i < 6; i++) bo = _subborrow_u64(bo, P[i], a->l[i], (unsigned long long*)&r->l[i]);
}
/* Montgomery product: operand scanning with the reduction row folded in behind every multiplier
 * word, the two carry chains of adcx / adox side by side, no seventh limb (p < 2^383) -- the
 * "no-carry" product gnark-crypto's amd64 assembly implements.  The instruction sequence is the
 * one the product's host layer generates for its own serial group operations
 * (go-curdleproofs_amd/csrc/gen_mont_x86.py -> mont_x86_64.inc, included here as data: the
 * baseline borrows the product's best CPU field multiplier rather than a slower one; results
 * are checked against the independent curdle_oracle.c).  ~2.7x the speed of the portable
 * __int128 form below, which stays as the fallback for builds without BMI2 + ADX. */
#if defined(__BMI2__) && defined(__ADX__)
static inline void fp_mul(fp* r, const fp* a, const fp* b) {
  static const u64 N0 = 0x89f3fffcfffcfffdull;
  const u64* x = a->l;
  const u64* y = b->l;
  u64 t0, t1, t2, t3, t4, t5, A, ax, bx, dx;
  __asm__(
#include "../go-curdleproofs_amd/csrc/mont_x86_64.inc"
      : [t0] "=&r"(t0), [t1] "=&r"(t1), [t2] "=&r"(t2), [t3] "=&r"(t3), [t4] "=&r"(t4), [t5] "=&r"(t5), [A] "=&r"(A),
        [ax] "=&r"(ax), [bx] "=&r"(bx), "=&d"(dx)
      : [x] "r"(x), [y] "r"(y), [p] "r"(P), [ninv] "m"(N0), "m"(*(const u64(*)[6])x), "m"(*(const u64(*)[6])y),
        "m"(*(const u64(*)[6])P)
      : "cc");
  const u64 t[6] = {t0, t1, t2, t3, t4, t5};
  fp_reduce(r, t);
}
#else
static inline void fp_mul(fp* r, const fp* a, const fp* b) {
  u64 t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
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
    const u64 m = t[0] * P_INV;
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
    t[7] = 0;
  }
  fp_reduce(r, t); /* t < 2p */
}
#endif
static inline void fp_sqr(fp* r, const fp* a) { fp_mul(r, a, a); }
static inline void fp_dbl(fp* r, const fp* a) { fp_add(r, a, a); }

static void fp_inv(fp* r, const fp* a) { /* a^(p-2), plain square-and-multiply: called once per MSM */
  fp acc, base = *a;
  memcpy(acc.l, ONE, 48);
  u64 e[6];
  memcpy(e, P, 48);
  e[0] -= 2;
  for (int i = 0; i < 384; i++) {
    if ((e[i >> 6] >> (i & 63)) & 1) fp_mul(&acc, &acc, &base);
    fp_sqr(&base, &base);
  }
  *r = acc;
}

/* ------------------------------------------------------------------ G1 --- */
static inline void x_set_inf(xyzz* p) { memset(p, 0, sizeof(*p)); }
static inline int x_is_inf(const xyzz* p) { return fp_is_zero(&p->zz); }

static void x_dbl(xyzz* p) { /* dbl-2008-s-1 */
  if (x_is_inf(p)) return;
  fp u, v, w, s, m, t, x3;
  fp_dbl(&u, &p->y);
  fp_sqr(&v, &u);
  fp_mul(&w, &u, &v);
  fp_mul(&s, &p->x, &v);
  fp_sqr(&t, &p->x);
  fp_dbl(&m, &t);
  fp_add(&m, &m, &t);
  fp_sqr(&x3, &m);
  fp_sub(&x3, &x3, &s);
  fp_sub(&x3, &x3, &s);
  fp_sub(&t, &s, &x3);
  fp_mul(&t, &m, &t);
  fp_mul(&u, &w, &p->y);
  fp_sub(&p->y, &t, &u);
  p->x = x3;
  fp_mul(&p->zz, &v, &p->zz);
  fp_mul(&p->zzz, &w, &p->zzz);
}
static void x_dbl_affine(xyzz* r, const fp* x1, const fp* y1) { /* mdbl-2008-s-1 */
  fp u, v, w, s, m, t;
  fp_dbl(&u, y1);
  fp_sqr(&v, &u);
  fp_mul(&w, &u, &v);
  fp_mul(&s, x1, &v);
  fp_sqr(&t, x1);
  fp_dbl(&m, &t);
  fp_add(&m, &m, &t);
  fp_sqr(&r->x, &m);
  fp_sub(&r->x, &r->x, &s);
  fp_sub(&r->x, &r->x, &s);
  fp_sub(&t, &s, &r->x);
  fp_mul(&t, &m, &t);
  fp_mul(&u, &w, y1);
  fp_sub(&r->y, &t, &u);
  r->zz = v;
  r->zzz = w;
}
/* acc += (x2, y2) affine, not infinity; neg: subtract instead.  madd-2008-s */
static void x_madd(xyzz* acc, const fp* x2, const fp* y2in, int neg) {
  fp y2 = *y2in;
  if (neg) fp_neg(&y2, y2in);
  if (x_is_inf(acc)) {
    acc->x = *x2;
    acc->y = y2;
    memcpy(acc->zz.l, ONE, 48);
    memcpy(acc->zzz.l, ONE, 48);
    return;
  }
  fp p, r, pp, ppp, q, t;
  fp_mul(&p, x2, &acc->zz);
  fp_sub(&p, &p, &acc->x);
  fp_mul(&r, &y2, &acc->zzz);
  fp_sub(&r, &r, &acc->y);
  if (fp_is_zero(&p)) {
    if (fp_is_zero(&r))
      x_dbl_affine(acc, x2, &y2);
    else
      x_set_inf(acc);
    return;
  }
  fp_sqr(&pp, &p);
  fp_mul(&ppp, &p, &pp);
  fp_mul(&q, &acc->x, &pp);
  fp_sqr(&t, &r);
  fp_sub(&t, &t, &ppp);
  fp_sub(&t, &t, &q);
  fp_sub(&t, &t, &q); /* X3 */
  fp_sub(&q, &q, &t);
  fp_mul(&q, &r, &q);
  fp_mul(&acc->y, &acc->y, &ppp);
  fp_sub(&acc->y, &q, &acc->y);
  acc->x = t;
  fp_mul(&acc->zz, &acc->zz, &pp);
  fp_mul(&acc->zzz, &acc->zzz, &ppp);
}
/* acc += b.  add-2008-s */
static void x_add(xyzz* acc, const xyzz* b) {
  if (x_is_inf(b)) return;
  if (x_is_inf(acc)) {
    *acc = *b;
    return;
  }
  fp u1, u2, s1, s2, p, r, pp, ppp, q, t;
  fp_mul(&u1, &acc->x, &b->zz);
  fp_mul(&u2, &b->x, &acc->zz);
  fp_mul(&s1, &acc->y, &b->zzz);
  fp_mul(&s2, &b->y, &acc->zzz);
  fp_sub(&p, &u2, &u1);
  fp_sub(&r, &s2, &s1);
  if (fp_is_zero(&p)) {
    if (fp_is_zero(&r))
      x_dbl(acc);
    else
      x_set_inf(acc);
    return;
  }
  fp_sqr(&pp, &p);
  fp_mul(&ppp, &p, &pp);
  fp_mul(&q, &u1, &pp);
  fp_sqr(&t, &r);
  fp_sub(&t, &t, &ppp);
  fp_sub(&t, &t, &q);
  fp_sub(&t, &t, &q);
  fp_sub(&q, &q, &t);
  fp_mul(&q, &r, &q);
  fp_mul(&s1, &s1, &ppp);
  fp_sub(&acc->y, &q, &s1);
  acc->x = t;
  fp_mul(&acc->zz, &acc->zz, &b->zz);
  fp_mul(&acc->zz, &acc->zz, &pp);
  fp_mul(&acc->zzz, &acc->zzz, &b->zzz);
  fp_mul(&acc->zzz, &acc->zzz, &ppp);
}

/* fr.Element (Montgomery) -> the integer: one Montgomery reduction */
static void fr_from_mont(u64 out[4], const u64 in[4]) {
  u64 t[5] = {in[0], in[1], in[2], in[3], 0};
  for (int i = 0; i < 4; i++) {
    const u64 m = t[0] * R_INV64;
    u128 s = (u128)m * R_MOD[0] + t[0];
    u64 c = (u64)(s >> 64);
    for (int j = 1; j < 4; j++) {
      s = (u128)m * R_MOD[j] + t[j] + c;
      t[j - 1] = (u64)s;
      c = (u64)(s >> 64);
    }
    s = (u128)t[4] + c;
    t[3] = (u64)s;
    t[4] = (u64)(s >> 64);
  }
  u64 d[4];
  unsigned char b = 0;
  for (int i = 0; i < 4; i++) b = _subborrow_u64(b, t[i], R_MOD[i], (unsigned long long*)&d[i]);
  for (int i = 0; i < 4; i++) out[i] = b ? t[i] : d[i];
}

/* --------------------------------------------------------------- tasks --- */
typedef struct {
  const aff* pts;
  const int32_t* digits; /* [W][n] signed digits */
  size_t n;
  int c, W;
  int chunks;            /* tasks per window */
  xyzz* buckets;         /* [W * chunks][nb] */
  size_t nb;
  xyzz* winsum;          /* [W] */
  int next;              /* work counter (atomic) */
  int phase;
} job_t;

static void* worker(void* arg) {
  job_t* j = (job_t*)arg;
  if (j->phase == 0) { /* bucket accumulation: one (window, chunk) per task */
    for (;;) {
      const int t = __atomic_fetch_add(&j->next, 1, __ATOMIC_RELAXED);
      if (t >= j->W * j->chunks) break;
      const int w = t / j->chunks, ch = t % j->chunks;
      const size_t lo = j->n * (size_t)ch / j->chunks, hi = j->n * (size_t)(ch + 1) / j->chunks;
      xyzz* b = j->buckets + (size_t)t * j->nb;
      memset(b, 0, j->nb * sizeof(xyzz));
      const int32_t* d = j->digits + (size_t)w * j->n;
      for (size_t i = lo; i < hi; i++) {
        const int32_t v = d[i];
        if (!v) continue;
        const aff* p = &j->pts[i];
        if (fp_is_zero(&p->x) && fp_is_zero(&p->y)) continue;
        if (i + 4 < hi) __builtin_prefetch(&b[(size_t)(abs(d[i + 4]) - 1) & (j->nb - 1)]);
        x_madd(&b[(size_t)(v < 0 ? -v : v) - 1], &p->x, &p->y, v < 0);
      }
    }
  } else { /* merge the chunks' buckets and run the window's running sum */
    for (;;) {
      const int w = __atomic_fetch_add(&j->next, 1, __ATOMIC_RELAXED);
      if (w >= j->W) break;
      xyzz run, sum;
      x_set_inf(&run);
      x_set_inf(&sum);
      for (size_t k = j->nb; k-- > 0;) {
        for (int ch = 0; ch < j->chunks; ch++) x_add(&run, &j->buckets[((size_t)w * j->chunks + ch) * j->nb + k]);
        x_add(&sum, &run);
      }
      j->winsum[w] = sum;
    }
  }
  return NULL;
}

static void run_phase(job_t* j, int phase, int threads) {
  j->phase = phase;
  j->next = 0;
  pthread_t th[256];
  if (threads > 256) threads = 256;
  for (int i = 1; i < threads; i++) pthread_create(&th[i], NULL, worker, j);
  worker(j);
  for (int i = 1; i < threads; i++) pthread_join(th[i], NULL);
}

/* out = sum_i scalars[i] * points[i] as the canonical Jacobian (x, y, 1) / (1, 1, 0), gnark layouts.
 * c = 0 picks the window width (16 from 2^18 pairs up).  Returns 0, or -1 on allocation failure. */
int fast_msm_g1(const u64* points, const u64* scalars, size_t n, int threads, int c, u64 out[18]) {
  if (threads < 1) threads = 1;
  if (c == 0) {
    int lg = 0;
    while (((size_t)1 << (lg + 1)) <= n) lg++;
    c = lg - 2;
    if (c < 4) c = 4;
    if (c > 16) c = 16;
  }
  const int W = (255 + c) / c + 0; /* one extra top window absorbs the last carry when 255 % c == 0 */
  const size_t nb = (size_t)1 << (c - 1);
  job_t j;
  memset(&j, 0, sizeof(j));
  j.pts = (const aff*)points;
  j.n = n;
  j.c = c;
  j.W = W;
  j.nb = nb;
  j.chunks = threads > W ? (threads + W - 1) / W : 1;
  int32_t* digits = (int32_t*)malloc((size_t)W * (n ? n : 1) * sizeof(int32_t));
  j.buckets = (xyzz*)malloc((size_t)W * j.chunks * nb * sizeof(xyzz));
  j.winsum = (xyzz*)malloc((size_t)W * sizeof(xyzz));
  if (!digits || !j.buckets || !j.winsum) {
    free(digits);
    free(j.buckets);
    free(j.winsum);
    return -1;
  }
  /* signed digits in [-2^(c-1), 2^(c-1)], carry into the next window */
  for (size_t i = 0; i < n; i++) {
    u64 k[5] = {0, 0, 0, 0, 0};
    fr_from_mont(k, scalars + 4 * i);
    int carry = 0;
    for (int w = 0; w < W; w++) {
      const int bit = w * c;
      u64 v = bit < 256 ? (k[bit >> 6] >> (bit & 63)) : 0;
      if ((bit & 63) + c > 64 && (bit >> 6) + 1 < 5) v |= k[(bit >> 6) + 1] << (64 - (bit & 63));
      int d = (int)(v & (((u64)1 << c) - 1)) + carry;
      carry = 0;
      if (d > (1 << (c - 1))) {
        d -= 1 << c;
        carry = 1;
      }
      digits[(size_t)w * n + i] = d;
    }
  }
  j.digits = digits;
  run_phase(&j, 0, threads);
  run_phase(&j, 1, threads < W ? threads : W);
  /* Horner over the windows */
  xyzz acc;
  x_set_inf(&acc);
  for (int w = W - 1; w >= 0; w--) {
    for (int k = 0; k < c; k++) x_dbl(&acc);
    x_add(&acc, &j.winsum[w]);
  }
  fp one;
  memcpy(one.l, ONE, 48);
  if (x_is_inf(&acc)) {
    memcpy(out, ONE, 48);
    memcpy(out + 6, ONE, 48);
    memset(out + 12, 0, 48);
  } else {
    fp t, inv, izz, izzz, x, y;
    fp_mul(&t, &acc.zz, &acc.zzz);
    fp_inv(&inv, &t);
    fp_mul(&izz, &inv, &acc.zzz);
    fp_mul(&izzz, &inv, &acc.zz);
    fp_mul(&x, &acc.x, &izz);
    fp_mul(&y, &acc.y, &izzz);
    memcpy(out, x.l, 48);
    memcpy(out + 6, y.l, 48);
    memcpy(out + 12, ONE, 48);
  }
  free(digits);
  free(j.buckets);
  free(j.winsum);
  return 0;
}
