/*
 * ORACLE (test infrastructure, NOT product code) -- plain-C CPU restatement of snarkjs 0.4.12
 * `groth16.prove` for BN254.  Used only by tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg, as the checker and as the timed CPU baseline ("kind": "port").
 *
 * PARITY UNPINNED against the reference: /root/reference holds no prover source or golden proof;
 * the algorithm lives in un-vendored npm packages pinned by /root/reference/yarn.lock
 * (snarkjs 0.4.12 :987-1001, ffjavascript 0.2.48 :408-416, wasmcurves 0.1.0 :1132-1138).
 * This file follows the published pipeline recorded in SURVEY.md section 3.3 / App. A-C:
 *   buildABC1 -> 3 x (ifft, batchApplyKey(w_2N), fft) -> joinABC -> 5 x multiExpAffine ->
 *   blinding -> affine.  It is pinned by tests/test_cpu_oracle_c.py against the Python big-int
 *   oracle (bit-exact), the trapdoor known-answer and the committed golden fixtures.
 *
 * Deliberately independent of the product code: 4 x 64-bit limbs with unsigned __int128 (the
 * product uses 8 x 32-bit), Jacobian accumulators like wasmcurves (the product uses XYZZ),
 * unsigned windows over per-thread point chunks like ffjavascript's multiexp (the product uses
 * signed digits + counting sort), bit-reversal + in-place radix-2 NTT.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef unsigned __int128 u128;
typedef struct { uint64_t v[4]; } fe;            /* Montgomery residue, R = 2^256 */
typedef struct { uint64_t p[4], r2[4], one[4], inv; } field;

static const field FQ = {
  {0x3c208c16d87cfd47ULL, 0x97816a916871ca8dULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL},
  {0xf32cfc5b538afa89ULL, 0xb5e71911d44501fbULL, 0x47ab1eff0a417ff6ULL, 0x06d89f71cab8351fULL},
  {0xd35d438dc58f0d9dULL, 0x0a78eb28f5c70b3dULL, 0x666ea36f7879462cULL, 0x0e0a77c19a07df2fULL},
  0x87d20782e4866389ULL};
static const field FR = {
  {0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL},
  {0x1bb8e645ae216da7ULL, 0x53fe3ab1e35c59e3ULL, 0x8c49833d53bb8085ULL, 0x0216d0b17f4e44a5ULL},
  {0xac96341c4ffffffbULL, 0x36fc76959f60cd29ULL, 0x666ea36f7879462eULL, 0x0e0a77c19a07df2fULL},
  0xc2e1f593efffffffULL};

static inline int fe_is_zero(const fe* a) { return (a->v[0] | a->v[1] | a->v[2] | a->v[3]) == 0; }
static inline int fe_eq(const fe* a, const fe* b) {
  return ((a->v[0] ^ b->v[0]) | (a->v[1] ^ b->v[1]) | (a->v[2] ^ b->v[2]) | (a->v[3] ^ b->v[3])) == 0;
}
static inline int ge_p(const uint64_t a[4], const uint64_t p[4]) {
  for (int i = 3; i >= 0; i--) { if (a[i] > p[i]) return 1; if (a[i] < p[i]) return 0; }
  return 1;
}
static inline void sub_p(uint64_t a[4], const uint64_t p[4]) {
  u128 br = 0;
  for (int i = 0; i < 4; i++) { u128 d = (u128)a[i] - p[i] - (uint64_t)br; a[i] = (uint64_t)d; br = (d >> 64) & 1; }
}
static inline void fe_add(const field* F, fe* r, const fe* a, const fe* b) {
  u128 c = 0;
  for (int i = 0; i < 4; i++) { c += (u128)a->v[i] + b->v[i]; r->v[i] = (uint64_t)c; c >>= 64; }
  if (ge_p(r->v, F->p)) sub_p(r->v, F->p);
}
static inline void fe_sub(const field* F, fe* r, const fe* a, const fe* b) {
  u128 br = 0; uint64_t t[4];
  for (int i = 0; i < 4; i++) { u128 d = (u128)a->v[i] - b->v[i] - (uint64_t)br; t[i] = (uint64_t)d; br = (d >> 64) & 1; }
  if (br) { u128 c = 0; for (int i = 0; i < 4; i++) { c += (u128)t[i] + F->p[i]; t[i] = (uint64_t)c; c >>= 64; } }
  memcpy(r->v, t, 32);
}
static inline void fe_neg(const field* F, fe* r, const fe* a) {
  if (fe_is_zero(a)) { *r = *a; return; }
  fe z = {{F->p[0], F->p[1], F->p[2], F->p[3]}};
  u128 br = 0;
  for (int i = 0; i < 4; i++) { u128 d = (u128)z.v[i] - a->v[i] - (uint64_t)br; r->v[i] = (uint64_t)d; br = (d >> 64) & 1; }
}
/* Montgomery product (operand scanning, 4 limbs) */
static inline void fe_mul(const field* F, fe* r, const fe* a, const fe* b) {
  uint64_t t[6] = {0, 0, 0, 0, 0, 0};
  for (int i = 0; i < 4; i++) {
    u128 c = 0;
    for (int j = 0; j < 4; j++) { c += (u128)a->v[j] * b->v[i] + t[j]; t[j] = (uint64_t)c; c >>= 64; }
    c += t[4]; t[4] = (uint64_t)c; t[5] = (uint64_t)(c >> 64);
    uint64_t m = t[0] * F->inv;
    c = (u128)m * F->p[0] + t[0]; c >>= 64;
    for (int j = 1; j < 4; j++) { c += (u128)m * F->p[j] + t[j]; t[j - 1] = (uint64_t)c; c >>= 64; }
    c += t[4]; t[3] = (uint64_t)c; t[4] = t[5] + (uint64_t)(c >> 64);
  }
  if (t[4] || ge_p(t, F->p)) sub_p(t, F->p);
  memcpy(r->v, t, 32);
}
static inline void fe_sqr(const field* F, fe* r, const fe* a) { fe_mul(F, r, a, a); }
static void fe_pow(const field* F, fe* r, const fe* a, const uint64_t e[4]) {
  fe acc; memcpy(acc.v, F->one, 32);
  for (int i = 255; i >= 0; i--) {
    fe_sqr(F, &acc, &acc);
    if ((e[i >> 6] >> (i & 63)) & 1) fe_mul(F, &acc, &acc, a);
  }
  *r = acc;
}
static void fe_inv(const field* F, fe* r, const fe* a) {
  uint64_t e[4] = {F->p[0] - 2, F->p[1], F->p[2], F->p[3]};
  fe_pow(F, r, a, e);
}
static inline void fe_to_mont(const field* F, fe* r, const fe* a) { fe r2; memcpy(r2.v, F->r2, 32); fe_mul(F, r, a, &r2); }
static inline void fe_from_mont(const field* F, fe* r, const fe* a) { fe one = {{1, 0, 0, 0}}; fe_mul(F, r, a, &one); }

/* ------------------------------------------------------------------ Fq2 = Fq[u]/(u^2+1) */
typedef struct { fe a, b; } fe2;
static inline void f2_add(fe2* r, const fe2* x, const fe2* y) { fe_add(&FQ, &r->a, &x->a, &y->a); fe_add(&FQ, &r->b, &x->b, &y->b); }
static inline void f2_sub(fe2* r, const fe2* x, const fe2* y) { fe_sub(&FQ, &r->a, &x->a, &y->a); fe_sub(&FQ, &r->b, &x->b, &y->b); }
static inline void f2_mul(fe2* r, const fe2* x, const fe2* y) {
  fe aa, bb, ab, ba; fe2 o;
  fe_mul(&FQ, &aa, &x->a, &y->a); fe_mul(&FQ, &bb, &x->b, &y->b);
  fe_mul(&FQ, &ab, &x->a, &y->b); fe_mul(&FQ, &ba, &x->b, &y->a);
  fe_sub(&FQ, &o.a, &aa, &bb); fe_add(&FQ, &o.b, &ab, &ba);
  *r = o;
}
static inline int f2_is_zero(const fe2* x) { return fe_is_zero(&x->a) && fe_is_zero(&x->b); }
static void f2_inv(fe2* r, const fe2* x) {
  fe t0, t1, d; fe2 o;
  fe_sqr(&FQ, &t0, &x->a); fe_sqr(&FQ, &t1, &x->b); fe_add(&FQ, &t0, &t0, &t1); fe_inv(&FQ, &d, &t0);
  fe_mul(&FQ, &o.a, &x->a, &d); fe_mul(&FQ, &t1, &x->b, &d); fe_neg(&FQ, &o.b, &t1);
  *r = o;
}

/* ------------------------------------------------------------------ curves (Jacobian, a = 0), generic via macros */
#define DEFINE_CURVE(PFX, T, ADD, SUB, MUL, ISZ, ONE_INIT)                                              \
  typedef struct { T x, y; } PFX##_aff;                                                                  \
  typedef struct { T x, y, z; } PFX##_jac;                                                               \
  static inline int PFX##_aff_is_inf(const PFX##_aff* p) { return ISZ(&p->x) && ISZ(&p->y); }            \
  static inline void PFX##_set_inf(PFX##_jac* p) { memset(p, 0, sizeof(*p)); }                            \
  static inline int PFX##_is_inf(const PFX##_jac* p) { return ISZ(&p->z); }                               \
  static void PFX##_dbl(PFX##_jac* r, const PFX##_jac* p) {                                               \
    if (PFX##_is_inf(p)) { *r = *p; return; }                                                             \
    T A, B, C, D, E, F, t, X3, Y3, Z3;                                                                    \
    MUL(&A, &p->x, &p->x); MUL(&B, &p->y, &p->y); MUL(&C, &B, &B);                                        \
    ADD(&t, &p->x, &B); MUL(&t, &t, &t); SUB(&t, &t, &A); SUB(&t, &t, &C); ADD(&D, &t, &t);               \
    ADD(&E, &A, &A); ADD(&E, &E, &A); MUL(&F, &E, &E);                                                    \
    SUB(&X3, &F, &D); SUB(&X3, &X3, &D);                                                                  \
    SUB(&t, &D, &X3); MUL(&Y3, &E, &t); ADD(&C, &C, &C); ADD(&C, &C, &C); ADD(&C, &C, &C); SUB(&Y3, &Y3, &C); \
    MUL(&Z3, &p->y, &p->z); ADD(&Z3, &Z3, &Z3);                                                           \
    r->x = X3; r->y = Y3; r->z = Z3;                                                                      \
  }                                                                                                       \
  static void PFX##_add(PFX##_jac* r, const PFX##_jac* p, const PFX##_jac* q) {                           \
    if (PFX##_is_inf(p)) { *r = *q; return; }                                                             \
    if (PFX##_is_inf(q)) { *r = *p; return; }                                                             \
    T Z1Z1, Z2Z2, U1, U2, S1, S2, H, Rr, HH, HHH, V, t, X3, Y3, Z3;                                       \
    MUL(&Z1Z1, &p->z, &p->z); MUL(&Z2Z2, &q->z, &q->z);                                                   \
    MUL(&U1, &p->x, &Z2Z2); MUL(&U2, &q->x, &Z1Z1);                                                       \
    MUL(&S1, &p->y, &q->z); MUL(&S1, &S1, &Z2Z2); MUL(&S2, &q->y, &p->z); MUL(&S2, &S2, &Z1Z1);           \
    SUB(&H, &U2, &U1); SUB(&Rr, &S2, &S1);                                                                \
    if (ISZ(&H)) { if (ISZ(&Rr)) { PFX##_dbl(r, p); } else { PFX##_set_inf(r); } return; }                \
    MUL(&HH, &H, &H); MUL(&HHH, &H, &HH); MUL(&V, &U1, &HH);                                              \
    MUL(&X3, &Rr, &Rr); SUB(&X3, &X3, &HHH); SUB(&X3, &X3, &V); SUB(&X3, &X3, &V);                        \
    SUB(&t, &V, &X3); MUL(&Y3, &Rr, &t); MUL(&t, &S1, &HHH); SUB(&Y3, &Y3, &t);                           \
    MUL(&Z3, &p->z, &q->z); MUL(&Z3, &Z3, &H);                                                            \
    r->x = X3; r->y = Y3; r->z = Z3;                                                                      \
  }                                                                                                       \
  static void PFX##_madd(PFX##_jac* r, const PFX##_jac* p, const PFX##_aff* q) {                          \
    if (PFX##_aff_is_inf(q)) { *r = *p; return; }                                                         \
    PFX##_jac qq; qq.x = q->x; qq.y = q->y; T one = ONE_INIT; qq.z = one;                                 \
    PFX##_add(r, p, &qq);                                                                                 \
  }                                                                                                       \
  static void PFX##_mul_scalar(PFX##_jac* r, const PFX##_aff* p, const uint64_t k[4]) {                   \
    PFX##_jac acc; PFX##_set_inf(&acc);                                                                   \
    for (int i = 255; i >= 0; i--) {                                                                      \
      PFX##_dbl(&acc, &acc);                                                                              \
      if ((k[i >> 6] >> (i & 63)) & 1) PFX##_madd(&acc, &acc, p);                                         \
    }                                                                                                     \
    *r = acc;                                                                                             \
  }

#define FQ_ADD(r, a, b) fe_add(&FQ, r, a, b)
#define FQ_SUB(r, a, b) fe_sub(&FQ, r, a, b)
#define FQ_MUL(r, a, b) fe_mul(&FQ, r, a, b)
#define FQ_ONE {{0xd35d438dc58f0d9dULL, 0x0a78eb28f5c70b3dULL, 0x666ea36f7879462cULL, 0x0e0a77c19a07df2fULL}}
#define FQ2_ONE {FQ_ONE, {{0, 0, 0, 0}}}
DEFINE_CURVE(g1, fe, FQ_ADD, FQ_SUB, FQ_MUL, fe_is_zero, FQ_ONE)
DEFINE_CURVE(g2, fe2, f2_add, f2_sub, f2_mul, f2_is_zero, FQ2_ONE)

static void g1_to_affine(g1_aff* r, const g1_jac* p) {
  if (g1_is_inf(p)) { memset(r, 0, sizeof(*r)); return; }
  fe zi, zi2, zi3; fe_inv(&FQ, &zi, &p->z); fe_sqr(&FQ, &zi2, &zi); fe_mul(&FQ, &zi3, &zi2, &zi);
  fe_mul(&FQ, &r->x, &p->x, &zi2); fe_mul(&FQ, &r->y, &p->y, &zi3);
}
static void g2_to_affine(g2_aff* r, const g2_jac* p) {
  if (g2_is_inf(p)) { memset(r, 0, sizeof(*r)); return; }
  fe2 zi, zi2, zi3; f2_inv(&zi, &p->z); f2_mul(&zi2, &zi, &zi); f2_mul(&zi3, &zi2, &zi);
  f2_mul(&r->x, &p->x, &zi2); f2_mul(&r->y, &p->y, &zi3);
}

/* ------------------------------------------------------------------ MSM: ffjavascript-style chunks x unsigned windows */
static inline uint32_t get_bits(const uint64_t s[4], int pos, int c) {
  if (pos >= 256) return 0;
  int w = pos >> 6, o = pos & 63;
  uint64_t v = s[w] >> o;
  if (o + c > 64 && w + 1 < 4) v |= s[w + 1] << (64 - o);
  return (uint32_t)(v & ((1u << c) - 1));
}
#define DEFINE_MSM(PFX)                                                                                   \
  static void PFX##_msm_chunk(PFX##_jac* out, const PFX##_aff* bases, const uint64_t* scalars /*4 per*/,  \
                              size_t n, int c) {                                                          \
    const int nwin = (254 + c - 1) / c;                                                                   \
    const size_t nb = (size_t)1 << c;                                                                     \
    PFX##_jac* buckets = (PFX##_jac*)malloc(nb * sizeof(PFX##_jac));                                      \
    PFX##_jac total; PFX##_set_inf(&total);                                                               \
    for (int w = nwin - 1; w >= 0; w--) {                                                                 \
      for (int k = 0; k < c; k++) PFX##_dbl(&total, &total);                                              \
      memset(buckets, 0, nb * sizeof(PFX##_jac));                                                         \
      for (size_t i = 0; i < n; i++) {                                                                    \
        uint32_t d = get_bits(scalars + 4 * i, w * c, c);                                                 \
        if (d) PFX##_madd(&buckets[d], &buckets[d], &bases[i]);                                           \
      }                                                                                                   \
      PFX##_jac run, acc; PFX##_set_inf(&run); PFX##_set_inf(&acc);                                       \
      for (size_t d = nb - 1; d >= 1; d--) { PFX##_add(&run, &run, &buckets[d]); PFX##_add(&acc, &acc, &run); } \
      PFX##_add(&total, &total, &acc);                                                                    \
    }                                                                                                     \
    free(buckets);                                                                                        \
    *out = total;                                                                                         \
  }                                                                                                       \
  static void PFX##_msm(PFX##_jac* out, const PFX##_aff* bases, const uint64_t* scalars, size_t n, int threads) { \
    PFX##_set_inf(out);                                                                                   \
    if (n == 0) return;                                                                                   \
    size_t nchunk = (size_t)threads * 4;                                                                  \
    if (nchunk > n) nchunk = n;                                                                           \
    size_t per = (n + nchunk - 1) / nchunk;                                                               \
    int c = 4; { size_t t = per; while (t > 40 && c < 16) { t >>= 1; c++; } if (c > 3) c -= 2; if (c < 2) c = 2; } \
    PFX##_jac* parts = (PFX##_jac*)calloc(nchunk, sizeof(PFX##_jac));                                     \
    _Pragma("omp parallel for schedule(dynamic, 1) num_threads(threads)")                                 \
    for (long k = 0; k < (long)nchunk; k++) {                                                             \
      size_t lo = (size_t)k * per, hi = lo + per < n ? lo + per : n;                                      \
      if (lo < hi) PFX##_msm_chunk(&parts[k], bases + lo, scalars + 4 * lo, hi - lo, c);                  \
    }                                                                                                     \
    for (size_t k = 0; k < nchunk; k++) PFX##_add(out, out, &parts[k]);                                   \
    free(parts);                                                                                          \
  }
DEFINE_MSM(g1)
DEFINE_MSM(g2)

/* ------------------------------------------------------------------ NTT over Fr (natural in/out) */
static void fr_root(fe* w, int power) { /* Fr.w[power] = 5^((r-1)/2^28) squared down */
  fe five = {{5, 0, 0, 0}}, t; fe_to_mont(&FR, &t, &five);
  uint64_t e[4]; /* (r-1) >> 28 */
  uint64_t rm1[4] = {FR.p[0] - 1, FR.p[1], FR.p[2], FR.p[3]};
  for (int i = 0; i < 4; i++) e[i] = (rm1[i] >> 28) | (i < 3 ? rm1[i + 1] << 36 : 0);
  fe_pow(&FR, w, &t, e);
  for (int i = 28; i > power; i--) fe_sqr(&FR, w, w);
}
static void ntt(fe* a, int logn, int inverse, int threads) {
  const size_t n = (size_t)1 << logn;
  for (size_t i = 1, j = 0; i < n; i++) {   /* bit reversal */
    size_t bit = n >> 1;
    for (; j & bit; bit >>= 1) j ^= bit;
    j |= bit;
    if (i < j) { fe t = a[i]; a[i] = a[j]; a[j] = t; }
  }
  fe w; fr_root(&w, logn);
  if (inverse) fe_inv(&FR, &w, &w);
  /* twiddle table w^k, k < n/2 */
  fe* tw = (fe*)malloc((n / 2 + 1) * sizeof(fe));
  memcpy(tw[0].v, FR.one, 32);
  for (size_t k = 1; k < n / 2; k++) fe_mul(&FR, &tw[k], &tw[k - 1], &w);
  for (int s = 1; s <= logn; s++) {
    const size_t len = (size_t)1 << s, half = len >> 1, stride = n >> s;
    _Pragma("omp parallel for schedule(static) num_threads(threads)")
    for (long idx = 0; idx < (long)(n / 2); idx++) {
      size_t blk = (size_t)idx / half, k = (size_t)idx % half;
      fe* u = &a[blk * len + k]; fe* v = u + half; fe t;
      fe_mul(&FR, &t, v, &tw[k * stride]);
      fe_sub(&FR, v, u, &t); fe_add(&FR, u, u, &t);
    }
  }
  if (inverse) {
    fe nn = {{n, 0, 0, 0}}, ninv; fe_to_mont(&FR, &nn, &nn); fe_inv(&FR, &ninv, &nn);
    _Pragma("omp parallel for schedule(static) num_threads(threads)")
    for (long i = 0; i < (long)n; i++) fe_mul(&FR, &a[i], &a[i], &ninv);
  }
  free(tw);
}

/* ------------------------------------------------------------------ containers */
typedef struct { const uint8_t* p; uint64_t size; } sec_t;
static int find_sections(const uint8_t* buf, size_t len, const char* magic, sec_t* secs, int maxid) {
  if (len < 12 || memcmp(buf, magic, 4) != 0) return -1;
  uint32_t nsec; memcpy(&nsec, buf + 8, 4);
  size_t pos = 12;
  for (int i = 0; i <= maxid; i++) { secs[i].p = NULL; secs[i].size = 0; }
  for (uint32_t i = 0; i < nsec; i++) {
    if (pos + 12 > len) return -1;
    uint32_t id; uint64_t sz; memcpy(&id, buf + pos, 4); memcpy(&sz, buf + pos + 4, 8); pos += 12;
    if (sz > len - pos) return -1;
    if ((int)id <= maxid && !secs[id].p) { secs[id].p = buf + pos; secs[id].size = sz; }
    pos += sz;
  }
  return 0;
}

/* proof_out: A.x|A.y|B.x0|B.x1|B.y0|B.y1|C.x|C.y, standard form LE (256 bytes); pub_out: nPublic*32 */
int g16o_prove(const uint8_t* zkey, size_t zlen, const uint8_t* wtns, size_t wlen, const uint8_t r_in[32],
               const uint8_t s_in[32], uint8_t* proof_out, uint8_t* pub_out, int threads) {
  sec_t zs[11], ws[3];
  if (threads < 1) threads = 1;
  if (find_sections(zkey, zlen, "zkey", zs, 10) || find_sections(wtns, wlen, "wtns", ws, 2)) return -1;
  for (int i = 1; i <= 9; i++) if (!zs[i].p) return -2;
  if (!ws[1].p || !ws[2].p) return -2;
  uint32_t proto; memcpy(&proto, zs[1].p, 4);
  if (proto != 1) return -3;
  const uint8_t* h = zs[2].p + 72;
  uint32_t nVars, nPublic, N; memcpy(&nVars, h, 4); memcpy(&nPublic, h + 4, 4); memcpy(&N, h + 8, 4); h += 12;
  uint32_t nw; memcpy(&nw, ws[1].p + 36, 4);
  if (nw != nVars) return -4;
  int logn = 0; while ((1u << logn) < N) logn++;
  g1_aff alpha1, beta1, delta1; g2_aff beta2, delta2;
  memcpy(&alpha1, h, 64); memcpy(&beta1, h + 64, 64); memcpy(&beta2, h + 128, 128);
  memcpy(&delta1, h + 384, 64); memcpy(&delta2, h + 448, 128);
  const uint64_t* w = (const uint64_t*)ws[2].p;   /* standard form, 4 x u64 per signal (x86: unaligned ok via memcpy below) */
  uint64_t* wcopy = (uint64_t*)malloc((size_t)nVars * 32);
  memcpy(wcopy, w, (size_t)nVars * 32);

  /* buildABC1 */
  fe* A = (fe*)calloc(N, sizeof(fe)); fe* B = (fe*)calloc(N, sizeof(fe)); fe* C = (fe*)calloc(N, sizeof(fe));
  uint32_t ncoef; memcpy(&ncoef, zs[4].p, 4);
  const uint8_t* recs = zs[4].p + 4;
  _Pragma("omp parallel num_threads(threads)")
  {
    int tid = 0, nt = 1;
#ifdef _OPENMP
    tid = omp_get_thread_num(); nt = omp_get_num_threads();
#endif
    uint32_t lo = (uint32_t)((uint64_t)N * tid / nt), hi = (uint32_t)((uint64_t)N * (tid + 1) / nt);
    for (uint32_t i = 0; i < ncoef; i++) {
      const uint8_t* rec = recs + (size_t)i * 44;
      uint32_t m, c, s; memcpy(&c, rec + 4, 4);
      if (c < lo || c >= hi) continue;
      memcpy(&m, rec, 4); memcpy(&s, rec + 8, 4);
      fe coef, ww, t; memcpy(coef.v, rec + 12, 32); memcpy(ww.v, wcopy + 4 * (size_t)s, 32);
      fe_mul(&FR, &t, &coef, &ww);
      fe* dst = m == 0 ? &A[c] : &B[c];
      fe_add(&FR, dst, dst, &t);
    }
  }
  _Pragma("omp parallel for num_threads(threads)")
  for (long i = 0; i < (long)N; i++) fe_mul(&FR, &C[i], &A[i], &B[i]);
  /* 3 x (ifft, shift by w_2N^i, fft) */
  fe inc; fr_root(&inc, logn + 1);
  fe* vecs[3] = {A, B, C};
  fe* incs = (fe*)malloc((size_t)N * sizeof(fe));
  memcpy(incs[0].v, FR.one, 32);
  for (size_t i = 1; i < N; i++) fe_mul(&FR, &incs[i], &incs[i - 1], &inc);
  for (int k = 0; k < 3; k++) {
    ntt(vecs[k], logn, 1, threads);
    _Pragma("omp parallel for num_threads(threads)")
    for (long i = 0; i < (long)N; i++) fe_mul(&FR, &vecs[k][i], &vecs[k][i], &incs[i]);
    ntt(vecs[k], logn, 0, threads);
  }
  free(incs);
  /* joinABC + fromMontgomery */
  uint64_t* P = (uint64_t*)malloc((size_t)N * 32);
  _Pragma("omp parallel for num_threads(threads)")
  for (long i = 0; i < (long)N; i++) {
    fe t; fe_mul(&FR, &t, &A[i], &B[i]); fe_sub(&FR, &t, &t, &C[i]); fe_from_mont(&FR, &t, &t);
    memcpy(P + 4 * i, t.v, 32);
  }
  free(A); free(B); free(C);
  /* five multiexps */
  g1_jac mA, mB1, mC, mH; g2_jac mB2;
  g1_msm(&mA, (const g1_aff*)zs[5].p, wcopy, nVars, threads);
  g1_msm(&mB1, (const g1_aff*)zs[6].p, wcopy, nVars, threads);
  g2_msm(&mB2, (const g2_aff*)zs[7].p, wcopy, nVars, threads);
  g1_msm(&mC, (const g1_aff*)zs[8].p, wcopy + 4 * ((size_t)nPublic + 1), nVars - nPublic - 1, threads);
  g1_msm(&mH, (const g1_aff*)zs[9].p, P, N, threads);
  free(P);
  /* blinding */
  uint64_t r[4], s[4]; memcpy(r, r_in, 32); memcpy(s, s_in, 32);
  g1_jac pa, pb1, pc, t1; g2_jac pb, t2;
  g1_madd(&pa, &mA, &alpha1); g1_mul_scalar(&t1, &delta1, r); g1_add(&pa, &pa, &t1);
  g2_madd(&pb, &mB2, &beta2); g2_mul_scalar(&t2, &delta2, s); g2_add(&pb, &pb, &t2);
  g1_madd(&pb1, &mB1, &beta1); g1_mul_scalar(&t1, &delta1, s); g1_add(&pb1, &pb1, &t1);
  g1_aff a_aff, b1_aff, c_aff; g2_aff b_aff;
  g1_to_affine(&a_aff, &pa); g1_to_affine(&b1_aff, &pb1); g2_to_affine(&b_aff, &pb);
  g1_add(&pc, &mC, &mH);
  g1_mul_scalar(&t1, &a_aff, s); g1_add(&pc, &pc, &t1);
  g1_mul_scalar(&t1, &b1_aff, r); g1_add(&pc, &pc, &t1);
  fe rm, sm, rs; memcpy(rm.v, r, 32); memcpy(sm.v, s, 32);
  fe_to_mont(&FR, &rm, &rm); fe_to_mont(&FR, &sm, &sm); fe_mul(&FR, &rs, &rm, &sm); fe_neg(&FR, &rs, &rs);
  fe_from_mont(&FR, &rs, &rs);
  g1_mul_scalar(&t1, &delta1, rs.v); g1_add(&pc, &pc, &t1);
  g1_to_affine(&c_aff, &pc);
  fe o[8];
  fe_from_mont(&FQ, &o[0], &a_aff.x); fe_from_mont(&FQ, &o[1], &a_aff.y);
  fe_from_mont(&FQ, &o[2], &b_aff.x.a); fe_from_mont(&FQ, &o[3], &b_aff.x.b);
  fe_from_mont(&FQ, &o[4], &b_aff.y.a); fe_from_mont(&FQ, &o[5], &b_aff.y.b);
  fe_from_mont(&FQ, &o[6], &c_aff.x); fe_from_mont(&FQ, &o[7], &c_aff.y);
  memcpy(proof_out, o, 256);
  if (pub_out && nPublic) memcpy(pub_out, wcopy + 4, (size_t)nPublic * 32);
  free(wcopy);
  return 0;
}
