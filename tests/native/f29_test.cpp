// Host differential test of the lazy 9x29-bit field / curve code (fq29.cuh, ec29.cuh) against the
// canonical 8x32-bit code (fp.cuh, ec.cuh), which is itself pinned against the Python oracle.
// Built with -DG16_F29_CHECK: every F29 result asserts its limb and value bounds.
//   g++ -O2 -std=c++17 -DG16_F29_CHECK -I nzcp-circom_amd/csrc tests/native/f29_test.cpp -o f29_test
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ec29.cuh"
#include "fr29.cuh"

using namespace g16;

static uint64_t rng_s = 0x9E3779B97F4A7C15ull;
static uint64_t rnd() { rng_s ^= rng_s << 13; rng_s ^= rng_s >> 7; rng_s ^= rng_s << 17; return rng_s; }
static Fq rand_fq() {
  Fq a;
  for (int i = 0; i < 8; i++) a.v[i] = (uint32_t)rnd();
  a.v[7] &= 0x1fffffffu;  // < 2^253 < p
  return a;
}
#define CHECK(c) do { if (!(c)) { printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #c); exit(1); } } while (0)

template <class F29ops, class FC> static bool same_point(const XYZZ<F29ops>& a, const XYZZ<FC>& b) {
  XYZZ<FC> ac;
  x29_to_canon<F29ops, FC>(ac, a);
  Affine<FC> pa, pb;
  xyzz_to_affine(pa, ac);
  xyzz_to_affine(pb, b);
  return FC::eq(pa.x, pb.x) && FC::eq(pa.y, pb.y);
}

template <class F29ops, class FC> static void curve_test(const Affine<FC>& gen, const char* name, int iters) {
  // a pool of points k*G
  const int NP = 64;
  Affine<FC> pool[NP];
  Affine<F29ops> pool29[NP];
  for (int i = 0; i < NP; i++) {
    uint32_t k[8];
    for (int j = 0; j < 8; j++) k[j] = (uint32_t)rnd();
    k[7] &= 0x0fffffffu;
    XYZZ<FC> t;
    xyzz_mul_scalar(t, gen, k);
    xyzz_to_affine(pool[i], t);
    a29_from_canon<F29ops, FC>(pool29[i], pool[i]);
    PackedAffine<F29ops> pk;           // the accumulate kernel adds the fast formula on unpacked points
    a29_pack(pk, pool29[i]);
    Affine<F29ops> back;
    a29_unpack(back, pk);
    CHECK(memcmp(&back, &pool29[i], sizeof(back)) == 0);
  }
  {  // x29_madd_fast == x29_madd away from the exceptional cases; flags a doubling and a cancellation
    XYZZ<F29ops> a1, a2;
    x29_set_inf(a1);
    x29_madd(a1, pool29[0]);
    a2 = a1;
    for (int i = 1; i < NP; i++) {
      x29_madd(a1, pool29[i]);
      CHECK(!x29_madd_fast(a2, pool29[i]));
      XYZZ<FC> c1;
      x29_to_canon<F29ops, FC>(c1, a1);
      CHECK((same_point<F29ops, FC>(a2, c1)));
    }
    XYZZ<F29ops> d;
    x29_set_inf(d);
    x29_madd(d, pool29[3]);
    XYZZ<F29ops> d2 = d;
    CHECK(x29_madd_fast(d2, pool29[3]));            // P + P
    Affine<F29ops> neg = pool29[3];
    a29_neg(neg);
    d2 = d;
    CHECK(x29_madd_fast(d2, neg));                   // P + (-P)
  }
  XYZZ<FC> acc, other;
  XYZZ<F29ops> acc29, other29;
  xyzz_set_inf(acc); x29_set_inf(acc29);
  xyzz_set_inf(other); x29_set_inf(other29);
  for (int it = 0; it < iters; it++) {
    const int op = rnd() % 16, i = rnd() % NP;
    Affine<FC> q = pool[i];
    Affine<F29ops> q29 = pool29[i];
    if (rnd() & 1) { aff_neg(q); a29_neg(q29); }
    if (op < 9) { xyzz_madd(acc, q); x29_madd(acc29, q29); }
    else if (op < 11) { xyzz_madd(other, q); x29_madd(other29, q29); }
    else if (op < 13) { xyzz_add(acc, other); x29_add(acc29, other29); }
    else if (op == 13) { xyzz_dbl(acc); x29_dbl(acc29); }
    else if (op == 14) {  // exceptional: acc + acc (via add) and acc + (-acc)
      XYZZ<FC> c = acc; XYZZ<F29ops> c29 = acc29;
      xyzz_add(acc, c); x29_add(acc29, c29);
    } else {
      // madd of the SAME affine point twice in a row from infinity: doubling inside madd, then P + (-P)
      XYZZ<FC> t; XYZZ<F29ops> t29;
      xyzz_set_inf(t); x29_set_inf(t29);
      xyzz_madd(t, q); x29_madd(t29, q29);
      xyzz_madd(t, q); x29_madd(t29, q29);
      CHECK((same_point<F29ops, FC>(t29, t)));
      aff_neg(q); a29_neg(q29);
      xyzz_madd(t, q); x29_madd(t29, q29);
      xyzz_madd(t, q); x29_madd(t29, q29);
      CHECK(xyzz_is_inf(t) && x29_is_inf(t29));
      xyzz_add(other, t); x29_add(other29, t29);
    }
    if ((it & 63) == 0 || it == iters - 1) {
      CHECK((same_point<F29ops, FC>(acc29, acc)));
      CHECK((same_point<F29ops, FC>(other29, other)));
    }
  }
  printf("%s: %d random curve ops agree\n", name, iters);
}

int main() {
  // ---- field
  for (int it = 0; it < 200000; it++) {
    const Fq a = rand_fq(), b = rand_fq();
    const F29 A = f29_from_fq(a), B = f29_from_fq(b);
    CHECK(fp_eq(f29_to_fq(A), a));
    {  // resident storage: nine 29-bit limbs <-> eight 32-bit words, lossless below 2^256
      const F29 U = f29_unpack(f29_pack(A));
      for (int i = 0; i < 9; i++) CHECK(U.l[i] == A.l[i]);
    }
    CHECK(fp_eq(f29_to_fq(f29_mul(A, B)), fp_mul(a, b)));
    CHECK(fp_eq(f29_to_fq(f29_sqr(A)), fp_mul(a, a)));
    CHECK(fp_eq(f29_to_fq(f29_sqr_mul(A, B, f29_add(A, B))), fp_add(fp_mul(a, a), fp_mul(b, fp_add(a, b)))));
    CHECK(fp_eq(f29_to_fq(f29_add(A, B)), fp_add(a, b)));
    CHECK(fp_eq(f29_to_fq(f29_sub<2>(A, B)), fp_sub(a, b)));
    CHECK(fp_eq(f29_to_fq(f29_neg<2>(A)), fp_neg(a)));
    // lazy chain: ((a+b)+(a+b)) * (a - b + 8p) and fused two-product form
    const F29 s = f29_add(f29_add(A, B), f29_add(A, B));
    const F29 d = f29_sub<8>(A, B);
    CHECK(fp_eq(f29_to_fq(f29_mul(s, d)), fp_mul(fp_dbl(fp_add(a, b)), fp_sub(a, b))));
    CHECK(fp_eq(f29_to_fq(f29_sqr(d)), fp_mul(fp_sub(a, b), fp_sub(a, b))));
    CHECK(fp_eq(f29_to_fq(f29_mul2(A, B, s, d)), fp_add(fp_mul(a, b), fp_mul(fp_dbl(fp_add(a, b)), fp_sub(a, b)))));
    CHECK(f29_is_zero(f29_sub<2>(A, A)) && f29_maybe_zero<7>(f29_sub<6>(A, A)));
    CHECK(!f29_is_zero(f29_sub<2>(A, B)));
  }
  {  // edge values
    Fq z = fp_zero<FqParams>(), one = fp_one<FqParams>();
    Fq pm1 = fp_neg(fp_from_mont(one));  // p - 1 as a plain residue image
    const Fq e[4] = {z, one, pm1, fp_neg(one)};
    for (int i = 0; i < 4; i++)
      for (int j = 0; j < 4; j++) {
        const F29 A = f29_from_fq(e[i]), B = f29_from_fq(e[j]);
        CHECK(fp_eq(f29_to_fq(f29_mul(A, B)), fp_mul(e[i], e[j])));
        CHECK(fp_eq(f29_to_fq(f29_sub<2>(A, B)), fp_sub(e[i], e[j])));
        CHECK(fp_eq(f29_to_fq(f29_add(A, B)), fp_add(e[i], e[j])));
      }
    CHECK(f29_is_zero(f29_from_fq(z)) && f29_is_literal_zero(f29_zero()));
    CHECK(fp_eq(f29_to_fq(f29_one()), one));
  }
  // ---- Fr in the same format: conversions, weak reduction, NTT-style lazy chains
  for (int it = 0; it < 100000; it++) {
    Fr a, b;
    for (int i = 0; i < 8; i++) { a.v[i] = (uint32_t)rnd(); b.v[i] = (uint32_t)rnd(); }
    a.v[7] &= 0x1fffffffu; b.v[7] &= 0x1fffffffu;       // canonical residues (< 2^253 < r)
    const F29 A = fr29_from_fr(a), B = fr29_from_fr(b);
    CHECK(fp_eq(fr29_to_fr(A), a));
    CHECK(fp_eq(fr29_to_fr(fr29_mul(A, B)), fp_mul(a, b)));
    CHECK(fp_eq(fr29_to_fr(fr29_sub<2>(A, B)), fp_sub(a, b)));
    // plain <-> Mont261: from_plain(x) represents x, to_plain inverts it
    CHECK(fp_eq(fr29_to_plain(fr29_from_plain(a)), a));
    CHECK(fp_eq(fr29_to_fr(fr29_from_plain(a)), fp_to_mont(a)));
    // zkey coefficient path: file word = coef*R256^2 (plain); product with plain witness word
    const Fr coefm = fp_to_mont(a);                 // Montgomery(coef)
    const Fr file = fp_to_mont(coefm);              // coef * R256^2 as stored by snarkjs
    const F29 t = fr29_mul(fr29_from_zkey_coef(file), fr29_repack(b));
    CHECK(fp_eq(fr29_to_fr(t), fp_mul(coefm, fp_to_mont(b))));   // Mont(coef * w)
    // DIF-like growth then weak reduction: ((A+B)+(A+B))+... up to < 16r
    F29 s = fr29_add(A, B);
    Fr sc = fp_add(a, b);
    for (int k = 0; k < 2; k++) { s = fr29_add(s, s); sc = fp_dbl(sc); }   // < 8r
    F29 d = fr29_sub<8>(A, s);                                            // A + 8r - s < 9.1r
    Fr dc = fp_sub(a, sc);
    const F29 wr = fr29_weak_reduce(d);
    CHECK(wr.l[8] <= Fr29C::P[8] + 1);                                    // < ~1.0001 r
    CHECK(fp_eq(fr29_to_fr(wr), dc));
    CHECK(fp_eq(fr29_to_fr(fr29_mul(d, s)), fp_mul(dc, sc)));
    CHECK(fp_eq(fr29_to_fr(fr29_weak_reduce(s)), sc));
  }
  {
    Fr z = fp_zero<FrParams>();
    CHECK(fp_eq(fr29_to_plain(fr29_weak_reduce(fr29_from_plain(z))), z));
    Fr one = fp_zero<FrParams>(); one.v[0] = 1;
    CHECK(fp_eq(fr29_to_plain(fr29_pow_u64(fr29_from_plain(one), 12345)), one));
    Fr three = fp_zero<FrParams>(); three.v[0] = 3;
    Fr e243 = fp_zero<FrParams>(); e243.v[0] = 243;
    CHECK(fp_eq(fr29_to_plain(fr29_pow_u64(fr29_from_plain(three), 5)), e243));
  }
  printf("field: ok\n");
  // ---- curves
  G1Affine g1;
  g1.x = fp_one<FqParams>();
  g1.y = fp_add(g1.x, g1.x);
  curve_test<Fq29Ops, FqOps>(g1, "G1", 60000);
  G2Affine g2;
  g2.x.a = Fq{G16_G2X0}; g2.x.b = Fq{G16_G2X1}; g2.y.a = Fq{G16_G2Y0}; g2.y.b = Fq{G16_G2Y1};
  curve_test<Fq2x29Ops, Fq2Ops>(g2, "G2", 20000);
  printf("ALL OK\n");
  return 0;
}
