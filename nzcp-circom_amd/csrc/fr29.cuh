// BN254 Fr in the 9 x 29-bit lazy format (see fq29.cuh) -- the QAP and NTT kernels' arithmetic.
//
// NTT butterflies add and subtract many times between products, so values drift upwards:
//   DIT:  t = v*w (< 1.1r);  u' = u + t;  v' = u + 2r - t      -> +2r per stage
//   DIF:  u' = u + v;  v' = (u + K r - v) * w                   -> x2 per stage on the sum branch
// fr29_weak_reduce brings any value below 16r back below 1.0001r with one 64-bit multiply for the
// quotient estimate and one limb-wise addition of the table entry 2^261 - q*r (mod 2^261); the NTT
// applies it on load and every 3 (DIF) / 7 (DIT) stages -- bounds in ntt.hip.
#pragma once
#include "fq29.cuh"

namespace g16 {

struct Fr29T {
  static constexpr uint32_t R2[9] = G16_FR29_R2;        // 2^522 mod r
  static constexpr uint32_t QAPK[9] = G16_FR29_QAPK;    // 2^783 / 2^512 mod r
  static constexpr uint32_t RECIP = G16_FR29_RECIP;
  static constexpr uint32_t NEGQ[17][9] = G16_FR29_NEGQ;
};

G16_HD F29 fr29_mul(const F29& a, const F29& b) { return f29_mul<Fr29C>(a, b); }
G16_HD F29 fr29_add(const F29& a, const F29& b) { return f29_add<Fr29C>(a, b); }
template <int K> G16_HD F29 fr29_sub(const F29& a, const F29& b) { return f29_sub<K, Fr29C>(a, b); }
G16_HD F29 fr29_one() { return f29_one<Fr29C>(); }

#if defined(__HIP_DEVICE_COMPILE__)
__device__ __constant__ uint32_t g_fr29_negq[17][9] = G16_FR29_NEGQ;
#endif

// x < 16r  ->  x - q*r < 1.0001 r, q = floor(top limb * RECIP / 2^40) <= floor(x / r)
G16_HD F29 fr29_weak_reduce(const F29& x) {
  const uint32_t q = (uint32_t)(((uint64_t)x.l[8] * Fr29T::RECIP) >> 40);   // 0..16
  F29 r;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
  for (int i = 0; i < 9; i++) r.l[i] = x.l[i] + g_fr29_negq[q][i];
#else
  for (int i = 0; i < 9; i++) r.l[i] = x.l[i] + Fr29T::NEGQ[q][i];
#endif
  f29_carry(r);
  r.l[8] &= kM29;   // drop 2^261
  r.pad_ = 0;
  return r;
}

// plain (standard-form, canonical 8 x 32) <-> Mont261
G16_HD F29 fr29_repack(const Fr& v) {   // same integer, 29-bit limbs
  F29 t;
#pragma unroll
  for (int i = 0; i < 9; i++) {
    const int bit = 29 * i, w = bit >> 5, o = bit & 31;
    uint64_t x = v.v[w];
    if (w + 1 < 8) x |= (uint64_t)v.v[w + 1] << 32;
    t.l[i] = (uint32_t)(x >> o) & (i < 8 ? kM29 : 0xffffffffu);
  }
  t.pad_ = 0;
  return t;
}
G16_HD F29 fr29_from_plain(const Fr& v) {
  F29 c;
#pragma unroll
  for (int i = 0; i < 9; i++) c.l[i] = Fr29T::R2[i];
  return fr29_mul(fr29_repack(v), c);
}
// Mont261 (any value < 16r) -> canonical plain integer as 8 x 32
G16_HD Fr fr29_to_plain(const F29& a) {
  F29 one = f29_zero();
  one.l[0] = 1;
  F29 t = fr29_mul(a, one);   // x mod r, in [0, r]
  uint32_t d[9];
  int32_t br = 0;
#pragma unroll
  for (int i = 0; i < 9; i++) {
    const int32_t x = (int32_t)t.l[i] - (int32_t)Fr29C::P[i] + br;
    d[i] = (uint32_t)x & (i < 8 ? kM29 : 0xffffffffu);
    br = (i < 8) ? (x >> 29) : (x < 0 ? -1 : 0);
  }
  const bool ge = (br == 0);
  Fr r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.v[i] = 0;
#pragma unroll
  for (int i = 0; i < 9; i++) {
    const uint32_t li = ge ? d[i] : t.l[i];
    const int bit = 29 * i, w = bit >> 5, o = bit & 31;
    r.v[w] |= li << o;
    if (o > 3 && w + 1 < 8) r.v[w + 1] |= li >> (32 - o);
  }
  return r;
}
// canonical Mont256 image (fp.cuh Fr) <-> Mont261
G16_HD F29 fr29_from_fr(const Fr& v) { return f29_from_fq<Fr29C, FrParams>(v); }
G16_HD Fr fr29_to_fr(const F29& a) { return f29_to_fq<Fr29C, FrParams>(a); }
// zkey section-4 word (coef * 2^512 as a plain integer) -> coef * 2^522, so that one product with
// the PLAIN witness word gives Mont261(coef * w)
G16_HD F29 fr29_from_zkey_coef(const Fr& v) {
  F29 c;
#pragma unroll
  for (int i = 0; i < 9; i++) c.l[i] = Fr29T::QAPK[i];
  return fr29_mul(fr29_repack(v), c);
}
G16_HD F29 fr29_pow_u64(const F29& a, uint64_t e) {
  F29 r = fr29_one();
  for (int i = 63; i >= 0; i--) {
    r = fr29_mul(r, r);
    if ((e >> i) & 1) r = fr29_mul(r, a);
  }
  return r;
}

}  // namespace g16
