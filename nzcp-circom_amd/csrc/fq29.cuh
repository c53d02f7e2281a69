// BN254 Fq in 9 x 29-bit limbs with lazy reduction -- the MSM kernels' arithmetic.
//
// Why (measured, profiles/r01_microbench_int_rates.txt): on gfx950 a v_mad_u64_u32 costs ~5.5
// cycles per wave-instruction, but so does every carry add (v_add_co/v_addc ~4.5 each,
// v_lshl_add_u64 ~4.6): the 8 x 32-bit CIOS product of fp.cuh spends more cycles on carries and on
// building register pairs than on multiplying (129 mad + 129 add64 + 275 mov).  With 29-bit limbs a
// product column sums 18 terms below 2^58 in ONE 64-bit accumulator without overflow, so the whole
// Montgomery product is 162 in-place v_mad_u64_u32 plus ~45 cheap ops, and additions are plain
// limb-wise v_add_u32 followed by one carry ripple.
//
// Representation: value = sum l[i] * 2^(29 i); l[0..7] < 2^29 always (every operation ends with a
// full carry ripple), l[8] holds the rest.  Montgomery radix R' = 2^261.  Values are NOT kept below
// p: R' leaves 7.4 spare bits, a product of inputs below 16p comes out below 2.51p (in practice
// ~1.1p), and the few additions/subtractions between products of the curve formulas stay below
// 16p (bound propagation: tools/f29_bounds.py, asserted at run time in the host test build).
// Subtraction a - b is a + K*p - b with K*p stored with inflated limbs (no borrows); K is a
// template parameter chosen per call site from the bound of b.
//
// Replaces nothing in the reference by itself: it is the internal number format of the kernels that
// replace wasmcurves' f1m/curve code (see fp.cuh, ec.cuh); conversion to/from the canonical
// 8 x 32-bit Montgomery(2^256) image happens when bases are uploaded and when window sums leave the GPU.
#pragma once
#include "fp.cuh"

namespace g16 {

struct alignas(8) F29 {   // 9 limbs + 1 pad word: 40 bytes, so points are 16-byte multiples (dwordx4 gathers)
  uint32_t l[9];
  uint32_t pad_;
};

static constexpr uint32_t kM29 = 0x1fffffffu;

template <int K> struct Fq29KP;
struct Fq29C {
  static constexpr uint32_t P[9] = G16_FQ29_P;
  static constexpr uint32_t ONE[9] = G16_FQ29_ONE;
  static constexpr uint32_t TO[9] = G16_FQ29_TO;
  static constexpr uint32_t FROM[9] = G16_FQ29_FROM;
  static constexpr uint32_t INV = G16_FQ29_INV;
  static constexpr uint32_t KPX4[9] = G16_FQ29_KPX4;
  template <int K> using KP = Fq29KP<K>;
};
template <int K> struct Fr29KP;
#define G16_DEF_KP(k) \
  template <> struct Fq29KP<k> { static constexpr uint32_t V[9] = G16_FQ29_KP##k; }; \
  template <> struct Fr29KP<k> { static constexpr uint32_t V[9] = G16_FR29_KP##k; };
G16_DEF_KP(1) G16_DEF_KP(2) G16_DEF_KP(3) G16_DEF_KP(4) G16_DEF_KP(5) G16_DEF_KP(6) G16_DEF_KP(7) G16_DEF_KP(8)
#undef G16_DEF_KP
// the same format for the scalar field (QAP / NTT kernels, fr29.cuh)
struct Fr29C {
  static constexpr uint32_t P[9] = G16_FR29_P;
  static constexpr uint32_t ONE[9] = G16_FR29_ONE;
  static constexpr uint32_t TO[9] = G16_FR29_TO;
  static constexpr uint32_t FROM[9] = G16_FR29_FROM;
  static constexpr uint32_t INV = G16_FR29_INV;
  static constexpr uint32_t KPX4[9] = G16_FR29_KPX4;
  template <int K> using KP = Fr29KP<K>;
};

#if defined(G16_F29_CHECK) && !defined(__HIP_DEVICE_COMPILE__)
#include <assert.h>
// host test build: every result must be below 16p (top limb comparison is enough: 16p < 2^258)
#define G16_F29_ASSERT_BOUND(x) assert((x).l[8] < (16u * C::P[8] + 16u))
#define G16_F29_ASSERT_LIMBS(x) do { for (int _i = 0; _i < 8; _i++) assert((x).l[_i] <= kM29); } while (0)
#else
#define G16_F29_ASSERT_BOUND(x) ((void)0)
#define G16_F29_ASSERT_LIMBS(x) ((void)0)
#endif

G16_HD F29 f29_zero() {
  F29 r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.l[i] = 0;
  return r;
}
template <class C = Fq29C> G16_HD F29 f29_one() {
  F29 r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.l[i] = C::ONE[i];
  return r;
}

// full carry ripple: limbs 0..7 below 2^29 afterwards (inputs: any limbs whose running sums fit 32 bits)
G16_HD void f29_carry(F29& a) {
  uint32_t c = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const uint32_t v = a.l[i] + c;
    a.l[i] = v & kM29;
    c = v >> 29;
  }
  a.l[8] += c;
}

template <class C = Fq29C> G16_HD F29 f29_add(const F29& a, const F29& b) {
  F29 r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.l[i] = a.l[i] + b.l[i];
  f29_carry(r);
  G16_F29_ASSERT_BOUND(r);
  return r;
}
template <class C = Fq29C> G16_HD F29 f29_dbl(const F29& a) { return f29_add<C>(a, a); }

// a + K*p - b; requires value(b) <= K*p
template <int K, class C = Fq29C> G16_HD F29 f29_sub(const F29& a, const F29& b) {
  F29 r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.l[i] = a.l[i] + C::template KP<K>::V[i] - b.l[i];
  f29_carry(r);
  G16_F29_ASSERT_BOUND(r);
  return r;
}
// a + 4p - b - 2c in one pass and one carry ripple (the X3 of the addition formulas: R^2 - PPP - 2Q); requires
// value(b) + 2 value(c) <= 4p; the constant's limbs are inflated by 2^31 to cover limbs of b + 2c below 3 * 2^29
template <class C = Fq29C> G16_HD F29 f29_sub_b_2c(const F29& a, const F29& b, const F29& c) {
  F29 r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.l[i] = a.l[i] + C::KPX4[i] - b.l[i] - 2u * c.l[i];
  f29_carry(r);
  G16_F29_ASSERT_BOUND(r);
  return r;
}
template <int K, class C = Fq29C> G16_HD F29 f29_neg(const F29& b) {
  F29 r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.l[i] = C::template KP<K>::V[i] - b.l[i];
  f29_carry(r);
  return r;
}

// acc + a * b as ONE v_mad_u64_u32 whose addend is the running column sum (-DG16_F29_CHAIN_MADS; off).  Written in C,
// `acc += (uint64_t)a * b` over a column that starts from the previous column's carry, the compiler reassociates: it starts every
// column in a fresh accumulator (a shorter dependency chain) and adds the shifted carry afterwards -- 16 v_lshl_add_u64 per
// product, 7 % of its instructions.  The inline-asm form pins ONE chain: 214 instructions per product instead of 228.  Measured
// (r03, profiles/r03_sweeps.txt 21), in the throughput kernels only (the tail kernels' products keep CHAIN = false: a wavefront
// alone on its SIMD needs the shorter chains -- H bucket reduce 0.43 ms against 0.36 with chained products): every stage alone
// is 0-3 % faster (H accumulate 1.03 -> 1.00 ms, serial sum 6.45 -> 6.32), 768-proof batches +2 % (308 -> 315 proofs/s) -- and a
// single proof in the product schedule 0.2 ms SLOWER (3.55 -> 3.78 ms: every accumulate and the NTT lengthen by 8-25 % when they
// share the chip).  The single proof is the headline; the option stays off.
#if defined(__HIP_DEVICE_COMPILE__) && defined(G16_F29_CHAIN_MADS)
#define G16_F29_CHAIN 1
__device__ __forceinline__ uint64_t f29_mad(uint32_t a, uint32_t b, uint64_t acc) {
  uint64_t r;
  asm("v_mad_u64_u32 %0, vcc, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(acc) : "vcc");
  return r;
}
// ... with a compile-time constant (a limb of p) in a scalar register
__device__ __forceinline__ uint64_t f29_mad_k(uint32_t a, uint32_t k, uint64_t acc) {
  uint64_t r;
  asm("v_mad_u64_u32 %0, vcc, %1, %2, %3" : "=v"(r) : "v"(a), "s"(k), "v"(acc) : "vcc");
  return r;
}
#else
G16_HD uint64_t f29_mad(uint32_t a, uint32_t b, uint64_t acc) { return acc + (uint64_t)a * b; }
G16_HD uint64_t f29_mad_k(uint32_t a, uint32_t k, uint64_t acc) { return acc + (uint64_t)a * k; }
#endif

// Operand-scanning (row-wise) Montgomery product: the same 162 multiply-adds as f29_mul below, but each
// goes to a DIFFERENT 64-bit column accumulator than its neighbours (18 short dependency chains instead
// of 17 long ones), so one wavefront keeps several v_mad_u64_u32 in flight.  Column bound: 9 a*b terms
// + 9 m*p terms < 2^58 each + a carry < 2^36, below 2^63.
template <class C = Fq29C> G16_HD F29 f29_mul_rows(const F29& a, const F29& b) {
  G16_F29_ASSERT_LIMBS(a); G16_F29_ASSERT_LIMBS(b);
  uint64_t t[18];
#pragma unroll
  for (int j = 0; j < 9; j++) t[j] = (uint64_t)a.l[0] * b.l[j];
#pragma unroll
  for (int j = 9; j < 18; j++) t[j] = 0;
#pragma unroll
  for (int i = 0; i < 9; i++) {
    if (i > 0) {
#pragma unroll
      for (int j = 0; j < 9; j++) t[i + j] += (uint64_t)a.l[i] * b.l[j];
    }
    const uint32_t m = ((uint32_t)t[i] * C::INV) & kM29;
#pragma unroll
    for (int j = 0; j < 9; j++) t[i + j] += (uint64_t)m * C::P[j];
    t[i + 1] += t[i] >> 29;
  }
  F29 r;
#pragma unroll
  for (int k = 9; k < 17; k++) {
    r.l[k - 9] = (uint32_t)t[k] & kM29;
    t[k + 1] += t[k] >> 29;
  }
  r.l[8] = (uint32_t)t[17];
  G16_F29_ASSERT_BOUND(r);
  return r;
}
template <class C = Fq29C> G16_HD F29 f29_sqr_rows(const F29& a) {
  G16_F29_ASSERT_LIMBS(a);
  uint64_t t[18];
#pragma unroll
  for (int j = 0; j < 18; j++) t[j] = 0;
#pragma unroll
  for (int i = 0; i < 9; i++) {
    const uint32_t a2 = a.l[i] << 1;
    t[2 * i] += (uint64_t)a.l[i] * a.l[i];
#pragma unroll
    for (int j = i + 1; j < 9; j++) t[i + j] += (uint64_t)a2 * a.l[j];
  }
#pragma unroll
  for (int i = 0; i < 9; i++) {
    const uint32_t m = ((uint32_t)t[i] * C::INV) & kM29;
#pragma unroll
    for (int j = 0; j < 9; j++) t[i + j] += (uint64_t)m * C::P[j];
    t[i + 1] += t[i] >> 29;
  }
  F29 r;
#pragma unroll
  for (int k = 9; k < 17; k++) {
    r.l[k - 9] = (uint32_t)t[k] & kM29;
    t[k + 1] += t[k] >> 29;
  }
  r.l[8] = (uint32_t)t[17];
  G16_F29_ASSERT_BOUND(r);
  return r;
}

// Montgomery product a*b / 2^261 (mod p), product scanning.
// Inputs: limbs < 2^29 (+ top limb), values < 16p.  Output: limbs exact, value < a*b/2^261 + p.
// The a*b terms and the m*p terms of a column go to two independent 64-bit accumulators (each
// < 2^63), so a wave that is alone on its SIMD (the reduce kernels, G2) has two mad chains in
// flight instead of one 162-long dependency chain (measured: 12 cycles/mad dependent vs 5.5 issue).
template <class C = Fq29C, bool CHAIN = true> G16_HD F29 f29_mul(const F29& a, const F29& b) {
  G16_F29_ASSERT_LIMBS(a); G16_F29_ASSERT_LIMBS(b);
#if defined(G16_F29_CHAIN)
  if constexpr (CHAIN) {  // one accumulator, one dependency chain (see f29_mad)
    uint64_t acc = 0;
    uint32_t m[9];
    F29 r;
#pragma unroll
    for (int k = 0; k < 9; k++) {
#pragma unroll
      for (int i = 0; i <= k; i++) acc = f29_mad(a.l[i], b.l[k - i], acc);
#pragma unroll
      for (int i = 0; i < k; i++) acc = f29_mad_k(m[i], C::P[k - i], acc);
      m[k] = ((uint32_t)acc * C::INV) & kM29;
      acc = f29_mad_k(m[k], C::P[0], acc);
      acc >>= 29;
    }
#pragma unroll
    for (int k = 9; k < 17; k++) {
#pragma unroll
      for (int i = k - 8; i <= 8; i++) acc = f29_mad(a.l[i], b.l[k - i], acc);
#pragma unroll
      for (int i = k - 8; i <= 8; i++) acc = f29_mad_k(m[i], C::P[k - i], acc);
      r.l[k - 9] = (uint32_t)acc & kM29;
      acc >>= 29;
    }
    r.l[8] = (uint32_t)acc;
    r.pad_ = 0;
    G16_F29_ASSERT_BOUND(r);
    return r;
  }
#endif
  uint64_t carry = 0;
  uint32_t m[9];
  F29 r;
#pragma unroll
  for (int k = 0; k < 9; k++) {
    uint64_t ab = 0, mp = carry;
#pragma unroll
    for (int i = 0; i <= k; i++) ab += (uint64_t)a.l[i] * b.l[k - i];
#pragma unroll
    for (int i = 0; i < k; i++) mp += (uint64_t)m[i] * C::P[k - i];
    uint64_t acc = ab + mp;
    m[k] = ((uint32_t)acc * C::INV) & kM29;
    acc += (uint64_t)m[k] * C::P[0];
    carry = acc >> 29;
  }
#pragma unroll
  for (int k = 9; k < 17; k++) {
    uint64_t ab = 0, mp = carry;
#pragma unroll
    for (int i = k - 8; i <= 8; i++) ab += (uint64_t)a.l[i] * b.l[k - i];
#pragma unroll
    for (int i = k - 8; i <= 8; i++) mp += (uint64_t)m[i] * C::P[k - i];
    const uint64_t acc = ab + mp;
    r.l[k - 9] = (uint32_t)acc & kM29;
    carry = acc >> 29;
  }
  r.l[8] = (uint32_t)carry;
  G16_F29_ASSERT_BOUND(r);
  return r;
}
// a^2 / 2^261: the cross terms a_i a_j (i < j) are taken once against the doubled limb 2 a_i, so
// 45 product terms instead of 81 (a column still sums below 2^63: 4 * 2^59 + 2^58 + 9 * 2^58).
template <class C = Fq29C, bool CHAIN = true> G16_HD F29 f29_sqr(const F29& a) {
  G16_F29_ASSERT_LIMBS(a);
  uint32_t a2[9];
#pragma unroll
  for (int i = 0; i < 9; i++) a2[i] = a.l[i] << 1;
#if defined(G16_F29_CHAIN)
  if constexpr (CHAIN) {
    uint64_t acc = 0;
    uint32_t m[9];
    F29 r;
#pragma unroll
    for (int k = 0; k < 17; k++) {
      const int lo = k < 9 ? 0 : k - 8, hi = k < 9 ? k : 8;
#pragma unroll
      for (int i = lo; i <= hi; i++) {
        const int j = k - i;
        if (i < j) acc = f29_mad(a2[i], a.l[j], acc);
        else if (i == j) acc = f29_mad(a.l[i], a.l[i], acc);
      }
      if (k < 9) {
#pragma unroll
        for (int i = 0; i < k; i++) acc = f29_mad_k(m[i], C::P[k - i], acc);
        m[k] = ((uint32_t)acc * C::INV) & kM29;
        acc = f29_mad_k(m[k], C::P[0], acc);
      } else {
#pragma unroll
        for (int i = k - 8; i <= 8; i++) acc = f29_mad_k(m[i], C::P[k - i], acc);
        r.l[k - 9] = (uint32_t)acc & kM29;
      }
      acc >>= 29;
    }
    r.l[8] = (uint32_t)acc;
    r.pad_ = 0;
    G16_F29_ASSERT_BOUND(r);
    return r;
  }
#endif
  uint64_t carry = 0;
  uint32_t m[9];
  F29 r;
#pragma unroll
  for (int k = 0; k < 17; k++) {
    uint64_t ab = 0, mp = carry;
    const int lo = k < 9 ? 0 : k - 8, hi = k < 9 ? k : 8;
#pragma unroll
    for (int i = lo; i <= hi; i++) {
      const int j = k - i;
      if (i < j) ab += (uint64_t)a2[i] * a.l[j];
      else if (i == j) ab += (uint64_t)a.l[i] * a.l[i];
    }
    if (k < 9) {
#pragma unroll
      for (int i = 0; i < k; i++) mp += (uint64_t)m[i] * C::P[k - i];
      uint64_t acc = ab + mp;
      m[k] = ((uint32_t)acc * C::INV) & kM29;
      acc += (uint64_t)m[k] * C::P[0];
      carry = acc >> 29;
    } else {
#pragma unroll
      for (int i = k - 8; i <= 8; i++) mp += (uint64_t)m[i] * C::P[k - i];
      const uint64_t acc = ab + mp;
      r.l[k - 9] = (uint32_t)acc & kM29;
      carry = acc >> 29;
    }
  }
  r.l[8] = (uint32_t)carry;
  G16_F29_ASSERT_BOUND(r);
  return r;
}
// (a^2 + c*d) / 2^261 with one reduction (real part of an Fq2 square)
template <class C = Fq29C, bool CHAIN = true> G16_HD F29 f29_sqr_mul(const F29& a, const F29& c, const F29& d) {
  uint32_t a2[9];
#pragma unroll
  for (int i = 0; i < 9; i++) a2[i] = a.l[i] << 1;
#if defined(G16_F29_CHAIN)
  if constexpr (CHAIN) {
    uint64_t acc = 0;
    uint32_t m[9];
    F29 r;
#pragma unroll
    for (int k = 0; k < 17; k++) {
      const int lo = k < 9 ? 0 : k - 8, hi = k < 9 ? k : 8;
#pragma unroll
      for (int i = lo; i <= hi; i++) {
        const int j = k - i;
        if (i < j) acc = f29_mad(a2[i], a.l[j], acc);
        else if (i == j) acc = f29_mad(a.l[i], a.l[i], acc);
        acc = f29_mad(c.l[i], d.l[j], acc);
      }
      if (k < 9) {
#pragma unroll
        for (int i = 0; i < k; i++) acc = f29_mad_k(m[i], C::P[k - i], acc);
        m[k] = ((uint32_t)acc * C::INV) & kM29;
        acc = f29_mad_k(m[k], C::P[0], acc);
      } else {
#pragma unroll
        for (int i = k - 8; i <= 8; i++) acc = f29_mad_k(m[i], C::P[k - i], acc);
        r.l[k - 9] = (uint32_t)acc & kM29;
      }
      acc >>= 29;
    }
    r.l[8] = (uint32_t)acc;
    r.pad_ = 0;
    G16_F29_ASSERT_BOUND(r);
    return r;
  }
#endif
  uint64_t carry = 0;
  uint32_t m[9];
  F29 r;
#pragma unroll
  for (int k = 0; k < 17; k++) {
    uint64_t ab = 0, cd = 0, mp = carry;
    const int lo = k < 9 ? 0 : k - 8, hi = k < 9 ? k : 8;
#pragma unroll
    for (int i = lo; i <= hi; i++) {
      const int j = k - i;
      if (i < j) ab += (uint64_t)a2[i] * a.l[j];
      else if (i == j) ab += (uint64_t)a.l[i] * a.l[i];
      cd += (uint64_t)c.l[i] * d.l[j];
    }
    if (k < 9) {
#pragma unroll
      for (int i = 0; i < k; i++) mp += (uint64_t)m[i] * C::P[k - i];
      uint64_t acc = ab + cd + mp;
      m[k] = ((uint32_t)acc * C::INV) & kM29;
      acc += (uint64_t)m[k] * C::P[0];
      carry = acc >> 29;
    } else {
#pragma unroll
      for (int i = k - 8; i <= 8; i++) mp += (uint64_t)m[i] * C::P[k - i];
      const uint64_t acc = ab + cd + mp;
      r.l[k - 9] = (uint32_t)acc & kM29;
      carry = acc >> 29;
    }
  }
  r.l[8] = (uint32_t)carry;
  G16_F29_ASSERT_BOUND(r);
  return r;
}

// (a*b + c*d) / 2^261 with ONE reduction (Fq2 products): 27 terms < 2^58 per column still fit
// (three independent accumulators: a*b, c*d, m*p).
template <class C = Fq29C, bool CHAIN = true> G16_HD F29 f29_mul2(const F29& a, const F29& b, const F29& c, const F29& d) {
#if defined(G16_F29_CHAIN)
  if constexpr (CHAIN) {
    uint64_t acc = 0;
    uint32_t m[9];
    F29 r;
#pragma unroll
    for (int k = 0; k < 17; k++) {
      const int lo = k < 9 ? 0 : k - 8, hi = k < 9 ? k : 8;
#pragma unroll
      for (int i = lo; i <= hi; i++) {
        acc = f29_mad(a.l[i], b.l[k - i], acc);
        acc = f29_mad(c.l[i], d.l[k - i], acc);
      }
      if (k < 9) {
#pragma unroll
        for (int i = 0; i < k; i++) acc = f29_mad_k(m[i], C::P[k - i], acc);
        m[k] = ((uint32_t)acc * C::INV) & kM29;
        acc = f29_mad_k(m[k], C::P[0], acc);
      } else {
#pragma unroll
        for (int i = k - 8; i <= 8; i++) acc = f29_mad_k(m[i], C::P[k - i], acc);
        r.l[k - 9] = (uint32_t)acc & kM29;
      }
      acc >>= 29;
    }
    r.l[8] = (uint32_t)acc;
    r.pad_ = 0;
    G16_F29_ASSERT_BOUND(r);
    return r;
  }
#endif
  uint64_t carry = 0;
  uint32_t m[9];
  F29 r;
#pragma unroll
  for (int k = 0; k < 9; k++) {
    uint64_t ab = 0, cd = 0, mp = carry;
#pragma unroll
    for (int i = 0; i <= k; i++) {
      ab += (uint64_t)a.l[i] * b.l[k - i];
      cd += (uint64_t)c.l[i] * d.l[k - i];
    }
#pragma unroll
    for (int i = 0; i < k; i++) mp += (uint64_t)m[i] * C::P[k - i];
    uint64_t acc = ab + cd + mp;
    m[k] = ((uint32_t)acc * C::INV) & kM29;
    acc += (uint64_t)m[k] * C::P[0];
    carry = acc >> 29;
  }
#pragma unroll
  for (int k = 9; k < 17; k++) {
    uint64_t ab = 0, cd = 0, mp = carry;
#pragma unroll
    for (int i = k - 8; i <= 8; i++) {
      ab += (uint64_t)a.l[i] * b.l[k - i];
      cd += (uint64_t)c.l[i] * d.l[k - i];
    }
#pragma unroll
    for (int i = k - 8; i <= 8; i++) mp += (uint64_t)m[i] * C::P[k - i];
    const uint64_t acc = ab + cd + mp;
    r.l[k - 9] = (uint32_t)acc & kM29;
    carry = acc >> 29;
  }
  r.l[8] = (uint32_t)carry;
  G16_F29_ASSERT_BOUND(r);
  return r;
}

// x == 0 as an integer
G16_HD bool f29_is_literal_zero(const F29& a) {
  uint32_t o = 0;
#pragma unroll
  for (int i = 0; i < 9; i++) o |= a.l[i];
  return o == 0;
}
// x == 0 (mod p), exact, for any x below 16p: x/2^261 mod p lands in [0, p]
template <class C = Fq29C> G16_HD bool f29_is_zero(const F29& a) {
  F29 one = f29_zero();
  one.l[0] = 1;
  const F29 y = f29_mul<C>(a, one);
  uint32_t z = 0, e = 0;
#pragma unroll
  for (int i = 0; i < 9; i++) { z |= y.l[i]; e |= y.l[i] ^ C::P[i]; }
  return z == 0 || e == 0;
}
// cheap necessary condition for x == 0 (mod p) when x < (KMAX+1)*p: the two low limbs are exact after the carry
// ripple, and x = k*p forces them to the low 58 bits of k*p.  (One limb alone fires on 2^-26 of ordinary values:
// with ~3 x 10^7 bucket additions per MSM launch that is a spurious redo task in every third launch, each
// costing a wavefront of complete additions on the critical chain; two limbs: 2^-55.)
template <int KMAX, class C = Fq29C> G16_HD bool f29_maybe_zero(const F29& a) {
  bool hit = false;
  const uint64_t p01 = (uint64_t)C::P[0] | ((uint64_t)C::P[1] << 29);   // p mod 2^58
#pragma unroll
  for (int k = 0; k <= KMAX; k++) {
    const uint64_t kp = (uint64_t)k * p01;
    hit |= (a.l[0] == (uint32_t)(kp & kM29)) & (a.l[1] == (uint32_t)((kp >> 29) & kM29));
  }
  return hit;
}

// canonical 8 x 32 Montgomery(2^256) image <-> F29 Montgomery(2^261)
template <class C = Fq29C, class PM = FqParams> G16_HD F29 f29_from_fq(const Fp<PM>& v) {
  F29 t;
#pragma unroll
  for (int i = 0; i < 9; i++) {
    const int bit = 29 * i, w = bit >> 5, o = bit & 31;
    uint64_t x = v.v[w];
    if (w + 1 < 8) x |= (uint64_t)v.v[w + 1] << 32;
    t.l[i] = (uint32_t)(x >> o) & (i < 8 ? kM29 : 0xffffffffu);
  }
  F29 c;
#pragma unroll
  for (int i = 0; i < 9; i++) c.l[i] = C::TO[i];
  return f29_mul<C>(t, c);
}
template <class C = Fq29C, class PM = FqParams> G16_HD Fp<PM> f29_to_fq(const F29& a) {
  F29 c;
#pragma unroll
  for (int i = 0; i < 9; i++) c.l[i] = C::FROM[i];
  F29 t = f29_mul<C>(a, c);  // x * 2^256 mod p, in [0, p]
  // t >= p ? t - p : t   (limbs are exact)
  uint32_t d[9];
  int32_t br = 0;
#pragma unroll
  for (int i = 0; i < 9; i++) {
    const int32_t x = (int32_t)t.l[i] - (int32_t)C::P[i] + br;
    d[i] = (uint32_t)x & (i < 8 ? kM29 : 0xffffffffu);
    br = x >> 29;  // arithmetic shift: 0 or -1 (for i < 8)
    if (i == 8) br = x < 0 ? -1 : 0;
  }
  const bool ge = (br == 0);
  Fp<PM> r;
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

// Optional out-of-line copies for the device loops (-DG16_F29_CALLS): fully inlined, one mixed
// addition is ~5.5k instructions (~40 KiB).  Measured on MI355X (r01): calls are NOT faster (serial
// proof 17.5 ms vs 16.3 ms inlined) -- the loop is bound by VALU issue (~3.4k instructions x ~5
// cycles per wave-iteration, profiles/r01_pmc_accumulate.txt), not by instruction fetch.  The
// variant stays because it cuts the G2 compile time by a third.
#if defined(__HIP_DEVICE_COMPILE__) && defined(G16_F29_CALLS)
// limbs travel as scalar arguments so that the ABI keeps them in VGPRs (two by-value structs spill
// the second one to scratch)
#define G16_L9(p) uint32_t p##0, uint32_t p##1, uint32_t p##2, uint32_t p##3, uint32_t p##4, uint32_t p##5, \
                  uint32_t p##6, uint32_t p##7, uint32_t p##8
#define G16_MK9(v, p) F29 v; v.l[0] = p##0; v.l[1] = p##1; v.l[2] = p##2; v.l[3] = p##3; v.l[4] = p##4; \
                      v.l[5] = p##5; v.l[6] = p##6; v.l[7] = p##7; v.l[8] = p##8; v.pad_ = 0
#define G16_X9(x) x.l[0], x.l[1], x.l[2], x.l[3], x.l[4], x.l[5], x.l[6], x.l[7], x.l[8]
__device__ __noinline__ F29 f29_mul_raw(G16_L9(a), G16_L9(b)) { G16_MK9(A, a); G16_MK9(B, b); return f29_mul(A, B); }
__device__ __noinline__ F29 f29_sqr_raw(G16_L9(a)) { G16_MK9(A, a); return f29_sqr(A); }
__device__ __noinline__ F29 f29_mul2_raw(G16_L9(a), G16_L9(b), G16_L9(c), G16_L9(d)) {
  G16_MK9(A, a); G16_MK9(B, b); G16_MK9(Cc, c); G16_MK9(D, d);
  return f29_mul2(A, B, Cc, D);
}
__device__ __noinline__ F29 f29_sqr_mul_raw(G16_L9(a), G16_L9(c), G16_L9(d)) {
  G16_MK9(A, a); G16_MK9(Cc, c); G16_MK9(D, d);
  return f29_sqr_mul(A, Cc, D);
}
__device__ __forceinline__ F29 f29_mul_call(const F29& a, const F29& b) { return f29_mul_raw(G16_X9(a), G16_X9(b)); }
__device__ __forceinline__ F29 f29_sqr_call(const F29& a) { return f29_sqr_raw(G16_X9(a)); }
__device__ __forceinline__ F29 f29_mul2_call(const F29& a, const F29& b, const F29& c, const F29& d) {
  return f29_mul2_raw(G16_X9(a), G16_X9(b), G16_X9(c), G16_X9(d));
}
__device__ __forceinline__ F29 f29_sqr_mul_call(const F29& a, const F29& c, const F29& d) {
  return f29_sqr_mul_raw(G16_X9(a), G16_X9(c), G16_X9(d));
}
#define G16_F29_MUL(a, b) f29_mul_call(a, b)
#define G16_F29_SQR(a) f29_sqr_call(a)
#define G16_F29_MUL2(a, b, c, d) f29_mul2_call(a, b, c, d)
#define G16_F29_SQR_MUL(a, c, d) f29_sqr_mul_call(a, c, d)
#else
#if defined(G16_F29_ROWS)
#define G16_F29_MUL(a, b) f29_mul_rows(a, b)
#define G16_F29_SQR(a) f29_sqr_rows(a)
#else
#define G16_F29_MUL(a, b) f29_mul(a, b)
#define G16_F29_SQR(a) f29_sqr(a)
#endif
#define G16_F29_MUL2(a, b, c, d) f29_mul2(a, b, c, d)
#define G16_F29_SQR_MUL(a, c, d) f29_sqr_mul(a, c, d)
#endif

// ------------------------------------------------------------------ Fq2 over F29
struct alignas(8) F29x2 {
  F29 a, b;  // a + b*u, u^2 = -1
};

// Field-ops bundles for ec29.cuh.  sub/neg carry the multiple of p that covers the subtrahend.
struct Fq29TailOps;
struct Fq2x29TailOps;
struct Fq29Ops {
  using Tail = Fq29TailOps;   // field ops of the latency-bound tail kernels (see the end of this file)
  using T = F29;
  using Canon = Fq;   // canonical twin (fp.cuh)
  using CanonOps = FqOps;
  static constexpr int kAccumWavesPerSimd = 4;   // msm_accumulate: fits 128 VGPRs
  static constexpr int kReduceThreads = 512;     // msm_bucket_reduce_kernel: workgroup cap (two wavefronts per SIMD)
  static G16_HD T zero() { return f29_zero(); }
  static G16_HD T one() { return f29_one(); }
  static G16_HD bool is_literal_zero(const T& x) { return f29_is_literal_zero(x); }
  static G16_HD bool is_zero(const T& x) { return f29_is_zero(x); }
  template <int KMAX> static G16_HD bool maybe_zero(const T& x) { return f29_maybe_zero<KMAX>(x); }
  static G16_HD T add(const T& x, const T& y) { return f29_add(x, y); }
  template <int K> static G16_HD T sub(const T& x, const T& y) { return f29_sub<K>(x, y); }
  template <int K> static G16_HD T neg(const T& x) { return f29_neg<K>(x); }
  static G16_HD T sub_b_2c(const T& a, const T& b, const T& c) { return f29_sub_b_2c(a, b, c); }
  static constexpr bool kFusedY3 = true;          // Y3 = R (Q - X3) + (-Y1) PPP with ONE Montgomery reduction
  static G16_HD T mul_add(const T& a, const T& b, const T& c, const T& d) { return G16_F29_MUL2(a, b, c, d); }
  static G16_HD T mul(const T& x, const T& y) { return G16_F29_MUL(x, y); }
  static G16_HD T sqr(const T& x) { return G16_F29_SQR(x); }
  static G16_HD T from_canon(const Fq& x) { return f29_from_fq(x); }
  static G16_HD Fq to_canon(const T& x) { return f29_to_fq(x); }
};

struct Fq2x29Ops {
  using Tail = Fq2x29TailOps;
  using T = F29x2;
  using Canon = Fq2;
  using CanonOps = Fq2Ops;
  static constexpr int kAccumWavesPerSimd = 2;
  static constexpr int kReduceThreads = 256;     // one wavefront per SIMD: its accumulators need AGPRs
  static G16_HD T zero() { return T{f29_zero(), f29_zero()}; }
  static G16_HD T one() { return T{f29_one(), f29_zero()}; }
  static G16_HD bool is_literal_zero(const T& x) { return f29_is_literal_zero(x.a) && f29_is_literal_zero(x.b); }
  static G16_HD bool is_zero(const T& x) { return f29_is_zero(x.a) && f29_is_zero(x.b); }
  template <int KMAX> static G16_HD bool maybe_zero(const T& x) {
    return f29_maybe_zero<KMAX>(x.a) && f29_maybe_zero<KMAX>(x.b);
  }
  static G16_HD T add(const T& x, const T& y) { return T{f29_add(x.a, y.a), f29_add(x.b, y.b)}; }
  template <int K> static G16_HD T sub(const T& x, const T& y) {
    return T{f29_sub<K>(x.a, y.a), f29_sub<K>(x.b, y.b)};
  }
  template <int K> static G16_HD T neg(const T& x) { return T{f29_neg<K>(x.a), f29_neg<K>(x.b)}; }
  static G16_HD T sub_b_2c(const T& a, const T& b, const T& c) {
    return T{f29_sub_b_2c(a.a, b.a, c.a), f29_sub_b_2c(a.b, b.b, c.b)};
  }
  static constexpr bool kFusedY3 = false;         // would need a four-product fused reduction per component
  // (a0 b0 - a1 b1) + (a0 b1 + a1 b0) u, two fused reductions; components of x below 8p
  static G16_HD T mul(const T& x, const T& y) {
    const F29 nb = f29_neg<8>(x.b);
    return T{G16_F29_MUL2(x.a, y.a, nb, y.b), G16_F29_MUL2(x.a, y.b, x.b, y.a)};
  }
  // (a^2 - b^2) + (2ab) u: one fused square+product reduction and one single product
  static G16_HD T sqr(const T& x) {
    return T{G16_F29_SQR_MUL(x.a, f29_neg<8>(x.b), x.b), G16_F29_MUL(x.a, f29_dbl(x.b))};
  }
  static G16_HD T from_canon(const Fq2& x) { return T{f29_from_fq(x.a), f29_from_fq(x.b)}; }
  static G16_HD Fq2 to_canon(const T& x) { return Fq2{f29_to_fq(x.a), f29_to_fq(x.b)}; }
};


// Field ops of the latency-bound MSM tail kernels (combine, bucket reduce, trees).  They are the inlined ops.  Measured and
// dropped: calling ONE out-of-line copy of the PRODUCTS (r02: G2 tails 10-25 % slower), and calling one copy of the
// complete point addition / doubling per code object so that the hot code of every tail kernel fits the instruction
// cache (r03: the kernels shrink from 0.5-1.2 MB to 18-33 KB + a 32 KB / 17 KB pair of functions, and a proof gets
// 0.2 ms SLOWER: profiles/r03_sweeps.txt).
// r03: under -DG16_F29_CHAIN_MADS (see f29_mad) their PRODUCTS keep the two-accumulator form (CHAIN = false): a wavefront alone
// on its SIMD does gain from the shorter dependency chains inside a product -- with the single chain the H bucket reduce took
// 0.43 ms instead of 0.36, the G2 lane's 0.84 instead of 0.63.
struct Fq29TailOps : Fq29Ops {
  static G16_HD T mul_add(const T& a, const T& b, const T& c, const T& d) { return f29_mul2<Fq29C, false>(a, b, c, d); }
  static G16_HD T mul(const T& x, const T& y) { return f29_mul<Fq29C, false>(x, y); }
  static G16_HD T sqr(const T& x) { return f29_sqr<Fq29C, false>(x); }
};
struct Fq2x29TailOps : Fq2x29Ops {
  static G16_HD T mul(const T& x, const T& y) {
    const F29 nb = f29_neg<8>(x.b);
    return T{f29_mul2<Fq29C, false>(x.a, y.a, nb, y.b), f29_mul2<Fq29C, false>(x.a, y.b, x.b, y.a)};
  }
  static G16_HD T sqr(const T& x) {
    return T{f29_sqr_mul<Fq29C, false>(x.a, f29_neg<8>(x.b), x.b), f29_mul<Fq29C, false>(x.a, f29_dbl(x.b))};
  }
};

}  // namespace g16
