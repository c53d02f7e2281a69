// 256-bit Montgomery prime-field arithmetic for BN254 Fq / Fr, 8 x 32-bit limbs, R = 2^256.
//
// Replaces wasmcurves 0.1.0 build_f1m.js + build_int.js
// (pin /root/reference/yarn.lock:1132-1138): same value domain and the same in-memory byte
// image (little-endian Montgomery residues), so zkey sections upload to HBM unmodified.
//
// The functions are __host__ __device__: the device build is the product hot path; the host
// build serves the O(1) proof tail, the zkey parser and the trapdoor setup tool -- and lets the
// CPU test-suite exercise exactly the arithmetic the kernels run.
//
// gfx950 notes: every 32x32->64 multiply-accumulate below lowers to v_mad_u64_u32
// (checked in the .s); values are kept fully reduced in [0, p) between operations.
#pragma once
#include <stdint.h>
#include "bn254_consts.h"

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define G16_HD __host__ __device__ __forceinline__
#else
#define G16_HD inline
#endif

namespace g16 {

struct FqParams {
  static constexpr uint32_t P[8] = G16_FQ_P;
  static constexpr uint32_t ONE[8] = G16_FQ_ONE;
  static constexpr uint32_t R2[8] = G16_FQ_R2;
  static constexpr uint32_t INV = G16_FQ_INV;
};
struct FrParams {
  static constexpr uint32_t P[8] = G16_FR_P;
  static constexpr uint32_t ONE[8] = G16_FR_ONE;
  static constexpr uint32_t R2[8] = G16_FR_R2;
  static constexpr uint32_t INV = G16_FR_INV;
};

template <class PM>
struct alignas(16) Fp {
  uint32_t v[8];
};
using Fq = Fp<FqParams>;
using Fr = Fp<FrParams>;

template <class PM> G16_HD Fp<PM> fp_zero() {
  Fp<PM> r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.v[i] = 0;
  return r;
}
template <class PM> G16_HD Fp<PM> fp_one() {
  Fp<PM> r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.v[i] = PM::ONE[i];
  return r;
}
template <class PM> G16_HD bool fp_is_zero(const Fp<PM>& a) {
  uint32_t o = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) o |= a.v[i];
  return o == 0;
}
template <class PM> G16_HD bool fp_eq(const Fp<PM>& a, const Fp<PM>& b) {
  uint32_t o = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) o |= a.v[i] ^ b.v[i];
  return o == 0;
}

// r = a - p if a >= p else a   (a < 2p)
template <class PM> G16_HD void fp_reduce_once(Fp<PM>& a) {
  uint32_t d[8];
  int64_t br = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    br += (int64_t)a.v[i] - (int64_t)PM::P[i];
    d[i] = (uint32_t)br;
    br >>= 32;  // arithmetic: 0 or -1
  }
  bool ge = (br == 0);
#pragma unroll
  for (int i = 0; i < 8; i++) a.v[i] = ge ? d[i] : a.v[i];
}

template <class PM> G16_HD Fp<PM> fp_add(const Fp<PM>& a, const Fp<PM>& b) {
  Fp<PM> r;
  uint64_t c = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    c += (uint64_t)a.v[i] + b.v[i];
    r.v[i] = (uint32_t)c;
    c >>= 32;
  }
  // p < 2^254 so a + b < 2^255: no carry out of limb 7
  fp_reduce_once(r);
  return r;
}

template <class PM> G16_HD Fp<PM> fp_sub(const Fp<PM>& a, const Fp<PM>& b) {
  Fp<PM> r;
  int64_t br = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    br += (int64_t)a.v[i] - (int64_t)b.v[i];
    r.v[i] = (uint32_t)br;
    br >>= 32;
  }
  uint32_t mask = (uint32_t)br;  // all ones when a < b
  uint64_t c = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    c += (uint64_t)r.v[i] + (PM::P[i] & mask);
    r.v[i] = (uint32_t)c;
    c >>= 32;
  }
  return r;
}

template <class PM> G16_HD Fp<PM> fp_neg(const Fp<PM>& a) {
  if (fp_is_zero(a)) return a;
  Fp<PM> r;
  int64_t br = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    br += (int64_t)PM::P[i] - (int64_t)a.v[i];
    r.v[i] = (uint32_t)br;
    br >>= 32;
  }
  return r;
}

template <class PM> G16_HD Fp<PM> fp_dbl(const Fp<PM>& a) { return fp_add(a, a); }

// Montgomery product a*b*R^-1 mod p, CIOS.  p < 2^254 so the running value stays < 2p and the
// ninth word never needs a second carry word.
template <class PM> G16_HD Fp<PM> fp_mul(const Fp<PM>& a, const Fp<PM>& b) {
#if !defined(__HIP_DEVICE_COMPILE__) && defined(__SIZEOF_INT128__) && !defined(G16_HOST_MUL32)
  // Host build (proof tail, window folding, setup tool): same CIOS on 4 x 64-bit limbs, ~3.5x
  // faster than the 32-bit form on x86.  The 32-bit form below is what the device runs.
  typedef unsigned __int128 u128;
  uint64_t A[4], B[4], P[4], t[5] = {0, 0, 0, 0, 0};
  for (int i = 0; i < 4; i++) {
    A[i] = a.v[2 * i] | ((uint64_t)a.v[2 * i + 1] << 32);
    B[i] = b.v[2 * i] | ((uint64_t)b.v[2 * i + 1] << 32);
    P[i] = PM::P[2 * i] | ((uint64_t)PM::P[2 * i + 1] << 32);
  }
  // -p^-1 mod 2^64 from the 32-bit constant by one Newton step: x' = x (2 + p x)  (x = -p^-1)
  const uint64_t inv32 = PM::INV;
  const uint64_t inv64 = inv32 * (2 + P[0] * inv32);
  for (int i = 0; i < 4; i++) {
    u128 c = 0;
    for (int j = 0; j < 4; j++) { c += (u128)A[j] * B[i] + t[j]; t[j] = (uint64_t)c; c >>= 64; }
    const uint64_t t4 = t[4] + (uint64_t)c;
    const uint64_t m = t[0] * inv64;
    c = (u128)m * P[0] + t[0];
    c >>= 64;
    for (int j = 1; j < 4; j++) { c += (u128)m * P[j] + t[j]; t[j - 1] = (uint64_t)c; c >>= 64; }
    c += t4;
    t[3] = (uint64_t)c;
    t[4] = (uint64_t)(c >> 64);
  }
  Fp<PM> r;
  for (int i = 0; i < 4; i++) { r.v[2 * i] = (uint32_t)t[i]; r.v[2 * i + 1] = (uint32_t)(t[i] >> 32); }
  fp_reduce_once(r);
  return r;
#else
  uint32_t t[9];
#pragma unroll
  for (int i = 0; i < 9; i++) t[i] = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    uint64_t c = 0;
    const uint32_t bi = b.v[i];
#pragma unroll
    for (int j = 0; j < 8; j++) {
      c += (uint64_t)a.v[j] * bi + t[j];
      t[j] = (uint32_t)c;
      c >>= 32;
    }
    uint32_t t8 = t[8] + (uint32_t)c;  // < 2^32: value < 2p*2^32 bound
    const uint32_t m = t[0] * PM::INV;
    c = (uint64_t)m * PM::P[0] + t[0];
    c >>= 32;
#pragma unroll
    for (int j = 1; j < 8; j++) {
      c += (uint64_t)m * PM::P[j] + t[j];
      t[j - 1] = (uint32_t)c;
      c >>= 32;
    }
    c += t8;
    t[7] = (uint32_t)c;
    t[8] = (uint32_t)(c >> 32);
  }
  Fp<PM> r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.v[i] = t[i];
  fp_reduce_once(r);
  return r;
#endif
}

template <class PM> G16_HD Fp<PM> fp_sqr(const Fp<PM>& a) { return fp_mul(a, a); }

// standard integer -> Montgomery and back
template <class PM> G16_HD Fp<PM> fp_to_mont(const Fp<PM>& a) {
  Fp<PM> r2;
#pragma unroll
  for (int i = 0; i < 8; i++) r2.v[i] = PM::R2[i];
  return fp_mul(a, r2);
}
template <class PM> G16_HD Fp<PM> fp_from_mont(const Fp<PM>& a) {
  Fp<PM> one = fp_zero<PM>();
  one.v[0] = 1;
  return fp_mul(a, one);
}

// a^e for a 256-bit little-endian exponent (host tail / table builders; not a hot path)
template <class PM> G16_HD Fp<PM> fp_pow(const Fp<PM>& a, const uint32_t e[8]) {
  Fp<PM> r = fp_one<PM>();
  for (int i = 255; i >= 0; i--) {
    r = fp_sqr(r);
    if ((e[i >> 5] >> (i & 31)) & 1) r = fp_mul(r, a);
  }
  return r;
}
template <class PM> G16_HD Fp<PM> fp_pow_u64(const Fp<PM>& a, uint64_t e) {
  Fp<PM> r = fp_one<PM>();
  for (int i = 63; i >= 0; i--) {
    r = fp_sqr(r);
    if ((e >> i) & 1) r = fp_mul(r, a);
  }
  return r;
}
// Fermat inverse a^(p-2); inv(0) = 0
template <class PM> G16_HD Fp<PM> fp_inv(const Fp<PM>& a) {
  uint32_t e[8];
#pragma unroll
  for (int i = 0; i < 8; i++) e[i] = PM::P[i];
  e[0] -= 2;  // p is odd and its low limb is > 2 for both fields
  return fp_pow(a, e);
}

// ------------------------------------------------------------------ Fq2 = Fq[u]/(u^2+1)
struct alignas(16) Fq2 {
  Fq a, b;  // a + b*u
};

// Field "ops" bundles: the curve templates in ec.cuh are written against these.
struct FqOps {
  using T = Fq;
  static G16_HD T zero() { return fp_zero<FqParams>(); }
  static G16_HD T one() { return fp_one<FqParams>(); }
  static G16_HD bool is_zero(const T& x) { return fp_is_zero(x); }
  static G16_HD bool eq(const T& x, const T& y) { return fp_eq(x, y); }
  static G16_HD T add(const T& x, const T& y) { return fp_add(x, y); }
  static G16_HD T sub(const T& x, const T& y) { return fp_sub(x, y); }
  static G16_HD T neg(const T& x) { return fp_neg(x); }
  static G16_HD T mul(const T& x, const T& y) { return fp_mul(x, y); }
  static G16_HD T sqr(const T& x) { return fp_sqr(x); }
  static G16_HD T inv(const T& x) { return fp_inv(x); }
};

struct Fq2Ops {
  using T = Fq2;
  static G16_HD T zero() { return T{fp_zero<FqParams>(), fp_zero<FqParams>()}; }
  static G16_HD T one() { return T{fp_one<FqParams>(), fp_zero<FqParams>()}; }
  static G16_HD bool is_zero(const T& x) { return fp_is_zero(x.a) && fp_is_zero(x.b); }
  static G16_HD bool eq(const T& x, const T& y) { return fp_eq(x.a, y.a) && fp_eq(x.b, y.b); }
  static G16_HD T add(const T& x, const T& y) { return T{fp_add(x.a, y.a), fp_add(x.b, y.b)}; }
  static G16_HD T sub(const T& x, const T& y) { return T{fp_sub(x.a, y.a), fp_sub(x.b, y.b)}; }
  static G16_HD T neg(const T& x) { return T{fp_neg(x.a), fp_neg(x.b)}; }
  static G16_HD T mul(const T& x, const T& y) {  // Karatsuba, 3 base muls
    Fq v0 = fp_mul(x.a, y.a), v1 = fp_mul(x.b, y.b);
    Fq s = fp_mul(fp_add(x.a, x.b), fp_add(y.a, y.b));
    return T{fp_sub(v0, v1), fp_sub(fp_sub(s, v0), v1)};
  }
  static G16_HD T sqr(const T& x) {  // (a+b)(a-b), 2ab
    Fq m = fp_mul(x.a, x.b);
    return T{fp_mul(fp_add(x.a, x.b), fp_sub(x.a, x.b)), fp_add(m, m)};
  }
  static G16_HD T inv(const T& x) {
    Fq d = fp_inv(fp_add(fp_sqr(x.a), fp_sqr(x.b)));
    return T{fp_mul(x.a, d), fp_neg(fp_mul(x.b, d))};
  }
};

}  // namespace g16
