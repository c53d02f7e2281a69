// BN254 G1 (over Fq) / G2 (over Fq2) point arithmetic, y^2 = x^3 + b, a = 0.
//
// Replaces wasmcurves 0.1.0 build_curve_jacobian_a0.js / build_f2m.js
// (pin /root/reference/yarn.lock:1132-1138).  wasmcurves keeps Jacobian accumulators; here the
// accumulator is extended-Jacobian "XYZZ" (x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2): a mixed add costs
// 8M+2S instead of 7M+4S and needs no squaring of Z, which matters when every M is ~136
// v_mad_u64_u32.  Group elements -- and therefore the affine proof bytes -- are identical.
//
// Affine infinity is the all-zero byte image, as in snarkjs zkey sections.
#pragma once
#include "fp.cuh"

namespace g16 {

template <class F> struct alignas(16) Affine {
  typename F::T x, y;
};
template <class F> struct alignas(16) XYZZ {
  typename F::T x, y, zz, zzz;
};
using G1Affine = Affine<FqOps>;
using G2Affine = Affine<Fq2Ops>;
using G1XYZZ = XYZZ<FqOps>;
using G2XYZZ = XYZZ<Fq2Ops>;

template <class F> G16_HD bool aff_is_inf(const Affine<F>& p) {
  return F::is_zero(p.x) && F::is_zero(p.y);
}
template <class F> G16_HD bool xyzz_is_inf(const XYZZ<F>& p) { return F::is_zero(p.zz); }
template <class F> G16_HD void xyzz_set_inf(XYZZ<F>& p) {
  p.x = F::zero(); p.y = F::zero(); p.zz = F::zero(); p.zzz = F::zero();
}
template <class F> G16_HD void xyzz_from_affine(XYZZ<F>& r, const Affine<F>& a) {
  if (aff_is_inf(a)) { xyzz_set_inf(r); return; }
  r.x = a.x; r.y = a.y; r.zz = F::one(); r.zzz = F::one();
}

// 2*P for affine P (mdbl-2008-s-1)
template <class F> G16_HD void xyzz_dbl_affine(XYZZ<F>& r, const Affine<F>& p) {
  typename F::T U = F::add(p.y, p.y);
  typename F::T V = F::sqr(U);
  typename F::T W = F::mul(U, V);
  typename F::T S = F::mul(p.x, V);
  typename F::T X2 = F::sqr(p.x);
  typename F::T M = F::add(F::add(X2, X2), X2);
  r.x = F::sub(F::sqr(M), F::add(S, S));
  r.y = F::sub(F::mul(M, F::sub(S, r.x)), F::mul(W, p.y));
  r.zz = V;
  r.zzz = W;
}

// P <- 2*P (dbl-2008-s-1); infinity stays infinity because ZZ3 = V*ZZ1
template <class F> G16_HD void xyzz_dbl(XYZZ<F>& p) {
  typename F::T U = F::add(p.y, p.y);
  typename F::T V = F::sqr(U);
  typename F::T W = F::mul(U, V);
  typename F::T S = F::mul(p.x, V);
  typename F::T X2 = F::sqr(p.x);
  typename F::T M = F::add(F::add(X2, X2), X2);
  typename F::T X3 = F::sub(F::sqr(M), F::add(S, S));
  p.y = F::sub(F::mul(M, F::sub(S, X3)), F::mul(W, p.y));
  p.x = X3;
  p.zz = F::mul(V, p.zz);
  p.zzz = F::mul(W, p.zzz);
}

// acc <- acc + q, q affine and NOT infinity (madd-2008-s), complete in the exceptional cases
template <class F> G16_HD void xyzz_madd(XYZZ<F>& acc, const Affine<F>& q) {
  if (xyzz_is_inf(acc)) {
    acc.x = q.x; acc.y = q.y; acc.zz = F::one(); acc.zzz = F::one();
    return;
  }
  typename F::T U2 = F::mul(q.x, acc.zz);
  typename F::T S2 = F::mul(q.y, acc.zzz);
  typename F::T P = F::sub(U2, acc.x);
  typename F::T R = F::sub(S2, acc.y);
  if (F::is_zero(P)) {
    if (F::is_zero(R)) xyzz_dbl_affine(acc, q);
    else xyzz_set_inf(acc);
    return;
  }
  typename F::T PP = F::sqr(P);
  typename F::T PPP = F::mul(P, PP);
  typename F::T Qv = F::mul(acc.x, PP);
  typename F::T X3 = F::sub(F::sub(F::sqr(R), PPP), F::add(Qv, Qv));
  acc.y = F::sub(F::mul(R, F::sub(Qv, X3)), F::mul(acc.y, PPP));
  acc.x = X3;
  acc.zz = F::mul(acc.zz, PP);
  acc.zzz = F::mul(acc.zzz, PPP);
}

// acc <- acc + q, both XYZZ (add-2008-s), complete
template <class F> G16_HD void xyzz_add(XYZZ<F>& acc, const XYZZ<F>& q) {
  if (xyzz_is_inf(q)) return;
  if (xyzz_is_inf(acc)) { acc = q; return; }
  typename F::T U1 = F::mul(acc.x, q.zz);
  typename F::T U2 = F::mul(q.x, acc.zz);
  typename F::T S1 = F::mul(acc.y, q.zzz);
  typename F::T S2 = F::mul(q.y, acc.zzz);
  typename F::T P = F::sub(U2, U1);
  typename F::T R = F::sub(S2, S1);
  if (F::is_zero(P)) {
    if (F::is_zero(R)) xyzz_dbl(acc);
    else xyzz_set_inf(acc);
    return;
  }
  typename F::T PP = F::sqr(P);
  typename F::T PPP = F::mul(P, PP);
  typename F::T Qv = F::mul(U1, PP);
  typename F::T X3 = F::sub(F::sub(F::sqr(R), PPP), F::add(Qv, Qv));
  acc.y = F::sub(F::mul(R, F::sub(Qv, X3)), F::mul(S1, PPP));
  acc.x = X3;
  acc.zz = F::mul(F::mul(acc.zz, q.zz), PP);
  acc.zzz = F::mul(F::mul(acc.zzz, q.zzz), PPP);
}

template <class F> G16_HD void aff_neg(Affine<F>& p) { p.y = F::neg(p.y); }
template <class F> G16_HD void xyzz_neg(XYZZ<F>& p) { p.y = F::neg(p.y); }

// XYZZ -> affine (one field inversion; host tail only)
template <class F> G16_HD void xyzz_to_affine(Affine<F>& r, const XYZZ<F>& p) {
  if (xyzz_is_inf(p)) { r.x = F::zero(); r.y = F::zero(); return; }
  // ZZ^3 = ZZZ^2  =>  1/ZZ = (ZZ/ZZZ)^2
  typename F::T zi = F::inv(p.zzz);
  typename F::T zzi = F::sqr(F::mul(zi, p.zz));
  r.x = F::mul(p.x, zzi);
  r.y = F::mul(p.y, zi);
}

// k * P for a 256-bit little-endian standard-form scalar (host tail: blinding terms)
template <class F> G16_HD void xyzz_mul_scalar(XYZZ<F>& r, const Affine<F>& p, const uint32_t k[8]) {
  xyzz_set_inf(r);
  if (aff_is_inf(p)) return;
  for (int i = 255; i >= 0; i--) {
    xyzz_dbl(r);
    if ((k[i >> 5] >> (i & 31)) & 1) xyzz_madd(r, p);
  }
}

}  // namespace g16
