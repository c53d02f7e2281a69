// XYZZ point arithmetic over the lazy 9 x 29-bit field (fq29.cuh) -- what the MSM kernels run.
// Same formulas as ec.cuh (madd-2008-s, add-2008-s, dbl-2008-s-1, mdbl-2008-s-1); the only
// difference is that every subtraction names the multiple of p that covers its subtrahend.
// The K constants and the resulting coordinate bounds (X < 5.4p, Y < 3.5p, ZZ/ZZZ < 1.1p for both
// G1 and G2; every product input < 16p) come from tools/f29_bounds.py and are asserted at run time
// by the host differential test (tests/native/f29_test.cpp, built with -DG16_F29_CHECK).
#pragma once
#include "ec.cuh"
#include "fq29.cuh"

namespace g16 {

using G1Affine29 = Affine<Fq29Ops>;
using G2Affine29 = Affine<Fq2x29Ops>;
using G1XYZZ29 = XYZZ<Fq29Ops>;
using G2XYZZ29 = XYZZ<Fq2x29Ops>;

template <class F> G16_HD bool x29_is_inf(const XYZZ<F>& p) { return F::is_literal_zero(p.zz); }
template <class F> G16_HD void x29_set_inf(XYZZ<F>& p) {
  p.x = F::zero(); p.y = F::zero(); p.zz = F::zero(); p.zzz = F::zero();
}
// x == 0 (mod p) for a difference below 8p: cheap low-limb filter, exact test only on a hit
template <class F> G16_HD bool x29_diff_is_zero(const typename F::T& d) {
  if (!F::template maybe_zero<7>(d)) return false;
  return F::is_zero(d);
}

// 2*P for affine P (coordinates below 2p)
template <class F> G16_HD void x29_dbl_affine(XYZZ<F>& r, const Affine<F>& p) {
  using T = typename F::T;
  const T U = F::add(p.y, p.y);
  const T V = F::sqr(U);
  const T W = F::mul(U, V);
  const T S = F::mul(p.x, V);
  const T X2 = F::sqr(p.x);
  const T M = F::add(F::add(X2, X2), X2);
  r.x = F::template sub<3>(F::sqr(M), F::add(S, S));
  r.y = F::template sub<2>(F::mul(M, F::template sub<5>(S, r.x)), F::mul(W, p.y));
  r.zz = V;
  r.zzz = W;
}

template <class F> G16_HD void x29_dbl(XYZZ<F>& p) {
  using T = typename F::T;
  if (x29_is_inf(p)) return;
  const T U = F::add(p.y, p.y);
  const T V = F::sqr(U);
  const T W = F::mul(U, V);
  const T S = F::mul(p.x, V);
  const T X2 = F::sqr(p.x);
  const T M = F::add(F::add(X2, X2), X2);
  const T X3 = F::template sub<3>(F::sqr(M), F::add(S, S));
  p.y = F::template sub<2>(F::mul(M, F::template sub<5>(S, X3)), F::mul(W, p.y));
  p.x = X3;
  p.zz = F::mul(V, p.zz);
  p.zzz = F::mul(W, p.zzz);
}

// acc <- acc + q, q affine (coordinates below 2p), not infinity
template <class F> G16_HD void x29_madd(XYZZ<F>& acc, const Affine<F>& q) {
  using T = typename F::T;
  if (x29_is_inf(acc)) {
    acc.x = q.x; acc.y = q.y; acc.zz = F::one(); acc.zzz = F::one();
    return;
  }
  const T U2 = F::mul(q.x, acc.zz);
  const T S2 = F::mul(q.y, acc.zzz);
  const T P = F::template sub<6>(U2, acc.x);
  const T R = F::template sub<4>(S2, acc.y);
  if (x29_diff_is_zero<F>(P)) {
    if (F::is_zero(R)) x29_dbl_affine(acc, q);
    else x29_set_inf(acc);
    return;
  }
  const T PP = F::sqr(P);
  const T PPP = F::mul(P, PP);
  const T Qv = F::mul(acc.x, PP);
  const T X3 = F::sub_b_2c(F::sqr(R), PPP, Qv);
  acc.y = F::template sub<2>(F::mul(R, F::template sub<6>(Qv, X3)), F::mul(acc.y, PPP));
  acc.x = X3;
  acc.zz = F::mul(acc.zz, PP);
  acc.zzz = F::mul(acc.zzz, PPP);
}

// acc <- acc + q by the plain madd-2008-s formula, no exceptional cases: the hot loop of the bucket
// accumulation.  Returns true when the result may be wrong -- x(q) == x(acc) (doubling or cancellation,
// caught by the low-limb filter on P, which also fires on a 2^-26 fraction of ordinary additions) -- and
// the caller then recomputes the whole task with x29_madd (msm_redo_kernel).  acc must not be infinity
// (a task starts from its first point; infinity only arises from a cancellation, which is flagged).
template <class F> G16_HD bool x29_madd_fast(XYZZ<F>& acc, const Affine<F>& q) {
  using T = typename F::T;
  const T U2 = F::mul(q.x, acc.zz);
  const T S2 = F::mul(q.y, acc.zzz);
  const T P = F::template sub<6>(U2, acc.x);
  const T R = F::template sub<4>(S2, acc.y);
  const bool suspicious = F::template maybe_zero<7>(P);
  const T PP = F::sqr(P);
  const T PPP = F::mul(P, PP);
  const T Qv = F::mul(acc.x, PP);
  const T X3 = F::sub_b_2c(F::sqr(R), PPP, Qv);   // R^2 + 4p - PPP - 2Q, one carry ripple instead of three
  if constexpr (F::kFusedY3) {
    // Y1 <= 2p here (an affine coordinate, or a previous Y3 < 1.3p), so 2p - Y1 is its negative; both products
    // share one reduction: (5.1p * 7.1p + 2p * 2.5p) / 2^261 + p < 1.3p
    acc.y = F::mul_add(R, F::template sub<6>(Qv, X3), F::template neg<2>(acc.y), PPP);
  } else {
    acc.y = F::template sub<2>(F::mul(R, F::template sub<6>(Qv, X3)), F::mul(acc.y, PPP));
  }
  acc.x = X3;
  acc.zz = F::mul(acc.zz, PP);
  acc.zzz = F::mul(acc.zzz, PPP);
  return suspicious;
}

// acc <- acc + q, both XYZZ
template <class F> G16_HD void x29_add(XYZZ<F>& acc, const XYZZ<F>& q) {
  using T = typename F::T;
  if (x29_is_inf(q)) return;
  if (x29_is_inf(acc)) { acc = q; return; }
  const T U1 = F::mul(acc.x, q.zz);
  const T U2 = F::mul(q.x, acc.zz);
  const T S1 = F::mul(acc.y, q.zzz);
  const T S2 = F::mul(q.y, acc.zzz);
  const T P = F::template sub<2>(U2, U1);
  const T R = F::template sub<2>(S2, S1);
  if (x29_diff_is_zero<F>(P)) {
    if (F::is_zero(R)) x29_dbl(acc);
    else x29_set_inf(acc);
    return;
  }
  const T PP = F::sqr(P);
  const T PPP = F::mul(P, PP);
  const T Qv = F::mul(U1, PP);
  const T X3 = F::sub_b_2c(F::sqr(R), PPP, Qv);
  acc.y = F::template sub<2>(F::mul(R, F::template sub<6>(Qv, X3)), F::mul(S1, PPP));
  acc.x = X3;
  acc.zz = F::mul(F::mul(acc.zz, q.zz), PP);
  acc.zzz = F::mul(F::mul(acc.zzz, q.zzz), PPP);
}

// -(x, y) for an affine point with y below 2p
template <class F> G16_HD void a29_neg(Affine<F>& p) { p.y = F::template neg<2>(p.y); }

// Resident storage of the base points: each lazy coordinate is below 2p < 2^255, so its nine 29-bit limbs
// repack losslessly into eight 32-bit words -- 64 bytes per G1 point (one 64-byte sector, never straddling a
// 128-byte line) and 128 per G2 point instead of 80 / 160: the accumulate kernel's gather traffic halves, for
// ~40 shift/mask instructions per gathered point.
struct alignas(16) F29Packed { uint32_t v[8]; };
G16_HD F29Packed f29_pack(const F29& a) {
  F29Packed r;
#pragma unroll
  for (int w = 0; w < 8; w++) {
    // word w = bits [32 w, 32 w + 32): spans limbs i = (32 w) / 29 and i + 1 (and i + 2 when 32 w % 29 > 26)
    const int bit = 32 * w, i = bit / 29, o = bit % 29;
    uint64_t x = a.l[i] >> o;
    int have = 29 - o;
    if (i + 1 < 9) { x |= (uint64_t)a.l[i + 1] << have; have += 29; }
    if (have < 32 && i + 2 < 9) x |= (uint64_t)a.l[i + 2] << have;
    r.v[w] = (uint32_t)x;
  }
  return r;
}
G16_HD F29 f29_unpack(const F29Packed& p) {
  F29 t;
#pragma unroll
  for (int i = 0; i < 9; i++) {
    const int bit = 29 * i, w = bit >> 5, o = bit & 31;
    uint64_t x = p.v[w];
    if (w + 1 < 8) x |= (uint64_t)p.v[w + 1] << 32;
    t.l[i] = (uint32_t)(x >> o) & (i < 8 ? kM29 : 0xffffffffu);
  }
  t.pad_ = 0;
  return t;
}
template <class F> struct PackedAffine;
template <> struct alignas(64) PackedAffine<Fq29Ops> { F29Packed x, y; };
template <> struct alignas(64) PackedAffine<Fq2x29Ops> { F29Packed xa, xb, ya, yb; };
G16_HD void a29_pack(PackedAffine<Fq29Ops>& r, const Affine<Fq29Ops>& p) { r.x = f29_pack(p.x); r.y = f29_pack(p.y); }
G16_HD void a29_unpack(Affine<Fq29Ops>& r, const PackedAffine<Fq29Ops>& p) { r.x = f29_unpack(p.x); r.y = f29_unpack(p.y); }
G16_HD void a29_pack(PackedAffine<Fq2x29Ops>& r, const Affine<Fq2x29Ops>& p) {
  r.xa = f29_pack(p.x.a); r.xb = f29_pack(p.x.b); r.ya = f29_pack(p.y.a); r.yb = f29_pack(p.y.b);
}
G16_HD void a29_unpack(Affine<Fq2x29Ops>& r, const PackedAffine<Fq2x29Ops>& p) {
  r.x.a = f29_unpack(p.xa); r.x.b = f29_unpack(p.xb); r.y.a = f29_unpack(p.ya); r.y.b = f29_unpack(p.yb);
}

// canonical (fp.cuh) <-> lazy conversions
template <class F, class FC> G16_HD void a29_from_canon(Affine<F>& r, const Affine<FC>& a) {
  r.x = F::from_canon(a.x);
  r.y = F::from_canon(a.y);
}
template <class F, class FC> G16_HD void x29_to_canon(XYZZ<FC>& r, const XYZZ<F>& p) {
  if (x29_is_inf(p)) { xyzz_set_inf(r); return; }
  r.x = F::to_canon(p.x); r.y = F::to_canon(p.y); r.zz = F::to_canon(p.zz); r.zzz = F::to_canon(p.zzz);
}

}  // namespace g16
