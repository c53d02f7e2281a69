// BN254 optimal-ate pairing for the Groth16 verifier ("GPU Groth16 batch verifier (pairing)";
// the acceptance check of SURVEY 3.4: e(-A,B) e(alpha,beta) e(vk_x,gamma) e(C,delta) = 1).
//
// Replaces wasmcurves 0.1.0 build_pairing.js / build_ftm.js as `snarkjs groth16 verify` drives them through
// curve.pairingEq ([EXT], pins /root/reference/yarn.lock:987-1001, 1132-1138; the reference's own call site is its
// README's verify step, there is no source in /root/reference).  Same group elements and the same accept/reject
// decision; the schedule is this repo's own:
//   tower   Fq2 = Fq[u]/(u^2+1),  Fq6 = Fq2[V]/(V^3 - xi),  Fq12 = Fq6[W]/(W^2 - V),  xi = 9 + u
//   twist   E'/Fq2: y^2 = x^3 + 3/xi (D-type); a line through points of E' evaluated at P = (xP, yP) in E(Fq) is the
//           sparse element  l0 + (lVV xP) W^4 + (lVW yP) W^3
//   Miller  f_{6z+2,Q}(P) over the bits of 6z+2 (z = 4965661367192848881), homogeneous projective doubling /
//           mixed-addition steps on E', then the two Frobenius steps Q1 = pi(Q), Q2 = -pi^2(Q)
//   final   easy part (p^6-1)(p^2+1), then the hard part raised to m (p^4-p^2+1)/r with the fixed multiplier
//           m = 2z(6z^2+3z+1) (coprime to r): three exponentiations by z, three Frobenius maps and a dozen products
//           instead of a 760-bit exponentiation.  The value is therefore e(P,Q)^m -- still a non-degenerate bilinear
//           pairing, so every product-equals-one decision is the decision of the plain pairing (tests compare the
//           value itself with oracle/bn254.py's pairing raised to m).
// All of it is __host__ __device__ on the canonical 8x32 Montgomery Fq (fp.cuh): the device build is the verifier's
// hot path (verify.hip), the host build computes the per-key constants (line coefficients of gamma2 / delta2,
// e(alpha,beta)) at g16_verifier_create and serves tests/native/pairing_test.cpp.
#pragma once
#include "ec.cuh"

namespace g16 {

struct Fq6 { Fq2 c0, c1, c2; };     // c0 + c1 V + c2 V^2
struct Fq12 { Fq6 c0, c1; };        // c0 + c1 W

// line coefficients of one Miller step
struct EllCoeffs { Fq2 l0, lVW, lVV; };

constexpr int kAteBits = 65;                                  // 6z+2 = 0x1_9D797039BE763BA8
constexpr uint64_t kAteLo = 0x9D797039BE763BA8ull;            // low 64 bits; bit 64 is the leading one
constexpr uint64_t kBnZ = 4965661367192848881ull;             // the curve parameter z (positive)
G16_HD constexpr int ate_popcount_low() {
  int c = 0;
  for (int i = 0; i < 64; i++) c += (int)((kAteLo >> i) & 1);
  return c;
}
constexpr int kEllSteps = 64 + ate_popcount_low() + 2;        // doublings + additions + the two Frobenius steps

// per-curve constants, computed once on the host (pairing_consts_init) and handed to the kernels by value
struct PairingConsts {
  Fq2 frob1[6];     // xi^(k (p-1)/6): W^k -> its image under pi, up to conjugation of the coefficient
  Fq2 frob2[6];     // xi^(k (p^2-1)/6)  (these lie in Fq)
  Fq2 frob3[6];     // xi^(k (p^3-1)/6)
  Fq2 twist_b;      // 3 / xi
  Fq two_inv;       // 1/2
};

// ------------------------------------------------------------------ Fq2 helpers
using F2 = Fq2Ops;
// The Fq2 product is the unit of code here: ~1.6 k instructions, CALLED from the tower arithmetic and the curve steps
// (inlined everywhere, the verifier's translation unit was ~2 M instructions and took a quarter of an hour to
// compile; the call costs a few dozen cycles per 1.6 k-instruction body).  The Fq12 operations are calls as well.
#if defined(__HIPCC__)
#define G16_F12_FN static __host__ __device__ __noinline__   // (internal linkage: two translation units include this header)
#else
#define G16_F12_FN inline
#endif
G16_F12_FN Fq2 f2m(const Fq2& a, const Fq2& b) { return Fq2Ops::mul(a, b); }
G16_F12_FN Fq2 f2s(const Fq2& a) { return Fq2Ops::sqr(a); }
G16_HD Fq2 f2_conj(const Fq2& a) { return Fq2{a.a, fp_neg(a.b)}; }
G16_HD Fq2 f2_dbl(const Fq2& a) { return F2::add(a, a); }
G16_HD Fq2 f2_mul_xi(const Fq2& a) {   // (a + b u)(9 + u) = (9a - b) + (9b + a) u
  const Fq a2 = fp_add(a.a, a.a), a4 = fp_add(a2, a2), a8 = fp_add(a4, a4), a9 = fp_add(a8, a.a);
  const Fq b2 = fp_add(a.b, a.b), b4 = fp_add(b2, b2), b8 = fp_add(b4, b4), b9 = fp_add(b8, a.b);
  return Fq2{fp_sub(a9, a.b), fp_add(b9, a.a)};
}
G16_HD Fq2 f2_mul_fq(const Fq2& a, const Fq& k) { return Fq2{fp_mul(a.a, k), fp_mul(a.b, k)}; }

// ------------------------------------------------------------------ Fq6
G16_HD Fq6 f6_zero() { return Fq6{F2::zero(), F2::zero(), F2::zero()}; }
G16_HD Fq6 f6_one() { return Fq6{F2::one(), F2::zero(), F2::zero()}; }
G16_HD Fq6 f6_add(const Fq6& a, const Fq6& b) { return Fq6{F2::add(a.c0, b.c0), F2::add(a.c1, b.c1), F2::add(a.c2, b.c2)}; }
G16_HD Fq6 f6_sub(const Fq6& a, const Fq6& b) { return Fq6{F2::sub(a.c0, b.c0), F2::sub(a.c1, b.c1), F2::sub(a.c2, b.c2)}; }
G16_HD Fq6 f6_neg(const Fq6& a) { return Fq6{F2::neg(a.c0), F2::neg(a.c1), F2::neg(a.c2)}; }
G16_HD Fq6 f6_mul_v(const Fq6& a) { return Fq6{f2_mul_xi(a.c2), a.c0, a.c1}; }   // times V
G16_HD bool f6_is_zero(const Fq6& a) { return F2::is_zero(a.c0) && F2::is_zero(a.c1) && F2::is_zero(a.c2); }
G16_HD bool f6_eq(const Fq6& a, const Fq6& b) { return F2::eq(a.c0, b.c0) && F2::eq(a.c1, b.c1) && F2::eq(a.c2, b.c2); }
// Karatsuba over Fq2: 6 products
G16_HD Fq6 f6_mul(const Fq6& a, const Fq6& b) {
  const Fq2 v0 = f2m(a.c0, b.c0), v1 = f2m(a.c1, b.c1), v2 = f2m(a.c2, b.c2);
  const Fq2 t12 = F2::sub(F2::sub(f2m(F2::add(a.c1, a.c2), F2::add(b.c1, b.c2)), v1), v2);   // a1 b2 + a2 b1
  const Fq2 t01 = F2::sub(F2::sub(f2m(F2::add(a.c0, a.c1), F2::add(b.c0, b.c1)), v0), v1);   // a0 b1 + a1 b0
  const Fq2 t02 = F2::sub(F2::sub(f2m(F2::add(a.c0, a.c2), F2::add(b.c0, b.c2)), v0), v2);   // a0 b2 + a2 b0
  return Fq6{F2::add(v0, f2_mul_xi(t12)), F2::add(t01, f2_mul_xi(v2)), F2::add(t02, v1)};
}
G16_HD Fq6 f6_sqr(const Fq6& a) { return f6_mul(a, a); }
G16_HD Fq6 f6_inv(const Fq6& a) {
  // c0 = a0^2 - xi a1 a2, c1 = xi a2^2 - a0 a1, c2 = a1^2 - a0 a2;  1/a = (c0, c1, c2) / (a0 c0 + xi (a2 c1 + a1 c2))
  const Fq2 c0 = F2::sub(f2s(a.c0), f2_mul_xi(f2m(a.c1, a.c2)));
  const Fq2 c1 = F2::sub(f2_mul_xi(f2s(a.c2)), f2m(a.c0, a.c1));
  const Fq2 c2 = F2::sub(f2s(a.c1), f2m(a.c0, a.c2));
  const Fq2 t = F2::add(f2m(a.c0, c0), f2_mul_xi(F2::add(f2m(a.c2, c1), f2m(a.c1, c2))));
  const Fq2 ti = F2::inv(t);
  return Fq6{f2m(c0, ti), f2m(c1, ti), f2m(c2, ti)};
}

// ------------------------------------------------------------------ Fq12
G16_HD Fq12 f12_one() { return Fq12{f6_one(), f6_zero()}; }
G16_HD bool f12_eq(const Fq12& a, const Fq12& b) { return f6_eq(a.c0, b.c0) && f6_eq(a.c1, b.c1); }
G16_HD bool f12_is_one(const Fq12& a) { return f12_eq(a, f12_one()); }
G16_HD Fq12 f12_conj(const Fq12& a) { return Fq12{a.c0, f6_neg(a.c1)}; }   // = a^(p^6); the inverse on the cyclotomic subgroup
G16_F12_FN Fq12 f12_mul(const Fq12& a, const Fq12& b) {
  const Fq6 v0 = f6_mul(a.c0, b.c0), v1 = f6_mul(a.c1, b.c1);
  const Fq6 s = f6_mul(f6_add(a.c0, a.c1), f6_add(b.c0, b.c1));
  return Fq12{f6_add(v0, f6_mul_v(v1)), f6_sub(f6_sub(s, v0), v1)};
}
G16_F12_FN Fq12 f12_sqr(const Fq12& a) {   // complex squaring: 2 Fq6 products
  const Fq6 ab = f6_mul(a.c0, a.c1);
  const Fq6 t = f6_mul(f6_add(a.c0, a.c1), f6_add(a.c0, f6_mul_v(a.c1)));
  return Fq12{f6_sub(f6_sub(t, ab), f6_mul_v(ab)), f6_add(ab, ab)};
}
// a times the sparse line value  l0 + l4 W^4 + l3 W^3  =  (l0, 0, l4) + (0, l3, 0) W
G16_F12_FN Fq12 f12_mul_line(const Fq12& a, const Fq2& l0, const Fq2& l3, const Fq2& l4) {
  // (a0 + a1 W)(b0 + b1 W), b0 = (l0, 0, l4), b1 = (0, l3, 0)
  const Fq6& x = a.c0;
  const Fq6& y = a.c1;
  // x * b0
  const Fq2 x0l0 = f2m(x.c0, l0), x1l0 = f2m(x.c1, l0), x2l0 = f2m(x.c2, l0);
  const Fq2 x0l4 = f2m(x.c0, l4), x1l4 = f2m(x.c1, l4), x2l4 = f2m(x.c2, l4);
  // (x0 + x1 V + x2 V^2)(l0 + l4 V^2) = x0 l0 + xi x1 l4 + (x1 l0 + xi x2 l4) V + (x2 l0 + x0 l4) V^2
  const Fq6 v0{F2::add(x0l0, f2_mul_xi(x1l4)), F2::add(x1l0, f2_mul_xi(x2l4)), F2::add(x2l0, x0l4)};
  // y * b1 = (y0 + y1 V + y2 V^2) l3 V = xi y2 l3 + y0 l3 V + y1 l3 V^2
  const Fq2 y0l3 = f2m(y.c0, l3), y1l3 = f2m(y.c1, l3), y2l3 = f2m(y.c2, l3);
  const Fq6 v1{f2_mul_xi(y2l3), y0l3, y1l3};
  // x * b1 and y * b0 for the W coefficient
  const Fq2 x0l3 = f2m(x.c0, l3), x1l3 = f2m(x.c1, l3), x2l3 = f2m(x.c2, l3);
  const Fq6 xb1{f2_mul_xi(x2l3), x0l3, x1l3};
  const Fq2 y0l0 = f2m(y.c0, l0), y1l0 = f2m(y.c1, l0), y2l0 = f2m(y.c2, l0);
  const Fq2 y0l4 = f2m(y.c0, l4), y1l4 = f2m(y.c1, l4), y2l4 = f2m(y.c2, l4);
  const Fq6 yb0{F2::add(y0l0, f2_mul_xi(y1l4)), F2::add(y1l0, f2_mul_xi(y2l4)), F2::add(y2l0, y0l4)};
  return Fq12{f6_add(v0, f6_mul_v(v1)), f6_add(xb1, yb0)};
}
G16_F12_FN Fq12 f12_inv(const Fq12& a) {
  const Fq6 t = f6_inv(f6_sub(f6_sqr(a.c0), f6_mul_v(f6_sqr(a.c1))));
  return Fq12{f6_mul(a.c0, t), f6_neg(f6_mul(a.c1, t))};
}
// a^(p^k), k = 1, 2, 3: coefficient of W^i is (conjugated for odd k and) scaled by frob_k[i]
G16_F12_FN Fq12 f12_frob(const Fq12& a, int k, const PairingConsts& pc) {
  const Fq2* g = k == 1 ? pc.frob1 : (k == 2 ? pc.frob2 : pc.frob3);
  const bool cj = (k & 1) != 0;
  auto m = [&](const Fq2& x, int i) { return f2m(cj ? f2_conj(x) : x, g[i]); };
  // W^0, W^2, W^4 live in c0 = (., V, V^2); W^1, W^3, W^5 in c1
  return Fq12{Fq6{m(a.c0.c0, 0), m(a.c0.c1, 2), m(a.c0.c2, 4)}, Fq6{m(a.c1.c0, 1), m(a.c1.c1, 3), m(a.c1.c2, 5)}};
}
// a^2 for a in the cyclotomic subgroup (a^(p^6+1) = 1: everything after the easy part of the final exponentiation):
// Granger-Scott squaring -- three squarings in Fq4 = Fq2[y]/(y^2 - xi) over the coefficient pairs (g0, h1), (h0, g2),
// (g1, h2) of a = (g0, g1, g2) + (h0, h1, h2) W: 6 Fq2 products instead of the 12 of the plain squaring.
G16_F12_FN Fq12 f12_cyclo_sqr(const Fq12& a) {
  const Fq2 &z0 = a.c0.c0, &z4 = a.c0.c1, &z3 = a.c0.c2, &z2 = a.c1.c0, &z1 = a.c1.c1, &z5 = a.c1.c2;
  auto fq4_sqr = [](const Fq2& x, const Fq2& y, Fq2& t0, Fq2& t1) {   // (x + y Y)^2, Y^2 = xi
    const Fq2 tmp = f2m(x, y);
    t0 = F2::sub(F2::sub(f2m(F2::add(x, y), F2::add(x, f2_mul_xi(y))), tmp), f2_mul_xi(tmp));
    t1 = F2::add(tmp, tmp);
  };
  Fq2 t0, t1, t2, t3, t4, t5;
  fq4_sqr(z0, z1, t0, t1);
  fq4_sqr(z2, z3, t2, t3);
  fq4_sqr(z4, z5, t4, t5);
  auto m3p2 = [](const Fq2& t, const Fq2& z) { const Fq2 s = F2::add(t, z); return F2::add(F2::add(s, s), t); };   // 3t + 2z
  auto m3m2 = [](const Fq2& t, const Fq2& z) { const Fq2 s = F2::sub(t, z); return F2::add(F2::add(s, s), t); };   // 3t - 2z
  const Fq2 xt5 = f2_mul_xi(t5);
  const Fq2 n0 = m3m2(t0, z0), n1 = m3p2(t1, z1), n2 = m3p2(xt5, z2), n3 = m3m2(t4, z3), n4 = m3m2(t2, z4), n5 = m3p2(t3, z5);
  return Fq12{Fq6{n0, n4, n3}, Fq6{n2, n1, n5}};
}
// a^z (z = kBnZ) for a in the cyclotomic subgroup
G16_F12_FN Fq12 f12_pow_z(const Fq12& a) {
  Fq12 r = a;
  for (int i = 61; i >= 0; i--) {   // z has 63 bits: bit 62 is the leading one
    r = f12_cyclo_sqr(r);
    if ((kBnZ >> i) & 1) r = f12_mul(r, a);
  }
  return r;
}

// ------------------------------------------------------------------ G2 steps (projective R = (X, Y, Z) on the twist)
struct G2Proj { Fq2 x, y, z; };

G16_HD void g2_double_step(G2Proj& r, EllCoeffs& c, const PairingConsts& pc) {
  const Fq2 A = f2_mul_fq(f2m(r.x, r.y), pc.two_inv);
  const Fq2 B = f2s(r.y);
  const Fq2 C = f2s(r.z);
  const Fq2 D = F2::add(f2_dbl(C), C);
  const Fq2 E = f2m(pc.twist_b, D);
  const Fq2 F = F2::add(f2_dbl(E), E);
  const Fq2 G = f2_mul_fq(F2::add(B, F), pc.two_inv);
  const Fq2 H = F2::sub(f2s(F2::add(r.y, r.z)), F2::add(B, C));
  const Fq2 I = F2::sub(E, B);
  const Fq2 J = f2s(r.x);
  const Fq2 E2 = f2s(E);
  r.x = f2m(A, F2::sub(B, F));
  r.y = F2::sub(f2s(G), F2::add(f2_dbl(E2), E2));
  r.z = f2m(B, H);
  c.l0 = f2_mul_xi(I);
  c.lVW = F2::neg(H);
  c.lVV = F2::add(f2_dbl(J), J);
}

G16_HD void g2_add_step(G2Proj& r, const Affine<Fq2Ops>& q, EllCoeffs& c) {
  const Fq2 D = F2::sub(r.x, f2m(q.x, r.z));
  const Fq2 E = F2::sub(r.y, f2m(q.y, r.z));
  const Fq2 F = f2s(D);
  const Fq2 G = f2s(E);
  const Fq2 H = f2m(D, F);
  const Fq2 I = f2m(r.x, F);
  const Fq2 J = F2::sub(F2::add(H, f2m(r.z, G)), f2_dbl(I));
  const Fq2 y1 = r.y;
  r.x = f2m(D, J);
  r.y = F2::sub(f2m(E, F2::sub(I, J)), f2m(H, y1));
  r.z = f2m(r.z, H);
  c.l0 = f2_mul_xi(F2::sub(f2m(E, q.x), f2m(D, q.y)));
  c.lVV = F2::neg(E);
  c.lVW = D;
}

// pi(Q) on the twist: (conj(x) xi^((p-1)/3), conj(y) xi^((p-1)/2))
G16_HD Affine<Fq2Ops> g2_frob(const Affine<Fq2Ops>& q, const PairingConsts& pc) {
  Affine<Fq2Ops> r;
  r.x = f2m(f2_conj(q.x), pc.frob1[2]);
  r.y = f2m(f2_conj(q.y), pc.frob1[3]);
  return r;
}

// The kEllSteps line coefficients of Q (not infinity), in the order the Miller loop consumes them.
template <class Sink> G16_HD void g2_line_schedule(const Affine<Fq2Ops>& q, const PairingConsts& pc, Sink&& sink) {
  G2Proj r{q.x, q.y, F2::one()};
  EllCoeffs c;
  for (int i = 63; i >= 0; i--) {
    g2_double_step(r, c, pc);
    sink(c);
    if ((kAteLo >> i) & 1) {
      g2_add_step(r, q, c);
      sink(c);
    }
  }
  const Affine<Fq2Ops> q1 = g2_frob(q, pc);
  Affine<Fq2Ops> q2 = g2_frob(q1, pc);
  q2.y = F2::neg(q2.y);
  g2_add_step(r, q1, c);
  sink(c);
  g2_add_step(r, q2, c);
  sink(c);
}

G16_HD void g2_precompute(const Affine<Fq2Ops>& q, const PairingConsts& pc, EllCoeffs* out) {
  int k = 0;
  g2_line_schedule(q, pc, [&](const EllCoeffs& c) { out[k++] = c; });
}

// f <- f * line(P) for one step
G16_HD void miller_apply(Fq12& f, const EllCoeffs& c, const Affine<FqOps>& p) {
  f = f12_mul_line(f, c.l0, f2_mul_fq(c.lVW, p.y), f2_mul_fq(c.lVV, p.x));
}

// Miller loop of ONE pair with precomputed coefficients (kEllSteps of them); P not infinity
G16_HD Fq12 miller_loop_pre(const Affine<FqOps>& p, const EllCoeffs* __restrict__ coeffs) {
  Fq12 f = f12_one();
  int k = 0;
  for (int i = 63; i >= 0; i--) {
    f = f12_sqr(f);
    miller_apply(f, coeffs[k++], p);
    if ((kAteLo >> i) & 1) miller_apply(f, coeffs[k++], p);
  }
  miller_apply(f, coeffs[k++], p);
  miller_apply(f, coeffs[k++], p);
  return f;
}

// Miller loop of one pair, the coefficients of Q computed on the fly; P, Q not infinity
G16_HD Fq12 miller_loop(const Affine<FqOps>& p, const Affine<Fq2Ops>& q, const PairingConsts& pc) {
  Fq12 f = f12_one();
  G2Proj r{q.x, q.y, F2::one()};
  EllCoeffs c;
  for (int i = 63; i >= 0; i--) {
    f = f12_sqr(f);
    g2_double_step(r, c, pc);
    miller_apply(f, c, p);
    if ((kAteLo >> i) & 1) {
      g2_add_step(r, q, c);
      miller_apply(f, c, p);
    }
  }
  const Affine<Fq2Ops> q1 = g2_frob(q, pc);
  Affine<Fq2Ops> q2 = g2_frob(q1, pc);
  q2.y = F2::neg(q2.y);
  g2_add_step(r, q1, c);
  miller_apply(f, c, p);
  g2_add_step(r, q2, c);
  miller_apply(f, c, p);
  return f;
}

// f^(m (p^12 - 1) / r), m = 2z(6z^2+3z+1)  (see the header)
G16_HD Fq12 final_exponentiation(const Fq12& f, const PairingConsts& pc) {
  // easy part: f^((p^6 - 1)(p^2 + 1))
  const Fq12 t0 = f12_mul(f12_conj(f), f12_inv(f));
  const Fq12 e = f12_mul(f12_frob(t0, 2, pc), t0);
  // hard part: e^(m (p^4 - p^2 + 1)/r) = e^(p^3 (12z^3+6z^2+4z-1) + p^2 (12z^3+6z^2+6z) + p (12z^3+6z^2+4z) + 12z^3+12z^2+6z+1);
  // inverses are conjugates from here on
  const Fq12 A = f12_conj(f12_pow_z(e));        // e^-z
  const Fq12 B = f12_cyclo_sqr(A);              // e^-2z
  const Fq12 C = f12_cyclo_sqr(B);              // e^-4z
  const Fq12 D = f12_mul(C, B);                 // e^-6z
  const Fq12 E = f12_conj(f12_pow_z(D));        // e^(6z^2)
  const Fq12 F = f12_cyclo_sqr(E);              // e^(12z^2)
  const Fq12 G = f12_conj(f12_pow_z(F));        // e^(-12z^3)
  const Fq12 H = f12_conj(D);                   // e^(6z)
  const Fq12 I = f12_conj(G);                   // e^(12z^3)
  const Fq12 J = f12_mul(I, E);                 // e^(12z^3 + 6z^2)
  const Fq12 K = f12_mul(J, H);                 // e^(12z^3 + 6z^2 + 6z)
  const Fq12 L = f12_mul(K, B);                 // e^(12z^3 + 6z^2 + 4z)
  const Fq12 M = f12_mul(K, E);                 // e^(12z^3 + 12z^2 + 6z)
  const Fq12 N = f12_mul(M, e);                 // e^(12z^3 + 12z^2 + 6z + 1)
  const Fq12 O = f12_frob(L, 1, pc);
  const Fq12 P = f12_mul(O, N);
  const Fq12 Q = f12_frob(K, 2, pc);
  const Fq12 R = f12_mul(Q, P);
  const Fq12 S = f12_conj(e);
  const Fq12 T = f12_mul(S, L);
  const Fq12 U = f12_frob(T, 3, pc);
  return f12_mul(U, R);
}

// on-curve checks of untrusted proof points (affine, Montgomery form); infinity (all zero) is not accepted here
G16_HD bool g1_on_curve(const Affine<FqOps>& p) {
  Fq three = fp_one<FqParams>();
  three = fp_add(fp_add(three, three), three);
  return fp_eq(fp_sqr(p.y), fp_add(fp_mul(fp_sqr(p.x), p.x), three));
}
G16_HD bool g2_on_curve(const Affine<Fq2Ops>& q, const PairingConsts& pc) {
  return F2::eq(f2s(q.y), F2::add(f2m(f2s(q.x), q.x), pc.twist_b));
}

// ------------------------------------------------------------------ host-only: the constants
inline Fq2 f2_pow_words(const Fq2& a, const uint32_t* e, int nwords) {
  Fq2 r = F2::one();
  for (int i = nwords * 32 - 1; i >= 0; i--) {
    r = f2s(r);
    if ((e[i >> 5] >> (i & 31)) & 1) r = f2m(r, a);
  }
  return r;
}
inline void pairing_consts_init(PairingConsts& pc) {
  // (p - 1) / 6 as little-endian words
  uint32_t e[8];
  {
    static const uint32_t P[8] = G16_FQ_P;
    uint32_t t[8];
    for (int i = 0; i < 8; i++) t[i] = P[i];
    t[0] -= 1;
    uint64_t rem = 0;
    for (int i = 7; i >= 0; i--) {
      const uint64_t cur = (rem << 32) | t[i];
      e[i] = (uint32_t)(cur / 6);
      rem = cur % 6;
    }
  }
  const Fq one = fp_one<FqParams>();
  Fq nine = one;
  for (int i = 0; i < 3; i++) nine = fp_add(nine, nine);   // 8
  nine = fp_add(nine, one);
  const Fq2 xi{nine, one};
  const Fq2 g1 = f2_pow_words(xi, e, 8);                   // xi^((p-1)/6)
  pc.frob1[0] = F2::one();
  for (int k = 1; k < 6; k++) pc.frob1[k] = f2m(pc.frob1[k - 1], g1);
  for (int k = 0; k < 6; k++) {
    pc.frob2[k] = f2m(pc.frob1[k], f2_conj(pc.frob1[k]));
    pc.frob3[k] = f2m(pc.frob2[k], pc.frob1[k]);
  }
  pc.twist_b = Fq2{Fq{G16_G2B_C0}, Fq{G16_G2B_C1}};
  pc.two_inv = fp_inv(fp_add(one, one));
}

}  // namespace g16
