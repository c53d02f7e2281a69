// PLONK prover on the device (second half: "PLONK prover (the only flow the Makefile actually
// scripts, /root/reference/Makefile:30-33) -- different protocol, reuse Fr NTT + G1 MSM kernels").
//
// Replaces `snarkjs plonk prove` ([EXT] snarkjs 0.4.12 plonk_prove.js, pin /root/reference/yarn.lock:987-1001) for a
// PLONK .zkey as `snarkjs plonk setup` writes it (protocol id 2; sections: header, additions, A/B/C signal maps,
// Qm Ql Qr Qo Qc, sigma 1-3, Lagrange polynomials, powers of tau; a polynomial = N coefficients + 4N evaluations).
// Same five rounds, same Keccak-256 transcript, same blinding convention (b1..b9), so that with the same blinding
// scalars the proof is the proof snarkjs computes; restated in oracle/plonk.py, which this file must match bit for bit.
//
// Device schedule (everything below on the canonical 8x32 Montgomery Fr of fp.cuh; transforms through ntt.hip's lazy
// 9x29 format; commitments through the MSM group machinery with the powers of tau as a dense, window-precomputed
// base section):
//   witness   additions by dependency level (the setup's queue-order reduction makes a balanced tree: <= ~10 levels),
//             A/B/C = w[map]
//   round 1   3 x (iNTT N, blinding tweak, NTT 4N), 3 commitments
//   round 2   grand product: per-lane chunks (batched inversion of the denominators, local prefix products), the chunk
//             carries by one workgroup, Z; iNTT N, NTT 4N, commitment
//   round 3   one elementwise kernel over the 4N domain (gate, permutation and L1 terms with the blinding parts tracked
//             apart: T and Tz), 2 iNTT 4N, division by Z_H as a stride-N recurrence, 3 commitments
//   round 4   evaluations at xi by chunked Horner + a one-workgroup carry combination, linearisation polynomial r
//   round 5   opening polynomials by chunked synthetic division, 2 commitments
// The public-input polynomial comes from an iNTT/NTT of the public signals, not from the zkey's Lagrange section
// (which is not read: 513 public signals at N = 2^22 make it 344 GB).
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <mutex>
#include <string>
#include <vector>

#include "fr29.cuh"
#include "internal.h"
#include "transcript.h"

namespace g16 {
namespace {

using transcript::FrM;   // Montgomery residue
constexpr uint32_t kChunk = 64;     // grand-product chunk per lane
constexpr uint32_t kHorner = 256;   // Horner / synthetic-division chunk per lane

// ------------------------------------------------------------------ host field helpers (fp.cuh host build)
FrM h_from_u64(uint64_t v) {
  Fr a = fp_zero<FrParams>();
  a.v[0] = (uint32_t)v;
  a.v[1] = (uint32_t)(v >> 32);
  return fp_to_mont(a);
}
FrM h_root(int L) {
  Fr w = {G16_FR_W28};
  for (int i = 28; i > L; i--) w = fp_sqr(w);
  return w;
}
FrM h_pow(FrM a, uint64_t e) { return fp_pow_u64(a, e); }
// a^e from the top set bit of e down (fp_pow_u64 walks all 64 bits: 64 squarings for an exponent of 4 bits -- a third
// of k_suffix_horner, 7 % of k_z_ratio)
__device__ __forceinline__ FrM frm_pow(const FrM& a, uint64_t e) {
  if (e == 0) return fp_one<FrParams>();
  FrM r = a;
  for (int i = 62 - (int)__clzll((long long)e); i >= 0; i--) {
    r = fp_sqr(r);
    if ((e >> i) & 1) r = fp_mul(r, a);
  }
  return r;
}

using namespace transcript;

// ------------------------------------------------------------------ device kernels
__global__ __launch_bounds__(256) void k_to_mont(const Fr* __restrict__ in, FrM* __restrict__ out, uint32_t n, uint32_t zero_first) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  out[i] = (zero_first && i == 0) ? fp_zero<FrParams>() : fp_to_mont(in[i]);
}
// Montgomery -> standard form, zero padded to n_out (MSM scalars)
__global__ __launch_bounds__(256) void k_from_mont_pad(const FrM* __restrict__ in, uint32_t n_in, Fr* __restrict__ out, uint32_t n_out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_out) return;
  out[i] = i < n_in ? fp_from_mont(in[i]) : fp_zero<FrParams>();
}
__global__ __launch_bounds__(256) void k_additions(const uint32_t* __restrict__ order, uint32_t lo, uint32_t hi,
                                                   const uint32_t* __restrict__ s1, const uint32_t* __restrict__ s2,
                                                   const FrM* __restrict__ f1, const FrM* __restrict__ f2,
                                                   FrM* __restrict__ w, uint32_t base) {
  const uint32_t t = lo + blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= hi) return;
  const uint32_t k = order[t];
  w[base + k] = fp_add(fp_mul(f1[k], w[s1[k]]), fp_mul(f2[k], w[s2[k]]));
}
__global__ __launch_bounds__(256) void k_gather(const FrM* __restrict__ w, const uint32_t* __restrict__ map, FrM* __restrict__ out, uint32_t n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  out[i] = w[map[i]];
}
// pol = coefs (n) with the blinding polynomial (pz[0] + pz[1] X + ...)(X^n - 1) added: pol has n + npz entries
struct Pz { FrM v[3]; uint32_t n; };
__global__ void k_blind(const FrM* __restrict__ coefs, uint32_t n, Pz pz, FrM* __restrict__ pol) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n + pz.n) return;
  FrM x = i < n ? coefs[i] : fp_zero<FrParams>();
  if (i < pz.n) x = fp_sub(x, pz.v[i]);
  if (i >= n) x = fp_add(x, pz.v[i - n]);
  pol[i] = x;
}
__global__ __launch_bounds__(256) void k_pad4(const FrM* __restrict__ coefs, uint32_t n, FrM* __restrict__ out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 4 * n) return;
  out[i] = i < n ? coefs[i] : fp_zero<FrParams>();
}
// g^j from two tables: hi[j >> 12] * lo[j & 4095] (Montgomery words)
struct PowTab { const FrM* lo; const FrM* hi; };
__device__ __forceinline__ FrM pow_tab(const PowTab& t, uint32_t j) { return fp_mul(t.hi[j >> 12], t.lo[j & 4095u]); }
// The coefficients of p(g X), zero padded to 4N -- the 4N-point transform of these is p on the coset g <w_4N> -- written
// straight into the forward transform's working vector: lazy format, bit-reversed order (what ntt_dit_forward
// reads).  A GATHER: output p takes coefficient bitrev(p), and only a quarter of the outputs have one
// (ntt_import scatters 40-byte elements: 1.3-2.1 ms per 2^24 vector against 0.3 here).
__global__ __launch_bounds__(256) void k_coset_import(const FrM* __restrict__ coefs, uint32_t len, int L4, PowTab g,
                                                      F29* __restrict__ out) {
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= (1u << L4)) return;
  const uint32_t src = __brev(p) >> (32 - L4);
  out[p] = src < len ? fr29_from_fr(fp_mul(coefs[src], pow_tab(g, src))) : f29_zero();
}
// Back from t(g X) to t: coefficient j of the inverse transform (its working vector is lazy, bit-reversed, unscaled: a
// gather again) times g^-j / 4N; bad[0] is set when a coefficient at or above `keep` is not zero (the numerator was not
// a multiple of X^n - 1: snarkjs' "T Polynomial is not divisible")
__global__ __launch_bounds__(256) void k_coset_export_check(const F29* __restrict__ in, F29 ninv, int L4, uint32_t keep,
                                                            PowTab ginv, FrM* __restrict__ t, uint32_t* __restrict__ bad) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= (1u << L4)) return;
  const FrM v = fr29_to_fr(fr29_mul(in[__brev(j) >> (32 - L4)], ninv));
  if (j >= keep) {
    if (!fp_is_zero(v)) atomicOr(&bad[0], 1u);
    t[j] = v;
    return;
  }
  t[j] = fp_mul(v, pow_tab(ginv, j));
}
__global__ __launch_bounds__(256) void k_mul_const(FrM* __restrict__ x, size_t n, FrM c) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) x[i] = fp_mul(x[i], c);
}
__global__ __launch_bounds__(256) void k_stride4(const FrM* __restrict__ in, uint32_t n, FrM* __restrict__ out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = in[4 * (size_t)i];
}
// out[i] = w^i, i < n (each lane: one power by square-and-multiply, then kChunk successive products)
__global__ __launch_bounds__(256) void k_powers(FrM w, uint32_t n, FrM* __restrict__ out) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t lo = t * kChunk;
  if (lo >= n) return;
  FrM x = frm_pow(w, lo);
  const uint32_t hi = lo + kChunk < n ? lo + kChunk : n;
  for (uint32_t i = lo; i < hi; i++) {
    out[i] = x;
    x = fp_mul(x, w);
  }
}

// Round 2, the grand product.  A wavefront owns kZBlock consecutive rows; lane l takes rows base + 64 k + l, so
// every load and store of a step is one coalesced 2 KB run (first version: a lane per 64 CONSECUTIVE rows -- fifteen
// strided passes over the vectors, 4.1 ms of k_z_local + 1.2 ms for the 65 k chunk carries on one workgroup).
constexpr uint32_t kZSteps = 32;              // rows per lane (r02, round 2 at N = 2^22: 64 -> 16.2 ms, 32 -> 12.0, 16 -> 13.5)
constexpr uint32_t kZBlock = 64 * kZSteps;
struct R2Args { FrM beta, gamma, k1, k2, w1, w64; };
// ratio[i] = num_i / den_i; the denominators of a lane's 64 rows are inverted in one batch (Montgomery's trick works on
// any set of elements, contiguous or not).  S1..S3: sigma evaluations at the N domain points.
__global__ __launch_bounds__(256) void k_z_ratio(const FrM* __restrict__ A, const FrM* __restrict__ B, const FrM* __restrict__ C,
                                                 const FrM* __restrict__ S1, const FrM* __restrict__ S2, const FrM* __restrict__ S3,
                                                 R2Args a, uint32_t n, FrM* __restrict__ num, FrM* __restrict__ den,
                                                 FrM* __restrict__ pre) {
  const uint32_t gt = blockIdx.x * blockDim.x + threadIdx.x;
  const size_t base = (size_t)(gt >> 6) * kZBlock + (gt & 63u);
  if (base >= n) return;
  FrM w = frm_pow(a.w1, base);
  FrM run = fp_one<FrParams>();
  uint32_t steps = 0;
  for (size_t i = base; i < n && steps < kZSteps; i += 64, steps++) {
    const FrM bw = fp_mul(a.beta, w);
    const FrM n1 = fp_add(fp_add(A[i], bw), a.gamma);
    const FrM n2 = fp_add(fp_add(B[i], fp_mul(a.k1, bw)), a.gamma);
    const FrM n3 = fp_add(fp_add(C[i], fp_mul(a.k2, bw)), a.gamma);
    num[i] = fp_mul(fp_mul(n1, n2), n3);
    const FrM d1 = fp_add(fp_add(A[i], fp_mul(a.beta, S1[i])), a.gamma);
    const FrM d2 = fp_add(fp_add(B[i], fp_mul(a.beta, S2[i])), a.gamma);
    const FrM d3 = fp_add(fp_add(C[i], fp_mul(a.beta, S3[i])), a.gamma);
    const FrM d = fp_mul(fp_mul(d1, d2), d3);
    den[i] = d;
    pre[i] = run;
    run = fp_mul(run, d);
    w = fp_mul(w, a.w64);
  }
  FrM inv = fp_inv(run);
  for (uint32_t k = steps; k-- > 0;) {
    const size_t i = base + 64 * (size_t)k;
    const FrM di = fp_mul(inv, pre[i]);
    inv = fp_mul(inv, den[i]);
    num[i] = fp_mul(num[i], di);   // the ratio
  }
}
__device__ __forceinline__ FrM frm_shfl_up(const FrM& v, int d) {
  FrM r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.v[i] = (uint32_t)__shfl_up((int)v.v[i], d, 64);
  return r;
}
__device__ __forceinline__ FrM frm_shfl(const FrM& v, int lane) {
  FrM r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.v[i] = (uint32_t)__shfl((int)v.v[i], lane, 64);
  return r;
}
// lp[i] = product of the block's ratios up to and including row i (64 rows per step: a shuffle scan across the
// wavefront, times the running product of the steps before); totals[block] = the block's product
__global__ __launch_bounds__(256) void k_z_scan(const FrM* __restrict__ ratio, uint32_t n, FrM* __restrict__ lp, FrM* __restrict__ totals) {
  const uint32_t gt = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t blk = gt >> 6, l = gt & 63u;
  const size_t base0 = (size_t)blk * kZBlock;
  if (base0 >= n) return;   // whole wavefronts
  FrM carry = fp_one<FrParams>();
  for (uint32_t k = 0; k < kZSteps && base0 + 64 * (size_t)k < n; k++) {
    const size_t i = base0 + 64 * (size_t)k + l;
    FrM v = i < n ? ratio[i] : fp_one<FrParams>();
    for (int d = 1; d < 64; d <<= 1) {
      const FrM u = frm_shfl_up(v, d);
      if (l >= (uint32_t)d) v = fp_mul(v, u);
    }
    v = fp_mul(v, carry);
    if (i < n) lp[i] = v;
    carry = frm_shfl(v, 63);
  }
  if (l == 0) totals[blk] = carry;
}
// Z[0] = 1, Z[i + 1] = carry[chunk(i)] * lp[i]
__global__ __launch_bounds__(256) void k_z_apply(const FrM* __restrict__ lp, const FrM* __restrict__ carry, uint32_t n, FrM* __restrict__ Z) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (i == 0) Z[0] = fp_one<FrParams>();
  if (i + 1 < n) Z[i + 1] = fp_mul(carry[i / kZBlock], lp[i]);
}

// Round 3.  snarkjs evaluates the UNBLINDED polynomials on the 4N subgroup and carries the blinding factors along as
// a second polynomial (T and Tz, `mul4` with its Z1/Z2/Z3 tables), because X^n - 1 vanishes on a quarter of that
// domain and the quotient has to be taken coefficient-wise.  The quotient polynomial t itself does not depend on how
// it is computed, so this prover evaluates the BLINDED polynomials on the coset g <w_4N> (g = w_8N: X^n - 1 takes four
// non-zero values there), forms numerator / (X^n - 1) point by point and transforms back: 21 products per point instead
// of 77, one inverse transform instead of two, and the same coefficients bit for bit.
//
// It runs on the 9 x 29-bit lazy Montgomery field of the NTT kernels (fr29.cuh: 241 instructions per product against
// ~530 of the canonical 8x32 product): its inputs are the 4N-point transforms left in the lazy format (no export), its
// output goes straight into the inverse transform's working vector (no import).  Discipline: every sum and difference
// is weak-reduced (< 1.0001 r), so every product input is < 2.5 r and every subtrahend < 3 r.
using L9 = F29;
__device__ __forceinline__ L9 lmul(const L9& a, const L9& b) { return fr29_mul(a, b); }
__device__ __forceinline__ L9 ladd(const L9& a, const L9& b) { return fr29_weak_reduce(fr29_add(a, b)); }
__device__ __forceinline__ L9 lsub(const L9& a, const L9& b) { return fr29_weak_reduce(fr29_sub<3>(a, b)); }
__global__ __launch_bounds__(256) void k_to_lazy(const FrM* __restrict__ in, L9* __restrict__ out, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  out[i] = fr29_from_fr(in[i]);
}
struct R3Args {
  L9 beta, gamma, alpha, alpha2, k1, k2, one;
  L9 zhinv[4];   // 1 / (x^n - 1) at x = g w_4N^i: depends on i mod 4 only
};
struct R3Ptrs {
  const L9 *A4, *B4, *C4, *Z4, *qm, *ql, *qr, *qo, *qc, *s1, *s2, *s3, *pi4, *l1, *x4;
};
__global__ __launch_bounds__(256) void k_round3(R3Ptrs q, R3Args a, uint32_t n4, L9* __restrict__ T) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  auto ld = [&](const L9* v, uint32_t k) { return fr29_weak_reduce(v[k]); };   // (transform outputs may sit anywhere below 16 r)
  const L9 A = ld(q.A4, i), B = ld(q.B4, i), C = ld(q.C4, i);
  // gate
  L9 e = lmul(lmul(A, B), ld(q.qm, i));
  e = ladd(e, ladd(lmul(A, ld(q.ql, i)), lmul(B, ld(q.qr, i))));
  e = ladd(e, ladd(lmul(C, ld(q.qo, i)), ladd(ld(q.pi4, i), ld(q.qc, i))));
  // permutation
  const L9 bx = lmul(a.beta, q.x4[i]);
  L9 e2 = lmul(ladd(ladd(A, bx), a.gamma), ladd(ladd(B, lmul(bx, a.k1)), a.gamma));
  e2 = lmul(e2, ladd(ladd(C, lmul(bx, a.k2)), a.gamma));
  const L9 Z = ld(q.Z4, i);
  e2 = lmul(e2, Z);
  L9 e3 = lmul(ladd(ladd(A, lmul(a.beta, ld(q.s1, i))), a.gamma), ladd(ladd(B, lmul(a.beta, ld(q.s2, i))), a.gamma));
  e3 = lmul(e3, ladd(ladd(C, lmul(a.beta, ld(q.s3, i))), a.gamma));
  e3 = lmul(e3, ld(q.Z4, (i + 4) % n4));   // z(w X)
  e = ladd(e, lmul(a.alpha, lsub(e2, e3)));
  e = ladd(e, lmul(lmul(lsub(Z, a.one), ld(q.l1, i)), a.alpha2));
  T[i] = lmul(e, a.zhinv[i & 3u]);
}
// H[c] = sum_{j < len_c} P[c kHorner + j] x^j
__global__ __launch_bounds__(256) void k_horner(const FrM* __restrict__ P, uint32_t n, FrM x, FrM* __restrict__ H) {
  const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t lo = c * kHorner;
  if (lo >= n) return;
  const uint32_t hi = lo + kHorner < n ? lo + kHorner : n;
  FrM r = fp_zero<FrParams>();
  for (uint32_t i = hi; i-- > lo;) r = fp_add(fp_mul(r, x), P[i]);
  H[c] = r;
}
// synthetic division by (X - d): res[i] = P[i+1] + d res[i+1]; E[c + 1] = res at the top index of chunk c
__global__ __launch_bounds__(256) void k_divpol(const FrM* __restrict__ P, uint32_t n, FrM d, const FrM* __restrict__ E, FrM* __restrict__ res) {
  const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t lo = c * kHorner;
  if (lo >= n) return;
  const uint32_t hi = lo + kHorner < n ? lo + kHorner : n;
  FrM r = E[c + 1];
  for (uint32_t i = hi; i-- > lo;) {
    res[i] = r;
    r = fp_add(P[i], fp_mul(d, r));
  }
}
// The carries between chunks, on the device (one workgroup of 1024 lanes, up to 65 536 chunk values; first version:
// copied to the host, combined there, copied back -- ~11 ms of a 113 ms proof).
// E[c] = H[c] + m E[c + 1], E[nc] = 0: the chunk values of a Horner evaluation / synthetic division combined.
__global__ __launch_bounds__(1024) void k_suffix_horner(const FrM* __restrict__ H, uint32_t nc, FrM m, FrM* __restrict__ E) {
  __shared__ FrM sm[1024];
  const uint32_t t = threadIdx.x;
  const uint32_t B = (nc + 1023) / 1024;
  const uint32_t lo = t * B < nc ? t * B : nc, hi = lo + B < nc ? lo + B : nc;
  FrM r = fp_zero<FrParams>();
  for (uint32_t i = hi; i-- > lo;) {
    r = fp_add(H[i], fp_mul(m, r));
    E[i] = r;
  }
  sm[t] = r;
  FrM pw = frm_pow(m, B);
  __syncthreads();
  for (uint32_t d = 1; d < 1024; d <<= 1) {
    const FrM v = t + d < 1024 ? sm[t + d] : fp_zero<FrParams>();
    __syncthreads();
    sm[t] = fp_add(sm[t], fp_mul(pw, v));
    pw = fp_sqr(pw);
    __syncthreads();
  }
  const FrM carry = t + 1 < 1024 ? sm[t + 1] : fp_zero<FrParams>();
  FrM q = m;
  for (uint32_t i = hi; i-- > lo;) {
    E[i] = fp_add(E[i], fp_mul(q, carry));
    q = fp_mul(q, m);
  }
  if (t == 0) E[nc] = fp_zero<FrParams>();
}
// carry[c] = prod_{j < c} tot[j]; flag set when the product of all is not one
__global__ __launch_bounds__(1024) void k_prefix_prod(const FrM* __restrict__ tot, uint32_t nc, FrM* __restrict__ carry,
                                                       uint32_t* __restrict__ flag) {
  __shared__ FrM sm[1024];
  const uint32_t t = threadIdx.x, T = blockDim.x;   // T <= 1024, a power of two
  const uint32_t B = (nc + T - 1) / T;
  const uint32_t lo = t * B < nc ? t * B : nc, hi = lo + B < nc ? lo + B : nc;
  FrM r = fp_one<FrParams>();
  for (uint32_t i = lo; i < hi; i++) {
    carry[i] = r;
    r = fp_mul(r, tot[i]);
  }
  sm[t] = r;
  __syncthreads();
  for (uint32_t d = 1; d < T; d <<= 1) {
    const FrM v = t >= d ? sm[t - d] : fp_one<FrParams>();
    __syncthreads();
    sm[t] = fp_mul(sm[t], v);
    __syncthreads();
  }
  const FrM c0 = t ? sm[t - 1] : fp_one<FrParams>();
  for (uint32_t i = lo; i < hi; i++) carry[i] = fp_mul(carry[i], c0);
  if (t == T - 1 && !fp_eq(sm[T - 1], fp_one<FrParams>())) atomicOr(flag, 1u);
}
__global__ void k_flag_nonzero(const FrM* __restrict__ x, uint32_t* __restrict__ flag) {
  if (threadIdx.x == 0 && blockIdx.x == 0 && !fp_is_zero(x[0])) atomicOr(flag, 1u);
}
__global__ void k_copy1(const FrM* __restrict__ src, FrM* __restrict__ dst) {
  if (threadIdx.x == 0 && blockIdx.x == 0) dst[0] = src[0];
}

// Several evaluations at once (round 4: a, b, c, s1, s2, t at xi and z at xi w): blockIdx.y = the job
struct EvalJob { const FrM* pol; uint32_t n; FrM x, xc; };
struct EvalJobs { EvalJob j[8]; uint32_t stride; };
__global__ __launch_bounds__(256) void k_horner_multi(EvalJobs jobs, FrM* __restrict__ H) {
  const EvalJob& jb = jobs.j[blockIdx.y];
  const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t lo = c * kHorner;
  if (lo >= jb.n) return;
  const uint32_t hi = lo + kHorner < jb.n ? lo + kHorner : jb.n;
  FrM r = fp_zero<FrParams>();
  for (uint32_t i = hi; i-- > lo;) r = fp_add(fp_mul(r, jb.x), jb.pol[i]);
  H[(size_t)blockIdx.y * jobs.stride + c] = r;
}
// one workgroup per job: the chunk values combined (as k_suffix_horner), only the total kept: out[job] = P_job(x_job)
__global__ __launch_bounds__(1024) void k_suffix_total_multi(EvalJobs jobs, const FrM* __restrict__ H, FrM* __restrict__ out) {
  __shared__ FrM sm[1024];
  const EvalJob& jb = jobs.j[blockIdx.x];
  const FrM* h = H + (size_t)blockIdx.x * jobs.stride;
  const uint32_t nc = (jb.n + kHorner - 1) / kHorner;
  const uint32_t t = threadIdx.x;
  const uint32_t B = (nc + 1023) / 1024;
  const uint32_t lo = t * B < nc ? t * B : nc, hi = lo + B < nc ? lo + B : nc;
  FrM r = fp_zero<FrParams>();
  for (uint32_t i = hi; i-- > lo;) r = fp_add(h[i], fp_mul(jb.xc, r));
  sm[t] = r;
  FrM pw = frm_pow(jb.xc, B);
  __syncthreads();
  for (uint32_t d = 1; d < 1024; d <<= 1) {
    const FrM v = t + d < 1024 ? sm[t + d] : fp_zero<FrParams>();
    __syncthreads();
    sm[t] = fp_add(sm[t], fp_mul(pw, v));
    pw = fp_sqr(pw);
    __syncthreads();
  }
  if (t == 0) out[blockIdx.x] = sm[0];
}

struct R4Args { FrM coefz, coef_ab, ea, eb, ec, coefs3; };
// pol_r[i] = coefz z[i] (+ coef_ab qm + ea ql + eb qr + ec qo + qc - coefs3 s3 for i < n), i < n + 3
__global__ __launch_bounds__(256) void k_pol_r(const FrM* __restrict__ z, const FrM* __restrict__ qm, const FrM* __restrict__ ql,
                                               const FrM* __restrict__ qr, const FrM* __restrict__ qo, const FrM* __restrict__ qc,
                                               const FrM* __restrict__ s3, R4Args a, uint32_t n, FrM* __restrict__ r) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n + 3) return;
  FrM v = fp_mul(a.coefz, z[i]);
  if (i < n) {
    v = fp_add(v, fp_mul(a.coef_ab, qm[i]));
    v = fp_add(v, fp_mul(a.ea, ql[i]));
    v = fp_add(v, fp_mul(a.eb, qr[i]));
    v = fp_add(v, fp_mul(a.ec, qo[i]));
    v = fp_add(v, qc[i]);
    v = fp_sub(v, fp_mul(a.coefs3, s3[i]));
  }
  r[i] = v;
}
struct R5Args { FrM v[7], xim, xi2m, sub0; };
__global__ __launch_bounds__(256) void k_pol_wxi(const FrM* __restrict__ t, const FrM* __restrict__ r, const FrM* __restrict__ pa,
                                                 const FrM* __restrict__ pb, const FrM* __restrict__ pc, const FrM* __restrict__ s1,
                                                 const FrM* __restrict__ s2, R5Args a, uint32_t n, FrM* __restrict__ out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n + 6) return;
  FrM w = fp_mul(a.xi2m, t[2 * (size_t)n + i]);
  if (i < n + 3) w = fp_add(w, fp_mul(a.v[1], r[i]));
  if (i < n + 2) {
    w = fp_add(w, fp_mul(a.v[2], pa[i]));
    w = fp_add(w, fp_mul(a.v[3], pb[i]));
    w = fp_add(w, fp_mul(a.v[4], pc[i]));
  }
  if (i < n) {
    w = fp_add(w, t[i]);
    w = fp_add(w, fp_mul(a.xim, t[(size_t)n + i]));
    w = fp_add(w, fp_mul(a.v[5], s1[i]));
    w = fp_add(w, fp_mul(a.v[6], s2[i]));
  }
  if (i == 0) w = fp_sub(w, a.sub0);
  out[i] = w;
}
__global__ void k_sub0(FrM* __restrict__ p, FrM v) {
  if (threadIdx.x == 0 && blockIdx.x == 0) p[0] = fp_sub(p[0], v);
}
// -pub[j] at j < n_public, 0 elsewhere (the N evaluations of the public-input polynomial)
__global__ __launch_bounds__(256) void k_pi_evals(const FrM* __restrict__ A, uint32_t n_public, uint32_t n, FrM* __restrict__ out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  out[i] = i < n_public ? fp_neg(A[i]) : fp_zero<FrParams>();
}

inline unsigned nblk(size_t n, unsigned per = 256) { return (unsigned)((n + per - 1) / per); }

}  // namespace
}  // namespace g16

using namespace g16;

struct g16_plonk {
  int device = 0;
  uint32_t N = 0, L = 0, nVars = 0, nPublic = 0, nAdd = 0, nCons = 0, nBase = 0;
  FrM k1, k2, w1;
  hipStream_t st = nullptr, st2 = nullptr;   // main chain (commitments); the 4N transforms of a round beside its commitments
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  NttTables ntt_n, ntt_4n;
  FrM* d_sig[3] = {};     // S1 S2 S3 at the N domain points (every 4th of the zkey's 4N evaluations), canonical: round 2
  F29* d_ext_l[8] = {};   // Qm Ql Qr Qo Qc S1 S2 S3, 4N evaluations in the lazy format: round 3
  FrM* d_pol[8] = {};     // ... N coefficients
  uint32_t* d_map[3] = {};
  uint32_t *d_add_s1 = nullptr, *d_add_s2 = nullptr, *d_add_order = nullptr;
  FrM *d_add_f1 = nullptr, *d_add_f2 = nullptr;
  std::vector<uint32_t> level_start;
  F29 *d_om4 = nullptr, *d_l1 = nullptr;   // the points g w_4N^i of the round-3 coset and L1 on it, lazy format
  FrM* d_gtab = nullptr;                   // g^j as (lo: 4096 words, hi: gtab_hi words), then g^-j likewise: see PowTab
  size_t gtab_hi = 1;
  FrM gN;                                  // g^N
  PowTab tab_g() const { return PowTab{d_gtab, d_gtab + 4096}; }
  PowTab tab_ginv() const { return PowTab{d_gtab + 4096 + gtab_hi, d_gtab + 8192 + gtab_hi}; }
  MsmGroup srs;
  MsmWorkspace* ws = nullptr;             // slot 0 of the commitment lanes (on the main stream)
  MsmWorkspace* wsx[2] = {nullptr, nullptr};   // slots 1, 2: independent commitments of a round run side by side
  hipStream_t mst[2] = {nullptr, nullptr};
  Fr* d_scalx[2] = {nullptr, nullptr};
  hipEvent_t ev_pol = nullptr;
  // per-proof scratch
  Fr* d_wraw = nullptr;                 // witness as uploaded (standard form)
  FrM* d_w = nullptr;                   // extended witness, Montgomery
  FrM *d_A = nullptr, *d_B = nullptr, *d_C = nullptr, *d_Z = nullptr;   // N evaluations
  FrM *d_pa = nullptr, *d_pb = nullptr, *d_pc = nullptr, *d_pz = nullptr;   // blinded coefficient forms (N + 3)
  F29 *d_A4 = nullptr, *d_B4 = nullptr, *d_C4 = nullptr, *d_Z4 = nullptr, *d_pi4 = nullptr;   // 4N evaluations, lazy format
  FrM* d_T = nullptr;                   // 4N canonical words: padding scratch, then the quotient's coefficients
  FrM *d_tmpN = nullptr, *d_tmpN2 = nullptr, *d_tmpN3 = nullptr, *d_tmpN4 = nullptr;   // N-sized scratch
  FrM *d_cA = nullptr, *d_cB = nullptr, *d_cC = nullptr, *d_cZ = nullptr;   // unblinded coefficients (the side stream pads them)
  FrM *d_pi_ev = nullptr, *d_pi_co = nullptr;
  FrM *d_r = nullptr, *d_wxi = nullptr, *d_q = nullptr;
  FrM *d_tot = nullptr, *d_tot2 = nullptr;   // chunk totals / Horner partials, and their combined carries
  FrM* d_evals = nullptr;               // 8 evaluation results, read back once per round
  F29 *d_lazy = nullptr, *d_lazy2 = nullptr;   // 4N lazy elements each: the transforms' working vectors (d_lazy: t in round 3)
  Fr* d_scal = nullptr;                 // N + 6 standard-form MSM scalars
  uint32_t* d_bad = nullptr;
  float last_ms[6] = {};
  std::mutex mu;
  ~g16_plonk() {
    (void)hipSetDevice(device);
    for (auto p : d_sig) if (p) (void)hipFree(p);
    for (auto p : d_ext_l) if (p) (void)hipFree(p);
    for (auto p : d_pol) if (p) (void)hipFree(p);
    for (auto p : d_map) if (p) (void)hipFree(p);
    void* v[] = {d_add_s1, d_add_s2, d_add_order, d_add_f1, d_add_f2, d_om4, d_l1, d_wraw, d_w, d_A, d_B, d_C, d_Z, d_pa, d_pb, d_pc,
                 d_pz, d_A4, d_B4, d_C4, d_Z4, d_T, d_gtab, d_pi4, d_tmpN, d_tmpN2, d_tmpN3, d_tmpN4, d_r, d_wxi, d_q, d_tot, d_lazy, d_lazy2, d_cA, d_cB, d_cC, d_cZ, d_pi_ev, d_pi_co, d_tot2, d_evals,
                 d_scal, d_bad};
    for (void* p : v) if (p) (void)hipFree(p);
    if (ws) msm_workspace_destroy(ws);
    for (auto w : wsx) if (w) msm_workspace_destroy(w);
    for (auto p : d_scalx) if (p) (void)hipFree(p);
    for (auto x : mst) if (x) (void)hipStreamDestroy(x);
    if (ev_pol) (void)hipEventDestroy(ev_pol);
    msm_group_destroy(srs);
    ntt_tables_destroy(ntt_n);
    ntt_tables_destroy(ntt_4n);
    if (ev_fork) (void)hipEventDestroy(ev_fork);
    if (ev_join) (void)hipEventDestroy(ev_join);
    if (st2) (void)hipStreamDestroy(st2);
    if (st) (void)hipStreamDestroy(st);
  }
};

namespace {

struct Sec { const uint8_t* p = nullptr; uint64_t size = 0; };
int find_sections(const uint8_t* buf, size_t len, const char* magic, Sec out[16], const char* name) {
  if (len < 12 || memcmp(buf, magic, 4) != 0) { set_error(std::string(name) + ": Invalid File format"); return G16_E_FORMAT; }
  uint32_t version, nsec;
  memcpy(&version, buf + 4, 4);
  memcpy(&nsec, buf + 8, 4);
  if (version > 2) { set_error("Version not supported"); return G16_E_FORMAT; }
  size_t pos = 12;
  for (uint32_t i = 0; i < nsec; i++) {
    if (pos + 12 > len) { set_error(std::string(name) + ": Invalid File format"); return G16_E_FORMAT; }
    uint32_t id;
    uint64_t size;
    memcpy(&id, buf + pos, 4);
    memcpy(&size, buf + pos + 4, 8);
    pos += 12;
    if (size > len - pos) { set_error(std::string(name) + ": Invalid File format"); return G16_E_FORMAT; }
    if (id < 16 && !out[id].p) { out[id].p = buf + pos; out[id].size = size; }
    pos += size;
  }
  return G16_OK;
}

// transforms of Montgomery vectors through the lazy working vector (natural order in and out)
int do_ifft(const NttTables& t, const FrM* in, FrM* out, F29* lazy, hipStream_t st) {
  int rc = ntt_import(t, in, lazy, false, st);
  F29* v[1] = {lazy};
  if (!rc) rc = ntt_dif_inverse(t, v, 1, st);
  if (!rc) rc = ntt_export(t, lazy, out, true, true, st);
  return rc;
}
int do_fft(const NttTables& t, const FrM* in, FrM* out, F29* lazy, hipStream_t st) {
  int rc = ntt_import(t, in, lazy, true, st);
  F29* v[1] = {lazy};
  if (!rc) rc = ntt_dit_forward(t, v, 1, st);
  if (!rc) rc = ntt_export(t, lazy, out, false, false, st);
  return rc;
}

// `len` Montgomery coefficients of p -> p on the round-3 coset g <w_4N>, 4N evaluations in the lazy format (natural order)
int coset_fft(g16_plonk* P, const FrM* d_coefs, uint32_t len, F29* out, hipStream_t st) {
  const size_t n4 = (size_t)P->N * 4;
  k_coset_import<<<nblk(n4), 256, 0, st>>>(d_coefs, len, (int)P->L + 2, P->tab_g(), out);
  G16_HIP(hipGetLastError());
  F29* v[1] = {out};
  return ntt_dit_forward(P->ntt_4n, v, 1, st);
}

int plonk_create_impl(const uint8_t* zkey, size_t len, int device, g16_plonk* P) {
  Sec s[16];
  int rc = find_sections(zkey, len, "zkey", s, "zkey");
  if (rc) return rc;
  if (!s[1].p || s[1].size < 4) { set_error("zkey: Invalid File format"); return G16_E_FORMAT; }
  uint32_t proto;
  memcpy(&proto, s[1].p, 4);
  if (proto != 2) { set_error("zkey file is not plonk"); return G16_E_FORMAT; }
  const size_t hdr = 4 + 32 + 4 + 32 + 20 + 64 + 8 * 64 + 128;
  if (!s[2].p || s[2].size < hdr) { set_error("zkey: Invalid File format"); return G16_E_FORMAT; }
  {
    static const uint32_t Qp[8] = G16_FQ_P, Rp[8] = G16_FR_P;
    uint32_t n8q, n8r;
    memcpy(&n8q, s[2].p, 4);
    memcpy(&n8r, s[2].p + 36, 4);
    if (n8q != 32 || n8r != 32 || memcmp(s[2].p + 4, Qp, 32) != 0 || memcmp(s[2].p + 40, Rp, 32) != 0) {
      set_error("zkey: curve not supported (bn128 only)");
      return G16_E_FORMAT;
    }
  }
  const uint8_t* h = s[2].p + 72;
  memcpy(&P->nVars, h, 4);
  memcpy(&P->nPublic, h + 4, 4);
  memcpy(&P->N, h + 8, 4);
  memcpy(&P->nAdd, h + 12, 4);
  memcpy(&P->nCons, h + 16, 4);
  memcpy(P->k1.v, h + 20, 32);
  memcpy(P->k2.v, h + 52, 32);
  const uint32_t N = P->N;
  if (N < 8 || (N & (N - 1)) || N > (1u << 24) || P->nCons > N || P->nAdd > P->nVars || P->nPublic >= P->nVars - P->nAdd) {
    set_error("zkey: Invalid File format");
    return G16_E_FORMAT;
  }
  while ((1u << P->L) < N) P->L++;
  P->nBase = P->nVars - P->nAdd;
  P->w1 = h_root((int)P->L);
  const size_t polb = (size_t)N * 32 * 5;
  if (!s[3].p || s[3].size != (uint64_t)P->nAdd * 72 || s[4].size != (uint64_t)P->nCons * 4 || s[5].size != s[4].size ||
      s[6].size != s[4].size || s[12].size != 3 * polb || s[14].size != ((uint64_t)N + 6) * 64) {
    set_error("zkey: Invalid File format");
    return G16_E_FORMAT;
  }
  for (int k = 7; k <= 11; k++)
    if (s[k].size != polb) { set_error("zkey: Invalid File format"); return G16_E_FORMAT; }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { set_error("no HIP device: the PLONK prover has no CPU path"); return G16_E_NOGPU; }
  if (device < 0 || device >= ndev) { set_error("bad device ordinal"); return G16_E_ARG; }
  P->device = device;
  G16_HIP(hipSetDevice(device));
  int prio_lo = 0, prio_hi = 0;
  G16_HIP(hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));
  G16_HIP(hipStreamCreateWithPriority(&P->st, hipStreamNonBlocking, prio_hi));
  G16_HIP(hipStreamCreateWithPriority(&P->st2, hipStreamNonBlocking, 0));
  G16_HIP(hipEventCreateWithFlags(&P->ev_fork, hipEventDisableTiming));
  G16_HIP(hipEventCreateWithFlags(&P->ev_join, hipEventDisableTiming));
  hipStream_t st = P->st;
  if ((rc = ntt_tables_create(P->ntt_n, (int)P->L, st))) return rc;
  if ((rc = ntt_tables_create(P->ntt_4n, (int)P->L + 2, st))) return rc;
  {   // the round-3 coset: g = w_8N (g^4N = -1, so X^N - 1 has no zero on g <w_4N>), powers of g and 1/g by two-level tables
    const FrM g = h_root((int)P->L + 3), gi = fp_inv(g);
    P->gN = h_pow(g, N);
    P->gtab_hi = std::max<size_t>(1, ((size_t)N * 4) >> 12);
    std::vector<FrM> tab(2 * (4096 + P->gtab_hi));
    const FrM base[2] = {g, gi};
    for (int t = 0; t < 2; t++) {
      FrM* lo = tab.data() + t * (4096 + P->gtab_hi);
      FrM* hi = lo + 4096;
      lo[0] = hi[0] = fp_one<FrParams>();
      for (int k = 1; k < 4096; k++) lo[k] = fp_mul(lo[k - 1], base[t]);
      const FrM step = fp_mul(lo[4095], base[t]);
      for (size_t k = 1; k < P->gtab_hi; k++) hi[k] = fp_mul(hi[k - 1], step);
    }
    G16_HIP(hipMalloc(&P->d_gtab, tab.size() * 32));
    G16_HIP(hipMemcpy(P->d_gtab, tab.data(), tab.size() * 32, hipMemcpyHostToDevice));
  }
  // polynomials: coefficients (N) then evaluations (4N)
  // the selectors' 4N evaluations pass through d_T (allocated here, scratch of the proofs later) on their way to the
  // lazy format
  G16_HIP(hipMalloc(&P->d_T, (size_t)N * 128));
  FrM* d_stage = P->d_T;
  for (int k = 0; k < 8; k++) {
    const uint8_t* src = k < 5 ? s[7 + k].p : s[12].p + (size_t)(k - 5) * polb;
    G16_HIP(hipMalloc(&P->d_pol[k], (size_t)N * 32));
    G16_HIP(hipMalloc(&P->d_ext_l[k], (size_t)N * 4 * sizeof(F29)));
    G16_HIP(hipMemcpyAsync(P->d_pol[k], src, (size_t)N * 32, hipMemcpyHostToDevice, st));
    if (k >= 5) {   // sigma: round 2 reads the canonical evaluations at the N-domain points (every 4th of the file's 4N)
      G16_HIP(hipMalloc(&P->d_sig[k - 5], (size_t)N * 32));
      G16_HIP(hipMemcpyAsync(d_stage, src + (size_t)N * 32, (size_t)N * 128, hipMemcpyHostToDevice, st));
      k_stride4<<<nblk(N), 256, 0, st>>>(d_stage, N, P->d_sig[k - 5]);
    }
    // round 3 wants the polynomial on the coset g <w_4N>, not the zkey's subgroup evaluations: transformed here
    if ((rc = coset_fft(P, P->d_pol[k], N, P->d_ext_l[k], st))) return rc;
  }
  G16_HIP(hipStreamSynchronize(st));
  // maps, zero padded to N
  for (int c = 0; c < 3; c++) {
    G16_HIP(hipMalloc(&P->d_map[c], (size_t)N * 4));
    G16_HIP(hipMemsetAsync(P->d_map[c], 0, (size_t)N * 4, st));
    if (P->nCons) G16_HIP(hipMemcpyAsync(P->d_map[c], s[4 + c].p, (size_t)P->nCons * 4, hipMemcpyHostToDevice, st));
    for (uint32_t i = 0; i < P->nCons; i++) {
      uint32_t v;
      memcpy(&v, s[4 + c].p + (size_t)i * 4, 4);
      if (v >= P->nVars) { set_error("zkey: signal map out of range"); return G16_E_FORMAT; }
    }
  }
  // additions: split into arrays, ordered by dependency level
  {
    const uint32_t na = P->nAdd;
    std::vector<uint32_t> s1(na), s2(na), lvl(na), order(na);
    std::vector<FrM> f1(na), f2(na);
    uint32_t maxl = 0;
    for (uint32_t k = 0; k < na; k++) {
      const uint8_t* r = s[3].p + (size_t)k * 72;
      memcpy(&s1[k], r, 4);
      memcpy(&s2[k], r + 4, 4);
      memcpy(f1[k].v, r + 8, 32);
      memcpy(f2[k].v, r + 40, 32);
      if (s1[k] >= P->nBase + k || s2[k] >= P->nBase + k) { set_error("zkey: addition refers to a later signal"); return G16_E_FORMAT; }
      const uint32_t l1 = s1[k] >= P->nBase ? lvl[s1[k] - P->nBase] + 1 : 0, l2 = s2[k] >= P->nBase ? lvl[s2[k] - P->nBase] + 1 : 0;
      lvl[k] = l1 > l2 ? l1 : l2;
      if (lvl[k] > maxl) maxl = lvl[k];
    }
    std::vector<uint32_t> cnt(maxl + 2, 0);
    for (uint32_t k = 0; k < na; k++) cnt[lvl[k] + 1]++;
    for (uint32_t l = 0; l <= maxl; l++) cnt[l + 1] += cnt[l];
    P->level_start.assign(cnt.begin(), cnt.end());
    std::vector<uint32_t> cur(cnt.begin(), cnt.end() - 1);
    for (uint32_t k = 0; k < na; k++) order[cur[lvl[k]]++] = k;
    if (na) {
      G16_HIP(hipMalloc(&P->d_add_s1, (size_t)na * 4));
      G16_HIP(hipMalloc(&P->d_add_s2, (size_t)na * 4));
      G16_HIP(hipMalloc(&P->d_add_order, (size_t)na * 4));
      G16_HIP(hipMalloc(&P->d_add_f1, (size_t)na * 32));
      G16_HIP(hipMalloc(&P->d_add_f2, (size_t)na * 32));
      G16_HIP(hipMemcpy(P->d_add_s1, s1.data(), (size_t)na * 4, hipMemcpyHostToDevice));
      G16_HIP(hipMemcpy(P->d_add_s2, s2.data(), (size_t)na * 4, hipMemcpyHostToDevice));
      G16_HIP(hipMemcpy(P->d_add_order, order.data(), (size_t)na * 4, hipMemcpyHostToDevice));
      G16_HIP(hipMemcpy(P->d_add_f1, f1.data(), (size_t)na * 32, hipMemcpyHostToDevice));
      G16_HIP(hipMemcpy(P->d_add_f2, f2.data(), (size_t)na * 32, hipMemcpyHostToDevice));
    }
  }
  // scratch
  const size_t n4 = (size_t)N * 4;
  F29** four_l[] = {&P->d_A4, &P->d_B4, &P->d_C4, &P->d_Z4, &P->d_pi4, &P->d_om4, &P->d_l1, &P->d_lazy, &P->d_lazy2};
  for (F29** p : four_l) G16_HIP(hipMalloc(p, n4 * sizeof(F29)));
  FrM** one[] = {&P->d_A, &P->d_B, &P->d_C, &P->d_Z, &P->d_tmpN, &P->d_tmpN2, &P->d_tmpN3, &P->d_tmpN4,
                 &P->d_cA, &P->d_cB, &P->d_cC, &P->d_cZ, &P->d_pi_ev, &P->d_pi_co};
  for (FrM** p : one) G16_HIP(hipMalloc(p, (size_t)N * 32));
  FrM** plus[] = {&P->d_pa, &P->d_pb, &P->d_pc, &P->d_pz, &P->d_r, &P->d_wxi, &P->d_q};
  for (FrM** p : plus) G16_HIP(hipMalloc(p, ((size_t)N + 8) * 32));
  G16_HIP(hipMalloc(&P->d_wraw, (size_t)P->nBase * 32));
  G16_HIP(hipMalloc(&P->d_w, (size_t)P->nVars * 32));
  G16_HIP(hipMalloc(&P->d_scal, ((size_t)N + 8) * 32));
  const size_t tot_n = std::max<size_t>(n4 / kChunk + 8, 8 * ((3 * (size_t)N + 8) / kHorner + 2));
  G16_HIP(hipMalloc(&P->d_tot, tot_n * 32));
  G16_HIP(hipMalloc(&P->d_tot2, tot_n * 32));
  G16_HIP(hipMalloc(&P->d_evals, 8 * 32));
  G16_HIP(hipMalloc(&P->d_bad, 64));
  // the coset points g w_4N^i and L1 = iNTT(e_0) on them
  k_powers<<<nblk(nblk(n4, kChunk)), 256, 0, st>>>(h_root((int)P->L + 2), (uint32_t)n4, P->d_T);
  k_mul_const<<<nblk(n4), 256, 0, st>>>(P->d_T, n4, h_root((int)P->L + 3));
  k_to_lazy<<<nblk(n4), 256, 0, st>>>(P->d_T, P->d_om4, n4);
  {
    G16_HIP(hipMemsetAsync(P->d_tmpN, 0, (size_t)N * 32, st));
    const FrM one_m = fp_one<FrParams>();
    G16_HIP(hipMemcpyAsync(P->d_tmpN, &one_m, 32, hipMemcpyHostToDevice, st));
    if ((rc = do_ifft(P->ntt_n, P->d_tmpN, P->d_tmpN2, P->d_lazy, st))) return rc;
    if ((rc = coset_fft(P, P->d_tmpN2, N, P->d_l1, st))) return rc;
  }
  G16_HIP(hipGetLastError());
  G16_HIP(hipStreamSynchronize(st));
  // powers of tau: one dense base section
  MsmSectionIn sec;
  sec.bases_host = s[14].p;
  sec.n_total = N + 6;
  MsmConfig cfg;
  cfg.dense = true;
  if ((rc = msm_group_create(P->srs, &sec, 1, cfg))) return rc;
  if (P->srs.n != N + 6) { set_error("zkey: a power of tau is the point at infinity"); return G16_E_FORMAT; }
  if ((rc = msm_workspace_create(&P->ws, P->srs))) return rc;
  // three of the accumulate kernel's four wavefronts per SIMD (as fast: it is issue-bound), so that the short kernels
  // of the other commitments -- task queues, reduce tails -- find a slot while it runs
  const uint32_t acc_waves = getenv("G16_PLONK_ACC_WAVES") ? (uint32_t)atoi(getenv("G16_PLONK_ACC_WAVES")) : 3u;
  msm_set_waves(P->ws, acc_waves, 0);
  for (int k = 0; k < 2; k++) {
    if ((rc = msm_workspace_create(&P->wsx[k], P->srs))) return rc;
    msm_set_waves(P->wsx[k], acc_waves, 0);
    // (HIP maps streams onto 4 hardware queues per priority level, FIFO within a queue: the main stream and the first
    // side stream take the high-priority pool, the others the normal one -- r02 trace: slot 0 and slot 2 shared a queue)
    G16_HIP(hipStreamCreateWithPriority(&P->mst[k], hipStreamNonBlocking, k == 0 ? prio_hi : 0));
    G16_HIP(hipMalloc(&P->d_scalx[k], ((size_t)N + 8) * 32));
  }
  G16_HIP(hipEventCreateWithFlags(&P->ev_pol, hipEventDisableTiming));
  return G16_OK;
}

struct PlonkProofM {   // points affine Montgomery, evaluations Montgomery
  G1Affine A, B, C, Z, T1, T2, T3, Wxi, Wxiw;
  FrM ea, eb, ec, es1, es2, ezw, er;
};

// commitment of `len` Montgomery coefficients starting at d_coefs: sum coef_i [tau^i]
// Commitments: slot 0 runs on the main stream, slots 1 and 2 on their own streams behind an event of the main stream
// (the coefficients are written there), so that the independent commitments of a round overlap their latency-bound
// front ends and reduce tails with each other's bucket accumulation.
int commit_launch(g16_plonk* P, int slot, const FrM* d_coefs, uint32_t len) {
  const uint32_t np = P->N + 6;
  hipStream_t s = slot ? P->mst[slot - 1] : P->st;
  Fr* scal = slot ? P->d_scalx[slot - 1] : P->d_scal;
  if (slot) {
    G16_HIP(hipEventRecord(P->ev_pol, P->st));
    G16_HIP(hipStreamWaitEvent(s, P->ev_pol, 0));
  }
  k_from_mont_pad<<<nblk(np), 256, 0, s>>>(d_coefs, len, scal, np);
  G16_HIP(hipGetLastError());
  return msm_launch_front(P->srs, slot ? P->wsx[slot - 1] : P->ws, scal, s);   // digits -> sort; commit_lanes: the rest
}
int commit_lanes(g16_plonk* P, int slot) {
  hipStream_t s = slot ? P->mst[slot - 1] : P->st;
  return msm_launch_lanes(P->srs, slot ? P->wsx[slot - 1] : P->ws, s, s, nullptr, nullptr);
}
int commit_collect(g16_plonk* P, int slot, G1Affine* out) {
  MsmResult res;
  int rc = msm_collect(P->srs, slot ? P->wsx[slot - 1] : P->ws, &res);
  if (rc) return rc;
  xyzz_to_affine(*out, res.g1[0]);
  return G16_OK;
}
int commit(g16_plonk* P, const FrM* d_coefs, uint32_t len, G1Affine* out) {
  int rc = commit_launch(P, 0, d_coefs, len);
  if (!rc) rc = commit_lanes(P, 0);
  return rc ? rc : commit_collect(P, 0, out);
}
// three (or two) independent commitments at once
int commit3(g16_plonk* P, const FrM* const coefs[3], const uint32_t lens[3], G1Affine* const outs[3], int count) {
  int rc = G16_OK;
  // every front end (digits, sorts: memory- and latency-bound, they overlap each other) before any bucket accumulation
  // (it fills the chip for ~4 ms at 2^22 points and would hold the next commitment's front end back until it is done)
  for (int k = count - 1; k >= 0 && !rc; k--) rc = commit_launch(P, k, coefs[k], lens[k]);   // side slots first: their event precedes slot 0's kernels
  for (int k = count - 1; k >= 0 && !rc; k--) rc = commit_lanes(P, k);
  for (int k = 0; k < count; k++) {
    const int r = commit_collect(P, k, outs[k]);
    if (r && !rc) rc = r;
  }
  return rc;
}

// sum_i P[i] x^i over n device coefficients -> d_evals[slot]: chunked Horner, the chunk values combined by one
// workgroup (nothing leaves the device; read_evals fetches the slots of a round at once)
int eval_pol_async(g16_plonk* P, const FrM* d_pol, uint32_t n, const FrM& x, int slot) {
  const uint32_t nc = (n + kHorner - 1) / kHorner;
  k_horner<<<nblk(nc), 256, 0, P->st>>>(d_pol, n, x, P->d_tot);
  k_suffix_horner<<<1, 1024, 0, P->st>>>(P->d_tot, nc, h_pow(x, kHorner), P->d_tot2);
  k_copy1<<<1, 1, 0, P->st>>>(P->d_tot2, P->d_evals + slot);
  G16_HIP(hipGetLastError());
  return G16_OK;
}
int read_evals(g16_plonk* P, FrM* out, int count) {
  G16_HIP(hipMemcpyAsync(out, P->d_evals, (size_t)count * 32, hipMemcpyDeviceToHost, P->st));
  G16_HIP(hipStreamSynchronize(P->st));
  return G16_OK;
}

// res = P / (X - d) for a P with P(d) == 0 (n coefficients; res has n entries, the top one zero); d_bad[flag] is set
// when the remainder is not zero
int div_pol1(g16_plonk* P, const FrM* d_pol, uint32_t n, const FrM& d, FrM* d_res, int flag) {
  const uint32_t nc = (n + kHorner - 1) / kHorner;
  k_horner<<<nblk(nc), 256, 0, P->st>>>(d_pol, n, d, P->d_tot);
  // E[c] = sum_{j >= lo_c} P[j] d^(j - lo_c) = H[c] + d^256 E[c + 1]  (only the last chunk is short, and E beyond it is 0)
  k_suffix_horner<<<1, 1024, 0, P->st>>>(P->d_tot, nc, h_pow(d, kHorner), P->d_tot2);
  k_flag_nonzero<<<1, 1, 0, P->st>>>(P->d_tot2, P->d_bad + flag);
  k_divpol<<<nblk(nc), 256, 0, P->st>>>(d_pol, n, d, P->d_tot2, d_res);
  G16_HIP(hipGetLastError());
  return G16_OK;
}

int random_fr(FrM* out) {
  int fd = open("/dev/urandom", O_RDONLY);
  if (fd < 0) { set_error("cannot open /dev/urandom"); return G16_E_STATE; }
  Fr x;
  for (;;) {
    if (read(fd, x.v, 32) != 32) { close(fd); set_error("short read from /dev/urandom"); return G16_E_STATE; }
    x.v[7] &= 0x3fffffffu;
    if (h_lt_r(x.v)) break;
  }
  close(fd);
  *out = fp_to_mont(x);
  return G16_OK;
}

int plonk_prove_impl(g16_plonk* P, const uint8_t* wtns, size_t wlen, const uint8_t* blind, PlonkProofM* pr, uint8_t* pub) {
  Sec s[16];
  int rc = find_sections(wtns, wlen, "wtns", s, "wtns");
  if (rc) return rc;
  if (!s[1].p || s[1].size < 40 || !s[2].p) { set_error("wtns: Invalid File format"); return G16_E_FORMAT; }
  {
    static const uint32_t Rp[8] = G16_FR_P;
    uint32_t n8;
    memcpy(&n8, s[1].p, 4);
    if (n8 != 32 || memcmp(s[1].p + 4, Rp, 32) != 0) {
      set_error("Curve of the witness does not match the curve of the proving key");
      return G16_E_FORMAT;
    }
    uint32_t nw;
    memcpy(&nw, s[1].p + 36, 4);
    if (nw != P->nBase) {
      set_error("Invalid witness length. Circuit: " + std::to_string(P->nVars) + ", witness: " + std::to_string(nw) + ", " +
                std::to_string(P->nAdd));
      return G16_E_FORMAT;
    }
    if (s[2].size != (uint64_t)nw * 32) { set_error("wtns: Invalid File format"); return G16_E_FORMAT; }
  }
  for (uint32_t i = 0; i < P->nBase; i++) {
    uint32_t v[8];
    memcpy(v, s[2].p + (size_t)i * 32, 32);
    if (!h_lt_r(v)) { set_error("wtns: signal " + std::to_string(i) + " is not reduced modulo the scalar field"); return G16_E_FORMAT; }
  }
  if (pub && P->nPublic) memcpy(pub, s[2].p + 32, (size_t)P->nPublic * 32);
  FrM b[10];
  b[0] = fp_zero<FrParams>();
  for (int i = 1; i <= 9; i++) {
    if (blind) {
      Fr x;
      memcpy(x.v, blind + (size_t)(i - 1) * 32, 32);
      if (!h_lt_r(x.v)) { set_error("blinding scalar not below r"); return G16_E_ARG; }
      b[i] = fp_to_mont(x);
    } else if ((rc = random_fr(&b[i]))) {
      return rc;
    }
  }
  G16_HIP(hipSetDevice(P->device));
  hipStream_t st = P->st;
  const uint32_t N = P->N;
  const size_t n4 = (size_t)N * 4;
  auto tprev = std::chrono::steady_clock::now();
  const auto tstart = tprev;
  auto lap = [&](int k) {   // host wall time of a round (each ends in a commitment, which waits for the device)
    const auto now = std::chrono::steady_clock::now();
    P->last_ms[k] = std::chrono::duration<float, std::milli>(now - tprev).count();
    tprev = now;
  };
  // ---- witness: first element zeroed ("not used in plonk"), additions level by level, A/B/C
  G16_HIP(hipMemsetAsync(P->d_bad, 0, 32, st));
  G16_HIP(hipMemcpyAsync(P->d_wraw, s[2].p, (size_t)P->nBase * 32, hipMemcpyHostToDevice, st));
  k_to_mont<<<nblk(P->nBase), 256, 0, st>>>(P->d_wraw, P->d_w, P->nBase, 1u);
  for (size_t l = 0; l + 1 < P->level_start.size(); l++) {
    const uint32_t lo = P->level_start[l], hi = P->level_start[l + 1];
    if (hi > lo)
      k_additions<<<nblk(hi - lo), 256, 0, st>>>(P->d_add_order, lo, hi, P->d_add_s1, P->d_add_s2, P->d_add_f1, P->d_add_f2, P->d_w,
                                                 P->nBase);
  }
  FrM* ev[3] = {P->d_A, P->d_B, P->d_C};
  for (int c = 0; c < 3; c++) k_gather<<<nblk(N), 256, 0, st>>>(P->d_w, P->d_map[c], ev[c], N);
  G16_HIP(hipGetLastError());
  // ---- round 1
  // coefficients + blinded polynomial on the main stream; the 4N evaluations (needed in round 3 only) on the side
  // stream, beside the round's commitments
  hipStream_t st2 = P->st2;
  auto to_pol = [&](const FrM* evals, Pz pz, FrM* pol, FrM* coefs) -> int {
    int r = do_ifft(P->ntt_n, evals, coefs, P->d_lazy, st);
    if (r) return r;
    k_blind<<<nblk(N + pz.n), 256, 0, st>>>(coefs, N, pz, pol);
    G16_HIP(hipGetLastError());
    return G16_OK;
  };
  auto ext_of = [&](const FrM* coefs, uint32_t len, F29* ext) -> int {   // `len` coefficients -> the round-3 coset
    return coset_fft(P, coefs, len, ext, st2);   // the evaluations stay in the lazy format for round 3
  };
  Pz pza{{b[2], b[1], fp_zero<FrParams>()}, 2}, pzb{{b[4], b[3], fp_zero<FrParams>()}, 2}, pzc{{b[6], b[5], fp_zero<FrParams>()}, 2};
  if ((rc = to_pol(P->d_A, pza, P->d_pa, P->d_cA))) return rc;
  if ((rc = to_pol(P->d_B, pzb, P->d_pb, P->d_cB))) return rc;
  if ((rc = to_pol(P->d_C, pzc, P->d_pc, P->d_cC))) return rc;
  G16_HIP(hipEventRecord(P->ev_fork, st));
  G16_HIP(hipStreamWaitEvent(st2, P->ev_fork, 0));
  if ((rc = ext_of(P->d_pa, N + 2, P->d_A4))) return rc;   // the BLINDED polynomials (see k_round3)
  if ((rc = ext_of(P->d_pb, N + 2, P->d_B4))) return rc;
  if ((rc = ext_of(P->d_pc, N + 2, P->d_C4))) return rc;
  {   // the public-input polynomial on the 4N domain
    k_pi_evals<<<nblk(N), 256, 0, st2>>>(P->d_A, P->nPublic, N, P->d_pi_ev);
    if ((rc = do_ifft(P->ntt_n, P->d_pi_ev, P->d_pi_co, P->d_lazy2, st2))) return rc;
    if ((rc = ext_of(P->d_pi_co, N, P->d_pi4))) return rc;
  }
  {
    const FrM* cf[3] = {P->d_pa, P->d_pb, P->d_pc};
    const uint32_t ln[3] = {N + 2, N + 2, N + 2};
    G1Affine* o[3] = {&pr->A, &pr->B, &pr->C};
    if ((rc = commit3(P, cf, ln, o, 3))) return rc;
  }
  lap(0);
  // ---- round 2
  std::vector<uint8_t> tr;
  put_g1_be(tr, pr->A);
  put_g1_be(tr, pr->B);
  put_g1_be(tr, pr->C);
  const FrM beta = hash_to_fr(tr);
  tr.clear();
  put_fr_be(tr, beta);
  const FrM gamma = hash_to_fr(tr);
  {
    R2Args a{beta, gamma, P->k1, P->k2, P->w1, h_pow(P->w1, 64)};
    const uint32_t nc = (N + kZBlock - 1) / kZBlock;   // blocks = wavefronts
    // num -> d_tmpN (ratios), den -> d_tmpN2, pre -> d_tmpN3, lp -> d_tmpN4
    k_z_ratio<<<nblk((size_t)nc * 64), 256, 0, st>>>(P->d_A, P->d_B, P->d_C, P->d_sig[0], P->d_sig[1], P->d_sig[2], a, N, P->d_tmpN,
                                                    P->d_tmpN2, P->d_tmpN3);
    k_z_scan<<<nblk((size_t)nc * 64), 256, 0, st>>>(P->d_tmpN, N, P->d_tmpN4, P->d_tot);
    // (a 256-lane workgroup finds a compute unit beside the transforms of the side stream sooner than a 1 024-lane one)
    k_prefix_prod<<<1, nc > 4096 ? 1024 : 256, 0, st>>>(P->d_tot, nc, P->d_tot2, P->d_bad + 2);   // "Copy constraints does not match": read in round 3
    k_z_apply<<<nblk(N), 256, 0, st>>>(P->d_tmpN4, P->d_tot2, N, P->d_Z);
    G16_HIP(hipGetLastError());
  }
  Pz pzz{{b[9], b[8], b[7]}, 3};
  if ((rc = to_pol(P->d_Z, pzz, P->d_pz, P->d_cZ))) return rc;
  G16_HIP(hipEventRecord(P->ev_fork, st));
  G16_HIP(hipStreamWaitEvent(st2, P->ev_fork, 0));
  if ((rc = ext_of(P->d_pz, N + 3, P->d_Z4))) return rc;
  G16_HIP(hipEventRecord(P->ev_join, st2));
  if ((rc = commit(P, P->d_pz, N + 3, &pr->Z))) return rc;
  lap(1);
  // ---- round 3
  tr.clear();
  put_g1_be(tr, pr->Z);
  const FrM alpha = hash_to_fr(tr);
  {
    G16_HIP(hipStreamWaitEvent(st, P->ev_join, 0));   // A4, B4, C4, the public-input polynomial and Z4 are in place
    R3Args a;
    auto lz = [](const FrM& x) { return fr29_from_fr(x); };
    a.beta = lz(beta); a.gamma = lz(gamma); a.alpha = lz(alpha); a.alpha2 = lz(fp_sqr(alpha)); a.k1 = lz(P->k1); a.k2 = lz(P->k2);
    a.one = lz(fp_one<FrParams>());
    {   // x^N - 1 at x = g w_4N^i is g^N i4^(i mod 4) - 1
      const FrM i4 = h_root(2);
      FrM v = P->gN;
      for (int k = 0; k < 4; k++) {
        a.zhinv[k] = lz(fp_inv(fp_sub(v, fp_one<FrParams>())));
        v = fp_mul(v, i4);
      }
    }
    R3Ptrs q{P->d_A4, P->d_B4, P->d_C4, P->d_Z4, P->d_ext_l[0], P->d_ext_l[1], P->d_ext_l[2], P->d_ext_l[3], P->d_ext_l[4],
             P->d_ext_l[5], P->d_ext_l[6], P->d_ext_l[7], P->d_pi4, P->d_l1, P->d_om4};
    k_round3<<<nblk(n4), 256, 0, st>>>(q, a, (uint32_t)n4, P->d_lazy);   // t on the coset, natural order, lazy
    G16_HIP(hipGetLastError());
    {
      F29* v[1] = {P->d_lazy};
      if ((rc = ntt_dif_inverse(P->ntt_4n, v, 1, st))) return rc;
      Fr n4s = fp_zero<FrParams>();
      n4s.v[0] = (uint32_t)n4;
      const F29 ninv = fr29_from_fr(fp_inv(fp_to_mont(n4s)));
      k_coset_export_check<<<nblk(n4), 256, 0, st>>>(P->d_lazy, ninv, (int)P->L + 2, 3 * N + 6, P->tab_ginv(), P->d_T, P->d_bad);
      G16_HIP(hipGetLastError());
    }
    uint32_t bad[3] = {0, 0, 0};
    G16_HIP(hipMemcpyAsync(bad, P->d_bad, 12, hipMemcpyDeviceToHost, st));
    G16_HIP(hipStreamSynchronize(st));
    if (bad[2]) { set_error("Copy constraints does not match"); return G16_E_STATE; }
    if (bad[0]) { set_error("T Polynomial is not divisible"); return G16_E_STATE; }
  }
  {
    const FrM* cf[3] = {P->d_T, P->d_T + N, P->d_T + 2 * (size_t)N};
    const uint32_t ln[3] = {N, N, N + 6};
    G1Affine* o[3] = {&pr->T1, &pr->T2, &pr->T3};
    if ((rc = commit3(P, cf, ln, o, 3))) return rc;
  }
  lap(2);
  // ---- round 4
  tr.clear();
  put_g1_be(tr, pr->T1);
  put_g1_be(tr, pr->T2);
  put_g1_be(tr, pr->T3);
  const FrM xi = hash_to_fr(tr);
  FrM et;
  const FrM xiw = fp_mul(xi, P->w1);
  {
    EvalJobs jobs;
    const FrM xic = h_pow(xi, kHorner), xiwc = h_pow(xiw, kHorner);
    jobs.j[0] = EvalJob{P->d_pa, N + 2, xi, xic};
    jobs.j[1] = EvalJob{P->d_pb, N + 2, xi, xic};
    jobs.j[2] = EvalJob{P->d_pc, N + 2, xi, xic};
    jobs.j[3] = EvalJob{P->d_pol[5], N, xi, xic};
    jobs.j[4] = EvalJob{P->d_pol[6], N, xi, xic};
    jobs.j[5] = EvalJob{P->d_T, 3 * N + 6, xi, xic};
    jobs.j[6] = EvalJob{P->d_pz, N + 3, xiw, xiwc};
    jobs.j[7] = jobs.j[6];
    jobs.stride = (3 * N + 8) / kHorner + 2;
    k_horner_multi<<<dim3(nblk(jobs.stride), 7), 256, 0, st>>>(jobs, P->d_tot);
    k_suffix_total_multi<<<7, 1024, 0, st>>>(jobs, P->d_tot, P->d_evals);
    G16_HIP(hipGetLastError());
    FrM evs[7];
    if ((rc = read_evals(P, evs, 7))) return rc;
    pr->ea = evs[0]; pr->eb = evs[1]; pr->ec = evs[2]; pr->es1 = evs[3]; pr->es2 = evs[4]; et = evs[5]; pr->ezw = evs[6];
  }
  FrM xim = xi;
  for (uint32_t i = 0; i < P->L; i++) xim = fp_sqr(xim);
  {
    const FrM one = fp_one<FrParams>();
    const FrM bxi = fp_mul(beta, xi);
    FrM e2 = fp_mul(fp_mul(fp_add(fp_add(pr->ea, bxi), gamma), fp_add(fp_add(pr->eb, fp_mul(bxi, P->k1)), gamma)),
                    fp_add(fp_add(pr->ec, fp_mul(bxi, P->k2)), gamma));
    e2 = fp_mul(e2, alpha);
    FrM e3 = fp_mul(fp_add(fp_add(pr->ea, fp_mul(beta, pr->es1)), gamma), fp_add(fp_add(pr->eb, fp_mul(beta, pr->es2)), gamma));
    e3 = fp_mul(fp_mul(fp_mul(e3, beta), pr->ezw), alpha);
    const FrM l1 = fp_mul(fp_sub(xim, one), fp_inv(fp_mul(fp_sub(xi, one), h_from_u64(N))));
    const FrM e4 = fp_mul(l1, fp_sqr(alpha));
    R4Args a{fp_add(e2, e4), fp_mul(pr->ea, pr->eb), pr->ea, pr->eb, pr->ec, e3};
    k_pol_r<<<nblk(N + 3), 256, 0, st>>>(P->d_pz, P->d_pol[0], P->d_pol[1], P->d_pol[2], P->d_pol[3], P->d_pol[4], P->d_pol[7], a, N,
                                         P->d_r);
    G16_HIP(hipGetLastError());
  }
  if ((rc = eval_pol_async(P, P->d_r, N + 3, xi, 0))) return rc;
  if ((rc = read_evals(P, &pr->er, 1))) return rc;
  lap(3);
  // ---- round 5
  tr.clear();
  put_fr_be(tr, pr->ea); put_fr_be(tr, pr->eb); put_fr_be(tr, pr->ec); put_fr_be(tr, pr->es1); put_fr_be(tr, pr->es2);
  put_fr_be(tr, pr->ezw); put_fr_be(tr, pr->er);
  R5Args a5;
  a5.v[0] = fp_zero<FrParams>();
  a5.v[1] = hash_to_fr(tr);
  for (int i = 2; i <= 6; i++) a5.v[i] = fp_mul(a5.v[i - 1], a5.v[1]);
  a5.xim = xim;
  a5.xi2m = fp_sqr(xim);
  a5.sub0 = fp_add(et, fp_add(fp_mul(a5.v[1], pr->er),
                   fp_add(fp_mul(a5.v[2], pr->ea), fp_add(fp_mul(a5.v[3], pr->eb), fp_add(fp_mul(a5.v[4], pr->ec),
                   fp_add(fp_mul(a5.v[5], pr->es1), fp_mul(a5.v[6], pr->es2)))))));
  k_pol_wxi<<<nblk(N + 6), 256, 0, st>>>(P->d_T, P->d_r, P->d_pa, P->d_pb, P->d_pc, P->d_pol[5], P->d_pol[6], a5, N, P->d_wxi);
  G16_HIP(hipGetLastError());
  if ((rc = div_pol1(P, P->d_wxi, N + 6, xi, P->d_q, 3))) return rc;
  G16_HIP(hipMemcpyAsync(P->d_r, P->d_pz, ((size_t)N + 3) * 32, hipMemcpyDeviceToDevice, st));   // (r is spent)
  k_sub0<<<1, 1, 0, st>>>(P->d_r, pr->ezw);
  if ((rc = div_pol1(P, P->d_r, N + 3, xiw, P->d_wxi, 4))) return rc;
  {
    const FrM* cf[3] = {P->d_q, P->d_wxi, nullptr};
    const uint32_t ln[3] = {N + 6, N + 3, 0};
    G1Affine* o[3] = {&pr->Wxi, &pr->Wxiw, nullptr};
    if ((rc = commit3(P, cf, ln, o, 2))) return rc;
  }
  {
    uint32_t bad[2] = {0, 0};
    G16_HIP(hipMemcpyAsync(bad, P->d_bad + 3, 8, hipMemcpyDeviceToHost, st));
    G16_HIP(hipStreamSynchronize(st));
    if (bad[0] || bad[1]) { set_error("Polinomial does not divide"); return G16_E_STATE; }
  }
  lap(4);
  P->last_ms[5] = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - tstart).count();
  return G16_OK;
}

}  // namespace

// setup helper (synth.cpp::g16_plonk_setup): for each of 8 evaluation vectors (N Montgomery words on the host), the N
// coefficients and the 4N evaluations, written back to back at out[k] (5N words)
namespace g16 {
int plonk_setup_polys(int device, int L, const Fr* const evals[8], uint8_t* const out[8]) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { set_error("plonk setup: no HIP device (the transforms run on the GPU)"); return G16_E_NOGPU; }
  if (device < 0 || device >= ndev) { set_error("plonk setup: bad device ordinal"); return G16_E_ARG; }
  G16_HIP(hipSetDevice(device));
  const size_t N = (size_t)1 << L, n4 = N * 4;
  NttTables tn, t4;
  hipStream_t st = nullptr;
  FrM *d_ev = nullptr, *d_co = nullptr, *d_ext = nullptr;
  F29* d_lazy = nullptr;
  int rc = G16_OK;
  auto fail = [&](hipError_t e) {
    if (e == hipSuccess) return false;
    set_error(std::string("plonk setup: ") + hipGetErrorString(e));
    rc = G16_E_HIP;
    return true;
  };
  do {
    if (fail(hipStreamCreate(&st))) break;
    if ((rc = ntt_tables_create(tn, L, st))) break;
    if ((rc = ntt_tables_create(t4, L + 2, st))) break;
    if (fail(hipMalloc(&d_ev, N * 32)) || fail(hipMalloc(&d_co, N * 32)) || fail(hipMalloc(&d_ext, n4 * 32)) ||
        fail(hipMalloc(&d_lazy, n4 * sizeof(F29)))) break;
    for (int k = 0; k < 8 && rc == G16_OK; k++) {
      if (fail(hipMemcpyAsync(d_ev, evals[k], N * 32, hipMemcpyHostToDevice, st))) break;
      if ((rc = do_ifft(tn, d_ev, d_co, d_lazy, st))) break;
      k_pad4<<<nblk(n4), 256, 0, st>>>(d_co, (uint32_t)N, d_ext);
      if ((rc = do_fft(t4, d_ext, d_ext, d_lazy, st))) break;
      if (fail(hipGetLastError())) break;
      if (fail(hipMemcpyAsync(out[k], d_co, N * 32, hipMemcpyDeviceToHost, st))) break;
      if (fail(hipMemcpyAsync(out[k] + N * 32, d_ext, n4 * 32, hipMemcpyDeviceToHost, st))) break;
      if (fail(hipStreamSynchronize(st))) break;
    }
  } while (false);
  if (st) (void)hipStreamSynchronize(st);
  void* bufs[] = {d_ev, d_co, d_ext, d_lazy};
  for (void* p : bufs) if (p) (void)hipFree(p);
  ntt_tables_destroy(tn);
  ntt_tables_destroy(t4);
  if (st) (void)hipStreamDestroy(st);
  return rc;
}
}  // namespace g16

// setup with a real .ptau (synth.cpp::g16_plonk_setup_ptau): the eight commitments sum coef_i [tau^i]G1 by MSM over the
// first N powers; coefs[k] = N Montgomery words on the host, out = 8 affine Montgomery points
namespace g16 {
int plonk_setup_commit(int device, const uint8_t* tau_g1, uint32_t N, const uint8_t* const coefs[8], uint8_t* out) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { set_error("plonk setup: no HIP device"); return G16_E_NOGPU; }
  if (device < 0 || device >= ndev) { set_error("plonk setup: bad device ordinal"); return G16_E_ARG; }
  G16_HIP(hipSetDevice(device));
  MsmGroup grp;
  MsmWorkspace* ws = nullptr;
  MsmSectionIn sec;
  sec.bases_host = tau_g1;
  sec.n_total = N;
  MsmConfig cfg;
  cfg.dense = true;
  int rc = msm_group_create(grp, &sec, 1, cfg);
  if (!rc && grp.n != N) { set_error("ptau: a power of tau is the point at infinity"); rc = G16_E_FORMAT; }
  if (!rc) rc = msm_workspace_create(&ws, grp);
  hipStream_t st = nullptr;
  FrM* d_co = nullptr;
  Fr* d_sc = nullptr;
  if (!rc && (hipStreamCreate(&st) != hipSuccess || hipMalloc(&d_co, (size_t)N * 32) != hipSuccess ||
              hipMalloc(&d_sc, (size_t)N * 32) != hipSuccess)) { set_error("plonk setup: HIP allocation failed"); rc = G16_E_HIP; }
  for (int k = 0; k < 8 && !rc; k++) {
    if (hipMemcpyAsync(d_co, coefs[k], (size_t)N * 32, hipMemcpyHostToDevice, st) != hipSuccess) { set_error("plonk setup: upload failed"); rc = G16_E_HIP; break; }
    k_from_mont_pad<<<nblk(N), 256, 0, st>>>(d_co, N, d_sc, N);
    rc = msm_launch(grp, ws, d_sc, st, st);
    MsmResult res;
    if (!rc) rc = msm_collect(grp, ws, &res);
    if (!rc) {
      G1Affine a;
      xyzz_to_affine(a, res.g1[0]);
      memcpy(out + (size_t)k * 64, &a, 64);
    }
  }
  if (st) (void)hipStreamSynchronize(st);
  if (d_co) (void)hipFree(d_co);
  if (d_sc) (void)hipFree(d_sc);
  if (ws) msm_workspace_destroy(ws);
  msm_group_destroy(grp);
  if (st) (void)hipStreamDestroy(st);
  return rc;
}
}  // namespace g16

namespace {

void g1_std(uint8_t out[64], const G1Affine& p) {
  if (aff_is_inf(p)) { memset(out, 0, 64); return; }
  const Fq x = fp_from_mont(p.x), y = fp_from_mont(p.y);
  memcpy(out, x.v, 32);
  memcpy(out + 32, y.v, 32);
}

}  // namespace

extern "C" int g16_plonk_create(const uint8_t* zkey, size_t zkey_len, int device, g16_plonk** out) {
  if (!zkey || !out) { set_error("NULL argument"); return G16_E_ARG; }
  g16_plonk* P = new g16_plonk();
  const int rc = plonk_create_impl(zkey, zkey_len, device, P);
  if (rc) { delete P; return rc; }
  *out = P;
  return G16_OK;
}

extern "C" int g16_plonk_prove(g16_plonk* P, const uint8_t* wtns, size_t wtns_len, const uint8_t* blinding,
                               g16_plonk_proof* out, uint8_t* pub) {
  if (!P || !wtns || !out) { set_error("NULL argument"); return G16_E_ARG; }
  std::lock_guard<std::mutex> lk(P->mu);
  PlonkProofM pr;
  const int rc = plonk_prove_impl(P, wtns, wtns_len, blinding, &pr, pub);
  if (rc) return rc;
  g1_std(out->A, pr.A); g1_std(out->B, pr.B); g1_std(out->C, pr.C); g1_std(out->Z, pr.Z);
  g1_std(out->T1, pr.T1); g1_std(out->T2, pr.T2); g1_std(out->T3, pr.T3); g1_std(out->Wxi, pr.Wxi); g1_std(out->Wxiw, pr.Wxiw);
  const FrM* evs[7] = {&pr.ea, &pr.eb, &pr.ec, &pr.es1, &pr.es2, &pr.ezw, &pr.er};
  uint8_t* dst[7] = {out->eval_a, out->eval_b, out->eval_c, out->eval_s1, out->eval_s2, out->eval_zw, out->eval_r};
  for (int k = 0; k < 7; k++) {
    const Fr sdt = fp_from_mont(*evs[k]);
    memcpy(dst[k], sdt.v, 32);
  }
  return G16_OK;
}

extern "C" int g16_plonk_get_info(const g16_plonk* P, uint32_t info[6]) {
  if (!P || !info) { set_error("NULL argument"); return G16_E_ARG; }
  info[0] = P->nVars; info[1] = P->nPublic; info[2] = P->N; info[3] = P->nAdd; info[4] = P->nCons;
  info[5] = (uint32_t)(P->level_start.empty() ? 0 : P->level_start.size() - 1);
  return G16_OK;
}

extern "C" int g16_plonk_timings(const g16_plonk* P, float ms[6]) {
  if (!P || !ms) { set_error("NULL argument"); return G16_E_ARG; }
  for (int k = 0; k < 6; k++) ms[k] = P->last_ms[k];
  return G16_OK;
}

extern "C" void g16_plonk_destroy(g16_plonk* P) { delete P; }
