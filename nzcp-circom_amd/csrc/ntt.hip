// Radix-2 NTT over BN254 Fr for the Groth16 H-polynomial pipeline.
//
// Replaces ffjavascript 0.2.48 engine_fft.js / engine_applykey.js and wasmcurves 0.1.0
// build_fft.js / build_qap.js (pins /root/reference/yarn.lock:408-416, 1132-1138):
//   Fr.ifft  -> ntt_dif_inverse  (natural in, bit-reversed out, w^-1, unscaled)
//   Fr.batchApplyKey(x, 1, w_2N) + the 1/N of ifft -> ntt_coset_scale (one fused table multiply)
//   Fr.fft   -> ntt_dit_forward  (bit-reversed in, natural out)
//   qap_joinABC + frm_batchFromMontgomery -> ntt_join_abc
// DIF-then-DIT removes both bit-reversal permutations the reference performs.
//
// HBM layout: a vector is N contiguous 40-byte elements (9 x 29-bit limbs + pad).  One launch covers up
// to 9 butterfly stages: a workgroup stages a 512-element tile (20 KiB, dynamic LDS) and runs the stages
// with __syncthreads between them, then writes the tile back, so a 2^21 transform is 3 passes (9 + 7 + 5
// stages) over HBM instead of 21.  Tiles of later passes are 2^S rows x T columns with T >= 4
// contiguous elements (160 B runs) for coalescing.  Index math validated by the Python model in
// tests/test_cpu_ntt_plan.py.
//
// Arithmetic: Fr in the 9 x 29-bit lazy format (fr29.cuh; 40-byte elements in HBM and LDS).  A
// butterfly is one 162-mad product + a limb-wise add and sub; values drift upwards between
// products, so every element is weak-reduced on load and after every 3rd DIF stage (sum branch
// doubles: 1 -> 2 -> 4 -> 8r, subtraction offsets K = 2, 3, 5) or after the 7th DIT stage (+2r per
// stage), keeping every product input below 16r.
#include <stdlib.h>

#include "fr29.cuh"
#include "internal.h"

namespace g16 {

static constexpr int kTileLogCap = 11;   // 2048 elements x 40 B = 80 KiB of LDS at most
static constexpr int kThreads = 256;
// tile size / minimum contiguous run, tunable for sweeps: G16_NTT_TILE_LOG (<= 11), G16_NTT_MIN_TB
static int env_int(const char* name, int dflt) {
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
}

struct VecPtrs { F29* p[4]; };

// Issue priority of the NTT wavefronts (s_setprio), G16_CHAIN_PRIO = 0..3 at create: the QAP -> NTT chain is the head of a
// proof's critical path and shares the chip with the witness group's bucket accumulation (prio 0, a persistent grid).
__device__ int g_ntt_prio = 0;
__device__ __forceinline__ void ntt_set_prio() {
  const int p = g_ntt_prio;
  if (p == 1) __builtin_amdgcn_s_setprio(1);
  else if (p == 2) __builtin_amdgcn_s_setprio(2);
  else if (p >= 3) __builtin_amdgcn_s_setprio(3);
}

__device__ __forceinline__ uint32_t bitrev_dev(uint32_t x, int bits) {
  return bits == 0 ? 0u : (__brev(x) >> (32 - bits));
}

// The butterfly stages of one pass on a tile staged in LDS (shared by the plain pass kernel and the fused last
// forward pass).  gidx maps a tile element to its index in the vector.
template <class GIdx>
__device__ __forceinline__ void ntt_tile_stages(F29* __restrict__ tile, const F29* __restrict__ tw, const GIdx& gidx,
                                                uint32_t tile_n, int L, int lo_bits, int S, int tb, int dif) {
  for (int k = 0; k < S; k++) {
    const int st = dif ? (S - 1 - k) : k;
    const int bit = tb + st;
    const int beta = lo_bits + st;  // global index bit this stage acts on
    const size_t hmask = ((size_t)1 << beta) - 1;
    const int sh = L - 1 - beta;
    const int phase = k % 3;        // DIF: stages since the last weak reduction
    for (uint32_t bf = threadIdx.x; bf < (tile_n >> 1); bf += kThreads) {
      const uint32_t e0 = ((bf >> bit) << (bit + 1)) | (bf & ((1u << bit) - 1));
      const uint32_t e1 = e0 | (1u << bit);
      const F29 u = tile[e0], v = tile[e1];
      if (beta == 0) {
        // the stage on global index bit 0: every twiddle is w^0 = 1, no product (1/21 of all butterflies at 2^21).
        // DIF: it is the last stage, d < 16r goes to the store / the coset product; DIT: it is the first stage,
        // v was weak-reduced on load (< 1.0001 r), so sub<2> covers it.
        tile[e0] = fr29_add(u, v);
        tile[e1] = dif ? (phase == 0 ? fr29_sub<2>(u, v) : (phase == 1 ? fr29_sub<3>(u, v) : fr29_sub<5>(u, v)))
                       : fr29_sub<2>(u, v);
        continue;
      }
      const F29 w = tw[(gidx(e0) & hmask) << sh];
      if (dif) {
        tile[e0] = fr29_add(u, v);
        const F29 d = phase == 0 ? fr29_sub<2>(u, v) : (phase == 1 ? fr29_sub<3>(u, v) : fr29_sub<5>(u, v));
        tile[e1] = fr29_mul(d, w);
      } else {
        const F29 tv = fr29_mul(v, w);
        tile[e0] = fr29_add(u, tv);
        tile[e1] = fr29_sub<2>(u, tv);
      }
    }
    __syncthreads();
    const bool reduce_now = dif ? (phase == 2 && k + 1 < S) : (k == 6 && S > 7);
    if (reduce_now) {
      for (uint32_t e = threadIdx.x; e < tile_n; e += kThreads) tile[e] = fr29_weak_reduce(tile[e]);
      __syncthreads();
    }
  }
}

// `post` (optional): table multiplied into every element on store (the coset/1-over-N table after the
// last inverse pass), saving one sweep over the vectors.
__global__ __launch_bounds__(kThreads) void ntt_pass_kernel(VecPtrs vecs, const F29* __restrict__ tw,
                                                            int L, int tile_log, int lo_bits, int S,
                                                            int tb, int dif, const F29* __restrict__ post) {
  ntt_set_prio();
  extern __shared__ __align__(16) unsigned char ntt_lds[];
  F29* tile = reinterpret_cast<F29*>(ntt_lds);
  F29* __restrict__ x = vecs.p[blockIdx.y];
  const uint32_t tile_n = 1u << tile_log;
  const uint32_t T = 1u << tb;
  const uint32_t t = blockIdx.x;
  size_t base;
  if (lo_bits == 0) {
    base = (size_t)t << tile_log;
  } else {
    const uint32_t per = (1u << lo_bits) >> tb;  // lo blocks per hi group
    const uint32_t lo_blk = t % per, hi = t / per;
    base = ((size_t)hi << (lo_bits + S)) + (size_t)lo_blk * T;
  }
  auto gidx = [&](uint32_t e) -> size_t {
    return base + ((size_t)(e >> tb) << lo_bits) + (e & (T - 1));
  };
  for (uint32_t e = threadIdx.x; e < tile_n; e += kThreads) tile[e] = fr29_weak_reduce(x[gidx(e)]);
  __syncthreads();
  ntt_tile_stages(tile, tw, gidx, tile_n, L, lo_bits, S, tb, dif);
  if (post) {
    for (uint32_t e = threadIdx.x; e < tile_n; e += kThreads) {
      const size_t g = gidx(e);
      x[g] = fr29_mul(tile[e], post[g]);
    }
  } else {
    for (uint32_t e = threadIdx.x; e < tile_n; e += kThreads) x[gidx(e)] = tile[e];
  }
}

// The LAST inverse (DIF) pass and the FIRST forward (DIT) pass act on the same tiles -- the contiguous 2^tile_log
// elements, index bits [0, tile_log) -- with the coset table between them: one kernel runs the inverse stages, the
// table product and the forward stages on the tile in LDS, so the half-way vectors (coefficients x coset factors, in
// bit-reversed order) never go to HBM: one read + one write of every vector less per odd-coset evaluation.
__global__ __launch_bounds__(kThreads) void ntt_mid_pass_kernel(VecPtrs vecs, const F29* __restrict__ tw_inv,
                                                                const F29* __restrict__ tw_fwd, int L, int tile_log,
                                                                const F29* __restrict__ post) {
  ntt_set_prio();
  extern __shared__ __align__(16) unsigned char ntt_lds[];
  F29* tile = reinterpret_cast<F29*>(ntt_lds);
  F29* __restrict__ x = vecs.p[blockIdx.y];
  const uint32_t tile_n = 1u << tile_log;
  const size_t base = (size_t)blockIdx.x << tile_log;
  auto gidx = [&](uint32_t e) -> size_t { return base + e; };
  for (uint32_t e = threadIdx.x; e < tile_n; e += kThreads) tile[e] = fr29_weak_reduce(x[base + e]);
  __syncthreads();
  ntt_tile_stages(tile, tw_inv, gidx, tile_n, L, 0, tile_log, 0, 1);
  // (< 16 r after the inverse stages; the product with the table entry is < 1.1 r: what the forward stages expect)
  for (uint32_t e = threadIdx.x; e < tile_n; e += kThreads) tile[e] = fr29_mul(tile[e], post[base + e]);
  __syncthreads();
  ntt_tile_stages(tile, tw_fwd, gidx, tile_n, L, 0, tile_log, 0, 0);
  for (uint32_t e = threadIdx.x; e < tile_n; e += kThreads) x[base + e] = tile[e];
}

// The LAST forward (DIT) pass of the three vectors A, B, C fused with qap_joinABC + batchFromMontgomery: a
// workgroup runs the pass on the same tile of A, then B, then C (one 20 KiB LDS tile, reused), keeps its own
// elements in registers and writes only P[g] = plain(A'[g] B'[g] - C'[g]): the three transformed vectors never
// go back to HBM (2 x 252 MB less traffic at N = 2^21, and the join kernel disappears).
template <int EPT>   // tile elements per thread = max(1, tile_n / 256)
__global__ __launch_bounds__(kThreads) void ntt_last_pass_join_kernel(VecPtrs vecs, const F29* __restrict__ tw, int L,
                                                                      int tile_log, int lo_bits, int S, int tb,
                                                                      Fr* __restrict__ p_out) {
  ntt_set_prio();
  extern __shared__ __align__(16) unsigned char ntt_lds[];
  F29* tile = reinterpret_cast<F29*>(ntt_lds);
  const uint32_t tile_n = 1u << tile_log;
  const uint32_t T = 1u << tb;
  const uint32_t t = blockIdx.x;
  size_t base;
  if (lo_bits == 0) {
    base = (size_t)t << tile_log;
  } else {
    const uint32_t per = (1u << lo_bits) >> tb;
    const uint32_t lo_blk = t % per, hi = t / per;
    base = ((size_t)hi << (lo_bits + S)) + (size_t)lo_blk * T;
  }
  auto gidx = [&](uint32_t e) -> size_t {
    return base + ((size_t)(e >> tb) << lo_bits) + (e & (T - 1));
  };
  F29 acc[EPT];   // a'[e], then a'[e] b'[e]
#pragma unroll 1
  for (int v = 0; v < 3; v++) {
    const F29* __restrict__ x = vecs.p[v];
    for (uint32_t e = threadIdx.x; e < tile_n; e += kThreads) tile[e] = fr29_weak_reduce(x[gidx(e)]);
    __syncthreads();
    ntt_tile_stages(tile, tw, gidx, tile_n, L, lo_bits, S, tb, 0);
#pragma unroll
    for (int k = 0; k < EPT; k++) {
      const uint32_t e = threadIdx.x + (uint32_t)k * kThreads;
      if (e < tile_n) {
        if (v == 0) acc[k] = tile[e];
        else if (v == 1) acc[k] = fr29_mul(acc[k], tile[e]);
        else p_out[gidx(e)] = fr29_to_plain(fr29_sub<2>(acc[k], fr29_weak_reduce(tile[e])));
      }
    }
    __syncthreads();
  }
}

// tw[i] = w^i
__global__ void ntt_pow_table_kernel(F29* out, F29 w, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = fr29_pow_u64(w, (uint64_t)i);
}
// coset[j] = ninv * inc^bitrev(j)
__global__ void ntt_coset_table_kernel(F29* out, F29 inc, F29 ninv, int L) {
  size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j < ((size_t)1 << L)) out[j] = fr29_mul(ninv, fr29_pow_u64(inc, (uint64_t)bitrev_dev((uint32_t)j, L)));
}
__global__ void ntt_scale_kernel(VecPtrs vecs, const F29* __restrict__ tab, size_t n) {
  size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j < n) {
    F29* x = vecs.p[blockIdx.y];
    x[j] = fr29_mul(x[j], tab[j]);
  }
}
// operator-level API glue: canonical Montgomery(2^256) image <-> lazy format, with the bit reversal
__global__ void ntt_import_bitrev_kernel(const Fr* __restrict__ in, F29* __restrict__ out, int bitrev, int L) {
  size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j < ((size_t)1 << L)) out[bitrev ? bitrev_dev((uint32_t)j, L) : j] = fr29_from_fr(in[j]);
}
__global__ void ntt_export_bitrev_kernel(const F29* __restrict__ in, Fr* __restrict__ out, F29 scale, int do_scale,
                                         int bitrev, int L) {
  size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j < ((size_t)1 << L)) {
    F29 v = in[j];
    if (do_scale) v = fr29_mul(v, scale);
    out[bitrev ? bitrev_dev((uint32_t)j, L) : j] = fr29_to_fr(v);
  }
}
// P[i] = a[i]*b[i] - c[i] as a plain (standard-form) integer: the H-MSM scalar
__global__ void ntt_join_kernel(const F29* __restrict__ a, const F29* __restrict__ b,
                                const F29* __restrict__ c, Fr* __restrict__ p, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = fr29_to_plain(fr29_sub<2>(fr29_mul(a[i], b[i]), fr29_weak_reduce(c[i])));
}

static Fr host_root(int L) {  // Fr.w[L] in Montgomery form
  Fr w = {G16_FR_W28};
  for (int i = 28; i > L; i--) w = fp_sqr(w);
  return w;
}
static Fr host_from_u64(uint64_t v) {
  Fr a = fp_zero<FrParams>();
  a.v[0] = (uint32_t)v;
  a.v[1] = (uint32_t)(v >> 32);
  return fp_to_mont(a);
}

int ntt_tables_create(NttTables& t, int L, hipStream_t st) {
  if (L < 0 || L > 27) { set_error("domainSize out of range (need 2^0..2^27)"); return G16_E_ARG; }
  t.L = L;
  {
    const int prio = env_int("G16_CHAIN_PRIO", 0);
    G16_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_ntt_prio), &prio, sizeof(int)));
  }
  // r01 sweep on MI355X (N = 2^21, three vectors): tile 2^9 = 1.64 ms, 2^10 = 1.84, 2^11 = 2.1, 2^8 = 2.1:
  // a 20 KiB tile keeps 8 workgroups resident per CU, which matters more than saving a pass.
  int tile_max = env_int("G16_NTT_TILE_LOG", 9);
  if (tile_max > kTileLogCap) tile_max = kTileLogCap;
  if (tile_max < 3) tile_max = 3;
  int min_tb = env_int("G16_NTT_MIN_TB", 2);
  if (min_tb < 0) min_tb = 0;
  if (min_tb > tile_max - 1) min_tb = tile_max - 1;
  t.tile_log = L < tile_max ? L : tile_max;
  t.passes.clear();
  t.passes.push_back({0, t.tile_log, 0});
  for (int done = t.tile_log; done < L;) {
    int S = L - done;
    if (S > t.tile_log - min_tb) S = t.tile_log - min_tb;
    t.passes.push_back({done, S, t.tile_log - S});
    done += S;
  }
  G16_HIP(hipFuncSetAttribute((const void*)ntt_pass_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)(sizeof(F29) << kTileLogCap)));
  G16_HIP(hipFuncSetAttribute((const void*)ntt_mid_pass_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)(sizeof(F29) << kTileLogCap)));
  G16_HIP(hipFuncSetAttribute((const void*)ntt_last_pass_join_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)(sizeof(F29) << kTileLogCap)));
  G16_HIP(hipFuncSetAttribute((const void*)ntt_last_pass_join_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)(sizeof(F29) << kTileLogCap)));
  const size_t N = (size_t)1 << L, half = N > 1 ? N / 2 : 1;
  G16_HIP(hipMalloc(&t.tw_fwd, half * sizeof(F29)));
  G16_HIP(hipMalloc(&t.tw_inv, half * sizeof(F29)));
  G16_HIP(hipMalloc(&t.coset, N * sizeof(F29)));
  const Fr wc = host_root(L);
  const F29 w = fr29_from_fr(wc), winv = fr29_from_fr(fp_inv(wc));
  const F29 inc = fr29_from_fr(host_root(L + 1));  // w_2N (L <= 27 so L+1 <= 28)
  const F29 ninv = fr29_from_fr(fp_inv(host_from_u64(N)));
  const int bs = 256;
  ntt_pow_table_kernel<<<(unsigned)((half + bs - 1) / bs), bs, 0, st>>>(t.tw_fwd, w, half);
  ntt_pow_table_kernel<<<(unsigned)((half + bs - 1) / bs), bs, 0, st>>>(t.tw_inv, winv, half);
  ntt_coset_table_kernel<<<(unsigned)((N + bs - 1) / bs), bs, 0, st>>>(t.coset, inc, ninv, L);
  G16_HIP(hipGetLastError());
  return G16_OK;
}

void ntt_tables_destroy(NttTables& t) {
  if (t.tw_fwd) (void)hipFree(t.tw_fwd);
  if (t.tw_inv) (void)hipFree(t.tw_inv);
  if (t.coset) (void)hipFree(t.coset);
  t.tw_fwd = t.tw_inv = t.coset = nullptr;
  t.L = -1;
}

static int run_passes(const NttTables& t, F29* const* vecs, int nvec, bool dif, const F29* post_last, hipStream_t st) {
  if (nvec < 1 || nvec > 4) { set_error("ntt: nvec must be 1..4"); return G16_E_ARG; }
  if (t.L == 0) return G16_OK;
  VecPtrs vp{};
  for (int i = 0; i < nvec; i++) vp.p[i] = vecs[i];
  const unsigned ntiles = 1u << (t.L - t.tile_log);
  const int np = (int)t.passes.size();
  for (int k = 0; k < np; k++) {
    const NttPass& p = t.passes[dif ? np - 1 - k : k];
    ntt_pass_kernel<<<dim3(ntiles, nvec), kThreads, sizeof(F29) << t.tile_log, st>>>(
        vp, dif ? t.tw_inv : t.tw_fwd, t.L, t.tile_log, p.lo_bits, p.S, p.tb, dif ? 1 : 0,
        k == np - 1 ? post_last : nullptr);
  }
  G16_HIP(hipGetLastError());
  return G16_OK;
}

int ntt_dif_inverse(const NttTables& t, F29* const* vecs, int nvec, hipStream_t st) {
  return run_passes(t, vecs, nvec, true, nullptr, st);
}
// inverse transform with the coset table (1/N * w_2N^i, bit-reversed order) fused into the last pass
int ntt_dif_inverse_coset(const NttTables& t, F29* const* vecs, int nvec, hipStream_t st) {
  if (t.L == 0) return ntt_coset_scale(t, vecs, nvec, st);
  return run_passes(t, vecs, nvec, true, t.coset, st);
}
int ntt_dit_forward(const NttTables& t, F29* const* vecs, int nvec, hipStream_t st) {
  return run_passes(t, vecs, nvec, false, nullptr, st);
}

// Forward transform of (a, b, c) with the join fused into the last pass: p_std[i] = plain(a'[i] b'[i] - c'[i]).
// The vectors are left half-transformed (their last pass is never written back).
int ntt_dit_forward_join(const NttTables& t, F29* a, F29* b, F29* c, Fr* p_std, hipStream_t st) {
  if (t.L == 0) return ntt_join_abc(a, b, c, p_std, 1, st);
  if (t.tile_log > 11) { set_error("ntt: tile too large for the fused join"); return G16_E_ARG; }
  VecPtrs vp{};
  vp.p[0] = a; vp.p[1] = b; vp.p[2] = c;
  const unsigned ntiles = 1u << (t.L - t.tile_log);
  const int np = (int)t.passes.size();
  for (int k = 0; k + 1 < np; k++) {
    const NttPass& p = t.passes[k];
    ntt_pass_kernel<<<dim3(ntiles, 3), kThreads, sizeof(F29) << t.tile_log, st>>>(vp, t.tw_fwd, t.L, t.tile_log, p.lo_bits,
                                                                                   p.S, p.tb, 0, nullptr);
  }
  const NttPass& p = t.passes[np - 1];
  const size_t lds = sizeof(F29) << t.tile_log;
#define G16_JOIN_LAUNCH(EPT) \
  ntt_last_pass_join_kernel<EPT><<<ntiles, kThreads, lds, st>>>(vp, t.tw_fwd, t.L, t.tile_log, p.lo_bits, p.S, p.tb, p_std)
  if (t.tile_log <= 8) G16_JOIN_LAUNCH(1);
  else if (t.tile_log == 9) G16_JOIN_LAUNCH(2);
  else if (t.tile_log == 10) G16_JOIN_LAUNCH(4);
  else G16_JOIN_LAUNCH(8);
#undef G16_JOIN_LAUNCH
  G16_HIP(hipGetLastError());
  return G16_OK;
}

// Odd-coset evaluation of `nvec` vectors in place: inverse transform, coset table, forward transform, with the
// middle two passes fused (ntt_mid_pass_kernel).  join_p != nullptr (three vectors a, b, c): the last forward pass
// joins as well and writes join_p[i] = plain(a'[i] b'[i] - c'[i]) (the vectors are then left half-transformed).
int ntt_coset_roundtrip(const NttTables& t, F29* const* vecs, int nvec, Fr* join_p, hipStream_t st) {
  if (nvec < 1 || nvec > 4 || (join_p && nvec != 3)) { set_error("ntt: bad vector count"); return G16_E_ARG; }
  const int np = (int)t.passes.size();
  if (t.L == 0 || np < 2 || t.passes[0].lo_bits != 0 || t.passes[0].S != t.tile_log) {   // tiny domains: unfused
    int rc = ntt_dif_inverse_coset(t, vecs, nvec, st);
    if (rc) return rc;
    return join_p ? ntt_dit_forward_join(t, vecs[0], vecs[1], vecs[2], join_p, st) : ntt_dit_forward(t, vecs, nvec, st);
  }
  if (join_p && t.tile_log > 11) { set_error("ntt: tile too large for the fused join"); return G16_E_ARG; }
  VecPtrs vp{};
  for (int i = 0; i < nvec; i++) vp.p[i] = vecs[i];
  const unsigned ntiles = 1u << (t.L - t.tile_log);
  const size_t lds = sizeof(F29) << t.tile_log;
  for (int k = np - 1; k >= 1; k--) {   // inverse passes on the high bits
    const NttPass& p = t.passes[k];
    ntt_pass_kernel<<<dim3(ntiles, nvec), kThreads, lds, st>>>(vp, t.tw_inv, t.L, t.tile_log, p.lo_bits, p.S, p.tb, 1, nullptr);
  }
  ntt_mid_pass_kernel<<<dim3(ntiles, nvec), kThreads, lds, st>>>(vp, t.tw_inv, t.tw_fwd, t.L, t.tile_log, t.coset);
  const int last_plain = join_p ? np - 2 : np - 1;
  for (int k = 1; k <= last_plain; k++) {   // forward passes on the high bits
    const NttPass& p = t.passes[k];
    ntt_pass_kernel<<<dim3(ntiles, nvec), kThreads, lds, st>>>(vp, t.tw_fwd, t.L, t.tile_log, p.lo_bits, p.S, p.tb, 0, nullptr);
  }
  if (join_p) {
    const NttPass& p = t.passes[np - 1];
#define G16_JOIN_LAUNCH(EPT) \
  ntt_last_pass_join_kernel<EPT><<<ntiles, kThreads, lds, st>>>(vp, t.tw_fwd, t.L, t.tile_log, p.lo_bits, p.S, p.tb, join_p)
    if (t.tile_log <= 8) G16_JOIN_LAUNCH(1);
    else if (t.tile_log == 9) G16_JOIN_LAUNCH(2);
    else if (t.tile_log == 10) G16_JOIN_LAUNCH(4);
    else G16_JOIN_LAUNCH(8);
#undef G16_JOIN_LAUNCH
  }
  G16_HIP(hipGetLastError());
  return G16_OK;
}

int ntt_coset_scale(const NttTables& t, F29* const* vecs, int nvec, hipStream_t st) {
  if (nvec < 1 || nvec > 4) { set_error("ntt: nvec must be 1..4"); return G16_E_ARG; }
  VecPtrs vp{};
  for (int i = 0; i < nvec; i++) vp.p[i] = vecs[i];
  const size_t N = (size_t)1 << t.L;
  ntt_scale_kernel<<<dim3((unsigned)((N + 255) / 256), nvec), 256, 0, st>>>(vp, t.coset, N);
  G16_HIP(hipGetLastError());
  return G16_OK;
}

int ntt_import(const NttTables& t, const Fr* in, F29* out, bool bitrev, hipStream_t st) {
  const size_t N = (size_t)1 << t.L;
  ntt_import_bitrev_kernel<<<(unsigned)((N + 255) / 256), 256, 0, st>>>(in, out, bitrev ? 1 : 0, t.L);
  G16_HIP(hipGetLastError());
  return G16_OK;
}
int ntt_export(const NttTables& t, const F29* in, Fr* out, bool bitrev, bool scale_ninv, hipStream_t st) {
  const size_t N = (size_t)1 << t.L;
  const F29 ninv = fr29_from_fr(fp_inv(host_from_u64(N)));
  ntt_export_bitrev_kernel<<<(unsigned)((N + 255) / 256), 256, 0, st>>>(in, out, ninv, scale_ninv ? 1 : 0,
                                                                        bitrev ? 1 : 0, t.L);
  G16_HIP(hipGetLastError());
  return G16_OK;
}

int ntt_join_abc(const F29* a, const F29* b, const F29* c, Fr* p_std, size_t n, hipStream_t st) {
  ntt_join_kernel<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(a, b, c, p_std, n);
  G16_HIP(hipGetLastError());
  return G16_OK;
}

}  // namespace g16
