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
// HBM layout: a vector is N contiguous 32-byte Montgomery residues.  One launch covers up to
// 10 butterfly stages: a workgroup stages a 1024-element tile (32 KiB) in LDS, runs the stages
// with __syncthreads between them and writes the tile back, so a 2^21 transform is 3 passes
// over HBM instead of 21.  Tiles of later passes are 2^S rows x T columns with T >= 4
// contiguous elements (128 B runs) for coalescing.  Index math validated by the Python model in
// tests/test_ntt_plan.py.
#include "internal.h"

namespace g16 {

static constexpr int kTileLogMax = 10;
static constexpr int kMinTb = 2;
static constexpr int kThreads = 256;

struct VecPtrs { Fr* p[4]; };

__device__ __forceinline__ uint32_t bitrev_dev(uint32_t x, int bits) {
  return bits == 0 ? 0u : (__brev(x) >> (32 - bits));
}

__global__ __launch_bounds__(kThreads) void ntt_pass_kernel(VecPtrs vecs, const Fr* __restrict__ tw,
                                                            int L, int tile_log, int lo_bits, int S,
                                                            int tb, int dif) {
  __shared__ Fr tile[1 << kTileLogMax];
  Fr* __restrict__ x = vecs.p[blockIdx.y];
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
  for (uint32_t e = threadIdx.x; e < tile_n; e += kThreads) tile[e] = x[gidx(e)];
  __syncthreads();
  for (int k = 0; k < S; k++) {
    const int st = dif ? (S - 1 - k) : k;
    const int bit = tb + st;
    const int beta = lo_bits + st;  // global index bit this stage acts on
    const size_t hmask = ((size_t)1 << beta) - 1;
    const int sh = L - 1 - beta;
    for (uint32_t bf = threadIdx.x; bf < (tile_n >> 1); bf += kThreads) {
      const uint32_t e0 = ((bf >> bit) << (bit + 1)) | (bf & ((1u << bit) - 1));
      const uint32_t e1 = e0 | (1u << bit);
      const Fr w = tw[(gidx(e0) & hmask) << sh];
      Fr u = tile[e0], v = tile[e1];
      if (dif) {
        tile[e0] = fp_add(u, v);
        tile[e1] = fp_mul(fp_sub(u, v), w);
      } else {
        v = fp_mul(v, w);
        tile[e0] = fp_add(u, v);
        tile[e1] = fp_sub(u, v);
      }
    }
    __syncthreads();
  }
  for (uint32_t e = threadIdx.x; e < tile_n; e += kThreads) x[gidx(e)] = tile[e];
}

// tw[i] = w^i
__global__ void ntt_pow_table_kernel(Fr* out, Fr w, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = fp_pow_u64(w, (uint64_t)i);
}
// coset[j] = ninv * inc^bitrev(j)
__global__ void ntt_coset_table_kernel(Fr* out, Fr inc, Fr ninv, int L) {
  size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j < ((size_t)1 << L)) out[j] = fp_mul(ninv, fp_pow_u64(inc, (uint64_t)bitrev_dev((uint32_t)j, L)));
}
__global__ void ntt_scale_kernel(VecPtrs vecs, const Fr* __restrict__ tab, size_t n) {
  size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j < n) {
    Fr* x = vecs.p[blockIdx.y];
    x[j] = fp_mul(x[j], tab[j]);
  }
}
__global__ void ntt_bitrev_kernel(const Fr* __restrict__ in, Fr* __restrict__ out, Fr ninv, int scale,
                                  int L) {
  size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j < ((size_t)1 << L)) {
    Fr v = in[j];
    if (scale) v = fp_mul(v, ninv);
    out[bitrev_dev((uint32_t)j, L)] = v;
  }
}
__global__ void ntt_join_kernel(const Fr* __restrict__ a, const Fr* __restrict__ b,
                                const Fr* __restrict__ c, Fr* __restrict__ p, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = fp_from_mont(fp_sub(fp_mul(a[i], b[i]), c[i]));
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
  t.tile_log = L < kTileLogMax ? L : kTileLogMax;
  t.passes.clear();
  t.passes.push_back({0, t.tile_log, 0});
  for (int done = t.tile_log; done < L;) {
    int S = L - done;
    if (S > t.tile_log - kMinTb) S = t.tile_log - kMinTb;
    t.passes.push_back({done, S, t.tile_log - S});
    done += S;
  }
  const size_t N = (size_t)1 << L, half = N > 1 ? N / 2 : 1;
  G16_HIP(hipMalloc(&t.tw_fwd, half * sizeof(Fr)));
  G16_HIP(hipMalloc(&t.tw_inv, half * sizeof(Fr)));
  G16_HIP(hipMalloc(&t.coset, N * sizeof(Fr)));
  const Fr w = host_root(L), winv = fp_inv(w);
  const Fr inc = host_root(L + 1);  // w_2N (L <= 27 so L+1 <= 28)
  const Fr ninv = fp_inv(host_from_u64(N));
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

static int run_passes(const NttTables& t, Fr* const* vecs, int nvec, bool dif, hipStream_t st) {
  if (nvec < 1 || nvec > 4) { set_error("ntt: nvec must be 1..4"); return G16_E_ARG; }
  if (t.L == 0) return G16_OK;
  VecPtrs vp{};
  for (int i = 0; i < nvec; i++) vp.p[i] = vecs[i];
  const unsigned ntiles = 1u << (t.L - t.tile_log);
  const int np = (int)t.passes.size();
  for (int k = 0; k < np; k++) {
    const NttPass& p = t.passes[dif ? np - 1 - k : k];
    ntt_pass_kernel<<<dim3(ntiles, nvec), kThreads, 0, st>>>(vp, dif ? t.tw_inv : t.tw_fwd, t.L,
                                                             t.tile_log, p.lo_bits, p.S, p.tb,
                                                             dif ? 1 : 0);
  }
  G16_HIP(hipGetLastError());
  return G16_OK;
}

int ntt_dif_inverse(const NttTables& t, Fr* const* vecs, int nvec, hipStream_t st) {
  return run_passes(t, vecs, nvec, true, st);
}
int ntt_dit_forward(const NttTables& t, Fr* const* vecs, int nvec, hipStream_t st) {
  return run_passes(t, vecs, nvec, false, st);
}

int ntt_coset_scale(const NttTables& t, Fr* const* vecs, int nvec, hipStream_t st) {
  if (nvec < 1 || nvec > 4) { set_error("ntt: nvec must be 1..4"); return G16_E_ARG; }
  VecPtrs vp{};
  for (int i = 0; i < nvec; i++) vp.p[i] = vecs[i];
  const size_t N = (size_t)1 << t.L;
  ntt_scale_kernel<<<dim3((unsigned)((N + 255) / 256), nvec), 256, 0, st>>>(vp, t.coset, N);
  G16_HIP(hipGetLastError());
  return G16_OK;
}

int ntt_scale_bitrev(const NttTables& t, const Fr* in, Fr* out, bool scale_ninv, hipStream_t st) {
  const size_t N = (size_t)1 << t.L;
  const Fr ninv = fp_inv(host_from_u64(N));
  ntt_bitrev_kernel<<<(unsigned)((N + 255) / 256), 256, 0, st>>>(in, out, ninv, scale_ninv ? 1 : 0, t.L);
  G16_HIP(hipGetLastError());
  return G16_OK;
}

int ntt_join_abc(const Fr* a, const Fr* b, const Fr* c, Fr* p_std, size_t n, hipStream_t st) {
  ntt_join_kernel<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(a, b, c, p_std, n);
  G16_HIP(hipGetLastError());
  return G16_OK;
}

}  // namespace g16
