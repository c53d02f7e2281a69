// Witness x QAP evaluation: A_T = A.w, B_T = B.w, C_T = A_T o B_T.
//
// Replaces snarkjs 0.4.12 groth16_prove.js `buildABC1` (pin /root/reference/yarn.lock:987-1001),
// which walks the zkey section-4 records {m, c, s, coef} on one JS thread with one WASM call per
// record.  Here the records are regrouped once (at g16_create) into a CSR per matrix that stays
// resident in HBM; one thread owns one constraint row and produces a_c, b_c and c_c = a_c*b_c.
//
// Arithmetic follows the reference's trick: the file stores coef*R^2 (R = 2^256) as a plain integer and
// the witness word is in standard form, so ONE Montgomery product gives Montgomery(coef*w).  Here the
// coefficients are converted once to the kernels' 9x29 lazy format with the radix-2^261 equivalent
// (coef * 2^522), and the sums are lazy (fr29.cuh).
#include <stdlib.h>

#include "fr29.cuh"
#include "internal.h"

namespace g16 {

// issue priority of the QAP wavefronts: see ntt.hip g_ntt_prio (G16_CHAIN_PRIO)
__device__ int g_qap_prio = 0;
__device__ __forceinline__ void qap_set_prio() {
  const int p = g_qap_prio;
  if (p == 1) __builtin_amdgcn_s_setprio(1);
  else if (p == 2) __builtin_amdgcn_s_setprio(2);
  else if (p >= 3) __builtin_amdgcn_s_setprio(3);
}

// One matrix as the kernel sees it, and the sum of one row's records [k0, k1) taken with stride `step` from lane offset `lane`:
// the +-1 records add or subtract the Montgomery image of their witness word (a 40-byte gather and nine limb additions instead
// of a repack and a 229-instruction product: r03, 79 % of the records), the general ones multiply.  Lazy sums: an added term is
// below 1.1r, a subtracted one adds 2r - x <= 2r; a weak reduction (-> 1.0001r) whenever seven units have gone in keeps the
// sum below 1 + 6 * 1.1 + 2 = 9.6r < 16r.
struct QapMat { const uint32_t *rp, *mid, *vptr, *col; const F29* val; };
__device__ __forceinline__ F29 qap_row_sum(const QapMat& M, uint32_t c, uint32_t lane, uint32_t step,
                                            const Fr* __restrict__ w, const F29* __restrict__ wm) {
  F29 s = f29_zero();
  uint32_t cnt = 0;
  const uint32_t k0 = M.rp[c], km = M.mid[c], k1 = M.rp[c + 1];
  for (uint32_t k = k0 + lane; k < km; k += step) {
    const uint32_t ci = M.col[k];
    const F29 x = wm[ci & 0x7fffffffu];
    if (ci >> 31) { s = fr29_sub<2>(s, x); cnt += 2; }
    else { s = fr29_add(s, x); cnt += 1; }
    if (cnt >= 7) { s = fr29_weak_reduce(s); cnt = 0; }
  }
  const uint32_t vp = M.vptr[c];
  for (uint32_t k = km + lane; k < k1; k += step) {
    s = fr29_add(s, fr29_mul(M.val[vp + (k - km)], fr29_repack(w[M.col[k]])));
    if (++cnt >= 7) { s = fr29_weak_reduce(s); cnt = 0; }
  }
  return s;
}

__device__ __forceinline__ void qap_eval_rows(uint32_t block, const QapMat& A, const QapMat& B, const Fr* __restrict__ w,
                                              const F29* __restrict__ wm, F29* __restrict__ a, F29* __restrict__ b,
                                              F29* __restrict__ cc, uint32_t N) {
  const uint32_t c = block * blockDim.x + threadIdx.x;
  if (c >= N) return;
  if (A.rp[c + 1] - A.rp[c] > kQapLongRow || B.rp[c + 1] - B.rp[c] > kQapLongRow) return;   // the grouped rows below
  const F29 sa = qap_row_sum(A, c, 0u, 1u, w, wm), sb = qap_row_sum(B, c, 0u, 1u, w, wm);
  a[c] = sa;
  b[c] = sb;
  cc[c] = fr29_mul(sa, sb);
}

__device__ __forceinline__ F29 f29_shfl_xor(const F29& v, int mask) {
  F29 r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.l[i] = (uint32_t)__shfl_xor((int)v.l[i], mask, 64);
  r.pad_ = 0;
  return r;
}
// GW lanes per long row (GW = 64: a wavefront; GW = 8: eight rows per wavefront): lanes stride over the row's records,
// partial sums meet in a log2(GW)-step __shfl_xor tree (weak-reduced on the way so they stay below 16r).
// The real NZCP circuit has 48 k rows of 17-32 terms (and a single B term) beside its 2.5 k rows of 65-356 terms: with a
// whole wavefront per row (r02) those 48 k rows cost 65 M wavefront-instructions -- 4 % of a proof, two 6-step trees over
// mostly empty lanes each; eight-lane groups do them for a tenth of that.
template <int GW>
__device__ __forceinline__ void qap_long_rows(uint32_t block, const uint32_t* __restrict__ rows, uint32_t n_long,
                                              const QapMat& A, const QapMat& B, const Fr* __restrict__ w,
                                              const F29* __restrict__ wm, F29* __restrict__ a, F29* __restrict__ b,
                                              F29* __restrict__ cc) {
  const uint32_t tid = block * blockDim.x + threadIdx.x, gid = tid / GW, lane = tid % GW;
  if (gid >= n_long) return;   // (whole groups leave together: the shuffles below stay inside a group)
  const uint32_t c = rows[gid];
  F29 s[2];
#pragma unroll
  for (int m = 0; m < 2; m++) {
    s[m] = fr29_weak_reduce(qap_row_sum(m ? B : A, c, lane, (uint32_t)GW, w, wm));
    for (int d = GW / 2; d >= 1; d >>= 1) {
      s[m] = fr29_add(s[m], f29_shfl_xor(s[m], d));
      if (d == 8 || d == 1) s[m] = fr29_weak_reduce(s[m]);   // at most 8 partials below ~2r between reductions
    }
  }
  if (lane == 0) {
    a[c] = s[0];
    b[c] = s[1];
    cc[c] = fr29_mul(s[0], s[1]);
  }
}

// ONE launch for the three kinds of rows (they write disjoint rows and depend on nothing but the witness): blocks
// [0, nb_long) take the rows above kQapWaveRow terms, a wavefront each -- first, they are the longest -- then nb_mid blocks
// the rows of 17-64 terms in eight-lane groups, the rest a row per thread.  r02's three back-to-back launches at the head
// of a proof's critical chain lasted 0.15 ms standalone; side by side they last as long as the longest.
struct QapRows { const uint32_t* rows; uint32_t n_mid, n_long, nb_long, nb_mid; };
__global__ __launch_bounds__(256) void qap_eval_kernel(QapRows lr, QapMat A, QapMat B, const Fr* __restrict__ w,
                                                       const F29* __restrict__ wm, F29* __restrict__ a,
                                                       F29* __restrict__ b, F29* __restrict__ cc, uint32_t N) {
  qap_set_prio();
  const uint32_t blk = blockIdx.x;   // (uniform per block: no divergence between the three bodies)
  if (blk < lr.nb_long)
    qap_long_rows<64>(blk, lr.rows + lr.n_mid, lr.n_long - lr.n_mid, A, B, w, wm, a, b, cc);
  else if (blk < lr.nb_long + lr.nb_mid)
    qap_long_rows<8>(blk - lr.nb_long, lr.rows, lr.n_mid, A, B, w, wm, a, b, cc);
  else
    qap_eval_rows(blk - lr.nb_long - lr.nb_mid, A, B, w, wm, a, b, cc, N);
}
// wm[i] = Montgomery image of the witness word i (lazy format): what a +-1 record adds
__global__ __launch_bounds__(256) void qap_witness_mont_kernel(const Fr* __restrict__ w, F29* __restrict__ wm, uint32_t n) {
  qap_set_prio();
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) wm[i] = fr29_from_plain(w[i]);
}

__global__ __launch_bounds__(256) void qap_convert_kernel(const Fr* __restrict__ in, F29* __restrict__ out, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = fr29_from_zkey_coef(in[i]);
}

int qap_eval(const QapCsr& q, const Fr* w_std, uint32_t n_w, F29* w_mont, F29* a, F29* b, F29* cc, hipStream_t st) {
  // long_rows = [the n_mid rows of kQapLongRow < terms <= kQapWaveRow][the rows above]
  QapRows lr;
  lr.rows = q.long_rows;
  lr.n_mid = q.n_mid;
  lr.n_long = q.n_long;
  lr.nb_long = (q.n_long - q.n_mid + 3) / 4;
  lr.nb_mid = (q.n_mid + 31) / 32;
  const QapMat A{q.row_ptr[0], q.mid[0], q.vptr[0], q.col[0], q.val[0]}, B{q.row_ptr[1], q.mid[1], q.vptr[1], q.col[1], q.val[1]};
  if (n_w) qap_witness_mont_kernel<<<(n_w + 255) / 256, 256, 0, st>>>(w_std, w_mont, n_w);
  qap_eval_kernel<<<lr.nb_long + lr.nb_mid + (q.N + 255) / 256, 256, 0, st>>>(lr, A, B, w_std, w_mont, a, b, cc, q.N);
  G16_HIP(hipGetLastError());
  return G16_OK;
}

// Witness words must be canonical residues (what every circom witness calculator writes): the signed-digit
// recoding of the MSM assumes scalars below r.  flag[0] <- the lowest index of a word >= r (0xffffffff: none).
// On the device: the words are in HBM anyway, and the host-side scan cost ~1 ms of every g16_prove (r01).
__global__ __launch_bounds__(256) void qap_check_witness_kernel(const Fr* __restrict__ w, uint32_t n, uint32_t* __restrict__ flag) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  static const uint32_t kR[8] = G16_FR_P;
  const Fr x = w[i];
  bool lt = false, decided = false;
#pragma unroll
  for (int k = 7; k >= 0; k--) {
    if (!decided && x.v[k] != kR[k]) { lt = x.v[k] < kR[k]; decided = true; }
  }
  if (!lt) atomicMin(flag, i);
}
int qap_check_witness(const Fr* w_std, uint32_t n, uint32_t* d_flag, uint32_t* h_flag, hipStream_t st) {
  G16_HIP(hipMemsetAsync(d_flag, 0xff, 4, st));
  if (n) qap_check_witness_kernel<<<(n + 255) / 256, 256, 0, st>>>(w_std, n, d_flag);
  G16_HIP(hipGetLastError());
  G16_HIP(hipMemcpyAsync(h_flag, d_flag, 4, hipMemcpyDeviceToHost, st));
  return G16_OK;
}

int qap_convert_coefs(const Fr* in, F29* out, size_t n, hipStream_t st) {
  if (n == 0) return G16_OK;
  {
    const int prio = getenv("G16_CHAIN_PRIO") ? atoi(getenv("G16_CHAIN_PRIO")) : 0;
    G16_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_qap_prio), &prio, sizeof(int)));
  }
  qap_convert_kernel<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(in, out, n);
  G16_HIP(hipGetLastError());
  return G16_OK;
}

}  // namespace g16
