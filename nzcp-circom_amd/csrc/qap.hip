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
#include "fr29.cuh"
#include "internal.h"

namespace g16 {

__global__ __launch_bounds__(256) void qap_eval_kernel(
    const uint32_t* __restrict__ rpA, const uint32_t* __restrict__ colA, const F29* __restrict__ valA,
    const uint32_t* __restrict__ rpB, const uint32_t* __restrict__ colB, const F29* __restrict__ valB,
    const Fr* __restrict__ w, F29* __restrict__ a, F29* __restrict__ b, F29* __restrict__ cc, uint32_t N) {
  const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= N) return;
  // lazy sums: each term is below 1.1r; weak-reduce every 8 terms so the sum stays below 16r
  F29 sa = f29_zero(), sb = f29_zero();
  uint32_t cnt = 0;
  for (uint32_t k = rpA[c], e = rpA[c + 1]; k < e; k++) {
    sa = fr29_add(sa, fr29_mul(valA[k], fr29_repack(w[colA[k]])));
    if ((++cnt & 7u) == 0) sa = fr29_weak_reduce(sa);
  }
  cnt = 0;
  for (uint32_t k = rpB[c], e = rpB[c + 1]; k < e; k++) {
    sb = fr29_add(sb, fr29_mul(valB[k], fr29_repack(w[colB[k]])));
    if ((++cnt & 7u) == 0) sb = fr29_weak_reduce(sb);
  }
  a[c] = sa;
  b[c] = sb;
  cc[c] = fr29_mul(sa, sb);
}

__global__ __launch_bounds__(256) void qap_convert_kernel(const Fr* __restrict__ in, F29* __restrict__ out, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = fr29_from_zkey_coef(in[i]);
}

int qap_eval(const QapCsr& q, const Fr* w_std, F29* a, F29* b, F29* cc, hipStream_t st) {
  qap_eval_kernel<<<(q.N + 255) / 256, 256, 0, st>>>(q.row_ptr[0], q.col[0], q.val[0], q.row_ptr[1],
                                                      q.col[1], q.val[1], w_std, a, b, cc, q.N);
  G16_HIP(hipGetLastError());
  return G16_OK;
}

int qap_convert_coefs(const Fr* in, F29* out, size_t n, hipStream_t st) {
  if (n == 0) return G16_OK;
  qap_convert_kernel<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(in, out, n);
  G16_HIP(hipGetLastError());
  return G16_OK;
}

}  // namespace g16
