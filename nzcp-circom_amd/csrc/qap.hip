// Witness x QAP evaluation: A_T = A.w, B_T = B.w, C_T = A_T o B_T.
//
// Replaces snarkjs 0.4.12 groth16_prove.js `buildABC1` (pin /root/reference/yarn.lock:987-1001),
// which walks the zkey section-4 records {m, c, s, coef} on one JS thread with one WASM call per
// record.  Here the records are regrouped once (at g16_create) into a CSR per matrix that stays
// resident in HBM; one thread owns one constraint row and produces a_c, b_c and c_c = a_c*b_c.
//
// Arithmetic is exactly the reference's: the file stores coef*R^2 as a plain integer and the
// witness word is in standard form, so one Montgomery product gives Montgomery(coef*w).
#include "internal.h"

namespace g16 {

__global__ __launch_bounds__(256) void qap_eval_kernel(
    const uint32_t* __restrict__ rpA, const uint32_t* __restrict__ colA, const Fr* __restrict__ valA,
    const uint32_t* __restrict__ rpB, const uint32_t* __restrict__ colB, const Fr* __restrict__ valB,
    const Fr* __restrict__ w, Fr* __restrict__ a, Fr* __restrict__ b, Fr* __restrict__ cc, uint32_t N) {
  const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= N) return;
  Fr sa = fp_zero<FrParams>(), sb = fp_zero<FrParams>();
  for (uint32_t k = rpA[c], e = rpA[c + 1]; k < e; k++) sa = fp_add(sa, fp_mul(valA[k], w[colA[k]]));
  for (uint32_t k = rpB[c], e = rpB[c + 1]; k < e; k++) sb = fp_add(sb, fp_mul(valB[k], w[colB[k]]));
  a[c] = sa;
  b[c] = sb;
  cc[c] = fp_mul(sa, sb);
}

int qap_eval(const QapCsr& q, const Fr* w_std, Fr* a, Fr* b, Fr* cc, hipStream_t st) {
  qap_eval_kernel<<<(q.N + 255) / 256, 256, 0, st>>>(q.row_ptr[0], q.col[0], q.val[0], q.row_ptr[1],
                                                      q.col[1], q.val[1], w_std, a, b, cc, q.N);
  G16_HIP(hipGetLastError());
  return G16_OK;
}

}  // namespace g16
