// Internal interfaces between the translation units of libg16hip.so (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include <string.h>
#include <string>
#include <vector>

#include "../../include/g16_prover.h"
#include "ec.cuh"
#include "fq29.cuh"

namespace g16 {

// ---------------------------------------------------------------- errors (thread-local text)
void set_error(const std::string& msg);
const char* get_error();

#define G16_HIP(expr)                                                                      \
  do {                                                                                     \
    hipError_t _e = (expr);                                                                \
    if (_e != hipSuccess) {                                                                \
      g16::set_error(std::string(#expr) + ": " + hipGetErrorString(_e));                   \
      return G16_E_HIP;                                                                    \
    }                                                                                      \
  } while (0)


// ---------------------------------------------------------------- NTT (ntt.hip)
struct NttPass { int lo_bits, S, tb; };
struct NttTables {
  int L = -1;                // log2 N
  int tile_log = 0;
  F29* tw_fwd = nullptr;     // w_N^i, i < N/2 (9x29 lazy format, Montgomery 2^261)
  F29* tw_inv = nullptr;     // w_N^-i
  F29* coset = nullptr;      // coset[j] = N^-1 * w_2N^bitrev(j)  (position order after the DIF iNTT)
  std::vector<NttPass> passes;  // DIT order; DIF runs them reversed
};
int ntt_tables_create(NttTables& t, int L, hipStream_t st);
void ntt_tables_destroy(NttTables& t);
// Batched in-place transforms over `nvec` vectors (device pointers in host array vecs).
// dif_inverse: natural in -> bit-reversed out with w^-1 (no scaling);  dit_forward: bit-reversed
// in -> natural out with w.
int ntt_dif_inverse(const NttTables& t, F29* const* vecs, int nvec, hipStream_t st);
int ntt_dit_forward(const NttTables& t, F29* const* vecs, int nvec, hipStream_t st);
// ntt_dif_inverse followed by ntt_coset_scale, the table multiply fused into the last pass's store
int ntt_dif_inverse_coset(const NttTables& t, F29* const* vecs, int nvec, hipStream_t st);
// x[j] *= coset[j]  (fused 1/N and w_2N^i shift in bit-reversed position order)
int ntt_coset_scale(const NttTables& t, F29* const* vecs, int nvec, hipStream_t st);
// operator-level glue: canonical Montgomery(2^256) Fr image <-> the kernels' lazy format (optionally
// through the bit-reversal permutation; export can fold in the 1/N of the inverse transform)
int ntt_import(const NttTables& t, const Fr* in, F29* out, bool bitrev, hipStream_t st);
int ntt_export(const NttTables& t, const F29* in, Fr* out, bool bitrev, bool scale_ninv, hipStream_t st);
// P[i] = plain(a[i]*b[i] - c[i])   (qap_joinABC + batchFromMontgomery)
int ntt_join_abc(const F29* a, const F29* b, const F29* c, Fr* p_std, size_t n, hipStream_t st);

// ---------------------------------------------------------------- QAP (qap.hip)
struct QapCsr {
  // CSR of zkey section 4 per matrix (0 = A, 1 = B): rows = domainSize
  uint32_t* row_ptr[2] = {nullptr, nullptr};  // [N+1]
  uint32_t* col[2] = {nullptr, nullptr};      // signal index per record
  F29* val[2] = {nullptr, nullptr};           // coef * 2^522 (lazy format): one product with the plain witness word
  size_t nnz[2] = {0, 0};
  uint32_t N = 0;
  // rows with more than kQapLongRow terms in A or B (the modular-addition rows of a SHA-256 circuit carry
  // ~260): one wavefront each instead of one lane
  uint32_t* long_rows = nullptr;
  uint32_t n_long = 0;
};
static constexpr uint32_t kQapLongRow = 16;
// a[c] = sum val*w[col] (lazy Montgomery), b likewise, cc = a*b; w is the standard-form witness
int qap_eval(const QapCsr& q, const Fr* w_std, F29* a, F29* b, F29* cc, hipStream_t st);
// zkey section-4 words (device, canonical) -> the lazy coefficient format, once at create
int qap_convert_coefs(const Fr* in, F29* out, size_t n, hipStream_t st);

// ---------------------------------------------------------------- MSM (msm_g1.hip / msm_g2.hip)
struct MsmConfig {
  int c = 0;           // window bits (0 = choose from n)
  int task_len = 0;    // max sorted entries per accumulation task (0 = default)
  bool dense = true;   // scalars uniform in Fr (H) vs NZCP witness mix (~1/3 full-width)
  int precomp = 0;     // window precomputation factor (0 = default, 1 = none): see MsmInstance::pf
};
struct MsmWorkspace;   // opaque, msm.cuh
// Fixed-base-set MSM instance: bases resident in HBM, infinity points compacted away.
struct MsmInstance {
  int curve = 1;               // 1 = G1, 2 = G2
  uint32_t n = 0;              // non-infinity bases
  void* d_bases = nullptr;     // Affine<F>[n] (Montgomery)
  uint32_t* d_src = nullptr;   // scalar index of base i (into the scalar vector handed to run)
  // Window precomputation: the base table also holds 2^(c W k) * P_i for k = 1..pf-1 (computed once at
  // create), so scalar window j = k W + r of point i lands in row r as entry k n + i: only W = ceil(Ws / pf)
  // rows of buckets are reduced instead of Ws, for the same number of bucket additions.
  int c = 0, W = 0;            // window bits; ROWS of buckets (= output window sums, Horner on the host)
  int Ws = 0;                  // scalar windows = ceil(256 / c)
  uint32_t pf = 1;             // precomputation factor
  uint32_t n_ext = 0;          // pf * n = entries per row = points in d_bases
  uint32_t nbuckets = 0;       // per window = 2^(c-1)
  uint32_t task_len = 0;
  bool dense = true;           // MsmConfig::dense (the H-MSM)
};
size_t msm_point_bytes(int curve);   // XYZZ bytes: 128 (G1) / 256 (G2)
// bases_host: n_total affine points in file layout; keeps only non-infinity ones.
int msm_instance_create(MsmInstance& m, int curve, const uint8_t* bases_host, uint32_t n_total,
                        uint32_t scalar_offset, const MsmConfig& cfg);
void msm_instance_destroy(MsmInstance& m);
int msm_workspace_create(MsmWorkspace** ws, const MsmInstance* insts, int ninst);
void msm_workspace_destroy(MsmWorkspace* ws);
// Runs the MSM of `m` against scalars d_scalars (standard form, 32 B each) and writes the W
// per-window sums (XYZZ, Montgomery) to host memory out_windows: (W + 1) * msm_point_bytes, the
// last entry being the unweighted sum of the scalar == 1 points (combine: msm_combine_windows below).
float msm_last_accum_ms(const MsmWorkspace* ws);
// accumulate kernel of the next msm_launch on `ws`: wait for `accum_gate` first (nullptr = none); persistent
// grid of `waves_per_simd` wavefronts per SIMD (0 = full occupancy)
void msm_set_schedule(MsmWorkspace* ws, hipEvent_t accum_gate, uint32_t waves_per_simd);
hipEvent_t msm_sorted_event(MsmWorkspace* ws);       // recorded when the last launch's sorted task list is ready
hipEvent_t msm_accum_done_event(MsmWorkspace* ws);   // recorded after the accumulate kernel of the last launch
float msm_accum_event_offset_ms(MsmWorkspace* ws, hipEvent_t base, int which);   // G16_TRACE_HOST timeline
int msm_run(const MsmInstance& m, MsmWorkspace* ws, const Fr* d_scalars, uint8_t* out_windows,
            hipStream_t st);
// The same split in two so several MSMs can be in flight on different streams: msm_launch only
// enqueues (each MSM needs its own workspace), msm_collect waits for that stream and copies out.
int msm_launch(const MsmInstance& m, MsmWorkspace* ws, const Fr* d_scalars, hipStream_t st);
int msm_collect(MsmWorkspace* ws, uint8_t* out_windows, hipStream_t st);

// total = sum_j 2^(c j) * windows[j]  (Horner, c doublings per window) + windows[W] (ones window)
template <class F> inline void msm_combine_windows(XYZZ<F>& total, const uint8_t* windows, int W, int c) {
  xyzz_set_inf(total);
  for (int j = W - 1; j >= 0; j--) {
    if (!xyzz_is_inf(total))
      for (int k = 0; k < c; k++) xyzz_dbl(total);
    XYZZ<F> w;
    memcpy(&w, windows + (size_t)j * sizeof(XYZZ<F>), sizeof(w));
    xyzz_add(total, w);
  }
  XYZZ<F> ones;
  memcpy(&ones, windows + (size_t)W * sizeof(XYZZ<F>), sizeof(ones));
  xyzz_add(total, ones);
}


}  // namespace g16
