// Internal interfaces between the translation units of libg16hip.so (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include <string.h>
#include <functional>
#include <string>
#include <vector>

#include "../../include/g16_prover.h"
#include "ec.cuh"
#include "fq29.cuh"

struct g16_prover;
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


// ---------------------------------------------------------------- sharded pipeline inside ONE process (prover.cpp -> multi.cpp)
// Enqueue-only pieces of the two-phase sharded proof (g16_shard_begin / g16_shard_end are the same pieces with host
// copies and synchronisation in between, for hosts that exchange the slices themselves: one process per GPU over RCCL).
// Every call works on the handle's context 0 and returns without waiting for the device unless it says otherwise.
struct ShardView {
  int device = 0;
  hipStream_t st = nullptr;        // the handle's main stream: QAP -> NTT -> (slices arrive) -> join -> H-MSM
  Fr* d_w = nullptr;               // witness slot (device, nVars words)
  F29* vec[3] = {nullptr, nullptr, nullptr};   // A, B, C vectors of the context (N lazy elements each)
  uint32_t lo = 0, hi = 0;         // the handle's range of the domain
  uint32_t N = 0, nVars = 0;
};
int shard_view(g16_prover* p, uint32_t slot, ShardView* out);                 // (allocates the slot)
// parse + upload + canonicity check of a wtns buffer into `slot`, all behind each other on the main stream
int shard_upload_witness(g16_prover* p, uint32_t slot, const uint8_t* wtns, size_t len);
int shard_witness_verdict(g16_prover* p);                                     // after the main stream has drained
// QAP + odd-coset evaluation of the vectors in `mask`, and the witness MSMs of the handle's point range
int shard_begin_async(g16_prover* p, uint32_t slot, uint32_t mask);
// join of [lo, hi), H-MSM, then WAIT and fold: the handle's partial sums
int shard_end_collect(g16_prover* p, uint8_t partial[G16_PARTIAL_BYTES]);
// after a failure between begin and end: wait for what begin launched
void shard_drain(g16_prover* p);

// ---------------------------------------------------------------- trapdoor setup on the device (setup_gpu.hip)
// out[i] = [k_i] G as affine Montgomery bytes; tbl = host table [nwin][2^wb - 1] of d * 2^(wb j) * G, k in Montgomery form
int setup_fixed_mul_g1(int device, const G1Affine* tbl, int wb, int nwin, const Fr* ks_mont, size_t n, uint8_t* out);
int setup_fixed_mul_g2(int device, const G2Affine* tbl, int wb, int nwin, const Fr* ks_mont, size_t n, uint8_t* out);

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
// inverse transform + coset table + forward transform of `nvec` vectors in place, the two middle passes fused in LDS;
// join_p != nullptr (nvec == 3: a, b, c): the last forward pass also joins, join_p[i] = plain(a'b' - c') (vectors consumed)
int ntt_coset_roundtrip(const NttTables& t, F29* const* vecs, int nvec, Fr* join_p, hipStream_t st);
// ntt_dit_forward of a, b, c with the join fused into the last pass (the vectors are consumed)
int ntt_dit_forward_join(const NttTables& t, F29* a, F29* b, F29* c, Fr* p_std, hipStream_t st);
// P[i] = plain(a[i]*b[i] - c[i])   (qap_joinABC + batchFromMontgomery)
int ntt_join_abc(const F29* a, const F29* b, const F29* c, Fr* p_std, size_t n, hipStream_t st);

// ---------------------------------------------------------------- QAP (qap.hip)
struct QapCsr {
  // CSR of zkey section 4 per matrix (0 = A, 1 = B): rows = domainSize
  uint32_t* row_ptr[2] = {nullptr, nullptr};  // [N+1]
  // A row's records are ordered: first those whose coefficient is +1 or -1 (79 % of the real NZCP circuit's 4.2 M records:
  // they ADD the witness word's Montgomery image, bit 31 of col = minus), then the general ones (a product each)
  uint32_t* col[2] = {nullptr, nullptr};      // signal index per record (| 0x80000000: coefficient -1, in the +-1 part)
  uint32_t* mid[2] = {nullptr, nullptr};      // [N] where the general records of a row start (an index into col)
  uint32_t* vptr[2] = {nullptr, nullptr};     // [N] ... and their first coefficient (an index into val)
  F29* val[2] = {nullptr, nullptr};           // general records only: coef * 2^522 (lazy format): one product with the plain witness word
  size_t nnz[2] = {0, 0}, ngen[2] = {0, 0};
  uint32_t N = 0;
  // rows with more than kQapLongRow terms in A or B (the modular-addition rows of a SHA-256 circuit carry
  // ~260): one wavefront each instead of one lane
  uint32_t* long_rows = nullptr;   // first the n_mid rows of at most kQapWaveRow terms (eight lanes each), then the rest
  uint32_t n_long = 0, n_mid = 0;
};
static constexpr uint32_t kQapLongRow = 16;
static constexpr uint32_t kQapWaveRow = 64;
// a[c] = sum val*w[col] (lazy Montgomery), b likewise, cc = a*b; w is the standard-form witness
// w_mont: scratch of q's witness length (filled here: the Montgomery image of every witness word, for the +-1 records)
int qap_eval(const QapCsr& q, const Fr* w_std, uint32_t n_w, F29* w_mont, F29* a, F29* b, F29* cc, hipStream_t st);
// canonicity of the staged witness words: *h_flag (pinned) <- lowest index of a word >= r, or 0xffffffff, once `st`
// has passed this point
int qap_check_witness(const Fr* w_std, uint32_t n, uint32_t* d_flag, uint32_t* h_flag, hipStream_t st);
// zkey section-4 words (device, canonical) -> the lazy coefficient format, once at create
int qap_convert_coefs(const Fr* in, F29* out, size_t n, hipStream_t st);

// ---------------------------------------------------------------- MSM (msm_g1.hip / msm_g2.hip)
struct MsmConfig {
  int c = 0;           // window bits (0 = choose from n)
  int task_len = 0;    // max sorted entries per accumulation task (0 = default)
  bool dense = true;   // scalars uniform in Fr (H) vs NZCP witness mix (~1/3 full-width, ~30 % equal to 1)
  int precomp = 0;     // window precomputation factor of a single-section group (0 = auto, 1 = none): MsmGroup::pf
  int dup_chunk = 0;   // bits per chunk of a repeated value (0 = kDupChunkBits): MsmGroup::dup_chunk
};
struct MsmWorkspace;   // opaque, msm.cuh
static constexpr int kMsmMaxSections = 3;
// One base section handed to msm_group_create: n_total affine points in zkey file layout (64 B G1); infinity
// points are compacted away.  bases2_host (optional) is the G2 twin of the SAME points (zkey section 7 next to
// section 6: [v_i]G2 beside [v_i]G1): it rides on the section's sorted bucket lists (the "G2 lane").
struct MsmSectionIn {
  const uint8_t* bases_host = nullptr;   // G1 points (or nullptr for a G2-only group, see bases2_host)
  const uint8_t* bases2_host = nullptr;  // G2 points (128 B)
  uint32_t n_total = 0;
  uint32_t scalar_offset = 0;            // scalar index of the section's first point
};
// A group of base sections that share ONE front end (digit extraction, bucket sort, task cut) over one scalar
// vector: the witness MSMs A, B1, C (+ B2 riding on B1's buckets) are one group, the H-MSM another.  Bases stay
// resident in HBM in the kernels' packed format.
struct MsmGroup {
  int nsec = 0;
  uint32_t sec_n[kMsmMaxSections] = {0, 0, 0};       // non-infinity points per section
  uint32_t sec_begin[kMsmMaxSections] = {0, 0, 0};   // their start in the concatenated point index space
  uint32_t n = 0;                                    // total points
  void* d_bases = nullptr;      // PackedAffine<G1>[pf * n], or nullptr (G2-only group)
  void* d_bases2 = nullptr;     // PackedAffine<G2>[pf * sec_n[g2_sec]], or nullptr
  int g2_sec = -1;              // section the G2 lane rides on
  uint32_t* d_src = nullptr;    // [n] scalar index of point g
  // Window precomputation (single-section groups): the base table also holds 2^(c W k) * P_i for k < pf, so
  // scalar window j = k W + r of point i lands in row r as table entry k n + i; with pf = Ws (the dense H-MSM's
  // default) ONE row of buckets collects every window and wider windows pay: c = 20 needs 13 additions per point
  // instead of 16.
  int c = 0, Ws = 0, W = 0;     // window bits; scalar windows ceil(254 / c); bucket rows of digit windows per section
  uint32_t pf = 1;
  // Width of the scalar windows.  Default (wb = c, wx = 0): window j = bits [c j, c j + c), the top one holds what is left of
  // the 254 bits -- 7 at c = 19, 2 at c = 21, and with ONE row of buckets (pf = Ws) those few digit values are a handful of
  // buckets that take all n top-window entries: a single bin of the two-level sort, one workgroup sorting n entries (r02: 1.4
  // ms at n = 2^20; H window 19 / 21 cost 5.5 / 7.1 ms per proof against 4.05 at 20).  Fully precomputed groups therefore
  // split the 255 bits (254 + the carry of the signed recoding) EVENLY: wx windows of wb + 1 bits, then Ws - wx of wb bits
  // (c = 20: 8 x 20 + 5 x 19; c = 19: 3 x 19 + 11 x 18), every level of the table at its own window's offset.
  uint32_t wb = 0, wx = 0;
  // Optional (G16_WINDOW_ORDER=1; fully precomputed groups of at most 16 windows): the bucket sort's key carries the window
  // index (4 bits) below the low bucket bits, so a bucket's entries -- and with them every accumulate task -- walk the levels
  // of the base table in ascending order, and the wavefronts, which start together and take ~two tasks each, gather from the
  // same one or two 64 MB levels of the 0.87 GB table at any moment (the idea: keep the live slab inside the 256 MB
  // Infinity Cache; r02: TCC_MISS x 64 B = 1.34 GB per H launch).  r03 measured it: H accumulate 1.042 ms with and 1.041
  // without, the sort 34 us slower -- the gathers' misses are already hidden behind the integer work.  Off by default.
  uint32_t wkb = 0;
  uint32_t B = 0;               // buckets per row 2^(c-1)
  uint32_t low_bits = 0, bins = 1;   // bucket = bin << low_bits | low: the two levels of the sort
  bool ones = false;            // extra unweighted row per section for the scalars equal to 1 (witness groups)
  // The TOP scalar window holds only 254 - c (Ws - 1) bits (7 of 12 at c = 13, and r >> 247 = 48: 49 buckets would
  // take every full-width scalar's top digit -- thousands of entries, ~150 task partials each, a latency-bound
  // wavefront-per-bucket combine: r02, 1.1 ms on the G2 lane of the real NZCP witness).  Its digits are spread
  // instead: bucket = (digit - 1) << salt_bits | (point index & mask), every bucket of a digit weighted alike.
  uint32_t salt_bits = 0;
  // REPEATED scalar values (witness groups).  A circom witness is made of few distinct values: 63 % zeros, 16 % ones
  // and -- the real NZCP witness, r02 -- 174 k full-width words that are only 663 distinct inverses 1/(i - index) of
  // its QuinSelector comparisons.  Points whose scalar value is shared by >= kDupMin points of the section take one
  // entry in a "dup" row (bucket = a hash of the value; a bucket qualifies when every scalar in it is equal, checked
  // exactly) instead of one entry per window: sum_i s P_i = s (sum_i P_i).  The bucket sums T are then combined by
  // 16-bit chunk of the value (a lane multiplies its T by chunk k of its value, one shuffle tree per chunk: U_k = sum
  // chunk_k(s) T) and the host runs Horner over the 16 chunk sums.  dup_rows rows of B hash buckets follow the digit (+ ones) rows of every section.
  uint32_t dup_rows = 0;        // 0 = off
  uint32_t dup_bits = 0;        // log2(dup_rows * B) hash buckets per section
  // a repeated value is cut into chunks of dup_chunk bits, one output row per chunk (dup_bit_rows of them); G16_DUP_CHUNK =
  // 4 / 8 / 16 (sweeps).  8 bits halve the dependent steps per lane for twice the lanes: measured on the whole key and on
  // shards of it (profiles/r03_sweeps.txt 10, 17), no gain either way
  // dup_chunk = the window width when c <= 16 (then the chunk sums join the window sums and the host runs ONE Horner pass,
  // msm_collect); dup_chunk_wide = 16, fewer rows of device work: the batch pipeline, which is device-bound, uses it
  // (msm_set_throughput).  dup_rows_cap = the larger row count of the two (buffer sizes).
  uint32_t dup_chunk = 16, dup_chunk_wide = 16, dup_rows_cap = 16;
  uint32_t rps = 0, rows = 0;   // rows per section (W + ones + dup_rows), rows in total
  uint32_t task_len = 0;        // of the G1 lane
  bool task_len_forced = false; // MsmConfig::task_len given: every lane uses it
  bool dense = true;
  bool src_identity = false;   // point g takes scalar g (no infinity points dropped, no offset): the front end skips d_src
  uint32_t chunks = 1, per = 0; // front-end geometry: workgroups over the points, points per workgroup
  uint64_t max_entries = 0;     // upper bound on sorted entries of one launch
};
size_t msm_point_bytes(int curve);   // canonical XYZZ bytes: 128 (G1) / 256 (G2)
int msm_group_create(MsmGroup& g, const MsmSectionIn* secs, int nsec, const MsmConfig& cfg);
void msm_group_destroy(MsmGroup& g);
int msm_workspace_create(MsmWorkspace** ws, const MsmGroup& g);
// priority of the streams the NEXT workspaces created on this thread make for themselves (the G2 lane's dup-row
// stream); kMsmPrioHighest = the device's highest (default).  See prover.cpp create_impl: hardware-queue pools.
static constexpr int kMsmPrioHighest = -1000;
void msm_set_aux_stream_priority(int prio);
void msm_workspace_destroy(MsmWorkspace* ws);
// Per-section results of one launch: XYZZ sums in the canonical Montgomery(2^256) image.
struct MsmResult {
  G1XYZZ g1[kMsmMaxSections];
  G2XYZZ g2;
};
static constexpr uint32_t kDupMin = 8;       // points sharing a value before the dup row pays (1 entry + ~127 tree
                                             // additions per value against one entry per window)
static constexpr uint32_t kDupChunkBits = 16; // default chunk width of a repeated value (MsmGroup::dup_chunk)
// msm_launch only enqueues: front end + G1 lane on `st`, the G2 lane (if any) forks onto `st2` after the sort
// (st2 == nullptr or == st: same stream).  msm_collect waits for both, folds the window sums on the host
// (Horner, c doublings per row) and fills `out`.  One launch in flight per workspace.
int msm_launch(const MsmGroup& g, MsmWorkspace* ws, const Fr* d_scalars, hipStream_t st, hipStream_t st2);
// the same in two steps, so that a caller can order the lanes of one group after events of another:
// gate1 / gate2 (optional) hold back the G1 / G2 bucket-accumulate kernel
int msm_launch_front(const MsmGroup& g, MsmWorkspace* ws, const Fr* d_scalars, hipStream_t st);
int msm_launch_lanes(const MsmGroup& g, MsmWorkspace* ws, hipStream_t st, hipStream_t st2, hipEvent_t gate1,
                     hipEvent_t gate2);
void msm_set_quota(MsmWorkspace* ws, uint32_t quota_g1, uint32_t quota_g2);   // accumulate wavefronts retire after this many chunks of 64 tasks (0 = persistent)
hipEvent_t msm_event(MsmWorkspace* ws, int which);   // 0 = sort done, 1 / 2 = G1 accumulate kernel started / done, 3 / 4 = G1 / G2 lane done (nullptr: no such lane)
// after_g1 (optional) runs once the G1 sums are folded and BEFORE the wait for the G2 lane: host work that needs only
// the G1 results overlaps the G2 lane's last kernels (prover.cpp: the G1 half of the proof assembly)
int msm_collect(const MsmGroup& g, MsmWorkspace* ws, MsmResult* out,
                const std::function<void(const MsmResult&)>* after_g1 = nullptr);
double trace_ms();   // G16_TRACE_HOST: milliseconds since the last proof was launched (prover.cpp)
float msm_last_accum_ms(const MsmWorkspace* ws, int lane);        // lane 0 = G1, 1 = G2: the accumulate kernel alone
// throughput mode of the next launch (the batch pipeline): 16-bit chunks of the repeated values (MsmGroup::dup_chunk_wide) and
// longer reduce segments on dense rows (MsmLaneWs::seg_len_thr) -- fewer instructions, longer chains
void msm_set_throughput(MsmWorkspace* ws, const MsmGroup& g, bool on);
void msm_set_waves(MsmWorkspace* ws, uint32_t waves_g1, uint32_t waves_g2);   // persistent accumulate grids, wavefronts per SIMD (0 = full occupancy)
float msm_event_offset_ms(MsmWorkspace* ws, hipEvent_t base, int lane, int which);   // G16_TRACE_HOST timeline

// total = sum_j 2^(c j) * windows[j]  (Horner, c doublings per window) [+ the unweighted ones row]
template <class F> inline void msm_combine_windows(XYZZ<F>& total, const XYZZ<F>* windows, int W, int c, bool ones) {
  xyzz_set_inf(total);
  for (int j = W - 1; j >= 0; j--) {
    if (!xyzz_is_inf(total))
      for (int k = 0; k < c; k++) xyzz_dbl(total);
    xyzz_add(total, windows[j]);
  }
  if (ones) xyzz_add(total, windows[W]);
}
// total += sum_k 2^(k L) * bits[k]  (the dup rows' chunk sums, L = chunk bits: Horner, L doublings per row)
template <class F> inline void msm_add_bit_sums(XYZZ<F>& total, const XYZZ<F>* bits, uint32_t chunk_bits, uint32_t nrows) {
  XYZZ<F> acc;
  xyzz_set_inf(acc);
  for (int b = (int)nrows - 1; b >= 0; b--) {
    if (!xyzz_is_inf(acc))
      for (uint32_t k = 0; k < chunk_bits; k++) xyzz_dbl(acc);
    xyzz_add(acc, bits[b]);
  }
  xyzz_add(total, acc);
}

}  // namespace g16
