// Pippenger multi-scalar multiplication on BN254 G1/G2.
//
// Replaces ffjavascript 0.2.48 engine_multiexp.js + wasmcurves 0.1.0 build_multiexp.js
// (`G1/G2.multiExpAffine`, pins /root/reference/yarn.lock:408-416, 1132-1138): same inputs
// (affine Montgomery bases in zkey-section byte layout, 32-byte standard-form scalars), same
// group element out.  The reference chunks points across web-workers and runs an unsigned-window
// bucket method per chunk; this is a different schedule for the same sum (all point arithmetic on
// the 9x29-bit lazy field, fq29.cuh / ec29.cuh; bases converted once at g16_create):
//
//   1. msm_digits_kernel: signed c-bit digits (carry-free: one 256-bit add of the constant
//      K = sum 2^(c-1) 2^(cj) turns signed recoding into plain bit extraction), written window-major.
//      The scalar value 1 (~30 % of an NZCP witness) goes to an extra UNWEIGHTED
//      pseudo-window spread over its buckets, so there is no giant bucket.
//   2. msm_sort_kernel<0/1> + msm_hist_* + msm_scan_*: counting sort of (point, sign) by bucket key
//      = window*2^(c-1) + |digit|-1 with per-workgroup LDS histograms / cursors: no global atomics.
//   3. msm_task_fill_kernel: buckets are cut into tasks of <= task_len sorted entries.
//   4. msm_accumulate_kernel: persistent wavefronts over a task queue; a lane walks its task's slice of
//      the sorted list, gathers the 80/160-byte affine point and mixed-adds it into an XYZZ
//      accumulator held in VGPRs; a finished lane takes the next task.
//   5. msm_combine_light/heavy_kernel: task partials -> one sum per bucket (a lane per light bucket,
//      one wavefront with a __shfl_down tree per heavy bucket).
//   6. msm_bucket_reduce_kernel: per window, sum k*S_k by running sums over segments of 16 buckets
//      plus a short double-and-add for the segment offset; msm_wave_reduce_kernel: 64 -> 1 tree per
//      wavefront with __shfl_down of the limbs; msm_to_canon_kernel: back to the canonical image.
//   Window sums (W + 1 points) go back to the host, which does the c*W doublings (internal.h).
//
// Roofline note: ~2.25k VALU instructions (1.47k v_mad_u64_u32) per
// gathered point addition, 16 additions per 96 algorithmic bytes: integer-issue bound, not HBM bound.
#pragma once
#include <stdlib.h>

#include <atomic>

#include "ec29.cuh"
#include "internal.h"

namespace g16 {

struct MsmWorkspace {
  uint32_t max_entries = 0, max_buckets = 0, max_tasks = 0;
  uint32_t* d_cnt = nullptr;
  uint32_t* d_off = nullptr;
  uint32_t* d_toff = nullptr;
  uint32_t* d_sorted = nullptr;
  uint2* d_task_desc = nullptr;   // by task id
  uint4* d_qdesc = nullptr;       // by queue position: full-length tasks first (msm_task_fill_kernel)
  uint32_t* d_foff = nullptr;     // exclusive scan of the full-length task counts, [nb] = their total
  uint32_t* d_tile_c = nullptr;
  uint32_t* d_class = nullptr;    // [2][kRemClasses]: remainder-class totals and cursors
  uint32_t* d_queue = nullptr;    // [0] work-queue head of the accumulate kernel, [1] number of flagged tasks
  uint32_t* d_redo = nullptr;     // tasks whose fast-path sum met an exceptional case (recomputed by msm_redo_kernel)
  uint32_t* d_tile_a = nullptr;
  uint32_t* d_tile_b = nullptr;
  uint32_t* d_dig = nullptr;      // [(W+1)][n] digit codes (window-major)
  uint32_t* d_hist = nullptr;     // [(W+1)][chunks][B] per-workgroup histograms -> start offsets
  uint32_t chunks = 1;
  void* d_partial = nullptr;
  void* d_bsum = nullptr;         // one XYZZ per bucket after the combine pass
  uint32_t* d_heavy = nullptr;    // [0] = count, [1..] = bucket ids with more than kLightTasks partials
  uint32_t max_heavy = 0;
  void* d_seg = nullptr;
  void* d_red = nullptr;
  void* d_canon = nullptr;        // (W+1) canonical XYZZ window sums
  uint8_t* h_pinned = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;  // around the bucket-accumulate kernel
  hipEvent_t ev_sorted = nullptr;           // after the sort + task cut (before the accumulate gate)
  float last_accum_ms = 0.f;
  uint32_t launched_n = 0;
  size_t out_bytes = 0;
  // scheduling (set by the prover before msm_launch; defaults = unconstrained)
  hipEvent_t accum_gate = nullptr;   // the accumulate kernel waits for this event (nullptr: none)
  uint32_t waves_per_simd = 0;       // persistent accumulate grid, 0 = the kernel's full occupancy
  hipEvent_t trace_ev[8] = {};       // G16_TRACE_HOST: stage boundaries inside msm_launch
};

struct U256 { uint32_t v[8]; };

static constexpr uint32_t kSegLenDefault = 16;   // buckets per reduce segment (G16_SEG_LEN overrides, sweeps)
// `dense` (the H-MSM): its reduce is the exposed tail of the proof, so shorter segments (more lanes, shorter
// chains) pay; the witness MSMs reduce while the H-MSM accumulates, where extra VALU work only competes.
inline uint32_t msm_seg_len(bool dense = false) {
  struct Cfg { uint32_t v, vd; };
  static const Cfg cfg = [] {   // thread-safe one-time initialisation (two host threads may prove on two handles)
    Cfg c;
    const char* e = getenv("G16_SEG_LEN");
    c.v = e ? (uint32_t)atoi(e) : kSegLenDefault;
    if (c.v < 1) c.v = 1;
    if (c.v > 64) c.v = 64;
    const char* ed = getenv("G16_SEG_LEN_DENSE");
    c.vd = ed ? (uint32_t)atoi(ed) : (e ? c.v : 8u);   // measured on the H-MSM: reduce + tree 0.62 ms at 16, 0.55 at 8, 0.80 at 4
    if (c.vd < 1) c.vd = 1;
    if (c.vd > 64) c.vd = 64;
    return c;
  }();
  return dense ? cfg.vd : cfg.v;
}

__device__ __forceinline__ uint32_t msm_extract(const uint32_t s[8], int pos, int c) {
  const int word = pos >> 5, off = pos & 31;
  if (word >= 8) return 0;
  uint64_t v = s[word];
  if (word + 1 < 8) v |= (uint64_t)s[word + 1] << 32;
  return (uint32_t)(v >> off) & ((1u << c) - 1);
}

// Loads scalar, adds K; returns true when the scalar is exactly 1.
__device__ __forceinline__ bool msm_load_scalar(const Fr* __restrict__ scalars, const uint32_t* __restrict__ src,
                                                uint32_t i, const U256& K, uint32_t s[8]) {
  const Fr x = scalars[src ? src[i] : i];
  uint32_t hi = 0;
#pragma unroll
  for (int k = 1; k < 8; k++) hi |= x.v[k];
  const bool one = (hi == 0 && x.v[0] == 1);
  uint64_t cy = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    cy += (uint64_t)x.v[k] + K.v[k];
    s[k] = (uint32_t)cy;
    cy >>= 32;
  }
  return one;
}

// digit of window j -> key (or 0xffffffff when the digit is zero) and sign
__device__ __forceinline__ uint32_t msm_key(const uint32_t s[8], int j, int c, int W, uint32_t B, uint32_t& neg) {
  const uint32_t e = msm_extract(s, j * c, c);
  int32_t d = (j == W - 1) ? (int32_t)e : (int32_t)e - (int32_t)B;
  neg = d < 0 ? 1u : 0u;
  const uint32_t mag = d < 0 ? (uint32_t)(-d) : (uint32_t)d;
  return mag == 0 ? 0xffffffffu : (uint32_t)j * B + (mag - 1);
}

// Digit codes, row-major: dig[r*ne + e] = bucket (|digit|-1) | sign<<31, or kSkip (ne = pf*n entries per row).
// Row W is the "ones" pseudo-window: scalars equal to 1 are spread over its buckets by point index.
static constexpr uint32_t kSkip = 0x7fffffffu;

static __global__ __launch_bounds__(256) void msm_digits_kernel(const Fr* __restrict__ scalars,
                                                         const uint32_t* __restrict__ src, uint32_t n,
                                                         int c, int Ws, int W, uint32_t pf, U256 K,
                                                         uint32_t* __restrict__ dig) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t s[8];
  const bool one = msm_load_scalar(scalars, src, i, K, s);
  const uint32_t B = 1u << (c - 1);
  const size_t ne = (size_t)pf * n;
  // scalar window j = k W + r -> row r, entry k n + i (the entry index IS the index into the base table)
  for (uint32_t k = 0; k < pf; k++) {
    for (int r = 0; r < W; r++) {
      const int j = (int)k * W + r;
      uint32_t code = kSkip;
      if (j < Ws && !one) {
        uint32_t neg;
        const uint32_t key = msm_key(s, j, c, Ws, B, neg);
        if (key != 0xffffffffu) code = (key - (uint32_t)j * B) | (neg << 31);
      }
      dig[(size_t)r * ne + (size_t)k * n + i] = code;
    }
    dig[(size_t)W * ne + (size_t)k * n + i] = (one && k == 0) ? (i & (B - 1)) : kSkip;
  }
}

// Counting sort without global atomics: workgroup (window j, chunk) histograms its slice of row j
// in LDS (MODE 0, writes hist[j][chunk][*]) and later scatters it with LDS cursors preloaded with
// the exclusive start offsets (MODE 1).
template <int MODE>
static __global__ __launch_bounds__(1024) void msm_sort_kernel(const uint32_t* __restrict__ dig, uint32_t n,
                                                        uint32_t B, uint32_t chunks, uint32_t per,
                                                        uint32_t* __restrict__ hist,
                                                        uint32_t* __restrict__ sorted) {
  extern __shared__ uint32_t lds[];
  const uint32_t j = blockIdx.x / chunks, chunk = blockIdx.x % chunks;
  uint32_t* __restrict__ h = hist + ((size_t)j * chunks + chunk) * B;
  for (uint32_t b = threadIdx.x; b < B; b += 1024) lds[b] = MODE ? h[b] : 0u;
  __syncthreads();
  const uint32_t lo = chunk * per;
  const uint32_t hi = (lo + per < n) ? lo + per : n;
  const uint32_t* __restrict__ row = dig + (size_t)j * n;
  for (uint32_t i = lo + threadIdx.x; i < hi; i += 1024) {
    const uint32_t d = row[i];
    if (d == kSkip) continue;
    const uint32_t pos = atomicAdd(&lds[d & 0x7fffffffu], 1u);
    if (MODE) sorted[pos] = i | (d & 0x80000000u);
  }
  if (!MODE) {
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < B; b += 1024) h[b] = lds[b];
  }
}

// cnt[key] = sum over chunks of hist[j][chunk][b]
static __global__ __launch_bounds__(256) void msm_hist_sum_kernel(const uint32_t* __restrict__ hist, uint32_t nb,
                                                           uint32_t B, uint32_t chunks,
                                                           uint32_t* __restrict__ cnt) {
  const uint32_t key = blockIdx.x * blockDim.x + threadIdx.x;
  if (key >= nb) return;
  const uint32_t j = key / B, b = key % B;
  uint32_t s = 0;
  for (uint32_t k = 0; k < chunks; k++) s += hist[((size_t)j * chunks + k) * B + b];
  cnt[key] = s;
}
// hist[j][chunk][b] <- off[key] + sum_{chunk' < chunk} hist[j][chunk'][b]
static __global__ __launch_bounds__(256) void msm_hist_start_kernel(uint32_t* __restrict__ hist, uint32_t nb,
                                                             uint32_t B, uint32_t chunks,
                                                             const uint32_t* __restrict__ off) {
  const uint32_t key = blockIdx.x * blockDim.x + threadIdx.x;
  if (key >= nb) return;
  const uint32_t j = key / B, b = key % B;
  uint32_t run = off[key];
  for (uint32_t k = 0; k < chunks; k++) {
    uint32_t* p = &hist[((size_t)j * chunks + k) * B + b];
    const uint32_t v = *p;
    *p = run;
    run += v;
  }
}

// Task length by bucket key: graded, so the END of the task queue is made of short tasks.  A persistent lane
// spends ~20 us per addition, i.e. ~0.7 ms on a 32-entry task; with uniform tasks the last ~0.7 ms of the
// kernel run at falling occupancy (lanes that found the queue empty wait for the stragglers).  The last
// 3/16 of the keys get half-length tasks and the last 1/16 quarter-length ones (floor 4): the drain shrinks
// 4x for ~30 % more task partials.
struct MsmTaskGrade { uint32_t len, split1, split2; };
__host__ __device__ __forceinline__ uint32_t msm_task_len_at(const MsmTaskGrade g, uint32_t key) {
  uint32_t l = g.len;
  if (key >= g.split1) l >>= 1;
  if (key >= g.split2) l >>= 1;
  return l < 4u ? 4u : l;
}
inline MsmTaskGrade msm_task_grade(uint32_t task_len, uint32_t rows, uint32_t B) {
  // graded over the digit rows only; the ones pseudo-row (keys >= rows * B) always gets the shortest tasks
  const uint32_t nk = rows * B;
  // measured (r01): the accumulate kernels get 7 % (G1) to 35 % (G2) shorter, but the lane-per-bucket combine
  // pass then walks up to 4x more partials in its last buckets, and throughput mode loses 10 %: off by default
  static const bool graded = getenv("G16_GRADED_TASKS") != nullptr;
  if (!graded) return MsmTaskGrade{task_len, 0xffffffffu, 0xffffffffu};
  return MsmTaskGrade{task_len, nk - 3 * (nk / 16), nk - nk / 16};
}

// Exclusive scans off = scan(cnt), toff = scan(ceil(cnt/task_len)) in three launches:
// per-tile sums (2048 counters per workgroup) -> one workgroup scans the tile sums -> per-tile
// local scan + tile offset.
static constexpr uint32_t kScanTile = 2048;   // 256 threads x 8 counters

static __global__ __launch_bounds__(256) void msm_scan_tiles_kernel(const uint32_t* __restrict__ cnt, uint32_t nb,
                                                             MsmTaskGrade tg, uint32_t* __restrict__ tile_a,
                                                             uint32_t* __restrict__ tile_b,
                                                             uint32_t* __restrict__ tile_c) {
  __shared__ uint32_t sh_a[256], sh_b[256], sh_c[256];
  const uint32_t tid = threadIdx.x, base = blockIdx.x * kScanTile + tid * 8;
  uint32_t sa = 0, sb = 0, sc = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const uint32_t v = (base + k < nb) ? cnt[base + k] : 0u;
    const uint32_t tl = msm_task_len_at(tg, base + k);
    sa += v;
    sb += (v + tl - 1) / tl;
    sc += v / tl;
  }
  sh_a[tid] = sa; sh_b[tid] = sb; sh_c[tid] = sc;
  __syncthreads();
  for (uint32_t d = 128; d > 0; d >>= 1) {
    if (tid < d) { sh_a[tid] += sh_a[tid + d]; sh_b[tid] += sh_b[tid + d]; sh_c[tid] += sh_c[tid + d]; }
    __syncthreads();
  }
  if (tid == 0) { tile_a[blockIdx.x] = sh_a[0]; tile_b[blockIdx.x] = sh_b[0]; tile_c[blockIdx.x] = sh_c[0]; }
}

// ntiles <= 1024 * chunk; one workgroup; writes exclusive tile offsets in place and the totals
static __global__ __launch_bounds__(1024) void msm_scan_top_kernel(uint32_t* __restrict__ tile_a,
                                                            uint32_t* __restrict__ tile_b,
                                                            uint32_t* __restrict__ tile_c, uint32_t ntiles,
                                                            uint32_t* __restrict__ total_a,
                                                            uint32_t* __restrict__ total_b,
                                                            uint32_t* __restrict__ total_c) {
  __shared__ uint32_t sh_a[1024], sh_b[1024], sh_c[1024];
  const uint32_t tid = threadIdx.x;
  const uint32_t chunk = (ntiles + 1023) / 1024;
  const uint32_t lo = tid * chunk, hi = (lo + chunk < ntiles) ? lo + chunk : ntiles;
  uint32_t sa = 0, sb = 0, sc = 0;
  for (uint32_t k = lo; k < hi; k++) { sa += tile_a[k]; sb += tile_b[k]; sc += tile_c[k]; }
  sh_a[tid] = sa; sh_b[tid] = sb; sh_c[tid] = sc;
  __syncthreads();
  for (uint32_t d = 1; d < 1024; d <<= 1) {
    uint32_t va = 0, vb = 0, vc = 0;
    if (tid >= d) { va = sh_a[tid - d]; vb = sh_b[tid - d]; vc = sh_c[tid - d]; }
    __syncthreads();
    sh_a[tid] += va; sh_b[tid] += vb; sh_c[tid] += vc;
    __syncthreads();
  }
  uint32_t pa = sh_a[tid] - sa, pb = sh_b[tid] - sb, pc = sh_c[tid] - sc;
  for (uint32_t k = lo; k < hi; k++) {
    const uint32_t va = tile_a[k], vb = tile_b[k], vc = tile_c[k];
    tile_a[k] = pa; tile_b[k] = pb; tile_c[k] = pc;
    pa += va; pb += vb; pc += vc;
  }
  if (tid == 1023) { *total_a = sh_a[1023]; *total_b = sh_b[1023]; *total_c = sh_c[1023]; }
}

static __global__ __launch_bounds__(256) void msm_scan_apply_kernel(const uint32_t* __restrict__ cnt, uint32_t nb,
                                                             MsmTaskGrade tg,
                                                             const uint32_t* __restrict__ tile_a,
                                                             const uint32_t* __restrict__ tile_b,
                                                             const uint32_t* __restrict__ tile_c,
                                                             uint32_t* __restrict__ off,
                                                             uint32_t* __restrict__ toff,
                                                             uint32_t* __restrict__ foff) {
  __shared__ uint32_t sh_a[256], sh_b[256], sh_c[256];
  const uint32_t tid = threadIdx.x, base = blockIdx.x * kScanTile + tid * 8;
  uint32_t v[8], sa = 0, sb = 0, sc = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    v[k] = (base + k < nb) ? cnt[base + k] : 0u;
    const uint32_t tl = msm_task_len_at(tg, base + k);
    sa += v[k];
    sb += (v[k] + tl - 1) / tl;
    sc += v[k] / tl;
  }
  sh_a[tid] = sa; sh_b[tid] = sb; sh_c[tid] = sc;
  __syncthreads();
  for (uint32_t d = 1; d < 256; d <<= 1) {
    uint32_t va = 0, vb = 0, vc = 0;
    if (tid >= d) { va = sh_a[tid - d]; vb = sh_b[tid - d]; vc = sh_c[tid - d]; }
    __syncthreads();
    sh_a[tid] += va; sh_b[tid] += vb; sh_c[tid] += vc;
    __syncthreads();
  }
  uint32_t pa = tile_a[blockIdx.x] + sh_a[tid] - sa, pb = tile_b[blockIdx.x] + sh_b[tid] - sb,
           pc = tile_c[blockIdx.x] + sh_c[tid] - sc;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    if (base + k < nb) {
      off[base + k] = pa;
      toff[base + k] = pb;
      foff[base + k] = pc;
    }
    pa += v[k];
    const uint32_t tl = msm_task_len_at(tg, base + k);
    pb += (v[k] + tl - 1) / tl;
    pc += v[k] / tl;
  }
}

// task descriptor = (first sorted entry, entry count); the tasks of one bucket have consecutive ids (their
// partial sums are consecutive for the combine pass).  The work QUEUE is a permutation of the tasks: every
// full-length task first, the remainders (one per bucket at most, shorter) after them -- the 64 lanes of a
// wavefront then start and finish their full tasks in the same iteration, so the flush / start / request code
// of the accumulate loop runs once per task instead of in nearly every iteration, and the queue ends with its
// shortest tasks (a shorter drain).  qdesc[q] = (first entry, count, task id, -).
static constexpr uint32_t kRemClasses = 32;   // remainder tasks are queued by relative length, longest class first
__device__ __forceinline__ uint32_t msm_rem_class(uint32_t len, uint32_t task_len) {
  return (len * kRemClasses) / task_len;      // len < task_len -> 0 .. kRemClasses - 1
}

// class_total[c] = number of remainder tasks (cnt % task_len != 0) of relative-length class c
static __global__ __launch_bounds__(256) void msm_rem_count_kernel(const uint32_t* __restrict__ cnt, uint32_t nb,
                                                            MsmTaskGrade tg, uint32_t* __restrict__ class_total) {
  __shared__ uint32_t h[kRemClasses];
  if (threadIdx.x < kRemClasses) h[threadIdx.x] = 0;
  __syncthreads();
  const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < nb) {
    const uint32_t tl = msm_task_len_at(tg, b), r = cnt[b] % tl;
    if (r) atomicAdd(&h[msm_rem_class(r, tl)], 1u);
  }
  __syncthreads();
  if (threadIdx.x < kRemClasses && h[threadIdx.x]) atomicAdd(&class_total[threadIdx.x], h[threadIdx.x]);
}

static __global__ __launch_bounds__(256) void msm_task_fill_kernel(const uint32_t* __restrict__ off,
                                                            const uint32_t* __restrict__ toff,
                                                            const uint32_t* __restrict__ foff, uint32_t nb,
                                                            MsmTaskGrade tg, uint2* __restrict__ task_desc,
                                                            uint4* __restrict__ qdesc,
                                                            const uint32_t* __restrict__ class_total,
                                                            uint32_t* __restrict__ class_cursor) {
  // remainders: after all the full tasks, by relative-length class (longest first) so that the lanes of a
  // wavefront hold remainders of (nearly) equal length; inside a class the order is whatever the atomics give
  __shared__ uint32_t h[kRemClasses], base[kRemClasses];
  if (threadIdx.x < kRemClasses) h[threadIdx.x] = 0;
  __syncthreads();
  const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t task_len = 1, start = 0, left = 0, rem = 0, cls = 0, rank = 0;
  if (b < nb) {
    task_len = msm_task_len_at(tg, b);
    start = off[b];
    left = off[b + 1] - start;
    rem = left % task_len;
    if (rem) {
      cls = msm_rem_class(rem, task_len);
      rank = atomicAdd(&h[cls], 1u);
    }
  }
  __syncthreads();
  if (threadIdx.x < kRemClasses) {
    const uint32_t c = threadIdx.x;
    uint32_t before = foff[nb];                              // all full tasks, then the longer classes
    for (uint32_t k = c + 1; k < kRemClasses; k++) before += class_total[k];
    base[c] = h[c] ? before + atomicAdd(&class_cursor[c], h[c]) : 0u;
  }
  __syncthreads();
  if (b >= nb) return;
  uint32_t fq = foff[b];
  for (uint32_t t = toff[b], e = toff[b + 1]; t < e; t++) {
    const uint32_t len = left < task_len ? left : task_len;
    task_desc[t] = make_uint2(start, len);
    qdesc[len == task_len ? fq++ : base[cls] + rank] = make_uint4(start, len, t, 0u);
    start += len;
    left -= len;
  }
}

// Bucket accumulation: persistent wavefronts over a work queue.  A wavefront pulls chunks of
// kTaskChunk consecutive tasks from a global counter; a lane that is about to finish its task is handed
// the chunk's next one (ballot + prefix popcount), so lanes stay busy instead of idling behind the
// longest bucket, and there is no grid-quantisation tail.  Every loop iteration is one REAL mixed
// addition for a busy lane: the descriptor of the next task is requested while the lane does the last
// addition of the current one (its load latency hides behind that addition), and a task starts by
// loading its first point straight into the accumulator (ZZ = ZZZ = 1) and adding the second in the
// same iteration -- a task of L entries costs L - 1 iterations (1 when L = 1).
// Exit: the queue counter passes `total` (every wave sees it) and no lane holds or awaits a task.
static constexpr uint32_t kTaskChunk = 64;

template <class F>
__global__ __launch_bounds__(64, F::kAccumWavesPerSimd) void msm_accumulate_kernel(const PackedAffine<F>* __restrict__ bases,
                                                            const uint32_t* __restrict__ sorted,
                                                            const uint32_t* __restrict__ toff, uint32_t nb,
                                                            const uint4* __restrict__ qdesc,
                                                            uint32_t* __restrict__ queue,
                                                            uint32_t* __restrict__ redo,
                                                            XYZZ<F>* __restrict__ partial) {
  const uint32_t total = toff[nb];
  const uint32_t lane = threadIdx.x;
  const unsigned long long lt_mask = (1ull << lane) - 1;
  constexpr uint32_t kNone = 0xffffffffu;
  uint32_t next = 0, chunk_end = 0;     // wave-uniform: the chunk being handed out
  bool exhausted = false;               // wave-uniform: the queue has no more chunks
  uint32_t my_task = kNone, pending = kNone, cur = 0, end = 0;
  bool bad = false;                     // the running task met an exceptional case: its sum is redone
  uint4 desc = make_uint4(0, 0, 0, 0);
  XYZZ<F> acc;
  x29_set_inf(acc);
  // (Staggering the wavefronts of a SIMD by HW_ID.WAVE_ID so that their gathers do not coincide was measured:
  // no change for G1 -- the kernel is issue-bound, ~5.5 cycles per VALU instruction with 13 % memory wait.)
  for (;;) {
    // 1. lanes whose task is complete write its partial sum
    if (my_task != kNone && cur == end) {
      partial[my_task] = acc;
      if (bad) redo[atomicAdd(&queue[1], 1u)] = my_task;   // at most one entry per task: cannot overflow
      my_task = kNone;
    }
    // 2. idle lanes whose next descriptor has arrived start it: the first entry IS the accumulator
    if (my_task == kNone && pending != kNone) {
      my_task = desc.z;     // the task id (where its partial sum goes); `pending` was its queue position
      pending = kNone;
      cur = desc.x;
      end = desc.x + desc.y;
      const uint32_t idx = sorted[cur++];
      Affine<F> p;
      a29_unpack(p, bases[idx & 0x7fffffffu]);
      if (idx >> 31) a29_neg(p);
      acc.x = p.x; acc.y = p.y; acc.zz = F::one(); acc.zzz = F::one();
      bad = false;
    }
    // 3. lanes with at most one addition left (or none at all) ask for their next task
    const bool want = pending == kNone && !exhausted && (my_task == kNone || end - cur <= 1u);
    const unsigned long long m = __ballot(want);
    if (m) {
      if (next == chunk_end) {   // wave-uniform: pull the next chunk (`exhausted` is false here: m != 0)
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(queue, kTaskChunk);
        base = __shfl(base, 0, 64);
        if (base >= total) {
          exhausted = true;
        } else {
          next = base;
          chunk_end = base + kTaskChunk < total ? base + kTaskChunk : total;
        }
      }
      if (!exhausted) {
        if (want) {
          const uint32_t cand = next + (uint32_t)__popcll(m & lt_mask);
          if (cand < chunk_end) {
            pending = cand;
            desc = qdesc[cand];
          }
        }
        next += (uint32_t)__popcll(m);
        if (next > chunk_end) next = chunk_end;
      }
    }
    // 4. done when no lane holds or awaits a task and the queue is empty
    if (exhausted && __ballot(my_task != kNone || pending != kNone) == 0) break;
    // 5. one mixed addition per lane that has entries left.  (Requesting the NEXT point before this addition --
    //    software prefetch, 18 more VGPRs -- was measured twice, before and after the queue was ordered (two-stage
    //    pipeline of packed point + next index): 2.77 -> 2.89 ms on the H-MSM.  Four wavefronts per SIMD already
    //    hide the two dependent loads; the extra moves and registers only cost issue slots.)
    if (my_task != kNone && cur != end) {
      const uint32_t idx = sorted[cur++];
      Affine<F> p;
      a29_unpack(p, bases[idx & 0x7fffffffu]);
      if (idx >> 31) a29_neg(p);
      bad |= x29_madd_fast(acc, p);
    }
  }
}

// The flagged tasks again, with the complete addition (doubling, cancellation, infinity): a handful per
// proof at most, one lane each.
template <class F>
__global__ __launch_bounds__(64) void msm_redo_kernel(const PackedAffine<F>* __restrict__ bases,
                                                      const uint32_t* __restrict__ sorted,
                                                      const uint2* __restrict__ task_desc,
                                                      const uint32_t* __restrict__ queue,
                                                      const uint32_t* __restrict__ redo,
                                                      XYZZ<F>* __restrict__ partial) {
  const uint32_t count = queue[1];
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += gridDim.x * blockDim.x) {
    const uint32_t t = redo[i];
    const uint2 d = task_desc[t];
    XYZZ<F> acc;
    x29_set_inf(acc);
    for (uint32_t e = d.x; e < d.x + d.y; e++) {
      const uint32_t idx = sorted[e];
      Affine<F> p;
      a29_unpack(p, bases[idx & 0x7fffffffu]);
      if (idx >> 31) a29_neg(p);
      x29_madd(acc, p);
    }
    partial[t] = acc;
  }
}

template <class F>
__device__ __forceinline__ void msm_mul_small(XYZZ<F>& r, const XYZZ<F>& p, uint32_t k) {
  x29_set_inf(r);
  if (k == 0) return;
  for (int i = 31 - __clz(k); i >= 0; i--) {
    x29_dbl(r);
    if ((k >> i) & 1) x29_add(r, p);
  }
}

static constexpr uint32_t kLightTasks = 6;   // buckets with more partials take the wavefront path (floor; see msm_light_max)
// With window precomputation a bucket collects pf times more entries, hence more task partials: the lane-per-
// bucket path takes up to ~2x the average (a few sequential adds on an otherwise idle machine), the wavefront
// path only the outliers.
inline uint32_t msm_light_max(const MsmInstance& m) {
  const uint64_t tasks = ((uint64_t)m.n * (uint32_t)m.Ws) / m.task_len;
  const uint64_t avg = tasks / ((uint64_t)m.W * m.nbuckets) + 1;
  uint64_t v = 2 * avg + 4;
  static const bool graded = getenv("G16_GRADED_TASKS") != nullptr;
  if (graded) v *= 4;   // graded task lengths: up to 4x more partials in the last buckets
  if (v < kLightTasks) v = kLightTasks;
  if (v > 48) v = 48;
  return (uint32_t)v;
}

template <class F> __device__ __forceinline__ XYZZ<F> xyzz_shfl_down(const XYZZ<F>& p, int delta) {
  XYZZ<F> r;
  constexpr int NW = sizeof(XYZZ<F>) / 4;
  const uint32_t* s = reinterpret_cast<const uint32_t*>(&p);
  uint32_t* d = reinterpret_cast<uint32_t*>(&r);
#pragma unroll
  for (int i = 0; i < NW; i++) d[i] = __shfl_down(s[i], delta, 64);
  return r;
}

// bsum[b] = sum of the task partials of bucket b (light buckets); heavy buckets are queued.
template <class F>
__global__ __launch_bounds__(64) void msm_combine_light_kernel(const XYZZ<F>* __restrict__ partial,
                                                               const uint32_t* __restrict__ toff, uint32_t nb,
                                                               XYZZ<F>* __restrict__ bsum,
                                                               uint32_t* __restrict__ heavy, uint32_t max_heavy,
                                                               uint32_t light_max) {
  const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nb) return;
  const uint32_t t0 = toff[b], t1 = toff[b + 1];
  XYZZ<F> acc;
  x29_set_inf(acc);
  if (t1 - t0 > light_max) {
    const uint32_t k = atomicAdd(&heavy[0], 1u);
    if (k < max_heavy) heavy[1 + k] = b;   // cannot overflow: max_heavy >= max_tasks / kLightTasks
    return;                                // bsum[b] written by the heavy kernel
  }
  for (uint32_t t = t0; t < t1; t++) {
    const XYZZ<F> s = partial[t];
    x29_add(acc, s);
  }
  bsum[b] = acc;
}

// One wavefront per heavy bucket: lanes stride over the partials, then a 6-step shuffle tree.
template <class F>
__global__ __launch_bounds__(64) void msm_combine_heavy_kernel(const XYZZ<F>* __restrict__ partial,
                                                               const uint32_t* __restrict__ toff,
                                                               XYZZ<F>* __restrict__ bsum,
                                                               const uint32_t* __restrict__ heavy, uint32_t max_heavy) {
  uint32_t count = heavy[0];
  if (count > max_heavy) count = max_heavy;
  const uint32_t lane = threadIdx.x;
  for (uint32_t h = blockIdx.x; h < count; h += gridDim.x) {
    const uint32_t b = heavy[1 + h];
    const uint32_t t0 = toff[b], t1 = toff[b + 1];
    XYZZ<F> acc;
    x29_set_inf(acc);
    for (uint32_t t = t0 + lane; t < t1; t += 64) {
      const XYZZ<F> s = partial[t];
      x29_add(acc, s);
    }
    for (int d = 32; d >= 1; d >>= 1) {
      const XYZZ<F> q = xyzz_shfl_down(acc, d);
      x29_add(acc, q);
    }
    if (lane == 0) bsum[b] = acc;
  }
}

// seg[j*nseg + g] = sum_{bi in segment g of window j} (bi+1) * S_bi   (j < W);  j == W: sum S_bi
template <class F>
__global__ __launch_bounds__(64) void msm_bucket_reduce_kernel(const XYZZ<F>* __restrict__ bsum, uint32_t B,
                                                               uint32_t nseg, uint32_t W, uint32_t seg_len,
                                                               XYZZ<F>* __restrict__ seg) {
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= (W + 1) * nseg) return;
  const uint32_t j = tid / nseg, g = tid % nseg;
  const bool plain = (j == W);   // the "ones" pseudo-window: plain sum of its buckets
  const uint32_t lo = g * seg_len;
  const uint32_t hi = (lo + seg_len < B) ? lo + seg_len : B;
  XYZZ<F> run, acc;
  x29_set_inf(run);
  x29_set_inf(acc);
  for (uint32_t bi = hi; bi-- > lo;) {
    const XYZZ<F> s = bsum[(size_t)j * B + bi];
    x29_add(run, s);
    if (!plain) x29_add(acc, run);
  }
  if (plain) {
    acc = run;
  } else if (lo != 0) {
    XYZZ<F> m;
    msm_mul_small(m, run, lo);
    x29_add(acc, m);
  }
  seg[tid] = acc;
}

// out[j*nout + blk] = sum of in[j*nin + blk*64 .. +64)
template <class F>
__global__ __launch_bounds__(64) void msm_wave_reduce_kernel(const XYZZ<F>* __restrict__ in, uint32_t nin,
                                                             XYZZ<F>* __restrict__ out, uint32_t nout) {
  const uint32_t j = blockIdx.y, blk = blockIdx.x, lane = threadIdx.x;
  const uint32_t i = blk * 64 + lane;
  XYZZ<F> p;
  if (i < nin) p = in[(size_t)j * nin + i];
  else x29_set_inf(p);
  for (int d = 32; d >= 1; d >>= 1) {
    const XYZZ<F> q = xyzz_shfl_down(p, d);
    x29_add(p, q);
  }
  if (lane == 0) out[(size_t)j * nout + blk] = p;
}

// bases: canonical affine image (zkey bytes) -> lazy 9x29 representation, once at create
template <class F>
__global__ __launch_bounds__(256) void msm_convert_bases_kernel(const Affine<typename F::CanonOps>* __restrict__ in,
                                                                PackedAffine<F>* __restrict__ out, uint32_t n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Affine<F> r;
  a29_from_canon<F, typename F::CanonOps>(r, in[i]);
  PackedAffine<F> pk;
  a29_pack(pk, r);
  out[i] = pk;
}
// Window precomputation (once at create): out[i] = 2^ndbl * in[i], affine, on the canonical field (exact
// arithmetic, fp.cuh / ec.cuh; one Fermat inversion per point -- create-time only).
template <class FC>
__global__ __launch_bounds__(256) void msm_precompute_kernel(const Affine<FC>* __restrict__ in,
                                                             Affine<FC>* __restrict__ out, uint32_t n, int ndbl) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  XYZZ<FC> r;
  xyzz_dbl_affine(r, in[i]);
  for (int k = 1; k < ndbl; k++) xyzz_dbl(r);
  Affine<FC> a;
  xyzz_to_affine(a, r);
  out[i] = a;
}
// window sums: lazy -> canonical XYZZ (what the host folds)
template <class F>
__global__ __launch_bounds__(64) void msm_to_canon_kernel(const XYZZ<F>* __restrict__ in,
                                                          XYZZ<typename F::CanonOps>* __restrict__ out, uint32_t n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  XYZZ<typename F::CanonOps> r;
  x29_to_canon<F, typename F::CanonOps>(r, in[i]);
  out[i] = r;
}

// ------------------------------------------------------------------ host side (per curve)
inline void msm_make_K(int c, int W, U256& K) {
  for (int i = 0; i < 8; i++) K.v[i] = 0;
  for (int j = 0; j + 1 < W; j++) {
    const int bit = c * j + c - 1;
    if (bit < 256) K.v[bit >> 5] |= 1u << (bit & 31);
  }
}

// Enqueues the whole MSM on `st` (no host synchronisation); results land in ws->h_pinned.
template <class F>
int msm_launch_t(const MsmInstance& m, MsmWorkspace* ws, const Fr* d_scalars, hipStream_t st) {
  using PT = XYZZ<F>;
  const uint32_t W = (uint32_t)m.W, B = m.nbuckets, WT = W + 1, nb = WT * B;  // rows + the ones window
  const uint32_t ne = m.n_ext;                                                // entries per row
  ws->last_accum_ms = 0.f;
  ws->launched_n = m.n;
  using CPT = XYZZ<typename F::CanonOps>;
  ws->out_bytes = (size_t)WT * sizeof(CPT);
  if (m.n == 0) return G16_OK;
  const uint32_t seg_len = msm_seg_len(m.dense);
  const uint32_t nseg = (B + seg_len - 1) / seg_len;
  U256 K;
  msm_make_K(m.c, m.Ws, K);
  const uint32_t nblk = (m.n + 255) / 256;
  const uint32_t chunks = ws->chunks, per = (ne + chunks - 1) / chunks;
  const size_t lds_bytes = (size_t)B * 4;
  {
    // > 64 KiB of dynamic LDS needs the opt-in (c = 16: 128 KiB histogram).  The attribute is set once PER DEVICE
    // (one process may hold handles on several GPUs: the Node host of BASELINE config 4) and per template
    // instantiation (the kernels are static to each translation unit).
    static std::atomic<uint64_t> attr_mask{0};
    int dev = 0;
    G16_HIP(hipGetDevice(&dev));
    const uint64_t bit = 1ull << (dev & 63);
    if (!(attr_mask.load(std::memory_order_acquire) & bit)) {
      G16_HIP(hipFuncSetAttribute((const void*)msm_sort_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      G16_HIP(hipFuncSetAttribute((const void*)msm_sort_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      attr_mask.fetch_or(bit, std::memory_order_release);
    }
  }
  static const bool trace = getenv("G16_TRACE_HOST") != nullptr;
  auto mark = [&](int k) {
    if (!trace) return;
    if (!ws->trace_ev[k]) (void)hipEventCreate(&ws->trace_ev[k]);
    (void)hipEventRecord(ws->trace_ev[k], st);
  };
  msm_digits_kernel<<<nblk, 256, 0, st>>>(d_scalars, m.d_src, m.n, m.c, m.Ws, m.W, m.pf, K, ws->d_dig);
  mark(0);
  msm_sort_kernel<0><<<WT * chunks, 1024, lds_bytes, st>>>(ws->d_dig, ne, B, chunks, per, ws->d_hist, nullptr);
  mark(1);
  msm_hist_sum_kernel<<<(nb + 255) / 256, 256, 0, st>>>(ws->d_hist, nb, B, chunks, ws->d_cnt);
  const uint32_t ntiles = (nb + kScanTile - 1) / kScanTile;
  const MsmTaskGrade tg = msm_task_grade(m.task_len, W, B);
  msm_scan_tiles_kernel<<<ntiles, 256, 0, st>>>(ws->d_cnt, nb, tg, ws->d_tile_a, ws->d_tile_b, ws->d_tile_c);
  msm_scan_top_kernel<<<1, 1024, 0, st>>>(ws->d_tile_a, ws->d_tile_b, ws->d_tile_c, ntiles, ws->d_off + nb, ws->d_toff + nb,
                                          ws->d_foff + nb);
  msm_scan_apply_kernel<<<ntiles, 256, 0, st>>>(ws->d_cnt, nb, tg, ws->d_tile_a, ws->d_tile_b, ws->d_tile_c, ws->d_off,
                                                ws->d_toff, ws->d_foff);
  msm_hist_start_kernel<<<(nb + 255) / 256, 256, 0, st>>>(ws->d_hist, nb, B, chunks, ws->d_off);
  mark(2);
  msm_sort_kernel<1><<<WT * chunks, 1024, lds_bytes, st>>>(ws->d_dig, ne, B, chunks, per, ws->d_hist, ws->d_sorted);
  mark(3);
  G16_HIP(hipMemsetAsync(ws->d_class, 0, 2 * 32 * 4, st));
  msm_rem_count_kernel<<<(nb + 255) / 256, 256, 0, st>>>(ws->d_cnt, nb, tg, ws->d_class);
  msm_task_fill_kernel<<<(nb + 255) / 256, 256, 0, st>>>(ws->d_off, ws->d_toff, ws->d_foff, nb, tg, ws->d_task_desc,
                                                         ws->d_qdesc, ws->d_class, ws->d_class + 32);
  // upper bound on tasks: every non-empty bucket has <= 1 short task + entries/task_len full ones
  // (the shortest graded tasks hold task_len / 4 >= 4 entries; ones: <= n entries, covered)
  const uint64_t max_tasks = (uint64_t)nb + ((uint64_t)m.n * (uint32_t)m.Ws) / (m.task_len >= 16 ? m.task_len / 4 : 4);
  // persistent grid: as many wavefronts as the chip holds for this kernel (4/SIMD G1, 2/SIMD G2), fewer
  // when there is little work
  const uint32_t full_occ = sizeof(typename F::T) > sizeof(F29) ? 2 : 4;
  const uint32_t occ = (ws->waves_per_simd && ws->waves_per_simd < full_occ) ? ws->waves_per_simd : full_occ;
  uint64_t waves = (uint64_t)256 * 4 * occ;
  if (waves > (max_tasks + kTaskChunk - 1) / kTaskChunk) waves = (max_tasks + kTaskChunk - 1) / kTaskChunk;
  if (waves == 0) waves = 1;
  G16_HIP(hipMemsetAsync(ws->d_queue, 0, 8, st));
  G16_HIP(hipEventRecord(ws->ev_sorted, st));
  if (ws->accum_gate) G16_HIP(hipStreamWaitEvent(st, ws->accum_gate, 0));
  G16_HIP(hipEventRecord(ws->ev0, st));
  msm_accumulate_kernel<F><<<(unsigned)waves, 64, 0, st>>>((const PackedAffine<F>*)m.d_bases, ws->d_sorted, ws->d_toff, nb,
                                                            ws->d_qdesc, ws->d_queue, ws->d_redo, (PT*)ws->d_partial);
  G16_HIP(hipEventRecord(ws->ev1, st));
  msm_redo_kernel<F><<<64, 64, 0, st>>>((const PackedAffine<F>*)m.d_bases, ws->d_sorted, ws->d_task_desc, ws->d_queue,
                                        ws->d_redo, (PT*)ws->d_partial);
  G16_HIP(hipMemsetAsync(ws->d_heavy, 0, 4, st));
  mark(4);
  msm_combine_light_kernel<F><<<(nb + 63) / 64, 64, 0, st>>>((const PT*)ws->d_partial, ws->d_toff, nb,
                                                             (PT*)ws->d_bsum, ws->d_heavy, ws->max_heavy,
                                                             msm_light_max(m));
  msm_combine_heavy_kernel<F><<<1024, 64, 0, st>>>((const PT*)ws->d_partial, ws->d_toff, (PT*)ws->d_bsum,
                                                   ws->d_heavy, ws->max_heavy);
  mark(5);
  msm_bucket_reduce_kernel<F><<<(WT * nseg + 63) / 64, 64, 0, st>>>((const PT*)ws->d_bsum, B, nseg, W, seg_len,
                                                                   (PT*)ws->d_seg);
  mark(6);
  // tree: d_seg (nseg per window) -> ... -> 1 per window, ping-pong between d_red halves
  PT* cur = (PT*)ws->d_seg;
  uint32_t cnt = nseg;
  PT* bufs[2] = {(PT*)ws->d_red, (PT*)ws->d_red + (size_t)WT * ((nseg + 63) / 64)};
  int flip = 0;
  while (cnt > 1) {
    const uint32_t nout = (cnt + 63) / 64;
    msm_wave_reduce_kernel<F><<<dim3(nout, WT), 64, 0, st>>>(cur, cnt, bufs[flip], nout);
    cur = bufs[flip];
    flip ^= 1;
    cnt = nout;
  }
  G16_HIP(hipGetLastError());
  msm_to_canon_kernel<F><<<(WT + 63) / 64, 64, 0, st>>>(cur, (CPT*)ws->d_canon, WT);
  G16_HIP(hipGetLastError());
  G16_HIP(hipMemcpyAsync(ws->h_pinned, ws->d_canon, (size_t)WT * sizeof(CPT), hipMemcpyDeviceToHost, st));
  return G16_OK;
}

}  // namespace g16
