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
// Roofline note: ~3.4k VALU instructions (1.75k v_mad_u64_u32) per
// gathered point addition, 16 additions per 96 algorithmic bytes: integer-issue bound, not HBM bound.
#pragma once
#include <stdlib.h>

#include "ec29.cuh"
#include "internal.h"

namespace g16 {

struct MsmWorkspace {
  uint32_t max_entries = 0, max_buckets = 0, max_tasks = 0;
  uint32_t* d_cnt = nullptr;
  uint32_t* d_off = nullptr;
  uint32_t* d_toff = nullptr;
  uint32_t* d_sorted = nullptr;
  uint2* d_task_desc = nullptr;
  uint32_t* d_queue = nullptr;    // work-queue head of the accumulate kernel
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
  float last_accum_ms = 0.f;
  uint32_t launched_n = 0;
  size_t out_bytes = 0;
};

struct U256 { uint32_t v[8]; };

static constexpr uint32_t kSegLenDefault = 16;   // buckets per reduce segment (G16_SEG_LEN overrides, sweeps)
inline uint32_t msm_seg_len() {
  static uint32_t v = 0;
  if (!v) { const char* e = getenv("G16_SEG_LEN"); v = e ? (uint32_t)atoi(e) : kSegLenDefault; if (v < 1) v = 1; if (v > 64) v = 64; }
  return v;
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

// Digit codes, window-major: dig[j*n + i] = bucket (|digit|-1) | sign<<31, or kSkip.  Row W is the
// "ones" pseudo-window: scalars equal to 1 are spread over its buckets by point index.
static constexpr uint32_t kSkip = 0x7fffffffu;

static __global__ __launch_bounds__(256) void msm_digits_kernel(const Fr* __restrict__ scalars,
                                                         const uint32_t* __restrict__ src, uint32_t n,
                                                         int c, int W, U256 K, uint32_t* __restrict__ dig) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t s[8];
  const bool one = msm_load_scalar(scalars, src, i, K, s);
  const uint32_t B = 1u << (c - 1);
  for (int j = 0; j < W; j++) {
    uint32_t neg;
    const uint32_t key = msm_key(s, j, c, W, B, neg);
    dig[(size_t)j * n + i] = (one || key == 0xffffffffu) ? kSkip : ((key - (uint32_t)j * B) | (neg << 31));
  }
  dig[(size_t)W * n + i] = one ? (i & (B - 1)) : kSkip;
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

// Exclusive scans off = scan(cnt), toff = scan(ceil(cnt/task_len)) in three launches:
// per-tile sums (2048 counters per workgroup) -> one workgroup scans the tile sums -> per-tile
// local scan + tile offset.
static constexpr uint32_t kScanTile = 2048;   // 256 threads x 8 counters

static __global__ __launch_bounds__(256) void msm_scan_tiles_kernel(const uint32_t* __restrict__ cnt, uint32_t nb,
                                                             uint32_t task_len, uint32_t* __restrict__ tile_a,
                                                             uint32_t* __restrict__ tile_b) {
  __shared__ uint32_t sh_a[256], sh_b[256];
  const uint32_t tid = threadIdx.x, base = blockIdx.x * kScanTile + tid * 8;
  uint32_t sa = 0, sb = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const uint32_t v = (base + k < nb) ? cnt[base + k] : 0u;
    sa += v;
    sb += (v + task_len - 1) / task_len;
  }
  sh_a[tid] = sa; sh_b[tid] = sb;
  __syncthreads();
  for (uint32_t d = 128; d > 0; d >>= 1) {
    if (tid < d) { sh_a[tid] += sh_a[tid + d]; sh_b[tid] += sh_b[tid + d]; }
    __syncthreads();
  }
  if (tid == 0) { tile_a[blockIdx.x] = sh_a[0]; tile_b[blockIdx.x] = sh_b[0]; }
}

// ntiles <= 1024 * chunk; one workgroup; writes exclusive tile offsets in place and the totals
static __global__ __launch_bounds__(1024) void msm_scan_top_kernel(uint32_t* __restrict__ tile_a,
                                                            uint32_t* __restrict__ tile_b, uint32_t ntiles,
                                                            uint32_t* __restrict__ total_a,
                                                            uint32_t* __restrict__ total_b) {
  __shared__ uint32_t sh_a[1024], sh_b[1024];
  const uint32_t tid = threadIdx.x;
  const uint32_t chunk = (ntiles + 1023) / 1024;
  const uint32_t lo = tid * chunk, hi = (lo + chunk < ntiles) ? lo + chunk : ntiles;
  uint32_t sa = 0, sb = 0;
  for (uint32_t k = lo; k < hi; k++) { sa += tile_a[k]; sb += tile_b[k]; }
  sh_a[tid] = sa; sh_b[tid] = sb;
  __syncthreads();
  for (uint32_t d = 1; d < 1024; d <<= 1) {
    uint32_t va = 0, vb = 0;
    if (tid >= d) { va = sh_a[tid - d]; vb = sh_b[tid - d]; }
    __syncthreads();
    sh_a[tid] += va; sh_b[tid] += vb;
    __syncthreads();
  }
  uint32_t pa = sh_a[tid] - sa, pb = sh_b[tid] - sb;
  for (uint32_t k = lo; k < hi; k++) {
    const uint32_t va = tile_a[k], vb = tile_b[k];
    tile_a[k] = pa; tile_b[k] = pb;
    pa += va; pb += vb;
  }
  if (tid == 1023) { *total_a = sh_a[1023]; *total_b = sh_b[1023]; }
}

static __global__ __launch_bounds__(256) void msm_scan_apply_kernel(const uint32_t* __restrict__ cnt, uint32_t nb,
                                                             uint32_t task_len,
                                                             const uint32_t* __restrict__ tile_a,
                                                             const uint32_t* __restrict__ tile_b,
                                                             uint32_t* __restrict__ off,
                                                             uint32_t* __restrict__ toff) {
  __shared__ uint32_t sh_a[256], sh_b[256];
  const uint32_t tid = threadIdx.x, base = blockIdx.x * kScanTile + tid * 8;
  uint32_t v[8], sa = 0, sb = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    v[k] = (base + k < nb) ? cnt[base + k] : 0u;
    sa += v[k];
    sb += (v[k] + task_len - 1) / task_len;
  }
  sh_a[tid] = sa; sh_b[tid] = sb;
  __syncthreads();
  for (uint32_t d = 1; d < 256; d <<= 1) {
    uint32_t va = 0, vb = 0;
    if (tid >= d) { va = sh_a[tid - d]; vb = sh_b[tid - d]; }
    __syncthreads();
    sh_a[tid] += va; sh_b[tid] += vb;
    __syncthreads();
  }
  uint32_t pa = tile_a[blockIdx.x] + sh_a[tid] - sa, pb = tile_b[blockIdx.x] + sh_b[tid] - sb;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    if (base + k < nb) {
      off[base + k] = pa;
      toff[base + k] = pb;
    }
    pa += v[k];
    pb += (v[k] + task_len - 1) / task_len;
  }
}

// task descriptor = (first sorted entry, entry count); tasks of one bucket are consecutive
static __global__ __launch_bounds__(256) void msm_task_fill_kernel(const uint32_t* __restrict__ off,
                                                            const uint32_t* __restrict__ toff, uint32_t nb,
                                                            uint32_t task_len, uint2* __restrict__ task_desc) {
  const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nb) return;
  uint32_t start = off[b], left = off[b + 1] - start;
  for (uint32_t t = toff[b], e = toff[b + 1]; t < e; t++) {
    const uint32_t len = left < task_len ? left : task_len;
    task_desc[t] = make_uint2(start, len);
    start += len;
    left -= len;
  }
}

// Bucket accumulation: persistent wavefronts over a work queue.  A wavefront pulls chunks of
// kTaskChunk consecutive tasks from a global counter; a lane that finishes its task takes the
// chunk's next one (ballot + prefix popcount), so lanes stay busy instead of idling behind the
// longest bucket, and there is no grid-quantisation tail.  The descriptor of a newly assigned task
// is loaded when the lane runs dry and consumed one iteration later: the wave never stalls on it
// while the other lanes are adding.  Exit: the queue counter passes `total` (every wave sees it).
static constexpr uint32_t kTaskChunk = 64;

template <class F>
__global__ __launch_bounds__(64, F::kAccumWavesPerSimd) void msm_accumulate_kernel(const Affine<F>* __restrict__ bases,
                                                            const uint32_t* __restrict__ sorted,
                                                            const uint32_t* __restrict__ toff, uint32_t nb,
                                                            const uint2* __restrict__ task_desc,
                                                            uint32_t* __restrict__ queue,
                                                            XYZZ<F>* __restrict__ partial) {
  const uint32_t total = toff[nb];
  const uint32_t lane = threadIdx.x;
  const unsigned long long lt_mask = (1ull << lane) - 1;
  constexpr uint32_t kNone = 0xffffffffu;
  uint32_t next = 0, chunk_end = 0;     // wave-uniform: the chunk being handed out
  bool exhausted = false;               // wave-uniform: the queue has no more chunks
  uint32_t my_task = kNone, pending = kNone, cur = 0, end = 0;
  uint2 desc = make_uint2(0, 0);
  XYZZ<F> acc;
  x29_set_inf(acc);
  for (;;) {
    // 1. lanes whose descriptor arrived start their task
    if (pending != kNone) {
      my_task = pending;
      pending = kNone;
      cur = desc.x;
      end = desc.x + desc.y;
      x29_set_inf(acc);
    }
    // 2. lanes that ran dry flush and ask for the next task
    const bool finished = (my_task != kNone && cur == end);
    const bool dry = finished || (my_task == kNone && pending == kNone && !exhausted);
    const unsigned long long m = __ballot(dry);
    if (m) {
      if (finished) {
        partial[my_task] = acc;
        my_task = kNone;
      }
      if (next == chunk_end && !exhausted) {   // wave-uniform: pull the next chunk
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
      if (dry && !exhausted) {
        const uint32_t cand = next + (uint32_t)__popcll(m & lt_mask);
        if (cand < chunk_end) {
          pending = cand;
          desc = task_desc[cand];
        }
      }
      if (!exhausted) {
        next += (uint32_t)__popcll(m);
        if (next > chunk_end) next = chunk_end;
      }
    }
    // 3. done when no lane holds or awaits a task and the queue is empty
    if (exhausted && __ballot(my_task != kNone || pending != kNone) == 0) break;
    // 4. one mixed addition per busy lane
    if (my_task != kNone) {
      const uint32_t idx = sorted[cur++];
      Affine<F> p = bases[idx & 0x7fffffffu];
      if (idx >> 31) a29_neg(p);
      x29_madd(acc, p);
    }
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

static constexpr uint32_t kLightTasks = 6;   // buckets with more partials take the wavefront path

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
                                                               uint32_t* __restrict__ heavy, uint32_t max_heavy) {
  const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nb) return;
  const uint32_t t0 = toff[b], t1 = toff[b + 1];
  XYZZ<F> acc;
  x29_set_inf(acc);
  if (t1 - t0 > kLightTasks) {
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
                                                                Affine<F>* __restrict__ out, uint32_t n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Affine<F> r;
  a29_from_canon<F, typename F::CanonOps>(r, in[i]);
  out[i] = r;
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
  const uint32_t W = (uint32_t)m.W, B = m.nbuckets, WT = W + 1, nb = WT * B;  // + the ones window
  ws->last_accum_ms = 0.f;
  ws->launched_n = m.n;
  using CPT = XYZZ<typename F::CanonOps>;
  ws->out_bytes = (size_t)WT * sizeof(CPT);
  if (m.n == 0) return G16_OK;
  const uint32_t seg_len = msm_seg_len();
  const uint32_t nseg = (B + seg_len - 1) / seg_len;
  U256 K;
  msm_make_K(m.c, m.W, K);
  const uint32_t nblk = (m.n + 255) / 256;
  const uint32_t chunks = ws->chunks, per = (m.n + chunks - 1) / chunks;
  const size_t lds_bytes = (size_t)B * 4;
  {
    static bool attr_done = false;   // > 64 KiB of dynamic LDS needs the opt-in (c = 16: 128 KiB histogram)
    if (!attr_done) {
      G16_HIP(hipFuncSetAttribute((const void*)msm_sort_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      G16_HIP(hipFuncSetAttribute((const void*)msm_sort_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      attr_done = true;
    }
  }
  msm_digits_kernel<<<nblk, 256, 0, st>>>(d_scalars, m.d_src, m.n, m.c, m.W, K, ws->d_dig);
  msm_sort_kernel<0><<<WT * chunks, 1024, lds_bytes, st>>>(ws->d_dig, m.n, B, chunks, per, ws->d_hist, nullptr);
  msm_hist_sum_kernel<<<(nb + 255) / 256, 256, 0, st>>>(ws->d_hist, nb, B, chunks, ws->d_cnt);
  const uint32_t ntiles = (nb + kScanTile - 1) / kScanTile;
  msm_scan_tiles_kernel<<<ntiles, 256, 0, st>>>(ws->d_cnt, nb, m.task_len, ws->d_tile_a, ws->d_tile_b);
  msm_scan_top_kernel<<<1, 1024, 0, st>>>(ws->d_tile_a, ws->d_tile_b, ntiles, ws->d_off + nb, ws->d_toff + nb);
  msm_scan_apply_kernel<<<ntiles, 256, 0, st>>>(ws->d_cnt, nb, m.task_len, ws->d_tile_a, ws->d_tile_b, ws->d_off,
                                                ws->d_toff);
  msm_hist_start_kernel<<<(nb + 255) / 256, 256, 0, st>>>(ws->d_hist, nb, B, chunks, ws->d_off);
  msm_sort_kernel<1><<<WT * chunks, 1024, lds_bytes, st>>>(ws->d_dig, m.n, B, chunks, per, ws->d_hist, ws->d_sorted);
  msm_task_fill_kernel<<<(nb + 255) / 256, 256, 0, st>>>(ws->d_off, ws->d_toff, nb, m.task_len, ws->d_task_desc);
  // upper bound on tasks: every non-empty bucket has <= 1 short task + entries/task_len full ones
  const uint64_t max_tasks = (uint64_t)nb + ((uint64_t)m.n * W) / m.task_len;  // ones: <= n entries, covered
  // persistent grid: as many wavefronts as the chip holds for this kernel (4/SIMD G1, 2/SIMD G2), fewer
  // when there is little work
  uint64_t waves = (uint64_t)256 * 4 * (sizeof(typename F::T) > sizeof(F29) ? 2 : 4);
  if (waves > (max_tasks + kTaskChunk - 1) / kTaskChunk) waves = (max_tasks + kTaskChunk - 1) / kTaskChunk;
  if (waves == 0) waves = 1;
  G16_HIP(hipMemsetAsync(ws->d_queue, 0, 4, st));
  G16_HIP(hipEventRecord(ws->ev0, st));
  msm_accumulate_kernel<F><<<(unsigned)waves, 64, 0, st>>>((const Affine<F>*)m.d_bases, ws->d_sorted, ws->d_toff, nb,
                                                            ws->d_task_desc, ws->d_queue, (PT*)ws->d_partial);
  G16_HIP(hipEventRecord(ws->ev1, st));
  G16_HIP(hipMemsetAsync(ws->d_heavy, 0, 4, st));
  msm_combine_light_kernel<F><<<(nb + 63) / 64, 64, 0, st>>>((const PT*)ws->d_partial, ws->d_toff, nb,
                                                             (PT*)ws->d_bsum, ws->d_heavy, ws->max_heavy);
  msm_combine_heavy_kernel<F><<<1024, 64, 0, st>>>((const PT*)ws->d_partial, ws->d_toff, (PT*)ws->d_bsum,
                                                   ws->d_heavy, ws->max_heavy);
  msm_bucket_reduce_kernel<F><<<(WT * nseg + 63) / 64, 64, 0, st>>>((const PT*)ws->d_bsum, B, nseg, W, seg_len,
                                                                   (PT*)ws->d_seg);
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
