// Pippenger multi-scalar multiplication on BN254 G1/G2.
//
// Replaces ffjavascript 0.2.48 engine_multiexp.js + wasmcurves 0.1.0 build_multiexp.js
// (`G1/G2.multiExpAffine`, pins /root/reference/yarn.lock:408-416, 1132-1138): same inputs
// (affine Montgomery bases in zkey-section byte layout, 32-byte standard-form scalars), same
// group elements out.  The reference chunks points across web-workers and runs an unsigned-window
// bucket method per chunk and per call; this is a different schedule for the same sums (all point arithmetic
// on the 9x29-bit lazy field, fq29.cuh / ec29.cuh; bases converted once at g16_create):
//
//   A GROUP of base sections shares one front end over one scalar vector (internal.h MsmGroup): the witness
//   MSMs A, B1, C of snarkjs groth16_prove.js are one group -- every witness word is recoded once per section
//   in one pass, ONE sort, ONE task cut -- and B2 (the G2 twin of B1's points) rides on B1's sorted bucket lists
//   as a second accumulate "lane"; the H-MSM is a group of its own.
//
//   front end (msm_g1.hip, curve independent)
//   1. witness groups -- msm_bin_pass_kernel<0/1>: signed c-bit digits straight from the scalars (carry-free: one
//      256-bit add of the constant K = sum 2^(c-1) 2^(cj) turns signed recoding into plain bit extraction; only the
//      windows a scalar reaches are walked), counted (pass 0) and then scattered (pass 1) into (row, bin) runs, bin =
//      the high bits of the bucket: every workgroup writes contiguous runs.  The scalar value 1 (~30 % of an NZCP
//      witness) goes to an extra UNWEIGHTED row per section, spread over its buckets; repeated values
//      to the dup rows (internal.h MsmGroup::dup_rows).
//      dense groups (H, PLONK commitments) -- msm_bin_direct_kernel: ONE pass into fixed-capacity bins (count in LDS,
//      claim a run per bin with one global atomic, scatter); windows of EVEN width (MsmGroup::wb); an overflowing bin
//      makes the launch inert and msm_collect repeats it on the two-pass path.
//   2. msm_bin_sort_kernel: one workgroup per (row, bin) counts the low bucket bits in LDS, writes the bucket
//      populations and the final sorted list of its own contiguous output range.  No global atomics, and the
//      HBM writes of the sort are the payload, not one sector per 4-byte entry.
//   3. msm_scan_*: exclusive scans of the bucket populations -> entry offsets, task ids, queue positions; the task
//      total is checked against the capacity of the task buffers there (over capacity: G16_E_STATE, not a fault).
//   per lane (this header, instantiated for G1 in msm_g1.hip and for G2 in msm_g2.hip)
//   4. msm_task_fill (msm_g1.hip): buckets are cut into tasks of <= task_len sorted entries,
//      queued full-length tasks first, remainders by length class.
//   5. msm_accumulate_kernel: persistent wavefronts over the task queue; a lane walks its task's slice of
//      the sorted list, gathers the packed 64/128-byte affine point and mixed-adds it into an XYZZ
//      accumulator held in VGPRs; a finished lane takes the next task.
//   6. msm_combine_light/medium/heavy_kernel: buckets cut into several tasks -> one sum per bucket.
//   7. row sums sum_b (b + 1) S_b.  Dense rows: msm_bucket_reduce_kernel (running sums per segment, the segments' weights
//      by suffix SCANS over the lanes -- no per-lane multiplication) + msm_pairs_fold_kernel + a host fold of 16 triples;
//      sparse rows (witness lanes): msm_bucket_reduce_mul_kernel (r02: a short double-and-add per segment, free for an
//      empty segment) + msm_row_final_kernel.  Dup rows: msm_dup_bits_kernel + msm_wave_reduce_kernel trees.
//   Row sums go back to the host, which does the c doublings per row (internal.h msm_combine_windows).
//
// Roofline note: ~2.2k VALU instructions (1.47k v_mad_u64_u32) per
// gathered point addition, 13-16 additions per 96 algorithmic bytes: integer-issue bound, not HBM bound.
#pragma once
#include <stdlib.h>

#include <atomic>

#include "ec29.cuh"
#include "internal.h"

namespace g16 {

// by-value kernel argument: the geometry of a group (see MsmGroup)
struct MsmPlan {
  uint32_t nsec, n;
  uint32_t sec_begin[kMsmMaxSections];
  int c, Ws;
  uint32_t W, pf, B, low_bits, bins, rps, rows, ones;
  uint32_t salt_bits;   // see MsmGroup::salt_bits
  uint32_t dup_rows, dup_bits;   // see MsmGroup::dup_rows
  uint32_t wb, wx;      // scalar window j covers wb + (j < wx) bits from bit j wb + min(j, wx): see MsmGroup::wb
  uint32_t wkb;         // window bits in the sort key: see MsmGroup::wkb
};
// the bit offset and width of scalar window j
__host__ __device__ __forceinline__ uint32_t msm_win_off(const MsmPlan& pl, uint32_t j) { return j * pl.wb + (j < pl.wx ? j : pl.wx); }
__host__ __device__ __forceinline__ uint32_t msm_win_bits(const MsmPlan& pl, uint32_t j) { return pl.wb + (j < pl.wx ? 1u : 0u); }

// the hash bucket of a scalar value in its section's dup rows
__host__ __device__ __forceinline__ uint32_t msm_dup_hash(const uint32_t x[8], uint32_t bits) {
  uint64_t h = 0x9E3779B97F4A7C15ull;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    h ^= (uint64_t)x[2 * k] | ((uint64_t)x[2 * k + 1] << 32);
    h *= 0xBF58476D1CE4E5B9ull;
    h ^= h >> 29;
  }
  return (uint32_t)(h >> 24) & ((1u << bits) - 1);
}

// Everything downstream of the sort exists once per curve ("lane"): lane 0 = G1 over every key of the group,
// lane 1 = G2 over the keys of the section it rides on.
struct MsmLaneWs {
  bool active = false;
  int curve = 1;
  uint32_t key_lo = 0, key_hi = 0;   // bucket keys [key_lo, key_hi) of the group
  uint32_t point_base = 0;           // subtracted from a sorted entry to index this lane's base table
  uint32_t rows = 0;                 // (key_hi - key_lo) / B
  uint64_t max_tasks = 0;
  uint32_t task_len = 0;             // entries per task in THIS lane (the G2 lane has fewer lanes to fill and cuts shorter)
  uint32_t task_len_min = 0;         // shortest task length a launch may pick (sizes max_tasks)
  uint32_t* h_stat = nullptr;        // pinned: [0] = end, [1] = start of the lane's sorted entries in the last launch;
                                     // [2] != 0: that launch needed this many tasks, more than max_tasks; [3] != 0: its
                                     // medium / heavy bucket lists overflowed (both: msm_collect fails with G16_E_STATE)
  uint32_t seg_len = 0;              // buckets per reduce segment (a power of two) of the NEXT launch: seg_len_lat, or
  uint32_t seg_len_lat = 0, seg_len_thr = 0;   // seg_len_thr in throughput mode (msm_set_throughput)
  uint32_t row_pts = 0;              // canonical points the row block of d_canon / h_pinned holds: `rows` row sums, or
                                     // rows * ngroups * 3 triples when a row is cut into several workgroups (msm_reduce_plan)
  uint32_t* d_off = nullptr;         // [nbk + 1] first sorted entry of a bucket (absolute position in d_sorted)
  uint32_t* d_toff = nullptr;        // [nbk + 1] exclusive scan of ceil(cnt / task_len): first task id of a bucket
  uint32_t* d_foff = nullptr;        // [nbk + 1] exclusive scan of floor(cnt / task_len): full-length tasks
  uint32_t* d_tile_a = nullptr;
  uint32_t* d_tile_b = nullptr;
  uint32_t* d_tile_c = nullptr;
  uint2* d_task_desc = nullptr;      // by task id
  uint4* d_qdesc = nullptr;          // by queue position: full-length tasks first
  uint32_t* d_class = nullptr;       // [2][kRemClasses]: remainder-class totals and cursors
  uint32_t* d_queue = nullptr;       // [0] work-queue head of the accumulate kernel, [1] number of flagged tasks,
                                     // [2] != 0: over capacity, no task exists (msm_scan_top_kernel)
  uint32_t* d_redo = nullptr;        // tasks whose fast-path sum met an exceptional case
  void* d_partial = nullptr;         // one XYZZ per task
  void* d_bsum = nullptr;            // one XYZZ per bucket cut into several tasks
  uint32_t* d_heavy = nullptr;       // [0] = count, [1..] = buckets with more than kMediumTasks partials
  uint32_t* d_medium = nullptr;      // the same for light_max < partials <= kMediumTasks
  uint32_t max_heavy = 0;
  void* d_seg = nullptr;             // bucket reduce: one (V, W) pair per workgroup (rows cut into several workgroups)
  void* d_red = nullptr;             // sparse rows: ping-pong buffers of the tree over the segment sums
  bool reduce_scan = false;          // dense rows: msm_bucket_reduce_kernel (suffix scans); sparse rows: msm_bucket_reduce_mul_kernel
  void* d_canon = nullptr;           // `rows` canonical XYZZ row sums, then nsec_lane * dup_bit_rows chunk sums of the dup rows
  void* d_dseg = nullptr;            // dup rows: per (section, bit, chunk of 64 hash buckets) sums
  void* d_dred = nullptr;            // ... and their tree
  hipStream_t st_dup = nullptr;      // the dup-row stage runs beside the bucket reduce of the digit rows
  hipEvent_t ev_dup_fork = nullptr, ev_dup_join = nullptr;
  uint32_t* d_dcount = nullptr;      // [nsec_lane] qualifying non-empty dup buckets
  uint32_t* d_dlist = nullptr;       // [nsec_lane][2^dup_bits] their hash-bucket ids, compacted
  uint32_t nsec_lane = 0;            // sections this lane covers (G1: all, G2: one)
  uint8_t* h_pinned = nullptr;
  bool rows_mapped = false;          // d_canon IS h_pinned seen from the device (lane_create)
  size_t out_bytes = 0;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;   // around the bucket-accumulate kernel
  hipEvent_t ev_done = nullptr;
  hipEvent_t trace_ev[4] = {};       // G16_TRACE_HOST: queue built, combined, reduced, end
  float last_accum_ms = 0.f;
  // scheduling of the next launch (msm_launch_lanes / msm_set_waves)
  hipEvent_t gate = nullptr;         // the accumulate kernel waits for this event (nullptr: none)
  uint32_t waves_per_simd = 0;       // persistent accumulate grid, 0 = the kernel's full occupancy
  uint32_t chunk_quota = 0;          // 0 = persistent wavefronts; k = wavefronts retire after k chunks of 64 tasks
};

struct MsmWorkspace {
  // front end
  uint32_t* d_hist = nullptr;        // [chunks][rows * bins] per-workgroup bin histograms -> run starts
  uint32_t* d_bin_cnt = nullptr;     // [rows * bins]
  uint32_t* d_bin_start = nullptr;   // [rows * bins + 1]
  uint2* d_tmp = nullptr;            // binned entries: (table index | sign << 31, low bucket bits)
  uint32_t* d_sorted = nullptr;      // final list: table index | sign << 31, grouped by bucket key
  uint32_t* d_cnt = nullptr;         // [nb] bucket populations
  uint32_t nb = 0;
  // repeated-value detection (MsmGroup::dup_rows): per section and hash bucket
  uint32_t* d_dup_cnt = nullptr;     // points hashed there (scalars other than 0 and 1)
  uint32_t* d_dup_rep = nullptr;     // one of them (global point index), 0xffffffff = none
  uint32_t* d_dup_mixed = nullptr;   // != 0: the bucket holds different values -> not used
  const Fr* d_scalars = nullptr;     // of the running launch
  // single-pass front end (dense single-row groups, msm_bin_direct_kernel): (row, bin) regions of bin_cap entries in d_tmp
  bool direct = false;
  uint32_t bin_cap = 0;
  uint32_t dup_chunk = 16, dup_bit_rows = 16;   // chunk width of the repeated values for the next launch (msm_set_throughput)
  uint32_t* h_over = nullptr;        // pinned: != 0 when a bin overflowed in the last launch (msm_collect repeats it two-pass)
  hipStream_t st_last = nullptr, st2_last = nullptr;   // streams of the running launch (for that repeat)
  MsmLaneWs lane[2];
  hipEvent_t ev_sorted = nullptr;    // sort + scans done: the lanes may start
  hipEvent_t trace_ev[4] = {};       // G16_TRACE_HOST: pass 0, bin scans, pass 1, bin sort
  bool launched = false;
  bool empty = false;                // the last launch had no points
};

struct U256 { uint32_t v[8]; };

// Buckets per reduce segment (a power of two): a lane sums its segment with running sums (2 additions per bucket);
// shorter segments = more lanes and shorter chains on a latency-bound kernel, but a row of more than kReduceMaxThreads
// segments no longer fits one workgroup (msm_bucket_reduce_kernel).
// which: 0 = witness group G1 lane, 1 = G2 lane, 2 = dense (H).  G16_SEG_LEN="w,g2,h" overrides (sweeps).
inline uint32_t msm_seg_len_cfg(int which) {
  struct Cfg { uint32_t v[3]; };
  static const Cfg cfg = [] {   // thread-safe one-time initialisation (two host threads may prove on two handles)
    Cfg c{{8u, 4u, 8u}};
    if (const char* e = getenv("G16_SEG_LEN")) {
      int a = 0, b = 0, d = 0;
      const int k = sscanf(e, "%d,%d,%d", &a, &b, &d);
      if (k >= 1 && a > 0) c.v[0] = (uint32_t)a;
      if (k >= 2 && b > 0) c.v[1] = (uint32_t)b;
      if (k >= 3 && d > 0) c.v[2] = (uint32_t)d;
    }
    for (auto& x : c.v) x = x > 64 ? 64 : x;
    return c;
  }();
  return cfg.v[which];
}

// The latency-bound tail kernels (combine, reduce, tree) run 256-thread workgroups: the four wavefronts of a
// workgroup sit on one CU and walk the same ~100 KB of unrolled point-addition code together, sharing its
// instruction-cache lines (64-thread workgroups scattered one lonely wavefront per CU: r01 measured 71 cycles per
// instruction on the G2 reduce).
static constexpr uint32_t kTailThreads = 256;
// kLatencyPrio: every kernel of the MSM except the bucket accumulate is a chain of dependent steps on few
// wavefronts.  Sharing a SIMD with three accumulate (or NTT) wavefronts under round-robin issue, such a wavefront
// gets a quarter of the issue slots and its chain takes 3x as long (r02 device timeline: the G2 bucket reduce 0.9 ms
// alone, 2.7 ms beside the H accumulate; the H front end 0.44 -> 2.0 ms) -- so these kernels raise their wavefronts'
// issue priority (s_setprio 3); the throughput kernels stay at 0 and lose a few percent of their slots.
static constexpr uint32_t kRemClasses = 32;   // remainder tasks are queued by relative length, longest class first
static constexpr uint32_t kTaskChunk = 64;
static constexpr uint32_t kLightTasks = 3;    // lower bound of the light/other split (sizes the bucket lists)
static constexpr uint32_t kMediumTasks = 64;  // up to this many partials: a 16-lane group per bucket

// what msm_scan_top_kernel resets / reports for its lane besides the scan (msm_build_queue)
struct MsmSmallInit { uint32_t *h_stat, *d_class, *d_queue, *d_heavy, *d_medium; };
// front end + queue construction (msm_g1.hip)
int msm_front_end(const MsmGroup& g, MsmWorkspace* ws, const Fr* d_scalars, hipStream_t st);
int msm_build_queue(const MsmGroup& g, MsmWorkspace* ws, MsmLaneWs& ln, hipStream_t st);
int msm_launch_lane_g2(const MsmGroup& g, MsmWorkspace* ws, MsmLaneWs& ln, hipStream_t st);
int msm_convert_bases_g2(const void* in, void* out, uint32_t n);
int msm_precompute_g2(const void* in, void* out, uint32_t n, int ndbl);

// Bucket accumulation: persistent wavefronts over a work queue.  A wavefront pulls chunks of
// kTaskChunk consecutive tasks from a global counter; a lane that is about to finish its task is handed
// the chunk's next one (ballot + prefix popcount), so lanes stay busy instead of idling behind the
// longest bucket, and there is no grid-quantisation tail.  Every loop iteration is one REAL mixed
// addition for a busy lane: the descriptor of the next task is requested while the lane does the last
// addition of the current one (its load latency hides behind that addition), and a task starts by
// loading its first point straight into the accumulator (ZZ = ZZZ = 1) and adding the second in the
// same iteration -- a task of L entries costs L - 1 iterations (1 when L = 1).
// Exit: the queue counter passes `total` (every wave sees it) -- or the wavefront has taken its quota of chunks -- and
// no lane holds or awaits a task.  chunk_quota = 0xffffffff: persistent wavefronts (a grid that fills the chip once);
// a small quota: a big grid of short-lived wavefronts, so that wave slots free up all the time and the latency-bound
// kernels of the other streams get in (r02: behind a persistent grid the H-MSM's front end took 2.6 ms instead of
// 0.44, the G2 reduce 2.7 instead of 0.9).
template <class F>
__global__ __launch_bounds__(64, F::kAccumWavesPerSimd) void msm_accumulate_kernel(const PackedAffine<F>* __restrict__ bases,
                                                            const uint32_t* __restrict__ sorted,
                                                            const uint32_t* __restrict__ toff, uint32_t nbk,
                                                            uint32_t point_base,
                                                            const uint4* __restrict__ qdesc,
                                                            uint32_t* __restrict__ queue,
                                                            uint32_t* __restrict__ redo,
                                                            XYZZ<F>* __restrict__ partial, uint32_t chunk_quota) {
  const uint32_t total = toff[nbk];
  uint32_t pulled = 0;                  // wave-uniform: chunks this wavefront has taken (it retires after chunk_quota)
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
      a29_unpack(p, bases[(idx & 0x7fffffffu) - point_base]);
      if (idx >> 31) a29_neg(p);
      acc.x = p.x; acc.y = p.y; acc.zz = F::one(); acc.zzz = F::one();
      bad = false;
    }
    // 3. lanes with at most one addition left (or none at all) ask for their next task
    const bool want = pending == kNone && !exhausted && (my_task == kNone || end - cur <= 1u);
    const unsigned long long m = __ballot(want);
    if (m) {
      if (next == chunk_end) {   // wave-uniform: pull the next chunk (`exhausted` is false here: m != 0)
        if (pulled >= chunk_quota) {
          exhausted = true;   // quota reached: finish what the lanes hold and retire (the slot goes to whoever waits)
        } else {
          uint32_t base = 0;
          if (lane == 0) base = atomicAdd(queue, kTaskChunk);
          base = __shfl(base, 0, 64);
          pulled++;
          if (base >= total) {
            exhausted = true;
          } else {
            next = base;
            chunk_end = base + kTaskChunk < total ? base + kTaskChunk : total;
          }
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
      a29_unpack(p, bases[(idx & 0x7fffffffu) - point_base]);
      if (idx >> 31) a29_neg(p);
      bad |= x29_madd_fast(acc, p);
    }
  }
}

// The flagged tasks again, with the complete addition (doubling, cancellation, infinity): a handful per proof at
// most.  One wavefront per task: the lanes take its entries (the first addition into an empty accumulator is a
// copy), then a shuffle tree of complete additions -- ~6 sequential additions instead of task_len.
template <class F>
__global__ __launch_bounds__(64) void msm_redo_kernel(const PackedAffine<F>* __restrict__ bases,
                                                      const uint32_t* __restrict__ sorted, uint32_t point_base,
                                                      const uint2* __restrict__ task_desc,
                                                      const uint32_t* __restrict__ queue,
                                                      const uint32_t* __restrict__ redo,
                                                      XYZZ<F>* __restrict__ partial) {
  __builtin_amdgcn_s_setprio(3);   // issue priority over the throughput kernels sharing the SIMD (msm.cuh, kLatencyPrio)
  const uint32_t count = queue[1];
  const uint32_t lane = threadIdx.x;
  for (uint32_t i = blockIdx.x; i < count; i += gridDim.x) {
    const uint32_t t = redo[i];
    const uint2 d = task_desc[t];
    XYZZ<F> acc;
    x29_set_inf(acc);
    for (uint32_t e = d.x + lane; e < d.x + d.y; e += 64) {
      const uint32_t idx = sorted[e];
      Affine<F> p;
      a29_unpack(p, bases[(idx & 0x7fffffffu) - point_base]);
      if (idx >> 31) a29_neg(p);
      x29_madd(acc, p);
    }
    for (int s = 32; s >= 1; s >>= 1) {
      XYZZ<F> q = xyzz_shfl_down(acc, s);
      if (lane >= (uint32_t)s) x29_set_inf(q);   // (see x29_tree_step)
      x29_add(acc, q);
    }
    if (lane == 0) partial[t] = acc;
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

template <class F, int WIDTH = 64> __device__ __forceinline__ XYZZ<F> xyzz_shfl_down(const XYZZ<F>& p, int delta) {
  XYZZ<F> r;
  constexpr int NW = sizeof(XYZZ<F>) / 4;
  const uint32_t* s = reinterpret_cast<const uint32_t*>(&p);
  uint32_t* d = reinterpret_cast<uint32_t*>(&r);
#pragma unroll
  for (int i = 0; i < NW; i++) d[i] = __shfl_down(s[i], delta, WIDTH);
  return r;
}
// One step of a shuffle tree: lanes [0, d) of a WIDTH-lane group take the value d lanes up.  The other lanes' result
// is never read, and a shuffle past the group's end returns the lane's OWN value: added blindly that is a doubling, and
// the whole wavefront then walks both the doubling and the addition path of x29_add (r02 trace: 15-20 us per step
// against 7.7 us per addition in the reduce kernel) -- they get the point at infinity instead (x29_add returns at once).
template <class F, int WIDTH = 64>
__device__ __forceinline__ void x29_tree_step(XYZZ<F>& acc, int d, uint32_t lane_in_group) {
  XYZZ<F> q = xyzz_shfl_down<F, WIDTH>(acc, d);
  if (lane_in_group >= (uint32_t)d) x29_set_inf(q);
  x29_add(acc, q);
}

// Buckets cut into SEVERAL tasks: bsum[b] = sum of the task partials of bucket b (2 .. light_max
// partials: a lane per bucket; more: queued for the wavefront kernel).  A bucket with one task needs no pass at
// all -- the reduce kernel reads its partial sum directly (msm_bucket_value).
template <class F>
__global__ __launch_bounds__(kTailThreads) void msm_combine_light_kernel(const XYZZ<F>* __restrict__ partial,
                                                               const uint32_t* __restrict__ toff, uint32_t nbk,
                                                               XYZZ<F>* __restrict__ bsum,
                                                               uint32_t* __restrict__ heavy, uint32_t max_heavy,
                                                               uint32_t* __restrict__ medium,
                                                               uint32_t light_max, uint32_t* __restrict__ h_stat) {
  __builtin_amdgcn_s_setprio(3);   // issue priority over the throughput kernels sharing the SIMD (msm.cuh, kLatencyPrio)
  const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nbk) return;
  const uint32_t t0 = toff[b], t1 = toff[b + 1];
  if (t1 - t0 < 2) return;
  if (t1 - t0 > light_max) {               // bsum written by the medium / heavy kernel
    uint32_t* list = (t1 - t0 <= kMediumTasks) ? medium : heavy;
    const uint32_t k = atomicAdd(&list[0], 1u);
    if (k < max_heavy) list[1 + k] = b;    // cannot overflow: max_heavy >= max_tasks / light_max ...
    else h_stat[3] = 1u;                   // ... and if it ever does, msm_collect reports it instead of a wrong sum
    return;
  }
  XYZZ<F> acc = partial[t0];
  for (uint32_t t = t0 + 1; t < t1; t++) {
    const XYZZ<F> s = partial[t];
    x29_add(acc, s);
  }
  bsum[b] = acc;
}

// Buckets with light_max < partials <= 64: FOUR buckets per wavefront, 16 lanes each -- up to four partials per
// lane, then a 4-step tree inside the 16-lane group.  (The real NZCP witness is made of few distinct values repeated hundreds of
// times -- the inverses 1/(i - index) of its QuinSelector comparisons -- so its buckets are few and hold ~10 task
// partials each: a whole wavefront per bucket spent 4 additions on 11 useful lanes, r02: 1.1 ms on the G2 lane.)
template <class F>
__global__ __launch_bounds__(kTailThreads) void msm_combine_medium_kernel(const XYZZ<F>* __restrict__ partial,
                                                                const uint32_t* __restrict__ toff,
                                                                XYZZ<F>* __restrict__ bsum,
                                                                const uint32_t* __restrict__ medium, uint32_t max_heavy) {
  __builtin_amdgcn_s_setprio(3);   // issue priority over the throughput kernels sharing the SIMD (msm.cuh, kLatencyPrio)
  uint32_t count = medium[0];
  if (count > max_heavy) count = max_heavy;
  const uint32_t lane = threadIdx.x & 63u, sub = lane >> 4, l = lane & 15u;
  const uint32_t wpb = blockDim.x >> 6;
  for (uint32_t h0 = (blockIdx.x * wpb + (threadIdx.x >> 6)) * 4; h0 < count; h0 += gridDim.x * wpb * 4) {
    const uint32_t h = h0 + sub;
    XYZZ<F> acc;
    x29_set_inf(acc);
    uint32_t b = 0;
    if (h < count) {
      b = medium[1 + h];
      const uint32_t t0 = toff[b], t1 = toff[b + 1];
      for (uint32_t t = t0 + l; t < t1; t += 16) {   // <= 4 per lane; the first lands in an empty accumulator: a copy
        const XYZZ<F> sp = partial[t];
        x29_add(acc, sp);
      }
    }
    for (int d = 8; d >= 1; d >>= 1) x29_tree_step<F, 16>(acc, d, l);
    if (h < count && l == 0) bsum[b] = acc;
  }
}

// One wavefront per heavy bucket: lanes stride over the partials, then a 6-step shuffle tree.
template <class F>
__global__ __launch_bounds__(kTailThreads) void msm_combine_heavy_kernel(const XYZZ<F>* __restrict__ partial,
                                                               const uint32_t* __restrict__ toff,
                                                               XYZZ<F>* __restrict__ bsum,
                                                               const uint32_t* __restrict__ heavy, uint32_t max_heavy) {
  __builtin_amdgcn_s_setprio(3);   // issue priority over the throughput kernels sharing the SIMD (msm.cuh, kLatencyPrio)
  uint32_t count = heavy[0];
  if (count > max_heavy) count = max_heavy;
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wpb = blockDim.x >> 6;
  for (uint32_t h = blockIdx.x * wpb + (threadIdx.x >> 6); h < count; h += gridDim.x * wpb) {
    const uint32_t b = heavy[1 + h];
    const uint32_t t0 = toff[b], t1 = toff[b + 1];
    XYZZ<F> acc;
    x29_set_inf(acc);
    for (uint32_t t = t0 + lane; t < t1; t += 64) {
      const XYZZ<F> s = partial[t];
      x29_add(acc, s);
    }
    for (int d = 32; d >= 1; d >>= 1) x29_tree_step<F>(acc, d, lane);
    if (lane == 0) bsum[b] = acc;
  }
}

// the sum of bucket b (lane-local index): nothing, its single task's partial sum, or the combined sum
template <class F>
__device__ __forceinline__ XYZZ<F> msm_bucket_value(const XYZZ<F>* __restrict__ partial, const XYZZ<F>* __restrict__ bsum,
                                                    const uint32_t* __restrict__ toff, uint32_t b) {
  const uint32_t t0 = toff[b], nt = toff[b + 1] - t0;
  if (nt == 0) {
    XYZZ<F> z;
    x29_set_inf(z);
    return z;
  }
  if (nt == 1) return partial[t0];
  return bsum[b];
}

// ---------------------------------------------------------------------------------------------- bucket reduce, sparse rows
// (the witness lanes: most buckets of a row are empty -- a lane whose segment is empty skips its whole weighting, which the
// scan-based kernel below cannot; r03 measured the scan-based reduce 20-45 % SLOWER on these rows, so they keep r02's kernel)
// seg[j*nseg + g] = sum_{bi in segment g of row j} (bi+1) * S_bi;  ones rows (j % rps == W): plain sum S_bi
template <class F>
__global__ __launch_bounds__(kTailThreads) void msm_bucket_reduce_mul_kernel(const XYZZ<F>* __restrict__ partial,
                                                               const XYZZ<F>* __restrict__ bsum,
                                                               const uint32_t* __restrict__ toff,
                                                               uint32_t B, uint32_t nseg, uint32_t rows, uint32_t rps,
                                                               uint32_t W, uint32_t ones, uint32_t salt_bits,
                                                               uint32_t seg_len, uint32_t wave_tree,
                                                               XYZZ<F>* __restrict__ seg) {
  __builtin_amdgcn_s_setprio(3);   // issue priority over the throughput kernels sharing the SIMD (msm.cuh, kLatencyPrio)
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= rows * nseg) return;
  // wave_tree (nseg a multiple of 64: a wavefront never straddles two rows): the first level of the tree over the
  // segment sums happens here, on code that is already running -- seg gets one point per wavefront, and the host
  // launches one msm_wave_reduce_kernel less (r02 trace, H lane: 104 us for that launch; its 6 steps cost ~50 here)
  // wave_tree == 2 (nseg a multiple of the workgroup size): the workgroup's wavefront sums are added up through LDS
  // as well, one point per workgroup
  __shared__ uint32_t sh_raw[(kTailThreads / 64) * sizeof(XYZZ<F>) / 4];
  XYZZ<F>* sh = reinterpret_cast<XYZZ<F>*>(sh_raw);
  const uint32_t out = wave_tree == 2 ? tid / kTailThreads : (wave_tree ? tid >> 6 : tid);
  const uint32_t j = tid / nseg, g = tid % nseg;
  if (j % rps >= W + ones) {                   // dup rows: combined by msm_dup_bits_kernel, not here
    XYZZ<F> z;
    x29_set_inf(z);
    if (!wave_tree || (wave_tree == 1 && (threadIdx.x & 63u) == 0) || threadIdx.x == 0) seg[out] = z;
    return;                                    // (the whole workgroup is in this row when it uses the barrier below)
  }
  const bool plain = ones && (j % rps == W);   // the "ones" pseudo-window: plain sum of its buckets
  // the salted top window: 2^salt_bits consecutive buckets share the weight (index >> salt_bits) + 1, and a
  // segment never straddles two weights (seg_len divides 2^salt_bits)
  const bool salted = salt_bits && (j % rps == W - 1);
  const uint32_t lo = g * seg_len;
  const uint32_t hi = (lo + seg_len < B) ? lo + seg_len : B;
  XYZZ<F> run, acc;
  x29_set_inf(run);
  x29_set_inf(acc);
  for (uint32_t bi = hi; bi-- > lo;) {
    const XYZZ<F> s = msm_bucket_value<F>(partial, bsum, toff, j * B + bi);
    x29_add(run, s);
    if (!plain && !salted) x29_add(acc, run);
  }
  if (plain) {
    acc = run;
  } else if (salted) {
    msm_mul_small(acc, run, (lo >> salt_bits) + 1);
  } else if (lo != 0) {
    XYZZ<F> m;
    msm_mul_small(m, run, lo);
    x29_add(acc, m);
  }
  if (wave_tree) {
    const uint32_t lane = threadIdx.x & 63u;
    for (int d = 32; d >= 1; d >>= 1) x29_tree_step<F>(acc, d, lane);
    if (wave_tree == 2) {
      if (lane == 0) sh[threadIdx.x >> 6] = acc;
      __syncthreads();
      if (threadIdx.x == 0) {
        for (uint32_t k = 1; k < kTailThreads / 64; k++) x29_add(acc, sh[k]);
        seg[out] = acc;
      }
    } else if (lane == 0) {
      seg[out] = acc;
    }
  } else {
    seg[tid] = acc;
  }
}

// The end of a lane's tree: one workgroup per row adds that row's `cnt` points (strided partial sums per lane, a
// shuffle tree per wavefront, the four wavefront sums through LDS) and writes the row sum in the canonical format the
// host folds -- instead of a msm_wave_reduce_kernel launch per factor of 64 plus msm_to_canon_kernel.
template <class F>
__global__ __launch_bounds__(kTailThreads) void msm_row_final_kernel(const XYZZ<typename F::Tail>* __restrict__ in, uint32_t cnt,
                                                                     XYZZ<typename F::CanonOps>* __restrict__ out) {
  using FT = typename F::Tail;
  __builtin_amdgcn_s_setprio(3);   // issue priority over the throughput kernels sharing the SIMD (msm.cuh, kLatencyPrio)
  __shared__ uint32_t sh_raw[(kTailThreads / 64) * sizeof(XYZZ<FT>) / 4];
  XYZZ<FT>* sh = reinterpret_cast<XYZZ<FT>*>(sh_raw);
  const uint32_t row = blockIdx.x, lane = threadIdx.x & 63u;
  XYZZ<FT> p;
  x29_set_inf(p);
  for (uint32_t i = threadIdx.x; i < cnt; i += kTailThreads) {
    const XYZZ<FT> q = in[(size_t)row * cnt + i];
    x29_add(p, q);
  }
  for (int d = 32; d >= 1; d >>= 1) x29_tree_step<FT>(p, d, lane);
  if (lane == 0) sh[threadIdx.x >> 6] = p;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (uint32_t k = 1; k < kTailThreads / 64; k++) x29_add(p, sh[k]);
    XYZZ<typename F::CanonOps> r;
    x29_to_canon<F, typename F::CanonOps>(r, *reinterpret_cast<const XYZZ<F>*>(&p));
    out[row] = r;
  }
}

// ---------------------------------------------------------------------------------------------- bucket reduce
// Row sum = sum_b w(b) S_b over the B buckets of a row, w(b) = b + 1 (digit rows), 1 (the "ones" row), or
// (b >> salt_bits) + 1 (the salted top window).  A lane owns a SEGMENT of 2^seg_log consecutive buckets and sums it with
// running sums (A_g = sum_j (j + 1) S_{lo + j}, R_g = sum_j S_{lo + j}: two additions per bucket); the weight of the
// segment's position, f(g) R_g with f(g) = 2^seg_log g (digit rows) or (g >> sh) + 1 (salted, sh = salt_bits - seg_log),
// is NOT a multiplication per lane (r02: msm_mul_small(run, lo), up to 19 doublings + additions, was half of this
// kernel's instructions): with the suffix sums T_g = sum_{g' >= g} R_g',
//     sum_g f(g) R_g = f(0) T_0 + sum_{g >= 1} (f(g) - f(g - 1)) T_g,
// and the increments are a constant power of two (2^seg_log at every g, or 1 at every multiple of 2^sh): a suffix SCAN
// over the lanes (6 shuffle steps in a wavefront, the wavefront totals through LDS), seg_log doublings, one tree.
// A workgroup covers `wg` consecutive segments of one row.  Rows of at most kReduceMaxThreads segments are one workgroup,
// which writes the row sum itself (canonical format); longer rows (the dense H-MSM: 2^19 buckets) are cut into workgroups
// of 256, each writes the pair (V_x, W_x) = (weighted sum relative to its first segment, plain sum), and
// msm_pairs_fold_kernel + the host (msm_fold_row) finish: sum_x V_x + F(x) W_x is the same problem on nwg points.
struct MsmReducePlan {
  uint32_t B, nseg, rows, rps, W, ones, salt_bits;
  uint32_t seg_log;   // log2(buckets per segment)
  uint32_t wg;        // segments (= threads) per workgroup: a power of two, 64 .. kReduceMaxThreads
  uint32_t nwg;       // workgroups per row
};
static constexpr uint32_t kReduceMaxThreads = 512;   // G1; a G2 accumulator pair needs the 512 registers of one wavefront per
                                                     // SIMD: 256-thread workgroups (F::kReduceThreads)
static constexpr uint32_t kPairGroup = 16;   // msm_pairs_fold_kernel: pairs per lane group (one triple out per group)
// the weight of pair x of a row cut into workgroups: F(x) = f(x wg), f as in msm_bucket_reduce_kernel.  kind: 0 digit row,
// 1 ones row, 2 salted top window
__host__ __device__ inline uint64_t msm_pair_weight(const MsmReducePlan& rp, int kind, uint64_t x) {
  if (kind == 1) return 1;
  if (kind == 2) return ((x * rp.wg) >> (rp.salt_bits - rp.seg_log)) + 1;
  return (x * rp.wg) << rp.seg_log;
}
// msm_pairs_fold_kernel: the weights of consecutive pairs of a row differ at every 2^step_log-th pair: digit rows step at
// every pair; a salted row too when a workgroup spans its 2^sh segments of equal weight, else at every (2^sh / wg)-th pair
// (the Y of a ones row is not used)
__host__ __device__ inline uint32_t msm_row_step_log(const MsmReducePlan& rp, int kind) {
  if (kind != 2) return 0;
  const uint32_t sh = rp.salt_bits - rp.seg_log;
  uint32_t wl = 0;
  while ((1u << wl) < rp.wg) wl++;
  return sh > wl ? sh - wl : 0u;
}
__host__ __device__ inline int msm_row_kind(const MsmReducePlan& rp, uint32_t row) {
  const uint32_t jr = row % rp.rps;
  if (rp.ones && jr == rp.W) return 1;
  if (rp.salt_bits && jr == rp.W - 1) return 2;
  return 0;   // (dup rows: every pair is the point at infinity, the kind does not matter)
}

template <class F> __device__ __forceinline__ XYZZ<F> xyzz_shfl(const XYZZ<F>& p, int src) {
  XYZZ<F> r;
  constexpr int NW = sizeof(XYZZ<F>) / 4;
  const uint32_t* s = reinterpret_cast<const uint32_t*>(&p);
  uint32_t* d = reinterpret_cast<uint32_t*>(&r);
#pragma unroll
  for (int i = 0; i < NW; i++) d[i] = __shfl(s[i], src, 64);
  return r;
}
// One step of a suffix scan over the WIDTH lanes of a group: lane l adds the value of lane l + d (nothing past the end)
template <class F, int WIDTH = 64>
__device__ __forceinline__ void x29_scan_step(XYZZ<F>& t, int d, uint32_t lane_in_group, uint32_t width) {
  XYZZ<F> q = xyzz_shfl_down<F, WIDTH>(t, d);
  if (lane_in_group + (uint32_t)d >= width) x29_set_inf(q);
  x29_add(t, q);
}

template <class F>
__global__ __launch_bounds__(F::kReduceThreads) void msm_bucket_reduce_kernel(const XYZZ<F>* __restrict__ partial,
                                                                    const XYZZ<F>* __restrict__ bsum,
                                                                    const uint32_t* __restrict__ toff, MsmReducePlan rp,
                                                                    XYZZ<F>* __restrict__ pairs,
                                                                    XYZZ<typename F::CanonOps>* __restrict__ canon) {
  __builtin_amdgcn_s_setprio(3);   // issue priority over the throughput kernels sharing the SIMD (msm.cuh, kLatencyPrio)
  __shared__ uint32_t sh_raw[(F::kReduceThreads / 64) * sizeof(XYZZ<F>) / 4];
  XYZZ<F>* sh = reinterpret_cast<XYZZ<F>*>(sh_raw);
  const uint32_t j = blockIdx.x / rp.nwg, x = blockIdx.x % rp.nwg;
  const uint32_t gl = threadIdx.x, g = x * blockDim.x + gl;      // segment: local to the workgroup, within the row
  const uint32_t lane = gl & 63u, wv = gl >> 6, nwv = blockDim.x >> 6;
  const uint32_t jr = j % rp.rps;
  if (jr >= rp.W + rp.ones) {   // dup rows: combined by msm_dup_bits_kernel, not here (the whole workgroup is in this row)
    if (gl == 0) {
      if (rp.nwg == 1) {
        xyzz_set_inf(canon[j]);
      } else {
        XYZZ<F> z;
        x29_set_inf(z);
        pairs[2 * (size_t)blockIdx.x] = z;
        pairs[2 * (size_t)blockIdx.x + 1] = z;
      }
    }
    return;
  }
  const bool plain = rp.ones && jr == rp.W;                  // the "ones" pseudo-window: plain sum of its buckets
  const bool salted = rp.salt_bits && jr == rp.W - 1;        // 2^salt_bits consecutive buckets share a weight
  const bool digit = !plain && !salted;
  // 1. the lane's segment: running sums from its top bucket down
  XYZZ<F> run, acc;
  x29_set_inf(run);
  x29_set_inf(acc);
  if (g < rp.nseg) {
    const uint32_t lo = g << rp.seg_log;
    const uint32_t hi = (lo + (1u << rp.seg_log) < rp.B) ? lo + (1u << rp.seg_log) : rp.B;
    for (uint32_t bi = hi; bi-- > lo;) {
      const XYZZ<F> s = msm_bucket_value<F>(partial, bsum, toff, j * rp.B + bi);
      x29_add(run, s);
      if (digit) x29_add(acc, run);
    }
  }
  // 2. suffix sums of the segment totals over the workgroup: in the wavefront, then the wavefront totals through LDS
  for (int d = 1; d < 64; d <<= 1) x29_scan_step<F>(run, d, lane, 64u);
  XYZZ<F> wtot;   // total of the workgroup (plain sum of its buckets)
  if (nwv > 1) {
    if (lane == 0) sh[wv] = run;
    __syncthreads();
    XYZZ<F> tw;
    if (lane < nwv) tw = sh[lane];
    else x29_set_inf(tw);
    for (int d = 1; d < (int)nwv; d <<= 1) x29_scan_step<F>(tw, d, lane, nwv);
    XYZZ<F> above = xyzz_shfl<F>(tw, (int)((wv + 1) & 63u));   // sum of the wavefronts after this one
    if (wv + 1 >= nwv) x29_set_inf(above);
    wtot = xyzz_shfl<F>(tw, 0);
    x29_add(run, above);
    __syncthreads();   // (sh is reused by the tree below)
  } else {
    wtot = xyzz_shfl<F>(run, 0);
  }
  // 3. the positions where the weight steps contribute their suffix sum, times the step
  const uint32_t sh_bits = salted ? rp.salt_bits - rp.seg_log : 0u;
  const bool sel = !plain && gl >= 1u && (gl & ((1u << sh_bits) - 1u)) == 0u;
  if (!sel) x29_set_inf(run);
  if (digit)
    for (uint32_t k = 0; k < rp.seg_log; k++) x29_dbl(run);
  x29_add(acc, run);
  // 4. tree over the workgroup
  for (int d = 32; d >= 1; d >>= 1) x29_tree_step<F>(acc, d, lane);
  if (nwv > 1) {
    if (lane == 0) sh[wv] = acc;
    __syncthreads();
    if (wv == 0) {
      if (lane < nwv) acc = sh[lane];
      else x29_set_inf(acc);
      for (int d = (int)nwv >> 1; d >= 1; d >>= 1) x29_tree_step<F>(acc, d, lane);
    }
  }
  if (gl == 0) {
    if (rp.nwg == 1) {
      if (!digit) x29_add(acc, wtot);   // f(0) = 1 for the ones row and the salted window, 0 for digit rows
      XYZZ<typename F::CanonOps> r;
      x29_to_canon<F, typename F::CanonOps>(r, acc);
      canon[j] = r;
    } else {
      pairs[2 * (size_t)blockIdx.x] = acc;
      pairs[2 * (size_t)blockIdx.x + 1] = wtot;
    }
  }
}

// Rows cut into several workgroups, second step.  A 16-lane group per kPairGroup consecutive pairs (V_x, W_x) of a row,
// four groups per wavefront (a lone wavefront pays ~9 us per dependent G1 addition, ~20 us per G2 addition, so the short
// scan and trees of a narrow group beat wider ones):
//   FINAL (the row has at most kPairGroup pairs -- the G2 lane's rows of two workgroups): the row sum itself,
//     sum V_x + step * (sum of the suffix sums of W at the pairs where the weight steps) + f(0) * sum W_x, canonical;
//   else (the H-MSM's one row of 2^19 buckets: 256 pairs): the triple (P = sum V_x, Y = that sum of suffix sums,
//     Wt = sum W_x) per group; the host adds sum P + step Y + sum_q F(group q) Wt_q (msm_fold_row: a few dozen additions).
// The triples variant runs as TWO wavefronts per four groups (gridDim.x = 2 x the group blocks): one sums the V_x (a 4-step tree),
// the other scans the W_x and sums the selected suffix sums (4 + 4 steps) -- 8 dependent additions on the proof's critical path
// instead of 12 in one wavefront, which cannot overlap them (a lone wavefront issues a multiply-add every ~11 cycles whatever
// the independent work at hand, fq29.cuh).
template <class F, bool FINAL>
__global__ __launch_bounds__(64) void msm_pairs_fold_kernel(const XYZZ<F>* __restrict__ pairs, MsmReducePlan rp,
                                                            uint32_t ngroups,
                                                            XYZZ<typename F::CanonOps>* __restrict__ out) {
  __builtin_amdgcn_s_setprio(3);   // issue priority over the throughput kernels sharing the SIMD (msm.cuh, kLatencyPrio)
  const uint32_t lane = threadIdx.x, k = lane & (kPairGroup - 1u), nwg = rp.nwg;
  const uint32_t nblk = FINAL ? gridDim.x : gridDim.x / 2u;
  const bool sum_v = FINAL || blockIdx.x < nblk, scan_w = FINAL || blockIdx.x >= nblk;   // (block-uniform)
  const uint32_t gid = (blockIdx.x % nblk) * (64u / kPairGroup) + lane / kPairGroup;   // FINAL: the row; else: (row, group)
  const uint32_t row = FINAL ? gid : gid / ngroups, grp = FINAL ? 0u : gid % ngroups;
  const bool live = row < rp.rows;
  const int kind = msm_row_kind(rp, live ? row : 0u);
  const uint32_t step_log = msm_row_step_log(rp, kind);
  const uint32_t x = grp * kPairGroup + k;
  XYZZ<F> v, t;
  x29_set_inf(v);
  x29_set_inf(t);
  if (live && x < nwg) {
    if (sum_v) v = pairs[2 * ((size_t)row * nwg + x)];
    if (scan_w) t = pairs[2 * ((size_t)row * nwg + x) + 1];
  }
  XYZZ<F> y;
  x29_set_inf(y);
  if (scan_w) {
    for (int d = 1; d < (int)kPairGroup; d <<= 1) x29_scan_step<F, (int)kPairGroup>(t, d, k, kPairGroup);
    if (kind != 1 && k >= 1u && (x & ((1u << step_log) - 1u)) == 0u) y = t;
  }
  if (FINAL) {
    const uint64_t step = msm_pair_weight(rp, kind, 1ull << step_log) - msm_pair_weight(rp, kind, 0);   // a power of two
    for (uint64_t m = step; m > 1; m >>= 1) x29_dbl(y);
    x29_add(v, y);
    for (int d = (int)kPairGroup >> 1; d >= 1; d >>= 1) x29_tree_step<F, (int)kPairGroup>(v, d, k);
    if (kind != 0) x29_add(v, t);   // lane k = 0 holds the total of W: f(0) = 1 for the ones row and the salted window
    if (k == 0 && live) {
      XYZZ<typename F::CanonOps> r;
      x29_to_canon<F, typename F::CanonOps>(r, v);
      out[row] = r;
    }
  } else {
    XYZZ<typename F::CanonOps>* o = out + 3 * ((size_t)row * ngroups + grp);
    XYZZ<typename F::CanonOps> r;
    if (sum_v) {
      for (int d = (int)kPairGroup >> 1; d >= 1; d >>= 1) x29_tree_step<F, (int)kPairGroup>(v, d, k);
      if (k == 0 && live) {
        x29_to_canon<F, typename F::CanonOps>(r, v);
        o[0] = r;
      }
    } else {
      for (int d = (int)kPairGroup >> 1; d >= 1; d >>= 1) x29_tree_step<F, (int)kPairGroup>(y, d, k);
      if (k == 0 && live) {
        x29_to_canon<F, typename F::CanonOps>(r, y);
        o[1] = r;
        x29_to_canon<F, typename F::CanonOps>(r, t);
        o[2] = r;
      }
    }
  }
}

// Dup rows, step 1: the qualifying non-empty hash buckets of every section of the lane, compacted into
// dlist[sl][k] (k < dcount[sl]; order is whatever the atomics give -- the sums commute).
static __global__ __launch_bounds__(256) void msm_dup_compact_kernel(const uint32_t* __restrict__ toff, uint32_t B, uint32_t rps,
                                                              uint32_t dup_row0, uint32_t dup_bits, uint32_t sec0,
                                                              uint32_t nsec_lane, const uint32_t* __restrict__ dup_cnt,
                                                              const uint32_t* __restrict__ dup_mixed,
                                                              uint32_t* __restrict__ dcount, uint32_t* __restrict__ dlist) {
  __builtin_amdgcn_s_setprio(3);   // issue priority over the throughput kernels sharing the SIMD (msm.cuh, kLatencyPrio)
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (nsec_lane << dup_bits)) return;
  const uint32_t sl = i >> dup_bits, hb = i & ((1u << dup_bits) - 1);
  const size_t di = ((size_t)(sec0 + sl) << dup_bits) + hb;
  if (dup_cnt[di] < kDupMin || dup_mixed[di]) return;
  const uint32_t key = (sl * rps + dup_row0) * B + hb;
  if (toff[key + 1] == toff[key]) return;
  const uint32_t k = atomicAdd(&dcount[sl], 1u);
  dlist[((size_t)sl << dup_bits) + k] = hb;
}

// Step 2: dseg[(sl * bit_rows + k) * nchunk + chunk] = sum over the list entries [64 chunk, +64) of section sl of
// (bits [k L, (k+1) L) of the entry's single scalar value) * (its bucket sum T), L = chunk_bits.  A lane multiplies
// its own entry by the L-bit chunk (L doublings + ~L/2 additions), then one shuffle tree per wavefront; the host
// weights row k with 2^(k L).  One wavefront per (sl, k, chunk); chunks beyond the list write infinity; the tree over
// the chunks is msm_wave_reduce_kernel with dup_bit_rows * nsec rows.  (First version: one row per BIT, a lane
// contributing T when its value has the bit -- 254 trees per section instead of 16 short multiplications and 16
// trees: 23 % of all VALU instructions of a proof, profiles/r02_pmc_accumulate.txt.)
template <class F>
__global__ __launch_bounds__(kTailThreads) void msm_dup_bits_kernel(const XYZZ<F>* __restrict__ partial,
                                                              const XYZZ<F>* __restrict__ bsum,
                                                              const uint32_t* __restrict__ toff, uint32_t B,
                                                              uint32_t rps, uint32_t dup_row0, uint32_t dup_bits,
                                                              uint32_t sec0, uint32_t nsec_lane, uint32_t nchunk,
                                                              uint32_t chunk_bits, uint32_t bit_rows,
                                                              const uint32_t* __restrict__ dcount,
                                                              const uint32_t* __restrict__ dlist,
                                                              const uint32_t* __restrict__ dup_rep,
                                                              const Fr* __restrict__ scalars,
                                                              const uint32_t* __restrict__ src,
                                                              XYZZ<F>* __restrict__ dseg) {
  __builtin_amdgcn_s_setprio(3);   // issue priority over the throughput kernels sharing the SIMD (msm.cuh, kLatencyPrio)
  // chunk-major wave order: the waves with work (the first count / 64 chunks of every (section, bit)) are then the
  // FIRST workgroups of the grid, spread over all CUs.  (Bit-major order put them at a fixed phase of every 64
  // consecutive workgroups, which the round-robin dispatch maps to the same few CUs: r02, 3.7 ms instead of 0.1.)
  const uint32_t wid = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
  const uint32_t nrow = nsec_lane * bit_rows;
  if (wid >= nrow * nchunk) return;
  const uint32_t chunk = wid / nrow, row = wid % nrow, bit = row % bit_rows, sl = row / bit_rows;
  const uint32_t wave = row * nchunk + chunk;   // output slot: [(sl, bit)][chunk]
  const uint32_t count = dcount[sl], k = chunk * 64 + lane;
  XYZZ<F> acc;
  x29_set_inf(acc);
  if (chunk * 64 < count) {   // wave-uniform
    if (k < count) {
      const uint32_t hb = dlist[((size_t)sl << dup_bits) + k];
      const Fr v = scalars[src[dup_rep[((size_t)(sec0 + sl) << dup_bits) + hb]]];
      // this row's chunk_bits-bit chunk of the value times the bucket sum (short double-and-add)
      const uint32_t pos = bit * chunk_bits;
      uint64_t w2 = v.v[pos >> 5];
      if ((pos >> 5) + 1 < 8) w2 |= (uint64_t)v.v[(pos >> 5) + 1] << 32;
      const uint32_t cv = (uint32_t)(w2 >> (pos & 31)) & ((1u << chunk_bits) - 1u);
      if (cv) {
        const XYZZ<F> t = msm_bucket_value<F>(partial, bsum, toff, (sl * rps + dup_row0) * B + hb);
        msm_mul_small(acc, t, cv);
      }
    }
    for (int d = 32; d >= 1; d >>= 1) x29_tree_step<F>(acc, d, lane);
  }
  if (lane == 0) dseg[wave] = acc;
}

// out[j*nout + blk] = sum of in[j*nin + blk*64 .. +64)
template <class F>
__global__ __launch_bounds__(kTailThreads) void msm_wave_reduce_kernel(const XYZZ<F>* __restrict__ in, uint32_t nin,
                                                             XYZZ<F>* __restrict__ out, uint32_t nout) {
  __builtin_amdgcn_s_setprio(3);   // issue priority over the throughput kernels sharing the SIMD (msm.cuh, kLatencyPrio)
  const uint32_t j = blockIdx.y, blk = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
  if (blk >= nout) return;   // whole wavefronts leave together
  const uint32_t i = blk * 64 + lane;
  XYZZ<F> p;
  if (i < nin) p = in[(size_t)j * nin + i];
  else x29_set_inf(p);
  for (int d = 32; d >= 1; d >>= 1) x29_tree_step<F>(p, d, lane);
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
// row sums: lazy -> canonical XYZZ (what the host folds)
template <class F>
__global__ __launch_bounds__(64) void msm_to_canon_kernel(const XYZZ<F>* __restrict__ in,
                                                          XYZZ<typename F::CanonOps>* __restrict__ out, uint32_t n) {
  __builtin_amdgcn_s_setprio(3);   // issue priority over the throughput kernels sharing the SIMD (msm.cuh, kLatencyPrio)
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  XYZZ<typename F::CanonOps> r;
  x29_to_canon<F, typename F::CanonOps>(r, in[i]);
  out[i] = r;
}

// geometry of a lane's bucket reduce (see msm_bucket_reduce_kernel)
// points per row that leave the device: the row sum (one workgroup per row, or few enough for the FINAL fold), else a
// triple per pair group
inline uint32_t msm_row_out_points(const MsmReducePlan& rp) {
  return rp.nwg <= kPairGroup ? 1u : 3u * ((rp.nwg + kPairGroup - 1) / kPairGroup);
}
inline MsmReducePlan msm_reduce_plan(const MsmGroup& g, const MsmLaneWs& ln) {
  MsmReducePlan rp{};
  rp.B = g.B; rp.rows = ln.rows; rp.rps = g.rps; rp.W = (uint32_t)g.W; rp.ones = g.ones ? 1u : 0u; rp.salt_bits = g.salt_bits;
  rp.seg_log = 0;
  while ((2u << rp.seg_log) <= ln.seg_len) rp.seg_log++;
  rp.nseg = (g.B + (1u << rp.seg_log) - 1) >> rp.seg_log;
  uint32_t pad = 64;
  while (pad < rp.nseg) pad <<= 1;
  uint32_t cap = ln.curve == 2 ? (uint32_t)Fq2x29Ops::kReduceThreads : (uint32_t)Fq29Ops::kReduceThreads;
  static const int cap_env = getenv("G16_REDUCE_WG") ? atoi(getenv("G16_REDUCE_WG")) : 0;   // sweeps: 64 / 128 / 256 / 512
  if (cap_env >= 64 && (uint32_t)cap_env < cap && !(cap_env & (cap_env - 1))) cap = (uint32_t)cap_env;
  if (pad <= cap) { rp.wg = pad; rp.nwg = 1; }
  else { rp.wg = kTailThreads; rp.nwg = pad / kTailThreads; }
  return rp;
}
// Row sum from the triples (P, Y, Wt) of its pair groups (host; canonical field): sum P + step * sum Y + sum_q F(q G) Wt_q,
// the last by suffix sums again -- the weights of consecutive groups differ by one constant or not at all.
template <class FC> inline void xyzz_mul_u64(XYZZ<FC>& r, const XYZZ<FC>& p, uint64_t k) {
  xyzz_set_inf(r);
  for (int i = 63; i >= 0; i--) {
    if (!xyzz_is_inf(r)) xyzz_dbl(r);
    if ((k >> i) & 1) xyzz_add(r, p);
  }
}
template <class FC>
inline void msm_fold_row(XYZZ<FC>& out, const XYZZ<FC>* tr, uint32_t ngroups, const MsmReducePlan& rp, int kind) {
  XYZZ<FC> sp, sy, tw, t;
  xyzz_set_inf(sp);
  xyzz_set_inf(sy);
  for (uint32_t q = 0; q < ngroups; q++) {
    xyzz_add(sp, tr[3 * q]);
    xyzz_add(sy, tr[3 * q + 1]);
  }
  out = sp;
  if (kind == 1) {            // ones row: plain sum of everything
    for (uint32_t q = 0; q < ngroups; q++) xyzz_add(out, tr[3 * q + 2]);
    return;
  }
  // (a salted row whose weight steps are wider than a pair group has Y = infinity: no lane is selected)
  const uint64_t step = msm_pair_weight(rp, kind, 1ull << msm_row_step_log(rp, kind)) - msm_pair_weight(rp, kind, 0);
  if (step) {
    xyzz_mul_u64(t, sy, step);
    xyzz_add(out, t);
  }
  // sum_q w_q Wt_q with w_q = F(q G): = sum_q (w_q - w_{q-1}) TW_q over the suffix sums TW; the differences take at most a
  // couple of distinct values, one multiplication each
  uint64_t dv[4] = {0, 0, 0, 0};
  XYZZ<FC> ds[4];
  int nd = 0;
  xyzz_set_inf(tw);
  for (uint32_t q = ngroups; q-- > 0;) {
    xyzz_add(tw, tr[3 * q + 2]);
    const uint64_t wq = msm_pair_weight(rp, kind, (uint64_t)q * kPairGroup);
    const uint64_t d = q ? wq - msm_pair_weight(rp, kind, (uint64_t)(q - 1) * kPairGroup) : wq;
    if (!d) continue;
    int k = 0;
    while (k < nd && dv[k] != d) k++;
    if (k == nd) {
      if (nd == 4) {           // (cannot happen with power-of-two geometry; stay exact anyway)
        xyzz_mul_u64(t, tw, d);
        xyzz_add(out, t);
        continue;
      }
      dv[nd] = d;
      xyzz_set_inf(ds[nd]);
      nd++;
    }
    xyzz_add(ds[k], tw);
  }
  for (int k = 0; k < nd; k++) {
    xyzz_mul_u64(t, ds[k], dv[k]);
    xyzz_add(out, t);
  }
}

// light/heavy split of the combine pass.  A lane sums the partials of a "light" bucket one after the other, so ONE
// bucket with many partials holds its whole wavefront (and, these kernels being latency-bound, the kernel) for
// that many sequential additions: r02, real NZCP witness, light_max = 18 -> 17 G2 additions = 1.15 ms for a
// kernel whose typical bucket has 2 partials.  Everything beyond a few partials takes the wavefront-per-bucket
// tree instead (its steps above the partial count cost nothing: an addition of infinity returns at once).
inline uint32_t msm_light_max(const MsmLaneWs& ln) { return ln.curve == 2 ? 3u : 4u; }

// ------------------------------------------------------------------ one lane, after the sort (per curve)
// Enqueues queue construction, accumulate, combine, reduce and the copy of the row sums on `st`.
template <class F>
int msm_launch_lane_t(const MsmGroup& g, MsmWorkspace* ws, MsmLaneWs& ln, const void* d_bases, hipStream_t st) {
  using PT = XYZZ<F>;
  using CPT = XYZZ<typename F::CanonOps>;
  using FT = typename F::Tail;   // the field ops of the tail kernels: same element layout
  using TPT = XYZZ<FT>;
  static_assert(sizeof(TPT) == sizeof(PT), "tail ops must share the point layout");
  static const bool trace = getenv("G16_TRACE_HOST") != nullptr;
  auto mark = [&](int k) {
    if (!trace) return;
    if (!ln.trace_ev[k]) (void)hipEventCreate(&ln.trace_ev[k]);
    (void)hipEventRecord(ln.trace_ev[k], st);
  };
  int rc = msm_build_queue(g, ws, ln, st);
  if (rc) return rc;
  mark(0);
  const uint32_t nbk = ln.key_hi - ln.key_lo;
  // persistent grid: as many wavefronts as the chip holds for this kernel (4/SIMD G1, 2/SIMD G2), fewer
  // when there is little work
  // (five wavefronts per SIMD for G1 -- __launch_bounds__(64, 5): 96 VGPRs + 16 spilled words -- was measured in r03:
  // 4.18-4.21 ms per proof against 4.10-4.17, 7.52 against 7.29 on the 1.7 M circuit)
  const uint32_t full_occ = (uint32_t)F::kAccumWavesPerSimd;
  const uint32_t occ = (ln.waves_per_simd && ln.waves_per_simd < full_occ) ? ln.waves_per_simd : full_occ;
  uint64_t waves = (uint64_t)256 * 4 * occ;
  const uint64_t max_chunks = (ln.max_tasks + kTaskChunk - 1) / kTaskChunk;
  uint32_t quota = 0xffffffffu;
  if (ln.chunk_quota) {   // short-lived wavefronts: enough of them to drain the queue, quota chunks each
    quota = ln.chunk_quota;
    waves = (max_chunks + quota - 1) / quota + 64;
  }
  if (waves > max_chunks) waves = max_chunks;
  if (waves == 0) waves = 1;
  if (ln.gate) G16_HIP(hipStreamWaitEvent(st, ln.gate, 0));
  G16_HIP(hipEventRecord(ln.ev0, st));
  msm_accumulate_kernel<F><<<(unsigned)waves, 64, 0, st>>>((const PackedAffine<F>*)d_bases, ws->d_sorted, ln.d_toff, nbk,
                                                            ln.point_base, ln.d_qdesc, ln.d_queue, ln.d_redo,
                                                            (PT*)ln.d_partial, quota);
  G16_HIP(hipEventRecord(ln.ev1, st));
  msm_redo_kernel<F><<<64, 64, 0, st>>>((const PackedAffine<F>*)d_bases, ws->d_sorted, ln.point_base, ln.d_task_desc,
                                        ln.d_queue, ln.d_redo, (PT*)ln.d_partial);
  msm_combine_light_kernel<FT><<<(nbk + kTailThreads - 1) / kTailThreads, kTailThreads, 0, st>>>((const TPT*)ln.d_partial, ln.d_toff, nbk,
                                                              (TPT*)ln.d_bsum, ln.d_heavy, ln.max_heavy, ln.d_medium,
                                                              msm_light_max(ln), ln.h_stat);
  msm_combine_medium_kernel<FT><<<1024, kTailThreads, 0, st>>>((const TPT*)ln.d_partial, ln.d_toff, (TPT*)ln.d_bsum, ln.d_medium,
                                                              ln.max_heavy);
  // one wavefront per heavy bucket, all at once (a real NZCP witness puts thousands of entries into the buckets of
  // its byte-valued scalars and of the narrow top window: r02, 1 024 looping wavefronts took 3 rounds)
  msm_combine_heavy_kernel<FT><<<2048, kTailThreads, 0, st>>>((const TPT*)ln.d_partial, ln.d_toff, (TPT*)ln.d_bsum,
                                                   ln.d_heavy, ln.max_heavy);
  mark(1);
  uint32_t nout_pts = ln.row_pts;
  if (g.dup_rows) {
    // beside the reduce + tree of the digit rows (both are chains of sequential additions on few wavefronts)
    hipStream_t sd = ln.st_dup ? ln.st_dup : st;
    if (sd != st) {
      G16_HIP(hipEventRecord(ln.ev_dup_fork, st));
      G16_HIP(hipStreamWaitEvent(sd, ln.ev_dup_fork, 0));
    }
    // dup rows: chunk-weighted sums of the repeated-value bucket sums (bucket keys are lane-local: section sl of the
    // lane starts at row sl * rps)
    // at most kDupChunks * 64 distinct repeated values per section take this path (more are simply left out of
    // the list... they cannot be: every qualifying bucket must be summed) -> the chunk count covers all buckets
    const uint32_t nchunk = (1u << g.dup_bits) >> 6, drows = ln.nsec_lane * ws->dup_bit_rows;
    const uint32_t waves_d = drows * nchunk;
    const uint32_t sec0 = ln.key_lo / (g.rps * g.B), dup_row0 = (uint32_t)g.W + (g.ones ? 1u : 0u);
    G16_HIP(hipMemsetAsync(ln.d_dcount, 0, 16, sd));
    msm_dup_compact_kernel<<<((ln.nsec_lane << g.dup_bits) + 255) / 256, 256, 0, sd>>>(
        ln.d_toff, g.B, g.rps, dup_row0, g.dup_bits, sec0, ln.nsec_lane, ws->d_dup_cnt, ws->d_dup_mixed, ln.d_dcount, ln.d_dlist);
    msm_dup_bits_kernel<FT><<<(waves_d + 3) / 4, kTailThreads, 0, sd>>>(
        (const TPT*)ln.d_partial, (const TPT*)ln.d_bsum, ln.d_toff, g.B, g.rps, dup_row0, g.dup_bits, sec0, ln.nsec_lane, nchunk,
        ws->dup_chunk, ws->dup_bit_rows, ln.d_dcount, ln.d_dlist, ws->d_dup_rep, ws->d_scalars, g.d_src, (TPT*)ln.d_dseg);
    TPT* dcur = (TPT*)ln.d_dseg;
    uint32_t dcnt = nchunk;
    TPT* dbufs[2] = {(TPT*)ln.d_dred, (TPT*)ln.d_dred + (size_t)drows * ((nchunk + 63) / 64)};
    int dflip = 0;
    while (dcnt > 1) {
      const uint32_t nout = (dcnt + 63) / 64;
      msm_wave_reduce_kernel<FT><<<dim3((nout + 3) / 4, drows), kTailThreads, 0, sd>>>(dcur, dcnt, dbufs[dflip], nout);
      dcur = dbufs[dflip];
      dflip ^= 1;
      dcnt = nout;
    }
    msm_to_canon_kernel<F><<<(drows + 63) / 64, 64, 0, sd>>>((const PT*)dcur, (CPT*)ln.d_canon + ln.row_pts, drows);
    G16_HIP(hipGetLastError());
    nout_pts += drows;
    if (sd != st) G16_HIP(hipEventRecord(ln.ev_dup_join, sd));
  }
  if (ln.reduce_scan) {
    const MsmReducePlan rp = msm_reduce_plan(g, ln);
    msm_bucket_reduce_kernel<FT><<<ln.rows * rp.nwg, rp.wg, 0, st>>>((const TPT*)ln.d_partial, (const TPT*)ln.d_bsum, ln.d_toff, rp,
                                                                     (TPT*)ln.d_seg, (CPT*)ln.d_canon);
    mark(2);
    if (rp.nwg > 1) {
      const uint32_t ngroups = (rp.nwg + kPairGroup - 1) / kPairGroup, gpw = 64u / kPairGroup;
      if (ngroups == 1)
        msm_pairs_fold_kernel<FT, true><<<(ln.rows + gpw - 1) / gpw, 64, 0, st>>>((const TPT*)ln.d_seg, rp, 1u, (CPT*)ln.d_canon);
      else
        msm_pairs_fold_kernel<FT, false><<<2 * ((ln.rows * ngroups + gpw - 1) / gpw), 64, 0, st>>>((const TPT*)ln.d_seg, rp, ngroups,
                                                                                            (CPT*)ln.d_canon);
    }
  } else {
    const uint32_t seg_len = ln.seg_len;
    const uint32_t nseg = (g.B + seg_len - 1) / seg_len;
    const uint32_t wave_tree = (nseg % kTailThreads == 0) ? 2u : ((nseg % 64 == 0) ? 1u : 0u);
    msm_bucket_reduce_mul_kernel<FT><<<(ln.rows * nseg + kTailThreads - 1) / kTailThreads, kTailThreads, 0, st>>>(
        (const TPT*)ln.d_partial, (const TPT*)ln.d_bsum, ln.d_toff, g.B, nseg, ln.rows, g.rps, (uint32_t)g.W, g.ones ? 1u : 0u,
        g.salt_bits, seg_len, wave_tree, (TPT*)ln.d_seg);
    mark(2);
    // tree: d_seg (nseg per row, or fewer after the fused first levels) -> ... -> 1 per row, ping-pong between d_red halves
    TPT* cur = (TPT*)ln.d_seg;
    uint32_t cnt = wave_tree == 2 ? nseg / kTailThreads : (wave_tree ? nseg / 64 : nseg);
    if (wave_tree == 2 && cnt <= 4096) {
      msm_row_final_kernel<F><<<ln.rows, kTailThreads, 0, st>>>((const TPT*)cur, cnt, (CPT*)ln.d_canon);
    } else {
      TPT* bufs[2] = {(TPT*)ln.d_red, (TPT*)ln.d_red + (size_t)ln.rows * ((nseg + 63) / 64)};
      int flip = 0;
      while (cnt > 1) {
        const uint32_t nout = (cnt + 63) / 64;
        msm_wave_reduce_kernel<FT><<<dim3((nout + 3) / 4, ln.rows), kTailThreads, 0, st>>>(cur, cnt, bufs[flip], nout);
        cur = bufs[flip];
        flip ^= 1;
        cnt = nout;
      }
      G16_HIP(hipGetLastError());
      msm_to_canon_kernel<F><<<(ln.rows + 63) / 64, 64, 0, st>>>((const PT*)cur, (CPT*)ln.d_canon, ln.rows);
    }
  }
  G16_HIP(hipGetLastError());
  if (g.dup_rows && ln.st_dup) G16_HIP(hipStreamWaitEvent(st, ln.ev_dup_join, 0));
  if (!ln.rows_mapped) G16_HIP(hipMemcpyAsync(ln.h_pinned, ln.d_canon, (size_t)nout_pts * sizeof(CPT), hipMemcpyDeviceToHost, st));
  mark(3);
  G16_HIP(hipEventRecord(ln.ev_done, st));
  return G16_OK;
}

}  // namespace g16
