// MSM front end (curve independent: digit extraction, two-level bucket sort, scans, task queues), the G1 lane
// instantiation, and the host plumbing of groups and workspaces.  See msm.cuh for the overall schedule.
#include <string.h>

#include "msm.cuh"

namespace g16 {

size_t msm_point_bytes(int curve) { return curve == 2 ? sizeof(G2XYZZ) : sizeof(G1XYZZ); }

// ====================================================================== front-end kernels
__device__ __forceinline__ uint32_t msm_extract(const uint32_t s[8], int pos, int c) {
  const int word = pos >> 5, off = pos & 31;
  if (word >= 8) return 0;
  uint64_t v = s[word];
  if (word + 1 < 8) v |= (uint64_t)s[word + 1] << 32;
  return (uint32_t)(v >> off) & ((1u << c) - 1);
}

// Loads the scalar of point g, adds K; returns true when the scalar is exactly 1.  *bits (optional) <- bit length of the scalar.
__device__ __forceinline__ bool msm_load_scalar(const Fr* __restrict__ scalars, const uint32_t* __restrict__ src,
                                                uint32_t g, const U256& K, uint32_t s[8], uint32_t* bits = nullptr) {
  const Fr x = scalars[src ? src[g] : g];   // (src == nullptr: the identity map)
  uint32_t hi = 0;
#pragma unroll
  for (int k = 1; k < 8; k++) hi |= x.v[k];
  const bool one = (hi == 0 && x.v[0] == 1);
  if (bits) {
    uint32_t b = 0;
#pragma unroll
    for (int k = 0; k < 8; k++)
      if (x.v[k]) b = 32u * (uint32_t)k + 32u - (uint32_t)__clz(x.v[k]);
    *bits = b;
  }
  uint64_t cy = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    cy += (uint64_t)x.v[k] + K.v[k];
    s[k] = (uint32_t)cy;
    cy >>= 32;
  }
  return one;
}

// Repeated-value detection (MsmGroup::dup_rows), two passes over the points of a group:
//   count: every scalar other than 0 and 1 is hashed to a bucket of its section; the bucket counts it and keeps one
//          representative point;
//   check: a point whose scalar differs from its bucket's representative marks the bucket mixed (exact 256-bit
//          comparison: a qualifying bucket holds ONE value).
// msm_bin_pass_kernel then routes the points of buckets with >= kDupMin equal scalars to the dup rows.
__device__ __forceinline__ bool msm_scalar_is_01(const Fr& x) {
  uint32_t hi = 0;
#pragma unroll
  for (int k = 1; k < 8; k++) hi |= x.v[k];
  return hi == 0 && x.v[0] <= 1u;
}
template <int CHECK>
static __global__ __launch_bounds__(256) void msm_dup_scan_kernel(const Fr* __restrict__ scalars,
                                                           const uint32_t* __restrict__ src, MsmPlan pl,
                                                           uint32_t* __restrict__ dup_cnt, uint32_t* __restrict__ dup_rep,
                                                           uint32_t* __restrict__ dup_mixed) {
  __builtin_amdgcn_s_setprio(3);   // issue priority over the throughput kernels sharing the SIMD (msm.cuh, kLatencyPrio)
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= pl.n) return;
  uint32_t s = 0;
  if (pl.nsec > 1 && g >= pl.sec_begin[1]) s = 1;
  if (pl.nsec > 2 && g >= pl.sec_begin[2]) s = 2;
  const Fr x = scalars[src[g]];
  if (msm_scalar_is_01(x)) return;
  const size_t di = ((size_t)s << pl.dup_bits) + msm_dup_hash(x.v, pl.dup_bits);
  if (!CHECK) {
    atomicAdd(&dup_cnt[di], 1u);
    if (dup_rep[di] == 0xffffffffu) atomicCAS(&dup_rep[di], 0xffffffffu, g);
  } else {
    if (dup_cnt[di] < kDupMin || dup_mixed[di]) return;
    const Fr y = scalars[src[dup_rep[di]]];
    uint32_t diff = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) diff |= x.v[k] ^ y.v[k];
    if (diff) dup_mixed[di] = 1u;
  }
}

// Pass 0 (MODE 0) counts, pass 1 (MODE 1) scatters the bucket entries of a chunk of points into (row, bin) runs.
// A workgroup owns points [blockIdx.x * per, +per); its LDS holds one counter / cursor per (row, bin).  Signed
// digit j of point i (section s) becomes the entry (table index | sign << 31, low bucket bits) in row
// s * rps + (j mod W), bin = bucket >> low_bits; table index = global point index, or k n + i for window
// j = k W + r of a precomputed table (single-section groups).  Lanes start at different windows so that the
// LDS atomics of one instruction spread over the rows; the scalars equal to 1 -- a third of an NZCP witness, and
// neighbours share their (row, bin) -- are counted with one atomic per wavefront and bin (ballot + popcount).
// 256-thread workgroups on purpose: a 1024-thread workgroup needs 16 free wave slots on ONE CU at once and
// starves behind the NTT's small workgroups when the two run concurrently (r02: 0.1 ms -> 3.5 ms).
// hist layout: [rb][chunk] (chunk-contiguous, for the scan below).
static constexpr uint32_t kBinThreads = 256;
template <int MODE>
static __global__ __launch_bounds__(kBinThreads) void msm_bin_pass_kernel(const Fr* __restrict__ scalars,
                                                            const uint32_t* __restrict__ src, MsmPlan pl, U256 K,
                                                            uint32_t per, uint32_t chunks, uint32_t* __restrict__ hist,
                                                            const uint32_t* __restrict__ bin_start,
                                                            const uint32_t* __restrict__ dup_cnt,
                                                            const uint32_t* __restrict__ dup_mixed,
                                                            uint2* __restrict__ tmp) {
  __builtin_amdgcn_s_setprio(3);   // issue priority over the throughput kernels sharing the SIMD (msm.cuh, kLatencyPrio)
  extern __shared__ uint32_t lds[];
  const uint32_t nrb = pl.rows * pl.bins;
  for (uint32_t b = threadIdx.x; b < nrb; b += kBinThreads)
    lds[b] = MODE ? bin_start[b] + hist[(size_t)b * chunks + blockIdx.x] : 0u;
  __syncthreads();
  const uint32_t lo = blockIdx.x * per;
  const uint32_t hi = (lo + per < pl.n) ? lo + per : pl.n;
  const uint32_t lowmask = (1u << pl.low_bits) - 1;
  const uint32_t Ws = (uint32_t)pl.Ws;
  const uint32_t lane = threadIdx.x & 63u;
  const unsigned long long lt_mask = (1ull << lane) - 1;
  for (uint32_t g0 = lo; g0 < hi; g0 += kBinThreads) {   // uniform trip count: the ballots below need every lane
    const uint32_t g = g0 + threadIdx.x;
    const bool live = g < hi;
    uint32_t s = 0, i = 0, row0 = 0, bits = 0;
    uint32_t sc[8];
    bool one = false;
    if (live) {
      if (pl.nsec > 1 && g >= pl.sec_begin[1]) s = 1;
      if (pl.nsec > 2 && g >= pl.sec_begin[2]) s = 2;
      i = g - pl.sec_begin[s];
      row0 = s * pl.rps;
      one = msm_load_scalar(scalars, src, g, K, sc, &bits);
    }
    if (pl.ones) {
      // wave-aggregated count of the ones: lanes with the same (row, bin) share one atomic
      const bool is_one = live && one;
      const uint32_t bucket = i & (pl.B - 1);
      const uint32_t rb = (row0 + pl.W) * pl.bins + (bucket >> pl.low_bits);
      unsigned long long todo = __ballot(is_one);
      while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const uint32_t key = (uint32_t)__shfl((int)rb, leader, 64);
        const unsigned long long m = __ballot(is_one && rb == key) & todo;
        uint32_t base = 0;
        if ((int)lane == leader) base = atomicAdd(&lds[key], (uint32_t)__popcll(m));
        base = (uint32_t)__shfl((int)base, leader, 64);
        if (MODE && ((m >> lane) & 1ull)) tmp[base + (uint32_t)__popcll(m & lt_mask)] = make_uint2(g, bucket & lowmask);
        todo &= ~m;
      }
    }
    if (!live || (one && pl.ones)) continue;
    if (pl.dup_rows) {
      // a value shared by >= kDupMin points of the section: ONE entry in the section's dup rows
      const Fr x = scalars[src[g]];
      if (!msm_scalar_is_01(x)) {
        const uint32_t hb = msm_dup_hash(x.v, pl.dup_bits);
        const size_t di = ((size_t)s << pl.dup_bits) + hb;
        if (dup_cnt[di] >= kDupMin && !dup_mixed[di]) {
          const uint32_t row = row0 + pl.W + pl.ones + hb / pl.B, bucket = hb & (pl.B - 1);
          const uint32_t rb = row * pl.bins + (bucket >> pl.low_bits);
          const uint32_t pos = atomicAdd(&lds[rb], 1u);
          if (MODE) tmp[pos] = make_uint2(g, bucket & lowmask);
          continue;
        }
      }
    }
    // Only the windows the scalar reaches: above its top bit, x + K holds the bits of K alone -- digit 0 -- except for ONE
    // carry into the next window.  A witness is mostly zeros and small values (63 % / ~20 % of the NZCP witness): r02 walked
    // all Ws = 20 windows of every point, 1 250 instructions per point and pass -- as many VALU instructions in the two bin
    // passes as in both witness bucket accumulations together (profiles/r03_sweeps.txt).
    uint32_t nwin = bits ? (bits - 1u) / pl.wb + 2u : 0u;
    if (nwin > Ws) nwin = Ws;
    uint32_t j = nwin ? threadIdx.x % nwin : 0u;
    for (uint32_t t = 0; t < nwin; t++) {
      const uint32_t wj = msm_win_bits(pl, j);
      const uint32_t e = msm_extract(sc, (int)msm_win_off(pl, j), (int)wj);
      const int32_t d = (j == Ws - 1) ? (int32_t)e : (int32_t)e - (int32_t)(1u << (wj - 1));   // the top window stays unsigned
      if (d != 0) {
        const uint32_t neg = d < 0 ? 1u : 0u;
        uint32_t bucket = (d < 0 ? (uint32_t)(-d) : (uint32_t)d) - 1u;
        if (pl.salt_bits && j == Ws - 1) bucket = (bucket << pl.salt_bits) | (i & ((1u << pl.salt_bits) - 1));
        const uint32_t r = pl.pf > 1 ? j % pl.W : j;
        const uint32_t eidx = pl.pf > 1 ? (j / pl.W) * pl.n + i : g;
        const uint32_t rb = (row0 + r) * pl.bins + (bucket >> pl.low_bits);
        const uint32_t pos = atomicAdd(&lds[rb], 1u);
        if (MODE) tmp[pos] = make_uint2(eidx | (neg << 31), ((bucket & lowmask) << pl.wkb) | (j & ((1u << pl.wkb) - 1u)));
      }
      j = (j + 1 == nwin) ? 0u : j + 1;
    }
  }
  if (!MODE) {
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < nrb; b += kBinThreads) hist[(size_t)b * chunks + blockIdx.x] = lds[b];
  }
}

// Dense single-row groups (the H-MSM, the PLONK commitments: uniform scalars, no value classes): ONE pass instead of
// count -> two scans -> scatter.  Every (row, bin) owns a fixed-capacity region of tmp; a workgroup counts the digits of
// its points in LDS, claims a run in every bin with one global atomic per bin, and scatters.  Uniform digits fill the bins
// evenly (capacity = a multiple of the expected population, MsmWorkspace::bin_cap); a bin that overflows raises
// over[0] and its entries are dropped -- msm_collect then repeats the launch on the two-pass path, which has no
// capacity (degenerate scalar vectors: every P_i equal, say).  r02 timeline: pass 0 + its scans were 0.1 ms of the H
// front end standalone, 0.2 ms in the product schedule, on the critical chain of a proof.
// (WS: the window count as a compile-time constant -- 13 for the H-MSM at c = 20 -- so that the window loop unrolls and the LDS
// atomics of a point's digits are in flight together instead of one at a time; 0 = run-time count)
template <int WS>
static __global__ __launch_bounds__(kBinThreads) void msm_bin_direct_kernel(const Fr* __restrict__ scalars,
                                                              const uint32_t* __restrict__ src, MsmPlan pl, U256 K,
                                                              uint32_t per, uint32_t cap, uint32_t* __restrict__ bin_cnt,
                                                              uint32_t* __restrict__ over, uint2* __restrict__ tmp) {
  __builtin_amdgcn_s_setprio(3);   // issue priority over the throughput kernels sharing the SIMD (msm.cuh, kLatencyPrio)
  extern __shared__ uint32_t lds[];
  const uint32_t nrb = pl.rows * pl.bins;
  for (uint32_t b = threadIdx.x; b < nrb; b += kBinThreads) lds[b] = 0u;
  __syncthreads();
  const uint32_t lo = blockIdx.x * per;
  const uint32_t hi = (lo + per < pl.n) ? lo + per : pl.n;
  const uint32_t lowmask = (1u << pl.low_bits) - 1;
  const uint32_t Ws = WS ? (uint32_t)WS : (uint32_t)pl.Ws;
  // pass 0 counts (LDS atomics WITHOUT a return value: fire and forget), pass 1 takes positions (returning atomics) and scatters
  auto sweep = [&](auto pass_tag) {
    constexpr bool kScatter = decltype(pass_tag)::value;
    // four points per lane and trip: their scalar loads (two dependent ones each when the group has a point map) are in
    // flight together -- the kernel runs one wavefront per SIMD and its time was the sum of those latencies
    for (uint32_t g0 = lo + threadIdx.x; g0 < hi; g0 += 4 * kBinThreads) {
      uint32_t scq[4][8];
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const uint32_t gq = g0 + (uint32_t)q * kBinThreads;
        (void)msm_load_scalar(scalars, src, gq < hi ? gq : g0, K, scq[q]);
      }
#pragma unroll
      for (int q = 0; q < 4; q++) {
      const uint32_t g = g0 + (uint32_t)q * kBinThreads;
      if (g >= hi) break;
      const uint32_t (&sc)[8] = scq[q];
      // every lane walks the windows in the same order (the rotation of msm_bin_pass_kernel spreads the LDS atomics of a
      // MULTI-row group over its rows; here the bins of a window are spread by the digits themselves), so the window's
      // offset, width and row are wave-uniform scalars
      auto digit = [&](uint32_t j) {
        const uint32_t wj = msm_win_bits(pl, j);
        const uint32_t e = msm_extract(sc, (int)msm_win_off(pl, j), (int)wj);
        const int32_t d = (j == Ws - 1) ? (int32_t)e : (int32_t)e - (int32_t)(1u << (wj - 1));
        if (d != 0) {
          const uint32_t neg = d < 0 ? 1u : 0u;
          uint32_t bucket = (d < 0 ? (uint32_t)(-d) : (uint32_t)d) - 1u;
          if (pl.salt_bits && j == Ws - 1) bucket = (bucket << pl.salt_bits) | (g & ((1u << pl.salt_bits) - 1));
          const uint32_t r = pl.pf > 1 ? j % pl.W : j;
          const uint32_t rb = r * pl.bins + (bucket >> pl.low_bits);
          if constexpr (!kScatter) {
            __hip_atomic_fetch_add(&lds[rb], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          } else {
            const uint32_t eidx = pl.pf > 1 ? (j / pl.W) * pl.n + g : g;
            const uint32_t pos = atomicAdd(&lds[rb], 1u);
            if (!(pos & 0x80000000u))
              tmp[pos] = make_uint2(eidx | (neg << 31), ((bucket & lowmask) << pl.wkb) | (j & ((1u << pl.wkb) - 1u)));
          }
        }
      };
      if constexpr (WS > 0) {
#pragma unroll
        for (uint32_t j = 0; j < (uint32_t)WS; j++) digit(j);
      } else {
        for (uint32_t j = 0; j < Ws; j++) digit(j);
      }
      }
    }
  };
  sweep(std::false_type{});
  __syncthreads();
  // claim: lds[b] <- where this workgroup's run of bin b starts (bit 31: the bin is full, its entries are dropped)
  for (uint32_t b = threadIdx.x; b < nrb; b += kBinThreads) {
    const uint32_t c = lds[b];
    uint32_t start = 0x80000000u;
    if (c) {
      const uint32_t at = atomicAdd(&bin_cnt[b], c);
      if (at + c <= cap) start = b * cap + at;
      else over[0] = 1u;
    }
    lds[b] = start;
  }
  __syncthreads();
  sweep(std::true_type{});
}

// One wavefront per (row, bin): exclusive scan of its per-chunk counts in place (the run starts of pass 1,
// relative to the bin's start), bin_cnt[rb] = the bin's total.
static __global__ __launch_bounds__(64) void msm_bin_chunkscan_kernel(uint32_t* __restrict__ hist, uint32_t chunks,
                                                               uint32_t* __restrict__ bin_cnt) {
  __builtin_amdgcn_s_setprio(3);   // issue priority over the throughput kernels sharing the SIMD (msm.cuh, kLatencyPrio)
  const uint32_t rb = blockIdx.x, lane = threadIdx.x;
  uint32_t* __restrict__ h = hist + (size_t)rb * chunks;
  const uint32_t per = (chunks + 63) / 64;
  const uint32_t lo = lane * per, hi = (lo + per < chunks) ? lo + per : chunks;
  uint32_t s = 0;
  for (uint32_t k = lo; k < hi; k++) s += h[k];
  uint32_t inc = s;   // inclusive scan over lanes
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t v = (uint32_t)__shfl_up((int)inc, d, 64);
    if ((int)lane >= d) inc += v;
  }
  uint32_t run = inc - s;
  for (uint32_t k = lo; k < hi; k++) {
    const uint32_t v = h[k];
    h[k] = run;
    run += v;
  }
  if (lane == 63) bin_cnt[rb] = inc;
}
// exclusive scan of bin_cnt[0, nrb) by one workgroup -> bin_start[0, nrb], bin_start[nrb] = total
static __global__ __launch_bounds__(1024) void msm_bin_scan_kernel(const uint32_t* __restrict__ bin_cnt, uint32_t nrb,
                                                            uint32_t* __restrict__ bin_start, uint32_t cap) {
  __builtin_amdgcn_s_setprio(3);   // issue priority over the throughput kernels sharing the SIMD (msm.cuh, kLatencyPrio)
  __shared__ uint32_t sh[1024];
  const uint32_t tid = threadIdx.x;
  const uint32_t per = (nrb + 1023) / 1024;
  const uint32_t lo = tid * per, hi = (lo + per < nrb) ? lo + per : nrb;
  // (cap: the single-pass front end's bins hold at most `cap` entries; an overflowing launch is repeated, stay in bounds)
  uint32_t s = 0;
  for (uint32_t k = lo; k < hi; k++) s += bin_cnt[k] < cap ? bin_cnt[k] : cap;
  sh[tid] = s;
  __syncthreads();
  for (uint32_t d = 1; d < 1024; d <<= 1) {
    uint32_t v = 0;
    if (tid >= d) v = sh[tid - d];
    __syncthreads();
    sh[tid] += v;
    __syncthreads();
  }
  uint32_t run = sh[tid] - s;
  for (uint32_t k = lo; k < hi; k++) {
    const uint32_t v = bin_cnt[k] < cap ? bin_cnt[k] : cap;
    bin_start[k] = run;
    run += v;
  }
  if (tid == 1023) bin_start[nrb] = sh[1023];
}

// One workgroup per (row, bin): the bin's entries are one contiguous run of tmp.  Count the low bucket bits in
// LDS, write the bucket populations, then scatter the table indices to the bin's OWN contiguous range of the
// final list (second read of the run comes from L2).
static constexpr uint32_t kMaxLowBits = 12;
// direct_cap != 0 (single-pass front end): the bin's entries sit at tmp[rb * direct_cap, + bin_cnt[rb]); the sorted list
// still starts at bin_start[rb].  pl.wkb != 0: the sort key carries the scalar WINDOW below the low bucket bits, so the
// entries of a bucket come out ordered by window = by level of the precomputed base table (MsmGroup::wkb).
static __global__ __launch_bounds__(256) void msm_bin_sort_kernel(const uint2* __restrict__ tmp,
                                                           const uint32_t* __restrict__ bin_start, MsmPlan pl,
                                                           const uint32_t* __restrict__ bin_cnt, uint32_t direct_cap,
                                                           const uint32_t* __restrict__ over,
                                                           uint32_t* __restrict__ cnt, uint32_t* __restrict__ sorted) {
  __builtin_amdgcn_s_setprio(3);   // issue priority over the throughput kernels sharing the SIMD (msm.cuh, kLatencyPrio)
  __shared__ uint32_t c[1u << kMaxLowBits];
  __shared__ uint32_t part[256];
  const uint32_t rb = blockIdx.x, tid = threadIdx.x;
  const uint32_t out_lo = bin_start[rb];
  uint32_t lo = out_lo, hi = bin_start[rb + 1];
  if (direct_cap) {
    // an overflowed launch is repeated on the two-pass path (msm_collect); until then its bins hold runs with holes -- every
    // bin counts as empty, so that no kernel downstream reads an entry nobody wrote
    lo = rb * direct_cap;
    hi = over[0] ? lo : lo + (bin_cnt[rb] < direct_cap ? bin_cnt[rb] : direct_cap);
  }
  const uint32_t nb = 1u << pl.low_bits;          // buckets of the bin
  const uint32_t nl = nb << pl.wkb;               // sort keys
  const uint32_t row = rb / pl.bins, bin = rb % pl.bins;
  uint32_t* __restrict__ cnt_out = cnt + (size_t)row * pl.B + ((size_t)bin << pl.low_bits);
  if (lo == hi) {   // empty bin (e.g. the ones row of a dense scalar vector)
    for (uint32_t l = tid; l < nb && ((bin << pl.low_bits) + l) < pl.B; l += 256) cnt_out[l] = 0;
    return;
  }
  for (uint32_t l = tid; l < nl; l += 256) c[l] = 0;
  __syncthreads();
  {
    uint32_t e = lo + tid;
    for (; e + 768 < hi; e += 1024) {   // four independent loads in flight per lane
      const uint32_t y0 = tmp[e].y, y1 = tmp[e + 256].y, y2 = tmp[e + 512].y, y3 = tmp[e + 768].y;
      atomicAdd(&c[y0], 1u); atomicAdd(&c[y1], 1u); atomicAdd(&c[y2], 1u); atomicAdd(&c[y3], 1u);
    }
    for (; e < hi; e += 256) atomicAdd(&c[tmp[e].y], 1u);
  }
  __syncthreads();
  // exclusive scan of c[0, nl): thread t owns nl / 256 consecutive counters (nl < 256: one each, rest idle)
  const uint32_t per = (nl + 255) / 256;
  const uint32_t l0 = tid * per, l1 = (l0 + per < nl) ? l0 + per : nl;
  uint32_t s = 0;
  for (uint32_t l = l0; l < l1; l++) s += c[l];
  part[tid] = s;
  __syncthreads();
  for (uint32_t d = 1; d < 256; d <<= 1) {
    uint32_t v = 0;
    if (tid >= d) v = part[tid - d];
    __syncthreads();
    part[tid] += v;
    __syncthreads();
  }
  uint32_t run = out_lo + part[tid] - s;
  for (uint32_t l = l0; l < l1; l++) {
    const uint32_t v = c[l];
    c[l] = run;
    run += v;
  }
  __syncthreads();
  // bucket populations: the keys of bucket b are [b << wkb, (b + 1) << wkb), consecutive in the scanned c
  for (uint32_t b = tid; b < nb; b += 256) {
    if (((bin << pl.low_bits) + b) >= pl.B) continue;
    const uint32_t first = c[b << pl.wkb];
    const uint32_t next = (b + 1 < nb) ? c[(b + 1) << pl.wkb] : out_lo + (hi - lo);
    cnt_out[b] = next - first;
  }
  __syncthreads();
  {
    uint32_t e = lo + tid;
    for (; e + 768 < hi; e += 1024) {
      const uint2 e0 = tmp[e], e1 = tmp[e + 256], e2 = tmp[e + 512], e3 = tmp[e + 768];
      sorted[atomicAdd(&c[e0.y], 1u)] = e0.x;
      sorted[atomicAdd(&c[e1.y], 1u)] = e1.x;
      sorted[atomicAdd(&c[e2.y], 1u)] = e2.x;
      sorted[atomicAdd(&c[e3.y], 1u)] = e3.x;
    }
    for (; e < hi; e += 256) {
      const uint2 en = tmp[e];
      sorted[atomicAdd(&c[en.y], 1u)] = en.x;
    }
  }
}

// Exclusive scans off = scan(cnt), toff = scan(ceil(cnt/task_len)), foff = scan(floor(cnt/task_len)) in three
// launches: per-tile sums (2048 counters per workgroup) -> one workgroup scans the tile sums -> per-tile local
// scan + tile offset.
static constexpr uint32_t kScanTile = 2048;   // 256 threads x 8 counters

// relative-length class of a remainder task (see the task queues below)
__device__ __forceinline__ uint32_t msm_rem_class(uint32_t len, uint32_t task_len) {
  return (len * kRemClasses) / task_len;      // len < task_len -> 0 .. kRemClasses - 1
}

static __global__ __launch_bounds__(256) void msm_scan_tiles_kernel(const uint32_t* __restrict__ cnt, uint32_t nb,
                                                             uint32_t tl, uint32_t* __restrict__ tile_a,
                                                             uint32_t* __restrict__ tile_b,
                                                             uint32_t* __restrict__ tile_c) {
  __builtin_amdgcn_s_setprio(3);   // issue priority over the throughput kernels sharing the SIMD (msm.cuh, kLatencyPrio)
  __shared__ uint32_t sh_a[256], sh_b[256], sh_c[256];
  const uint32_t tid = threadIdx.x, base = blockIdx.x * kScanTile + tid * 8;
  uint32_t sa = 0, sb = 0, sc = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const uint32_t v = (base + k < nb) ? cnt[base + k] : 0u;
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

// one workgroup; writes exclusive tile offsets in place and the totals
static __global__ __launch_bounds__(1024) void msm_scan_top_kernel(uint32_t* __restrict__ tile_a,
                                                            uint32_t* __restrict__ tile_b,
                                                            uint32_t* __restrict__ tile_c, uint32_t ntiles,
                                                            const uint32_t* __restrict__ off_base,
                                                            uint32_t* __restrict__ total_a,
                                                            uint32_t* __restrict__ total_b,
                                                            uint32_t* __restrict__ total_c, uint32_t max_tasks,
                                                            MsmSmallInit init) {
  __builtin_amdgcn_s_setprio(3);   // issue priority over the throughput kernels sharing the SIMD (msm.cuh, kLatencyPrio)
  __shared__ uint32_t sh_a[1024], sh_b[1024], sh_c[1024];
  const uint32_t tid = threadIdx.x;
  // the lane's small per-launch state, zeroed here instead of by four hipMemsetAsync (each a ~6 us kernel of its own on
  // the lane's chain): class counters, queue counters, the medium / heavy bucket lists' counts
  if (tid < 2 * kRemClasses) init.d_class[tid] = 0;
  if (tid < 2) init.d_queue[tid] = 0;
  if (tid == 0) { init.d_heavy[0] = 0; init.d_medium[0] = 0; init.h_stat[3] = 0; }
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
  if (tid == 1023) {
    // The task buffers (task_desc, qdesc, redo, partial) hold max_tasks entries.  Their size is derived from the points
    // and the SHORTEST task length a launch may pick, so the total cannot exceed it -- but the task length is a
    // run-time quantity (msm_build_queue), and r02 once got that arithmetic wrong (core dump in g16_g2_multiexp).  So the
    // invariant is checked where the total is known: over capacity, the lane's kernels see d_queue[2] and do nothing
    // that indexes by task, and msm_collect turns h_stat[2] into G16_E_STATE.
    const uint32_t over = sh_b[1023] > max_tasks ? 1u : 0u;
    init.d_queue[2] = over;
    init.h_stat[2] = over ? sh_b[1023] : 0u;
    *total_a = *off_base + sh_a[1023];
    *total_b = over ? 0u : sh_b[1023];
    *total_c = over ? 0u : sh_c[1023];
    // end and start of the lane's sorted entries, straight into pinned host memory (read by the NEXT launch, after this
    // one was collected): two 4-byte copies less on the chain
    init.h_stat[0] = *off_base + sh_a[1023];
    init.h_stat[1] = *off_base;
  }
}

static __global__ __launch_bounds__(256) void msm_scan_apply_kernel(const uint32_t* __restrict__ cnt, uint32_t nb,
                                                             uint32_t tl, const uint32_t* __restrict__ tile_a,
                                                             const uint32_t* __restrict__ tile_b,
                                                             const uint32_t* __restrict__ tile_c,
                                                             const uint32_t* __restrict__ off_base,
                                                             const uint32_t* __restrict__ queue,
                                                             uint32_t* __restrict__ off,
                                                             uint32_t* __restrict__ toff,
                                                             uint32_t* __restrict__ foff,
                                                             uint32_t* __restrict__ class_total) {
  __builtin_amdgcn_s_setprio(3);   // issue priority over the throughput kernels sharing the SIMD (msm.cuh, kLatencyPrio)
  __shared__ uint32_t sh_a[256], sh_b[256], sh_c[256];
  // class_total[c] += remainder tasks (cnt % task_len != 0) of relative-length class c among this tile's buckets (zeroed by
  // msm_scan_top_kernel, which runs before; r02 had a kernel of its own for this: one dependent launch less per lane)
  __shared__ uint32_t sh_cls[kRemClasses];
  if (threadIdx.x < kRemClasses) sh_cls[threadIdx.x] = 0;
  const uint32_t tid = threadIdx.x, base = blockIdx.x * kScanTile + tid * 8;
  uint32_t v[8], sa = 0, sb = 0, sc = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    v[k] = (base + k < nb) ? cnt[base + k] : 0u;
    sa += v[k];
    sb += (v[k] + tl - 1) / tl;
    sc += v[k] / tl;
  }
  sh_a[tid] = sa; sh_b[tid] = sb; sh_c[tid] = sc;
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const uint32_t r = v[k] % tl;
    if (r) atomicAdd(&sh_cls[msm_rem_class(r, tl)], 1u);
  }
  for (uint32_t d = 1; d < 256; d <<= 1) {
    uint32_t va = 0, vb = 0, vc = 0;
    if (tid >= d) { va = sh_a[tid - d]; vb = sh_b[tid - d]; vc = sh_c[tid - d]; }
    __syncthreads();
    sh_a[tid] += va; sh_b[tid] += vb; sh_c[tid] += vc;
    __syncthreads();
  }
  uint32_t pa = *off_base + tile_a[blockIdx.x] + sh_a[tid] - sa, pb = tile_b[blockIdx.x] + sh_b[tid] - sb,
           pc = tile_c[blockIdx.x] + sh_c[tid] - sc;
  const bool over = queue[2] != 0;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    if (base + k < nb) {
      off[base + k] = pa;
      toff[base + k] = over ? 0u : pb;   // over capacity (msm_scan_top_kernel): no bucket has a task
      foff[base + k] = over ? 0u : pc;
    }
    pa += v[k];
    pb += (v[k] + tl - 1) / tl;
    pc += v[k] / tl;
  }
  // (the scan loop above ends with a barrier: every lane's class counts are in)
  if (tid < kRemClasses && sh_cls[tid] && !over) atomicAdd(&class_total[tid], sh_cls[tid]);
}

// ---------------------------------------------------------------------- task queues (per lane)
// task descriptor = (first sorted entry, entry count); the tasks of one bucket have consecutive ids (their
// partial sums are consecutive for the combine pass).  The work QUEUE is a permutation of the tasks: every
// full-length task first, the remainders (one per bucket at most, shorter) after them -- the 64 lanes of a
// wavefront then start and finish their full tasks in the same iteration, so the flush / start / request code
// of the accumulate loop runs once per task instead of in nearly every iteration, and the queue ends with its
// shortest tasks (a shorter drain).  qdesc[q] = (first entry, count, task id, -).  Task ids and queue positions
// are local to the lane (its key range [key_lo, key_hi) of the group, its own task length).

static __global__ __launch_bounds__(256) void msm_task_fill_kernel(const uint32_t* __restrict__ off,
                                                            const uint32_t* __restrict__ toff,
                                                            const uint32_t* __restrict__ foff, uint32_t nbk,
                                                            uint32_t task_len,
                                                            uint2* __restrict__ task_desc, uint4* __restrict__ qdesc,
                                                            const uint32_t* __restrict__ class_total,
                                                            uint32_t* __restrict__ class_cursor, uint32_t max_tasks) {
  __builtin_amdgcn_s_setprio(3);   // issue priority over the throughput kernels sharing the SIMD (msm.cuh, kLatencyPrio)
  // remainders: after all the full tasks, by relative-length class (longest first) so that the lanes of a
  // wavefront hold remainders of (nearly) equal length; inside a class the order is whatever the atomics give
  __shared__ uint32_t h[kRemClasses], base[kRemClasses];
  if (threadIdx.x < kRemClasses) h[threadIdx.x] = 0;
  __syncthreads();
  const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t start = 0, left = 0, rem = 0, cls = 0, rank = 0;
  if (b < nbk) {
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
    uint32_t before = foff[nbk];                               // all full tasks, then the longer classes
    for (uint32_t k = c + 1; k < kRemClasses; k++) before += class_total[k];
    base[c] = h[c] ? before + atomicAdd(&class_cursor[c], h[c]) : 0u;
  }
  __syncthreads();
  if (b >= nbk) return;
  uint32_t fq = foff[b];
  for (uint32_t t = toff[b], e = toff[b + 1]; t < e; t++) {
    const uint32_t len = left < task_len ? left : task_len;
    const uint32_t q = len == task_len ? fq++ : base[cls] + rank;
    if (t < max_tasks && q < max_tasks) {   // (always: the totals were checked by msm_scan_top_kernel)
      task_desc[t] = make_uint2(start, len);
      qdesc[q] = make_uint4(start, len, t, 0u);
    }
    start += len;
    left -= len;
  }
}

// ====================================================================== host side
static void msm_make_K(const MsmPlan& pl, U256& K) {
  for (int i = 0; i < 8; i++) K.v[i] = 0;
  for (int j = 0; j + 1 < pl.Ws; j++) {
    const int bit = (int)(msm_win_off(pl, (uint32_t)j) + msm_win_bits(pl, (uint32_t)j)) - 1;
    if (bit < 256) K.v[bit >> 5] |= 1u << (bit & 31);
  }
}

// Entries per accumulate task from the entries one lane of the persistent grid gets: one task per lane, within [lo, 32]
// (short tasks keep the drain tail of the persistent kernel small: r01 sweep) -- but once a lane gets more than four
// 32-entry tasks, a quarter of its share, up to 128: at N = 2^22 a bucket holds 100-150 entries, 32-entry tasks cut every
// bucket into 4-5 partial sums and the combine pass took as long as the accumulation itself (r03, SHA-256 chain of 163
// compressions, serial stages: accumulate 4.09 ms, combine 3.74 ms).
static uint32_t msm_task_len_for(uint64_t per_lane, uint32_t lo) {
  uint64_t t = per_lane < lo ? lo : (per_lane > 32 ? 32 : per_lane);
  if (per_lane / 4 > 32) t = per_lane / 4 > 128 ? 128 : per_lane / 4;
  return (uint32_t)t;
}

static MsmPlan msm_plan_of(const MsmGroup& g) {
  MsmPlan pl{};
  pl.nsec = (uint32_t)g.nsec;
  pl.n = g.n;
  for (int s = 0; s < kMsmMaxSections; s++) pl.sec_begin[s] = g.sec_begin[s];
  pl.c = g.c;
  pl.Ws = g.Ws;
  pl.W = (uint32_t)g.W;
  pl.pf = g.pf;
  pl.B = g.B;
  pl.low_bits = g.low_bits;
  pl.bins = g.bins;
  pl.rps = g.rps;
  pl.rows = g.rows;
  pl.ones = g.ones ? 1u : 0u;
  pl.salt_bits = g.salt_bits;
  pl.dup_rows = g.dup_rows;
  pl.dup_bits = g.dup_bits;
  pl.wb = g.wb;
  pl.wx = g.wx;
  pl.wkb = g.wkb;
  return pl;
}

static int msm_convert_bases_g1(const void* in, void* out, uint32_t n) {
  msm_convert_bases_kernel<Fq29Ops><<<(n + 255) / 256, 256>>>((const G1Affine*)in, (PackedAffine<Fq29Ops>*)out, n);
  G16_HIP(hipGetLastError());
  G16_HIP(hipDeviceSynchronize());
  return G16_OK;
}
static int msm_precompute_g1(const void* in, void* out, uint32_t n, int ndbl) {
  msm_precompute_kernel<FqOps><<<(n + 255) / 256, 256>>>((const G1Affine*)in, (G1Affine*)out, n, ndbl);
  G16_HIP(hipGetLastError());
  G16_HIP(hipDeviceSynchronize());
  return G16_OK;
}

// Window bits.  Without precomputation: minimise Ws * (n + 2.5 * 2^(c-1)) over c, skipping window sizes whose
// TOP window holds only 1..5 bits of the 254-bit scalar (its handful of buckets would each collect n / 2^bits
// entries: r01 sweep, c = 14 made the H-MSM 3x slower).  With full precomputation (one row of buckets):
// minimise Ws * n + 2.5 * 2^(c-1); the windows are then of even width (MsmGroup::wb), no top window is narrow.
static int choose_c(uint32_t n, bool full_precomp) {
  int best = 13;
  double best_cost = 1e300;
  for (int c = 4; c <= (full_precomp ? 21 : 16); c++) {
    const int Ws = (255 + c - 1) / c;
    const int top_bits = 254 - c * (Ws - 1);
    if (!full_precomp && n >= 4096 && top_bits > 0 && top_bits < 6) continue;   // (full precomputation: even windows, MsmGroup::wb)
    const double cost = full_precomp ? (double)Ws * (double)n + 2.5 * (double)(1u << (c - 1))
                                     : (double)Ws * ((double)n + 2.5 * (double)(1u << (c - 1)));
    if (cost < best_cost) { best_cost = cost; best = c; }
  }
  return best;
}

static bool is_inf_bytes(const uint8_t* p, size_t psz) {
  for (size_t k = 0; k < psz; k += 8) {
    uint64_t w;
    memcpy(&w, p + k, 8);
    if (w) return false;
  }
  return true;
}

// upload `count` canonical affine points, convert to the packed lazy format, and (pf > 1) append the levels
// 2^(ndbl[0] + .. + ndbl[k-1]) * P: dst holds pf * count packed points, level-major
static int upload_table(int curve, const std::vector<uint8_t>& canon, uint32_t count, uint32_t pf, const std::vector<int>& ndbl, void** dst) {
  const size_t lazy_pt = curve == 2 ? sizeof(PackedAffine<Fq2x29Ops>) : sizeof(PackedAffine<Fq29Ops>);
  void* tmp = nullptr;
  void* tmp2 = nullptr;
  G16_HIP(hipMalloc(&tmp, canon.size()));
  if (pf > 1) G16_HIP(hipMalloc(&tmp2, canon.size()));
  G16_HIP(hipMalloc(dst, (size_t)pf * count * lazy_pt));
  G16_HIP(hipMemcpy(tmp, canon.data(), canon.size(), hipMemcpyHostToDevice));
  int rc = G16_OK;
  for (uint32_t k = 0; k < pf && !rc; k++) {
    void* cur = (k & 1) ? tmp2 : tmp;
    void* nxt = (k & 1) ? tmp : tmp2;
    void* d = (uint8_t*)*dst + (size_t)k * count * lazy_pt;
    rc = curve == 2 ? msm_convert_bases_g2(cur, d, count) : msm_convert_bases_g1(cur, d, count);
    if (!rc && k + 1 < pf) rc = curve == 2 ? msm_precompute_g2(cur, nxt, count, ndbl[k]) : msm_precompute_g1(cur, nxt, count, ndbl[k]);
  }
  (void)hipFree(tmp);
  if (tmp2) (void)hipFree(tmp2);
  return rc;
}

int msm_group_create(MsmGroup& g, const MsmSectionIn* secs, int nsec, const MsmConfig& cfg) {
  if (nsec < 1 || nsec > kMsmMaxSections) { set_error("msm: 1..3 sections per group"); return G16_E_ARG; }
  g = MsmGroup();
  g.nsec = nsec;
  g.dense = cfg.dense;
  bool has_g1 = false;
  int g2_sec = -1;
  for (int s = 0; s < nsec; s++) {
    if (secs[s].bases_host) has_g1 = true;
    if (secs[s].bases2_host) {
      if (g2_sec >= 0) { set_error("msm: one G2 lane per group"); return G16_E_ARG; }
      g2_sec = s;
    }
    if (!secs[s].bases_host && !secs[s].bases2_host) { set_error("msm: section without bases"); return G16_E_ARG; }
  }
  if (has_g1)
    for (int s = 0; s < nsec; s++)
      if (!secs[s].bases_host) { set_error("msm: a G2-only section cannot share a group with G1 sections"); return G16_E_ARG; }
  // compaction: a point is kept when it is not infinity (both twins agree in any well-formed key; a point that
  // is infinity in exactly one of the two tables is rejected -- the caller then builds two separate groups)
  std::vector<uint32_t> src;
  std::vector<uint8_t> canon1, canon2;
  uint64_t total = 0;
  for (int s = 0; s < nsec; s++) total += secs[s].n_total;
  src.reserve(total);
  for (int s = 0; s < nsec; s++) {
    const MsmSectionIn& in = secs[s];
    g.sec_begin[s] = (uint32_t)src.size();
    for (uint32_t i = 0; i < in.n_total; i++) {
      const bool inf1 = in.bases_host ? is_inf_bytes(in.bases_host + (size_t)i * 64, 64) : true;
      const bool inf2 = in.bases2_host ? is_inf_bytes(in.bases2_host + (size_t)i * 128, 128) : true;
      if (in.bases_host && in.bases2_host && inf1 != inf2) {
        set_error("msm: G1/G2 twin sections disagree on a point at infinity");
        return G16_E_FORMAT;
      }
      if (in.bases_host ? inf1 : inf2) continue;
      src.push_back(in.scalar_offset + i);
      if (in.bases_host) canon1.insert(canon1.end(), in.bases_host + (size_t)i * 64, in.bases_host + (size_t)i * 64 + 64);
      if (in.bases2_host) canon2.insert(canon2.end(), in.bases2_host + (size_t)i * 128, in.bases2_host + (size_t)i * 128 + 128);
    }
    g.sec_n[s] = (uint32_t)src.size() - g.sec_begin[s];
  }
  for (int s = nsec; s < kMsmMaxSections; s++) g.sec_begin[s] = (uint32_t)src.size();
  g.n = (uint32_t)src.size();
  g.src_identity = true;
  for (size_t k = 0; k < src.size() && g.src_identity; k++) g.src_identity = (src[k] == (uint32_t)k);
  if (g2_sec >= 0 && g.sec_n[g2_sec] == 0) g2_sec = -1;   // nothing rides: the G2 sum is the point at infinity
  g.g2_sec = g2_sec;
  if (g.n >= 0x40000000u) { set_error("msm: too many bases"); return G16_E_ARG; }
  // ---- geometry
  // witness groups: only ~1/3 of the scalars are full-width -> size the windows for that
  uint32_t n_max = 0;
  for (int s = 0; s < nsec; s++) n_max = g.sec_n[s] > n_max ? g.sec_n[s] : n_max;
  const uint32_t n_eff = cfg.dense ? n_max : n_max / 3 + 1;
  int pf_req = cfg.precomp;
  if (pf_req <= 0) {
    const char* e = getenv("G16_PRECOMP");
    pf_req = e ? atoi(e) : 0;
  }
  // default: full window precomputation for a dense single-section group that is large enough to pay for its
  // table (Ws x 64 B per point, e.g. 1.75 GB for the 2^21-point H section; capped at 8 GiB)
  bool full = false;
  if (nsec == 1 && g.n > 0) {
    if (pf_req == 0) full = cfg.dense && g.n >= (1u << 15);
    else if (pf_req >= 255) full = true;
  }
  g.c = cfg.c ? cfg.c : choose_c(n_eff ? n_eff : 1, full);
  // witness groups: c = 15/16 would minimise the addition count by a few percent, but every bucket costs scan,
  // queue and reduce work on the latency-bound part of the chain: stay at <= 13 (4096 buckets per row)
  if (!cfg.c && !cfg.dense && g.c > 13) g.c = 13;
  if (g.c < 2 || g.c > 22) { set_error("msm: window bits must be in [2,22]"); return G16_E_ARG; }
  // scalar windows: the top one stays unsigned and must fit the buckets: 254 - c (Ws - 1) <= c - 1
  g.Ws = (255 + g.c - 1) / g.c;
  uint32_t pf = 1;
  if (nsec == 1) {
    if (full) pf = (uint32_t)g.Ws;
    else if (pf_req > 1) pf = (uint32_t)pf_req;
    if (pf > (uint32_t)g.Ws) pf = (uint32_t)g.Ws;
    const size_t pt = (g2_sec >= 0 && !has_g1) ? 128 : 64;
    while (pf > 1 && (uint64_t)pf * g.n * pt > (8ull << 30)) pf--;
  }
  g.W = (g.Ws + (int)pf - 1) / (int)pf;
  g.pf = (uint32_t)((g.Ws + g.W - 1) / g.W);   // drop empty trailing levels (Ws = 16, pf = 5 -> W = 4, pf = 4)
  if ((uint64_t)g.pf * g.n >= 0x7fffffffull) { set_error("msm: too many precomputed bases"); return G16_E_ARG; }
  g.B = 1u << (g.c - 1);
  g.wb = (uint32_t)g.c;
  g.wx = 0;
  {
    // even windows for a fully precomputed group (see MsmGroup::wb): the top window, unsigned, must stay below c bits
    const bool uneven_off = getenv("G16_UNIFORM_WINDOWS") && atoi(getenv("G16_UNIFORM_WINDOWS"));
    if (g.pf == (uint32_t)g.Ws && g.W == 1 && g.Ws > 1 && g.Ws * g.c > 255 && !uneven_off) {
      g.wb = 255u / (uint32_t)g.Ws;
      g.wx = 255u - g.wb * (uint32_t)g.Ws;
    }
  }
  {
    // window-ordered buckets (MsmGroup::wkb): opt-in, G16_WINDOW_ORDER=1 -- measured, no gain (see there)
    const bool on = getenv("G16_WINDOW_ORDER") && atoi(getenv("G16_WINDOW_ORDER"));
    if (g.pf == (uint32_t)g.Ws && g.W == 1 && g.Ws > 1 && g.Ws <= 16 && on) g.wkb = 4;
  }
  g.ones = !cfg.dense;
  {
    // salted top window (see MsmGroup::salt_bits): only without window precomputation (a row = one window), when
    // the top window is narrow enough to leave >= 4 salt bits (reduce segments of up to 16 buckets stay inside one
    // weight)
    const int top_bits = 254 - g.c * (g.Ws - 1);
    const bool salt_off = getenv("G16_NO_SALT") && atoi(getenv("G16_NO_SALT"));
    if (g.pf == 1 && top_bits >= 1 && (g.c - 1) - top_bits >= 4 && !salt_off) g.salt_bits = (uint32_t)((g.c - 1) - top_bits);
  }
  // dup rows (see MsmGroup::dup_rows): witness groups with at least 1024 buckets per row; 2^14 hash buckets per section
  {
    const bool dup_off = getenv("G16_NO_DUP") && atoi(getenv("G16_NO_DUP"));
    if (!cfg.dense && g.pf == 1 && g.B >= 1024 && !dup_off) {
      g.dup_rows = g.B >= 16384 ? 1u : 16384u / g.B;
      g.dup_bits = 0;
      while ((1u << g.dup_bits) < g.dup_rows * g.B) g.dup_bits++;
      // chunk width = window width (c <= 16): chunk k of a repeated value then weighs 2^(c k) like window k, its sum joins
      // the window's row sum on the host and ONE Horner pass does for both (msm_collect) -- r03: the host's share of a
      // proof was 0.8 ms, half of it the 240 doublings per section and curve of a second Horner over 16-bit chunks
      int ch = cfg.dup_chunk ? cfg.dup_chunk : (g.c <= 16 ? g.c : (int)kDupChunkBits);
      if (const char* e = getenv("G16_DUP_CHUNK")) ch = atoi(e);
      if (ch >= 2 && ch <= 16) g.dup_chunk = (uint32_t)ch;
      g.dup_chunk_wide = (getenv("G16_DUP_CHUNK") || cfg.dup_chunk) ? g.dup_chunk : (uint32_t)kDupChunkBits;
      g.dup_rows_cap = (254 + std::min(g.dup_chunk, g.dup_chunk_wide) - 1) / std::min(g.dup_chunk, g.dup_chunk_wide);
    }
  }
  g.rps = (uint32_t)g.W + (g.ones ? 1u : 0u) + g.dup_rows;
  g.rows = (uint32_t)nsec * g.rps;
  // two-level sort: bucket = bin << low_bits | low; 8 low bits unless that leaves too many (row, bin) counters
  // for the LDS of the binning passes (<= 12288)
  // (witness groups: 6 -- their bins are uneven (small scalars crowd the low buckets of window 0, r02: the largest
  // bin 3x the average), and one workgroup sorts one bin)
  uint32_t want_low = cfg.dense ? 8u : 6u;
  if (const char* e = getenv("G16_LOW_BITS")) {
    int a = 0, b = 0;
    const int k = sscanf(e, "%d,%d", &a, &b);
    if (!cfg.dense && k >= 1 && a > 0) want_low = (uint32_t)a;
    if (cfg.dense && k >= 2 && b > 0) want_low = (uint32_t)b;
  }
  g.low_bits = (uint32_t)(g.c - 1) < want_low ? (uint32_t)(g.c - 1) : want_low;
  while (g.low_bits < kMaxLowBits && (uint64_t)g.rows * (g.B >> g.low_bits) > 12288) g.low_bits++;
  if (g.low_bits + g.wkb > kMaxLowBits) g.wkb = 0;
  g.bins = g.B >> g.low_bits;
  if ((uint64_t)g.rows * g.bins > 12288) { set_error("msm: window bits too large for this group"); return G16_E_ARG; }
  // Task length of the G1 lane (the G2 lane derives its own in lane_create): enough tasks to fill ~256k lanes
  // (256 CUs x 4 SIMDs x 4 waves x 64), within [16, 32]: short tasks keep the drain tail of the persistent kernel
  // small (sweep r01)
  if (cfg.task_len) {
    g.task_len = (uint32_t)cfg.task_len;
  } else {
    uint64_t entries = 0;
    for (int s = 0; s < nsec; s++) entries += (uint64_t)(cfg.dense ? g.sec_n[s] : g.sec_n[s] / 3 + 1) * g.Ws;
    g.task_len = msm_task_len_for(entries / 262144, 16);
  }
  g.task_len_forced = cfg.task_len != 0;
  g.max_entries = (uint64_t)g.n * (uint32_t)g.Ws;
  if (g.max_entries >= 0x7fffffffull) { set_error("msm: too many bucket entries"); return G16_E_ARG; }
  // front-end chunks: >= 4096 points per 256-thread workgroup (long (row, bin) runs), at most 1024 workgroups
  {
    uint32_t per = 4096;
    if (const char* e = getenv("G16_BIN_PER")) per = (uint32_t)atoi(e) >= 256 ? (uint32_t)atoi(e) : 256;
    uint64_t chunks = ((uint64_t)g.n + per - 1) / per;
    if (chunks > 1024) chunks = 1024;
    if (chunks < 1) chunks = 1;
    g.chunks = (uint32_t)chunks;
    g.per = (uint32_t)(((uint64_t)g.n + chunks - 1) / chunks);
    if (g.per == 0) g.per = 1;
  }
  if (g.n == 0) return G16_OK;
  // ---- resident tables
  G16_HIP(hipMalloc(&g.d_src, (size_t)g.n * 4));
  G16_HIP(hipMemcpy(g.d_src, src.data(), (size_t)g.n * 4, hipMemcpyHostToDevice));
  int rc = G16_OK;
  std::vector<int> ndbl(g.pf, g.c * g.W);   // level k + 1 = 2^(bits of the windows of level k) * level k
  if (g.W == 1) {
    const MsmPlan pl = msm_plan_of(g);
    for (uint32_t k = 0; k < g.pf; k++) ndbl[k] = (int)msm_win_bits(pl, k);
  }
  if (has_g1) rc = upload_table(1, canon1, g.n, g.pf, ndbl, &g.d_bases);
  if (!rc && g2_sec >= 0) rc = upload_table(2, canon2, g.sec_n[g2_sec], g.pf, ndbl, &g.d_bases2);
  return rc;
}

void msm_group_destroy(MsmGroup& g) {
  if (g.d_bases) (void)hipFree(g.d_bases);
  if (g.d_bases2) (void)hipFree(g.d_bases2);
  if (g.d_src) (void)hipFree(g.d_src);
  g.d_bases = g.d_bases2 = nullptr;
  g.d_src = nullptr;
  g.n = 0;
}

static thread_local int t_aux_prio = kMsmPrioHighest;
void msm_set_aux_stream_priority(int prio) { t_aux_prio = prio; }

static int lane_create(MsmLaneWs& ln, const MsmGroup& g, int curve, uint32_t key_lo, uint32_t key_hi, uint32_t point_base,
                       uint64_t entries, uint64_t entries_eff) {
  ln.active = true;
  ln.curve = curve;
  ln.key_lo = key_lo;
  ln.key_hi = key_hi;
  ln.point_base = point_base;
  ln.rows = (key_hi - key_lo) / g.B;
  const uint64_t nbk = (uint64_t)key_hi - key_lo;
  // task length: fill the lane's persistent grid (G1: 4 wavefronts per SIMD = 262144 lanes, G2: 2 = 131072)
  ln.task_len = g.task_len;
  if (curve == 2 && !g.task_len_forced) {
    ln.task_len = msm_task_len_for(entries_eff / 131072, 16);
  }
  ln.seg_len = msm_seg_len_cfg(curve == 2 ? 1 : (g.dense ? 2 : 0));
  while (ln.seg_len & (ln.seg_len - 1)) ln.seg_len &= ln.seg_len - 1;   // a power of two (msm_bucket_reduce_kernel)
  if (g.salt_bits) {   // a segment must not straddle two weights of the salted top window
    while (ln.seg_len > (1u << g.salt_bits) || ((1u << g.salt_bits) % ln.seg_len) != 0) ln.seg_len >>= 1;
    if (ln.seg_len == 0) ln.seg_len = 1;
  }
  // A launch re-derives the task length from the sorted entries the PREVIOUS launch of this workspace really had
  // (msm_build_queue): the create-time estimate above knows the points, not the witness -- the real NZCP witness
  // leaves ~1.5 M entries of an estimated 12.8 M after the zero / one / repeated-value classes, and 32-entry tasks
  // then fill a fifth of the persistent grid (r02 sweep: 4.40 -> 4.18 ms per proof at 16; the synthetic witness, with
  // its 32 % full-width scalars, wants 32: 7.9 against 9.4 ms).
  ln.task_len_min = ln.task_len < 16u ? ln.task_len : 16u;   // (8 on the G2 lane: measured worse, r02 sweep)
  if (g.task_len_forced) ln.task_len_min = ln.task_len;
  // every non-empty bucket has <= 1 short task + entries / task_len full ones
  ln.max_tasks = nbk + entries / ln.task_len_min + 64;
  // test hook (tests/test_gpu_edges.py): undersized task buffers, to see the device-side capacity check fire
  if (const char* e = getenv("G16_TEST_MAX_TASKS"))
    if (atoll(e) > 0 && (uint64_t)atoll(e) < ln.max_tasks) ln.max_tasks = (uint64_t)atoll(e);
  G16_HIP(hipHostMalloc((void**)&ln.h_stat, 64));
  for (int k = 0; k < 16; k++) ln.h_stat[k] = 0;
  const size_t pb = curve == 2 ? sizeof(G2XYZZ29) : sizeof(G1XYZZ29);   // device-side (lazy) points
  const size_t cpb = msm_point_bytes(curve);                             // canonical, host-visible
  const size_t ntiles = (size_t)nbk / kScanTile + 2;
  G16_HIP(hipMalloc(&ln.d_off, (nbk + 4) * 4));
  G16_HIP(hipMalloc(&ln.d_toff, (nbk + 4) * 4));
  G16_HIP(hipMalloc(&ln.d_foff, (nbk + 4) * 4));
  G16_HIP(hipMalloc(&ln.d_tile_a, ntiles * 4));
  G16_HIP(hipMalloc(&ln.d_tile_b, ntiles * 4));
  G16_HIP(hipMalloc(&ln.d_tile_c, ntiles * 4));
  G16_HIP(hipMalloc(&ln.d_task_desc, (ln.max_tasks + 1) * sizeof(uint2)));
  G16_HIP(hipMalloc(&ln.d_qdesc, (ln.max_tasks + 1) * sizeof(uint4)));
  G16_HIP(hipMalloc(&ln.d_class, 2 * kRemClasses * 4));
  G16_HIP(hipMalloc(&ln.d_queue, 64));
  G16_HIP(hipMalloc(&ln.d_redo, (ln.max_tasks + 1) * 4));
  G16_HIP(hipMalloc(&ln.d_partial, ln.max_tasks * pb + 256));
  G16_HIP(hipMalloc(&ln.d_bsum, nbk * pb + 256));
  ln.max_heavy = (uint32_t)(ln.max_tasks / kLightTasks + 16);
  G16_HIP(hipMalloc(&ln.d_heavy, ((size_t)ln.max_heavy + 2) * 4));
  G16_HIP(hipMalloc(&ln.d_medium, ((size_t)ln.max_heavy + 2) * 4));
  // dense rows: the scan-based reduce; sparse rows (witness lanes): r02's per-lane weighting (see msm.cuh).  G16_REDUCE_SCAN
  // = 0 / 1 forces one of them for every lane (sweeps).
  {
    const int force = getenv("G16_REDUCE_SCAN") ? atoi(getenv("G16_REDUCE_SCAN")) : -1;   // (read per handle: tests flip it)
    ln.reduce_scan = force < 0 ? g.dense : force != 0;
  }
  // Throughput mode (g16_prove_batch: the device is the bottleneck, not the depth of a proof's chains): dense rows take
  // segments twice as long -- (32 + 19) / 16 additions per bucket instead of (16 + 19) / 8, a quarter of the reduce's
  // instructions less, on half the wavefronts: H reduce 0.36 -> 0.50 ms for a single proof, 512-proof batches 305-307 ->
  // 311-312 proofs/s (profiles/r03_sweeps.txt 20).  Buffers are sized for the shorter segments.
  ln.seg_len_lat = ln.seg_len_thr = ln.seg_len;
  if (ln.reduce_scan && g.dense && !getenv("G16_SEG_LEN")) {
    uint32_t t = ln.seg_len * 2;
    if (g.salt_bits)
      while (t > ln.seg_len && (t > (1u << g.salt_bits) || ((1u << g.salt_bits) % t) != 0)) t >>= 1;
    if (t * 64u <= g.B) ln.seg_len_thr = t;   // (a row of at least one wavefront of segments)
  }
  const MsmReducePlan rp = msm_reduce_plan(g, ln);
  if (ln.reduce_scan) {
    G16_HIP(hipMalloc(&ln.d_seg, 2 * (size_t)ln.rows * rp.nwg * pb + 256));
  } else {
    const uint64_t nseg = (g.B + ln.seg_len - 1) / ln.seg_len;
    G16_HIP(hipMalloc(&ln.d_seg, (size_t)ln.rows * nseg * pb + 256));
    G16_HIP(hipMalloc(&ln.d_red, 2 * (size_t)ln.rows * ((nseg + 63) / 64) * pb + 256));
  }
  ln.nsec_lane = ln.rows / g.rps;
  ln.row_pts = ln.reduce_scan ? ln.rows * msm_row_out_points(rp) : ln.rows;
  size_t out_pts = ln.row_pts;
  if (g.dup_rows) {
    const size_t nchunk = ((size_t)1 << g.dup_bits) >> 6, drows = (size_t)ln.nsec_lane * g.dup_rows_cap;
    out_pts += drows;
    G16_HIP(hipMalloc(&ln.d_dseg, drows * nchunk * pb + 256));
    G16_HIP(hipMalloc(&ln.d_dred, 2 * drows * ((nchunk + 63) / 64) * pb + 256));
    G16_HIP(hipMalloc(&ln.d_dcount, 64));
    if (curve == 2) {   // G2: its own stream beside the bucket reduce (G1's dup stage is short: it stays on the lane's
      int plo = 0, phi = 0;   // stream -- hardware queues are scarce, see prover.cpp create_impl)
      G16_HIP(hipDeviceGetStreamPriorityRange(&plo, &phi));
      if (t_aux_prio != kMsmPrioHighest) phi = t_aux_prio;
      G16_HIP(hipStreamCreateWithPriority(&ln.st_dup, hipStreamNonBlocking, phi));
    }
    G16_HIP(hipEventCreateWithFlags(&ln.ev_dup_fork, hipEventDisableTiming));
    G16_HIP(hipEventCreateWithFlags(&ln.ev_dup_join, hipEventDisableTiming));
    G16_HIP(hipMalloc(&ln.d_dlist, ((size_t)ln.nsec_lane << g.dup_bits) * 4 + 64));
  }
  ln.out_bytes = out_pts * cpb;
  G16_HIP(hipHostMalloc((void**)&ln.h_pinned, ln.out_bytes + 256));
  // the last kernels of a lane write their few KB of row sums straight into the pinned host block (it is mapped into the
  // device's address space): no copy-engine hop between the last kernel and the host's wake-up.  G16_ROWS_COPY=1: r02's
  // device buffer + hipMemcpyAsync
  ln.rows_mapped = !(getenv("G16_ROWS_COPY") && atoi(getenv("G16_ROWS_COPY")));
  if (ln.rows_mapped) G16_HIP(hipHostGetDevicePointer(&ln.d_canon, ln.h_pinned, 0));
  else G16_HIP(hipMalloc(&ln.d_canon, ln.out_bytes + 256));
  G16_HIP(hipEventCreate(&ln.ev0));
  G16_HIP(hipEventCreate(&ln.ev1));
  G16_HIP(hipEventCreate(&ln.ev_done));
  return G16_OK;
}

static void lane_destroy(MsmLaneWs& ln) {
  void* ptrs[] = {ln.d_task_desc, ln.d_qdesc, ln.d_class, ln.d_queue, ln.d_redo, ln.d_partial, ln.d_bsum, ln.d_heavy, ln.d_medium,
                  ln.d_seg, ln.d_red, ln.rows_mapped ? nullptr : ln.d_canon, ln.d_off, ln.d_toff, ln.d_foff, ln.d_tile_a, ln.d_tile_b, ln.d_tile_c,
                  ln.d_dseg, ln.d_dred, ln.d_dcount, ln.d_dlist};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  if (ln.h_pinned) (void)hipHostFree(ln.h_pinned);
  if (ln.h_stat) (void)hipHostFree(ln.h_stat);
  if (ln.st_dup) (void)hipStreamDestroy(ln.st_dup);
  hipEvent_t evs[] = {ln.ev0, ln.ev1, ln.ev_done, ln.trace_ev[0], ln.trace_ev[1], ln.trace_ev[2], ln.trace_ev[3],
                      ln.ev_dup_fork, ln.ev_dup_join};
  for (hipEvent_t e : evs)
    if (e) (void)hipEventDestroy(e);
  ln = MsmLaneWs();
}

int msm_workspace_create(MsmWorkspace** out, const MsmGroup& g) {
  MsmWorkspace* ws = new MsmWorkspace();
  *out = ws;
  if (g.n == 0) return G16_OK;
  msm_set_throughput(ws, g, false);
  const uint32_t nrb = g.rows * g.bins;
  ws->nb = g.rows * g.B;
  G16_HIP(hipMalloc(&ws->d_hist, ((size_t)g.chunks * nrb + 4) * 4));
  G16_HIP(hipMalloc(&ws->d_bin_cnt, ((size_t)nrb + 4) * 4));
  G16_HIP(hipMalloc(&ws->d_bin_start, ((size_t)nrb + 4) * 4));
  {
    // single-pass front end for dense single-section groups: fixed-capacity bins, 2.5x the average population (even
    // windows fill the low buckets ~1.4x the average, msm_bin_direct_kernel) + slack for small groups
    const bool off = getenv("G16_NO_DIRECT_BIN") && atoi(getenv("G16_NO_DIRECT_BIN"));
    uint64_t cap = (g.max_entries * 5 / 2) / nrb + 1024;
    if (const char* e = getenv("G16_TEST_BIN_CAP"))   // test hook (tests/test_gpu_edges.py): bins that overflow at once
      if (atoll(e) > 0) cap = (uint64_t)atoll(e);
    ws->direct = g.dense && g.nsec == 1 && !g.ones && !g.dup_rows && !off && cap * nrb < 0x7fffffffull;
    ws->bin_cap = ws->direct ? (uint32_t)cap : 0u;
  }
  G16_HIP(hipHostMalloc((void**)&ws->h_over, 64));
  ws->h_over[0] = 0;
  // (the two-pass path -- the fallback of an overflowing single-pass launch -- needs room for every entry whatever the capacity)
  uint64_t tmp_entries = g.max_entries;
  if (ws->direct && (uint64_t)ws->bin_cap * nrb > tmp_entries) tmp_entries = (uint64_t)ws->bin_cap * nrb;
  G16_HIP(hipMalloc(&ws->d_tmp, (tmp_entries + 4) * sizeof(uint2)));
  G16_HIP(hipMalloc(&ws->d_sorted, (g.max_entries + 4) * 4));
  G16_HIP(hipMalloc(&ws->d_cnt, ((size_t)ws->nb + 4) * 4));
  if (g.dup_rows) {
    const size_t nd = (size_t)g.nsec << g.dup_bits;
    G16_HIP(hipMalloc(&ws->d_dup_cnt, nd * 4));
    G16_HIP(hipMalloc(&ws->d_dup_rep, nd * 4));
    G16_HIP(hipMalloc(&ws->d_dup_mixed, nd * 4));
  }
  G16_HIP(hipEventCreate(&ws->ev_sorted));
  int rc = G16_OK;
  const uint32_t div = g.dense ? 1 : 3;   // effective (full-width) share of the scalars
  if (g.d_bases) rc = lane_create(ws->lane[0], g, 1, 0, ws->nb, 0, g.max_entries, g.max_entries / div);
  if (!rc && g.g2_sec >= 0) {
    const uint32_t lo = (uint32_t)g.g2_sec * g.rps * g.B;
    const uint64_t e2 = (uint64_t)g.sec_n[g.g2_sec] * (uint32_t)g.Ws;
    rc = lane_create(ws->lane[1], g, 2, lo, lo + g.rps * g.B, g.pf > 1 ? 0u : g.sec_begin[g.g2_sec], e2, e2 / div);
  }
  return rc;
}

void msm_workspace_destroy(MsmWorkspace* ws) {
  if (!ws) return;
  void* ptrs[] = {ws->d_hist, ws->d_bin_cnt, ws->d_bin_start, ws->d_tmp, ws->d_sorted, ws->d_cnt,
                  ws->d_dup_cnt, ws->d_dup_rep, ws->d_dup_mixed};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  for (auto& ln : ws->lane) lane_destroy(ln);
  if (ws->h_over) (void)hipHostFree(ws->h_over);
  if (ws->ev_sorted) (void)hipEventDestroy(ws->ev_sorted);
  for (hipEvent_t e : ws->trace_ev)
    if (e) (void)hipEventDestroy(e);
  delete ws;
}

int msm_front_end(const MsmGroup& g, MsmWorkspace* ws, const Fr* d_scalars, hipStream_t st) {
  const MsmPlan pl = msm_plan_of(g);
  const uint32_t nrb = g.rows * g.bins;
  U256 K;
  msm_make_K(pl, K);
  static const bool trace = getenv("G16_TRACE_HOST") != nullptr;
  auto mark = [&](int k) {
    if (!trace) return;
    if (!ws->trace_ev[k]) (void)hipEventCreate(&ws->trace_ev[k]);
    (void)hipEventRecord(ws->trace_ev[k], st);
  };
  const size_t lds = (size_t)nrb * 4;
  ws->d_scalars = d_scalars;
  if (g.dup_rows) {
    const size_t nd = (size_t)g.nsec << g.dup_bits;
    G16_HIP(hipMemsetAsync(ws->d_dup_cnt, 0, nd * 4, st));
    G16_HIP(hipMemsetAsync(ws->d_dup_rep, 0xff, nd * 4, st));
    G16_HIP(hipMemsetAsync(ws->d_dup_mixed, 0, nd * 4, st));
    msm_dup_scan_kernel<0><<<(g.n + 255) / 256, 256, 0, st>>>(d_scalars, g.d_src, pl, ws->d_dup_cnt, ws->d_dup_rep, ws->d_dup_mixed);
    msm_dup_scan_kernel<1><<<(g.n + 255) / 256, 256, 0, st>>>(d_scalars, g.d_src, pl, ws->d_dup_cnt, ws->d_dup_rep, ws->d_dup_mixed);
  }
  if (ws->direct) {
    G16_HIP(hipMemsetAsync(ws->d_bin_cnt, 0, (size_t)nrb * 4, st));
    if (g.Ws == 13)
      msm_bin_direct_kernel<13><<<g.chunks, kBinThreads, lds, st>>>(d_scalars, g.src_identity ? nullptr : g.d_src, pl, K, g.per, ws->bin_cap, ws->d_bin_cnt,
                                                                   ws->h_over, ws->d_tmp);
    else
      msm_bin_direct_kernel<0><<<g.chunks, kBinThreads, lds, st>>>(d_scalars, g.src_identity ? nullptr : g.d_src, pl, K, g.per, ws->bin_cap, ws->d_bin_cnt,
                                                                  ws->h_over, ws->d_tmp);
    mark(0);
    mark(1);
    msm_bin_scan_kernel<<<1, 1024, 0, st>>>(ws->d_bin_cnt, nrb, ws->d_bin_start, ws->bin_cap);
    mark(2);
    msm_bin_sort_kernel<<<nrb, 256, 0, st>>>(ws->d_tmp, ws->d_bin_start, pl, ws->d_bin_cnt, ws->bin_cap, ws->h_over, ws->d_cnt,
                                             ws->d_sorted);
    mark(3);
  } else {
    msm_bin_pass_kernel<0><<<g.chunks, kBinThreads, lds, st>>>(d_scalars, g.d_src, pl, K, g.per, g.chunks, ws->d_hist, nullptr,
                                                               ws->d_dup_cnt, ws->d_dup_mixed, nullptr);
    mark(0);
    msm_bin_chunkscan_kernel<<<nrb, 64, 0, st>>>(ws->d_hist, g.chunks, ws->d_bin_cnt);
    msm_bin_scan_kernel<<<1, 1024, 0, st>>>(ws->d_bin_cnt, nrb, ws->d_bin_start, 0xffffffffu);
    mark(1);
    msm_bin_pass_kernel<1><<<g.chunks, kBinThreads, lds, st>>>(d_scalars, g.d_src, pl, K, g.per, g.chunks, ws->d_hist,
                                                               ws->d_bin_start, ws->d_dup_cnt, ws->d_dup_mixed, ws->d_tmp);
    mark(2);
    msm_bin_sort_kernel<<<nrb, 256, 0, st>>>(ws->d_tmp, ws->d_bin_start, pl, nullptr, 0u, nullptr, ws->d_cnt, ws->d_sorted);
    mark(3);
  }
  G16_HIP(hipGetLastError());
  G16_HIP(hipEventRecord(ws->ev_sorted, st));
  return G16_OK;
}

// Per lane: exclusive scans of its bucket populations (entry offsets, task ids, queue positions of the full-length
// tasks, for THIS lane's task length), then the task descriptors and the work queue.
int msm_build_queue(const MsmGroup& g, MsmWorkspace* ws, MsmLaneWs& ln, hipStream_t st) {
  const uint32_t nbk = ln.key_hi - ln.key_lo;
  const uint32_t* cnt = ws->d_cnt + ln.key_lo;
  // the lane's first entry = the start of the first bin of its first row
  const uint32_t* off_base = ws->d_bin_start + (size_t)(ln.key_lo / g.B) * g.bins;
  const uint32_t ntiles = (nbk + kScanTile - 1) / kScanTile;
  if (!g.task_len_forced && ln.h_stat[0] > ln.h_stat[1]) {   // entries of the previous launch (its copies completed
    const uint64_t e = ln.h_stat[0] - ln.h_stat[1];          // before msm_collect returned): one task per lane of the
    ln.task_len = msm_task_len_for(e / (ln.curve == 2 ? 131072u : 262144u), ln.task_len_min);   // persistent grid
  }
  msm_scan_tiles_kernel<<<ntiles, 256, 0, st>>>(cnt, nbk, ln.task_len, ln.d_tile_a, ln.d_tile_b, ln.d_tile_c);
  msm_scan_top_kernel<<<1, 1024, 0, st>>>(ln.d_tile_a, ln.d_tile_b, ln.d_tile_c, ntiles, off_base, ln.d_off + nbk,
                                          ln.d_toff + nbk, ln.d_foff + nbk, (uint32_t)ln.max_tasks,
                                          MsmSmallInit{ln.h_stat, ln.d_class, ln.d_queue, ln.d_heavy, ln.d_medium});
  msm_scan_apply_kernel<<<ntiles, 256, 0, st>>>(cnt, nbk, ln.task_len, ln.d_tile_a, ln.d_tile_b, ln.d_tile_c, off_base,
                                                ln.d_queue, ln.d_off, ln.d_toff, ln.d_foff, ln.d_class);
  msm_task_fill_kernel<<<(nbk + 255) / 256, 256, 0, st>>>(ln.d_off, ln.d_toff, ln.d_foff, nbk, ln.task_len, ln.d_task_desc,
                                                          ln.d_qdesc, ln.d_class, ln.d_class + kRemClasses,
                                                          (uint32_t)ln.max_tasks);
  G16_HIP(hipGetLastError());
  return G16_OK;
}

int msm_launch_front(const MsmGroup& g, MsmWorkspace* ws, const Fr* d_scalars, hipStream_t st) {
  ws->launched = true;
  ws->st_last = st;
  ws->st2_last = nullptr;
  ws->empty = (g.n == 0);
  for (auto& ln : ws->lane) ln.last_accum_ms = 0.f;
  if (g.n == 0) return G16_OK;
  return msm_front_end(g, ws, d_scalars, st);
}

int msm_launch_lanes(const MsmGroup& g, MsmWorkspace* ws, hipStream_t st, hipStream_t st2, hipEvent_t gate1,
                     hipEvent_t gate2) {
  if (g.n == 0) return G16_OK;
  int rc;
  ws->st2_last = st2;
  if (ws->lane[1].active) {
    hipStream_t s2 = st2 ? st2 : st;
    if (s2 != st) G16_HIP(hipStreamWaitEvent(s2, ws->ev_sorted, 0));
    ws->lane[1].gate = gate2;
    // the longer chain first when both share a stream
    if ((rc = msm_launch_lane_g2(g, ws, ws->lane[1], s2))) return rc;
  }
  if (ws->lane[0].active) {
    ws->lane[0].gate = gate1;
    rc = msm_launch_lane_t<Fq29Ops>(g, ws, ws->lane[0], g.d_bases, st);
    if (rc) return rc;
  }
  return G16_OK;
}

int msm_launch(const MsmGroup& g, MsmWorkspace* ws, const Fr* d_scalars, hipStream_t st, hipStream_t st2) {
  int rc = msm_launch_front(g, ws, d_scalars, st);
  if (rc) return rc;
  return msm_launch_lanes(g, ws, st, st2, nullptr, nullptr);
}

hipEvent_t msm_event(MsmWorkspace* ws, int which) {
  switch (which) {
    case 0: return ws->ev_sorted;
    case 1: return ws->lane[0].ev0;
    case 2: return ws->lane[0].ev1;
    case 3: return ws->lane[0].active ? ws->lane[0].ev_done : nullptr;
    default: return ws->lane[1].active ? ws->lane[1].ev_done : nullptr;
  }
}

int msm_collect(const MsmGroup& g, MsmWorkspace* ws, MsmResult* out, const std::function<void(const MsmResult&)>* after_g1) {
  for (auto& p : out->g1) xyzz_set_inf(p);
  xyzz_set_inf(out->g2);
  if (!ws->launched || ws->empty) return G16_OK;
  bool overflow = false;
  for (int l = 0; l < 2; l++) {
    MsmLaneWs& ln = ws->lane[l];
    if (!ln.active) continue;
    G16_HIP(hipEventSynchronize(ln.ev_done));
    static const bool trace = getenv("G16_TRACE_HOST") != nullptr;
    if (trace) fprintf(stderr, "[g16 tail] %s lane %d: host saw its row sums at %.3f ms\n", g.dup_rows ? "W" : "H", l, trace_ms());
    (void)hipEventElapsedTime(&ln.last_accum_ms, ln.ev0, ln.ev1);
    if (ln.h_stat[2] || ln.h_stat[3]) {   // (checked on the device, msm_scan_top_kernel / msm_combine_light_kernel)
      set_error(ln.h_stat[2] ? "msm: lane " + std::to_string(l) + " needed " + std::to_string(ln.h_stat[2]) +
                                   " bucket tasks, its buffers hold " + std::to_string(ln.max_tasks)
                             : "msm: lane " + std::to_string(l) + " overflowed its heavy-bucket list");
      overflow = true;
      continue;   // (drain the other lane as well before reporting)
    }
    if (g.dup_rows && getenv("G16_DEBUG_DUP")) {
      uint32_t dc[4] = {0, 0, 0, 0};
      (void)hipMemcpy(dc, ln.d_dcount, 16, hipMemcpyDeviceToHost);
      fprintf(stderr, "[g16 dup] lane %d: repeated-value buckets per section: %u %u %u\n", l, dc[0], dc[1], dc[2]);
    }
    const MsmReducePlan rp = msm_reduce_plan(g, ln);
    const uint32_t ngroups = (rp.nwg + kPairGroup - 1) / kPairGroup;
    if (l == 0) {
      const G1XYZZ* rows = reinterpret_cast<const G1XYZZ*>(ln.h_pinned);
      std::vector<G1XYZZ> folded;
      if (ln.reduce_scan && rp.nwg > kPairGroup) {   // long rows (the H-MSM's): the device left (P, Y, Wt) triples per pair group
        folded.resize(ln.rows);
        for (uint32_t r = 0; r < ln.rows; r++)
          msm_fold_row<FqOps>(folded[r], rows + (size_t)r * ngroups * 3, ngroups, rp, msm_row_kind(rp, r));
      }
      const G1XYZZ* rsum = !folded.empty() ? folded.data() : rows;
      const bool merged = g.dup_rows && ws->dup_chunk == (uint32_t)g.c && ws->dup_bit_rows == (uint32_t)g.W;
      std::vector<G1XYZZ> wrow;
      for (int s = 0; s < g.nsec; s++) {
        const G1XYZZ* rs = rsum + (size_t)s * g.rps;
        const G1XYZZ* dup = rows + ln.row_pts + (size_t)s * ws->dup_bit_rows;
        if (merged) {   // chunk k of the repeated values weighs like window k: one Horner pass
          wrow.assign(rs, rs + g.W + (g.ones ? 1 : 0));
          for (int j = 0; j < g.W; j++) xyzz_add(wrow[j], dup[j]);
          rs = wrow.data();
        }
        msm_combine_windows<FqOps>(out->g1[s], rs, g.W, g.c, g.ones);
        if (g.dup_rows && !merged) msm_add_bit_sums<FqOps>(out->g1[s], dup, ws->dup_chunk, ws->dup_bit_rows);
      }
      if (after_g1 && *after_g1) (*after_g1)(*out);
    } else {
      const G2XYZZ* rows = reinterpret_cast<const G2XYZZ*>(ln.h_pinned);
      std::vector<G2XYZZ> folded;
      if (ln.reduce_scan && rp.nwg > kPairGroup) {
        folded.resize(ln.rows);
        for (uint32_t r = 0; r < ln.rows; r++)
          msm_fold_row<Fq2Ops>(folded[r], rows + (size_t)r * ngroups * 3, ngroups, rp, msm_row_kind(rp, r));
      }
      const G2XYZZ* rsum = !folded.empty() ? folded.data() : rows;
      const bool merged = g.dup_rows && ws->dup_chunk == (uint32_t)g.c && ws->dup_bit_rows == (uint32_t)g.W;
      std::vector<G2XYZZ> wrow;
      if (merged) {
        wrow.assign(rsum, rsum + g.W + (g.ones ? 1 : 0));
        for (int j = 0; j < g.W; j++) xyzz_add(wrow[j], rows[ln.row_pts + j]);
        rsum = wrow.data();
      }
      msm_combine_windows<Fq2Ops>(out->g2, rsum, g.W, g.c, g.ones);
      if (g.dup_rows && !merged) msm_add_bit_sums<Fq2Ops>(out->g2, rows + ln.row_pts, ws->dup_chunk, ws->dup_bit_rows);
    }
  }
  if (overflow) return G16_E_STATE;
  if (ws->direct && ws->h_over[0]) {
    // a bin of the single-pass front end overflowed (far-from-uniform scalars): its entries were dropped -- repeat the
    // launch on the two-pass path, which has no capacity, and stay there (the workload is what it is)
    ws->direct = false;
    ws->h_over[0] = 0;
    int rc = msm_launch_front(g, ws, ws->d_scalars, ws->st_last);
    if (!rc) rc = msm_launch_lanes(g, ws, ws->st_last, ws->st2_last, nullptr, nullptr);
    if (rc) return rc;
    return msm_collect(g, ws, out, after_g1);
  }
  return G16_OK;
}

float msm_last_accum_ms(const MsmWorkspace* ws, int lane) { return ws->lane[lane & 1].last_accum_ms; }
void msm_set_throughput(MsmWorkspace* ws, const MsmGroup& g, bool on) {
  if (!ws) return;
  ws->dup_chunk = on ? g.dup_chunk_wide : g.dup_chunk;
  ws->dup_bit_rows = (254 + ws->dup_chunk - 1) / ws->dup_chunk;
  for (auto& ln : ws->lane)
    if (ln.active) ln.seg_len = on ? ln.seg_len_thr : ln.seg_len_lat;
}
void msm_set_waves(MsmWorkspace* ws, uint32_t waves_g1, uint32_t waves_g2) {
  ws->lane[0].waves_per_simd = waves_g1;
  ws->lane[1].waves_per_simd = waves_g2;
}
void msm_set_quota(MsmWorkspace* ws, uint32_t quota_g1, uint32_t quota_g2) {
  ws->lane[0].chunk_quota = quota_g1;
  ws->lane[1].chunk_quota = quota_g2;
}
// which: 0..3 front-end marks (lane ignored), 4/5 accumulate start/end, 6..9 the lane's marks
float msm_event_offset_ms(MsmWorkspace* ws, hipEvent_t base, int lane, int which) {
  float t = 0.f;
  const MsmLaneWs& ln = ws->lane[lane & 1];
  hipEvent_t e = which < 4 ? ws->trace_ev[which] : which == 4 ? ln.ev0 : which == 5 ? ln.ev1 : ln.trace_ev[(which - 6) & 3];
  if (e && hipEventQuery(e) == hipSuccess) (void)hipEventElapsedTime(&t, base, e);
  return t;
}

}  // namespace g16
