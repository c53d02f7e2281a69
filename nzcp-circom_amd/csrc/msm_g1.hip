// G1 instantiation of the MSM kernels + the curve-independent host plumbing (instances, workspace).
#include <string.h>

#include "msm.cuh"

namespace g16 {

size_t msm_point_bytes(int curve) { return curve == 2 ? sizeof(G2XYZZ) : sizeof(G1XYZZ); }

int msm_convert_bases_g2(const void* in, void* out, uint32_t n);
static int msm_convert_bases_g1(const void* in, void* out, uint32_t n) {
  msm_convert_bases_kernel<Fq29Ops><<<(n + 255) / 256, 256>>>((const G1Affine*)in, (PackedAffine<Fq29Ops>*)out, n);
  G16_HIP(hipGetLastError());
  G16_HIP(hipDeviceSynchronize());
  return G16_OK;
}

int msm_precompute_g2(const void* in, void* out, uint32_t n, int ndbl);
static int msm_precompute_g1(const void* in, void* out, uint32_t n, int ndbl) {
  msm_precompute_kernel<FqOps><<<(n + 255) / 256, 256>>>((const G1Affine*)in, (G1Affine*)out, n, ndbl);
  G16_HIP(hipGetLastError());
  G16_HIP(hipDeviceSynchronize());
  return G16_OK;
}

// window precomputation factor: MsmConfig::precomp, else G16_PRECOMP, else the default
static uint32_t choose_pf(const MsmConfig& cfg, int Ws) {
  int pf = cfg.precomp;
  if (pf <= 0) {
    const char* e = getenv("G16_PRECOMP");
    pf = e ? atoi(e) : 1;
  }
  if (pf < 1) pf = 1;
  if (pf > Ws) pf = Ws;
  return (uint32_t)pf;
}

static int choose_c(uint32_t n) {
  // minimise W * (n + 2.5 * 2^(c-1)) over c in [4, 16], W = ceil(256 / c).  For large n skip window
  // sizes whose TOP window holds only 1..5 bits of the 254-bit scalar (c = 14, 12, 11, 10, 9, 7, 6, 4):
  // its handful of buckets would each collect n / 2^bits entries (r01 sweep: c = 14 made the H-MSM 3x
  // slower).  c = 16 (14 top bits), 15 (top window empty), 13 (7 bits) and 8 (6 bits) remain.
  int best = 16;
  double best_cost = 1e300;
  for (int c = 4; c <= 16; c++) {
    const int W = (256 + c - 1) / c;
    const int top_bits = 254 - c * (W - 1);
    if (n >= 4096 && top_bits > 0 && top_bits < 6) continue;
    const double cost = (double)W * ((double)n + 2.5 * (double)(1u << (c - 1)));
    if (cost < best_cost) { best_cost = cost; best = c; }
  }
  return best;
}

int msm_instance_create(MsmInstance& m, int curve, const uint8_t* bases_host, uint32_t n_total,
                        uint32_t scalar_offset, const MsmConfig& cfg) {
  const size_t psz = curve == 2 ? sizeof(G2Affine) : sizeof(G1Affine);
  m.curve = curve;
  m.dense = cfg.dense;
  std::vector<uint32_t> src;
  std::vector<uint8_t> packed;
  src.reserve(n_total);
  packed.reserve((size_t)n_total * psz);
  for (uint32_t i = 0; i < n_total; i++) {
    const uint8_t* p = bases_host + (size_t)i * psz;
    bool inf = true;
    for (size_t k = 0; k < psz; k += 8) {
      uint64_t w;
      memcpy(&w, p + k, 8);
      if (w) { inf = false; break; }
    }
    if (inf) continue;
    src.push_back(scalar_offset + i);
    packed.insert(packed.end(), p, p + psz);
  }
  m.n = (uint32_t)src.size();
  if (m.n >= 0x7fffffffu) { set_error("msm: too many bases"); return G16_E_ARG; }
  // witness MSMs: only ~1/3 of the scalars are full-width -> size the windows for that
  const uint32_t n_eff = cfg.dense ? m.n : m.n / 3 + 1;
  m.c = cfg.c ? cfg.c : choose_c(n_eff ? n_eff : 1);
  if (m.c < 2 || m.c > 16) { set_error("msm: window bits must be in [2,16]"); return G16_E_ARG; }
  m.Ws = (256 + m.c - 1) / m.c;
  m.pf = choose_pf(cfg, m.Ws);
  m.W = (m.Ws + (int)m.pf - 1) / (int)m.pf;
  m.pf = (uint32_t)((m.Ws + m.W - 1) / m.W);   // drop empty trailing levels (Ws = 16, pf = 5 -> W = 4, pf = 4)
  m.n_ext = m.pf * m.n;
  if ((uint64_t)m.pf * m.n >= 0x7fffffffull) { set_error("msm: too many precomputed bases"); return G16_E_ARG; }
  m.nbuckets = 1u << (m.c - 1);
  // Task length: enough tasks to fill ~256k lanes (256 CUs x 4 SIMDs x 4 waves x 64), within [16, 256].
  if (cfg.task_len) {
    m.task_len = (uint32_t)cfg.task_len;
  } else {
    const uint64_t entries = (uint64_t)n_eff * m.Ws;
    uint64_t t = entries / 262144;
    m.task_len = (uint32_t)(t < 16 ? 16 : (t > 32 ? 32 : t));   // short tasks: small drain tail (sweep r01)
  }
  if (m.n) {
    // upload the canonical image, convert once on the device to the kernels' 9x29 representation
    void* tmp = nullptr;
    void* tmp2 = nullptr;
    const size_t lazy_pt = curve == 2 ? sizeof(PackedAffine<Fq2x29Ops>) : sizeof(PackedAffine<Fq29Ops>);
    G16_HIP(hipMalloc(&tmp, packed.size()));
    if (m.pf > 1) G16_HIP(hipMalloc(&tmp2, packed.size()));
    G16_HIP(hipMalloc(&m.d_bases, (size_t)m.n_ext * lazy_pt));
    G16_HIP(hipMalloc(&m.d_src, (size_t)m.n * 4));
    G16_HIP(hipMemcpy(tmp, packed.data(), packed.size(), hipMemcpyHostToDevice));
    G16_HIP(hipMemcpy(m.d_src, src.data(), (size_t)m.n * 4, hipMemcpyHostToDevice));
    int rc = G16_OK;
    for (uint32_t k = 0; k < m.pf && !rc; k++) {
      // level k = 2^(c W) * level k-1 (canonical, ping-pong between tmp and tmp2), then to the lazy format
      void* cur = (k & 1) ? tmp2 : tmp;
      void* nxt = (k & 1) ? tmp : tmp2;
      void* dst = (uint8_t*)m.d_bases + (size_t)k * m.n * lazy_pt;
      rc = curve == 2 ? msm_convert_bases_g2(cur, dst, m.n) : msm_convert_bases_g1(cur, dst, m.n);
      if (!rc && k + 1 < m.pf)
        rc = curve == 2 ? msm_precompute_g2(cur, nxt, m.n, m.c * m.W) : msm_precompute_g1(cur, nxt, m.n, m.c * m.W);
    }
    (void)hipFree(tmp);
    if (tmp2) (void)hipFree(tmp2);
    if (rc) return rc;
  }
  return G16_OK;
}

void msm_instance_destroy(MsmInstance& m) {
  if (m.d_bases) (void)hipFree(m.d_bases);
  if (m.d_src) (void)hipFree(m.d_src);
  m.d_bases = nullptr;
  m.d_src = nullptr;
  m.n = 0;
}

int msm_workspace_create(MsmWorkspace** out, const MsmInstance* insts, int ninst) {
  MsmWorkspace* ws = new MsmWorkspace();
  size_t part_bytes = 0, seg_bytes = 0, red_bytes = 0, pin_bytes = 0, bsum_bytes = 0;
  for (int i = 0; i < ninst; i++) {
    const MsmInstance& m = insts[i];
    const uint64_t nb = (uint64_t)(m.W + 1) * m.nbuckets;   // + the ones window
    const uint64_t entries = (uint64_t)m.n * m.Ws;
    const uint64_t tasks = nb + entries / (m.task_len >= 16 ? m.task_len / 4 : 4) + 64;   // worst case of the graded lengths
    const uint32_t sl = msm_seg_len(true) < msm_seg_len(false) ? msm_seg_len(true) : msm_seg_len(false);   // the shorter: more segments
    const uint64_t nseg = (m.nbuckets + sl - 1) / sl;
    const size_t pb = m.curve == 2 ? sizeof(G2XYZZ29) : sizeof(G1XYZZ29);   // device-side (lazy) points
    const size_t cpb = msm_point_bytes(m.curve);                             // canonical, host-visible
    if (entries > ws->max_entries) ws->max_entries = (uint32_t)entries;
    if (nb > ws->max_buckets) ws->max_buckets = (uint32_t)nb;
    if (tasks > ws->max_tasks) ws->max_tasks = (uint32_t)tasks;
    if (tasks * pb > part_bytes) part_bytes = tasks * pb;
    if (nb * pb > bsum_bytes) bsum_bytes = nb * pb;
    if ((m.W + 1) * nseg * pb > seg_bytes) seg_bytes = (m.W + 1) * nseg * pb;
    const size_t rb = 2 * (size_t)(m.W + 1) * ((nseg + 63) / 64) * pb;
    if (rb > red_bytes) red_bytes = rb;
    if ((size_t)(m.W + 1) * cpb > pin_bytes) pin_bytes = (size_t)(m.W + 1) * cpb;
  }
  *out = ws;
  // sort geometry: ~4 workgroups of 1024 threads per CU when the LDS histogram allows it
  size_t dig_words = 0, hist_words = 0;
  for (int i = 0; i < ninst; i++) {
    const MsmInstance& m = insts[i];
    const uint32_t WT = (uint32_t)m.W + 1;
    const size_t lds = (size_t)m.nbuckets * 4;
    uint32_t per_cu = (uint32_t)(160 * 1024 / (lds ? lds : 1));
    if (per_cu > 2) per_cu = 2;        // 1024-thread workgroups: at most 2 per CU
    if (per_cu < 1) per_cu = 1;
    uint32_t chunks = (256 * per_cu + WT - 1) / WT;
    const uint32_t max_chunks = m.n_ext / 4096 + 1;
    if (chunks > max_chunks) chunks = max_chunks;
    if (chunks > ws->chunks) ws->chunks = chunks;
    if ((size_t)WT * m.n_ext > dig_words) dig_words = (size_t)WT * m.n_ext;
    if ((size_t)WT * chunks * m.nbuckets > hist_words) hist_words = (size_t)WT * chunks * m.nbuckets;
  }
  G16_HIP(hipMalloc(&ws->d_dig, (dig_words + 4) * 4));
  G16_HIP(hipMalloc(&ws->d_hist, (hist_words + 4) * 4));
  G16_HIP(hipMalloc(&ws->d_cnt, ((size_t)ws->max_buckets + 1) * 4));
  G16_HIP(hipMalloc(&ws->d_off, ((size_t)ws->max_buckets + 1) * 4));
  G16_HIP(hipMalloc(&ws->d_toff, ((size_t)ws->max_buckets + 1) * 4));
  G16_HIP(hipMalloc(&ws->d_sorted, ((size_t)ws->max_entries + 1) * 4));
  G16_HIP(hipMalloc(&ws->d_task_desc, ((size_t)ws->max_tasks + 1) * sizeof(uint2)));
  G16_HIP(hipMalloc(&ws->d_qdesc, ((size_t)ws->max_tasks + 1) * sizeof(uint4)));
  G16_HIP(hipMalloc(&ws->d_foff, ((size_t)ws->max_buckets + 1) * 4));
  G16_HIP(hipMalloc(&ws->d_tile_c, ((size_t)ws->max_buckets / kScanTile + 2) * 4));
  G16_HIP(hipMalloc(&ws->d_class, 2 * 32 * 4));
  G16_HIP(hipMalloc(&ws->d_queue, 64));
  G16_HIP(hipMalloc(&ws->d_redo, ((size_t)ws->max_tasks + 1) * 4));
  G16_HIP(hipMalloc(&ws->d_tile_a, ((size_t)ws->max_buckets / kScanTile + 2) * 4));
  G16_HIP(hipMalloc(&ws->d_tile_b, ((size_t)ws->max_buckets / kScanTile + 2) * 4));
  G16_HIP(hipMalloc(&ws->d_partial, part_bytes + 256));
  G16_HIP(hipMalloc(&ws->d_bsum, bsum_bytes + 256));
  ws->max_heavy = ws->max_tasks / kLightTasks + 16;
  G16_HIP(hipMalloc(&ws->d_heavy, ((size_t)ws->max_heavy + 2) * 4));
  G16_HIP(hipMalloc(&ws->d_seg, seg_bytes + 256));
  G16_HIP(hipMalloc(&ws->d_red, red_bytes + 256));
  G16_HIP(hipHostMalloc((void**)&ws->h_pinned, pin_bytes + 256));
  G16_HIP(hipMalloc(&ws->d_canon, pin_bytes + 256));
  G16_HIP(hipEventCreate(&ws->ev0));
  G16_HIP(hipEventCreate(&ws->ev1));
  G16_HIP(hipEventCreate(&ws->ev_sorted));
  return G16_OK;
}

void msm_workspace_destroy(MsmWorkspace* ws) {
  if (!ws) return;
  void* ptrs[] = {ws->d_cnt, ws->d_off, ws->d_toff, ws->d_sorted, ws->d_task_desc, ws->d_queue, ws->d_tile_a, ws->d_tile_b,
                  ws->d_partial, ws->d_seg, ws->d_red, ws->d_bsum, ws->d_heavy, ws->d_dig, ws->d_hist, ws->d_canon, ws->d_redo,
                  ws->d_qdesc, ws->d_foff, ws->d_tile_c, ws->d_class};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  if (ws->h_pinned) (void)hipHostFree(ws->h_pinned);
  if (ws->ev0) (void)hipEventDestroy(ws->ev0);
  if (ws->ev1) (void)hipEventDestroy(ws->ev1);
  if (ws->ev_sorted) (void)hipEventDestroy(ws->ev_sorted);
  delete ws;
}

int msm_launch_g2(const MsmInstance& m, MsmWorkspace* ws, const Fr* d_scalars, hipStream_t st);

float msm_last_accum_ms(const MsmWorkspace* ws) { return ws->last_accum_ms; }
void msm_set_schedule(MsmWorkspace* ws, hipEvent_t accum_gate, uint32_t waves_per_simd) {
  ws->accum_gate = accum_gate;
  ws->waves_per_simd = waves_per_simd;
}
hipEvent_t msm_accum_done_event(MsmWorkspace* ws) { return ws->ev1; }
hipEvent_t msm_sorted_event(MsmWorkspace* ws) { return ws->ev_sorted; }
float msm_accum_event_offset_ms(MsmWorkspace* ws, hipEvent_t base, int which) {
  float t = 0.f;
  hipEvent_t e = which == 0 ? ws->ev0 : which == 1 ? ws->ev1 : ws->trace_ev[which - 2];
  if (e) (void)hipEventElapsedTime(&t, base, e);
  return t;
}

int msm_launch(const MsmInstance& m, MsmWorkspace* ws, const Fr* d_scalars, hipStream_t st) {
  if (m.curve == 2) return msm_launch_g2(m, ws, d_scalars, st);
  return msm_launch_t<Fq29Ops>(m, ws, d_scalars, st);
}

int msm_collect(MsmWorkspace* ws, uint8_t* out_windows, hipStream_t st) {
  if (ws->launched_n == 0) {
    memset(out_windows, 0, ws->out_bytes);
    return G16_OK;
  }
  G16_HIP(hipStreamSynchronize(st));
  (void)hipEventElapsedTime(&ws->last_accum_ms, ws->ev0, ws->ev1);
  memcpy(out_windows, ws->h_pinned, ws->out_bytes);
  return G16_OK;
}

int msm_run(const MsmInstance& m, MsmWorkspace* ws, const Fr* d_scalars, uint8_t* out_windows,
            hipStream_t st) {
  int rc = msm_launch(m, ws, d_scalars, st);
  if (rc) return rc;
  return msm_collect(ws, out_windows, st);
}

}  // namespace g16
