// Host side of libg16hip.so: snarkjs container parsing, device orchestration, proof tail, C ABI.
//
// Mirrors snarkjs 0.4.12 `groth16.prove` (groth16_prove.js; pin /root/reference/yarn.lock:987-1001;
// call stack SURVEY.md section 3.3): the same checks in the same order with the same Error texts,
// the same five multi-exponentiations and the same blinding equations.
// There is NO CPU fallback: without a HIP device every compute entry point fails with G16_E_NOGPU.
#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include <map>
#include <memory>
#include <mutex>

#include "../../include/g16_prover.h"
#include "internal.h"

namespace g16 {

// ------------------------------------------------------------------ errors
static thread_local std::string g_err;
void set_error(const std::string& msg) { g_err = msg; }
const char* get_error() { return g_err.c_str(); }

// ------------------------------------------------------------------ binfile (App. A.1)
struct Section { const uint8_t* p = nullptr; uint64_t size = 0; bool present = false; };
struct BinFile { std::map<uint32_t, Section> secs; uint32_t version = 0; };

static uint32_t rd32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }
static uint64_t rd64(const uint8_t* p) { uint64_t v; memcpy(&v, p, 8); return v; }

// @iden3/binfileutils readBinFile(fileName, type, maxVersion) [EXT]
static int read_binfile(const uint8_t* buf, size_t len, const char* magic, uint32_t max_version,
                        const char* name, BinFile& out) {
  if (!buf || len < 12 || memcmp(buf, magic, 4) != 0) {
    set_error(std::string(name) + ": Invalid File format");
    return G16_E_FORMAT;
  }
  out.version = rd32(buf + 4);
  if (out.version > max_version) { set_error("Version not supported"); return G16_E_FORMAT; }
  const uint32_t nsec = rd32(buf + 8);
  size_t pos = 12;
  for (uint32_t i = 0; i < nsec; i++) {
    if (pos + 12 > len) { set_error(std::string(name) + ": Invalid File format"); return G16_E_FORMAT; }
    const uint32_t id = rd32(buf + pos);
    const uint64_t sz = rd64(buf + pos + 4);
    pos += 12;
    if (sz > len - pos) { set_error(std::string(name) + ": Invalid File format"); return G16_E_FORMAT; }
    Section& s = out.secs[id];
    if (!s.present) { s.p = buf + pos; s.size = sz; s.present = true; }
    pos += sz;
  }
  return G16_OK;
}
static int need_section(const BinFile& f, uint32_t id, const char* name, Section& out) {
  auto it = f.secs.find(id);
  if (it == f.secs.end()) {
    set_error(std::string(name) + ": Missing section " + std::to_string(id));
    return G16_E_FORMAT;
  }
  out = it->second;
  return G16_OK;
}

static const uint32_t kQ[8] = G16_FQ_P;
static const uint32_t kR[8] = G16_FR_P;

// ------------------------------------------------------------------ host point helpers
static void g1_out(uint8_t out[64], const G1Affine& p) {  // Montgomery affine -> standard LE bytes
  Fq x = fp_from_mont(p.x), y = fp_from_mont(p.y);
  memcpy(out, x.v, 32);
  memcpy(out + 32, y.v, 32);
}
static void g2_out(uint8_t out[128], const G2Affine& p) {
  Fq v[4] = {fp_from_mont(p.x.a), fp_from_mont(p.x.b), fp_from_mont(p.y.a), fp_from_mont(p.y.b)};
  for (int i = 0; i < 4; i++) memcpy(out + 32 * i, v[i].v, 32);
}
static bool scalar_lt_r(const uint32_t s[8]) {
  for (int i = 7; i >= 0; i--) {
    if (s[i] < kR[i]) return true;
    if (s[i] > kR[i]) return false;
  }
  return false;
}
static int random_scalar(uint32_t out[8]) {  // snarkjs Fr.random() counterpart: uniform in [0, r)
  int fd = open("/dev/urandom", O_RDONLY);
  if (fd < 0) { set_error("cannot open /dev/urandom"); return G16_E_STATE; }
  for (;;) {
    if (read(fd, out, 32) != 32) { close(fd); set_error("short read from /dev/urandom"); return G16_E_STATE; }
    out[7] &= 0x3fffffffu;  // r < 2^254
    if (scalar_lt_r(out)) break;
  }
  close(fd);
  return G16_OK;
}

struct KeyPoints {  // the zkey header points the proof tail needs (affine Montgomery)
  G1Affine alpha1, beta1, delta1;
  G2Affine beta2, delta2;
};

struct Partial {  // XYZZ sums of one shard, Montgomery
  G1XYZZ A, B1, C, H;
  G2XYZZ B2;
};
static_assert(sizeof(Partial) == G16_PARTIAL_BYTES, "partial blob layout");

}  // namespace g16

using namespace g16;

// ====================================================================== the prover handle
struct g16_prover {
  int device = 0;
  int shard_rank = 0, shard_count = 1;
  hipStream_t st = nullptr;     // = ctx[0].st: create-time work and witness staging
  uint32_t nVars = 0, nPublic = 0, N = 0, nCoefs = 0;
  int L = 0;
  KeyPoints kp;
  QapCsr csr;
  NttTables ntt;
  // Resident bases, shared by every context.  grp[0] = the witness group: sections A, B1, C share one front end,
  // B2 (the G2 twin of B1's points) rides on B1's sorted buckets; grp[1] = H; grp[2] = B2 on its own (only for a
  // malformed key whose sections 6 and 7 disagree on the points at infinity, else empty).
  MsmGroup grp[3];
  bool b2_solo = false;
  bool shard_begun = false;   // between g16_shard_begin and g16_shard_end
  // Per-proof scratch.  Several contexts so that g16_prove_batch can have proof i+1 on the GPU while the
  // host collects and finishes proof i (BASELINE config 3); single proofs use ctx[0].
  struct ProofCtx {
    hipStream_t st = nullptr;                        // QAP -> NTT -> H-MSM (critical chain)
    hipStream_t wst = nullptr, wst2 = nullptr;       // witness group: front end + G1 lane; G2 lane
    MsmWorkspace* ws[3] = {nullptr, nullptr, nullptr};   // one per group: they run concurrently
    hipEvent_t ev[8] = {};
    hipEvent_t mev[3][2] = {};
    F29 *d_a = nullptr, *d_b = nullptr, *d_c = nullptr;   // QAP/NTT vectors, lazy 9x29 format
    F29* d_wm = nullptr;                                  // Montgomery image of the witness (qap_eval's +-1 records)
    Fr* d_p = nullptr;                                    // H-MSM scalars, standard form
    Fr* d_w = nullptr;                                    // batch mode: this context's witness copy
    uint32_t* d_flag = nullptr;                           // canonicity check of the witness this context proves
    uint32_t* h_flag = nullptr;                           // ... its pinned host copy
    g16_timings tm{};
  };
  static constexpr int kCtx = 3;   // at most; `nctx` are created (a fourth context has to share hardware queues: r03, 272 proofs/s against 299)
  ProofCtx ctx[kCtx];
  int nctx = 3;   // r02 sweep, 512-proof batches: 205 / 257 / 267 proofs/s with 1 / 2 / 3 contexts (each on its own hardware queues)
  std::vector<Fr*> slot_dev;
  std::vector<std::vector<uint8_t>> slot_pub;
  g16_timings tm{};    // of the last completed proof (refresh_timings)
  int tm_ctx = -1;     // context whose events hold newer timings than `tm`, or -1
  std::mutex mu;

  ~g16_prover() {
    (void)hipSetDevice(device);
    for (Fr* p : slot_dev) if (p) (void)hipFree(p);
    for (auto& c : ctx) {
      void* vs[] = {c.d_a, c.d_b, c.d_c, c.d_p, c.d_w, c.d_flag, c.d_wm};
      for (void* p : vs) if (p) (void)hipFree(p);
      if (c.h_flag) (void)hipHostFree(c.h_flag);
      for (auto& w : c.ws) msm_workspace_destroy(w);
      if (c.wst && c.wst != c.st) (void)hipStreamDestroy(c.wst);
      if (c.wst2 && c.wst2 != c.st) (void)hipStreamDestroy(c.wst2);
      for (auto& e : c.mev) { if (e[0]) (void)hipEventDestroy(e[0]); if (e[1]) (void)hipEventDestroy(e[1]); }
      for (auto& e : c.ev) if (e) (void)hipEventDestroy(e);
      if (c.st) (void)hipStreamDestroy(c.st);
    }
    for (int m = 0; m < 2; m++) {
      if (csr.row_ptr[m]) (void)hipFree(csr.row_ptr[m]);
      if (csr.col[m]) (void)hipFree(csr.col[m]);
      if (csr.mid[m]) (void)hipFree(csr.mid[m]);
      if (csr.vptr[m]) (void)hipFree(csr.vptr[m]);
      if (csr.val[m]) (void)hipFree(csr.val[m]);
    }
    if (csr.long_rows) (void)hipFree(csr.long_rows);
    ntt_tables_destroy(ntt);
    for (auto& m : grp) msm_group_destroy(m);
  }
};

static int check_device(int device) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
    set_error("no HIP device available: libg16hip has no CPU fallback");
    return G16_E_NOGPU;
  }
  if (device < 0 || device >= n) { set_error("device ordinal out of range"); return G16_E_ARG; }
  G16_HIP(hipSetDevice(device));
  return G16_OK;
}

static void shard_range(uint32_t total, int rank, int count, uint32_t& lo, uint32_t& hi) {
  lo = (uint32_t)((uint64_t)total * rank / count);
  hi = (uint32_t)((uint64_t)total * (rank + 1) / count);
}

static int build_csr(g16_prover* P, const Section& s4) {
  if (s4.size < 4) { set_error("zkey: Invalid File format"); return G16_E_FORMAT; }
  const uint32_t nc = rd32(s4.p);
  if ((uint64_t)nc * 44 + 4 > s4.size) { set_error("zkey: Invalid File format"); return G16_E_FORMAT; }
  P->nCoefs = nc;
  const uint32_t N = P->N;
  std::vector<uint32_t> rp[2];
  rp[0].assign((size_t)N + 1, 0);
  rp[1].assign((size_t)N + 1, 0);
  for (uint32_t i = 0; i < nc; i++) {
    const uint8_t* rec = s4.p + 4 + (size_t)i * 44;
    const uint32_t m = rd32(rec), c = rd32(rec + 4), s = rd32(rec + 8);
    if (m > 1 || c >= N || s >= P->nVars) { set_error("zkey: coefficient record out of range"); return G16_E_FORMAT; }
    rp[m][c + 1]++;
  }
  for (int m = 0; m < 2; m++)
    for (uint32_t c = 0; c < N; c++) rp[m][c + 1] += rp[m][c];
  std::vector<uint32_t> col[2];
  std::vector<Fr> val[2];
  std::vector<uint32_t> cur[2];
  for (int m = 0; m < 2; m++) {
    col[m].resize(rp[m][N]);
    val[m].resize(rp[m][N]);
    cur[m].assign(rp[m].begin(), rp[m].end() - 1);
  }
  for (uint32_t i = 0; i < nc; i++) {
    const uint8_t* rec = s4.p + 4 + (size_t)i * 44;
    const uint32_t m = rd32(rec), c = rd32(rec + 4), s = rd32(rec + 8);
    const uint32_t k = cur[m][c]++;
    col[m][k] = s;
    memcpy(val[m][k].v, rec + 12, 32);
  }
  P->csr.N = N;
  {
    std::vector<uint32_t> long_rows, wave_rows;
    for (uint32_t c = 0; c < N; c++) {
      const uint32_t la = rp[0][c + 1] - rp[0][c], lb = rp[1][c + 1] - rp[1][c], mx = la > lb ? la : lb;
      if (mx > kQapWaveRow) wave_rows.push_back(c);
      else if (mx > kQapLongRow) long_rows.push_back(c);
    }
    P->csr.n_mid = (uint32_t)long_rows.size();
    long_rows.insert(long_rows.end(), wave_rows.begin(), wave_rows.end());
    P->csr.n_long = (uint32_t)long_rows.size();
    if (!long_rows.empty()) {
      G16_HIP(hipMalloc(&P->csr.long_rows, long_rows.size() * 4));
      G16_HIP(hipMemcpy(P->csr.long_rows, long_rows.data(), long_rows.size() * 4, hipMemcpyHostToDevice));
    }
  }
  // a row's records: the +-1 coefficients first (the file stores coef * R^2 mod r: +1 is R^2, -1 is r - R^2), then the rest
  Fr one, mone;
  for (int i = 0; i < 8; i++) one.v[i] = FrParams::R2[i];
  fp_reduce_once(one);
  mone = fp_neg(one);
  for (int m = 0; m < 2; m++) {
    const size_t nnz = col[m].size();
    std::vector<uint32_t> col2(nnz), mid(N), vptr(N);
    std::vector<Fr> gen;
    gen.reserve(nnz / 2 + 1);
    for (uint32_t c = 0; c < N; c++) {
      uint32_t k2 = rp[m][c];
      for (uint32_t k = rp[m][c]; k < rp[m][c + 1]; k++) {
        if (fp_eq(val[m][k], one)) col2[k2++] = col[m][k];
        else if (fp_eq(val[m][k], mone)) col2[k2++] = col[m][k] | 0x80000000u;
      }
      mid[c] = k2;
      vptr[c] = (uint32_t)gen.size();
      for (uint32_t k = rp[m][c]; k < rp[m][c + 1]; k++)
        if (!fp_eq(val[m][k], one) && !fp_eq(val[m][k], mone)) {
          col2[k2++] = col[m][k];
          gen.push_back(val[m][k]);
        }
    }
    const size_t ngen = gen.size();
    P->csr.nnz[m] = nnz;
    P->csr.ngen[m] = ngen;
    G16_HIP(hipMalloc(&P->csr.row_ptr[m], ((size_t)N + 1) * 4));
    G16_HIP(hipMalloc(&P->csr.mid[m], ((size_t)N + 1) * 4));
    G16_HIP(hipMalloc(&P->csr.vptr[m], ((size_t)N + 1) * 4));
    G16_HIP(hipMalloc(&P->csr.col[m], (nnz + 1) * 4));
    G16_HIP(hipMalloc(&P->csr.val[m], (ngen + 1) * sizeof(F29)));
    G16_HIP(hipMemcpy(P->csr.row_ptr[m], rp[m].data(), ((size_t)N + 1) * 4, hipMemcpyHostToDevice));
    G16_HIP(hipMemcpy(P->csr.mid[m], mid.data(), (size_t)N * 4, hipMemcpyHostToDevice));
    G16_HIP(hipMemcpy(P->csr.vptr[m], vptr.data(), (size_t)N * 4, hipMemcpyHostToDevice));
    if (nnz) G16_HIP(hipMemcpy(P->csr.col[m], col2.data(), nnz * 4, hipMemcpyHostToDevice));
    if (ngen) {
      Fr* tmp = nullptr;   // file words -> lazy coefficient format, once
      G16_HIP(hipMalloc(&tmp, ngen * sizeof(Fr)));
      G16_HIP(hipMemcpy(tmp, gen.data(), ngen * sizeof(Fr), hipMemcpyHostToDevice));
      int rc = qap_convert_coefs(tmp, P->csr.val[m], ngen, P->st);
      if (!rc && hipStreamSynchronize(P->st) != hipSuccess) { set_error("coefficient conversion failed"); rc = G16_E_HIP; }
      (void)hipFree(tmp);
      if (rc) return rc;
    }
  }
  return G16_OK;
}

static int create_impl(const uint8_t* zkey, size_t len, const g16_opts* opts, g16_prover* P) {
  BinFile f;
  int rc = read_binfile(zkey, len, "zkey", 2, "zkey", f);
  if (rc) return rc;
  Section s1, s2, s4, sb[5];
  if ((rc = need_section(f, 1, "zkey", s1))) return rc;
  if (s1.size < 4 || rd32(s1.p) != 1) { set_error("zkey file is not groth16"); return G16_E_FORMAT; }
  if ((rc = need_section(f, 2, "zkey", s2))) return rc;
  // zkey_utils.readHeaderGroth16 [EXT]
  const size_t hdr = 4 + 32 + 4 + 32 + 12 + 64 + 64 + 128 + 128 + 64 + 128;
  if (s2.size < hdr || rd32(s2.p) != 32 || rd32(s2.p + 36) != 32) {
    set_error("zkey: Invalid File format");
    return G16_E_FORMAT;
  }
  if (memcmp(s2.p + 4, kQ, 32) != 0 || memcmp(s2.p + 40, kR, 32) != 0) {
    set_error("Curve not supported: zkey is not over bn128");
    return G16_E_FORMAT;
  }
  const uint8_t* h = s2.p + 72;
  P->nVars = rd32(h);
  P->nPublic = rd32(h + 4);
  P->N = rd32(h + 8);
  h += 12;
  if (P->N == 0 || (P->N & (P->N - 1)) || (uint64_t)P->nPublic + 1 > (uint64_t)P->nVars) {
    set_error("zkey: Invalid File format");
    return G16_E_FORMAT;
  }
  P->L = 0;
  while ((1u << P->L) < P->N) P->L++;
  memcpy(&P->kp.alpha1, h, 64); h += 64;
  memcpy(&P->kp.beta1, h, 64); h += 64;
  memcpy(&P->kp.beta2, h, 128); h += 128;
  h += 128;  // gamma2 (verifier only)
  memcpy(&P->kp.delta1, h, 64); h += 64;
  memcpy(&P->kp.delta2, h, 128);
  if ((rc = need_section(f, 4, "zkey", s4))) return rc;
  static const uint32_t ids[5] = {5, 6, 7, 8, 9};
  for (int i = 0; i < 5; i++)
    if ((rc = need_section(f, ids[i], "zkey", sb[i]))) return rc;
  const uint32_t nC = P->nVars - P->nPublic - 1;
  const uint64_t expect[5] = {(uint64_t)P->nVars * 64, (uint64_t)P->nVars * 64, (uint64_t)P->nVars * 128,
                              (uint64_t)nC * 64, (uint64_t)P->N * 64};
  for (int i = 0; i < 5; i++)
    if (sb[i].size != expect[i]) { set_error("zkey: Invalid File format"); return G16_E_FORMAT; }

  // ---- device side
  P->device = opts ? opts->device : 0;
  P->shard_count = (opts && opts->shard_count > 1) ? opts->shard_count : 1;
  P->shard_rank = opts ? opts->shard_rank : 0;
  if (P->shard_rank < 0 || P->shard_rank >= P->shard_count) { set_error("shard_rank out of range"); return G16_E_ARG; }
  if ((rc = check_device(P->device))) return rc;
  // main stream (QAP -> NTT -> H-MSM, the critical path) at high priority, witness MSM streams low
  int prio_lo = 0, prio_hi = 0;
  G16_HIP(hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));
  if (getenv("G16_FLAT_PRIO") && atoi(getenv("G16_FLAT_PRIO"))) prio_lo = prio_hi;   // sweeps: every stream alike
  // HIP multiplexes streams onto a few hardware queues (4 per priority level by default, least-used first): two
  // streams on one queue run FIFO, whatever their events say.  r02 found the witness G2 lane on the main stream's
  // queue (its 0.9 ms bucket reduce "took" 2.7 ms behind the H-MSM's kernels), so context 0 -- the single-proof path
  // -- creates its streams first and back to back: main, G2 lane and the G2 lane's dup-row stream (high priority:
  // three distinct queues), witness front end + G1 lane (low priority: its own pool).
  const bool serial_mode = getenv("G16_SERIAL_MSM") && atoi(getenv("G16_SERIAL_MSM"));
  for (auto& c : P->ctx) {
    G16_HIP(hipStreamCreateWithPriority(&c.st, hipStreamNonBlocking, prio_hi));
    if (serial_mode) {
      c.wst = c.wst2 = c.st;
    } else {
      G16_HIP(hipStreamCreateWithPriority(&c.wst2, hipStreamNonBlocking, prio_hi));
      G16_HIP(hipStreamCreateWithPriority(&c.wst, hipStreamNonBlocking, prio_lo));
    }
    for (auto& e : c.ev) G16_HIP(hipEventCreate(&e));
    break;   // context 1 (batch pipelining) after context 0's workspaces, below
  }
  P->st = P->ctx[0].st;
  if ((rc = build_csr(P, s4))) return rc;
  if ((rc = ntt_tables_create(P->ntt, P->L, P->st))) return rc;
  MsmConfig cfg;
  cfg.task_len = opts ? opts->task_len : 0;
  // tuning overrides: G16_WINDOW_BITS="witness,h", G16_TASK_LEN="witness,h" (0 = auto)
  int c_over[2] = {0, 0}, tl_over[2] = {0, 0};
  if (const char* e = getenv("G16_WINDOW_BITS")) sscanf(e, "%d,%d", &c_over[0], &c_over[1]);
  if (const char* e = getenv("G16_TASK_LEN")) {
    if (sscanf(e, "%d,%d", &tl_over[0], &tl_over[1]) == 1) tl_over[1] = tl_over[0];
  }
  const int tl_opt = cfg.task_len;
  {
    // witness group: A over w, B1 (+ its G2 twin B2) over w, C over w[p+1:]; each section's point range sharded
    uint32_t lo[3], hi[3];
    shard_range(P->nVars, P->shard_rank, P->shard_count, lo[0], hi[0]);
    lo[1] = lo[0]; hi[1] = hi[0];
    shard_range(nC, P->shard_rank, P->shard_count, lo[2], hi[2]);
    MsmSectionIn secs[3];
    secs[0].bases_host = sb[0].p + (size_t)lo[0] * 64; secs[0].n_total = hi[0] - lo[0]; secs[0].scalar_offset = lo[0];
    secs[1].bases_host = sb[1].p + (size_t)lo[1] * 64; secs[1].bases2_host = sb[2].p + (size_t)lo[1] * 128;
    secs[1].n_total = hi[1] - lo[1]; secs[1].scalar_offset = lo[1];
    secs[2].bases_host = sb[3].p + (size_t)lo[2] * 64; secs[2].n_total = hi[2] - lo[2];
    secs[2].scalar_offset = P->nPublic + 1 + lo[2];
    cfg.c = c_over[0] ? c_over[0] : (opts ? opts->window_bits : 0);
    cfg.task_len = tl_over[0] ? tl_over[0] : tl_opt;
    cfg.dense = false;    // witness scalars are mostly 0/1/small
    cfg.precomp = 1;
    rc = msm_group_create(P->grp[0], secs, 3, cfg);
    if (rc == G16_E_FORMAT) {   // sections 6 / 7 disagree on infinity: B2 gets its own front end
      msm_group_destroy(P->grp[0]);
      MsmSectionIn b2 = secs[1];
      b2.bases_host = nullptr;
      secs[1].bases2_host = nullptr;
      if ((rc = msm_group_create(P->grp[0], secs, 3, cfg))) return rc;
      if ((rc = msm_group_create(P->grp[2], &b2, 1, cfg))) return rc;
      P->b2_solo = true;
    } else if (rc) {
      return rc;
    }
    // H over the quotient evaluations: dense (uniform in Fr)
    MsmSectionIn hsec;
    uint32_t hlo, hhi;
    shard_range(P->N, P->shard_rank, P->shard_count, hlo, hhi);
    hsec.bases_host = sb[4].p + (size_t)hlo * 64; hsec.n_total = hhi - hlo; hsec.scalar_offset = hlo;
    cfg.c = c_over[1] ? c_over[1] : (opts ? opts->window_bits : 0);
    cfg.task_len = tl_over[1] ? tl_over[1] : tl_opt;
    cfg.dense = true;
    cfg.precomp = opts ? (int)((opts->flags >> 8) & 0xffu) : 0;
    if ((rc = msm_group_create(P->grp[1], &hsec, 1, cfg))) return rc;
  }
  // G16_SERIAL_MSM=1 (profiling aid): every MSM on the main stream, so kernel times are standalone
  const bool serial = getenv("G16_SERIAL_MSM") && atoi(getenv("G16_SERIAL_MSM"));
  if (const char* e = getenv("G16_BATCH_CTX")) P->nctx = atoi(e) < 1 ? 1 : (atoi(e) > g16_prover::kCtx ? g16_prover::kCtx : atoi(e));
  for (int ci = 0; ci < P->nctx; ci++) {
    auto& c = P->ctx[ci];
    if (ci > 0) {
      // later contexts (batch pipelining): their three busy streams come from the NORMAL-priority pool.  HIP has 4
      // hardware queues per priority level; context 0 holds three of the high ones, so a second high-priority trio
      // shares queues with it -- and with itself: r02 kernel trace of g16_prove_batch, context 1's G2 lane sat on its
      // own main stream's queue, FIFO in front of the H-MSM (two proofs in flight took 2 x the time of one).
      const bool pools = prio_lo > prio_hi + 1 && !getenv("G16_CTX_SAME_PRIO");
      const int prio_ctx = pools ? prio_hi + 1 : prio_hi;
      // (a third context -- G16_BATCH_CTX=3, sweeps -- takes what is left: the 4th high queue, the 4th normal one, two low ones)
      const int p_main = ci == 1 ? prio_ctx : prio_hi, p_g2 = prio_ctx, p_dup = (ci == 1 || !pools) ? prio_ctx : prio_lo;
      G16_HIP(hipStreamCreateWithPriority(&c.st, hipStreamNonBlocking, p_main));
      if (serial) {
        c.wst = c.wst2 = c.st;
      } else {
        G16_HIP(hipStreamCreateWithPriority(&c.wst2, hipStreamNonBlocking, p_g2));
        G16_HIP(hipStreamCreateWithPriority(&c.wst, hipStreamNonBlocking, prio_lo));
      }
      for (auto& e : c.ev) G16_HIP(hipEventCreate(&e));
      msm_set_aux_stream_priority(p_dup);
    }
    for (int i = 0; i < 3; i++) {
      if ((rc = msm_workspace_create(&c.ws[i], P->grp[i]))) return rc;   // (creates the G2 lane's dup-row stream)
      G16_HIP(hipEventCreate(&c.mev[i][0]));
      G16_HIP(hipEventCreate(&c.mev[i][1]));
    }
    const size_t vb = (size_t)P->N * sizeof(F29);
    G16_HIP(hipMalloc(&c.d_a, vb));
    G16_HIP(hipMalloc(&c.d_b, vb));
    G16_HIP(hipMalloc(&c.d_c, vb));
    G16_HIP(hipMalloc(&c.d_p, (size_t)P->N * sizeof(Fr)));
    G16_HIP(hipMalloc(&c.d_wm, ((size_t)P->nVars + 1) * sizeof(F29)));
    G16_HIP(hipMalloc(&c.d_flag, 64));
    G16_HIP(hipHostMalloc((void**)&c.h_flag, 64));
    *c.h_flag = 0xffffffffu;
  }
  msm_set_aux_stream_priority(kMsmPrioHighest);
  G16_HIP(hipStreamSynchronize(P->st));
  return G16_OK;
}

// wtns_utils.readHeader + the checks of groth16.prove [EXT]
static int parse_wtns(const g16_prover* P, const uint8_t* wtns, size_t len, const uint8_t** body) {
  BinFile f;
  int rc = read_binfile(wtns, len, "wtns", 2, "wtns", f);
  if (rc) return rc;
  Section s1, s2;
  if ((rc = need_section(f, 1, "wtns", s1))) return rc;
  if (s1.size < 8) { set_error("wtns: Invalid File format"); return G16_E_FORMAT; }
  const uint32_t n8 = rd32(s1.p);
  if (s1.size < 8 + (uint64_t)n8) { set_error("wtns: Invalid File format"); return G16_E_FORMAT; }
  if (n8 != 32 || memcmp(s1.p + 4, kR, 32) != 0) {
    set_error("Curve of the witness does not match the curve of the proving key");
    return G16_E_FORMAT;
  }
  const uint32_t nw = rd32(s1.p + 4 + n8);
  if (nw != P->nVars) {
    set_error("Invalid witness length. Circuit: " + std::to_string(P->nVars) + ", witness: " + std::to_string(nw));
    return G16_E_FORMAT;
  }
  if ((rc = need_section(f, 2, "wtns", s2))) return rc;
  if (s2.size != (uint64_t)nw * 32) { set_error("wtns: Invalid File format"); return G16_E_FORMAT; }
  // (canonicity of the words -- every one below r -- is checked on the device after the upload: qap_check_witness)
  *body = s2.p;
  return G16_OK;
}

// the device-side canonicity verdict of the witness context `c` last checked (valid once its stream has drained)
static int witness_ok(const g16_prover::ProofCtx& c) {
  if (*c.h_flag == 0xffffffffu) return G16_OK;
  set_error("wtns: signal " + std::to_string(*c.h_flag) + " is not reduced modulo the scalar field");
  return G16_E_FORMAT;
}

static int stage_impl(g16_prover* P, uint32_t slot, const uint8_t* wtns, size_t len, bool sync = true) {
  const uint8_t* body = nullptr;
  int rc = parse_wtns(P, wtns, len, &body);
  if (rc) return rc;
  if (slot >= 65536) { set_error("slot out of range"); return G16_E_ARG; }
  G16_HIP(hipSetDevice(P->device));
  if (P->slot_dev.size() <= slot) { P->slot_dev.resize(slot + 1, nullptr); P->slot_pub.resize(slot + 1); }
  if (!P->slot_dev[slot]) G16_HIP(hipMalloc(&P->slot_dev[slot], (size_t)P->nVars * sizeof(Fr)));
  auto& c0 = P->ctx[0];
  G16_HIP(hipEventRecord(c0.ev[0], P->st));
  G16_HIP(hipMemcpyAsync(P->slot_dev[slot], body, (size_t)P->nVars * 32, hipMemcpyHostToDevice, P->st));
  G16_HIP(hipEventRecord(c0.ev[1], P->st));
  if ((rc = qap_check_witness(P->slot_dev[slot], P->nVars, c0.d_flag, c0.h_flag, P->st))) return rc;
  P->slot_pub[slot].assign(body + 32, body + 32 + (size_t)P->nPublic * 32);
  if (!sync) return G16_OK;   // g16_prove: the proof pipeline follows on the same stream; witness_ok() after it
  G16_HIP(hipStreamSynchronize(P->st));
  (void)hipEventElapsedTime(&c0.tm.upload_ms, c0.ev[0], c0.ev[1]);
  P->tm.upload_ms = c0.tm.upload_ms;
  return witness_ok(c0);
}

// Proof assembly on the host: O(1) work, in two halves.  The blinding terms that
// depend only on (r, s) -- r*delta1, s*delta1, (rs)*delta1, s*delta2: four of the six scalar
// multiplications -- are computed while the GPU is still busy (prepare_blinding, between launch and
// collect); assemble_proof needs the MSM sums.
struct Blinding {
  uint32_t r[8], s[8];
  G1XYZZ r_delta1, s_delta1, neg_rs_delta1;
  G2XYZZ s_delta2;
};

static int prepare_blinding(const KeyPoints* P, const uint8_t* r_in, const uint8_t* s_in, Blinding& b) {
  int rc;
  if (r_in) memcpy(b.r, r_in, 32); else if ((rc = random_scalar(b.r))) return rc;
  if (s_in) memcpy(b.s, s_in, 32); else if ((rc = random_scalar(b.s))) return rc;
  if (!scalar_lt_r(b.r) || !scalar_lt_r(b.s)) { set_error("blinding scalar not reduced mod r"); return G16_E_ARG; }
  xyzz_mul_scalar(b.r_delta1, P->delta1, b.r);
  xyzz_mul_scalar(b.s_delta1, P->delta1, b.s);
  xyzz_mul_scalar(b.s_delta2, P->delta2, b.s);
  Fr rm, sm;
  memcpy(rm.v, b.r, 32);
  memcpy(sm.v, b.s, 32);
  const Fr rs = fp_from_mont(fp_neg(fp_mul(fp_to_mont(rm), fp_to_mont(sm))));
  xyzz_mul_scalar(b.neg_rs_delta1, P->delta1, rs.v);
  return G16_OK;
}

// Everything of the proof that does not need the C and H sums: pi_a, pi_b, and the s*pi_a + r*pib1
// part of pi_c.  Runs on the host while the H-MSM (the last thing to finish) is still on the GPU.
struct EarlyTail {
  G1Affine a_aff;
  G2Affine b_aff;
  G1XYZZ c_terms;   // s*pi_a + r*pib1 - (rs)*delta1
};
// ... in two halves: the G1 half (pi_a and the s pi_a + r pib1 terms: two 254-bit scalar multiplications, 0.15 ms) needs
// only the G1 lane's sums and runs while the witness G2 lane finishes; the G2 half (pi_b) after it.
static void assemble_early_g1(const KeyPoints* P, const Blinding& b, const G1XYZZ& sumA, const G1XYZZ& sumB1, EarlyTail& e) {
  // pi_a = alpha1 + sum w_i A_i + r delta1
  G1XYZZ pa = sumA;
  xyzz_madd(pa, P->alpha1);
  xyzz_add(pa, b.r_delta1);
  // pib1 = beta1 + sum w_i B1_i + s delta1
  G1XYZZ pb1 = sumB1;
  xyzz_madd(pb1, P->beta1);
  xyzz_add(pb1, b.s_delta1);
  G1Affine b1_aff;
  xyzz_to_affine(e.a_aff, pa);
  xyzz_to_affine(b1_aff, pb1);
  G1XYZZ tmp;
  xyzz_mul_scalar(e.c_terms, e.a_aff, b.s);
  xyzz_mul_scalar(tmp, b1_aff, b.r);
  xyzz_add(e.c_terms, tmp);
  xyzz_add(e.c_terms, b.neg_rs_delta1);
}
static void assemble_early_g2(const KeyPoints* P, const Blinding& b, const G2XYZZ& sumB2, EarlyTail& e) {
  // pi_b = beta2 + sum w_i B2_i + s delta2
  G2XYZZ pb = sumB2;
  xyzz_madd(pb, P->beta2);
  xyzz_add(pb, b.s_delta2);
  xyzz_to_affine(e.b_aff, pb);
}
static void assemble_early(const KeyPoints* P, const Blinding& b, const G1XYZZ& sumA, const G1XYZZ& sumB1,
                           const G2XYZZ& sumB2, EarlyTail& e) {
  assemble_early_g1(P, b, sumA, sumB1, e);
  assemble_early_g2(P, b, sumB2, e);
}
// pi_c = sum_{i>p} w_i C_i + sum P_i H_i + s pi_a + r pib1 - (r s) delta1
static void assemble_late(const EarlyTail& e, const G1XYZZ& sumC, const G1XYZZ& sumH, g16_proof* out) {
  G1XYZZ pc = sumC;
  xyzz_add(pc, sumH);
  xyzz_add(pc, e.c_terms);
  G1Affine c_aff;
  xyzz_to_affine(c_aff, pc);
  g1_out(out->a, e.a_aff);
  g2_out(out->b, e.b_aff);
  g1_out(out->c, c_aff);
}

static int assemble_proof(const KeyPoints* P, const Blinding& b, const Partial* parts, uint32_t count, g16_proof* out) {
  Partial t = parts[0];
  for (uint32_t k = 1; k < count; k++) {
    xyzz_add(t.A, parts[k].A);
    xyzz_add(t.B1, parts[k].B1);
    xyzz_add(t.C, parts[k].C);
    xyzz_add(t.H, parts[k].H);
    xyzz_add(t.B2, parts[k].B2);
  }
  EarlyTail e;
  assemble_early(P, b, t.A, t.B1, t.B2, e);
  assemble_late(e, t.C, t.H, out);
  return G16_OK;
}

static int finish_impl(const KeyPoints* P, const Partial* parts, uint32_t count, const uint8_t* r_in,
                       const uint8_t* s_in, g16_proof* out) {
  Blinding b;
  int rc = prepare_blinding(P, r_in, s_in, b);
  if (rc) return rc;
  return assemble_proof(P, b, parts, count, out);
}

using ProofCtx = g16_prover::ProofCtx;

// The device pipeline of one proof on context `c`, in pieces that only enqueue (no host synchronisation).  The
// witness group (A, B1, B2, C) does not depend on the H polynomial: it runs on its own streams and overlaps with
// QAP -> NTTs -> join -> H-MSM on the context's main stream.
static void trace_host(const char* what, const std::chrono::steady_clock::time_point& t0) {
  static const bool on = getenv("G16_TRACE_HOST") != nullptr;
  if (on)
    fprintf(stderr, "[g16 host] %s enqueued at %.3f ms\n", what,
            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
}
// ... and the host's second half of a proof (waits, folds, assembly) on the same clock: origin = the last launch_ctx
static thread_local std::chrono::steady_clock::time_point g_trace_origin;   // (per host thread: two handles may prove from two threads)
namespace g16 {
double trace_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - g_trace_origin).count(); }
}
static void trace_tail(const char* what) {
  static const bool on = getenv("G16_TRACE_HOST") != nullptr;
  if (on)
    fprintf(stderr, "[g16 tail] %s at %.3f ms\n", what,
            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - g_trace_origin).count());
}
// Scheduling knobs (sweeps; the defaults are the measured optimum):
//   G16_ACC_WAVES = "abc": wavefronts per SIMD of the persistent accumulate grids of the witness G1 lane, the
//                   witness G2 lane and the H-MSM (0 = the kernel's full occupancy)
//   G16_ACC_QUOTA = "abc": chunks of 64 tasks after which an accumulate wavefront retires (0 = persistent grid)
//   G16_GATE      = "ab":  what the witness G1 / G2 accumulate kernels wait for: 0 nothing, 1 the NTT chain,
//                   2 the H-MSM's sort
static uint32_t sched_digit(const char* name, int which, int len, uint32_t dflt) {
  const char* e = getenv(name);
  if (!e || (int)strlen(e) != len || e[which] < '0' || e[which] > '9') return dflt;
  return (uint32_t)(e[which] - '0');
}
// witness group (+ the solo B2 group of a malformed key): front end on its own stream, after c.ev[2]
static int launch_witness_front(g16_prover* P, ProofCtx& c, const Fr* d_w) {
  int rc;
  if (c.wst != c.st) G16_HIP(hipStreamWaitEvent(c.wst, c.ev[2], 0));
  G16_HIP(hipEventRecord(c.mev[0][0], c.wst));
  if ((rc = msm_launch_front(P->grp[0], c.ws[0], d_w, c.wst))) return rc;
  if (P->b2_solo) {
    if (c.wst2 != c.st) G16_HIP(hipStreamWaitEvent(c.wst2, c.ev[2], 0));
    G16_HIP(hipEventRecord(c.mev[2][0], c.wst2));
    if ((rc = msm_launch_front(P->grp[2], c.ws[2], d_w, c.wst2))) return rc;
  }
  return G16_OK;
}
// ... and its lanes (accumulate, combine, reduce); `h_launched`: the H-MSM of this proof is already enqueued, so
// its events may gate the witness accumulates
static int launch_witness_lanes(g16_prover* P, ProofCtx& c, bool h_launched) {
  int rc;
  hipEvent_t gates[2] = {nullptr, nullptr};
  for (int l = 0; l < 2; l++) {
    // default: no gate.  (r02 sweeps, 40-proof medians, once every busy stream had its own hardware queue: real
    // nzcp_live circuit 5.2 ms ungated against 6.1 with the G2 accumulate held back until the NTT chain is done and
    // 6.3 with both; the 1.7 M synthetic circuit 8.1 ms either way -- holding work back only moves the collision.)
    const uint32_t gsel = sched_digit("G16_GATE", l, 2, 0u);
    if (gsel == 1) gates[l] = c.ev[4];
    else if (gsel == 2 && P->grp[1].n) gates[l] = msm_event(c.ws[1], 0);   // (recorded by the H front end, enqueued before)
  }
  msm_set_waves(c.ws[0], sched_digit("G16_ACC_WAVES", 0, 3, 0), sched_digit("G16_ACC_WAVES", 1, 3, 0));
  msm_set_quota(c.ws[0], sched_digit("G16_ACC_QUOTA", 0, 3, 0), sched_digit("G16_ACC_QUOTA", 1, 3, 0));
  if ((rc = msm_launch_lanes(P->grp[0], c.ws[0], c.wst, c.wst2, gates[0], gates[1]))) return rc;
  G16_HIP(hipEventRecord(c.mev[0][1], c.wst));
  if (P->b2_solo) {
    msm_set_waves(c.ws[2], 0, sched_digit("G16_ACC_WAVES", 1, 3, 0));
    if ((rc = msm_launch_lanes(P->grp[2], c.ws[2], c.wst2, c.wst2, nullptr, gates[1]))) return rc;
    G16_HIP(hipEventRecord(c.mev[2][1], c.wst2));
  }
  return G16_OK;
}
// QAP evaluation, then the odd-coset evaluation (iNTT, coset shift, NTT) of the vectors in `mask` (bit 0 = A,
// 1 = B, 2 = C) on the main stream
static int launch_qap_ntt(g16_prover* P, ProofCtx& c, const Fr* d_w, uint32_t mask, bool fuse_join = false) {
  int rc;
  if ((rc = qap_eval(P->csr, d_w, P->nVars, c.d_wm, c.d_a, c.d_b, c.d_c, c.st))) return rc;
  G16_HIP(hipEventRecord(c.ev[3], c.st));
  static const bool mid = !(getenv("G16_NO_FUSED_MID") && atoi(getenv("G16_NO_FUSED_MID")));
  if (fuse_join) {   // all three vectors here: the last forward pass writes P directly
    F29* v3[3] = {c.d_a, c.d_b, c.d_c};
    if (mid) return ntt_coset_roundtrip(P->ntt, v3, 3, c.d_p, c.st);
    if ((rc = ntt_dif_inverse_coset(P->ntt, v3, 3, c.st))) return rc;
    return ntt_dit_forward_join(P->ntt, c.d_a, c.d_b, c.d_c, c.d_p, c.st);
  }
  F29* all[3] = {c.d_a, c.d_b, c.d_c};
  F29* vecs[3];
  int nv = 0;
  for (int v = 0; v < 3; v++)
    if (mask & (1u << v)) vecs[nv++] = all[v];
  if (nv) {
    if (mid) return ntt_coset_roundtrip(P->ntt, vecs, nv, nullptr, c.st);
    if ((rc = ntt_dif_inverse_coset(P->ntt, vecs, nv, c.st))) return rc;   // iNTT + (1/N, w_2N^i) table
    if ((rc = ntt_dit_forward(P->ntt, vecs, nv, c.st))) return rc;
  }
  return G16_OK;
}
// P = A'.B' - C' over [lo, hi) of the domain, then the front end (sort) of the H-MSM of this handle's point range
static int launch_join_h_front(g16_prover* P, ProofCtx& c, uint32_t lo, uint32_t hi, bool joined = false) {
  int rc;
  if (!joined && hi > lo && (rc = ntt_join_abc(c.d_a + lo, c.d_b + lo, c.d_c + lo, c.d_p + lo, hi - lo, c.st))) return rc;
  G16_HIP(hipEventRecord(c.ev[4], c.st));
  G16_HIP(hipEventRecord(c.mev[1][0], c.st));
  return msm_launch_front(P->grp[1], c.ws[1], c.d_p, c.st);
}
// ... and its accumulate / combine / reduce.  G16_HGATE (sweeps): what the H accumulate waits for: 0 nothing, 1 the
// end of the witness G2 lane, 2 both witness lanes
static int launch_h_lanes(g16_prover* P, ProofCtx& c, bool w_launched) {
  int rc;
  msm_set_waves(c.ws[1], sched_digit("G16_ACC_WAVES", 2, 3, 0), 0);
  msm_set_quota(c.ws[1], sched_digit("G16_ACC_QUOTA", 2, 3, 0), 0);
  const uint32_t hg = w_launched ? sched_digit("G16_HGATE", 0, 1, 0) : 0;
  if (hg >= 2 && P->grp[0].n && msm_event(c.ws[0], 3)) G16_HIP(hipStreamWaitEvent(c.st, msm_event(c.ws[0], 3), 0));
  hipEvent_t gate = (hg >= 1 && P->grp[0].n) ? msm_event(c.ws[0], 4) : nullptr;
  if ((rc = msm_launch_lanes(P->grp[1], c.ws[1], c.st, c.st, gate, nullptr))) return rc;
  G16_HIP(hipEventRecord(c.mev[1][1], c.st));
  G16_HIP(hipEventRecord(c.ev[5], c.st));   // end of the main stream's share of this proof (refresh_timings)
  return G16_OK;
}
// `pipelined`: one of several proofs in flight (g16_prove_batch) -- the device is then the bottleneck, not the host's
// share of one proof nor the depth of its chains: the repeated-value stage and the H-MSM's bucket reduce run in their
// cheaper-on-the-device forms (msm_set_throughput: MsmGroup::dup_chunk_wide, MsmLaneWs::seg_len_thr)
static int launch_ctx(g16_prover* P, ProofCtx& c, const Fr* d_w, bool pipelined = false) {
  G16_HIP(hipSetDevice(P->device));
  int rc;
  msm_set_throughput(c.ws[0], P->grp[0], pipelined);
  msm_set_throughput(c.ws[1], P->grp[1], pipelined);
  if (P->b2_solo) msm_set_throughput(c.ws[2], P->grp[2], pipelined);
  const auto th0 = std::chrono::steady_clock::now();
  g_trace_origin = th0;
  if (P->tm_ctx == (int)(&c - P->ctx)) P->tm_ctx = -1;   // (its events are about to be recorded again)
  G16_HIP(hipEventRecord(c.ev[2], c.st));
  // critical chain first (host launch order matters: the witness group's ~35 launches cost host time)
  static const bool fuse = !(getenv("G16_NO_FUSED_JOIN") && atoi(getenv("G16_NO_FUSED_JOIN")));
  if ((rc = launch_qap_ntt(P, c, d_w, 7u, fuse))) return rc;
  trace_host("qap+ntt", th0);
  if ((rc = launch_witness_front(P, c, d_w))) return rc;
  trace_host("witness front end", th0);
  // a sharded handle joins only the slice of the domain its H bases cover
  uint32_t lo, hi;
  shard_range(P->N, P->shard_rank, P->shard_count, lo, hi);
  if ((rc = launch_join_h_front(P, c, lo, hi, fuse))) return rc;
  trace_host("h front end", th0);
  if ((rc = launch_witness_lanes(P, c, false))) return rc;
  trace_host("witness lanes", th0);
  if ((rc = launch_h_lanes(P, c, true))) return rc;
  trace_host("h lanes", th0);
  return G16_OK;
}

// Stream timings of the last collected proof, from the events its launch recorded (c.ev[5] closes the main stream in
// launch_h_lanes); valid until that context launches its next proof
static int refresh_timings(g16_prover* P) {
  if (P->tm_ctx < 0) return G16_OK;
  ProofCtx& c = P->ctx[P->tm_ctx];
  P->tm_ctx = -1;
  G16_HIP(hipSetDevice(P->device));
  G16_HIP(hipEventSynchronize(c.ev[5]));
  if (c.wst != c.st) G16_HIP(hipStreamSynchronize(c.wst));
  if (c.wst2 != c.st) G16_HIP(hipStreamSynchronize(c.wst2));
  (void)hipEventElapsedTime(&c.tm.qap_ms, c.ev[2], c.ev[3]);
  (void)hipEventElapsedTime(&c.tm.ntt_ms, c.ev[3], c.ev[4]);
  (void)hipEventElapsedTime(&c.tm.total_ms, c.ev[2], c.ev[5]);
  // msm_ms: [0] = witness group on its main stream (front end + G1 lane over A, B1, C), [2] = until the G2 lane
  // (B2) is done, [4] = H group; [1], [3] unused since the witness MSMs share one front end
  float wg = 0.f, hg = 0.f, g2 = 0.f;
  (void)hipEventElapsedTime(&wg, c.mev[0][0], c.mev[0][1]);
  (void)hipEventElapsedTime(&hg, c.mev[1][0], c.mev[1][1]);
  g2 = msm_event_offset_ms(c.ws[P->b2_solo ? 2 : 0], c.mev[P->b2_solo ? 2 : 0][0], 1, 9);
  c.tm.msm_ms[0] = wg; c.tm.msm_ms[1] = 0.f; c.tm.msm_ms[2] = g2; c.tm.msm_ms[3] = 0.f; c.tm.msm_ms[4] = hg;
  static const bool trace_dev = getenv("G16_TRACE_HOST") != nullptr;
  if (trace_dev) {
    float t_ntt0 = 0, t_ntt1 = 0;
    (void)hipEventElapsedTime(&t_ntt0, c.ev[2], c.ev[3]);
    (void)hipEventElapsedTime(&t_ntt1, c.ev[2], c.ev[4]);
    fprintf(stderr, "[g16 dev] qap 0..%.3f  ntt+join ..%.3f  total %.3f\n", t_ntt0, t_ntt1, c.tm.total_ms);
    static const char* nm[2] = {"W", "H"};
    for (int gi = 0; gi < 2; gi++) {
      float s0 = 0, s1 = 0;
      (void)hipEventElapsedTime(&s0, c.ev[2], c.mev[gi][0]);
      (void)hipEventElapsedTime(&s1, c.ev[2], c.mev[gi][1]);
      auto off = [&](int lane, int k) { return msm_event_offset_ms(c.ws[gi], c.ev[2], lane, k); };
      fprintf(stderr, "[g16 dev] %s  start %.3f pass0 %.3f binscan %.3f pass1 %.3f binsort+scans %.3f | G1 queue %.3f accumulate "
              "%.3f..%.3f combine %.3f reduce %.3f end %.3f (stream end %.3f)\n", nm[gi], s0, off(0, 0), off(0, 1), off(0, 2),
              off(0, 3), off(0, 6), off(0, 4), off(0, 5), off(0, 7), off(0, 8), off(0, 9), s1);
      if (gi == 0)
        fprintf(stderr, "[g16 dev] W2 (G2 lane) queue %.3f accumulate %.3f..%.3f combine %.3f reduce %.3f end %.3f\n",
                off(1, 6), off(1, 4), off(1, 5), off(1, 7), off(1, 8), off(1, 9));
    }
  }
  const float up = P->tm.upload_ms;
  P->tm = c.tm;
  P->tm.upload_ms = up;
  return G16_OK;
}

// Wait for context `c` and fold each MSM's row sums, in two halves: the witness group (A, B1, B2, C finish
// long before the H-MSM), then H.  The caller does the H-independent part of the proof assembly between the two.
static int collect_witness_msms(g16_prover* P, ProofCtx& c, Partial& out,
                                const std::function<void(const MsmResult&)>* after_g1 = nullptr) {
  MsmResult r;
  int rc = msm_collect(P->grp[0], c.ws[0], &r, after_g1);
  if (rc) return rc;
  out.A = r.g1[0];
  out.B1 = r.g1[1];
  out.C = r.g1[2];
  out.B2 = r.g2;
  c.tm.msm_accum_kernel_ms[0] = msm_last_accum_ms(c.ws[0], 0);
  c.tm.msm_accum_kernel_ms[1] = c.tm.msm_accum_kernel_ms[3] = 0.f;
  c.tm.msm_accum_kernel_ms[2] = msm_last_accum_ms(c.ws[0], 1);
  if (P->b2_solo) {
    if ((rc = msm_collect(P->grp[2], c.ws[2], &r))) return rc;
    out.B2 = r.g2;
    c.tm.msm_accum_kernel_ms[2] = msm_last_accum_ms(c.ws[2], 1);
  }
  return G16_OK;
}
static int collect_h_msm(g16_prover* P, ProofCtx& c, Partial& out) {
  MsmResult r;
  int rc = msm_collect(P->grp[1], c.ws[1], &r);
  if (rc) return rc;
  out.H = r.g1[0];
  trace_tail("H sum folded");
  c.tm.msm_accum_kernel_ms[4] = msm_last_accum_ms(c.ws[1], 0);
  // the stream timings of this proof are read from its events when somebody asks (g16_get_timings): the record +
  // synchronise + eight elapsed-time queries cost 0.05 ms of every proof when they sat here (r03 host trace).  The
  // streams need no draining either: each lane's ev_done, just waited for, is the last work on its stream.
  P->tm_ctx = (int)(&c - P->ctx);
  static const bool trace_dev = getenv("G16_TRACE_HOST") != nullptr;
  if (trace_dev) return refresh_timings(P);
  return G16_OK;
}
static int collect_ctx(g16_prover* P, ProofCtx& c, Partial& out) {
  int rc = collect_witness_msms(P, c, out);
  if (rc) return rc;
  return collect_h_msm(P, c, out);
}
// launch-independent second half of a proof on context `c`: collect, and assemble around the H wait
static int collect_and_assemble(g16_prover* P, ProofCtx& c, const Blinding& bl, g16_proof* out) {
  Partial part;
  EarlyTail e;
  bool g1_half = false;
  const std::function<void(const MsmResult&)> after_g1 = [&](const MsmResult& r) {   // (between the two lanes' waits)
    assemble_early_g1(&P->kp, bl, r.g1[0], r.g1[1], e);
    g1_half = true;
    trace_tail("G1 sums folded, pi_a and the pi_c terms assembled");
  };
  int rc = collect_witness_msms(P, c, part, &after_g1);
  trace_tail("witness sums folded");
  if (!rc) {   // host work while the H-MSM finishes
    if (!g1_half) assemble_early_g1(&P->kp, bl, part.A, part.B1, e);
    assemble_early_g2(&P->kp, bl, part.B2, e);
  }
  trace_tail("pi_a, pi_b assembled");
  const int rch = collect_h_msm(P, c, part);                           // always drain what was launched
  trace_tail("H sum folded, streams drained");
  if (rc) return rc;
  if (rch) return rch;
  assemble_late(e, part.C, part.H, out);
  trace_tail("pi_c assembled");
  return G16_OK;
}

// One proof on a staged witness (context 0).
static int device_impl(g16_prover* P, uint32_t slot, Partial& out) {
  if (slot >= P->slot_dev.size() || !P->slot_dev[slot]) { set_error("witness slot not staged"); return G16_E_STATE; }
  int rc = launch_ctx(P, P->ctx[0], P->slot_dev[slot]);
  if (rc) return rc;
  return collect_ctx(P, P->ctx[0], out);
}

// ====================================================================== in-process sharding (multi.cpp)
namespace g16 {

int shard_view(g16_prover* P, uint32_t slot, ShardView* out) {
  if (slot >= 65536) { set_error("slot out of range"); return G16_E_ARG; }
  G16_HIP(hipSetDevice(P->device));
  if (P->slot_dev.size() <= slot) { P->slot_dev.resize(slot + 1, nullptr); P->slot_pub.resize(slot + 1); }
  if (!P->slot_dev[slot]) G16_HIP(hipMalloc(&P->slot_dev[slot], (size_t)P->nVars * sizeof(Fr)));
  ProofCtx& c = P->ctx[0];
  out->device = P->device;
  out->st = c.st;
  out->d_w = P->slot_dev[slot];
  out->vec[0] = c.d_a; out->vec[1] = c.d_b; out->vec[2] = c.d_c;
  shard_range(P->N, P->shard_rank, P->shard_count, out->lo, out->hi);
  out->N = P->N;
  out->nVars = P->nVars;
  return G16_OK;
}

int shard_upload_witness(g16_prover* P, uint32_t slot, const uint8_t* wtns, size_t len) {
  return stage_impl(P, slot, wtns, len, /*sync=*/false);
}

int shard_witness_verdict(g16_prover* P) { return witness_ok(P->ctx[0]); }

int shard_begin_async(g16_prover* P, uint32_t slot, uint32_t mask) {
  if (slot >= P->slot_dev.size() || !P->slot_dev[slot]) { set_error("witness slot not staged"); return G16_E_STATE; }
  G16_HIP(hipSetDevice(P->device));
  ProofCtx& c = P->ctx[0];
  int rc;
  G16_HIP(hipEventRecord(c.ev[2], c.st));
  if (mask) {
    if ((rc = launch_qap_ntt(P, c, P->slot_dev[slot], mask))) return rc;
  } else {
    G16_HIP(hipEventRecord(c.ev[3], c.st));
  }
  if ((rc = launch_witness_front(P, c, P->slot_dev[slot]))) return rc;
  if ((rc = launch_witness_lanes(P, c, false))) return rc;
  P->shard_begun = true;
  return G16_OK;
}

int shard_end_collect(g16_prover* P, uint8_t partial[G16_PARTIAL_BYTES]) {
  if (!P->shard_begun) { set_error("shard_end without shard_begin"); return G16_E_STATE; }
  P->shard_begun = false;
  G16_HIP(hipSetDevice(P->device));
  ProofCtx& c = P->ctx[0];
  uint32_t lo, hi;
  shard_range(P->N, P->shard_rank, P->shard_count, lo, hi);
  int rc = launch_join_h_front(P, c, lo, hi);
  if (!rc) rc = launch_h_lanes(P, c, false);
  Partial part;
  if (rc) {
    (void)collect_witness_msms(P, c, part);
    return rc;
  }
  if ((rc = collect_ctx(P, c, part))) return rc;
  memcpy(partial, &part, sizeof(part));
  return G16_OK;
}

void shard_drain(g16_prover* P) {
  if (!P->shard_begun) return;
  P->shard_begun = false;
  (void)hipSetDevice(P->device);
  Partial part;
  (void)collect_witness_msms(P, P->ctx[0], part);
  (void)hipStreamSynchronize(P->ctx[0].st);
}

}  // namespace g16

// ====================================================================== C ABI
extern "C" {

const char* g16_last_error(void) { return get_error(); }

int g16_create(const uint8_t* zkey, size_t zkey_len, const g16_opts* opts, g16_prover** out) {
  if (!out) { set_error("out is NULL"); return G16_E_ARG; }
  *out = nullptr;
  std::unique_ptr<g16_prover> P(new g16_prover());
  int rc = create_impl(zkey, zkey_len, opts, P.get());
  if (rc) return rc;
  *out = P.release();
  return G16_OK;
}

void g16_destroy(g16_prover* p) { delete p; }

int g16_stage_witness(g16_prover* p, uint32_t slot, const uint8_t* wtns, size_t wtns_len) {
  if (!p) { set_error("prover is NULL"); return G16_E_ARG; }
  std::lock_guard<std::mutex> lk(p->mu);
  return stage_impl(p, slot, wtns, wtns_len);
}

int g16_prove_partial(g16_prover* p, uint32_t slot, uint8_t partial[G16_PARTIAL_BYTES]) {
  if (!p || !partial) { set_error("NULL argument"); return G16_E_ARG; }
  std::lock_guard<std::mutex> lk(p->mu);
  Partial part;
  int rc = device_impl(p, slot, part);
  if (rc) return rc;
  memcpy(partial, &part, sizeof(part));
  return G16_OK;
}

// Sharded H pipeline (BASELINE config 4): see include/g16_prover.h.
int g16_shard_begin(g16_prover* p, uint32_t slot, uint32_t vec_mask, void* const out_vecs[3]) {
  if (!p || (vec_mask & ~7u)) { set_error("g16_shard_begin: bad argument"); return G16_E_ARG; }
  std::lock_guard<std::mutex> lk(p->mu);
  if (slot >= p->slot_dev.size() || !p->slot_dev[slot]) { set_error("witness slot not staged"); return G16_E_STATE; }
  for (int v = 0; v < 3; v++)
    if ((vec_mask & (1u << v)) && (!out_vecs || !out_vecs[v])) { set_error("g16_shard_begin: missing output vector"); return G16_E_ARG; }
  G16_HIP(hipSetDevice(p->device));
  ProofCtx& c = p->ctx[0];
  int rc;
  G16_HIP(hipEventRecord(c.ev[2], c.st));
  if (vec_mask) {
    if ((rc = launch_qap_ntt(p, c, p->slot_dev[slot], vec_mask))) return rc;
  } else {
    G16_HIP(hipEventRecord(c.ev[3], c.st));
  }
  if ((rc = launch_witness_front(p, c, p->slot_dev[slot]))) return rc;   // keeps running behind the exchange
  if ((rc = launch_witness_lanes(p, c, false))) return rc;
  const F29* src[3] = {c.d_a, c.d_b, c.d_c};
  for (int v = 0; v < 3; v++)
    if (vec_mask & (1u << v))
      G16_HIP(hipMemcpyAsync(out_vecs[v], src[v], (size_t)p->N * sizeof(F29), hipMemcpyDefault, c.st));
  G16_HIP(hipStreamSynchronize(c.st));
  p->shard_begun = true;
  return G16_OK;
}

int g16_shard_end(g16_prover* p, uint32_t slot, const void* const slices[3], uint8_t partial[G16_PARTIAL_BYTES]) {
  if (!p || !slices || !partial) { set_error("NULL argument"); return G16_E_ARG; }
  std::lock_guard<std::mutex> lk(p->mu);
  if (!p->shard_begun) { set_error("g16_shard_end without g16_shard_begin"); return G16_E_STATE; }
  p->shard_begun = false;
  G16_HIP(hipSetDevice(p->device));
  ProofCtx& c = p->ctx[0];
  uint32_t lo, hi;
  shard_range(p->N, p->shard_rank, p->shard_count, lo, hi);
  F29* dst[3] = {c.d_a, c.d_b, c.d_c};
  int rc = G16_OK;
  for (int v = 0; v < 3 && !rc; v++) {
    if (hi > lo && !slices[v]) { set_error("g16_shard_end: missing slice"); rc = G16_E_ARG; break; }
    if (hi > lo && hipMemcpyAsync(dst[v] + lo, slices[v], (size_t)(hi - lo) * sizeof(F29), hipMemcpyDefault, c.st) != hipSuccess) {
      set_error("g16_shard_end: copy of a slice failed");
      rc = G16_E_HIP;
    }
  }
  if (!rc) rc = launch_join_h_front(p, c, lo, hi);
  if (!rc) rc = launch_h_lanes(p, c, false);
  Partial part;
  if (rc) {   // drain the witness group that g16_shard_begin started, then report
    (void)collect_witness_msms(p, c, part);
    return rc;
  }
  if ((rc = collect_ctx(p, c, part))) return rc;
  memcpy(partial, &part, sizeof(part));
  return G16_OK;
}

int g16_prove_finish(g16_prover* p, uint32_t slot, const uint8_t* partials, uint32_t count, const uint8_t r[32],
                     const uint8_t s[32], g16_proof* out, uint8_t* pub) {
  if (!p || !partials || !count || !out) { set_error("NULL argument"); return G16_E_ARG; }
  std::lock_guard<std::mutex> lk(p->mu);
  if (slot >= p->slot_pub.size()) { set_error("witness slot not staged"); return G16_E_STATE; }
  if (count != (uint32_t)p->shard_count) {   // a missing (or extra) partial would silently give an invalid proof
    set_error("g16_prove_finish: expected " + std::to_string(p->shard_count) + " partial sums, got " + std::to_string(count));
    return G16_E_ARG;
  }
  std::vector<Partial> parts(count);
  memcpy(parts.data(), partials, (size_t)count * sizeof(Partial));
  int rc = finish_impl(&p->kp, parts.data(), count, r, s, out);
  if (!rc && pub && p->nPublic) memcpy(pub, p->slot_pub[slot].data(), (size_t)p->nPublic * 32);
  return rc;
}

static int prove_staged_locked(g16_prover* p, uint32_t slot, const uint8_t r[32], const uint8_t s[32], g16_proof* out,
                               uint8_t* pub) {
  if (p->shard_count != 1) { set_error("sharded handle: use g16_prove_partial/g16_prove_finish"); return G16_E_STATE; }
  if (slot >= p->slot_dev.size() || !p->slot_dev[slot]) { set_error("witness slot not staged"); return G16_E_STATE; }
  Blinding bl;
  int rc = launch_ctx(p, p->ctx[0], p->slot_dev[slot]);
  const int rcb = prepare_blinding(&p->kp, r, s, bl);    // host work while the GPU runs
  if (rc || rcb) {                                       // drain whatever was launched, then report
    Partial part;
    if (!rc) (void)collect_ctx(p, p->ctx[0], part);
    return rc ? rc : rcb;
  }
  rc = collect_and_assemble(p, p->ctx[0], bl, out);
  if (!rc && pub && p->nPublic) memcpy(pub, p->slot_pub[slot].data(), (size_t)p->nPublic * 32);
  return rc;
}

int g16_prove_staged(g16_prover* p, uint32_t slot, const uint8_t r[32], const uint8_t s[32], g16_proof* out,
                     uint8_t* pub) {
  if (!p || !out) { set_error("NULL argument"); return G16_E_ARG; }
  std::lock_guard<std::mutex> lk(p->mu);
  return prove_staged_locked(p, slot, r, s, out, pub);
}

int g16_prove(g16_prover* p, const uint8_t* wtns, size_t wtns_len, const uint8_t r[32], const uint8_t s[32],
              g16_proof* out, uint8_t* pub) {
  if (!p || !out) { set_error("NULL argument"); return G16_E_ARG; }
  if (p->shard_count != 1) { set_error("sharded handle: use g16_prove_partial/g16_prove_finish"); return G16_E_STATE; }
  std::lock_guard<std::mutex> lk(p->mu);
  // upload and proof pipeline back to back on the main stream: no host synchronisation in between (the caller's
  // buffer stays valid until this call returns), the canonicity verdict of the words is read after the proof
  int rc = stage_impl(p, 0, wtns, wtns_len, /*sync=*/false);
  if (rc) return rc;
  rc = prove_staged_locked(p, 0, r, s, out, pub);
  (void)hipEventElapsedTime(&p->ctx[0].tm.upload_ms, p->ctx[0].ev[0], p->ctx[0].ev[1]);
  p->tm.upload_ms = p->ctx[0].tm.upload_ms;
  const int wrc = witness_ok(p->ctx[0]);
  return wrc ? wrc : rc;
}

int g16_prove_batch(g16_prover* p, const uint8_t* const* wtns, const size_t* wtns_lens, size_t count,
                    const uint8_t* rs, g16_proof* out, uint8_t* pub) {
  if (!p || !wtns || !wtns_lens || !out) { set_error("NULL argument"); return G16_E_ARG; }
  std::lock_guard<std::mutex> lk(p->mu);
  if (p->shard_count != 1) { set_error("sharded handle: use g16_prove_partial/g16_prove_finish"); return G16_E_STATE; }
  // Software pipeline over the contexts: while the host waits for, folds and finishes proof i-1,
  // proof i is already running on the GPU.
  const size_t wbytes = (size_t)p->nVars * sizeof(Fr);
  const size_t nctx = (size_t)p->nctx;
  Blinding bl[g16_prover::kCtx];
  int bl_rc[g16_prover::kCtx] = {};
  auto finish_one = [&](size_t i) -> int {
    ProofCtx& c = p->ctx[i % nctx];
    if (bl_rc[i % nctx]) {
      Partial part;
      (void)collect_ctx(p, c, part);
      return bl_rc[i % nctx];
    }
    const int frc = collect_and_assemble(p, c, bl[i % nctx], &out[i]);
    const int wrc = witness_ok(c);
    return wrc ? wrc : frc;
  };
  static const bool trace = getenv("G16_TRACE_HOST") != nullptr;
  const auto tb0 = std::chrono::steady_clock::now();
  auto now_ms = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tb0).count(); };
  for (size_t i = 0; i < count; i++) {
    ProofCtx& c = p->ctx[i % nctx];
    const double t_a = now_ms();
    if (i >= nctx) {   // the context is still busy with proof i - kCtx
      int rc = finish_one(i - nctx);
      if (rc) return rc;
    }
    const double t_b = now_ms();
    const uint8_t* body = nullptr;
    int rc = parse_wtns(p, wtns[i], wtns_lens[i], &body);
    if (rc) return rc;
    G16_HIP(hipSetDevice(p->device));
    if (!c.d_w) G16_HIP(hipMalloc(&c.d_w, wbytes));
    G16_HIP(hipMemcpyAsync(c.d_w, body, wbytes, hipMemcpyHostToDevice, c.st));
    const double t_c = now_ms();
    if ((rc = qap_check_witness(c.d_w, p->nVars, c.d_flag, c.h_flag, c.st))) return rc;
    if (pub && p->nPublic) memcpy(pub + i * (size_t)p->nPublic * 32, body + 32, (size_t)p->nPublic * 32);
    if ((rc = launch_ctx(p, c, c.d_w, count > 1))) return rc;
    const double t_d = now_ms();
    bl_rc[i % nctx] = prepare_blinding(&p->kp, rs ? rs + i * 64 : nullptr, rs ? rs + i * 64 + 32 : nullptr,
                                                    bl[i % nctx]);
    if (trace)
      fprintf(stderr, "[g16 batch] proof %zu: at %.3f ms  finish(i-%d) %.3f  upload call %.3f  launch %.3f  blinding %.3f\n", i,
              t_a, (int)nctx, t_b - t_a, t_c - t_b, t_d - t_c, now_ms() - t_d);
  }
  for (size_t i = count > nctx ? count - nctx : 0; i < count; i++) {
    int rc = finish_one(i);
    if (rc) return rc;
  }
  return G16_OK;
}

// Host-only proof assembly from gathered partial sums: needs no GPU handle (the assembling rank
// may hold only the zkey header).  Same arithmetic as g16_prove_finish.
int g16_finish_host(const uint8_t* zkey, size_t zkey_len, const uint8_t* partials, uint32_t count,
                    const uint8_t r[32], const uint8_t s[32], g16_proof* out) {
  if (!zkey || !partials || !count || !out) { set_error("NULL argument"); return G16_E_ARG; }
  BinFile f;
  int rc = read_binfile(zkey, zkey_len, "zkey", 2, "zkey", f);
  if (rc) return rc;
  Section s1, s2;
  if ((rc = need_section(f, 1, "zkey", s1))) return rc;
  if (s1.size < 4 || rd32(s1.p) != 1) { set_error("zkey file is not groth16"); return G16_E_FORMAT; }
  if ((rc = need_section(f, 2, "zkey", s2))) return rc;
  if (s2.size < 84 + 576) { set_error("zkey: Invalid File format"); return G16_E_FORMAT; }
  const uint8_t* h = s2.p + 84;
  KeyPoints kp;
  memcpy(&kp.alpha1, h, 64);
  memcpy(&kp.beta1, h + 64, 64);
  memcpy(&kp.beta2, h + 128, 128);
  memcpy(&kp.delta1, h + 384, 64);
  memcpy(&kp.delta2, h + 448, 128);
  std::vector<Partial> parts(count);
  memcpy(parts.data(), partials, (size_t)count * sizeof(Partial));
  return finish_impl(&kp, parts.data(), count, r, s, out);
}

// The point range [lo, hi) of a `total`-point base section owned by shard `rank` of `count`.
void g16_shard_range(uint32_t total, int32_t rank, int32_t count, uint32_t* lo, uint32_t* hi) {
  shard_range(total, rank, count < 1 ? 1 : count, *lo, *hi);
}

int g16_get_info(const g16_prover* p, g16_info* o) {
  if (!p || !o) { set_error("NULL argument"); return G16_E_ARG; }
  o->n_vars = p->nVars; o->n_public = p->nPublic; o->domain_size = p->N; o->n_coefs = p->nCoefs;
  const MsmGroup& w = p->grp[0];
  o->n_a = w.sec_n[0]; o->n_b1 = w.sec_n[1]; o->n_c = w.sec_n[2]; o->n_h = p->grp[1].n;
  o->n_b2 = p->b2_solo ? p->grp[2].n : w.sec_n[1];
  const uint32_t wc = (uint32_t)w.c, b2c = p->b2_solo ? (uint32_t)p->grp[2].c : wc;
  const uint32_t cs[5] = {wc, wc, b2c, wc, (uint32_t)p->grp[1].c};
  for (int i = 0; i < 5; i++) o->window_bits[i] = cs[i];
  return G16_OK;
}

int g16_get_timings(const g16_prover* p, g16_timings* o) {
  if (!p || !o) { set_error("NULL argument"); return G16_E_ARG; }
  g16_prover* P = const_cast<g16_prover*>(p);   // (reads events only; the handle's results are untouched)
  std::lock_guard<std::mutex> lk(P->mu);
  int rc = refresh_timings(P);
  if (rc) return rc;
  *o = p->tm;
  return G16_OK;
}

// Operator-level twin of snarkjs `buildABC1` (layer test): A_T, B_T, C_T of the staged witness as
// canonical Montgomery(2^256) residues, domainSize * 32 bytes each (what wasmcurves holds after buildABC1).
int g16_qap_eval(g16_prover* p, uint32_t slot, uint8_t* a, uint8_t* b, uint8_t* c) {
  if (!p || !a || !b || !c) { set_error("NULL argument"); return G16_E_ARG; }
  std::lock_guard<std::mutex> lk(p->mu);
  if (slot >= p->slot_dev.size() || !p->slot_dev[slot]) { set_error("witness slot not staged"); return G16_E_STATE; }
  G16_HIP(hipSetDevice(p->device));
  ProofCtx& cx = p->ctx[0];
  int rc = qap_eval(p->csr, p->slot_dev[slot], p->nVars, cx.d_wm, cx.d_a, cx.d_b, cx.d_c, cx.st);
  if (rc) return rc;
  const F29* src[3] = {cx.d_a, cx.d_b, cx.d_c};
  uint8_t* dst[3] = {a, b, c};
  for (int k = 0; k < 3; k++) {
    if ((rc = ntt_export(p->ntt, src[k], cx.d_p, false, false, cx.st))) return rc;
    G16_HIP(hipMemcpyAsync(dst[k], cx.d_p, (size_t)p->N * sizeof(Fr), hipMemcpyDeviceToHost, cx.st));
    G16_HIP(hipStreamSynchronize(cx.st));
  }
  return G16_OK;
}

void g16_free(void* p) { free(p); }

}  // extern "C"
