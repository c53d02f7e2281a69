// Groth16 batch verifier on the device (the acceptance check of SURVEY 3.4).
//
// Replaces `snarkjs groth16 verify` ([EXT] snarkjs 0.4.12 groth16_verify.js: cpub = IC[0] + sum pub_i IC[i+1] by
// G1.timesFr / add, then curve.pairingEq(-A, B, cpub, gamma2, C, delta2, alpha1, beta2); pins
// /root/reference/yarn.lock:987-1001; in the reference repo only the README's "verify" step and the PLONK twin
// Makefile:30-33 point at it) for MANY proofs against ONE verification key: proof i is accepted iff
//     e(-A_i, B_i) e(vk_x_i, gamma2) e(C_i, delta2) e(alpha1, beta2) = 1,   vk_x_i = IC_0 + sum_j pub_ij IC_{j+1}.
// Every proof gets its own verdict (no random linear combination: a rejected proof is identified, and the verdicts
// are exactly the one-at-a-time verdicts).
//
// Device schedule (all on the canonical 8x32 Montgomery field, pairing.cuh / ec.cuh):
//   create   verify_ic_table_kernel : 4-bit window tables d * 16^w * IC_j (d = 1..15, w < 64), affine, resident
//            host                    : line coefficients of gamma2 and delta2 (the G2 sides that never change), the
//                                      Miller value of (alpha1, beta2)
//   batch 1  verify_vkx_kernel      : a 256-lane workgroup per proof: lanes stride over the public signals, add the
//                                      table entries of the non-zero digits, LDS tree of XYZZ sums, -> affine vk_x
//         2  verify_miller_kernel   : a lane per (proof, pairing): (-A, B) with the lines of B computed on the fly,
//                                      (vk_x, gamma2) and (C, delta2) from the resident coefficients
//         3  verify_final_kernel    : a lane per proof: product of the three Miller values and the key's fourth,
//                                      final exponentiation, == 1
// The pairing work is ~35 k Fq products per proof on lanes that each own a 384-byte Fq12 accumulator: the tower
// products are called (not inlined) and spill to scratch -- a latency-bound, embarrassingly parallel kernel whose
// throughput comes from the batch size (BASELINE config 3's 1 024 proofs fill 48 + 16 wavefronts).
#include <hip/hip_runtime.h>

#include <mutex>
#include <vector>

#include "internal.h"
#include "pairing.cuh"

namespace g16 {
namespace {

constexpr int kIcWin = 64;          // 4-bit windows of a 256-bit scalar
constexpr int kIcRow = 15;          // non-zero digits

// tbl[(j * 64 + w) * 15 + d - 1] = d * 16^w * IC_j  (affine; all-zero = infinity)
__global__ __launch_bounds__(64) void verify_ic_table_kernel(const G1Affine* __restrict__ ic, uint32_t n,
                                                             G1Affine* __restrict__ tbl) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * kIcWin) return;
  const uint32_t j = t / kIcWin, w = t % kIcWin;
  G1XYZZ b;
  xyzz_from_affine(b, ic[j]);
  for (uint32_t k = 0; k < 4 * w; k++) xyzz_dbl(b);
  G1Affine base;
  xyzz_to_affine(base, b);
  G1XYZZ acc;
  xyzz_set_inf(acc);
  for (int d = 0; d < kIcRow; d++) {
    G1Affine a;
    if (aff_is_inf(base)) {
      a.x = FqOps::zero();
      a.y = FqOps::zero();
    } else {
      xyzz_madd(acc, base);
      xyzz_to_affine(a, acc);
    }
    tbl[(size_t)t * kIcRow + d] = a;
  }
}

// vkx[i] = IC_0 + sum_j pub[i][j] * IC_{j+1}   (pub: standard-form 256-bit integers, any value)
__global__ __launch_bounds__(256) void verify_vkx_kernel(const G1Affine* __restrict__ ic, const G1Affine* __restrict__ tbl,
                                                          const uint32_t* __restrict__ pub, uint32_t n_public,
                                                          G1Affine* __restrict__ vkx) {
  __shared__ G1XYZZ sm[256];
  const uint32_t i = blockIdx.x, t = threadIdx.x;
  G1XYZZ acc;
  xyzz_set_inf(acc);
  for (uint32_t j = t; j < n_public; j += 256) {
    const uint32_t* k = pub + ((size_t)i * n_public + j) * 8;
    for (int w = 0; w < kIcWin; w++) {
      const uint32_t d = (k[w >> 3] >> (4 * (w & 7))) & 15u;
      if (!d) continue;
      const G1Affine q = tbl[((size_t)(j + 1) * kIcWin + w) * kIcRow + d - 1];
      if (!aff_is_inf(q)) xyzz_madd(acc, q);
    }
  }
  sm[t] = acc;
  __syncthreads();
  for (uint32_t s = 128; s >= 1; s >>= 1) {
    if (t < s) {
      G1XYZZ a = sm[t];
      const G1XYZZ b = sm[t + s];
      xyzz_add(a, b);
      sm[t] = a;
    }
    __syncthreads();
  }
  if (t == 0) {
    G1XYZZ a = sm[0];
    const G1Affine ic0 = ic[0];
    if (!aff_is_inf(ic0)) xyzz_madd(a, ic0);
    G1Affine r;
    xyzz_to_affine(r, a);
    vkx[i] = r;
  }
}

struct ProofM {           // a proof in Montgomery form
  G1Affine a;
  G2Affine b;
  G1Affine c;
};

// standard-form proof bytes -> Montgomery, validity flags: bit 0 = a coordinate is not below q or a point is off its
// curve (the proof is rejected), bit 1 = A or B is infinity (their pairing is 1), bit 2 = C is infinity
__global__ __launch_bounds__(64) void verify_load_kernel(const g16_proof* __restrict__ proofs, uint32_t count,
                                                         PairingConsts pc, ProofM* __restrict__ out,
                                                         uint32_t* __restrict__ flags) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  static const uint32_t kQ[8] = G16_FQ_P;
  const uint32_t* w = reinterpret_cast<const uint32_t*>(&proofs[i]);
  Fq v[8];
  bool bad = false;
  for (int k = 0; k < 8; k++) {
    Fq x;
    bool lt = false, decided = false;
    for (int l = 7; l >= 0; l--) {
      x.v[l] = w[k * 8 + l];
      if (!decided && x.v[l] != kQ[l]) { lt = x.v[l] < kQ[l]; decided = true; }
    }
    bad |= !lt;
    v[k] = fp_to_mont(x);
  }
  ProofM p;
  p.a.x = v[0]; p.a.y = v[1];
  p.b.x = Fq2{v[2], v[3]}; p.b.y = Fq2{v[4], v[5]};
  p.c.x = v[6]; p.c.y = v[7];
  uint32_t f = 0;
  const bool a_inf = aff_is_inf(p.a), b_inf = aff_is_inf(p.b), c_inf = aff_is_inf(p.c);
  if (!a_inf && !g1_on_curve(p.a)) bad = true;
  if (!b_inf && !g2_on_curve(p.b, pc)) bad = true;
  if (!c_inf && !g1_on_curve(p.c)) bad = true;
  if (bad) f |= 1u;
  if (a_inf || b_inf) f |= 2u;
  if (c_inf) f |= 4u;
  out[i] = p;
  flags[i] = f;
}

// blockIdx.y = pairing: 0 = (-A, B), 1 = (vk_x, gamma2), 2 = (C, delta2); ml[3 i + pairing]
__global__ __launch_bounds__(64) void verify_miller_kernel(const ProofM* __restrict__ proofs, const uint32_t* __restrict__ flags,
                                                           const G1Affine* __restrict__ vkx,
                                                           const EllCoeffs* __restrict__ co_gamma,
                                                           const EllCoeffs* __restrict__ co_delta, uint32_t count,
                                                           PairingConsts pc, Fq12* __restrict__ ml) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  const uint32_t which = blockIdx.y;
  const uint32_t f = flags[i];
  Fq12 r = f12_one();
  if (!(f & 1u)) {
    if (which == 0) {
      if (!(f & 2u)) {
        G1Affine p = proofs[i].a;
        p.y = fp_neg(p.y);
        const G2Affine q = proofs[i].b;
        r = miller_loop(p, q, pc);
      }
    } else if (which == 1) {
      const G1Affine p = vkx[i];
      if (!aff_is_inf(p)) r = miller_loop_pre(p, co_gamma);
    } else {
      if (!(f & 4u)) {
        const G1Affine p = proofs[i].c;
        r = miller_loop_pre(p, co_delta);
      }
    }
  }
  ml[(size_t)i * 3 + which] = r;
}

__global__ __launch_bounds__(64) void verify_final_kernel(const Fq12* __restrict__ ml, const uint32_t* __restrict__ flags,
                                                          uint32_t count, PairingConsts pc, Fq12 ml_alphabeta,
                                                          uint8_t* __restrict__ ok) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  if (flags[i] & 1u) { ok[i] = 0; return; }
  Fq12 f = f12_mul(ml[(size_t)i * 3], ml[(size_t)i * 3 + 1]);
  f = f12_mul(f, ml[(size_t)i * 3 + 2]);
  f = f12_mul(f, ml_alphabeta);
  ok[i] = f12_is_one(final_exponentiation(f, pc)) ? 1 : 0;
}

// layer-test operator: out[i] = final_exponentiation(miller_loop(P_i, Q_i)) as 12 standard-form Fq words per pair,
// in the order of the W^k coefficients (a, b), k = 0..5 (tests/test_gpu_verify.py compares with oracle/bn254.py)
__global__ __launch_bounds__(64) void verify_pairing_op_kernel(const uint32_t* __restrict__ in, uint32_t count,
                                                               PairingConsts pc, uint32_t* __restrict__ out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  Fq v[6];
  for (int k = 0; k < 6; k++) {
    Fq x;
    for (int l = 0; l < 8; l++) x.v[l] = in[(size_t)i * 48 + k * 8 + l];
    v[k] = fp_to_mont(x);
  }
  const G1Affine p{v[0], v[1]};
  const G2Affine q{Fq2{v[2], v[3]}, Fq2{v[4], v[5]}};
  const Fq12 e = final_exponentiation(miller_loop(p, q, pc), pc);
  const Fq2* w[6] = {&e.c0.c0, &e.c1.c0, &e.c0.c1, &e.c1.c1, &e.c0.c2, &e.c1.c2};
  for (int k = 0; k < 6; k++) {
    const Fq a = fp_from_mont(w[k]->a), b = fp_from_mont(w[k]->b);
    for (int l = 0; l < 8; l++) {
      out[(size_t)i * 96 + k * 16 + l] = a.v[l];
      out[(size_t)i * 96 + k * 16 + 8 + l] = b.v[l];
    }
  }
}

}  // namespace
}  // namespace g16

using namespace g16;

struct g16_verifier {
  int device = 0;
  uint32_t n_public = 0;
  PairingConsts pc;
  Fq12 ml_alphabeta;
  G1Affine* d_ic = nullptr;
  G1Affine* d_tbl = nullptr;
  EllCoeffs* d_co_gamma = nullptr;
  EllCoeffs* d_co_delta = nullptr;
  // per-batch scratch, grown on demand
  size_t cap = 0;
  g16_proof* d_raw = nullptr;
  ProofM* d_proofs = nullptr;
  uint32_t* d_flags = nullptr;
  uint32_t* d_pub = nullptr;
  G1Affine* d_vkx = nullptr;
  Fq12* d_ml = nullptr;
  uint8_t* d_ok = nullptr;
  hipStream_t st = nullptr;
  hipEvent_t ev[4] = {};
  float last_ms[3] = {0, 0, 0};   // vk_x, Miller loops, final exponentiation of the last batch
  std::mutex mu;
  ~g16_verifier() {
    (void)hipSetDevice(device);
    void* bufs[] = {d_ic, d_tbl, d_co_gamma, d_co_delta, d_raw, d_proofs, d_flags, d_pub, d_vkx, d_ml, d_ok};
    for (void* p : bufs) if (p) (void)hipFree(p);
    for (auto& e : ev) if (e) (void)hipEventDestroy(e);
    if (st) (void)hipStreamDestroy(st);
  }
};

static bool fq_canonical(const uint8_t* p) {
  static const uint32_t kQ[8] = G16_FQ_P;
  uint32_t w[8];
  memcpy(w, p, 32);
  for (int l = 7; l >= 0; l--)
    if (w[l] != kQ[l]) return w[l] < kQ[l];
  return false;
}

static int verifier_create_impl(const uint8_t* vkey, size_t vkey_len, uint32_t n_public, int montgomery, int device,
                                g16_verifier* V) {
  const size_t need = 64 + 3 * 128 + ((size_t)n_public + 1) * 64;
  if (vkey_len != need) {
    set_error("verification key: expected " + std::to_string(need) + " bytes (alpha1 | beta2 | gamma2 | delta2 | IC[0.." +
              std::to_string(n_public) + "]), got " + std::to_string(vkey_len));
    return G16_E_FORMAT;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { set_error("no HIP device: the verifier has no CPU path"); return G16_E_NOGPU; }
  if (device < 0 || device >= ndev) { set_error("bad device ordinal"); return G16_E_ARG; }
  V->device = device;
  V->n_public = n_public;
  // every coordinate as a Montgomery Fq
  const size_t nfq = vkey_len / 32;
  std::vector<Fq> co(nfq);
  for (size_t k = 0; k < nfq; k++) {
    if (!fq_canonical(vkey + 32 * k)) { set_error("verification key: coordinate not below the field modulus"); return G16_E_FORMAT; }
    Fq x;
    memcpy(x.v, vkey + 32 * k, 32);
    co[k] = montgomery ? x : fp_to_mont(x);
  }
  pairing_consts_init(V->pc);
  const G1Affine alpha{co[0], co[1]};
  const G2Affine beta{Fq2{co[2], co[3]}, Fq2{co[4], co[5]}};
  const G2Affine gamma{Fq2{co[6], co[7]}, Fq2{co[8], co[9]}};
  const G2Affine delta{Fq2{co[10], co[11]}, Fq2{co[12], co[13]}};
  if (aff_is_inf(alpha) || aff_is_inf(beta) || aff_is_inf(gamma) || aff_is_inf(delta) || !g1_on_curve(alpha) ||
      !g2_on_curve(beta, V->pc) || !g2_on_curve(gamma, V->pc) || !g2_on_curve(delta, V->pc)) {
    set_error("verification key: alpha1 / beta2 / gamma2 / delta2 is not a point of its curve");
    return G16_E_FORMAT;
  }
  std::vector<G1Affine> ic(n_public + 1);
  for (uint32_t j = 0; j <= n_public; j++) {
    ic[j] = G1Affine{co[14 + 2 * j], co[15 + 2 * j]};
    if (!aff_is_inf(ic[j]) && !g1_on_curve(ic[j])) { set_error("verification key: IC[" + std::to_string(j) + "] is not on the curve"); return G16_E_FORMAT; }
  }
  std::vector<EllCoeffs> cg(kEllSteps), cd(kEllSteps);
  g2_precompute(gamma, V->pc, cg.data());
  g2_precompute(delta, V->pc, cd.data());
  V->ml_alphabeta = miller_loop(alpha, beta, V->pc);
  G16_HIP(hipSetDevice(device));
  G16_HIP(hipStreamCreate(&V->st));
  for (auto& e : V->ev) G16_HIP(hipEventCreate(&e));
  const size_t tbl_n = (size_t)(n_public + 1) * kIcWin * kIcRow;
  G16_HIP(hipMalloc(&V->d_ic, ic.size() * sizeof(G1Affine)));
  G16_HIP(hipMalloc(&V->d_tbl, tbl_n * sizeof(G1Affine)));
  G16_HIP(hipMalloc(&V->d_co_gamma, kEllSteps * sizeof(EllCoeffs)));
  G16_HIP(hipMalloc(&V->d_co_delta, kEllSteps * sizeof(EllCoeffs)));
  G16_HIP(hipMemcpyAsync(V->d_ic, ic.data(), ic.size() * sizeof(G1Affine), hipMemcpyHostToDevice, V->st));
  G16_HIP(hipMemcpyAsync(V->d_co_gamma, cg.data(), kEllSteps * sizeof(EllCoeffs), hipMemcpyHostToDevice, V->st));
  G16_HIP(hipMemcpyAsync(V->d_co_delta, cd.data(), kEllSteps * sizeof(EllCoeffs), hipMemcpyHostToDevice, V->st));
  const uint32_t nt = (n_public + 1) * kIcWin;
  verify_ic_table_kernel<<<(nt + 63) / 64, 64, 0, V->st>>>(V->d_ic, n_public + 1, V->d_tbl);
  G16_HIP(hipGetLastError());
  G16_HIP(hipStreamSynchronize(V->st));
  return G16_OK;
}

extern "C" int g16_verifier_create(const uint8_t* vkey, size_t vkey_len, uint32_t n_public, int montgomery, int device,
                                   g16_verifier** out) {
  if (!vkey || !out) { set_error("NULL argument"); return G16_E_ARG; }
  if (n_public > (1u << 20)) { set_error("verification key: too many public signals"); return G16_E_ARG; }
  g16_verifier* V = new g16_verifier();
  const int rc = verifier_create_impl(vkey, vkey_len, n_public, montgomery, device, V);
  if (rc) { delete V; return rc; }
  *out = V;
  return G16_OK;
}

static int ensure_scratch(g16_verifier* V, size_t count) {
  if (count <= V->cap) return G16_OK;
  void** bufs[] = {(void**)&V->d_raw, (void**)&V->d_proofs, (void**)&V->d_flags, (void**)&V->d_pub, (void**)&V->d_vkx,
                   (void**)&V->d_ml, (void**)&V->d_ok};
  for (void** b : bufs) if (*b) { (void)hipFree(*b); *b = nullptr; }
  V->cap = 0;
  G16_HIP(hipMalloc(&V->d_raw, count * sizeof(g16_proof)));
  G16_HIP(hipMalloc(&V->d_proofs, count * sizeof(ProofM)));
  G16_HIP(hipMalloc(&V->d_flags, count * 4));
  G16_HIP(hipMalloc(&V->d_pub, count * (size_t)(V->n_public ? V->n_public : 1) * 32));
  G16_HIP(hipMalloc(&V->d_vkx, count * sizeof(G1Affine)));
  G16_HIP(hipMalloc(&V->d_ml, count * 3 * sizeof(Fq12)));
  G16_HIP(hipMalloc(&V->d_ok, count));
  V->cap = count;
  return G16_OK;
}

extern "C" int g16_verify_batch(g16_verifier* V, const g16_proof* proofs, const uint8_t* pubs, size_t count, uint8_t* ok) {
  if (!V || !ok || (count && (!proofs || (V->n_public && !pubs)))) { set_error("NULL argument"); return G16_E_ARG; }
  if (count == 0) return G16_OK;
  if (count > (1u << 24)) { set_error("verify: batch too large"); return G16_E_ARG; }
  std::lock_guard<std::mutex> lk(V->mu);
  G16_HIP(hipSetDevice(V->device));
  int rc = ensure_scratch(V, count);
  if (rc) return rc;
  const uint32_t n = (uint32_t)count;
  hipStream_t st = V->st;
  G16_HIP(hipMemcpyAsync(V->d_raw, proofs, count * sizeof(g16_proof), hipMemcpyHostToDevice, st));
  if (V->n_public) G16_HIP(hipMemcpyAsync(V->d_pub, pubs, count * (size_t)V->n_public * 32, hipMemcpyHostToDevice, st));
  G16_HIP(hipEventRecord(V->ev[0], st));
  verify_load_kernel<<<(n + 63) / 64, 64, 0, st>>>(V->d_raw, n, V->pc, V->d_proofs, V->d_flags);
  verify_vkx_kernel<<<n, 256, 0, st>>>(V->d_ic, V->d_tbl, V->d_pub, V->n_public, V->d_vkx);
  G16_HIP(hipEventRecord(V->ev[1], st));
  verify_miller_kernel<<<dim3((n + 63) / 64, 3), 64, 0, st>>>(V->d_proofs, V->d_flags, V->d_vkx, V->d_co_gamma,
                                                              V->d_co_delta, n, V->pc, V->d_ml);
  G16_HIP(hipEventRecord(V->ev[2], st));
  verify_final_kernel<<<(n + 63) / 64, 64, 0, st>>>(V->d_ml, V->d_flags, n, V->pc, V->ml_alphabeta, V->d_ok);
  G16_HIP(hipEventRecord(V->ev[3], st));
  G16_HIP(hipGetLastError());
  G16_HIP(hipMemcpyAsync(ok, V->d_ok, count, hipMemcpyDeviceToHost, st));
  G16_HIP(hipStreamSynchronize(st));
  for (int k = 0; k < 3; k++) (void)hipEventElapsedTime(&V->last_ms[k], V->ev[k], V->ev[k + 1]);
  return G16_OK;
}

extern "C" int g16_verifier_timings(const g16_verifier* V, float ms[3]) {
  if (!V || !ms) { set_error("NULL argument"); return G16_E_ARG; }
  for (int k = 0; k < 3; k++) ms[k] = V->last_ms[k];
  return G16_OK;
}

extern "C" void g16_verifier_destroy(g16_verifier* V) { delete V; }

// Layer-test operator (tests/test_gpu_verify.py): count pairs (P in G1, Q in G2) as 6 standard-form 32-byte words
// each -> the 12 standard-form words of the pairing value the verifier kernels compute.
extern "C" int g16_pairing_op(int device, const uint8_t* in, uint32_t count, uint8_t* out) {
  if (!in || !out) { set_error("NULL argument"); return G16_E_ARG; }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { set_error("no HIP device"); return G16_E_NOGPU; }
  if (device < 0 || device >= ndev) { set_error("bad device ordinal"); return G16_E_ARG; }
  if (count == 0) return G16_OK;
  G16_HIP(hipSetDevice(device));
  PairingConsts pc;
  pairing_consts_init(pc);
  uint32_t *d_in = nullptr, *d_out = nullptr;
  G16_HIP(hipMalloc(&d_in, (size_t)count * 192));
  if (hipMalloc(&d_out, (size_t)count * 384) != hipSuccess) { (void)hipFree(d_in); set_error("hipMalloc"); return G16_E_HIP; }
  int rc = G16_OK;
  if (hipMemcpy(d_in, in, (size_t)count * 192, hipMemcpyHostToDevice) != hipSuccess) rc = G16_E_HIP;
  if (!rc) {
    verify_pairing_op_kernel<<<(count + 63) / 64, 64>>>(d_in, count, pc, d_out);
    if (hipGetLastError() != hipSuccess || hipMemcpy(out, d_out, (size_t)count * 384, hipMemcpyDeviceToHost) != hipSuccess) rc = G16_E_HIP;
  }
  (void)hipFree(d_in);
  (void)hipFree(d_out);
  if (rc) set_error("pairing operator: HIP error");
  return rc;
}
