// PLONK batch verifier on the device -- the counterpart of plonk.hip's prover, so that the reference's PLONK flow
// (/root/reference/Makefile:30-33: setup, verification-key export) closes on the GPU: setup -> export -> prove -> verify.
//
// Replaces `snarkjs plonk verify` ([EXT] snarkjs 0.4.12 plonk_verify.js, pin /root/reference/yarn.lock:987-1001) for many
// proofs against one verification key, one verdict per proof.  Per proof, as snarkjs does it:
//   challenges beta, gamma, alpha, xi, v, u from the Keccak-256 transcript; the Lagrange evaluations L_1..L_n(xi) and
//   the public-input value; t(xi) from the quotient identity; then ONE pairing equation
//       e(W_xi + u W_xiw, [tau]_2) = e(xi W_xi + u xi w W_xiw + F - E, [1]_2),
//   F = the commitment of the batched polynomial (T1 + xi^n T2 + xi^2n T3 + v1 R + v2 A + v3 B + v4 C + v5 S1 + v6 S2 +
//   u Z with R linear in Z, Qm, Ql, Qr, Qo, Qc, S3), E = [the batched evaluation]_1.
// Restated in oracle/plonk.py::verify (derived from the KZG opening identity); every verdict must equal the oracle's.
//
// Split: the transcript and the O(nPublic) scalar arithmetic run on host threads (a few thousand Fr products per
// proof); the group arithmetic -- twenty 254-bit scalar multiplications per proof, two Miller loops against the
// precomputed lines of [tau]_2 and the G2 generator, one final exponentiation -- on the device:
//   pv_terms_kernel  a lane per (proof, term): double-and-add on the canonical field, complete formulas
//   pv_sum_kernel    a lane per proof: the two sums, to affine
//   pv_miller_kernel a lane per (proof, pairing); pv_final_kernel a lane per proof: product, final exponentiation, == 1
#include <hip/hip_runtime.h>

#include <mutex>
#include <thread>
#include <vector>

#include "internal.h"
#include "pairing.cuh"
#include "transcript.h"

namespace g16 {
namespace {

constexpr int kTerms = 20;   // 2 for the left-hand point, 18 for the right-hand one

__global__ __launch_bounds__(64) void pv_terms_kernel(const G1Affine* __restrict__ bases, const Fr* __restrict__ scalars,
                                                      uint32_t n, G1XYZZ* __restrict__ out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const G1Affine p = bases[i];
  const Fr k = scalars[i];
  G1XYZZ r;
  xyzz_set_inf(r);
  if (!aff_is_inf(p)) {
    int top = 255;
    while (top >= 0 && !((k.v[top >> 5] >> (top & 31)) & 1u)) top--;
    for (int b = top; b >= 0; b--) {
      xyzz_dbl(r);
      if ((k.v[b >> 5] >> (b & 31)) & 1u) xyzz_madd(r, p);
    }
  }
  out[i] = r;
}

// pts[2 i] = sum of terms [0, 2), pts[2 i + 1] = sum of terms [2, 20) of proof i, affine (all-zero = infinity)
__global__ __launch_bounds__(64) void pv_sum_kernel(const G1XYZZ* __restrict__ terms, uint32_t count, G1Affine* __restrict__ pts) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  const G1XYZZ* t = terms + (size_t)i * kTerms;
  G1XYZZ a = t[0];
  xyzz_add(a, t[1]);
  G1XYZZ b = t[2];
  for (int k = 3; k < kTerms; k++) xyzz_add(b, t[k]);
  G1Affine pa, pb;
  xyzz_to_affine(pa, a);
  xyzz_to_affine(pb, b);
  pts[2 * (size_t)i] = pa;
  pts[2 * (size_t)i + 1] = pb;
}

// blockIdx.y = 0: (left point, [tau]_2), 1: (right point, G2 generator)
__global__ __launch_bounds__(64) void pv_miller_kernel(const G1Affine* __restrict__ pts, const uint8_t* __restrict__ reject,
                                                       const EllCoeffs* __restrict__ co_x2, const EllCoeffs* __restrict__ co_g2,
                                                       uint32_t count, Fq12* __restrict__ ml) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  const uint32_t which = blockIdx.y;
  Fq12 r = f12_one();
  if (!reject[i]) {
    const G1Affine p = pts[2 * (size_t)i + which];
    if (!aff_is_inf(p)) r = miller_loop_pre(p, which ? co_g2 : co_x2);
  }
  ml[2 * (size_t)i + which] = r;
}

__global__ __launch_bounds__(64) void pv_final_kernel(const Fq12* __restrict__ ml, const uint8_t* __restrict__ reject, uint32_t count,
                                                      PairingConsts pc, uint8_t* __restrict__ ok) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  if (reject[i]) { ok[i] = 0; return; }
  const Fq12 f = f12_mul(ml[2 * (size_t)i], ml[2 * (size_t)i + 1]);
  ok[i] = f12_is_one(final_exponentiation(f, pc)) ? 1 : 0;
}

// ---- host: scalar helpers (the Keccak transcript lives in transcript.h, shared with the prover)
using transcript::hash_to_fr;
using transcript::put_fr_be;
bool lt_words(const uint32_t v[8], const uint32_t m[8]) {
  for (int l = 7; l >= 0; l--)
    if (v[l] != m[l]) return v[l] < m[l];
  return false;
}
void put_be(std::vector<uint8_t>& t, const uint8_t le[32]) {   // a 32-byte LE standard-form word, big-endian
  for (int k = 31; k >= 0; k--) t.push_back(le[k]);
}
Fr root_of_unity(uint32_t power) {
  Fr w = {G16_FR_W28};
  for (uint32_t i = 28; i > power; i--) w = fp_sqr(w);
  return w;
}
Fr fr_u64(uint64_t v) {
  Fr a = fp_zero<FrParams>();
  a.v[0] = (uint32_t)v;
  a.v[1] = (uint32_t)(v >> 32);
  return fp_to_mont(a);
}

}  // namespace
}  // namespace g16

using namespace g16;

struct g16_plonk_verifier {
  int device = 0;
  uint32_t power = 0, n_public = 0;
  Fr k1, k2, w1;                 // Montgomery
  G1Affine cm[8];                // Qm Ql Qr Qo Qc S1 S2 S3, Montgomery affine
  PairingConsts pc;
  EllCoeffs *d_co_x2 = nullptr, *d_co_g2 = nullptr;
  size_t cap = 0;
  G1Affine *d_bases = nullptr, *d_pts = nullptr;
  Fr* d_scal = nullptr;
  G1XYZZ* d_terms = nullptr;
  Fq12* d_ml = nullptr;
  uint8_t *d_rej = nullptr, *d_ok = nullptr;
  hipStream_t st = nullptr;
  std::mutex mu;
  ~g16_plonk_verifier() {
    (void)hipSetDevice(device);
    void* v[] = {d_co_x2, d_co_g2, d_bases, d_pts, d_scal, d_terms, d_ml, d_rej, d_ok};
    for (void* p : v) if (p) (void)hipFree(p);
    if (st) (void)hipStreamDestroy(st);
  }
};

static bool fq_word_ok(const uint8_t* p) {
  static const uint32_t kQ[8] = G16_FQ_P;
  uint32_t w[8];
  memcpy(w, p, 32);
  return lt_words(w, kQ);
}
static bool read_g1_std(const uint8_t* p, G1Affine* out) {   // standard-form x | y; all zero = infinity; false: bad encoding / off curve
  if (!fq_word_ok(p) || !fq_word_ok(p + 32)) return false;
  Fq x, y;
  memcpy(x.v, p, 32);
  memcpy(y.v, p + 32, 32);
  out->x = fp_to_mont(x);
  out->y = fp_to_mont(y);
  return aff_is_inf(*out) || g1_on_curve(*out);
}

extern "C" int g16_plonk_verifier_create(const uint8_t* vkey, size_t vkey_len, int device, g16_plonk_verifier** out) {
  if (!vkey || !out) { set_error("NULL argument"); return G16_E_ARG; }
  const size_t need = 8 + 64 + 8 * 64 + 128;
  if (vkey_len != need) { set_error("plonk verification key: expected " + std::to_string(need) + " bytes"); return G16_E_FORMAT; }
  g16_plonk_verifier* V = new g16_plonk_verifier();
  auto fail = [&](int rc) { delete V; return rc; };
  memcpy(&V->power, vkey, 4);
  memcpy(&V->n_public, vkey + 4, 4);
  if (V->power < 2 || V->power > 28) { set_error("plonk verification key: power out of range"); return fail(G16_E_FORMAT); }
  static const uint32_t kR[8] = G16_FR_P;
  for (int k = 0; k < 2; k++) {
    Fr x;
    memcpy(x.v, vkey + 8 + 32 * k, 32);
    if (!lt_words(x.v, kR)) { set_error("plonk verification key: k1 / k2 not below r"); return fail(G16_E_FORMAT); }
    (k ? V->k2 : V->k1) = fp_to_mont(x);
  }
  for (int k = 0; k < 8; k++)
    if (!read_g1_std(vkey + 72 + 64 * k, &V->cm[k])) { set_error("plonk verification key: a commitment is not a curve point"); return fail(G16_E_FORMAT); }
  pairing_consts_init(V->pc);
  G2Affine x2;
  {
    const uint8_t* p = vkey + 72 + 512;
    Fq c[4];
    for (int k = 0; k < 4; k++) {
      if (!fq_word_ok(p + 32 * k)) { set_error("plonk verification key: X_2 coordinate not below q"); return fail(G16_E_FORMAT); }
      Fq x;
      memcpy(x.v, p + 32 * k, 32);
      c[k] = fp_to_mont(x);
    }
    x2 = G2Affine{Fq2{c[0], c[1]}, Fq2{c[2], c[3]}};
    if (aff_is_inf(x2) || !g2_on_curve(x2, V->pc)) { set_error("plonk verification key: X_2 is not a point of the twist"); return fail(G16_E_FORMAT); }
  }
  V->w1 = root_of_unity(V->power);
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { set_error("no HIP device: the PLONK verifier has no CPU path"); return fail(G16_E_NOGPU); }
  if (device < 0 || device >= ndev) { set_error("bad device ordinal"); return fail(G16_E_ARG); }
  V->device = device;
  G2Affine g2;
  g2.x.a = Fq{G16_G2X0}; g2.x.b = Fq{G16_G2X1}; g2.y.a = Fq{G16_G2Y0}; g2.y.b = Fq{G16_G2Y1};
  std::vector<EllCoeffs> cx(kEllSteps), cg(kEllSteps);
  g2_precompute(x2, V->pc, cx.data());
  g2_precompute(g2, V->pc, cg.data());
  if (hipSetDevice(device) != hipSuccess || hipStreamCreate(&V->st) != hipSuccess ||
      hipMalloc(&V->d_co_x2, kEllSteps * sizeof(EllCoeffs)) != hipSuccess || hipMalloc(&V->d_co_g2, kEllSteps * sizeof(EllCoeffs)) != hipSuccess ||
      hipMemcpy(V->d_co_x2, cx.data(), kEllSteps * sizeof(EllCoeffs), hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(V->d_co_g2, cg.data(), kEllSteps * sizeof(EllCoeffs), hipMemcpyHostToDevice) != hipSuccess) {
    set_error("plonk verifier: HIP initialisation failed");
    return fail(G16_E_HIP);
  }
  *out = V;
  return G16_OK;
}

// the host half of one proof: false = rejected outright (bad encoding, point off the curve, xi in the domain)
static bool plonk_verify_prepare(const g16_plonk_verifier* V, const g16_plonk_proof* pr, const uint8_t* pub, G1Affine* bases, Fr* scal) {
  static const uint32_t kR[8] = G16_FR_P;
  G1Affine A, B, C, Z, T1, T2, T3, Wxi, Wxiw;
  G1Affine* pts[9] = {&A, &B, &C, &Z, &T1, &T2, &T3, &Wxi, &Wxiw};
  const uint8_t* src[9] = {pr->A, pr->B, pr->C, pr->Z, pr->T1, pr->T2, pr->T3, pr->Wxi, pr->Wxiw};
  for (int k = 0; k < 9; k++)
    if (!read_g1_std(src[k], pts[k])) return false;
  Fr ev[7];   // a b c s1 s2 zw r, Montgomery
  const uint8_t* es[7] = {pr->eval_a, pr->eval_b, pr->eval_c, pr->eval_s1, pr->eval_s2, pr->eval_zw, pr->eval_r};
  for (int k = 0; k < 7; k++) {
    Fr x;
    memcpy(x.v, es[k], 32);
    if (!lt_words(x.v, kR)) {   // snarkjs reduces (Fr.e): so does this
      int64_t br = 0;
      while (!lt_words(x.v, kR)) {
        br = 0;
        for (int i = 0; i < 8; i++) { br += (int64_t)x.v[i] - (int64_t)kR[i]; x.v[i] = (uint32_t)br; br >>= 32; }
      }
    }
    ev[k] = fp_to_mont(x);
  }
  std::vector<uint8_t> tr;
  put_be(tr, pr->A); put_be(tr, pr->A + 32); put_be(tr, pr->B); put_be(tr, pr->B + 32); put_be(tr, pr->C); put_be(tr, pr->C + 32);
  const Fr beta = hash_to_fr(tr);
  tr.clear();
  put_fr_be(tr, beta);
  const Fr gamma = hash_to_fr(tr);
  tr.clear();
  put_be(tr, pr->Z); put_be(tr, pr->Z + 32);
  const Fr alpha = hash_to_fr(tr);
  tr.clear();
  put_be(tr, pr->T1); put_be(tr, pr->T1 + 32); put_be(tr, pr->T2); put_be(tr, pr->T2 + 32); put_be(tr, pr->T3); put_be(tr, pr->T3 + 32);
  const Fr xi = hash_to_fr(tr);
  tr.clear();
  for (int k = 0; k < 7; k++) put_fr_be(tr, ev[k]);
  Fr v[7];
  v[0] = fp_zero<FrParams>();
  v[1] = hash_to_fr(tr);
  for (int i = 2; i <= 6; i++) v[i] = fp_mul(v[i - 1], v[1]);
  tr.clear();
  put_be(tr, pr->Wxi); put_be(tr, pr->Wxi + 32); put_be(tr, pr->Wxiw); put_be(tr, pr->Wxiw + 32);
  const Fr u = hash_to_fr(tr);
  const Fr one = fp_one<FrParams>();
  Fr xin = xi;
  for (uint32_t i = 0; i < V->power; i++) xin = fp_sqr(xin);
  const Fr zh = fp_sub(xin, one);
  if (fp_is_zero(zh)) return false;
  // L_j(xi) = w^j zh / (n (xi - w^j)), j < max(1, nPublic): one batched inversion
  const uint32_t nl = V->n_public > 0 ? V->n_public : 1;
  const Fr nn = fr_u64((uint64_t)1 << V->power);
  std::vector<Fr> den(nl), pre(nl), wj(nl);
  Fr w = one, run = one;
  for (uint32_t j = 0; j < nl; j++) {
    wj[j] = w;
    den[j] = fp_mul(nn, fp_sub(xi, w));
    if (fp_is_zero(den[j])) return false;
    pre[j] = run;
    run = fp_mul(run, den[j]);
    w = fp_mul(w, V->w1);
  }
  Fr inv = fp_inv(run);
  Fr pl = fp_zero<FrParams>(), l1 = fp_zero<FrParams>();
  for (uint32_t j = nl; j-- > 0;) {
    const Fr dj = fp_mul(inv, pre[j]);
    inv = fp_mul(inv, den[j]);
    const Fr lj = fp_mul(fp_mul(wj[j], zh), dj);
    if (j == 0) l1 = lj;
    if (j < V->n_public) {
      Fr x;
      memcpy(x.v, pub + (size_t)j * 32, 32);
      while (!lt_words(x.v, kR)) {
        int64_t br = 0;
        for (int i = 0; i < 8; i++) { br += (int64_t)x.v[i] - (int64_t)kR[i]; x.v[i] = (uint32_t)br; br >>= 32; }
      }
      pl = fp_sub(pl, fp_mul(fp_to_mont(x), lj));
    }
  }
  const Fr &a = ev[0], &b = ev[1], &c = ev[2], &s1 = ev[3], &s2 = ev[4], &zw = ev[5], &r = ev[6];
  const Fr alpha2 = fp_sqr(alpha);
  const Fr e3a = fp_add(fp_add(a, fp_mul(beta, s1)), gamma), e3b = fp_add(fp_add(b, fp_mul(beta, s2)), gamma);
  Fr t = fp_add(r, pl);
  t = fp_sub(t, fp_mul(fp_mul(fp_mul(fp_mul(e3a, e3b), fp_add(c, gamma)), zw), alpha));
  t = fp_sub(t, fp_mul(l1, alpha2));
  t = fp_mul(t, fp_inv(zh));
  const Fr bxi = fp_mul(beta, xi);
  Fr coefz = fp_mul(fp_mul(fp_add(fp_add(a, bxi), gamma), fp_add(fp_add(b, fp_mul(bxi, V->k1)), gamma)),
                    fp_add(fp_add(c, fp_mul(bxi, V->k2)), gamma));
  coefz = fp_add(fp_mul(coefz, alpha), fp_mul(l1, alpha2));
  const Fr coefs3 = fp_mul(fp_mul(fp_mul(fp_mul(e3a, e3b), beta), zw), alpha);
  Fr e = fp_add(t, fp_mul(v[1], r));
  e = fp_add(e, fp_add(fp_mul(v[2], a), fp_add(fp_mul(v[3], b), fp_add(fp_mul(v[4], c), fp_add(fp_mul(v[5], s1), fp_mul(v[6], s2))))));
  e = fp_add(e, fp_mul(u, zw));
  G1Affine gen;
  gen.x = fp_one<FqParams>();
  gen.y = fp_add(gen.x, gen.x);
  // left: W_xi + u W_xiw;  right: -(xi W_xi + u xi w W_xiw + F - E)
  const Fr neg1 = fp_neg(one);
  auto put = [&](int k, const G1Affine& p, const Fr& s_mont) {
    bases[k] = p;
    scal[k] = fp_from_mont(s_mont);
  };
  put(0, Wxi, one);
  put(1, Wxiw, u);
  put(2, Wxi, fp_neg(xi));
  put(3, Wxiw, fp_neg(fp_mul(fp_mul(u, xi), V->w1)));
  put(4, T1, neg1);
  put(5, T2, fp_neg(xin));
  put(6, T3, fp_neg(fp_sqr(xin)));
  put(7, V->cm[0], fp_neg(fp_mul(v[1], fp_mul(a, b))));
  put(8, V->cm[1], fp_neg(fp_mul(v[1], a)));
  put(9, V->cm[2], fp_neg(fp_mul(v[1], b)));
  put(10, V->cm[3], fp_neg(fp_mul(v[1], c)));
  put(11, V->cm[4], fp_neg(v[1]));
  put(12, V->cm[7], fp_mul(v[1], coefs3));
  put(13, Z, fp_neg(fp_add(fp_mul(v[1], coefz), u)));
  put(14, A, fp_neg(v[2]));
  put(15, B, fp_neg(v[3]));
  put(16, C, fp_neg(v[4]));
  put(17, V->cm[5], fp_neg(v[5]));
  put(18, V->cm[6], fp_neg(v[6]));
  put(19, gen, e);
  // a scalar above r/2 with a negated base costs the same additions but half the doublings on small negatives (-1, ...)
  static const uint32_t kHalf[8] = {0xf8000001u, 0xa1f0fac9u, 0x3cdcb848u, 0x9419f424u, 0x40c0ac2eu, 0xdc2822dbu, 0x7098d014u, 0x18322739u};
  for (int k = 0; k < kTerms; k++) {
    if (!lt_words(scal[k].v, kHalf)) {   // s >= (r + 1) / 2: s P = (r - s)(-P)
      int64_t br = 0;
      Fr m;
      for (int i = 0; i < 8; i++) { br += (int64_t)kR[i] - (int64_t)scal[k].v[i]; m.v[i] = (uint32_t)br; br >>= 32; }
      scal[k] = m;
      if (!aff_is_inf(bases[k])) bases[k].y = fp_neg(bases[k].y);
    }
  }
  return true;
}

extern "C" int g16_plonk_verify_batch(g16_plonk_verifier* V, const g16_plonk_proof* proofs, const uint8_t* pubs, size_t count,
                                      uint8_t* ok) {
  if (!V || !ok || (count && (!proofs || (V->n_public && !pubs)))) { set_error("NULL argument"); return G16_E_ARG; }
  if (count == 0) return G16_OK;
  if (count > (1u << 22)) { set_error("plonk verify: batch too large"); return G16_E_ARG; }
  std::lock_guard<std::mutex> lk(V->mu);
  std::vector<G1Affine> bases(count * kTerms);
  std::vector<Fr> scal(count * kTerms);
  std::vector<uint8_t> rej(count, 0);
  {
    unsigned nt = std::thread::hardware_concurrency();
    if (nt == 0) nt = 1;
    if (nt > 32) nt = 32;
    if (nt > count) nt = (unsigned)count;
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nt; t++)
      th.emplace_back([&, t] {
        for (size_t i = t; i < count; i += nt) {
          if (!plonk_verify_prepare(V, &proofs[i], pubs ? pubs + i * (size_t)V->n_public * 32 : nullptr, &bases[i * kTerms],
                                    &scal[i * kTerms])) {
            rej[i] = 1;
            for (int k = 0; k < kTerms; k++) {
              bases[i * kTerms + k] = G1Affine{fp_zero<FqParams>(), fp_zero<FqParams>()};
              scal[i * kTerms + k] = fp_zero<FrParams>();
            }
          }
        }
      });
    for (auto& x : th) x.join();
  }
  G16_HIP(hipSetDevice(V->device));
  if (count > V->cap) {
    void** bufs[] = {(void**)&V->d_bases, (void**)&V->d_pts, (void**)&V->d_scal, (void**)&V->d_terms, (void**)&V->d_ml, (void**)&V->d_rej,
                     (void**)&V->d_ok};
    for (void** b : bufs) if (*b) { (void)hipFree(*b); *b = nullptr; }
    V->cap = 0;
    G16_HIP(hipMalloc(&V->d_bases, count * kTerms * sizeof(G1Affine)));
    G16_HIP(hipMalloc(&V->d_scal, count * kTerms * sizeof(Fr)));
    G16_HIP(hipMalloc(&V->d_terms, count * kTerms * sizeof(G1XYZZ)));
    G16_HIP(hipMalloc(&V->d_pts, count * 2 * sizeof(G1Affine)));
    G16_HIP(hipMalloc(&V->d_ml, count * 2 * sizeof(Fq12)));
    G16_HIP(hipMalloc(&V->d_rej, count));
    G16_HIP(hipMalloc(&V->d_ok, count));
    V->cap = count;
  }
  hipStream_t st = V->st;
  const uint32_t n = (uint32_t)count, nt20 = n * kTerms;
  G16_HIP(hipMemcpyAsync(V->d_bases, bases.data(), bases.size() * sizeof(G1Affine), hipMemcpyHostToDevice, st));
  G16_HIP(hipMemcpyAsync(V->d_scal, scal.data(), scal.size() * sizeof(Fr), hipMemcpyHostToDevice, st));
  G16_HIP(hipMemcpyAsync(V->d_rej, rej.data(), count, hipMemcpyHostToDevice, st));
  pv_terms_kernel<<<(nt20 + 63) / 64, 64, 0, st>>>(V->d_bases, V->d_scal, nt20, V->d_terms);
  pv_sum_kernel<<<(n + 63) / 64, 64, 0, st>>>(V->d_terms, n, V->d_pts);
  pv_miller_kernel<<<dim3((n + 63) / 64, 2), 64, 0, st>>>(V->d_pts, V->d_rej, V->d_co_x2, V->d_co_g2, n, V->d_ml);
  pv_final_kernel<<<(n + 63) / 64, 64, 0, st>>>(V->d_ml, V->d_rej, n, V->pc, V->d_ok);
  G16_HIP(hipGetLastError());
  G16_HIP(hipMemcpyAsync(ok, V->d_ok, count, hipMemcpyDeviceToHost, st));
  G16_HIP(hipStreamSynchronize(st));   // (bases / scal / rej are host temporaries)
  return G16_OK;
}

extern "C" void g16_plonk_verifier_destroy(g16_plonk_verifier* V) { delete V; }
