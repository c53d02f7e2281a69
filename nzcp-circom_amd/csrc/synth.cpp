// Test-only trapdoor Groth16 setup over a shape-matched synthetic R1CS (host code, threaded).
//
// Why it exists: the reference ships no .zkey/.wtns
// (/root/reference/.gitignore:2-4,15-16) and its circuit cannot be compiled offline
// (/root/reference/Makefile:14-19 fetches sha256-var-circom with curl), so benchmark- and
// parity-sized proving keys are fabricated here with a known trapdoor, in snarkjs's zkey layout
// (H basis App. C.3).  Plays the role of `snarkjs groth16 setup` [EXT] for tests.
//
// The generator follows, draw for draw, the spec in oracle/synth.py's docstring; tests check the
// two produce byte-identical zkey/wtns for the same seed.  The arithmetic is the product's own
// fp.cuh / ec.cuh compiled for the host.
#include <stdlib.h>
#include <algorithm>
#include <array>
#include <string.h>

#include <atomic>
#include <new>
#include <thread>

#include "internal.h"

namespace g16 {
namespace {

struct Xo {
  uint64_t s[4];
  explicit Xo(uint64_t seed) {
    uint64_t z = seed;
    for (int i = 0; i < 4; i++) {
      z += 0x9E3779B97F4A7C15ull;
      uint64_t x = z;
      x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
      x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
      s[i] = x ^ (x >> 31);
    }
  }
  static uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
  uint64_t next() {
    const uint64_t res = rotl(s[1] * 5, 7) * 9;
    const uint64_t t = s[1] << 17;
    s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3];
    s[2] ^= t;
    s[3] = rotl(s[3], 45);
    return res;
  }
  uint64_t below(uint64_t k) { return next() % k; }
  Fr rand_fr_std() {  // (u0 | u1<<64 | u2<<128 | u3<<192) mod r, standard form
    Fr x;
    for (int i = 0; i < 4; i++) {
      const uint64_t u = next();
      x.v[2 * i] = (uint32_t)u;
      x.v[2 * i + 1] = (uint32_t)(u >> 32);
    }
    static const uint32_t R[8] = G16_FR_P;
    for (;;) {  // 2^256 / r < 6
      bool ge = true;
      for (int i = 7; i >= 0; i--) {
        if (x.v[i] > R[i]) break;
        if (x.v[i] < R[i]) { ge = false; break; }
      }
      if (!ge) break;
      int64_t br = 0;
      for (int i = 0; i < 8; i++) {
        br += (int64_t)x.v[i] - (int64_t)R[i];
        x.v[i] = (uint32_t)br;
        br >>= 32;
      }
    }
    return x;
  }
  Fr rand_fr() { return fp_to_mont(rand_fr_std()); }  // Montgomery
};

using FrM = Fr;  // Montgomery-form Fr throughout this file

// where the fixed-base multiplications of the *_setup entry points run: -1 = host threads, >= 0 = that HIP device
static std::atomic<int> g_setup_device{-1};

FrM fr_u64(uint64_t v) {
  Fr a = fp_zero<FrParams>();
  a.v[0] = (uint32_t)v;
  a.v[1] = (uint32_t)(v >> 32);
  return fp_to_mont(a);
}
FrM fr_one() { return fp_one<FrParams>(); }
FrM fr_neg_one() { return fp_neg(fp_one<FrParams>()); }

FrM coef(Xo& rng) {
  const uint64_t u = rng.below(100);
  if (u < 40) return fr_one();
  if (u < 80) return fr_neg_one();
  if (u < 95) return fr_u64(1 + rng.below(65535));
  return rng.rand_fr();
}

enum : uint8_t { CONST = 0, PUB, BIT, SMALL, SLACK };
constexpr uint64_t EXP_EXAMPLE = 1951416330ull;

struct Term { uint32_t s; FrM cf; };
struct Circuit {
  uint32_t n, p, m;
  std::vector<uint8_t> cls;
  std::vector<uint32_t> rowA, rowB, rowC;  // row offsets (m+1) into the term arrays
  std::vector<Term> tA, tB, tC;
  struct Slack { uint32_t row, sw, j1; FrM c1; };
  std::vector<Slack> slacks;  // in constraint order
};

void gen_classes(Circuit& c, uint64_t seed) {
  Xo rng(seed);
  c.cls.assign(c.n, CONST);
  for (uint32_t i = 1; i < c.n; i++) {
    if (i <= c.p) c.cls[i] = PUB;
    else {
      const uint64_t u = rng.below(100);
      c.cls[i] = u < 60 ? BIT : (u < 68 ? SMALL : SLACK);
    }
  }
}

void gen_circuit(Circuit& c, uint64_t seed) {
  gen_classes(c, seed);
  Xo rng(seed + 3);
  const uint32_t n = c.n, p = c.p, m = c.m;
  std::vector<uint32_t> slack, bits, rank(n, 0);
  for (uint32_t i = 0; i < n; i++) {
    if (c.cls[i] == SLACK) { rank[i] = (uint32_t)slack.size(); slack.push_back(i); }
    if (((c.cls[i] == PUB && i < p) || c.cls[i] == BIT) && i % 20 < 7) bits.push_back(i);
  }
  if (bits.empty()) bits.push_back(0);
  uint32_t nxt = 0;
  auto fix = [&](uint32_t s) { return (c.cls[s] == SLACK && rank[s] >= nxt) ? 0u : s; };
  auto pick = [&]() { return fix((uint32_t)rng.below(n)); };
  auto pick_b = [&]() {
    const uint32_t t = (uint32_t)rng.below(n);
    uint32_t s = t - t % 20 + (uint32_t)rng.below(7);
    if (s > n - 1) s = n - 1;
    return fix(s);
  };
  c.rowA.assign(1, 0); c.rowB.assign(1, 0); c.rowC.assign(1, 0);
  for (uint32_t r = 0; r < m; r++) {
    const uint64_t u = rng.below(100);
    const uint32_t rem = (uint32_t)slack.size() - nxt;
    if (rem > 0 && (u < 32 || rem >= m - r)) {
      const uint32_t sw = slack[nxt];
      const uint32_t ta = 1 + (uint32_t)rng.below(4);
      for (uint32_t k = 0; k < ta; k++) { const uint32_t s = pick(); c.tA.push_back({s, coef(rng)}); }
      const uint32_t tb = 1 + (uint32_t)rng.below(2);
      for (uint32_t k = 0; k < tb; k++) { const uint32_t s = pick_b(); c.tB.push_back({s, coef(rng)}); }
      const uint32_t j1 = pick();
      const FrM c1 = rng.rand_fr();
      c.tC.push_back({sw, fr_one()});
      c.tC.push_back({j1, c1});
      c.slacks.push_back({r, sw, j1, c1});
      nxt++;
    } else {
      const uint32_t x = bits[rng.below(bits.size())];
      const uint32_t y = bits[rng.below(bits.size())];
      const FrM ca = coef(rng), cb = coef(rng);
      c.tA.push_back({x, ca});
      c.tA.push_back({y, fp_neg(ca)});
      c.tB.push_back({x, cb});
      c.tB.push_back({y, cb});
      c.tB.push_back({0, fp_neg(cb)});
    }
    c.rowA.push_back((uint32_t)c.tA.size());
    c.rowB.push_back((uint32_t)c.tB.size());
    c.rowC.push_back((uint32_t)c.tC.size());
  }
}

// witness in Montgomery form
void gen_witness(const Circuit& c, uint64_t wseed, std::vector<FrM>& w) {
  Xo rng(wseed);
  w.assign(c.n, fp_zero<FrParams>());
  w[0] = fr_one();
  for (uint32_t i = 1; i < c.n; i++) {
    switch (c.cls[i]) {
      case PUB: w[i] = fr_u64(i == c.p ? EXP_EXAMPLE : rng.below(2)); break;
      case BIT: w[i] = fr_u64(rng.below(2)); break;
      case SMALL: w[i] = fr_u64(rng.below(1024)); break;
      default: w[i] = rng.rand_fr(); break;
    }
  }
  for (const auto& sl : c.slacks) {
    FrM a = fp_zero<FrParams>(), b = fp_zero<FrParams>();
    for (uint32_t k = c.rowA[sl.row]; k < c.rowA[sl.row + 1]; k++) a = fp_add(a, fp_mul(c.tA[k].cf, w[c.tA[k].s]));
    for (uint32_t k = c.rowB[sl.row]; k < c.rowB[sl.row + 1]; k++) b = fp_add(b, fp_mul(c.tB[k].cf, w[c.tB[k].s]));
    w[sl.sw] = fp_sub(fp_mul(a, b), fp_mul(sl.c1, w[sl.j1]));
  }
}

// ------------------------------------------------------------------ binfile writer
struct Buf {
  uint8_t* p = nullptr;
  size_t len = 0, cap = 0;
  bool reserve(size_t c) {
    p = (uint8_t*)malloc(c ? c : 1);
    cap = c;
    return p != nullptr;
  }
  void put(const void* src, size_t n) { memcpy(p + len, src, n); len += n; }
  void u32(uint32_t v) { put(&v, 4); }
  void u64(uint64_t v) { put(&v, 8); }
  uint8_t* skip(size_t n) { uint8_t* q = p + len; len += n; return q; }
};

void write_wtns(const std::vector<FrM>& w, Buf& b) {
  const size_t n = w.size();
  b.reserve(12 + 12 + 40 + 12 + n * 32);
  b.put("wtns", 4); b.u32(2); b.u32(2);
  b.u32(1); b.u64(40);
  static const uint32_t R[8] = G16_FR_P;
  b.u32(32); b.put(R, 32); b.u32((uint32_t)n);
  b.u32(2); b.u64((uint64_t)n * 32);
  for (size_t i = 0; i < n; i++) {
    const Fr s = fp_from_mont(w[i]);
    b.put(s.v, 32);
  }
}

// ------------------------------------------------------------------ fixed-base multiplication
template <class F> struct FixedBase {
  int wb = 8, nwin = 32;
  std::vector<Affine<F>> tbl;  // [nwin][2^wb - 1]
  size_t row() const { return ((size_t)1 << wb) - 1; }
};

template <class F> void batch_to_affine(const XYZZ<F>* in, Affine<F>* out, size_t n) {
  // one inversion per batch: x = X*(ZZ/ZZZ)^2, y = Y/ZZZ
  std::vector<typename F::T> pref(n);
  typename F::T acc = F::one();
  for (size_t i = 0; i < n; i++) {
    pref[i] = acc;
    if (!xyzz_is_inf(in[i])) acc = F::mul(acc, in[i].zzz);
  }
  typename F::T inv = F::inv(acc);
  for (size_t i = n; i-- > 0;) {
    if (xyzz_is_inf(in[i])) { out[i].x = F::zero(); out[i].y = F::zero(); continue; }
    const typename F::T zi = F::mul(inv, pref[i]);
    inv = F::mul(inv, in[i].zzz);
    const typename F::T zzi = F::sqr(F::mul(zi, in[i].zz));
    out[i].x = F::mul(in[i].x, zzi);
    out[i].y = F::mul(in[i].y, zi);
  }
}

template <class Fn> void parallel_for(size_t n, int threads, Fn fn) {
  if (threads < 1) threads = 1;
  if ((size_t)threads > n) threads = n ? (int)n : 1;
  std::vector<std::thread> th;
  const size_t chunk = (n + threads - 1) / threads;
  for (int t = 0; t < threads; t++) {
    const size_t lo = (size_t)t * chunk, hi = lo + chunk < n ? lo + chunk : n;
    if (lo >= hi) break;
    th.emplace_back([=]() { fn(lo, hi); });
  }
  for (auto& x : th) x.join();
}

template <class F> void build_table(FixedBase<F>& fb, const Affine<F>& gen, int wb, int threads) {
  fb.wb = wb;
  fb.nwin = (254 + wb - 1) / wb;
  const size_t row = fb.row();
  fb.tbl.resize((size_t)fb.nwin * row);
  std::vector<Affine<F>> bases(fb.nwin);
  XYZZ<F> b;
  xyzz_from_affine(b, gen);
  for (int j = 0; j < fb.nwin; j++) {
    xyzz_to_affine(bases[j], b);
    for (int k = 0; k < wb; k++) xyzz_dbl(b);
  }
  parallel_for((size_t)fb.nwin, threads, [&](size_t lo, size_t hi) {
    std::vector<XYZZ<F>> tmp(row);
    for (size_t j = lo; j < hi; j++) {
      XYZZ<F> acc;
      xyzz_set_inf(acc);
      for (size_t d = 0; d < row; d++) {
        xyzz_madd(acc, bases[j]);
        tmp[d] = acc;
      }
      batch_to_affine<F>(tmp.data(), &fb.tbl[j * row], row);
    }
  });
}

// out[i] = [k_i] G, k in Montgomery Fr; affine Montgomery bytes written at out + i*sizeof(Affine)
template <class F>
void fixed_mul_many(const FixedBase<F>& fb, const FrM* ks, size_t n, uint8_t* out, int threads) {
  const size_t row = fb.row();
  const uint32_t mask = (1u << fb.wb) - 1;
  parallel_for(n, threads, [&](size_t lo, size_t hi) {
    const size_t B = 512;
    std::vector<XYZZ<F>> acc(B);
    std::vector<Affine<F>> aff(B);
    for (size_t base = lo; base < hi; base += B) {
      const size_t cnt = base + B < hi ? B : hi - base;
      for (size_t i = 0; i < cnt; i++) {
        const Fr k = fp_from_mont(ks[base + i]);
        XYZZ<F>& a = acc[i];
        xyzz_set_inf(a);
        for (int j = 0; j < fb.nwin; j++) {
          const int pos = j * fb.wb;
          uint64_t v = k.v[pos >> 5];
          if ((pos >> 5) + 1 < 8) v |= (uint64_t)k.v[(pos >> 5) + 1] << 32;
          const uint32_t d = (uint32_t)(v >> (pos & 31)) & mask;
          if (d) xyzz_madd(a, fb.tbl[(size_t)j * row + d - 1]);
        }
      }
      batch_to_affine<F>(acc.data(), aff.data(), cnt);
      memcpy(out + base * sizeof(Affine<F>), aff.data(), cnt * sizeof(Affine<F>));
    }
  });
}

void batch_inverse(std::vector<FrM>& v) {
  const size_t n = v.size();
  std::vector<FrM> pref(n);
  FrM acc = fr_one();
  for (size_t i = 0; i < n; i++) { pref[i] = acc; acc = fp_mul(acc, v[i]); }
  FrM inv = fp_inv(acc);
  for (size_t i = n; i-- > 0;) {
    const FrM t = fp_mul(inv, pref[i]);
    inv = fp_mul(inv, v[i]);
    v[i] = t;
  }
}

FrM host_root(int L) {
  Fr w = {G16_FR_W28};
  for (int i = 28; i > L; i--) w = fp_sqr(w);
  return w;
}

// L_c(tau) over the size-2^L domain, c = first, first+step, ... (count values)
void lagrange_at(int L, const FrM& tau, size_t first, size_t step, size_t count, std::vector<FrM>& out) {
  const size_t N = (size_t)1 << L;
  const FrM w = host_root(L);
  const FrM zt = fp_sub(fp_pow_u64(tau, N), fr_one());
  const FrM scale = fp_mul(zt, fp_inv(fr_u64(N)));
  const FrM wstep = fp_pow_u64(w, step);
  std::vector<FrM> wc(count), den(count);
  FrM cur = fp_pow_u64(w, first);
  for (size_t i = 0; i < count; i++) {
    wc[i] = cur;
    den[i] = fp_sub(tau, cur);
    cur = fp_mul(cur, wstep);
  }
  batch_inverse(den);
  out.resize(count);
  for (size_t i = 0; i < count; i++) out[i] = fp_mul(fp_mul(scale, wc[i]), den[i]);
}

}  // namespace
}  // namespace g16

using namespace g16;

extern "C" int g16_synth_witness(uint32_t n, uint32_t p, uint32_t m, uint64_t seed, uint64_t wseed,
                                 uint8_t** wtns, size_t* wtns_len) {
  if (!wtns || !wtns_len || n < p + 1 || n < 2) { set_error("synth: bad arguments"); return G16_E_ARG; }
  Circuit c;
  c.n = n; c.p = p; c.m = m;
  gen_circuit(c, seed);
  std::vector<FrM> w;
  gen_witness(c, wseed, w);
  Buf b;
  write_wtns(w, b);
  *wtns = b.p;
  *wtns_len = b.len;
  return G16_OK;
}

// Trapdoor Groth16 setup of an arbitrary R1CS held in `c` (rows of (signal, Montgomery coefficient)
// terms): snarkjs zkey layout out, plus the verification-key points.  Trapdoor from stream seed+1.
static int setup_core(const Circuit& c, uint64_t seed, int threads, uint8_t** zkey, size_t* zkey_len,
                      uint8_t** vkey, size_t* vkey_len) {
  const uint32_t n = c.n, p = c.p, m = c.m;
  if (threads <= 0) threads = (int)std::thread::hardware_concurrency();
  if (threads <= 0) threads = 1;
  int L = 0;
  while (((uint64_t)1 << L) < (uint64_t)m + p + 1) L++;
  if (L > 27) { set_error("setup: circuit too large"); return G16_E_ARG; }
  const size_t N = (size_t)1 << L;

  // trapdoor (stream seed+1): tau, alpha, beta, gamma, delta, all non-zero
  Xo trng(seed + 1);
  FrM td[5];
  for (int k = 0; k < 5;) {
    const FrM v = trng.rand_fr();
    if (!fp_is_zero(v)) td[k++] = v;
  }
  const FrM tau = td[0], alpha = td[1], beta = td[2], gamma = td[3], delta = td[4];
  std::vector<FrM> Lg;
  lagrange_at(L, tau, 0, 1, N, Lg);
  std::vector<FrM> u(n, fp_zero<FrParams>()), v(n, fp_zero<FrParams>()), t(n, fp_zero<FrParams>());
  const size_t ncoef = c.tA.size() + c.tB.size() + (size_t)p + 1;
  // section 4 image: records in constraint order, A terms then B terms, then the binding rows
  std::vector<uint8_t> s4(4 + ncoef * 44);
  {
    uint32_t nc32 = (uint32_t)ncoef;
    memcpy(s4.data(), &nc32, 4);
    uint8_t* q = s4.data() + 4;
    auto rec = [&](uint32_t mm, uint32_t cc, uint32_t ss, const FrM& cf) {
      memcpy(q, &mm, 4); memcpy(q + 4, &cc, 4); memcpy(q + 8, &ss, 4);
      const Fr raw = fp_to_mont(cf);  // Montgomery(coef) * R = coef * R^2, stored as a plain integer
      memcpy(q + 12, raw.v, 32);
      q += 44;
    };
    for (uint32_t r = 0; r < m; r++) {
      for (uint32_t k = c.rowA[r]; k < c.rowA[r + 1]; k++) {
        rec(0, r, c.tA[k].s, c.tA[k].cf);
        u[c.tA[k].s] = fp_add(u[c.tA[k].s], fp_mul(c.tA[k].cf, Lg[r]));
      }
      for (uint32_t k = c.rowB[r]; k < c.rowB[r + 1]; k++) {
        rec(1, r, c.tB[k].s, c.tB[k].cf);
        v[c.tB[k].s] = fp_add(v[c.tB[k].s], fp_mul(c.tB[k].cf, Lg[r]));
      }
      for (uint32_t k = c.rowC[r]; k < c.rowC[r + 1]; k++)
        t[c.tC[k].s] = fp_add(t[c.tC[k].s], fp_mul(c.tC[k].cf, Lg[r]));
    }
    for (uint32_t i = 0; i <= p; i++) {
      rec(0, m + i, i, fr_one());
      u[i] = fp_add(u[i], Lg[m + i]);
    }
  }
  const FrM ginv = fp_inv(gamma), dinv = fp_inv(delta);
  std::vector<FrM> kic(p + 1), kc(n - p - 1), hs;
  for (uint32_t i = 0; i < n; i++) {
    const FrM kk = fp_add(fp_add(fp_mul(beta, u[i]), fp_mul(alpha, v[i])), t[i]);
    if (i <= p) kic[i] = fp_mul(kk, ginv);
    else kc[i - p - 1] = fp_mul(kk, dinv);
  }
  lagrange_at(L + 1, tau, 1, 2, N, hs);  // L^(2N)_{2i+1}(tau)
  for (auto& x : hs) x = fp_mul(x, dinv);

  // fixed-base multiplications: host threads, or the device selected by g16_setup_device (setup_gpu.hip; a
  // small-window table then -- the device has the lanes, the table should stay in its L2)
  const int dev = g_setup_device.load();
  const int wb = dev >= 0 ? 8 : (n >= 20000 ? 16 : 8);
  FixedBase<FqOps> fb1;
  FixedBase<Fq2Ops> fb2;
  G1Affine g1;
  g1.x = fp_one<FqParams>();
  g1.y = fp_add(g1.x, g1.x);
  G2Affine g2;
  g2.x.a = Fq{G16_G2X0}; g2.x.b = Fq{G16_G2X1}; g2.y.a = Fq{G16_G2Y0}; g2.y.b = Fq{G16_G2Y1};
  build_table(fb1, g1, wb, threads);
  build_table(fb2, g2, wb, threads);
  int mul_rc = G16_OK;
  auto mul1 = [&](const FrM* ks, size_t cnt, uint8_t* out) {
    if (dev < 0 || cnt < 64) { fixed_mul_many(fb1, ks, cnt, out, threads); return; }
    const int r = setup_fixed_mul_g1(dev, fb1.tbl.data(), fb1.wb, fb1.nwin, ks, cnt, out);
    if (r && !mul_rc) mul_rc = r;
  };
  auto mul2 = [&](const FrM* ks, size_t cnt, uint8_t* out) {
    if (dev < 0 || cnt < 64) { fixed_mul_many(fb2, ks, cnt, out, threads); return; }
    const int r = setup_fixed_mul_g2(dev, fb2.tbl.data(), fb2.wb, fb2.nwin, ks, cnt, out);
    if (r && !mul_rc) mul_rc = r;
  };

  const size_t nC = (size_t)n - p - 1;
  const size_t hdr2 = 4 + 32 + 4 + 32 + 12 + 64 + 64 + 128 + 128 + 64 + 128;
  const size_t sizes[11] = {0, 4, hdr2, (size_t)(p + 1) * 64, s4.size(), (size_t)n * 64, (size_t)n * 64,
                            (size_t)n * 128, nC * 64, N * 64, 64 + 4};
  size_t total = 12;
  for (int i = 1; i <= 10; i++) total += 12 + sizes[i];
  Buf z;
  if (!z.reserve(total)) { set_error("synth: out of memory"); return G16_E_STATE; }
  z.put("zkey", 4); z.u32(1); z.u32(10);
  auto sec = [&](uint32_t id) { z.u32(id); z.u64(sizes[id]); return z.skip(sizes[id]); };
  uint8_t* p1 = sec(1);
  { uint32_t one = 1; memcpy(p1, &one, 4); }
  uint8_t* p2 = sec(2);
  {
    static const uint32_t Qp[8] = G16_FQ_P, Rp[8] = G16_FR_P;
    uint32_t v32 = 32;
    uint8_t* q = p2;
    memcpy(q, &v32, 4); q += 4; memcpy(q, Qp, 32); q += 32;
    memcpy(q, &v32, 4); q += 4; memcpy(q, Rp, 32); q += 32;
    uint32_t dom = (uint32_t)N;
    memcpy(q, &n, 4); memcpy(q + 4, &p, 4); memcpy(q + 8, &dom, 4); q += 12;
    const FrM hk[3] = {alpha, beta, delta};
    uint8_t g1pts[3 * 64], g2pts[3 * 128];
    fixed_mul_many(fb1, hk, 3, g1pts, 1);
    const FrM hk2[3] = {beta, gamma, delta};
    fixed_mul_many(fb2, hk2, 3, g2pts, 1);
    memcpy(q, g1pts, 64); q += 64;            // alpha1
    memcpy(q, g1pts + 64, 64); q += 64;       // beta1
    memcpy(q, g2pts, 128); q += 128;          // beta2
    memcpy(q, g2pts + 128, 128); q += 128;    // gamma2
    memcpy(q, g1pts + 128, 64); q += 64;      // delta1
    memcpy(q, g2pts + 256, 128);              // delta2
  }
  uint8_t* p3 = sec(3);
  mul1(kic.data(), kic.size(), p3);
  uint8_t* p4 = sec(4);
  memcpy(p4, s4.data(), s4.size());
  uint8_t* p5 = sec(5);
  mul1(u.data(), n, p5);
  uint8_t* p6 = sec(6);
  mul1(v.data(), n, p6);
  uint8_t* p7 = sec(7);
  mul2(v.data(), n, p7);
  uint8_t* p8 = sec(8);
  mul1(kc.data(), kc.size(), p8);
  uint8_t* p9 = sec(9);
  mul1(hs.data(), hs.size(), p9);
  uint8_t* p10 = sec(10);
  memset(p10, 0, sizes[10]);
  if (mul_rc) { free(z.p); return mul_rc; }
  *zkey = z.p;
  *zkey_len = z.len;
  if (vkey && vkey_len) {
    // alpha1 | beta2 | gamma2 | delta2 | IC[0..p]   (affine Montgomery LE)
    Buf b;
    b.reserve(64 + 3 * 128 + (size_t)(p + 1) * 64);
    const uint8_t* h = p2 + 84;
    b.put(h, 64);              // alpha1
    b.put(h + 128, 128);       // beta2
    b.put(h + 256, 128);       // gamma2
    b.put(h + 448, 128);       // delta2
    b.put(p3, (size_t)(p + 1) * 64);
    *vkey = b.p;
    *vkey_len = b.len;
  }
  return G16_OK;
}

extern "C" int g16_synth_setup(uint32_t n, uint32_t p, uint32_t m, uint64_t seed, int threads, uint8_t** zkey,
                               size_t* zkey_len, uint8_t** wtns, size_t* wtns_len, uint8_t** vkey,
                               size_t* vkey_len) {
  if (!zkey || !zkey_len || n < p + 1 || n < 2 || m == 0) { set_error("synth: bad arguments"); return G16_E_ARG; }
  Circuit c;
  c.n = n; c.p = p; c.m = m;
  gen_circuit(c, seed);
  int rc = setup_core(c, seed, threads, zkey, zkey_len, vkey, vkey_len);
  if (rc) return rc;
  if (wtns && wtns_len) {
    std::vector<FrM> w;
    gen_witness(c, seed, w);
    Buf b;
    write_wtns(w, b);
    *wtns = b.p;
    *wtns_len = b.len;
  }
  return G16_OK;
}

// ------------------------------------------------------------------ SHA-256 chain circuit
// A REAL constraint system instead of the shape-matched random one: `blocks` chained SHA-256 compressions,
//   d_0 = the 32-byte private message,  d_{i+1} = SHA-256(d_i)   (one padded 64-byte block each),
// public outputs = the 256 bits of d_blocks, MSB-first per byte -- the bit order of the NZCP circuit's
// sha256 outputs (/root/reference/test/nzcp.js:41-47).  Bit-level R1CS in the style of circomlib's sha256
// gadgets that nzcptpl.circom includes (xor3 / ch / maj as one or two products per bit, modular additions as
// one linear row plus a booleanity row per result and carry bit): ~27 k constraints per block, 155 blocks
// fill a 2^22 domain.  Every wire is a bit, so the witness is bits only.
namespace g16 {
namespace {

struct Lin { std::vector<std::pair<uint32_t, int64_t>> t; };   // sum of coef * wire (wire 0 = the constant 1)
struct Bit { int32_t wire; int8_t a, b; };                     // value = a * w[wire] + b,  a in {0, 1, -1}
inline Bit bit_const(int v) { return Bit{0, 0, (int8_t)v}; }
inline Bit bit_wire(uint32_t w) { return Bit{(int32_t)w, 1, 0}; }
inline Bit bit_not(const Bit& x) { return Bit{x.wire, (int8_t)-x.a, (int8_t)(1 - x.b)}; }
inline bool bit_is_const(const Bit& x) { return x.a == 0; }

struct ShaBuilder {
  std::vector<uint64_t> w;       // witness: one bit per wire (the NZCP circuit's `exp` output is the one wider value)
  Circuit c;
  FrM pow2[40];                  // 2^k in Montgomery form, and small-coefficient cache
  ShaBuilder() {
    w.push_back(1);              // wire 0
    c.rowA.assign(1, 0); c.rowB.assign(1, 0); c.rowC.assign(1, 0);
  }
  int val(const Bit& x) const { return x.a * (int)w[x.wire] + x.b; }
  uint32_t new_wire(uint64_t v) { w.push_back(v); return (uint32_t)w.size() - 1; }
  static FrM coef_of(int64_t v) { return v >= 0 ? fr_u64((uint64_t)v) : fp_neg(fr_u64((uint64_t)(-v))); }
  static void add(Lin& l, const Bit& x, int64_t mul) {
    if (x.a) l.t.push_back({(uint32_t)x.wire, mul * x.a});
    if (x.b) l.t.push_back({0u, mul * x.b});
  }
  void push(std::vector<Term>& dst, std::vector<uint32_t>& rows, Lin& l) {
    // merge duplicate wires (the constant wire shows up several times), drop zeros
    std::sort(l.t.begin(), l.t.end());
    size_t i = 0;
    while (i < l.t.size()) {
      int64_t sum = 0;
      const uint32_t wire = l.t[i].first;
      while (i < l.t.size() && l.t[i].first == wire) sum += l.t[i++].second;
      if (sum) dst.push_back({wire, coef_of(sum)});
    }
    rows.push_back((uint32_t)dst.size());
  }
  void constrain(Lin a, Lin b, Lin cc) {   // <a,w> * <b,w> = <cc,w>
    push(c.tA, c.rowA, a); push(c.tB, c.rowB, b); push(c.tC, c.rowC, cc);
  }
  void boolean(uint32_t wire) {   // b * (b - 1) = 0
    Lin a, b, z;
    a.t.push_back({wire, 1});
    b.t.push_back({wire, 1}); b.t.push_back({0u, -1});
    constrain(a, b, z);
  }
  Bit xor2(const Bit& x, const Bit& y) {
    if (bit_is_const(x)) return x.b ? bit_not(y) : y;
    if (bit_is_const(y)) return y.b ? bit_not(x) : x;
    const uint32_t z = new_wire(val(x) ^ val(y));
    Lin a, b, cc;                 // (2x) * y = x + y - z
    add(a, x, 2); add(b, y, 1); add(cc, x, 1); add(cc, y, 1); cc.t.push_back({z, -1});
    constrain(a, b, cc);
    return bit_wire(z);
  }
  Bit xor3(const Bit& x, const Bit& y, const Bit& z) { return xor2(xor2(x, y), z); }
  Bit ch(const Bit& e, const Bit& f, const Bit& g) {   // e ? f : g  =  g + e (f - g)
    const uint32_t o = new_wire(val(e) ? val(f) : val(g));
    Lin a, b, cc;
    add(a, e, 1); add(b, f, 1); add(b, g, -1); cc.t.push_back({o, 1}); add(cc, g, -1);
    constrain(a, b, cc);
    return bit_wire(o);
  }
  Bit maj(const Bit& x, const Bit& y, const Bit& z) {  // mid = x y ; out = mid + z (x + y - 2 mid)
    const uint32_t mid = new_wire(val(x) & val(y));
    {
      Lin a, b, cc;
      add(a, x, 1); add(b, y, 1); cc.t.push_back({mid, 1});
      constrain(a, b, cc);
    }
    const int vx = val(x), vy = val(y), vz = val(z);
    const uint32_t o = new_wire((vx & vy) | (vx & vz) | (vy & vz));
    Lin a, b, cc;
    add(a, z, 1); add(b, x, 1); add(b, y, 1); b.t.push_back({mid, -2}); cc.t.push_back({o, 1}); cc.t.push_back({mid, -1});
    constrain(a, b, cc);
    return bit_wire(o);
  }
  using Word = std::array<Bit, 32>;   // bit i has weight 2^i
  static Word word_const(uint32_t v) {
    Word r;
    for (int i = 0; i < 32; i++) r[i] = bit_const((v >> i) & 1);
    return r;
  }
  static Word rotr(const Word& x, int k) { Word r; for (int i = 0; i < 32; i++) r[i] = x[(i + k) & 31]; return r; }
  static Word shr(const Word& x, int k) { Word r; for (int i = 0; i < 32; i++) r[i] = i + k < 32 ? x[i + k] : bit_const(0); return r; }
  Word xor3w(const Word& a, const Word& b, const Word& d) { Word r; for (int i = 0; i < 32; i++) r[i] = xor3(a[i], b[i], d[i]); return r; }
  uint32_t word_val(const Word& x) const { uint32_t v = 0; for (int i = 0; i < 32; i++) v |= (uint32_t)val(x[i]) << i; return v; }
  // sum of the operands mod 2^32: result bits (fresh wires, or `out_wires` when given) and carry bits are
  // constrained boolean; one linear row ties them to the operands
  Word add_mod32(const std::vector<Word>& ops, const uint32_t* out_wires = nullptr) {
    uint64_t sum = 0;
    for (const Word& o : ops) sum += word_val(o);
    int ncarry = 0;
    while (((uint64_t)ops.size() << 32) > ((uint64_t)1 << (32 + ncarry))) ncarry++;
    Lin a, b, z;
    for (const Word& o : ops)
      for (int i = 0; i < 32; i++) add(a, o[i], (int64_t)1 << i);
    Word r;
    for (int i = 0; i < 32 + ncarry; i++) {
      const int v = (int)((sum >> i) & 1);
      uint32_t wire;
      if (i < 32 && out_wires) { wire = out_wires[i]; w[wire] = (uint64_t)v; }
      else wire = new_wire(v);
      boolean(wire);
      a.t.push_back({wire, -((int64_t)1 << i)});
      if (i < 32) r[i] = bit_wire(wire);
    }
    b.t.push_back({0u, 1});
    constrain(a, b, z);
    return r;
  }
};

const uint32_t kShaK[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01,
    0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc,
    0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147,
    0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
    0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08,
    0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
    0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
const uint32_t kShaIV[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};

// One compression: out = st + rounds(st, W16).
void sha_compress(ShaBuilder& sb, const ShaBuilder::Word st[8], const ShaBuilder::Word W16[16], ShaBuilder::Word out[8],
                  uint32_t out_base) {   // out_base != 0: the result bits are the 256 wires from out_base on
  using Word = ShaBuilder::Word;
  Word W[64];
  for (int j = 0; j < 16; j++) W[j] = W16[j];
  for (int t = 16; t < 64; t++) {
    const Word s0 = sb.xor3w(ShaBuilder::rotr(W[t - 15], 7), ShaBuilder::rotr(W[t - 15], 18), ShaBuilder::shr(W[t - 15], 3));
    const Word s1 = sb.xor3w(ShaBuilder::rotr(W[t - 2], 17), ShaBuilder::rotr(W[t - 2], 19), ShaBuilder::shr(W[t - 2], 10));
    W[t] = sb.add_mod32({W[t - 16], s0, W[t - 7], s1});
  }
  Word a = st[0], b = st[1], c = st[2], d = st[3], e = st[4], f = st[5], g = st[6], h = st[7];
  for (int t = 0; t < 64; t++) {
    const Word S1 = sb.xor3w(ShaBuilder::rotr(e, 6), ShaBuilder::rotr(e, 11), ShaBuilder::rotr(e, 25));
    Word chw, mjw;
    for (int i = 0; i < 32; i++) chw[i] = bit_is_const(e[i]) ? (e[i].b ? f[i] : g[i]) : sb.ch(e[i], f[i], g[i]);
    const Word S0 = sb.xor3w(ShaBuilder::rotr(a, 2), ShaBuilder::rotr(a, 13), ShaBuilder::rotr(a, 22));
    for (int i = 0; i < 32; i++) {
      if (bit_is_const(a[i]) && bit_is_const(b[i]) && bit_is_const(c[i]))
        mjw[i] = bit_const((a[i].b & b[i].b) | (a[i].b & c[i].b) | (b[i].b & c[i].b));
      else
        mjw[i] = sb.maj(a[i], b[i], c[i]);
    }
    const Word kw = ShaBuilder::word_const(kShaK[t]);
    const Word ne = sb.add_mod32({d, h, S1, chw, kw, W[t]});
    const Word na = sb.add_mod32({h, S1, chw, kw, W[t], S0, mjw});
    h = g; g = f; f = e; e = ne; d = c; c = b; b = a; a = na;
  }
  const Word fin[8] = {a, b, c, d, e, f, g, h};
  for (int j = 0; j < 8; j++) {
    if (out_base) {
      uint32_t outw[32];   // result bit i (weight 2^i) of word j is output bit 32 j + (31 - i)
      for (int i = 0; i < 32; i++) outw[i] = out_base + 32 * j + (31 - i);
      out[j] = sb.add_mod32({st[j], fin[j]}, outw);
    } else {
      out[j] = sb.add_mod32({st[j], fin[j]});
    }
  }
}

// plain SHA-256 (FIPS 180-4 padding, multi-block) of a message given as bits (MSB-first per byte; wires or
// constants); the digest bits land on the 256 wires from out_base on
void sha256_bits(ShaBuilder& sb, const std::vector<Bit>& mbits, uint32_t out_base) {
  using Word = ShaBuilder::Word;
  const uint64_t bitlen = mbits.size();
  const uint32_t nb = (uint32_t)((bitlen / 8 + 9 + 63) / 64);
  auto padded_bit = [&](uint64_t k) -> Bit {   // bit k (MSB-first) of the padded message
    if (k < bitlen) return mbits[k];
    if (k == bitlen) return bit_const(1);
    const uint64_t total = (uint64_t)nb * 512;
    if (k >= total - 64) return bit_const((int)((bitlen >> (total - 1 - k)) & 1));
    return bit_const(0);
  };
  Word st[8];
  for (int j = 0; j < 8; j++) st[j] = ShaBuilder::word_const(kShaIV[j]);
  for (uint32_t blk = 0; blk < nb; blk++) {
    Word W16[16], out[8];
    for (int j = 0; j < 16; j++)
      for (int k = 0; k < 32; k++) W16[j][31 - k] = padded_bit((uint64_t)blk * 512 + 32 * j + k);
    sha_compress(sb, st, W16, out, blk + 1 == nb ? out_base : 0u);
    for (int j = 0; j < 8; j++) st[j] = out[j];
  }
}

// wires: 0 = one, 1..256 = public outputs (digest bits, MSB-first), then the private message bits (MSB-first,
// boolean-constrained), then gates.  chain = true: digest_{i+1} = SHA-256(digest_i), `blocks` times, 32-byte
// message.  chain = false: plain SHA-256 of the `len`-byte message (padding per FIPS 180-4, len is a
// compile-time constant of the circuit, like the fixed-length Sha256 gadgets of circomlib).
void build_sha256(ShaBuilder& sb, bool chain, uint32_t blocks, const uint8_t* msg, uint32_t len) {
  using Word = ShaBuilder::Word;
  for (int i = 0; i < 256; i++) sb.new_wire(0);   // outputs, values filled by the last addition
  std::vector<Bit> mbits((size_t)len * 8);
  for (uint32_t k = 0; k < len * 8; k++) {
    const uint32_t wire = sb.new_wire((msg[k / 8] >> (7 - (k & 7))) & 1);
    sb.boolean(wire);
    mbits[k] = bit_wire(wire);
  }
  Word iv[8];
  for (int j = 0; j < 8; j++) iv[j] = ShaBuilder::word_const(kShaIV[j]);
  if (chain) {
    Word m[8];
    for (int j = 0; j < 8; j++)
      for (int k = 0; k < 32; k++) m[j][31 - k] = mbits[32 * j + k];
    for (uint32_t blk = 0; blk < blocks; blk++) {
      Word W16[16], out[8];
      for (int j = 0; j < 8; j++) W16[j] = m[j];
      W16[8] = ShaBuilder::word_const(0x80000000u);
      for (int j = 9; j < 15; j++) W16[j] = ShaBuilder::word_const(0);
      W16[15] = ShaBuilder::word_const(256);
      sha_compress(sb, iv, W16, out, blk + 1 == blocks ? 1u : 0u);
      for (int j = 0; j < 8; j++) m[j] = out[j];
    }
  } else {
    sha256_bits(sb, mbits, 1);
  }
  sb.c.n = (uint32_t)sb.w.size();
  sb.c.p = 256;
  sb.c.m = (uint32_t)sb.c.rowA.size() - 1;
}

// The NZCP circuit's PUBLIC INTERFACE on a fixed pass layout (/root/reference/circuits/nzcptpl.circom:447-602,
// /root/reference/test/nzcp.js:41-47): public signals [0..255] = SHA-256("given,family,dob") bits,
// [256..511] = SHA-256(ToBeSigned) bits, [512] = exp.  The reference finds the three strings and `exp` by CBOR
// parsing inside the circuit (cbortpl.circom, not restated here); this circuit takes their byte offsets as
// circuit constants instead -- sound for passes of that layout: the credential string is wired to the SAME
// ToBeSigned bit wires at seg_off[k] (no copies), commas are constants, and exp is one linear row over the
// 32 ToBeSigned bits at exp_off.
void build_nzcp_fixed_layout(ShaBuilder& sb, const uint8_t* tbs, uint32_t len, const uint32_t seg_off[3],
                             const uint32_t seg_len[3], uint32_t exp_off) {
  for (int i = 0; i < 512; i++) sb.new_wire(0);   // the two digests
  const uint32_t exp_wire = sb.new_wire(0);
  std::vector<Bit> mbits((size_t)len * 8);
  for (uint32_t k = 0; k < len * 8; k++) {
    const uint32_t wire = sb.new_wire((tbs[k / 8] >> (7 - (k & 7))) & 1);
    sb.boolean(wire);
    mbits[k] = bit_wire(wire);
  }
  std::vector<Bit> subj;
  for (int sgi = 0; sgi < 3; sgi++) {
    if (sgi)
      for (int k = 0; k < 8; k++) subj.push_back(bit_const((',' >> (7 - k)) & 1));
    for (uint32_t k = 0; k < seg_len[sgi] * 8; k++) subj.push_back(mbits[(size_t)seg_off[sgi] * 8 + k]);
  }
  sha256_bits(sb, subj, 1);
  sha256_bits(sb, mbits, 257);
  uint64_t exp = 0;
  Lin a, b, z;
  for (int k = 0; k < 32; k++) {
    const Bit& bt = mbits[(size_t)exp_off * 8 + k];
    exp |= (uint64_t)sb.val(bt) << (31 - k);
    ShaBuilder::add(a, bt, (int64_t)1 << (31 - k));
  }
  sb.w[exp_wire] = exp;
  a.t.push_back({exp_wire, -1});
  b.t.push_back({0u, 1});
  sb.constrain(a, b, z);
  sb.c.n = (uint32_t)sb.w.size();
  sb.c.p = 513;
  sb.c.m = (uint32_t)sb.c.rowA.size() - 1;
}

#include "nzcp_gadgets.h"

// every row of the builder's R1CS evaluated on its witness: the index of the first row with <A,w><B,w> != <C,w>,
// or -1
int64_t first_unsatisfied_row(const CBuilder& cb) {
  const Circuit& c = cb.c;
  const uint32_t m = (uint32_t)c.rowA.size() - 1;
  auto dot = [&](const std::vector<Term>& t, uint32_t lo, uint32_t hi) {
    FrM s = fp_zero<FrParams>();
    for (uint32_t k = lo; k < hi; k++) s = fp_add(s, fp_mul(t[k].cf, cb.wire_val(t[k].s)));
    return s;
  };
  for (uint32_t r = 0; r < m; r++) {
    const FrM a = dot(c.tA, c.rowA[r], c.rowA[r + 1]), b = dot(c.tB, c.rowB[r], c.rowB[r + 1]);
    const FrM cc = dot(c.tC, c.rowC[r], c.rowC[r + 1]);
    if (!CBuilder::fr_eq(fp_mul(a, b), cc)) return (int64_t)r;
  }
  return -1;
}

void write_r1cs(const Circuit& c, uint32_t n_pub_out, uint32_t n_pub_in, Buf& b) {
  const size_t nnz = c.tA.size() + c.tB.size() + c.tC.size();
  const size_t s1 = 4 + 32 + 16 + 8 + 4, s2 = (size_t)c.m * 12 + nnz * 36, s3 = (size_t)c.n * 8;
  b.reserve(12 + 3 * 12 + s1 + s2 + s3);
  b.put("r1cs", 4); b.u32(1); b.u32(3);
  static const uint32_t R[8] = G16_FR_P;
  b.u32(1); b.u64(s1);
  b.u32(32); b.put(R, 32); b.u32(c.n); b.u32(n_pub_out); b.u32(n_pub_in); b.u32(c.n - 1 - n_pub_out - n_pub_in);
  b.u64(c.n); b.u32(c.m);
  b.u32(2); b.u64(s2);
  const std::vector<Term>* ts[3] = {&c.tA, &c.tB, &c.tC};
  const std::vector<uint32_t>* rs[3] = {&c.rowA, &c.rowB, &c.rowC};
  for (uint32_t r = 0; r < c.m; r++)
    for (int k = 0; k < 3; k++) {
      const uint32_t lo = (*rs[k])[r], hi = (*rs[k])[r + 1];
      b.u32(hi - lo);
      for (uint32_t t = lo; t < hi; t++) {
        b.u32((*ts[k])[t].s);
        const Fr plain = fp_from_mont((*ts[k])[t].cf);
        b.put(plain.v, 32);
      }
    }
  b.u32(3); b.u64(s3);
  for (uint32_t i = 0; i < c.n; i++) b.u64(i);
}

}  // namespace
}  // namespace g16

static int sha_emit(ShaBuilder& sb, uint64_t seed, int threads, uint8_t** zkey, size_t* zkey_len, uint8_t** wtns,
                    size_t* wtns_len, uint8_t** vkey, size_t* vkey_len, uint8_t** r1cs, size_t* r1cs_len) {
  if ((uint64_t)sb.c.m + sb.c.p + 1 > ((uint64_t)1 << 27)) { set_error("sha256 circuit too large"); return G16_E_ARG; }
  if (wtns && wtns_len) {
    std::vector<FrM> w(sb.w.size());
    const FrM one = fr_one(), zero = fp_zero<FrParams>();
    for (size_t i = 0; i < w.size(); i++) w[i] = sb.w[i] == 0 ? zero : (sb.w[i] == 1 ? one : fr_u64(sb.w[i]));
    Buf b;
    write_wtns(w, b);
    *wtns = b.p;
    *wtns_len = b.len;
  }
  if (r1cs && r1cs_len) {
    Buf b;
    write_r1cs(sb.c, sb.c.p, 0, b);
    *r1cs = b.p;
    *r1cs_len = b.len;
  }
  if (zkey && zkey_len) return setup_core(sb.c, seed, threads, zkey, zkey_len, vkey, vkey_len);
  return G16_OK;
}

// Test-only: the SHA-256 circuits above, keyed with a known trapdoor.  Any output pointer may be NULL.
// r1cs: iden3 .r1cs v1 image of the same constraint system (for snarkjs / tools/r1cs_setup.py).
extern "C" int g16_sha256_chain_setup(uint32_t blocks, const uint8_t msg[32], uint64_t seed, int threads,
                                      uint8_t** zkey, size_t* zkey_len, uint8_t** wtns, size_t* wtns_len,
                                      uint8_t** vkey, size_t* vkey_len, uint8_t** r1cs, size_t* r1cs_len) {
  if (!msg || blocks == 0 || blocks > 4096) { set_error("sha256 chain: bad arguments"); return G16_E_ARG; }
  ShaBuilder sb;
  build_sha256(sb, true, blocks, msg, 32);
  return sha_emit(sb, seed, threads, zkey, zkey_len, wtns, wtns_len, vkey, vkey_len, r1cs, r1cs_len);
}

extern "C" int g16_sha256_message_setup(const uint8_t* msg, uint32_t len, uint64_t seed, int threads,
                                        uint8_t** zkey, size_t* zkey_len, uint8_t** wtns, size_t* wtns_len,
                                        uint8_t** vkey, size_t* vkey_len, uint8_t** r1cs, size_t* r1cs_len) {
  if ((!msg && len) || len > (1u << 20)) { set_error("sha256 message: bad arguments"); return G16_E_ARG; }
  ShaBuilder sb;
  const uint8_t none = 0;
  build_sha256(sb, false, 0, msg ? msg : &none, len);
  return sha_emit(sb, seed, threads, zkey, zkey_len, wtns, wtns_len, vkey, vkey_len, r1cs, r1cs_len);
}

extern "C" int g16_nzcp_fixed_layout_setup(const uint8_t* tbs, uint32_t len, const uint32_t seg_off[3],
                                          const uint32_t seg_len[3], uint32_t exp_off, uint64_t seed, int threads,
                                          uint8_t** zkey, size_t* zkey_len, uint8_t** wtns, size_t* wtns_len,
                                          uint8_t** vkey, size_t* vkey_len, uint8_t** r1cs, size_t* r1cs_len) {
  if (!tbs || !seg_off || !seg_len || len == 0 || len > 4096 || (uint64_t)exp_off + 4 > len) {
    set_error("nzcp fixed layout: bad arguments");
    return G16_E_ARG;
  }
  for (int k = 0; k < 3; k++)
    if ((uint64_t)seg_off[k] + seg_len[k] > len) { set_error("nzcp fixed layout: segment out of range"); return G16_E_ARG; }
  ShaBuilder sb;
  build_nzcp_fixed_layout(sb, tbs, len, seg_off, seg_len, exp_off);
  return sha_emit(sb, seed, threads, zkey, zkey_len, wtns, wtns_len, vkey, vkey_len, r1cs, r1cs_len);
}

// ---------------------------------------------------------------------------------------------------------
// The NZCP circuit library as native gadgets (nzcp_gadgets.h), one template at a time -- the twins of the
// reference's *_test.circom entry points (/root/reference/circuits/*_test.circom), so that its test vectors
// (/root/reference/test/cbor.js, quinSelector.js, nzcp.js) can be replayed against the natively built rows.
namespace g16 {
namespace {
using V = CBuilder::V;

int run_gadget(CBuilder& cb, const std::string& name, const uint32_t* prm, uint32_t nprm, const uint64_t* in, uint32_t nin,
               std::vector<V>& outs) {
  uint32_t pos_in = 0;
  auto P = [&](uint32_t i) -> uint32_t { return i < nprm ? prm[i] : 0u; };
  auto input = [&]() -> V {   // a private input wire
    const uint64_t v = pos_in < nin ? in[pos_in] : 0;
    pos_in++;
    return cb.of_wire(cb.new_wire(v));
  };
  auto inputs = [&](uint32_t n) { std::vector<V> r; for (uint32_t i = 0; i < n; i++) r.push_back(input()); return r; };
  if (name == "getType") { outs = {cb.get_type(input())}; }
  else if (name == "getX") { outs = {cb.get_x(input())}; }
  else if (name == "quinSelector") { const std::vector<V> arr = inputs(P(0)); const V idx = input(); outs = {cb.quin_selector(arr, idx)}; }
  else if (name == "getV") { const std::vector<V> b = inputs(P(0)); const V pos = input(); outs = {cb.get_v(b, pos)}; }
  else if (name == "decodeUint23") { outs = {cb.decode_uint23(input())}; }
  else if (name == "decodeUint") {   // inputs: bytes[N], pos, v
    const std::vector<V> b = inputs(P(0)); const V pos = input(); const V v = input();
    const CBuilder::UintOut o = cb.decode_uint(b, pos, v);
    outs = {o.value, o.next_pos};
  } else if (name == "readType") {
    const std::vector<V> b = inputs(P(0)); const V pos = input();
    const CBuilder::TypeOut o = cb.read_type(b, pos);
    outs = {o.next_pos, o.type, o.v};
  } else if (name == "skipValueScalar") { const std::vector<V> b = inputs(P(0)); const V pos = input(); outs = {cb.skip_value_scalar(b, pos)}; }
  else if (name == "skipValue") { const std::vector<V> b = inputs(P(0)); const V pos = input(); outs = {cb.skip_value(b, pos, P(1))}; }
  else if (name == "stringEquals") {   // params: N, constLen, const bytes...; inputs: bytes[N], pos, len
    const uint32_t cl = P(1);
    std::vector<uint8_t> cbts(cl);
    for (uint32_t i = 0; i < cl; i++) cbts[i] = (uint8_t)P(2 + i);
    const std::vector<V> b = inputs(P(0)); const V pos = input(); const V len = input();
    outs = {cb.string_equals(b, pos, len, cbts.data(), cl)};
  } else if (name == "readStringLength") {
    const std::vector<V> b = inputs(P(0)); const V pos = input();
    const CBuilder::LenOut o = cb.read_string_length(b, pos);
    outs = {o.len, o.next_pos};
  } else if (name == "readMapLength") {
    const std::vector<V> b = inputs(P(0)); const V pos = input();
    const CBuilder::LenOut o = cb.read_map_length(b, pos);
    outs = {o.len, o.next_pos};
  } else if (name == "copyString") {
    const std::vector<V> b = inputs(P(0)); const V pos = input();
    const CBuilder::CopyOut o = cb.copy_string(b, pos, P(1));
    outs = o.out; outs.push_back(o.next_pos); outs.push_back(o.len);
  } else if (name == "findVCAndExp" || name == "findCredSubj") {   // params: N, maxArr, maxMap; inputs: bytes[N], pos, mapLen
    static const uint8_t kVC[2] = {118, 99};
    static const uint8_t kCS[17] = {99, 114, 101, 100, 101, 110, 116, 105, 97, 108, 83, 117, 98, 106, 101, 99, 116};
    const std::vector<V> b = inputs(P(0)); const V pos = input(); const V ml = input();
    const bool vc = name == "findVCAndExp";
    const CBuilder::FindOut o = cb.find_in_map(b, pos, ml, P(1), P(2), vc ? kVC : kCS, vc ? 2 : 17, vc);
    outs = {o.needle_pos};
    if (vc) outs.push_back(o.exp_pos);
  } else if (name == "readCredSubj") {   // params: N, maxBufferLen; inputs: bytes[N], pos, mapLen
    const std::vector<V> b = inputs(P(0)); const V pos = input(); const V ml = input();
    const CBuilder::CredSubj o = cb.read_cred_subj(b, pos, ml, P(1));
    outs = o.given; outs.push_back(o.given_len);
    outs.insert(outs.end(), o.family.begin(), o.family.end()); outs.push_back(o.family_len);
    outs.insert(outs.end(), o.dob.begin(), o.dob.end()); outs.push_back(o.dob_len);
  } else if (name == "concatCredSubj") {   // params: maxBufferLen; inputs: given[M], givenLen, family[M], familyLen, dob[M], dobLen
    CBuilder::CredSubj cs;
    cs.given = inputs(P(0)); cs.given_len = input();
    cs.family = inputs(P(0)); cs.family_len = input();
    cs.dob = inputs(P(0)); cs.dob_len = input();
    const CBuilder::Concat o = cb.concat_cred_subj(cs, P(0));
    outs = o.result; outs.push_back(o.result_len);
  } else if (name == "sha256Var") {   // params: blockSpace; inputs: len_bits, then the message BYTES (bits are derived)
    const int bs = (int)P(0);
    const V len = input();
    std::vector<Bit> bits((size_t)512 << bs, bit_const(0));
    const uint32_t out_base = cb.new_wire(0);
    for (int i = 1; i < 256; i++) cb.new_wire(0);
    for (uint32_t j = 0; j + 1 < nin && j < (64u << bs); j++)
      for (int i = 0; i < 8; i++) {
        const uint32_t wire = cb.new_wire((in[1 + j] >> (7 - i)) & 1);
        cb.boolean(wire);
        bits[(size_t)j * 8 + (size_t)i] = bit_wire(wire);
      }
    cb.sha256_var(bits, len, bs, out_base);
    for (int i = 0; i < 256; i++) outs.push_back(cb.of_wire(out_base + (uint32_t)i));
  } else {
    set_error("unknown gadget: " + name);
    return G16_E_ARG;
  }
  return G16_OK;
}
}  // namespace
}  // namespace g16

// Test-only: build ONE template of the NZCP circuit library over the given private inputs, check every emitted
// R1CS row on the computed witness and return the template's outputs.  G16_E_STATE + "constraint not satisfied:
// ..." when the inputs violate one of the template's `===` / Num2Bits range constraints (where circom's witness
// generator throws).  outputs: *nout in = capacity, out = count.
extern "C" int g16_nzcp_gadget(const char* name, const uint32_t* params, uint32_t nparams, const uint64_t* inputs,
                               uint32_t nin, uint64_t* outputs, uint32_t* nout, uint32_t* n_constraints) {
  if (!name || !nout || (nparams && !params) || (nin && !inputs)) { set_error("NULL argument"); return G16_E_ARG; }
  CBuilder cb;
  std::vector<CBuilder::V> outs;
  int rc = run_gadget(cb, name, params, nparams, inputs, nin, outs);
  if (rc) return rc;
  if (n_constraints) *n_constraints = (uint32_t)cb.c.rowA.size() - 1;
  const int64_t bad = first_unsatisfied_row(cb);
  if (!cb.ok) {
    set_error("constraint not satisfied: " + cb.fail);
    return G16_E_STATE;
  }
  if (bad >= 0) { set_error("internal: R1CS row " + std::to_string(bad) + " is not satisfied by the computed witness"); return G16_E_HIP; }
  if (outs.size() > *nout) { set_error("output buffer too small"); return G16_E_ARG; }
  for (size_t i = 0; i < outs.size(); i++) {
    uint64_t v = 0;
    if (!CBuilder::small_of(outs[i].val, v)) { set_error("gadget output is not a small integer"); return G16_E_STATE; }
    if (outputs) outputs[i] = v;
  }
  *nout = (uint32_t)outs.size();
  return G16_OK;
}

// Test-only: the full NZCPPubIdentity(IsLive, MaxToBeSignedBytes, MaxCborArrayLenVC, MaxCborMapLenVC,
// MaxCborArrayLenCredSubj, MaxCborMapLenCredSubj, CredSubjMaxBufferSpace) constraint system
// (/root/reference/circuits/nzcptpl.circom:433; nzcp_exampleTest.circom = (0, 314, 0, 4, 2, 4, 5),
// nzcp_liveTest.circom = (1, 355, 0, 4, 2, 4, 6)) with the CBOR search IN the circuit, its witness for the given
// ToBeSigned bytes, and a trapdoor proving key.  params = the seven template parameters in that order.
extern "C" int g16_nzcp_circuit_setup(const uint32_t params[7], const uint8_t* tbs, uint32_t len, uint64_t seed,
                                      int threads, uint8_t** zkey, size_t* zkey_len, uint8_t** wtns, size_t* wtns_len,
                                      uint8_t** vkey, size_t* vkey_len, uint8_t** r1cs, size_t* r1cs_len,
                                      uint32_t* n_constraints) {
  if (!params || !tbs || params[1] == 0 || params[1] > 503 || params[6] < 2 || params[6] > 6 || len > params[1]) {
    set_error("nzcp circuit: bad arguments");
    return G16_E_ARG;
  }
  CBuilder cb;
  cb.nzcp_pub_identity(params[0] != 0, params[1], params[2], params[3], params[4], params[5], params[6], tbs, len);
  if (n_constraints) *n_constraints = cb.c.m;
  if (!cb.ok) { set_error("constraint not satisfied: " + cb.fail); return G16_E_STATE; }
  const int64_t bad = first_unsatisfied_row(cb);
  if (bad >= 0) { set_error("internal: R1CS row " + std::to_string(bad) + " is not satisfied by the computed witness"); return G16_E_HIP; }
  if ((uint64_t)cb.c.m + cb.c.p + 1 > ((uint64_t)1 << 27)) { set_error("nzcp circuit too large"); return G16_E_ARG; }
  if (wtns && wtns_len) {
    std::vector<FrM> w(cb.w.size());
    for (size_t i = 0; i < w.size(); i++) w[i] = cb.wire_val((uint32_t)i);
    Buf b;
    write_wtns(w, b);
    *wtns = b.p;
    *wtns_len = b.len;
  }
  if (r1cs && r1cs_len) {
    Buf b;
    write_r1cs(cb.c, cb.c.p, 0, b);
    *r1cs = b.p;
    *r1cs_len = b.len;
  }
  if (zkey && zkey_len) return setup_core(cb.c, seed, threads, zkey, zkey_len, vkey, vkey_len);
  return G16_OK;
}

// ------------------------------------------------------------------ .r1cs reader
// iden3 r1cs v1: section 1 header {n8, prime, nWires, nPubOut, nPubIn, nPrvIn, nLabels u64,
// nConstraints}, section 2 constraints: A, B, C each {nTerms u32, nTerms x (wireId u32, coef n8 LE)}.
// nPublic of the zkey = nPubOut + nPubIn ([EXT] r1csfile 0.0.35, pin /root/reference/yarn.lock:909-917).
static int read_r1cs(const uint8_t* buf, size_t len, Circuit& c) {
  auto bad = [](const char* why) { set_error(std::string("r1cs: ") + why); return G16_E_FORMAT; };
  if (!buf || len < 12 || memcmp(buf, "r1cs", 4) != 0) { set_error("r1cs: Invalid File format"); return G16_E_FORMAT; }
  uint32_t version, nsec;
  memcpy(&version, buf + 4, 4);
  memcpy(&nsec, buf + 8, 4);
  if (version > 1) { set_error("Version not supported"); return G16_E_FORMAT; }
  const uint8_t *s1 = nullptr, *s2 = nullptr;
  uint64_t l1 = 0, l2 = 0;
  size_t pos = 12;
  for (uint32_t i = 0; i < nsec; i++) {
    if (pos + 12 > len) return bad("truncated section table");
    uint32_t id; uint64_t sz;
    memcpy(&id, buf + pos, 4); memcpy(&sz, buf + pos + 4, 8); pos += 12;
    if (sz > len - pos) return bad("truncated section");
    if (id == 1 && !s1) { s1 = buf + pos; l1 = sz; }
    if (id == 2 && !s2) { s2 = buf + pos; l2 = sz; }
    pos += sz;
  }
  if (!s1 || !s2 || l1 < 4 + 32 + 16 + 8 + 4) return bad("missing header or constraint section");
  uint32_t n8; memcpy(&n8, s1, 4);
  static const uint32_t Rp[8] = G16_FR_P;
  if (n8 != 32 || memcmp(s1 + 4, Rp, 32) != 0) return bad("field is not the bn128 scalar field");
  uint32_t nWires, nPubOut, nPubIn, nPrvIn, nCons;
  memcpy(&nWires, s1 + 36, 4); memcpy(&nPubOut, s1 + 40, 4); memcpy(&nPubIn, s1 + 44, 4);
  memcpy(&nPrvIn, s1 + 48, 4); memcpy(&nCons, s1 + 60, 4);
  (void)nPrvIn;
  c.n = nWires; c.p = nPubOut + nPubIn; c.m = nCons;
  if (c.n < c.p + 1 || c.m == 0 || (uint64_t)nPubOut + nPubIn >= nWires) return bad("inconsistent header");
  // an untrusted header must not size the allocations: every constraint takes >= 12 bytes of section 2, and a wire
  // that appears nowhere still has its 8-byte entry in the wire map (section 3) when the file carries one
  if ((uint64_t)nCons * 12 > l2) return bad("constraint count exceeds the constraint section");
  if (nWires > (1u << 28)) return bad("too many wires");
  {
    const uint8_t* s3 = nullptr;
    uint64_t l3 = 0;
    size_t q3 = 12;
    for (uint32_t i = 0; i < nsec; i++) {
      uint32_t id; uint64_t sz;
      memcpy(&id, buf + q3, 4); memcpy(&sz, buf + q3 + 4, 8); q3 += 12;
      if (id == 3 && !s3) { s3 = buf + q3; l3 = sz; }
      q3 += sz;
    }
    if (s3 && l3 != (uint64_t)nWires * 8) return bad("wire map does not match the wire count");
    if (!s3 && (uint64_t)nWires > l2) return bad("wire count exceeds the file");
  }
  c.rowA.assign(1, 0); c.rowB.assign(1, 0); c.rowC.assign(1, 0);
  const uint8_t* q = s2;
  const uint8_t* end = s2 + l2;
  std::vector<Term>* dst[3] = {&c.tA, &c.tB, &c.tC};
  std::vector<uint32_t>* rows[3] = {&c.rowA, &c.rowB, &c.rowC};
  for (uint32_t r = 0; r < nCons; r++) {
    for (int k = 0; k < 3; k++) {
      if (q + 4 > end) return bad("truncated constraint");
      uint32_t nt; memcpy(&nt, q, 4); q += 4;
      if ((uint64_t)nt * 36 > (uint64_t)(end - q)) return bad("truncated constraint");
      for (uint32_t t = 0; t < nt; t++) {
        uint32_t wire; memcpy(&wire, q, 4);
        if (wire >= nWires) return bad("wire id out of range");
        Fr cf; memcpy(cf.v, q + 4, 32);
        q += 36;
        dst[k]->push_back({wire, fp_to_mont(cf)});
      }
      rows[k]->push_back((uint32_t)dst[k]->size());
    }
  }
  return G16_OK;
}

// Test-only trapdoor setup of a REAL circuit: .r1cs in, snarkjs-layout .zkey (+ vkey points) out.
extern "C" int g16_r1cs_setup(const uint8_t* r1cs, size_t r1cs_len, uint64_t seed, int threads, uint8_t** zkey,
                              size_t* zkey_len, uint8_t** vkey, size_t* vkey_len) {
  if (!zkey || !zkey_len) { set_error("NULL argument"); return G16_E_ARG; }
  try {
    Circuit c;
    int rc = read_r1cs(r1cs, r1cs_len, c);
    if (rc) return rc;
    uint64_t need = (uint64_t)c.m + c.p + 1;
    if (need > ((uint64_t)1 << 27)) { set_error("r1cs: circuit too large"); return G16_E_ARG; }
    return setup_core(c, seed, threads, zkey, zkey_len, vkey, vkey_len);
  } catch (const std::bad_alloc&) {   // no C++ exception crosses the C ABI
    set_error("setup: out of memory");
    return G16_E_STATE;
  }
}

// ------------------------------------------------------------------ PLONK setup (test-only: tau is known)
// Stands in for `snarkjs plonk setup c.r1cs pot.ptau c.zkey` (/root/reference/Makefile:31; [EXT] snarkjs 0.4.12
// plonk_setup.js): R1CS -> PLONK gates as snarkjs does it (public-input gates first; linear combinations reduced to
// one signal by addition gates taken from the front of a queue, results appended), copy-constraint permutation,
// selector / sigma polynomials as N coefficients + 4N evaluations, N + 6 powers of tau -- here from a KNOWN tau
// (seed), so the commitments are [Q(tau)]G by one fixed-base multiplication each.  Restated in oracle/plonk.py::setup,
// whose zkey this must equal byte for byte for the same tau (tests/test_gpu_plonk.py).  The transforms run on the
// device (plonk.hip::plonk_setup_polys); with_lagrange = 0 writes an EMPTY section 13.
namespace g16 {
namespace {

struct PlonkGate { uint32_t sl, sr, so; FrM qm, ql, qr, qo, qc; };
struct PlonkAdd { uint32_t s1, s2; FrM f1, f2; };

struct PlonkBuilder {
  std::vector<PlonkGate> gates;
  std::vector<PlonkAdd> adds;
  uint32_t nv = 0;
  FrM zero = fp_zero<FrParams>(), one = fr_one();
  struct LC { FrM k; std::vector<Term> t; };   // constant + terms on distinct non-zero wires (first-appearance order)
  // merge duplicate wires, split the constant off, drop zero coefficients; keeps the order of first appearance (what
  // iterating a JS object with integer keys does NOT do -- snarkjs walks ascending signal ids -- so sort by id)
  LC lc_of(const Term* b, const Term* e) {
    LC r;
    r.k = zero;
    std::vector<Term> v(b, e);
    std::sort(v.begin(), v.end(), [](const Term& x, const Term& y) { return x.s < y.s; });
    size_t i = 0;
    while (i < v.size()) {
      FrM sum = zero;
      const uint32_t wire = v[i].s;
      while (i < v.size() && v[i].s == wire) sum = fp_add(sum, v[i++].cf);
      if (fp_is_zero(sum)) continue;
      if (wire == 0) r.k = sum;
      else r.t.push_back({wire, sum});
    }
    return r;
  }
  // reduceCoefs: while more than max_c terms, the first two become one addition gate whose output goes to the back
  void reduce(LC& lc, size_t max_c) {
    size_t head = 0;
    while (lc.t.size() - head > max_c) {
      const Term c1 = lc.t[head], c2 = lc.t[head + 1];
      head += 2;
      const uint32_t so = nv++;
      gates.push_back({c1.s, c2.s, so, zero, fp_neg(c1.cf), fp_neg(c2.cf), one, zero});
      adds.push_back({c1.s, c2.s, c1.cf, c2.cf});
      lc.t.push_back({so, one});
    }
    lc.t.erase(lc.t.begin(), lc.t.begin() + head);
    while (lc.t.size() < max_c) lc.t.push_back({0u, zero});
  }
  void add_sum(LC lc) {
    reduce(lc, 3);
    gates.push_back({lc.t[0].s, lc.t[1].s, lc.t[2].s, zero, lc.t[0].cf, lc.t[1].cf, lc.t[2].cf, lc.k});
  }
};

}  // namespace
// plonk.hip
int plonk_setup_polys(int device, int L, const Fr* const evals[8], uint8_t* const out[8]);
}  // namespace g16

namespace g16 {
int plonk_setup_commit(int device, const uint8_t* tau_g1, uint32_t N, const uint8_t* const coefs[8], uint8_t* out);   // plonk.hip
}
// where the powers of tau come from: a known tau (test-only), or the points of a .ptau file
struct PlonkTauSrc {
  bool known = true;
  uint64_t seed = 0;
  const uint8_t* tau_g1 = nullptr;   // .ptau section 2: [tau^i]G1, affine Montgomery LE
  uint64_t n_g1 = 0;
  const uint8_t* tau_g2_1 = nullptr; // .ptau section 3, point 1: [tau]G2
  uint32_t power = 0;
};
static int plonk_setup_core(const uint8_t* r1cs, size_t r1cs_len, const PlonkTauSrc& src, int device, int with_lagrange,
                            uint8_t** zkey, size_t* zkey_len) {
  const uint64_t seed = src.seed;
  Circuit c;
  int rc = read_r1cs(r1cs, r1cs_len, c);
  if (rc) return rc;
  PlonkBuilder pb;
  pb.nv = c.n;
  for (uint32_t s = 1; s <= c.p; s++) pb.gates.push_back({s, 0u, 0u, pb.zero, pb.one, pb.zero, pb.zero, pb.zero});
  for (uint32_t r = 0; r < c.m; r++) {
    PlonkBuilder::LC a = pb.lc_of(c.tA.data() + c.rowA[r], c.tA.data() + c.rowA[r + 1]);
    PlonkBuilder::LC b = pb.lc_of(c.tB.data() + c.rowB[r], c.tB.data() + c.rowB[r + 1]);
    PlonkBuilder::LC cc = pb.lc_of(c.tC.data() + c.rowC[r], c.tC.data() + c.rowC[r + 1]);
    const bool a0 = a.t.empty() && fp_is_zero(a.k), b0 = b.t.empty() && fp_is_zero(b.k);
    if (a0 || b0) {
      pb.add_sum(cc);
    } else if (a.t.empty() || b.t.empty()) {   // a constant times a linear combination: k * other - C = 0
      const FrM kk = a.t.empty() ? a.k : b.k;
      const PlonkBuilder::LC& other = a.t.empty() ? b : a;
      std::vector<Term> j;
      j.push_back({0u, fp_sub(fp_mul(kk, other.k), cc.k)});
      for (const Term& t : other.t) j.push_back({t.s, fp_mul(kk, t.cf)});
      for (const Term& t : cc.t) j.push_back({t.s, fp_neg(t.cf)});
      pb.add_sum(pb.lc_of(j.data(), j.data() + j.size()));
    } else {
      pb.reduce(a, 1);
      pb.reduce(b, 1);
      pb.reduce(cc, 1);
      pb.gates.push_back({a.t[0].s, b.t[0].s, cc.t[0].s, fp_mul(a.t[0].cf, b.t[0].cf), fp_mul(a.t[0].cf, b.k),
                          fp_mul(a.k, b.t[0].cf), fp_neg(cc.t[0].cf), fp_sub(fp_mul(a.k, b.k), cc.k)});
    }
  }
  const size_t ng = pb.gates.size();
  int L = 3;   // (the quotient polynomial has 3N + 6 coefficients and must fit 4N: snarkjs, too, starts at 2^3)
  while (((size_t)1 << L) < ng) L++;
  if (L > 24) { set_error("plonk setup: circuit too large (more than 2^24 gates)"); return G16_E_ARG; }
  const size_t N = (size_t)1 << L;
  if (!src.known && ((uint32_t)L > src.power || N + 6 > src.n_g1)) {
    set_error("circuit too big for this power of tau ceremony. " + std::to_string(ng) + " > 2**" + std::to_string(src.power));
    return G16_E_ARG;
  }
  Xo trng(seed + 1);
  FrM tau;
  do { tau = trng.rand_fr(); } while (fp_is_zero(tau));
  const FrM w1 = host_root(L);
  // k1, k2: smallest values whose cosets are disjoint from H and from each other
  auto pow_n = [&](FrM x) { for (int i = 0; i < L; i++) x = fp_sqr(x); return x; };
  const FrM one = fr_one();
  uint64_t k1v = 2;
  while (fp_eq(pow_n(fr_u64(k1v)), one)) k1v++;
  uint64_t k2v = k1v + 1;
  while (fp_eq(pow_n(fr_u64(k2v)), one) || fp_eq(pow_n(fp_mul(fr_u64(k2v), fp_inv(fr_u64(k1v)))), one)) k2v++;
  const FrM k1 = fr_u64(k1v), k2 = fr_u64(k2v);
  // evaluation vectors: 5 selectors, 3 sigmas
  std::vector<std::vector<Fr>> ev(8, std::vector<Fr>(N, fp_zero<FrParams>()));
  std::vector<uint32_t> maps[3];
  for (auto& m : maps) m.assign(N, 0u);
  for (size_t i = 0; i < ng; i++) {
    const PlonkGate& g = pb.gates[i];
    maps[0][i] = g.sl; maps[1][i] = g.sr; maps[2][i] = g.so;
    ev[0][i] = g.qm; ev[1][i] = g.ql; ev[2][i] = g.qr; ev[3][i] = g.qo; ev[4][i] = g.qc;
  }
  {
    std::vector<FrM> last(pb.nv);
    std::vector<uint32_t> first(pb.nv, 0xffffffffu);
    std::vector<uint8_t> seen(pb.nv, 0);
    FrM w = one;
    for (size_t i = 0; i < N; i++) {
      const FrM vals[3] = {w, fp_mul(w, k1), fp_mul(w, k2)};
      for (int col = 0; col < 3; col++) {
        const uint32_t sgn = maps[col][i];
        const size_t ppos = (size_t)col * N + i;
        if (seen[sgn]) ev[5 + col][i] = last[sgn];
        else { first[sgn] = (uint32_t)ppos; seen[sgn] = 1; }
        last[sgn] = vals[col];
      }
      w = fp_mul(w, w1);
    }
    for (uint32_t sgn = 0; sgn < pb.nv; sgn++)
      if (seen[sgn]) ev[5 + first[sgn] / N][first[sgn] % N] = last[sgn];
  }
  // file image
  const size_t nlag = with_lagrange ? (c.p > 0 ? c.p : 1) : 0;
  const size_t polb = N * 32 * 5;
  const size_t hdr = 4 + 32 + 4 + 32 + 20 + 64 + 8 * 64 + 128;
  const size_t sizes[15] = {0, 4, hdr, pb.adds.size() * 72, ng * 4, ng * 4, ng * 4, polb, polb, polb, polb, polb, 3 * polb,
                            nlag * polb, (N + 6) * 64};
  size_t total = 12;
  for (int i = 1; i <= 14; i++) total += 12 + sizes[i];
  Buf z;
  if (!z.reserve(total)) { set_error("plonk setup: out of memory"); return G16_E_STATE; }
  z.put("zkey", 4); z.u32(1); z.u32(14);
  uint8_t* sp[15] = {};
  for (uint32_t id = 1; id <= 14; id++) { z.u32(id); z.u64(sizes[id]); sp[id] = z.skip(sizes[id]); }
  { uint32_t two = 2; memcpy(sp[1], &two, 4); }
  for (size_t k = 0; k < pb.adds.size(); k++) {
    uint8_t* q = sp[3] + k * 72;
    memcpy(q, &pb.adds[k].s1, 4); memcpy(q + 4, &pb.adds[k].s2, 4);
    memcpy(q + 8, pb.adds[k].f1.v, 32); memcpy(q + 40, pb.adds[k].f2.v, 32);
  }
  for (int col = 0; col < 3; col++) memcpy(sp[4 + col], maps[col].data(), ng * 4);
  // polynomials on the device: coefficients + 4N evaluations straight into the sections
  {
    const Fr* evp[8];
    uint8_t* outp[8];
    for (int k = 0; k < 8; k++) {
      evp[k] = ev[k].data();
      outp[k] = k < 5 ? sp[7 + k] : sp[12] + (size_t)(k - 5) * polb;
    }
    if ((rc = plonk_setup_polys(device, L, evp, outp))) { free(z.p); return rc; }
    if (nlag) {   // Lagrange polynomials of the public inputs, 8 at a time
      std::vector<std::vector<Fr>> le(8, std::vector<Fr>(N));
      for (size_t j0 = 0; j0 < nlag; j0 += 8) {
        for (int k = 0; k < 8; k++) {
          std::fill(le[k].begin(), le[k].end(), fp_zero<FrParams>());
          const size_t j = j0 + k < nlag ? j0 + k : nlag - 1;
          le[k][j] = one;
          evp[k] = le[k].data();
          outp[k] = sp[13] + j * polb;
        }
        if ((rc = plonk_setup_polys(device, L, evp, outp))) { free(z.p); return rc; }
      }
    }
  }
  auto write_header = [&](const uint8_t* commitments /* 8 x 64 */, const uint8_t* x2 /* 128 */) {
    uint8_t* q = sp[2];
    static const uint32_t Qp[8] = G16_FQ_P, Rp[8] = G16_FR_P;
    uint32_t v32 = 32;
    memcpy(q, &v32, 4); memcpy(q + 4, Qp, 32); memcpy(q + 36, &v32, 4); memcpy(q + 40, Rp, 32);
    q += 72;
    const uint32_t hv[5] = {pb.nv, c.p, (uint32_t)N, (uint32_t)pb.adds.size(), (uint32_t)ng};
    memcpy(q, hv, 20); q += 20;
    memcpy(q, k1.v, 32); memcpy(q + 32, k2.v, 32); q += 64;
    memcpy(q, commitments, 8 * 64);
    memcpy(q + 8 * 64, x2, 128);
  };
  if (!src.known) {
    // a real ceremony's points: the first N + 6 powers are copied, the commitments are MSMs over them on the device
    memcpy(sp[14], src.tau_g1, (N + 6) * 64);
    const uint8_t* cf[8];
    for (int k = 0; k < 8; k++) cf[k] = k < 5 ? sp[7 + k] : sp[12] + (size_t)(k - 5) * polb;
    uint8_t cm[8 * 64];
    if ((rc = plonk_setup_commit(device, src.tau_g1, (uint32_t)N, cf, cm))) { free(z.p); return rc; }
    write_header(cm, src.tau_g2_1);
    *zkey = z.p;
    *zkey_len = z.len;
    return G16_OK;
  }
  // powers of tau and the commitments [P(tau)]G (tau is known: one fixed-base multiplication each)
  const int threads = (int)std::thread::hardware_concurrency() > 0 ? (int)std::thread::hardware_concurrency() : 1;
  FixedBase<FqOps> fb1;
  FixedBase<Fq2Ops> fb2;
  G1Affine g1;
  g1.x = fp_one<FqParams>();
  g1.y = fp_add(g1.x, g1.x);
  G2Affine g2;
  g2.x.a = Fq{G16_G2X0}; g2.x.b = Fq{G16_G2X1}; g2.y.a = Fq{G16_G2Y0}; g2.y.b = Fq{G16_G2Y1};
  build_table(fb1, g1, 8, threads);
  build_table(fb2, g2, 8, threads);
  {
    std::vector<FrM> pw(N + 6);
    FrM x = one;
    for (size_t i = 0; i < N + 6; i++) { pw[i] = x; x = fp_mul(x, tau); }
    if (device >= 0) {
      if ((rc = setup_fixed_mul_g1(device, fb1.tbl.data(), fb1.wb, fb1.nwin, pw.data(), N + 6, sp[14]))) { free(z.p); return rc; }
    } else {
      fixed_mul_many(fb1, pw.data(), N + 6, sp[14], threads);
    }
  }
  {
    FrM cm[8];
    for (int k = 0; k < 8; k++) {   // P(tau) by Horner over the coefficients just written
      const uint8_t* co = k < 5 ? sp[7 + k] : sp[12] + (size_t)(k - 5) * polb;
      FrM acc = fp_zero<FrParams>();
      for (size_t i = N; i-- > 0;) {
        FrM cf;
        memcpy(cf.v, co + i * 32, 32);
        acc = fp_add(fp_mul(acc, tau), cf);
      }
      cm[k] = acc;
    }
    uint8_t cmb[8 * 64], x2[128];
    fixed_mul_many(fb1, cm, 8, cmb, 1);
    fixed_mul_many(fb2, &tau, 1, x2, 1);
    write_header(cmb, x2);
  }
  *zkey = z.p;
  *zkey_len = z.len;
  return G16_OK;
}

extern "C" int g16_plonk_setup(const uint8_t* r1cs, size_t r1cs_len, uint64_t seed, int device, int with_lagrange,
                               uint8_t** zkey, size_t* zkey_len) {
  if (!r1cs || !zkey || !zkey_len) { set_error("NULL argument"); return G16_E_ARG; }
  PlonkTauSrc src;
  src.known = true;
  src.seed = seed;
  try {
    return plonk_setup_core(r1cs, r1cs_len, src, device, with_lagrange, zkey, zkey_len);
  } catch (const std::bad_alloc&) {   // no C++ exception crosses the C ABI
    set_error("plonk setup: out of memory");
    return G16_E_STATE;
  }
}

// `snarkjs plonk setup c.r1cs pot.ptau c.zkey` (/root/reference/Makefile:31) with a REAL powers-of-tau file: .ptau v1
// ([EXT] snarkjs powersoftau_utils.js: section 1 = n8, q, power, ceremonyPower; section 2 = 2^(power+1) - 1 points
// [tau^i]G1; section 3 = 2^power points [tau^i]G2; affine Montgomery LE).  The N + 6 powers are copied into the key and
// the eight selector / sigma commitments are MSMs over them on the device.
extern "C" int g16_plonk_setup_ptau(const uint8_t* r1cs, size_t r1cs_len, const uint8_t* ptau, size_t ptau_len, int device,
                                    int with_lagrange, uint8_t** zkey, size_t* zkey_len) {
  if (!r1cs || !ptau || !zkey || !zkey_len) { set_error("NULL argument"); return G16_E_ARG; }
  if (ptau_len < 12 || memcmp(ptau, "ptau", 4) != 0) { set_error("ptau: Invalid File format"); return G16_E_FORMAT; }
  uint32_t version, nsec;
  memcpy(&version, ptau + 4, 4);
  memcpy(&nsec, ptau + 8, 4);
  if (version > 1) { set_error("Version not supported"); return G16_E_FORMAT; }
  const uint8_t* sp[4] = {nullptr, nullptr, nullptr, nullptr};
  uint64_t sl[4] = {0, 0, 0, 0};
  size_t pos = 12;
  for (uint32_t i = 0; i < nsec; i++) {
    if (pos + 12 > ptau_len) { set_error("ptau: Invalid File format"); return G16_E_FORMAT; }
    uint32_t id;
    uint64_t sz;
    memcpy(&id, ptau + pos, 4);
    memcpy(&sz, ptau + pos + 4, 8);
    pos += 12;
    if (sz > ptau_len - pos) { set_error("ptau: Invalid File format"); return G16_E_FORMAT; }
    if (id >= 1 && id <= 3 && !sp[id]) { sp[id] = ptau + pos; sl[id] = sz; }
    pos += sz;
  }
  static const uint32_t Qp[8] = G16_FQ_P;
  uint32_t n8 = 0;
  if (sp[1] && sl[1] >= 4) memcpy(&n8, sp[1], 4);
  if (!sp[1] || !sp[2] || !sp[3] || sl[1] < 4 + 32 + 8 || n8 != 32 || memcmp(sp[1] + 4, Qp, 32) != 0) {
    set_error("ptau: Invalid File format (bn128 powers of tau expected)");
    return G16_E_FORMAT;
  }
  PlonkTauSrc src;
  src.known = false;
  memcpy(&src.power, sp[1] + 36, 4);
  if (src.power > 28 || sl[2] < (((uint64_t)2 << src.power) - 1) * 64 || sl[3] < 2 * 128) {
    set_error("ptau: Invalid File format");
    return G16_E_FORMAT;
  }
  src.tau_g1 = sp[2];
  src.n_g1 = sl[2] / 64;
  src.tau_g2_1 = sp[3] + 128;
  try {
    return plonk_setup_core(r1cs, r1cs_len, src, device, with_lagrange, zkey, zkey_len);
  } catch (const std::bad_alloc&) {
    set_error("plonk setup: out of memory");
    return G16_E_STATE;
  }
}

// File-path form of g16_plonk_setup_ptau for hosts that cannot hold a ceremony file in one buffer (a Node.js Buffer
// ends at 2 GB; powersOfTau28_hez_final_22.ptau is 4.6 GB): the inputs are mapped read-only, the key is written out.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
extern "C" int g16_plonk_setup_files(const char* r1cs_path, const char* ptau_path, const char* zkey_path, int device,
                                     int with_lagrange) {
  if (!r1cs_path || !ptau_path || !zkey_path) { set_error("NULL argument"); return G16_E_ARG; }
  struct Map {
    void* p = MAP_FAILED;
    size_t len = 0;
    int open_ro(const char* path) {
      const int fd = open(path, O_RDONLY);
      if (fd < 0) { set_error(std::string(path) + ": cannot open"); return G16_E_ARG; }
      struct stat sb;
      if (fstat(fd, &sb) != 0 || sb.st_size <= 0) { close(fd); set_error(std::string(path) + ": Invalid File format"); return G16_E_FORMAT; }
      len = (size_t)sb.st_size;
      p = mmap(nullptr, len, PROT_READ, MAP_PRIVATE, fd, 0);
      close(fd);
      if (p == MAP_FAILED) { set_error(std::string(path) + ": cannot map"); return G16_E_STATE; }
      return G16_OK;
    }
    ~Map() { if (p != MAP_FAILED) munmap(p, len); }
  } r1cs, ptau;
  int rc = r1cs.open_ro(r1cs_path);
  if (!rc) rc = ptau.open_ro(ptau_path);
  if (rc) return rc;
  uint8_t* z = nullptr;
  size_t zl = 0;
  rc = g16_plonk_setup_ptau((const uint8_t*)r1cs.p, r1cs.len, (const uint8_t*)ptau.p, ptau.len, device, with_lagrange, &z, &zl);
  if (rc) return rc;
  FILE* f = fopen(zkey_path, "wb");
  if (!f) { free(z); set_error(std::string(zkey_path) + ": cannot create"); return G16_E_ARG; }
  size_t off = 0;
  while (off < zl) {
    const size_t chunk = zl - off < ((size_t)1 << 28) ? zl - off : ((size_t)1 << 28);
    if (fwrite(z + off, 1, chunk, f) != chunk) { fclose(f); free(z); set_error(std::string(zkey_path) + ": write failed"); return G16_E_STATE; }
    off += chunk;
  }
  free(z);
  if (fclose(f) != 0) { set_error(std::string(zkey_path) + ": write failed"); return G16_E_STATE; }
  return G16_OK;
}

extern "C" int g16_setup_device(int device) {
  if (device < -1) { set_error("setup: bad device ordinal"); return G16_E_ARG; }
  g_setup_device.store(device);
  return G16_OK;
}
