// The Keccak-256 transcript of snarkjs's PLONK (plonk_prove.js / plonk_verify.js `hashToFr`: the digest read as a
// big-endian integer, reduced modulo r) and its big-endian encoders -- shared by the prover (plonk.hip) and the
// verifier (verify_plonk.hip).  Host code only.
#pragma once
#include <stdint.h>
#include <string.h>

#include <vector>

#include "ec.cuh"

namespace g16 {
namespace transcript {

using FrM = Fr;   // Montgomery residue

inline bool h_lt_r(const uint32_t v[8]) {
  static const uint32_t kR[8] = G16_FR_P;
  for (int l = 7; l >= 0; l--)
    if (v[l] != kR[l]) return v[l] < kR[l];
  return false;
}

// ------------------------------------------------------------------ Keccak-256 (original 0x01 padding)
inline void keccak_f(uint64_t s[25]) {
  static const uint64_t RC[24] = {
      0x0000000000000001ull, 0x0000000000008082ull, 0x800000000000808Aull, 0x8000000080008000ull, 0x000000000000808Bull,
      0x0000000080000001ull, 0x8000000080008081ull, 0x8000000000008009ull, 0x000000000000008Aull, 0x0000000000000088ull,
      0x0000000080008009ull, 0x000000008000000Aull, 0x000000008000808Bull, 0x800000000000008Bull, 0x8000000000008089ull,
      0x8000000000008003ull, 0x8000000000008002ull, 0x8000000000000080ull, 0x000000000000800Aull, 0x800000008000000Aull,
      0x8000000080008081ull, 0x8000000000008080ull, 0x0000000080000001ull, 0x8000000080008008ull};
  static const int ROT[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};
  auto rol = [](uint64_t x, int n) { return n ? (x << n) | (x >> (64 - n)) : x; };
  for (int r = 0; r < 24; r++) {
    uint64_t c[5], d[5], b[25];
    for (int x = 0; x < 5; x++) c[x] = s[x] ^ s[x + 5] ^ s[x + 10] ^ s[x + 15] ^ s[x + 20];
    for (int x = 0; x < 5; x++) d[x] = c[(x + 4) % 5] ^ rol(c[(x + 1) % 5], 1);
    for (int i = 0; i < 25; i++) s[i] ^= d[i % 5];
    for (int x = 0; x < 5; x++)
      for (int y = 0; y < 5; y++) b[y + 5 * ((2 * x + 3 * y) % 5)] = rol(s[x + 5 * y], ROT[x + 5 * y]);
    for (int x = 0; x < 5; x++)
      for (int y = 0; y < 5; y++) s[x + 5 * y] = b[x + 5 * y] ^ ((~b[(x + 1) % 5 + 5 * y]) & b[(x + 2) % 5 + 5 * y]);
    s[0] ^= RC[r];
  }
}
inline void keccak256(const uint8_t* data, size_t len, uint8_t out[32]) {
  const size_t rate = 136;
  std::vector<uint8_t> m(data, data + len);
  m.push_back(0x01);
  while (m.size() % rate) m.push_back(0);
  m.back() |= 0x80;
  uint64_t s[25] = {0};
  for (size_t off = 0; off < m.size(); off += rate) {
    for (size_t i = 0; i < rate / 8; i++) {
      uint64_t v = 0;
      for (int k = 7; k >= 0; k--) v = (v << 8) | m[off + 8 * i + k];
      s[i] ^= v;
    }
    keccak_f(s);
  }
  for (int i = 0; i < 4; i++)
    for (int k = 0; k < 8; k++) out[8 * i + k] = (uint8_t)(s[i] >> (8 * k));
}
// hashToFr: the digest as a big-endian integer, reduced modulo r -> Montgomery
inline FrM hash_to_fr(const std::vector<uint8_t>& t) {
  uint8_t h[32];
  keccak256(t.data(), t.size(), h);
  Fr x;
  for (int l = 0; l < 8; l++) {
    uint32_t w = 0;
    for (int k = 0; k < 4; k++) w = (w << 8) | h[32 - 4 * (l + 1) + k];
    x.v[l] = w;
  }
  static const uint32_t kR[8] = G16_FR_P;
  while (!h_lt_r(x.v)) {   // 2^256 / r < 6: a few subtractions
    int64_t br = 0;
    for (int i = 0; i < 8; i++) {
      br += (int64_t)x.v[i] - (int64_t)kR[i];
      x.v[i] = (uint32_t)br;
      br >>= 32;
    }
  }
  return fp_to_mont(x);
}
inline void put_be(std::vector<uint8_t>& t, const uint32_t v[8]) {   // 32-byte big-endian image of a standard-form integer
  for (int l = 7; l >= 0; l--)
    for (int k = 3; k >= 0; k--) t.push_back((uint8_t)(v[l] >> (8 * k)));
}
inline void put_fr_be(std::vector<uint8_t>& t, const FrM& x) {
  const Fr s = fp_from_mont(x);
  put_be(t, s.v);
}
inline void put_g1_be(std::vector<uint8_t>& t, const G1Affine& p) {   // Montgomery affine -> G1.toRprUncompressed
  const Fq x = fp_from_mont(p.x), y = fp_from_mont(p.y);
  put_be(t, x.v);
  put_be(t, y.v);
}


}  // namespace transcript
}  // namespace g16
