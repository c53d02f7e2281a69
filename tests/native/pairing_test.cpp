// Host build of csrc/pairing.cuh (the arithmetic the verifier kernels run) for the CPU test-suite:
// reads "px py qx0 qx1 qy0 qy1" (hex, standard form) per line, prints the 12 Fq coordinates (hex, standard form) of
// final_exponentiation(miller_loop(P, Q)) in the order of the W^i coefficients a_i = (a, b), i = 0..5 -- once with the
// line coefficients computed on the fly and once from g2_precompute (must agree) -- tests/test_cpu_pairing.py compares
// them with oracle/bn254.py.
#include <stdio.h>
#include <string.h>

#include <string>
#include <vector>

#include "pairing.cuh"

using namespace g16;

static Fq parse_fq(const char* hex) {
  Fq r = fp_zero<FqParams>();
  std::string s(hex);
  while (s.size() < 64) s = "0" + s;
  for (int i = 0; i < 8; i++) r.v[i] = (uint32_t)strtoul(s.substr(64 - 8 * (i + 1), 8).c_str(), nullptr, 16);
  return fp_to_mont(r);
}
static void print_fq(const Fq& m) {
  const Fq s = fp_from_mont(m);
  for (int i = 7; i >= 0; i--) printf("%08x", s.v[i]);
}
static void print_f12(const Fq12& f) {
  const Fq2* w[6] = {&f.c0.c0, &f.c1.c0, &f.c0.c1, &f.c1.c1, &f.c0.c2, &f.c1.c2};
  for (int i = 0; i < 6; i++) {
    print_fq(w[i]->a); printf(" ");
    print_fq(w[i]->b); printf(i == 5 ? "\n" : " ");
  }
}

int main() {
  PairingConsts pc;
  pairing_consts_init(pc);
  char a[6][80];
  while (scanf("%79s %79s %79s %79s %79s %79s", a[0], a[1], a[2], a[3], a[4], a[5]) == 6) {
    Affine<FqOps> p{parse_fq(a[0]), parse_fq(a[1])};
    Affine<Fq2Ops> q{Fq2{parse_fq(a[2]), parse_fq(a[3])}, Fq2{parse_fq(a[4]), parse_fq(a[5])}};
    if (!g1_on_curve(p) || !g2_on_curve(q, pc)) { printf("offcurve\n"); continue; }
    const Fq12 e1 = final_exponentiation(miller_loop(p, q, pc), pc);
    std::vector<EllCoeffs> co(kEllSteps);
    g2_precompute(q, pc, co.data());
    const Fq12 e2 = final_exponentiation(miller_loop_pre(p, co.data()), pc);
    if (!f12_eq(e1, e2)) { printf("mismatch\n"); continue; }
    print_f12(e1);
  }
  return 0;
}
