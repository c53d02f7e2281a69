// Cost floor of the MSM inner loop on gfx950: a resident-data loop of x29_madd (G1) per lane, 4 waves/SIMD,
// no work queue, no sort indirection -- compare with msm_accumulate_kernel's time per sorted entry.
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Inzcp-circom_amd/csrc -Iinclude -o madd_bench tools/madd_bench.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

#include "ec29.cuh"
using namespace g16;

#define ITERS 256

template <int MODE, int OCC = 4>
__global__ __launch_bounds__(64, OCC) void k_loop(const Affine<Fq29Ops>* __restrict__ tab, uint32_t mask,
                                                XYZZ<Fq29Ops>* __restrict__ out) {
  const uint32_t t = blockIdx.x * 64 + threadIdx.x;
  XYZZ<Fq29Ops> acc;
  const Affine<Fq29Ops> p0 = tab[t & mask];
  acc.x = p0.x; acc.y = p0.y; acc.zz = f29_one(); acc.zzz = f29_one();
  uint32_t idx = t * 2654435761u;
  for (int it = 0; it < ITERS; it++) {
    idx = idx * 1664525u + 1013904223u;
    const Affine<Fq29Ops> p = tab[(idx >> 8) & mask];
    if (MODE == 0) {
      x29_madd(acc, p);
    } else if (MODE >= 2 && MODE <= 6) {   // the addition formula with selected exceptional-case checks
      using F = Fq29Ops;
      bool skip = false;
      if (MODE == 3 || MODE == 5) {
        if (x29_is_inf(acc)) { acc.x = p.x; acc.y = p.y; acc.zz = F::one(); acc.zzz = F::one(); skip = true; }
      }
      if (!skip) {
        const F29 U2 = F::mul(p.x, acc.zz);
        const F29 S2 = F::mul(p.y, acc.zzz);
        const F29 P = F::sub<6>(U2, acc.x);
        const F29 R = F::sub<4>(S2, acc.y);
        bool exc = false;
        if (MODE == 4 || MODE == 5) {
          if (x29_diff_is_zero<F>(P)) {
            if (F::is_zero(R)) x29_dbl_affine(acc, p);
            else x29_set_inf(acc);
            exc = true;
          }
        }
        if (MODE == 6) {   // filter only, exceptional lanes just flagged
          if (F::maybe_zero<7>(P)) acc.zz.pad_ = 1;
        }
        if (!exc) {
          const F29 PP = F::sqr(P);
          const F29 PPP = F::mul(P, PP);
          const F29 Qv = F::mul(acc.x, PP);
          const F29 X3 = F::sub<4>(F::sqr(R), F::add(PPP, F::add(Qv, Qv)));
          acc.y = F::sub<2>(F::mul(R, F::sub<6>(Qv, X3)), F::mul(acc.y, PPP));
          acc.x = X3;
          acc.zz = F::mul(acc.zz, PP);
          acc.zzz = F::mul(acc.zzz, PPP);
        }
      }
    } else if (MODE == 1) {   // the ten field products of a mixed addition, no subtractions / checks
      F29 a = Fq29Ops::mul(p.x, acc.zz), b = Fq29Ops::mul(p.y, acc.zzz);
      F29 c = Fq29Ops::sqr(a), d = Fq29Ops::mul(a, c), e = Fq29Ops::mul(acc.x, c), f = Fq29Ops::sqr(b);
      acc.y = Fq29Ops::mul(b, e); acc.x = Fq29Ops::mul(f, d); acc.zz = Fq29Ops::mul(acc.zz, c); acc.zzz = Fq29Ops::mul(acc.zzz, d);
    } else if (MODE >= 7) {   // ten products + 208 injected simple VALU ops of one kind
      F29 a = Fq29Ops::mul(p.x, acc.zz), b = Fq29Ops::mul(p.y, acc.zzz);
      uint32_t d0 = idx, d1 = idx + 1, d2 = idx + 2, d3 = idx + 3;
      const uint32_t sc = __builtin_amdgcn_readfirstlane(idx) | 1u;   // an SGPR operand
#pragma unroll
      for (int k = 0; k < 52; k++) {
        if (MODE == 7) {
          asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4"
                       : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(idx));
        } else if (MODE == 8) {
          asm volatile("v_and_b32 %0, 0x1fffffff, %0\n v_and_b32 %1, 0x1fffffff, %1\n v_and_b32 %2, 0x1fffffff, %2\n v_and_b32 %3, 0x1fffffff, %3"
                       : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));
        } else if (MODE == 9) {
          asm volatile("v_lshrrev_b32 %0, 29, %0\n v_lshrrev_b32 %1, 29, %1\n v_lshrrev_b32 %2, 29, %2\n v_lshrrev_b32 %3, 29, %3"
                       : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));
        } else if (MODE == 10) {
          asm volatile("v_add3_u32 %0, %0, %1, %4\n v_add3_u32 %1, %1, %2, %4\n v_add3_u32 %2, %2, %3, %4\n v_add3_u32 %3, %3, %0, %4"
                       : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "s"(sc));
        } else if (MODE == 11) {
          asm volatile("v_sub_u32 %0, %0, %1\n v_sub_u32 %1, %1, %2\n v_sub_u32 %2, %2, %3\n v_sub_u32 %3, %3, %0"
                       : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));
        } else if (MODE == 12) {   // the serial carry ripple as compiled: add3 -> and / lshr -> add3 ...
          asm volatile("v_add3_u32 %0, %1, %0, %2\n v_and_b32 %1, 0x1fffffff, %0\n v_lshrrev_b32 %0, 29, %0\n v_sub_u32 %1, %1, %0"
                       : "+v"(d0), "+v"(d1) : "s"(sc));
        }
      }
      F29 c = Fq29Ops::sqr(a), d = Fq29Ops::mul(a, c), e = Fq29Ops::mul(acc.x, c), f = Fq29Ops::sqr(b);
      acc.y = Fq29Ops::mul(b, e); acc.x = Fq29Ops::mul(f, d); acc.zz = Fq29Ops::mul(acc.zz, c); acc.zzz = Fq29Ops::mul(acc.zzz, d);
      acc.zzz.pad_ = d0 ^ d1 ^ d2 ^ d3;
    }
  }
  out[t] = acc;
}

template <int MODE, int OCC = 4> static void run(const char* name, const Affine<Fq29Ops>* tab, uint32_t mask, XYZZ<Fq29Ops>* out, double ghz) {
  const int waves = 256 * 4 * OCC;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k_loop<MODE, OCC><<<waves, 64>>>(tab, mask, out);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k_loop<MODE, OCC><<<waves, 64>>>(tab, mask, out);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double per = ms * 1e-3 * ghz * 1e9 / ((double)OCC * ITERS);   // cycles per wave-iteration per SIMD
  printf("%-48s table %8u pts: %.3f ms, %.0f cycles per wave-iteration per SIMD, %.2f G lane-iterations/s\n", name,
         mask + 1, ms, per, waves * 64.0 * ITERS / ms / 1e6);
}

int main() {
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const double ghz = prop.clockRate / 1e6;
  printf("device %s, clock %.2f GHz\n", prop.gcnArchName, ghz);
  const uint32_t sizes[1] = {1u << 12};
  for (uint32_t n : sizes) {
    std::vector<Affine<Fq29Ops>> h(n);
    uint64_t s = 88172645463325252ull;
    for (auto& p : h)
      for (F29* f : {&p.x, &p.y}) {
        for (int i = 0; i < 9; i++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; f->l[i] = (uint32_t)s & (i == 8 ? 0x1fffffu : 0x1fffffffu); }
        f->pad_ = 0;
      }
    Affine<Fq29Ops>* tab; XYZZ<Fq29Ops>* out;
    hipMalloc(&tab, sizeof(h[0]) * n);
    hipMalloc(&out, sizeof(XYZZ<Fq29Ops>) * 256 * 4 * 8 * 64);
    hipMemcpy(tab, h.data(), sizeof(h[0]) * n, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; rep++) {
      printf("-- pass %d\n", rep);
      run<0>("x29_madd loop (gather + mixed add)", tab, n - 1, out, ghz);
      run<1>("ten field products only (8 mul + 2 sqr)", tab, n - 1, out, ghz);
      run<2>("addition formula, no checks", tab, n - 1, out, ghz);
      run<3>("  + accumulator-is-infinity check", tab, n - 1, out, ghz);
      run<4>("  + P == 0 check (doubling / cancellation path)", tab, n - 1, out, ghz);
      run<5>("  + both checks", tab, n - 1, out, ghz);
      run<6>("  + low-limb filter only", tab, n - 1, out, ghz);
      run<1, 5>("ten products, 5 waves/SIMD", tab, n - 1, out, ghz);
      run<1, 6>("ten products, 6 waves/SIMD (VGPR cap 80)", tab, n - 1, out, ghz);
      run<2, 5>("formula no checks, 5 waves/SIMD", tab, n - 1, out, ghz);
      run<2, 6>("formula no checks, 6 waves/SIMD (VGPR cap 80)", tab, n - 1, out, ghz);
      run<5, 5>("formula both checks, 5 waves/SIMD", tab, n - 1, out, ghz);
      run<1, 3>("ten products, 3 waves/SIMD", tab, n - 1, out, ghz);
      run<2, 3>("formula no checks, 3 waves/SIMD", tab, n - 1, out, ghz);
      run<1, 2>("ten products, 2 waves/SIMD", tab, n - 1, out, ghz);
      run<2, 2>("formula no checks, 2 waves/SIMD", tab, n - 1, out, ghz);
      run<7>("ten products + 208 v_add_u32", tab, n - 1, out, ghz);
      run<8>("ten products + 208 v_and_b32 literal", tab, n - 1, out, ghz);
      run<9>("ten products + 208 v_lshrrev_b32", tab, n - 1, out, ghz);
      run<10>("ten products + 208 v_add3_u32 sgpr", tab, n - 1, out, ghz);
      run<11>("ten products + 208 v_sub_u32", tab, n - 1, out, ghz);
      run<12>("ten products + 208 ops in carry-ripple pattern", tab, n - 1, out, ghz);
    }
    hipFree(tab); hipFree(out);
  }
  return 0;
}
