// Integer / fp64 issue-rate microbenchmarks for gfx950.
// The MI355X guides list no integer-multiply rates, and the MSM/NTT kernels are bound by them, so
// this measures, per instruction, cycles per wave-instruction per SIMD (lower = faster) at 1..8
// waves/SIMD, with 8 independent dependency chains per lane.
//   build: hipcc --offload-arch=gfx950 -O3 -o microbench tools/microbench.hip ; run: ./microbench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

#define CHAINS 8
#define ITERS 2048

enum { OP_MAD_DISTINCT, OP_MAD_DISTINCT_2CH, OP_LSHR_B64, OP_F29_MUL_LIKE, OP_MAD_U64_U32, OP_MUL_LO, OP_MUL_HI, OP_MAD_U32_U24, OP_MUL_HI_U24, OP_ADD_CO, OP_LSHL_ADD_U64, OP_FMA_F64,
       OP_ADD_U32, OP_MAD_U64_U32_C, OP_COUNT };
static const char* kNames[] = {"v_mad_u64_u32 distinct srcs, 8 chains", "v_mad_u64_u32 distinct srcs, 2 chains", "v_lshrrev_b64", "81-mad column product (C, 2 chains)", "v_mad_u64_u32 (asm)", "v_mul_lo_u32", "v_mul_hi_u32", "v_mad_u32_u24", "v_mul_hi_u32_u24",
                               "v_add_co_u32+v_addc_co_u32 (pair)", "v_lshl_add_u64", "v_fma_f64", "v_add_u32",
                               "(u64)a*b+c in C"};

template <int OP>
__global__ __launch_bounds__(256) void bench(uint64_t* out, uint32_t a0, uint32_t b0) {
  uint32_t a = a0 + threadIdx.x, b = b0 ^ threadIdx.x;
  uint64_t acc[CHAINS];
  uint32_t x[CHAINS];
  double d[CHAINS];
#pragma unroll
  for (int k = 0; k < CHAINS; k++) { acc[k] = a * (k + 1); x[k] = b + k; d[k] = 1.0 + k * 1e-3; }
  const double da = 1.0000001, db = 1e-9;
  for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int k = 0; k < CHAINS; k++) {
      if (OP == OP_MAD_DISTINCT) {
        uint64_t cy;
        asm volatile("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(acc[k]), "=s"(cy) : "v"(x[k]), "v"(x[(k + 3) & 7]));
      } else if (OP == OP_MAD_DISTINCT_2CH) {
        uint64_t cy;
        asm volatile("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(acc[k & 1]), "=s"(cy) : "v"(x[k]), "v"(x[(k + 3) & 7]));
      } else if (OP == OP_LSHR_B64) {
        asm volatile("v_lshrrev_b64 %0, 29, %0" : "+v"(acc[k]));
      } else if (OP == OP_F29_MUL_LIKE) {
        // one "column" per k: 9 products into two accumulators, like fq29.cuh
        uint64_t ab = 0, mp = acc[k];
#pragma unroll
        for (int i = 0; i < 5; i++) ab += (uint64_t)x[i] * x[7 - i];
#pragma unroll
        for (int i = 0; i < 4; i++) mp += (uint64_t)x[(i + k) & 7] * (0x12345u + i);
        acc[k] = (ab + mp) >> 29;
        x[k] = (uint32_t)acc[k] & 0x1fffffffu;
      } else if (OP == OP_MAD_U64_U32) {
        uint64_t cy;
        asm volatile("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(acc[k]), "=s"(cy) : "v"(a), "v"(b));
      } else if (OP == OP_MUL_LO) {
        asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x[k]) : "v"(a));
      } else if (OP == OP_MUL_HI) {
        asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(x[k]) : "v"(a));
      } else if (OP == OP_MAD_U32_U24) {
        asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(x[k]) : "v"(a), "v"(b));
      } else if (OP == OP_MUL_HI_U24) {
        asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(x[k]) : "v"(a));
      } else if (OP == OP_ADD_CO) {
        uint32_t lo = (uint32_t)acc[k], hi = (uint32_t)(acc[k] >> 32);
        asm volatile("v_add_co_u32 %0, vcc, %0, %2\n\tv_addc_co_u32 %1, vcc, %1, %3, vcc" : "+v"(lo), "+v"(hi) : "v"(a), "v"(b) : "vcc");
        acc[k] = ((uint64_t)hi << 32) | lo;
      } else if (OP == OP_LSHL_ADD_U64) {
        uint64_t o = ((uint64_t)a << 32) | b;
        asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(acc[k]) : "v"(o));
      } else if (OP == OP_FMA_F64) {
        asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[k]) : "v"(da), "v"(db));
      } else if (OP == OP_ADD_U32) {
        asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[k]) : "v"(a));
      } else if (OP == OP_MAD_U64_U32_C) {
        acc[k] = (uint64_t)a * (uint32_t)(acc[k] >> 7) + acc[k];
      }
    }
  }
  uint64_t r = 0;
#pragma unroll
  for (int k = 0; k < CHAINS; k++) r += acc[k] + x[k] + (uint64_t)d[k];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int OP> static void run_one(uint64_t* dout, double clock_ghz) {
  printf("%-36s", kNames[OP]);
  for (int wps = 1; wps <= 8; wps *= 2) {   // waves per SIMD: blocks of 256 threads = 4 waves = 1 per SIMD
    const int blocks = 256 * wps;           // one block per CU per wave-per-SIMD step
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    bench<OP><<<blocks, 256>>>(dout, 12345u, 678u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    bench<OP><<<blocks, 256>>>(dout, 12345u, 678u);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    // wave-instructions issued per SIMD = wps * ITERS * CHAINS (x2 for the add pair)
    const double insts = (double)wps * ITERS * CHAINS;
    const double cycles = ms * 1e-3 * clock_ghz * 1e9;
    printf("  %d w/SIMD: %6.2f cyc/inst", wps, cycles / insts);
    hipEventDestroy(e0); hipEventDestroy(e1);
  }
  printf("\n");
}

int main() {
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const double ghz = prop.clockRate / 1e6;
  printf("device %s, %d CUs, clock %.2f GHz (cycles below assume this clock)\n", prop.gcnArchName, prop.multiProcessorCount, ghz);
  uint64_t* dout;
  hipMalloc(&dout, sizeof(uint64_t) * 256 * 256 * 8);
  run_one<OP_MAD_DISTINCT>(dout, ghz);
  run_one<OP_MAD_DISTINCT_2CH>(dout, ghz);
  run_one<OP_LSHR_B64>(dout, ghz);
  run_one<OP_F29_MUL_LIKE>(dout, ghz);
  run_one<OP_MAD_U64_U32>(dout, ghz);
  run_one<OP_MAD_U64_U32_C>(dout, ghz);
  run_one<OP_MUL_LO>(dout, ghz);
  run_one<OP_MUL_HI>(dout, ghz);
  run_one<OP_MAD_U32_U24>(dout, ghz);
  run_one<OP_MUL_HI_U24>(dout, ghz);
  run_one<OP_ADD_CO>(dout, ghz);
  run_one<OP_LSHL_ADD_U64>(dout, ghz);
  run_one<OP_ADD_U32>(dout, ghz);
  run_one<OP_FMA_F64>(dout, ghz);
  hipFree(dout);
  return 0;
}
