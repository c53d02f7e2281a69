// VGPR bank-conflict probe for v_mad_u64_u32 on gfx950: same instruction stream with sources in distinct
// banks (register index mod 4) vs all in one bank.  build: hipcc --offload-arch=gfx950 -O3 -o bank_bench tools/bank_bench.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define REP8(x) x x x x x x x x
template <int MODE>
__global__ __launch_bounds__(64, 4) void k(uint32_t* out, uint32_t a) {
  uint32_t r = a + threadIdx.x;
  for (int it = 0; it < 2048; it++) {
    if (MODE == 0) {   // src0 bank 2, src1 bank 3, src2 banks 0/1
      asm volatile(
          "v_mov_b32 v30, %0\n v_mov_b32 v31, %0\n v_mov_b32 v24, %0\n v_mov_b32 v28, %0\n"
          REP8("v_mad_u64_u32 v[20:21], s[10:11], v30, v31, v[20:21]\n v_mad_u64_u32 v[32:33], s[10:11], v30, v31, v[32:33]\n"
               "v_mad_u64_u32 v[36:37], s[10:11], v30, v31, v[36:37]\n v_mad_u64_u32 v[40:41], s[10:11], v30, v31, v[40:41]\n")
          "v_xor_b32 %0, %0, v20\n"
          : "+v"(r) : : "v20", "v21", "v24", "v28", "v30", "v31", "v32", "v33", "v36", "v37", "v40", "v41", "s10", "s11");
    } else if (MODE == 1) {   // src0, src1, src2.lo all bank 0
      asm volatile(
          "v_mov_b32 v30, %0\n v_mov_b32 v31, %0\n v_mov_b32 v24, %0\n v_mov_b32 v28, %0\n"
          REP8("v_mad_u64_u32 v[20:21], s[10:11], v24, v28, v[20:21]\n v_mad_u64_u32 v[32:33], s[10:11], v24, v28, v[32:33]\n"
               "v_mad_u64_u32 v[36:37], s[10:11], v24, v28, v[36:37]\n v_mad_u64_u32 v[40:41], s[10:11], v24, v28, v[40:41]\n")
          "v_xor_b32 %0, %0, v20\n"
          : "+v"(r) : : "v20", "v21", "v24", "v28", "v30", "v31", "v32", "v33", "v36", "v37", "v40", "v41", "s10", "s11");
    } else if (MODE == 2) {   // one SGPR multiplicand (like m * P[j])
      asm volatile(
          "v_mov_b32 v30, %0\n v_mov_b32 v31, %0\n v_mov_b32 v24, %0\n v_mov_b32 v28, %0\n s_mov_b32 s12, 0x12345\n"
          REP8("v_mad_u64_u32 v[20:21], s[10:11], v30, s12, v[20:21]\n v_mad_u64_u32 v[32:33], s[10:11], v30, s12, v[32:33]\n"
               "v_mad_u64_u32 v[36:37], s[10:11], v30, s12, v[36:37]\n v_mad_u64_u32 v[40:41], s[10:11], v30, s12, v[40:41]\n")
          "v_xor_b32 %0, %0, v20\n"
          : "+v"(r) : : "v20", "v21", "v24", "v28", "v30", "v31", "v32", "v33", "v36", "v37", "v40", "v41", "s10", "s11", "s12");
    } else if (MODE == 3) {   // zero addend (first product of a column)
      asm volatile(
          "v_mov_b32 v30, %0\n v_mov_b32 v31, %0\n v_mov_b32 v24, %0\n v_mov_b32 v28, %0\n"
          REP8("v_mad_u64_u32 v[20:21], s[10:11], v30, v31, 0\n v_mad_u64_u32 v[32:33], s[10:11], v30, v31, 0\n"
               "v_mad_u64_u32 v[36:37], s[10:11], v30, v31, 0\n v_mad_u64_u32 v[40:41], s[10:11], v30, v31, 0\n")
          "v_xor_b32 %0, %0, v20\n"
          : "+v"(r) : : "v20", "v21", "v24", "v28", "v30", "v31", "v32", "v33", "v36", "v37", "v40", "v41", "s10", "s11");
    } else if (MODE == 4) {   // same accumulator back to back (one dependency chain)
      asm volatile(
          "v_mov_b32 v30, %0\n v_mov_b32 v31, %0\n v_mov_b32 v24, %0\n v_mov_b32 v28, %0\n"
          REP8("v_mad_u64_u32 v[20:21], s[10:11], v30, v31, v[20:21]\n v_mad_u64_u32 v[20:21], s[10:11], v30, v31, v[20:21]\n"
               "v_mad_u64_u32 v[20:21], s[10:11], v30, v31, v[20:21]\n v_mad_u64_u32 v[20:21], s[10:11], v30, v31, v[20:21]\n")
          "v_xor_b32 %0, %0, v20\n"
          : "+v"(r) : : "v20", "v21", "v24", "v28", "v30", "v31", "v32", "v33", "v36", "v37", "v40", "v41", "s10", "s11");
    }
  }
  out[blockIdx.x * 64 + threadIdx.x] = r;
}
template <int MODE> static void run(const char* name, uint32_t* out) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int occ = 1; occ <= 4; occ *= 2) {
    k<MODE><<<1024 * occ, 64>>>(out, 3);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<1024 * occ, 64>>>(out, 3);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-52s %d w/SIMD: %.2f cycles per v_mad_u64_u32 per SIMD\n", name, occ, ms * 1e-3 * 2.4e9 / (2048.0 * 32 * occ));
  }
}
int main() {
  uint32_t* out;
  hipMalloc(&out, 4 * 64 * 4096);
  run<0>("sources in banks 2,3 + accumulator 0/1", out);
  run<1>("sources and accumulator.lo all in bank 0", out);
  run<2>("one SGPR multiplicand", out);
  run<3>("constant-zero addend", out);
  run<4>("single dependency chain", out);
  return 0;
}
