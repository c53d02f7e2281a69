// Calibration of rocprofv3's FETCH_SIZE for THIS repo's dominant access pattern (MI355X_MICROARCH.md, HBM section:
// "Other access widths are uncalibrated: calibrate on a known byte count in your own access pattern"): the bucket
// accumulate gathers one aligned 64-byte record per lane (4 x global_load_dwordx4) at a data-dependent index.
//   hipcc --offload-arch=gfx950 -O3 tools/fetch_calib.hip -o nzcp-circom_amd/lib/fetch_calib
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE -d out -o run --output-format csv -- ./fetch_calib
// Three kernels over a 2 GiB table (8x the Infinity Cache), each reading exactly 2^24 records = 1 GiB:
//   gather64_random   : record index = a bijective hash of the lane id (every record at most once)
//   gather64_sorted   : record index = lane id (the same 64-byte loads, consecutive)
//   stream16          : 16 B per lane, consecutive (the guide's calibrated case: FETCH_SIZE reports half)
// Expected bytes per kernel: 1 073 741 824.  tools/profile_round.sh prints FETCH_SIZE x 1024 next to it.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

struct alignas(16) Rec { uint4 q[4]; };

__global__ __launch_bounds__(256) void gather64_random(const Rec* __restrict__ tab, uint32_t mask, uint32_t* __restrict__ out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t idx = (i * 2654435761u) & mask;          // odd multiplier: a bijection modulo 2^k
  const Rec r = tab[idx];
  out[i] = r.q[0].x ^ r.q[1].y ^ r.q[2].z ^ r.q[3].w;
}
__global__ __launch_bounds__(256) void gather64_sorted(const Rec* __restrict__ tab, uint32_t mask, uint32_t* __restrict__ out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const Rec r = tab[i & mask];
  out[i] = r.q[0].x ^ r.q[1].y ^ r.q[2].z ^ r.q[3].w;
}
__global__ __launch_bounds__(256) void stream16(const uint4* __restrict__ tab, uint32_t* __restrict__ out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t acc = 0;
  for (int k = 0; k < 4; k++) {
    const uint4 v = tab[(size_t)k * (1u << 24) + i];
    acc ^= v.x ^ v.w;
  }
  out[i] = acc;
}

int main() {
  const size_t nrec = (size_t)1 << 25;      // 2 GiB of 64-byte records
  const uint32_t n = 1u << 24;              // records read per kernel
  Rec* tab = nullptr;
  uint32_t* out = nullptr;
  if (hipMalloc(&tab, nrec * sizeof(Rec)) != hipSuccess || hipMalloc(&out, (size_t)n * 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
  (void)hipMemset(tab, 0x5a, nrec * sizeof(Rec));
  (void)hipDeviceSynchronize();
  for (int rep = 0; rep < 2; rep++) {
    gather64_random<<<n / 256, 256>>>(tab, (uint32_t)(nrec - 1), out);
    gather64_sorted<<<n / 256, 256>>>(tab + ((size_t)rep << 24), (uint32_t)(n - 1), out);
    stream16<<<n / 256, 256>>>((const uint4*)tab + ((size_t)rep << 26), out);
  }
  if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
  printf("expected bytes per kernel: %llu\n", (unsigned long long)n * 64ull);
  (void)hipFree(tab);
  (void)hipFree(out);
  return 0;
}
