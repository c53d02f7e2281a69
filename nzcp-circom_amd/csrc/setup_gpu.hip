// Fixed-base scalar multiplication on the device: out[i] = [k_i] G for the millions of trapdoor scalars of a
// Groth16 setup (".r1cs reader + native Groth16 trapdoor setup at scale on GPU (fixed-base MSM
// kernel)").  What it stands in for: `snarkjs groth16 setup` ([EXT] snarkjs 0.4.12, pin /root/reference/yarn.lock:987-1001;
// the reference records only its PLONK twin and the 2^22-power ptau it used, /root/reference/Makefile:30-31) --
// there the section points come from a powers-of-tau file, here (test-only, known trapdoor) from the scalars
// u_i(tau), v_i(tau), (beta u_i + alpha v_i + w_i)/gamma|delta, L_{2i+1}(tau)/delta that synth.cpp::setup_core
// evaluates on the host.
//
// Two kernels per chunk of points, both on the canonical 8x32-bit Montgomery field (fp.cuh / ec.cuh: exact,
// complete additions -- this is create-time work, the bytes must equal the host path's bytes):
//   setup_fixed_mul_kernel : one lane per scalar; the scalar leaves Montgomery form, its `nwin` wb-bit digits index
//                            the table [nwin][2^wb - 1] of d * 2^(wb j) * G (a few hundred KB: L2 resident) and are
//                            mixed-added into an XYZZ accumulator.
//   setup_to_affine_kernel : one lane per kBatch consecutive points: Montgomery's trick on their ZZZ (3 products per
//                            point + one Fermat inversion per batch), then x = X (ZZ/ZZZ)^2, y = Y / ZZZ.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include "ec.cuh"
#include "internal.h"

namespace g16 {
namespace {

constexpr int kBatch = 8;

template <class FC>
__global__ __launch_bounds__(256) void setup_fixed_mul_kernel(const Affine<FC>* __restrict__ tbl, int wb, int nwin,
                                                               const Fr* __restrict__ ks_mont, uint32_t n,
                                                               XYZZ<FC>* __restrict__ out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const Fr k = fp_from_mont(ks_mont[i]);
  const uint32_t row = (1u << wb) - 1, mask = row;
  XYZZ<FC> acc;
  xyzz_set_inf(acc);
  for (int j = 0; j < nwin; j++) {
    const int pos = j * wb;
    uint64_t v = k.v[pos >> 5];
    if ((pos >> 5) + 1 < 8) v |= (uint64_t)k.v[(pos >> 5) + 1] << 32;
    const uint32_t d = (uint32_t)(v >> (pos & 31)) & mask;
    if (d) {
      const Affine<FC> q = tbl[(size_t)j * row + d - 1];
      xyzz_madd(acc, q);
    }
  }
  out[i] = acc;
}

template <class FC>
__global__ __launch_bounds__(256) void setup_to_affine_kernel(const XYZZ<FC>* __restrict__ in, Affine<FC>* __restrict__ out,
                                                               uint32_t n) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t lo = t * kBatch;
  if (lo >= n) return;
  const uint32_t cnt = n - lo < (uint32_t)kBatch ? n - lo : (uint32_t)kBatch;
  typename FC::T pref[kBatch];
  typename FC::T acc = FC::one();
  for (uint32_t e = 0; e < cnt; e++) {
    pref[e] = acc;
    const typename FC::T zzz = in[lo + e].zzz;
    if (!FC::is_zero(zzz)) acc = FC::mul(acc, zzz);
  }
  typename FC::T inv = FC::inv(acc);
  for (uint32_t e = cnt; e-- > 0;) {
    const XYZZ<FC> p = in[lo + e];
    Affine<FC> a;
    if (xyzz_is_inf(p)) {
      a.x = FC::zero();
      a.y = FC::zero();
    } else {
      const typename FC::T zi = FC::mul(inv, pref[e]);
      inv = FC::mul(inv, p.zzz);
      const typename FC::T zzi = FC::sqr(FC::mul(zi, p.zz));
      a.x = FC::mul(p.x, zzi);
      a.y = FC::mul(p.y, zi);
    }
    out[lo + e] = a;
  }
}

template <class FC>
int fixed_mul_device(int device, const Affine<FC>* h_tbl, int wb, int nwin, const Fr* ks_mont, size_t n, uint8_t* out) {
  if (n == 0) return G16_OK;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    set_error("setup: no HIP device (g16_setup_device selected the GPU path)");
    return G16_E_NOGPU;
  }
  if (device < 0 || device >= ndev) { set_error("setup: bad device ordinal"); return G16_E_ARG; }
  G16_HIP(hipSetDevice(device));
  const size_t tbl_n = (size_t)nwin * (((size_t)1 << wb) - 1);
  const size_t chunk = n < ((size_t)1 << 20) ? n : ((size_t)1 << 20);
  Affine<FC>* d_tbl = nullptr;
  Fr* d_k = nullptr;
  XYZZ<FC>* d_acc = nullptr;
  Affine<FC>* d_aff = nullptr;
  hipStream_t st = nullptr;
  int rc = G16_OK;
  auto fail = [&](hipError_t e) {
    if (e == hipSuccess) return false;
    set_error(std::string("setup (device): ") + hipGetErrorString(e));
    rc = G16_E_HIP;
    return true;
  };
  static const bool trace = getenv("G16_TRACE_HOST") != nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  float kern_ms = 0.f;
  do {
    if (fail(hipStreamCreate(&st))) break;
    if (trace && (fail(hipEventCreate(&e0)) || fail(hipEventCreate(&e1)))) break;
    if (fail(hipMalloc(&d_tbl, tbl_n * sizeof(Affine<FC>)))) break;
    if (fail(hipMalloc(&d_k, chunk * sizeof(Fr)))) break;
    if (fail(hipMalloc(&d_acc, chunk * sizeof(XYZZ<FC>)))) break;
    if (fail(hipMalloc(&d_aff, chunk * sizeof(Affine<FC>)))) break;
    if (fail(hipMemcpyAsync(d_tbl, h_tbl, tbl_n * sizeof(Affine<FC>), hipMemcpyHostToDevice, st))) break;
    for (size_t base = 0; base < n && rc == G16_OK; base += chunk) {
      const uint32_t cnt = (uint32_t)(base + chunk < n ? chunk : n - base);
      if (fail(hipMemcpyAsync(d_k, ks_mont + base, (size_t)cnt * sizeof(Fr), hipMemcpyHostToDevice, st))) break;
      if (trace) (void)hipEventRecord(e0, st);
      setup_fixed_mul_kernel<FC><<<(cnt + 255) / 256, 256, 0, st>>>(d_tbl, wb, nwin, d_k, cnt, d_acc);
      const uint32_t nb = (cnt + kBatch - 1) / kBatch;
      setup_to_affine_kernel<FC><<<(nb + 255) / 256, 256, 0, st>>>(d_acc, d_aff, cnt);
      if (trace) (void)hipEventRecord(e1, st);
      if (fail(hipGetLastError())) break;
      if (fail(hipMemcpyAsync(out + base * sizeof(Affine<FC>), d_aff, (size_t)cnt * sizeof(Affine<FC>),
                              hipMemcpyDeviceToHost, st))) break;
      if (fail(hipStreamSynchronize(st))) break;
      if (trace) {
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        kern_ms += ms;
      }
    }
  } while (false);
  if (trace && rc == G16_OK)
    fprintf(stderr, "[g16 setup] %s fixed-base: %zu scalars, kernels %.3f ms (%.1f M points/s)\n",
            sizeof(Affine<FC>) == 64 ? "G1" : "G2", n, kern_ms, kern_ms > 0 ? n / kern_ms / 1e3 : 0.0);
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  if (st) (void)hipStreamSynchronize(st);
  void* bufs[] = {d_tbl, d_k, d_acc, d_aff};
  for (void* p : bufs) if (p) (void)hipFree(p);
  if (st) (void)hipStreamDestroy(st);
  return rc;
}

}  // namespace

int setup_fixed_mul_g1(int device, const G1Affine* tbl, int wb, int nwin, const Fr* ks_mont, size_t n, uint8_t* out) {
  return fixed_mul_device<FqOps>(device, tbl, wb, nwin, ks_mont, n, out);
}
int setup_fixed_mul_g2(int device, const G2Affine* tbl, int wb, int nwin, const Fr* ks_mont, size_t n, uint8_t* out) {
  return fixed_mul_device<Fq2Ops>(device, tbl, wb, nwin, ks_mont, n, out);
}

}  // namespace g16
