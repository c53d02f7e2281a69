// G2 (Fq2 coordinates) instantiation of the MSM lane kernels (msm.cuh) and of the base-table conversions.
#include <string.h>

#include "msm.cuh"

namespace g16 {
int msm_launch_lane_g2(const MsmGroup& g, MsmWorkspace* ws, MsmLaneWs& ln, hipStream_t st) {
  return msm_launch_lane_t<Fq2x29Ops>(g, ws, ln, g.d_bases2, st);
}
int msm_convert_bases_g2(const void* in, void* out, uint32_t n) {
  msm_convert_bases_kernel<Fq2x29Ops><<<(n + 255) / 256, 256>>>((const G2Affine*)in, (PackedAffine<Fq2x29Ops>*)out, n);
  G16_HIP(hipGetLastError());
  G16_HIP(hipDeviceSynchronize());
  return G16_OK;
}
int msm_precompute_g2(const void* in, void* out, uint32_t n, int ndbl) {
  msm_precompute_kernel<Fq2Ops><<<(n + 255) / 256, 256>>>((const G2Affine*)in, (G2Affine*)out, n, ndbl);
  G16_HIP(hipGetLastError());
  G16_HIP(hipDeviceSynchronize());
  return G16_OK;
}
}  // namespace g16
