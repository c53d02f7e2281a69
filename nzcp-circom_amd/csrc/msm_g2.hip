// G2 (Fq2 coordinates) instantiation of the MSM kernels.
#include <string.h>

#include "msm.cuh"

namespace g16 {
int msm_launch_g2(const MsmInstance& m, MsmWorkspace* ws, const Fr* d_scalars, hipStream_t st) {
  return msm_launch_t<Fq2Ops>(m, ws, d_scalars, st);
}
}  // namespace g16
