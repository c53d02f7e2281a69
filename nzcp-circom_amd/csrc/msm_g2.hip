// G2 (Fq2 coordinates) instantiation of the MSM kernels.
#include <string.h>

#include "msm.cuh"

namespace g16 {
int msm_run_g2(const MsmInstance& m, MsmWorkspace* ws, const Fr* d_scalars, uint8_t* out, hipStream_t st) {
  return msm_run_t<Fq2Ops>(m, ws, d_scalars, out, st);
}
}  // namespace g16
