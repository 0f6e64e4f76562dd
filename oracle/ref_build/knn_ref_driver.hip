// knn_ref_driver.hip -- TEST INFRASTRUCTURE.  C entry point around the REFERENCE's own SimpleKNN::knn
// (gaussiansplatting/submodules/simple-knn/simple_knn.cu:185-221), whose unmodified source is compiled for gfx950 by
// the Makefile next to this file: device pointers in, device pointer out.
#include <hip/hip_runtime.h>

#include "simple_knn.h"

extern "C" int gsrref_knn(int P, const float* points_dev, float* mean_dists_dev) {
  try {
    SimpleKNN::knn(P, reinterpret_cast<float3*>(const_cast<float*>(points_dev)), mean_dists_dev);
  } catch (...) {
    return -1;
  }
  return hipDeviceSynchronize() == hipSuccess ? 0 : -2;
}
