// oracle/ref_hip/knn_driver.cpp -- TEST INFRASTRUCTURE ONLY.  C-ABI around the REAL reference simple-knn
// (SimpleKNN::knn, /root/reference/submodules/simple-knn/simple_knn.h:15-19, compiled in place for gfx950 by
// oracle/ref_hip/Makefile), playing the part of its torch binding distCUDA2 (spatial.cu:15-26): device pointers in,
// P floats out (mean squared distance to the 3 nearest neighbours).  Used by tests/golden/make_golden_ref_hip.py to
// produce the fixture that pins oracle/wg_knn_oracle.c.
#include <hip/hip_runtime.h>
#include "simple_knn.h"

extern "C" void refhip_knn(int P, const float* points, float* mean_dists) {
    hipMemset(mean_dists, 0, (size_t)P * sizeof(float));  // torch::full(0) of the binding
    if (P == 0) return;
    SimpleKNN::knn(P, reinterpret_cast<float3*>(const_cast<float*>(points)), mean_dists);
}
