// Activation shared by the GroupNorm kernels and the GEMM kernels that apply a fused GroupNorm to their input.
#pragma once
#include <hip/hip_runtime.h>

namespace gd {

// nn.swish: x * sigmoid(x) (cld_jax/models/layers.py:40)
__device__ __forceinline__ float swish_e(float x) { return x * __frcp_rn(1.0f + __expf(-x)); }

}  // namespace gd
