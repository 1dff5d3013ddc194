// Activation shared by the GroupNorm kernels and the GEMM kernels that apply a fused GroupNorm to their input.
#pragma once
#include <hip/hip_runtime.h>

namespace gd {

// nn.swish: x * sigmoid(x) (cld_jax/models/layers.py:40) on the hardware transcendentals: v_exp_f32 (2^x) and v_rcp_f32, 1 ulp each
// -- five VALU instructions.  (__expf / __frcp_rn expand to a range-checked exponential and a correctly rounded IEEE division,
// about four times the instructions, which shows up where this runs inside a GEMM's staging path.)
__device__ __forceinline__ float swish_e(float x) {
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}

}  // namespace gd
