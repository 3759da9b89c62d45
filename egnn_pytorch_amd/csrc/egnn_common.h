// Shared device helpers for the gfx950 EGNN kernels.  wave = 64 lanes everywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/egnn_hip.h"

#define EGNN_WAVE 64

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// SiLU(x) = x * sigmoid(x) (reference: nn.SiLU, egnn_pytorch.py:56-60).
// v_exp_f32 + v_rcp_f32; |error| ~1e-7 relative, far inside the 1e-4 parity budget.
__device__ __forceinline__ float egnn_silu(float x) {
    return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x));
}
__device__ __forceinline__ float egnn_sigmoid(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __expf(-x));
}

// squared distance exactly as the reference's CPU path produces it:
// ((dx*dx + dy*dy) + dz*dz), each operation rounded separately -- no FMA contraction.
__device__ __forceinline__ float egnn_sqdist(float xi, float yi, float zi, float xj, float yj, float zj,
                                             float& dx, float& dy, float& dz) {
    // plain operators under an explicit contract(off): hipcc's default -ffp-contract=fast would fuse
    // (and re-associate) these into v_fmac_f32; the __fmul_rn/__fadd_rn wrappers do NOT prevent that.
#pragma clang fp contract(off)
    dx = xi - xj;
    dy = yi - yj;
    dz = zi - zj;
    const float sx = dx * dx;
    const float sy = dy * dy;
    const float sz = dz * dz;
    const float sxy = sx + sy;
    return sxy + sz;
}

static inline int egnn_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? EGNN_OK : (int)e;
}
