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

// General coordinate dimension C <= CDM (components >= C of a, b must be 0).  Summation order = what the reference's
// `(rel_coors ** 2).sum(-1)` does on the CPU (measured, torch 2.10 fp32): left to
// right for C in {1, 2, 3, 4, 8}; s0, s4, ..., s_{C-1}, s1, s2, s3 for C in {5, 6, 7}.  No FMA contraction.
template <int CDM>
__device__ __forceinline__ float egnn_sqdist_n(const float (&a)[CDM], const float (&b)[CDM], int C, float (&rel)[CDM]) {
#pragma clang fp contract(off)
    float sq[CDM];
#pragma unroll
    for (int c = 0; c < CDM; ++c) {
        rel[c] = a[c] - b[c];
        sq[c] = rel[c] * rel[c];
    }
    float d = sq[0];
    if (C >= 5 && C <= 7) {
#pragma unroll
        for (int c = 4; c < CDM; ++c)
            if (c < C) d = d + sq[c];
#pragma unroll
        for (int c = 1; c < 4; ++c) d = d + sq[c];
    } else {
#pragma unroll
        for (int c = 1; c < CDM; ++c)
            if (c < C) d = d + sq[c];
    }
    return d;
}

// Packed ("tile-major") layout of the fp16 GEMM operands: an (R x Kp) matrix, R padded to 32 rows, Kp % 32 == 0, is
// stored as [R/32][Kp/16][32 rows][2 chunks][8 halves]; the chunk index is XOR-swizzled by ((row >> 3) & 1) so that the
// 1 KB (row block, K-tile) piece is exactly the bank-conflict-free LDS image the GEMM wants.  nkt = Kp / 16.
__host__ __device__ __forceinline__ size_t egnn_pk_off(int64_t row, int k, int nkt) {
    const int r = (int)(row & 31);
    const int64_t rb = row >> 5;
    const int kt = k >> 4, ck = (k >> 3) & 1, e = k & 7;
    return (size_t)((((rb * nkt + kt) * 32 + r) * 2 + (ck ^ ((r >> 3) & 1))) * 8 + e);
}

// internal (node_ops.hip): producer of the packed (hi, lo) layout, shared by egnn_node_prep_hl and egnn_split_f16
int egnn_pack_rows_launch(const float* X, int64_t ldx, const float* m_i, const float* gamma, const float* beta, float eps,
                          void* hi, void* lo, int Kp, void* raw_hi, void* raw_lo, int raw_Kp, int64_t rows, int dim, int m_dim,
                          void* stream);

static inline int egnn_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? EGNN_OK : (int)e;
}
