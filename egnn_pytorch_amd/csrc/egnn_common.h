// Shared device helpers for the gfx950 EGNN kernels.  wave = 64 lanes everywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/egnn_hip.h"

#define EGNN_WAVE 64

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---- cross-lane primitives.  All of them are DPP (VALU) operations.  The LDS-pipe shuffles (`__shfl*` =
// ds_bpermute_b32) are NOT used anywhere: with other launches of these kernels co-resident on the CU (LDS-DMA traffic in
// flight) ds_bpermute_b32 occasionally returned another value -- tools/concurrency_check.py; the hip shuffle intrinsics are
// poisoned below so that they cannot come back.
#define EGNN_DPP(v, ctrl, row_mask, bound) \
    __builtin_bit_cast(decltype(v), __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, row_mask, 0xF, bound))

// Sum over the 16 lanes of a DPP row; every lane of the row ends with the same bits.
template <typename T>
__device__ __forceinline__ T egnn_row16_sum(T v)
{
    v += EGNN_DPP(v, 0xB1, 0xF, true);       // quad_perm [1,0,3,2]
    v += EGNN_DPP(v, 0x4E, 0xF, true);       // quad_perm [2,3,0,1]
    v += EGNN_DPP(v, 0x141, 0xF, true);      // row_half_mirror
    v += EGNN_DPP(v, 0x140, 0xF, true);      // row_mirror
    return v;
}

// Sum over the wave; the result is wave-uniform (built from the four row sums with v_readlane).
__device__ __forceinline__ float egnn_wave_sum(float v)
{
    v = egnn_row16_sum(v);
    const int b = __builtin_bit_cast(int, v);
    return ((__builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16))) +
            __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32))) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48));
}
__device__ __forceinline__ int egnn_wave_sum(int v)
{
    v = egnn_row16_sum(v);
    return ((__builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16)) + __builtin_amdgcn_readlane(v, 32)) +
           __builtin_amdgcn_readlane(v, 48);
}

// For each column e = lane & 15: the sum over the four 16-lane rows, i.e. over lanes e, e+16, e+32, e+48 (what
// `v += shfl_xor(v, 16); v += shfl_xor(v, 32)` computes), in every lane -- through 64 floats of wave-private LDS
// (plain ds_write_b32 / ds_read_b32; same-wave DS operations execute in issue order, the waits pin the compiler).
// (The gfx950 v_permlane16/32_swap builtins would do it in registers, but hipcc 7.2 folds their second result away
// when both operands are the same value.)
__device__ __forceinline__ float egnn_column_sum4(float v, float* scratch64, int lane)
{
    scratch64[lane] = v;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    const int e = lane & 15;
    const float t = ((scratch64[e] + scratch64[16 + e]) + scratch64[32 + e]) + scratch64[48 + e];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    return t;
}

// For each column e = lane & 15: the sum over the four 16-lane rows, in every lane, in registers: gfx950's row-swap instructions
// (v_permlane16_swap: odd rows of the first operand <-> even rows of the second; v_permlane32_swap: upper half <-> lower half) on two
// copies of the value, as inline assembly (see above: the builtin's second result is folded away).  Order: (row0 + row1) + (row2 + row3).
#ifndef EGNN_NO_PERMLANE_SWAP
__device__ __forceinline__ float egnn_column_sum4_reg(float v)
{
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    const float s = a + b;
    float c = s, d = s;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(c), "+v"(d));
    return c + d;
}
#endif

// Inclusive prefix sum over the 64 lanes (row_shr 1/2/4/8 inside the rows, then row_bcast15 / row_bcast31).
__device__ __forceinline__ int egnn_wave_inclusive_scan(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, true);     // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, true);     // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, true);     // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, true);     // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);    // row_bcast:15 -> rows 1, 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);    // row_bcast:31 -> rows 2, 3
    return v;
}

// max |x| by-products of kernels whose output is the next gradient GEMM's operand (the host picks the operand's power-of-two scale
// from it, egnn_absmax_f32's contract: the bit pattern of the float, integer atomicMax, order independent; a NaN's pattern wins).
// m = the thread's running maximum of (bits & 0x7fffffff); slot = one word of workgroup LDS, zeroed here.
__device__ __forceinline__ uint32_t egnn_abs_bits(float x) { return __builtin_bit_cast(uint32_t, x) & 0x7fffffffu; }
__device__ __forceinline__ void egnn_block_absmax_commit(uint32_t m, uint32_t* slot, uint32_t* out_bits)
{
    if (threadIdx.x == 0) *slot = 0u;
    __syncthreads();
    if (m) atomicMax(slot, m);
    __syncthreads();
    // (tens of thousands of workgroups on one word: only those that would raise it go to the atomic unit -- a stale read costs one
    // redundant atomic, never a wrong result)
    if (threadIdx.x == 0 && *slot > __atomic_load_n(out_bits, __ATOMIC_RELAXED)) atomicMax(out_bits, *slot);
}

#pragma GCC poison __shfl __shfl_xor __shfl_up __shfl_down

// Range status (include/egnn_hip.h: EGNN_RANGE_*): an atomic from the lanes that saw a violation, nothing otherwise.
__device__ __forceinline__ void egnn_flag_range(int32_t* status, bool bad, int bit)
{
    if (status && bad) atomicOr(status, bit);
}
// finite but beyond what an fp16 (hi, lo) pair carries
__device__ __forceinline__ bool egnn_beyond_f16(float x) { const float a = fabsf(x); return a >= 65504.f && a < __builtin_inff(); }

// SiLU(x) = x * sigmoid(x) (reference: nn.SiLU, egnn_pytorch.py:56-60).
// v_exp_f32 + v_rcp_f32; |error| ~1e-7 relative, far inside the 1e-4 parity budget.
__device__ __forceinline__ float egnn_silu(float x) {
    return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x));
}
__device__ __forceinline__ float egnn_sigmoid(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __expf(-x));
}

// Training-mode dropout (egnn_pytorch.py:176: ONE nn.Dropout shared by edge_mlp, node_mlp and coors_mlp, each time behind the
// first Linear).  The E x H activations of edge_mlp never exist in memory, so neither can a mask: the keep decision of element
// (row, col) of a site is a counter-based hash of (seed, site, row, col) -- the same function in the forward kernels and in
// whatever differentiates them (egnn_pytorch_amd/_dropout.py is its torch twin).  row = edge id b N K + i K + k (sites 0, 1) or
// node id b N + i (site 2); col = the unit's index in the reference's hidden dimension.  keep <=> hash >= thr, thr = p 2^32.
#define EGNN_DROP_SITE_EDGE 0u
#define EGNN_DROP_SITE_COORS 1u
#define EGNN_DROP_SITE_NODE 2u
__host__ __device__ __forceinline__ uint32_t egnn_drop_base(uint32_t seed, uint32_t site, uint32_t row)
{
    return row * 0x9E3779B1u + seed + site * 0x27D4EB2Fu;
}
__host__ __device__ __forceinline__ uint32_t egnn_drop_hash(uint32_t base, uint32_t col)
{
    uint32_t x = base + col * 0x85EBCA77u;
    x ^= x >> 15;
    x *= 0x2C1B3C6Du;
    x ^= x >> 12;
    x *= 0x297A2D39u;
    x ^= x >> 15;
    return x;
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

// the same from the differences (egnn_slot_prep_f32 stores x_i - x_j): identical operations, identical bits
__device__ __forceinline__ float egnn_sqdist_rel(float dx, float dy, float dz) {
#pragma clang fp contract(off)
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

// Any coordinate dimension (C > 8 on the plain kernels; every C in float64): the summation tree of the reference's
// `(rel_coors ** 2).sum(dim=-1)` on the CPU -- ATen's inner-dimension sum (aten/src/ATen/native/cpu/SumKernel.cpp), restated and
// checked bit for bit against torch 2.10 for C = 1 .. 513 in float32 and 1 .. 200 in float64 (the tests' CPU checker holds the same
// statement and is compared with torch itself in tests/).  V = lanes of its vector type: 8 for float32, 4 for float64.
//   C <  V: four interleaved partial sums p[j] = x[j] + x[4 + j] + ..., the tail x[4 (C / 4) ..] added to p[0], then ((p0 + p1) + p2) + p3;
//   C >= V: whole vectors v_i = x[V i .. V i + V): four interleaved partial vectors over groups of four, the remaining vectors added to
//           the first, the four folded into it; then a scalar that starts at 0, takes the tail x[V (C / V) ..] first and the V lanes after.
// (long rows cascade in blocks of 16 groups: the statement holds for C <= 512 in float32, 256 in float64.)  Every operation rounded
// separately -- no FMA contraction.  For C <= 8 in float32 it coincides with egnn_sqdist / egnn_sqdist_n above.
template <typename T, int V>
__device__ __forceinline__ T egnn_sqdist_any(const T* __restrict__ a, const T* __restrict__ b, int C) {
#pragma clang fp contract(off)
    auto sq = [&](int c) { const T r = a[c] - b[c]; return r * r; };
    if (C < V) {
        T p0 = (T)0, p1 = (T)0, p2 = (T)0, p3 = (T)0;
        const int groups = C >> 2;
        for (int i = 0; i < groups; ++i) {
            p0 = p0 + sq(4 * i);
            p1 = p1 + sq(4 * i + 1);
            p2 = p2 + sq(4 * i + 2);
            p3 = p3 + sq(4 * i + 3);
        }
        for (int k = 4 * groups; k < C; ++k) p0 = p0 + sq(k);
        return ((p0 + p1) + p2) + p3;
    }
    const int nv = C / V, groups = nv >> 2;
    T p[V];
#pragma unroll
    for (int k = 0; k < V; ++k) p[k] = (T)0;
    if (groups > 0) {                                            // (C >= 4 V only: three more partial vectors)
        T q1[V], q2[V], q3[V];
#pragma unroll
        for (int k = 0; k < V; ++k) { q1[k] = (T)0; q2[k] = (T)0; q3[k] = (T)0; }
        for (int i = 0; i < groups; ++i) {
#pragma unroll
            for (int k = 0; k < V; ++k) {
                p[k] = p[k] + sq(V * (4 * i) + k);
                q1[k] = q1[k] + sq(V * (4 * i + 1) + k);
                q2[k] = q2[k] + sq(V * (4 * i + 2) + k);
                q3[k] = q3[k] + sq(V * (4 * i + 3) + k);
            }
        }
        for (int i = 4 * groups; i < nv; ++i) {
#pragma unroll
            for (int k = 0; k < V; ++k) p[k] = p[k] + sq(V * i + k);
        }
#pragma unroll
        for (int k = 0; k < V; ++k) p[k] = ((p[k] + q1[k]) + q2[k]) + q3[k];
    } else {
        for (int i = 0; i < nv; ++i) {
#pragma unroll
            for (int k = 0; k < V; ++k) p[k] = p[k] + sq(V * i + k);
        }
    }
    T acc = (T)0;
    for (int k = V * nv; k < C; ++k) acc = acc + sq(k);
#pragma unroll
    for (int k = 0; k < V; ++k) acc = acc + p[k];
    return acc;
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
                          int32_t* status, void* stream);

static inline int egnn_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? EGNN_OK : (int)e;
}
