// Per-graph spatial (Morton / Z-order) permutation of the nodes -- a scheduling aid for the fused edge pass.
//
// The edge kernel gathers one row of P_j per edge.  When the nodes a workgroup owns are close in space their k-NN
// sets overlap (measured on the north-star shape: 8 nodes x 32 neighbours touch 91 distinct rows instead of 227),
// so the gathered rows are shared through the CU's L1 instead of being fetched again from L2.  The permutation only
// changes WHICH nodes a workgroup processes together; every node's result is computed exactly as before (same
// neighbour list, same summation order), so outputs are bit-identical with or without it.
//
// One 256-thread workgroup per graph: bounding box by LDS reduction, 10 bits per axis, 30-bit Morton code,
// bitonic sort of (code << 32 | index) keys in LDS.  N <= 4096.
#include "egnn_common.h"

namespace {

__device__ __forceinline__ uint32_t spread3(uint32_t v) {          // 10 bits -> every third bit
    v &= 0x3ffu;
    v = (v | (v << 16)) & 0x030000FFu;
    v = (v | (v << 8)) & 0x0300F00Fu;
    v = (v | (v << 4)) & 0x030C30C3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

__global__ __launch_bounds__(256) void spatial_order_kernel(const float* __restrict__ coors, int N, int Np,
                                                            int32_t* __restrict__ order)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint64_t* keys = reinterpret_cast<uint64_t*>(smem);            // [Np]
    float* red = reinterpret_cast<float*>(smem + (size_t)Np * 8);  // [6][256]
    const int tid = threadIdx.x;
    const int b = blockIdx.x;
    const float* c = coors + (size_t)b * N * 3;

    float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
    for (int i = tid; i < N; i += 256)
        for (int a = 0; a < 3; ++a) {
            const float v = c[i * 3 + a];
            lo[a] = fminf(lo[a], v);
            hi[a] = fmaxf(hi[a], v);
        }
    for (int a = 0; a < 3; ++a) { red[a * 256 + tid] = lo[a]; red[(3 + a) * 256 + tid] = hi[a]; }
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s)
            for (int a = 0; a < 3; ++a) {
                red[a * 256 + tid] = fminf(red[a * 256 + tid], red[a * 256 + tid + s]);
                red[(3 + a) * 256 + tid] = fmaxf(red[(3 + a) * 256 + tid], red[(3 + a) * 256 + tid + s]);
            }
        __syncthreads();
    }
    float mn[3], sc[3];
    for (int a = 0; a < 3; ++a) {
        mn[a] = red[a * 256];
        const float ext = red[(3 + a) * 256] - mn[a];
        sc[a] = ext > 0.f ? 1023.0f / ext : 0.f;
    }
    for (int i = tid; i < Np; i += 256) {
        uint64_t k = ~0ull;
        if (i < N) {
            uint32_t code = 0;
            for (int a = 0; a < 3; ++a) {
                float q = (c[i * 3 + a] - mn[a]) * sc[a];
                q = fminf(fmaxf(q, 0.f), 1023.f);                    // NaN -> 0
                code |= spread3((uint32_t)q) << a;
            }
            k = ((uint64_t)code << 32) | (uint32_t)i;
        }
        keys[i] = k;
    }
    __syncthreads();
    for (int size = 2; size <= Np; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = tid; t < Np / 2; t += 256) {
                const int pos = 2 * t - (t & (stride - 1));
                const bool up = ((pos & size) == 0);
                const uint64_t x = keys[pos], y = keys[pos + stride];
                if ((x > y) == up) { keys[pos] = y; keys[pos + stride] = x; }
            }
            __syncthreads();
        }
    for (int i = tid; i < N; i += 256) order[(size_t)b * N + i] = (int32_t)(uint32_t)(keys[i] & 0xffffffffull);
}

}  // namespace

extern "C" int egnn_spatial_order_f32(const float* coors, int B, int N, int32_t* order_out, void* stream)
{
    if (!coors || !order_out) return EGNN_E_NULLPTR;
    if (B <= 0 || N <= 0) return EGNN_E_SHAPE;
    if (N > 4096) return EGNN_E_UNSUPPORTED;
    int Np = 2;
    while (Np < N) Np <<= 1;
    const size_t lds = (size_t)Np * 8 + 6 * 256 * sizeof(float);
    hipLaunchKernelGGL(spatial_order_kernel, dim3(B), dim3(256), lds, static_cast<hipStream_t>(stream), coors, N, Np,
                       order_out);
    return egnn_launch_status();
}
