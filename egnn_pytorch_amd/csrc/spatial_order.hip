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

__global__ __launch_bounds__(256) void spatial_order_kernel(const float* __restrict__ coors, const uint8_t* __restrict__ mask, int N, int Np,
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
            // padded nodes (mask = 0) behind the real ones: the edge pass skips a group of four whose nodes are all padding (bit 62; the
            // padding entries of the sort carry ~0 and stay last)
            if (mask && !mask[(size_t)b * N + i]) code |= 0x40000000u;
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

// One thread per edge slot, in the edge pass's consumption order: the dependent loads of its setup done once, coalesced.
__global__ __launch_bounds__(256) void slot_prep_kernel(const float* __restrict__ coors, const uint8_t* __restrict__ mask,
                                                        const int32_t* __restrict__ idx, const float* __restrict__ rank,
                                                        const int32_t* __restrict__ order, float valid_radius, int N, int K,
                                                        int64_t total, uint4* __restrict__ slots)
{
    const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (q >= total) return;
    const int64_t total_nodes = total / K;
    const int64_t node = q / K;                              // b * N + pos
    const int k = (int)(q - node * K);
    const int64_t bN = node / N * N;
    const int i = order ? order[node] : (int)(node - bN);
    const int64_t e = (bN + i) * K + k;
    const int j = idx ? idx[e] : k;                          // idx NULL: dense all-pairs (K == N), neighbour k is node k
    const float* ci = coors + (bN + i) * 3;
    const float* cj = coors + (bN + j) * 3;
    bool ok = true;
    if (mask) {
        ok = mask[bN + i] && mask[bN + j];
        if (rank) ok = ok && (rank[e] <= valid_radius);
    }
    uint4 r;
    r.x = (uint32_t)j | (ok ? 0x80000000u : 0u);
    if (mask && (k & 31) == 0) {
        // bit 30 of the first record of every 32-slot round: all four nodes of this node's group (positions 4 (node / 4) .. + 3 of the
        // consumption order, over the whole batch) are padding -- the wave-per-node edge kernel (csrc/edge_pw.hip) then skips the
        // workgroup's round without touching its staging ring.  The same four masks for the four nodes of a group, by construction.
        const int64_t g0 = node & ~(int64_t)3;
        bool dead = true;
        for (int m = 0; m < 4; ++m) {
            const int64_t nm = g0 + m < total_nodes ? g0 + m : total_nodes - 1;
            const int64_t bm = nm / N * N;
            const int im = order ? order[nm] : (int)(nm - bm);
            dead = dead && !mask[bm + im];
        }
        if (dead) r.x |= 0x40000000u;
    }
    r.y = __float_as_uint(ci[0] - cj[0]);
    r.z = __float_as_uint(ci[1] - cj[1]);
    r.w = __float_as_uint(ci[2] - cj[2]);
    slots[q] = r;
}

}  // namespace

extern "C" int egnn_slot_prep_f32(const float* coors, const uint8_t* mask, const int32_t* idx, const float* rank, const int32_t* order,
                                  float valid_radius, int B, int N, int K, void* slots, void* stream)
{
    if (!coors || !slots) return EGNN_E_NULLPTR;
    if (B <= 0 || N <= 0 || K <= 0 || (!idx && K != N)) return EGNN_E_SHAPE;
    if (reinterpret_cast<uintptr_t>(slots) & 15) return EGNN_E_ALIGN;
    const int64_t total = (int64_t)B * N * K;
    const int64_t blocks = (total + 255) / 256;
    if (blocks > 0x7fffffffLL) return EGNN_E_UNSUPPORTED;
    hipLaunchKernelGGL(slot_prep_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), coors, mask, idx, rank,
                       order, valid_radius, N, K, total, static_cast<uint4*>(slots));
    return egnn_launch_status();
}

extern "C" int egnn_spatial_order_f32(const float* coors, int B, int N, int32_t* order_out, void* stream)
{
    return egnn_spatial_order_masked_f32(coors, nullptr, B, N, order_out, stream);
}

extern "C" int egnn_spatial_order_masked_f32(const float* coors, const uint8_t* mask, int B, int N, int32_t* order_out, void* stream)
{
    if (!coors || !order_out) return EGNN_E_NULLPTR;
    if (B <= 0 || N <= 0) return EGNN_E_SHAPE;
    if (N > 4096) return EGNN_E_UNSUPPORTED;
    int Np = 2;
    while (Np < N) Np <<= 1;
    const size_t lds = (size_t)Np * 8 + 6 * 256 * sizeof(float);
    hipLaunchKernelGGL(spatial_order_kernel, dim3(B), dim3(256), lds, static_cast<hipStream_t>(stream), coors, mask, N, Np,
                       order_out);
    return egnn_launch_status();
}
