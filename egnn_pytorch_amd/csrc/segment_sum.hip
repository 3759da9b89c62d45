// Deterministic row gather-sum: out[r, :] = sum over p in [seg_ptr[r], seg_ptr[r+1]) of rows[order[p], :], in that order.
//
// The backward of the neighbour gather (egnn_pytorch.py:275, feats_j = batched_index_select(feats, nbhd_indices)):
// d loss / d P_j[j] is the sum of dz over every edge (i, k) whose neighbour is j.  `order` lists the edges sorted by
// destination (stable, so ties stay in edge order) -- the transposed neighbour list -- which makes the sum a fixed-order
// read-only reduction: no float atomics, bit-reproducible.  One WAVE per destination row (four rows per workgroup: narrow rows --
// the c3 layer's 1152 bytes -- would leave most of a 256-thread workgroup idle); every wave instruction reads 1 KB of one source
// row; HBM-bound (each source row is read exactly once).
#include "egnn_common.h"

namespace {

__global__ __launch_bounds__(256) void rows_gather_sum_kernel(const float* __restrict__ rows, int64_t ld, const int64_t* __restrict__ order,
                                                              const int64_t* __restrict__ seg_ptr, int64_t n_out, int cols,
                                                              float* __restrict__ out, int64_t ldo, uint32_t* __restrict__ amax_bits)
{
    __shared__ uint32_t amax_slot;
    uint32_t mx = 0u;
    const int64_t r_raw = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const bool live = r_raw < n_out;
    const int64_t r = live ? r_raw : n_out - 1;
    const int64_t p0 = seg_ptr[r], p1 = live ? seg_ptr[r + 1] : p0;
    for (int c = (threadIdx.x & 63) * 4; live && c < cols; c += 64 * 4) {
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
        int64_t p = p0;
        for (; p + 1 < p1; p += 2) {                                   // two rows in flight
            const f32x4 a = *reinterpret_cast<const f32x4*>(rows + order[p] * ld + c);
            const f32x4 b = *reinterpret_cast<const f32x4*>(rows + order[p + 1] * ld + c);
            acc += a;
            acc += b;
        }
        if (p < p1) acc += *reinterpret_cast<const f32x4*>(rows + order[p] * ld + c);
        *reinterpret_cast<f32x4*>(out + r * ldo + c) = acc;
        if (amax_bits) {
#pragma unroll
            for (int u = 0; u < 4; ++u) { const uint32_t t = egnn_abs_bits(acc[u]); mx = mx > t ? mx : t; }
        }
    }
    if (amax_bits) egnn_block_absmax_commit(mx, &amax_slot, amax_bits);
}

}  // namespace

extern "C" int egnn_rows_gather_sum_f32(const float* rows, int64_t ld, const int64_t* order, const int64_t* seg_ptr, int64_t n_out,
                                        int cols, float* out, int64_t ldo, uint32_t* amax_bits, void* stream)
{
    if (!rows || !order || !seg_ptr || !out) return EGNN_E_NULLPTR;
    if (n_out <= 0 || cols <= 0 || (cols % 4) != 0 || ld < cols || ldo < cols || (ld % 4) != 0 || (ldo % 4) != 0) return EGNN_E_SHAPE;
    if (n_out > 0x7fffffffLL) return EGNN_E_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(rows) & 15) || (reinterpret_cast<uintptr_t>(out) & 15)) return EGNN_E_ALIGN;
    if (amax_bits && hipMemsetAsync(amax_bits, 0, sizeof(uint32_t), static_cast<hipStream_t>(stream)) != hipSuccess) return (int)hipGetLastError();
    hipLaunchKernelGGL(rows_gather_sum_kernel, dim3((unsigned)((n_out + 3) / 4)), dim3(256), 0, static_cast<hipStream_t>(stream), rows, ld, order,
                       seg_ptr, n_out, cols, out, ldo, amax_bits);
    return egnn_launch_status();
}


// ---------------------------------------------------------------------------------------------------------------------------
// EGNN_Network front-end (egnn_pytorch.py:410-432): the per-pair edge features of the K selected pairs of every node, looked up
// from the embedding tables -- see include/egnn_hip.h::egnn_edge_features_gather_f32.  One thread per (edge, column).
namespace {

__global__ __launch_bounds__(256) void edge_features_gather_kernel(const float* __restrict__ edges, const int64_t* __restrict__ tok,
                                                                   const float* __restrict__ tok_emb, int d1,
                                                                   const uint8_t* __restrict__ deg, const float* __restrict__ deg_emb,
                                                                   int d2, const int32_t* __restrict__ idx, int N, int K, int64_t total,
                                                                   float* __restrict__ out)
{
    const int w = d1 + d2;
    for (int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x; o < total; o += (int64_t)gridDim.x * 256) {
        const int64_t edge = o / w;
        const int col = (int)(o - edge * w);
        const int64_t node = edge / K;                              // b * N + i
        const int k = (int)(edge - node * K);
        const int j = idx ? idx[edge] : k;
        const int64_t pair = node * N + j;
        float v;
        if (col < d1) v = tok ? tok_emb[tok[pair] * d1 + col] : edges[pair * d1 + col];
        else v = deg_emb[(int64_t)deg[pair] * d2 + (col - d1)];
        out[o] = v;
    }
}

}  // namespace

extern "C" int egnn_edge_features_gather_f32(const float* edges, const int64_t* edge_tok, const float* edge_tok_emb, int d1,
                                             const uint8_t* adj_deg, const float* adj_deg_emb, int d2, const int32_t* idx,
                                             int B, int N, int K, float* out, void* stream)
{
    if (!out) return EGNN_E_NULLPTR;
    if (B <= 0 || N <= 0 || K <= 0 || d1 < 0 || d2 < 0 || d1 + d2 <= 0) return EGNN_E_SHAPE;
    if (d1 > 0 && !edges && !(edge_tok && edge_tok_emb)) return EGNN_E_NULLPTR;
    if (d2 > 0 && (!adj_deg || !adj_deg_emb)) return EGNN_E_NULLPTR;
    if (!idx && K != N) return EGNN_E_SHAPE;
    const int64_t total = (int64_t)B * N * K * (d1 + d2);
    int64_t blocks = (total + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(edge_features_gather_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), edges,
                       edge_tok, edge_tok_emb, d1, adj_deg, adj_deg_emb, d2, idx, N, K, total, out);
    return egnn_launch_status();
}
