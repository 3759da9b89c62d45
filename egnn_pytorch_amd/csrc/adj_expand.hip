// N-degree adjacency expansion of EGNN_Network (reference: egnn_pytorch/egnn_pytorch.py:414-427) as bit-set algebra.
//
// The reference labels 2nd, 3rd, ... degree neighbours by repeatedly squaring the adjacency with a FLOAT matmul
// ((adj.float() @ adj.float()) > 0: 2 N^3 flops per graph and degree) and marks the entries where the squared
// adjacency differs from the current one ((next.float() - adj.float()).bool() -- an XOR).  Here a row of the adjacency
// is a bit set of W = ceil(N/64) 64-bit words held one word per lane, and
//     next[i] = OR over { j : adj[i][j] } of adj[j]
// is N/64 wave-wide OR instructions per set bit: integer/byte work, bounded by reading adj once and writing the labels
// and the expanded adjacency once (3 bytes per ordered pair).  Results are bit-exact by construction.
//
// One wave per row; rows of the current bit matrix live in a ping-pong workspace (2 x B x N x W words).  N <= 4096.
#include "egnn_common.h"

namespace {

__global__ __launch_bounds__(256) void adj_pack_kernel(const uint8_t* __restrict__ adj, int64_t bstride, int N, int W,
                                                       uint64_t* __restrict__ bits, uint8_t* __restrict__ labels, int B)
{
    const int lane = threadIdx.x & 63;
    const int64_t row_g = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);       // global row = b * N + i
    const int b = (int)(row_g / N), i = (int)(row_g % N);
    if (b >= B) return;
    const uint8_t* src = adj + (size_t)b * bstride + (size_t)i * N;
    for (int w = 0; w < W; ++w) {
        const int j = w * 64 + lane;
        const bool v = j < N && src[j] != 0;
        const uint64_t word = __ballot(v);
        if (lane == 0) bits[((size_t)b * N + i) * W + w] = word;
        if (j < N) labels[((size_t)b * N + i) * N + j] = v ? 1 : 0;           // adj_indices = adj_mat.long()  (:419)
    }
}

__global__ __launch_bounds__(256) void adj_square_kernel(const uint64_t* __restrict__ cur, uint64_t* __restrict__ nxt, int N, int W,
                                                         int degree, uint8_t* __restrict__ labels, uint8_t* __restrict__ adj_out, int B)
{
    const int lane = threadIdx.x & 63;
    const int64_t row_g = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int b = (int)(row_g / N), i = (int)(row_g % N);
    if (b >= B) return;
    const uint64_t* gb = cur + (size_t)b * N * W;
    const uint64_t mine = lane < W ? gb[(size_t)i * W + lane] : 0ull;         // word `lane` of row i
    uint64_t acc = 0ull;
    for (int w = 0; w < W; ++w) {
        uint64_t word = (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(mine & 0xffffffffu), w) |
                        ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(mine >> 32), w) << 32);
        while (word) {                                                         // wave-uniform
            const int j = w * 64 + __builtin_ctzll(word);
            word &= word - 1;
            if (lane < W) acc |= gb[(size_t)j * W + lane];                     // :422  (adj @ adj) > 0
        }
    }
    if (lane < W) nxt[((size_t)b * N + i) * W + lane] = acc;
    const uint64_t changed = acc ^ mine;                                       // :423  next != adj
    for (int w = 0; w < W; ++w) {
        const uint64_t cw = (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(changed & 0xffffffffu), w) |
                            ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(changed >> 32), w) << 32);
        const uint64_t aw = (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(acc & 0xffffffffu), w) |
                            ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(acc >> 32), w) << 32);
        const int j = w * 64 + lane;
        if (j < N) {
            if ((cw >> lane) & 1ull) labels[((size_t)b * N + i) * N + j] = (uint8_t)degree;     // :424
            if (adj_out) adj_out[((size_t)b * N + i) * N + j] = (uint8_t)((aw >> lane) & 1ull); // :425 (last step)
        }
    }
}

__global__ __launch_bounds__(256) void adj_copy_kernel(const uint8_t* __restrict__ adj, int64_t bstride, int N,
                                                       uint8_t* __restrict__ adj_out)
{
    const int64_t per = (int64_t)N * N;
    const int b = blockIdx.y;
    for (int64_t x = (int64_t)blockIdx.x * 256 + threadIdx.x; x < per; x += (int64_t)gridDim.x * 256)
        adj_out[(size_t)b * per + x] = adj[(size_t)b * bstride + x] ? 1 : 0;
}

}  // namespace

extern "C" size_t egnn_adj_expand_workspace_bytes(int B, int N)
{
    const size_t W = (size_t)(N + 63) / 64;
    return 2 * (size_t)B * N * W * sizeof(uint64_t);
}

extern "C" int egnn_adj_expand_u8(const uint8_t* adj, int64_t adj_batch_stride, int B, int N, int num_adj_degrees,
                                  uint8_t* adj_out, uint8_t* degree_out, void* workspace, void* stream)
{
    if (!adj || !adj_out || !degree_out || !workspace) return EGNN_E_NULLPTR;
    if (B <= 0 || N <= 0 || num_adj_degrees < 1) return EGNN_E_SHAPE;
    if (N > 4096 || num_adj_degrees > 255 || B > 65535) return EGNN_E_UNSUPPORTED;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int W = (N + 63) / 64;
    uint64_t* buf0 = static_cast<uint64_t*>(workspace);
    uint64_t* buf1 = buf0 + (size_t)B * N * W;
    const int64_t rows = (int64_t)B * N;
    dim3 g2((unsigned)((rows + 3) / 4));                              // one wave per global row b * N + i
    hipLaunchKernelGGL(adj_pack_kernel, g2, dim3(256), 0, s, adj, adj_batch_stride, N, W, buf0, degree_out, B);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    if (num_adj_degrees == 1) {
        hipLaunchKernelGGL(adj_copy_kernel, dim3(256, B), dim3(256), 0, s, adj, adj_batch_stride, N, adj_out);
        return egnn_launch_status();
    }
    uint64_t *cur = buf0, *nxt = buf1;
    for (int d = 2; d <= num_adj_degrees; ++d) {
        hipLaunchKernelGGL(adj_square_kernel, g2, dim3(256), 0, s, cur, nxt, N, W, d, degree_out,
                           d == num_adj_degrees ? adj_out : nullptr, B);
        e = hipGetLastError();
        if (e != hipSuccess) return (int)e;
        uint64_t* t = cur; cur = nxt; nxt = t;
    }
    return EGNN_OK;
}
