// The edge pass in plain fp32 (reference: egnn_pytorch/egnn_pytorch.py:262-333) -- the WIDE-RANGE path.
//
// The fast kernels (edge_fused.hip / edge_pw.hip) evaluate every product as a split-fp16 product on the matrix cores: fp32-class
// accuracy, but an activation beyond fp16's 65504 cannot be carried and the call ends with a range status bit.
// The reference computes in plain fp32 and has no such limit, so inputs that trip a bit -- features x 1e6, coordinates x 1e5, edge
// features x 1e9 -- are reference-legal.  This kernel is what the host binding re-runs such a call on: every operation an fp32
// VALU instruction on unscaled operands, exactly the arithmetic class of the reference (fp32 products, fp32 accumulation, overflow
// only where fp32 itself overflows), with the weights read as the module holds them (no re-layout).  It is several times slower
// than the fast path (16 FMAs per hidden value instead of 3/16 of an MFMA) and is never what bench.py times.
//
//   kernel 1, one thread per edge (b, i, k):  x[h] = P_i[i,h] + P_j[j,h] + sum_s scal[s] W_s[h,s];  a = SiLU(x);
//             u[c] += W2[c,h] a  (the weights are wave-uniform: scalar loads);  m = SiLU(u + b2) (* gate);  coors_mlp;  masks, clamp,
//             CoorsNorm  ->  per-edge rows [m (m_dim) | w_ij * rel (C) | kept (1)] in a caller-owned workspace
//   kernel 2, one thread per (node, channel): the K rows of the node summed in k order (deterministic), mean / safe_div, outputs.
#include "egnn_common.h"

namespace {

constexpr int EX_THREADS = 256;
constexpr int EX_SMAX = 64;                  // per-edge scalars 2 F + 1 + edge_dim (the fast path stops at 16)

// (expf, not the fast exponential of the split-f16 kernels: this path trades speed for the reference's arithmetic class)
__device__ __forceinline__ float ex_silu(float x) { return x / (1.0f + expf(-x)); }

template <int MB>                            // message channels held in registers: 16 / 32 / 64
__global__ __launch_bounds__(EX_THREADS) void edge_exact_kernel(const egnn_edge_exact_args p)
{
    const int64_t E = (int64_t)p.B * p.N * p.K;
    const int64_t q = (int64_t)blockIdx.x * EX_THREADS + threadIdx.x;
    if (q >= E) return;
    const int N = p.N, K = p.K, C = p.coor_dim, F = p.fourier, m_dim = p.m_dim, H = p.H;
    const int S = 2 * F + 1 + p.edge_dim;
    const int64_t node = q / K;                                          // b N + i
    const int k = (int)(q - node * K);
    const int64_t bN = node / N * N;
    const int i = (int)(node - bN);
    const int j = p.idx ? p.idx[q] : k;

    // x_i - x_j and the squared distance, in the reference's operation order (egnn_common.h): what the neighbour selection ranked by
    float rel[8];
    float d;
    {
        const float* ci = p.coors + (bN + i) * C;
        const float* cj = p.coors + (bN + j) * C;
        if (C == 3) {
            d = egnn_sqdist(ci[0], ci[1], ci[2], cj[0], cj[1], cj[2], rel[0], rel[1], rel[2]);
#pragma unroll
            for (int c = 3; c < 8; ++c) rel[c] = 0.f;
        } else {
            float a[8], bb[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) { a[c] = c < C ? ci[c] : 0.f; bb[c] = c < C ? cj[c] : 0.f; }
            d = egnn_sqdist_n<8>(a, bb, C, rel);
        }
    }
    // per-edge scalars [sin(d / 2^f) ..., cos(d / 2^f) ..., d, edge features ...]  (:34-41, :270-272, :282-285): a column of LDS per
    // thread (S x 256 floats, dynamic; scalar s of thread t at [s * 256 + t]: conflict-free) -- a run-time indexed private array
    // would live in scratch memory
    extern __shared__ float scal_lds[];
    float* const scal = scal_lds + threadIdx.x;
    for (int f = 0; f < F; ++f) {
        const float x = d / exp2f((float)f);
        scal[f * EX_THREADS] = sinf(x);
        scal[(F + f) * EX_THREADS] = cosf(x);
    }
    scal[2 * F * EX_THREADS] = d;
    if (p.edge_dim > 0) {
        const float* ep = p.edges + (p.edges_by_k ? (size_t)q : ((size_t)(bN + i) * N + j)) * p.edge_dim;
        for (int s = 0; s < p.edge_dim; ++s) scal[(2 * F + 1 + s) * EX_THREADS] = ep[s];
    }

    // ---- edge_mlp: Linear (factorised: node-level projections + the scalars' columns), SiLU, Linear
    const float* pi = p.Pi + (bN + i) * p.ldp;
    const float* pj = p.Pj + (bN + j) * p.ldp;
    float u[MB];
#pragma unroll
    for (int c = 0; c < MB; ++c) u[c] = 0.f;
    for (int h = 0; h < H; ++h) {
        float x = pi[h] + pj[h];
        const float* ws = p.Ws + (size_t)h * p.ldws;
        if (S == 1) x = fmaf(d, ws[0], x);                               // (the common case stays in a register)
        else for (int s = 0; s < S; ++s) x = fmaf(scal[s * EX_THREADS], ws[s], x);
        const float a = ex_silu(x);
        const float* w2 = p.W2 + h;
#pragma unroll
        for (int c = 0; c < MB; ++c)
            if (c < m_dim) u[c] = fmaf(w2[(size_t)c * H], a, u[c]);
    }
    float m[MB];
#pragma unroll
    for (int c = 0; c < MB; ++c) m[c] = c < m_dim ? ex_silu(u[c] + p.b2[c]) : 0.f;
    if (p.gate_w) {                                                      // soft_edges (:289-290)
        float gsum = p.gate_b[0];
#pragma unroll
        for (int c = 0; c < MB; ++c)
            if (c < m_dim) gsum = fmaf(p.gate_w[c], m[c], gsum);
        const float gt = 1.0f / (1.0f + expf(-gsum));
#pragma unroll
        for (int c = 0; c < MB; ++c) m[c] *= gt;
    }

    // ---- masks (:292-300): applied only when a mask is given (the reference's quirk); a wave-uniform "has mask"
    const bool has_mask = p.mask != nullptr;
    bool keep = true;
    if (has_mask) {
        keep = p.mask[bN + i] && p.mask[bN + j];
        if (p.rank && p.idx) keep = keep && (p.rank[q] <= p.valid_radius);
    }

    float* row = p.edge_ws + (size_t)q * (m_dim + C + 1);
    // ---- coors_mlp, CoorsNorm, clamp (:302-317)
    if (p.W3) {
        const int hid = 4 * m_dim;
        float cw = p.b4[0];
        for (int r = 0; r < hid; ++r) {
            float z = p.b3[r];
            const float* w3 = p.W3 + (size_t)r * m_dim;
#pragma unroll
            for (int c = 0; c < MB; ++c)
                if (c < m_dim) z = fmaf(w3[c], m[c], z);
            cw = fmaf(p.W4[r], ex_silu(z), cw);
        }
        float inv = 1.f;
        if (p.coors_scale) {                                             // CoorsNorm (:67-77)
            float n2 = 0.f;
#pragma unroll
            for (int c = 0; c < 8; ++c) n2 = fmaf(rel[c], rel[c], n2);
            inv = p.coors_scale[0] / fmaxf(sqrtf(n2), 1e-8f);
        }
        if (has_mask && !keep) cw = 0.f;                                 // :308-309
        if (p.clamp >= 0.f) cw = fminf(fmaxf(cw, -p.clamp), p.clamp);    // :311-313
#pragma unroll
        for (int c = 0; c < 8; ++c)
            if (c < C) row[m_dim + c] = cw * (rel[c] * inv);
    } else {
#pragma unroll
        for (int c = 0; c < 8; ++c)
            if (c < C) row[m_dim + c] = 0.f;
    }
#pragma unroll
    for (int c = 0; c < MB; ++c)
        if (c < m_dim) row[c] = (has_mask && !keep) ? 0.f : m[c];        // masked_fill (:320-322)
    row[m_dim + C] = keep ? 1.f : 0.f;
}

__global__ __launch_bounds__(EX_THREADS) void edge_exact_pool_kernel(const egnn_edge_exact_args p)
{
    const int C = p.coor_dim, m_dim = p.m_dim, K = p.K;
    const int nch = m_dim + C;
    const int64_t total = (int64_t)p.B * p.N * nch;
    const int64_t o = (int64_t)blockIdx.x * EX_THREADS + threadIdx.x;
    if (o >= total) return;
    const int64_t node = o / nch;
    const int ch = (int)(o - node * nch);
    const int ld = m_dim + C + 1;
    const float* rows = p.edge_ws + (size_t)node * K * ld;
    float s = 0.f, cnt = 0.f;
    for (int k = 0; k < K; ++k) {                                        // k order: deterministic
        s += rows[(size_t)k * ld + ch];
        cnt += rows[(size_t)k * ld + m_dim + C];
    }
    if (ch < m_dim) {
        if (p.pool_mean) {
            if (p.mask) s = (cnt == 0.f) ? 0.f : s / fmaxf(cnt, 1e-8f);  // safe_div (:13-16, :326-327)
            else s = s / (float)K;                                       // :330
        }
        if (p.m_i) p.m_i[node * m_dim + ch] = s;
    } else if (p.coors_out) {
        const int c = ch - m_dim;
        p.coors_out[node * C + c] = p.coors[node * C + c] + s;           // :315
    }
}

}  // namespace

extern "C" size_t egnn_edge_exact_workspace_bytes(int B, int N, int K, int m_dim, int coor_dim)
{
    if (B <= 0 || N <= 0 || K <= 0 || m_dim <= 0 || coor_dim <= 0) return 0;
    return (size_t)B * N * K * (size_t)(m_dim + coor_dim + 1) * sizeof(float);
}

extern "C" int egnn_edge_exact_f32(const egnn_edge_exact_args* args, void* stream)
{
    if (!args) return EGNN_E_NULLPTR;
    const egnn_edge_exact_args& a = *args;
    if (!a.Pi || !a.Pj || !a.Ws || !a.W2 || !a.b2 || !a.coors || !a.edge_ws) return EGNN_E_NULLPTR;
    if (!a.m_i && !a.coors_out) return EGNN_E_NULLPTR;
    if (a.B <= 0 || a.N <= 0 || a.K <= 0 || a.H <= 0 || a.ldp < a.H || a.ldws < 2 * a.fourier + 1 + a.edge_dim) return EGNN_E_SHAPE;
    if (a.m_dim < 1 || a.m_dim > 64 || a.coor_dim < 1 || a.coor_dim > 8) return EGNN_E_UNSUPPORTED;
    if (a.fourier < 0 || a.edge_dim < 0 || 2 * a.fourier + 1 + a.edge_dim > EX_SMAX) return EGNN_E_UNSUPPORTED;
    if (a.edge_dim > 0 && !a.edges) return EGNN_E_NULLPTR;
    if (a.coors_out && (!a.W3 || !a.b3 || !a.W4 || !a.b4)) return EGNN_E_NULLPTR;
    if (a.gate_w && !a.gate_b) return EGNN_E_NULLPTR;
    if (!a.idx && a.K != a.N) return EGNN_E_SHAPE;                        // dense path: K == N
    const int64_t E = (int64_t)a.B * a.N * a.K;
    const int64_t blocks = (E + EX_THREADS - 1) / EX_THREADS;
    if (blocks > 0x7fffffffLL) return EGNN_E_UNSUPPORTED;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t lds = (size_t)(2 * a.fourier + 1 + a.edge_dim) * EX_THREADS * sizeof(float);      // <= 64 KB
    if (a.m_dim <= 16) hipLaunchKernelGGL(edge_exact_kernel<16>, dim3((unsigned)blocks), dim3(EX_THREADS), lds, s, a);
    else if (a.m_dim <= 32) hipLaunchKernelGGL(edge_exact_kernel<32>, dim3((unsigned)blocks), dim3(EX_THREADS), lds, s, a);
    else hipLaunchKernelGGL(edge_exact_kernel<64>, dim3((unsigned)blocks), dim3(EX_THREADS), lds, s, a);
    int rc = egnn_launch_status();
    if (rc != EGNN_OK) return rc;
    const int64_t total = (int64_t)a.B * a.N * (a.m_dim + a.coor_dim);
    hipLaunchKernelGGL(edge_exact_pool_kernel, dim3((unsigned)((total + EX_THREADS - 1) / EX_THREADS)), dim3(EX_THREADS), 0, s, a);
    return egnn_launch_status();
}
