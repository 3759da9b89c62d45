// egnn_edge_tail_bwd_f32 -- the per-edge part of the backward behind edge_mlp's second Linear, in closed form (gfx950).
//
// Reference semantics: autograd through egnn_pytorch.py:287 (second SiLU), :292-317 (pair mask, coors_mlp, CoorsNorm :67-77,
// clamp, coordinate update) and :319-333 (message pooling), without the edge gate.  egnn_pytorch_amd/autograd.py::
// tail_edge_backward is the specification (float64-equal to autograd of the restated layer, tests/test_autograd.py); this is
// the same arithmetic, one edge per lane:
//     m = SiLU(u),  hid = W3 m + b3,  a3 = SiLU(hid),  w = W4 a3 + b4,  rel' = CoorsNorm(x_i - x_j),  w_c = clamp(mask(w))
//     g_w = mask(clamp'(g_coors_out[i] . rel')),   g_hid = g_w W4 SiLU'(hid),   g_u = (W3^T g_hid + mask(g_msum[i])) SiLU'(u),
//     g_rel = CoorsNorm'(w_c g_coors_out[i])
// Outputs per edge: g_u (what the E x H backward kernels take as gU), g_rel, and the two E x 64 arrays the parameter
// gradients of coors_mlp are tall products of (g_hid, a3) plus g_w and the per-edge term of d/d CoorsNorm.scale.  In ATen this
// chain is some thirty elementwise / reduction / small-GEMM passes over E x 16 and E x 64 tensors (6 ms at the north-star shape);
// here one lane walks its edge's 64 hidden values with the weights broadcast from LDS.
//
// REDUCE (args.part): the parameter gradients of coors_mlp / CoorsNorm / the edge gate / edge_mlp's last bias are sums over ALL edges
// of per-edge products.  Instead of writing the E x 64 factors out for library reductions (1.07 GB at the north-star shape, read
// back by two split-K products and three column sums), every wave reduces its 64 edges in LDS -- 16 hidden columns at a time, lane
// (t, c-group) owning four entries of d/d W3 -- and leaves one row of 1192 partial sums; egnn_sum_parts_f32 adds the rows up in
// fixed order.
#include "egnn_common.h"

#include <cstdlib>

namespace {

constexpr int TM = 16;       // padded m_dim
constexpr int TH = 64;       // padded hidden width of coors_mlp (4 m_dim)
constexpr int SLD = 36;      // floats per row of the store staging (144 B: the 16-byte writes of 16 lanes hit 64 distinct banks)
constexpr int RLD = 20;      // REDUCE: floats per row of a 16-column block of g_hid / a3
constexpr int QLD = 40;      // REDUCE: floats per row of the per-edge scalar terms
constexpr int PART = 1192;   // REDUCE: partial sums per wave: d/d W3 (64 x 16) | d/d b3 (64) | d/d W4 (64) | 40 scalar-term sums (EGNN_TAIL_PART_FLOATS)

template <bool REDUCE>
__global__ __launch_bounds__(256) void edge_tail_bwd_kernel(const egnn_edge_tail_args p)
{
    // weights in LDS (broadcast reads): as plain global loads they would sit in vector registers -- the compiler cannot prove
    // them read-only next to the kernel's stores and does not use scalar loads
    __shared__ __attribute__((aligned(16))) float sW3[TH * TM];
    __shared__ float sb3[TH], sW4[TH], sgw[TM];
    if (threadIdx.x < TM) sgw[threadIdx.x] = p.gate_w ? p.gate_w[threadIdx.x] : 0.f;
    for (int o = threadIdx.x; o < TH * TM; o += 256) sW3[o] = p.W3[o];
    if (threadIdx.x < TH) { sb3[threadIdx.x] = p.b3[threadIdx.x]; sW4[threadIdx.x] = p.W4[threadIdx.x]; }
    __syncthreads();
    // the two E x 64 outputs leave through LDS: a lane produces 4 columns of ITS row per iteration (16 bytes at a 256-byte stride
    // across the wave), staged per wave as [64 rows][32 columns] and written out as whole 128-byte lines, 8 rows per instruction
    constexpr int STAGE_FLOATS = REDUCE ? 4 * (64 * TM + 64 * QLD + 64) : 4 * 2 * 64 * SLD;
    __shared__ __attribute__((aligned(16))) float stage_raw[STAGE_FLOATS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float (*stage)[2][64][SLD] = reinterpret_cast<float (*)[2][64][SLD]>(stage_raw);          // (!REDUCE)
    float* sm = stage_raw + wave * (64 * TM + 64 * QLD + 64);                                   // REDUCE: [64][16] post-gate messages
    float* sblk = sm + 64 * TM;                                                                // [64][RLD] g_hid block | [64][RLD] a3 block; later [64][QLD]
    float* sgwe = sblk + 64 * QLD;                                                             // [64] g_w per edge
    const int64_t E = (int64_t)p.B * p.N * p.K;
    const int64_t e_raw = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool live = e_raw < E;                                               // (lanes behind the last edge compute a copy of it with every gradient zero, and store nothing)
    const int64_t e = live ? e_raw : E - 1;
    const int K = p.K, N = p.N;
    const int64_t ig = e / K;                                                  // global node (b N + i)
    const int64_t jg = p.idx ? (ig / N) * N + p.idx[e] : (ig / N) * N + (e - ig * K);
    const bool pm = live && (p.pair_mask ? p.pair_mask[e] != 0 : true);
    float* wpart = REDUCE ? p.part + ((size_t)blockIdx.x * 4 + wave) * PART : nullptr;      // this wave's row of partial sums

    float u[TM], sgu[TM], m[TM];
    {
        const f32x4* up = reinterpret_cast<const f32x4*>(p.u + e * TM);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 v = up[q];
#pragma unroll
            for (int c = 0; c < 4; ++c) u[4 * q + c] = v[c];
        }
    }
#pragma unroll
    for (int c = 0; c < TM; ++c) {
        sgu[c] = egnn_sigmoid(u[c]);
        m[c] = u[c] * sgu[c];
    }
    // edge gate (soft_edges, :289-290): m_ij = m0 * gt, gt = sigmoid(gate_w . m0 + gate_b); m0 is rebuilt from u in the backward
    float gt = 1.f, gtc = 0.f;                    // gate and 1 - gate (as sigmoid(-s): no cancellation where the gate saturates)
    if (p.gate_w) {
        float sgate = p.gate_b[0];
#pragma unroll
        for (int c = 0; c < TM; ++c) sgate = __builtin_fmaf(sgw[c], m[c], sgate);
        gt = egnn_sigmoid(sgate);
        gtc = egnn_sigmoid(-sgate);
#pragma unroll
        for (int c = 0; c < TM; ++c) m[c] *= gt;
    }
    // coors_mlp forward (weights zero padded to 64 x 16 by the host: no run-time bounds).  The 64 hidden values are not kept:
    // the backward loop below recomputes each (16 FMAs) instead of holding 64 registers across the scalar section.
    float w = p.b4[0];
#pragma unroll 2
    for (int t = 0; t < TH; ++t) {
        float h = sb3[t];
#pragma unroll
        for (int c = 0; c < TM; ++c) h = __builtin_fmaf(sW3[t * TM + c], m[c], h);
        w = __builtin_fmaf(sW4[t], egnn_silu(h), w);
    }
    // relative coordinate, CoorsNorm
    float rel[3], relp[3], g[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        rel[d] = p.coors[ig * 3 + d] - p.coors[jg * 3 + d];
        g[d] = p.g_coors_out[ig * 3 + d];
    }
    float rn = 0.f, den = 1.f, scale = 1.f;
    if (p.norm_coors) {
        rn = sqrtf(rel[0] * rel[0] + rel[1] * rel[1] + rel[2] * rel[2]);
        den = fmaxf(rn, p.eps);
        scale = p.scale[0];
#pragma unroll
        for (int d = 0; d < 3; ++d) relp[d] = rel[d] / den * scale;
    } else {
#pragma unroll
        for (int d = 0; d < 3; ++d) relp[d] = rel[d];
    }
    const float wm = pm ? w : 0.f;
    const bool clamped = p.clamp >= 0.f && !(wm >= -p.clamp && wm <= p.clamp);
    const float wc = p.clamp >= 0.f ? fminf(fmaxf(wm, -p.clamp), p.clamp) : wm;
    const float g_wc = g[0] * relp[0] + g[1] * relp[1] + g[2] * relp[2];
    const float g_w = (pm && !clamped) ? g_wc : 0.f;
    float g_relp[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) g_relp[d] = wc * g[d];
    f32x4 grel;
    float gsc_e = 0.f;                                                         // this edge's term of d/d CoorsNorm.scale
    if (p.rel_out && live) {
        *reinterpret_cast<f32x4*>(p.rel_out + e * 4) = f32x4{rel[0], rel[1], rel[2], 0.f};
        p.dist_out[e] = (rel[0] * rel[0] + rel[1] * rel[1]) + rel[2] * rel[2];
    }
    if (p.norm_coors) {
        const float dot = g_relp[0] * rel[0] + g_relp[1] * rel[1] + g_relp[2] * rel[2];
        gsc_e = dot / den;
        if (!REDUCE && p.g_scale && live) p.g_scale[e] = gsc_e;
        const float k1 = scale / den;
        const float k2 = rn >= p.eps ? dot * scale / (den * den * fmaxf(rn, 1e-30f)) : 0.f;
#pragma unroll
        for (int d = 0; d < 3; ++d) grel[d] = g_relp[d] * k1 - k2 * rel[d];
    } else {
#pragma unroll
        for (int d = 0; d < 3; ++d) grel[d] = g_relp[d];
    }
    grel[3] = 0.f;
    // A self pair (j == i: only_sparse_neighbors with the diagonal in the adjacency, dense all-pairs) has rel = x_i - x_i: its
    // gradient reaches x_i once with each sign and cancels identically.  Written out as zero -- the two copies would otherwise
    // travel through two different sums (per source, per neighbour) and, under CoorsNorm (|rel| < eps: a factor 1 / eps = 1e8),
    // leave rounding noise of order one behind.
    if (jg == ig) grel = f32x4{0.f, 0.f, 0.f, 0.f};
    if (live) {
        *reinterpret_cast<f32x4*>(p.g_rel + e * 4) = grel;
        if (!REDUCE) p.g_w[e] = g_w;
    }
    if constexpr (REDUCE) {
#pragma unroll
        for (int q = 0; q < 4; ++q) *reinterpret_cast<f32x4*>(sm + lane * TM + 4 * q) = f32x4{m[4 * q], m[4 * q + 1], m[4 * q + 2], m[4 * q + 3]};
        sgwe[lane] = g_w;
    }

    // coors_mlp backward; g_m accumulates W3^T g_hid
    float gm[TM];
#pragma unroll
    for (int c = 0; c < TM; ++c) gm[c] = pm ? p.g_msum[ig * TM + c] : 0.f;
    const int64_t e_wave = (int64_t)blockIdx.x * 256 + wave * 64;              // first edge of this wave
#pragma unroll 1
    for (int t0 = 0; t0 < TH; t0 += 4) {
        f32x4 ghv, a3v;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int t = t0 + q;
            float h = sb3[t];
#pragma unroll
            for (int c = 0; c < TM; ++c) h = __builtin_fmaf(sW3[t * TM + c], m[c], h);
            const float sg = egnn_sigmoid(h);
            a3v[q] = h * sg;
            const float gh = g_w * sW4[t] * (sg * (1.0f + h * (1.0f - sg)));
            ghv[q] = gh;
#pragma unroll
            for (int c = 0; c < TM; ++c) gm[c] = __builtin_fmaf(sW3[t * TM + c], gh, gm[c]);
        }
        if constexpr (REDUCE) {
            const int q4 = (t0 >> 2) & 3;                                         // 4 iterations fill a block of 16 hidden columns
            *reinterpret_cast<f32x4*>(sblk + lane * RLD + 4 * q4) = ghv;
            *reinterpret_cast<f32x4*>(sblk + 64 * RLD + lane * RLD + 4 * q4) = a3v;
            if (q4 == 3) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
                // lane (tt, cg): d/d W3[t][4 cg .. 4 cg + 3], and the column sums d/d b3[t], d/d W4[t], over the wave's 64 edges in order
                const int tt = lane & 15, cg = lane >> 4, tb = t0 >> 4;
                f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
                float accb = 0.f, accw = 0.f;
#pragma unroll 4
                for (int ee = 0; ee < 64; ++ee) {
                    const float gh = sblk[ee * RLD + tt];
                    const float a3 = sblk[64 * RLD + ee * RLD + tt];
                    const f32x4 m4 = *reinterpret_cast<const f32x4*>(sm + ee * TM + 4 * cg);
                    acc += m4 * gh;
                    accb += gh;
                    accw = __builtin_fmaf(sgwe[ee], a3, accw);
                }
                *reinterpret_cast<f32x4*>(wpart + (16 * tb + tt) * TM + 4 * cg) = acc;
                if (cg == 0) wpart[TH * TM + 16 * tb + tt] = accb;
                if (cg == 1) wpart[TH * TM + TH + 16 * tb + tt] = accw;
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
            }
            continue;
        }
        const int part = (t0 >> 2) & 7;                                          // 8 iterations fill 32 columns = one line per row
        *reinterpret_cast<f32x4*>(&stage[wave][0][lane][4 * part]) = ghv;
        *reinterpret_cast<f32x4*>(&stage[wave][1][lane][4 * part]) = a3v;
        if (part == 7) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            const int col0 = t0 - 28;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int row = (lane >> 3) + 8 * j;
                if (e_wave + row < E) {
                    const size_t o = (size_t)(e_wave + row) * TH + col0 + 4 * (lane & 7);
                    *reinterpret_cast<f32x4*>(p.g_hid + o) = *reinterpret_cast<const f32x4*>(&stage[wave][0][row][4 * (lane & 7)]);
                    *reinterpret_cast<f32x4*>(p.a3 + o) = *reinterpret_cast<const f32x4*>(&stage[wave][1][row][4 * (lane & 7)]);
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
        }
    }
    float gs = 0.f;
    if (p.gate_w) {
        // through the gate: m = m0 gt  =>  d/d m0 = gm gt + (gm . m0) gt (1 - gt) gate_w,   d/d (gate pre-activation) = (gm . m0) gt (1 - gt)
        float dot = 0.f;
#pragma unroll
        for (int c = 0; c < TM; ++c) dot = __builtin_fmaf(gm[c], u[c] * sgu[c], dot);
        gs = dot * gt * gtc;
#pragma unroll
        for (int c = 0; c < TM; ++c) gm[c] = __builtin_fmaf(gs, sgw[c], gm[c] * gt);
        if (!REDUCE && live) p.g_gate[e] = gs;
    }
    f32x4* gup = reinterpret_cast<f32x4*>(p.gU + e * TM);
    uint32_t gu_max = 0u;                                                      // max |gU| over this lane's edge (dead lanes: all zero)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        f32x4 v;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int cc = 4 * q + c;
            v[c] = gm[cc] * (sgu[cc] * (1.0f + u[cc] * (1.0f - sgu[cc])));
        }
        if (live) gup[q] = v;
#pragma unroll
        for (int c = 0; c < 4; ++c) { const uint32_t tb = egnn_abs_bits(v[c]); gu_max = gu_max > tb ? gu_max : tb; }
        if constexpr (REDUCE) {
            // per-edge scalar terms, summed over the wave's edges below: [0, 16) d loss / d u (-> edge_mlp's last bias),
            // [16, 32) gate term x SiLU(u) (-> d/d gate weight), 32 g_w (-> d/d b4), 33 CoorsNorm.scale term, 34 gate term (-> d/d gate bias)
            *reinterpret_cast<f32x4*>(sblk + lane * QLD + 4 * q) = v;
            *reinterpret_cast<f32x4*>(sblk + lane * QLD + 16 + 4 * q) =
                f32x4{gs * (u[4 * q] * sgu[4 * q]), gs * (u[4 * q + 1] * sgu[4 * q + 1]), gs * (u[4 * q + 2] * sgu[4 * q + 2]), gs * (u[4 * q + 3] * sgu[4 * q + 3])};
        }
    }
    if constexpr (REDUCE) {
        *reinterpret_cast<f32x4*>(sblk + lane * QLD + 32) = f32x4{g_w, gsc_e, gs, 0.f};
        *reinterpret_cast<f32x4*>(sblk + lane * QLD + 36) = f32x4{0.f, 0.f, 0.f, 0.f};
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        if (lane < QLD) {
            float acc = 0.f;
#pragma unroll 8
            for (int ee = 0; ee < 64; ++ee) acc += sblk[ee * QLD + lane];
            wpart[TH * TM + 2 * TH + lane] = acc;
        }
    }
    if (p.amax_gu) {
        __shared__ uint32_t amax_slot;
        egnn_block_absmax_commit(gu_max, &amax_slot, p.amax_gu);
    }
}

// Pooled messages for the backward's node-level part: m_sum[node] = sum over the node's K edges of pair_mask * SiLU(u) * gate
// (egnn_pytorch.py:287-290, :319-326) -- 16 lanes (channels) per node, the edge gate's dot product as a DPP row sum.
__global__ __launch_bounds__(256) void edge_pool_kernel(const float* __restrict__ u, const float* __restrict__ gate_w, const float* __restrict__ gate_b,
                                                        const uint8_t* __restrict__ pair_mask, int64_t nodes, int K, float* __restrict__ m_sum)
{
    const int c = threadIdx.x & 15;
    const int64_t node_raw = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
    const int64_t node = node_raw < nodes ? node_raw : nodes - 1;
    const float gw = gate_w ? gate_w[c] : 0.f;
    const float gb = gate_w ? gate_b[0] : 0.f;
    float acc = 0.f;
    for (int k = 0; k < K; ++k) {
        const int64_t e = node * K + k;
        float m = egnn_silu(u[e * TM + c]);
        if (gate_w) m *= egnn_sigmoid(egnn_row16_sum(gw * m) + gb);
        if (pair_mask && !pair_mask[e]) m = 0.f;
        acc += m;
    }
    if (node_raw < nodes) m_sum[node * TM + c] = acc;
}

// ---------------------------------------------------------------------------------------------------------------------------
// The REDUCE variant with coors_mlp on the matrix cores.  In the kernel above every lane owns an edge and walks coors_mlp's 64 x 16
// weights three times with scalar FMAs (forward, recompute, backward: ~3100 of its ~6100 VALU instructions per wave).  Here a wave
// takes its 64 edges as four tiles of 16 with the EDGES as the N dimension of v_mfma_f32_16x16x16_f16 (the forward kernel's
// transposed formulation): lane (g, e) holds channels 4g .. 4g+3 of edge e --
//     hid^T (64 x 16 e) = W3 (64 x 16) m^T          A = W3 rows as split-f16 fragments (LDS), B = the lane's 4 messages, split f16 x 3
//     gm^T  (16 x 16 e) = W3^T (16 x 64) q^T        q = W4 SiLU'(hid), O(1); its D layout IS the B layout of the second product;
//                                                   the edge's g_w is a per-column factor applied to the fp32 result
// -- and the per-edge scalar section runs once per tile (replicated over the four lane groups).  The sums over edges (REDUCE) take
// each tile's g_hid / a3 / m through LDS: lane t owns row t of d/d W3 over the wave's edges.  Same outputs and `part` layout as
// edge_tail_bwd_kernel<true>; the summation order differs (tests compare both with the float64 specification).
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void tail_split4(const f32x4 v, h16x4& hi, h16x4& lo)
{
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const _Float16 h = (_Float16)v[u];
        hi[u] = h;
        lo[u] = (_Float16)(v[u] - (float)h);
    }
}

constexpr int GLD = 68;      // floats per row of a tile's g_hid / a3 staging (64 columns + 4: rows 16 banks apart for the 16-byte writes)

// DROP: training-mode dropout in coors_mlp (its own instantiation: the keep bits and the hash cost the plain one a wave per SIMD)
template <bool DROP>
__global__ __launch_bounds__(256) void edge_tail_mfma_kernel(const egnn_edge_tail_args p)
{
    __shared__ uint2 fragA[4][2][64];            // W3 rows: [t block][hi | lo][lane (g, m)] = W3[16 tb + m][4g .. 4g+3] x s3
    __shared__ uint2 fragB[4][2][64];            // W3^T:    [t block][hi | lo][lane (g, m)] = W3[16 tb + 4g .. +3][m] x s3
    __shared__ float sb3[TH], sW4[TH], sgw[TM];
    __shared__ uint32_t w3max;
    __shared__ __attribute__((aligned(16))) float stage_raw[4 * (2 * 16 * GLD + 16 * TM + 16)];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, ee = lane & 15;
    // ---- weights: power-of-two scale from max |W3| (hi / lo halves in fp16's normal range), fragments, biases
    if (tid == 0) w3max = 0u;
    __syncthreads();
    {
        uint32_t m = 0u;
        for (int o = tid; o < TH * TM; o += 256) { const uint32_t t = egnn_abs_bits(p.W3[o]); m = m > t ? m : t; }
        if (m) atomicMax(&w3max, m);
    }
    if (tid < TM) sgw[tid] = p.gate_w ? p.gate_w[tid] : 0.f;
    if (tid < TH) { sb3[tid] = p.b3[tid]; sW4[tid] = p.W4[tid]; }
    __syncthreads();
    const uint32_t ebits = w3max & 0x7f800000u;                                 // exponent field of max |W3|
    const float s3 = (ebits && ebits < 0x7e800000u) ? __uint_as_float(0x7f000000u - ebits) : 1.0f;       // 2^-floor(log2 max): max |W3| s3 in [1, 2)
    const float inv_s3 = 1.0f / s3;
    {
        const int tb = tid >> 6, l = tid & 63, lg = l >> 4, lm = l & 15;
        f32x4 a, b;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            a[r] = p.W3[(16 * tb + lm) * TM + 4 * lg + r] * s3;
            b[r] = p.W3[(16 * tb + 4 * lg + r) * TM + lm] * s3;
        }
        h16x4 hi, lo;
        tail_split4(a, hi, lo);
        fragA[tb][0][l] = __builtin_bit_cast(uint2, hi); fragA[tb][1][l] = __builtin_bit_cast(uint2, lo);
        tail_split4(b, hi, lo);
        fragB[tb][0][l] = __builtin_bit_cast(uint2, hi); fragB[tb][1][l] = __builtin_bit_cast(uint2, lo);
    }
    __syncthreads();

    float* sgh = stage_raw + wave * (2 * 16 * GLD + 16 * TM + 16);            // [16 e][GLD] g_hid of the tile
    float* sa3 = sgh + 16 * GLD;                                              // [16 e][GLD] a3
    float* smm = sa3 + 16 * GLD;                                              // [16 e][16]  post-gate messages
    float* sgwe = smm + 16 * TM;                                              // [16] g_w
    const int64_t E = (int64_t)p.B * p.N * p.K;
    const int K = p.K, N = p.N;
    float* wpart = p.part + ((size_t)blockIdx.x * 4 + wave) * PART;
    const float gb = p.gate_w ? p.gate_b[0] : 0.f;
    const float b4 = p.b4[0];
    const float cscale = p.norm_coors ? p.scale[0] : 1.f;

    float acc3[TM];                             // lane t = lane: d/d W3[t][0..15] over the wave's 64 edges
#pragma unroll
    for (int c = 0; c < TM; ++c) acc3[c] = 0.f;
    float accb = 0.f, accw = 0.f;               // d/d b3[t], d/d W4[t]
    f32x4 s_gu = f32x4{0.f, 0.f, 0.f, 0.f}, s_gate = f32x4{0.f, 0.f, 0.f, 0.f};     // per (g, ee): sums over the four tiles of gU[4g+r], gs m0[4g+r]
    float s_gw = 0.f, s_sc = 0.f, s_gs = 0.f;
    uint32_t gu_max = 0u;

#pragma unroll 1
    for (int tt = 0; tt < 4; ++tt) {
        const int64_t e_raw = (int64_t)blockIdx.x * 256 + wave * 64 + 16 * tt + ee;
        const bool live = e_raw < E;
        const int64_t e = live ? e_raw : E - 1;
        const int64_t ig = e / K;
        const int64_t jg = p.idx ? (ig / N) * N + p.idx[e] : (ig / N) * N + (e - ig * K);
        const bool pm = live && (p.pair_mask ? p.pair_mask[e] != 0 : true);
        const f32x4 u4 = *reinterpret_cast<const f32x4*>(p.u + e * TM + 4 * g);
        f32x4 sgu, m0, m4;
#pragma unroll
        for (int r = 0; r < 4; ++r) { sgu[r] = egnn_sigmoid(u4[r]); m0[r] = u4[r] * sgu[r]; }
        float gt = 1.f, gtc = 0.f;
        m4 = m0;
        if (p.gate_w) {
            float ps = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) ps = __builtin_fmaf(sgw[4 * g + r], m0[r], ps);
            const float sgate = egnn_column_sum4_reg(ps) + gb;
            gt = egnn_sigmoid(sgate);
            gtc = egnn_sigmoid(-sgate);
            m4 = m0 * gt;
        }
        h16x4 mh, ml;
        tail_split4(m4, mh, ml);
        // ---- coors_mlp forward: hid^T, a3, q = W4 SiLU'(hid); w = W4 . a3 + b4
        // (training-mode dropout behind coors_mlp's first Linear, egnn_pytorch.py:203-208: the forward's hash mask -- row = global edge
        // id, column = hidden unit -- re-evaluated here; the keep bits of the lane's 16 units are reused by the backward below)
        float hid[4][4], sgh4[4][4];
        float wpartial = 0.f;
        uint32_t kbits = 0xffffu;
        const uint32_t ckey = egnn_drop_base(p.drop_seed, EGNN_DROP_SITE_COORS, (uint32_t)(e + p.drop_eid0));
        if constexpr (DROP) kbits = 0u;
#pragma unroll
        for (int tb = 0; tb < 4; ++tb) {
            const h16x4 ah = __builtin_bit_cast(h16x4, fragA[tb][0][lane]), al = __builtin_bit_cast(h16x4, fragA[tb][1][lane]);
            f32x4 d = f32x4{0.f, 0.f, 0.f, 0.f};
            d = __builtin_amdgcn_mfma_f32_16x16x16f16(ah, mh, d, 0, 0, 0);
            d = __builtin_amdgcn_mfma_f32_16x16x16f16(ah, ml, d, 0, 0, 0);
            d = __builtin_amdgcn_mfma_f32_16x16x16f16(al, mh, d, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int t = 16 * tb + 4 * g + r;
                float h = __builtin_fmaf(d[r], inv_s3, sb3[t]);
                if constexpr (DROP) {
                    const bool keep = egnn_drop_hash(ckey, (uint32_t)t) >= p.drop_thr;
                    h = keep ? h * p.drop_inv_keep : 0.f;
                    kbits |= keep ? (1u << (4 * tb + r)) : 0u;
                }
                const float sg = egnn_sigmoid(h);
                hid[tb][r] = h;
                sgh4[tb][r] = sg;
                wpartial = __builtin_fmaf(sW4[t], h * sg, wpartial);
            }
        }
        const float w = egnn_column_sum4_reg(wpartial) + b4;
        // ---- the edge's scalar section (every lane group computes its edge's copy)
        float rel[3], relp[3], gco[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            rel[d] = p.coors[ig * 3 + d] - p.coors[jg * 3 + d];
            gco[d] = p.g_coors_out[ig * 3 + d];
        }
        float rn = 0.f, den = 1.f;
        if (p.norm_coors) {
            rn = sqrtf(rel[0] * rel[0] + rel[1] * rel[1] + rel[2] * rel[2]);
            den = fmaxf(rn, p.eps);
#pragma unroll
            for (int d = 0; d < 3; ++d) relp[d] = rel[d] / den * cscale;
        } else {
#pragma unroll
            for (int d = 0; d < 3; ++d) relp[d] = rel[d];
        }
        const float wm = pm ? w : 0.f;
        const bool clamped = p.clamp >= 0.f && !(wm >= -p.clamp && wm <= p.clamp);
        const float wc = p.clamp >= 0.f ? fminf(fmaxf(wm, -p.clamp), p.clamp) : wm;
        const float g_wc = gco[0] * relp[0] + gco[1] * relp[1] + gco[2] * relp[2];
        const float g_w = (pm && !clamped) ? g_wc : 0.f;
        float g_relp[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) g_relp[d] = wc * gco[d];
        f32x4 grel;
        float gsc_e = 0.f;
        if (p.norm_coors) {
            const float dot = g_relp[0] * rel[0] + g_relp[1] * rel[1] + g_relp[2] * rel[2];
            gsc_e = dot / den;
            const float k1 = cscale / den;
            const float k2 = rn >= p.eps ? dot * cscale / (den * den * fmaxf(rn, 1e-30f)) : 0.f;
#pragma unroll
            for (int d = 0; d < 3; ++d) grel[d] = g_relp[d] * k1 - k2 * rel[d];
        } else {
#pragma unroll
            for (int d = 0; d < 3; ++d) grel[d] = g_relp[d];
        }
        grel[3] = 0.f;
        if (jg == ig) grel = f32x4{0.f, 0.f, 0.f, 0.f};                          // (self pair: see edge_tail_bwd_kernel)
        if (live && g == 0) {
            *reinterpret_cast<f32x4*>(p.g_rel + e * 4) = grel;
            if (p.rel_out) {
                *reinterpret_cast<f32x4*>(p.rel_out + e * 4) = f32x4{rel[0], rel[1], rel[2], 0.f};
                p.dist_out[e] = (rel[0] * rel[0] + rel[1] * rel[1]) + rel[2] * rel[2];
            }
        }
        // ---- coors_mlp backward: gm^T = g_w W3^T q^T; the tile's g_hid / a3 / m / g_w go to LDS for the sums over edges
        f32x4 gmacc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int tb = 0; tb < 4; ++tb) {
            f32x4 q, ghv, a3v;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float h = hid[tb][r], sg = sgh4[tb][r];
                q[r] = sW4[16 * tb + 4 * g + r] * (sg * (1.0f + h * (1.0f - sg)));
                if constexpr (DROP) q[r] = ((kbits >> (4 * tb + r)) & 1u) ? q[r] * p.drop_inv_keep : 0.f;       // d (dropped hid) / d hid
                ghv[r] = g_w * q[r];
                a3v[r] = h * sg;
            }
            *reinterpret_cast<f32x4*>(sgh + ee * GLD + 16 * tb + 4 * g) = ghv;
            *reinterpret_cast<f32x4*>(sa3 + ee * GLD + 16 * tb + 4 * g) = a3v;
            h16x4 qh, ql;
            tail_split4(q, qh, ql);
            const h16x4 bh = __builtin_bit_cast(h16x4, fragB[tb][0][lane]), bl = __builtin_bit_cast(h16x4, fragB[tb][1][lane]);
            gmacc = __builtin_amdgcn_mfma_f32_16x16x16f16(bh, qh, gmacc, 0, 0, 0);
            gmacc = __builtin_amdgcn_mfma_f32_16x16x16f16(bh, ql, gmacc, 0, 0, 0);
            gmacc = __builtin_amdgcn_mfma_f32_16x16x16f16(bl, qh, gmacc, 0, 0, 0);
        }
        *reinterpret_cast<f32x4*>(smm + ee * TM + 4 * g) = m4;
        if (g == 0) sgwe[ee] = g_w;
        f32x4 gm;
        const float gk = g_w * inv_s3;
#pragma unroll
        for (int r = 0; r < 4; ++r) gm[r] = __builtin_fmaf(gmacc[r], gk, pm ? p.g_msum[ig * TM + 4 * g + r] : 0.f);
        float gs = 0.f;
        if (p.gate_w) {
            float dp = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) dp = __builtin_fmaf(gm[r], m0[r], dp);
            gs = egnn_column_sum4_reg(dp) * gt * gtc;
#pragma unroll
            for (int r = 0; r < 4; ++r) gm[r] = __builtin_fmaf(gs, sgw[4 * g + r], gm[r] * gt);
        }
        f32x4 gu;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            gu[r] = gm[r] * (sgu[r] * (1.0f + u4[r] * (1.0f - sgu[r])));
            const uint32_t tbits = egnn_abs_bits(gu[r]);
            gu_max = gu_max > tbits ? gu_max : tbits;
        }
        if (live) *reinterpret_cast<f32x4*>(p.gU + e * TM + 4 * g) = gu;
        s_gu += gu;
        s_gate += m0 * gs;
        if (g == 0) { s_gw += g_w; s_sc += gsc_e; s_gs += gs; }
        // ---- sums over the tile's 16 edges: lane t = lane owns row t of d/d W3, d/d b3[t], d/d W4[t]
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
#pragma unroll 4
        for (int x = 0; x < 16; ++x) {
            const float gh = sgh[x * GLD + lane];
            accb += gh;
            accw = __builtin_fmaf(sgwe[x], sa3[x * GLD + lane], accw);
#pragma unroll
            for (int c4 = 0; c4 < 4; ++c4) {
                const f32x4 mv = *reinterpret_cast<const f32x4*>(smm + x * TM + 4 * c4);
#pragma unroll
                for (int r = 0; r < 4; ++r) acc3[4 * c4 + r] = __builtin_fmaf(gh, mv[r], acc3[4 * c4 + r]);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
    }
    // ---- this wave's row of partial sums
#pragma unroll
    for (int c4 = 0; c4 < 4; ++c4)
        *reinterpret_cast<f32x4*>(wpart + lane * TM + 4 * c4) = f32x4{acc3[4 * c4], acc3[4 * c4 + 1], acc3[4 * c4 + 2], acc3[4 * c4 + 3]};
    wpart[TH * TM + lane] = accb;
    wpart[TH * TM + TH + lane] = accw;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float a = egnn_row16_sum(s_gu[r]), b = egnn_row16_sum(s_gate[r]);
        if (ee == 0) { wpart[TH * TM + 2 * TH + 4 * g + r] = a; wpart[TH * TM + 2 * TH + 16 + 4 * g + r] = b; }
    }
    {
        const float a = egnn_row16_sum(s_gw), b = egnn_row16_sum(s_sc), c = egnn_row16_sum(s_gs);
        if (lane == 0) {
            wpart[TH * TM + 2 * TH + 32] = a; wpart[TH * TM + 2 * TH + 33] = b; wpart[TH * TM + 2 * TH + 34] = c;
#pragma unroll
            for (int o = 35; o < 40; ++o) wpart[TH * TM + 2 * TH + o] = 0.f;
        }
    }
    if (p.amax_gu) {
        __shared__ uint32_t amax_slot;
        egnn_block_absmax_commit(gu_max, &amax_slot, p.amax_gu);
    }
}

}  // namespace

extern "C" int egnn_edge_tail_bwd_f32(const egnn_edge_tail_args* args, void* stream)
{
    if (!args) return EGNN_E_NULLPTR;
    const egnn_edge_tail_args& a = *args;
    if (!a.u || !a.coors || !a.g_coors_out || !a.g_msum || !a.W3 || !a.b3 || !a.W4 || !a.b4 || !a.gU || !a.g_rel) return EGNN_E_NULLPTR;
    if (!a.part && (!a.g_hid || !a.a3 || !a.g_w)) return EGNN_E_NULLPTR;
    if (a.norm_coors && (!a.scale || (!a.part && !a.g_scale))) return EGNN_E_NULLPTR;
    if (a.gate_w && (!a.gate_b || (!a.part && !a.g_gate))) return EGNN_E_NULLPTR;
    if ((a.rel_out == nullptr) != (a.dist_out == nullptr)) return EGNN_E_NULLPTR;
    if (a.drop_thr && !(a.drop_inv_keep >= 1.f)) return EGNN_E_SHAPE;
    if (a.B <= 0 || a.N <= 0 || a.K <= 0) return EGNN_E_SHAPE;
    if (a.idx == nullptr && a.K != a.N) return EGNN_E_SHAPE;
    if ((reinterpret_cast<uintptr_t>(a.u) & 15) || (reinterpret_cast<uintptr_t>(a.gU) & 15) || (reinterpret_cast<uintptr_t>(a.g_rel) & 15) ||
        (reinterpret_cast<uintptr_t>(a.g_hid) & 15) || (reinterpret_cast<uintptr_t>(a.a3) & 15) || (reinterpret_cast<uintptr_t>(a.part) & 15) ||
        (reinterpret_cast<uintptr_t>(a.rel_out) & 15))
        return EGNN_E_ALIGN;
    const int64_t E = (int64_t)a.B * a.N * a.K;
    const int64_t blocks = (E + 255) / 256;
    if (blocks >= ((int64_t)1 << 31)) return EGNN_E_SHAPE;
    if (a.amax_gu && hipMemsetAsync(a.amax_gu, 0, sizeof(uint32_t), static_cast<hipStream_t>(stream)) != hipSuccess) return (int)hipGetLastError();
    // (EGNN_TAIL_SCALAR=1: the reduce variant with coors_mlp as per-lane FMAs, for A/B and debugging)
    static const bool scalar_tail = [] { const char* v = getenv("EGNN_TAIL_SCALAR"); return v && v[0] == '1'; }();
    if (a.drop_thr && (!a.part || scalar_tail)) return EGNN_E_UNSUPPORTED;       // (dropout: the matrix-core variant only)
    if (a.part && !scalar_tail && a.drop_thr) hipLaunchKernelGGL(edge_tail_mfma_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    else if (a.part && !scalar_tail) hipLaunchKernelGGL(edge_tail_mfma_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    else if (a.part) hipLaunchKernelGGL(edge_tail_bwd_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    else hipLaunchKernelGGL(edge_tail_bwd_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    return egnn_launch_status();
}

extern "C" int egnn_edge_tail_part_floats(void) { return PART; }

extern "C" int egnn_edge_pool_f32(const float* u, const float* gate_w, const float* gate_b, const uint8_t* pair_mask, int B, int N, int K,
                                  float* m_sum, void* stream)
{
    if (!u || !m_sum || (gate_w && !gate_b)) return EGNN_E_NULLPTR;
    if (B <= 0 || N <= 0 || K <= 0) return EGNN_E_SHAPE;
    const int64_t nodes = (int64_t)B * N;
    const int64_t blocks = (nodes + 15) / 16;
    if (blocks >= ((int64_t)1 << 31)) return EGNN_E_SHAPE;
    hipLaunchKernelGGL(edge_pool_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), u, gate_w, gate_b, pair_mask, nodes, K, m_sum);
    return egnn_launch_status();
}
