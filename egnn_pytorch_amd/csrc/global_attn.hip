// The two attention cores of EGNN_Network's induced-set ("global linear") attention block (reference:
// egnn_pytorch/egnn_pytorch.py:83-113 Attention, :115-144 GlobalLinearAttention).
//
//   attn1: T global tokens (4 by default) attend over the N (masked) nodes of their graph   -> egnn_induced_attn_f32
//   attn2: every node attends over the T induced tokens                                      -> egnn_token_attn_f32
//
// The projections around them (to_q / to_kv / to_out, the feed-forward) are dense GEMMs over the nodes and run on
// egnn_linear_hl_f32; what is left here is O(N T heads dim_head) per graph: one pass over the K / V projections of the nodes
// (attn1, HBM-bound: each row is read once) and one over their Q projections (attn2).  fp32 throughout; softmax with the
// reference's masking convention (masked logits = -FLT_MAX, so a graph whose mask is all False gets a uniform distribution
// and finite outputs, :102-107).
#include "egnn_common.h"
#include <float.h>

namespace {

constexpr int TMAX = 8;            // global tokens per graph
constexpr int DPL_MAX = 4;         // dim_head <= 256: floats per lane

// ---- attn1: one workgroup per (graph, head); each wave streams a quarter of the nodes, lane l owns dims l, l+64, ...
template <int DPL>
__global__ __launch_bounds__(256) void induced_attn_kernel(const float* __restrict__ q, const float* __restrict__ kv, int64_t ldkv,
                                                           const uint8_t* __restrict__ mask, int N, int T, int heads, int dh, float scale,
                                                           float* __restrict__ out)
{
    __shared__ float sm[4][TMAX][2];                    // per wave: running max, running sum
    __shared__ float so[4][TMAX][64 * DPL_MAX];         // per wave: un-normalised output
    const int b = blockIdx.x / heads, h = blockIdx.x % heads;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int inner = heads * dh;
    float qr[TMAX][DPL];
    for (int t = 0; t < TMAX; ++t)
#pragma unroll
        for (int u = 0; u < DPL; ++u) {
            const int d = lane + 64 * u;
            qr[t][u] = (t < T && d < dh) ? q[((size_t)b * T + t) * inner + h * dh + d] * scale : 0.f;
        }
    float m[TMAX], l[TMAX], o[TMAX][DPL];
    for (int t = 0; t < TMAX; ++t) {
        m[t] = -FLT_MAX; l[t] = 0.f;
#pragma unroll
        for (int u = 0; u < DPL; ++u) o[t][u] = 0.f;
    }
    for (int n = wave; n < N; n += 4) {
        const float* row = kv + ((size_t)b * N + n) * ldkv + h * dh;
        float kr[DPL], vr[DPL];
#pragma unroll
        for (int u = 0; u < DPL; ++u) {
            const int d = lane + 64 * u;
            kr[u] = d < dh ? row[d] : 0.f;
            vr[u] = d < dh ? row[inner + d] : 0.f;
        }
        const bool keep = mask ? mask[(size_t)b * N + n] != 0 : true;
        for (int t = 0; t < T; ++t) {
            float s = 0.f;
#pragma unroll
            for (int u = 0; u < DPL; ++u) s += qr[t][u] * kr[u];
            s = egnn_wave_sum(s);
            if (!keep) s = -FLT_MAX;                                   // masked_fill_(~mask, -finfo.max), :102-105
            const float mn = fmaxf(m[t], s);
            const float corr = __expf(m[t] - mn), p = __expf(s - mn);
            l[t] = l[t] * corr + p;
#pragma unroll
            for (int u = 0; u < DPL; ++u) o[t][u] = o[t][u] * corr + p * vr[u];
            m[t] = mn;
        }
    }
    for (int t = 0; t < T; ++t) {
        if (lane == 0) { sm[wave][t][0] = m[t]; sm[wave][t][1] = l[t]; }
#pragma unroll
        for (int u = 0; u < DPL; ++u) so[wave][t][lane + 64 * u] = o[t][u];
    }
    __syncthreads();
    // merge the four partial softmaxes (fixed order) and normalise
    for (int idx = threadIdx.x; idx < T * dh; idx += 256) {
        const int t = idx / dh, d = idx - t * dh;
        float mm = -FLT_MAX;
        for (int w = 0; w < 4; ++w) mm = fmaxf(mm, sm[w][t][0]);
        float num = 0.f, den = 0.f;
        for (int w = 0; w < 4; ++w) {
            const float c = __expf(sm[w][t][0] - mm);
            num += so[w][t][d] * c;
            den += sm[w][t][1] * c;
        }
        out[((size_t)b * T + t) * inner + h * dh + d] = num / den;
    }
}

// ---- attn2: one wave per (node, head): T logits, softmax over T, weighted sum of the T value rows
template <int DPL>
__global__ __launch_bounds__(256) void token_attn_kernel(const float* __restrict__ q, int64_t ldq, const float* __restrict__ kv_tok,
                                                         int N, int T, int heads, int dh, float scale, int64_t rows,
                                                         float* __restrict__ out, int64_t ldo)
{
    const int lane = threadIdx.x & 63;
    const int inner = heads * dh;
    const int64_t w0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (int64_t)gridDim.x * 4;
    for (int64_t item = w0; item < rows * heads; item += nw) {
        const int64_t r = item / heads;
        const int h = (int)(item - r * heads);
        const int64_t b = r / N;
        float qr[DPL];
#pragma unroll
        for (int u = 0; u < DPL; ++u) {
            const int d = lane + 64 * u;
            qr[u] = d < dh ? q[r * ldq + h * dh + d] * scale : 0.f;
        }
        float s[TMAX], mx = -FLT_MAX;
        for (int t = 0; t < T; ++t) {
            const float* krow = kv_tok + ((size_t)b * T + t) * 2 * inner + h * dh;
            float acc = 0.f;
#pragma unroll
            for (int u = 0; u < DPL; ++u) { const int d = lane + 64 * u; acc += d < dh ? qr[u] * krow[d] : 0.f; }
            s[t] = egnn_wave_sum(acc);
            mx = fmaxf(mx, s[t]);
        }
        float den = 0.f;
        for (int t = 0; t < T; ++t) { s[t] = __expf(s[t] - mx); den += s[t]; }
        const float inv = 1.0f / den;
#pragma unroll
        for (int u = 0; u < DPL; ++u) {
            const int d = lane + 64 * u;
            if (d >= dh) continue;
            float acc = 0.f;
            for (int t = 0; t < T; ++t) acc += s[t] * kv_tok[((size_t)b * T + t) * 2 * inner + inner + h * dh + d];
            out[r * ldo + h * dh + d] = acc * inv;
        }
    }
}

}  // namespace

extern "C" int egnn_induced_attn_f32(const float* q, const float* kv, int64_t ldkv, const uint8_t* mask, int B, int N, int T, int heads,
                                     int dim_head, float scale, float* out, void* stream)
{
    if (!q || !kv || !out) return EGNN_E_NULLPTR;
    if (B <= 0 || N <= 0 || T < 1 || heads < 1 || dim_head < 1 || ldkv < 2 * (int64_t)heads * dim_head) return EGNN_E_SHAPE;
    if (T > TMAX || dim_head > 64 * DPL_MAX) return EGNN_E_UNSUPPORTED;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const dim3 grid((unsigned)(B * heads)), block(256);
    if (dim_head <= 64) hipLaunchKernelGGL(induced_attn_kernel<1>, grid, block, 0, s, q, kv, ldkv, mask, N, T, heads, dim_head, scale, out);
    else if (dim_head <= 128) hipLaunchKernelGGL(induced_attn_kernel<2>, grid, block, 0, s, q, kv, ldkv, mask, N, T, heads, dim_head, scale, out);
    else hipLaunchKernelGGL(induced_attn_kernel<4>, grid, block, 0, s, q, kv, ldkv, mask, N, T, heads, dim_head, scale, out);
    return egnn_launch_status();
}

extern "C" int egnn_token_attn_f32(const float* q, int64_t ldq, const float* kv_tok, int B, int N, int T, int heads, int dim_head,
                                   float scale, float* out, int64_t ldo, void* stream)
{
    if (!q || !kv_tok || !out) return EGNN_E_NULLPTR;
    const int64_t inner = (int64_t)heads * dim_head;
    if (B <= 0 || N <= 0 || T < 1 || heads < 1 || dim_head < 1 || ldq < inner || ldo < inner) return EGNN_E_SHAPE;
    if (T > TMAX || dim_head > 64 * DPL_MAX) return EGNN_E_UNSUPPORTED;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int64_t rows = (int64_t)B * N;
    int64_t blocks = (rows * heads + 3) / 4;
    if (blocks > 65536) blocks = 65536;
    const dim3 grid((unsigned)blocks), block(256);
    if (dim_head <= 64) hipLaunchKernelGGL(token_attn_kernel<1>, grid, block, 0, s, q, ldq, kv_tok, N, T, heads, dim_head, scale, rows, out, ldo);
    else if (dim_head <= 128) hipLaunchKernelGGL(token_attn_kernel<2>, grid, block, 0, s, q, ldq, kv_tok, N, T, heads, dim_head, scale, rows, out, ldo);
    else hipLaunchKernelGGL(token_attn_kernel<4>, grid, block, 0, s, q, ldq, kv_tok, N, T, heads, dim_head, scale, rows, out, ldo);
    return egnn_launch_status();
}
