// The E x H work of the backward in plain fp32 / float64 (autograd of egnn_pytorch/egnn_pytorch.py:277-287): the backward of the layers
// that run on csrc/edge_exact.hip -- float64 modules (what the reference's own training recipe is, denoise_sparse.py:11, 23-32), calls
// re-run on the wide-range path because their values leave the split-fp16 range, and the shapes beyond the fused kernels' compile-time
// limits (more than 16 per-edge scalars, more than 8 coordinates, heads wider than 64 channels).
//
// With z = P_i[i] + P_j[j] + W_s s (the factorised first Linear of edge_mlp), a = SiLU(z), u = W2 a + b2 and gU = d loss / d u (from the
// small per-edge tail, which the host differentiates on E x m tensors):
//     dz[e, h] = (sum_c W2[c, h] gU[e, c]) SiLU'(z[e, h])
//     d/d P_i[n]  = sum of dz over the edges LEAVING n        d/d P_j[n] = sum of dz over the edges ARRIVING at n
//     d/d W2 = gU^T a        d/d W_s = dz^T s        d/d s[e] = dz[e] W_s
//   kernel 1, one thread per edge (the forward kernel's shape: weights wave-uniform, the edge's scalars in a column of LDS): recomputes
//             z and a, writes a and dz TRANSPOSED -- (H, E): consecutive threads write consecutive addresses, and the two all-edge
//             contractions become plain C = X W^T products of egnn_linear_f32 / _f64 with the edges as the K dimension -- and d/d s;
//   kernel 2, one thread per (hidden unit, node): the node's K outgoing edges, and its incoming edges through the CSR lists of
//             egnn_dest_lists_i32, summed in a fixed order (no atomics: deterministic); both sums in both layouts, (nodes, H) for the
//             d/d feats product and (H, nodes) for the d/d W_i, W_j products.
// The a / dz tables are 2 H E elements: the host cuts the batch into chunks of graphs that keep them within its budget.  Correct first,
// like the forward of this path: arithmetic in the parameters' precision, never what bench.py times.
#include "egnn_common.h"

namespace {

constexpr int XB_THREADS = 256;

__device__ __forceinline__ float xb_exp(float x) { return expf(x); }
__device__ __forceinline__ double xb_exp(double x) { return exp(x); }
__device__ __forceinline__ float xb_sin(float x) { return sinf(x); }
__device__ __forceinline__ double xb_sin(double x) { return sin(x); }
__device__ __forceinline__ float xb_cos(float x) { return cosf(x); }
__device__ __forceinline__ double xb_cos(double x) { return cos(x); }
__device__ __forceinline__ float xb_fma(float a, float b, float c) { return fmaf(a, b, c); }
__device__ __forceinline__ double xb_fma(double a, double b, double c) { return fma(a, b, c); }
__device__ __forceinline__ float xb_sqdist(const float* ci, const float* cj, int C)
{
    if (C == 3) {
        float dx, dy, dz;
        return egnn_sqdist(ci[0], ci[1], ci[2], cj[0], cj[1], cj[2], dx, dy, dz);
    }
    return egnn_sqdist_any<float, 8>(ci, cj, C);
}
__device__ __forceinline__ double xb_sqdist(const double* ci, const double* cj, int C) { return egnn_sqdist_any<double, 4>(ci, cj, C); }

template <typename T>
struct XbArgs {
    int B, N, K, m_dim, H, fourier, edge_dim, coor_dim, edges_by_k;
    const T *Pi, *Pj, *Ws, *W2, *coors, *edges, *gU;
    int64_t ldp, ldws;
    const int32_t* idx;
    T *A_T, *DZ_T, *g_scal;
    uint32_t drop_thr, drop_seed;
    float drop_inv_keep;
    int64_t drop_eid0;
};

// MB: message channels whose gU the thread keeps in registers (16 / 32 / 64); 0: any m_dim, gU re-read per hidden unit
template <typename T, int MB>
__global__ __launch_bounds__(XB_THREADS) void edge_exact_bwd_kernel(const XbArgs<T> p)
{
    const int64_t E = (int64_t)p.B * p.N * p.K;
    const int64_t q = (int64_t)blockIdx.x * XB_THREADS + threadIdx.x;
    if (q >= E) return;
    const int N = p.N, K = p.K, C = p.coor_dim, F = p.fourier, m_dim = p.m_dim, H = p.H;
    const int S = 2 * F + 1 + p.edge_dim;
    const int64_t node = q / K;
    const int k = (int)(q - node * K);
    const int64_t bN = node / N * N;
    const int i = (int)(node - bN);
    const int j = p.idx ? p.idx[q] : k;
    const T* const ci = p.coors + (bN + i) * C;
    const T* const cj = p.coors + (bN + j) * C;
    const T d = xb_sqdist(ci, cj, C);                                    // (the forward's operation order: the same z bit for bit)
    // the edge's scalars and its d/d scalars: two columns of LDS per thread (S x 256 each, conflict-free)
    extern __shared__ __attribute__((aligned(16))) char scal_raw[];
    T* const scal = reinterpret_cast<T*>(scal_raw) + threadIdx.x;
    T* const gs = scal + (size_t)S * XB_THREADS;
    for (int f = 0; f < F; ++f) {
        const T x = d / (T)(1u << f);
        scal[f * XB_THREADS] = xb_sin(x);
        scal[(F + f) * XB_THREADS] = xb_cos(x);
    }
    scal[2 * F * XB_THREADS] = d;
    if (p.edge_dim > 0) {
        const T* ep = p.edges + (p.edges_by_k ? (size_t)q : ((size_t)(bN + i) * N + j)) * p.edge_dim;
        for (int s = 0; s < p.edge_dim; ++s) scal[(2 * F + 1 + s) * XB_THREADS] = ep[s];
    }
    for (int s = 0; s < S; ++s) gs[s * XB_THREADS] = (T)0;

    const T* pi = p.Pi + (bN + i) * p.ldp;
    const T* pj = p.Pj + (bN + j) * p.ldp;
    const T* gu = p.gU + (size_t)q * m_dim;
    constexpr int MR = MB > 0 ? MB : 1;
    T g[MR];
    if (MB > 0) {
#pragma unroll
        for (int c = 0; c < MR; ++c) g[c] = c < m_dim ? gu[c] : (T)0;
    }
    const uint32_t ekey = p.drop_thr ? egnn_drop_base(p.drop_seed, EGNN_DROP_SITE_EDGE, (uint32_t)(q + p.drop_eid0)) : 0u;
    T gd = (T)0;                                                         // S == 1: d/d (squared distance) stays in a register
    for (int h = 0; h < H; ++h) {
        T x = pi[h] + pj[h];
        const T* ws = p.Ws + (size_t)h * p.ldws;
        if (S == 1) x = xb_fma(d, ws[0], x);
        else for (int s = 0; s < S; ++s) x = xb_fma(scal[s * XB_THREADS], ws[s], x);
        T dk = (T)1;                                                     // d (dropped pre-activation) / d (pre-activation)
        if (p.drop_thr) {                                                // training-mode dropout (:178-184): the forward's mask of this row
            dk = egnn_drop_hash(ekey, (uint32_t)h) >= p.drop_thr ? (T)p.drop_inv_keep : (T)0;
            x *= dk;
        }
        const T sig = (T)1 / ((T)1 + xb_exp(-x));
        const T a = x * sig;                                             // SiLU(x), as the forward (x / (1 + exp(-x)))
        const T* w2 = p.W2 + h;
        T da = (T)0;
        if (MB > 0) {
#pragma unroll
            for (int c = 0; c < MR; ++c)
                if (c < m_dim) da = xb_fma(w2[(size_t)c * H], g[c], da);
        } else {
            for (int c = 0; c < m_dim; ++c) da = xb_fma(w2[(size_t)c * H], gu[c], da);
        }
        const T dz = da * (sig * ((T)1 + x * ((T)1 - sig))) * dk;        // SiLU'(x) = sig (1 + x (1 - sig))
        p.A_T[(size_t)h * E + q] = a;
        p.DZ_T[(size_t)h * E + q] = dz;
        if (S == 1) gd = xb_fma(dz, ws[0], gd);
        else for (int s = 0; s < S; ++s) gs[s * XB_THREADS] = xb_fma(dz, ws[s], gs[s * XB_THREADS]);
    }
    T* out = p.g_scal + (size_t)q * S;
    if (S == 1) out[0] = gd;
    else for (int s = 0; s < S; ++s) out[s] = gs[s * XB_THREADS];
}

// thread (h, n), n fastest: sums of DZ_T[h] over the node's outgoing edges (n K .. n K + K - 1) and over its incoming edges (CSR)
template <typename T>
__global__ __launch_bounds__(XB_THREADS) void edge_exact_node_sums_kernel(const T* __restrict__ dzt, int64_t E, int H, int64_t nodes, int K,
                                                                          const int64_t* __restrict__ order, const int64_t* __restrict__ seg,
                                                                          T* __restrict__ gpi, T* __restrict__ gpi_t, T* __restrict__ gpj,
                                                                          T* __restrict__ gpj_t)
{
    const int64_t o = (int64_t)blockIdx.x * XB_THREADS + threadIdx.x;
    if (o >= (int64_t)H * nodes) return;
    const int64_t h = o / nodes, n = o - h * nodes;
    const T* row = dzt + (size_t)h * E;
    T si = (T)0;
    for (int k = 0; k < K; ++k) si += row[n * K + k];                    // k order
    T sj = (T)0;
    for (int64_t t = seg[n]; t < seg[n + 1]; ++t) sj += row[order[t]];   // ascending edge id (the stable sort of egnn_dest_lists_i32)
    gpi[n * H + h] = si;
    gpi_t[h * nodes + n] = si;
    gpj[n * H + h] = sj;
    gpj_t[h * nodes + n] = sj;
}

template <typename T>
int edge_exact_bwd_launch(const egnn_edge_exact_bwd_args* args, void* stream)
{
    if (!args) return EGNN_E_NULLPTR;
    const egnn_edge_exact_bwd_args& a = *args;
    if (!a.Pi || !a.Pj || !a.Ws || !a.W2 || !a.coors || !a.gU || !a.A_T || !a.DZ_T || !a.g_scal) return EGNN_E_NULLPTR;
    if (a.B <= 0 || a.N <= 0 || a.K <= 0 || a.H <= 0 || a.ldp < a.H || a.ldws < 2 * a.fourier + 1 + a.edge_dim) return EGNN_E_SHAPE;
    if (a.m_dim < 1 || a.m_dim > 1024 || a.coor_dim < 1 || a.coor_dim > 64) return EGNN_E_UNSUPPORTED;
    if (a.fourier < 0 || a.fourier > 31 || a.edge_dim < 0) return EGNN_E_UNSUPPORTED;
    if (a.edge_dim > 0 && !a.edges) return EGNN_E_NULLPTR;
    if (!a.idx && a.K != a.N) return EGNN_E_SHAPE;
    const int S = 2 * a.fourier + 1 + a.edge_dim;
    const size_t lds = (size_t)2 * S * XB_THREADS * sizeof(T);           // the scalars and their gradients: 80 per edge in fp32, 40 in float64
    if (lds > 160 * 1024) return EGNN_E_UNSUPPORTED;
    const int64_t E = (int64_t)a.B * a.N * a.K;
    const int64_t blocks = (E + XB_THREADS - 1) / XB_THREADS;
    if (blocks > 0x7fffffffLL) return EGNN_E_UNSUPPORTED;
    XbArgs<T> p;
    p.B = a.B; p.N = a.N; p.K = a.K; p.m_dim = a.m_dim; p.H = a.H; p.fourier = a.fourier; p.edge_dim = a.edge_dim; p.coor_dim = a.coor_dim;
    p.edges_by_k = a.edges_by_k;
    p.Pi = static_cast<const T*>(a.Pi); p.Pj = static_cast<const T*>(a.Pj); p.Ws = static_cast<const T*>(a.Ws); p.W2 = static_cast<const T*>(a.W2);
    p.coors = static_cast<const T*>(a.coors); p.edges = static_cast<const T*>(a.edges); p.gU = static_cast<const T*>(a.gU);
    p.ldp = a.ldp; p.ldws = a.ldws; p.idx = a.idx;
    p.A_T = static_cast<T*>(a.A_T); p.DZ_T = static_cast<T*>(a.DZ_T); p.g_scal = static_cast<T*>(a.g_scal);
    if (a.drop_thr && (!(a.drop_inv_keep >= 1.f) || a.drop_eid0 < 0 || a.drop_eid0 + E > 0xffffffffLL)) return EGNN_E_SHAPE;
    p.drop_thr = a.drop_thr; p.drop_seed = a.drop_seed; p.drop_inv_keep = a.drop_inv_keep; p.drop_eid0 = a.drop_eid0;
    hipStream_t s = static_cast<hipStream_t>(stream);
    auto run = [&](auto kern) -> int {
        if (lds > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return (int)e;
        }
        hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(XB_THREADS), lds, s, p);
        return egnn_launch_status();
    };
    if (a.m_dim <= 16) return run(edge_exact_bwd_kernel<T, 16>);
    if (a.m_dim <= 32) return run(edge_exact_bwd_kernel<T, 32>);
    if (a.m_dim <= 64) return run(edge_exact_bwd_kernel<T, 64>);
    return run(edge_exact_bwd_kernel<T, 0>);
}

template <typename T>
int node_sums_launch(const void* DZ_T, int64_t E, int H, int64_t nodes, int K, const int64_t* csr_order, const int64_t* csr_seg,
                     void* gPi, void* gPi_T, void* gPj, void* gPj_T, void* stream)
{
    if (!DZ_T || !csr_order || !csr_seg || !gPi || !gPi_T || !gPj || !gPj_T) return EGNN_E_NULLPTR;
    if (E <= 0 || H <= 0 || nodes <= 0 || K <= 0 || nodes * K != E) return EGNN_E_SHAPE;
    const int64_t total = (int64_t)H * nodes;
    const int64_t blocks = (total + XB_THREADS - 1) / XB_THREADS;
    if (blocks > 0x7fffffffLL) return EGNN_E_UNSUPPORTED;
    hipLaunchKernelGGL(edge_exact_node_sums_kernel<T>, dim3((unsigned)blocks), dim3(XB_THREADS), 0, static_cast<hipStream_t>(stream),
                       static_cast<const T*>(DZ_T), E, H, nodes, K, csr_order, csr_seg, static_cast<T*>(gPi), static_cast<T*>(gPi_T),
                       static_cast<T*>(gPj), static_cast<T*>(gPj_T));
    return egnn_launch_status();
}

}  // namespace

extern "C" int egnn_edge_exact_bwd_f32(const egnn_edge_exact_bwd_args* args, void* stream) { return edge_exact_bwd_launch<float>(args, stream); }
extern "C" int egnn_edge_exact_bwd_f64(const egnn_edge_exact_bwd_args* args, void* stream) { return edge_exact_bwd_launch<double>(args, stream); }

extern "C" int egnn_edge_exact_node_sums_f32(const void* DZ_T, int64_t E, int H, int64_t nodes, int K, const int64_t* csr_order,
                                             const int64_t* csr_seg, void* gPi, void* gPi_T, void* gPj, void* gPj_T, void* stream)
{
    return node_sums_launch<float>(DZ_T, E, H, nodes, K, csr_order, csr_seg, gPi, gPi_T, gPj, gPj_T, stream);
}
extern "C" int egnn_edge_exact_node_sums_f64(const void* DZ_T, int64_t E, int H, int64_t nodes, int K, const int64_t* csr_order,
                                             const int64_t* csr_seg, void* gPi, void* gPi_T, void* gPj, void* gPj_T, void* stream)
{
    return node_sums_launch<double>(DZ_T, E, H, nodes, K, csr_order, csr_seg, gPi, gPi_T, gPj, gPj_T, stream);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The per-edge chain BEHIND u in closed form, any head width up to 64 channels, any coordinate dimension, fp32 or float64 (autograd of
// egnn_pytorch.py:287 second SiLU, :289-290 gate, :292-317 pair mask / coors_mlp / CoorsNorm / clamp / coordinate update, :319-333
// pooling).  csrc/edge_tail.hip is the matrix-core version of this for the standard layer (16 channels, 3-D, fp32); this one serves
// the shapes it does not: heads of 17 .. 64 channels and the other coordinate dimensions on the fast path, and everything on the plain
// fp32 / float64 path.  egnn_pytorch_amd/autograd.py::tail_edge_backward is the specification.  One thread per edge, coors_mlp's hidden
// units walked twice (its output first, then their gradients); the operands of the parameter gradients -- sums over all edges -- are
// written TRANSPOSED, (rows, E), so that the host contracts them with egnn_linear_f32 / _f64 (the edges as the K dimension).
template <typename T>
struct XtArgs {
    int B, N, K, m_dim, coor_dim, norm_coors;
    const T *u, *coors, *g_coors_out, *g_msum, *W3, *b3, *W4, *b4, *scale, *gate_w, *gate_b;
    const int32_t* idx;
    const uint8_t* pair_mask;
    T eps, clamp;
    T *gU, *g_rel, *ghid_t, *a3_t, *mm_t, *m0_t, *g_w, *g_scale, *g_gate;
    uint32_t drop_thr, drop_seed;
    float drop_inv_keep;
    int64_t drop_eid0;
};

template <typename T, int MB>
__global__ __launch_bounds__(XB_THREADS) void edge_tail_exact_bwd_kernel(const XtArgs<T> p)
{
    const int64_t E = (int64_t)p.B * p.N * p.K;
    const int64_t q = (int64_t)blockIdx.x * XB_THREADS + threadIdx.x;
    if (q >= E) return;
    const int N = p.N, K = p.K, C = p.coor_dim, m_dim = p.m_dim, hid = 4 * p.m_dim;
    const int64_t node = q / K;
    const int k = (int)(q - node * K);
    const int64_t bN = node / N * N;
    const int j = p.idx ? p.idx[q] : k;
    const bool keep = p.pair_mask ? p.pair_mask[q] != 0 : true;
    const bool has_mask = p.pair_mask != nullptr;

    // m0 = SiLU(u), the gate, m = m0 * gate (:287-290)
    // (two arrays of MB values live through the kernel -- the messages and their gradients; u and SiLU(u) are re-read / recomputed where
    // they are needed: with 64 channels in float64 even these two are the whole register file)
    const T* const urow = p.u + (size_t)q * m_dim;
    T mm[MB];
    T gsum = p.gate_w ? p.gate_b[0] : (T)0;
#pragma unroll
    for (int c = 0; c < MB; ++c) {
        const T uc = c < m_dim ? urow[c] : (T)0;
        mm[c] = uc / ((T)1 + xb_exp(-uc));                              // m0 = SiLU(u)
        if (p.gate_w && c < m_dim) gsum = xb_fma(p.gate_w[c], mm[c], gsum);
    }
    const T gt = p.gate_w ? (T)1 / ((T)1 + xb_exp(-gsum)) : (T)1;
#pragma unroll
    for (int c = 0; c < MB; ++c) {
        if (c < m_dim && p.m0_t) p.m0_t[(size_t)c * E + q] = mm[c];
        mm[c] *= gt;
        if (c < m_dim) p.mm_t[(size_t)c * E + q] = mm[c];
    }
    // g_m: d loss / d m, starting with the pooled messages' share (masked_fill of :320-322)
    T gm[MB];
#pragma unroll
    for (int c = 0; c < MB; ++c) gm[c] = (c < m_dim && p.g_msum && keep) ? p.g_msum[(size_t)node * m_dim + c] : (T)0;

    T g_w = (T)0;
    // (training-mode dropout between coors_mlp's Linear and its SiLU: the forward's hash mask of this edge's row, per hidden unit)
    const uint32_t ckey = p.drop_thr ? egnn_drop_base(p.drop_seed, EGNN_DROP_SITE_COORS, (uint32_t)(q + p.drop_eid0)) : 0u;
    if (p.W3) {
        // coors_mlp forward (:203-208): w = W4 SiLU(drop(W3 m + b3)) + b4
        T w = p.b4[0];
        for (int r = 0; r < hid; ++r) {
            T z = p.b3[r];
            const T* w3 = p.W3 + (size_t)r * m_dim;
#pragma unroll
            for (int c = 0; c < MB; ++c)
                if (c < m_dim) z = xb_fma(w3[c], mm[c], z);
            if (p.drop_thr) z = egnn_drop_hash(ckey, (uint32_t)r) >= p.drop_thr ? z * (T)p.drop_inv_keep : (T)0;
            const T sg = (T)1 / ((T)1 + xb_exp(-z));
            w = xb_fma(p.W4[r], z * sg, w);
        }
        // rel, CoorsNorm (:67-77), mask, clamp (:308-313); g = d loss / d coors_out[i]
        const T* ci = p.coors + (size_t)node * C;
        const T* cj = p.coors + (size_t)(bN + j) * C;
        const T* g = p.g_coors_out + (size_t)node * C;
        T n2 = (T)0;
        for (int c = 0; c < C; ++c) { const T r = ci[c] - cj[c]; n2 = xb_fma(r, r, n2); }
        const T rn = p.norm_coors ? (T)sqrt((double)n2) : (T)1;
        const T den = p.norm_coors ? (rn > p.eps ? rn : p.eps) : (T)1;
        const T sc = p.norm_coors ? p.scale[0] : (T)1;
        const T wm = (has_mask && !keep) ? (T)0 : w;
        const bool clamped = p.clamp >= (T)0 && (wm < -p.clamp || wm > p.clamp);
        const T wc = p.clamp >= (T)0 ? (wm < -p.clamp ? -p.clamp : (wm > p.clamp ? p.clamp : wm)) : wm;
        T g_wc = (T)0, dot = (T)0;                                      // g . relp  and  g_relp . rel
        for (int c = 0; c < C; ++c) {
            const T r = ci[c] - cj[c];
            g_wc = xb_fma(g[c], r / den * sc, g_wc);
            dot = xb_fma(wc * g[c], r, dot);
        }
        g_w = clamped ? (T)0 : g_wc;
        if (has_mask && !keep) g_w = (T)0;
        const bool self_pair = (bN + j) == node;                        // x_i - x_i is identically 0: its gradient reaches x_i once with each
        for (int c = 0; c < C; ++c) {                                   // sign -- under CoorsNorm two 1 / eps-sized terms whose fp32 sum is noise,
            const T r = ci[c] - cj[c];                                  // written as the exact zero they add up to (as csrc/edge_tail.hip does)
            T gr = wc * g[c];                                            // g_relp
            if (p.norm_coors) {
                gr = gr * (sc / den);
                if (rn >= p.eps) gr -= dot * sc / (den * den) * (r / (rn > (T)1e-30 ? rn : (T)1e-30));
            }
            p.g_rel[(size_t)c * E + q] = self_pair ? (T)0 : gr;
        }
        if (p.g_scale) p.g_scale[q] = p.norm_coors ? dot / den : (T)0;
        // coors_mlp backward: g_hid = g_w W4 SiLU'(hid);  g_m += g_hid W3
        for (int r = 0; r < hid; ++r) {
            T z = p.b3[r];
            const T* w3 = p.W3 + (size_t)r * m_dim;
#pragma unroll
            for (int c = 0; c < MB; ++c)
                if (c < m_dim) z = xb_fma(w3[c], mm[c], z);
            T dk = (T)1;                                               // d (dropped pre-activation) / d (pre-activation)
            if (p.drop_thr) {
                dk = egnn_drop_hash(ckey, (uint32_t)r) >= p.drop_thr ? (T)p.drop_inv_keep : (T)0;
                z *= dk;
            }
            const T sg = (T)1 / ((T)1 + xb_exp(-z));
            const T gh = g_w * p.W4[r] * (sg * ((T)1 + z * ((T)1 - sg))) * dk;
            p.a3_t[(size_t)r * E + q] = z * sg;
            p.ghid_t[(size_t)r * E + q] = gh;
#pragma unroll
            for (int c = 0; c < MB; ++c)
                if (c < m_dim) gm[c] = xb_fma(gh, w3[c], gm[c]);
        }
    } else {
        for (int c = 0; c < C; ++c) p.g_rel[(size_t)c * E + q] = (T)0;
        if (p.g_scale) p.g_scale[q] = (T)0;
    }
    if (p.g_w) p.g_w[q] = g_w;
    // the gate (:289-290): m = m0 sigmoid(gate_w . m0 + gate_b)
    if (p.gate_w) {
        T s = (T)0;
#pragma unroll
        for (int c = 0; c < MB; ++c)
            if (c < m_dim) s = xb_fma(gm[c], mm[c] / (gt > (T)0 ? gt : (T)1), s);   // m0 = m / gate (a sigmoid that underflowed to 0: the
        const T gs = s * gt * ((T)1 - gt);                                          //  factor gate (1 - gate) below is 0 as well)
#pragma unroll
        for (int c = 0; c < MB; ++c)
            if (c < m_dim) gm[c] = gm[c] * gt + gs * p.gate_w[c];
        if (p.g_gate) p.g_gate[q] = gs;
    }
    // the second SiLU of edge_mlp (:183)
#pragma unroll
    for (int c = 0; c < MB; ++c)
        if (c < m_dim) {
            const T uc = urow[c];
            const T sg = (T)1 / ((T)1 + xb_exp(-uc));
            p.gU[(size_t)q * m_dim + c] = gm[c] * (sg * ((T)1 + uc * ((T)1 - sg)));
        }
}

template <typename T>
int edge_tail_exact_launch(const egnn_edge_tail_exact_args* args, void* stream)
{
    if (!args) return EGNN_E_NULLPTR;
    const egnn_edge_tail_exact_args& a = *args;
    if (!a.u || !a.coors || !a.gU || !a.g_rel || !a.mm_t) return EGNN_E_NULLPTR;
    if (a.B <= 0 || a.N <= 0 || a.K <= 0) return EGNN_E_SHAPE;
    if (a.m_dim < 1 || a.m_dim > 64 || a.coor_dim < 1 || a.coor_dim > 64) return EGNN_E_UNSUPPORTED;
    if (!a.idx && a.K != a.N) return EGNN_E_SHAPE;
    if (a.W3 && (!a.b3 || !a.W4 || !a.b4 || !a.g_coors_out || !a.ghid_t || !a.a3_t)) return EGNN_E_NULLPTR;
    if (a.norm_coors && !a.scale) return EGNN_E_NULLPTR;
    if (a.gate_w && !a.gate_b) return EGNN_E_NULLPTR;
    const int64_t E = (int64_t)a.B * a.N * a.K;
    const int64_t blocks = (E + XB_THREADS - 1) / XB_THREADS;
    if (blocks > 0x7fffffffLL) return EGNN_E_UNSUPPORTED;
    XtArgs<T> p;
    p.B = a.B; p.N = a.N; p.K = a.K; p.m_dim = a.m_dim; p.coor_dim = a.coor_dim; p.norm_coors = a.norm_coors;
    p.u = static_cast<const T*>(a.u); p.coors = static_cast<const T*>(a.coors); p.g_coors_out = static_cast<const T*>(a.g_coors_out);
    p.g_msum = static_cast<const T*>(a.g_msum); p.W3 = static_cast<const T*>(a.W3); p.b3 = static_cast<const T*>(a.b3);
    p.W4 = static_cast<const T*>(a.W4); p.b4 = static_cast<const T*>(a.b4); p.scale = static_cast<const T*>(a.scale);
    p.gate_w = static_cast<const T*>(a.gate_w); p.gate_b = static_cast<const T*>(a.gate_b);
    p.idx = a.idx; p.pair_mask = a.pair_mask; p.eps = (T)a.eps; p.clamp = (T)a.clamp;
    p.gU = static_cast<T*>(a.gU); p.g_rel = static_cast<T*>(a.g_rel); p.ghid_t = static_cast<T*>(a.ghid_t); p.a3_t = static_cast<T*>(a.a3_t);
    p.mm_t = static_cast<T*>(a.mm_t); p.m0_t = static_cast<T*>(a.m0_t); p.g_w = static_cast<T*>(a.g_w); p.g_scale = static_cast<T*>(a.g_scale);
    p.g_gate = static_cast<T*>(a.g_gate);
    if (a.drop_thr && (!(a.drop_inv_keep >= 1.f) || a.drop_eid0 < 0 || a.drop_eid0 + E > 0xffffffffLL)) return EGNN_E_SHAPE;    // (32-bit row counter)
    p.drop_thr = a.drop_thr; p.drop_seed = a.drop_seed; p.drop_inv_keep = a.drop_inv_keep; p.drop_eid0 = a.drop_eid0;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (a.m_dim <= 16) hipLaunchKernelGGL((edge_tail_exact_bwd_kernel<T, 16>), dim3((unsigned)blocks), dim3(XB_THREADS), 0, s, p);
    else if (a.m_dim <= 32) hipLaunchKernelGGL((edge_tail_exact_bwd_kernel<T, 32>), dim3((unsigned)blocks), dim3(XB_THREADS), 0, s, p);
    else hipLaunchKernelGGL((edge_tail_exact_bwd_kernel<T, 64>), dim3((unsigned)blocks), dim3(XB_THREADS), 0, s, p);
    return egnn_launch_status();
}

extern "C" int egnn_edge_tail_exact_bwd_f32(const egnn_edge_tail_exact_args* args, void* stream) { return edge_tail_exact_launch<float>(args, stream); }
extern "C" int egnn_edge_tail_exact_bwd_f64(const egnn_edge_tail_exact_args* args, void* stream) { return edge_tail_exact_launch<double>(args, stream); }
