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
    T gd = (T)0;                                                         // S == 1: d/d (squared distance) stays in a register
    for (int h = 0; h < H; ++h) {
        T x = pi[h] + pj[h];
        const T* ws = p.Ws + (size_t)h * p.ldws;
        if (S == 1) x = xb_fma(d, ws[0], x);
        else for (int s = 0; s < S; ++s) x = xb_fma(scal[s * XB_THREADS], ws[s], x);
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
        const T dz = da * (sig * ((T)1 + x * ((T)1 - sig)));             // SiLU'(x) = sig (1 + x (1 - sig))
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
