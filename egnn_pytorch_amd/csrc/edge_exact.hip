// The edge pass in plain fp32 / fp64 (reference: egnn_pytorch/egnn_pytorch.py:262-333) -- the WIDE-RANGE path and the float64 path.
//
// The fast kernels (edge_fused.hip / edge_pw.hip) evaluate every product as a split-fp16 product on the matrix cores: fp32-class
// accuracy, but an activation beyond fp16's 65504 cannot be carried and the call ends with a range status bit.
// The reference computes in plain fp32 and has no such limit, so inputs that trip a bit -- features x 1e6, coordinates x 1e5, edge
// features x 1e9 -- are reference-legal.  This kernel is what the host binding re-runs such a call on: every operation an fp32
// VALU instruction on unscaled operands, exactly the arithmetic class of the reference (fp32 products, fp32 accumulation, overflow
// only where fp32 itself overflows), with the weights read as the module holds them (no re-layout).  It is several times slower
// than the fast path (16 FMAs per hidden value instead of 3/16 of an MFMA) and is never what bench.py times.
// Instantiated for double as well (egnn_edge_exact_f64): a float64 module -- the reference is dtype-generic and its own tests run in
// float64, tests/test_equivariance.py:6 -- computes in float64 on these kernels, not at the fast path's fp32-class precision.
//
//   kernel 1, one thread per edge (b, i, k):  x[h] = P_i[i,h] + P_j[j,h] + sum_s scal[s] W_s[h,s];  a = SiLU(x);
//             u[c] += W2[c,h] a  (the weights are wave-uniform: scalar loads);  m = SiLU(u + b2) (* gate);  coors_mlp;  masks, clamp,
//             CoorsNorm  ->  per-edge rows [m (m_dim) | w_ij * rel (C) | kept (1)] in a caller-owned workspace
//   kernel 2, one thread per (node, channel): the K rows of the node summed in k order (deterministic), mean / safe_div, outputs.
#include "egnn_common.h"

namespace {

constexpr int EX_THREADS = 256;

// (exp, not the fast exponential of the split-f16 kernels: this path trades speed for the reference's arithmetic class)
__device__ __forceinline__ float ex_exp(float x) { return expf(x); }
__device__ __forceinline__ double ex_exp(double x) { return exp(x); }
__device__ __forceinline__ float ex_sin(float x) { return sinf(x); }
__device__ __forceinline__ double ex_sin(double x) { return sin(x); }
__device__ __forceinline__ float ex_cos(float x) { return cosf(x); }
__device__ __forceinline__ double ex_cos(double x) { return cos(x); }
__device__ __forceinline__ float ex_sqrt(float x) { return sqrtf(x); }
__device__ __forceinline__ double ex_sqrt(double x) { return sqrt(x); }
__device__ __forceinline__ float ex_fma(float a, float b, float c) { return fmaf(a, b, c); }
__device__ __forceinline__ double ex_fma(double a, double b, double c) { return fma(a, b, c); }
template <typename T>
__device__ __forceinline__ T ex_silu(T x) { return x / ((T)1 + ex_exp(-x)); }

// the squared distance in the reference's operation order (what the neighbour selection ranked by): up to 8 coordinates in fp32 through
// the helpers that are bit-exact against ATen (egnn_common.h), otherwise -- more coordinates, float64 -- the general summation tree
__device__ __forceinline__ float ex_sqdist(const float* ci, const float* cj, int C)
{
    if (C == 3) {
        float dx, dy, dz;
        return egnn_sqdist(ci[0], ci[1], ci[2], cj[0], cj[1], cj[2], dx, dy, dz);
    }
    return egnn_sqdist_any<float, 8>(ci, cj, C);
}
__device__ __forceinline__ double ex_sqdist(const double* ci, const double* cj, int C) { return egnn_sqdist_any<double, 4>(ci, cj, C); }

// the argument block with its data pointers typed (float for egnn_edge_exact_f32, double for egnn_edge_exact_f64)
template <typename T>
struct ExArgs {
    int B, N, K, m_dim, H, fourier, edge_dim, coor_dim, pool_mean, edges_by_k;
    const T *Pi, *Pj, *Ws, *W2, *b2, *gate_w, *gate_b, *W3, *b3, *W4, *b4, *coors_scale, *coors, *edges, *rank;
    int64_t ldp, ldws;
    const uint8_t* mask;
    const int32_t* idx;
    T valid_radius, clamp;
    T *m_i, *coors_out, *edge_ws, *U_out;
    uint32_t drop_thr, drop_seed;
    float drop_inv_keep;
    int64_t drop_eid0;
};
template <typename T>
ExArgs<T> ex_args(const egnn_edge_exact_args& a)
{
    ExArgs<T> p;
    p.B = a.B; p.N = a.N; p.K = a.K; p.m_dim = a.m_dim; p.H = a.H; p.fourier = a.fourier; p.edge_dim = a.edge_dim;
    p.coor_dim = a.coor_dim; p.pool_mean = a.pool_mean; p.edges_by_k = a.edges_by_k;
    p.Pi = static_cast<const T*>(a.Pi); p.Pj = static_cast<const T*>(a.Pj); p.Ws = static_cast<const T*>(a.Ws);
    p.W2 = static_cast<const T*>(a.W2); p.b2 = static_cast<const T*>(a.b2);
    p.gate_w = static_cast<const T*>(a.gate_w); p.gate_b = static_cast<const T*>(a.gate_b);
    p.W3 = static_cast<const T*>(a.W3); p.b3 = static_cast<const T*>(a.b3); p.W4 = static_cast<const T*>(a.W4); p.b4 = static_cast<const T*>(a.b4);
    p.coors_scale = static_cast<const T*>(a.coors_scale); p.coors = static_cast<const T*>(a.coors);
    p.edges = static_cast<const T*>(a.edges); p.rank = static_cast<const T*>(a.rank);
    p.ldp = a.ldp; p.ldws = a.ldws; p.mask = a.mask; p.idx = a.idx;
    p.valid_radius = (T)a.valid_radius; p.clamp = (T)a.clamp;
    p.m_i = static_cast<T*>(a.m_i); p.coors_out = static_cast<T*>(a.coors_out); p.edge_ws = static_cast<T*>(a.edge_ws);
    p.U_out = static_cast<T*>(a.U_out);
    p.drop_thr = a.drop_thr; p.drop_seed = a.drop_seed; p.drop_inv_keep = a.drop_inv_keep; p.drop_eid0 = a.drop_eid0;
    return p;
}

template <typename T, int MB>                // message channels held in registers: 16 / 32 / 64
__global__ __launch_bounds__(EX_THREADS) void edge_exact_kernel(const ExArgs<T> p)
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
    // (training-mode dropout: the fused kernels' hash masks -- row = global edge id -- of the edge_mlp and coors_mlp sites)
    const uint32_t ekey = p.drop_thr ? egnn_drop_base(p.drop_seed, EGNN_DROP_SITE_EDGE, (uint32_t)(q + p.drop_eid0)) : 0u;
    const uint32_t ckey = p.drop_thr ? egnn_drop_base(p.drop_seed, EGNN_DROP_SITE_COORS, (uint32_t)(q + p.drop_eid0)) : 0u;

    // x_i - x_j and the squared distance, in the reference's operation order (egnn_common.h): what the neighbour selection ranked by
    const T* const ci = p.coors + (bN + i) * C;
    const T* const cj = p.coors + (bN + j) * C;
    const T d = ex_sqdist(ci, cj, C);                                    // (x_i - x_j is re-read per coordinate below: any C)
    // per-edge scalars [sin(d / 2^f) ..., cos(d / 2^f) ..., d, edge features ...]  (:34-41, :270-272, :282-285): a column of LDS per
    // thread (S x 256 floats, dynamic; scalar s of thread t at [s * 256 + t]: conflict-free) -- a run-time indexed private array
    // would live in scratch memory
    extern __shared__ __attribute__((aligned(16))) char scal_raw[];
    T* const scal = reinterpret_cast<T*>(scal_raw) + threadIdx.x;
    for (int f = 0; f < F; ++f) {
        const T x = d / (T)(1u << f);                                    // (x / 2^f, :36-38; F <= 31)
        scal[f * EX_THREADS] = ex_sin(x);
        scal[(F + f) * EX_THREADS] = ex_cos(x);
    }
    scal[2 * F * EX_THREADS] = d;
    if (p.edge_dim > 0) {
        const T* ep = p.edges + (p.edges_by_k ? (size_t)q : ((size_t)(bN + i) * N + j)) * p.edge_dim;
        for (int s = 0; s < p.edge_dim; ++s) scal[(2 * F + 1 + s) * EX_THREADS] = ep[s];
    }

    // ---- edge_mlp: Linear (factorised: node-level projections + the scalars' columns), SiLU, Linear
    const T* pi = p.Pi + (bN + i) * p.ldp;
    const T* pj = p.Pj + (bN + j) * p.ldp;
    T u[MB];
#pragma unroll
    for (int c = 0; c < MB; ++c) u[c] = (T)0;
    for (int h = 0; h < H; ++h) {
        T x = pi[h] + pj[h];
        const T* ws = p.Ws + (size_t)h * p.ldws;
        if (S == 1) x = ex_fma(d, ws[0], x);                             // (the common case stays in a register)
        else for (int s = 0; s < S; ++s) x = ex_fma(scal[s * EX_THREADS], ws[s], x);
        if (p.drop_thr) x = egnn_drop_hash(ekey, (uint32_t)h) >= p.drop_thr ? x * (T)p.drop_inv_keep : (T)0;      // :178-184 (training mode)
        const T a = ex_silu(x);
        const T* w2 = p.W2 + h;
#pragma unroll
        for (int c = 0; c < MB; ++c)
            if (c < m_dim) u[c] = ex_fma(w2[(size_t)c * H], a, u[c]);
    }
    if (p.U_out) {                                                       // forward under autograd: what the backward differentiates from
#pragma unroll
        for (int c = 0; c < MB; ++c)
            if (c < m_dim) p.U_out[(size_t)q * m_dim + c] = u[c] + p.b2[c];
    }
    T m[MB];
#pragma unroll
    for (int c = 0; c < MB; ++c) m[c] = c < m_dim ? ex_silu(u[c] + p.b2[c]) : (T)0;
    if (p.gate_w) {                                                      // soft_edges (:289-290)
        T gsum = p.gate_b[0];
#pragma unroll
        for (int c = 0; c < MB; ++c)
            if (c < m_dim) gsum = ex_fma(p.gate_w[c], m[c], gsum);
        const T gt = (T)1 / ((T)1 + ex_exp(-gsum));
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

    T* row = p.edge_ws + (size_t)q * (m_dim + C + 1);
    // ---- coors_mlp, CoorsNorm, clamp (:302-317)
    if (p.W3) {
        const int hid = 4 * m_dim;
        T cw = p.b4[0];
        for (int r = 0; r < hid; ++r) {
            T z = p.b3[r];
            const T* w3 = p.W3 + (size_t)r * m_dim;
#pragma unroll
            for (int c = 0; c < MB; ++c)
                if (c < m_dim) z = ex_fma(w3[c], m[c], z);
            if (p.drop_thr) z = egnn_drop_hash(ckey, (uint32_t)r) >= p.drop_thr ? z * (T)p.drop_inv_keep : (T)0;   // :203-208 (training mode)
            cw = ex_fma(p.W4[r], ex_silu(z), cw);
        }
        T inv = (T)1;
        if (p.coors_scale) {                                             // CoorsNorm (:67-77)
            T n2 = (T)0;
            for (int c = 0; c < C; ++c) { const T r = ci[c] - cj[c]; n2 = ex_fma(r, r, n2); }
            const T nrm = ex_sqrt(n2);
            inv = p.coors_scale[0] / (nrm > (T)1e-8 ? nrm : (T)1e-8);
        }
        if (has_mask && !keep) cw = (T)0;                                // :308-309
        if (p.clamp >= (T)0) cw = cw < -p.clamp ? -p.clamp : (cw > p.clamp ? p.clamp : cw);      // :311-313
        for (int c = 0; c < C; ++c) row[m_dim + c] = cw * ((ci[c] - cj[c]) * inv);
    } else {
        for (int c = 0; c < C; ++c) row[m_dim + c] = (T)0;
    }
#pragma unroll
    for (int c = 0; c < MB; ++c)
        if (c < m_dim) row[c] = (has_mask && !keep) ? (T)0 : m[c];       // masked_fill (:320-322)
    row[m_dim + C] = keep ? (T)1 : (T)0;
}

// The same for heads wider than 64 channels (the reference has no m_dim limit, egnn_pytorch.py:153): the messages do not fit the
// registers, so the second Linear runs in blocks of 64 channels -- the hidden loop is repeated per block -- and the edge's messages live
// in its workspace row, which the gate, coors_mlp and the masks read back (same thread: program order).
template <typename T>
__global__ __launch_bounds__(EX_THREADS) void edge_exact_wide_kernel(const ExArgs<T> p)
{
    const int64_t E = (int64_t)p.B * p.N * p.K;
    const int64_t q = (int64_t)blockIdx.x * EX_THREADS + threadIdx.x;
    if (q >= E) return;
    const int N = p.N, K = p.K, C = p.coor_dim, F = p.fourier, m_dim = p.m_dim, H = p.H;
    const int S = 2 * F + 1 + p.edge_dim;
    const int64_t node = q / K;
    const int k = (int)(q - node * K);
    const int64_t bN = node / N * N;
    const int i = (int)(node - bN);
    const int j = p.idx ? p.idx[q] : k;
    // (training-mode dropout: the fused kernels' hash masks -- row = global edge id -- of the edge_mlp and coors_mlp sites)
    const uint32_t ekey = p.drop_thr ? egnn_drop_base(p.drop_seed, EGNN_DROP_SITE_EDGE, (uint32_t)(q + p.drop_eid0)) : 0u;
    const uint32_t ckey = p.drop_thr ? egnn_drop_base(p.drop_seed, EGNN_DROP_SITE_COORS, (uint32_t)(q + p.drop_eid0)) : 0u;
    const T* const ci = p.coors + (bN + i) * C;
    const T* const cj = p.coors + (bN + j) * C;
    const T d = ex_sqdist(ci, cj, C);
    extern __shared__ __attribute__((aligned(16))) char scal_raw[];
    T* const scal = reinterpret_cast<T*>(scal_raw) + threadIdx.x;
    for (int f = 0; f < F; ++f) {
        const T x = d / (T)(1u << f);
        scal[f * EX_THREADS] = ex_sin(x);
        scal[(F + f) * EX_THREADS] = ex_cos(x);
    }
    scal[2 * F * EX_THREADS] = d;
    if (p.edge_dim > 0) {
        const T* ep = p.edges + (p.edges_by_k ? (size_t)q : ((size_t)(bN + i) * N + j)) * p.edge_dim;
        for (int s = 0; s < p.edge_dim; ++s) scal[(2 * F + 1 + s) * EX_THREADS] = ep[s];
    }
    const T* pi = p.Pi + (bN + i) * p.ldp;
    const T* pj = p.Pj + (bN + j) * p.ldp;
    T* row = p.edge_ws + (size_t)q * (m_dim + C + 1);
    for (int cb = 0; cb < m_dim; cb += 64) {
        T u[64];
#pragma unroll
        for (int c = 0; c < 64; ++c) u[c] = (T)0;
        for (int h = 0; h < H; ++h) {
            T x = pi[h] + pj[h];
            const T* ws = p.Ws + (size_t)h * p.ldws;
            for (int s = 0; s < S; ++s) x = ex_fma(scal[s * EX_THREADS], ws[s], x);
            if (p.drop_thr) x = egnn_drop_hash(ekey, (uint32_t)h) >= p.drop_thr ? x * (T)p.drop_inv_keep : (T)0;
            const T a = ex_silu(x);
            const T* w2 = p.W2 + (size_t)cb * H + h;
#pragma unroll
            for (int c = 0; c < 64; ++c)
                if (cb + c < m_dim) u[c] = ex_fma(w2[(size_t)c * H], a, u[c]);
        }
#pragma unroll
        for (int c = 0; c < 64; ++c)
            if (cb + c < m_dim) {
                if (p.U_out) p.U_out[(size_t)q * m_dim + cb + c] = u[c] + p.b2[cb + c];
                row[cb + c] = ex_silu(u[c] + p.b2[cb + c]);
            }
    }
    T gt = (T)1;
    if (p.gate_w) {                                                      // soft_edges (:289-290)
        T gsum = p.gate_b[0];
        for (int c = 0; c < m_dim; ++c) gsum = ex_fma(p.gate_w[c], row[c], gsum);
        gt = (T)1 / ((T)1 + ex_exp(-gsum));
    }
    const bool has_mask = p.mask != nullptr;
    bool keep = true;
    if (has_mask) {
        keep = p.mask[bN + i] && p.mask[bN + j];
        if (p.rank && p.idx) keep = keep && (p.rank[q] <= p.valid_radius);
    }
    if (p.W3) {
        const int hid = 4 * m_dim;
        T cw = p.b4[0];
        for (int r = 0; r < hid; ++r) {
            T z = p.b3[r];
            const T* w3 = p.W3 + (size_t)r * m_dim;
            for (int c = 0; c < m_dim; ++c) z = ex_fma(w3[c], row[c] * gt, z);
            if (p.drop_thr) z = egnn_drop_hash(ckey, (uint32_t)r) >= p.drop_thr ? z * (T)p.drop_inv_keep : (T)0;
            cw = ex_fma(p.W4[r], ex_silu(z), cw);
        }
        T inv = (T)1;
        if (p.coors_scale) {
            T n2 = (T)0;
            for (int c = 0; c < C; ++c) { const T r = ci[c] - cj[c]; n2 = ex_fma(r, r, n2); }
            const T nrm = ex_sqrt(n2);
            inv = p.coors_scale[0] / (nrm > (T)1e-8 ? nrm : (T)1e-8);
        }
        if (has_mask && !keep) cw = (T)0;
        if (p.clamp >= (T)0) cw = cw < -p.clamp ? -p.clamp : (cw > p.clamp ? p.clamp : cw);
        for (int c = 0; c < C; ++c) row[m_dim + c] = cw * ((ci[c] - cj[c]) * inv);
    } else {
        for (int c = 0; c < C; ++c) row[m_dim + c] = (T)0;
    }
    for (int c = 0; c < m_dim; ++c) row[c] = (has_mask && !keep) ? (T)0 : row[c] * gt;
    row[m_dim + C] = keep ? (T)1 : (T)0;
}

template <typename T>
__global__ __launch_bounds__(EX_THREADS) void edge_exact_pool_kernel(const ExArgs<T> p)
{
    const int C = p.coor_dim, m_dim = p.m_dim, K = p.K;
    const int nch = m_dim + C;
    const int64_t total = (int64_t)p.B * p.N * nch;
    const int64_t o = (int64_t)blockIdx.x * EX_THREADS + threadIdx.x;
    if (o >= total) return;
    const int64_t node = o / nch;
    const int ch = (int)(o - node * nch);
    const int ld = m_dim + C + 1;
    const T* rows = p.edge_ws + (size_t)node * K * ld;
    T s = (T)0, cnt = (T)0;
    for (int k = 0; k < K; ++k) {                                        // k order: deterministic
        s += rows[(size_t)k * ld + ch];
        cnt += rows[(size_t)k * ld + m_dim + C];
    }
    if (ch < m_dim) {
        if (p.pool_mean) {
            if (p.mask) s = (cnt == (T)0) ? (T)0 : s / (cnt > (T)1e-8 ? cnt : (T)1e-8);  // safe_div (:13-16, :326-327)
            else s = s / (T)K;                                           // :330
        }
        if (p.m_i) p.m_i[node * m_dim + ch] = s;
    } else if (p.coors_out) {
        const int c = ch - m_dim;
        p.coors_out[node * C + c] = p.coors[node * C + c] + s;           // :315
    }
}

template <typename T>
int edge_exact_launch(const egnn_edge_exact_args* args, void* stream)
{
    if (!args) return EGNN_E_NULLPTR;
    const egnn_edge_exact_args& a = *args;
    if (!a.Pi || !a.Pj || !a.Ws || !a.W2 || !a.b2 || !a.coors || !a.edge_ws) return EGNN_E_NULLPTR;
    if (!a.m_i && !a.coors_out) return EGNN_E_NULLPTR;
    if (a.B <= 0 || a.N <= 0 || a.K <= 0 || a.H <= 0 || a.ldp < a.H || a.ldws < 2 * a.fourier + 1 + a.edge_dim) return EGNN_E_SHAPE;
    if (a.m_dim < 1 || a.m_dim > 1024 || a.coor_dim < 1 || a.coor_dim > 64) return EGNN_E_UNSUPPORTED;
    if (a.fourier < 0 || a.fourier > 31 || a.edge_dim < 0) return EGNN_E_UNSUPPORTED;
    // (the scalars of a workgroup's 256 edges live in LDS: up to 160 per edge in fp32, 80 in float64)
    if ((size_t)(2 * a.fourier + 1 + a.edge_dim) * EX_THREADS * sizeof(T) > 160 * 1024) return EGNN_E_UNSUPPORTED;
    if (a.edge_dim > 0 && !a.edges) return EGNN_E_NULLPTR;
    if (a.coors_out && (!a.W3 || !a.b3 || !a.W4 || !a.b4)) return EGNN_E_NULLPTR;
    if (a.gate_w && !a.gate_b) return EGNN_E_NULLPTR;
    if (!a.idx && a.K != a.N) return EGNN_E_SHAPE;                        // dense path: K == N
    const int64_t E = (int64_t)a.B * a.N * a.K;
    const int64_t blocks = (E + EX_THREADS - 1) / EX_THREADS;
    if (blocks > 0x7fffffffLL) return EGNN_E_UNSUPPORTED;
    if (a.drop_thr && (!(a.drop_inv_keep >= 1.f) || a.drop_eid0 < 0 || a.drop_eid0 + E > 0xffffffffLL)) return EGNN_E_SHAPE;      // (32-bit mask rows)
    hipStream_t s = static_cast<hipStream_t>(stream);
    const ExArgs<T> p = ex_args<T>(a);
    const size_t lds = (size_t)(2 * a.fourier + 1 + a.edge_dim) * EX_THREADS * sizeof(T);
    auto run = [&](auto kern) -> int {
        if (lds > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return (int)e;
        }
        hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(EX_THREADS), lds, s, p);
        return egnn_launch_status();
    };
    int rc;
    if (a.m_dim <= 16) rc = run(edge_exact_kernel<T, 16>);
    else if (a.m_dim <= 32) rc = run(edge_exact_kernel<T, 32>);
    else if (a.m_dim <= 64) rc = run(edge_exact_kernel<T, 64>);
    else rc = run(edge_exact_wide_kernel<T>);
    if (rc != EGNN_OK) return rc;
    const int64_t total = (int64_t)a.B * a.N * (a.m_dim + a.coor_dim);
    hipLaunchKernelGGL(edge_exact_pool_kernel<T>, dim3((unsigned)((total + EX_THREADS - 1) / EX_THREADS)), dim3(EX_THREADS), 0, s, p);
    return egnn_launch_status();
}

// Z <- SiLU(dropout(Z)), element-wise (nn.Dropout between node_mlp's first Linear and its SiLU on this path, egnn_pytorch.py:196-201)
template <typename T>
__global__ __launch_bounds__(EX_THREADS) void drop_silu_kernel(T* __restrict__ Z, int64_t ld, int64_t rows, int cols, uint32_t thr, uint32_t seed,
                                                               float inv_keep, int64_t row0)
{
    const int64_t total = rows * cols;
    for (int64_t o = (int64_t)blockIdx.x * EX_THREADS + threadIdx.x; o < total; o += (int64_t)gridDim.x * EX_THREADS) {
        const int64_t r = o / cols;
        const int c = (int)(o - r * cols);
        T z = Z[r * ld + c];
        if (thr) z = egnn_drop_hash(egnn_drop_base(seed, EGNN_DROP_SITE_NODE, (uint32_t)(row0 + r)), (uint32_t)c) >= thr ? z * (T)inv_keep : (T)0;
        Z[r * ld + c] = ex_silu(z);
    }
}

template <typename T>
int drop_silu_launch(void* Z, int64_t ld, int64_t rows, int cols, uint32_t thr, uint32_t seed, float inv_keep, int64_t row0, void* stream)
{
    if (!Z) return EGNN_E_NULLPTR;
    if (rows <= 0 || cols <= 0 || ld < cols || row0 < 0 || row0 + rows > 0xffffffffLL) return EGNN_E_SHAPE;
    if (thr && !(inv_keep >= 1.f)) return EGNN_E_SHAPE;
    int64_t blocks = (rows * cols + EX_THREADS - 1) / EX_THREADS;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(drop_silu_kernel<T>, dim3((unsigned)blocks), dim3(EX_THREADS), 0, static_cast<hipStream_t>(stream), static_cast<T*>(Z), ld,
                       rows, cols, thr, seed, inv_keep, row0);
    return egnn_launch_status();
}

}  // namespace

extern "C" int egnn_drop_silu_f32(void* Z, int64_t ld, int64_t rows, int cols, uint32_t drop_thr, uint32_t drop_seed, float drop_inv_keep,
                                  int64_t row0, void* stream)
{
    return drop_silu_launch<float>(Z, ld, rows, cols, drop_thr, drop_seed, drop_inv_keep, row0, stream);
}
extern "C" int egnn_drop_silu_f64(void* Z, int64_t ld, int64_t rows, int cols, uint32_t drop_thr, uint32_t drop_seed, float drop_inv_keep,
                                  int64_t row0, void* stream)
{
    return drop_silu_launch<double>(Z, ld, rows, cols, drop_thr, drop_seed, drop_inv_keep, row0, stream);
}

extern "C" size_t egnn_edge_exact_workspace_bytes(int B, int N, int K, int m_dim, int coor_dim)
{
    if (B <= 0 || N <= 0 || K <= 0 || m_dim <= 0 || coor_dim <= 0) return 0;
    return (size_t)B * N * K * (size_t)(m_dim + coor_dim + 1) * sizeof(float);
}

extern "C" int egnn_edge_exact_f32(const egnn_edge_exact_args* args, void* stream) { return edge_exact_launch<float>(args, stream); }

extern "C" int egnn_edge_exact_f64(const egnn_edge_exact_args* args, void* stream) { return edge_exact_launch<double>(args, stream); }
