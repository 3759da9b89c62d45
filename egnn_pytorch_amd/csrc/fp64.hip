// The float64 path (gfx950): neighbour selection, the dense layers and node_norm of a float64 module in float64 arithmetic.
//
// The reference is dtype-generic and its own tests run in float64 (tests/test_equivariance.py:6, 1e-6 bars at default-scale
// weights): a module converted with .double() computes everything in float64.  The fast kernels carry ~22 significant bits per
// product, so a float64 module takes these kernels instead (egnn_pytorch_amd/layer.py::_forward_exact with dtype float64), together
// with the float64 instantiation of the plain edge pass (edge_exact.hip: egnn_edge_exact_f64):
//   egnn_knn_select_f64  egnn_pytorch.py:230-260   squared distances, ranking edits, exact top-K with the lowest-index tie policy
//   egnn_linear_f64      :178-179, :196-201, :287, :336-337   C = act(A W^T + bias) (+ residual) on v_mfma_f64_16x16x4_f64
//   egnn_node_prep_f64   :335-336                  [LayerNorm(feats) | m_i]
// Correct and deterministic first, fast second (the float64 matrix rate of this part is 1/32 of its fp16 rate): never what
// bench.py times.
#include "egnn_common.h"

namespace {

typedef double f64x4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------------ dense layer
constexpr int DM = 64, DN = 64, DK = 16;       // workgroup tile; 4 waves as 2 x 2, each 32 x 32 = 2 x 2 MFMA tiles of 16 x 16
constexpr int DLD = DK + 2;                    // padded LDS row, doubles (144 B)
constexpr int D_THREADS = 256;

__device__ __forceinline__ double silu64(double x) { return x / (1.0 + exp(-x)); }

template <int ACT, bool HAS_RES>
__global__ __launch_bounds__(D_THREADS) void linear_f64_kernel(
    const double* __restrict__ A, int64_t lda, const double* __restrict__ W, int64_t ldw, const double* __restrict__ bias,
    const double* __restrict__ R, int64_t ldr, double* __restrict__ C, int64_t ldc, int64_t M, int N, int K, int ntn)
{
    __shared__ __attribute__((aligned(16))) double lds[2 * 2 * DM * DLD];          // [buf][A | W][64][18]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int64_t m0 = (int64_t)(blockIdx.x / ntn) * DM;
    const int n0 = (int)(blockIdx.x % ntn) * DN;
    // staging: thread t moves doubles [row t / 4][4 (t % 4) .. + 3] of both tiles
    const int srow = tid >> 2, sk = (tid & 3) * 4;
    double ra[4], rw[4];
    auto gload = [&](int kt) {
        const int k0 = kt * DK + sk;
        const int64_t ar = m0 + srow;
        const int64_t wr = (int64_t)n0 + srow;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            ra[u] = (ar < M && k0 + u < K) ? A[ar * lda + k0 + u] : 0.0;
            rw[u] = (wr < N && k0 + u < K) ? W[wr * ldw + k0 + u] : 0.0;
        }
    };
    auto lstore = [&](int buf) {
        double* as = lds + buf * (2 * DM * DLD) + srow * DLD + sk;
        double* ws = as + DM * DLD;
#pragma unroll
        for (int u = 0; u < 4; ++u) { as[u] = ra[u]; ws[u] = rw[u]; }
    };
    f64x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f64x4{0.0, 0.0, 0.0, 0.0};
    const int nk = (K + DK - 1) / DK;
    gload(0);
    lstore(0);
    __syncthreads();
    const int fr = lane & 15, fk = lane >> 4;       // fragment row / k slot of this lane (A: [m = fr][k = fk], B: [k = fk][n = fr])
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) gload(kt + 1);
        const double* as = lds + buf * (2 * DM * DLD) + (wm * 32 + fr) * DLD + fk;
        const double* ws = lds + buf * (2 * DM * DLD) + DM * DLD + (wn * 32 + fr) * DLD + fk;
#pragma unroll
        for (int s = 0; s < DK / 4; ++s) {
            const double a0 = as[4 * s], a1 = as[16 * DLD + 4 * s];
            const double b0 = ws[4 * s], b1 = ws[16 * DLD + 4 * s];
            acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
        }
        if (kt + 1 < nk) lstore(buf ^ 1);
        __syncthreads();
    }
    // D of v_mfma_f64_16x16x4_f64: lane (g = lane / 16, c = lane % 16), register r: row 4 r + g, column c (not the fp32 shape's 4 g + r)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int gn = n0 + wn * 32 + j * 16 + fr;
            if (gn >= N) continue;
            const double bv = bias ? bias[gn] : 0.0;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t gm = m0 + wm * 32 + i * 16 + 4 * r + fk;
                if (gm >= M) continue;
                double x = acc[i][j][r] + bv;
                if (ACT == 1) x = silu64(x);
                if (HAS_RES) x += R[gm * ldr + gn];
                C[gm * ldc + gn] = x;
            }
        }
}

// ------------------------------------------------------------------------------------------------ node_norm + concat
// butterfly sum of a double over the wave: the two 32-bit halves through ds_bpermute (every lane ends with the same bits)
__device__ __forceinline__ double wave_sum64(double v)
{
    const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const uint64_t u = (uint64_t)__double_as_longlong(v);
        const int src = (lane ^ o) << 2;
        const uint32_t lo = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)(uint32_t)u);
        const uint32_t hi = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)(uint32_t)(u >> 32));
        v += __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
    }
    return v;
}

__global__ __launch_bounds__(256) void node_prep_f64_kernel(const double* __restrict__ feats, const double* __restrict__ m_i,
                                                            const double* __restrict__ gamma, const double* __restrict__ beta,
                                                            double eps, double* __restrict__ out, int64_t rows, int dim, int m_dim)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    const int od = dim + m_dim;
    for (int64_t r = wave0; r < rows; r += nwaves) {
        const double* x = feats + r * dim;
        double* y = out + r * od;
        if (gamma) {                                   // two-pass statistics, as torch's LayerNorm
            double s = 0.0;
            for (int c = lane; c < dim; c += 64) s += x[c];
            const double mean = wave_sum64(s) / (double)dim;
            double v = 0.0;
            for (int c = lane; c < dim; c += 64) { const double d = x[c] - mean; v += d * d; }
            const double var = wave_sum64(v) / (double)dim;
            const double rstd = 1.0 / sqrt(var + eps);
            for (int c = lane; c < dim; c += 64) y[c] = (x[c] - mean) * rstd * gamma[c] + beta[c];
        } else {
            for (int c = lane; c < dim; c += 64) y[c] = x[c];
        }
        for (int c = lane; c < m_dim; c += 64) y[dim + c] = m_i ? m_i[r * m_dim + c] : 0.0;
    }
}

// ------------------------------------------------------------------------------------------------ neighbour selection
// One workgroup per row: the row's N ranking keys (order-preserving integer images of the values) in LDS, the exact K-th smallest by
// bitwise radix descent, ties at the threshold resolved towards the lowest index, the K selected sorted by (value, index) -- the tie
// policy of the fp32 kernels (knn_select.hip).  Any coordinate dimension: the squared distances follow the
// reference's summation tree (egnn_common.h::egnn_sqdist_any).  Instantiated for double (egnn_knn_select_f64) and for float with more
// than 8 coordinates (egnn_knn_select_f32 hands those over: knn_select.hip keeps a row's coordinates in registers up to 8).
constexpr int KN_THREADS = 256, KN_WAVES = 4;

__device__ __forceinline__ uint64_t to_key(double f)
{
    const uint64_t u = (uint64_t)__double_as_longlong(f);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
__device__ __forceinline__ uint32_t to_key(float f)
{
    const uint32_t u = __float_as_uint(f);
    return (u >> 31) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ double from_key(uint64_t k)
{
    const uint64_t u = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
    return __longlong_as_double((long long)u);
}
__device__ __forceinline__ float from_key(uint32_t k)
{
    const uint32_t u = (k >> 31) ? (k & 0x7fffffffu) : ~k;
    return __uint_as_float(u);
}
template <typename T> struct KnnKey;
template <> struct KnnKey<double> { typedef uint64_t type; static constexpr int V = 4; };
template <> struct KnnKey<float> { typedef uint32_t type; static constexpr int V = 8; };

template <typename T>
__global__ __launch_bounds__(KN_THREADS) void knn_select_any_kernel(
    const T* __restrict__ coors, const uint8_t* __restrict__ mask, const uint8_t* __restrict__ adj, int64_t adj_bstride,
    int N, int K, int C, int32_t* __restrict__ idx_out, T* __restrict__ rank_out)
{
    typedef typename KnnKey<T>::type key_t;
    constexpr int BITS = 8 * (int)sizeof(key_t);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    key_t* keys = reinterpret_cast<key_t*>(smem);                        // [N]
    key_t* selk = keys + N;                                              // [K] keys of the selected
    int* selj = reinterpret_cast<int*>(selk + K);                        // [K] their indices
    int* red = selj + K;                                                 // [2 * KN_WAVES]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.y, i = blockIdx.x;
    const T* cb = coors + (size_t)b * N * C;
    const uint8_t* mb = mask ? mask + (size_t)b * N : nullptr;
    const bool mi = mb ? mb[i] != 0 : true;
    const uint8_t* adjrow = adj ? adj + (size_t)b * adj_bstride + (size_t)i * N : nullptr;
    const size_t obase = ((size_t)b * N + i) * K;
    if (!mi && !adjrow) {                                                // a masked row: all keys 1e5, the first K indices
        for (int k = tid; k < K; k += KN_THREADS) { idx_out[obase + k] = k; rank_out[obase + k] = (T)1e5; }
        return;
    }
    for (int j = tid; j < N; j += KN_THREADS) {
        T rk = egnn_sqdist_any<T, KnnKey<T>::V>(cb + (size_t)i * C, cb + (size_t)j * C, C);
        if (!(mi && (mb ? mb[j] != 0 : true))) rk = (T)1e5;              // :240-242
        if (adjrow) {
            if (j == i) rk = (T)-1;                                      // :255
            else if (adjrow[j]) rk = (T)0;                               // :256
        }
        keys[j] = to_key(rk);
    }
    __syncthreads();
    auto block_sum = [&](int v, int slot) {
        v = egnn_wave_sum(v);
        if (lane == 0) red[slot * KN_WAVES + wave] = v;
        __syncthreads();
        int t = 0;
#pragma unroll
        for (int w = 0; w < KN_WAVES; ++w) t += red[slot * KN_WAVES + w];
        return t;
    };
    // contiguous slice of candidate indices per thread (index order = thread order: the tie pick below relies on it)
    const int per = (N + KN_THREADS - 1) / KN_THREADS;
    const int j0 = tid * per < N ? tid * per : N, j1 = (j0 + per) < N ? (j0 + per) : N;
    key_t T_ = 0;
    int below = 0;
    for (int bit = BITS - 1; bit >= 0; --bit) {
        const key_t want = T_ >> bit;
        int cnt = 0;
        for (int j = j0; j < j1; ++j) cnt += (keys[j] >> bit) == want ? 1 : 0;
        cnt = block_sum(cnt, bit & 1);                                   // (alternating slots: one barrier per step)
        if (below + cnt < K) {
            below += cnt;
            T_ |= ((key_t)1 << bit);
        }
    }
    const int need = K - below;
    int nless = 0, neq = 0;
    for (int j = j0; j < j1; ++j) {
        nless += keys[j] < T_ ? 1 : 0;
        neq += keys[j] == T_ ? 1 : 0;
    }
    __syncthreads();
    const int il = egnn_wave_inclusive_scan(nless), ie = egnn_wave_inclusive_scan(neq);
    if (lane == 63) { red[wave] = il; red[KN_WAVES + wave] = ie; }
    __syncthreads();
    int offl = il - nless, offe = ie - neq;
    for (int w = 0; w < wave; ++w) { offl += red[w]; offe += red[KN_WAVES + w]; }
    int pl = offl, pe = offe;
    for (int j = j0; j < j1; ++j) {
        const key_t kj = keys[j];
        if (kj < T_) { selk[pl] = kj; selj[pl] = j; ++pl; }
        else if (kj == T_) {
            if (pe < need) { selk[below + pe] = kj; selj[below + pe] = j; }
            ++pe;
        }
    }
    __syncthreads();
    for (int t = tid; t < K; t += KN_THREADS) {                          // sort by (value, index): rank by counting
        const key_t mk = selk[t];
        const int mj = selj[t];
        int rnk = 0;
        for (int u = 0; u < K; ++u) rnk += (selk[u] < mk || (selk[u] == mk && selj[u] < mj)) ? 1 : 0;
        idx_out[obase + rnk] = mj;
        rank_out[obase + rnk] = from_key(mk);
    }
}

template <typename T>
int knn_select_any(const T* coors, const uint8_t* mask, const uint8_t* adj, int64_t adj_batch_stride, int B, int N, int K, int coor_dim,
                   int32_t* idx_out, T* rank_out, void* stream)
{
    if (!coors || !idx_out || !rank_out) return EGNN_E_NULLPTR;
    if (B <= 0 || N <= 0 || K <= 0) return EGNN_E_SHAPE;
    if (coor_dim < 1 || coor_dim > 64) return EGNN_E_UNSUPPORTED;
    if (K > N) return EGNN_E_K_GT_N;
    if (K > 1024 || B > 65535) return EGNN_E_UNSUPPORTED;
    typedef typename KnnKey<T>::type key_t;
    const size_t lds = ((size_t)N * sizeof(key_t) + 7) / 8 * 8 + (size_t)K * (sizeof(key_t) + 4) + 2 * KN_WAVES * sizeof(int) + 8;
    if (lds > 160 * 1024) return EGNN_E_UNSUPPORTED;                      // N <= ~ 20 000 (double) / 40 000 (float)
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(knn_select_any_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(knn_select_any_kernel<T>, dim3(N, B), dim3(KN_THREADS), lds, static_cast<hipStream_t>(stream), coors, mask, adj,
                       adj_batch_stride, N, K, coor_dim, idx_out, rank_out);
    return egnn_launch_status();
}

}  // namespace

extern "C" int egnn_linear_f64(const double* A, int64_t lda, const double* W, int64_t ldw, const double* bias, const double* residual,
                               int64_t ldr, double* C, int64_t ldc, int64_t M, int N, int K, int act, void* stream)
{
    if (!A || !W || !C) return EGNN_E_NULLPTR;
    if (M <= 0 || N <= 0 || K <= 0 || lda < K || ldw < K || ldc < N) return EGNN_E_SHAPE;
    if (residual && ldr < N) return EGNN_E_SHAPE;
    if (act != 0 && act != 1) return EGNN_E_UNSUPPORTED;
    const int64_t ntm = (M + DM - 1) / DM, ntn = (N + DN - 1) / DN;
    if (ntm * ntn > 0x7fffffffLL) return EGNN_E_UNSUPPORTED;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const dim3 grid((unsigned)(ntm * ntn)), block(D_THREADS);
    if (act == 0) {
        if (residual) hipLaunchKernelGGL((linear_f64_kernel<0, true>), grid, block, 0, s, A, lda, W, ldw, bias, residual, ldr, C, ldc, M, N, K, (int)ntn);
        else hipLaunchKernelGGL((linear_f64_kernel<0, false>), grid, block, 0, s, A, lda, W, ldw, bias, residual, ldr, C, ldc, M, N, K, (int)ntn);
    } else {
        if (residual) hipLaunchKernelGGL((linear_f64_kernel<1, true>), grid, block, 0, s, A, lda, W, ldw, bias, residual, ldr, C, ldc, M, N, K, (int)ntn);
        else hipLaunchKernelGGL((linear_f64_kernel<1, false>), grid, block, 0, s, A, lda, W, ldw, bias, residual, ldr, C, ldc, M, N, K, (int)ntn);
    }
    return egnn_launch_status();
}

extern "C" int egnn_node_prep_f64(const double* feats, const double* m_i, const double* gamma, const double* beta, double eps,
                                  double* out, int64_t rows, int dim, int m_dim, void* stream)
{
    if (!feats || !out) return EGNN_E_NULLPTR;
    if ((gamma == nullptr) != (beta == nullptr)) return EGNN_E_NULLPTR;
    if (rows <= 0 || dim <= 0 || m_dim < 0) return EGNN_E_SHAPE;
    int64_t blocks = (rows + 3) / 4;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(node_prep_f64_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), feats, m_i, gamma, beta,
                       eps, out, rows, dim, m_dim);
    return egnn_launch_status();
}

extern "C" int egnn_knn_select_f64(const double* coors, const uint8_t* mask, const uint8_t* adj, int64_t adj_batch_stride, int B, int N,
                                   int K, int coor_dim, int32_t* idx_out, double* rank_out, void* stream)
{
    return knn_select_any<double>(coors, mask, adj, adj_batch_stride, B, N, K, coor_dim, idx_out, rank_out, stream);
}

// internal (knn_select.hip: egnn_knn_select_f32 with more than 8 coordinates)
int egnn_knn_select_any_f32(const float* coors, const uint8_t* mask, const uint8_t* adj, int64_t adj_batch_stride, int B, int N, int K,
                            int coor_dim, int32_t* idx_out, float* rank_out, void* stream)
{
    return knn_select_any<float>(coors, mask, adj, adj_batch_stride, B, N, K, coor_dim, idx_out, rank_out, stream);
}
