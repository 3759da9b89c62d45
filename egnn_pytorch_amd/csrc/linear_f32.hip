// C = act(A * W^T + bias) (+ residual) on the fp32 matrix cores of gfx950.
//
// v_mfma_f32_32x32x2_f32: f32 in, f32 accumulate, bit-for-bit an fmaf chain (exact fp32 numerics --
// bf16 would miss the 1e-4 parity bar, and gfx950 has no xf32).  Peak 157.3 TFLOP/s.
//
// Tile: 128 x 128 x 32 per 256-thread workgroup; 4 waves as 2 x 2, each wave 64 x 64 = 2 x 2 MFMA
// tiles (64 accumulator registers).  A and W tiles are staged global -> registers -> LDS with the next
// tile's global loads in flight during the current tile's MFMAs (one barrier per K-tile, double-buffered
// LDS).  LDS rows are padded to 36 floats so that the ds_read_b128 fragment reads (lane = row) and the
// ds_write_b128 staging writes are both bank-conflict free.  The k index inside a K-tile is permuted
// (MFMA k-slot kk of step t holds k = 8s + 4kk + t) so that every lane fetches its A/W fragments as one
// 16-byte LDS read per 4 MFMAs.
//
// Block -> tile map: XCD-contiguous (block b runs on XCD b % 8; each XCD gets a contiguous range of
// tiles) and grouped over M (8 M-tiles share each W panel while it is L2-hot).
#include "egnn_common.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int LDT = BK + 4;              // padded LDS row, floats
constexpr int LIN_THREADS = 256;
constexpr int GROUP_M = 8;

struct TileRegs {
    float4 a[4];
    float4 w[4];
};

template <bool ALIGNED>
__device__ __forceinline__ float4 load_row4(const float* __restrict__ base, int64_t ld, int64_t row, int64_t nrows,
                                            int k, int K)
{
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row < nrows) {
        const float* p = base + row * ld + k;
        if (ALIGNED) {
            if (k < K) v = *reinterpret_cast<const float4*>(p);
        } else {
            if (k + 0 < K) v.x = p[0];
            if (k + 1 < K) v.y = p[1];
            if (k + 2 < K) v.z = p[2];
            if (k + 3 < K) v.w = p[3];
        }
    }
    return v;
}

template <int ACT, bool HAS_RES, bool ALIGNED>
__global__ __launch_bounds__(LIN_THREADS, 2) void linear_kernel(
    const float* __restrict__ A, int64_t lda, const float* __restrict__ W, int64_t ldw,
    const float* __restrict__ bias, const float* __restrict__ R, int64_t ldr, float* __restrict__ C, int64_t ldc,
    int64_t M, int N, int K, int ntm, int ntn)
{
    __shared__ __attribute__((aligned(16))) float lds[2 * 2 * BM * LDT];   // [buf][A|W][128][36]
    float* const As0 = lds;
    float* const Ws0 = lds + BM * LDT;
    constexpr int BUF_STRIDE = 2 * BM * LDT;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    // ---- block -> tile (XCD-contiguous, bijective for any block count; then grouped over M)
    const int nblk = ntm * ntn;
    const int bid = blockIdx.x;
    const int q = nblk >> 3, rr = nblk & 7;
    const int xcd = bid & 7;
    const int v = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (bid >> 3);
    const int width = GROUP_M * ntn;
    const int gid = v / width;
    const int first_m = gid * GROUP_M;
    const int gsz = (ntm - first_m) < GROUP_M ? (ntm - first_m) : GROUP_M;
    const int tile_m = first_m + (v % width) % gsz;
    const int tile_n = (v % width) / gsz;
    const int64_t m0 = (int64_t)tile_m * BM;
    const int n0 = tile_n * BN;

    // staging coordinates: float4 index f = tid + 256u -> row f>>3, k-quad f&7
    int srow[4], skq[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int f = tid + LIN_THREADS * u;
        srow[u] = f >> 3;
        skq[u] = (f & 7) * 4;
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = (K + BK - 1) / BK;
    TileRegs tr;

    auto gload = [&](int kt) {
        const int k0 = kt * BK;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            tr.a[u] = load_row4<ALIGNED>(A, lda, m0 + srow[u], M, k0 + skq[u], K);
            tr.w[u] = load_row4<ALIGNED>(W, ldw, (int64_t)n0 + srow[u], (int64_t)N, k0 + skq[u], K);
        }
    };
    auto lstore = [&](int buf) {
        float* as = As0 + buf * BUF_STRIDE;
        float* ws = Ws0 + buf * BUF_STRIDE;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            *reinterpret_cast<float4*>(as + srow[u] * LDT + skq[u]) = tr.a[u];
            *reinterpret_cast<float4*>(ws + srow[u] * LDT + skq[u]) = tr.w[u];
        }
    };

    gload(0);
    lstore(0);
    __syncthreads();

    const int fi = lane & 31;       // fragment row inside a 32-row MFMA tile
    const int fk = (lane >> 5) * 4; // k-quad owned by this half-wave

    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) gload(kt + 1);

        const float* as = As0 + buf * BUF_STRIDE + (wm * 64 + fi) * LDT + fk;
        const float* ws = Ws0 + buf * BUF_STRIDE + (wn * 64 + fi) * LDT + fk;
#pragma unroll
        for (int s = 0; s < BK / 8; ++s) {
            const f32x4 a0 = *reinterpret_cast<const f32x4*>(as + s * 8);
            const f32x4 a1 = *reinterpret_cast<const f32x4*>(as + 32 * LDT + s * 8);
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(ws + s * 8);
            const f32x4 b1 = *reinterpret_cast<const f32x4*>(ws + 32 * LDT + s * 8);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[t], b0[t], acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[t], b1[t], acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[t], b0[t], acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[t], b1[t], acc[1][1], 0, 0, 0);
            }
        }

        if (kt + 1 < nk) lstore(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: C/D map of the 32x32 MFMA: col = lane & 31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    const int col = lane & 31;
    const int rbase = 4 * (lane >> 5);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int gn = n0 + wn * 64 + j * 32 + col;
            if (gn >= N) continue;
            const float bv = bias ? bias[gn] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t gm = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + rbase;
                if (gm >= M) continue;
                float x = acc[i][j][r] + bv;
                if (ACT == 1) x = egnn_silu(x);
                if (HAS_RES) x += R[gm * ldr + gn];
                C[gm * ldc + gn] = x;
            }
        }
    }
}

template <int ACT, bool HAS_RES, bool ALIGNED>
int launch_linear(const float* A, int64_t lda, const float* W, int64_t ldw, const float* bias, const float* R,
                  int64_t ldr, float* C, int64_t ldc, int64_t M, int N, int K, hipStream_t s)
{
    const int64_t ntm = (M + BM - 1) / BM;
    const int64_t ntn = (N + BN - 1) / BN;
    if (ntm * ntn > 0x7fffffffLL) return EGNN_E_UNSUPPORTED;
    hipLaunchKernelGGL((linear_kernel<ACT, HAS_RES, ALIGNED>), dim3((unsigned)(ntm * ntn)), dim3(LIN_THREADS), 0, s,
                       A, lda, W, ldw, bias, R, ldr, C, ldc, M, N, K, (int)ntm, (int)ntn);
    return egnn_launch_status();
}

template <int ACT, bool HAS_RES>
int dispatch_aligned(bool aligned, const float* A, int64_t lda, const float* W, int64_t ldw, const float* bias,
                     const float* R, int64_t ldr, float* C, int64_t ldc, int64_t M, int N, int K, hipStream_t s)
{
    if (aligned) return launch_linear<ACT, HAS_RES, true>(A, lda, W, ldw, bias, R, ldr, C, ldc, M, N, K, s);
    return launch_linear<ACT, HAS_RES, false>(A, lda, W, ldw, bias, R, ldr, C, ldc, M, N, K, s);
}

}  // namespace

extern "C" int egnn_linear_f32(const float* A, int64_t lda, const float* W, int64_t ldw, const float* bias,
                               const float* residual, int64_t ldr, float* C, int64_t ldc, int64_t M, int N, int K,
                               int act, void* stream)
{
    if (!A || !W || !C) return EGNN_E_NULLPTR;
    if (M <= 0 || N <= 0 || K <= 0 || lda < K || ldw < K || ldc < N) return EGNN_E_SHAPE;
    if (residual && ldr < N) return EGNN_E_SHAPE;
    if (act != 0 && act != 1) return EGNN_E_UNSUPPORTED;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool aligned = (K % 4 == 0) && (lda % 4 == 0) && (ldw % 4 == 0) &&
                         ((reinterpret_cast<uintptr_t>(A) & 15) == 0) && ((reinterpret_cast<uintptr_t>(W) & 15) == 0);
    if (act == 0) {
        if (residual) return dispatch_aligned<0, true>(aligned, A, lda, W, ldw, bias, residual, ldr, C, ldc, M, N, K, s);
        return dispatch_aligned<0, false>(aligned, A, lda, W, ldw, bias, residual, ldr, C, ldc, M, N, K, s);
    }
    if (residual) return dispatch_aligned<1, true>(aligned, A, lda, W, ldw, bias, residual, ldr, C, ldc, M, N, K, s);
    return dispatch_aligned<1, false>(aligned, A, lda, W, ldw, bias, residual, ldr, C, ldc, M, N, K, s);
}
