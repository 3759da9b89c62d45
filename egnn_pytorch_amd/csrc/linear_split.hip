// C = act(A * W^T + bias) (+ residual), fp32 in / fp32 out, on the gfx950 MATRIX cores.
//
// Why not v_mfma_f32_32x32x2_f32: measured on MI355X (tools/ubench/gen_overlap_asm.py) the f32-input MFMA runs at
// the f32 vector rate on the vector datapath (157 TFLOP/s peak, blocks the VALU); the f16 MFMA runs on the matrix
// cores at 16x that rate.  So every fp32 product is evaluated as a 3-term split-f16 product with fp32 accumulation:
//     a = a_hi + a_lo,  w = w_hi + w_lo   (f16 pairs: 22 significant bits)
//     a*w ~= a_hi*w_hi + a_lo*w_hi + a_hi*w_lo          (dropped a_lo*w_lo <= 2^-22 |a w|)
// i.e. fp32-class accuracy at 3/16 of the f32-MFMA cost.  W is split once on the host (and scaled by a power of two
// into fp16's normal range; the epilogue multiplies by the inverse); A is split on the fly while it is staged
// (v_cvt_pkrtz_f16_f32 + v_fma_mix_f32, ~2 VALU ops per element, amortised over the 128 output columns).
// Range: |A| must stay below 65504 (fp16 max); activations / features of this model are O(1..100).
//
// Tile: 128 x 128 x 32 per 256-thread workgroup, 4 waves as 2 x 2, each wave 64 x 64 = 2 x 2 tiles of
// v_mfma_f32_32x32x16_f16 (64 accumulator registers), 24 MFMAs per wave per K-tile.  The next K-tile's global loads
// are in flight while the current one is computed.  LDS holds hi and lo images of A and W with rows padded to
// 40 halves (80 B): ds_read_b128 fragment reads and the staging writes are bank-conflict free.
// Block -> tile map: XCD-contiguous, grouped over M (as linear_f32.hip).
#include "egnn_common.h"
#include "../../include/egnn_hip_ref.h"       // TEST-ONLY library (tests/libegnn_hip_ref.so)

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int LDH = BK + 8;              // padded LDS row, halves (80 bytes)
constexpr int LS_THREADS = 256;
constexpr int GROUP_M = 8;

__device__ __forceinline__ void split4(const float4 v, f16x4& hi, f16x4& lo)
{
    const f16x2 h01 = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(v.x, v.y));
    const f16x2 h23 = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(v.z, v.w));
    const f16x2 l01 = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(v.x - (float)h01[0], v.y - (float)h01[1]));
    const f16x2 l23 = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(v.z - (float)h23[0], v.w - (float)h23[1]));
    hi = f16x4{h01[0], h01[1], h23[0], h23[1]};
    lo = f16x4{l01[0], l01[1], l23[0], l23[1]};
}

template <bool ALIGNED>
__device__ __forceinline__ float4 load_a4(const float* __restrict__ base, int64_t ld, int64_t row, int64_t nrows, int k, int K)
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
__global__ __launch_bounds__(LS_THREADS, 2) void linear_split_kernel(
    const float* __restrict__ A, int64_t lda, const _Float16* __restrict__ Whi, const _Float16* __restrict__ Wlo,
    int64_t ldw, const float* __restrict__ bias, const float* __restrict__ R, int64_t ldr, float* __restrict__ C,
    int64_t ldc, int64_t M, int N, int K, int ntm, int ntn, float out_scale)
{
    __shared__ __attribute__((aligned(16))) _Float16 lds[4 * BM * LDH];   // Ah | Al | Bh | Bl, [128][40] each
    _Float16* const Ah = lds;
    _Float16* const Al = lds + BM * LDH;
    _Float16* const Bh = lds + 2 * BM * LDH;
    _Float16* const Bl = lds + 3 * BM * LDH;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    // ---- block -> tile (XCD-contiguous, bijective; grouped over M)
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

    // staging coordinates
    int arow[4], akq[4];                     // A: float4 index f = tid + 256u -> row f>>3, k-quad (f&7)*4
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int f = tid + LS_THREADS * u;
        arow[u] = f >> 3;
        akq[u] = (f & 7) * 4;
    }
    int wrow[2], wk8[2];                     // W: 16-byte index f = tid + 256u -> row f>>2, k-oct (f&3)*8
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int f = tid + LS_THREADS * u;
        wrow[u] = f >> 2;
        wk8[u] = (f & 3) * 8;
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = (K + BK - 1) / BK;
    struct Stage {
        float4 a[4];
        uint4 wh[2], wl[2];                  // 8 halves each, carried as 16 raw bytes
    } st;

    auto gload = [&](int kt) {
        const int k0 = kt * BK;
#pragma unroll
        for (int u = 0; u < 4; ++u) st.a[u] = load_a4<ALIGNED>(A, lda, m0 + arow[u], M, k0 + akq[u], K);
#pragma unroll
        for (int u = 0; u < 2; ++u) {        // W images are zero padded to (ntn*128, nk*32): no guards
            const size_t off = (size_t)(n0 + wrow[u]) * ldw + k0 + wk8[u];
            st.wh[u] = *reinterpret_cast<const uint4*>(Whi + off);
            st.wl[u] = *reinterpret_cast<const uint4*>(Wlo + off);
        }
    };
    auto lstore = [&]() {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            f16x4 hi, lo;
            split4(st.a[u], hi, lo);
            *reinterpret_cast<f16x4*>(Ah + arow[u] * LDH + akq[u]) = hi;
            *reinterpret_cast<f16x4*>(Al + arow[u] * LDH + akq[u]) = lo;
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            *reinterpret_cast<uint4*>(Bh + wrow[u] * LDH + wk8[u]) = st.wh[u];
            *reinterpret_cast<uint4*>(Bl + wrow[u] * LDH + wk8[u]) = st.wl[u];
        }
    };

    gload(0);
    lstore();
    __syncthreads();

    const int fi = lane & 31;                // fragment row inside a 32-row MFMA tile
    const int fk = (lane >> 5) * 8;          // k-oct owned by this half-wave
    const int aoff = (wm * 64 + fi) * LDH + fk;
    const int boff = (wn * 64 + fi) * LDH + fk;

    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) gload(kt + 1);

        // fragments of k16-step s+1 are fetched while the 12 MFMAs of step s run; consecutive MFMAs hit different
        // accumulators (the three terms of one product are issued 4 MFMAs apart).
        f16x8 ah[2][2], al[2][2], bh[2][2], bl[2][2];
        auto ldfrag = [&](int s, int slot) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                ah[slot][i] = *reinterpret_cast<const f16x8*>(Ah + aoff + i * 32 * LDH + s * 16);
                al[slot][i] = *reinterpret_cast<const f16x8*>(Al + aoff + i * 32 * LDH + s * 16);
                bh[slot][i] = *reinterpret_cast<const f16x8*>(Bh + boff + i * 32 * LDH + s * 16);
                bl[slot][i] = *reinterpret_cast<const f16x8*>(Bl + boff + i * 32 * LDH + s * 16);
            }
        };
        ldfrag(0, 0);
#pragma unroll
        for (int s = 0; s < BK / 16; ++s) {
            const int cur = s & 1;
            if (s + 1 < BK / 16) ldfrag(s + 1, cur ^ 1);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[cur][i], bh[cur][j], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[cur][i], bh[cur][j], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[cur][i], bl[cur][j], acc[i][j], 0, 0, 0);
        }

        __syncthreads();                      // every wave is done reading this K-tile
        if (kt + 1 < nk) {
            lstore();
            __syncthreads();
        }
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
                float x = acc[i][j][r] * out_scale + bv;
                if (ACT == 1) x = egnn_silu(x);
                if (HAS_RES) x += R[gm * ldr + gn];
                C[gm * ldc + gn] = x;
            }
        }
    }
}

template <int ACT, bool HAS_RES>
int launch_ls(bool aligned, const float* A, int64_t lda, const _Float16* Whi, const _Float16* Wlo, int64_t ldw,
              const float* bias, const float* R, int64_t ldr, float* C, int64_t ldc, int64_t M, int N, int K,
              float out_scale, hipStream_t s)
{
    const int64_t ntm = (M + BM - 1) / BM;
    const int64_t ntn = (N + BN - 1) / BN;
    if (ntm * ntn > 0x7fffffffLL) return EGNN_E_UNSUPPORTED;
    if (aligned)
        hipLaunchKernelGGL((linear_split_kernel<ACT, HAS_RES, true>), dim3((unsigned)(ntm * ntn)), dim3(LS_THREADS), 0, s,
                           A, lda, Whi, Wlo, ldw, bias, R, ldr, C, ldc, M, N, K, (int)ntm, (int)ntn, out_scale);
    else
        hipLaunchKernelGGL((linear_split_kernel<ACT, HAS_RES, false>), dim3((unsigned)(ntm * ntn)), dim3(LS_THREADS), 0, s,
                           A, lda, Whi, Wlo, ldw, bias, R, ldr, C, ldc, M, N, K, (int)ntm, (int)ntn, out_scale);
    return egnn_launch_status();
}

}  // namespace

extern "C" int egnn_linear_split_f32(const float* A, int64_t lda, const void* W_hi, const void* W_lo, int64_t ldw,
                                     float w_inv_scale, const float* bias, const float* residual, int64_t ldr,
                                     float* C, int64_t ldc, int64_t M, int N, int K, int act, void* stream)
{
    if (!A || !W_hi || !W_lo || !C) return EGNN_E_NULLPTR;
    if (M <= 0 || N <= 0 || K <= 0 || lda < K || ldc < N) return EGNN_E_SHAPE;
    if (ldw < (K + BK - 1) / BK * BK || (ldw % 8) != 0) return EGNN_E_SHAPE;     // K padded to 32 in the W images
    if (residual && ldr < N) return EGNN_E_SHAPE;
    if (act != 0 && act != 1) return EGNN_E_UNSUPPORTED;
    if (!(w_inv_scale > 0.f)) return EGNN_E_SHAPE;
    if ((reinterpret_cast<uintptr_t>(W_hi) & 15) || (reinterpret_cast<uintptr_t>(W_lo) & 15)) return EGNN_E_ALIGN;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const _Float16* wh = static_cast<const _Float16*>(W_hi);
    const _Float16* wl = static_cast<const _Float16*>(W_lo);
    const bool aligned = (K % 4 == 0) && (lda % 4 == 0) && ((reinterpret_cast<uintptr_t>(A) & 15) == 0);
    if (act == 0) {
        if (residual) return launch_ls<0, true>(aligned, A, lda, wh, wl, ldw, bias, residual, ldr, C, ldc, M, N, K, w_inv_scale, s);
        return launch_ls<0, false>(aligned, A, lda, wh, wl, ldw, bias, residual, ldr, C, ldc, M, N, K, w_inv_scale, s);
    }
    if (residual) return launch_ls<1, true>(aligned, A, lda, wh, wl, ldw, bias, residual, ldr, C, ldc, M, N, K, w_inv_scale, s);
    return launch_ls<1, false>(aligned, A, lda, wh, wl, ldw, bias, residual, ldr, C, ldc, M, N, K, w_inv_scale, s);
}
