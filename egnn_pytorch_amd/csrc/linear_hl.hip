// C = act(A * W^T + bias) (+ residual) with BOTH operands pre-split into f16 (hi, lo) pairs -- the production GEMM.
//
// Same arithmetic as linear_split.hip (3-term split-f16 product, fp32 accumulation on v_mfma_f32_32x32x16_f16:
// fp32-class accuracy at 3/16 of the f32-MFMA cost) but A arrives already split (egnn_split_f16 /
// egnn_node_prep_hl / the previous layer's epilogue), so the kernel does no VALU work on its operands and stages
// them with the LDS-DMA path:
//   * global_load_lds_dwordx4: each wave instruction drops 16 rows x 64 B (one K-tile of 32 halves) straight into LDS,
//     no VGPRs, no ds_write; 8 instructions per wave per K-tile, double-buffered (2 x 32 KB), so the next tile streams
//     in while the 24 MFMAs per wave of the current one run.  Raw s_barrier + counted vmcnt: the DMA of tile t+1 stays
//     in flight across the barrier that publishes tile t.
//   * the LDS image is lane-linear, so the bank-conflict fix is an XOR swizzle of the 16-byte chunk index applied on
//     the per-lane SOURCE address and again on the fragment read: phys = chunk ^ ((row >> 2) & 3)  -> ds_read_b128
//     fragment reads are conflict free.
// Tile 128 x 128 x 32, 256 threads = 4 waves (2 x 2), each 64 x 64 = 2 x 2 MFMA tiles; XCD-contiguous, M-grouped
// block -> tile map.  Optional second output: the result re-split into (hi, lo) f16 for the next GEMM of the chain.
#include "egnn_common.h"

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

constexpr int BK = 32;
constexpr int ROWB = BK * 2;                 // bytes per LDS row (64)
constexpr int GROUP_M = 8;

// Tile configurations: BT x BT output tile (BT = 128: 4 waves as 2 x 2; BT = 256: 8 waves as 2 x 4), K-tile 32.
// The 256 tile halves the L2 -> LDS traffic per flop (the 128 tile streams 8.3 TB/s out of L2 on the north-star
// projection and is bound by it); the 128 tile keeps small problems from idling most of the chip.
template <int BT> struct Cfg {
    static constexpr int BM = BT, BN = BT;
    static constexpr int WAVES = BT == 256 ? 8 : 4;
    static constexpr int THREADS = WAVES * 64;
    static constexpr int WN = BT == 256 ? 4 : 2;          // waves along N
    static constexpr int TI = BM / 2 / 32;                 // MFMA tiles per wave along M (2 waves along M)
    static constexpr int TJ = BN / WN / 32;                // ... along N
    static constexpr int ARR = BT * ROWB;                  // bytes per operand image (hi or lo of A or W)
    static constexpr int BUF = 4 * ARR;                    // Ah | Al | Bh | Bl
    static constexpr int STAGE_Q = BT / (WAVES * 16);      // 16-row DMA instructions per wave per image
};

template <int BT, int ACT, bool HAS_RES>
__global__ __launch_bounds__(Cfg<BT>::THREADS, BT == 256 ? 2 : 2) void linear_hl_kernel(
    const _Float16* __restrict__ Ahi, const _Float16* __restrict__ Alo, int64_t lda,
    const _Float16* __restrict__ Whi, const _Float16* __restrict__ Wlo, int64_t ldw,
    const float* __restrict__ bias, const float* __restrict__ R, int64_t ldr,
    float* __restrict__ C, int64_t ldc, _Float16* __restrict__ Chi, _Float16* __restrict__ Clo, int64_t ldch,
    int64_t M, int N, int Kp, int ntm, int ntn, float out_scale)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];      // 2 x BUF
    using C_ = Cfg<BT>;
    constexpr int BM = C_::BM, BN = C_::BN, ARR = C_::ARR, BUF = C_::BUF, TI = C_::TI, TJ = C_::TJ, SQ = C_::STAGE_Q;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / C_::WN, wn = wave % C_::WN;

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

    // ---- LDS-DMA sources: wave w stages rows [16*SQ*w, 16*SQ*(w+1)) of each of the 4 operand images, 16 rows per instruction
    const _Float16* srcA[2][SQ];             // [hi|lo][q]
    const _Float16* srcW[2][SQ];
#pragma unroll
    for (int qq = 0; qq < SQ; ++qq) {
        const int row = wave * (16 * SQ) + qq * 16 + (lane >> 2);
        const int chunk = (lane & 3) ^ ((row >> 2) & 3);               // logical 16-byte chunk parked at physical lane&3
        int64_t ar = m0 + row;
        if (ar >= M) ar = M - 1;                                       // clamp: valid memory, result rows discarded
        srcA[0][qq] = Ahi + ar * lda + chunk * 8;
        srcA[1][qq] = Alo + ar * lda + chunk * 8;
        const int64_t wr = (int64_t)n0 + row;                          // W images are padded to ntn*128 rows
        srcW[0][qq] = Whi + wr * ldw + chunk * 8;
        srcW[1][qq] = Wlo + wr * ldw + chunk * 8;
    }
    const int dst_off = wave * (16 * SQ) * ROWB;                       // + qq*16*ROWB, + array, + buffer

    auto stage = [&](int kt, int buf) {
        const int k0 = kt * BK;
        char* base = smem + buf * BUF + dst_off;
#pragma unroll
        for (int qq = 0; qq < SQ; ++qq) {
            __builtin_amdgcn_global_load_lds((glb_void*)(srcA[0][qq] + k0), (lds_void*)(base + 0 * ARR + qq * 16 * ROWB), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((glb_void*)(srcA[1][qq] + k0), (lds_void*)(base + 1 * ARR + qq * 16 * ROWB), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((glb_void*)(srcW[0][qq] + k0), (lds_void*)(base + 2 * ARR + qq * 16 * ROWB), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((glb_void*)(srcW[1][qq] + k0), (lds_void*)(base + 3 * ARR + qq * 16 * ROWB), 16, 0, 0);
        }
    };

    f32x16 acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // ---- fragment addressing: lane (fi = l & 31, kk = l >> 5) reads logical chunk 2s + kk of its row
    const int fi = lane & 31, kk = lane >> 5;
    const int sw = (fi >> 2) & 3;                                      // row-dependent XOR (tile rows are 32-aligned)
    int foff[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) foff[s] = fi * ROWB + (((2 * s + kk) ^ sw) * 16);
    const int a_base = (wm * TI * 32) * ROWB;
    const int b_base = 2 * ARR + (wn * TJ * 32) * ROWB;

    const int nk = Kp / BK;
    stage(0, 0);

    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) {
            stage(kt + 1, buf ^ 1);
            if (SQ == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // this wave's DMAs of tile kt have landed,
            else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");            // those of tile kt+1 (4*SQ) stay in flight
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();                                  // ... and so have everyone else's
        __builtin_amdgcn_sched_barrier(0);

        const char* tb = smem + buf * BUF;
#pragma unroll
        for (int s = 0; s < BK / 16; ++s) {
            f16x8 ah[TI], al[TI], bh[TJ], bl[TJ];
#pragma unroll
            for (int i = 0; i < TI; ++i) {
                ah[i] = *reinterpret_cast<const f16x8*>(tb + a_base + 0 * ARR + i * 32 * ROWB + foff[s]);
                al[i] = *reinterpret_cast<const f16x8*>(tb + a_base + 1 * ARR + i * 32 * ROWB + foff[s]);
            }
#pragma unroll
            for (int j = 0; j < TJ; ++j) {
                bh[j] = *reinterpret_cast<const f16x8*>(tb + b_base + 0 * ARR + j * 32 * ROWB + foff[s]);
                bl[j] = *reinterpret_cast<const f16x8*>(tb + b_base + 1 * ARR + j * 32 * ROWB + foff[s]);
            }
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");            // fragment reads done before the buffer is recycled
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    }

    // ---- epilogue: C/D map of the 32x32 MFMA: col = lane & 31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    const int col = lane & 31;
    const int rbase = 4 * (lane >> 5);
#pragma unroll
    for (int i = 0; i < TI; ++i) {
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
            const int gn = n0 + wn * (TJ * 32) + j * 32 + col;
            if (gn >= N) continue;
            const float bv = bias ? bias[gn] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t gm = m0 + wm * (TI * 32) + i * 32 + (r & 3) + 8 * (r >> 2) + rbase;
                if (gm >= M) continue;
                float x = acc[i][j][r] * out_scale + bv;
                if (ACT == 1) x = egnn_silu(x);
                if (HAS_RES) x += R[gm * ldr + gn];
                if (C) C[gm * ldc + gn] = x;
                if (Chi) {
                    const _Float16 h = (_Float16)x;
                    Chi[gm * ldch + gn] = h;
                    Clo[gm * ldch + gn] = (_Float16)(x - (float)h);
                }
            }
        }
    }
}

// elementwise (hi, lo) split with zero padding of the trailing columns: one wave per row
__global__ __launch_bounds__(256) void split_f16_kernel(const float* __restrict__ X, int64_t ldx, int64_t rows, int cols,
                                                        _Float16* __restrict__ hi, _Float16* __restrict__ lo, int64_t ldh)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    for (int64_t r = wave0; r < rows; r += nwaves) {
        const float* x = X + r * ldx;
        for (int c = lane * 2; c < ldh; c += 128) {                   // ldh is even (multiple of 32)
            const float v0 = c < cols ? x[c] : 0.f;
            const float v1 = c + 1 < cols ? x[c + 1] : 0.f;
            const _Float16 h0 = (_Float16)v0, h1 = (_Float16)v1;
            typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
            *reinterpret_cast<f16x2*>(hi + r * ldh + c) = f16x2{h0, h1};
            *reinterpret_cast<f16x2*>(lo + r * ldh + c) = f16x2{(_Float16)(v0 - (float)h0), (_Float16)(v1 - (float)h1)};
        }
    }
}

template <int BT, int ACT, bool HAS_RES>
int launch_hl_bt(const _Float16* Ahi, const _Float16* Alo, int64_t lda, const _Float16* Whi, const _Float16* Wlo, int64_t ldw,
              const float* bias, const float* R, int64_t ldr, float* C, int64_t ldc, _Float16* Chi, _Float16* Clo,
              int64_t ldch, int64_t M, int N, int Kp, float out_scale, hipStream_t s)
{
    const int64_t ntm = (M + BT - 1) / BT;
    const int64_t ntn = (N + BT - 1) / BT;
    if (ntm * ntn > 0x7fffffffLL) return EGNN_E_UNSUPPORTED;
    const size_t lds = 2 * Cfg<BT>::BUF;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(linear_hl_kernel<BT, ACT, HAS_RES>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL((linear_hl_kernel<BT, ACT, HAS_RES>), dim3((unsigned)(ntm * ntn)), dim3(Cfg<BT>::THREADS), lds, s, Ahi,
                       Alo, lda, Whi, Wlo, ldw, bias, R, ldr, C, ldc, Chi, Clo, ldch, M, N, Kp, (int)ntm, (int)ntn, out_scale);
    return egnn_launch_status();
}

template <int ACT, bool HAS_RES>
int launch_hl(const _Float16* Ahi, const _Float16* Alo, int64_t lda, const _Float16* Whi, const _Float16* Wlo, int64_t ldw,
              const float* bias, const float* R, int64_t ldr, float* C, int64_t ldc, _Float16* Chi, _Float16* Clo,
              int64_t ldch, int64_t M, int N, int Kp, float out_scale, int w_rows, hipStream_t s)
{
    // 256 x 256 tiles once they fill the chip (>= 256 tiles) and the padded W image covers them; else 128 x 128
    const int64_t t256 = ((M + 255) / 256) * ((N + 255) / 256);
    if (t256 >= 256 && w_rows >= (N + 255) / 256 * 256)
        return launch_hl_bt<256, ACT, HAS_RES>(Ahi, Alo, lda, Whi, Wlo, ldw, bias, R, ldr, C, ldc, Chi, Clo, ldch, M, N, Kp, out_scale, s);
    return launch_hl_bt<128, ACT, HAS_RES>(Ahi, Alo, lda, Whi, Wlo, ldw, bias, R, ldr, C, ldc, Chi, Clo, ldch, M, N, Kp, out_scale, s);
}

}  // namespace

extern "C" int egnn_linear_hl_f32(const void* A_hi, const void* A_lo, int64_t lda, const void* W_hi, const void* W_lo,
                                  int64_t ldw, float w_inv_scale, const float* bias, const float* residual, int64_t ldr,
                                  float* C, int64_t ldc, void* C_hi, void* C_lo, int64_t ldch, int64_t M, int N, int Kp,
                                  int w_rows, int act, void* stream)
{
    if (!A_hi || !A_lo || !W_hi || !W_lo) return EGNN_E_NULLPTR;
    if (!C && !C_hi) return EGNN_E_NULLPTR;
    if ((C_hi == nullptr) != (C_lo == nullptr)) return EGNN_E_NULLPTR;
    if (M <= 0 || N <= 0 || Kp <= 0 || (Kp % BK) != 0 || lda < Kp || ldw < Kp || (lda % 8) || (ldw % 8)) return EGNN_E_SHAPE;
    if (w_rows < (N + 127) / 128 * 128) return EGNN_E_SHAPE;          // W images must cover whole 128-row tiles
    if (C && ldc < N) return EGNN_E_SHAPE;
    if (C_hi && ldch < N) return EGNN_E_SHAPE;
    if (residual && ldr < N) return EGNN_E_SHAPE;
    if (act != 0 && act != 1) return EGNN_E_UNSUPPORTED;
    if (!(w_inv_scale > 0.f)) return EGNN_E_SHAPE;
    if ((reinterpret_cast<uintptr_t>(A_hi) & 15) || (reinterpret_cast<uintptr_t>(A_lo) & 15) ||
        (reinterpret_cast<uintptr_t>(W_hi) & 15) || (reinterpret_cast<uintptr_t>(W_lo) & 15))
        return EGNN_E_ALIGN;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const _Float16 *ah = static_cast<const _Float16*>(A_hi), *al = static_cast<const _Float16*>(A_lo);
    const _Float16 *wh = static_cast<const _Float16*>(W_hi), *wl = static_cast<const _Float16*>(W_lo);
    _Float16 *ch = static_cast<_Float16*>(C_hi), *cl = static_cast<_Float16*>(C_lo);
    if (act == 0) {
        if (residual) return launch_hl<0, true>(ah, al, lda, wh, wl, ldw, bias, residual, ldr, C, ldc, ch, cl, ldch, M, N, Kp, w_inv_scale, w_rows, s);
        return launch_hl<0, false>(ah, al, lda, wh, wl, ldw, bias, residual, ldr, C, ldc, ch, cl, ldch, M, N, Kp, w_inv_scale, w_rows, s);
    }
    if (residual) return launch_hl<1, true>(ah, al, lda, wh, wl, ldw, bias, residual, ldr, C, ldc, ch, cl, ldch, M, N, Kp, w_inv_scale, w_rows, s);
    return launch_hl<1, false>(ah, al, lda, wh, wl, ldw, bias, residual, ldr, C, ldc, ch, cl, ldch, M, N, Kp, w_inv_scale, w_rows, s);
}

extern "C" int egnn_split_f16(const float* X, int64_t ldx, int64_t rows, int cols, void* hi, void* lo, int64_t ldh,
                              void* stream)
{
    if (!X || !hi || !lo) return EGNN_E_NULLPTR;
    if (rows <= 0 || cols <= 0 || ldx < cols || ldh < cols || (ldh % 32) != 0) return EGNN_E_SHAPE;
    int64_t blocks = (rows + 3) / 4;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(split_f16_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), X, ldx, rows,
                       cols, static_cast<_Float16*>(hi), static_cast<_Float16*>(lo), ldh);
    return egnn_launch_status();
}
