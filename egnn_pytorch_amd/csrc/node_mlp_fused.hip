// node_mlp in ONE kernel for narrow layers (dim <= 256; reference: egnn_pytorch/egnn_pytorch.py:196-201, 336-337):
//     out = W6 SiLU(W5 [LayerNorm(h) | m_i] + b5) + b6 + h
// As two launches of linear_hl.hip the hidden activation (B N x 2 dim) leaves the chip as a packed (hi, lo) image and comes back:
// with a contraction of 144 ... 512 both GEMMs are bound by that traffic, not by the matrix cores (c3 / c5 of BASELINE.json: 0.41 - 0.45
// of the HBM roofline at 0.14 - 0.37 of the MFMA peak, profiles/r05_final).  Here the hidden activation never leaves registers:
//   * one wave owns 16 nodes.  Both products run TRANSPOSED on v_mfma_f32_16x16x32_f16 -- D1[h][node] = W5 x X^T, D2[d][node] = W6 x A^T --
//     so the weights are the A operands (staged in LDS, shared by the workgroup's 8 waves = 128 nodes) and the node side is the B operand:
//     X^T (the packed [LayerNorm(h) | m_i] image, read from memory ONCE per wave into registers) for the first product, and for the second
//     the first product's ACCUMULATORS: lane (node, q) of a D tile holds rows 4q .. 4q+3 of its node's column, which -- two 16-unit tiles
//     side by side -- are exactly the eight K-slots 8q .. 8q+7 that lane supplies as a B fragment, if the 32 hidden units of the block
//     are taken in the order pi(8q + t) = 4q + t (t < 4), 16 + 4q + (t - 4) (t >= 4).  The contraction does not care about the order, so
//     W6's fragments are packed with that permutation (egnn_node_mlp_fused_pack_f16) and no transpose, LDS round trip or shuffle is needed.
//   * same arithmetic as the two-launch path: 3-term split-f16 products with fp32 accumulation, bias and SiLU in fp32, the activation
//     split into (hi, lo) in registers; results differ from it only by the order of the fp32 sums.
//   * hidden units in blocks of 32: per block the W5 rows (2 tiles x K1S k-steps x (hi, lo) KB) and the W6 columns (dim / 16 tiles x (hi, lo)
//     KB) arrive by LDS-DMA into one of two stages, requested a whole block ahead (one raw barrier per block; with a single stage and the
//     request half a block ahead the DMA had not landed when the barrier came: 0.137 -> ... ms at dim 256).
#include <type_traits>
#include <utility>
#include "egnn_common.h"
#include "egnn_lds_dma.h"

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

#ifndef EGNN_NMF_D
#define EGNN_NMF_D 4                            // fragment pairs requested ahead of the MFMAs (dim 256)
#endif
constexpr int NMF_WAVES = 8;
constexpr int NMF_THREADS = NMF_WAVES * 64;
constexpr int NMF_NODES = NMF_WAVES * 16;                // nodes per workgroup

__host__ __device__ constexpr int nmf_k1s(int ndt) { return (16 * ndt + 16 + 31) / 32; }           // 32-wide k-steps of [h | m_i] (m_dim = 16)
__host__ __device__ constexpr int nmf_f5_kb(int ndt) { return 2 * nmf_k1s(ndt) * 2; }              // KB of W5 fragments per block of 32 hidden units
__host__ __device__ constexpr int nmf_f6_kb(int ndt) { return ndt * 2; }                           // KB of W6 fragments per block

// hidden unit (within its block of 32) that K-slot s of the second product carries
__host__ __device__ __forceinline__ int nmf_pi(int s) { const int q = s >> 3, t = s & 7; return t < 4 ? 4 * q + t : 16 + 4 * q + (t - 4); }

// The fused image from the standard packed (hi, lo) images of scale * W5 (rows 2 dim, K padded to 32 K1S) and scale * W6 (rows dim, K = 2 dim):
// per block hb of 32 hidden units: [W5 fragments (ht, ks, part)] [W6 fragments (dt, part)], every fragment 64 lanes x 8 halves in lane order.
__global__ __launch_bounds__(256) void node_mlp_pack_kernel(const _Float16* __restrict__ w5h, const _Float16* __restrict__ w5l, int nkt5,
                                                            const _Float16* __restrict__ w6h, const _Float16* __restrict__ w6l, int nkt6,
                                                            _Float16* __restrict__ img, int ndt)
{
    const int k1s = nmf_k1s(ndt), nhb = ndt;             // 2 dim / 32 = dim / 16 blocks
    const int f5 = nmf_f5_kb(ndt) * 64, f6 = nmf_f6_kb(ndt) * 64;      // 16-byte chunks per block and section
    const int64_t total = (int64_t)nhb * (f5 + f6);
    for (int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x; c < total; c += (int64_t)gridDim.x * 256) {
        const int hb = (int)(c / (f5 + f6));
        int w = (int)(c - (int64_t)hb * (f5 + f6));
        f16x8 v;
        if (w < f5) {                                     // W5: fragment (ht, ks, part), lane (r, kq): row 32 hb + 16 ht + r, k = 32 ks + 8 kq ..
            const int lane = w & 63, frag = w >> 6;
            const int part = frag & 1, ks = (frag >> 1) % k1s, ht = (frag >> 1) / k1s;
            const int r = lane & 15, kq = lane >> 4;
            const _Float16* src = part ? w5l : w5h;
            v = *reinterpret_cast<const f16x8*>(src + egnn_pk_off(32 * hb + 16 * ht + r, 32 * ks + 8 * kq, nkt5));
        } else {                                          // W6: fragment (dt, part), lane (r, kq): row 16 dt + r, K-slots 8 kq .. = hidden 32 hb + pi(.)
            w -= f5;
            const int lane = w & 63, frag = w >> 6;
            const int part = frag & 1, dt = frag >> 1;
            const int r = lane & 15, kq = lane >> 4;
            const _Float16* src = part ? w6l : w6h;
            const f16x4 a = *reinterpret_cast<const f16x4*>(src + egnn_pk_off(16 * dt + r, 32 * hb + 4 * kq, nkt6));
            const f16x4 b = *reinterpret_cast<const f16x4*>(src + egnn_pk_off(16 * dt + r, 32 * hb + 16 + 4 * kq, nkt6));
            v = f16x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
        }
        *reinterpret_cast<f16x8*>(img + c * 8) = v;
    }
}

// compile-time loop: f(std::integral_constant<int, I>) for I = A .. B-1
template <int A, int B, typename F>
__device__ __forceinline__ void nmf_for(F&& f)
{
    if constexpr (A < B) {
        f(std::integral_constant<int, A>{});
        nmf_for<A + 1, B>(f);
    }
}
// LDS read / counted wait hidden from the compiler's wait-count model (cdna_hip_programming.md 5.7; as in linear_hl.hip's K loop)
template <int OFF>
__device__ __forceinline__ void nmf_lds_rd(f16x8& d, uint32_t addr)
{
    static_assert(OFF >= 0 && OFF < 65536, "ds_read offset field");
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "i"(OFF) : "memory");
}
template <int N>
__device__ __forceinline__ void nmf_wait_lgkm(f16x8& a, f16x8& b)
{
    static_assert(N >= 0 && N <= 15, "lgkmcnt field");
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "i"(N) : "memory");
}

template <int NDT>
// (second launch bound = waves per SIMD: two workgroups per CU for dim <= 128 -- at most 128 registers --, one for dim 256)
__global__ __launch_bounds__(NMF_THREADS, NDT <= 8 ? 4 : 2) void node_mlp_fused_kernel(
    const _Float16* __restrict__ xhi, const _Float16* __restrict__ xlo, const _Float16* __restrict__ img,
    const float* __restrict__ b5, const float* __restrict__ b6, const float* __restrict__ R, float* __restrict__ out,
    float w5_inv, float w6_inv, int64_t M, int32_t* __restrict__ status)
{
    constexpr int K1S = nmf_k1s(NDT), NHB = NDT, DIM = 16 * NDT;
    constexpr int S5 = nmf_f5_kb(NDT) * 1024, S6 = nmf_f6_kb(NDT) * 1024;
    constexpr int P5 = nmf_f5_kb(NDT), P6 = nmf_f6_kb(NDT);              // 1 KB DMA pieces per block
    extern __shared__ __attribute__((aligned(16))) char smem[];          // two stages of [W5 fragments | W6 fragments], then b5 (2 dim floats)
    constexpr int STG = S5 + S6;
    float* const b5s = reinterpret_cast<float*>(smem + 2 * STG);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nd = lane & 15, q = lane >> 4;
    const int64_t node0 = (int64_t)blockIdx.x * NMF_NODES + 16 * wave;
    const int64_t node = node0 + nd;
    const bool live = node < M;
    const int64_t nrow = live ? node : (M - 1);                           // (clamped: valid memory, nothing stored)

    const char* const img_b = reinterpret_cast<const char*>(img);
    const uint32_t lane16 = (uint32_t)lane * 16u;
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(size_t)(__attribute__((address_space(3))) char*)smem);
    // block hb's fragments (its W5 rows and W6 columns are contiguous in the image) -> stage hb & 1
    constexpr int NPW = (P5 + P6 + NMF_WAVES - 1) / NMF_WAVES;            // 1 KB pieces per wave and block
    auto dma_piece = [&](int hb, int i) {                                 // this wave's i-th piece of block hb
        const int pc = wave + NMF_WAVES * i;
        if (pc < P5 + P6) lds_dma16_s(img_b + (size_t)hb * STG + pc * 1024, lane16, smem + (hb & 1) * STG + pc * 1024);
    };
    auto dma = [&](int hb) {
        for (int i = 0; i < NPW; ++i) dma_piece(hb, i);
    };

    dma(0);
    for (int t = tid; t < 2 * DIM; t += NMF_THREADS) b5s[t] = b5[t];
    // the wave's 16 rows of [LayerNorm(h) | m_i] as B fragments: lane (node, q) holds k = 32 ks + 8 q .. + 7
    f16x8 xh[K1S], xl[K1S];
    {
        constexpr int nkt = 2 * K1S;
#pragma unroll
        for (int ks = 0; ks < K1S; ++ks) {
            const size_t o = egnn_pk_off(nrow, 32 * ks + 8 * q, nkt);
            xh[ks] = *reinterpret_cast<const f16x8*>(xhi + o);
            xl[ks] = *reinterpret_cast<const f16x8*>(xlo + o);
        }
    }
    f32x4 acc2[NDT];
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) acc2[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    float amax = 0.f;                                   // max |activation| (one range check per wave, after the loop)

    for (int hb = 0; hb < NHB; ++hb) {
        // ---- this block's fragments have landed (requested a whole block ago); every wave is done with the previous block's stage
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // (the next block's pieces are issued one per fragment pair below, between its MFMAs: a wave that issues its 5 - 9 LDS-DMA
        // instructions in a row behind the barrier feeds the matrix pipe nothing for ~1000 cycles, both waves of a SIMD at once)
        // The block's fragment pairs are consumed in the order they sit in the stage: G1 = 2 K1S pairs of W5 (tile, k-step), then NDT pairs of
        // W6 -- 2 KB apart.  Their reads run D pairs ahead of the MFMAs through a ring of D + 1 register pairs, as asm statements with
        // hand-counted waits (LDS returns in order): left to the compiler the loop degenerates into read -> lgkmcnt(0) -> MFMA on two
        // fragment registers and one accumulator chain (MfmaUtil 34 %, waves parked 55 % of their cycles: profiles/r06_experiments).
        const uint32_t base5 = lds0 + (uint32_t)((hb & 1) * STG) + lane16;
        const uint32_t base6 = base5 + (uint32_t)S5;
        constexpr int G1 = 2 * K1S, GT = G1 + NDT, D = NDT <= 8 ? 2 : EGNN_NMF_D, RING = (D + 1) / 2 * 2 + 2;
        f16x8 fh[RING], fl[RING];
        f32x4 acc1[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
        f16x8 ahi, alo;
        auto issue = [&](auto pc) {
            constexpr int p = decltype(pc)::value;
            if constexpr (p < G1) {
                constexpr int pos = (p & 1) * K1S + (p >> 1);             // pair p: tile p & 1, k-step p >> 1 (the two tiles' chains alternate)
                nmf_lds_rd<pos * 2048>(fh[p % RING], base5);
                nmf_lds_rd<pos * 2048 + 1024>(fl[p % RING], base5);
            } else {
                nmf_lds_rd<(p - G1) * 2048>(fh[p % RING], base6);
                nmf_lds_rd<(p - G1) * 2048 + 1024>(fl[p % RING], base6);
            }
        };
        static_assert(D % 2 == 0 || D == 3, "pairs are consumed two at a time");
        constexpr int DD = (D + 1) / 2 * 2;                                 // (whole pairs of pairs ahead)
        nmf_for<0, DD>([&](auto pc) { issue(pc); });
        nmf_for<0, GT / 2>([&](auto jc) {
            constexpr int p = 2 * decltype(jc)::value;                      // fragment pairs p, p + 1: two INDEPENDENT accumulators, their MFMAs
            if constexpr (p + DD < GT) {                                    // alternate (three back-to-back MFMAs on one accumulator wait for
                issue(std::integral_constant<int, p + DD>{});               // each other's results)
                issue(std::integral_constant<int, p + DD + 1>{});
            }
            constexpr int ahead = (GT - 2 - p) < DD ? (GT - 2 - p) : DD;    // pairs requested behind these two
            nmf_wait_lgkm<2 * ahead>(fh[p % RING], fl[p % RING]);
            nmf_wait_lgkm<2 * ahead>(fh[(p + 1) % RING], fl[(p + 1) % RING]);
            __builtin_amdgcn_sched_barrier(0);
            const f16x8 ah0 = fh[p % RING], al0 = fl[p % RING], ah1 = fh[(p + 1) % RING], al1 = fl[(p + 1) % RING];
            if constexpr (p < G1) {
                constexpr int ks = p >> 1;                                  // (pair p: tile 0, pair p + 1: tile 1 of k-step ks)
                acc1[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah0, xh[ks], acc1[0], 0, 0, 0);
                acc1[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah1, xh[ks], acc1[1], 0, 0, 0);
                acc1[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al0, xh[ks], acc1[0], 0, 0, 0);
                acc1[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al1, xh[ks], acc1[1], 0, 0, 0);
                acc1[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah0, xl[ks], acc1[0], 0, 0, 0);
                acc1[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah1, xl[ks], acc1[1], 0, 0, 0);
            } else {
                constexpr int dt = p - G1;
                acc2[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah0, ahi, acc2[dt], 0, 0, 0);
                acc2[dt + 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah1, ahi, acc2[dt + 1], 0, 0, 0);
                acc2[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al0, ahi, acc2[dt], 0, 0, 0);
                acc2[dt + 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al1, ahi, acc2[dt + 1], 0, 0, 0);
                acc2[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah0, alo, acc2[dt], 0, 0, 0);
                acc2[dt + 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah1, alo, acc2[dt + 1], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (p / 2 < NPW) {
                if (hb + 1 < NHB) dma_piece(hb + 1, p / 2);
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (p == G1 - 2) {
                // bias + SiLU (:196-201) and the (hi, lo) split of the activation, in the accumulator layout = the second product's B
                // fragment (the first W6 pairs are already on their way)
#pragma unroll
                for (int ht = 0; ht < 2; ++ht) {
                    const f32x4 bb = *reinterpret_cast<const f32x4*>(b5s + 32 * hb + 16 * ht + 4 * q);
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const float a = egnn_silu(acc1[ht][t] * w5_inv + bb[t]);
                        amax = fmaxf(amax, fabsf(a));    // (fmaxf drops a NaN: an activation beyond fp32 shows as inf, a NaN input as NaN outputs)
                        const _Float16 h = (_Float16)a;
                        ahi[4 * ht + t] = h;
                        alo[4 * ht + t] = (_Float16)(a - (float)h);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        });
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    egnn_flag_range(status, live && egnn_beyond_f16(amax), EGNN_RANGE_A_OPERAND);
    // ---- second bias + residual (:336-337).  Lane (node, q) holds output features 16 dt + 4 q .. + 3 of its node: stored from there a
    // wave instruction writes sixteen 64-byte pieces.  Through a wave-private LDS strip (the weight buffers are free now), half the
    // output columns at a time, the rows leave -- and the residual arrives -- as whole lines.
    constexpr int CG = NDT / 2, SEGF = 16 * CG, LD = SEGF + 4;             // d-tiles / floats per node and group; strip row stride
    constexpr int LPN = SEGF / 4, NPI = 64 / LPN, NI = 16 / NPI;           // lanes per node row, node rows per instruction, instructions
    static_assert(NDT >= 2 && NMF_WAVES * 16 * LD * 4 <= 2 * STG, "the strips live in the weight buffers");
    __builtin_amdgcn_s_barrier();                                          // every wave has left the weight buffers (lgkmcnt(0) above)
    float* const strip = reinterpret_cast<float*>(smem) + wave * (16 * LD);
#pragma unroll
    for (int grp = 0; grp < 2; ++grp) {
#pragma unroll
        for (int dl = 0; dl < CG; ++dl) {
            const int dt = grp * CG + dl;
            const f32x4 bb = *reinterpret_cast<const f32x4*>(b6 + 16 * dt + 4 * q);
            f32x4 o;
#pragma unroll
            for (int t = 0; t < 4; ++t) o[t] = acc2[dt][t] * w6_inv + bb[t];
            *reinterpret_cast<f32x4*>(strip + nd * LD + 16 * dl + 4 * q) = o;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int nl = i * NPI + lane / LPN, c = lane % LPN;
            const int64_t gn = node0 + nl;
            if (gn < M) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(strip + nl * LD + 4 * c);
                const f32x4 rr = *reinterpret_cast<const f32x4*>(R + gn * DIM + grp * SEGF + 4 * c);
                *reinterpret_cast<f32x4*>(out + gn * DIM + grp * SEGF + 4 * c) = v + rr;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                 // the strip is read before the next group overwrites it
        __builtin_amdgcn_wave_barrier();
    }
}

template <int NDT>
int launch_fused(const _Float16* xhi, const _Float16* xlo, const _Float16* img, const float* b5, const float* b6, const float* R, float* out,
                 float w5_inv, float w6_inv, int64_t M, int32_t* status, hipStream_t s)
{
    const size_t lds = (size_t)2 * (nmf_f5_kb(NDT) + nmf_f6_kb(NDT)) * 1024 + (size_t)2 * 16 * NDT * 4;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(node_mlp_fused_kernel<NDT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    const int64_t blocks = (M + NMF_NODES - 1) / NMF_NODES;
    if (blocks > 0x7fffffffLL) return EGNN_E_UNSUPPORTED;
    hipLaunchKernelGGL((node_mlp_fused_kernel<NDT>), dim3((unsigned)blocks), dim3(NMF_THREADS), lds, s, xhi, xlo, img, b5, b6, R, out, w5_inv, w6_inv, M, status);
    return egnn_launch_status();
}

}  // namespace

// dims the fused kernel is built for (m_dim = 16): dim % 16 == 0 with dim / 16 one of the instantiations below
static bool nmf_supported(int dim, int m_dim) { return m_dim == 16 && (dim == 32 || dim == 64 || dim == 128 || dim == 256); }

extern "C" int64_t egnn_node_mlp_fused_halves(int dim, int m_dim)
{
    if (!nmf_supported(dim, m_dim)) return 0;
    const int ndt = dim / 16;
    return (int64_t)ndt * (nmf_f5_kb(ndt) + nmf_f6_kb(ndt)) * 512;
}

extern "C" int egnn_node_mlp_fused_pack_f16(const void* W5_hi, const void* W5_lo, const void* W6_hi, const void* W6_lo, int dim, int m_dim,
                                            void* image, void* stream)
{
    if (!W5_hi || !W5_lo || !W6_hi || !W6_lo || !image) return EGNN_E_NULLPTR;
    if (!nmf_supported(dim, m_dim)) return EGNN_E_UNSUPPORTED;
    const int ndt = dim / 16;
    const int64_t chunks = egnn_node_mlp_fused_halves(dim, m_dim) / 8;
    hipLaunchKernelGGL(node_mlp_pack_kernel, dim3((unsigned)((chunks + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const _Float16*>(W5_hi), static_cast<const _Float16*>(W5_lo), 2 * nmf_k1s(ndt),
                       static_cast<const _Float16*>(W6_hi), static_cast<const _Float16*>(W6_lo), 2 * dim / 16,
                       static_cast<_Float16*>(image), ndt);
    return egnn_launch_status();
}

extern "C" int egnn_node_mlp_fused_f32(const void* X_hi, const void* X_lo, const void* image, float w5_inv_scale, const float* b5,
                                       float w6_inv_scale, const float* b6, const float* residual, float* out, int64_t M, int dim, int m_dim,
                                       int32_t* status, void* stream)
{
    if (!X_hi || !X_lo || !image || !b5 || !b6 || !residual || !out) return EGNN_E_NULLPTR;
    if (M <= 0) return EGNN_E_SHAPE;
    if (!nmf_supported(dim, m_dim)) return EGNN_E_UNSUPPORTED;
    if (!(w5_inv_scale > 0.f) || !(w6_inv_scale > 0.f)) return EGNN_E_SHAPE;
    if ((reinterpret_cast<uintptr_t>(X_hi) | reinterpret_cast<uintptr_t>(X_lo) | reinterpret_cast<uintptr_t>(image) | reinterpret_cast<uintptr_t>(b6) |
         reinterpret_cast<uintptr_t>(residual) | reinterpret_cast<uintptr_t>(out)) & 15)
        return EGNN_E_ALIGN;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const _Float16 *xh = static_cast<const _Float16*>(X_hi), *xl = static_cast<const _Float16*>(X_lo), *im = static_cast<const _Float16*>(image);
    switch (dim / 16) {
    case 2: return launch_fused<2>(xh, xl, im, b5, b6, residual, out, w5_inv_scale, w6_inv_scale, M, status, s);
    case 4: return launch_fused<4>(xh, xl, im, b5, b6, residual, out, w5_inv_scale, w6_inv_scale, M, status, s);
    case 8: return launch_fused<8>(xh, xl, im, b5, b6, residual, out, w5_inv_scale, w6_inv_scale, M, status, s);
    case 16: return launch_fused<16>(xh, xl, im, b5, b6, residual, out, w5_inv_scale, w6_inv_scale, M, status, s);
    }
    return EGNN_E_UNSUPPORTED;
}
