// C = act(A * W^T + bias) (+ residual) with BOTH operands pre-split into f16 (hi, lo) pairs -- the production GEMM.
//
// Same arithmetic as linear_split.hip (3-term split-f16 product, fp32 accumulation on v_mfma_f32_32x32x16_f16:
// fp32-class accuracy at 3/16 of the f32-MFMA cost) but A arrives already split (egnn_split_f16 /
// egnn_node_prep_hl / the previous layer's epilogue), so the kernel does no VALU work on its operands and stages
// them with the LDS-DMA path:
//   * operands live in HBM in a PACKED tile-major layout (egnn_common.h: egnn_pk_off): [row/32][k/16][32 rows][2 x 16 B],
//     so that one K-tile of one 32-row block is 1 KB of contiguous memory.  A wave-level load is processed line by line
//     (tools/ubench/gather.hip): with row-major operands every DMA instruction touched 32 partly-used lines and the
//     stream ran at ~10 TB/s out of L2 (this kernel's main loop was bound by it); packed, it touches 8 full lines.
//   * global_load_lds_dwordx4: each wave instruction drops 32 rows x 32 B (one K-tile of 16 halves) straight into LDS,
//     no VGPRs, no ds_write; 4 instructions per wave per K-tile into a 4-deep LDS ring, ONE raw s_barrier per K-tile and
//     a counted vmcnt, so three tiles are in flight behind the one being multiplied (the 2-barrier, 1-tile-ahead
//     version of this kernel sat at 0.31 of the f16 MFMA peak: L2 latency ~ one tile of MFMAs).
//   * the LDS image is lane-linear = the packed HBM image, which already carries the bank-conflict fix: the 16-byte
//     chunk index is XOR-swizzled, phys = chunk ^ ((row >> 3) & 1), and the fragment read applies the same XOR
//     -> conflict-free ds_read_b128.
// Tiles 256 x 256 (8 waves, 2 x 4) or 128 x 128 (4 waves, 2 x 2), K-tile 16; XCD-contiguous, M-grouped block -> tile
// map.  Optional second output: the result re-split into (hi, lo) f16 for the next GEMM of the chain.
#include <cstdlib>
#include "egnn_common.h"

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

// training-mode dropout between the Linear and its activation (egnn_linear_hl_drop_f32); thr = 0: off
struct DropArgs {
    uint32_t thr, seed; float inv_keep;
    const uint8_t* row_mask;     // (M) bytes or NULL: an M-tile none of whose rows is set is not computed (egnn_linear_hl_lda_rows_f32)
};

// Build knobs (tools/variants.py build src=linear_hl ...).  Until round 6 the defaults of ILV and PRIO were defined BELOW the kernel body,
// which therefore saw them undefined (= 0): the production library ran without either, whatever the comments said.
#ifndef EGNN_HL_ILV
#define EGNN_HL_ILV 0                        // classic loop: a K-tile's DMA issue spread over its three MFMA groups
#endif
#ifndef EGNN_HL_PRIO
#define EGNN_HL_PRIO 0                       // s_setprio 3 for the K loop, 0 for the epilogue
#endif
#ifndef EGNN_HL_CFG
#define EGNN_HL_CFG 1
#endif
#ifndef EGNN_HL_NT
#define EGNN_HL_NT 0                         // experiment: `nt` on the LDS-DMA pieces of the hand-scheduled loop (1: A pieces, 2: all)
#endif
#ifndef EGNN_HL_DMAPOS
#define EGNN_HL_DMAPOS 0                     // experiment: 1 = a tile's DMA pieces right behind the barrier instead of between group 3's MFMAs
#endif
#ifndef EGNN_HL_NTSTORE
#define EGNN_HL_NTSTORE 2                    // non-temporal output stores in the staged epilogue: 0 never, 1 always, 2 the projection table only
#endif
#ifndef EGNN_HL_LOOP
#define EGNN_HL_LOOP 2                       // 2: software-pipelined, hand-scheduled K loop (64 x 64 wave tiles); 0: the classic loop
#endif
constexpr int BK = 16;                       // one v_mfma_f32_32x32x16_f16 step per K-tile
constexpr int ROWB = BK * 2;                 // bytes per LDS row (32)
#ifndef EGNN_HL_GROUP_M
#define EGNN_HL_GROUP_M 8
#endif
constexpr int GROUP_M = EGNN_HL_GROUP_M;

// Tile configurations (BM x BN output tile, K-tile 16, every wave owns a 64 x 64 sub-tile = 2 x 2 MFMA tiles):
//   0: 128 x 128, 4 waves (2 x 2), 4-deep ring of 16 KB  -> 64 KB LDS, two workgroups per CU
//   1: 256 x 128, 8 waves (4 x 2), 3-deep ring of 24 KB  -> 72 KB LDS, two workgroups per CU; 1.37x the flops per byte
//      streamed out of L2 (the 128 x 128 tile draws ~11 TB/s from L2 on the north-star projection, which is what
//      holds its MFMA utilisation near 45 %)
//   2: 256 x 256, 8 waves (2 x 4, 128 x 64 per wave), 4-deep ring of 32 KB -> 128 KB LDS, one workgroup per CU
// ---- hand-scheduled K loop (EGNN_HL_LOOP == 2): LDS reads, LDS-DMA and their waits as inline assembly, hidden from the compiler's
// wait-count model (cdna_hip_programming.md 5.7: asm loads are not counted, so every wait below is placed and counted by hand).
template <int OFF>
__device__ __forceinline__ void hl_lds_rd(f16x8& d, uint32_t addr)
{
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "i"(OFF) : "memory");
}
// one LDS-DMA piece: lane l's 16 bytes at (sbase + voff) land at lds_dst + 16 l; wave-uniform 64-bit base in scalar registers
template <bool NT = false>
__device__ __forceinline__ void hl_dma16(const char* sbase, uint32_t voff, uint32_t lds_dst)
{
    if constexpr (NT)                        // (experiment EGNN_HL_NT: non-temporal policy for the streamed operand)
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 nt" :: "v"(voff), "s"(sbase), "s"(lds_dst) : "memory", "m0");
    else
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff), "s"(sbase), "s"(lds_dst) : "memory", "m0");
}
__device__ __forceinline__ const char* hl_uniform_ptr(const void* p)
{
    const uint64_t a = (uint64_t)(size_t)p;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a), hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
    return reinterpret_cast<const char*>((size_t)(((uint64_t)hi << 32) | lo));
}

template <int CFG> struct Cfg;
template <> struct Cfg<0> { static constexpr int BM = 128, BN = 128, WM = 2, WN = 2, TI = 2, TJ = 2, STAGES = 4; };
template <> struct Cfg<1> { static constexpr int BM = 256, BN = 128, WM = 4, WN = 2, TI = 2, TJ = 2, STAGES = 3; };
template <> struct Cfg<2> { static constexpr int BM = 256, BN = 256, WM = 2, WN = 4, TI = 4, TJ = 2, STAGES = 4; };
// Measured and dropped (north-star projection, 0.82 ms with configuration 1): 128 x 64 per wave with 4-wave workgroups
// (256 x 128 or 128 x 256 tiles, 2 waves per SIMD) 0.89-0.90 ms; 128 x 128 per wave (256 x 256 tile, accumulators in
// AGPRs, 1 wave per SIMD) 2.07 ms -- a wave that issues 8 LDS-DMA pieces per K-tile stalls its own MFMA stream and
// nothing else is resident to cover it; 128 x 128 tiles with a 3- or 2-deep ring (3 - 4 workgroups per CU) 0.90-0.91 ms.

// Pipeline (per K-tile of 16, ONE barrier), S = STAGES:
//     s_waitcnt vmcnt((S-2) * DPW)   this wave's DMAs of tile kt have landed (tiles kt+1 .. kt+S-2 stay in flight)
//     s_barrier                      ... and everyone else's; also: every wave is done reading the ring slot of tile kt-1
//     issue the DMAs of tile kt+S-1 into that slot
//     fragments of tile kt -> 3 x TI x TJ MFMAs
// (Device function of the block index: see edge_fused.hip::edge_body.)
template <int CFG, int ACT, bool HAS_RES, bool DROP = false>
__device__ __forceinline__ void linear_hl_body(
    const _Float16* __restrict__ Ahi, const _Float16* __restrict__ Alo,
    const _Float16* __restrict__ Whi, const _Float16* __restrict__ Wlo,
    const float* __restrict__ bias, const float* __restrict__ R, int64_t ldr,
    float* __restrict__ C, int64_t ldc, _Float16* __restrict__ Chi, _Float16* __restrict__ Clo, int nkt_out,
    int64_t M, int N, int Kp, int ntm, int ntn, float out_scale, int split_cols, int32_t* __restrict__ status, char* smem, const int bid,
    const int kt0 = 0, const int kt_count = -1, const DropArgs drop = DropArgs{0u, 0u, 1.f}, const int a_nkt = 0, const int group_m = GROUP_M)
{
    using C_ = Cfg<CFG>;
    constexpr int BM = C_::BM, BN = C_::BN, TI = C_::TI, TJ = C_::TJ, STAGES = C_::STAGES;
    constexpr int WAVES = C_::WM * C_::WN;
    constexpr int ARB = BM / 32, WRB = BN / 32;                       // 32-row blocks per image
    constexpr int AARR = BM * ROWB, WARR = BN * ROWB;                 // bytes per A / W image
    constexpr int BUF = 2 * AARR + 2 * WARR;                          // Ah | Al | Wh | Wl
    constexpr int NDMA = 2 * ARB + 2 * WRB;                           // 1 KB pieces per K-tile
    constexpr int DPW = NDMA / WAVES;                                 // ... per wave
    static_assert(NDMA % WAVES == 0, "DMA pieces must divide evenly over the waves");

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / C_::WN, wn = wave % C_::WN;

    // ---- block -> tile (XCD-contiguous, bijective; grouped over M)
    const int nblk = ntm * ntn;
    const int q = nblk >> 3, rr = nblk & 7;
    const int xcd = bid & 7;
    const int v = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (bid >> 3);
    const int width = group_m * ntn;
    const int gid = v / width;
    const int first_m = gid * group_m;
    const int gsz = (ntm - first_m) < group_m ? (ntm - first_m) : group_m;
    const int tile_m = first_m + (v % width) % gsz;
    const int tile_n = (v % width) / gsz;
    const int64_t m0 = (int64_t)tile_m * BM;
    const int n0 = tile_n * BN;
    if (drop.row_mask) {
        // rows nobody reads: every wave looks at the tile's BM flags and comes to the same answer (no barrier has been passed yet)
        bool any = false;
#pragma unroll
        for (int j = 0; j < BM / 64; ++j) {
            const int64_t row = m0 + lane + 64 * j;
            any = any || (row < M && drop.row_mask[row] != 0);
        }
        if (__builtin_amdgcn_ballot_w64(any) == 0ull) return;
    }

    // ---- LDS-DMA pieces: piece d = wave + WAVES * j (j < DPW) of the list [Ah blocks | Al blocks | Wh blocks | Wl blocks].
    // In the packed layout a (row block, K-tile) piece is 1 KB of contiguous memory that is ALREADY the LDS image
    // (chunk swizzle included): lane l copies bytes [16 l, 16 l + 16).
    const int nkt = Kp / BK;
    const _Float16* src[DPW];
    int dst[DPW];
    {
        const int64_t rbA_max = (M - 1) >> 5;
#pragma unroll
        for (int j = 0; j < DPW; ++j) {
            const int d = wave + WAVES * j;
            const _Float16* base;
            int64_t rb;
            int off;
            if (d < 2 * ARB) {
                const int blk = d % ARB;
                rb = (m0 >> 5) + blk;
                if (rb > rbA_max) rb = rbA_max;                        // clamp: valid memory, result rows discarded
                base = d < ARB ? Ahi : Alo;
                off = (d < ARB ? 0 : AARR) + blk * 32 * ROWB;
            } else {
                const int e = d - 2 * ARB;
                const int blk = e % WRB;
                rb = (int64_t)(n0 >> 5) + blk;                         // W images are padded to whole tiles
                base = e < WRB ? Whi : Wlo;
                off = 2 * AARR + (e < WRB ? 0 : WARR) + blk * 32 * ROWB;
            }
            // (a_nkt: K-tiles per row block of the A image when it is wider than the contraction -- egnn_linear_hl_lda_f32)
            src[j] = base + rb * ((d < 2 * ARB && a_nkt) ? a_nkt : nkt) * 512 + lane * 8;
            dst[j] = off;
        }
    }
    auto stage = [&](int kt_src, int slot) {
        char* base = smem + slot * BUF;
#pragma unroll
        for (int j = 0; j < DPW; ++j)
            __builtin_amdgcn_global_load_lds((glb_void*)(src[j] + kt_src * 512), (lds_void*)(base + dst[j]), 16, 0, 0);
    };
    // pieces j = part, part + 3, ... (the main loop spreads a K-tile's DMA issue over its three MFMA groups)
    auto stage_part = [&](int kt_src, int slot, int part) {
        char* base = smem + slot * BUF;
#pragma unroll
        for (int j = 0; j < DPW; ++j)
            if (j % 3 == part)
                __builtin_amdgcn_global_load_lds((glb_void*)(src[j] + kt_src * 512), (lds_void*)(base + dst[j]), 16, 0, 0);
    };

    f32x16 acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // ---- fragment addressing: lane (fi = l & 31, kk = l >> 5) reads logical chunk kk of its row
    const int fi = lane & 31, kk = lane >> 5;
    const int foff = fi * ROWB + ((kk ^ ((fi >> 3) & 1)) * 16);        // tile rows are 32-aligned
    const int a_base = (wm * TI * 32) * ROWB + foff;
    const int b_base = 2 * AARR + (wn * TJ * 32) * ROWB + foff;

    // (split-K launches -- egnn_linear_hl_splitk_f32 -- give every workgroup a range [kt0, kt0 + kt_count) of the K-tiles)
    const int nk = kt_count < 0 ? nkt : kt_count;
    constexpr bool HAND = EGNN_HL_LOOP == 2 && TI == 2 && TJ == 2 && DPW <= 4;      // the hand-scheduled loop: 64 x 64 wave tiles
    // prologue: tiles 0 .. S-2 in flight (dummies past the end keep the vmcnt bookkeeping uniform)
    if constexpr (!HAND) {
#pragma unroll
        for (int t = 0; t < STAGES - 1; ++t) stage(kt0 + (t < nk ? t : nk - 1), t);
    }

    int slot = 0;                                                      // ring slot of tile kt
#if EGNN_HL_PRIO
    asm volatile("s_setprio 3");             // the K loop issues ahead of the CU's other workgroup when that one is in its epilogue
#endif
    auto wait_tile = [&]() {                 // this wave's DMAs of the oldest tile in flight have landed; its own LDS reads are complete
        if (STAGES == 4 && DPW == 4) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
        else if (STAGES == 3 && DPW == 3) asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory");
        else if (STAGES == 3 && DPW == 4) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    };
    // ---- The K loop, software-pipelined and hand-scheduled (round 6; profiles/r06_experiments/gemm_k_loop.txt).
    // The classic loop (the else branch) puts the barrier at the HEAD of a K-tile: behind it every wave issues its DMA pieces, reads four
    // fragments and waits for them before its first MFMA -- a few hundred cycles in which the wave feeds the matrix pipe nothing, per 384
    // cycles of its own MFMA work, all eight waves of the workgroup at the same time.  Here the barrier that publishes tile kt+1 sits
    // INSIDE tile kt, between its second and third MFMA group, and the hi fragments of tile kt+1 are read behind it into a second
    // register set (+16 VGPRs: 114) while group 3 runs:
    //     [ah, bh of tile kt in registers]
    //     reads al0 al1 bl0 bl1 (tile kt)       outstanding: [ah' bh' x 4 of the previous tile's tail] + 4
    //     lgkmcnt(4)  -> ah, bh                 g1: ah x bh
    //     lgkmcnt(2)  -> al                     g2: al x bh
    //     vmcnt((S-2) DPW) lgkmcnt(0); s_barrier         own reads of tile kt complete, own DMAs of tile kt+1 landed
    //                                                    -> tile kt+1 visible, ring slot of tile kt free
    //     reads ah' bh' (tile kt+1)             g3: ah x bl, the DMA pieces of tile kt+S (into the freed slot) between its MFMAs
    // A wave arrives at the barrier with MFMAs in the pipe, has four more ready behind it, and every fragment read is issued at least one
    // MFMA group (128 matrix-pipe cycles) ahead of its first use.  The same schedule written with compiler-visible loads does NOT come
    // out this way: the register allocator re-uses fragment registers across the loop's back edge and the wait-count pass then puts
    // lgkmcnt(0) right behind the reads, in front of the tile's first MFMA (measured: -3 % instead of -9 ... -11 %).  So the fragment reads
    // and the DMA pieces are asm statements in program order (the compiler does not count them: cdna_hip_programming.md 5.7), the waits are
    // counted by hand (LDS returns in order) and sched_barrier pins the MFMAs between them; DMA sources are a scalar base + one shared
    // lane offset (no 64-bit vector address arithmetic per piece).  Same products in the same order per accumulator: the same bits.
    if constexpr (HAND) {
        const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(size_t)(__attribute__((address_space(3))) char*)smem);
        const char* sb[DPW];
        uint32_t sdst[DPW];
#pragma unroll
        for (int j = 0; j < DPW; ++j) {
            sb[j] = hl_uniform_ptr(src[j] - lane * 8);
            sdst[j] = lds0 + (uint32_t)dst[j];
        }
        const uint32_t voff = (uint32_t)lane * 16u;
        const uint32_t a_addr = lds0 + (uint32_t)a_base, b_addr = lds0 + (uint32_t)b_base;
        auto dma = [&](int kt_src, int sl, int j) {
#if defined(EGNN_HL_ABL) && (EGNN_HL_ABL & 4)
            kt_src = 0;                                                // ablation: always the same (L1/L2-hot) tile
#endif
            // (piece j of every wave is an A piece for WAVES j < 2 ARB, a W piece behind that)
#if EGNN_HL_NT == 2
            hl_dma16<true>(sb[j] + (size_t)kt_src * 1024, voff, sdst[j] + (uint32_t)(sl * BUF));
#elif EGNN_HL_NT == 1
            if (WAVES * j < 2 * ARB) hl_dma16<true>(sb[j] + (size_t)kt_src * 1024, voff, sdst[j] + (uint32_t)(sl * BUF));
            else hl_dma16<false>(sb[j] + (size_t)kt_src * 1024, voff, sdst[j] + (uint32_t)(sl * BUF));
#else
            hl_dma16<false>(sb[j] + (size_t)kt_src * 1024, voff, sdst[j] + (uint32_t)(sl * BUF));
#endif
        };
#pragma unroll
        for (int t = 0; t < STAGES - 1; ++t)
#pragma unroll
            for (int j = 0; j < DPW; ++j) dma(kt0 + (t < nk ? t : nk - 1), t, j);
        wait_tile();
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < DPW; ++j) dma(kt0 + (STAGES - 1 < nk ? STAGES - 1 : nk - 1), STAGES - 1, j);
        f16x8 ahA[TI], bhA[TJ], ahB[TI], bhB[TJ];
        hl_lds_rd<0>(ahA[0], a_addr); hl_lds_rd<32 * ROWB>(ahA[1], a_addr);
        hl_lds_rd<0>(bhA[0], b_addr); hl_lds_rd<32 * ROWB>(bhA[1], b_addr);
        auto tile = [&](f16x8 (&ah)[TI], f16x8 (&bh)[TJ], f16x8 (&ahn)[TI], f16x8 (&bhn)[TJ], const int kt) {
            const uint32_t so = (uint32_t)(slot * BUF);
            f16x8 al[TI], bl[TJ];
            hl_lds_rd<AARR>(al[0], a_addr + so); hl_lds_rd<AARR + 32 * ROWB>(al[1], a_addr + so);
            hl_lds_rd<WARR>(bl[0], b_addr + so); hl_lds_rd<WARR + 32 * ROWB>(bl[1], b_addr + so);
            asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(ah[0]), "+v"(ah[1]), "+v"(bh[0]), "+v"(bh[1]) :: "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(al[0]), "+v"(al[1]) :: "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (STAGES == 4 && DPW == 4) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" : "+v"(bl[0]), "+v"(bl[1]) :: "memory");
            else if (STAGES == 3 && DPW == 3) asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" : "+v"(bl[0]), "+v"(bl[1]) :: "memory");
            else if (STAGES == 3 && DPW == 4) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" : "+v"(bl[0]), "+v"(bl[1]) :: "memory");
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : "+v"(bl[0]), "+v"(bl[1]) :: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            const int nslot = slot + 1 == STAGES ? 0 : slot + 1;
            const int nsrc = kt0 + (kt + STAGES < nk ? kt + STAGES : nk - 1);      // past the end: harmless re-fetch of the last tile
            const uint32_t sn = (uint32_t)(nslot * BUF);
            hl_lds_rd<0>(ahn[0], a_addr + sn); hl_lds_rd<32 * ROWB>(ahn[1], a_addr + sn);
            hl_lds_rd<0>(bhn[0], b_addr + sn); hl_lds_rd<32 * ROWB>(bhn[1], b_addr + sn);
#if EGNN_HL_DMAPOS == 1
#pragma unroll
            for (int j = 0; j < DPW; ++j) dma(nsrc, slot, j);
#endif
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
#if !(defined(EGNN_HL_ABL) && (EGNN_HL_ABL & 8)) && EGNN_HL_DMAPOS == 0  // (ablation 8: no DMA in the loop -- timing only)
                    if (i * TJ + j < DPW) dma(nsrc, slot, i * TJ + j);
#endif
                    __builtin_amdgcn_sched_barrier(0);
                }
            slot = nslot;
        };
        for (int kt = 0; kt < nk; kt += 2) {
            tile(ahA, bhA, ahB, bhB, kt);
            if (kt + 1 < nk) tile(ahB, bhB, ahA, bhA, kt + 1);
        }
        // (the last tile's look-ahead reads: their destinations may not be re-used before they have landed)
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ahA[0]), "+v"(ahA[1]), "+v"(bhA[0]), "+v"(bhA[1]), "+v"(ahB[0]), "+v"(ahB[1]), "+v"(bhB[0]), "+v"(bhB[1]) :: "memory");
    } else {
    for (int kt = 0; kt < nk; ++kt) {
        wait_tile();
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        const int nt = kt + STAGES - 1;
        int nslot = slot - 1;
        if (nslot < 0) nslot += STAGES;                                // (kt + S - 1) % S
        const int nsrc = kt0 + (nt < nk ? nt : nk - 1);                // past the end: harmless re-fetch of the last tile
#if !(defined(EGNN_HL_ILV) && EGNN_HL_ILV)
        {
#if defined(EGNN_HL_ABL) && (EGNN_HL_ABL & 4)
            stage(0, nslot);                                           // ablation: always the same (L1/L2-hot) tile
#elif defined(EGNN_HL_ABL) && (EGNN_HL_ABL & 8)
            if (kt == 0) stage(0, nslot); else { asm volatile("" ::: "memory"); }   // ablation: (almost) no DMA -- vmcnt bookkeeping breaks, timing only
#else
            stage(nsrc, nslot);
#endif
        }
#endif
        const char* tb = smem + slot * BUF;
        f16x8 ah[TI], al[TI], bh[TJ], bl[TJ];
#pragma unroll
        for (int i = 0; i < TI; ++i) {
            ah[i] = *reinterpret_cast<const f16x8*>(tb + a_base + i * 32 * ROWB);
            al[i] = *reinterpret_cast<const f16x8*>(tb + a_base + AARR + i * 32 * ROWB);
        }
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
            bh[j] = *reinterpret_cast<const f16x8*>(tb + b_base + j * 32 * ROWB);
            bl[j] = *reinterpret_cast<const f16x8*>(tb + b_base + WARR + j * 32 * ROWB);
        }
#if defined(EGNN_HL_ABL) && (EGNN_HL_ABL & 2)
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int j = 0; j < TJ; ++j)                                // ablation: one MFMA per (i, j) instead of three
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i] + al[i], bh[j] + bl[j], acc[i][j], 0, 0, 0);
        slot = slot + 1 == STAGES ? 0 : slot + 1;
        continue;
#endif
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int j = 0; j < TJ; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
#if defined(EGNN_HL_ILV) && EGNN_HL_ILV
        __builtin_amdgcn_sched_barrier(0);
        stage_part(nsrc, nslot, 0);
        __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int j = 0; j < TJ; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
#if defined(EGNN_HL_ILV) && EGNN_HL_ILV
        __builtin_amdgcn_sched_barrier(0);
        stage_part(nsrc, nslot, 1);
        __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int j = 0; j < TJ; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
#if defined(EGNN_HL_ILV) && EGNN_HL_ILV
        __builtin_amdgcn_sched_barrier(0);
        stage_part(nsrc, nslot, 2);
        __builtin_amdgcn_sched_barrier(0);
#endif
        slot = slot + 1 == STAGES ? 0 : slot + 1;
    }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // drain the dummy DMAs before the LDS is released
#if EGNN_HL_PRIO
    asm volatile("s_setprio 0");
#endif

#if defined(EGNN_HL_ABL) && (EGNN_HL_ABL & 16)
    {                                                                  // ablation: no epilogue at all (keeps the accumulators alive)
        float sacc = 0.f;
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int j = 0; j < TJ; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) sacc += acc[i][j][r];
        if (sacc == 12345.678f && C) C[0] = sacc;
        return;
    }
#endif
#ifndef EGNN_HL_STAGED
#define EGNN_HL_STAGED 1
#endif
#if EGNN_HL_STAGED
    // ---- epilogue of a full tile, staged through LDS (the ring is free now).  The MFMA leaves a lane one COLUMN of its 64 x 64
    // sub-tile: storing from there means 64 scalar store instructions per wave (two 128-byte row pieces each; 2-byte scatters for the
    // packed (hi, lo) output) behind ~18 VALU instructions of addressing / bounds / range checks per element -- measured (ablations,
    // profiles/r04_experiments/gemm_ablations.txt): the epilogue was 19 % of the projection GEMM and 30 - 37 % of the two node_mlp
    // GEMMs, almost all of it the stores.  Here each wave transposes 32 rows at a time through a private 8.5 KB strip: transformed
    // values go in as the MFMA holds them (conflict-free ds_write_b32), come out row-major (ds_read_b128) and leave as 16-byte stores
    // that fill whole lines -- fp32 rows: 4 rows x 256 B per instruction; packed (hi, lo) images: one contiguous 1 KB (row block, K-tile)
    // piece per instruction.  Same arithmetic per element, same bits.  Partial tiles (M or N edge) keep the per-element path below.
    if (m0 + BM <= M && n0 + BN <= N && (ldc & 3) == 0 && (ldr & 3) == 0 && (reinterpret_cast<uintptr_t>(C) & 15) == 0 &&
        (reinterpret_cast<uintptr_t>(R) & 15) == 0 && (split_cols & 3) == 0) {
        static_assert(TJ == 2, "the staged epilogue reads 64-column strips");
        __syncthreads();                                               // every wave has left the ring (no DMA in flight: vmcnt(0) above)
        const int LD = Chi ? 68 : 64;                                  // strip row stride in dwords (packed output: rows spread over banks)
        float* const stg = reinterpret_cast<float*>(smem) + wave * (32 * 68);
        const int c = lane & 31, q4 = 4 * (lane >> 5);
        const int nw0 = n0 + wn * (TJ * 32);                           // first column / row of this wave's sub-tile
        const int64_t mw0 = m0 + wm * (TI * 32);
        float bv[TJ];
#pragma unroll
        for (int j = 0; j < TJ; ++j) bv[j] = bias ? bias[nw0 + 32 * j + c] : 0.f;
        bool bad_p = false, bad_a = false;
#pragma unroll
        for (int i = 0; i < TI; ++i) {
            // 1. transform, in the accumulator layout: lane = column c of block j, register r = row (r&3) + 8 (r>>2) + 4 (lane>>5)
#pragma unroll
            for (int j = 0; j < TJ; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (r & 3) + 8 * (r >> 2) + q4;
                    float x = acc[i][j][r] * out_scale + bv[j];
                    if constexpr (DROP)
                        x = egnn_drop_hash(egnn_drop_base(drop.seed, EGNN_DROP_SITE_NODE, (uint32_t)(mw0 + i * 32 + row)), (uint32_t)(nw0 + 32 * j + c)) >= drop.thr
                                ? x * drop.inv_keep : 0.f;
                    if (ACT == 1) x = egnn_silu(x);
                    if (ACT == 2) x = 0.5f * x * (1.0f + erff(x * 0.70710678118654752f));
                    stg[row * LD + 32 * j + c] = x;
                }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            // 2a. fp32 rows: lane -> 4 consecutive columns of row (lane >> 4) + 4 k
            if (C) {
                const int col4 = 4 * (lane & 15);
                const int gn = nw0 + col4;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int row = (lane >> 4) + 4 * k;
                    const int64_t gm = mw0 + i * 32 + row;
                    f32x4 v = *reinterpret_cast<const f32x4*>(stg + row * LD + col4);
                    if (HAS_RES) v += *reinterpret_cast<const f32x4*>(R + gm * ldr + gn);
                    if (gn < split_cols) {                               // these columns as (fp16 hi, fp16 lo) words
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const float x = v[u];
                            bad_p = bad_p || egnn_beyond_f16(x);
                            const _Float16 h = (_Float16)x;
                            typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
                            const f16x2 w = {h, (_Float16)(x - (float)h)};
                            v[u] = __builtin_bit_cast(float, w);
                        }
                    }
#if !(defined(EGNN_HL_ABL) && (EGNN_HL_ABL & 1))
                    // the projection table (the call with split_cols: 1.09 GB at the north-star shape, read back by the edge pass long
                    // after it has left every cache) goes out non-temporal: its lines no longer push the A / W panels out of L2 (round 6,
                    // one process: north star -2.7 %, c5 -5 %, c3 -13 %; node_mlp's outputs, which the next kernel reads at once: +4 %, left alone)
                    if (EGNN_HL_NTSTORE == 1 || (EGNN_HL_NTSTORE == 2 && split_cols > 0))
                        __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(C + gm * ldc + gn));
                    else
                        *reinterpret_cast<f32x4*>(C + gm * ldc + gn) = v;
#else
                    if (v[0] == 123.456f) *reinterpret_cast<f32x4*>(C + gm * ldc + gn) = v;
#endif
                }
            }
            // 2b. packed (hi, lo) images: one 1 KB (row block, K-tile) piece per instruction -- lane l = bytes [16 l, 16 l + 16) of the
            // piece = row l >> 1, physical chunk l & 1 (logical chunk XOR-swizzled by (row >> 3) & 1, egnn_pk_off)
            if (Chi) {
                const int row = lane >> 1;
                const int ck = (lane & 1) ^ ((row >> 3) & 1);
                const int64_t gm = mw0 + i * 32 + row;
                const int64_t rbg = (mw0 + i * 32) >> 5;
#pragma unroll
                for (int kt = 0; kt < TJ * 2; ++kt) {
                    const int cc = 16 * kt + 8 * ck;
                    f32x4 v0 = *reinterpret_cast<const f32x4*>(stg + row * LD + cc);
                    f32x4 v1 = *reinterpret_cast<const f32x4*>(stg + row * LD + cc + 4);
                    if (HAS_RES) {
                        v0 += *reinterpret_cast<const f32x4*>(R + gm * ldr + nw0 + cc);
                        v1 += *reinterpret_cast<const f32x4*>(R + gm * ldr + nw0 + cc + 4);
                    }
                    f16x8 h8, l8;
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const float x = u < 4 ? v0[u & 3] : v1[u & 3];
                        bad_a = bad_a || egnn_beyond_f16(x);
                        const _Float16 h = (_Float16)x;
                        h8[u] = h;
                        l8[u] = (_Float16)(x - (float)h);
                    }
                    const size_t o = (size_t)((rbg * nkt_out + (nw0 >> 4) + kt) * 512 + lane * 8);
#if EGNN_HL_NTSTORE == 1
                    __builtin_nontemporal_store(h8, reinterpret_cast<f16x8*>(Chi + o));
                    __builtin_nontemporal_store(l8, reinterpret_cast<f16x8*>(Clo + o));
#else
                    *reinterpret_cast<f16x8*>(Chi + o) = h8;
                    *reinterpret_cast<f16x8*>(Clo + o) = l8;
#endif
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         // the strip is read before the next 32 rows overwrite it
            __builtin_amdgcn_wave_barrier();
        }
        egnn_flag_range(status, bad_p, EGNN_RANGE_PROJ);
        egnn_flag_range(status, bad_a, EGNN_RANGE_A_OPERAND);
        return;
    }
#endif
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
                if constexpr (DROP)                                 // nn.Dropout behind the Linear (egnn_pytorch.py:196-201); its own
                    x = egnn_drop_hash(egnn_drop_base(drop.seed, EGNN_DROP_SITE_NODE, (uint32_t)gm), (uint32_t)gn) >= drop.thr ? x * drop.inv_keep : 0.f;
                if (ACT == 1) x = egnn_silu(x);
                if (ACT == 2) x = 0.5f * x * (1.0f + erff(x * 0.70710678118654752f));     // exact GELU (nn.GELU default, :130)
                if (HAS_RES) x += R[gm * ldr + gn];
#if defined(EGNN_HL_ABL) && (EGNN_HL_ABL & 1)
                if (x == 123.456f)                                   // ablation: no output stores
#endif
                if (C) {
                    if (gn < split_cols) {                            // this column as an (fp16 hi, fp16 lo) word
                        egnn_flag_range(status, egnn_beyond_f16(x), EGNN_RANGE_PROJ);
                        const _Float16 h = (_Float16)x;
                        typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
                        const f16x2 w = {h, (_Float16)(x - (float)h)};
                        C[gm * ldc + gn] = __builtin_bit_cast(float, w);
                    } else {
                        C[gm * ldc + gn] = x;
                    }
                }
                if (Chi) {
                    egnn_flag_range(status, egnn_beyond_f16(x), EGNN_RANGE_A_OPERAND);
                    const _Float16 h = (_Float16)x;
                    const size_t o = egnn_pk_off(gm, gn, nkt_out);
                    Chi[o] = h;
                    Clo[o] = (_Float16)(x - (float)h);
                }
            }
        }
    }
}

template <int CFG, int ACT, bool HAS_RES, bool DROP = false>
__global__ __launch_bounds__(Cfg<CFG>::WM * Cfg<CFG>::WN * 64, 2) void linear_hl_kernel(
    const _Float16* __restrict__ Ahi, const _Float16* __restrict__ Alo,
    const _Float16* __restrict__ Whi, const _Float16* __restrict__ Wlo,
    const float* __restrict__ bias, const float* __restrict__ R, int64_t ldr,
    float* __restrict__ C, int64_t ldc, _Float16* __restrict__ Chi, _Float16* __restrict__ Clo, int nkt_out,
    int64_t M, int N, int Kp, int ntm, int ntn, float out_scale, int split_cols, int32_t* __restrict__ status, const DropArgs drop, const int a_nkt,
    const int group_m)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];      // STAGES x BUF
    linear_hl_body<CFG, ACT, HAS_RES, DROP>(Ahi, Alo, Whi, Wlo, bias, R, ldr, C, ldc, Chi, Clo, nkt_out, M, N, Kp, ntm, ntn,
                                            out_scale, split_cols, status, smem, blockIdx.x, 0, -1, drop, a_nkt, group_m);
}

// Split-K: blockIdx.y = part; the part's partial product goes to its own (M, ldc) slab (summed afterwards in fixed order)
template <int CFG>
__global__ __launch_bounds__(Cfg<CFG>::WM * Cfg<CFG>::WN * 64, 2) void linear_hl_splitk_kernel(
    const _Float16* __restrict__ Ahi, const _Float16* __restrict__ Alo, const _Float16* __restrict__ Whi, const _Float16* __restrict__ Wlo,
    float* __restrict__ Cpart, int64_t ldc, int64_t M, int N, int Kp, int ntm, int ntn, float out_scale, int tiles_per_part)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int part = blockIdx.y;
    const int nkt = Kp / BK;
    const int kt0 = part * tiles_per_part;
    int cnt = nkt - kt0;
    if (cnt > tiles_per_part) cnt = tiles_per_part;
    linear_hl_body<CFG, 0, false>(Ahi, Alo, Whi, Wlo, nullptr, nullptr, 0, Cpart + (size_t)part * M * ldc, ldc, nullptr, nullptr, 0,
                                  M, N, Kp, ntm, ntn, out_scale, 0, nullptr, smem, blockIdx.x, kt0, cnt);
}

__global__ __launch_bounds__(256) void sum_parts_kernel(const float* __restrict__ parts, int nparts, int64_t count, float scale, float* __restrict__ out)
{
    for (int64_t o = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; o < count; o += (int64_t)gridDim.x * 1024) {
        f32x4 acc = *reinterpret_cast<const f32x4*>(parts + o);
#pragma unroll 8
        for (int p = 1; p < nparts; ++p) acc += *reinterpret_cast<const f32x4*>(parts + (size_t)p * count + o);     // fixed order
        *reinterpret_cast<f32x4*>(out + o) = acc * scale;
    }
}

// M-tiles per group of the block -> tile map (the tiles of a group -- group_m M-tiles x all N-tiles, M fastest -- are consecutive block
// indices of one XCD).  EGNN_HL_GROUP_M in the environment overrides the build default (tools/gemm_lab.py gm=... sweeps it in one process).
// Measured in round 6 (profiles/r06_experiments/gemm_group_m.txt, every value on every GEMM shape of the north star, c3 and c5 in one
// process): the best group holds ~128 tiles -- 4 M-tiles for the projection's 33 N-tiles (-3 % against the former constant 8), 16 for
// node_mlp.0's 8 (-3 %), flat for four N-tiles and fewer, 32 M-tiles always worse.
static int hl_group_m(int ntm, int ntn)
{
    (void)ntm;
    if (const char* e = getenv("EGNN_HL_GROUP_M")) {
        const int v = atoi(e);
        if (v >= 1 && v <= 1024) return v;
    }
    if (ntn <= 4) return GROUP_M;
    const int g = (128 + ntn / 2) / ntn;
    return g < 2 ? 2 : (g > 16 ? 16 : g);
}

template <int CFG, int ACT, bool HAS_RES>
int launch_hl_cfg(const _Float16* Ahi, const _Float16* Alo, const _Float16* Whi, const _Float16* Wlo,
                  const float* bias, const float* R, int64_t ldr, float* C, int64_t ldc, _Float16* Chi, _Float16* Clo,
                  int nkt_out, int64_t M, int N, int Kp, float out_scale, int split_cols, int32_t* status, hipStream_t s,
                  const DropArgs drop = DropArgs{0u, 0u, 1.f}, const int a_nkt = 0)
{
    using C_ = Cfg<CFG>;
    const int64_t ntm = (M + C_::BM - 1) / C_::BM;
    const int64_t ntn = (N + C_::BN - 1) / C_::BN;
    if (ntm * ntn > 0x7fffffffLL) return EGNN_E_UNSUPPORTED;
    const size_t lds = (size_t)C_::STAGES * (2 * C_::BM + 2 * C_::BN) * ROWB;
    const int gm = hl_group_m((int)ntm, (int)ntn);
    // (the dropout epilogue -- a hash per output element -- is its own instantiation: as a run-time branch in the common kernel it cost
    // the residual GEMM 18 % (0.25 -> 0.30 ms) through register pressure alone)
    if constexpr (ACT == 1 && !HAS_RES) {
        if (drop.thr) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(linear_hl_kernel<CFG, ACT, HAS_RES, true>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return (int)e;
            hipLaunchKernelGGL((linear_hl_kernel<CFG, ACT, HAS_RES, true>), dim3((unsigned)(ntm * ntn)), dim3(C_::WM * C_::WN * 64), lds, s,
                               Ahi, Alo, Whi, Wlo, bias, R, ldr, C, ldc, Chi, Clo, nkt_out, M, N, Kp, (int)ntm, (int)ntn, out_scale, split_cols, status, drop, a_nkt, gm);
            return egnn_launch_status();
        }
    } else if (drop.thr) {
        return EGNN_E_UNSUPPORTED;                       // dropout sits behind node_mlp's first Linear only (SiLU, no residual)
    }
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(linear_hl_kernel<CFG, ACT, HAS_RES>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL((linear_hl_kernel<CFG, ACT, HAS_RES>), dim3((unsigned)(ntm * ntn)), dim3(C_::WM * C_::WN * 64), lds, s,
                       Ahi, Alo, Whi, Wlo, bias, R, ldr, C, ldc, Chi, Clo, nkt_out, M, N, Kp, (int)ntm, (int)ntn, out_scale, split_cols, status, drop, a_nkt, gm);
    return egnn_launch_status();
}

template <int ACT, bool HAS_RES>
int launch_hl(const _Float16* Ahi, const _Float16* Alo, const _Float16* Whi, const _Float16* Wlo,
              const float* bias, const float* R, int64_t ldr, float* C, int64_t ldc, _Float16* Chi, _Float16* Clo,
              int nkt_out, int64_t M, int N, int Kp, float out_scale, int w_rows, int split_cols, int32_t* status, hipStream_t s,
              const DropArgs drop = DropArgs{0u, 0u, 1.f}, const int a_nkt = 0)
{
    // Large problems (enough 256 x 128 tiles to fill the chip twice) use the larger tile; small ones the 128 x 128 tile.
    constexpr int BIG = EGNN_HL_CFG;
    const int64_t tbig = ((M + Cfg<BIG>::BM - 1) / Cfg<BIG>::BM) * ((N + Cfg<BIG>::BN - 1) / Cfg<BIG>::BN);
    if (BIG != 0 && tbig >= 512 && w_rows >= (N + Cfg<BIG>::BN - 1) / Cfg<BIG>::BN * Cfg<BIG>::BN)
        return launch_hl_cfg<BIG, ACT, HAS_RES>(Ahi, Alo, Whi, Wlo, bias, R, ldr, C, ldc, Chi, Clo, nkt_out, M, N, Kp, out_scale, split_cols, status, s, drop, a_nkt);
    return launch_hl_cfg<0, ACT, HAS_RES>(Ahi, Alo, Whi, Wlo, bias, R, ldr, C, ldc, Chi, Clo, nkt_out, M, N, Kp, out_scale, split_cols, status, s, drop, a_nkt);
}

}  // namespace

static int linear_hl_entry(const void* A_hi, const void* A_lo, const void* W_hi, const void* W_lo,
                           float w_inv_scale, const float* bias, const float* residual, int64_t ldr,
                           float* C, int64_t ldc, void* C_hi, void* C_lo, int Kp_out, int64_t M, int N, int Kp,
                           int w_rows, int act, int split_cols, int32_t* status, void* stream, const DropArgs drop, const int Kp_a = 0)
{
    if (Kp_a && (Kp_a < Kp || (Kp_a % 32) != 0)) return EGNN_E_SHAPE;
    const int a_nkt = Kp_a / 16;
    if (!A_hi || !A_lo || !W_hi || !W_lo) return EGNN_E_NULLPTR;
    if (!C && !C_hi) return EGNN_E_NULLPTR;
    if ((C_hi == nullptr) != (C_lo == nullptr)) return EGNN_E_NULLPTR;
    if (M <= 0 || N <= 0 || Kp <= 0 || (Kp % 32) != 0) return EGNN_E_SHAPE;
    if (w_rows < (N + 127) / 128 * 128) return EGNN_E_SHAPE;          // W images must cover whole 128-row tiles
    if (C && ldc < N) return EGNN_E_SHAPE;
    if (C_hi && (Kp_out < N || (Kp_out % 32) != 0)) return EGNN_E_SHAPE;
    if (residual && ldr < N) return EGNN_E_SHAPE;
    if (act < 0 || act > 2) return EGNN_E_UNSUPPORTED;
    if (act == 2 && residual) return EGNN_E_UNSUPPORTED;
    if (!(w_inv_scale > 0.f)) return EGNN_E_SHAPE;
    if (split_cols < 0 || split_cols > N || (split_cols % 32) != 0 || (split_cols && (!C || residual))) return EGNN_E_SHAPE;
    if ((reinterpret_cast<uintptr_t>(A_hi) & 15) || (reinterpret_cast<uintptr_t>(A_lo) & 15) ||
        (reinterpret_cast<uintptr_t>(W_hi) & 15) || (reinterpret_cast<uintptr_t>(W_lo) & 15))
        return EGNN_E_ALIGN;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const _Float16 *ah = static_cast<const _Float16*>(A_hi), *al = static_cast<const _Float16*>(A_lo);
    const _Float16 *wh = static_cast<const _Float16*>(W_hi), *wl = static_cast<const _Float16*>(W_lo);
    _Float16 *ch = static_cast<_Float16*>(C_hi), *cl = static_cast<_Float16*>(C_lo);
    const int nkt_out = Kp_out / 16;
    if (act == 0) {
        if (residual) return launch_hl<0, true>(ah, al, wh, wl, bias, residual, ldr, C, ldc, ch, cl, nkt_out, M, N, Kp, w_inv_scale, w_rows, split_cols, status, s, drop, a_nkt);
        return launch_hl<0, false>(ah, al, wh, wl, bias, residual, ldr, C, ldc, ch, cl, nkt_out, M, N, Kp, w_inv_scale, w_rows, split_cols, status, s, drop, a_nkt);
    }
    if (act == 2) return launch_hl<2, false>(ah, al, wh, wl, bias, residual, ldr, C, ldc, ch, cl, nkt_out, M, N, Kp, w_inv_scale, w_rows, split_cols, status, s, drop, a_nkt);
    if (residual) return launch_hl<1, true>(ah, al, wh, wl, bias, residual, ldr, C, ldc, ch, cl, nkt_out, M, N, Kp, w_inv_scale, w_rows, split_cols, status, s, drop, a_nkt);
    return launch_hl<1, false>(ah, al, wh, wl, bias, residual, ldr, C, ldc, ch, cl, nkt_out, M, N, Kp, w_inv_scale, w_rows, split_cols, status, s, drop, a_nkt);
}

extern "C" int egnn_linear_hl_f32(const void* A_hi, const void* A_lo, const void* W_hi, const void* W_lo,
                                  float w_inv_scale, const float* bias, const float* residual, int64_t ldr,
                                  float* C, int64_t ldc, void* C_hi, void* C_lo, int Kp_out, int64_t M, int N, int Kp,
                                  int w_rows, int act, int split_cols, int32_t* status, void* stream)
{
    return linear_hl_entry(A_hi, A_lo, W_hi, W_lo, w_inv_scale, bias, residual, ldr, C, ldc, C_hi, C_lo, Kp_out, M, N, Kp, w_rows, act,
                           split_cols, status, stream, DropArgs{0u, 0u, 1.f});
}

// ... with the A image wider than the contraction: Kp_a (>= Kp, % 32 == 0) is the K padding the image was written with, the product
// runs over its first Kp columns (the projection reads feats out of the [feats | m_i] image of node_mlp's input: no second image)
extern "C" int egnn_linear_hl_lda_f32(const void* A_hi, const void* A_lo, int Kp_a, const void* W_hi, const void* W_lo,
                                      float w_inv_scale, const float* bias, const float* residual, int64_t ldr,
                                      float* C, int64_t ldc, void* C_hi, void* C_lo, int Kp_out, int64_t M, int N, int Kp,
                                      int w_rows, int act, int split_cols, int32_t* status, void* stream)
{
    return linear_hl_entry(A_hi, A_lo, W_hi, W_lo, w_inv_scale, bias, residual, ldr, C, ldc, C_hi, C_lo, Kp_out, M, N, Kp, w_rows, act,
                           split_cols, status, stream, DropArgs{0u, 0u, 1.f}, Kp_a);
}

// ... and with a row mask: M-tiles (128 or 256 rows) none of whose rows has row_mask != 0 are skipped -- their rows of C are NOT written.
// For the projection table of a padded batch (egnn_pytorch.py:279-287 factorised): a padded node's rows are read by masked-out edges
// only, whose values the edge pass drops by select (csrc/edge_pw.hip, csrc/edge_fused.hip), never under autograd (the backward
// differentiates through every edge's u).
extern "C" int egnn_linear_hl_lda_rows_f32(const void* A_hi, const void* A_lo, int Kp_a, const void* W_hi, const void* W_lo,
                                           float w_inv_scale, const float* bias, const float* residual, int64_t ldr,
                                           float* C, int64_t ldc, void* C_hi, void* C_lo, int Kp_out, int64_t M, int N, int Kp,
                                           int w_rows, int act, int split_cols, const uint8_t* row_mask, int32_t* status, void* stream)
{
    return linear_hl_entry(A_hi, A_lo, W_hi, W_lo, w_inv_scale, bias, residual, ldr, C, ldc, C_hi, C_lo, Kp_out, M, N, Kp, w_rows, act,
                           split_cols, status, stream, DropArgs{0u, 0u, 1.f, row_mask}, Kp_a);
}

extern "C" int egnn_linear_hl_drop_f32(const void* A_hi, const void* A_lo, const void* W_hi, const void* W_lo,
                                       float w_inv_scale, const float* bias, const float* residual, int64_t ldr,
                                       float* C, int64_t ldc, void* C_hi, void* C_lo, int Kp_out, int64_t M, int N, int Kp,
                                       int w_rows, int act, int split_cols, uint32_t drop_thr, uint32_t drop_seed, float drop_inv_keep,
                                       int32_t* status, void* stream)
{
    if (drop_thr && !(drop_inv_keep >= 1.f)) return EGNN_E_SHAPE;
    if (M > 0xffffffffLL) return EGNN_E_UNSUPPORTED;                          // the mask's row counter is 32 bits
    return linear_hl_entry(A_hi, A_lo, W_hi, W_lo, w_inv_scale, bias, residual, ldr, C, ldc, C_hi, C_lo, Kp_out, M, N, Kp, w_rows, act,
                           split_cols, status, stream, DropArgs{drop_thr, drop_seed, drop_inv_keep});
}

extern "C" int egnn_linear_hl_splitk_f32(const void* A_hi, const void* A_lo, const void* W_hi, const void* W_lo, float w_inv_scale,
                                         float* C_parts, int64_t ldc, int64_t M, int N, int Kp, int w_rows, int k_splits, void* stream)
{
    if (!A_hi || !A_lo || !W_hi || !W_lo || !C_parts) return EGNN_E_NULLPTR;
    if (M <= 0 || N <= 0 || Kp <= 0 || (Kp % 32) != 0 || ldc < N || k_splits < 1) return EGNN_E_SHAPE;
    if (w_rows < (N + 127) / 128 * 128 || !(w_inv_scale > 0.f)) return EGNN_E_SHAPE;
    const int nkt = Kp / BK;
    const int per = (nkt + k_splits - 1) / k_splits;
    if ((int64_t)per * (k_splits - 1) >= nkt) return EGNN_E_SHAPE;                 // every part must own at least one K-tile
    using C_ = Cfg<0>;
    const int64_t ntm = (M + C_::BM - 1) / C_::BM, ntn = (N + C_::BN - 1) / C_::BN;
    if (ntm * ntn > 0x7fffffffLL || k_splits > 65535) return EGNN_E_UNSUPPORTED;
    const size_t lds = (size_t)C_::STAGES * (2 * C_::BM + 2 * C_::BN) * ROWB;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(linear_hl_splitk_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL((linear_hl_splitk_kernel<0>), dim3((unsigned)(ntm * ntn), (unsigned)k_splits), dim3(C_::WM * C_::WN * 64), lds,
                       static_cast<hipStream_t>(stream), static_cast<const _Float16*>(A_hi), static_cast<const _Float16*>(A_lo),
                       static_cast<const _Float16*>(W_hi), static_cast<const _Float16*>(W_lo), C_parts, ldc, M, N, Kp, (int)ntm, (int)ntn,
                       w_inv_scale, per);
    return egnn_launch_status();
}

extern "C" int egnn_sum_parts_f32(const float* parts, int nparts, int64_t count, float scale, float* out, void* stream)
{
    if (!parts || !out) return EGNN_E_NULLPTR;
    if (nparts < 1 || count <= 0 || (count % 4) != 0) return EGNN_E_SHAPE;
    if ((reinterpret_cast<uintptr_t>(parts) & 15) || (reinterpret_cast<uintptr_t>(out) & 15)) return EGNN_E_ALIGN;
    int64_t blocks = (count / 4 + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(sum_parts_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), parts, nparts, count, scale, out);
    return egnn_launch_status();
}

extern "C" int64_t egnn_packed_halves(int64_t rows, int Kp) { return (rows + 31) / 32 * 32 * (int64_t)Kp; }

extern "C" int egnn_split_f16(const float* X, int64_t ldx, int64_t rows, int cols, void* hi, void* lo, int Kp, int32_t* status, void* stream)
{
    if (!X || !hi || !lo) return EGNN_E_NULLPTR;
    if (rows <= 0 || cols <= 0 || ldx < cols || Kp < cols || (Kp % 32) != 0) return EGNN_E_SHAPE;
    return egnn_pack_rows_launch(X, ldx, nullptr, nullptr, nullptr, 0.f, hi, lo, Kp, nullptr, nullptr, 0, rows, cols, 0, status, stream);
}
