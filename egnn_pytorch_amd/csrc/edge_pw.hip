// The edge pass of the k-NN layers -- and of dense all-pairs layers with N % 32 == 0 -- with ONE WAVE PER NODE (reference: egnn_pytorch/egnn_pytorch.py:262-333).
//
// Same arithmetic, same operand layouts and -- for K <= 128 -- the same bits as edge_fused.hip's general kernel, for the shape every
// k-NN configuration of BASELINE.json has: K % 32 == 0 neighbours, squared distance as the only per-edge scalar (no fourier
// features, no edge features), m_dim <= 16, 3-D coordinates, the per-slot records of egnn_slot_prep_f32, no training-mode dropout
// (inference, and the forward under autograd: args.U_out).  The hidden loop is the general kernel's; what differs is everything
// AROUND it, which at 32 neighbours was 16 % of the general kernel's VALU instructions at dim 512 and 40 % at dim 128 (505 M issued
// against 426 M in the loop, profiles/r03_final) -- here 483 M / 159 M against 505 M / 173 M (profiles/r04_final):
//   * one wave owns one node: its 32 k-slots per round are the wave's two MFMA tiles, rounds (K / 32) run back to back in the same
//     wave and the per-node sums stay in registers (DPP butterfly over a tile's 16 edges) -- no cross-wave reduction, no LDS
//     accumulators, no workgroup barrier outside the staging ring's one per chunk of 64 hidden units;
//   * the setup is written for this shape only (the general kernel carries fourier / edge-feature / dense / ragged-K paths as
//     run-time branches through every tile); a round's 32 slot records (512 B) arrive by LDS-DMA in a wave-private buffer -- the next
//     round's while the current one computes -- and the epilogue re-reads x_i - x_j from the same LDS copy;
//   * kernel arguments are read per phase through a pointer the optimiser cannot see through (pw_args): nothing of the ~40 fields
//     stays live -- or spilled -- across the hidden loop; per-lane addresses of the epilogue constants are not hoisted (pw_opaque);
//     the staging ring's LDS-DMA uses a scalar base + 32-bit lane offset (no 64-bit vector address arithmetic per chunk);
//   * the residual of the hidden value's hi/lo split runs as v_mfma_f32_4x4x4_16B_f16 (lane-local: D = C - B with A = -I4, two
//     passes on the matrix pipe) instead of v_mfma_f32_16x16x16_f16 (four): bit-identical, half the matrix-pipe time (-2 %);
//   * the layer's last step, of which only 2 hidden units are real whenever dim % 8 == 0, skips its padding block (-2 %);
//   * the node's 16 message channels leave as one 8-byte store per packed image.
// Scheduling: one workgroup (4 waves = 4 Morton-adjacent nodes) per node group, dispatched dynamically.  The kernel CAN walk several
// groups per workgroup (EGNN_PW_GRID_MULT; ring and record prefetch continue across groups) and was first built fully persistent --
// 5 workgroups per CU, equal static shares: 10 % SLOWER (1.54 vs 1.39 ms at the north-star shape, same box): with equal shares
// fixed at launch the five workgroups of a CU stay in step, their latency-bound setups and epilogues coincide instead of hiding under
// each other's hidden loops, and it is the dispatcher's staggered refill that de-phases them (egnn_edge_pw_launch below).
// Round 5 (profiles/r05_experiments/edge_pw_ablations_and_pipelining.txt; code in the history at commit aaff455): software pipelining
// of the step head without extra registers (tile 0 of the next step picked up at the end of the current one), the four SiLU chains of
// a register quad interleaved, conflict-free A-fragment reads, the P_i words through the staging ring instead of two loads per step,
// four workgroups per CU with a barrier per three steps: all within the box's +-1.5 % or slower.  Timing-only ablations (EGNN_PW_ABL):
// without any MFMA 1.10 ms of 1.33 -- the matrix-pipe work costs 0.23 ms of VALU time however it is scheduled; without the chunk
// barrier a dim-128 layer is 2.8x SLOWER (the barrier keeps the four nodes' gathers on the same lines).
// Also measured and not kept (profiles/r04_experiments/): W2 / W_s fragments straight from global memory instead of the LDS ring
// (no barriers, but +2.5 KB of vector-memory traffic per wave-step: +22 %); a 32-column ring at six workgroups per CU (a barrier per
// step: +3 %); the first Linear as one v_mfma_f32_32x32x16_f16 per (tile, 64 hidden units) (+8 %: a tile at a time leaves the wave
// nothing to issue under the MFMA's latency).
#include <type_traits>
#include "egnn_common.h"
#include "egnn_lds_dma.h"

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2v __attribute__((ext_vector_type(2)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4v __attribute__((ext_vector_type(4)));

#ifndef EGNN_PW_WGS
#define EGNN_PW_WGS 5                        // workgroups per CU the register allocation must allow
#endif
#ifndef EGNN_PW_SKIP_MASKED
#define EGNN_PW_SKIP_MASKED 1                // a round none of whose 32 edges contributes (a padded node: mask_i = 0) skips its hidden loop
#endif
#ifndef EGNN_PW_RESID4
#define EGNN_PW_RESID4 1                     // residual of the split on v_mfma_f32_4x4x4_16B_f16 (0: v_mfma_f32_16x16x16_f16)
#endif

constexpr int PW_THREADS = 256;
constexpr int PW_WAVES = 4;
#ifndef EGNN_PW_EARLY_W
#define EGNN_PW_EARLY_W 1
#endif
// experiment: wave priorities -- 1: setup / epilogue high, hidden loop low; 2: the reverse; 0: none (measured: see the header comment)
#ifndef EGNN_PW_PRIO
#define EGNN_PW_PRIO 0
#endif
#ifndef EGNN_PW_HC
#define EGNN_PW_HC 64
#endif
// Timing-only ablations (results are wrong; profiles/r05_experiments/edge_pw_ablations.txt): 1 no gather DMA issue, 2 no pick-up of the
// gathered lines (no vmcnt wait, no exchange-row reads), 4 no first-layer MFMAs, 8 no residual MFMAs, 16 no second-layer MFMAs,
// 32 no transcendentals (a = y), 64 no per-edge epilogue (coors_mlp, second SiLU), 128 no staging ring (no chunk DMA, no barrier),
// 256 no P_i loads, 512 no hi / lo conversions
#ifndef EGNN_PW_ABL
#define EGNN_PW_ABL 0
#endif
constexpr int PW_HC = EGNN_PW_HC;            // hidden columns per slot of the staging ring (64: two steps of 32)
constexpr int PW_XLD = 32;                   // floats per row of the gather exchange buffer (one 128-byte line, chunk-swizzled)
// LDS (31 KB: five workgroups per CU):  W2 fragments, two ring slots | first-layer A fragments, two ring slots |
// per-wave gather exchange rows (32 slots x 128 B) | per-wave slot records, two buffers of 32 x 16 B | per-wave 64-float scratch
constexpr int PW_W2S = 0;
constexpr int PW_WST = PW_W2S + 2 * PW_HC * 64;
constexpr int PW_XCH = PW_WST + 2 * PW_HC * 16;
constexpr int PW_REC = PW_XCH + PW_WAVES * 32 * PW_XLD * 4;
constexpr int PW_SCR = PW_REC + PW_WAVES * 2 * 512;
constexpr int PW_LDS = PW_SCR + PW_WAVES * 256;

__device__ __forceinline__ uint32_t pw_pack_h2(_Float16 a, _Float16 b)
{
    const f16x2 v = {a, b};
    return __builtin_bit_cast(uint32_t, v);
}

// A wave-uniform value the optimiser must treat as changed at this point: keeps per-lane addresses that depend only on kernel
// arguments (epilogue constants, output offsets) from being hoisted out of the persistent loop, where they would sit in -- or be
// spilled from -- vector registers across the hidden loop.
template <typename T>
__device__ __forceinline__ T pw_opaque(T v)
{
    asm volatile("" : "+s"(v));
    return v;
}

__device__ __forceinline__ uint32_t lds_addr_of(const void* p)
{
    return __builtin_amdgcn_readfirstlane((uint32_t)(size_t)(__attribute__((address_space(3))) char*)(char*)p);
}

// Kernel arguments read where they are used, through a pointer the optimiser cannot see through: as ordinary by-value arguments all
// ~40 fields are loaded up front and stay live -- spilled to vector-register lanes -- across the persistent loop; this way the setup
// and the epilogue fetch theirs with a handful of scalar loads per round (scalar cache) and the hidden loop keeps its registers.
typedef const __attribute__((address_space(4))) egnn_edge_args* pw_args_ptr;
__device__ __forceinline__ pw_args_ptr pw_args()
{
    pw_args_ptr a = (pw_args_ptr)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(a));
    return a;
}

// MULTI: K > 32 -- a node's rounds accumulate in registers that live through the hidden loop (K = 32: nothing does)
template <bool MULTI>
__global__ __launch_bounds__(PW_THREADS, EGNN_PW_WGS) void edge_pw_kernel(const egnn_edge_args p)
{
    __shared__ __attribute__((aligned(16))) char smem[PW_LDS];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int e = lane & 15;
    const int g = lane >> 4;
    // (`p` is read at kernel entry only; everything the round loop needs comes through pw_args(), per phase)
    const int N = p.N, K = p.K;
    const int R = MULTI ? (K >> 5) : 1;                       // rounds of 32 k-slots per node
    const int T = p.B * N;                                    // nodes (B N K 16 < 2^32, checked by the launcher)

    // ---- the workgroup's node groups (4 nodes, one per wave): XCD x owns groups [gstart, gstart + gcount), its workgroups interleave
    const int ngroups = (T + PW_WAVES - 1) / PW_WAVES;
    const int nper = (int)(gridDim.x >> 3);
    const int xcd = (int)(blockIdx.x & 7), wi = (int)(blockIdx.x >> 3);
    const int gq = ngroups >> 3, grem = ngroups & 7;
    const int gstart = xcd * gq + (xcd < grem ? xcd : grem);
    const int gcount = gq + (xcd < grem ? 1 : 0);
    if (wi >= gcount) return;                                 // (the whole workgroup)
    const int mygroups = (gcount - wi + nper - 1) / nper;
    const int total_rounds = mygroups * R;

    _Float16* const w2s = reinterpret_cast<_Float16*>(smem + PW_W2S);
    char* const wst = smem + PW_WST;
    float* const xch = reinterpret_cast<float*>(smem + PW_XCH) + wave * (32 * PW_XLD);
    char* const recb = smem + PW_REC + wave * 1024;
    float* const scr = reinterpret_cast<float*>(smem + PW_SCR) + wave * 64;
    const uint32_t xch_lds = lds_addr_of(xch);
    const uint32_t rec_lds = lds_addr_of(recb);

    const int Hp = p.Hp;
    const int nchunks = (Hp + PW_HC - 1) / PW_HC;
#if defined(EGNN_PW_NO_HALF_TAIL) && EGNN_PW_NO_HALF_TAIL
    const bool half_tail = false;
#else
    const bool half_tail = Hp - p.H >= 16;                    // the last step's upper 16 units are padding (exact zeros)
#endif

    // ---- staging ring: chunk c of the hidden dimension (PW_HC columns of W2 fragments + first-layer A fragments) -> slot
    const char* const w2h_g = reinterpret_cast<const char*>(p.W2h);
    const char* const wst_g = reinterpret_cast<const char*>(p.Wst);
    const uint32_t lane16 = (uint32_t)lane * 16u;
    auto stage = [&](int c, int slot) {
        const int c0s = c * PW_HC;
        const int hcs = (Hp - c0s) < PW_HC ? (Hp - c0s) : PW_HC;
        // (wave-uniform base + per-lane 32-bit offset: no vector address arithmetic per piece)
        const char* src = w2h_g + (size_t)c0s * 64;
        char* dst = reinterpret_cast<char*>(w2s) + slot * (PW_HC * 64);
        for (int pc = wave; pc < hcs / 16; pc += PW_WAVES) lds_dma16_s(src + pc * 1024, lane16, dst + pc * 1024);
        if (wave == 0) {
            const int tbytes = hcs * 16;                                   // one 16-byte row of four terms per hidden unit
            for (int o = 0; o < tbytes; o += 1024)                         // (one instruction per 64 hidden units)
                if ((int)lane16 + o < tbytes) lds_dma16_s(wst_g + (size_t)c0s * 16 + o, lane16, wst + slot * (PW_HC * 16) + o);
        }
    };
    // the 32 slot records of round r of node tau -> record buffer `buf` (lanes 0 .. 31: 512 bytes)
    auto rec_dma = [&](int tau, int r, int buf) {
        const pw_args_ptr a = pw_args();
        const u32x4s slot_words = make_rsrc_words(a->slots, (uint32_t)((size_t)a->B * a->N * a->K * 16));
        const uint32_t soff = ((uint32_t)tau * (uint32_t)a->K + 32u * (uint32_t)r) * 16u;
        if (lane < 32) gather_dma16(slot_words, (uint32_t)lane * 16u, soff, rec_lds + (uint32_t)buf * 512u);
    };
    // node of this wave in the workgroup's kg-th group (a wave past the last node repeats the last one and stores nothing)
    auto node_of = [&](int kg) { return (gstart + wi + kg * nper) * PW_WAVES + wave; };
    // its feature row i = order[tau], requested a whole round before it is needed: the load is made to look per-lane (an opaque zero in
    // the address) so that it stays an ordinary vector load whose wait sits at the first use -- as a uniform load the compiler reads
    // it back into a scalar register, and waits for it, on the spot
    int zero_v = 0;
    asm volatile("" : "+v"(zero_v));
    auto row_of = [&](int tau) {
        const pw_args_ptr a = pw_args();
        return a->order ? a->order[tau + zero_v] : tau % a->N;
    };

    // ---- lane constants
    // pick-up: lane (e, g) reads hidden rows 16 hb + 4 g .. + 3 of slot 16 t + e; the swizzle depends on e only
    const float* xr[2];
#pragma unroll
    for (int hb = 0; hb < 2; ++hb) xr[hb] = xch + e * PW_XLD + 4 * ((4 * hb + g) ^ ((e >> 1) & 7));
    // first-layer A fragments: row (hidden unit) e of the 16-block, split term g
    // (units e and e + 8 of a 16-block are 32 dwords apart: read at position g they would share a bank -- a 2-way conflict on this
    // ds_read_b32, the 12 % of LDS cycles VERDICT r4 asked about (16.9 M of 138 M).  The table therefore stores the terms of units
    // 8 .. 15 of a block with the pairs (0, 1) and (2, 3) swapped (_weights.scalar_table / egnn_pack_weights_host) and the lane reads
    // position g ^ 2 there: SQ_LDS_BANK_CONFLICT 0, same time -- the LDS pipe is busy a fifth of it -- profiles/r05_experiments/)
    const char* const tl = wst + (e * 4 + (g ^ ((e >> 2) & 2))) * 4;
    constexpr int tstep = 16 * 4 * 4;                                      // bytes per 16 hidden units
#if EGNN_PW_RESID4
    f16x4 neg_identity;                                                    // A operand of the 4x4x4 residual MFMA: row (lane & 3) of -I4
#pragma unroll
    for (int u = 0; u < 4; ++u) neg_identity[u] = ((lane & 3) == u) ? (_Float16)-1.f : (_Float16)0.f;
#else
    f16x4 neg_identity;                                                    // -I16: row e, K-slots 4g .. 4g+3
#pragma unroll
    for (int u = 0; u < 4; ++u) neg_identity[u] = (e == 4 * g + u) ? (_Float16)-1.f : (_Float16)0.f;
#endif
#if defined(EGNN_PW_STAGGER) && EGNN_PW_STAGGER
    // experiment: the five workgroups of a CU start together and run rounds of equal length -- are their latency-bound phases (epilogue,
    // setup) phase-locked?  Delay the workgroup by its (presumed) slot on the CU
    for (int q = 0; q < ((wi / 32) % EGNN_PW_WGS) * EGNN_PW_STAGGER; ++q) __builtin_amdgcn_s_sleep(127);
#endif
    // ---- prologue: first chunk of the ring, first records
    stage(0, 0);
    int tau_next = node_of(0);
    bool live_next = tau_next < T;
    if (!live_next) tau_next = T - 1;
    rec_dma(tau_next, 0, 0);
    int i_next_v = row_of(tau_next);
    int ring = 0;                                                          // chunks staged so far - 1 = index of the chunk the loop consumes next

    f32x4 nms = f32x4{0.f, 0.f, 0.f, 0.f};                                 // MULTI: the node's message sums, channels 4g .. 4g+3 (every lane of row g)
    float ncs[4] = {0.f, 0.f, 0.f, 0.f};                                   //        coordinate update (3) and edge count

    int kg = 0, r = 0;
    for (int rho = 0; rho < total_rounds; ++rho) {
        // ------------------------------------------------------------------ round setup
        const pw_args_ptr pa = pw_args();                                   // setup arguments (scalar loads, this round only)
        const int64_t ldp = pa->ldp;
        const int tau = tau_next;
        const bool live = live_next;
        const int b = tau / N;
        const size_t bN = (size_t)b * N;
        const int i = __builtin_amdgcn_readfirstlane(i_next_v);
        const uint32_t prow_bytes = (uint32_t)((size_t)N * ldp * 4);
        const __amdgpu_buffer_rsrc_t pi_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(pa->Pi + bN * ldp), 0, prow_bytes, 0x00020000);
        const u32x4s pj_words = make_rsrc_words(pa->Pj + bN * ldp, prow_bytes);
        const int cur = rho & 1;
        const char* const rb = recb + cur * 512;

        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                    // this round's records have landed
        __builtin_amdgcn_wave_barrier();
        uint32_t goff[4];
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
            const uint32_t j2 = *reinterpret_cast<const uint32_t*>(rb + (8 * qq + (lane >> 3)) * 16) & 0x3fffffffu;   // (bit 31: pair mask, bit 30: group flag)
            // the DMA drops lane l's 16 bytes at position l & 7 of row 8 qq + (l >> 3); the exchange rows' swizzle (chunk c at position
            // c ^ ((row >> 1) & 7)) moves to the global side: fetch the chunk that belongs at that position
            const uint32_t gchunk = (uint32_t)(((lane & 7) ^ ((4 * qq + (lane >> 4)) & 7)) * 16);
            goff[qq] = (uint32_t)((size_t)j2 * ldp * 4) + gchunk;
        }
        u32x4v rec[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) rec[t] = *reinterpret_cast<const u32x4v*>(rb + (16 * t + e) * 16);
        // the first step's gathered lines and P_i words
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) gather_dma16(pj_words, goff[qq], 0u, xch_lds + qq * 1024);
        const uint32_t piw = (uint32_t)(((size_t)i * ldp + e) * 4);
        uint32_t piv[2];
        piv[0] = buf_load1(pi_rsrc, piw, 0);
        piv[1] = buf_load1(pi_rsrc, piw, 64);
        // the records (and, for a new node, the feature row) of the round after this one
        if (rho + 1 < total_rounds) {
            if (r + 1 == R) {
                tau_next = node_of(kg + 1);
                live_next = tau_next < T;
                if (!live_next) tau_next = T - 1;
                i_next_v = row_of(tau_next);
                rec_dma(tau_next, 0, cur ^ 1);
            } else {
                rec_dma(tau, r + 1, cur ^ 1);
            }
        }

        u32x2 bq[2];                                                        // B fragments of the first-layer MFMA
        bool fm[2];                                                         // edge contributes (unmasked)
        // lane-group masks, rebuilt every round (three registers that need not live through the hidden loop) and applied with bitwise
        // operations: as selects on g the compiler turns the split below into divergent branches
        const int gq_ = g + pw_opaque(0);
        const uint32_t gm0 = gq_ == 0 ? 0xffffffffu : 0u, gm1 = gq_ == 1 ? 0xffffffffu : 0u, gm2 = gq_ == 2 ? 0xffffffffu : 0u;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const uint32_t r0 = rec[t][0], r1 = rec[t][1], r2 = rec[t][2], r3 = rec[t][3];   // (by value: see edge_fused.hip)
            const float d = egnn_sqdist_rel(__uint_as_float(r1), __uint_as_float(r2), __uint_as_float(r3));
            // d' = d / ws_scale = 2^10 s1 + r_hi + r_lo: lane group g carries split term g in K-slots 4g+2, 4g+3
            float val = d * pa->ws_inv_scale;
            egnn_flag_range(pa->status, live && fabsf(val) >= 6.0e7f && fabsf(val) < __builtin_inff(), EGNN_RANGE_SCALAR);
            if (fabsf(val) >= 6.0e7f) val = __builtin_nanf("");
            const _Float16 s1 = (_Float16)(val * (1.0f / 1024.0f));
            const float rem = val - (float)s1 * 1024.0f;
            const _Float16 rh = (_Float16)rem;
            const _Float16 rl = (_Float16)(rem - (float)rh);
            u32x2 bw;
            bw[0] = 0x3c003c00u & gm0;                                           // (1, 1) x (P_i hi, P_i lo), lane group 0 only
            bw[1] = (pw_pack_h2(s1, s1) & gm0) | (pw_pack_h2(rh, rh) & gm1) | (pw_pack_h2(rl, (_Float16)0.f) & gm2);
            bq[t] = bw;
            fm[t] = live && (r0 >> 31) != 0u;                                    // mask_i & mask_j & (rank <= radius), or 1 without a mask
        }

        f32x4 acc[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        // A round whose 32 edges are all masked out (the node is padding -- mask_i = 0 --, or past the grid's last node) contributes exact
        // zeros to m_i and to the coordinate update whatever its hidden values are (the epilogue drops masked edges): its hidden loop
        // -- gathers, first-layer MFMAs, 32 x 2080 SiLUs, second-layer MFMAs -- is skipped; the wave keeps its part of the workgroup's
        // staging ring and barriers.  Not under autograd (U_out: the backward reads u of every edge).  Wave-uniform.
        // When all FOUR nodes of the workgroup's group are padding (bit 30 of the first record of every round: egnn_slot_prep_f32 sets it
        // from the four nodes' masks, so the four waves read the same answer), the ring is left alone as well: no barrier, no staging --
        // chunk 0 stays where the previous round (or the prologue) put it.  The Morton order lists padded nodes last (egnn_spatial_order_
        // masked_f32), so that padding fills whole groups.
#if EGNN_PW_SKIP_MASKED
        const bool skip_round = pa->U_out == nullptr && __builtin_amdgcn_ballot_w64(fm[0] || fm[1]) == 0ull;
        const bool skip_group = skip_round && ((__builtin_amdgcn_readfirstlane(*reinterpret_cast<const int*>(rb)) >> 30) & 1) != 0;
#else
        constexpr bool skip_round = false, skip_group = false;
#endif

        // ------------------------------------------------------------------ hidden loop
        const bool last_round = rho + 1 == total_rounds;
        // One step of 32 hidden units for the wave's 32 edges.  HALF: the layer's last step when at most 16 of its units are real
        // (H = 2 (2 dim + 1 + ...) leaves TWO units in the last step whenever dim % 8 == 0 -- every BASELINE.json configuration): the
        // upper 16-unit block is padding -- P, W_s and W2 are exactly zero there, so y = 0, a = 0 / (1 + 1) = 0, hi = lo = 0 -- and its
        // pick-up reads, first-layer MFMAs, 32 SiLU evaluations, conversions and residual MFMAs are skipped: the same bits, 40 VALU
        // instructions instead of 86 in that step (1 step of 17 at dim 128, of 33 at dim 256, of 65 at dim 512).
        auto step = [&](auto half_tag, const int c, const int st, const int slot, const int hoff, const _Float16* w2c, const char* tlc) {
            constexpr bool HALF = decltype(half_tag)::value;
            constexpr int NHB = HALF ? 1 : 2;
            const bool more = hoff + 32 < Hp;
            const int hnext = more ? hoff + 32 : hoff;                       // last step: harmless re-read of the P_i words
            f32x4 x[2][2];
            uint32_t pivn[2];
#if EGNN_PW_EARLY_W
            // this step's staged operands first (in LDS since the chunk's barrier): their LDS latency runs under the wait for the
            // gathered lines instead of after it
            u32x2 a0[2];
#pragma unroll
            for (int hb = 0; hb < NHB; ++hb)
                a0[hb] = u32x2{0u, *reinterpret_cast<const uint32_t*>(tlc + st * 2 * tstep + hb * tstep)};
            const f16x8 whi = *reinterpret_cast<const f16x8*>(w2c + ((st * 2 + 0) * 64 + lane) * 8);
            const f16x8 wlo = *reinterpret_cast<const f16x8*>(w2c + ((st * 2 + 1) * 64 + lane) * 8);
#endif
            // the lines of this step were requested a step ago and land in the wave's exchange rows by themselves
#if EGNN_PW_ABL & 2
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int hb = 0; hb < NHB; ++hb) { x[t][hb] = acc[t] * 0.25f; asm volatile("" : "+v"(x[t][hb])); }
#else
            asm volatile("" ::: "memory");
            __builtin_amdgcn_s_waitcnt(0x0F70);                              // vmcnt(0)
            asm volatile("" ::: "memory");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int hb = 0; hb < NHB; ++hb) x[t][hb] = *reinterpret_cast<const f32x4*>(xr[hb] + t * 16 * PW_XLD);
            // the rows are in registers before the next step's lines may overwrite them
            asm volatile("" ::: "memory");
            __builtin_amdgcn_s_waitcnt(0xC07F);                              // lgkmcnt(0)
            asm volatile("" ::: "memory");
            __builtin_amdgcn_wave_barrier();
#endif
            if (!HALF) {
#if EGNN_PW_ABL & 256
                pivn[0] = piv[0]; pivn[1] = piv[1];
#else
                pivn[0] = buf_load1(pi_rsrc, piw, hnext * 4);
                pivn[1] = buf_load1(pi_rsrc, piw, hnext * 4 + 64);
#endif
            }
#if !(EGNN_PW_ABL & 128)
            if (st == 0) {
                // the next chunk of the ring: the next one of this round, or -- the ring does not drain between the rounds of a
                // workgroup -- the first one of the round that follows (its setup and this round's epilogue run with the chunk in flight)
                if (c + 1 < nchunks) stage(c + 1, slot ^ 1);
                else if (!last_round) stage(0, slot ^ 1);
            }
#endif
#if !(EGNN_PW_ABL & 1)
            if (more) {
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) gather_dma16(pj_words, goff[qq], (uint32_t)(hnext * 4), xch_lds + qq * 1024);
            }
#endif

#if EGNN_PW_EARLY_W
#pragma unroll
            for (int hb = 0; hb < NHB; ++hb) a0[hb][0] = piv[hb];
#else
            u32x2 a0[2];
#pragma unroll
            for (int hb = 0; hb < NHB; ++hb)
                a0[hb] = u32x2{piv[hb], *reinterpret_cast<const uint32_t*>(tlc + st * 2 * tstep + hb * tstep)};
            const f16x8 whi = *reinterpret_cast<const f16x8*>(w2c + ((st * 2 + 0) * 64 + lane) * 8);
            const f16x8 wlo = *reinterpret_cast<const f16x8*>(w2c + ((st * 2 + 1) * 64 + lane) * 8);
#endif
            if (!HALF) {
                piv[0] = pivn[0];
                piv[1] = pivn[1];
            }
            // first Linear of edge_mlp on the matrix cores: x += [P_i | W_s] x [1 | split d]
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int hb = 0; hb < NHB; ++hb) {
#if EGNN_PW_ABL & 4
                    asm volatile("" :: "v"(a0[hb]), "v"(bq[t]));
#else
                    x[t][hb] = __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(f16x4, a0[hb]), __builtin_bit_cast(f16x4, bq[t]), x[t][hb], 0, 0, 0);
#endif
                }
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                // x holds y = -log2(e) * (pre-activation); a = y / (1 + 2^y) = SiLU(pre) / (-ln 2); hi = fp16(a) (IEEE: beyond 65504 ->
                // inf, never a silently saturated number); lo32 = a - hi exactly, on the matrix cores
                f16x8 bhi, blo;
#pragma unroll
                for (int u = 0; u < 8; ++u) { bhi[u] = (_Float16)0.f; blo[u] = (_Float16)0.f; }
#pragma unroll
                for (int hb = 0; hb < NHB; ++hb) {
                    f32x4 a4;
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const float y = x[t][hb][u];
#if EGNN_PW_ABL & 32
                        float h = y;
#else
                        float h = y * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(y));
#endif
                        asm("" : "+v"(h));           // keeps the products scalar (v_pk_mul_f32 costs 9.3 cycles per pair against 2 x 2.8)
                        a4[u] = h;
                    }
#if EGNN_PW_ABL & 512
                    const f16x2 h01 = __builtin_bit_cast(f16x2, a4[0]), h23 = __builtin_bit_cast(f16x2, a4[2]);
#else
                    const f16x2 h01 = __builtin_convertvector((f32x2v){a4[0], a4[1]}, f16x2);
                    const f16x2 h23 = __builtin_convertvector((f32x2v){a4[2], a4[3]}, f16x2);
#endif
                    const f16x4 hi4 = {h01[0], h01[1], h23[0], h23[1]};
#if EGNN_PW_ABL & 8
                    const f32x4 l4 = a4;
                    asm volatile("" :: "v"(neg_identity));
#elif EGNN_PW_RESID4
                    const f32x4 l4 = __builtin_amdgcn_mfma_f32_4x4x4f16(neg_identity, hi4, a4, 0, 0, 0);
#else
                    const f32x4 l4 = __builtin_amdgcn_mfma_f32_16x16x16f16(neg_identity, hi4, a4, 0, 0, 0);
#endif
#if EGNN_PW_ABL & 512
                    const f16x2 l01 = __builtin_bit_cast(f16x2, l4[1]), l23 = __builtin_bit_cast(f16x2, l4[3]);
#else
                    const f16x2 l01 = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(l4[0], l4[1]));
                    const f16x2 l23 = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(l4[2], l4[3]));
#endif
                    bhi[4 * hb + 0] = h01[0]; bhi[4 * hb + 1] = h01[1]; bhi[4 * hb + 2] = h23[0]; bhi[4 * hb + 3] = h23[1];
                    blo[4 * hb + 0] = l01[0]; blo[4 * hb + 1] = l01[1]; blo[4 * hb + 2] = l23[0]; blo[4 * hb + 3] = l23[1];
                }
#if EGNN_PW_ABL & 16
                f32x4 at = acc[t];
                asm volatile("" : "+v"(at) : "v"(whi), "v"(wlo), "v"(bhi), "v"(blo));
                acc[t] = at;
#else
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(whi, bhi, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wlo, bhi, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(whi, blo, acc[t], 0, 0, 0);
#endif
            }
        };
#if EGNN_PW_PRIO == 1
        asm volatile("s_setprio 0");
#elif EGNN_PW_PRIO == 2
        asm volatile("s_setprio 3");
#endif
        if (skip_group) {
        } else if (skip_round) {
            // the ring's protocol without the steps: wait for the chunk, meet the other waves, request this wave's share of the next one
            for (int c = 0; c < nchunks; ++c, ++ring) {
                const int slot = ring & 1;
#if !(EGNN_PW_ABL & 128)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (c + 1 < nchunks) stage(c + 1, slot ^ 1);
                else if (!last_round) stage(0, slot ^ 1);
#endif
            }
        } else
        for (int c = 0; c < nchunks; ++c, ++ring) {
            const int slot = ring & 1;
            const int c0 = c * PW_HC;
            const int hc = (Hp - c0) < PW_HC ? (Hp - c0) : PW_HC;
            // chunk `ring` was requested one chunk ago; the only other loads in flight are the gathers of the coming step
#if !(EGNN_PW_ABL & 128)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();                 // every wave's pieces have landed, and every wave has left the other slot
#endif
            const _Float16* w2c = w2s + slot * (PW_HC * 32);
            const char* tlc = tl + slot * (PW_HC * 16);
            const int nst = hc >> 5;
            for (int st = 0; st < nst; ++st) {
                const int hoff = c0 + st * 32;
                if (half_tail && hoff + 32 >= Hp) step(std::true_type{}, c, st, slot, hoff, w2c, tlc);
                else step(std::false_type{}, c, st, slot, hoff, w2c, tlc);
            }
        }

#if EGNN_PW_PRIO == 1
        asm volatile("s_setprio 3");
#elif EGNN_PW_PRIO == 2
        asm volatile("s_setprio 0");
#endif
        // ------------------------------------------------------------------ per-edge epilogue (registers; channel of (lane group g, register u) = 4 g + u)
        const pw_args_ptr pe = pw_args();                                   // epilogue arguments (scalar loads, this round only)
        const bool has_mask = pe->mask != nullptr;
        int32_t* const status = pe->status;
        const float* const gate_w = pe->gate_w;
        const float* const coors_scale = pe->coors_scale;
        if (skip_round) {
            // every edge of the round is masked out: its sums are exact zeros (what the code below computes for masked edges)
            if (!MULTI) {
                nms = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int c = 0; c < 4; ++c) ncs[c] = 0.f;
            }
        } else {
        f32x4 b2r, gwr = f32x4{0.f, 0.f, 0.f, 0.f};
        float gb = 0.f;
        const int g4 = pw_opaque(4) * g;                                     // (opaque: see pw_opaque)
        b2r = *reinterpret_cast<const f32x4*>(pe->b2 + g4);
        if (gate_w) {
            gwr = *reinterpret_cast<const f32x4*>(gate_w + g4);
            gb = pe->gate_b[0];
        }
        float cscale = 0.f;
        if (coors_scale) cscale = coors_scale[0];
        const float w2_inv_scale = pe->w2_inv_scale;
        float* const u_out = pe->U_out;

        float cw[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            f32x4 m;
            bool bad = false;
            f32x4 uu;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                bad = bad || !(fabsf(acc[t][u]) < __builtin_inff());
                uu[u] = acc[t][u] * w2_inv_scale + b2r[u];
                m[u] = egnn_silu(uu[u]);
            }
            // forward under autograd: u = W2 SiLU(x) + b2, what the backward differentiates from (one 16-float row per edge (b, i, k))
            if (u_out && live) *reinterpret_cast<f32x4*>(u_out + ((bN + i) * (size_t)K + 32 * r + 16 * t + e) * 16 + g4) = uu;
            float part = gwr[0] * m[0] + gwr[1] * m[1] + gwr[2] * m[2] + gwr[3] * m[3];
            egnn_flag_range(status, bad && fm[t], EGNN_RANGE_HIDDEN);
            if (gate_w) {                                                      // soft_edges (:289-290)
                part = egnn_column_sum4(part, scr, lane);
                const float gt = egnn_sigmoid(part + gb);
                m *= gt;
            }
            acc[t] = m;
            cw[t] = 0.f;
        }

#if EGNN_PW_ABL & 64
        const _Float16* const w3h = nullptr;
#else
        const _Float16* const w3h = static_cast<const _Float16*>(pe->W3h);
#endif
        if (w3h) {
            // coors_mlp (:203-208): first Linear (16 -> 64) on the matrix cores, split-f16: lane (e, g) holds channels 4g .. 4g+3 of its
            // edge = the B fragment of v_mfma_f32_16x16x16_f16; A = rows 16 blk + e of W3
            float part[2] = {0.f, 0.f};
            f16x4 mhi[2], mlo[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                bool bad = false;
#pragma unroll
                for (int u = 0; u < 4; ++u) bad = bad || egnn_beyond_f16(acc[t][u]);
                egnn_flag_range(status, bad && fm[t], EGNN_RANGE_MESSAGE);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const _Float16 h = (_Float16)acc[t][u];
                    mhi[t][u] = h;
                    mlo[t][u] = (_Float16)(acc[t][u] - (float)h);
                }
            }
            constexpr int W3LD = 16, W3IMG = 64 * 16;
            const int w3o = e * W3LD + g4;
            const float* const b3p = pe->b3;
            const float* const w4p = pe->W4;
            const float w3_inv_scale = pe->w3_inv_scale;
            const float clampv = pe->clamp;
#pragma unroll
            for (int blk = 0; blk < 4; ++blk) {
                const f32x4 b3 = *reinterpret_cast<const f32x4*>(b3p + 16 * blk + g4);
                const f32x4 w4 = *reinterpret_cast<const f32x4*>(w4p + 16 * blk + g4);
                const f16x4 w3hi = *reinterpret_cast<const f16x4*>(w3h + 16 * blk * W3LD + w3o);
                const f16x4 w3lo = *reinterpret_cast<const f16x4*>(w3h + W3IMG + 16 * blk * W3LD + w3o);
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    f32x4 a2 = f32x4{0.f, 0.f, 0.f, 0.f};
                    a2 = __builtin_amdgcn_mfma_f32_16x16x16f16(w3hi, mhi[t], a2, 0, 0, 0);
                    a2 = __builtin_amdgcn_mfma_f32_16x16x16f16(w3lo, mhi[t], a2, 0, 0, 0);
                    a2 = __builtin_amdgcn_mfma_f32_16x16x16f16(w3hi, mlo[t], a2, 0, 0, 0);
#pragma unroll
                    for (int u = 0; u < 4; ++u) part[t] += w4[u] * egnn_silu(a2[u] * w3_inv_scale + b3[u]);
                }
            }
            const float b4 = pe->b4[0];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                float s = egnn_column_sum4(part[t], scr, lane);
                s += b4;
                if (has_mask && !fm[t]) s = 0.f;                                 // :308-309
                if (clampv >= 0.f) s = fminf(fmaxf(s, -clampv), clampv);         // :311-313
                if (!fm[t] && !has_mask) s = 0.f;                                // a wave past the last node
                cw[t] = s;
            }
        }

        // x_i - x_j again, from the wave's LDS copy of the records
        float rn[2][3], keep[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const u32x4v rc = *reinterpret_cast<const u32x4v*>(rb + (16 * t + e) * 16);
            const uint32_t rw[3] = {rc[1], rc[2], rc[3]};
            float rel[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) rel[c] = __uint_as_float(rw[c]);
            keep[t] = fm[t] ? 1.f : 0.f;
            float inv = 1.f;
            if (coors_scale) {                                                   // CoorsNorm, egnn_pytorch.py:67-77
                float n2 = 0.f;                                                  // (explicit fma chain, as in edge_fused.hip: the same bits)
#pragma unroll
                for (int c = 0; c < 3; ++c) n2 = __builtin_fmaf(rel[c], rel[c], n2);
                inv = cscale / fmaxf(sqrtf(n2), 1e-8f);
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) rn[t][c] = rel[c] * inv;
        }
        // the round's 32 edges belong to this wave's node: sum them in registers (DPP butterfly over the 16 edges of a tile, fixed order)
        const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f};
        f32x4 ms = (fm[0] ? acc[0] : zero4) + (fm[1] ? acc[1] : zero4);          // select: masked_fill semantics (:322)
        float cs[4];
#pragma unroll
        for (int c = 0; c < 3; ++c) cs[c] = __builtin_fmaf(cw[1], rn[1][c], cw[0] * rn[0][c]);
        cs[3] = keep[0] + keep[1];
#pragma unroll
        for (int u = 0; u < 4; ++u) ms[u] = egnn_row16_sum(ms[u]);
#pragma unroll
        for (int c = 0; c < 4; ++c) cs[c] = egnn_row16_sum(cs[c]);
        // rounds in ascending order, each added to a running sum that starts at +0 (the order -- and the bits -- of the general kernel's
        // cross-wave reduction for K <= 128)
        if (MULTI) {
#pragma unroll
            for (int u = 0; u < 4; ++u) nms[u] += ms[u];
#pragma unroll
            for (int c = 0; c < 4; ++c) ncs[c] += cs[c];
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u) nms[u] = 0.f + ms[u];
#pragma unroll
            for (int c = 0; c < 4; ++c) ncs[c] = 0.f + cs[c];
        }
        }

        // ------------------------------------------------------------------ node outputs (after the node's last round)
        if (r + 1 == R) {
            if (live) {
                const size_t row = bN + i;
                const int m_dim = pe->m_dim;
                float* const m_i = pe->m_i;
                _Float16* const node_hi = static_cast<_Float16*>(pe->node_hi);
                _Float16* const node_lo = static_cast<_Float16*>(pe->node_lo);
                const int dim = pe->dim;
                if (e == 0 && node_hi && !m_i && (dim & 3) == 0 && m_dim == 16) {
                    // the common case (the layer's own launch sequence, m_dim = 16, dim % 4 == 0): the lane's four channels are four
                    // consecutive halves of one 16-byte chunk of the packed images -- one 8-byte store per image instead of four
                    // per-channel rounds of offset arithmetic
                    const int gch = pw_opaque(4) * g;
                    const int pool_mean = pe->pool_mean, nkt = pe->node_kp / 16;
                    f32x4 val;
                    bool bad = false;
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        float v = 0.f + nms[u];
                        if (pool_mean) {
                            if (has_mask) {                                      // safe_div, egnn_pytorch.py:13-16
                                const float cnt = 0.f + ncs[3];
                                v = (cnt == 0.f) ? 0.f : v / fmaxf(cnt, 1e-8f);
                            } else {
                                v = v / (float)K;                                // :330
                            }
                        }
                        bad = bad || egnn_beyond_f16(v);
                        val[u] = v;
                    }
                    egnn_flag_range(status, bad, EGNN_RANGE_MESSAGE);
                    f16x4 h4, l4;
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        h4[u] = (_Float16)val[u];
                        l4[u] = (_Float16)(val[u] - (float)h4[u]);
                    }
                    const size_t off = egnn_pk_off((int64_t)row, dim + gch, nkt);
                    *reinterpret_cast<f16x4*>(node_hi + off) = h4;
                    *reinterpret_cast<f16x4*>(node_lo + off) = l4;
                } else if (e == 0 && (m_i || node_hi)) {
                    const int gch = pw_opaque(4) * g;                            // (opaque: the packed offsets below are not loop invariants)
                    const int pool_mean = pe->pool_mean, nkt = pe->node_kp / 16;
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int ch = gch + u;
                        float val = 0.f + nms[u];
                        if (ch < m_dim) {
                            if (pool_mean) {
                                if (has_mask) {                                  // safe_div, egnn_pytorch.py:13-16
                                    const float cnt = 0.f + ncs[3];
                                    val = (cnt == 0.f) ? 0.f : val / fmaxf(cnt, 1e-8f);
                                } else {
                                    val = val / (float)K;                        // :330
                                }
                            }
                            if (m_i) m_i[row * m_dim + ch] = val;
                            if (node_hi) {                                       // straight into the node_mlp input, as a (hi, lo) pair
                                egnn_flag_range(status, egnn_beyond_f16(val), EGNN_RANGE_MESSAGE);
                                const _Float16 h = (_Float16)val;
                                const size_t off = egnn_pk_off((int64_t)row, dim + ch, nkt);
                                node_hi[off] = h;
                                node_lo[off] = (_Float16)(val - (float)h);
                            }
                        }
                    }
                }
                float* const coors_out = pe->coors_out;
                if (lane < 3 && coors_out) {
                    const float val = 0.f + (lane == 0 ? ncs[0] : (lane == 1 ? ncs[1] : ncs[2]));
                    coors_out[row * 3 + lane] = pe->coors[row * 3 + lane] + val;
                }
            }
            if (MULTI) {
                nms = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int c = 0; c < 4; ++c) ncs[c] = 0.f;
            }
            r = 0;
            ++kg;
        } else {
            ++r;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                        // no DMA may land after the LDS is released
}

}  // namespace

// internal (called by egnn_edge_fused_f32's dispatcher, edge_fused.hip): EGNN_E_UNSUPPORTED = "not this kernel's shape, use the general one"
// 1 when egnn_edge_fused_f32 (algo = 0) runs THIS kernel for a layer of the given shape -- provided it is handed slot records
// (egnn_slot_prep_f32), a split P_i table (K >= 6) and no training-mode dropout; else the general kernel runs.  Exported: callers that
// skip the projection rows of padded nodes (egnn_linear_hl_lda_rows_f32) may do so only here -- the general kernel's tiles mix the P_i
// rows of several nodes in one MFMA operand, where an unwritten (NaN) row would reach the edges of its neighbours in the tile.
extern "C" int egnn_edge_pw_covers(int B, int N, int K, int S, int fourier, int edge_dim, int m_dim, int coor_dim, int64_t ldp)
{
    if (coor_dim != 3 || K < 32 || (K % 32) != 0 || K > 4096) return 0;
    if (S != 1 || fourier != 0 || edge_dim != 0 || m_dim > 16) return 0;
    if ((int64_t)B * N * K * 16 > 0xffffffffLL) return 0;                 // slot records behind one 32-bit buffer resource
    if ((int64_t)N * ldp * 4 > 0xffffffffLL) return 0;
    return 1;
}

int egnn_edge_pw_launch(const egnn_edge_args* args, void* stream)
{
    const egnn_edge_args& a = *args;
    if (!egnn_edge_pw_covers(a.B, a.N, a.K, a.S, a.fourier, a.edge_dim, a.m_dim, a.coor_dim, a.ldp) || a.wst_terms != 4) return EGNN_E_UNSUPPORTED;
    if (!a.slots || !a.pi_split || (!a.idx && a.K != a.N)) return EGNN_E_UNSUPPORTED;       // (idx NULL: the dense all-pairs layer, records with j = k)
    if (a.drop_thr) return EGNN_E_UNSUPPORTED;                            // training-mode dropout keeps the general kernel (its MODE 3)
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
        return (int)hipGetLastError();
    const int64_t T = (int64_t)a.B * a.N;
    const int64_t ngroups = (T + PW_WAVES - 1) / PW_WAVES;
    // One workgroup per node group by default.  The kernel can walk several groups per workgroup (a grid of EGNN_PW_GRID_MULT x 5
    // workgroups per CU; 1 = fully persistent) -- measured on the north-star shape / c3, same box: persistent 1.54 / 0.484 ms, two
    // workgroups per slot 1.45 / 0.463, four 1.40 / 0.446, one workgroup per group 1.39 / 0.443 (profiles/r04_experiments/): with equal
    // shares fixed at launch the five workgroups of a CU stay in step -- their latency-bound setups and epilogues coincide instead of
    // hiding under each other's hidden loops -- and dynamic dispatch is what de-phases them.
#if defined(EGNN_PW_GRID_MULT)
    int64_t grid = (int64_t)cus * EGNN_PW_WGS * EGNN_PW_GRID_MULT;
    if (grid > ngroups) grid = ngroups;
#else
    int64_t grid = ngroups;
#endif
    grid = (grid + 7) / 8 * 8;                                             // eight XCDs; surplus workgroups return at once
    if (a.K == 32) hipLaunchKernelGGL(edge_pw_kernel<false>, dim3((unsigned)grid), dim3(PW_THREADS), 0, static_cast<hipStream_t>(stream), a);
    else hipLaunchKernelGGL(edge_pw_kernel<true>, dim3((unsigned)grid), dim3(PW_THREADS), 0, static_cast<hipStream_t>(stream), a);
    return egnn_launch_status();
}
