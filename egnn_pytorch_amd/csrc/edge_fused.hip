// The fused edge pass of EGNN.forward on gfx950 (reference: egnn_pytorch/egnn_pytorch.py:262-333).
//
// Per edge (b, i, k) with neighbour j = idx[b,i,k] (or j = k on the dense all-pairs path):
//     x[h]   = Pi[i,h] + Pj[j,h] + sum_s scal[s] * Ws[s,h]        (first Linear of edge_mlp, factorised:
//                                                                    Pi/Pj are node-level projections)
//     m_ij   = SiLU(W2 * SiLU(x) + b2)                             (second Linear, H -> m_dim)
//     m_ij  *= sigmoid(gate_w . m_ij + gate_b)                     (soft_edges)
//     w_ij   = W4 . SiLU(W3 * m_ij + b3) + b4                      (coors_mlp)
//     mask, clamp, CoorsNorm;  x_i' = x_i + sum_k w_ij * rel_ij;  m_i = sum_k m_ij  (or mean)
//
// What bounds it on MI355X (tools/ubench/gen_overlap_asm.py, silu_seq.hip; measured): the E x H SiLU evaluations.  The
// f32-input MFMA (v_mfma_f32_16x16x4_f32) executes on the SAME datapath as the f32 VALU -- MFMA time and VALU time add --
// while f16/bf16 MFMAs run on the matrix cores and hide completely behind VALU work.  So BOTH Linears of edge_mlp run as
// split-f16 products on the f16 matrix cores (fp32 accumulation: per-product error ~2^-22, i.e. f32 class) and the VALU
// is left with exactly 4 instructions for SiLU (v_exp_f32, v_rcp_f32 cost 8.6 cycles, a plain op 2.8) and 2.5 for the
// hi/lo split of the hidden value (v_cvt_pkrtz_f16_f32 4.65, v_fma_mix_f32 4.6 cycles):
//   * first Linear: x = C + A B with C = the gathered P_j values (accumulator input), A = per hidden unit the (hi, lo)
//     pair of P_i and split weight pairs of W_s, B = per edge (1, 1) and the three-part fp16 split of each scalar
//     (v_mfma_f32_16x16x16_f16, NM chained MFMAs of 4 terms);
//   * second Linear: hid = hi + lo, W2 = hi + lo in f16; hi*hi + lo*hi + hi*lo on v_mfma_f32_16x16x32_f16.
//   * Pi/Pj/Ws arrive pre-scaled by -log2(e) (folded into the projection weights) so that SiLU(x) is
//     -ln2 * y * rcp(1 + exp2(y)) with y the MFMA result; the -ln2 and a power-of-two range scale are folded
//     into the f16 W2 fragments and undone once per tile (w2_inv_scale).
//
// Mapping to the machine
//   * one 256-thread workgroup owns G consecutive nodes (in Morton order) of one graph = up to 128 edge slots
//     per round; each wave owns 32 slots = 2 MFMA tiles of 16 edges; 128 VGPRs, 39 KB LDS -> 4 workgroups per CU.
//   * swapped orientation D[row][edge]: lane l (e = l & 15, g = l >> 4) owns edge e of its tile; after the first-layer
//     MFMA it holds hidden units 16 hb + 4 g .. + 3 (hb = 0, 1) of every 32-wide step, so
//        - the gather of Pj fetches WHOLE 128-byte lines (lane l: chunk l&7 of the row of slot 8q + (l>>3)); a
//          wave-level load is processed line by line (tools/ubench/gather.hip: 16 half-used lines per instruction
//          9.7 TB/s, 8 full lines 20.9 TB/s), the rows are then re-dealt through a wave-private, chunk-swizzled LDS buffer,
//        - W2 is read from LDS in pre-built fragment order (lane-linear, conflict free),
//        - the H -> 16 result lands as D[4g + r][e]: every lane keeps "its" edge for the whole epilogue and holds
//          exactly the B-operand fragments the coors_mlp MFMAs (16 -> 64) need -- no transposes, no LDS.
//   * hidden activations (E x H) never leave registers; per-node sums are a DPP butterfly + a fixed-order cross-wave
//     sum (K % 32 == 0) or go through LDS and are summed in k order -- deterministic, no float atomics.
//   * W2 fragments and the W_s table are staged by LDS-DMA in chunks of HC hidden columns, shared by the 4 waves; the
//     Pi/Pj rows of step s+1 are requested before step s is computed.
//   * blocks are remapped so that each XCD works on a contiguous range of graphs (Pj rows of a graph stay
//     in that XCD's L2; measured hit rate 85 %).
#include "egnn_common.h"
#include "egnn_lds_dma.h"

// (internal: the dropout translation units of this file, -DEGNN_EDGE_DROP_TU)
int egnn_edge_fused_c3_drop(const egnn_edge_args* args, void* stream);
int egnn_edge_fused_generic_c_drop(const egnn_edge_args* args, void* stream);

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2v __attribute__((ext_vector_type(2)));

#ifndef EGNN_EDGE_THREADS
#define EGNN_EDGE_THREADS 256
#endif
#ifndef EGNN_EDGE_HC
#define EGNN_EDGE_HC 256
#endif
// workgroups per CU of the per-lane-P_i variants (K < 6: adjacency neighbours) with 2 .. 5 per-edge scalars
#ifndef EGNN_EDGE_T0_BLOCKS
#define EGNN_EDGE_T0_BLOCKS 4
#endif
#ifndef EGNN_EDGE_MINW
#define EGNN_EDGE_MINW 5
#endif
// W2 / W_s staging as a two-slot ring of HCT/2 columns: the LDS-DMA of chunk c+1 is in flight while chunk c is computed
// (one barrier per chunk and no exposed DMA latency) instead of barrier, DMA, wait, barrier per HCT columns
#ifndef EGNN_EDGE_RING
#define EGNN_EDGE_RING 1
#endif
#ifndef EGNN_EDGE_PRIO
#define EGNN_EDGE_PRIO 0
#endif
// Gathered P_j lines go straight from L2 into the wave's exchange rows by LDS-DMA (`buffer_load_dwordx4 ... lds`): no VGPR
// round trip, no ds_write_b128 parking stores (4 KB of LDS-pipe traffic and 16 registers per wave and step)
#ifndef EGNN_EDGE_GDMA
#define EGNN_EDGE_GDMA 1
#endif
// The residual of the hi/lo split, lo = a - fp16(a), on the matrix cores: D = (-I) x hi + a is exact and the accumulator
// layout of `a` is the B-fragment layout of `hi` -- one v_mfma_f32_16x16x16_f16 per 4 values instead of a v_fma_mix_f32 per value
#ifndef EGNN_EDGE_LO_MFMA
#define EGNN_EDGE_LO_MFMA 1
#endif
constexpr int EDGE_THREADS = EGNN_EDGE_THREADS;
constexpr int EDGE_WAVES = EDGE_THREADS / 64;
constexpr int TILES = 2;                 // MFMA tiles (16 edges) per wave
#ifndef EGNN_EDGE_DENSE_TERMS
#define EGNN_EDGE_DENSE_TERMS 1
#endif
constexpr int SLOTS_PER_WAVE = TILES * 16;
constexpr int SLOTS_PER_ROUND = EDGE_WAVES * SLOTS_PER_WAVE;     // 256
constexpr int HC = EGNN_EDGE_HC;          // hidden columns per LDS chunk (steps of 32)
constexpr int KSTEP = 32;                // hidden units per v_mfma_f32_16x16x32_f16
// Coordinate dimension: this translation unit is compiled twice -- CDM = 3 (the fast path, every BASELINE config) and,
// with -DEGNN_EDGE_GENERIC_C, CDM = 8 for 1 <= C <= 8 at run time (egnn_pytorch.py works for any C; its tests use C = 5).
// -DEGNN_EDGE_DROP_TU: a translation unit of its own for the training-mode dropout instantiations (MODE 3) beyond the standard layer's --
// 17 .. 64 message channels with 3-D coordinates, every head width with the other coordinate dimensions -- so that the two main
// compilations keep their size (csrc/build.sh compiles all four in parallel).
#ifdef EGNN_EDGE_GENERIC_C
constexpr int CDM = 8;
#ifdef EGNN_EDGE_DROP_TU
#define EGNN_EDGE_ENTRY egnn_edge_fused_generic_c_drop
#else
#define EGNN_EDGE_ENTRY egnn_edge_fused_generic_c
#endif
#else
constexpr int CDM = 3;
#ifdef EGNN_EDGE_DROP_TU
#define EGNN_EDGE_ENTRY egnn_edge_fused_c3_drop
#else
#define EGNN_EDGE_ENTRY egnn_edge_fused_c3
#endif
#endif
// per-edge channels reduced per node: 16 NB message channels | CDM coords | 1 count (NB = 16-channel blocks of m_dim)
constexpr int nch_of(int nb) { return 16 * nb + CDM + 1; }
constexpr int GMAX = 64;                 // nodes per workgroup
constexpr int XLD = 32;                  // floats per row of the gather exchange buffer (one 128 B line, chunk-swizzled)

__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
    const int q = nblk >> 3, rr = nblk & 7, xcd = bid & 7;
    return (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (bid >> 3);
}

// Workgroups per CU the register allocator must allow.  Nothing may spill inside the hidden loop (csrc/build.sh checks the
// ISA: csrc/check_scratch.py).  The K % 32 == 0 kernel of the standard layer runs FIVE workgroups per CU (96 registers, 26 KB of
// LDS with 128-column staging chunks): a workgroup's setup and epilogue are latency chains (index -> neighbour list ->
// coordinates -> first gathered lines; constants, barriers, stores: ~0.2 ms of 1.47 at the north-star shape with the loop
// compiled out) that only the other workgroups of the CU can cover -- measured 1.470 -> 1.390 ms.  Its 13 spilled dwords are
// per-slot values of the setup that the epilogue reads back (one store, one load each, outside every loop).
constexpr int edge_min_blocks(int nm, int tpi, int nb)
{
    if (nb >= 4) return 1;                      // m_dim > 32: four accumulator tiles per edge tile
    if (nb == 2) return nm > 4 ? 1 : 2;
    if (nm >= 12) return 1;
    if (nm > 4) return 2;
    if (nm > 1) return (CDM == 3 && tpi == 2) ? 3 : ((CDM == 3 && tpi == 0) ? EGNN_EDGE_T0_BLOCKS : 2);
    if (CDM == 3 && tpi == 2) return EGNN_EDGE_MINW;
    return (CDM == 3 && tpi >= 1) ? 4 : 3;
}
// (the training forward also writes u and keeps the edge index: one workgroup fewer)
constexpr int edge_launch_blocks(int nm, int tpi, int nb, int mode)
{
    const int mb = edge_min_blocks(nm, tpi, nb);
    if (mode == 1 || mode == 3) return mb > 1 ? mb - 1 : mb;
    return mb;
}
// staging chunk of the five-workgroup kernel
#ifndef EGNN_EDGE_HC5
#define EGNN_EDGE_HC5 128
#endif

__device__ __forceinline__ float row16_sum(float v) { return egnn_row16_sum(v); }

typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t pack_h2(_Float16 a, _Float16 b)
{
    const f16x2 v = {a, b};
    return __builtin_bit_cast(uint32_t, v);
}

// NM: chained first-layer MFMAs (4 split terms each, 3 terms per per-edge scalar); HCT: hidden columns per LDS chunk;
// TPI: how P_i reaches x.  2 (K % 32 == 0): the 32 slots of a wave belong to one node -- its (hi, lo) row rides in the
// first-layer MFMA (K-slots 0, 1).  1 (K >= 6): the 16 slots of a tile touch at most 4 nodes -- their rows sit in the
// K-slots 4g, 4g+1 of lane group g and every edge selects its node with a (1, 1) there.  0 (K < 6): fp32 P_i, added per
// lane on the VALU.
// (The body is a device function of the block index so that a dispatcher kernel can give a workgroup slot either an edge
// group or a GEMM tile: tools/ubench/mix_probe.hip.)
template <int NM, int HCT, int TPI, int NB, int MODE = 0>
__device__ __forceinline__ void edge_body(const egnn_edge_args& p, const int G, const int gpg, char* smem, const int bid, const int nblk)
{
    // MODE 0: inference forward.  1: forward that also writes u (args.U_out) for the backward -- its own instantiation, the
    // inference kernels sit at the register limit.  3: training-mode dropout.
    constexpr bool DROP = MODE == 3;                         // training-mode dropout (args.drop_thr); writes u like MODE 1 if asked
    // (wide heads -- m_dim > 16 -- and the generic coordinate dimension have no instantiation of their own for it: a run-time branch)
    constexpr bool WRITE_U = MODE == 1 || MODE == 3 || NB > 1 || CDM != 3;
    constexpr bool GDMA = EGNN_EDGE_GDMA && EGNN_EDGE_RING;           // gathers by LDS-DMA
    // first-layer MFMAs: with P_i on the VALU (TPI == 0: K < 6) both words of a lane group's K-slots are free for scalar terms -- two
    // terms per lane group and MFMA, half as many MFMAs (same table, read two words at a time; round 5, late)
    constexpr bool DENSE_TERMS = EGNN_EDGE_DENSE_TERMS && TPI == 0 && NM > 1;
    constexpr int NMX = DENSE_TERMS ? (NM + 1) / 2 : NM;
    constexpr int HC = EGNN_EDGE_RING ? HCT / 2 : HCT;     // columns per staged chunk (ring: two slots of HCT / 2)
    constexpr int NCH = nch_of(NB);
    constexpr int W2B = 64 * NB;                           // bytes of W2 fragments per hidden column: NB blocks x (hi | lo) x 16 channels
    const int S = p.S;
    _Float16* w2s = reinterpret_cast<_Float16*>(smem);                  // [HCT/32][NB][hi|lo][64][8] halves = HCT * 64 NB bytes
    float* xchall = reinterpret_cast<float*>(smem + HCT * W2B);           // [EDGE_WAVES][32 slots][XLD]: per-wave gather exchange
    float* ebuf = xchall;                                                // [slots][NCH] aliases it (TPI != 2 epilogue only)
    constexpr int XCH_FLOATS = (TPI != 2 && SLOTS_PER_ROUND * NCH > SLOTS_PER_ROUND * XLD) ? SLOTS_PER_ROUND * NCH : SLOTS_PER_ROUND * XLD;
    float* nodeacc = xchall + XCH_FLOATS;                               // [G][NCH]
    char* wst = reinterpret_cast<char*>(nodeacc + G * NCH);             // [HCT][4 NM] dwords: first-layer A fragments

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int e = lane & 15;
    const int g = lane >> 4;

    const int v = xcd_remap(bid, nblk);
    const int b = v / gpg;
    const int node0 = (v % gpg) * G;
    const int N = p.N, K = p.K;
    const int slots_total = G * K;                           // per node group (one or several rounds of 128 slots)
    const int rounds = (slots_total + SLOTS_PER_ROUND - 1) / SLOTS_PER_ROUND;
    const bool has_mask = p.mask != nullptr;
    const bool has_rank = p.rank != nullptr && p.idx != nullptr;
    // per-slot records of egnn_slot_prep_f32 (neighbour path, C = 3): {j | pair_ok << 31, x_i - x_j} in consumption order -- the
    // setup then has no dependent loads (order -> idx -> coors -> mask / rank), just one coalesced 16-byte load per slot
    typedef uint32_t u32x4v __attribute__((ext_vector_type(4)));
    const u32x4v* slot_rec = (CDM == 3 && p.idx) ? static_cast<const u32x4v*>(p.slots) : nullptr;
    const size_t bN = (size_t)b * N;
    // Buffer resources over this graph's rows of P_j / P_i: the gathers are `buffer_load ... offen` with a 32-bit per-lane
    // byte offset (one address register per stream) and the hidden-unit offset of the step in the SCALAR offset operand --
    // no vector address arithmetic inside the loop.
    const uint32_t prow_bytes = (uint32_t)((size_t)N * p.ldp * 4);
    const __amdgpu_buffer_rsrc_t pj_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.Pj + bN * p.ldp), 0, prow_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t pi_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.Pi + bN * p.ldp), 0, prow_bytes, 0x00020000);
    const u32x4s pj_words = make_rsrc_words(p.Pj + bN * p.ldp, prow_bytes);      // the same descriptor as four scalars (GDMA)

    for (int o = tid; o < G * NCH; o += EDGE_THREADS) nodeacc[o] = 0.f;

    for (int round = 0; round < rounds; ++round) {
        // ------------------------------------------------------------------ per-slot setup
        uint32_t pip[TILES];                                 // TPI == 0: byte offset of this lane's fp32 P_i row (+ 4 g)
        uint32_t piw[TPI == 1 ? TILES : 1] = {};             // byte offset of: TPI == 2: the wave's P_i row as (hi, lo) words
                                                             // (+ lane & 15); TPI == 1: per tile, the row of the tile's g-th node
        u32x2 bq[TILES][NMX];                                // B fragments of the first-layer MFMAs (constant over the hidden loop)
        int ei[TILES], ej[TILES];                            // node / neighbour of this lane's edge: x_i - x_j is recomputed in the
                                                             // epilogue instead of living in 2 x CDM registers across the hidden loop
        bool fm[TILES];                                      // edge contributes (valid slot and unmasked)
        uint32_t ekey[DROP ? TILES : 1] = {};                // DROP: mask row key of this lane's edge (+ the lane's unit offset)

        // TPI == 2 (K % 32 == 0): the 32 slots of a wave belong to ONE node -> node index and first k are wave-uniform
        const int qwave = round * SLOTS_PER_ROUND + wave * SLOTS_PER_WAVE;
        const int nl_w = (TPI == 2) ? qwave / K : 0;
        const int k_w = qwave - nl_w * K;

#pragma unroll
        for (int t = 0; t < TILES; ++t) {
            const int q = qwave + t * 16 + e;
            int nl = (TPI == 2) ? nl_w : q / K;
            int k = (TPI == 2) ? k_w + t * 16 + e : q - nl * K;
            int pos = node0 + nl;                                    // position in the (optionally permuted) node order
            bool valid = (q < slots_total) && (pos < N);
            if (!valid) { pos = node0 < N ? node0 : 0; k = 0; }
#if defined(EGNN_EDGE_ABL) && (EGNN_EDGE_ABL & 1024)
            const int i = pos;                                    // ablation: no index chain (order -> idx -> coors)
            const int j = (pos + k) & (N - 1);
#else
            const int i = p.order ? p.order[bN + pos] : pos;
            u32x4v rec = u32x4v{0u, 0u, 0u, 0u};
            if (slot_rec) rec = slot_rec[(bN + pos) * (size_t)K + k];
            const int j = slot_rec ? (int)(rec[0] & 0x3fffffffu) : (p.idx ? p.idx[(bN + i) * K + k] : k);
#endif
            const int C = (CDM == 3) ? 3 : p.coor_dim;
            const float* ci = p.coors + (bN + i) * C;
            const float* cj = p.coors + (bN + j) * C;
            float d;
#if !(defined(EGNN_EDGE_ABL) && (EGNN_EDGE_ABL & 1024))
            if (slot_rec) {
                // (by value: __builtin_bit_cast of a vector-element lvalue reads the vector's first bytes -- hipcc 7.2 -- i.e. element 0)
                const uint32_t r1 = rec[1], r2 = rec[2], r3 = rec[3];
                d = egnn_sqdist_rel(__uint_as_float(r1), __uint_as_float(r2), __uint_as_float(r3));
            } else
#endif
            {
                float rel0[CDM];
                if (CDM == 3) {
                    d = egnn_sqdist(ci[0], ci[1], ci[2], cj[0], cj[1], cj[2], rel0[0], rel0[1], rel0[2]);
                } else {
                    float a[CDM], bb[CDM];
#pragma unroll
                    for (int c = 0; c < CDM; ++c) { a[c] = c < C ? ci[c] : 0.f; bb[c] = c < C ? cj[c] : 0.f; }
                    d = egnn_sqdist_n<CDM>(a, bb, C, rel0);
                }
            }
            ei[t] = i; ej[t] = j;
            if constexpr (DROP)                                   // the edge's mask row (csrc/egnn_common.h), + this lane's 4 g of the unit index
                ekey[t] = egnn_drop_base(p.drop_seed, EGNN_DROP_SITE_EDGE, (uint32_t)((bN + i) * (size_t)K + k)) + (uint32_t)(4 * g) * 0x85EBCA77u;

            // Per-edge scalars [sin(d/2^f)..., cos(d/2^f)..., d, edges...] (egnn_pytorch.py:34-41, 282-285) as B
            // fragments of v_mfma_f32_16x16x16_f16: lane group g of MFMA m carries split term tau = 4 m + g of scalar
            // tau / 3 in K-slots 4g+2, 4g+3 (K-slots 4g, 4g+1 belong to P_i):
            //     s' = s / ws_scale = 2^10 s1 + r_hi + r_lo      (s1 coarse; r the exact remainder, |s'| clamped to 6e7)
            // meeting the (hi, lo) weight pairs of egnn_pytorch_amd/_weights.py::scalar_table.
            const int F = p.fourier;
            // split term tau (scalar tau / 3, part tau % 3) of this lane's edge as the (fp16, fp16) word of a B fragment; 0 beyond the scalars
            auto term_word = [&](const int tau) -> uint32_t {
                const int sidx = tau / 3, kind = tau - 3 * sidx;
                if (sidx >= S) return 0u;
                float val;
                if (NM == 1 && TPI != 2) val = d;                   // NM == 1 <=> S == 1: d is the only scalar (TPI == 2 keeps the generic
                                                                    // form: the shortcut tips its register allocation into a spill)
                else if (sidx < F) val = sinf(d * exp2f(-(float)sidx));
                else if (sidx < 2 * F) val = cosf(d * exp2f(-(float)(sidx - F)));
                else if (sidx == 2 * F) val = d;
                else val = p.edges[(p.edges_by_k ? (bN + i) * (size_t)K + k : (bN + i) * (size_t)N + j) * p.edge_dim + (sidx - 2 * F - 1)];
                // |s'| beyond 2^10 * 65504 = 6.7e7 does not fit the three fp16 parts: the coarse part overflows to inf and
                // the edge's result is NaN (never a silently clamped number); the status word says why
                val = val * p.ws_inv_scale;
                egnn_flag_range(p.status, valid && fabsf(val) >= 6.0e7f && fabsf(val) < __builtin_inff(), EGNN_RANGE_SCALAR);
                if (fabsf(val) >= 6.0e7f) val = __builtin_nanf("");
                const _Float16 s1 = (_Float16)(val * (1.0f / 1024.0f));
                const float r = val - (float)s1 * 1024.0f;
                const _Float16 rh = (_Float16)r;
                const _Float16 rl = (_Float16)(r - (float)rh);
                return kind == 0 ? pack_h2(s1, s1) : (kind == 1 ? pack_h2(rh, rh) : pack_h2(rl, (_Float16)0.f));
            };
#pragma unroll
            for (int m = 0; m < NMX; ++m) {
                u32x2 bw = u32x2{0u, 0u};
                if constexpr (DENSE_TERMS) {
                    // P_i is added on the VALU here (TPI == 0), so BOTH words of a lane group's K-slots carry scalar terms: terms 8 m + 2 g and
                    // 8 m + 2 g + 1 -- half the first-layer MFMAs of the one-term-per-group form (c4: K = 3, five scalars: 4 -> 2)
                    bw[0] = term_word(8 * m + 2 * g);
                    bw[1] = term_word(8 * m + 2 * g + 1);
                } else {
                    bw[1] = term_word(4 * m + g);
                    if (TPI == 2 && m == 0 && g == 0) bw[0] = pack_h2((_Float16)1.f, (_Float16)1.f);   // x (P_i hi, P_i lo)
                    if (TPI == 1 && m == 0) {
                        // the tile's slots start in node nf and end at most 3 nodes later (K >= 6): lane group g carries node nf + g
                        const int nf = (qwave + t * 16) / K;
                        if (nl - nf == g) bw[0] = pack_h2((_Float16)1.f, (_Float16)1.f);
                    }
                }
                bq[t][m] = bw;
            }

            bool em = valid;
#if !(defined(EGNN_EDGE_ABL) && (EGNN_EDGE_ABL & 1024))
            if (slot_rec) {
                em = em && (rec[0] >> 31) != 0u;                     // mask_i & mask_j & (rank <= radius), or 1 without a mask
            } else
#endif
            if (has_mask) {
                em = em && p.mask[bN + i] && p.mask[bN + j];
                if (has_rank) em = em && (p.rank[(bN + i) * K + k] <= p.valid_radius);
            }
            fm[t] = em;
            pip[t] = (uint32_t)(((size_t)i * p.ldp + 4 * g) * 4);
            if (TPI == 2 && t == 0) piw[0] = (uint32_t)(((size_t)i * p.ldp + e) * 4);
            if (TPI == 1) {
                // the tile's g-th node.  Its row meets a zero B coefficient in every edge column that belongs to another node,
                // and 0 x NaN is NaN on the matrix core: a row that is not needed (no such node, or a masked node -- all its
                // edges are zeroed anyway, :322) must read as zeros, or non-finite padding behind the mask would leak into
                // the valid edges of the same tile.  An offset past the buffer resource's range makes the load return 0.
                const int posg = node0 + (qwave + t * 16) / K + g;
                const bool exists = posg < N && posg < node0 + G;
                const int ig = exists ? (p.order ? p.order[bN + posg] : posg) : 0;
                const bool needed = exists && (!has_mask || p.mask[bN + ig]);
                piw[t] = needed ? (uint32_t)(((size_t)ig * p.ldp + e) * 4) : 0x80000000u;
            }
        }

        // Gather addressing.  A wave-level load instruction is processed line by line (measured,
        // tools/ubench/gather.hip: 16 half-used 128-B lines per instruction run at 9.7 TB/s, 8 fully used lines at
        // 20.9 TB/s), so P_j is fetched as whole lines -- lane l reads 16-byte chunk (l & 7) of the row of slot
        // 8*q + (l >> 3) -- and redistributed to the MFMA accumulator layout through a wave-private LDS buffer.
        uint32_t goff[4];
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
            const int q = qwave + qq * 8 + (lane >> 3);
            int nl = (TPI == 2) ? nl_w : q / K;
            int k = (TPI == 2) ? k_w + qq * 8 + (lane >> 3) : q - nl * K;
            int pos = node0 + nl;
            if (!((q < slots_total) && (pos < N))) { pos = node0 < N ? node0 : 0; k = 0; }
#if defined(EGNN_EDGE_ABL) && (EGNN_EDGE_ABL & 1024)
            const int i2 = pos;
            const int j2 = (pos + k) & (N - 1);
#else
            int j2;
            if (slot_rec) {
                j2 = (int)(reinterpret_cast<const uint32_t*>(slot_rec + ((bN + pos) * (size_t)K + k))[0] & 0x3fffffffu);
            } else {
                const int i2 = p.order ? p.order[bN + pos] : pos;
                j2 = p.idx ? p.idx[(bN + i2) * K + k] : k;
            }
#endif
#if defined(EGNN_EDGE_ABL) && (EGNN_EDGE_ABL & 1)
            goff[qq] = (uint32_t)(((size_t)(i2 & ~7) * p.ldp + 4 * (lane & 7)) * 4);
#else
            // GDMA: the DMA drops lane l's chunk at byte 16 l of the 1 KB piece, i.e. at position l & 7 of row 8 qq + (l >> 3);
            // the exchange buffer's swizzle (chunk c at position c ^ ((row >> 1) & 7)) moves to the global side: fetch the
            // chunk that belongs at that position
            const int chunk = GDMA ? ((lane & 7) ^ ((4 * qq + (lane >> 4)) & 7)) : (lane & 7);
            goff[qq] = (uint32_t)(((size_t)j2 * p.ldp + 4 * chunk) * 4);
#endif
        }
        // Exchange buffer: row = slot (128 B), 16-byte chunk c stored at position c ^ ((row >> 1) & 7): the parking
        // stores (2 rows x 8 chunks per 16 lanes) and the pick-up loads (16 rows x 1 chunk per 16 lanes) are both
        // bank-conflict free.
        float* xch = xchall + wave * (SLOTS_PER_WAVE * XLD);
        // parking: row = 8 qq + (lane >> 3); the swizzle (row >> 1) & 7 = (4 qq + (lane >> 4)) & 7 repeats every 16 rows, so
        // qq and qq + 2 differ by a constant 16 rows (an immediate offset): two address registers instead of four
        float* xw[2];
#pragma unroll
        for (int qq = 0; qq < 2; ++qq) {
            const int row = 8 * qq + (lane >> 3);
            xw[qq] = xch + row * XLD + 4 * ((lane & 7) ^ ((row >> 1) & 7));
        }
        // pick-up: lane (e, g) reads hidden rows 16 hb + 4 g .. + 3 of slot 16 t + e; the swizzle depends on e only
        const float* xr[2];
#pragma unroll
        for (int hb = 0; hb < 2; ++hb) xr[hb] = xch + e * XLD + 4 * ((4 * hb + g) ^ ((e >> 1) & 7));

        // A operand of the residual MFMA: -I (row e, K-slots 4g .. 4g+3)
        f16x4 neg_identity;
#pragma unroll
        for (int u = 0; u < 4; ++u) neg_identity[u] = (e == 4 * g + u) ? (_Float16)-1.f : (_Float16)0.f;
        // first-layer A fragments: row (hidden unit) e of the 16-block, split term 4 m + g
        const char* tl = wst + (e * (4 * NM) + (g ^ ((e >> 2) & 2))) * 4;      // (units 8 .. 15 of a block: term pairs swapped in the table, no bank conflict)
        constexpr int tstep = 16 * 4 * NM * 4;                          // bytes per 16 hidden units
        // DENSE_TERMS: from this lane's one-word position to its two-word position (bytes; 8-byte aligned: even word index)
        const int dense_delta = ((4 * (g >> 1) + 2 * ((g & 1) ^ ((e >> 3) & 1))) - (g ^ ((e >> 2) & 2))) * 4;

        f32x4 acc[TILES][NB];
#pragma unroll
        for (int t = 0; t < TILES; ++t)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) acc[t][nb] = f32x4{0.f, 0.f, 0.f, 0.f};

        // ------------------------------------------------------------------ main loop over hidden units
        // Software pipeline: the Pi/Pj rows of step st+1 are requested before step st is computed, so the
        // gather latency (L2 / Infinity Cache) hides under the SiLU work of the current step.
        f32x4 gl[GDMA ? 1 : 4];
        f32x4 pin[TPI == 0 ? TILES : 1][2];
        uint32_t piv[TPI == 1 ? TILES : 1][2] = {};
        const uint32_t xch_lds = __builtin_amdgcn_readfirstlane((uint32_t)(size_t)(__attribute__((address_space(3))) char*)(char*)xch);
        if constexpr (GDMA) {
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) gather_dma16(pj_words, goff[qq], 0u, xch_lds + qq * 1024);
        } else {
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) gl[qq] = buf_load4(pj_rsrc, goff[qq], 0);
        }
        if (TPI == 0) {
#pragma unroll
            for (int t = 0; t < TILES; ++t) {
                pin[t][0] = buf_load4(pi_rsrc, pip[t], 0);
                pin[t][1] = buf_load4(pi_rsrc, pip[t], 64);
            }
        } else {
#pragma unroll
            for (int t = 0; t < (TPI == 1 ? TILES : 1); ++t) {
                piv[t][0] = buf_load1(pi_rsrc, piw[t], 0);
                piv[t][1] = buf_load1(pi_rsrc, piw[t], 64);
            }
        }
#if defined(EGNN_EDGE_ABL) && (EGNN_EDGE_ABL & 4)
        const int HpLoop = 0;                                            // ablation: setup + epilogue only
#else
        const int HpLoop = p.Hp;
#endif
#if EGNN_EDGE_RING
        // Two-slot ring of HC = HCT / 2 columns: the LDS-DMA of chunk c+1 flies while chunk c is computed out of slot c & 1
        // -- one barrier per chunk and no exposed load latency.  The DMA is issued from inline asm (lds_dma16): with the
        // builtin the compiler, which cannot tell LDS-DMA writes from the other LDS traffic, waits vmcnt(0) before every
        // LDS access of the step that follows.  Hidden from its counters the extra loads can only make its own vmcnt
        // waits stricter (the counter retires in order), never weaker; their completion is waited for explicitly below.
        // W2 fragments: (Hp/32, 2, 64, 8) halves = 2048 bytes per step; scalar table: NM * 16 bytes per hidden unit.
        auto stage = [&](int c0s, int slot) {
            const int hcs = (p.Hp - c0s) < HC ? (p.Hp - c0s) : HC;
            const char* src = reinterpret_cast<const char*>(p.W2h) + (size_t)c0s * W2B + lane * 16;
            char* dst = reinterpret_cast<char*>(w2s) + slot * (HC * W2B);
            for (int pc = wave; pc < hcs * NB / 16; pc += EDGE_WAVES) lds_dma16(src + pc * 1024, dst + pc * 1024);
            const int tbytes = hcs * NM * 16;
            const char* tsrc = reinterpret_cast<const char*>(p.Wst) + (size_t)c0s * NM * 16 + lane * 16;
            char* tdst = wst + slot * (HC * NM * 16);
            for (int pc = wave; pc * 1024 < tbytes; pc += EDGE_WAVES)
                if (pc * 1024 + lane * 16 < tbytes) lds_dma16(tsrc + pc * 1024, tdst + pc * 1024);
        };
        stage(0, 0);               // (the barrier that ended the previous round freed both slots)
        int slot = 0;
        for (int c0 = 0; c0 < HpLoop; c0 += HC, slot ^= 1) {
            const int hc = (p.Hp - c0) < HC ? (p.Hp - c0) : HC;
            // chunk c0 was requested one chunk ago; the only other loads in flight are the gathers of the coming step, which
            // the step consumes first thing anyway -> waiting for everything costs nothing extra
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#if !(defined(EGNN_EDGE_ABL) && (EGNN_EDGE_ABL & 64))                         // (ablation 64: no chunk barrier -- timing only)
            __syncthreads();       // every wave's pieces have landed, and every wave has left the other slot
#endif
#if defined(EGNN_EDGE_RING_DBG) && (EGNN_EDGE_RING_DBG & 1)
            if (c0 + HC < p.Hp) { stage(c0 + HC, slot ^ 1); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); }
#else
            // (GDMA: issued inside the chunk's first step, behind that step's wait for its gathered lines -- every wait of the
            // loop is then a plain vmcnt(0) and none of them sits out a DMA that has only just been issued)
            if (!GDMA && c0 + HC < p.Hp) stage(c0 + HC, slot ^ 1);
#endif
            const _Float16* w2c = w2s + slot * (HC * (W2B / 2));
            const char* tlc = tl + slot * (HC * NM * 16);
#else
        for (int c0 = 0; c0 < HpLoop; c0 += HC) {
            const int hc = (p.Hp - c0) < HC ? (p.Hp - c0) : HC;
            __syncthreads();
            {
                // LDS-DMA, 1 KB (64 lanes x 16 B) per instruction: no VGPR round trip, no VALU address loop.
                // W2 fragments: (Hp/32, 2, 64, 8) halves = 2048 bytes per step; scalar table: NM * 16 bytes per hidden unit.
                const char* src = reinterpret_cast<const char*>(p.W2h) + (size_t)c0 * W2B + lane * 16;
                for (int pc = wave; pc < hc * NB / 16; pc += EDGE_WAVES)
                    __builtin_amdgcn_global_load_lds((glb_void*)(src + pc * 1024),
                                                     (lds_void*)(reinterpret_cast<char*>(w2s) + pc * 1024), 16, 0, 0);
                const int tbytes = hc * NM * 16;
                const char* tsrc = reinterpret_cast<const char*>(p.Wst) + (size_t)c0 * NM * 16 + lane * 16;
                for (int pc = wave; pc * 1024 < tbytes; pc += EDGE_WAVES)
                    if (pc * 1024 + lane * 16 < tbytes)
                        __builtin_amdgcn_global_load_lds((glb_void*)(tsrc + pc * 1024), (lds_void*)(wst + pc * 1024), 16, 0, 0);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __syncthreads();
            const _Float16* w2c = w2s;
            const char* tlc = tl;
#endif

            const int nst = hc / KSTEP;
            for (int st = 0; st < nst; ++st) {
                const int hoff = c0 + st * KSTEP;
#if defined(EGNN_EDGE_STEPSYNC) && EGNN_EDGE_STEPSYNC
                __builtin_amdgcn_s_barrier();      // keep the workgroup's waves on the same step: gathered rows shared via L1
#endif
#if EGNN_EDGE_PRIO
                __builtin_amdgcn_s_setprio(EGNN_EDGE_PRIO);          // hurry through the LDS / MFMA head of the step
#endif
                int hnext = hoff + KSTEP;
                if (hnext >= p.Hp) hnext = hoff;                 // last step: harmless re-read (GDMA: nothing is issued)
                // x starts as the gathered P_j values, in the MFMA accumulator layout: lane (e, g) = edge e, hidden rows
                // 16 hb + 4 g .. + 3 of this step
                f32x4 x[TILES][2];
                uint32_t pivn[TPI == 1 ? TILES : 1][2] = {};
                if constexpr (GDMA) {
                    // the lines of this step were requested a step ago and land in the wave's exchange rows by themselves.
                    // (The waits are the builtin, not inline asm: the compiler then KNOWS that no load of its own -- the P_i
                    // words -- is outstanding; guessing, it waits vmcnt(3) for them, which with four untracked DMA
                    // instructions behind them means sitting out the latency of the lines just requested.)
                    asm volatile("" ::: "memory");
                    __builtin_amdgcn_s_waitcnt(0x0F70);             // vmcnt(0)
                    asm volatile("" ::: "memory");
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int t = 0; t < TILES; ++t) {
#if defined(EGNN_EDGE_ABL) && (EGNN_EDGE_ABL & 32)
                        x[t][0] = f32x4{0.f, 0.f, 0.f, 0.f};                // ablation: no pick-up reads
                        x[t][1] = f32x4{0.f, 0.f, 0.f, 0.f};
#else
                        x[t][0] = *reinterpret_cast<const f32x4*>(xr[0] + t * 16 * XLD);
                        x[t][1] = *reinterpret_cast<const f32x4*>(xr[1] + t * 16 * XLD);
#endif
                    }
                    // the rows are in registers before the next step's lines may overwrite them
                    asm volatile("" ::: "memory");
                    __builtin_amdgcn_s_waitcnt(0xC07F);             // lgkmcnt(0)
                    asm volatile("" ::: "memory");
                    __builtin_amdgcn_wave_barrier();
                    // the step's P_i words are requested here, between two asm statements that order memory operations: left
                    // to itself the scheduler sinks them to the end of the step, right in front of the wait that needs them
                    if constexpr (TPI != 0) {
#pragma unroll
                        for (int t = 0; t < (TPI == 1 ? TILES : 1); ++t) {
                            pivn[t][0] = buf_load1(pi_rsrc, piw[t], hnext * 4);
                            pivn[t][1] = buf_load1(pi_rsrc, piw[t], hnext * 4 + 64);
                        }
                    }
                    if (st == 0 && c0 + HC < p.Hp) stage(c0 + HC, slot ^ 1);
#if !(defined(EGNN_EDGE_ABL) && (EGNN_EDGE_ABL & 128))                     // (ablation 128: no gathers)
                    if (hoff + KSTEP < p.Hp) {
#pragma unroll
                        for (int qq = 0; qq < 4; ++qq) gather_dma16(pj_words, goff[qq], (uint32_t)(hnext * 4), xch_lds + qq * 1024);
                    }
#endif
                } else {
                // park the lines fetched for this step, then (same wave, DS ops execute in order) pick the rows up
#if defined(EGNN_EDGE_ABL) && (EGNN_EDGE_ABL & 32)
                const f32x4 glx[4] = {gl[0], gl[1], gl[2], gl[3]};      // ablation: no park / pick-up through LDS
#else
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) *reinterpret_cast<f32x4*>(xw[qq & 1] + (qq >> 1) * 16 * XLD) = gl[qq];
#endif
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) gl[qq] = buf_load4(pj_rsrc, goff[qq], hnext * 4);
                // Same-wave hand-off through LDS: DS operations of one wave execute in issue order; the explicit
                // lgkmcnt(0) makes the store -> other-lane load dependency independent of that (4 stores, negligible).
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int t = 0; t < TILES; ++t) {
#if defined(EGNN_EDGE_ABL) && (EGNN_EDGE_ABL & 32)
                    x[t][0] = glx[2 * t];
                    x[t][1] = glx[2 * t + 1];
#else
                    x[t][0] = *reinterpret_cast<const f32x4*>(xr[0] + t * 16 * XLD);
                    x[t][1] = *reinterpret_cast<const f32x4*>(xr[1] + t * 16 * XLD);
#endif
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                }

                u32x2 av[NMX][2];
                if constexpr (DENSE_TERMS) {
                    // two consecutive table words per lane group: logical 4-word group 2 m + (g >> 1), its words 2 (g & 1), 2 (g & 1) + 1
                    // (units 8 .. 15 of a block store their word pairs swapped: _weights.py::swizzle_terms); a group beyond NM (odd NM): zero
#pragma unroll
                    for (int m = 0; m < NMX; ++m) {
                        const bool in = 2 * m + (g >> 1) < NM;
                        const char* tp = tlc + dense_delta + m * 32 + st * 2 * tstep;
                        av[m][0] = in ? *reinterpret_cast<const u32x2*>(tp) : u32x2{0u, 0u};
                        av[m][1] = in ? *reinterpret_cast<const u32x2*>(tp + tstep) : u32x2{0u, 0u};
                    }
                } else {
#pragma unroll
                    for (int m = 0; m < NM; ++m) {
                        av[m][0] = u32x2{0u, *reinterpret_cast<const uint32_t*>(tlc + m * 16 + st * 2 * tstep)};
                        av[m][1] = u32x2{0u, *reinterpret_cast<const uint32_t*>(tlc + m * 16 + st * 2 * tstep + tstep)};
                    }
                }
                f16x8 whi[NB], wlo[NB];
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    whi[nb] = *reinterpret_cast<const f16x8*>(w2c + (((st * NB + nb) * 2 + 0) * 64 + lane) * 8);
                    wlo[nb] = *reinterpret_cast<const f16x8*>(w2c + (((st * NB + nb) * 2 + 1) * 64 + lane) * 8);
                }
                u32x2 a0[TILES][2];                                // first MFMA's A operand: K-slots 4g, 4g+1 = P_i (hi, lo)
                if (TPI == 0) {
#pragma unroll
                    for (int t = 0; t < TILES; ++t) {
                        x[t][0] += pin[t][0];
                        x[t][1] += pin[t][1];
                        pin[t][0] = buf_load4(pi_rsrc, pip[t], hnext * 4);
                        pin[t][1] = buf_load4(pi_rsrc, pip[t], hnext * 4 + 64);
                        a0[t][0] = av[0][0];
                        a0[t][1] = av[0][1];
                    }
                } else {
#pragma unroll
                    for (int t = 0; t < TILES; ++t) {
                        const int pt = TPI == 1 ? t : 0;           // (TPI == 2: one row for both tiles; B is zero there for g > 0)
                        a0[t][0] = u32x2{piv[pt][0], av[0][0][1]};
                        a0[t][1] = u32x2{piv[pt][1], av[0][1][1]};
                    }
#pragma unroll
                    for (int t = 0; t < (TPI == 1 ? TILES : 1); ++t) {
                        if constexpr (GDMA) {
                            piv[t][0] = pivn[t][0];
                            piv[t][1] = pivn[t][1];
                        } else {
                            piv[t][0] = buf_load1(pi_rsrc, piw[t], hnext * 4);
                            piv[t][1] = buf_load1(pi_rsrc, piw[t], hnext * 4 + 64);
                        }
                    }
                }
                // First Linear of edge_mlp on the matrix cores: x += [P_i | W_s] x [1 | scalars]  (split-f16 products)
#pragma unroll
                for (int t = 0; t < TILES; ++t)
#pragma unroll
                    for (int hb = 0; hb < 2; ++hb)
#pragma unroll
                        for (int m = 0; m < NMX; ++m)
#if defined(EGNN_EDGE_ABL) && (EGNN_EDGE_ABL & 16)
                            x[t][hb][0] += __builtin_bit_cast(float, (m == 0 ? a0[t][hb] : av[m][hb])[0] ^ bq[t][m][1]);   // ablation: no first-layer MFMAs
#else
                            x[t][hb] = __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(f16x4, m == 0 ? a0[t][hb] : av[m][hb]),
                                                                            __builtin_bit_cast(f16x4, bq[t][m]), x[t][hb], 0, 0, 0);
#endif
#if EGNN_EDGE_PRIO
                __builtin_amdgcn_s_setprio(0);
#endif
                {
#pragma unroll
                for (int t = 0; t < TILES; ++t) {
                    // x holds y = -log2(e) * (pre-activation); hv = y / (1 + 2^y) = SiLU(pre) / (-ln 2)
                    f16x8 bhi, blo;
#if defined(EGNN_EDGE_STAGEWISE) && EGNN_EDGE_STAGEWISE
                    // stage-wise over SW of the tile's 8 values: SW independent chains between dependent instructions
                    constexpr int SW = EGNN_EDGE_STAGEWISE;
#pragma unroll
                    for (int u0 = 0; u0 < 8; u0 += SW) {
                        float yv[SW], rv[SW], hv[SW];
#pragma unroll
                        for (int u = 0; u < SW; ++u) yv[u] = x[t][(u0 + u) >> 2][(u0 + u) & 3];
#pragma unroll
                        for (int u = 0; u < SW; ++u) rv[u] = __builtin_amdgcn_exp2f(yv[u]);
#pragma unroll
                        for (int u = 0; u < SW; ++u) rv[u] = 1.0f + rv[u];
#pragma unroll
                        for (int u = 0; u < SW; ++u) rv[u] = __builtin_amdgcn_rcpf(rv[u]);
#pragma unroll
                        for (int u = 0; u < SW; ++u) { hv[u] = yv[u] * rv[u]; asm("" : "+v"(hv[u])); }
#pragma unroll
                        for (int u = 0; u < SW; u += 2) {
                            const f16x2 hi = __builtin_convertvector((f32x2v){hv[u], hv[u + 1]}, f16x2);
                            const uint32_t hw = __builtin_bit_cast(uint32_t, hi);
                            float l0, l1;
                            asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(l0) : "v"(yv[u]), "v"(rv[u]), "v"(hw));
                            asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(l1) : "v"(yv[u + 1]), "v"(rv[u + 1]), "v"(hw));
                            const f16x2 lo = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(l0, l1));
                            bhi[u0 + u] = hi[0]; bhi[u0 + u + 1] = hi[1];
                            blo[u0 + u] = lo[0]; blo[u0 + u + 1] = lo[1];
                        }
                    }
#elif EGNN_EDGE_LO_MFMA
                    // a = y / (1 + 2^y) per value (4 VALU instructions), hi = fp16(a) (v_cvt_pk_f16_f32, IEEE: beyond 65504 -> inf,
                    // so an overflowing hidden value poisons the edge's message instead of saturating silently), and the
                    // residual on the matrix cores: lo32 = (-I) x hi + a, exact (a - hi has at most 13 significant bits)
#pragma unroll
                    for (int hb = 0; hb < 2; ++hb) {
                        f32x4 a4;
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            float y = x[t][hb][u];
                            if constexpr (DROP) {
                                // nn.Dropout behind edge_mlp's first Linear (:180): unit hoff + 16 hb + 4 g + u of this lane's edge.
                                // (y = -log2(e) z: scaling commutes; a dropped unit gives SiLU(0) = 0 like the reference's)
                                const uint32_t hsh = egnn_drop_hash(ekey[t], (uint32_t)(hoff + 16 * hb + u));
                                y = hsh >= p.drop_thr ? y * p.drop_inv_keep : 0.f;
                            }
#if defined(EGNN_EDGE_ABL) && (EGNN_EDGE_ABL & 2)
                            float h = y * (1.0f + y);                       // ablation: no transcendentals
#else
                            float h = y * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(y));
#endif
                            asm("" : "+v"(h));           // keeps the products scalar (v_pk_mul_f32 costs 9.3 cycles per pair against 2 x 2.8)
                            a4[u] = h;
                        }
                        const f16x2 h01 = __builtin_convertvector((f32x2v){a4[0], a4[1]}, f16x2);
                        const f16x2 h23 = __builtin_convertvector((f32x2v){a4[2], a4[3]}, f16x2);
                        const f16x4 hi4 = {h01[0], h01[1], h23[0], h23[1]};
                        const f32x4 l4 = __builtin_amdgcn_mfma_f32_16x16x16f16(neg_identity, hi4, a4, 0, 0, 0);
                        const f16x2 l01 = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(l4[0], l4[1]));
                        const f16x2 l23 = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(l4[2], l4[3]));
                        bhi[4 * hb + 0] = h01[0]; bhi[4 * hb + 1] = h01[1]; bhi[4 * hb + 2] = h23[0]; bhi[4 * hb + 3] = h23[1];
                        blo[4 * hb + 0] = l01[0]; blo[4 * hb + 1] = l01[1]; blo[4 * hb + 2] = l23[0]; blo[4 * hb + 3] = l23[1];
                    }
#else
#pragma unroll
                    for (int u = 0; u < 8; u += 2) {
                        const float y0 = x[t][u >> 2][u & 3], y1 = x[t][u >> 2][(u & 3) + 1];
#if defined(EGNN_EDGE_ABL) && (EGNN_EDGE_ABL & 2)
                        const float h0 = y0 * (1.0f + y0);                  // ablation: no transcendentals
                        const float h1 = y1 * (1.0f + y1);
#else
                        const float h0 = y0 * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(y0));
                        const float h1 = y1 * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(y1));
#endif
                        // v_cvt_pk_f16_f32 (IEEE: beyond 65504 -> inf, so an overflowing hidden value poisons the edge's
                        // message instead of saturating silently as v_cvt_pkrtz would; same issue cost)
                        // (the empty asm keeps the two products scalar: SLP-packed into v_pk_mul_f32 they cost 9.3 cycles per
                        // pair against 2 x 2.8, tools/ubench/mix_rates.hip)
                        float h0s = h0, h1s = h1;
                        asm("" : "+v"(h0s));
                        asm("" : "+v"(h1s));
                        const f16x2 hi = __builtin_convertvector((f32x2v){h0s, h1s}, f16x2);
#if (defined(EGNN_EDGE_ABL) && (EGNN_EDGE_ABL & 2)) || (defined(EGNN_EDGE_LO_PLAIN) && EGNN_EDGE_LO_PLAIN)
                        const float l0 = h0 - (float)hi[0];
                        const float l1 = h1 - (float)hi[1];
#else
                        // lo = y * sigma - hi as ONE v_fma_mix_f32 per value (the f16 operand is read in place).  Written out:
                        // behind the round-to-nearest conversion above hipcc no longer forms it by itself and converts hi
                        // back with two extra v_cvt_f32_f16 per pair (+8 % VALU instructions on this VALU-bound loop).
                        const float r0 = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(y0));
                        const float r1 = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(y1));
                        const uint32_t hw = __builtin_bit_cast(uint32_t, hi);
                        float l0, l1;
                        asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(l0) : "v"(y0), "v"(r0), "v"(hw));
                        asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(l1) : "v"(y1), "v"(r1), "v"(hw));
#endif
                        const f16x2 lo = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(l0, l1));
                        bhi[u] = hi[0]; bhi[u + 1] = hi[1];
                        blo[u] = lo[0]; blo[u + 1] = lo[1];
                    }
#endif
#if defined(EGNN_EDGE_ABL) && (EGNN_EDGE_ABL & 8)
                    acc[t][0][0] += (float)bhi[0] + (float)blo[1] + (float)bhi[2] + (float)blo[3] + (float)bhi[4] + (float)blo[5] + (float)bhi[6] + (float)blo[7] + (float)whi[0][0] + (float)wlo[0][0];   // ablation: no second-layer MFMAs
#else
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) {
#if defined(EGNN_EDGE_L2_LEGACY) && EGNN_EDGE_L2_LEGACY
                        // experiment: each K = 32 product as two legacy K = 16 MFMAs (the fragment halves are the two 16-row blocks)
                        const f16x4 wh0 = {whi[nb][0], whi[nb][1], whi[nb][2], whi[nb][3]}, wh1 = {whi[nb][4], whi[nb][5], whi[nb][6], whi[nb][7]};
                        const f16x4 wl0 = {wlo[nb][0], wlo[nb][1], wlo[nb][2], wlo[nb][3]}, wl1 = {wlo[nb][4], wlo[nb][5], wlo[nb][6], wlo[nb][7]};
                        const f16x4 bh0 = {bhi[0], bhi[1], bhi[2], bhi[3]}, bh1 = {bhi[4], bhi[5], bhi[6], bhi[7]};
                        const f16x4 bl0 = {blo[0], blo[1], blo[2], blo[3]}, bl1 = {blo[4], blo[5], blo[6], blo[7]};
                        acc[t][nb] = __builtin_amdgcn_mfma_f32_16x16x16f16(wh0, bh0, acc[t][nb], 0, 0, 0);
                        acc[t][nb] = __builtin_amdgcn_mfma_f32_16x16x16f16(wh1, bh1, acc[t][nb], 0, 0, 0);
                        acc[t][nb] = __builtin_amdgcn_mfma_f32_16x16x16f16(wl0, bh0, acc[t][nb], 0, 0, 0);
                        acc[t][nb] = __builtin_amdgcn_mfma_f32_16x16x16f16(wl1, bh1, acc[t][nb], 0, 0, 0);
                        acc[t][nb] = __builtin_amdgcn_mfma_f32_16x16x16f16(wh0, bl0, acc[t][nb], 0, 0, 0);
                        acc[t][nb] = __builtin_amdgcn_mfma_f32_16x16x16f16(wh1, bl1, acc[t][nb], 0, 0, 0);
#else
                        acc[t][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(whi[nb], bhi, acc[t][nb], 0, 0, 0);
                        acc[t][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wlo[nb], bhi, acc[t][nb], 0, 0, 0);
                        acc[t][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(whi[nb], blo, acc[t][nb], 0, 0, 0);
#endif
                    }
#endif
                }
                }
            }
        }
        if constexpr (GDMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no DMA may land in rows the epilogue reuses
#if defined(EGNN_EDGE_ABL) && (EGNN_EDGE_ABL & 512)
        {                                                     // ablation: no epilogue (keeps the accumulators alive)
            float sacc = 0.f;
#pragma unroll
            for (int t = 0; t < TILES; ++t) sacc += acc[t][0][0] + acc[t][0][1] + acc[t][0][2] + acc[t][0][3] + (fm[t] ? 1.f : 0.f) + (float)(ei[t] + ej[t]);
            if (sacc == 12345.678f) nodeacc[tid] = sacc;
            __syncthreads();
            continue;
        }
#endif

        // ------------------------------------------------------------------ per-edge epilogue (registers)
        // channel of (block nb, lane group g, register u) = 16 nb + 4 g + u
        f32x4 b2r[NB], gwr[NB];
        float gb = 0.f;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            gwr[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int u = 0; u < 4; ++u) b2r[nb][u] = p.b2[16 * nb + 4 * g + u];
            if (p.gate_w) {
#pragma unroll
                for (int u = 0; u < 4; ++u) gwr[nb][u] = p.gate_w[16 * nb + 4 * g + u];
            }
        }
        if (p.gate_w) gb = p.gate_b[0];
        float cscale = 0.f;
        if (p.coors_scale) cscale = p.coors_scale[0];

        float cw[TILES];
#pragma unroll
        for (int t = 0; t < TILES; ++t) {
            f32x4 m[NB];
            bool bad = false;
            float part = 0.f;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    bad = bad || !(fabsf(acc[t][nb][u]) < __builtin_inff());
                    m[nb][u] = egnn_silu(acc[t][nb][u] * p.w2_inv_scale + b2r[nb][u]);
                }
                part += gwr[nb][0] * m[nb][0] + gwr[nb][1] * m[nb][1] + gwr[nb][2] * m[nb][2] + gwr[nb][3] * m[nb][3];
            }
            egnn_flag_range(p.status, bad && fm[t], EGNN_RANGE_HIDDEN);
            if (WRITE_U && (MODE == 1 || p.U_out)) {         // u = W2 SiLU(x) + b2: what the backward differentiates from
                // (the slot -> (node, k) decode of the setup again: keeping the edge index live through the hidden loop costs
                // four registers the TPI = 1 variant does not have)
                const int q = qwave + t * 16 + e;
                const int nl = (TPI == 2) ? nl_w : q / K;
                const int kk = (TPI == 2) ? k_w + t * 16 + e : q - nl * K;
                if (q < slots_total && node0 + nl < N) {
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) {          // rows of 16 NB channels (m_dim rounded up to whole accumulator tiles)
                        f32x4 uu;
#pragma unroll
                        for (int u = 0; u < 4; ++u) uu[u] = acc[t][nb][u] * p.w2_inv_scale + b2r[nb][u];
                        *reinterpret_cast<f32x4*>(p.U_out + ((bN + ei[t]) * (size_t)K + kk) * (16 * NB) + 16 * nb + 4 * g) = uu;
                    }
                }
            }
            if (p.gate_w) {
                part = egnn_column_sum4(part, xch + 64 * t, lane);          // this wave's exchange rows are free now
                const float gt = egnn_sigmoid(part + gb);
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) m[nb] *= gt;
            }
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) acc[t][nb] = m[nb];
            cw[t] = 0.f;
        }

#if defined(EGNN_EDGE_ABL) && (EGNN_EDGE_ABL & 256)
        if (false) {                                          // ablation: no coors_mlp
#else
        if (p.W3h) {
#endif
            float part[TILES];
#pragma unroll
            for (int t = 0; t < TILES; ++t) part[t] = 0.f;
            // coors_mlp first Linear (16 -> 64) on the matrix cores, same split-f16 scheme: lane (e, g) already holds
            // channels 4g..4g+3 of its edge = the B fragment of v_mfma_f32_16x16x16_f16; A = rows 16 blk + e of W3.
            // (Twice during development 1-2 % of the coordinate weights of dense multi-round launches came out wrong,
            // differently on every run, with the node features intact; it also showed when two launches ran concurrently
            // on different streams.  Bisected (tools/concurrency_check.py) to the two ds_bpermute_b32 = `__shfl_xor`
            // that summed the weight over the four lane groups below: with LDS-DMA traffic of co-resident workgroups in
            // flight they occasionally returned another value.  No kernel uses the LDS-pipe shuffles any more
            // (egnn_common.h poisons them); regressions: tests/test_gpu_parity.py::test_multi_round_stress... and
            // ::test_concurrent_launches_do_not_change_results.)
            f16x4 mhi[TILES][NB], mlo[TILES][NB];
#pragma unroll
            for (int t = 0; t < TILES; ++t) {
                bool bad = false;
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                    for (int u = 0; u < 4; ++u) bad = bad || egnn_beyond_f16(acc[t][nb][u]);
                egnn_flag_range(p.status, bad && fm[t], EGNN_RANGE_MESSAGE);
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const _Float16 h = (_Float16)acc[t][nb][u];
                        mhi[t][nb][u] = h;
                        mlo[t][nb][u] = (_Float16)(acc[t][nb][u] - (float)h);
                    }
            }
            // W3h: (2, 64 NB, 16 NB) fp16, hi image then lo image of coors_mlp.0.weight zero padded; the (16 NB -> 64 NB)
            // product runs as 4 NB row blocks x NB K-blocks of v_mfma_f32_16x16x16_f16
            const _Float16* w3h = static_cast<const _Float16*>(p.W3h);
            constexpr int W3LD = 16 * NB, W3IMG = 64 * NB * 16 * NB;
#pragma unroll
            for (int blk = 0; blk < 4 * NB; ++blk) {
                const f32x4 b3 = *reinterpret_cast<const f32x4*>(p.b3 + 16 * blk + 4 * g);
                const f32x4 w4 = *reinterpret_cast<const f32x4*>(p.W4 + 16 * blk + 4 * g);
                f32x4 a2t[TILES];
#pragma unroll
                for (int t = 0; t < TILES; ++t) a2t[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kb = 0; kb < NB; ++kb) {
                    const f16x4 w3hi = *reinterpret_cast<const f16x4*>(w3h + (16 * blk + e) * W3LD + 16 * kb + 4 * g);
                    const f16x4 w3lo = *reinterpret_cast<const f16x4*>(w3h + W3IMG + (16 * blk + e) * W3LD + 16 * kb + 4 * g);
#pragma unroll
                    for (int t = 0; t < TILES; ++t) {
                        a2t[t] = __builtin_amdgcn_mfma_f32_16x16x16f16(w3hi, mhi[t][kb], a2t[t], 0, 0, 0);
                        a2t[t] = __builtin_amdgcn_mfma_f32_16x16x16f16(w3lo, mhi[t][kb], a2t[t], 0, 0, 0);
                        a2t[t] = __builtin_amdgcn_mfma_f32_16x16x16f16(w3hi, mlo[t][kb], a2t[t], 0, 0, 0);
                    }
                }
#pragma unroll
                for (int t = 0; t < TILES; ++t) {
                    const f32x4 a2 = a2t[t];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        float hpre = a2[u] * p.w3_inv_scale + b3[u];
                        if constexpr (DROP) {                            // nn.Dropout behind coors_mlp's first Linear (:205), unit 16 blk + 4 g + u
                            const uint32_t base = ekey[t] + (EGNN_DROP_SITE_COORS - EGNN_DROP_SITE_EDGE) * 0x27D4EB2Fu;
                            hpre = egnn_drop_hash(base, (uint32_t)(16 * blk + u)) >= p.drop_thr ? hpre * p.drop_inv_keep : 0.f;
                        }
                        part[t] += w4[u] * egnn_silu(hpre);
                    }
                }
            }
            const float b4 = p.b4[0];
#pragma unroll
            for (int t = 0; t < TILES; ++t) {
                float s = part[t];
                s = egnn_column_sum4(s, xch + 64 * (TILES + t), lane);
                s += b4;
                if (has_mask && !fm[t]) s = 0.f;                         // :308-309
                if (p.clamp >= 0.f) s = fminf(fmaxf(s, -p.clamp), p.clamp);   // :311-313
                if (!fm[t] && !has_mask) s = 0.f;                        // padding slot (not a real edge)
                cw[t] = s;
            }
        }

        float rel[TILES][CDM];                               // x_i - x_j again (components >= C are 0)
        {
            const int C = (CDM == 3) ? 3 : p.coor_dim;
#pragma unroll
            for (int t = 0; t < TILES; ++t) {
                if (CDM == 3 && slot_rec) {                              // (the record again: coalesced, L2)
                    const int q = qwave + t * 16 + e;
                    int nl = (TPI == 2) ? nl_w : q / K;
                    int kk = (TPI == 2) ? k_w + t * 16 + e : q - nl * K;
                    int pos = node0 + nl;
                    if (!((q < slots_total) && (pos < N))) { pos = node0 < N ? node0 : 0; kk = 0; }
                    const u32x4v rec = slot_rec[(bN + pos) * (size_t)K + kk];
                    const uint32_t rw[3] = {rec[1], rec[2], rec[3]};     // (by value, see the setup)
#pragma unroll
                    for (int c = 0; c < CDM; ++c) rel[t][c] = c < 3 ? __uint_as_float(rw[c < 3 ? c : 0]) : 0.f;
                } else {
                    const float* ci = p.coors + (bN + ei[t]) * C;
                    const float* cj = p.coors + (bN + ej[t]) * C;
#pragma unroll
                    for (int c = 0; c < CDM; ++c) rel[t][c] = c < C ? ci[c] - cj[c] : 0.f;
                }
            }
        }
        if (TPI == 2) {
            // The wave's 32 edges belong to one node: sum them in registers (DPP butterfly over the 16 edges of a
            // tile, fixed order -> deterministic) and hand 20 partials per wave to the cross-wave reduction.
            float rn[TILES][CDM], keep[TILES];
#pragma unroll
            for (int t = 0; t < TILES; ++t) {
                keep[t] = fm[t] ? 1.f : 0.f;
                float inv = 1.f;
                if (p.coors_scale) {                                    // CoorsNorm, egnn_pytorch.py:67-77
                    float n2 = 0.f;                                     // (explicit fma chain: the same bits in every kernel variant)
#pragma unroll
                    for (int c = 0; c < CDM; ++c) n2 = __builtin_fmaf(rel[t][c], rel[t][c], n2);
                    inv = cscale / fmaxf(sqrtf(n2), 1e-8f);
                }
#pragma unroll
                for (int c = 0; c < CDM; ++c) rn[t][c] = rel[t][c] * inv;
            }
            const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f};
            f32x4 ms[NB];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
                ms[nb] = (fm[0] ? acc[0][nb] : zero4) + (fm[1] ? acc[1][nb] : zero4);     // select: masked_fill semantics (:322)
            float cs[CDM + 1];
#pragma unroll
            for (int c = 0; c < CDM; ++c) cs[c] = __builtin_fmaf(cw[1], rn[1][c], cw[0] * rn[0][c]);
            cs[CDM] = keep[0] + keep[1];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int u = 0; u < 4; ++u) ms[nb][u] = row16_sum(ms[nb][u]);
#pragma unroll
            for (int c = 0; c <= CDM; ++c) cs[c] = row16_sum(cs[c]);
            // the wave's own exchange rows double as its partial-sum row (no other wave touches them)
            if (e == 0) {
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) *reinterpret_cast<f32x4*>(xch + 16 * nb + 4 * g) = ms[nb];
            }
            if (lane == 0) {
#pragma unroll
                for (int c = 0; c <= CDM; ++c) xch[16 * NB + c] = cs[c];
            }
            __syncthreads();
            const int kw = K / SLOTS_PER_WAVE;                           // waves per node
            const int wbase = round * EDGE_WAVES;
            for (int o = tid; o < G * NCH; o += EDGE_THREADS) {
                const int nl = o / NCH, ch = o - nl * NCH;
                int w0 = nl * kw, w1 = w0 + kw;
                if (w0 < wbase) w0 = wbase;
                if (w1 > wbase + EDGE_WAVES) w1 = wbase + EDGE_WAVES;
                float s = 0.f;
                for (int w = w0; w < w1; ++w) s += xchall[(w - wbase) * (SLOTS_PER_WAVE * XLD) + ch];
                if (w1 > w0) nodeacc[o] += s;
            }
        } else {
            __syncthreads();                                     // previous round's reduction has read ebuf
#pragma unroll
            for (int t = 0; t < TILES; ++t) {
                const int slot = wave * SLOTS_PER_WAVE + t * 16 + e;
                const float keep = fm[t] ? 1.f : 0.f;
                float* row = ebuf + slot * NCH;
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)                              // masked_fill semantics (:322)
                    *reinterpret_cast<f32x4*>(row + 16 * nb + 4 * g) = fm[t] ? acc[t][nb] : f32x4{0.f, 0.f, 0.f, 0.f};
                if (g == 0) {
                    float inv = 1.f;
                    if (p.coors_scale) {                                    // CoorsNorm, egnn_pytorch.py:67-77
                        float n2 = 0.f;
#pragma unroll
                        for (int c = 0; c < CDM; ++c) n2 = __builtin_fmaf(rel[t][c], rel[t][c], n2);
                        inv = cscale / fmaxf(sqrtf(n2), 1e-8f);
                    }
#pragma unroll
                    for (int c = 0; c < CDM; ++c) row[16 * NB + c] = cw[t] * (rel[t][c] * inv);
                    row[16 * NB + CDM] = keep;
                }
            }
            __syncthreads();

            // ------------------------------------------------------------------ per-node reduction, k order
            const int qbase = round * SLOTS_PER_ROUND;
            for (int o = tid; o < G * NCH; o += EDGE_THREADS) {
                const int nl = o / NCH, ch = o - nl * NCH;
                int q0 = nl * K, q1 = q0 + K;
                if (q0 < qbase) q0 = qbase;
                if (q1 > qbase + SLOTS_PER_ROUND) q1 = qbase + SLOTS_PER_ROUND;
                float s = 0.f;
                for (int q = q0; q < q1; ++q) s += ebuf[(q - qbase) * NCH + ch];
                if (q1 > q0) nodeacc[o] += s;
            }
        }
        __syncthreads();          // multi-round groups: ebuf (aliasing the exchange buffers) is free again
    }
    __syncthreads();

    // ---------------------------------------------------------------------- node outputs
    for (int o = tid; o < G * NCH; o += EDGE_THREADS) {
        const int nl = o / NCH, ch = o - nl * NCH;
        if (node0 + nl >= N) continue;
        const int i = p.order ? p.order[bN + node0 + nl] : node0 + nl;
        float val = nodeacc[o];
        if (ch < 16 * NB) {
            if (ch < p.m_dim && (p.m_i || p.node_hi)) {
                if (p.pool_mean) {
                    if (has_mask) {                                     // safe_div, egnn_pytorch.py:13-16
                        const float cnt = nodeacc[nl * NCH + (NCH - 1)];
                        val = (cnt == 0.f) ? 0.f : val / fmaxf(cnt, 1e-8f);
                    } else {
                        val = val / (float)K;                           // :330
                    }
                }
                if (p.m_i) p.m_i[(bN + i) * p.m_dim + ch] = val;
                if (p.node_hi) {                                        // straight into the node_mlp input, as a (hi, lo) pair
                    egnn_flag_range(p.status, egnn_beyond_f16(val), EGNN_RANGE_MESSAGE);
                    const _Float16 h = (_Float16)val;
                    const size_t off = egnn_pk_off((int64_t)(bN + i), p.dim + ch, p.node_kp / 16);
                    static_cast<_Float16*>(p.node_hi)[off] = h;
                    static_cast<_Float16*>(p.node_lo)[off] = (_Float16)(val - (float)h);
                }
            }
        } else if (ch < 16 * NB + ((CDM == 3) ? 3 : p.coor_dim)) {
            const int C = (CDM == 3) ? 3 : p.coor_dim;
            if (p.coors_out) p.coors_out[(bN + i) * C + (ch - 16 * NB)] = p.coors[(bN + i) * C + (ch - 16 * NB)] + val;
        }
    }
}

template <int NM, int HCT, int TPI, int NB, int MODE = 0>
__global__ __launch_bounds__(EDGE_THREADS, edge_launch_blocks(NM, TPI, NB, MODE)) void edge_kernel(const egnn_edge_args p, const int G, const int gpg)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    edge_body<NM, HCT, TPI, NB, MODE>(p, G, gpg, smem, blockIdx.x, gridDim.x);
}

template <int NM, int HCT, int TPI, int NB, int MODE = 0>
int launch_edge(const egnn_edge_args& a, hipStream_t s)
{
    constexpr int NCH = nch_of(NB);
    // nodes per workgroup: as many as fit one round of 128 slots -- or, when that would leave slots idle (K = 24: 120 of
    // 128, K = 48: 96), the smallest group whose slots fill whole rounds (K = 48: 8 nodes = 3 rounds), up to 16 nodes
    int G = SLOTS_PER_ROUND / a.K;
#if defined(EGNN_EDGE_GMULT)
    if (a.K % 32 == 0 && a.K <= SLOTS_PER_ROUND) G *= EGNN_EDGE_GMULT;          // experiment: several rounds per workgroup
#else
    // narrow layers (dim <= 256: at most 40 steps of the hidden loop): two rounds per workgroup -- the per-workgroup part of the
    // fixed cost (launch, arguments, node sums and outputs) is a third of the pass there (c3: 0.533 -> 0.510 ms; north star +1 %)
    if (a.K % 32 == 0 && a.K <= SLOTS_PER_ROUND && a.Hp <= 1280 && (int64_t)a.B * a.N / (2 * G) >= 4096) G *= 2;
#endif
    if (G < 1) G = 1;
    if (G > GMAX) G = GMAX;
    if (a.K < SLOTS_PER_ROUND && (G * a.K) % SLOTS_PER_ROUND != 0)
        for (int g2 = G + 1; g2 <= 16; ++g2)
            if ((g2 * a.K) % SLOTS_PER_ROUND == 0) { G = g2; break; }
    if (G > a.N) G = a.N;
    const int gpg = (a.N + G - 1) / G;
    const int64_t nblk = (int64_t)a.B * gpg;
    if (nblk > 0x7fffffffLL) return EGNN_E_UNSUPPORTED;
    // W2 fragments | gather exchange | node accumulators | first-layer A fragments (40 KB = 4 workgroups per CU at S = 1)
    const size_t xch_floats = (TPI != 2 && NCH > XLD) ? (size_t)SLOTS_PER_ROUND * NCH : (size_t)SLOTS_PER_ROUND * XLD;
    const size_t lds = (size_t)HCT * 64 * NB + sizeof(float) * (xch_floats + (size_t)G * NCH) + (size_t)HCT * NM * 16;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(edge_kernel<NM, HCT, TPI, NB, MODE>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL((edge_kernel<NM, HCT, TPI, NB, MODE>), dim3((unsigned)nblk), dim3(EDGE_THREADS), lds, s, a, G, gpg);
    return egnn_launch_status();
}

template <int NM, int HCT, int NB, int MODE = 0>
int dispatch_tpi_nb(const egnn_edge_args& a, hipStream_t s)
{
    // K % 32 == 0: both tiles of a wave share node i; K >= 6: a tile touches <= 4 nodes -- either way P_i rides in the
    // first-layer MFMA as (hi, lo) words (pi_split); K < 6: it is added per lane from the fp32 projection
    if (a.K % 32 == 0) return launch_edge<NM, (NM == 1 && NB == 1 && edge_min_blocks(NM, 2, NB) >= 5) ? EGNN_EDGE_HC5 : HCT, 2, NB, MODE>(a, s);
    if (a.K >= 6) return launch_edge<NM, HCT, 1, NB, MODE>(a, s);
    return launch_edge<NM, HCT, 0, NB, MODE>(a, s);
}

// m_dim <= 16: one accumulator tile per edge tile (every BASELINE config); 17..32: two; 33..64: four, with the staged
// chunk shrunk so that the W2 fragments keep their LDS footprint (HCT * 64 NB bytes)
template <int NM, int HCT>
int dispatch_tpi(const egnn_edge_args& a, hipStream_t s)
{
#ifdef EGNN_EDGE_DROP_TU
    // training-mode dropout (egnn_pytorch.py:176-184, 203-208) for the shapes the main translation units do not instantiate it for
    if (!a.drop_thr) return EGNN_E_UNSUPPORTED;
#ifdef EGNN_EDGE_GENERIC_C
    if (a.m_dim <= 16) return dispatch_tpi_nb<NM, HCT, 1, 3>(a, s);
#else
    if (a.m_dim <= 16) return EGNN_E_UNSUPPORTED;                                   // (the main translation unit's)
#endif
    if (a.m_dim <= 32) return dispatch_tpi_nb<NM, (HCT / 2 >= 64 ? HCT / 2 : 64), 2, 3>(a, s);
    return dispatch_tpi_nb<NM, 64, 4, 3>(a, s);
#else
#ifndef EGNN_EDGE_GENERIC_C
    if (a.drop_thr) return a.m_dim <= 16 ? dispatch_tpi_nb<NM, HCT, 1, 3>(a, s) : egnn_edge_fused_c3_drop(&a, s);   // training-mode dropout
    if (a.m_dim <= 16 && a.U_out) return dispatch_tpi_nb<NM, HCT, 1, 1>(a, s);      // forward under autograd: also writes u
#else
    if (a.drop_thr) return egnn_edge_fused_generic_c_drop(&a, s);
#endif
    if (a.m_dim <= 16) return dispatch_tpi_nb<NM, HCT, 1>(a, s);
#ifndef EGNN_EDGE_TUNING_BUILD
    if (a.m_dim <= 32) return dispatch_tpi_nb<NM, (HCT / 2 >= 64 ? HCT / 2 : 64), 2>(a, s);
    return dispatch_tpi_nb<NM, 64, 4>(a, s);
#else
    return EGNN_E_UNSUPPORTED;
#endif
#endif  // EGNN_EDGE_DROP_TU
}

}  // namespace

#if !defined(EGNN_EDGE_GENERIC_C) && !defined(EGNN_EDGE_DROP_TU)
extern "C" int egnn_padded_hidden(int H) { return (H + 31) / 32 * 32; }

extern "C" int egnn_edge_mfmas(int S) { return S <= 1 ? 1 : (S <= 4 ? 3 : (S <= 5 ? 4 : (S <= 8 ? 6 : 12))); }

#endif

// internal: the two compilations of this file (coordinate dimension 3 / generic), and the persistent wave-per-node kernel of the
// K % 32 == 0 inference layers (edge_pw.hip; EGNN_E_UNSUPPORTED = not its shape)
int egnn_edge_fused_c3(const egnn_edge_args* args, void* stream);
int egnn_edge_fused_generic_c(const egnn_edge_args* args, void* stream);
int egnn_edge_pw_launch(const egnn_edge_args* args, void* stream);
#ifndef EGNN_EDGE_PW
#define EGNN_EDGE_PW 1
#endif

#if !defined(EGNN_EDGE_GENERIC_C) && !defined(EGNN_EDGE_DROP_TU)
extern "C" int egnn_edge_fused_f32(const egnn_edge_args* args, void* stream)
{
    if (!args) return EGNN_E_NULLPTR;
    if (args->coor_dim < 1 || args->coor_dim > 8) return EGNN_E_UNSUPPORTED;
    return args->coor_dim == 3 ? egnn_edge_fused_c3(args, stream) : egnn_edge_fused_generic_c(args, stream);
}

#endif

int EGNN_EDGE_ENTRY(const egnn_edge_args* args, void* stream)
{
    const egnn_edge_args& a = *args;
    if (!a.Pi || !a.Pj || !a.Wst || !a.W2h || !a.b2 || !a.coors) return EGNN_E_NULLPTR;
    if (!a.m_i && !a.coors_out && !a.node_hi) return EGNN_E_NULLPTR;
    if ((a.node_hi == nullptr) != (a.node_lo == nullptr)) return EGNN_E_NULLPTR;
    if (a.node_hi && (a.node_kp < a.dim + a.m_dim || (a.node_kp % 32) != 0 || a.dim <= 0)) return EGNN_E_SHAPE;
    if (a.coors_out && (!a.W3h || !a.b3 || !a.W4 || !a.b4 || !(a.w3_inv_scale > 0.f))) return EGNN_E_NULLPTR;
    if (a.gate_w && !a.gate_b) return EGNN_E_NULLPTR;
    if (a.B <= 0 || a.N <= 0 || a.K <= 0 || a.H <= 0) return EGNN_E_SHAPE;
    if (a.Hp != egnn_padded_hidden(a.H) || a.ldp < a.Hp || (a.ldp % 4) != 0 || !(a.w2_inv_scale > 0.f)) return EGNN_E_SHAPE;
    if (a.m_dim < 1 || a.m_dim > 64) return EGNN_E_UNSUPPORTED;
    if (a.S != 2 * a.fourier + 1 + a.edge_dim) return EGNN_E_SHAPE;
    if (a.S > 16) return EGNN_E_UNSUPPORTED;
    if (!(a.ws_inv_scale > 0.f)) return EGNN_E_SHAPE;
    if ((a.pi_split != 0) != (a.K >= 6)) return EGNN_E_SHAPE;          // P_i format must match the kernel variant
    if (a.edge_dim > 0 && !a.edges) return EGNN_E_NULLPTR;
    if (a.idx == nullptr && a.K != a.N) return EGNN_E_SHAPE;          // dense path: K == N
    if (a.slots && ((!a.idx && a.K != a.N) || a.coor_dim != 3)) return EGNN_E_SHAPE;  // records: 3-D coordinates; dense (idx NULL) means K == N
    if (a.drop_thr && !(a.drop_inv_keep >= 1.f)) return EGNN_E_SHAPE;
    if (a.slots && (reinterpret_cast<uintptr_t>(a.slots) & 15)) return EGNN_E_ALIGN;
    if ((reinterpret_cast<uintptr_t>(a.Pi) & 15) || (reinterpret_cast<uintptr_t>(a.Pj) & 15) ||
        (reinterpret_cast<uintptr_t>(a.Wst) & 15) || (reinterpret_cast<uintptr_t>(a.W2h) & 15))
        return EGNN_E_ALIGN;
    if (a.W3h && ((reinterpret_cast<uintptr_t>(a.W3h) & 15) || (reinterpret_cast<uintptr_t>(a.b3) & 15) ||
                  (reinterpret_cast<uintptr_t>(a.W4) & 15)))
        return EGNN_E_ALIGN;
    hipStream_t s = static_cast<hipStream_t>(stream);
    // NM = ceil(3 S / 4) first-layer MFMAs, rounded up to an instantiated value; Wst must be laid out for that NM
    if (a.wst_terms != 4 * egnn_edge_mfmas(a.S)) return EGNN_E_SHAPE;
#if !defined(EGNN_EDGE_GENERIC_C) && !defined(EGNN_EDGE_DROP_TU) && EGNN_EDGE_PW
    if (a.algo == 0) {
        const int rc = egnn_edge_pw_launch(args, stream);
        if (rc != EGNN_E_UNSUPPORTED) return rc;
    } else if (a.algo != 1) {
        return EGNN_E_SHAPE;
    }
#endif
    if (a.S == 1) return dispatch_tpi<1, HC>(a, s);
#ifndef EGNN_EDGE_TUNING_BUILD
    if (a.S <= 4) return dispatch_tpi<3, 128>(a, s);
    if (a.S <= 5) return dispatch_tpi<4, 128>(a, s);          // edge_dim = 4 (the README's configuration): 15 terms
    if (a.S <= 8) return dispatch_tpi<6, 64>(a, s);
    return dispatch_tpi<12, 64>(a, s);
#else
    return EGNN_E_UNSUPPORTED;
#endif
}
