// egnn_edge_bwd_pass_f32 -- the backward of the edge pass WITHOUT anything of size E x H in memory (gfx950).
//
// Reference semantics: autograd through egnn_pytorch.py:279-287 (edge_mlp's first Linear and SiLU and the second Linear's
// weight); include/egnn_hip.h documents the entry.  With z = P_i[i] + P_j[j] + W_s s the pre-activation of the first SiLU,
// a = SiLU(z), gU = d loss / d (second Linear's output) and dz = (W2^T gU) * SiLU'(z), the gradients that need all E x H
// values are contractions -- over the hidden index (d/d s = dz W_s) or over edges:
//     d/d P_i[n] = sum of dz over the edges that leave n,      d/d P_j[n] = sum of dz over the edges that arrive at n,
//     d/d W_2    = gU^T a,                                      d/d W_s    = dz^T s.
// The first backward kernel (round 2, since removed) wrote dz and a to HBM (35 GB at the north-star shape) for library
// reductions to read back five times: traffic-bound at 45 + 87 GB.  This one recomputes z and contracts in registers.
//
// Layout.  The forward kernel holds z as (hidden rows) x (edge columns) in the MFMA accumulator layout; contracting over
// edges on the matrix cores needs the transpose -- edges along the K dimension of the B operand.  Swapping the operands of
// every MFMA of the forward's first layer gives exactly that: with A = the per-edge fragment and B = the per-hidden-unit
// fragment, D[m = edge][n = hidden unit] lands with lane (g, h) holding edges 4g .. 4g+3 of the tile for hidden unit h, and
// those four registers ARE the B fragment [k = edge][n = hidden] of v_mfma_f32_16x16x16_f16.  Then
//     d/d P partial = Ind x dz      (A = the tile's valid-entry flags: the sum over the 16 entries of a tile, which the host's
//                                    list layout makes the entries of ONE node; a node's tiles are consecutive partial rows that
//                                    egnn_rows_gather_sum_f32 adds up in fixed order -- no float atomics, any K, ragged in-degrees),
//     d/d W_2 +=     gU^T x a       (A = this tile's gU transposed; accumulated in registers),
// dz and a as split-f16 pairs (the flags are exact, gU is split as well -> fp32-class sums; all three pre-scaled by powers of
// two: the lo half of a value below ~0.25 is an fp16 subnormal, resolved to 2^-24 absolute only -- d/d W_2 1e-6 -> 3e-7 against
// float64 in the kernel's test, tools/r02_exp28.sh; conversions and MFMA inputs do keep subnormals, tools/ubench/f16_subnormal.hip),
// while d/d W_s and d/d s are plain FMAs on the values (S of them each).
//
// Persistence.  d/d W_2 and d/d W_s are sums over ALL edges per hidden column, so a workgroup owns CH x 32 hidden columns
// (its W2^T / W_s fragments staged in LDS once) and streams a slab of the entry list through them: grid = slabs x column
// chunks, the per-entry setup is redone per chunk (cheap next to 4 steps of SiLU work), and every (slab, wave) ends with one
// small partial of d/d W_2 / d/d W_s that the host sums in fixed order.  Slabs are short (8 rounds of 128 entries) and the
// block order keeps the workgroups of one graph and one column chunk side by side on an XCD: what they gather is the same
// 512-byte pieces of that graph's rows (L2 hit rate 38 % -> forward-like with the order, 5.3 -> 4.4 ms per pass).
//
// Two passes over the edges: grouped by source node (d/d P_i; carries d/d W_s and d/d s) and grouped by neighbour, i.e. over
// the edge list sorted stably by destination (d/d P_j; carries d/d W_2) -- each all-edge contraction keeps its accumulators
// in registers, and one pass carrying all of them drops from 3-4 to 2 workgroups per CU (measured: slower).  z is rebuilt
// from fp32 P_i and P_j rows: the other endpoint's rows as whole 128-byte lines parked in wave-private LDS and picked up
// transposed, the tile's own row (wave-uniform offset in the scalar operand) by one load per tile and half step -- both one
// step ahead.
#include "egnn_common.h"

namespace {

typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2v __attribute__((ext_vector_type(2)));

#ifndef EGNN_BWD_DEST_BLOCKS
#define EGNN_BWD_DEST_BLOCKS 4
#endif
#ifndef EGNN_BWD_W2_BLOCKS
#define EGNN_BWD_W2_BLOCKS 3
#endif
#ifndef EGNN_BWD_S_BLOCKS
#define EGNN_BWD_S_BLOCKS 4
#endif
#ifndef EGNN_BWD_CHUNK_STEPS
#define EGNN_BWD_CHUNK_STEPS 8
#endif
#ifndef EGNN_BWD_GROUP_SLABS
#define EGNN_BWD_GROUP_SLABS 32
#endif
constexpr int GS = EGNN_BWD_GROUP_SLABS;     // slabs whose workgroups run next to each other on one XCD, chunk by chunk
constexpr int BW_THREADS = 256;
constexpr int BW_WAVES = 4;
constexpr int CH_S = EGNN_BWD_CHUNK_STEPS;   // steps of 32 hidden columns a workgroup owns (the variant that writes ds_part: sizes it)
#ifndef EGNN_BWD_CH_W2
#define EGNN_BWD_CH_W2 6
#endif
constexpr int CH_W2 = EGNN_BWD_CH_W2;                     // ... of the variant with the d/d W_2 tiles only: one step fewer keeps it at 3 workgroups per CU
constexpr int XLD = 36;                      // floats per exchange row: 144 B -> rows 4 apart sit 16 banks apart (transposed pick-up)
constexpr float DZ_UP = 256.f;               // dz (scaled units, < 2^7) x 2^8 before the f16 split: keeps small values off the subnormals
#ifndef EGNN_BWD_LO_MFMA
#define EGNN_BWD_LO_MFMA 1
#endif
#ifndef EGNN_BWD_SKIP_DEAD
#define EGNN_BWD_SKIP_DEAD 0                 // 1: a wave's round whose 32 entries all carry gU = 0 (padded nodes' edges, list padding) writes zeros and
                                             // moves on.  Measured (profiles/r06_experiments/padded_nodes.txt): ragged masks -0.25 ms of 13.4, all-true
                                             // masks +0.3 ms (the branch changes the loop's code) -- off
#endif
#ifndef EGNN_BWD_VALU_TILE_SUM
#define EGNN_BWD_VALU_TILE_SUM 1
#endif
#ifndef EGNN_BWD_A_UP
#define EGNN_BWD_A_UP 64.f
#endif
#ifndef EGNN_BWD_GT_UP
#define EGNN_BWD_GT_UP 64.f
#endif
constexpr float A_UP = EGNN_BWD_A_UP;        // SiLU(z) and the transposed gU likewise (x 2^6 each): lo halves out of the subnormal range
constexpr float GT_UP = EGNN_BWD_GT_UP;      //   (d/d W_2: 1e-6 -> 3e-7 of its scale; tools/r02_exp28.sh)

__device__ __forceinline__ uint32_t pack_h2(_Float16 a, _Float16 b)
{
    const f16x2 v = {a, b};
    return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ f32x4 buf_load4(__amdgpu_buffer_rsrc_t r, uint32_t voff, int soff)
{
    typedef unsigned int u32x4v __attribute__((ext_vector_type(4)));
    return __builtin_bit_cast(f32x4, (u32x4v)__builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, soff, 0));
}
__device__ __forceinline__ float buf_load1f(__amdgpu_buffer_rsrc_t r, uint32_t voff, int soff)
{
    return __builtin_bit_cast(float, (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, soff, 0));
}
__device__ __forceinline__ void buf_store1f(__amdgpu_buffer_rsrc_t r, uint32_t voff, int soff, float v)
{
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v), r, (int)voff, soff, 0);
}
__device__ __forceinline__ int xcd_remap(int bid, int nblk)
{
    const int q = nblk >> 3, rr = nblk & 7, xcd = bid & 7;
    return (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (bid >> 3);
}
// (hi, lo) split of four fp32 values into two f16x4 fragments
__device__ __forceinline__ void split4(const f32x4 v, f16x4& hi, f16x4& lo)
{
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const _Float16 h = (_Float16)v[u];
        hi[u] = h;
        lo[u] = (_Float16)(v[u] - (float)h);
    }
}

// Per-entry records of one pass, in list order (edge_bwd_prep_kernel).  A workgroup owns a chunk of hidden columns, so every
// 128-entry round is set up once per column chunk (4 - 8 times per pass at the north-star width): index arithmetic (two integer
// divisions per entry), the fp16 splits of gU and of the scalars' first-layer terms used to be redone each time -- a third of the
// pass's VALU instructions.  Now they are computed once per pass and the round's setup is a handful of loads.
struct BwdPrep {
    uint2* gu_frag;         // [L][8]: fp16 hi (4 x uint2: channels 0-15) | lo of gU x gu_scale x DZ_UP -- the A fragments [m = entry][k = channel]
    uint2* gt_frag;         // [L/16][2][16][4]: per tile hi | lo of (gU x gu_scale x GT_UP)^T -- the A fragments [m = channel][k = entry]; or NULL
    uint32_t* sq;           // [L][4 NM]: the scalars' split terms as the first layer's (fp16, fp16) words, per (m, g)
    int32_t* oth;           // [L]: row (b N + node) of the entry's other endpoint, -1 = padding
    int32_t* own;           // [L/16]: row of the tile's key node
    float* scal_l;          // [L]: the entry's scalar (S == 1, with d/d W_s), or NULL
};

__host__ __device__ inline size_t prep_bytes(int64_t L, int NM, bool w2, bool s1)
{
    return (size_t)L * 64 + (w2 ? (size_t)L * 64 : 0) + (size_t)L * 16 * NM + (size_t)L * 4 + (size_t)(L / 16) * 4 + (s1 ? (size_t)L * 4 : 0);
}

inline BwdPrep prep_carve(void* work, int64_t L, int NM, bool w2, bool s1)
{
    char* c = static_cast<char*>(work);
    BwdPrep w;
    w.gu_frag = reinterpret_cast<uint2*>(c); c += (size_t)L * 64;
    w.gt_frag = w2 ? reinterpret_cast<uint2*>(c) : nullptr; c += w2 ? (size_t)L * 64 : 0;
    w.sq = reinterpret_cast<uint32_t*>(c); c += (size_t)L * 16 * NM;
    w.oth = reinterpret_cast<int32_t*>(c); c += (size_t)L * 4;
    w.scal_l = s1 ? reinterpret_cast<float*>(c) : nullptr; c += s1 ? (size_t)L * 4 : 0;
    w.own = reinterpret_cast<int32_t*>(c);
    return w;
}

template <int NM>
__global__ __launch_bounds__(256) void edge_bwd_prep_kernel(const egnn_edge_bwd_args p, const BwdPrep w)
{
    __shared__ _Float16 th[256][18], tl[256][18];           // (gU x GT_UP) halves of the workgroup's 16 tiles, for the transposed table
    const int tid = threadIdx.x;
    const int64_t q = (int64_t)blockIdx.x * 256 + tid;
    const bool inside = q < p.L;
    const int eid = inside ? p.ent[q] : -1;
    const bool valid = eid >= 0;
    const int ev = valid ? eid : 0;
    const int ig = ev / p.K;                                 // global node (b N + i)
    const int jg = (ig / p.N) * p.N + (p.idx ? p.idx[ev] : ev - ig * p.K);
    f32x4 gv[4];
#pragma unroll
    for (int c4 = 0; c4 < 4; ++c4) gv[c4] = valid ? *reinterpret_cast<const f32x4*>(p.gU + (size_t)eid * 16 + 4 * c4) : f32x4{0.f, 0.f, 0.f, 0.f};
    if (inside) {
        w.oth[q] = valid ? (p.by_dest ? ig : jg) : -1;
        if ((q & 15) == 0) w.own[q >> 4] = p.by_dest ? jg : ig;             // (padding sits behind a node's entries; whole padding tiles: row 0)
        if (w.scal_l) w.scal_l[q] = valid ? p.scal[(size_t)eid * p.S] : 0.f;
        f16x4 hi[4], lo[4];
#pragma unroll
        for (int c4 = 0; c4 < 4; ++c4) split4(gv[c4] * (p.gu_scale * DZ_UP), hi[c4], lo[c4]);
#pragma unroll
        for (int c4 = 0; c4 < 4; ++c4) {
            w.gu_frag[q * 8 + c4] = __builtin_bit_cast(uint2, hi[c4]);
            w.gu_frag[q * 8 + 4 + c4] = __builtin_bit_cast(uint2, lo[c4]);
        }
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            uint32_t wd[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                // split term tau = 4 m + g of scalar tau / 3, exactly the forward's (csrc/edge_fused.hip, _weights.py::scalar_table)
                const int tau = 4 * m + g;
                const int sidx = tau / 3, kind = tau - 3 * sidx;
                wd[g] = 0u;
                if (sidx < p.S && valid) {
                    float val = p.scal[(size_t)eid * p.S + sidx] * p.ws_inv_scale;
                    if (fabsf(val) >= 6.0e7f) val = __builtin_nanf("");
                    const _Float16 s1 = (_Float16)(val * (1.0f / 1024.0f));
                    const float r = val - (float)s1 * 1024.0f;
                    const _Float16 rh = (_Float16)r;
                    const _Float16 rl = (_Float16)(r - (float)rh);
                    wd[g] = kind == 0 ? pack_h2(s1, s1) : (kind == 1 ? pack_h2(rh, rh) : pack_h2(rl, (_Float16)0.f));
                }
            }
            *reinterpret_cast<uint4*>(w.sq + (q * NM + m) * 4) = uint4{wd[0], wd[1], wd[2], wd[3]};
        }
    }
    if (w.gt_frag) {
#pragma unroll
        for (int c4 = 0; c4 < 4; ++c4) {
            f16x4 h, l;
            split4(gv[c4] * (p.gu_scale * GT_UP), h, l);
#pragma unroll
            for (int u = 0; u < 4; ++u) { th[tid][4 * c4 + u] = h[u]; tl[tid][4 * c4 + u] = l[u]; }
        }
        __syncthreads();
        // 16 tiles x (hi | lo) x 16 channels x 2 half-rows of 8 entries = 1024 chunks of 16 bytes
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int x = tid + 256 * i;
            const int tile_l = x >> 6, half = (x >> 5) & 1, c = (x >> 1) & 15, eh = x & 1;
            const int64_t tile = (int64_t)blockIdx.x * 16 + tile_l;
            typedef _Float16 f16x8v __attribute__((ext_vector_type(8)));
            f16x8v v;
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = half ? tl[tile_l * 16 + 8 * eh + u][c] : th[tile_l * 16 + 8 * eh + u][c];
            if (tile < (p.L >> 4)) *reinterpret_cast<f16x8v*>(w.gt_frag + ((tile * 2 + half) * 16 + c) * 4 + 2 * eh) = v;
        }
    }
}

// NM: first-layer MFMAs of the scalar term (egnn_edge_mfmas(S)); ST: register bound on S (per-edge scalars); WANT_W2: also
// d/d W_2; WANT_S: also d/d W_s and d/d s; CH: steps of 32 hidden columns the workgroup owns
// PAIR: the two tiles of a round belong to one node (by source, 16 < K <= 32): one partial row per round = the node's row
// DROP: training-mode dropout behind edge_mlp's first Linear (egnn_pytorch.py:178-184): the forward's hash mask (csrc/egnn_common.h:
// row = global edge id, column = hidden unit) re-evaluated on z and on SiLU'
// DSM (more than five per-edge scalars): d/d s = dz W_s contracts over the HIDDEN index, which the transposed tiles hold as their N
// dimension -- per-lane FMAs (the S <= 5 variants) need 8 S registers and S multiply-adds per value.  Here every dz tile goes through
// wave-private LDS once, comes back as the B fragment [k = hidden][n = entry] and meets A = W_s^T fragments (split f16 x 3, staged in
// LDS like W2^T): one accumulator tile per entry tile whatever S is.
template <int NM, int ST, bool WANT_W2, bool WANT_S, int CH, bool PAIR = false, bool DROP = false, bool DSM = false>
__global__ __launch_bounds__(BW_THREADS, ((WANT_W2 && WANT_S) || ((WANT_W2 || WANT_S) && ST > 1)) ? 2 : (WANT_W2 ? EGNN_BWD_W2_BLOCKS : (WANT_S ? (DROP ? 3 : EGNN_BWD_S_BLOCKS) : EGNN_BWD_DEST_BLOCKS))) void edge_bwd_kernel(const egnn_edge_bwd_args p, const BwdPrep w)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    _Float16* w2t = reinterpret_cast<_Float16*>(smem);                                  // [CH][hb][hi|lo][64][4] halves
    uint32_t* wstl = reinterpret_cast<uint32_t*>(smem + CH * 2048);                     // [CH * 32][4 NM] words
    float* xchall = reinterpret_cast<float*>(smem + CH * 2048 + CH * 32 * 4 * NM * 4);  // [waves][32][XLD]
    constexpr int DLD = 20;                                                             // floats per row of a transposed dz tile (80 B: 16-byte reads)
    _Float16* wsT = reinterpret_cast<_Float16*>(smem + CH * 2048 + CH * 32 * 4 * NM * 4 + BW_WAVES * 32 * XLD * 4);     // DSM: [CH][hb][hi|lo][64][4] halves
    float* dztall = reinterpret_cast<float*>(smem + CH * 2048 + CH * 32 * 4 * NM * 4 + BW_WAVES * 32 * XLD * 4 + CH * 2048);   // DSM: [waves][2][16][DLD]
    static_assert(!DSM || (WANT_S && ST > 1), "DSM: d/d s on the matrix cores goes with the MFMA form of d/d W_s");

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hq = lane & 15;                    // n index of the MFMA layouts: hidden unit (data), edge / row (A fragments)
    const int g = lane >> 4;
    const int S = p.S, N = p.N;
    const float rows_scale = p.inv_scale * (1.0f / DZ_UP);      // everything derived from dz leaves in natural units
    const float w2_out_scale = 1.0f / (p.gu_scale * GT_UP * A_UP);

    // Workgroup -> (slab of the entry list, column chunk).  What a workgroup gathers is the chunk's CH x 128 bytes of the rows
    // its entries point at; rows are shared between the entries of one graph only.  So the workgroups that are resident on an
    // XCD together (its L2) should work on the same graph and the same chunk: blocks go round robin over the 8 XCDs, v is the
    // position within the XCD's contiguous share of the grid, and consecutive v walk the GS slabs of a group before the next
    // chunk (measured: with chunk fastest the gathers hit L2 at 38 %, the forward's, which walks all columns of a graph's rows
    // together, at 85 %).
    const int n_chunks = (p.Hp / 32 + CH - 1) / CH;
    const int nblk = gridDim.x;
    const int v = xcd_remap(blockIdx.x, nblk);
    const int per_group = n_chunks * GS;
    const int group = v / per_group, rem = v - group * per_group;
    const int chunk = rem / GS;
    const int slab = group * GS + (rem - chunk * GS);
    if (slab >= p.n_slabs) return;                               // (the last group may be partial)
    // the hidden steps are dealt out evenly (9 steps in two chunks: 5 + 4, not 8 + 1 -- a chunk of one step pays a round's setup and
    // its exposed first gather for 32 columns)
    const int steps_total = p.Hp / 32;
    const int base = steps_total / n_chunks, extra = steps_total - base * n_chunks;
    const int st0 = chunk * base + (chunk < extra ? chunk : extra);
    const int nst = base + (chunk < extra ? 1 : 0);

    // the chunk's W2^T fragments and scalar-weight table: once per workgroup
    {
        const uint4* src = reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(p.W2Th) + (size_t)st0 * 2048);
        for (int o = tid; o < nst * 128; o += BW_THREADS) reinterpret_cast<uint4*>(w2t)[o] = src[o];
        const uint32_t* ts = reinterpret_cast<const uint32_t*>(p.Wst) + (size_t)st0 * 32 * 4 * NM;
        for (int o = tid; o < nst * 32 * 4 * NM; o += BW_THREADS) wstl[o] = ts[o];
        if constexpr (DSM) {
            const uint4* srcs = reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(p.WsTh) + (size_t)st0 * 2048);
            for (int o = tid; o < nst * 128; o += BW_THREADS) reinterpret_cast<uint4*>(wsT)[o] = srcs[o];
        }
    }
    __syncthreads();

    const float* Town = p.by_dest ? p.Pj : p.Pi;
    const float* Toth = p.by_dest ? p.Pi : p.Pj;
    const uint32_t tab_bytes = (uint32_t)((size_t)p.B * N * p.ldp * 4);
    const __amdgpu_buffer_rsrc_t own_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Town), 0, tab_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t oth_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Toth), 0, tab_bytes, 0x00020000);

    // partial rows: row = tile index; wave-uniform, so it rides in the scalar offset and the lane contributes 4 hq only
    const uint32_t row_bytes = (uint32_t)(p.ld_rows * 4);
    const __amdgpu_buffer_rsrc_t rows_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.part_rows, 0, (uint32_t)(p.L >> 4) * row_bytes, 0x00020000);
    const uint32_t hq4 = hq * 4;
    // (32-bit offsets here as well: a 64-bit per-lane address per half step would be hoisted out of the round loop -- 2 registers each)
    const __amdgpu_buffer_rsrc_t ws_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(WANT_S ? p.Ws : p.Pi), 0, (uint32_t)((size_t)p.Hp * S * 4), 0x00020000);

    // -I_16 as an A fragment (lane (g, m): k = 4g .. 4g+3): the residual of a split on the matrix cores
    f16x4 neg_ident;
#pragma unroll
    for (int r = 0; r < 4; ++r) neg_ident[r] = (hq == 4 * g + r) ? (_Float16)-1.f : (_Float16)0.f;
    float* xch = xchall + wave * (32 * XLD);
    // parking: lane l holds 16-byte chunk l & 7 of the line of entry 8 qq + (l >> 3)
    float* xw = xch + (lane >> 3) * XLD + 4 * (lane & 7);
    // transposed pick-up: lane (g, hq) reads hidden unit 16 hb + hq of entries 16 t + 4 g + r
    const float* xr = xch + (4 * g) * XLD + hq;

    f32x4 dW2[WANT_W2 ? 2 * CH : 1];              // d/d W_2: rows c = 4g + r, column = chunk column 16 blk + hq
    // d/d W_s.  S = 1: per-lane FMAs against the entry's scalar (one register per column block).  S > 1: like d/d W_2, three
    // MFMAs per tile with A = the tile's scalars transposed (split f16, columns pre-scaled by the host's powers of two) -- one
    // accumulator tile per column block whatever S is, instead of S registers of scalars per entry and S per block.
    constexpr bool MW = WANT_S && ST > 1;
    float dws[(WANT_S && !MW) ? 2 * CH : 1][ST];  // d/d W_s partial over this lane group's edges
    f32x4 dWsm[MW ? 2 * CH : 1];                  // rows c = 4g + r (scalar index), column = chunk column 16 blk + hq
#pragma unroll
    for (int bl = 0; bl < (WANT_W2 ? 2 * CH : 1); ++bl) dW2[bl] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int bl = 0; bl < ((WANT_S && !MW) ? 2 * CH : 1); ++bl)
#pragma unroll
        for (int c = 0; c < ST; ++c) dws[bl][c] = 0.f;
#pragma unroll
    for (int bl = 0; bl < (MW ? 2 * CH : 1); ++bl) dWsm[bl] = f32x4{0.f, 0.f, 0.f, 0.f};

    uint32_t rows_max = 0u;                      // max |partial row value| (rows_amax: the rows are final node rows with one or two tiles per node)
    const int64_t rounds_total = p.L / 128;
    const int64_t per_slab = (rounds_total + p.n_slabs - 1) / p.n_slabs;
    const int64_t rd0 = (int64_t)slab * per_slab;
    const int64_t rd1 = (rd0 + per_slab) < rounds_total ? (rd0 + per_slab) : rounds_total;

    for (int64_t round = rd0; round < rd1; ++round) {
        const int q0 = (int)round * 128 + wave * 32;            // (L < 2^31: 32-bit entry indices keep the list addresses in scalar base + offset form)
        // ---------------------------------------------------------------- per-entry setup
        // A fragments indexed by edge (m = hq): gU (channels 4g .. 4g+3) and the scalar terms of the first layer
        // (all of it read from the pass's per-entry records, edge_bwd_prep_kernel)
        f16x4 guhi[2], gulo[2];
        u32x2 bq[2][NM];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int q = q0 + 16 * t + hq;
            guhi[t] = __builtin_bit_cast(f16x4, w.gu_frag[(size_t)q * 8 + g]);
            gulo[t] = __builtin_bit_cast(f16x4, w.gu_frag[(size_t)q * 8 + 4 + g]);
#pragma unroll
            for (int m = 0; m < NM; ++m) bq[t][m] = u32x2{0u, w.sq[((size_t)q * NM + m) * 4 + g]};
        }
#if EGNN_BWD_SKIP_DEAD
        {
            // Every entry of the wave's two tiles has gU = 0 -- the edges of a padded node (mask = 0: their messages reach no output), or
            // list padding: dz = (W2^T gU) SiLU'(z) is an exact zero whatever z is, and so is everything contracted from it.  The wave
            // writes the zeros the pass owes for these entries (their partial rows, d/d s) and takes the next round; the waves of a
            // workgroup meet only after the loop.  A batch of ragged graphs spends a fifth of both passes here.
            const u32x2 z0 = __builtin_bit_cast(u32x2, guhi[0]), z1 = __builtin_bit_cast(u32x2, guhi[1]);
            const u32x2 z2 = __builtin_bit_cast(u32x2, gulo[0]), z3 = __builtin_bit_cast(u32x2, gulo[1]);
            // (sign bits aside: a masked edge's gU is (+0) x SiLU'(u), which is -0 where SiLU' is negative)
            const uint32_t nz = (z0[0] | z0[1] | z1[0] | z1[1] | z2[0] | z2[1] | z3[0] | z3[1]) & 0x7fff7fffu;
            if (__builtin_amdgcn_ballot_w64(nz != 0u) == 0ull) {
                for (int st = 0; st < nst; ++st)
#pragma unroll
                    for (int hb = 0; hb < 2; ++hb) {
                        const int col4 = ((st0 + st) * 32 + 16 * hb) * 4;
                        if constexpr (PAIR) {
                            buf_store1f(rows_rsrc, hq4, (int)(((uint32_t)q0 >> 5) * row_bytes) + col4, 0.f);
                        } else {
                            buf_store1f(rows_rsrc, hq4, (int)(((uint32_t)q0 >> 4) * row_bytes) + col4, 0.f);
                            buf_store1f(rows_rsrc, hq4, (int)(((uint32_t)(q0 + 16) >> 4) * row_bytes) + col4, 0.f);
                        }
                    }
                if constexpr (WANT_S) {
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        if constexpr (DSM) {
                            const int eid = p.ent[q0 + 16 * t + hq];
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                if (eid >= 0 && 4 * g + r < S) p.ds_part[((size_t)chunk * p.E + eid) * S + 4 * g + r] = 0.f;
                        } else {
                            const i32x4 e4 = *reinterpret_cast<const i32x4*>(p.ent + q0 + 16 * t + 4 * g);
#pragma unroll
                            for (int r = 0; r < 4; ++r)
#pragma unroll
                                for (int c = 0; c < ST; ++c)
                                    if (hq == 0 && c < S && e4[r] >= 0) p.ds_part[((size_t)chunk * p.E + e4[r]) * S + c] = 0.f;
                        }
                    }
                }
                continue;
            }
        }
#endif
        // per register r: entry 16 t + 4 g + r (the edges this lane's data registers belong to).  The 16 entries of a tile share
        // their key node (the host pads every node's entries to whole tiles): one own row and one partial row per tile.
        int ownoff[2];                             // byte offset of the tile's own row: wave-uniform, rides in the scalar offset
        float sv[2][4][MW ? 1 : ST];
        f16x4 sth[2], stl[2];                      // MW: the tile's scalars transposed, A fragment [m = scalar][k = entry]
        f16x4 gth[2], gtl[2], ind[2];
        uint32_t ekey[DROP ? 2 : 1][4];             // DROP: the mask's row key of entry (t, r)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const i32x4 e4 = *reinterpret_cast<const i32x4*>(p.ent + q0 + 16 * t + 4 * g);
            if constexpr (DROP) {
#pragma unroll
                for (int r = 0; r < 4; ++r) ekey[t][r] = egnn_drop_base(p.drop_seed, EGNN_DROP_SITE_EDGE, (uint32_t)((int64_t)e4[r] + p.drop_eid0));
            }
            const int tile = (q0 >> 4) + t;
            ownoff[t] = (int)((size_t)__builtin_amdgcn_readfirstlane(w.own[tile]) * p.ldp * 4);
            f32x4 st4 = f32x4{0.f, 0.f, 0.f, 0.f};
            if constexpr (WANT_S && !MW) {                            // (ST == 1: the entries' scalar, zero for padding)
                const f32x4 s4 = *reinterpret_cast<const f32x4*>(w.scal_l + q0 + 16 * t + 4 * g);
#pragma unroll
                for (int r = 0; r < 4; ++r) sv[t][r][0] = s4[r];
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int eid = e4[r];
                const bool valid = eid >= 0;
                // rows 0, 4, 8, 12 of D = the tile's sum over its valid entries: register 0 of every lane group holds it, and all
                // four write the same value to the tile's partial row (no per-lane row select, no spare row)
                ind[t][r] = (valid && (hq & 3) == 0) ? (_Float16)1.f : (_Float16)0.f;
                if constexpr (MW) {
                    if (valid && hq < S) st4[r] = p.scal[(size_t)eid * S + hq] * p.scal_scale[hq];
                }
            }
            if constexpr (WANT_W2) {
                gth[t] = __builtin_bit_cast(f16x4, w.gt_frag[((size_t)tile * 2 * 16 + hq) * 4 + g]);
                gtl[t] = __builtin_bit_cast(f16x4, w.gt_frag[((size_t)tile * 2 * 16 + 16 + hq) * 4 + g]);
            }
            if constexpr (MW) split4(st4 * GT_UP, sth[t], stl[t]);
        }
        // whole-line gathers of the other endpoint's rows: lane l fetches chunk l & 7 of entry 8 qq + (l >> 3)
        uint32_t goff[4];
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
            const int o = w.oth[q0 + 8 * qq + (lane >> 3)];
            const int othrow = o >= 0 ? o : 0;
            goff[qq] = (uint32_t)(((size_t)othrow * p.ldp + 4 * (lane & 7)) * 4);
        }

        float ps[(WANT_S && !DSM) ? 2 : 1][4][ST];          // d/d s of entry (t, r): this lane's hidden units only (summed over hq at the end)
        f32x4 dsacc[DSM ? 2 : 1];                  // DSM: d/d s^T of tile t: rows s = 4g + r, column = entry hq
#pragma unroll
        for (int t = 0; t < (DSM ? 2 : 1); ++t) dsacc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < ((WANT_S && !DSM) ? 2 : 1); ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < ST; ++c) ps[t][r][c] = 0.f;

        // ---------------------------------------------------------------- the chunk's steps, one step of loads ahead
        f32x4 gl[4];
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) gl[qq] = buf_load4(oth_rsrc, goff[qq], st0 * 32 * 4);
        float ownpf[2][2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int hb = 0; hb < 2; ++hb) ownpf[t][hb] = buf_load1f(own_rsrc, hq4, ownoff[t] + (st0 * 32 + 16 * hb) * 4);
        // One step of 32 hidden columns (a lambda so that the unrolled loop below indexes the accumulator tiles with constants)
        auto step = [&](const int st) __attribute__((always_inline)) {
            const int hoff = (st0 + st) * 32;
            const int hnext = (st + 1 < nst) ? hoff + 32 : hoff;            // last step: harmless re-read
            // The tile's own row: every step reads a new 128-byte line of it (an L1 miss each time), fetched one step ahead like
            // the gathers.
            float ow[2][2];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int hb = 0; hb < 2; ++hb) {
                    ow[t][hb] = ownpf[t][hb];
                    ownpf[t][hb] = buf_load1f(own_rsrc, hq4, ownoff[t] + (hnext + 16 * hb) * 4);
                }
            // park the neighbour lines of this step, pick them up transposed (same wave: DS operations execute in order)
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) *reinterpret_cast<f32x4*>(xw + qq * 8 * XLD) = gl[qq];
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) gl[qq] = buf_load4(oth_rsrc, goff[qq], hnext * 4);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();

#pragma unroll
            for (int hb = 0; hb < 2; ++hb) {
                const int col = hoff + 16 * hb + hq;
                // z^T of this half step: the parked neighbour rows, transposed, + the own rows
                f32x4 x[2];
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) x[t][r] = xr[(16 * t + r) * XLD + 16 * hb] + ow[t][hb];
                // first Linear's scalar term, operands swapped against the forward: D[edge][hidden] += [scalars] x [W_s]
#pragma unroll
                for (int m = 0; m < NM; ++m) {
                    const u32x2 bv = u32x2{0u, wstl[(st * 32 + 16 * hb + hq) * (4 * NM) + 4 * m + (g ^ ((hq >> 2) & 2))]};   // (the table's pair swap for units 8 .. 15)
#pragma unroll
                    for (int t = 0; t < 2; ++t)
                        x[t] = __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(f16x4, bq[t][m]), __builtin_bit_cast(f16x4, bv), x[t], 0, 0, 0);
                }
                // W2^T fragments: [step][hb][hi|lo][lane = 16 g + hq][u] = W2[4 g + u][32 step + 16 hb + hq]  (B operand: k = channel)
                const f16x4* wt = reinterpret_cast<const f16x4*>(w2t) + (size_t)(st * 2 + hb) * 2 * 64;
                const f16x4 wthi = wt[lane], wtlo = wt[64 + lane];
                float wsn[ST];
                if constexpr (WANT_S && !DSM) {
#pragma unroll
                    for (int c = 0; c < ST; ++c) wsn[c] = c < S ? buf_load1f(ws_rsrc, (uint32_t)((hq * S + c) * 4), (hoff + 16 * hb) * S * 4) : 0.f;
                }
                f32x4 dPc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    f32x4 ga = f32x4{0.f, 0.f, 0.f, 0.f};
                    ga = __builtin_amdgcn_mfma_f32_16x16x16f16(guhi[t], wthi, ga, 0, 0, 0);
                    ga = __builtin_amdgcn_mfma_f32_16x16x16f16(guhi[t], wtlo, ga, 0, 0, 0);
                    ga = __builtin_amdgcn_mfma_f32_16x16x16f16(gulo[t], wthi, ga, 0, 0, 0);
                    // y = -log2(e) z;  sigma = 1 / (1 + 2^y);  a = SiLU(z) = z sigma;  SiLU'(z) = sigma + a (1 - sigma).
                    // Units: a carries A_UP, dz = ga SiLU' carries gu_scale x w2t_scale x DZ_UP (DZ_UP rides in the gU fragments),
                    // so that the lo halves of small values stay off fp16's subnormal range (2^-24 absolute resolution); undone at the outputs.
                    float sgv[4], znv[4], spv[4];
                    f32x4 av4, dz4;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float y = x[t][r];
                        bool keep = true;
                        if constexpr (DROP) {                      // (y = -log2(e) z: the rescaling commutes; a dropped unit sees z = 0)
                            keep = egnn_drop_hash(ekey[t][r], (uint32_t)col) >= p.drop_thr;
                            y = keep ? y * p.drop_inv_keep : 0.f;
                        }
                        sgv[r] = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(y));
                        if constexpr (WANT_W2) {
                            znv[r] = y * (-0.6931471805599453f * A_UP);
                            av4[r] = znv[r] * sgv[r];
                            const float t1 = __builtin_fmaf(sgv[r], -1.0f / A_UP, 1.0f / A_UP);      // (1 - sigma) / A_UP
                            spv[r] = __builtin_fmaf(av4[r], t1, sgv[r]);
                        } else {
                            // a = SiLU(z) is not needed without d/d W_2: SiLU'(z) = sigma (1 + z (1 - sigma)), z = -ln2 y -- one instruction fewer
                            const float tc = __builtin_fmaf(sgv[r], 0.6931471805599453f, -0.6931471805599453f);      // -ln2 (1 - sigma)
                            spv[r] = sgv[r] * __builtin_fmaf(y, tc, 1.0f);
                            znv[r] = 0.f; av4[r] = 0.f;
                        }
                        if constexpr (DROP) spv[r] = keep ? spv[r] * p.drop_inv_keep : 0.f;      // d (dropped z) / d z
                        dz4[r] = ga[r] * spv[r];
                        asm("" : "+v"(dz4[r]));                    // (scalar multiplies: v_pk_mul_f32 costs 9.3 cycles per pair against 2 x 2.9)
                    }
                    if constexpr (DSM) {
                        // dz tile -> LDS [entry][hidden] -> B fragment [k = hidden 4g + r][n = entry hq]; A = W_s^T [m = scalar][k = hidden]
                        float* tl = dztall + (wave * 2 + t) * (16 * DLD);
#pragma unroll
                        for (int r = 0; r < 4; ++r) tl[(4 * g + r) * DLD + hq] = dz4[r];
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        __builtin_amdgcn_wave_barrier();
                        const f32x4 bt = *reinterpret_cast<const f32x4*>(tl + hq * DLD + 4 * g);
                        f16x4 bh, bl;
                        split4(bt, bh, bl);
                        const f16x4* wf = reinterpret_cast<const f16x4*>(wsT) + (size_t)(st * 2 + hb) * 2 * 64;
                        const f16x4 ash = wf[lane], asl = wf[64 + lane];
                        f32x4 dd = dsacc[t];
                        dd = __builtin_amdgcn_mfma_f32_16x16x16f16(ash, bh, dd, 0, 0, 0);
                        dd = __builtin_amdgcn_mfma_f32_16x16x16f16(ash, bl, dd, 0, 0, 0);
                        dd = __builtin_amdgcn_mfma_f32_16x16x16f16(asl, bh, dd, 0, 0, 0);
                        dsacc[t] = dd;
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        __builtin_amdgcn_wave_barrier();
                    }
                    if constexpr (WANT_S && !DSM) {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
#pragma unroll
                            for (int c = 0; c < ST; ++c) {
                                ps[t][r][c] = __builtin_fmaf(dz4[r], wsn[c], ps[t][r][c]);
                                if constexpr (!MW) dws[2 * st + hb][c] = __builtin_fmaf(dz4[r], sv[t][r][c], dws[2 * st + hb][c]);
                            }
                    }
                    // The tile's sum of dz.  Without a second use of dz on the matrix cores (d/d W_s at S > 1) it is plain fp32 adds: the
                    // lane's four entries, then the four lane groups through the row-swap instructions (egnn_column_sum4_reg) -- no
                    // (hi, lo) split of dz (a v_fma_mix_f32 and a conversion per value) and two MFMAs per tile fewer; padding entries carry
                    // gU = 0, so their dz is 0.
                    if constexpr (!MW && EGNN_BWD_VALU_TILE_SUM) {
                        float tsum = egnn_column_sum4_reg((dz4[0] + dz4[1]) + (dz4[2] + dz4[3]));
                        if constexpr (PAIR) {
                            if (t == 0) dPc[0] = tsum;
                            else {
                                const float rv = (dPc[0] + tsum) * rows_scale;
                                buf_store1f(rows_rsrc, hq4, (int)(((uint32_t)q0 >> 5) * row_bytes) + (hoff + 16 * hb) * 4, rv);
                                const uint32_t rb = egnn_abs_bits(rv);
                                rows_max = rows_max > rb ? rows_max : rb;
                            }
                        } else {
                            const float rv = tsum * rows_scale;
                            buf_store1f(rows_rsrc, hq4, (int)(((uint32_t)(q0 + 16 * t) >> 4) * row_bytes) + (hoff + 16 * hb) * 4, rv);
                            const uint32_t rb = egnn_abs_bits(rv);
                            rows_max = rows_max > rb ? rows_max : rb;
                        }
                    }
                    // (hi, lo) halves: hi = RNE pair conversion, lo = product - hi as ONE v_fma_mix_f32 per value (the f16 operand read in place)
                    f16x4 dh, dl;
                    if constexpr (MW || !EGNN_BWD_VALU_TILE_SUM) {
#pragma unroll
                    for (int r = 0; r < 4; r += 2) {
                        const f16x2 hi = __builtin_convertvector((f32x2v){dz4[r], dz4[r + 1]}, f16x2);
                        const uint32_t hw = __builtin_bit_cast(uint32_t, hi);
                        float l0, l1;
                        asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(l0) : "v"(ga[r]), "v"(spv[r]), "v"(hw));
                        asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(l1) : "v"(ga[r + 1]), "v"(spv[r + 1]), "v"(hw));
                        const f16x2 lo = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(l0, l1));
                        dh[r] = hi[0]; dh[r + 1] = hi[1];
                        dl[r] = lo[0]; dl[r + 1] = lo[1];
                    }
                    // sum over each node's edges of the tile: rows = the tile's local nodes
                    }
                    if constexpr (MW || !EGNN_BWD_VALU_TILE_SUM) {
                    f32x4 dP = (PAIR && t == 1) ? dPc : f32x4{0.f, 0.f, 0.f, 0.f};
                    dP = __builtin_amdgcn_mfma_f32_16x16x16f16(ind[t], dh, dP, 0, 0, 0);
                    dP = __builtin_amdgcn_mfma_f32_16x16x16f16(ind[t], dl, dP, 0, 0, 0);
                    // (no branch around the store: the step stays one basic block and the scheduler interleaves the two tiles)
                    if constexpr (PAIR) {
                        if (t == 0) dPc = dP;
                        else buf_store1f(rows_rsrc, hq4, (int)(((uint32_t)q0 >> 5) * row_bytes) + (hoff + 16 * hb) * 4, dP[0] * rows_scale);
                    } else {
                        buf_store1f(rows_rsrc, hq4, (int)(((uint32_t)(q0 + 16 * t) >> 4) * row_bytes) + (hoff + 16 * hb) * 4, dP[0] * rows_scale);
                    }
                    }
                    if constexpr (MW) {
                        f32x4 d = dWsm[2 * st + hb];
                        d = __builtin_amdgcn_mfma_f32_16x16x16f16(sth[t], dh, d, 0, 0, 0);
                        d = __builtin_amdgcn_mfma_f32_16x16x16f16(sth[t], dl, d, 0, 0, 0);
                        d = __builtin_amdgcn_mfma_f32_16x16x16f16(stl[t], dh, d, 0, 0, 0);
                        dWsm[2 * st + hb] = d;
                    }
                    if constexpr (WANT_W2 && EGNN_BWD_LO_MFMA) {
                        // (hi, lo) of a = SiLU(z): hi by pair conversion, the residual a - hi on the matrix cores (A = -I, B = hi, C = a:
                        // exact), like the forward's split -- no v_fma_mix_f32 per value
                        f16x4 ah, al;
#pragma unroll
                        for (int r = 0; r < 4; r += 2) {
                            const f16x2 hi = __builtin_convertvector((f32x2v){av4[r], av4[r + 1]}, f16x2);
                            ah[r] = hi[0]; ah[r + 1] = hi[1];
                        }
                        const f32x4 lo32 = __builtin_amdgcn_mfma_f32_16x16x16f16(neg_ident, ah, av4, 0, 0, 0);
#pragma unroll
                        for (int r = 0; r < 4; r += 2) {
                            const f16x2 lo = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(lo32[r], lo32[r + 1]));
                            al[r] = lo[0]; al[r + 1] = lo[1];
                        }
                        f32x4 d = dW2[2 * st + hb];
                        d = __builtin_amdgcn_mfma_f32_16x16x16f16(gth[t], ah, d, 0, 0, 0);
                        d = __builtin_amdgcn_mfma_f32_16x16x16f16(gth[t], al, d, 0, 0, 0);
                        d = __builtin_amdgcn_mfma_f32_16x16x16f16(gtl[t], ah, d, 0, 0, 0);
                        dW2[2 * st + hb] = d;
                    }
                    if constexpr (WANT_W2 && !EGNN_BWD_LO_MFMA) {
                        f16x4 ah, al;
#pragma unroll
                        for (int r = 0; r < 4; r += 2) {
                            float a0 = av4[r], a1 = av4[r + 1];
                            asm("" : "+v"(a0));
                            asm("" : "+v"(a1));
                            const f16x2 hi = __builtin_convertvector((f32x2v){a0, a1}, f16x2);
                            const uint32_t hw = __builtin_bit_cast(uint32_t, hi);
                            float l0, l1;
                            asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(l0) : "v"(znv[r]), "v"(sgv[r]), "v"(hw));
                            asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(l1) : "v"(znv[r + 1]), "v"(sgv[r + 1]), "v"(hw));
                            const f16x2 lo = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(l0, l1));
                            ah[r] = hi[0]; ah[r + 1] = hi[1];
                            al[r] = lo[0]; al[r + 1] = lo[1];
                        }
                        f32x4 d = dW2[2 * st + hb];
                        d = __builtin_amdgcn_mfma_f32_16x16x16f16(gth[t], ah, d, 0, 0, 0);
                        d = __builtin_amdgcn_mfma_f32_16x16x16f16(gth[t], al, d, 0, 0, 0);
                        d = __builtin_amdgcn_mfma_f32_16x16x16f16(gtl[t], ah, d, 0, 0, 0);
                        dW2[2 * st + hb] = d;
                    }
                }
            }
            // the next step's parking stores come after this step's pick-up loads (same wave: DS operations execute in order)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_sched_barrier(0);          // keep the unrolled steps apart: hoisting the next step's loads costs registers
        };
#pragma unroll
        for (int st = 0; st < CH; ++st)
            if (st < nst) step(st);
        if constexpr (DSM) {
            // d/d s of this chunk's columns, from the accumulator tiles: lane (g, hq) holds scalars 4g .. 4g+3 of entry hq
            const float ds_scale = rows_scale * p.wst_inv_scale;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int eid = p.ent[q0 + 16 * t + hq];
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (eid >= 0 && 4 * g + r < S) p.ds_part[((size_t)chunk * p.E + eid) * S + 4 * g + r] = dsacc[t][r] * ds_scale;
            }
        }
        if constexpr (WANT_S && !DSM) {
            // d/d s of this chunk's columns: sum over the 16 hidden units of the row, one partial per (chunk, edge)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const i32x4 e4 = *reinterpret_cast<const i32x4*>(p.ent + q0 + 16 * t + 4 * g);
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int c = 0; c < ST; ++c) {
                        const float v = egnn_row16_sum(ps[t][r][c]) * rows_scale;
                        if (hq == 0 && c < S && e4[r] >= 0) p.ds_part[((size_t)chunk * p.E + e4[r]) * S + c] = v;
                    }
            }
        }
    }

    if constexpr (!(WANT_S && ST > 1) && EGNN_BWD_VALU_TILE_SUM) {
        if (p.rows_amax) {
            __shared__ uint32_t amax_slot;
            egnn_block_absmax_commit(rows_max, &amax_slot, p.rows_amax);
        }
    }
    // The all-edge partials: one per WORKGROUP for d/d W_2 and (S = 1) d/d W_s -- the four waves' accumulators (and the four lane
    // groups' of d/d W_s) are added up through LDS in a fixed order (wave 0 .. 3, lane group 0 .. 3): 4x / 16x smaller partial
    // arrays for the host's final sum (1.1 GB -> 0.27 GB at the north-star shape).
    if constexpr (WANT_W2) {
        f32x4* red = reinterpret_cast<f32x4*>(xchall);
#pragma unroll
        for (int bl = 0; bl < 2 * CH; ++bl) {
            __syncthreads();
            if (wave > 0) red[(wave - 1) * 64 + lane] = dW2[bl];
            __syncthreads();
            if (wave == 0 && bl < 2 * nst) {
                f32x4 sum = dW2[bl];
                sum += red[lane];
                sum += red[64 + lane];
                sum += red[128 + lane];
                const int col = st0 * 32 + 16 * bl + hq;
#pragma unroll
                for (int r = 0; r < 4; ++r) p.dW2_part[((size_t)slab * 16 + 4 * g + r) * p.Hp + col] = sum[r] * w2_out_scale;
            }
        }
    }
    if constexpr (WANT_S && !MW) {
        static_assert(2 * CH * BW_THREADS * 4 <= BW_WAVES * 32 * XLD * 4, "the exchange buffer holds the d/d W_s partials");
        float* red = xchall;
        __syncthreads();
#pragma unroll
        for (int bl = 0; bl < 2 * CH; ++bl) red[bl * BW_THREADS + tid] = dws[bl][0];
        __syncthreads();
        for (int o = tid; o < 2 * nst * 16; o += BW_THREADS) {
            const int bl = o >> 4, h16 = o & 15;
            float sum = 0.f;
#pragma unroll
            for (int u = 0; u < 16; ++u) sum += red[bl * BW_THREADS + u * 16 + h16];       // u = 4 wave + lane group
            p.dWs_part[(size_t)slab * p.Hp + st0 * 32 + 16 * bl + h16] = sum * rows_scale;
        }
    }
    if constexpr (MW) {
        const size_t w = (size_t)slab * BW_WAVES + wave;
#pragma unroll
        for (int bl = 0; bl < 2 * CH; ++bl) {
            if (bl < 2 * nst) {
                const int col = st0 * 32 + 16 * bl + hq;
                {                             // one partial per wave (rows = scalars); the host undoes scal_scale
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (4 * g + r < S) p.dWs_part[((w * 4) * S + 4 * g + r) * p.Hp + col] = dWsm[bl][r] * (rows_scale * (1.0f / GT_UP));
                }
            }
        }
    }
}

template <int NM, int ST, bool W2, bool SS, int CH, bool PAIR = false, bool DROP = false, bool DSM = false>
int launch_v(const egnn_edge_bwd_args& a, hipStream_t s)
{
    if (a.row_pairs && !PAIR) return EGNN_E_UNSUPPORTED;
    const int n_chunks = (a.Hp / 32 + CH - 1) / CH;
    const size_t lds = (size_t)CH * 2048 + (size_t)CH * 32 * 4 * NM * 4 + (size_t)BW_WAVES * 32 * XLD * 4 +
                       (DSM ? (size_t)CH * 2048 + (size_t)BW_WAVES * 2 * 16 * 20 * 4 : 0);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(edge_bwd_kernel<NM, ST, W2, SS, CH, PAIR, DROP, DSM>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    const dim3 grid((unsigned)((a.n_slabs + GS - 1) / GS * GS * n_chunks));
    // the pass's per-entry records first (list order; read once per column chunk by the kernel below)
    constexpr bool S1 = SS && ST == 1;
    if (!a.work || a.work_bytes < (int64_t)prep_bytes(a.L, NM, W2, S1)) return EGNN_E_SHAPE;
    const BwdPrep w = prep_carve(a.work, a.L, NM, W2, S1);
    if (a.rows_amax) {
        if ((SS && ST > 1) || !EGNN_BWD_VALU_TILE_SUM) return EGNN_E_UNSUPPORTED;          // (the by-product exists where the tile sums are fp32 adds)
        if (hipMemsetAsync(a.rows_amax, 0, sizeof(uint32_t), s) != hipSuccess) return (int)hipGetLastError();
    }
    hipLaunchKernelGGL((edge_bwd_prep_kernel<NM>), dim3((unsigned)((a.L + 255) / 256)), dim3(256), 0, s, a, w);
    hipLaunchKernelGGL((edge_bwd_kernel<NM, ST, W2, SS, CH, PAIR, DROP, DSM>), grid, dim3(BW_THREADS), lds, s, a, w);
    return egnn_launch_status();
}

// more than five per-edge scalars (NM = 6: S <= 8, NM = 12: S <= 16): d/d W_s and d/d s on the matrix cores (DSM), d/d W_2 with the other pass
template <int NM>
int launch_many(const egnn_edge_bwd_args& a, hipStream_t s)
{
    if (a.dW2_part && a.dWs_part) return EGNN_E_UNSUPPORTED;
    if (a.drop_thr) {
        // training-mode dropout: the same variants with the forward's mask of z re-evaluated
        if (a.dW2_part) return launch_v<NM, 2, true, false, CH_W2, false, true>(a, s);
        if (a.dWs_part) {
            if (!a.WsTh || !(a.wst_inv_scale > 0.f)) return EGNN_E_NULLPTR;
            return a.row_pairs ? launch_v<NM, 2, false, true, CH_S, true, true, true>(a, s) : launch_v<NM, 2, false, true, CH_S, false, true, true>(a, s);
        }
        return launch_v<NM, 2, false, false, CH_S, false, true>(a, s);
    }
    if (a.dW2_part) return launch_v<NM, 2, true, false, CH_W2>(a, s);
    if (a.dWs_part) {
        if (!a.WsTh || !(a.wst_inv_scale > 0.f)) return EGNN_E_NULLPTR;
        return a.row_pairs ? launch_v<NM, 2, false, true, CH_S, true, false, true>(a, s) : launch_v<NM, 2, false, true, CH_S, false, false, true>(a, s);
    }
    return launch_v<NM, 2, false, false, CH_S>(a, s);
}

template <int NM, int ST>
int launch(const egnn_edge_bwd_args& a, hipStream_t s)
{
    if (a.drop_thr) {
        // training-mode dropout: the same variants with the forward's mask re-evaluated (their own instantiations: a hash per value)
        if (a.dW2_part && a.dWs_part) {
            if constexpr (ST == 1) return launch_v<NM, ST, true, true, CH_S, false, true>(a, s);
            else return EGNN_E_UNSUPPORTED;
        }
        if (a.dW2_part) return launch_v<NM, ST, true, false, CH_W2, false, true>(a, s);
        if (a.dWs_part) return a.row_pairs ? launch_v<NM, ST, false, true, CH_S, true, true>(a, s) : launch_v<NM, ST, false, true, CH_S, false, true>(a, s);
        return launch_v<NM, ST, false, false, CH_S, false, true>(a, s);
    }
    if (a.dW2_part && a.dWs_part) {
        // (everything in one pass: built for S = 1 only; with more scalars its registers do not fit 2 workgroups per CU)
        if constexpr (ST == 1) return launch_v<NM, ST, true, true, CH_S>(a, s);
        else return EGNN_E_UNSUPPORTED;
    }
    if (a.dW2_part) return launch_v<NM, ST, true, false, CH_W2>(a, s);
    if (a.dWs_part) return a.row_pairs ? launch_v<NM, ST, false, true, CH_S, true>(a, s) : launch_v<NM, ST, false, true, CH_S>(a, s);
    return launch_v<NM, ST, false, false, CH_S>(a, s);
}

}  // namespace

extern "C" int egnn_edge_bwd_chunk_steps(void) { return CH_S; }

extern "C" size_t egnn_edge_bwd_work_bytes(int64_t L, int S, int want_w2, int want_s)
{
    if (L <= 0 || S < 1 || S > 16) return 0;
    const int nm = egnn_edge_mfmas(S);
    return prep_bytes(L, nm, want_w2 != 0, want_s != 0 && S == 1);
}

extern "C" int egnn_edge_bwd_pass_f32(const egnn_edge_bwd_args* args, void* stream)
{
    if (!args) return EGNN_E_NULLPTR;
    const egnn_edge_bwd_args& a = *args;
    if (!a.ent || !a.Pi || !a.Pj || !a.Wst || !a.W2Th || !a.gU || !a.scal || !a.part_rows || !a.work) return EGNN_E_NULLPTR;
    if ((a.dWs_part == nullptr) != (a.ds_part == nullptr)) return EGNN_E_NULLPTR;
    if (a.dWs_part && !a.Ws) return EGNN_E_NULLPTR;
    if (a.row_pairs && (a.by_dest || a.K <= 16 || a.K > 32 || a.L != (((int64_t)a.B * a.N * 32 + 127) / 128) * 128)) return EGNN_E_SHAPE;
    if (a.B <= 0 || a.N <= 0 || a.K <= 0 || a.Hp <= 0 || (a.Hp % 32) != 0 || a.S < 1 || a.n_slabs < 1) return EGNN_E_SHAPE;
    if (a.L <= 0 || (a.L % 128) != 0 || a.ldp < a.Hp || (a.ldp % 4) != 0 || a.ld_rows < a.Hp) return EGNN_E_SHAPE;
    if (a.E != (int64_t)a.B * a.N * a.K || a.E >= ((int64_t)1 << 31) || a.L >= ((int64_t)1 << 31)) return EGNN_E_SHAPE;
    if ((size_t)a.B * a.N * a.ldp * 4 >= ((size_t)1 << 31)) return EGNN_E_UNSUPPORTED;           // 32-bit (signed scalar) buffer offsets into the P tables
    if ((size_t)(a.L >> 4) * a.ld_rows * 4 >= ((size_t)1 << 31)) return EGNN_E_UNSUPPORTED;         // (scalar offsets are signed)    // ... and into the partial rows
    if (a.S > 16 || a.wst_terms != 4 * egnn_edge_mfmas(a.S)) return EGNN_E_UNSUPPORTED;
    if (a.dWs_part && a.S > 1 && !a.scal_scale) return EGNN_E_NULLPTR;
    if (!(a.ws_inv_scale > 0.f) || !(a.gu_scale > 0.f) || !(a.inv_scale > 0.f)) return EGNN_E_SHAPE;
    if (a.drop_thr && !(a.drop_inv_keep >= 1.f)) return EGNN_E_SHAPE;
    if ((reinterpret_cast<uintptr_t>(a.Pi) & 15) || (reinterpret_cast<uintptr_t>(a.Pj) & 15) || (reinterpret_cast<uintptr_t>(a.W2Th) & 15) ||
        (reinterpret_cast<uintptr_t>(a.gU) & 15) || (reinterpret_cast<uintptr_t>(a.ent) & 15) ||
        (reinterpret_cast<uintptr_t>(a.Wst) & 3) || (reinterpret_cast<uintptr_t>(a.work) & 15))
        return EGNN_E_ALIGN;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (a.S == 1) return launch<1, 1>(a, s);
    if (a.S <= 4) return launch<3, 4>(a, s);
    if (a.S <= 5) return launch<4, 5>(a, s);
    if (a.S <= 8) return launch_many<6>(a, s);
    return launch_many<12>(a, s);
}
