// Fused pairwise squared distance + ranking overrides + per-row top-K selection (gfx950).
//
// Replaces egnn_pytorch/egnn_pytorch.py:232-233, 237-256, 258 of the reference without ever
// materialising an (N x N) tensor: one wavefront owns one query row i at a time; the graph's
// coordinates live in LDS as SoA (conflict-free: lane l reads x[c*64+l]); every lane keeps
// CPL = ceil(N/64) candidate keys in registers.
//
// Selection works on the order-preserving uint32 image of the fp32 ranking value.  Fast path (K <= 64): the K-th
// smallest of the 64 per-lane minima bounds the K-th smallest candidate, which prunes the row to ~1.3 K survivors;
// they are compacted into LDS (ballot prefix) and ranked by counting on the composite key (value, index).  General
// path (K > 64, or heavily tied rows such as fully masked ones): exact radix descent over all candidates (32
// wave-uniform steps of ballot+popcount), index-ordered pick among the ties with the K-th value, same compaction and
// ranking.  Both are deterministic: ascending value, ties by ascending index -- the tie policy of SURVEY.md §8c(5).
//
// Bound: VALU (N compares x 32 bits per row); HBM traffic is only coors in + (idx, rank) out.
#include "egnn_common.h"

namespace {

__device__ __forceinline__ uint32_t f2key(float f) {
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key2f(uint32_t k) {
    uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    return __uint_as_float(u);
}
__device__ __forceinline__ void wave_lds_sync() {
    // same-wave LDS hand-off: DS operations of one wave execute in issue order; the explicit wait makes the
    // store -> other-lane load dependency independent of that, the wave barrier pins the compiler's ordering.
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

#ifndef EGNN_KNN_BITS
#define EGNN_KNN_BITS 16
#endif
#ifndef EGNN_KNN_PAIR
#define EGNN_KNN_PAIR 1                      // two query rows per wave on the standard shape (see process_pair; 0: one row per wave everywhere)
#endif
#ifndef EGNN_KNN_ADJ_FAST
#define EGNN_KNN_ADJ_FAST 1                  // rows with >= K - 1 adjacent nodes straight from the adjacency row (see the row loop)
#endif
constexpr int KNN_PREFIX_BITS = EGNN_KNN_BITS;   // key bits resolved by the pruning threshold of the fast path
constexpr int KNN_SURVIVORS = 128;               // ... and survivors the fast path ranks directly (two per lane)
constexpr int KNN_THREADS = 256;
constexpr int KNN_WAVES = KNN_THREADS / 64;

// CDM: 3 = the fast path (C == 3, distance bit-exact as ((dx^2 + dy^2) + dz^2)); 8 = any 1 <= C <= 8 at run time.
template <int CPL, int CDM>
__global__ __launch_bounds__(KNN_THREADS) void knn_select_kernel(
    const float* __restrict__ coors, const uint8_t* __restrict__ mask, const uint8_t* __restrict__ adj,
    int64_t adj_bstride, int N, int K, int Npad, int Kpad, int rows_per_wg, int Cdim,
    int32_t* __restrict__ idx_out, float* __restrict__ rank_out)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int C = (CDM == 3) ? 3 : Cdim;
    float* xs = reinterpret_cast<float*>(smem);                         // [C][Npad], component-major
    uint8_t* ms = reinterpret_cast<uint8_t*>(xs + (size_t)C * Npad);
    const size_t cbytes = (size_t)Npad * (4 * C + 1);
    uint64_t* selall = reinterpret_cast<uint64_t*>(smem + cbytes + 8 - cbytes % 8);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.y;
    const int row0 = blockIdx.x * rows_per_wg;
    uint64_t* selbuf = selall + (size_t)wave * Kpad;

    const float* cb = coors + (size_t)b * N * C;
    if constexpr (CDM == 3) {
        // the graph's 3 N floats as one flat, coalesced stream, a batch of loads in flight at a time (round 6: one 12-byte-strided load
        // per coordinate with its LDS store behind it was eight dependent L2 round trips per workgroup at N = 2048 -- two thirds of
        // what was left of the kernel once the adjacency decides most rows)
        constexpr int BL = 12;
        const int total = 3 * Npad;
        for (int f0 = 0; f0 < total; f0 += BL * KNN_THREADS) {
            float v[BL];
#pragma unroll
            for (int t = 0; t < BL; ++t) {
                const int f = f0 + t * KNN_THREADS + tid;
                v[t] = f < 3 * N ? cb[f] : 0.f;
            }
#pragma unroll
            for (int t = 0; t < BL; ++t) {
                const int f = f0 + t * KNN_THREADS + tid;
                if (f < total) {
                    const int j = f / 3;
                    xs[(f - 3 * j) * Npad + j] = v[t];
                }
            }
        }
        for (int j = tid; j < Npad; j += KNN_THREADS) ms[j] = j < N ? (mask ? mask[(size_t)b * N + j] : (uint8_t)1) : (uint8_t)0;
    } else {
        for (int j = tid; j < Npad; j += KNN_THREADS) {
            for (int c = 0; c < C; ++c) xs[c * Npad + j] = j < N ? cb[j * C + c] : 0.f;
            ms[j] = j < N ? (mask ? mask[(size_t)b * N + j] : (uint8_t)1) : (uint8_t)0;
        }
    }
    __syncthreads();

    const uint64_t lt_mask = (1ull << lane) - 1ull;

#if EGNN_KNN_ADJ_FAST
    // bit r: some OTHER node has the first coordinate of this workgroup's row r (then a non-adjacent node could sit at distance exactly
    // 0.0 and tie with the adjacent ones: the row takes the general path) -- once per workgroup, every thread its <= 8 candidates
    // against the <= 32 rows (branch-free; per row and wave it was two thirds of a decided row's instructions)
    uint32_t* const dupflags = reinterpret_cast<uint32_t*>(selall + (size_t)KNN_WAVES * Kpad);
    if constexpr (CPL % 4 == 0 && CPL <= 32) {
        if (adj) {                                                       // (uniform)
            if (tid == 0) *dupflags = 0u;
            __syncthreads();
            uint32_t m = rows_per_wg > 32 ? 0xFFFFFFFFu : 0u;
            if (rows_per_wg <= 32) {
                constexpr int PT = CPL / 4;                              // candidates per thread: N <= 64 CPL = 256 PT
                float xc[PT];
#pragma unroll
                for (int t = 0; t < PT; ++t) xc[t] = xs[tid + KNN_THREADS * t];          // (past N: pad zeros / the next plane, masked below)
                // lane r holds the first coordinate of row r; v_readlane hands it to every lane (no LDS round trip per row)
                const int xrow_bits = __float_as_int((lane < 32 && row0 + lane < N) ? xs[row0 + lane] : 0.f);
#pragma unroll
                for (int r = 0; r < 32; ++r) {
                    const int i = row0 + r;
                    const float xi = __int_as_float(__builtin_amdgcn_readlane(xrow_bits, r));
                    uint32_t hit = 0u;
#pragma unroll
                    for (int t = 0; t < PT; ++t) {
                        const int j = tid + KNN_THREADS * t;
                        hit |= (uint32_t)(xc[t] == xi) & (uint32_t)(j != i) & (uint32_t)(j < N);
                    }
                    if (r < rows_per_wg && i < N) m |= hit << r;         // (uniform)
                }
            }
            if (m) atomicOr(dupflags, m);
            __syncthreads();
        }
    }
#endif

#if EGNN_KNN_PAIR
    // Two query rows per wave (round 5): the candidates' coordinates and mask bytes are read from LDS once for both rows, and the two
    // serial ballot / popcount chains of the threshold search interleave.  The standard shape only -- 3-D coordinates, K <= 32, no
    // adjacency, both rows unmasked, at most one survivor per lane in each row -- anything else returns false and the rows take the
    // single-row code below.  Same keys, same (value, index) ranking: the same bits.
    auto process_pair = [&](const int i0, const int i1) -> bool {
        const float ax = xs[i0], ay = xs[Npad + i0], az = xs[2 * Npad + i0];
        const float bx = xs[i1], by = xs[Npad + i1], bz = xs[2 * Npad + i1];
        uint32_t k0[CPL], k1[CPL];
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
            const int j = c * 64 + lane;
            uint32_t q0 = 0xFFFFFFFFu, q1 = 0xFFFFFFFFu;
            if (j < N) {
                const float xj = xs[j], yj = xs[Npad + j], zj = xs[2 * Npad + j];
                float dx, dy, dz;
                float r0 = egnn_sqdist(ax, ay, az, xj, yj, zj, dx, dy, dz);
                float r1 = egnn_sqdist(bx, by, bz, xj, yj, zj, dx, dy, dz);
                if (ms[j] == 0) { r0 = 1e5f; r1 = 1e5f; }            // :240-242 (both query rows are unmasked)
                q0 = f2key(r0);
                q1 = f2key(r1);
            }
            k0[c] = q0;
            k1[c] = q1;
        }
        uint32_t m0 = k0[0], m1 = k1[0];
#pragma unroll
        for (int c = 1; c < CPL; ++c) {
            m0 = k0[c] < m0 ? k0[c] : m0;
            m1 = k1[c] < m1 ? k1[c] : m1;
        }
        uint32_t M0 = 0, M1 = 0;
        int below0 = 0, below1 = 0;
        for (int bit = 31; bit >= 32 - KNN_PREFIX_BITS; --bit) {
            const int c0 = __popcll(__ballot((m0 >> bit) == (M0 >> bit)));
            const int c1 = __popcll(__ballot((m1 >> bit) == (M1 >> bit)));
            if (below0 + c0 < K) { below0 += c0; M0 |= (1u << bit); }
            if (below1 + c1 < K) { below1 += c1; M1 |= (1u << bit); }
        }
        M0 |= (1u << (32 - KNN_PREFIX_BITS)) - 1u;
        M1 |= (1u << (32 - KNN_PREFIX_BITS)) - 1u;
        int cl0 = 0, cl1 = 0;
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
            cl0 += k0[c] <= M0 ? 1 : 0;
            cl1 += k1[c] <= M1 ? 1 : 0;
        }
        const int incl = egnn_wave_inclusive_scan(cl0 | (cl1 << 16));   // (both prefix sums in one scan: counts stay below 2^16)
        const int tot = __builtin_amdgcn_readlane(incl, 63);
        const int S0 = tot & 0xffff, S1 = tot >> 16;
        if (S0 > 64 || S1 > 64) return false;                           // wave-uniform
        uint64_t* const sel0 = selbuf;
        uint64_t* const sel1 = selbuf + 64;
        int p0 = (incl & 0xffff) - cl0, p1 = (incl >> 16) - cl1;
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
            if (k0[c] <= M0) sel0[p0++] = ((uint64_t)k0[c] << 32) | (uint32_t)(c * 64 + lane);
            if (k1[c] <= M1) sel1[p1++] = ((uint64_t)k1[c] << 32) | (uint32_t)(c * 64 + lane);
        }
        wave_lds_sync();
        const uint64_t mine0 = lane < S0 ? sel0[lane] : ~0ull;
        const uint64_t mine1 = lane < S1 ? sel1[lane] : ~0ull;
        int rnk0 = 0, rnk1 = 0;
        const int Smin = S0 < S1 ? S0 : S1;
        for (int u = 0; u < Smin; ++u) {
            rnk0 += (sel0[u] < mine0) ? 1 : 0;
            rnk1 += (sel1[u] < mine1) ? 1 : 0;
        }
        for (int u = Smin; u < S0; ++u) rnk0 += (sel0[u] < mine0) ? 1 : 0;
        for (int u = Smin; u < S1; ++u) rnk1 += (sel1[u] < mine1) ? 1 : 0;
        const size_t ob0 = ((size_t)b * N + i0) * K, ob1 = ((size_t)b * N + i1) * K;
        if (lane < S0 && rnk0 < K) {
            idx_out[ob0 + rnk0] = (int32_t)(uint32_t)(mine0 & 0xFFFFFFFFull);
            rank_out[ob0 + rnk0] = key2f((uint32_t)(mine0 >> 32));
        }
        if (lane < S1 && rnk1 < K) {
            idx_out[ob1 + rnk1] = (int32_t)(uint32_t)(mine1 & 0xFFFFFFFFull);
            rank_out[ob1 + rnk1] = key2f((uint32_t)(mine1 >> 32));
        }
        wave_lds_sync();
        return true;
    };
#endif

    for (int r = wave; r < rows_per_wg; r += KNN_WAVES) {
        const int i = row0 + r;
        if (i >= N) break;                       // wave-uniform
#if EGNN_KNN_PAIR
        if constexpr (CDM == 3 && CPL <= 16) {                           // (N <= 1024: two rows' keys are 32 registers)
            if (K <= 32 && !adj && r + KNN_WAVES < rows_per_wg && i + KNN_WAVES < N && ms[i] != 0 && ms[i + KNN_WAVES] != 0) {
                if (process_pair(i, i + KNN_WAVES)) {
                    r += KNN_WAVES;
                    continue;
                }
            }
        }
#endif
        float ci[CDM];
#pragma unroll
        for (int c = 0; c < CDM; ++c) ci[c] = c < C ? xs[c * Npad + i] : 0.f;
        const bool mi = ms[i] != 0;
        const uint8_t* adjrow = adj ? adj + (size_t)b * adj_bstride + (size_t)i * N : nullptr;
        if (!mi && !adjrow) {
            // a masked (padded) row: every pair is masked, the whole ranking row is 1e5 (:240-242) and the selection is the
            // first K indices (ties by ascending index) -- what the general path below would find after a 32-step descent
            // over N equal keys.  Ragged batches spend a third of their rows here.
            const size_t ob = ((size_t)b * N + i) * K;
            for (int k = lane; k < K; k += 64) { idx_out[ob + k] = k; rank_out[ob + k] = 1e5f; }
            continue;
        }

#if EGNN_KNN_ADJ_FAST
        // ---- rows the adjacency alone decides (round 6).  With an adjacency matrix the ranking row is -1 for the node itself, 0 for
        // every adjacent node and a distance >= 0 (or 1e5) for the others (:255-256), and ties go by ascending index: a row with at
        // least K - 1 adjacent nodes is [i, its first K - 1 adjacent nodes in index order] with ranks -1, 0, 0, ... whatever the
        // coordinates are -- unless a NON-adjacent node sits at distance exactly 0.0 and ties with them, which a row rules out by
        // finding no candidate with the query's first coordinate (checked conservatively: any j != i; only rows of unmasked nodes can
        // have such a tie, a masked row's non-adjacent pairs are all 1e5).  The chain adjacency of BASELINE.json's c4 (K = 3,
        // N = 2048): every interior row, 2046 of 2048 -- no distance, no key, no threshold search for them.
        // Adjacent nodes are counted from whole-line loads of the row's 0 / 1 bytes with one wave scan for the positions -- no per-chunk
        // ballots, which run on the CU's one scalar unit.
#if defined(EGNN_KNN_ABL) && (EGNN_KNN_ABL & 1)
        if (adjrow) continue;                                             // timing-only ablation: prologue only
#endif
        if constexpr (CPL % 4 == 0 && CPL <= 32) {
            if (adjrow && (N & 15) == 0 && (reinterpret_cast<uintptr_t>(adjrow) & 15) == 0) {
                // the row's N 0 / 1 bytes in chunks of 1 KB: lane l holds bytes [16 l, 16 l + 16) of every chunk -- one fully coalesced
                // 16-byte load per lane and chunk (as dwords at a 32-byte lane stride the same bytes cost eight instructions of sixteen
                // partly used lines each: 59 of the kernel's 93 us at c4's shape)
                constexpr int NCH = (CPL * 64 + 1023) / 1024;            // chunks: 1 (N <= 1024) or 2
                typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                uint32_t nz[NCH][4];
                int cl[NCH];
#pragma unroll
                for (int h = 0; h < NCH; ++h) {
                    const int jb = 1024 * h + 16 * lane;                  // (N % 16 == 0: a lane's 16 bytes are inside the row or outside it)
                    u32x4 w4 = u32x4{0u, 0u, 0u, 0u};
                    if (jb < N) w4 = *reinterpret_cast<const u32x4*>(adjrow + jb);
                    cl[h] = 0;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        uint32_t w = w4[q];
                        w = (((w & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | w) & 0x80808080u;       // bit 8 t + 7: byte t is non-zero
                        const int d = i - (jb + 4 * q);
                        if ((unsigned)d < 4u) w &= ~(0x80u << (8 * d));                  // the diagonal is cleared (:252-254)
                        nz[h][q] = w;
                        cl[h] += __popc(w);
                    }
                }
                // both chunks' prefix sums in one scan (a lane has at most 16 adjacent nodes per chunk)
                const int packed = cl[0] | ((NCH > 1 ? cl[NCH - 1] : 0) << 16);
                const int incl = egnn_wave_inclusive_scan(packed);
                const int tot = __builtin_amdgcn_readlane(incl, 63);
                const int total0 = tot & 0xffff, total = total0 + (tot >> 16);
                bool decided = total >= K - 1;                            // wave-uniform
                if (decided && mi) decided = ((*dupflags >> r) & 1u) == 0u;      // (no other node with this row's first coordinate)
#if defined(EGNN_KNN_ABL) && (EGNN_KNN_ABL & 2)
                if (decided || total == 12345) continue;                   // timing-only ablation: no emission, no general rows
#endif
                if (decided) {
                    const size_t ob = ((size_t)b * N + i) * K;
                    if (lane == 0) { idx_out[ob] = i; rank_out[ob] = -1.0f; }
#pragma unroll
                    for (int h = 0; h < NCH; ++h) {
                        int pos = h == 0 ? (incl & 0xffff) - cl[0] : total0 + (incl >> 16) - cl[NCH - 1];
                        const int jb = 1024 * h + 16 * lane;
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            uint32_t w = nz[h][q];
                            while (w && pos < K - 1) {
                                const int t = __builtin_ctz(w) >> 3;
                                w &= w - 1;
                                idx_out[ob + 1 + pos] = jb + 4 * q + t;
                                rank_out[ob + 1 + pos] = 0.0f;
                                ++pos;
                            }
                        }
                    }
                    continue;
                }
            }
        }
#endif
        uint32_t key[CPL];
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
            const int j = c * 64 + lane;
            uint32_t k = 0xFFFFFFFFu;            // padding candidates sort last
            if (j < N) {
                float rk;
                if (CDM == 3) {
                    float dx, dy, dz;
                    rk = egnn_sqdist(ci[0], ci[1], ci[2], xs[j], xs[Npad + j], xs[2 * Npad + j], dx, dy, dz);
                } else {
                    float cj[CDM], rel[CDM];
#pragma unroll
                    for (int c = 0; c < CDM; ++c) cj[c] = c < C ? xs[c * Npad + j] : 0.f;
                    rk = egnn_sqdist_n<CDM>(ci, cj, C, rel);
                }
                if (!(mi && ms[j] != 0)) rk = 1e5f;                 // :240-242
                if (adjrow) {
                    if (j == i) rk = -1.0f;                         // :255
                    else if (adjrow[j]) rk = 0.0f;                  // :256
                }
                k = f2key(rk);
            }
            key[c] = k;
        }

        const size_t obase = ((size_t)b * N + i) * K;

        // ---- fast path (K <= 64): prune with the lane minima.  Let M be the K-th smallest of the 64 per-lane minima (for
        // K > 32: of the 128 smallest-two-per-lane keys): at least K candidates are <= M, so the K smallest all are.  Typically only ~1.3 K candidates
        // survive (N = 1024, K = 32: ~43); if there are at most two per lane they are ranked by counting directly.
        bool done = false;
        if (K <= 64) {
            uint32_t lmin = key[0], lmin2 = 0xFFFFFFFFu;
            const bool two = K > 32;                          // wave-uniform: K > 32 prunes with the TWO smallest keys per lane
            if (two) {
#pragma unroll
                for (int c = 1; c < CPL; ++c) {
                    const uint32_t kc = key[c];
                    lmin2 = kc < lmin ? lmin : (kc < lmin2 ? kc : lmin2);
                    lmin = kc < lmin ? kc : lmin;
                }
            } else {
#pragma unroll
                for (int c = 1; c < CPL; ++c) lmin = key[c] < lmin ? key[c] : lmin;
            }
            // The threshold only has to keep >= K candidates and <= 64 survivors, so 16 key bits (sign, exponent, 7
            // mantissa bits: 0.8 % granularity) are enough -- half the serial ballot/popcount chain, which runs on the
            // CU's single scalar unit and was what bound this kernel.
            uint32_t M = 0;
            int belowm = 0;
            for (int bit = 31; bit >= 32 - KNN_PREFIX_BITS; --bit) {
                int cnt = __popcll(__ballot((lmin >> bit) == (M >> bit)));
                if (two) cnt += __popcll(__ballot((lmin2 >> bit) == (M >> bit)));
                if (belowm + cnt < K) {
                    belowm += cnt;
                    M |= (1u << bit);
                }
            }
            M |= (1u << (32 - KNN_PREFIX_BITS)) - 1u;
            // survivors per lane, exclusive prefix over the wave (vector ops only: no ballots)
            int cl = 0;
#pragma unroll
            for (int c = 0; c < CPL; ++c) cl += key[c] <= M ? 1 : 0;
            const int incl = egnn_wave_inclusive_scan(cl);
            const int S = __builtin_amdgcn_readlane(incl, 63);
            if (S <= KNN_SURVIVORS) {                        // wave-uniform; up to two survivors per lane
                int pos = incl - cl;
#pragma unroll
                for (int c = 0; c < CPL; ++c)
                    if (key[c] <= M) selbuf[pos++] = ((uint64_t)key[c] << 32) | (uint32_t)(c * 64 + lane);
                wave_lds_sync();
                const uint64_t mine0 = lane < S ? selbuf[lane] : ~0ull;
                const uint64_t mine1 = lane + 64 < S ? selbuf[lane + 64] : ~0ull;
                int rnk0 = 0, rnk1 = 0;
                if (S <= 64) {                                // (wave-uniform) the common case: one survivor per lane
                    for (int u = 0; u < S; ++u) rnk0 += (selbuf[u] < mine0) ? 1 : 0;
                } else {
                    for (int u = 0; u < S; ++u) {
                        const uint64_t o = selbuf[u];
                        rnk0 += (o < mine0) ? 1 : 0;
                        rnk1 += (o < mine1) ? 1 : 0;
                    }
                }
                if (lane < S && rnk0 < K) {
                    idx_out[obase + rnk0] = (int32_t)(uint32_t)(mine0 & 0xFFFFFFFFull);
                    rank_out[obase + rnk0] = key2f((uint32_t)(mine0 >> 32));
                }
                if (lane + 64 < S && rnk1 < K) {
                    idx_out[obase + rnk1] = (int32_t)(uint32_t)(mine1 & 0xFFFFFFFFull);
                    rank_out[obase + rnk1] = key2f((uint32_t)(mine1 >> 32));
                }
                wave_lds_sync();
                done = true;
            }
        }
        if (done) continue;

        // ---- general path: exact K-th smallest key by bitwise radix descent, all control flow wave-uniform
        uint32_t T = 0;
        int below = 0;
        for (int bit = 31; bit >= 0; --bit) {
            const uint32_t want = T >> bit;
            int cnt = 0;
#pragma unroll
            for (int c = 0; c < CPL; ++c)
                cnt += __popcll(__ballot((key[c] >> bit) == want));
            if (below + cnt < K) {
                below += cnt;
                T |= (1u << bit);
            }
        }

        // ---- pick: everything below T, then the lowest-index `need` candidates equal to T
        int need = K - below;
        int base = 0;
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
            const bool less = key[c] < T;
            const bool eq = key[c] == T;
            const uint64_t beq = __ballot(eq);
            const int ceq = __popcll(beq);
            const bool take_eq = eq && (__popcll(beq & lt_mask) < need);
            need -= (need < ceq ? need : ceq);
            const bool sel = less || take_eq;
            const uint64_t bs = __ballot(sel);
            if (sel)
                selbuf[base + __popcll(bs & lt_mask)] =
                    ((uint64_t)key[c] << 32) | (uint32_t)(c * 64 + lane);
            base += __popcll(bs);
        }
        wave_lds_sync();

        // ---- sort the K survivors by (value, index): rank by counting
        for (int t = lane; t < K; t += 64) {
            const uint64_t mine = selbuf[t];
            int rnk = 0;
            for (int u = 0; u < K; ++u) rnk += (selbuf[u] < mine) ? 1 : 0;
            idx_out[obase + rnk] = (int32_t)(uint32_t)(mine & 0xFFFFFFFFull);
            rank_out[obase + rnk] = key2f((uint32_t)(mine >> 32));
        }
        wave_lds_sync();
    }
}

// ---- large graphs (N beyond what a wave keeps in registers: 8192 nodes with 3-D coordinates, 4096 otherwise).  One WORKGROUP per
// query row: the row's N ranking keys (the same bit-exact values) live in LDS -- up to 32 768 -- instead of registers; the K-th
// smallest key by the same 32-step radix descent (per-thread counts over a contiguous slice, wave sums, one cross-wave sum per step),
// the index-ordered pick among the ties with the K-th value through a block-wide exclusive scan, rank-by-counting on (value, index).
// Coordinates are read from global memory (a graph's 12 N bytes are L2-resident).  Deterministic, same tie policy; ~10 us per row,
// rows spread over the chip: a rarely used path (the reference's topk has no size limit, egnn_pytorch.py:258), not a fast one.
template <int CDM>
__global__ __launch_bounds__(KNN_THREADS) void knn_select_large_kernel(
    const float* __restrict__ coors, const uint8_t* __restrict__ mask, const uint8_t* __restrict__ adj,
    int64_t adj_bstride, int N, int K, int Cdim, int32_t* __restrict__ idx_out, float* __restrict__ rank_out)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint32_t* keys = reinterpret_cast<uint32_t*>(smem);                  // [N]
    uint64_t* sel = reinterpret_cast<uint64_t*>(smem + ((size_t)N * 4 + 7) / 8 * 8);     // [K]
    int* red = reinterpret_cast<int*>(sel + K);                          // [2 * KNN_WAVES + 2]
    const int C = (CDM == 3) ? 3 : Cdim;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.y, i = blockIdx.x;
    const float* cb = coors + (size_t)b * N * C;
    const uint8_t* mb = mask ? mask + (size_t)b * N : nullptr;
    const bool mi = mb ? mb[i] != 0 : true;
    const uint8_t* adjrow = adj ? adj + (size_t)b * adj_bstride + (size_t)i * N : nullptr;
    const size_t obase = ((size_t)b * N + i) * K;
    if (!mi && !adjrow) {                                                // a masked row: all keys 1e5, the first K indices (see above)
        for (int k = tid; k < K; k += KNN_THREADS) { idx_out[obase + k] = k; rank_out[obase + k] = 1e5f; }
        return;
    }
    float ci[CDM];
#pragma unroll
    for (int c = 0; c < CDM; ++c) ci[c] = c < C ? cb[(size_t)i * C + c] : 0.f;
    for (int j = tid; j < N; j += KNN_THREADS) {
        float rk;
        if (CDM == 3) {
            float dx, dy, dz;
            rk = egnn_sqdist(ci[0], ci[1], ci[2], cb[(size_t)j * 3], cb[(size_t)j * 3 + 1], cb[(size_t)j * 3 + 2], dx, dy, dz);
        } else {
            float cj[CDM], rel[CDM];
#pragma unroll
            for (int c = 0; c < CDM; ++c) cj[c] = c < C ? cb[(size_t)j * C + c] : 0.f;
            rk = egnn_sqdist_n<CDM>(ci, cj, C, rel);
        }
        if (!(mi && (mb ? mb[j] != 0 : true))) rk = 1e5f;                // :240-242
        if (adjrow) {
            if (j == i) rk = -1.0f;                                      // :255
            else if (adjrow[j]) rk = 0.0f;                               // :256
        }
        keys[j] = f2key(rk);
    }
    __syncthreads();

    // block-wide sum of a per-thread count (every thread gets it)
    auto block_sum = [&](int v, int slot) {
        v = egnn_wave_sum(v);
        if (lane == 0) red[slot * KNN_WAVES + wave] = v;
        __syncthreads();
        int t = 0;
#pragma unroll
        for (int w = 0; w < KNN_WAVES; ++w) t += red[slot * KNN_WAVES + w];
        return t;
    };
    // contiguous slice of candidate indices per thread (index order = thread order: the tie pick below relies on it)
    const int per = (N + KNN_THREADS - 1) / KNN_THREADS;
    const int j0 = tid * per, j1 = (j0 + per) < N ? (j0 + per) : N;

    // ---- exact K-th smallest key T by bitwise radix descent
    uint32_t T = 0;
    int below = 0;
    for (int bit = 31; bit >= 0; --bit) {
        const uint32_t want = T >> bit;
        int cnt = 0;
        for (int j = j0; j < j1; ++j) cnt += (keys[j] >> bit) == want ? 1 : 0;
        cnt = block_sum(cnt, bit & 1);                                   // (alternating slots: one barrier per step)
        if (below + cnt < K) {
            below += cnt;
            T |= (1u << bit);
        }
    }
    // ---- pick: everything below T, then the lowest-index `need` candidates equal to T
    const int need = K - below;
    int nless = 0, neq = 0;
    for (int j = j0; j < j1; ++j) {
        nless += keys[j] < T ? 1 : 0;
        neq += keys[j] == T ? 1 : 0;
    }
    // exclusive scans over the threads (wave scan + cross-wave offsets)
    __syncthreads();
    const int il = egnn_wave_inclusive_scan(nless), ie = egnn_wave_inclusive_scan(neq);
    if (lane == 63) { red[wave] = il; red[KNN_WAVES + wave] = ie; }
    __syncthreads();
    int offl = il - nless, offe = ie - neq;
    for (int w = 0; w < wave; ++w) { offl += red[w]; offe += red[KNN_WAVES + w]; }
    // selected entries: the `below` smaller ones first (any order), then the ties in index order
    int pl = offl, pe = offe;
    for (int j = j0; j < j1; ++j) {
        const uint32_t kj = keys[j];
        if (kj < T) sel[pl++] = ((uint64_t)kj << 32) | (uint32_t)j;
        else if (kj == T) {
            if (pe < need) sel[below + pe] = ((uint64_t)kj << 32) | (uint32_t)j;
            ++pe;
        }
    }
    __syncthreads();
    // ---- sort the K selected by (value, index): rank by counting
    for (int t = tid; t < K; t += KNN_THREADS) {
        const uint64_t mine = sel[t];
        int rnk = 0;
        for (int u = 0; u < K; ++u) rnk += (sel[u] < mine) ? 1 : 0;
        idx_out[obase + rnk] = (int32_t)(uint32_t)(mine & 0xFFFFFFFFull);
        rank_out[obase + rnk] = key2f((uint32_t)(mine >> 32));
    }
}

template <int CDM>
int launch_knn_large(const float* coors, const uint8_t* mask, const uint8_t* adj, int64_t adj_bstride, int B, int N,
                     int K, int C, int32_t* idx_out, float* rank_out, hipStream_t s)
{
    const size_t lds = ((size_t)N * 4 + 7) / 8 * 8 + (size_t)K * 8 + (2 * KNN_WAVES + 2) * sizeof(int);
    if (lds > 160 * 1024) return EGNN_E_UNSUPPORTED;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(knn_select_large_kernel<CDM>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL((knn_select_large_kernel<CDM>), dim3(N, B), dim3(KNN_THREADS), lds, s, coors, mask, adj, adj_bstride, N, K, C,
                       idx_out, rank_out);
    return egnn_launch_status();
}

__global__ __launch_bounds__(256) void adj_max_degree_kernel(const uint8_t* __restrict__ adj, int64_t rows,
                                                              int N, int32_t* __restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave_global = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    int best = 0;
    for (int64_t r = wave_global; r < rows; r += nwaves) {
        const uint8_t* row = adj + r * N;
        int cnt = 0;
        for (int j = lane; j < N; j += 64) cnt += row[j] ? 1 : 0;
        cnt = egnn_wave_sum(cnt);
        best = cnt > best ? cnt : best;
    }
    if (lane == 0 && best > 0) atomicMax(out, best);
}

template <int CPL, int CDM>
int launch_knn_c(const float* coors, const uint8_t* mask, const uint8_t* adj, int64_t adj_bstride, int B, int N,
                 int K, int C, int32_t* idx_out, float* rank_out, hipStream_t s)
{
    const int Npad = (N + 63) / 64 * 64;
    const int Kpad = K > KNN_SURVIVORS ? (K + 1) / 2 * 2 : KNN_SURVIVORS;      // the fast path parks up to 128 survivors
#ifndef EGNN_KNN_ROWS_PER_WG
#define EGNN_KNN_ROWS_PER_WG 32
#endif
    int rows_per_wg = EGNN_KNN_ROWS_PER_WG;
    if (N < rows_per_wg) rows_per_wg = (N + 3) / 4 * 4;
    const size_t cbytes = (size_t)Npad * (4 * C + 1);
    const size_t coord_bytes = cbytes + 8 - cbytes % 8;
    const size_t lds = coord_bytes + (size_t)KNN_WAVES * Kpad * 8 + 8;          // (+ the duplicate-coordinate flags of the adjacency path)
    if (lds > 160 * 1024) return EGNN_E_UNSUPPORTED;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(knn_select_kernel<CPL, CDM>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    dim3 grid((N + rows_per_wg - 1) / rows_per_wg, B);
    hipLaunchKernelGGL((knn_select_kernel<CPL, CDM>), grid, dim3(KNN_THREADS), lds, s, coors, mask, adj, adj_bstride, N, K,
                       Npad, Kpad, rows_per_wg, C, idx_out, rank_out);
    return egnn_launch_status();
}

template <int CPL>
int launch_knn(const float* coors, const uint8_t* mask, const uint8_t* adj, int64_t adj_bstride, int B, int N,
               int K, int C, int32_t* idx_out, float* rank_out, hipStream_t s)
{
    if (C == 3) return launch_knn_c<CPL, 3>(coors, mask, adj, adj_bstride, B, N, K, C, idx_out, rank_out, s);
    return launch_knn_c<CPL, 8>(coors, mask, adj, adj_bstride, B, N, K, C, idx_out, rank_out, s);
}

}  // namespace

// internal (fp64.hip: the one-workgroup-per-row kernel for any coordinate dimension)
int egnn_knn_select_any_f32(const float* coors, const uint8_t* mask, const uint8_t* adj, int64_t adj_batch_stride, int B, int N, int K,
                            int coor_dim, int32_t* idx_out, float* rank_out, void* stream);

extern "C" int egnn_knn_select_f32(const float* coors, const uint8_t* mask, const uint8_t* adj,
                                   int64_t adj_batch_stride, int B, int N, int K, int coor_dim, int32_t* idx_out,
                                   float* rank_out, void* stream)
{
    if (!coors || !idx_out || !rank_out) return EGNN_E_NULLPTR;
    if (B <= 0 || N <= 0 || K <= 0) return EGNN_E_SHAPE;
    if (coor_dim < 1 || coor_dim > 64) return EGNN_E_UNSUPPORTED;
    // more than 8 coordinates: one workgroup per row, coordinates read from memory, the reference's summation tree for any length
    if (coor_dim > 8) return egnn_knn_select_any_f32(coors, mask, adj, adj_batch_stride, B, N, K, coor_dim, idx_out, rank_out, stream);
    const int C = coor_dim;
    if (K > N) return EGNN_E_K_GT_N;
    if (K > 1024 || B > 65535) return EGNN_E_UNSUPPORTED;
    hipStream_t s = static_cast<hipStream_t>(stream);
    // candidate keys live in registers (ceil(N / 64) per lane) and the graph's coordinates in LDS: N <= 8192 for 3-D coordinates
    // (128 keys per lane, 104 KB), N <= 4096 otherwise; larger graphs: one workgroup per row with the keys in LDS, up to 32 768 nodes
    if (N > (C == 3 ? 8192 : 4096)) {
        if (N > 32768) return EGNN_E_UNSUPPORTED;
        return C == 3 ? launch_knn_large<3>(coors, mask, adj, adj_batch_stride, B, N, K, C, idx_out, rank_out, s)
                      : launch_knn_large<8>(coors, mask, adj, adj_batch_stride, B, N, K, C, idx_out, rank_out, s);
    }
    if (N <= 64) return launch_knn<1>(coors, mask, adj, adj_batch_stride, B, N, K, C, idx_out, rank_out, s);
    if (N <= 128) return launch_knn<2>(coors, mask, adj, adj_batch_stride, B, N, K, C, idx_out, rank_out, s);
    if (N <= 256) return launch_knn<4>(coors, mask, adj, adj_batch_stride, B, N, K, C, idx_out, rank_out, s);
    if (N <= 512) return launch_knn<8>(coors, mask, adj, adj_batch_stride, B, N, K, C, idx_out, rank_out, s);
    if (N <= 1024) return launch_knn<16>(coors, mask, adj, adj_batch_stride, B, N, K, C, idx_out, rank_out, s);
    if (N <= 2048) return launch_knn<32>(coors, mask, adj, adj_batch_stride, B, N, K, C, idx_out, rank_out, s);
    if (N <= 4096) return launch_knn<64>(coors, mask, adj, adj_batch_stride, B, N, K, C, idx_out, rank_out, s);
    return launch_knn_c<128, 3>(coors, mask, adj, adj_batch_stride, B, N, K, C, idx_out, rank_out, s);
}

extern "C" int egnn_adj_max_degree_u8(const uint8_t* adj, int64_t rows, int N, int32_t* out_dev, void* stream)
{
    if (!adj || !out_dev) return EGNN_E_NULLPTR;
    if (rows <= 0 || N <= 0) return EGNN_E_SHAPE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipError_t e = hipMemsetAsync(out_dev, 0, sizeof(int32_t), s);
    if (e != hipSuccess) return (int)e;
    int64_t blocks = (rows + 3) / 4;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(adj_max_degree_kernel, dim3((unsigned)blocks), dim3(256), 0, s, adj, rows, N, out_dev);
    return egnn_launch_status();
}
