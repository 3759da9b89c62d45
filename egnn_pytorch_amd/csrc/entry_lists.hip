// The transposed neighbour list of the backward, built by ONE pair of launches instead of sort / bincount /
// cumsum / searchsorted passes (~1.5 ms of small ATen kernels per step at the north-star shape).
//
// d loss / d P_j[n] (autograd of the neighbour gather, egnn_pytorch.py:275) is a sum over the edges (i, k) that ARRIVE at n.  The
// backward walks them in a fixed order -- ascending edge id -- so that the sums are bit-reproducible (no float atomics).  What it
// needs is the edge ids sorted stably by destination, in two forms:
//     csr:  order[p] = edge id, p in [csr_seg[n], csr_seg[n+1])                       (egnn_rows_gather_sum_f32)
//     ent:  the same, every destination's entries padded with -1 to whole 16-entry tiles: tile range [tile_seg[n], tile_seg[n+1])
//           (the entry list of egnn_edge_bwd_pass_f32 with by_dest = 1; the list is padded with -1 to a multiple of 128 entries)
// A counting sort per graph: the destinations of one source row are distinct (top-k of distinct candidates; j = k on the dense
// path), so a wave that walks its rows in order can claim list positions with LDS counters without two lanes of one instruction
// ever meeting on a counter -- positions come out in ascending edge id with no sort.  One workgroup per graph; four waves own
// consecutive quarters of the rows, each with its own histogram (pass 1) turned into its own start offsets (pass 2).
#include "egnn_common.h"

namespace {

constexpr int EL_MAX_THREADS = 1024;     // up to 16 waves per graph, each with its own histogram (as many as fit in LDS)
#ifndef EGNN_EL_ROWS_AHEAD
#define EGNN_EL_ROWS_AHEAD 16
#endif
constexpr int EL_ROWS_AHEAD = EGNN_EL_ROWS_AHEAD;   // rows whose index loads are in flight together (round 5, with the parallel scan below: 0.18 -> 0.09 ms per call at the north-star shape)

// hist[w][j] = number of edges of wave w's rows that arrive at j;  then (pass 2) per destination j: deg, tiles, and the
// graph-local exclusive scans first[j] (entries) / tseg[j] (tiles); returns the graph's totals through LDS slots
__device__ __forceinline__ void hist_and_scan(const int32_t* __restrict__ idx, int b, int N, int K, int* hist, int* scan_e, int* scan_t,
                                              int* totals)
{
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int EL_THREADS = blockDim.x, EL_WAVES = blockDim.x >> 6;
    for (int o = tid; o < EL_WAVES * N; o += EL_THREADS) hist[o] = 0;
    __syncthreads();
    const int rows_per_wave = (N + EL_WAVES - 1) / EL_WAVES;
    const int r0 = wave * rows_per_wave, r1 = (r0 + rows_per_wave) < N ? (r0 + rows_per_wave) : N;
    // (K <= 64 -- one column per lane: four rows' indices in flight per step; the loop is bound by the latency of the index loads -- one
    // workgroup per graph -- and only the LDS counters have to be bumped in row order)
    if (idx && K <= 64) {
        for (int i = r0; i < r1; i += EL_ROWS_AHEAD) {
            int jv[EL_ROWS_AHEAD];
#pragma unroll
            for (int u = 0; u < EL_ROWS_AHEAD; ++u) jv[u] = (lane < K && i + u < r1) ? idx[((size_t)b * N + i + u) * K + lane] : -1;
#pragma unroll
            for (int u = 0; u < EL_ROWS_AHEAD; ++u)
                if (jv[u] >= 0) atomicAdd(&hist[wave * N + jv[u]], 1);
        }
    } else {
        for (int i = r0; i < r1; ++i)
            for (int k = lane; k < K; k += 64) {
                const int j = idx ? idx[((size_t)b * N + i) * K + k] : k;
                atomicAdd(&hist[wave * N + j], 1);
            }
    }
    __syncthreads();
    // per thread: a contiguous range of destinations; local sums, then a scan over the 256 threads
    const int per = (N + EL_THREADS - 1) / EL_THREADS;
    const int j0 = tid * per, j1 = (j0 + per) < N ? (j0 + per) : N;
    int se = 0, st = 0;
    for (int j = j0; j < j1; ++j) {
        int deg = 0;
        for (int w = 0; w < EL_WAVES; ++w) deg += hist[w * N + j];
        se += deg;
        st += (deg + 15) >> 4;
    }
    // exclusive scans over the workgroup's threads: inside the waves on the DPP network, the (up to 16) wave totals through LDS (round 5:
    // this was one thread walking all 1024 entries -- half of the two kernels' time)
    const int ie = egnn_wave_inclusive_scan(se), it = egnn_wave_inclusive_scan(st);
    int* wtot = totals + 2;                              // [2][16]
    if (lane == 63) { wtot[wave] = ie; wtot[16 + wave] = it; }
    __syncthreads();
    int be = 0, bt = 0;
    for (int w = 0; w < wave; ++w) { be += wtot[w]; bt += wtot[16 + w]; }
    scan_e[tid] = be + ie - se;
    scan_t[tid] = bt + it - st;
    if (tid == EL_THREADS - 1) { totals[0] = be + ie; totals[1] = bt + it; }
    __syncthreads();
}

__global__ __launch_bounds__(EL_MAX_THREADS) void dest_totals_kernel(const int32_t* __restrict__ idx, int N, int K, int64_t* __restrict__ tiles_per_graph)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int EL_THREADS = blockDim.x, EL_WAVES = blockDim.x >> 6;
    int* hist = reinterpret_cast<int*>(smem);
    int* scan_e = hist + EL_WAVES * N;
    int* scan_t = scan_e + EL_THREADS;
    int* totals = scan_t + EL_THREADS;
    hist_and_scan(idx, blockIdx.x, N, K, hist, scan_e, scan_t, totals);
    if (threadIdx.x == 0) tiles_per_graph[blockIdx.x] = totals[1];
}

__global__ __launch_bounds__(EL_MAX_THREADS) void dest_lists_kernel(const int32_t* __restrict__ idx, int B, int N, int K,
                                                                const int64_t* __restrict__ tiles_per_graph, int32_t* __restrict__ ent,
                                                                int64_t* __restrict__ tile_seg, int64_t* __restrict__ csr_order,
                                                                int64_t* __restrict__ csr_seg)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int EL_THREADS = blockDim.x, EL_WAVES = blockDim.x >> 6;
    int* hist = reinterpret_cast<int*>(smem);
    int* scan_e = hist + EL_WAVES * N;
    int* scan_t = scan_e + EL_THREADS;
    int* totals = scan_t + EL_THREADS;
    int* first = totals + 2 + 32;                    // [N] graph-local first entry of destination j in the csr list (behind the wave totals)
    const int b = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    hist_and_scan(idx, b, N, K, hist, scan_e, scan_t, totals);
    // tiles of the graphs in front of this one: one graph per thread, summed over the workgroup (integer: any order gives the same sum)
    int64_t tbase = 0;
    {
        int part = 0;
        for (int g = tid; g < b; g += EL_THREADS) part += (int)tiles_per_graph[g];
        part = egnn_wave_inclusive_scan(part);
        int* wtot = totals + 2;                              // (free again: hist_and_scan ended with a barrier behind its last use)
        if (lane == 63) wtot[wave] = part;
        __syncthreads();
        for (int w = 0; w < EL_WAVES; ++w) tbase += wtot[w];
    }
    const int64_t ebase = (int64_t)b * N * K;        // every graph has exactly N K edges
    // start offsets per (wave, destination): hist[w][j] <- position of wave w's first entry of j inside j's tiles (graph-local,
    // in entries), first[j] <- csr start; tile_seg / csr_seg of this graph's destinations
    const int per = (N + EL_THREADS - 1) / EL_THREADS;
    const int j0 = tid * per, j1 = (j0 + per) < N ? (j0 + per) : N;
    int ae = scan_e[tid], at = scan_t[tid];
    for (int j = j0; j < j1; ++j) {
        int run = 0;
        for (int w = 0; w < EL_WAVES; ++w) {
            const int h = hist[w * N + j];
            hist[w * N + j] = at * 16 + run;         // padded position (entries) of this wave's first entry of j
            run += h;
        }
        first[j] = ae - at * 16;                      // csr position = first[j] + padded position  (entries of j are contiguous in both)
        tile_seg[(int64_t)b * N + j] = tbase + at;
        csr_seg[(int64_t)b * N + j] = ebase + ae;
        ae += run;
        at += (run + 15) >> 4;
    }
    if (b == B - 1 && tid == EL_THREADS - 1) {       // (the last thread's range ends at N: its running sums are the graph totals)
        tile_seg[(int64_t)B * N] = tbase + totals[1];
        csr_seg[(int64_t)B * N] = ebase + totals[0];
    }
    __syncthreads();
    // placement: rows in ascending order within the wave, the waves' row ranges in ascending order -> ascending edge id per destination
    const int rows_per_wave = (N + EL_WAVES - 1) / EL_WAVES;
    const int r0 = wave * rows_per_wave, r1 = (r0 + rows_per_wave) < N ? (r0 + rows_per_wave) : N;
    if (idx && K <= 64) {
        // (as in the histogram pass: four rows' index loads in flight, the counters bumped in row order)
        for (int i = r0; i < r1; i += EL_ROWS_AHEAD) {
            int jv[EL_ROWS_AHEAD];
#pragma unroll
            for (int u = 0; u < EL_ROWS_AHEAD; ++u) jv[u] = (lane < K && i + u < r1) ? idx[ebase + (int64_t)(i + u) * K + lane] : -1;
#pragma unroll
            for (int u = 0; u < EL_ROWS_AHEAD; ++u) {
                if (jv[u] < 0) continue;
                const int64_t eid = ebase + (int64_t)(i + u) * K + lane;
                const int pos = atomicAdd(&hist[wave * N + jv[u]], 1);   // distinct destinations within a row: no two lanes share a counter
                ent[tbase * 16 + pos] = (int32_t)eid;
                csr_order[ebase + first[jv[u]] + pos] = eid;
            }
        }
    } else {
        for (int i = r0; i < r1; ++i)
            for (int k = lane; k < K; k += 64) {
                const int64_t eid = ebase + (int64_t)i * K + k;
                const int j = idx ? idx[eid] : k;
                const int pos = atomicAdd(&hist[wave * N + j], 1);       // distinct destinations within a row: no two lanes share a counter
                ent[tbase * 16 + pos] = (int32_t)eid;
                csr_order[ebase + first[j] + pos] = eid;
            }
    }
}

}  // namespace

extern "C" size_t egnn_dest_lists_capacity(int B, int N, int K)
{
    // upper bound on the padded list: every destination wastes less than one tile; rounded up to whole rounds of 128 entries
    const int64_t tiles = (int64_t)B * ((int64_t)N * K / 16 + N) + 8;
    return (size_t)((tiles * 16 + 127) / 128 * 128);
}

extern "C" int egnn_dest_lists_i32(const int32_t* idx, int B, int N, int K, int32_t* ent, size_t ent_capacity, int64_t* tile_seg,
                                   int64_t* csr_order, int64_t* csr_seg, int64_t* tiles_per_graph, void* stream)
{
    if (!ent || !tile_seg || !csr_order || !csr_seg || !tiles_per_graph) return EGNN_E_NULLPTR;
    if (B <= 0 || N <= 0 || K <= 0) return EGNN_E_SHAPE;
    if (!idx && K != N) return EGNN_E_SHAPE;
    if ((int64_t)B * N * K > 0x7fffffffLL) return EGNN_E_UNSUPPORTED;       // edge ids are int32 in the entry list
    if (ent_capacity < egnn_dest_lists_capacity(B, N, K)) return EGNN_E_SHAPE;
    // as many waves per graph as histograms fit in LDS (16, 8 or 4): one workgroup per graph is all the parallelism there is
    // (large graphs -- N beyond ~8000 -- trade waves for histogram space: 2 waves up to N = 13 500, 1 wave up to N = 20 400; the
    // forward's k-NN select stops at 8192 nodes per graph, dense graphs reach the int32 edge-id limit long before)
    int waves = 16;
    auto lds_for = [&](int w) { return ((size_t)w * N + 2 * (size_t)w * 64 + 2 + 32 + N) * sizeof(int); };
    while (waves > 4 && lds_for(waves) > 96 * 1024) waves >>= 1;
    while (waves > 1 && lds_for(waves) > 160 * 1024) waves >>= 1;
    const size_t lds = lds_for(waves);
    const int EL_THREADS = waves * 64;
    if (lds > 160 * 1024) return EGNN_E_UNSUPPORTED;                        // N <= 20 415 destinations per graph
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(dest_totals_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(dest_lists_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    if (hipMemsetAsync(ent, 0xFF, ent_capacity * sizeof(int32_t), s) != hipSuccess) return (int)hipGetLastError();     // -1 = padding
    hipLaunchKernelGGL(dest_totals_kernel, dim3(B), dim3(EL_THREADS), lds, s, idx, N, K, tiles_per_graph);
    hipLaunchKernelGGL(dest_lists_kernel, dim3(B), dim3(EL_THREADS), lds, s, idx, B, N, K, tiles_per_graph, ent, tile_seg, csr_order, csr_seg);
    return egnn_launch_status();
}
