// Micro-benchmark 4: throughput of the P_j row gather (random rows of 8320 B, walked 128 B per step) as a function of
// the per-instruction access shape and of the number of loads in flight.  Table = 64 graphs x 1024 rows (545 MB).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/gather tools/ubench/gather.hip && /tmp/gather
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <random>
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int ROWF = 2080, NROW = 1024, NGRAPH = 64, STEPS = 65;

// PAT 0: lane (e=l&15, g=l>>4): two loads at byte 32g and 32g+16 of row[e]   (the edge kernel today)
// PAT 1: lane (e, g): two loads at 16g and 64+16g                             (4 lanes = 64 contiguous bytes)
// PAT 2: lane (r=l>>3, c=l&7): ONE load of row[r] chunk c, rows 0..7, then rows 8..15  (8 full lines / instr)
// PAT 3: all 16 edges read the same row (L1 broadcast; the ablation)
template <int PAT, int DEPTH>
__global__ __launch_bounds__(256) void gather_k(const float* __restrict__ tab, const int* __restrict__ nbr, float* out, int groups_per_graph)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int bid = blockIdx.x;
    const int graph = (bid & 7) * (gridDim.x / 8 / groups_per_graph) + (bid >> 3) / groups_per_graph;   // XCD-contiguous graphs
    const int grp = (bid >> 3) % groups_per_graph;
    const int e = lane & 15, g = lane >> 4;
    const float* base = tab + (size_t)graph * NROW * ROWF;
    const int* nb = nbr + ((size_t)(graph * groups_per_graph + grp) * 4 + wave) * 32;
    const float* p[2][2];
    for (int t = 0; t < 2; ++t) {
        if (PAT == 0) { const float* r = base + (size_t)nb[t * 16 + e] * ROWF; p[t][0] = r + 8 * g; p[t][1] = r + 8 * g + 4; }
        if (PAT == 1) { const float* r = base + (size_t)nb[t * 16 + e] * ROWF; p[t][0] = r + 4 * g; p[t][1] = r + 16 + 4 * g; }
        if (PAT == 2) { p[t][0] = base + (size_t)nb[t * 16 + (lane >> 3)] * ROWF + 4 * (lane & 7);
                        p[t][1] = base + (size_t)nb[t * 16 + 8 + (lane >> 3)] * ROWF + 4 * (lane & 7); }
        if (PAT == 3) { const float* r = base + (size_t)nb[t * 16] * ROWF; p[t][0] = r + 8 * g; p[t][1] = r + 8 * g + 4; }
    }
    f32x4 acc = {0, 0, 0, 0};
    f32x4 buf[DEPTH][4];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
#pragma unroll
        for (int q = 0; q < 4; ++q) buf[d][q] = *reinterpret_cast<const f32x4*>(p[q >> 1][q & 1] + d * 32);
    for (int s = 0; s < STEPS; s += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
#pragma unroll
            for (int q = 0; q < 4; ++q) acc += buf[d][q];
            int nxt = s + d + DEPTH; if (nxt >= STEPS) nxt = STEPS - 1;
#pragma unroll
            for (int q = 0; q < 4; ++q) buf[d][q] = *reinterpret_cast<const f32x4*>(p[q >> 1][q & 1] + nxt * 32);
            // stand-in for the per-step compute (keeps the loads one "step" apart in time)
            for (int w = 0; w < 24; ++w) acc = acc * 1.0001f + 0.5f;
        }
    }
    out[(size_t)bid * 256 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}

template <int PAT, int DEPTH>
void run(const float* tab, const int* nbr, float* out, const char* name)
{
    const int gpg = 256;                       // 1024 nodes / 4 nodes per 256-thread WG
    const int blocks = NGRAPH * gpg;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((gather_k<PAT, DEPTH>), dim3(blocks), dim3(256), 0, 0, tab, nbr, out, gpg);
    (void)hipDeviceSynchronize();
    float best = 1e9;
    for (int r = 0; r < 3; ++r) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((gather_k<PAT, DEPTH>), dim3(blocks), dim3(256), 0, 0, tab, nbr, out, gpg);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
    }
    const double bytes = (double)blocks * 4 * 32 * STEPS * 128;
    printf("%-46s depth %d: %7.3f ms  %6.2f TB/s useful\n", name, DEPTH, best, bytes / best / 1e9);
}

int main()
{
    float* tab; int* nbr; float* out;
    const size_t tabn = (size_t)NGRAPH * NROW * ROWF;
    (void)hipMalloc(&tab, tabn * 4); (void)hipMemset(tab, 0, tabn * 4);
    std::vector<int> h((size_t)NGRAPH * 256 * 4 * 32);
    std::mt19937 rng(1);
    for (auto& v : h) v = rng() % NROW;
    (void)hipMalloc(&nbr, h.size() * 4); (void)hipMemcpy(nbr, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    (void)hipMalloc(&out, (size_t)NGRAPH * 256 * 256 * 4);
    run<0, 1>(tab, nbr, out, "PAT0 16 rows x {32g, 32g+16} (today)");
    run<0, 2>(tab, nbr, out, "PAT0");
    run<0, 4>(tab, nbr, out, "PAT0");
    run<1, 1>(tab, nbr, out, "PAT1 16 rows x 64 contiguous bytes per instr");
    run<1, 2>(tab, nbr, out, "PAT1");
    run<1, 4>(tab, nbr, out, "PAT1");
    run<2, 1>(tab, nbr, out, "PAT2 8 rows x full 128-byte line per instr");
    run<2, 2>(tab, nbr, out, "PAT2");
    run<2, 4>(tab, nbr, out, "PAT2");
    run<3, 1>(tab, nbr, out, "PAT3 one row per tile (no divergence)");
    run<3, 2>(tab, nbr, out, "PAT3");
    return 0;
}
