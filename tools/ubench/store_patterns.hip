// Micro-benchmark: HBM store throughput of the backward edge kernel's dz / a streams (two E x Hp fp32 arrays, E = 2^20 rows of
// Hp = 2080 floats, walked 32 columns per step by workgroups that own 256 consecutive rows) as a function of the per-instruction
// store shape and of the array layout.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/store_patterns tools/ubench/store_patterns.hip && /tmp/store_patterns
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int HP = 2080, STEPS = HP / 32;
constexpr long E = 1 << 20;

// PAT 0: today: lane (e = l & 15, g = l >> 4) stores 16 B at row e of its tile, column 32 st + 16 hb + 4 g (64-byte half lines)
// PAT 1: row-major, full lines: lane (r = l >> 3, c = l & 7) stores chunk c of the step's 128-byte line of row 8 q + r
// PAT 2: step-blocked layout [step][E][32], today's lanes
// PAT 3: step-blocked layout, full lines (1 KB contiguous per instruction)
// PAT 4: plain streaming fill of the same bytes (upper bound)
// INFL > 0: at most INFL stores of a wave in flight (s_waitcnt vmcnt(INFL - 8) before each step's 8 stores) at 4 workgroups per
// CU (36 KB of LDS each) -- the edge kernel's situation, where the step's gathers sit behind its stores on the in-order counter
template <int PAT, int INFL = 0>
__global__ __launch_bounds__(256) void k(float* __restrict__ a, float* __restrict__ d, float seed)
{
    extern __shared__ char smem_[];
    if (INFL > 0 && seed < 0.f) smem_[threadIdx.x] = 1;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int e = lane & 15, g = lane >> 4, r = lane >> 3, c = lane & 7;
    const long row0 = (long)blockIdx.x * 128 + wave * 32;          // 128 rows per workgroup round, 2 rounds
    for (int round = 0; round < 2; ++round) {
        const long rw = row0 + (long)round * (E / 2);
        for (int st = 0; st < STEPS; ++st) {
            f32x4 v = {seed + st, seed + lane, seed, seed * 2.f};
            if (INFL == 8) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (INFL == 16) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            if (INFL == 24) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            if (INFL == 32) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
            if (INFL == 48) asm volatile("s_waitcnt vmcnt(40)" ::: "memory");
            if (INFL == 63) asm volatile("s_waitcnt vmcnt(55)" ::: "memory");
            if (PAT == 0 || PAT == 2) {
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int hb = 0; hb < 2; ++hb) {
                        const long row = rw + 16 * t + e;
                        const size_t o = PAT == 0 ? (size_t)row * HP + 32 * st + 16 * hb + 4 * g
                                                  : ((size_t)st * E + row) * 32 + 16 * hb + 4 * g;
                        *reinterpret_cast<f32x4*>(a + o) = v;
                        *reinterpret_cast<f32x4*>(d + o) = v;
                    }
            } else if (PAT == 1 || PAT == 3) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const long row = rw + 8 * q + r;
                    const size_t o = PAT == 1 ? (size_t)row * HP + 32 * st + 4 * c : ((size_t)st * E + row) * 32 + 4 * c;
                    *reinterpret_cast<f32x4*>(a + o) = v;
                    *reinterpret_cast<f32x4*>(d + o) = v;
                }
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const size_t o = (((size_t)blockIdx.x * 2 + round) * STEPS + st) * 4096 + (wave * 4 + q) * 256 + lane * 4;
                    *reinterpret_cast<f32x4*>(a + o) = v;
                    *reinterpret_cast<f32x4*>(d + o) = v;
                }
            }
        }
    }
}

template <int PAT, int INFL = 0>
void run(const char* name, float* a, float* d)
{
    hipEvent_t t0, t1; (void)hipEventCreate(&t0); (void)hipEventCreate(&t1);
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        (void)hipEventRecord(t0);
        hipLaunchKernelGGL((k<PAT, INFL>), dim3(E / 256), dim3(256), INFL > 0 ? 36 * 1024 : 0, 0, a, d, 1.0f + rep);
        (void)hipEventRecord(t1); (void)hipEventSynchronize(t1);
        float ms; (void)hipEventElapsedTime(&ms, t0, t1);
        if (rep && ms < best) best = ms;
    }
    const double gb = 2.0 * E * HP * 4 / 1e9;
    printf("%-64s %7.3f ms  %6.2f TB/s\n", name, best, gb / best);
}

int main()
{
    float *a, *d;
    (void)hipMalloc(&a, (size_t)E * HP * 4); (void)hipMalloc(&d, (size_t)E * HP * 4);
    run<4>("streaming fill (1 KB contiguous per instruction)", a, d);
    run<0>("row-major, 64-byte half lines (today)", a, d);
    run<1>("row-major, full 128-byte lines", a, d);
    run<2>("step-blocked [step][E][32], half lines", a, d);
    run<3>("step-blocked, full lines (1 KB contiguous per instruction)", a, d);
    run<0>("row-major, 64-byte half lines (today), again", a, d);
    run<0, 8>("  16 waves / CU, <=  8 stores of a wave in flight", a, d);
    run<0, 16>("  16 waves / CU, <= 16", a, d);
    run<0, 24>("  16 waves / CU, <= 24", a, d);
    run<0, 32>("  16 waves / CU, <= 32", a, d);
    run<0, 48>("  16 waves / CU, <= 48", a, d);
    run<0, 63>("  16 waves / CU, <= 63", a, d);
    run<1, 16>("  full lines, 16 waves / CU, <= 16", a, d);
    run<3, 16>("  step-blocked full lines, 16 waves / CU, <= 16", a, d);
    return 0;
}
