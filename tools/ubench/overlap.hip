// Micro-benchmark 2: how many VALU / transcendental instructions hide beside one v_mfma_f32_16x16x4_f32
// (a) inside ONE wave (interleaved in program order, no data dependence between the two streams),
// (b) across TWO waves of one SIMD (one MFMA-only wave + one VALU-only wave, 512-thread workgroup).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/overlap tools/ubench/overlap.hip && /tmp/overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define ITERS 2048

template <int KP, int KT>
__device__ __forceinline__ void body(f32x4 (&acc)[4], float (&v)[16], float (&w)[4], float a, float b)
{
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[c], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < KP; ++i) v[i] = __builtin_fmaf(v[i], a, b);
#pragma unroll
        for (int i = 0; i < KT; ++i) w[i] = __builtin_amdgcn_exp2f(w[i]);
        __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x2, KP + KT, 0);
    }
}

template <int KP, int KT, bool MFMA>
__global__ __launch_bounds__(256) void k_one(float* out, float seed)
{
    f32x4 acc[4];
    float v[16], w[4];
    for (int c = 0; c < 4; ++c) acc[c] = f32x4{seed, seed, seed, seed};
    for (int i = 0; i < 16; ++i) v[i] = seed + i + threadIdx.x;
    for (int i = 0; i < 4; ++i) w[i] = seed * 0.1f;
    const float a = seed * 0.5f, b = seed * 0.25f;
    for (int it = 0; it < ITERS; ++it) {
        if (MFMA) body<KP, KT>(acc, v, w, a, b);
        else {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
#pragma unroll
                for (int i = 0; i < KP; ++i) v[i] = __builtin_fmaf(v[i], a, b);
#pragma unroll
                for (int i = 0; i < KT; ++i) w[i] = __builtin_amdgcn_exp2f(w[i]);
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += v[i];
    for (int i = 0; i < 4; ++i) s += w[i] + acc[i][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// two waves per SIMD: waves 0-3 MFMA-only, waves 4-7 VALU-only (KP fma + KT exp per "slot")
template <int KP, int KT>
__global__ __launch_bounds__(512) void k_two(float* out, float seed, int mode)
{
    f32x4 acc[4];
    float v[16], w[4];
    for (int c = 0; c < 4; ++c) acc[c] = f32x4{seed, seed, seed, seed};
    for (int i = 0; i < 16; ++i) v[i] = seed + i + threadIdx.x;
    for (int i = 0; i < 4; ++i) w[i] = seed * 0.1f;
    const float a = seed * 0.5f, b = seed * 0.25f;
    const bool mf = __builtin_amdgcn_readfirstlane(threadIdx.x) < 256;
    if (mf) {
        if (mode & 1)
            for (int it = 0; it < ITERS; ++it) {
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[c], 0, 0, 0);
            }
    } else {
        if (mode & 2)
            for (int it = 0; it < ITERS; ++it) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
#pragma unroll
                    for (int i = 0; i < KP; ++i) v[i] = __builtin_fmaf(v[i], a, b);
#pragma unroll
                    for (int i = 0; i < KT; ++i) w[i] = __builtin_amdgcn_exp2f(w[i]);
                }
            }
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += v[i];
    for (int i = 0; i < 4; ++i) s += w[i] + acc[i][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename F>
float timeit(F f)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    f(); (void)hipDeviceSynchronize();
    float best = 1e9;
    for (int r = 0; r < 3; ++r) {
        (void)hipEventRecord(e0); f(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    return best;
}

template <int KP, int KT>
void one(float* d, float base_ms)
{
    float ms = timeit([&] { hipLaunchKernelGGL((k_one<KP, KT, true>), dim3(256), dim3(256), 0, 0, d, 1.0001f); });
    float alone = (KP + KT) ? timeit([&] { hipLaunchKernelGGL((k_one<KP, KT, false>), dim3(256), dim3(256), 0, 0, d, 1.0001f); }) : 0.f;
    printf("one wave/SIMD: 1 mfma + %2d fma + %d exp : %7.3f ms  = %5.2fx mfma-only   (valu stream alone %7.3f ms)\n", KP, KT, ms,
           ms / base_ms, alone);
}

template <int KP, int KT>
void two(float* d)
{
    float m = timeit([&] { hipLaunchKernelGGL((k_two<KP, KT>), dim3(256), dim3(512), 0, 0, d, 1.0001f, 1); });
    float v = timeit([&] { hipLaunchKernelGGL((k_two<KP, KT>), dim3(256), dim3(512), 0, 0, d, 1.0001f, 2); });
    float both = timeit([&] { hipLaunchKernelGGL((k_two<KP, KT>), dim3(256), dim3(512), 0, 0, d, 1.0001f, 3); });
    printf("two waves/SIMD: mfma wave %7.3f ms | valu wave (%2d fma + %d exp per mfma slot) %7.3f ms | together %7.3f ms\n", m, KP, KT,
           v, both);
}

int main()
{
    float* d; (void)hipMalloc(&d, 256 * 512 * sizeof(float));
    float base = timeit([&] { hipLaunchKernelGGL((k_one<0, 0, true>), dim3(256), dim3(256), 0, 0, d, 1.0001f); });
    printf("mfma only: %.3f ms for %d mfma per wave -> %.2f cycles each at 2.4 GHz\n", base, ITERS * 4, base * 1e-3 * 2.4e9 / (ITERS * 4));
    one<2, 0>(d, base); one<4, 0>(d, base); one<6, 0>(d, base); one<8, 0>(d, base); one<10, 0>(d, base); one<12, 0>(d, base);
    one<16, 0>(d, base);
    one<0, 1>(d, base); one<0, 2>(d, base); one<0, 3>(d, base); one<4, 2>(d, base); one<6, 2>(d, base); one<8, 2>(d, base);
    two<6, 0>(d); two<12, 0>(d); two<16, 0>(d); two<4, 2>(d); two<8, 2>(d);
    return 0;
}
