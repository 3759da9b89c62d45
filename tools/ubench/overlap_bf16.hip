// Micro-benchmark 3: does a bf16 MFMA (real matrix pipe) overlap with f32 VALU work on the same SIMD?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/overlap_bf16 tools/ubench/overlap_bf16.hip && /tmp/overlap_bf16
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define ITERS 2048

// SHAPE 0: 16x16x32 bf16, SHAPE 1: 32x32x16 bf16
template <int SHAPE, int KP, int KT, bool MFMA>
__global__ __launch_bounds__(256) void k_one(float* out, float seed)
{
    f32x4 acc4[4];
    f32x16 acc16[2];
    float v[16], w[4];
    for (int c = 0; c < 4; ++c) acc4[c] = f32x4{seed, seed, seed, seed};
    for (int c = 0; c < 2; ++c) for (int r = 0; r < 16; ++r) acc16[c][r] = seed;
    for (int i = 0; i < 16; ++i) v[i] = seed + i + threadIdx.x;
    for (int i = 0; i < 4; ++i) w[i] = seed * 0.1f;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(seed + i); b[i] = (__bf16)(seed - i); }
    const float fa = seed * 0.5f, fb = seed * 0.25f;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (MFMA) {
                if (SHAPE == 0) acc4[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc4[c], 0, 0, 0);
                else acc16[c & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc16[c & 1], 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < KP; ++i) v[i] = __builtin_fmaf(v[i], fa, fb);
#pragma unroll
            for (int i = 0; i < KT; ++i) w[i] = __builtin_amdgcn_exp2f(w[i]);
            if (MFMA) {
                __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x2, KP + KT, 0);
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += v[i];
    for (int i = 0; i < 4; ++i) s += w[i] + acc4[i][i];
    s += acc16[0][3] + acc16[1][5];
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

template <int SHAPE, int KP, int KT>
void one(float* d, int wps)
{
    const int blocks = 256 * wps;
    float m = timeit([&] { hipLaunchKernelGGL((k_one<SHAPE, 0, 0, true>), dim3(blocks), dim3(256), 0, 0, d, 1.0001f); });
    float v = timeit([&] { hipLaunchKernelGGL((k_one<SHAPE, KP, KT, false>), dim3(blocks), dim3(256), 0, 0, d, 1.0001f); });
    float both = timeit([&] { hipLaunchKernelGGL((k_one<SHAPE, KP, KT, true>), dim3(blocks), dim3(256), 0, 0, d, 1.0001f); });
    printf("%s waves/SIMD=%d: mfma-only %7.3f ms (%5.1f cyc@2.4GHz each) | valu-only (%2d fma + %d exp per mfma) %7.3f ms | interleaved %7.3f ms\n",
           SHAPE == 0 ? "16x16x32_bf16" : "32x32x16_bf16", wps, m, m * 1e-3 * 2.4e9 / (ITERS * 4 * wps), KP, KT, v, both);
}

int main()
{
    float* d; (void)hipMalloc(&d, 256 * 4 * 256 * sizeof(float));
    for (int wps : {1, 2}) {
        one<0, 2, 0>(d, wps); one<0, 4, 0>(d, wps); one<0, 6, 0>(d, wps); one<0, 4, 1>(d, wps); one<0, 8, 2>(d, wps);
        one<1, 4, 0>(d, wps); one<1, 8, 0>(d, wps); one<1, 12, 0>(d, wps); one<1, 8, 2>(d, wps); one<1, 12, 4>(d, wps);
    }
    return 0;
}
