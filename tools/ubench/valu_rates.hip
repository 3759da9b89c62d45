// Micro-benchmark: issue cost (cycles per wave64 instruction per SIMD) of the VALU / transcendental / MFMA
// instructions the edge kernel is made of, alone and mixed.  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_rates tools/ubench/valu_rates.hip && /tmp/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define ITERS 4096
#define NV 8

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, float seed)
{
    float v[NV];
    for (int i = 0; i < NV; ++i) v[i] = seed + threadIdx.x * 1e-3f + i;
    f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0}, acc3 = {0, 0, 0, 0};
    const float a = seed * 0.5f, b = seed * 0.25f;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            if (MODE == 0) v[i] = __builtin_fmaf(v[i], a, b);                       // v_fma_f32
            if (MODE == 1) v[i] = __builtin_amdgcn_exp2f(v[i]);                     // v_exp_f32
            if (MODE == 2) v[i] = __builtin_amdgcn_rcpf(v[i]);                      // v_rcp_f32
            if (MODE == 3) {                                                        // silu body (6 ops, 2 trans)
                float y = v[i];
                float e = __builtin_amdgcn_exp2f(y);
                float r = __builtin_amdgcn_rcpf(1.0f + e);
                v[i] = __builtin_fmaf(y, r, a);
            }
        }
        if (MODE == 4 || MODE == 5 || MODE == 6) {                                   // MFMA 16x16x4 f32, 4 independent chains
#pragma unroll
            for (int i = 0; i < NV / 4; ++i) {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(v[0], a, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(v[1], a, acc1, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(v[2], a, acc2, 0, 0, 0);
                acc3 = __builtin_amdgcn_mfma_f32_16x16x4f32(v[3], a, acc3, 0, 0, 0);
            }
        }
        if (MODE == 5) {                                                            // + 8 silu bodies beside 8 MFMAs
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                float y = v[i];
                float e = __builtin_amdgcn_exp2f(y);
                float r = __builtin_amdgcn_rcpf(1.0f + e);
                v[i] = __builtin_fmaf(y, r, a);
            }
        }
        if (MODE == 6) {                                                            // + 8 x 6 plain fma beside 8 MFMAs
#pragma unroll
            for (int i = 0; i < NV; ++i) {
#pragma unroll
                for (int j = 0; j < 6; ++j) v[i] = __builtin_fmaf(v[i], a, b);
            }
        }
        if (MODE == 7) {                                                            // packed fma: 2 values / instr
#pragma unroll
            for (int i = 0; i < NV; i += 2) {
                f32x2 x = {v[i], v[i + 1]};
                f32x2 aa = {a, a}, bb = {b, b};
                x = __builtin_elementwise_fma(x, aa, bb);
                v[i] = x[0]; v[i + 1] = x[1];
            }
        }
    }
    float s = 0;
    for (int i = 0; i < NV; ++i) s += v[i];
    s += acc0[0] + acc1[1] + acc2[2] + acc3[3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, int instr_per_iter, int blocks_per_cu, float* d)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * blocks_per_cu;
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 1.0001f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 1.0001f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // waves per SIMD = blocks_per_cu (256 threads = 4 waves = 1 per SIMD)
    const double wave_instr_per_simd = (double)blocks_per_cu * ITERS * instr_per_iter;
    const double cyc = ms * 1e-3 * 2.4e9 / wave_instr_per_simd;     // at the 2.4 GHz max clock (upper bound)
    printf("%-34s waves/SIMD=%d  %8.3f ms  %6.2f cycles per wave-instr (at 2.4 GHz)\n", name, blocks_per_cu, ms, cyc);
}

int main()
{
    float* d; hipMalloc(&d, 256 * 8 * 256 * sizeof(float));
    for (int w : {1, 2, 4}) {
        run<0>("v_fma_f32", NV, w, d);
        run<7>("v_pk_fma_f32 (per pk instr)", NV / 2, w, d);
        run<1>("v_exp_f32", NV, w, d);
        run<2>("v_rcp_f32", NV, w, d);
        run<3>("silu body (exp,add,rcp,fma) per value", NV, w, d);
        run<4>("mfma_f32_16x16x4 (4 chains)", NV, w, d);
        run<5>("8 mfma + 8 silu, per mfma", NV, w, d);
        run<6>("8 mfma + 48 fma, per mfma", NV, w, d);
    }
    return 0;
}
