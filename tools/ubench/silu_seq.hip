// Micro-benchmark 5: cost (cycles per wave-instruction per SIMD, 4 waves/SIMD) of the instructions of the edge kernel's
// per-value sequence, alone and as the full sequence.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/silu_seq tools/ubench/silu_seq.hip && /tmp/silu_seq
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITERS 4096
#define REP8(x) x x x x x x x x

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, float seed)
{
    float v0 = seed, v1 = seed + 1, v2 = seed + 2, v3 = seed + 3, v4 = seed + 4, v5 = seed + 5, v6 = seed + 6, v7 = seed + 7;
    float a = seed * 0.5f, b = seed * 0.25f;
    unsigned p0 = 0, p1 = 0, p2 = 0, p3 = 0;
    for (int it = 0; it < ITERS; ++it) {
        if (MODE == 0) asm volatile("v_fma_f32 %0,%0,%8,%9\n v_fma_f32 %1,%1,%8,%9\n v_fma_f32 %2,%2,%8,%9\n v_fma_f32 %3,%3,%8,%9\n v_fma_f32 %4,%4,%8,%9\n v_fma_f32 %5,%5,%8,%9\n v_fma_f32 %6,%6,%8,%9\n v_fma_f32 %7,%7,%8,%9"
                                    : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "v"(a), "v"(b));
        if (MODE == 1) asm volatile("v_cvt_pkrtz_f16_f32 %8,%0,%1\n v_cvt_pkrtz_f16_f32 %9,%2,%3\n v_cvt_pkrtz_f16_f32 %10,%4,%5\n v_cvt_pkrtz_f16_f32 %11,%6,%7\n v_cvt_pkrtz_f16_f32 %8,%1,%0\n v_cvt_pkrtz_f16_f32 %9,%3,%2\n v_cvt_pkrtz_f16_f32 %10,%5,%4\n v_cvt_pkrtz_f16_f32 %11,%7,%6"
                                    : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7), "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));
        if (MODE == 2) asm volatile("v_fma_mix_f32 %0,-%8,1.0,%0 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %1,-%8,1.0,%1 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 %2,-%9,1.0,%2 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %3,-%9,1.0,%3 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 %4,-%10,1.0,%4 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %5,-%10,1.0,%5 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 %6,-%11,1.0,%6 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %7,-%11,1.0,%7 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
                                    : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "v"(p0), "v"(p1), "v"(p2), "v"(p3));
        if (MODE == 3) asm volatile("v_add_f32 %0,%0,%8\n v_add_f32 %1,%1,%8\n v_add_f32 %2,%2,%8\n v_add_f32 %3,%3,%8\n v_mul_f32 %4,%4,%9\n v_mul_f32 %5,%5,%9\n v_mul_f32 %6,%6,%9\n v_mul_f32 %7,%7,%9"
                                    : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "v"(a), "v"(b));
        if (MODE == 4) {   // full sequence for 2 values (16 instructions): add, fmac, exp, add, rcp, mul x2 ; cvt, 2 mix, cvt
            asm volatile(
                "v_add_f32 %0,%0,%8\n v_add_f32 %1,%1,%8\n v_fmac_f32 %0,%9,%2\n v_fmac_f32 %1,%9,%3\n"
                "v_exp_f32 %4,%0\n v_exp_f32 %5,%1\n v_add_f32 %4,1.0,%4\n v_add_f32 %5,1.0,%5\n v_rcp_f32 %4,%4\n v_rcp_f32 %5,%5\n"
                "v_mul_f32 %4,%0,%4\n v_mul_f32 %5,%1,%5\n v_cvt_pkrtz_f16_f32 %10,%4,%5\n"
                "v_fma_mix_f32 %6,-%10,1.0,%4 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %7,-%10,1.0,%5 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n v_cvt_pkrtz_f16_f32 %11,%6,%7"
                : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7), "+v"(a), "+v"(b), "+v"(p0), "+v"(p1));
        }
        if (MODE == 5) asm volatile("v_exp_f32 %0,%0\n v_rcp_f32 %1,%1\n v_exp_f32 %2,%2\n v_rcp_f32 %3,%3\n v_exp_f32 %4,%4\n v_rcp_f32 %5,%5\n v_exp_f32 %6,%6\n v_rcp_f32 %7,%7"
                                    : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7 + p0 + p1 + p2 + p3;
}

template <int MODE>
void run(const char* name, int n_instr, float* d)
{
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int wps = 4;
    hipLaunchKernelGGL(k<MODE>, dim3(256 * wps), dim3(256), 0, 0, d, 1.0001f); (void)hipDeviceSynchronize();
    float best = 1e9;
    for (int r = 0; r < 3; ++r) { (void)hipEventRecord(e0); hipLaunchKernelGGL(k<MODE>, dim3(256 * wps), dim3(256), 0, 0, d, 1.0001f);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); float ms; (void)hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best; }
    printf("%-44s %7.3f ms  %6.2f cycles@2.4GHz per instruction\n", name, best, best * 1e-3 * 2.4e9 / ((double)wps * ITERS * n_instr));
}

int main()
{
    float* d; (void)hipMalloc(&d, 256 * 4 * 256 * sizeof(float));
    run<0>("v_fma_f32", 8, d);
    run<3>("v_add_f32 / v_mul_f32", 8, d);
    run<1>("v_cvt_pkrtz_f16_f32", 8, d);
    run<2>("v_fma_mix_f32 (f16 operand)", 8, d);
    run<5>("v_exp_f32 / v_rcp_f32 alternating", 8, d);
    run<4>("full per-value sequence (16 instr / 2 values)", 16, d);
    return 0;
}
