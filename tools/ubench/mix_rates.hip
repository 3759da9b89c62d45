// Micro-benchmark 6 (round 2): issue cost of the candidate instructions for the edge kernel's (hi, lo) split, 4 waves/SIMD,
// cycles per wave-instruction at the nominal 2.4 GHz (same convention as silu_seq.hip).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mix_rates tools/ubench/mix_rates.hip && /tmp/mix_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITERS 4096

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, float seed)
{
    float v0 = seed, v1 = seed + 1, v2 = seed + 2, v3 = seed + 3, v4 = seed + 4, v5 = seed + 5, v6 = seed + 6, v7 = seed + 7;
    float a = seed * 0.5f, b = seed * 0.25f;
    unsigned p0 = 0, p1 = 0, p2 = 0, p3 = 0;
    for (int it = 0; it < ITERS; ++it) {
        if (MODE == 0)   // v_fma_mixlo_f16 / v_fma_mixhi_f16, f32 sources
            asm volatile("v_fma_mixlo_f16 %8,%0,%1,0 op_sel_hi:[0,0,0]\n v_fma_mixhi_f16 %8,%2,%3,0 op_sel_hi:[0,0,0]\n"
                         "v_fma_mixlo_f16 %9,%4,%5,0 op_sel_hi:[0,0,0]\n v_fma_mixhi_f16 %9,%6,%7,0 op_sel_hi:[0,0,0]\n"
                         "v_fma_mixlo_f16 %10,%1,%0,0 op_sel_hi:[0,0,0]\n v_fma_mixhi_f16 %10,%3,%2,0 op_sel_hi:[0,0,0]\n"
                         "v_fma_mixlo_f16 %11,%5,%4,0 op_sel_hi:[0,0,0]\n v_fma_mixhi_f16 %11,%7,%6,0 op_sel_hi:[0,0,0]"
                         : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7), "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));
        if (MODE == 1)   // the same with an f16 third source (the lo form)
            asm volatile("v_fma_mixlo_f16 %8,%0,%1,-%9 op_sel_hi:[0,0,1]\n v_fma_mixhi_f16 %8,%2,%3,-%9 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n"
                         "v_fma_mixlo_f16 %10,%4,%5,-%11 op_sel_hi:[0,0,1]\n v_fma_mixhi_f16 %10,%6,%7,-%11 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n"
                         "v_fma_mixlo_f16 %9,%1,%0,-%8 op_sel_hi:[0,0,1]\n v_fma_mixhi_f16 %9,%3,%2,-%8 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n"
                         "v_fma_mixlo_f16 %11,%5,%4,-%10 op_sel_hi:[0,0,1]\n v_fma_mixhi_f16 %11,%7,%6,-%10 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
                         : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7), "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));
        if (MODE == 2)   // v_cvt_pk_f16_f32 (gfx950, round to nearest)
            asm volatile("v_cvt_pk_f16_f32 %8,%0,%1\n v_cvt_pk_f16_f32 %9,%2,%3\n v_cvt_pk_f16_f32 %10,%4,%5\n v_cvt_pk_f16_f32 %11,%6,%7\n"
                         "v_cvt_pk_f16_f32 %8,%1,%0\n v_cvt_pk_f16_f32 %9,%3,%2\n v_cvt_pk_f16_f32 %10,%5,%4\n v_cvt_pk_f16_f32 %11,%7,%6"
                         : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7), "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));
        if (MODE == 3)   // old per-pair sequence: exp, add, rcp x2; mul x2, cvt_pkrtz, fma_mix x2, cvt_pkrtz  (12 instr / 2 values)
            asm volatile("v_exp_f32 %4,%0\n v_exp_f32 %5,%1\n v_add_f32 %4,1.0,%4\n v_add_f32 %5,1.0,%5\n v_rcp_f32 %4,%4\n v_rcp_f32 %5,%5\n"
                         "v_mul_f32 %6,%0,%4\n v_mul_f32 %7,%1,%5\n v_cvt_pkrtz_f16_f32 %8,%6,%7\n"
                         "v_fma_mix_f32 %6,%0,%4,-%8 op_sel_hi:[0,0,1]\n v_fma_mix_f32 %7,%1,%5,-%8 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n v_cvt_pkrtz_f16_f32 %9,%6,%7"
                         : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7), "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));
        if (MODE == 4)   // new per-pair sequence: exp, add, rcp x2; mixlo, mixhi, mixlo, mixhi  (10 instr / 2 values)
            asm volatile("v_exp_f32 %4,%0\n v_exp_f32 %5,%1\n v_add_f32 %4,1.0,%4\n v_add_f32 %5,1.0,%5\n v_rcp_f32 %4,%4\n v_rcp_f32 %5,%5\n"
                         "v_fma_mixlo_f16 %8,%0,%4,0 op_sel_hi:[0,0,0]\n v_fma_mixhi_f16 %8,%1,%5,0 op_sel_hi:[0,0,0]\n"
                         "v_fma_mixlo_f16 %9,%0,%4,-%8 op_sel_hi:[0,0,1]\n v_fma_mixhi_f16 %9,%1,%5,-%8 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
                         : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7), "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));
        if (MODE == 5)   // v_pk_mul_f32 (2 values per instruction)
            asm volatile("v_pk_mul_f32 %0,%0,%4\n v_pk_mul_f32 %2,%2,%4\n v_pk_mul_f32 %0,%0,%6\n v_pk_mul_f32 %2,%2,%6\n"
                         "v_pk_mul_f32 %0,%0,%4\n v_pk_mul_f32 %2,%2,%4\n v_pk_mul_f32 %0,%0,%6\n v_pk_mul_f32 %2,%2,%6"
                         : "+v"(*(double*)&v0), "+v"(*(double*)&v1), "+v"(*(double*)&v2), "+v"(*(double*)&v3), "+v"(*(double*)&v4),
                           "+v"(*(double*)&v5), "+v"(*(double*)&v6), "+v"(*(double*)&v7));
        if (MODE == 6)   // v_exp_f32 only / MODE 7: v_rcp_f32 only
            asm volatile("v_exp_f32 %0,%0\n v_exp_f32 %1,%1\n v_exp_f32 %2,%2\n v_exp_f32 %3,%3\n v_exp_f32 %4,%4\n v_exp_f32 %5,%5\n v_exp_f32 %6,%6\n v_exp_f32 %7,%7"
                         : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7));
        if (MODE == 7)
            asm volatile("v_rcp_f32 %0,%0\n v_rcp_f32 %1,%1\n v_rcp_f32 %2,%2\n v_rcp_f32 %3,%3\n v_rcp_f32 %4,%4\n v_rcp_f32 %5,%5\n v_rcp_f32 %6,%6\n v_rcp_f32 %7,%7"
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
    printf("%-58s %7.3f ms  %6.2f cycles@2.4GHz per instruction, %6.2f per sequence\n", name, best,
           best * 1e-3 * 2.4e9 / ((double)wps * ITERS * n_instr), best * 1e-3 * 2.4e9 / ((double)wps * ITERS));
}

int main()
{
    float* d; (void)hipMalloc(&d, 256 * 4 * 256 * sizeof(float));
    run<0>("v_fma_mixlo/hi_f16 (f32 sources)", 8, d);
    run<1>("v_fma_mixlo/hi_f16 (f16 third source)", 8, d);
    run<2>("v_cvt_pk_f16_f32", 8, d);
    run<5>("v_pk_mul_f32", 8, d);
    run<6>("v_exp_f32", 8, d);
    run<7>("v_rcp_f32", 8, d);
    run<3>("old pair: exp,add,rcp x2 + mul,mul,cvt,mix,mix,cvt (12)", 12, d);
    run<4>("new pair: exp,add,rcp x2 + mixlo,mixhi,mixlo,mixhi (10)", 10, d);
    return 0;
}
