// Do fp16 subnormals survive (a) the fp32 -> fp16 conversions the kernels use and (b) the MFMA's A / B inputs?  (edge_bwd.hip found
// d/d W_2 at hi-only accuracy until the (hi, lo) operands were pre-scaled: the lo halves of values below ~0.25 are subnormal.)
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/f16_subnormal tools/ubench/f16_subnormal.hip && /tmp/f16_subnormal
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2v __attribute__((ext_vector_type(2)));

__global__ void k(float x, float* out)
{
    const int lane = threadIdx.x;
    // (a) conversions of a value in fp16's subnormal range (x = 3e-6: 6e-8 * 50)
    const _Float16 c1 = (_Float16)x;                                                           // v_cvt_f16_f32
    const f16x2 c2 = __builtin_convertvector((f32x2v){x, x}, f16x2);                           // v_cvt_pk_f16_f32
    const f16x2 c3 = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(x, x));              // v_cvt_pkrtz_f16_f32
    // (b) MFMA with one operand = 1.0 in k-slot 0 and the other = a subnormal built from its bit pattern (50 * 2^-24)
    const _Float16 sub = __builtin_bit_cast(_Float16, (unsigned short)50);
    f16x4 one = {(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f}, sb = one;
    if (lane < 16) { one[0] = (_Float16)1.f; sb[0] = sub; }
    f32x4 z = {0.f, 0.f, 0.f, 0.f};
    const f32x4 dB = __builtin_amdgcn_mfma_f32_16x16x16f16(one, sb, z, 0, 0, 0);               // subnormal in B
    const f32x4 dA = __builtin_amdgcn_mfma_f32_16x16x16f16(sb, one, z, 0, 0, 0);               // subnormal in A
    f16x8 one8 = {0, 0, 0, 0, 0, 0, 0, 0}, sb8 = one8;
    if (lane < 16) { one8[0] = (_Float16)1.f; sb8[0] = sub; }
    const f32x4 d32 = __builtin_amdgcn_mfma_f32_16x16x32_f16(one8, sb8, z, 0, 0, 0);           // the forward's second-layer shape, subnormal in B
    if (lane == 0) {
        out[0] = (float)c1; out[1] = (float)c2[0]; out[2] = (float)c3[0];
        out[3] = dB[0]; out[4] = dA[0]; out[5] = d32[0]; out[6] = (float)sub;
    }
}

int main()
{
    float* d; (void)hipMalloc(&d, 64);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, 3.0e-6f, d);
    float h[7]; (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("x = 3e-6 (fp16 subnormal range; nearest fp16 = %.4e)\n", 50 * 5.9604645e-8);
    printf("v_cvt_f16_f32        -> %.4e\nv_cvt_pk_f16_f32     -> %.4e\nv_cvt_pkrtz_f16_f32  -> %.4e\n", h[0], h[1], h[2]);
    printf("mfma 16x16x16 f16, subnormal %.4e in B x 1.0 -> %.4e\nmfma 16x16x16 f16, subnormal in A x 1.0          -> %.4e\nmfma 16x16x32 f16, subnormal in B x 1.0          -> %.4e\n", h[6], h[3], h[4], h[5]);
    return 0;
}
