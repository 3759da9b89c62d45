// Stand-alone reproducer attempt for the round-1 observation "ds_bpermute_b32 occasionally returns another value while
// LDS-DMA (global_load_lds) traffic of co-resident workgroups is in flight" (VERDICT r1 item 9).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/bpermute_dma tools/ubench/bpermute_dma.hip && /tmp/bpermute_dma
// Every CU hosts workgroups of two roles (block parity): "shufflers" run the exact pattern the edge kernel used -- the sum over
// the four 16-lane groups with two ds_bpermute_b32 (xor 16, xor 32) -- on values whose correct result is known, and count
// mismatches; "loaders" keep 1-KB LDS-DMA pieces in flight into their own LDS.  MODE 0: shufflers only (control);
// MODE 1: shufflers + loaders co-resident; MODE 2: every workgroup does both (two waves shuffle, two waves load).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;
#define ITERS 20000

__device__ __forceinline__ float bperm(float v, int src_lane) {
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src_lane << 2, __builtin_bit_cast(int, v)));
}

template <int MODE>
__global__ __launch_bounds__(256) void k(const char* __restrict__ src, unsigned long long* __restrict__ bad, float* __restrict__ sink)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];          // 16 KB: the loaders' landing zone
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool loader = MODE == 1 ? (blockIdx.x & 1) : (MODE == 2 ? wave >= 2 : false);
    if (loader) {
        const char* g = src + ((size_t)(blockIdx.x * 4 + wave) % 4096) * 1024 + lane * 16;
        for (int it = 0; it < ITERS; ++it) {
#pragma unroll
            for (int p = 0; p < 4; ++p)
                __builtin_amdgcn_global_load_lds((glb_void*)(g + (size_t)((it * 4 + p) & 1023) * 4096), (lds_void*)(smem + (wave * 4 + p) * 1024), 16, 0, 0);
            if ((it & 3) == 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) sink[blockIdx.x * 4 + wave] = *reinterpret_cast<float*>(smem + wave * 4096);
        return;
    }
    unsigned long long mism = 0;
    for (int it = 0; it < ITERS; ++it) {
        // value = f(lane, it); the 4-group sum of column e = lane & 15 is known in closed form
        const float v = (float)((lane * 7 + it) & 1023);
        float s = v + bperm(v, lane ^ 16);
        s = s + bperm(s, lane ^ 32);
        const int e = lane & 15;
        float want = 0.f;
#pragma unroll
        for (int g = 0; g < 4; ++g) want += (float)((((e + 16 * g) * 7) + it) & 1023);
        mism += (s != want);
    }
    if (mism) atomicAdd(bad, mism);
}

template <int MODE>
void run(const char* name, const char* src, unsigned long long* bad, float* sink)
{
    (void)hipMemset(bad, 0, 8);
    hipLaunchKernelGGL(k<MODE>, dim3(256 * 8), dim3(256), 16384, 0, src, bad, sink);
    (void)hipDeviceSynchronize();
    unsigned long long h = 0;
    (void)hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost);
    printf("%-58s mismatching shuffles: %llu\n", name, h);
}

int main()
{
    char* src; unsigned long long* bad; float* sink;
    (void)hipMalloc(&src, (size_t)8 << 20); (void)hipMemset(src, 1, (size_t)8 << 20);
    (void)hipMalloc(&bad, 8); (void)hipMalloc(&sink, 256 * 8 * 4 * sizeof(float));
    for (int rep = 0; rep < 3; ++rep) {
        run<0>("shufflers only (control)", src, bad, sink);
        run<1>("shufflers + LDS-DMA loaders co-resident (block parity)", src, bad, sink);
        run<2>("both roles inside every workgroup (waves 0-1 / 2-3)", src, bad, sink);
    }
    return 0;
}
