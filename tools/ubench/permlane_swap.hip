// Semantics check of the row-swap reduction used by the edge backward (csrc/egnn_common.h::egnn_column_sum4_reg):
// every lane must end with v[e] + v[16 + e] + v[32 + e] + v[48 + e], e = lane & 15, added as (row0 + row1) + (row2 + row3).
//   hipcc --offload-arch=gfx950 -O3 -I egnn_pytorch_amd/csrc tools/ubench/permlane_swap.hip -o /tmp/pls && /tmp/pls
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "egnn_common.h"

__global__ void k(const float* in, float* out) { out[threadIdx.x] = egnn_column_sum4_reg(in[threadIdx.x]); }

int main()
{
    float h[256], r[256], *d_in, *d_out;
    for (int i = 0; i < 256; ++i) h[i] = 1.0f + 0.37f * i + 1e-3f * (i * i % 17);
    hipMalloc(&d_in, sizeof(h)); hipMalloc(&d_out, sizeof(h));
    hipMemcpy(d_in, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, d_in, d_out);
    hipMemcpy(r, d_out, sizeof(r), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int w = 0; w < 4; ++w)
        for (int l = 0; l < 64; ++l) {
            const float* v = h + 64 * w;
            const int e = l & 15;
            const float want = (v[e] + v[16 + e]) + (v[32 + e] + v[48 + e]);
            if (r[64 * w + l] != want) ++bad;
        }
    printf("permlane swap column sum: %d mismatches of 256\n", bad);
    return bad != 0;
}
