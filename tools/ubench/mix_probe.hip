// Experiment (not part of libegnn_hip.so): ONE launch whose workgroup slots alternate between the VALU-bound edge pass
// and the MFMA-bound projection GEMM of ANOTHER chunk of the batch -- do the two pipes overlap when waves of both kinds
// share a SIMD?  Built by tools/mix_probe.py into its own shared object; this file includes the two kernel sources of the package.
#define EGNN_EDGE_TUNING_BUILD
#include "../../egnn_pytorch_amd/csrc/edge_fused.hip"
#include "../../egnn_pytorch_amd/csrc/linear_hl.hip"

int egnn_edge_fused_generic_c(const egnn_edge_args*, void*) { return EGNN_E_UNSUPPORTED; }   // (not built into the probe)

namespace {

struct MixGemm {
    const _Float16 *Ahi, *Alo, *Whi, *Wlo;
    const float* bias;
    float* C;
    int64_t ldc, M;
    int N, Kp, ntm, ntn, split_cols;
    float out_scale;
};

// 128 x 128 tiles, 4 waves, 3-deep ring: 48 KB -- the same 256-thread workgroup shape as the edge pass
template <> struct Cfg<6> { static constexpr int BM = 128, BN = 128, WM = 2, WN = 2, TI = 2, TJ = 2, STAGES = 3; };

// blocks come in groups of 16: 8 GEMM tiles, then 8 edge groups (so that each role keeps bid % 8 = XCD)
__global__ __launch_bounds__(256, 3) void mix_kernel(const egnn_edge_args ea, const int G, const int gpg, const int n_edge,
                                                    const MixGemm ga, const int n_gemm, const int mode)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int bid = blockIdx.x;
    int role, id;
    if (mode == 0) {                        // interleaved
        const int grp = bid >> 4, in = bid & 15;
        role = in >> 3;
        id = grp * 8 + (in & 7);
    } else {                                // all GEMM tiles first, then all edge groups (same kernel, no mixing)
        role = bid >= n_gemm;
        id = role ? bid - n_gemm : bid;
    }
    if (role == 0) {
        if (id < n_gemm)
            linear_hl_body<6, 0, false>(ga.Ahi, ga.Alo, ga.Whi, ga.Wlo, ga.bias, nullptr, 0, ga.C, ga.ldc, nullptr, nullptr, 0,
                                        ga.M, ga.N, ga.Kp, ga.ntm, ga.ntn, ga.out_scale, ga.split_cols, nullptr, smem, id);
    } else {
        if (id < n_edge) edge_body<1, 256, 2, 1>(ea, G, gpg, smem, id, n_edge);
    }
}

}  // namespace

extern "C" int egnn_mix_probe(const egnn_edge_args* ea, const void* A_hi, const void* A_lo, const void* W_hi, const void* W_lo,
                              float w_inv_scale, const float* bias, float* C, int64_t ldc, int64_t M, int N, int Kp,
                              int split_cols, int mode, int which, void* stream)
{
    const egnn_edge_args& a = *ea;
    int G = SLOTS_PER_ROUND / a.K;
    if (G > a.N) G = a.N;
    const int gpg = (a.N + G - 1) / G;
    const int n_edge = (which & 1) ? a.B * gpg : 0;
    MixGemm g;
    g.Ahi = (const _Float16*)A_hi; g.Alo = (const _Float16*)A_lo; g.Whi = (const _Float16*)W_hi; g.Wlo = (const _Float16*)W_lo;
    g.bias = bias; g.C = C; g.ldc = ldc; g.M = M; g.N = N; g.Kp = Kp;
    g.ntm = (int)((M + 127) / 128); g.ntn = (N + 127) / 128; g.split_cols = split_cols; g.out_scale = w_inv_scale;
    const int n_gemm = (which & 2) ? g.ntm * g.ntn : 0;
    const int mx = n_edge > n_gemm ? n_edge : n_gemm;
    const int nblk = mode == 0 ? ((mx + 7) / 8) * 16 : n_edge + n_gemm;
    const size_t lds_edge = (size_t)256 * 64 + sizeof(float) * ((size_t)SLOTS_PER_ROUND * XLD + (size_t)G * NCH) + (size_t)256 * 16;
    const size_t lds_gemm = (size_t)Cfg<6>::STAGES * (2 * 128 + 2 * 128) * ROWB;
    const size_t lds = lds_edge > lds_gemm ? lds_edge : lds_gemm;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(mix_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(mix_kernel, dim3(nblk), dim3(256), lds, static_cast<hipStream_t>(stream), a, G, gpg, n_edge, g, n_gemm, mode);
    return (int)hipGetLastError();
}
