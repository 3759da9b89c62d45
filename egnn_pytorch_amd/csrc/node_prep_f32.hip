// node_norm + concat with fp32 output (the wide-range path, include/egnn_hip.h: egnn_node_prep_f32),
//     out[r] = [ LayerNorm(feats[r]) (or feats[r]) | m_i[r] ]        (egnn_pytorch/egnn_pytorch.py:335-336)
// One wavefront per row; row statistics by DPP reductions (two-pass: mean, then centred variance, as torch's LayerNorm).
// The fast path writes the packed fp16 (hi, lo) pair instead (node_ops.hip); this kernel is also its A/B reference in the tests.
#include "egnn_common.h"

namespace {

__device__ __forceinline__ float wave_sum(float v) { return egnn_wave_sum(v); }

__global__ __launch_bounds__(256) void node_prep_kernel(const float* __restrict__ feats, const float* __restrict__ m_i,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        float eps, float* __restrict__ out, int64_t rows, int dim,
                                                        int m_dim)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    const int od = dim + m_dim;
    for (int64_t r = wave0; r < rows; r += nwaves) {
        const float* x = feats + r * dim;
        float* y = out + r * od;
        if (gamma) {
            float s = 0.f;
            for (int c = lane; c < dim; c += 64) s += x[c];
            const float mean = wave_sum(s) / (float)dim;
            float v = 0.f;
            for (int c = lane; c < dim; c += 64) { const float d = x[c] - mean; v += d * d; }
            const float var = wave_sum(v) / (float)dim;
            const float rstd = 1.0f / sqrtf(var + eps);
            for (int c = lane; c < dim; c += 64) y[c] = (x[c] - mean) * rstd * gamma[c] + beta[c];
        } else {
            for (int c = lane; c < dim; c += 64) y[c] = x[c];
        }
        if (m_i) {
            for (int c = lane; c < m_dim; c += 64) y[dim + c] = m_i[r * m_dim + c];
        } else {
            for (int c = lane; c < m_dim; c += 64) y[dim + c] = 0.f;
        }
    }
}

}  // namespace

extern "C" int egnn_node_prep_f32(const float* feats, const float* m_i, const float* gamma, const float* beta, float eps,
                                  float* out, int64_t rows, int dim, int m_dim, void* stream)
{
    if (!feats || !out) return EGNN_E_NULLPTR;
    if ((gamma == nullptr) != (beta == nullptr)) return EGNN_E_NULLPTR;
    if (rows <= 0 || dim <= 0 || m_dim < 0) return EGNN_E_SHAPE;
    int64_t blocks = (rows + 3) / 4;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(node_prep_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), feats,
                       m_i, gamma, beta, eps, out, rows, dim, m_dim);
    return egnn_launch_status();
}

