// node_norm + concat feeding node_mlp (reference: egnn_pytorch/egnn_pytorch.py:335-336):
//     out[r] = [ LayerNorm(feats[r]) (or feats[r]) | m_i[r] ]
// as the packed fp16 (hi, lo) operand pair of the GEMM; row statistics two-pass (mean, then centred variance, as torch's
// LayerNorm); HBM-bound streaming kernel.  (The fp32-output variant is test-only: csrc/node_prep_ref.hip.)
#include "egnn_common.h"

namespace {

// Packed-layout producer: the row [LayerNorm(x) | m_i] (or just x) as the (hi, lo) f16 pair the matrix-core GEMM consumes.
// 16 lanes per row, 4 consecutive rows per wave: a lane converts 8 consecutive columns (one 16-byte chunk) at a time, so a
// wave load covers 4 x 512 contiguous bytes and a wave store fills whole 128-byte lines of the packed layout (rows r..r+3
// of one K-tile are adjacent).  Row statistics: two-pass (mean, centred variance) with 16-lane DPP butterflies.
__device__ __forceinline__ float sum16(float v) { return egnn_row16_sum(v); }

__global__ __launch_bounds__(256) void node_prep_hl_kernel(const float* __restrict__ feats, int64_t ldx, const float* __restrict__ m_i,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           float eps, _Float16* __restrict__ hi, _Float16* __restrict__ lo,
                                                           int Kp, _Float16* __restrict__ raw_hi, _Float16* __restrict__ raw_lo,
                                                           int raw_Kp, int64_t rows, int dim, int m_dim, int32_t* __restrict__ status)
{
    typedef _Float16 f16x8v __attribute__((ext_vector_type(8)));
    const int lane = threadIdx.x & 63;
    const int sub = lane & 15;
    const int64_t quad0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6));         // one wave = 4 rows
    const int64_t nquads = (int64_t)gridDim.x * 4;
    const int nkt = Kp / 16;
    const int raw_nkt = raw_Kp / 16;
    const int Kmax = raw_hi && raw_Kp > Kp ? raw_Kp : Kp;
    const int64_t rows_p = (rows + 31) / 32 * 32;                                  // pad rows are written as zeros
    for (int64_t qd = quad0; qd * 4 < rows_p; qd += nquads) {
        const int64_t r = qd * 4 + (lane >> 4);
        const bool live = r < rows;
        const float* x = feats + (live ? r : 0) * ldx;
        float mean = 0.f, rstd = 1.f;
        if (gamma) {
            // statistics over 16-byte chunks per lane (whole 128-B lines per wave instruction), remainder scalar
            const bool vec = (dim % 4 == 0) && ((ldx % 4) == 0);
            float s = 0.f;
            if (vec) {
                for (int c = sub * 4; c < dim; c += 64) {
                    const f32x4 q = *reinterpret_cast<const f32x4*>(x + c);
                    s += (q[0] + q[1]) + (q[2] + q[3]);
                }
            } else {
                for (int c = sub; c < dim; c += 16) s += x[c];
            }
            mean = sum16(s) / (float)dim;
            float v = 0.f;
            if (vec) {
                for (int c = sub * 4; c < dim; c += 64) {
                    const f32x4 q = *reinterpret_cast<const f32x4*>(x + c) - mean;
                    v += (q[0] * q[0] + q[1] * q[1]) + (q[2] * q[2] + q[3] * q[3]);
                }
            } else {
                for (int c = sub; c < dim; c += 16) { const float d = x[c] - mean; v += d * d; }
            }
            rstd = 1.0f / sqrtf(sum16(v) / (float)dim + eps);
        }
        for (int c0 = sub * 8; c0 < Kmax; c0 += 128) {
            f16x8v h8, l8, rh8, rl8;
            float xv[8], yv[8];
            if (live && c0 + 8 <= dim && (dim % 4 == 0) && (ldx % 4) == 0) {      // whole chunk inside the row: 16-byte loads
                const f32x4 q0 = *reinterpret_cast<const f32x4*>(x + c0), q1 = *reinterpret_cast<const f32x4*>(x + c0 + 4);
#pragma unroll
                for (int u = 0; u < 4; ++u) { xv[u] = q0[u]; xv[4 + u] = q1[u]; }
                if (gamma) {
                    const f32x4 g0 = *reinterpret_cast<const f32x4*>(gamma + c0), g1 = *reinterpret_cast<const f32x4*>(gamma + c0 + 4);
                    const f32x4 b0 = *reinterpret_cast<const f32x4*>(beta + c0), b1 = *reinterpret_cast<const f32x4*>(beta + c0 + 4);
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        yv[u] = (xv[u] - mean) * rstd * g0[u] + b0[u];
                        yv[4 + u] = (xv[4 + u] - mean) * rstd * g1[u] + b1[u];
                    }
                } else {
#pragma unroll
                    for (int u = 0; u < 8; ++u) yv[u] = xv[u];
                }
            } else {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int c = c0 + u;
                    float y = 0.f, xr = 0.f;
                    if (live) {
                        if (c < dim) {
                            xr = x[c];
                            y = gamma ? (xr - mean) * rstd * gamma[c] + beta[c] : xr;
                        } else if (c < dim + m_dim) {
                            y = m_i ? m_i[r * m_dim + (c - dim)] : 0.f;
                        }
                    }
                    xv[u] = xr; yv[u] = y;
                }
            }
            bool beyond = false;                                   // finite values the fp16 pair cannot carry (they turn into inf / NaN)
#pragma unroll
            for (int u = 0; u < 8; ++u) beyond = beyond || egnn_beyond_f16(yv[u]) || (raw_hi && egnn_beyond_f16(xv[u]));
            egnn_flag_range(status, beyond, EGNN_RANGE_A_OPERAND);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const _Float16 h = (_Float16)yv[u];
                h8[u] = h;
                l8[u] = (_Float16)(yv[u] - (float)h);
                const _Float16 rh = (_Float16)xv[u];
                rh8[u] = rh;
                rl8[u] = (_Float16)(xv[u] - (float)rh);
            }
            if (c0 < Kp) {
                const size_t o = egnn_pk_off(r, c0, nkt);
                *reinterpret_cast<f16x8v*>(hi + o) = h8;
                *reinterpret_cast<f16x8v*>(lo + o) = l8;
            }
            if (raw_hi && c0 < raw_Kp) {                              // the un-normalised row as a second (hi, lo) pair
                const size_t o = egnn_pk_off(r, c0, raw_nkt);
                *reinterpret_cast<f16x8v*>(raw_hi + o) = rh8;
                *reinterpret_cast<f16x8v*>(raw_lo + o) = rl8;
            }
        }
    }
}

// X (rows, cols) fp32 -> packed (hi, lo) images of scale * X (transposed = 0) or of (scale * X)^T (transposed = 1: the image has
// `cols` rows and K = rows) -- the operands of the backward's node-level gradient GEMMs (egnn_split_scaled_f16).
// One workgroup = 64 X-rows x 32 X-columns through LDS; every thread emits one 16-byte chunk (8 consecutive K values of one
// image row) per image, so that the workgroup's stores fill whole 1 KB (row block, K-tile) pieces.
// transposed = 2: both images from one read of X (hi / lo = the plain image, hiT / loT = the transposed one).
// colsum_parts (or NULL): row blockIdx.y of a (gridDim.y, ld_cs) array receives the tile's column sums of scale * X -- a gradient matrix's
// column sums are the gradient of the Linear's bias, and this pass reads every element anyway (summed over the row blocks in fixed
// order by egnn_sum_parts_f32; the scale is a power of two: exact).
__global__ __launch_bounds__(256) void split_scaled_kernel(const float* __restrict__ X, int64_t ldx, int64_t rows, int cols, float scale,
                                                           int transposed, _Float16* __restrict__ hi, _Float16* __restrict__ lo, int nkt,
                                                           int64_t img_rows_p, _Float16* __restrict__ hiT, _Float16* __restrict__ loT, int nktT,
                                                           int64_t img_rows_pT, int32_t* __restrict__ status,
                                                           float* __restrict__ colsum_parts = nullptr, int64_t ld_cs = 0)
{
    typedef _Float16 f16x8v __attribute__((ext_vector_type(8)));
    __shared__ float tile[64][33];
    __shared__ float csum[8][32];
    const int tid = threadIdx.x;
    const int64_t r0 = (int64_t)blockIdx.y * 64;
    const int c0 = blockIdx.x * 32;
    for (int o = tid; o < 64 * 32; o += 256) {
        const int r = o >> 5, c = o & 31;
        float v = 0.f;
        if (r0 + r < rows && c0 + c < cols) v = X[(r0 + r) * ldx + c0 + c] * scale;
        tile[r][c] = v;
    }
    __syncthreads();
    if (colsum_parts) {                       // (uniform branch) rows 8 q .. 8 q + 7 of column c, then the eight partial sums in order
        const int c = tid & 31, q = tid >> 5;
        float sacc = 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u) sacc += tile[8 * q + u][c];
        csum[q][c] = sacc;
        __syncthreads();
        if (tid < 32 && c0 + tid < ld_cs) {
            float t = csum[0][tid];
#pragma unroll
            for (int u = 1; u < 8; ++u) t += csum[u][tid];
            colsum_parts[(int64_t)blockIdx.y * ld_cs + c0 + tid] = t;
        }
    }
    for (int pass = 0; pass < (transposed == 2 ? 2 : 1); ++pass) {
    const bool tr = transposed == 2 ? pass == 1 : transposed != 0;
    _Float16* ohi = (transposed == 2 && pass == 1) ? hiT : hi;
    _Float16* olo = (transposed == 2 && pass == 1) ? loT : lo;
    const int onkt = (transposed == 2 && pass == 1) ? nktT : nkt;
    const int64_t orows = (transposed == 2 && pass == 1) ? img_rows_pT : img_rows_p;
    float x[8];
    int64_t out_row;
    int out_k;
    if (tr) {                                 // image row = X column c0 + (tid & 31), K = X rows r0 + 8 q .. + 7
        const int c = tid & 31, q = tid >> 5;
#pragma unroll
        for (int u = 0; u < 8; ++u) x[u] = tile[8 * q + u][c];
        out_row = c0 + c;
        out_k = (int)(r0 + 8 * q);
    } else {                                  // image row = X row r0 + (tid >> 2), K = X columns c0 + 8 (tid & 3) .. + 7
        const int r = tid >> 2, q = tid & 3;
#pragma unroll
        for (int u = 0; u < 8; ++u) x[u] = tile[r][8 * q + u];
        out_row = r0 + r;
        out_k = c0 + 8 * q;
    }
    bool beyond = false;
    f16x8v h8, l8;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        beyond = beyond || egnn_beyond_f16(x[u]);
        const _Float16 h = (_Float16)x[u];
        h8[u] = h;
        l8[u] = (_Float16)(x[u] - (float)h);
    }
    egnn_flag_range(status, beyond, EGNN_RANGE_A_OPERAND);
    if (out_k < onkt * 16 && out_row < orows) {           // (inside the padded image: rows / K beyond the matrix are written as zeros)
        const size_t o = egnn_pk_off(out_row, out_k, onkt);
        *reinterpret_cast<f16x8v*>(ohi + o) = h8;
        *reinterpret_cast<f16x8v*>(olo + o) = l8;
    }
    }
}

// a = SiLU(z) and gz = g SiLU'(z) in one pass (the backward of node_mlp's activation, egnn_pytorch.py:196-201); a_out may be z, gz_out may be g
// drop_thr != 0: nn.Dropout sits between the Linear and the SiLU (egnn_pytorch.py:196-201): z is the Linear's output, the forward's
// hash mask (site node, row = row0 + element / cols, column = element % cols) is re-evaluated: z_d = keep ? z k : 0, a = SiLU(z_d),
// gz = g SiLU'(z_d) (keep ? k : 0).
__global__ __launch_bounds__(256) void silu_bwd_kernel(const float* z, const float* g, float* a_out, float* gz_out, int64_t quads, uint32_t* amax_bits,
                                                       uint32_t drop_thr, uint32_t drop_seed, float drop_inv_keep, int64_t row0, int cols)
{
    __shared__ uint32_t slot_a, slot_g;
    uint32_t ma = 0u, mg = 0u;
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < quads; q += (int64_t)gridDim.x * 256) {
        const f32x4 zv = reinterpret_cast<const f32x4*>(z)[q];
        const f32x4 gv = reinterpret_cast<const f32x4*>(g)[q];
        f32x4 av, dv;
        uint32_t key = 0u;
        int col0 = 0;
        int64_t row = 0;
        if (drop_thr) {
            row = (q * 4) / cols;
            col0 = (int)(q * 4 - row * cols);
            key = egnn_drop_base(drop_seed, EGNN_DROP_SITE_NODE, (uint32_t)(row0 + row));
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float zz = zv[u], gk = 1.0f;
            if (drop_thr) {
                if (u && col0 + u == cols) {                     // cols % 4 != 0 (node_mlp's hidden width 2 dim with an odd dim): the quad
                    col0 -= cols;                                // crosses into the next row
                    key = egnn_drop_base(drop_seed, EGNN_DROP_SITE_NODE, (uint32_t)(row0 + ++row));
                }
                const bool keep = egnn_drop_hash(key, (uint32_t)(col0 + u)) >= drop_thr;
                zz = keep ? zz * drop_inv_keep : 0.f;
                gk = keep ? drop_inv_keep : 0.f;
            }
            const float sg = 1.0f / (1.0f + __expf(-zz));
            av[u] = zz * sg;
            dv[u] = gv[u] * (sg * (1.0f + zz * (1.0f - sg))) * gk;
            const uint32_t ta = egnn_abs_bits(av[u]), tg = egnn_abs_bits(dv[u]);
            ma = ma > ta ? ma : ta;
            mg = mg > tg ? mg : tg;
        }
        reinterpret_cast<f32x4*>(a_out)[q] = av;
        reinterpret_cast<f32x4*>(gz_out)[q] = dv;
    }
    if (amax_bits) {
        egnn_block_absmax_commit(ma, &slot_a, amax_bits);
        egnn_block_absmax_commit(mg, &slot_g, amax_bits + 1);
    }
}

// max |x| over a flat array as the bit pattern of the float (non-negative floats order like unsigned integers; a NaN's pattern
// is above infinity's, so it wins and the host sees it): integer atomicMax -- the result does not depend on the order
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ X, int64_t count, uint32_t* __restrict__ out_bits)
{
    const int64_t quads = count >> 2;
    const int64_t stride = (int64_t)gridDim.x * 256;
    uint32_t m = 0u;
    auto fold = [&](const uint4 v) {
        const uint32_t a = v.x & 0x7fffffffu, b = v.y & 0x7fffffffu, c = v.z & 0x7fffffffu, d = v.w & 0x7fffffffu;
        const uint32_t ab = a > b ? a : b, cd = c > d ? c : d;
        const uint32_t t = ab > cd ? ab : cd;
        m = m > t ? m : t;
    };
    int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
    for (; q + 3 * stride < quads; q += 4 * stride) {                  // four independent 16-byte loads in flight per thread
        const uint4 v0 = reinterpret_cast<const uint4*>(X)[q], v1 = reinterpret_cast<const uint4*>(X)[q + stride];
        const uint4 v2 = reinterpret_cast<const uint4*>(X)[q + 2 * stride], v3 = reinterpret_cast<const uint4*>(X)[q + 3 * stride];
        fold(v0); fold(v1); fold(v2); fold(v3);
    }
    for (; q < quads; q += stride) fold(reinterpret_cast<const uint4*>(X)[q]);
    if (blockIdx.x == 0 && threadIdx.x < (count & 3)) {
        const uint32_t t = __float_as_uint(X[(quads << 2) + threadIdx.x]) & 0x7fffffffu;
        m = m > t ? m : t;
    }
    __shared__ uint32_t block_max;
    if (threadIdx.x == 0) block_max = 0u;
    __syncthreads();
    if (m) atomicMax(&block_max, m);
    __syncthreads();
    if (threadIdx.x == 0 && block_max) atomicMax(out_bits, block_max);
}

// columns [0, cols) of X hold (fp16 hi, fp16 lo) words (egnn_linear_hl_f32 with split_cols): rewritten in place as the fp32 values
// hi + lo.  One thread per four words.
__global__ __launch_bounds__(256) void unsplit_words_kernel(float* __restrict__ X, int64_t ldx, int64_t rows, int cols)
{
    typedef _Float16 f16x2v __attribute__((ext_vector_type(2)));
    const int qpr = cols >> 2;                                          // quads per row
    const int64_t total = rows * qpr;
    for (int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x; o < total; o += (int64_t)gridDim.x * 256) {
        const int64_t r = o / qpr;
        const int c = (int)(o - r * qpr) * 4;
        uint4* p = reinterpret_cast<uint4*>(X + r * ldx + c);
        const uint4 v = *p;
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
        f32x4 f;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const f16x2v h = __builtin_bit_cast(f16x2v, w[u]);
            f[u] = (float)h[0] + (float)h[1];
        }
        *reinterpret_cast<f32x4*>(p) = f;
    }
}

}  // namespace

extern "C" int egnn_unsplit_words_f32(float* X, int64_t ldx, int64_t rows, int cols, void* stream)
{
    if (!X) return EGNN_E_NULLPTR;
    if (rows <= 0 || cols <= 0 || (cols % 4) != 0 || ldx < cols || (ldx % 4) != 0) return EGNN_E_SHAPE;
    if (reinterpret_cast<uintptr_t>(X) & 15) return EGNN_E_ALIGN;
    int64_t blocks = (rows * (cols >> 2) + 256 * 4 - 1) / (256 * 4);
    blocks = blocks < 1 ? 1 : (blocks > 8192 ? 8192 : blocks);
    hipLaunchKernelGGL(unsplit_words_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), X, ldx, rows, cols);
    return egnn_launch_status();
}

extern "C" int egnn_absmax_f32(const float* X, int64_t count, uint32_t* out_bits, void* stream)
{
    if (!X || !out_bits) return EGNN_E_NULLPTR;
    if (count <= 0) return EGNN_E_SHAPE;
    if (reinterpret_cast<uintptr_t>(X) & 15) return EGNN_E_ALIGN;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (hipMemsetAsync(out_bits, 0, sizeof(uint32_t), s) != hipSuccess) return (int)hipGetLastError();
    int64_t blocks = ((count >> 2) + 256 * 16 - 1) / (256 * 16);         // ~16 quads per thread
    blocks = blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks);
    hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)blocks), dim3(256), 0, s, X, count, out_bits);
    return egnn_launch_status();
}

extern "C" int egnn_split_scaled_f16(const float* X, int64_t ldx, int64_t rows, int cols, float scale, int transposed, void* hi, void* lo,
                                     int Kp, int32_t* status, void* stream)
{
    if (!X || !hi || !lo) return EGNN_E_NULLPTR;
    const int64_t k_extent = transposed ? rows : cols;                  // the K dimension of the image
    const int64_t img_rows = transposed ? cols : rows;
    if (rows <= 0 || cols <= 0 || ldx < cols || Kp < k_extent || (Kp % 32) != 0 || !(scale != 0.f) || !(fabsf(scale) < __builtin_inff()))
        return EGNN_E_SHAPE;                                                // (a negative factor -- the forward's -log2 e -- is fine)
    // the grid covers the padded image exactly: image rows up to a multiple of 32, K up to Kp (tiles of 64 x 32 or 32 x 64 of X)
    const int64_t rows_cover = transposed ? Kp : (img_rows + 31) / 32 * 32;
    const int64_t cols_cover = transposed ? (img_rows + 31) / 32 * 32 : Kp;
    const int64_t gy = (rows_cover + 63) / 64, gx = (cols_cover + 31) / 32;
    if (gy > 65535 * 16LL || gx > 0x7fffffffLL) return EGNN_E_UNSUPPORTED;
    if (gy > 65535) return EGNN_E_UNSUPPORTED;
    hipLaunchKernelGGL(split_scaled_kernel, dim3((unsigned)gx, (unsigned)gy), dim3(256), 0, static_cast<hipStream_t>(stream), X, ldx, rows, cols,
                       scale, transposed, static_cast<_Float16*>(hi), static_cast<_Float16*>(lo), Kp / 16, (img_rows + 31) / 32 * 32,
                       static_cast<_Float16*>(nullptr), static_cast<_Float16*>(nullptr), 0, (int64_t)0, status);
    return egnn_launch_status();
}

extern "C" int64_t egnn_split_scaled_colsum_rows(int64_t rows, int KpT)
{
    const int64_t rows32 = (rows + 31) / 32 * 32;
    return ((rows32 > KpT ? rows32 : KpT) + 63) / 64;
}

extern "C" int egnn_split_scaled_both_f16(const float* X, int64_t ldx, int64_t rows, int cols, float scale, void* hi, void* lo, int Kp,
                                          void* hiT, void* loT, int KpT, int32_t* status, float* colsum_parts, int64_t ld_colsum, void* stream)
{
    if (colsum_parts && ld_colsum < cols) return EGNN_E_SHAPE;
    if (!X || !hi || !lo || !hiT || !loT) return EGNN_E_NULLPTR;
    if (rows <= 0 || cols <= 0 || ldx < cols || Kp < cols || (Kp % 32) != 0 || KpT < rows || (KpT % 32) != 0 || !(scale > 0.f) || !(scale < __builtin_inff())) return EGNN_E_SHAPE;
    // the grid covers both padded images: X rows up to max(rows | 32, KpT), X columns up to max(Kp, cols | 32)
    const int64_t rows32 = (rows + 31) / 32 * 32, cols32 = ((int64_t)cols + 31) / 32 * 32;
    const int64_t rows_cover = rows32 > KpT ? rows32 : KpT, cols_cover = Kp > cols32 ? Kp : cols32;
    const int64_t gy = (rows_cover + 63) / 64, gx = (cols_cover + 31) / 32;
    if (gy > 65535 || gx > 0x7fffffffLL) return EGNN_E_UNSUPPORTED;
    hipLaunchKernelGGL(split_scaled_kernel, dim3((unsigned)gx, (unsigned)gy), dim3(256), 0, static_cast<hipStream_t>(stream), X, ldx, rows, cols,
                       scale, 2, static_cast<_Float16*>(hi), static_cast<_Float16*>(lo), Kp / 16, rows32,
                       static_cast<_Float16*>(hiT), static_cast<_Float16*>(loT), KpT / 16, cols32, status, colsum_parts, ld_colsum);
    return egnn_launch_status();
}

extern "C" int egnn_silu_bwd_drop_f32(const float* z, const float* g, float* a_out, float* gz_out, int64_t count, uint32_t* amax_bits,
                                      uint32_t drop_thr, uint32_t drop_seed, float drop_inv_keep, int64_t row0, int cols, void* stream);

extern "C" int egnn_silu_bwd_f32(const float* z, const float* g, float* a_out, float* gz_out, int64_t count, uint32_t* amax_bits, void* stream)
{
    return egnn_silu_bwd_drop_f32(z, g, a_out, gz_out, count, amax_bits, 0u, 0u, 1.f, 0, 4, stream);
}

extern "C" int egnn_silu_bwd_drop_f32(const float* z, const float* g, float* a_out, float* gz_out, int64_t count, uint32_t* amax_bits,
                                      uint32_t drop_thr, uint32_t drop_seed, float drop_inv_keep, int64_t row0, int cols, void* stream)
{
    if (drop_thr && (!(drop_inv_keep >= 1.f) || cols < 4 || (count % cols) != 0 || row0 < 0 || row0 + count / cols > 0xffffffffLL)) return EGNN_E_SHAPE;
    if (!z || !g || !a_out || !gz_out) return EGNN_E_NULLPTR;
    if (count <= 0 || (count % 4) != 0) return EGNN_E_SHAPE;
    if ((reinterpret_cast<uintptr_t>(z) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(a_out) | reinterpret_cast<uintptr_t>(gz_out)) & 15)
        return EGNN_E_ALIGN;
    int64_t blocks = (count / 4 + 256 * 4 - 1) / (256 * 4);
    blocks = blocks < 1 ? 1 : (blocks > 8192 ? 8192 : blocks);
    if (amax_bits && hipMemsetAsync(amax_bits, 0, 2 * sizeof(uint32_t), static_cast<hipStream_t>(stream)) != hipSuccess) return (int)hipGetLastError();
    hipLaunchKernelGGL(silu_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), z, g, a_out, gz_out, count / 4, amax_bits,
                       drop_thr, drop_seed, drop_inv_keep, row0, cols);
    return egnn_launch_status();
}

// internal: shared by egnn_node_prep_hl and egnn_split_f16
int egnn_pack_rows_launch(const float* X, int64_t ldx, const float* m_i, const float* gamma, const float* beta, float eps,
                          void* hi, void* lo, int Kp, void* raw_hi, void* raw_lo, int raw_Kp, int64_t rows, int dim, int m_dim,
                          int32_t* status, void* stream)
{
    const int64_t quads = ((rows + 31) / 32 * 32) / 4;
    int64_t blocks = (quads + 3) / 4;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(node_prep_hl_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), X, ldx, m_i,
                       gamma, beta, eps, static_cast<_Float16*>(hi), static_cast<_Float16*>(lo), Kp,
                       static_cast<_Float16*>(raw_hi), static_cast<_Float16*>(raw_lo), raw_Kp, rows, dim, m_dim, status);
    return egnn_launch_status();
}


extern "C" int egnn_node_prep_hl(const float* feats, const float* m_i, const float* gamma, const float* beta, float eps,
                                 void* out_hi, void* out_lo, int Kp, void* raw_hi, void* raw_lo, int raw_Kp,
                                 int64_t rows, int dim, int m_dim, int32_t* status, void* stream)
{
    if (!feats || !out_hi || !out_lo) return EGNN_E_NULLPTR;
    if ((gamma == nullptr) != (beta == nullptr)) return EGNN_E_NULLPTR;
    if ((raw_hi == nullptr) != (raw_lo == nullptr)) return EGNN_E_NULLPTR;
    if (rows <= 0 || dim <= 0 || m_dim < 0 || Kp < dim + m_dim || (Kp % 32) != 0) return EGNN_E_SHAPE;
    if (raw_hi && (raw_Kp < dim || (raw_Kp % 32) != 0)) return EGNN_E_SHAPE;
    return egnn_pack_rows_launch(feats, dim, m_i, gamma, beta, eps, out_hi, out_lo, Kp, raw_hi, raw_lo, raw_hi ? raw_Kp : 0,
                                 rows, dim, m_dim, status, stream);
}

// The range status words handed to the host without a copy engine and without a stream synchronisation: one thread copies them into
// pinned (host-coherent) memory and then writes a sequence number behind them; the host spins on the sequence number.
__global__ void status_publish_kernel(const int32_t* __restrict__ st, volatile int32_t* host, int32_t nwords, int32_t seq)
{
    for (int i = 0; i < nwords; ++i) host[i] = st[i];
    __threadfence_system();
    host[nwords] = seq;
}

extern "C" int egnn_status_publish(const int32_t* status_dev, int32_t* host_pinned, int nwords, int32_t seq, void* stream)
{
    if (!status_dev || !host_pinned) return EGNN_E_NULLPTR;
    if (nwords < 1 || nwords > 8) return EGNN_E_SHAPE;
    hipLaunchKernelGGL(status_publish_kernel, dim3(1), dim3(1), 0, static_cast<hipStream_t>(stream), status_dev, host_pinned, nwords, seq);
    return egnn_launch_status();
}

extern "C" int egnn_abi_version(void) { return EGNN_ABI_VERSION; }

extern "C" int64_t egnn_struct_bytes(int which)
{
    switch (which) {
    case 0: return (int64_t)sizeof(egnn_edge_args);
    case 1: return (int64_t)sizeof(egnn_edge_bwd_args);
    case 2: return (int64_t)sizeof(egnn_edge_tail_args);
    case 3: return (int64_t)sizeof(egnn_layer_desc);
    case 4: return (int64_t)sizeof(egnn_packed_info);
    case 5: return (int64_t)sizeof(egnn_edge_exact_args);
    case 6: return (int64_t)sizeof(egnn_edge_exact_bwd_args);
    case 7: return (int64_t)sizeof(egnn_edge_tail_exact_args);
    case 8: return (int64_t)sizeof(egnn_forward_opts);
    default: return -1;
    }
}

extern "C" const char* egnn_error_string(int code)
{
    switch (code) {
        case EGNN_OK: return "ok";
        case EGNN_E_NULLPTR: return "a required pointer is NULL";
        case EGNN_E_SHAPE: return "non-positive or inconsistent sizes";
        case EGNN_E_UNSUPPORTED: return "shape outside what the gfx950 kernels are built for";
        case EGNN_E_ALIGN: return "pointer or leading dimension not 16-byte aligned";
        case EGNN_E_K_GT_N: return "selected index k out of range (K > N)";
        default: return code > 0 ? "HIP runtime error (hipError_t)" : "unknown error";
    }
}
