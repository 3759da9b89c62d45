// Deterministic row gather-sum: out[r, :] = sum over p in [seg_ptr[r], seg_ptr[r+1]) of rows[order[p], :], in that order.
//
// The backward of the neighbour gather (egnn_pytorch.py:275, feats_j = batched_index_select(feats, nbhd_indices)):
// d loss / d P_j[j] is the sum of dz over every edge (i, k) whose neighbour is j.  `order` lists the edges sorted by
// destination (stable, so ties stay in edge order) -- the transposed neighbour list -- which makes the sum a fixed-order
// read-only reduction: no float atomics, bit-reproducible.  One workgroup per destination row; every wave instruction reads
// 1 KB of one source row; HBM-bound (each source row is read exactly once).
#include "egnn_common.h"

namespace {

__global__ __launch_bounds__(256) void rows_gather_sum_kernel(const float* __restrict__ rows, int64_t ld, const int64_t* __restrict__ order,
                                                              const int64_t* __restrict__ seg_ptr, int cols, float* __restrict__ out,
                                                              int64_t ldo)
{
    const int64_t r = blockIdx.x;
    const int64_t p0 = seg_ptr[r], p1 = seg_ptr[r + 1];
    for (int c = threadIdx.x * 4; c < cols; c += 256 * 4) {
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
        int64_t p = p0;
        for (; p + 1 < p1; p += 2) {                                   // two rows in flight
            const f32x4 a = *reinterpret_cast<const f32x4*>(rows + order[p] * ld + c);
            const f32x4 b = *reinterpret_cast<const f32x4*>(rows + order[p + 1] * ld + c);
            acc += a;
            acc += b;
        }
        if (p < p1) acc += *reinterpret_cast<const f32x4*>(rows + order[p] * ld + c);
        *reinterpret_cast<f32x4*>(out + r * ldo + c) = acc;
    }
}

}  // namespace

extern "C" int egnn_rows_gather_sum_f32(const float* rows, int64_t ld, const int64_t* order, const int64_t* seg_ptr, int64_t n_out,
                                        int cols, float* out, int64_t ldo, void* stream)
{
    if (!rows || !order || !seg_ptr || !out) return EGNN_E_NULLPTR;
    if (n_out <= 0 || cols <= 0 || (cols % 4) != 0 || ld < cols || ldo < cols || (ld % 4) != 0 || (ldo % 4) != 0) return EGNN_E_SHAPE;
    if (n_out > 0x7fffffffLL) return EGNN_E_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(rows) & 15) || (reinterpret_cast<uintptr_t>(out) & 15)) return EGNN_E_ALIGN;
    hipLaunchKernelGGL(rows_gather_sum_kernel, dim3((unsigned)n_out), dim3(256), 0, static_cast<hipStream_t>(stream), rows, ld, order,
                       seg_ptr, cols, out, ldo);
    return egnn_launch_status();
}
