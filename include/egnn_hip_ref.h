/* egnn_hip_ref.h -- TEST-ONLY companion of egnn_hip.h: a reference implementation the shipped library does not contain.
 *
 * Built into tests/libegnn_hip_ref.so by egnn_pytorch_amd/csrc/build.sh and loaded only by tests/_reflib.py (A/B checks of
 * the production kernels: operands split on the fly).  Nothing under
 * egnn_pytorch_amd/ binds these symbols.  Same conventions as egnn_hip.h (device pointers, stream, return codes).
 */
#ifndef EGNN_HIP_REF_H
#define EGNN_HIP_REF_H

#include "egnn_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------------------------------------------------------------------------------------
 * The same operation on the matrix cores with A split on the fly (predecessor of egnn_linear_hl_f32): fp32 in / fp32 out, every product evaluated as
 * a 3-term split-f16 product with fp32 accumulation on v_mfma_f32_32x32x16_f16
 * (a = a_hi + a_lo, w = w_hi + w_lo;  a_hi w_hi + a_lo w_hi + a_hi w_lo;  dropped term <= 2^-22 |a w|: fp32-class
 * accuracy at 3/16 of the f32-MFMA cost -- on gfx950 the f32-input MFMA runs at vector rate on the vector datapath).
 *   W_hi, W_lo: (Np, ldw) fp16 images of w_scale * W, split on the host (egnn_pytorch_amd/_weights.py::split_f16):
 *               Np = N rounded up to 128, ldw = K rounded up to 32, zero padded; w_inv_scale = 1 / w_scale
 *               (a power of two that brings max|W| into [1,2)).
 *   A is split on the fly; requires |A| < 65504.  Other arguments as egnn_linear_f32.
 */
int egnn_linear_split_f32(const float* A, int64_t lda, const void* W_hi, const void* W_lo, int64_t ldw,
                          float w_inv_scale, const float* bias, const float* residual, int64_t ldr,
                          float* C, int64_t ldc, int64_t M, int N, int K, int act, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* EGNN_HIP_REF_H */
