// Whole-layer C interface (include/egnn_hip.h, "Whole-layer interface"): the weight re-layout as a host function, workspace
// sizing, and one EGNN.forward (egnn_pytorch/egnn_pytorch.py:224-341) as a chain of the library's own launches.
//
// egnn_pack_weights_host restates egnn_pytorch_amd/_weights.py::pack operation for operation (same fp32 products, same
// round-to-nearest fp16 conversions), so the blob is bit-identical to what the Python module builds on the device
// (tests/test_host_logic.py::test_c_weight_packer_matches_python); egnn_layer_forward_f32 mirrors
// egnn_pytorch_amd/layer.py::_forward_hip (tests/test_gpu_parity.py::test_c_layer_forward_matches_module).
#include "egnn_common.h"

#include <cmath>
#include <cstring>
#include <vector>

namespace {

constexpr float NEG_LOG2E = -1.4426950408889634f;
constexpr float NEG_LN2 = -0.6931471805599453f;
constexpr int M_MAX = 64;         // largest m_dim the edge kernel is instantiated for
constexpr int S_MAX = 16;

inline int pad32(int x) { return (x + 31) / 32 * 32; }
inline size_t align256(size_t x) { return (x + 255) / 256 * 256; }

// the power of two that brings amax into [1, 2) (1 for 0 / non-finite): _weights.py::pow2_scale
inline float pow2_scale(float amax)
{
    if (!(amax > 0.f) || !std::isfinite(amax)) return 1.f;
    int e;
    std::frexp(amax, &e);                       // amax = m * 2^e, m in [0.5, 1)  ->  floor(log2 amax) = e - 1
    return std::ldexp(1.f, 1 - e);
}

struct Dims {
    int dim, m, F, edge_dim, din, H, Hp, S, NM, kp_dim, kp_node, kp_hid;
    int NB, M_PAD, C_PAD;          // 16-channel blocks of m_dim (1, 2 or 4); padded channels 16 NB; coors_mlp hidden 64 NB
    bool ok;
};

Dims dims_of(const egnn_layer_desc* d)
{
    Dims x{};
    if (!d || d->dim <= 0 || d->m_dim < 1 || d->m_dim > M_MAX || d->edge_dim < 0 || d->fourier_features < 0) return x;
    x.dim = d->dim; x.m = d->m_dim; x.F = d->fourier_features; x.edge_dim = d->edge_dim;
    x.S = 2 * x.F + 1 + x.edge_dim;
    if (x.S > S_MAX) return x;
    x.din = 2 * x.dim + x.S;
    x.H = 2 * x.din;
    x.Hp = egnn_padded_hidden(x.H);
    x.NM = egnn_edge_mfmas(x.S);
    x.NB = x.m <= 16 ? 1 : (x.m <= 32 ? 2 : 4);
    x.M_PAD = 16 * x.NB;
    x.C_PAD = 64 * x.NB;
    x.kp_dim = pad32(x.dim);
    x.kp_node = pad32(x.dim + x.m);
    x.kp_hid = pad32(2 * x.dim);
    x.ok = true;
    return x;
}

// byte offsets of every piece of the blob (sizes depend on the descriptor only)
void layout(const egnn_layer_desc* d, const Dims& x, egnn_packed_info* info)
{
    std::memset(info, 0, sizeof(*info));
    info->H = x.H; info->Hp = x.Hp; info->S = x.S; info->NM = x.NM;
    info->wcat_rows = (2 * x.Hp + 255) / 256 * 256;
    info->w5_rows = (2 * x.dim + 255) / 256 * 256;
    info->w6_rows = (x.dim + 255) / 256 * 256;
    size_t off = 0;
    auto take = [&](uint64_t& field, size_t bytes) { field = off; off = align256(off + bytes); };
    take(info->wcat_hi, (size_t)info->wcat_rows * x.kp_dim * 2);
    take(info->wcat_lo, (size_t)info->wcat_rows * x.kp_dim * 2);
    take(info->bcat, (size_t)2 * x.Hp * 4);
    take(info->wst, (size_t)x.Hp * 4 * x.NM * 2 * 2);
    take(info->w2h, (size_t)(x.Hp / 32) * x.NB * 2 * 64 * 8 * 2);
    take(info->b2, (size_t)x.M_PAD * 4);
    if (d->soft_edges) { take(info->gate_w, (size_t)x.M_PAD * 4); take(info->gate_b, 4); }
    if (d->update_coors) {
        take(info->w3h, (size_t)2 * x.C_PAD * x.M_PAD * 2);
        take(info->b3, (size_t)x.C_PAD * 4);
        take(info->w4, (size_t)x.C_PAD * 4);
        take(info->b4, 4);
    }
    if (d->norm_coors) take(info->coors_scale, 4);
    if (d->update_feats) {
        take(info->w5_hi, (size_t)info->w5_rows * x.kp_node * 2);
        take(info->w5_lo, (size_t)info->w5_rows * x.kp_node * 2);
        take(info->b5, (size_t)2 * x.dim * 4);
        take(info->w6_hi, (size_t)info->w6_rows * x.kp_hid * 2);
        take(info->w6_lo, (size_t)info->w6_rows * x.kp_hid * 2);
        take(info->b6, (size_t)x.dim * 4);
        if (d->norm_feats) { take(info->gamma, (size_t)x.dim * 4); take(info->beta, (size_t)x.dim * 4); }
    }
    info->bytes = off;
}

// _weights.py::split_f16: (n x k) fp32 (row stride ld) -> packed tile-major fp16 images of scale * W, zero padded to
// (rows_p x kp); returns 1 / scale
float split_packed(const float* w, int n, int k, int64_t ld, int rows_p, int kp, _Float16* hi, _Float16* lo)
{
    float amax = 0.f;
    for (int r = 0; r < n; ++r)
        for (int c = 0; c < k; ++c) amax = std::fmax(amax, std::fabs(w[r * ld + c]));
    const float scale = pow2_scale(amax);
    std::memset(hi, 0, (size_t)rows_p * kp * 2);
    std::memset(lo, 0, (size_t)rows_p * kp * 2);
    const int nkt = kp / 16;
    for (int r = 0; r < n; ++r)
        for (int c = 0; c < k; ++c) {
            const float v = w[r * ld + c] * scale;
            const _Float16 h = (_Float16)v;
            const size_t o = egnn_pk_off(r, c, nkt);
            hi[o] = h;
            lo[o] = (_Float16)(v - (float)h);
        }
    return 1.f / scale;
}

}  // namespace

extern "C" int egnn_packed_layout(const egnn_layer_desc* desc, egnn_packed_info* info)
{
    if (!desc || !info) return EGNN_E_NULLPTR;
    const Dims x = dims_of(desc);
    if (!x.ok) return EGNN_E_UNSUPPORTED;
    layout(desc, x, info);
    return EGNN_OK;
}

extern "C" size_t egnn_packed_weights_bytes(const egnn_layer_desc* desc)
{
    const Dims x = dims_of(desc);
    if (!x.ok) return 0;
    egnn_packed_info info;
    layout(desc, x, &info);
    return (size_t)info.bytes;
}

extern "C" int egnn_pack_weights_host(const egnn_layer_desc* desc, const egnn_layer_params* p, void* blob, egnn_packed_info* info)
{
    if (!desc || !p || !blob || !info) return EGNN_E_NULLPTR;
    const Dims x = dims_of(desc);
    if (!x.ok) return EGNN_E_UNSUPPORTED;
    if (!p->edge_mlp_0_weight || !p->edge_mlp_0_bias || !p->edge_mlp_3_weight || !p->edge_mlp_3_bias) return EGNN_E_NULLPTR;
    if (desc->soft_edges && (!p->edge_gate_0_weight || !p->edge_gate_0_bias)) return EGNN_E_NULLPTR;
    if (desc->update_coors && (!p->coors_mlp_0_weight || !p->coors_mlp_0_bias || !p->coors_mlp_3_weight || !p->coors_mlp_3_bias))
        return EGNN_E_NULLPTR;
    if (desc->norm_coors && !p->coors_norm_scale) return EGNN_E_NULLPTR;
    if (desc->update_feats && (!p->node_mlp_0_weight || !p->node_mlp_0_bias || !p->node_mlp_3_weight || !p->node_mlp_3_bias))
        return EGNN_E_NULLPTR;
    if (desc->update_feats && desc->norm_feats && (!p->node_norm_weight || !p->node_norm_bias)) return EGNN_E_NULLPTR;
    if (!desc->update_feats && !desc->update_coors) return EGNN_E_SHAPE;      // the reference asserts this too (:171)
    layout(desc, x, info);
    char* base = static_cast<char*>(blob);
    std::memset(base, 0, (size_t)info->bytes);
    const int dim = x.dim, H = x.H, Hp = x.Hp, S = x.S, m = x.m, din = x.din;
    const float* w1 = p->edge_mlp_0_weight;                                   // (H, din): [h_i | h_j | scalars]

    // ---- node-level projection weights: Wcat (2 Hp, dim): rows [0, H) = -log2e W_i, rows [Hp, Hp + H) = -log2e W_j
    {
        std::vector<float> wcat((size_t)2 * Hp * dim, 0.f);
        for (int h = 0; h < H; ++h)
            for (int c = 0; c < dim; ++c) {
                wcat[(size_t)h * dim + c] = w1[(size_t)h * din + c] * NEG_LOG2E;
                wcat[(size_t)(Hp + h) * dim + c] = w1[(size_t)h * din + dim + c] * NEG_LOG2E;
            }
        info->wcat_inv_scale = split_packed(wcat.data(), 2 * Hp, dim, dim, info->wcat_rows, x.kp_dim,
                                            reinterpret_cast<_Float16*>(base + info->wcat_hi), reinterpret_cast<_Float16*>(base + info->wcat_lo));
        float* bcat = reinterpret_cast<float*>(base + info->bcat);
        for (int h = 0; h < H; ++h) bcat[h] = p->edge_mlp_0_bias[h] * NEG_LOG2E;
    }
    // ---- per-edge scalar columns as first-layer MFMA A fragments: _weights.py::scalar_table
    {
        float amax = 0.f;
        for (int h = 0; h < H; ++h)
            for (int s = 0; s < S; ++s) amax = std::fmax(amax, std::fabs(w1[(size_t)h * din + 2 * dim + s] * NEG_LOG2E));
        const float c = pow2_scale(amax);
        info->ws_inv_scale = 1.f / c;
        _Float16* tab = reinterpret_cast<_Float16*>(base + info->wst);     // (Hp, 4 NM, 2)
        const int terms = 4 * x.NM;
        for (int h = 0; h < H; ++h)
            for (int s = 0; s < S; ++s) {
                const float w = (w1[(size_t)h * din + 2 * dim + s] * NEG_LOG2E) * c;
                const float wa = w * 1024.0f;
                const _Float16 hi = (_Float16)w, ahi = (_Float16)wa;
                const _Float16 lo = (_Float16)(w - (float)hi), alo = (_Float16)(wa - (float)ahi);
                // term index ti = 3 s + kind sits at position (ti & ~3) | ((ti & 3) ^ sw): units 8 .. 15 of a 16-block keep the pairs
                // (0, 1) and (2, 3) of every four-term group swapped (the kernels' conflict-free LDS read, _weights.scalar_table)
                const int sw = (h & 8) ? 2 : 0;
                auto at = [&](int ti) { return tab + ((size_t)h * terms + ((ti & ~3) | ((ti & 3) ^ sw))) * 2; };
                _Float16* t0 = at(3 * s), *t1 = at(3 * s + 1), *t2 = at(3 * s + 2);
                t0[0] = ahi; t0[1] = alo;                                     // kind 0: (hi, lo) of 2^10 c W
                t1[0] = hi; t1[1] = lo;                                       // kind 1: (hi, lo) of c W
                t2[0] = hi; t2[1] = (_Float16)0.f;                            // kind 2: (hi, 0)  of c W
            }
    }
    // ---- second Linear of edge_mlp in v_mfma_f32_16x16x32_f16 fragment order
    {
        float amax = 0.f;
        for (int c = 0; c < m; ++c)
            for (int h = 0; h < H; ++h) amax = std::fmax(amax, std::fabs(p->edge_mlp_3_weight[(size_t)c * H + h] * NEG_LN2));
        const float scale = pow2_scale(amax);
        info->w2_inv_scale = 1.f / scale;
        _Float16* w2h = reinterpret_cast<_Float16*>(base + info->w2h);     // (Hp/32, NB, 2, 64, 8)
        for (int c = 0; c < m; ++c)
            for (int h = 0; h < H; ++h) {
                const float v = (p->edge_mlp_3_weight[(size_t)c * H + h] * NEG_LN2) * scale;
                const _Float16 hi = (_Float16)v, lo = (_Float16)(v - (float)hi);
                const int step = h / 32, hb = (h % 32) / 16, g = (h % 16) / 4, r = h % 4;
                const size_t o = ((((size_t)step * x.NB + c / 16) * 2 + 0) * 64 + (16 * g + c % 16)) * 8 + (4 * hb + r);
                w2h[o] = hi;
                w2h[o + 64 * 8] = lo;
            }
        float* b2 = reinterpret_cast<float*>(base + info->b2);
        for (int c = 0; c < m; ++c) b2[c] = p->edge_mlp_3_bias[c];
    }
    if (desc->soft_edges) {
        float* gw = reinterpret_cast<float*>(base + info->gate_w);
        for (int c = 0; c < m; ++c) gw[c] = p->edge_gate_0_weight[c];
        *reinterpret_cast<float*>(base + info->gate_b) = p->edge_gate_0_bias[0];
    }
    if (desc->update_coors) {
        float amax = 0.f;
        for (int j = 0; j < 4 * m; ++j)
            for (int c = 0; c < m; ++c) amax = std::fmax(amax, std::fabs(p->coors_mlp_0_weight[(size_t)j * m + c]));
        const float scale = pow2_scale(amax);
        info->w3_inv_scale = 1.f / scale;
        _Float16* w3h = reinterpret_cast<_Float16*>(base + info->w3h);     // (2, 64 NB, 16 NB): hi image | lo image
        const int M_PAD = x.M_PAD, C_PAD = x.C_PAD;
        for (int j = 0; j < 4 * m; ++j)
            for (int c = 0; c < m; ++c) {
                const float v = p->coors_mlp_0_weight[(size_t)j * m + c] * scale;
                const _Float16 hi = (_Float16)v;
                w3h[(size_t)j * M_PAD + c] = hi;
                w3h[(size_t)C_PAD * M_PAD + (size_t)j * M_PAD + c] = (_Float16)(v - (float)hi);
            }
        float* b3 = reinterpret_cast<float*>(base + info->b3);
        float* w4 = reinterpret_cast<float*>(base + info->w4);
        for (int j = 0; j < 4 * m; ++j) { b3[j] = p->coors_mlp_0_bias[j]; w4[j] = p->coors_mlp_3_weight[j]; }
        *reinterpret_cast<float*>(base + info->b4) = p->coors_mlp_3_bias[0];
    }
    if (desc->norm_coors) *reinterpret_cast<float*>(base + info->coors_scale) = p->coors_norm_scale[0];
    if (desc->update_feats) {
        info->w5_inv_scale = split_packed(p->node_mlp_0_weight, 2 * dim, dim + m, dim + m, info->w5_rows, x.kp_node,
                                          reinterpret_cast<_Float16*>(base + info->w5_hi), reinterpret_cast<_Float16*>(base + info->w5_lo));
        info->w6_inv_scale = split_packed(p->node_mlp_3_weight, dim, 2 * dim, 2 * dim, info->w6_rows, x.kp_hid,
                                          reinterpret_cast<_Float16*>(base + info->w6_hi), reinterpret_cast<_Float16*>(base + info->w6_lo));
        std::memcpy(base + info->b5, p->node_mlp_0_bias, (size_t)2 * dim * 4);
        std::memcpy(base + info->b6, p->node_mlp_3_bias, (size_t)dim * 4);
        if (desc->norm_feats) {
            std::memcpy(base + info->gamma, p->node_norm_weight, (size_t)dim * 4);
            std::memcpy(base + info->beta, p->node_norm_bias, (size_t)dim * 4);
        }
    }
    return EGNN_OK;
}

namespace {

struct Workspace {
    size_t idx, rank, order, slots, raw_hi, raw_lo, node_hi, node_lo, proj, hid_hi, hid_lo, nmf_img, bytes;
};

Workspace carve(const egnn_layer_desc* d, const Dims& x, int64_t B, int64_t N, int64_t K)
{
    Workspace w{};
    size_t off = 0;
    auto take = [&](size_t& field, size_t bytes) { field = off; off = align256(off + bytes); };
    const int64_t rows = B * N;
    const bool nearest = d->num_nearest_neighbors > 0 || d->only_sparse_neighbors;
    if (nearest) { take(w.idx, (size_t)rows * K * 4); take(w.rank, (size_t)rows * K * 4); take(w.slots, (size_t)rows * K * 16); }
    take(w.order, (size_t)rows * 4);
    take(w.raw_hi, (size_t)egnn_packed_halves(rows, x.kp_dim) * 2);
    take(w.raw_lo, (size_t)egnn_packed_halves(rows, x.kp_dim) * 2);
    take(w.proj, (size_t)rows * 2 * x.Hp * 4);
    if (d->update_feats) {
        take(w.node_hi, (size_t)egnn_packed_halves(rows, x.kp_node) * 2);
        take(w.node_lo, (size_t)egnn_packed_halves(rows, x.kp_node) * 2);
        if (egnn_node_mlp_fused_halves(x.dim, x.m) > 0) {
            // narrow layers: node_mlp in one launch (csrc/node_mlp_fused.hip) -- no hidden image, the fused weight image instead
            take(w.nmf_img, (size_t)egnn_node_mlp_fused_halves(x.dim, x.m) * 2);
        } else {
            take(w.hid_hi, (size_t)egnn_packed_halves(rows, x.kp_hid) * 2);
            take(w.hid_lo, (size_t)egnn_packed_halves(rows, x.kp_hid) * 2);
        }
    }
    w.bytes = off;
    return w;
}

}  // namespace

extern "C" size_t egnn_workspace_bytes(const egnn_layer_desc* desc, int B, int N, int K)
{
    const Dims x = dims_of(desc);
    if (!x.ok || B <= 0 || N <= 0 || K < 0) return 0;
    return carve(desc, x, B, N, K).bytes;
}

#define EGNN_TRY(call) do { const int rc__ = (call); if (rc__ != EGNN_OK) return rc__; } while (0)

extern "C" int egnn_layer_forward_f32(const egnn_layer_desc* desc, const egnn_packed_info* info, const void* blob_dev,
                                      const float* feats, const float* coors, const float* edges, const uint8_t* mask,
                                      const uint8_t* adj, int64_t adj_batch_stride, int B, int N, int K, int coor_dim,
                                      float* feats_out, float* coors_out, void* workspace, size_t workspace_bytes,
                                      int32_t* status, void* stream)
{
    return egnn_layer_forward_opts_f32(desc, info, blob_dev, feats, coors, edges, mask, adj, adj_batch_stride, B, N, K, coor_dim, feats_out,
                                       coors_out, workspace, workspace_bytes, status, stream, nullptr);
}

#define EGNN_HIP_TRY(call) do { const hipError_t e__ = (call); if (e__ != hipSuccess) return (int)e__; } while (0)

extern "C" int egnn_layer_forward_opts_f32(const egnn_layer_desc* desc, const egnn_packed_info* info, const void* blob_dev,
                                           const float* feats, const float* coors, const float* edges, const uint8_t* mask,
                                           const uint8_t* adj, int64_t adj_batch_stride, int B, int N, int K, int coor_dim,
                                           float* feats_out, float* coors_out, void* workspace, size_t workspace_bytes,
                                           int32_t* status, void* stream, const egnn_forward_opts* opts)
{
    if (!desc || !info || !blob_dev || !feats || !coors || !feats_out || !coors_out || !workspace) return EGNN_E_NULLPTR;
    // the neighbour selection on a second stream (opts->side_stream): it reads the coordinates only, so it forks at THIS call's entry
    // (ev_fork, recorded before the node-level launches) and is joined in front of the edge pass (ev_join)
    hipStream_t side = opts ? static_cast<hipStream_t>(opts->side_stream) : nullptr;
    if (side && (!opts->ev_fork || !opts->ev_join)) return EGNN_E_NULLPTR;
    const Dims x = dims_of(desc);
    if (!x.ok) return EGNN_E_UNSUPPORTED;
    if (B <= 0 || N <= 0 || K < 0 || coor_dim < 1 || coor_dim > 8) return EGNN_E_SHAPE;
    if ((edges != nullptr) != (x.edge_dim > 0)) return EGNN_E_SHAPE;
    if (info->H != x.H || info->Hp != x.Hp || info->S != x.S) return EGNN_E_SHAPE;
    const bool nearest = desc->num_nearest_neighbors > 0 || desc->only_sparse_neighbors;
    if (!nearest && K != N) return EGNN_E_SHAPE;                               // dense all-pairs path
    if (K > N) return EGNN_E_K_GT_N;                                           // torch.topk's error upstream (:258)
    const Workspace w = carve(desc, x, B, N, K);
    if (workspace_bytes < w.bytes) return EGNN_E_SHAPE;
    if (reinterpret_cast<uintptr_t>(workspace) & 255) return EGNN_E_ALIGN;
    hipStream_t s = static_cast<hipStream_t>(stream);
    char* ws = static_cast<char*>(workspace);
    const char* blob = static_cast<const char*>(blob_dev);
    auto F = [&](uint64_t off) { return reinterpret_cast<const float*>(blob + off); };
    const int64_t rows = (int64_t)B * N;
    const int dim = x.dim;

    // outputs default to the inputs (update_feats / update_coors False, or no edges at all)
    if (!desc->update_feats && feats_out != feats)
        if (hipMemcpyAsync(feats_out, feats, (size_t)rows * dim * 4, hipMemcpyDeviceToDevice, s) != hipSuccess) return (int)hipGetLastError();
    if ((!desc->update_coors || K == 0) && coors_out != coors)
        if (hipMemcpyAsync(coors_out, coors, (size_t)rows * coor_dim * 4, hipMemcpyDeviceToDevice, s) != hipSuccess) return (int)hipGetLastError();

    // ---- neighbour selection (:230-260)
    int32_t* idx = nullptr;
    float* rank = nullptr;
    float valid_radius = desc->valid_radius;
    if (nearest) {
        if (adj && desc->only_sparse_neighbors) valid_radius = 0.f;            // (:250)
        if (K > 0) {
            idx = reinterpret_cast<int32_t*>(ws + w.idx);
            rank = reinterpret_cast<float*>(ws + w.rank);
        }
    }
    const float* gamma = desc->update_feats && desc->norm_feats ? F(info->gamma) : nullptr;
    const float* beta = desc->update_feats && desc->norm_feats ? F(info->beta) : nullptr;
    void *node_hi = desc->update_feats ? ws + w.node_hi : nullptr, *node_lo = desc->update_feats ? ws + w.node_lo : nullptr;

    if (K > 0) {
        if (side && idx) EGNN_HIP_TRY(hipEventRecord(static_cast<hipEvent_t>(opts->ev_fork), s));
        // ---- operand prep + node-level projections P = feats [W_i ; W_j]^T + [b1 ; 0]
        // (node_norm = Identity, the reference's default: [feats | 0] for node_mlp and feats for the projection hold the same values --
        // one packed image, the projection contracts over its first kp_dim columns: egnn_linear_hl_lda_f32)
        const bool shared = desc->update_feats && !gamma;
        if (shared)
            EGNN_TRY(egnn_node_prep_hl(feats, nullptr, nullptr, nullptr, desc->ln_eps, node_hi, node_lo, x.kp_node, nullptr, nullptr, 0,
                                       rows, dim, x.m, status, stream));
        else if (desc->update_feats)
            EGNN_TRY(egnn_node_prep_hl(feats, nullptr, gamma, beta, desc->ln_eps, node_hi, node_lo, x.kp_node, ws + w.raw_hi, ws + w.raw_lo,
                                       x.kp_dim, rows, dim, x.m, status, stream));
        else
            EGNN_TRY(egnn_split_f16(feats, dim, rows, dim, ws + w.raw_hi, ws + w.raw_lo, x.kp_dim, status, stream));
        const int pi_split = K >= 6;
        float* proj = reinterpret_cast<float*>(ws + w.proj);
        // (mask: a padded node's rows of P are read by masked-out edges only -- M-tiles of padded nodes are not computed; only where the
        // wave-per-node edge kernel runs: egnn_edge_pw_covers)
        const uint8_t* row_mask = (mask && idx && coor_dim == 3 && pi_split &&
                                   egnn_edge_pw_covers(B, N, K, x.S, x.F, x.edge_dim, x.m, coor_dim, 2 * (int64_t)x.Hp)) ? mask : nullptr;
        if (shared)
            EGNN_TRY(egnn_linear_hl_lda_rows_f32(node_hi, node_lo, x.kp_node, blob + info->wcat_hi, blob + info->wcat_lo, info->wcat_inv_scale,
                                                 F(info->bcat), nullptr, 0, proj, 2 * x.Hp, nullptr, nullptr, 0, rows, 2 * x.Hp, x.kp_dim,
                                                 info->wcat_rows, 0, pi_split ? x.Hp : 0, row_mask, status, stream));
        else
            EGNN_TRY(egnn_linear_hl_lda_rows_f32(ws + w.raw_hi, ws + w.raw_lo, 0, blob + info->wcat_hi, blob + info->wcat_lo,
                                                 info->wcat_inv_scale, F(info->bcat), nullptr, 0, proj, 2 * x.Hp, nullptr, nullptr, 0, rows,
                                                 2 * x.Hp, x.kp_dim, info->wcat_rows, 0, pi_split ? x.Hp : 0, row_mask, status, stream));
        egnn_edge_args a;
        std::memset(&a, 0, sizeof(a));
        a.B = B; a.N = N; a.K = K; a.dim = dim; a.m_dim = x.m; a.H = x.H; a.Hp = x.Hp;
        a.fourier = x.F; a.edge_dim = x.edge_dim; a.S = x.S; a.pi_split = pi_split;
        a.Pi = proj; a.Pj = proj + x.Hp; a.ldp = 2 * x.Hp;
        a.Wst = blob + info->wst; a.wst_terms = 4 * x.NM; a.ws_inv_scale = info->ws_inv_scale;
        a.W2h = blob + info->w2h; a.w2_inv_scale = info->w2_inv_scale; a.b2 = F(info->b2);
        if (desc->soft_edges) { a.gate_w = F(info->gate_w); a.gate_b = F(info->gate_b); }
        if (desc->update_coors) {
            a.W3h = blob + info->w3h; a.w3_inv_scale = info->w3_inv_scale;
            a.b3 = F(info->b3); a.W4 = F(info->w4); a.b4 = F(info->b4);
            a.coors_out = coors_out;
        }
        if (desc->norm_coors) a.coors_scale = F(info->coors_scale);
        a.coors = coors; a.coor_dim = coor_dim; a.edges = edges; a.mask = mask; a.idx = idx; a.rank = rank;
        // ---- neighbour selection (:230-260), enqueued BEHIND the node-level launches (with a side stream it runs beside them: it
        // waits for the entry event only) -- the first kernel of the forward starts earlier, and nothing it writes is read before the
        // edge pass
        void* sel_stream = (side && idx) ? static_cast<void*>(side) : stream;
        if (side && idx) EGNN_HIP_TRY(hipStreamWaitEvent(side, static_cast<hipEvent_t>(opts->ev_fork), 0));
        if (idx) EGNN_TRY(egnn_knn_select_f32(coors, mask, adj, adj_batch_stride, B, N, K, coor_dim, idx, rank, sel_stream));
        if (idx && !adj && N >= 64 && N <= 4096 && coor_dim == 3) {            // scheduling aid only
            int32_t* order = (opts && opts->order) ? opts->order : reinterpret_cast<int32_t*>(ws + w.order);
            if (!(opts && opts->order && opts->order_is_hint))                 // (a stack of layers reuses the first layer's order)
                EGNN_TRY(egnn_spatial_order_masked_f32(coors, mask, B, N, order, sel_stream));
            a.order = order;
        }
        if (idx && coor_dim == 3) {                                             // the setup's index chain, flattened (egnn_slot_prep_f32)
            EGNN_TRY(egnn_slot_prep_f32(coors, mask, idx, rank, a.order, valid_radius < 3.0e38f ? valid_radius : 3.0e38f, B, N, K,
                                        ws + w.slots, sel_stream));
            a.slots = ws + w.slots;
        }
        if (side && idx) {
            EGNN_HIP_TRY(hipEventRecord(static_cast<hipEvent_t>(opts->ev_join), side));
            EGNN_HIP_TRY(hipStreamWaitEvent(s, static_cast<hipEvent_t>(opts->ev_join), 0));
        }
        a.valid_radius = valid_radius < 3.0e38f ? valid_radius : 3.0e38f;
        a.clamp = desc->coor_weights_clamp_value < 0.f ? -1.f : desc->coor_weights_clamp_value;
        a.pool_mean = desc->pool_mean;
        a.node_hi = node_hi; a.node_lo = node_lo; a.node_kp = desc->update_feats ? x.kp_node : 0;
        a.status = status;
        EGNN_TRY(egnn_edge_fused_f32(&a, stream));
    } else if (desc->update_feats) {                                          // K == 0: no messages, m_i = 0
        EGNN_TRY(egnn_node_prep_hl(feats, nullptr, gamma, beta, desc->ln_eps, node_hi, node_lo, x.kp_node, nullptr, nullptr, 0,
                                   rows, dim, x.m, status, stream));
    }

    // ---- node update (:335-337)
    if (desc->update_feats && egnn_node_mlp_fused_halves(dim, x.m) > 0) {
        // (the Python module packs the fused image once per parameter version; this entry keeps no state between calls and re-derives
        // it -- ~1 MB -- from the blob's two packed images)
        const void* img = (opts && opts->nmf_img) ? opts->nmf_img : ws + w.nmf_img;
        if (!(opts && opts->nmf_img))
            EGNN_TRY(egnn_node_mlp_fused_pack_f16(blob + info->w5_hi, blob + info->w5_lo, blob + info->w6_hi, blob + info->w6_lo, dim, x.m,
                                                  ws + w.nmf_img, stream));
        EGNN_TRY(egnn_node_mlp_fused_f32(node_hi, node_lo, img, info->w5_inv_scale, F(info->b5), info->w6_inv_scale, F(info->b6),
                                         feats, feats_out, rows, dim, x.m, status, stream));
    } else if (desc->update_feats) {
        if (x.kp_hid != 2 * dim) {                                             // pad columns must read as zero in the next GEMM
            if (hipMemsetAsync(ws + w.hid_hi, 0, (size_t)egnn_packed_halves(rows, x.kp_hid) * 2, s) != hipSuccess) return (int)hipGetLastError();
            if (hipMemsetAsync(ws + w.hid_lo, 0, (size_t)egnn_packed_halves(rows, x.kp_hid) * 2, s) != hipSuccess) return (int)hipGetLastError();
        }
        EGNN_TRY(egnn_linear_hl_f32(node_hi, node_lo, blob + info->w5_hi, blob + info->w5_lo, info->w5_inv_scale, F(info->b5), nullptr, 0,
                                    nullptr, 0, ws + w.hid_hi, ws + w.hid_lo, x.kp_hid, rows, 2 * dim, x.kp_node, info->w5_rows, 1, 0,
                                    status, stream));
        EGNN_TRY(egnn_linear_hl_f32(ws + w.hid_hi, ws + w.hid_lo, blob + info->w6_hi, blob + info->w6_lo, info->w6_inv_scale, F(info->b6),
                                    feats, dim, feats_out, dim, nullptr, nullptr, 0, rows, dim, x.kp_hid, info->w6_rows, 0, 0, status,
                                    stream));
    }
    return EGNN_OK;
}
