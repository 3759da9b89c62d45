/* egnn_hip.h -- C ABI of libegnn_hip.so: the MI355X (gfx950) EGNN.forward hot path.
 *
 * The reference (lucidrains/egnn-pytorch v0.2.8) has no FFI layer: its boundary is the Python
 * nn.Module API `EGNN.forward(feats, coors, edges, mask, adj_mat)` (egnn_pytorch/egnn_pytorch.py:224).
 * The entry points below are what a binding for that path calls (the shipped binding is
 * egnn_pytorch_amd/_abi.py, ctypes; INTEGRATION.md shows the stub a reference maintainer would add).
 * Each entry point names the reference lines it replaces.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (HBM) owned by the caller; the library allocates nothing,
 *     keeps no global state and never synchronises (egnn_adj_max_degree_u8 excepted: see below);
 *   - all tensors are contiguous row-major fp32 unless a leading dimension (ld*) is given;
 *     masks / adjacency are 1 byte per element (torch.bool);
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); work is enqueued
 *     asynchronously on it;
 *   - return value: 0 = ok, <0 = EGNN_E_* (bad argument / unsupported shape), >0 = hipError_t of
 *     the failed launch.  egnn_error_string() maps the negative codes to text.
 *   - re-entrant and thread-safe (no shared mutable state); one process per GPU in multi-GPU runs.
 */
#ifndef EGNN_HIP_H
#define EGNN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EGNN_ABI_VERSION 37

enum {
    EGNN_OK = 0,
    EGNN_E_NULLPTR = -1,      /* a required pointer is NULL */
    EGNN_E_SHAPE = -2,        /* non-positive or inconsistent sizes */
    EGNN_E_UNSUPPORTED = -3,  /* shape outside what the kernels are built for (see each entry) */
    EGNN_E_ALIGN = -4,        /* pointer / leading dimension not aligned as required */
    EGNN_E_K_GT_N = -5        /* K > N: the reference's topk raises here too (egnn_pytorch.py:258) */
};

/* Numerical-range status word (every product on this path is a split-fp16 product: DESIGN.md §2).  Kernels that
 * take a `status` pointer (device int32, may be NULL) OR one of these bits into it when a FINITE value leaves the range
 * the fp16 split can carry; the value itself then becomes inf / NaN (never a silently saturated finite number), so the
 * outputs of the call are visibly unusable and the host binding can raise.  The caller zeroes the word and reads it
 * back when it chooses to synchronise (egnn_pytorch_amd: `range_check`).  The reference has no such limits
 * (egnn_pytorch.py:232-233, 287: plain fp32). */
enum {
    EGNN_RANGE_A_OPERAND = 1,  /* a GEMM input (feats, [LayerNorm(feats) | m_i], node_mlp hidden) with |x| >= 65504 */
    EGNN_RANGE_PROJ = 2,       /* a node projection P_i with |-log2(e) P| >= 65504 (needed as an fp16 pair when K >= 6) */
    EGNN_RANGE_SCALAR = 4,     /* a per-edge scalar (squared distance, fourier term, edge feature) with |s / ws_scale| > 6e7 */
    EGNN_RANGE_HIDDEN = 8,     /* an edge_mlp hidden activation beyond fp16 (the edge message came out non-finite) */
    EGNN_RANGE_MESSAGE = 16    /* an edge message / pooled message m_i with |m| >= 65504 */
};

int egnn_abi_version(void);
/* sizeof of the argument structs as this library was compiled -- 0: egnn_edge_args, 1: egnn_edge_bwd_args, 2: egnn_edge_tail_args,
 * 3: egnn_layer_desc, 4: the packed-weights info struct, 5: egnn_edge_exact_args, 6: egnn_edge_exact_bwd_args, 7: egnn_edge_tail_exact_args,
 * 8: egnn_forward_opts; -1 otherwise -- so that a binding that mirrors them (ctypes, cgo, JNI) can verify its
 * layout at load time instead of corrupting a call. */
int64_t egnn_struct_bytes(int which);
/* Hands `nwords` (<= 8) int32 status words to the host without a copy engine and without a stream synchronisation: a one-thread kernel on
 * `stream` copies them into host_pinned[0 .. nwords) -- pinned, host-coherent memory the device can address (hipHostMalloc; torch's
 * pin_memory()) -- and then writes `seq` to host_pinned[nwords]; the caller spins until it reads `seq` there.  What the Python layer's
 * default range check does after every forward. */
int egnn_status_publish(const int32_t* status_dev, int32_t* host_pinned, int nwords, int32_t seq, void* stream);
const char* egnn_error_string(int code);

/* Padded hidden width the projection / edge kernels use for H = 2*edge_input_dim:
 * H rounded up to a multiple of 32 floats (128-byte rows). */
int egnn_padded_hidden(int H);

/* ---------------------------------------------------------------------------------------------
 * Neighbour selection: replaces egnn_pytorch.py:232-233 (pairwise rel_coors / rel_dist), :237-256
 * (ranking: masked pairs -> 1e5, with adj_mat: self -> -1, adjacent -> 0) and :258 (topk smallest K,
 * ascending).  Nothing of size N*N is materialised.
 *   coors  (B,N,coor_dim) fp32, 1 <= coor_dim <= 64.  For coor_dim == 3 rel_dist is computed as
 *          ((dx*dx + dy*dy) + dz*dz) with every multiply and add rounded separately (no FMA) -- bit-identical to the
 *          reference's CPU result; other dimensions follow the summation tree of the reference's sum(dim=-1) (ATen's inner-dimension
 *          sum: left to right for C in {1,2,4,8}; s0, s4 .. s_{C-1}, s1, s2, s3 for C in {5,6,7}; eight interleaved lanes with the
 *          tail first beyond 8 -- csrc/egnn_common.h::egnn_sqdist_any).  More than 8 coordinates: one workgroup per row.
 *   mask   (B,N) bytes or NULL.
 *   adj    (N,N) bytes (adj_batch_stride = 0) or (B,N,N) bytes (adj_batch_stride = N*N), or NULL.
 *          The diagonal is ignored (the reference clears it, :254).
 *   idx_out  (B,N,K) int32   neighbour indices, ascending rank; ties broken by ascending index
 *   rank_out (B,N,K) fp32    the ranking values of the selected neighbours (reference `nbhd_ranking`)
 * Limits: 1 <= K <= min(N, 1024), N <= 32 768 (up to 8192 nodes with 3-D coordinates, 4096 otherwise, a wave keeps a row's candidate
 * keys in registers; beyond that one workgroup per row keeps them in LDS: ~10 us per row).
 */
int egnn_knn_select_f32(const float* coors, const uint8_t* mask, const uint8_t* adj,
                        int64_t adj_batch_stride, int B, int N, int K, int coor_dim,
                        int32_t* idx_out, float* rank_out, void* stream);

/* Replaces `int(adj_mat.float().sum(dim=-1).max().item())` (egnn_pytorch.py:249; the diagonal is
 * counted when set).  adj: (rows, N) bytes.  *out_dev (device int32) receives the maximum row sum;
 * the call zeroes it first on `stream`.  The caller reads it back (that read is the host sync the
 * reference also has at :249). */
int egnn_adj_max_degree_u8(const uint8_t* adj, int64_t rows, int N, int32_t* out_dev, void* stream);

/* Scheduling aid for egnn_edge_fused_f32 (no reference counterpart): per-graph Morton (Z-order) permutation of
 * the nodes, order_out (B,N) int32.  N <= 4096. */
int egnn_spatial_order_f32(const float* coors, int B, int N, int32_t* order_out, void* stream);
/* The same with the padded nodes (mask (B,N) bytes = 0; NULL: none) listed BEHIND the real ones of their graph: padding then fills whole
 * groups of four consecutive positions, which the wave-per-node edge kernel skips (egnn_slot_prep_f32's bit 30). */
int egnn_spatial_order_masked_f32(const float* coors, const uint8_t* mask, int B, int N, int32_t* order_out, void* stream);

/* The transposed neighbour list of the backward (autograd of the gather at egnn_pytorch.py:275): the edge ids (b, i, k) -> b N K + i K + k
 * sorted stably by destination node b N + idx[b,i,k] (idx NULL: dense, destination = k, K = N), in two forms:
 *     csr_order (B N K) int64 with csr_seg (B N + 1): the edges that arrive at node n are csr_order[csr_seg[n] .. csr_seg[n+1])  (the
 *         `order` / `seg_ptr` of egnn_rows_gather_sum_f32)
 *     ent (capacity) int32 with tile_seg (B N + 1): the same, every node's entries padded with -1 to whole 16-entry tiles -- node n owns
 *         tiles [tile_seg[n], tile_seg[n+1]) -- and -1 behind the last tile: the entry list of egnn_edge_bwd_pass_f32 (by_dest = 1),
 *         of which the caller uses the first (tile_seg[B N] * 16 rounded up to 128) entries
 * ent_capacity >= egnn_dest_lists_capacity(B, N, K) entries; tiles_per_graph: (B) int64 scratch.  Counting sort per graph (the
 * destinations of one source row are distinct), two launches, deterministic.  Limits: B N K < 2^31, N <= 20 415 (the per-wave histograms live in LDS). */
size_t egnn_dest_lists_capacity(int B, int N, int K);
int egnn_dest_lists_i32(const int32_t* idx, int B, int N, int K, int32_t* ent, size_t ent_capacity, int64_t* tile_seg,
                        int64_t* csr_order, int64_t* csr_seg, int64_t* tiles_per_graph, void* stream);

/* Flattens the index chain of the edge pass's setup (no reference counterpart; coordinate dimension 3, neighbour path).  A
 * workgroup of egnn_edge_fused_f32 starts with dependent loads -- order -> neighbour list -> coordinates -> mask / rank -- that
 * only the other workgroups of its CU can cover (~0.2 ms of 1.47 at the north-star shape).  This pass does
 * them once per edge slot, in the order the edge pass consumes the slots (position pos of `order`, neighbour k), and leaves one
 * 16-byte record per slot:
 *     slots[(b*N + pos)*K + k] = { j | (pair_ok << 31) | (group_dead << 30),  x_i - x_j  (3 floats, the reference's :232 subtraction bit for bit) }
 * with i = order ? order[b,pos] : pos, j = idx[b,i,k] (< 2^30), pair_ok = mask ? mask[b,i] && mask[b,j] && (rank ? rank[b,i,k] <= valid_radius : 1) : 1
 * (:292-300); group_dead (records with k % 32 == 0 only, 0 elsewhere and without a mask; round 6): the four nodes at positions
 * 4 ((b*N + pos) / 4) .. + 3 of the consumption order are ALL padding (mask = 0) -- every edge of theirs is masked out, their pooled
 * messages are exact zeros and their coordinates unchanged whatever the hidden values are, so the wave-per-node edge kernel, whose
 * workgroup owns exactly those four nodes, skips the round (a single padded node among real ones skips its own hidden loop and keeps
 * the workgroup's barriers).  The edge pass then needs one coalesced load per slot (egnn_edge_args.slots).  slots: B*N*K records of 4 dwords.
 * idx NULL = the dense all-pairs layer (K == N, j = k): with the records a dense layer with N % 32 == 0 runs the wave-per-node edge kernel. */
int egnn_slot_prep_f32(const float* coors, const uint8_t* mask, const int32_t* idx, const float* rank, const int32_t* order,
                       float valid_radius, int B, int N, int K, void* slots, void* stream);

/* ---------------------------------------------------------------------------------------------
 * EGNN_Network front-end: N-degree adjacency expansion, egnn_pytorch.py:414-427.
 * Replaces the float matmuls `(adj.float() @ adj.float()) > 0` and the XOR labelling by bit-set algebra (bit-exact).
 *   adj         (N,N) bytes (adj_batch_stride = 0) or (B,N,N) bytes (adj_batch_stride = N*N)
 *   adj_out     (B,N,N) bytes: the expanded adjacency the layers receive (:425)
 *   degree_out  (B,N,N) bytes: adj_indices (:419-424): 0 = not connected, 1 = adjacent, d = first reached at degree d
 *   workspace   egnn_adj_expand_workspace_bytes(B, N) bytes
 * Limits: N <= 4096, 1 <= num_adj_degrees <= 255. */
size_t egnn_adj_expand_workspace_bytes(int B, int N);
int egnn_adj_expand_u8(const uint8_t* adj, int64_t adj_batch_stride, int B, int N, int num_adj_degrees,
                       uint8_t* adj_out, uint8_t* degree_out, void* workspace, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Dense layers: C = act(A * W^T + bias) (+ residual).  Used for (a) the node-level projections P = feats * [W_i ; W_j]^T +
 * [b1 ; 0] that replace the per-edge first Linear of edge_mlp (egnn_pytorch.py:178-179, 279-287: Linear(cat(h_i,h_j,d,e)) =
 * W_i h_i + W_j h_j + w_d d + W_e e + b), and (b) node_mlp (egnn_pytorch.py:196-201, 336-337).
 *   A (M,K);  W (N,K) (nn.Linear weight layout);  bias (N) or NULL;  residual (M,N) ldr or NULL;  C (M,N) ldc;
 *   act: 0 = identity, 1 = SiLU, 2 = exact GELU (egnn_linear_hl_f32 only, no residual).
 * (The exact-fp32 and split-on-the-fly reference implementations of this operation, kept for A/B tests, live in the
 * test-only library: include/egnn_hip_ref.h.) */

/* Packed ("tile-major") layout of the fp16 GEMM operands.  An (R x Kp) fp16 matrix -- R padded up to a multiple of 32
 * rows, Kp % 32 == 0, both pads zero -- is stored as [R/32][Kp/16][32 rows][2 chunks][8 halves] with the 16-byte chunk
 * index XOR-swizzled by ((row >> 3) & 1):
 *     offset(row, k) = ((((row>>5) * (Kp/16) + (k>>4)) * 32 + (row&31)) * 2 + (((k>>3)&1) ^ (((row&31)>>3)&1))) * 8 + (k&7)
 * so one K-tile of one 32-row block is 1 KB of contiguous memory that is already the bank-conflict-free LDS image.
 * (Why: a wave-level load is processed line by line; row-major operands made every staging instruction touch 32 partly
 * used 128-B lines and capped the stream at ~10 TB/s out of L2.)  egnn_packed_halves = number of fp16 elements. */
int64_t egnn_packed_halves(int64_t rows, int Kp);

/* The production GEMM: C = act(A * W^T + bias) (+ residual), fp32 semantics on the f16 matrix cores.
 * Both operands arrive pre-split into fp16 (hi, lo) pairs in the packed layout (A from egnn_split_f16 /
 * egnn_node_prep_hl / a previous call's C_hi, C_lo; W from egnn_pytorch_amd/_weights.py::split_f16), are staged by
 * LDS-DMA (global_load_lds_dwordx4, 4-deep ring, one barrier per K-tile) and multiplied with three
 * v_mfma_f32_32x32x16_f16 per fragment pair (a_hi w_hi + a_lo w_hi + a_hi w_lo; fp32 accumulation: fp32-class accuracy).
 *   A_hi, A_lo: packed (M, Kp);  W_hi, W_lo: packed (w_rows, Kp), w_rows >= ceil(N/128)*128 (256 x 256 output tiles are
 *   used when w_rows also covers ceil(N/256)*256), holding w_scale * W with w_inv_scale = 1 / w_scale (a power of two);
 *   C (M,N) fp32 row-major and/or C_hi, C_lo packed (M, Kp_out) (the result re-split for the next GEMM; their pad
 *   columns [N, Kp_out) must be zero on entry);  |A| must stay below 65504 (the producers flag violations, see `status`);
 *   status: optional range status word -- EGNN_RANGE_A_OPERAND for C_hi / C_lo values, EGNN_RANGE_PROJ for split_cols words.
 *   split_cols (multiple of 32, <= N, needs C and no residual): columns [0, split_cols) of C are written as 32-bit words
 *   holding (fp16 hi | fp16 lo << 16) of the value instead of the fp32 value -- the form in which the edge pass feeds
 *   P_i to its first-layer MFMA (egnn_edge_args.pi_split).  Other arguments as above. */
int egnn_linear_hl_f32(const void* A_hi, const void* A_lo, const void* W_hi, const void* W_lo,
                       float w_inv_scale, const float* bias, const float* residual, int64_t ldr,
                       float* C, int64_t ldc, void* C_hi, void* C_lo, int Kp_out, int64_t M, int N, int Kp,
                       int w_rows, int act, int split_cols, int32_t* status, void* stream);
/* The same with an A image that is wider than the contraction (ABI 34): A_hi / A_lo were written with K padding Kp_a (>= Kp, % 32 == 0),
 * the product runs over their first Kp columns.  The projection (egnn_pytorch.py:279: feats enters edge_mlp) reads feats out of the
 * [feats | m_i] image that egnn_node_prep_hl writes for node_mlp's first Linear when node_norm is the identity: one image, not two. */
int egnn_linear_hl_lda_f32(const void* A_hi, const void* A_lo, int Kp_a, const void* W_hi, const void* W_lo,
                           float w_inv_scale, const float* bias, const float* residual, int64_t ldr,
                           float* C, int64_t ldc, void* C_hi, void* C_lo, int Kp_out, int64_t M, int N, int Kp,
                           int w_rows, int act, int split_cols, int32_t* status, void* stream);
/* The same (Kp_a = 0: a plain A image) with a row mask (round 6): M-tiles -- 128 or 256 consecutive rows -- none of whose rows has
 * row_mask[row] != 0 are skipped, their rows of C are NOT written.  Made for the projection table of a padded batch: a padded node's rows
 * are read by masked-out edges only, whose values both edge kernels drop by select; with padding at the end of every graph about a
 * tile in five is skipped at the parity protocol's ragged masks.  Inference only (the backward differentiates through every edge's u). */
/* (1 when egnn_edge_fused_f32 runs the wave-per-node kernel for a layer of this shape, given slot records, K >= 6 and no dropout: the
 * one case in which a caller may hand the projection a row mask -- the general kernel's tiles mix the P_i rows of several nodes in one
 * MFMA operand, where an unwritten row would reach its tile neighbours' edges.) */
int egnn_edge_pw_covers(int B, int N, int K, int S, int fourier, int edge_dim, int m_dim, int coor_dim, int64_t ldp);
int egnn_linear_hl_lda_rows_f32(const void* A_hi, const void* A_lo, int Kp_a, const void* W_hi, const void* W_lo,
                                float w_inv_scale, const float* bias, const float* residual, int64_t ldr,
                                float* C, int64_t ldc, void* C_hi, void* C_lo, int Kp_out, int64_t M, int N, int Kp,
                                int w_rows, int act, int split_cols, const uint8_t* row_mask, int32_t* status, void* stream);
/* The same with training-mode dropout between the Linear and the activation (node_mlp, egnn_pytorch.py:196-201): element (row, col) of
 * A W^T + bias is kept iff the hash of [seed, site = node, row, col] (csrc/egnn_common.h) is >= drop_thr and multiplied by drop_inv_keep,
 * else zeroed. */
int egnn_linear_hl_drop_f32(const void* A_hi, const void* A_lo, const void* W_hi, const void* W_lo,
                            float w_inv_scale, const float* bias, const float* residual, int64_t ldr,
                            float* C, int64_t ldc, void* C_hi, void* C_lo, int Kp_out, int64_t M, int N, int Kp,
                            int w_rows, int act, int split_cols, uint32_t drop_thr, uint32_t drop_seed, float drop_inv_keep,
                            int32_t* status, void* stream);

/* The backward's node-level gradient products (autograd of egnn_pytorch.py:287, 336: d/d feats = dP W, d/d W = dP^T feats) on the same
 * split-f16 matrix-core GEMM as the forward.
 * egnn_split_scaled_f16: X (rows, cols) fp32 -> packed (hi, lo) images of scale * X (transposed = 0: `rows` image rows, K = cols) or of
 *     its transpose (transposed = 1: `cols` image rows, K = rows); scale finite and non-zero (for gradients a power of two that lifts
 *     small values off fp16's subnormals; for the weight images of the Python module's re-layout a power of two times the kernels'
 *     unit factor); image rows padded to a multiple of 32 and K to Kp with zeros (both written).
 * egnn_linear_hl_splitk_f32: C_parts[p] (M, ldc) = w_inv_scale * A[:, K range p] W[:, K range p]^T for p < k_splits -- a contraction
 *     that is deep (K = B N nodes) and has a small output fills the chip only when K is cut; parts are summed in fixed order by
 * egnn_sum_parts_f32: out[o] = scale * sum_p parts[p][o], o < count (count % 4 == 0).
 * egnn_absmax_f32: *out_bits = the bit pattern of max |X[o]|, o < count (what the power-of-two scales are chosen from; a NaN anywhere
 *     comes back as a NaN pattern); X 16-byte aligned; one pass, integer atomicMax (order independent). */
int egnn_split_scaled_f16(const float* X, int64_t ldx, int64_t rows, int cols, float scale, int transposed, void* hi, void* lo, int Kp,
                          int32_t* status, void* stream);
/* both images of egnn_split_scaled_f16 from one read of X: hi / lo (`rows` image rows, K = cols padded to Kp) and hiT / loT (`cols` image
 * rows, K = rows padded to KpT) -- a gradient matrix enters one NN product (d/d input) and one TN product (d/d weight).
 * colsum_parts (or NULL; ABI 34): a (egnn_split_scaled_colsum_rows(rows, KpT), ld_colsum >= cols) fp32 array; row r receives the sums of
 * scale * X over X rows 64 r .. 64 r + 63 per column (columns >= cols up to min(ld_colsum, the grid's cover): 0) -- the column sums of
 * a gradient matrix are the gradient of the Linear's bias and this pass reads every element anyway; egnn_sum_parts_f32 adds the rows
 * up in fixed order (scale: a power of two, so 1 / scale there is exact). */
int egnn_split_scaled_both_f16(const float* X, int64_t ldx, int64_t rows, int cols, float scale, void* hi, void* lo, int Kp,
                               void* hiT, void* loT, int KpT, int32_t* status, float* colsum_parts, int64_t ld_colsum, void* stream);
int64_t egnn_split_scaled_colsum_rows(int64_t rows, int KpT);
/* backward of an MLP's SiLU (node_mlp, egnn_pytorch.py:196-201) in one pass: a_out = SiLU(z), gz_out = g * SiLU'(z); count % 4 == 0,
 * 16-byte aligned; a_out may be z and gz_out may be g (element-wise); amax_bits (2 words) or NULL: the bit patterns of max |a_out| and
 * max |gz_out| (egnn_absmax_f32's contract) -- both are operands of the next gradient GEMMs. */
int egnn_silu_bwd_f32(const float* z, const float* g, float* a_out, float* gz_out, int64_t count, uint32_t* amax_bits, void* stream);
/* ... with training-mode dropout between the Linear and the SiLU (node_mlp, egnn_pytorch.py:196-201): z (rows, cols) is the Linear's
 * output; the forward's mask (egnn_linear_hl_drop_f32: site node, row = row0 + r, column c) is re-evaluated: z_d = keep ? z *
 * drop_inv_keep : 0, a_out = SiLU(z_d), gz_out = g SiLU'(z_d) (keep ? drop_inv_keep : 0).  cols >= 4, count % cols == 0 (count % 4 == 0 as
 * above: the array is walked four elements at a time, across row ends). */
int egnn_silu_bwd_drop_f32(const float* z, const float* g, float* a_out, float* gz_out, int64_t count, uint32_t* amax_bits,
                           uint32_t drop_thr, uint32_t drop_seed, float drop_inv_keep, int64_t row0, int cols, void* stream);
int egnn_linear_hl_splitk_f32(const void* A_hi, const void* A_lo, const void* W_hi, const void* W_lo, float w_inv_scale, float* C_parts,
                              int64_t ldc, int64_t M, int N, int Kp, int w_rows, int k_splits, void* stream);
int egnn_sum_parts_f32(const float* parts, int nparts, int64_t count, float scale, float* out, void* stream);
int egnn_absmax_f32(const float* X, int64_t count, uint32_t* out_bits, void* stream);
/* The P_i half of the forward's projection table (egnn_linear_hl_f32 with split_cols: columns [0, cols) are [fp16 hi, fp16 lo] words) turned
 * into the fp32 values hi + lo IN PLACE, for the backward kernels, which read both halves as fp32: the table the forward used is kept
 * for the backward instead of being recomputed.  cols % 4 == 0, ldx % 4 == 0, X 16-byte aligned. */
int egnn_unsplit_words_f32(float* X, int64_t ldx, int64_t rows, int cols, void* stream);

/* node_mlp in one launch for narrow layers (reference: egnn_pytorch.py:196-201, 336-337) -- csrc/node_mlp_fused.hip:
 *     out = W6 SiLU(W5 x + b5) + b6 + residual,   x = the packed [LayerNorm(h) | m_i] image of egnn_node_prep_hl (Kp = pad32(dim + 16))
 * with the hidden activation (rows x 2 dim) kept in registers instead of written out as a packed image and read back by a second
 * egnn_linear_hl_f32.  Built for m_dim = 16 and dim in {32, 64, 128, 256} (egnn_node_mlp_fused_halves returns 0 otherwise; the two-launch
 * path serves every shape).  Same arithmetic as that path (3-term split-fp16 products, fp32 accumulation, bias / SiLU in fp32); results
 * differ from it by the order of the fp32 sums only.  An activation beyond fp16's range sets EGNN_RANGE_A_OPERAND.
 *   egnn_node_mlp_fused_halves: fp16 elements of the fused weight image.
 *   egnn_node_mlp_fused_pack_f16: the image from the packed (hi, lo) images of scale5 * W5 (2 dim rows, Kp = pad32(dim + 16)) and
 *     scale6 * W6 (dim rows, Kp = 2 dim) that egnn_linear_hl_f32 takes (power-of-two scales; their inverses are passed to the kernel).
 *   egnn_node_mlp_fused_f32: residual / out (M, dim) fp32 contiguous, 16-byte aligned; b5 (2 dim), b6 (dim). */
int64_t egnn_node_mlp_fused_halves(int dim, int m_dim);
int egnn_node_mlp_fused_pack_f16(const void* W5_hi, const void* W5_lo, const void* W6_hi, const void* W6_lo, int dim, int m_dim,
                                 void* image, void* stream);
int egnn_node_mlp_fused_f32(const void* X_hi, const void* X_lo, const void* image, float w5_inv_scale, const float* b5,
                            float w6_inv_scale, const float* b6, const float* residual, float* out, int64_t M, int dim, int m_dim,
                            int32_t* status, void* stream);

/* X (rows, cols) fp32 row-major -> packed (rows, Kp) images hi = fp16(x), lo = fp16(x - hi); pads zero; Kp % 32 == 0,
 * Kp >= cols.  |X| >= 65504 (finite) sets EGNN_RANGE_A_OPERAND in *status (optional) and turns into inf / NaN. */
int egnn_split_f16(const float* X, int64_t ldx, int64_t rows, int cols, void* hi, void* lo, int Kp, int32_t* status, void* stream);

/* node_norm + concat (egnn_pytorch.py:335-336), out[r] = [ LayerNorm(feats[r]) | m_i[r] ] (gamma / beta NULL -> Identity,
 * norm_feats=False), written as the packed (hi, lo) pair the GEMM consumes: (rows, Kp), Kp >= dim + m_dim, Kp % 32 == 0.
 * m_i NULL: those columns are written as zeros (egnn_edge_fused_f32 fills them in place: egnn_edge_args.node_hi).
 * raw_hi / raw_lo (optional, both or neither): additionally the un-normalised feats as a packed (rows, raw_Kp) pair --
 * the A operand of the projection GEMM -- so that one pass over feats serves both consumers. */
int egnn_node_prep_hl(const float* feats, const float* m_i, const float* gamma, const float* beta, float eps,
                      void* out_hi, void* out_lo, int Kp, void* raw_hi, void* raw_lo, int raw_Kp,
                      int64_t rows, int dim, int m_dim, int32_t* status, void* stream);

/* ---------------------------------------------------------------------------------------------
 * The fused edge pass: replaces egnn_pytorch.py:262-266, 270-285 (gathers, fourier, concat), :287
 * (edge_mlp), :289-290 (edge_gate), :292-300 (mask combine), :302-317 (coors_mlp, CoorsNorm, clamp,
 * coordinate reduction) and :319-333 (message pooling).  Nothing of size E x H reaches HBM.
 */
typedef struct egnn_edge_args {
    /* shapes */
    int32_t B, N, K;            /* K = neighbours per node (= N on the dense all-pairs path) */
    int32_t dim;                /* feature width (first message column of node_hi / node_lo) */
    int32_t m_dim;              /* <= 64; NB = 1 (m_dim <= 16), 2 (<= 32) or 4 blocks of 16 channels */
    int32_t H, Hp;              /* hidden width 2*Din and its padding (egnn_padded_hidden) */
    int32_t fourier;            /* F = fourier_features */
    int32_t edge_dim;           /* width of `edges` (0 if none) */
    int32_t S;                  /* per-edge scalar inputs: 2F + 1 + edge_dim  (<= 16) */
    int32_t pi_split;           /* format of Pi: 0 = fp32, 1 = (fp16 hi, fp16 lo) words (egnn_linear_hl_f32 split_cols);
                                   must be 1 exactly when K >= 6 (P_i then rides in the first-layer MFMA) */
    /* node-level projections, produced by egnn_linear_hl_f32 */
    const float* Pi;            /* (B*N, ldp): -log2(e) * (W_i h_i + b1)      (pad columns must be 0) */
    const float* Pj;            /* (B*N, ldp): -log2(e) * W_j h_j, fp32 */
    int64_t ldp;
    /* re-laid-out weights (egnn_pytorch_amd/_weights.py) */
    const void* Wst;            /* (Hp, wst_terms, 2) fp16: columns [2dim .. 2dim+S) of edge_mlp.0.weight x -log2(e) x ws_scale as
                                   A fragments of the first-layer MFMA (v_mfma_f32_16x16x16_f16, K-slots 4g+2, 4g+3 of lane
                                   group g of MFMA m hold term 4m+g); per hidden unit and scalar s three (fp16, fp16) words,
                                   terms 3s .. 3s+2:  (hi, lo) of 2^10 w | (hi, lo) of w | (hi, 0) of w;  rest zero.  Position of term
                                   4m + g inside unit h's row: 4m + (g ^ (h & 8 ? 2 : 0)) -- units 8 .. 15 of a 16-block keep the pairs
                                   (0, 1) and (2, 3) swapped so that the kernels' LDS read of units e and e + 8 is conflict-free */
    int32_t wst_terms;          /* = 4 * egnn_edge_mfmas(S) */
    float ws_inv_scale;         /* 1 / ws_scale (a power of two): the kernel multiplies the per-edge scalars by it */
    const void* W2h;            /* (Hp/32, NB, 2, 64, 8) fp16: -ln2 * w2_scale * edge_mlp.3.weight split into hi | lo halves,
                                   in v_mfma_f32_16x16x32_f16 fragment order: [step][nb][hi|lo][lane = 16 g + c][t] =
                                   W2[16 nb + c][32 step + (t < 4 ? 4 g + t : 16 + 4 g + t - 4)] */
    float w2_inv_scale;         /* 1 / w2_scale (a power of two), applied to the accumulated H -> m_dim product */
    const float* b2;            /* (16 NB) edge_mlp.3.bias, zero padded */
    const float* gate_w;        /* (16 NB) edge_gate.0.weight or NULL (soft_edges=False) */
    const float* gate_b;        /* (1) */
    const void* W3h;            /* (2, 64 NB, 16 NB) fp16: w3_scale * coors_mlp.0.weight zero padded, hi image then lo image,
                                   or NULL (update_coors=False) */
    float w3_inv_scale;         /* 1 / w3_scale (a power of two) */
    const float* b3;            /* (64 NB) */
    const float* W4;            /* (64 NB) coors_mlp.3.weight */
    const float* b4;            /* (1) */
    const float* coors_scale;   /* (1) CoorsNorm.scale or NULL (norm_coors=False) */
    /* inputs */
    const float* coors;         /* (B,N,coor_dim) */
    int32_t coor_dim;           /* 1..8 (3 = the fast path); coors_out has the same width */
    const float* edges;         /* (B,N,N,edge_dim) or NULL */
    const uint8_t* mask;        /* (B,N) or NULL */
    const int32_t* idx;         /* (B,N,K) from egnn_knn_select_f32, or NULL = dense (j = k) */
    const float* rank;          /* (B,N,K) or NULL */
    const int32_t* order;       /* (B,N) permutation from egnn_spatial_order_f32 or NULL: workgroups own nodes that are
                                   consecutive in this order (L1 sharing of gathered rows); results do not depend on it */
    float valid_radius;         /* nbhd_mask = rank <= valid_radius, applied only when mask != NULL (:292) */
    float clamp;                /* coor_weights_clamp_value; < 0 = no clamp */
    int32_t pool_mean;          /* m_pool_method == 'mean' */
    /* outputs */
    float* m_i;                 /* (B*N, m_dim) or NULL */
    float* coors_out;           /* (B,N,coor_dim) or NULL (update_coors=False) */
    void* node_hi;              /* optional: packed (B*N, node_kp) fp16 (hi, lo) pair = the node_mlp input prepared by */
    void* node_lo;              /*   egnn_node_prep_hl(m_i = NULL); the pooled messages are written into its columns  */
    int32_t node_kp;            /*   [dim, dim + m_dim) (both or neither; node_kp % 32 == 0, >= dim + m_dim)          */
    int32_t* status;            /* optional range status word (EGNN_RANGE_SCALAR / _HIDDEN / _MESSAGE), see the enum above */
    /* autograd support (NULL for plain inference) */
    float* U_out;               /* forward, optional: (B*N*K, 16 * NB) fp32, NB = 1 / 2 / 4 accumulator blocks for m_dim <= 16 / <= 32 / <= 64
                                   (so m_dim 33 .. 48 has rows of 64 floats, not 48): u = edge_mlp.3(SiLU(edge_mlp.0(.))) before the second
                                   SiLU (egnn_pytorch.py:181-183), one row per edge (b, i, k), pad channels 0: what the backward
                                   (egnn_edge_tail_bwd_f32 / egnn_edge_bwd_pass_f32) differentiates from */
    int32_t edges_by_k;         /* 0: `edges` is (B,N,N,edge_dim), read at [b,i,j];  1: `edges` is (B,N,K,edge_dim), the features of the
                                   selected pairs in neighbour-list order (egnn_edge_features_gather_f32), read at [b,i,k] */
    const void* slots;          /* optional (idx != NULL, coor_dim == 3): the records of egnn_slot_prep_f32 for the SAME idx / rank / mask /
                                   order / valid_radius -- the setup reads them instead of walking order -> idx -> coors -> mask */
    /* training-mode dropout (egnn_pytorch.py:176, 178-184, 203-208; drop_thr = 0: off).  Element (edge, unit) of the pre-activation of
     * edge_mlp's and of coors_mlp's first SiLU is kept iff the counter-based hash of [seed, site, edge id, unit] is >= drop_thr (csrc/egnn_common.h;
     * the torch twin is egnn_pytorch_amd/_dropout.py) and multiplied by drop_inv_keep = 1 / (1 - p), else zeroed.  Every shape of this entry
     * (coor_dim <= 8, m_dim <= 64, up to 16 scalars); the plain kernels carry the same masks (egnn_edge_exact_args.drop_*). */
    uint32_t drop_thr;          /* round(p * 2^32), p in (0, 1) */
    uint32_t drop_seed;
    float drop_inv_keep;
    /* Kernel selection.  0 = automatic: inference calls with K % 32 == 0, S = 1 (squared distance only), m_dim <= 16, coor_dim = 3 and
     * slot records run the persistent wave-per-node kernel (csrc/edge_pw.hip), everything else the general kernel (csrc/edge_fused.hip).
     * 1 = the general kernel always (A/B measurements, and the tests that check the two against each other: same bits for K <= 128). */
    int32_t algo;
} egnn_edge_args;

int egnn_edge_fused_f32(const egnn_edge_args* args, void* stream);

/* EGNN_Network's per-pair edge features (egnn_pytorch.py:410-432: cat(edge_emb(edge tokens) | float edges, adj_emb(adjacency-degree
 * labels))) for the K SELECTED pairs of every node only -- the (B,N,N,edge_dim+adj_dim) tensor is never materialised:
 *   out[b,i,k,:] = [ edge_tok ? edge_tok_emb[edge_tok[b,i,j]] : edges[b,i,j,:d1]  |  adj_deg_emb[adj_deg[b,i,j]] ],  j = idx[b,i,k]
 * (idx NULL: dense, j = k, K = N).  edges (B,N,N,d1) fp32 or NULL; edge_tok (B,N,N) int64 or NULL with edge_tok_emb (T,d1);
 * adj_deg (B,N,N) bytes or NULL with adj_deg_emb (D+1,d2);  out (B,N,K,d1+d2) fp32, fed to egnn_edge_fused_f32 with edges_by_k = 1. */
int egnn_edge_features_gather_f32(const float* edges, const int64_t* edge_tok, const float* edge_tok_emb, int d1,
                                  const uint8_t* adj_deg, const float* adj_deg_emb, int d2, const int32_t* idx,
                                  int B, int N, int K, float* out, void* stream);

/* Backward of the edge pass without anything of size E x H in memory (autograd of egnn_pytorch.py:279-287;
 * csrc/edge_bwd.hip).  One call = one pass over a list of L entries (edges) grouped by a key node: by the source node i
 * (by_dest = 0) or by the neighbour j (by_dest = 1: the edge ids sorted stably by destination).  Layout of the list: the entries
 * of one key node are consecutive and padded with -1 to a multiple of 16, so that every 16-entry tile belongs to one node;
 * L is a multiple of 128 (whole tiles of -1 at the end).  The pass recomputes z = P_i[i] + P_j[j] + W_s s per edge, a = SiLU(z),
 * dz = (W2^T gU) SiLU'(z), and contracts them in registers:
 *     part_rows[q / 16, :]   = sum of dz over tile q / 16     (d/d P_i or d/d P_j after egnn_rows_gather_sum_f32 over each node's
 *                              consecutive tiles -- fixed order, no float atomics; row_pairs: one row per node directly)
 *     dW2_part (n_slabs, 16, Hp),  if not NULL:  partial sums of gU^T a  -> d loss / d edge_mlp.3.weight  = sum over dim 0
 *     dWs_part (S = 1: (n_slabs, Hp);  S > 1: (n_slabs * 16, S, Hp), rows multiples of 4 only, times scal_scale[c]) and ds_part (n_chunks, E, S),
 *                              if not NULL (both or neither): partial sums of s^T dz
 *                              -> d loss / d (scalar columns of edge_mlp.0.weight), and dz W_s over each column chunk
 *                              -> d loss / d scalars = sum over dim 0   (n_chunks = ceil(Hp / 32 / egnn_edge_bwd_chunk_steps()))
 * The all-edge contractions can ride with either pass (each edge appears once in both lists).  Every element of the outputs is
 * written (no zero-fill needed) except the unused rows of dWs_part at S > 1.  Limits: S <= 16 (beyond five scalars d/d s goes to the
 * matrix cores: WsTh; the all-edge contractions must then be split over the two passes), m_dim <= 16 per call (wider heads: once per
 * block of 16 channels), B*N*K < 2^31, the P table and the partial rows below 4 GB each.  All pointers
 * device memory. */
typedef struct egnn_edge_bwd_args {
    int B, N, K;
    int Hp, S;                  /* padded hidden width (egnn_padded_hidden), per-edge scalars */
    int by_dest;                /* key of the grouping: 0 = source node i (own row P_i, gathered row P_j), 1 = neighbour j */
    int n_slabs;                /* the entry list is cut into n_slabs contiguous slabs of 128-entry rounds (grid = n_slabs x n_chunks) */
    int wst_terms;              /* 4 * egnn_edge_mfmas(S) */
    int64_t L;                  /* entries, a multiple of 128 */
    int64_t E;                  /* B * N * K */
    const int32_t* ent;         /* (L) edge id (b*N + i)*K + k of each entry, -1 = padding */
    const float* Pi;            /* (B*N, ldp) fp32 P_i rows incl. bias, in the forward's units (x -log2 e) */
    const float* Pj;            /* (B*N, ldp) fp32 P_j rows */
    int64_t ldp;
    const void* Wst;            /* (Hp, wst_terms, 2) fp16: the forward's scalar-weight table */
    float ws_inv_scale;
    const int32_t* idx;         /* (B*N*K) neighbour of each edge, NULL = dense (K == N, j = k) */
    const void* W2Th;           /* (Hp/32, 2, 2, 64, 4) fp16: W2^T fragments of ONE 16-channel block (hi | lo images; built by egnn_pytorch_amd/_weights.py::pack, key "W2Th_blocks") */
    const float* gU;            /* (E, 16) fp32 d loss / d u */
    float gu_scale;             /* power of two applied to gU before its fp16 split */
    float inv_scale;            /* 1 / (gu_scale * scale of W2Th) */
    const float* scal;          /* (E, S) fp32 per-edge scalars [fourier..., dist, edges...] in natural units */
    const float* Ws;            /* (Hp, S) fp32 scalar columns of edge_mlp.0.weight, rows >= H zero (with dWs_part) */
    const float* scal_scale;    /* (S) powers of two that bring each column of scal into [1, 2) at its maximum (with dWs_part, S > 1) */
    float* part_rows;           /* out: (L / 16 + 1, ld_rows) fp32, one row per tile, the last row scratch */
    int64_t ld_rows;
    float* dW2_part;            /* out or NULL */
    float* dWs_part;            /* out or NULL (with ds_part) */
    float* ds_part;             /* out or NULL (with dWs_part) */
    const void* WsTh;           /* S > 5 with dWs_part: (Hp/32, 2, 2, 64, 4) fp16 fragments of wst_scale * W_s^T, [step][hb][hi|lo][lane = 16 g + s][u] = */
    float wst_inv_scale;        /*   W_s[32 step + 16 hb + 4 g + u][s] (0 for s >= S): d/d s on the matrix cores;  1 / wst_scale */
    uint32_t drop_thr;          /* training-mode dropout behind edge_mlp's first Linear (egnn_pytorch.py:178-184): 0 = none, else the forward's */
    uint32_t drop_seed;         /*   mask (egnn_edge_args.drop_*) is re-evaluated: keep iff hash(seed, site edge, drop_eid0 + edge id, hidden unit) */
    float drop_inv_keep;        /*   >= drop_thr, kept units times drop_inv_keep */
    int64_t drop_eid0;          /*   global id of this call's first edge (the forward numbered the whole batch's edges) */
    int row_pairs;              /* 1 (by source with d/d W_s, 16 < K <= 32, the list = 32 entries per node): the two tiles of a node are summed */
                                /*   in the kernel -- part_rows (L / 32 + 1, ld_rows) holds one row per NODE and no gather-sum is needed */
    uint32_t* rows_amax;        /* out or NULL: the bit pattern of max |part_rows| (egnn_absmax_f32's contract; S = 1 or without dWs_part) -- with */
                                /*   row_pairs / one tile per node the rows ARE d/d P_i and this is the scale of its gradient GEMM operands */
    void* work;                 /* scratch, 16-byte aligned: the pass's per-entry records (other endpoint's row, fp16 fragments of gU and of */
    int64_t work_bytes;         /*   the scalars' first-layer terms, in list order), written by a first launch, read once per column chunk */
} egnn_edge_bwd_args;

int egnn_edge_tail_part_floats(void);
/* The pooled messages the backward's node-level part starts from (egnn_pytorch.py:287-290, :319-326, sum pooling): m_sum (B*N, 16) =
 * sum over k of pair_mask * SiLU(u) * gate, from u (E, 16) = edge_mlp's second Linear output; gate_w (16) / gate_b (1) or NULL,
 * pair_mask (E) bytes or NULL. */
int egnn_edge_pool_f32(const float* u, const float* gate_w, const float* gate_b, const uint8_t* pair_mask, int B, int N, int K,
                       float* m_sum, void* stream);
int egnn_edge_bwd_pass_f32(const egnn_edge_bwd_args* args, void* stream);
int egnn_edge_bwd_chunk_steps(void);    /* hidden steps (of 32 columns) one workgroup owns: sizes ds_part */
/* bytes of egnn_edge_bwd_args.work for a list of L entries; want_w2 / want_s = dW2_part / dWs_part not NULL; 0 = outside the limits */
size_t egnn_edge_bwd_work_bytes(int64_t L, int S, int want_w2, int want_s);

/* The per-edge part of the backward behind edge_mlp's second Linear in closed form (csrc/edge_tail.hip; autograd of
 * egnn_pytorch.py:287 second SiLU, :289-290 edge gate (soft_edges), :292-317 pair mask / coors_mlp / CoorsNorm / clamp / coordinate
 * update, :319-333 pooling; coordinate dimension 3).  One edge per lane: from u = the second Linear's output (E, 16), g_coors_out = d loss / d coors_out
 * and g_msum = d loss / d (sum over k of the pair-masked messages; mean pooling: already divided by the count) it writes
 * gU = d loss / d u (the input of egnn_edge_bwd_pass_f32), g_rel = d loss / d (x_i - x_j) without the distance path, and what the
 * parameter gradients of coors_mlp are tall products of: g_hid (E, 64), a3 (E, 64), g_w (E), and the per-edge terms of
 * d loss / d coors_norm.scale.  W3 / b3 / W4 zero padded to 64 x 16 / 64 / 64 by the caller. */
typedef struct egnn_edge_tail_args {
    int B, N, K;
    int norm_coors;             /* CoorsNorm on (scale, eps below) */
    float clamp;                /* coor_weights_clamp_value, < 0 = none */
    float eps;                  /* CoorsNorm eps */
    const float* u;             /* (E, 16) fp32, columns >= m_dim zero */
    const float* coors;         /* (B*N, 3) */
    const int32_t* idx;         /* (E) neighbour of each edge, NULL = dense (K == N) */
    const uint8_t* pair_mask;   /* (E) mask_i & mask_j & (rank <= valid_radius), NULL = all pairs count */
    const float* g_coors_out;   /* (B*N, 3) */
    const float* g_msum;        /* (B*N, 16) zero padded */
    const float* W3;            /* (64, 16) coors_mlp.0.weight zero padded */
    const float* b3;            /* (64) */
    const float* W4;            /* (64) coors_mlp.3.weight */
    const float* b4;            /* (1) */
    const float* scale;         /* (1) coors_norm.scale or NULL */
    float* gU;                  /* out (E, 16) */
    float* g_rel;               /* out (E, 4), last column 0 */
    float* g_hid;               /* out (E, 64) */
    float* a3;                  /* out (E, 64) */
    float* g_w;                 /* out (E) */
    float* g_scale;             /* out (E) or NULL */
    const float* gate_w;        /* (16) edge_gate.0.weight zero padded, or NULL (soft_edges=False): m_ij = SiLU(u) sigmoid(gate_w . SiLU(u) + gate_b) */
    const float* gate_b;        /* (1) */
    float* g_gate;              /* out (E): d loss / d (gate pre-activation) -- d/d gate_w = sum_e g_gate[e] SiLU(u_e), d/d gate_b = sum_e g_gate[e] */
    float* part;                /* out or NULL: (ceil(E / 256) * 4, egnn_edge_tail_part_floats()) -- per wave of 64 edges the sums the parameter gradients */
                                /*   are made of, INSTEAD of g_hid / a3 / g_w / g_scale / g_gate (those may be NULL): [0, 1024) d/d W3 (64 x 16), */
                                /*   [1024, 1088) d/d b3, [1088, 1152) d/d W4, then 40 scalars: [0, 16) column sums of gU, [16, 32) d/d gate_w, */
                                /*   32 d/d b4, 33 d/d CoorsNorm.scale, 34 d/d gate_b, rest 0.  Summed over rows by egnn_sum_parts_f32 (fixed order). */
    float* rel_out;             /* out or NULL: (E, 4) x_i - x_j (4th 0) and, with it, dist_out (E) = |x_i - x_j|^2 -- what the backward of the */
    float* dist_out;            /*   distance path needs (d loss / d rel += 2 g_dist rel), by-products of this pass */
    uint32_t* amax_gu;          /* out or NULL: the bit pattern of max |gU| (egnn_absmax_f32's contract) -- the scale of the next pass's fp16 split */
    uint32_t drop_thr;          /* training-mode dropout behind coors_mlp's first Linear (egnn_pytorch.py:203-208): 0 = none, else the forward's mask */
    uint32_t drop_seed;         /*   (site coors, row = drop_eid0 + edge id, column = hidden unit) is re-evaluated; with `part` only */
    float drop_inv_keep;
    int64_t drop_eid0;
} egnn_edge_tail_args;

int egnn_edge_tail_bwd_f32(const egnn_edge_tail_args* args, void* stream);

/* Backward of the neighbour gather (egnn_pytorch.py:275): out[r, :] = sum of rows[order[p], :] for p in [seg_ptr[r], seg_ptr[r+1]),
 * in that order -- with `order` = the edges sorted (stably) by destination node this is d loss / d P_j from dZ, a fixed-order
 * read-only reduction (no float atomics: bit-reproducible).  rows (n_rows, ld) fp32, out (n_out, ldo) fp32, cols % 4 == 0,
 * order / seg_ptr int64 (seg_ptr has n_out + 1 entries).  amax_bits or NULL: the bit pattern of max |out| (egnn_absmax_f32's contract). */
int egnn_rows_gather_sum_f32(const float* rows, int64_t ld, const int64_t* order, const int64_t* seg_ptr, int64_t n_out,
                             int cols, float* out, int64_t ldo, uint32_t* amax_bits, void* stream);

/* Number of chained first-layer MFMAs the edge kernel is instantiated with for S per-edge scalars (>= ceil(3 S / 4); one of 1, 3, 4, 6, 12). */
int egnn_edge_mfmas(int S);


/* ---------------------------------------------------------------------------------------------
 * EGNN_Network's induced-set attention block (egnn_pytorch.py:83-144): the two attention cores.  The
 * projections around them are egnn_linear_hl_f32 calls (act = 2: exact GELU for the feed-forward).
 *   egnn_induced_attn_f32: out[b,t,(h d)] = softmax_n(scale q[b,t,h,:] . k[b,n,h,:], masked nodes -> -FLT_MAX) v[b,n,h,:]
 *       q (B,T,heads*dim_head) fp32;  kv (B*N, ldkv) fp32 = attn1.to_kv(LayerNorm(x)): k in columns [0, inner), v in [inner, 2 inner);
 *       mask (B,N) bytes or NULL;  out (B,T,inner).  T <= 8, dim_head <= 256.
 *   egnn_token_attn_f32:   out[r,(h d)] = softmax_t(scale q[r,h,:] . k_tok[b,t,h,:]) v_tok[b,t,h,:],  r = b N + n
 *       q (B*N, ldq);  kv_tok (B,T,2*inner) fp32 = attn2.to_kv(induced);  out (B*N, ldo). */
int egnn_induced_attn_f32(const float* q, const float* kv, int64_t ldkv, const uint8_t* mask, int B, int N, int T, int heads,
                          int dim_head, float scale, float* out, void* stream);
int egnn_token_attn_f32(const float* q, int64_t ldq, const float* kv_tok, int B, int N, int T, int heads, int dim_head,
                        float scale, float* out, int64_t ldo, void* stream);

/* =============================================================================================
 * Whole-layer interface: everything a binding that is NOT the shipped Python one needs to
 * run `EGNN.forward` (egnn_pytorch.py:224-341) from the reference's own parameters -- no torch, no re-implementation of
 * the weight re-layout.  tests/c_abi/layer_forward_test.c drives a golden case through exactly these four calls.
 */
typedef struct egnn_layer_desc {            /* the constructor arguments of EGNN (egnn_pytorch.py:149-168) */
    int32_t dim, edge_dim, m_dim, fourier_features, num_nearest_neighbors;
    int32_t norm_feats, norm_coors, update_feats, update_coors, only_sparse_neighbors, soft_edges;
    int32_t pool_mean;                      /* m_pool_method == 'mean' */
    float valid_radius;                     /* +inf (or >= 3e38) = none */
    float coor_weights_clamp_value;         /* < 0 = None */
    float ln_eps;                           /* node_norm.eps (1e-5) */
} egnn_layer_desc;

typedef struct egnn_layer_params {          /* the reference's state_dict, fp32 row-major, HOST pointers; NULL where the module */
    const float *edge_mlp_0_weight, *edge_mlp_0_bias;      /* (H, Din), (H): Din = 2 dim + 2 F + 1 + edge_dim, H = 2 Din   */
    const float *edge_mlp_3_weight, *edge_mlp_3_bias;      /* (m_dim, H), (m_dim)                                           */
    const float *edge_gate_0_weight, *edge_gate_0_bias;    /* (1, m_dim), (1)            -- soft_edges                      */
    const float *node_norm_weight, *node_norm_bias;        /* (dim), (dim)               -- norm_feats                      */
    const float *coors_norm_scale;                         /* (1)                        -- norm_coors                      */
    const float *node_mlp_0_weight, *node_mlp_0_bias;      /* (2 dim, dim + m_dim), (2 dim)   -- update_feats                */
    const float *node_mlp_3_weight, *node_mlp_3_bias;      /* (dim, 2 dim), (dim)                                           */
    const float *coors_mlp_0_weight, *coors_mlp_0_bias;    /* (4 m_dim, m_dim), (4 m_dim)     -- update_coors                */
    const float *coors_mlp_3_weight, *coors_mlp_3_bias;    /* (1, 4 m_dim), (1)                                             */
} egnn_layer_params;                                       /* does not have the tensor */

/* Where each re-laid-out tensor sits inside the packed weight blob (byte offsets; 0 size = absent), plus the power-of-two
 * scales the kernels undo.  Filled by egnn_pack_weights_host, consumed by egnn_layer_forward_f32. */
typedef struct egnn_packed_info {
    int32_t H, Hp, S, NM;                   /* hidden width, its padding, per-edge scalars, first-layer MFMAs */
    int32_t wcat_rows, w5_rows, w6_rows;    /* padded row counts of the packed GEMM weights */
    float wcat_inv_scale, ws_inv_scale, w2_inv_scale, w3_inv_scale, w5_inv_scale, w6_inv_scale;
    uint64_t wcat_hi, wcat_lo, bcat, wst, w2h, b2, gate_w, gate_b, w3h, b3, w4, b4, coors_scale,
             w5_hi, w5_lo, b5, w6_hi, w6_lo, b6, gamma, beta;
    uint64_t bytes;                         /* total blob size */
} egnn_packed_info;

/* Size of the packed blob for a layer (0 on an invalid descriptor). */
size_t egnn_packed_weights_bytes(const egnn_layer_desc* desc);

/* The blob's layout alone: dimensions, padded row counts, byte offsets and `bytes` of `info`; the scales are left 0.  For a binding
 * that re-lays the weights itself ON THE DEVICE in the formats of egnn_pack_weights_host (the shipped module does: its torch packer
 * needs ~1.5 ms where the host function needs 60 ms at dim 512, which matters when the parameters change between forwards) and only
 * has to place the pieces where egnn_layer_forward_f32 looks for them. */
int egnn_packed_layout(const egnn_layer_desc* desc, egnn_packed_info* info);

/* The weight re-layout of egnn_pytorch_amd/_weights.py::pack as a HOST function (pure CPU work, no HIP call): factorised
 * first Linear of edge_mlp (W_i | W_j | W_s, x -log2 e), fp16 (hi, lo) splits with power-of-two scales, packed tile-major
 * GEMM operands, MFMA fragment orders (see egnn_edge_args).  `blob` (host, egnn_packed_weights_bytes bytes) is what the
 * caller copies to the device once per set of parameters; bit-identical to the Python packer
 * (tests/test_host_logic.py::test_c_weight_packer_matches_python). */
int egnn_pack_weights_host(const egnn_layer_desc* desc, const egnn_layer_params* params, void* blob, egnn_packed_info* info);

/* Device workspace one forward needs (neighbour list, projections P, packed GEMM operands ...) for B graphs of N nodes with
 * K neighbours per node (K = N on the dense path; K = the largest adjacency row sum with only_sparse_neighbors).  0 on
 * invalid arguments.  The library still allocates nothing: the caller provides the buffer. */
size_t egnn_workspace_bytes(const egnn_layer_desc* desc, int B, int N, int K);

/* One EGNN layer forward = the 7 launches of DESIGN.md §1 chained on `stream` (neighbour select, operand prep, projection
 * GEMM, Morton order, fused edge pass, node_mlp GEMMs); semantics and quirks of egnn_pytorch.py:224-341.
 *   blob_dev: the packed blob on the device;  feats (B,N,dim), coors (B,N,coor_dim) fp32;  edges (B,N,N,edge_dim) or NULL;
 *   mask (B,N) bytes or NULL;  adj (N,N) / (B,N,N) bytes or NULL (adj_batch_stride 0 / N*N);
 *   feats_out (B,N,dim), coors_out (B,N,coor_dim): always written (copies of the inputs when the layer does not update them);
 *   K: neighbours per node actually used -- pass num_nearest_neighbors, or N on the dense path; with only_sparse_neighbors
 *      pass the largest adjacency row sum (egnn_adj_max_degree_u8; that device read is the host sync the reference has
 *      too, :249) -- workspace_bytes must be >= egnn_workspace_bytes(desc, B, N, K);
 *   status: optional range status word (EGNN_RANGE_*).
 * Returns EGNN_E_K_GT_N when K > N (the reference's topk error). */
int egnn_layer_forward_f32(const egnn_layer_desc* desc, const egnn_packed_info* info, const void* blob_dev,
                           const float* feats, const float* coors, const float* edges, const uint8_t* mask,
                           const uint8_t* adj, int64_t adj_batch_stride, int B, int N, int K, int coor_dim,
                           float* feats_out, float* coors_out, void* workspace, size_t workspace_bytes,
                           int32_t* status, void* stream);

/* The same forward with what a binding that calls it per layer of a running model wants (round 6: the shipped Python module's
 * inference path is this one call -- egnn_pytorch_amd/layer.py::_forward_c -- instead of a dozen Python-side launches: the host
 * time of a forward falls from ~200 us to a few tens, which is what a synchronous range check exposes per call and what bounds
 * small graphs).  NULL opts = egnn_layer_forward_f32.  All handles belong to the caller; the library still creates nothing.
 *   side_stream: a second hipStream_t for the neighbour selection (k-NN select, Morton order, slot records: they read the
 *                coordinates only) -- it waits for ev_fork, which this call records on `stream` at its entry, runs beside the
 *                operand prep and the projection GEMM, and is joined in front of the edge pass through ev_join (both hipEvent_t,
 *                required with side_stream; reusable by the next call once this one has been enqueued).  The workspace must have been
 *                allocated in `stream`'s order (the side stream touches it only between the two events).
 *   order / order_is_hint: (B,N) int32 buffer for the Morton order instead of the workspace's -- written by this call, or, with
 *                order_is_hint = 1, read as is (a stack of layers reuses the first layer's order: scheduling only, any permutation
 *                of each graph's nodes is valid);
 *   nmf_img:     the fused node_mlp weight image (egnn_node_mlp_fused_pack_f16) kept by the caller across calls; NULL: re-derived
 *                from the blob into the workspace at every call. */
typedef struct egnn_forward_opts {
    void* side_stream;
    void* ev_fork;
    void* ev_join;
    int32_t* order;
    const void* nmf_img;
    int32_t order_is_hint;
    int32_t reserved;
} egnn_forward_opts;

int egnn_layer_forward_opts_f32(const egnn_layer_desc* desc, const egnn_packed_info* info, const void* blob_dev,
                                const float* feats, const float* coors, const float* edges, const uint8_t* mask,
                                const uint8_t* adj, int64_t adj_batch_stride, int B, int N, int K, int coor_dim,
                                float* feats_out, float* coors_out, void* workspace, size_t workspace_bytes,
                                int32_t* status, void* stream, const egnn_forward_opts* opts);


/* =============================================================================================
 * The wide-range path: the same layer in plain fp32 (no split-fp16 products, no re-laid-out weights).
 *
 * The fast kernels above carry every product as a split-fp16 pair: a finite activation beyond fp16's 65504 at one of their cast
 * sites sets a bit of the range status word instead of producing a number (EGNN_RANGE_*).  The reference computes in plain fp32
 * (egnn_pytorch.py:232-233, 287) and has no such limit.  A binding re-runs a call that tripped a bit on the entries below (the shipped
 * binding does so automatically, egnn_pytorch_amd/layer.py): exact-fp32 GEMMs (v_mfma_f32_32x32x2_f32 = an fmaf chain), fp32
 * node_norm, and the edge pass as fp32 VALU arithmetic on the module's own weight tensors -- the reference's arithmetic class,
 * overflowing only where fp32 itself does.  Several times slower than the fast path; never what bench.py times.
 * ============================================================================================= */
/*  * Dense layer, exact-fp32 variant (v_mfma_f32_32x32x2_f32): C = act(A * W^T + bias) (+ residual).
 * Used for (a) the node-level projections P = feats * [W_i ; W_j]^T + [b1 ; 0] that replace the
 * per-edge first Linear of edge_mlp (egnn_pytorch.py:178-179, 279-287: Linear(cat(h_i,h_j,d,e)) =
 * W_i h_i + W_j h_j + w_d d + W_e e + b), and (b) node_mlp (egnn_pytorch.py:196-201, 336-337).
 *   A (M,K) lda;  W (N,K) ldw (nn.Linear weight layout);  bias (N) or NULL;
 *   residual (M,N) ldr or NULL;  C (M,N) ldc;  act: 0 = identity, 1 = SiLU.
 * v_mfma_f32_32x32x2_f32: exact fp32 products/accumulation.
 */
int egnn_linear_f32(const float* A, int64_t lda, const float* W, int64_t ldw, const float* bias,
                    const float* residual, int64_t ldr, float* C, int64_t ldc,
                    int64_t M, int N, int K, int act, void* stream);


/* node_norm + concat (egnn_pytorch.py:335-336): out[r] = [ LayerNorm(feats[r]) | m_i[r] ].
 * gamma/beta NULL -> Identity (norm_feats=False).  out: (rows, dim + m_dim). */
int egnn_node_prep_f32(const float* feats, const float* m_i, const float* gamma, const float* beta,
                       float eps, float* out, int64_t rows, int dim, int m_dim, void* stream);


/* The edge pass of egnn_pytorch.py:262-333 in fp32.  Weights are the module's own tensors:
 *   Pi, Pj   (B*N, ldp) fp32, natural units: Pi = feats W_i^T + b1, Pj = feats W_j^T with W_i = edge_mlp.0.weight[:, :dim],
 *            W_j = edge_mlp.0.weight[:, dim:2dim] (two egnn_linear_f32 calls with ldw = Din);
 *   Ws       = edge_mlp.0.weight + 2 dim: row h holds the S = 2 fourier + 1 + edge_dim scalar columns, ldws = Din;
 *   W2, b2   edge_mlp.3 (m_dim, H), (m_dim);  gate_w (m_dim), gate_b (1) or NULL;
 *   W3 (4 m_dim, m_dim), b3 (4 m_dim), W4 (4 m_dim), b4 (1): coors_mlp, or NULL (update_coors=False);  coors_scale (1) or NULL;
 *   coors (B,N,coor_dim), 1 <= coor_dim <= 64;  edges / edges_by_k / mask / idx / rank / valid_radius / clamp / pool_mean as in egnn_edge_args;
 *   m_i (B*N, m_dim) and / or coors_out (B,N,coor_dim);  edge_ws: egnn_edge_exact_workspace_bytes() of scratch (per-edge rows,
 *   summed per node in k order: deterministic).
 * Limits: m_dim <= 1024 (beyond 64 the second Linear runs in blocks of 64 channels and the messages live in the workspace row),
 * per-edge scalars: 160 in fp32, 80 in float64 (a workgroup's 256 edges keep theirs in LDS), coor_dim <= 64. */
typedef struct egnn_edge_exact_args {
    int32_t B, N, K, m_dim, H, fourier, edge_dim, coor_dim, pool_mean, edges_by_k;
    /* data pointers: float for egnn_edge_exact_f32, double for egnn_edge_exact_f64 */
    const void* Pi;
    const void* Pj;
    int64_t ldp;
    const void* Ws;
    int64_t ldws;
    const void* W2;
    const void* b2;
    const void* gate_w;
    const void* gate_b;
    const void* W3;
    const void* b3;
    const void* W4;
    const void* b4;
    const void* coors_scale;
    const void* coors;
    const void* edges;
    const uint8_t* mask;
    const int32_t* idx;
    const void* rank;
    double valid_radius, clamp;
    void* m_i;
    void* coors_out;
    void* edge_ws;
    void* U_out;                /* optional (forward under autograd): (B*N*K, m_dim) u = edge_mlp.3(SiLU(edge_mlp.0(.))) incl. its bias, before the
                                   second SiLU (egnn_pytorch.py:181-183) -- what egnn_edge_exact_bwd_* and the per-edge tail differentiate from */
    /* training-mode dropout behind the first Linear of edge_mlp and of coors_mlp (egnn_pytorch.py:178-184, 203-208; ABI 34): the same
     * counter-based hash masks as the fused kernels (csrc/egnn_common.h: sites edge / coors, row = drop_eid0 + edge, column = hidden
     * unit), kept values x drop_inv_keep; drop_thr = 0: none.  node_mlp's mask: egnn_drop_silu_f32 / _f64 between its two Linears. */
    uint32_t drop_thr, drop_seed;
    float drop_inv_keep;
    int64_t drop_eid0;
} egnn_edge_exact_args;

size_t egnn_edge_exact_workspace_bytes(int B, int N, int K, int m_dim, int coor_dim);      /* fp32; twice that for egnn_edge_exact_f64 */
int egnn_edge_exact_f32(const egnn_edge_exact_args* args, void* stream);
/* Z (rows, ld >= cols) <- SiLU(dropout(Z)) in place: nn.Dropout between node_mlp's first Linear and its SiLU on the plain kernels
 * (egnn_pytorch.py:196-201) -- element (row, col) kept iff the hash of [seed, site = node, row0 + row, col] >= drop_thr (csrc/egnn_common.h)
 * and multiplied by drop_inv_keep, else zero; drop_thr = 0: plain SiLU. */
int egnn_drop_silu_f32(void* Z, int64_t ld, int64_t rows, int cols, uint32_t drop_thr, uint32_t drop_seed, float drop_inv_keep, int64_t row0,
                       void* stream);
int egnn_drop_silu_f64(void* Z, int64_t ld, int64_t rows, int cols, uint32_t drop_thr, uint32_t drop_seed, float drop_inv_keep, int64_t row0,
                       void* stream);

/* The E x H work of the BACKWARD of that path (csrc/edge_exact_bwd.hip; autograd of egnn_pytorch.py:277-287): plain fp32 for calls that
 * were answered by the wide-range path and for the shapes beyond the fused kernels' limits, float64 for float64 modules (the reference's
 * own training recipe, denoise_sparse.py:11, 23-32).  With z = P_i[i] + P_j[j] + W_s s, a = SiLU(z) and gU = d loss / d u:
 *     dz = (W2^T gU) SiLU'(z);  A_T (H, E) = a and DZ_T (H, E) = dz, TRANSPOSED so that d/d W2 = gU^T a and d/d W_s = dz^T s are plain
 *     C = X W^T products of egnn_linear_f32 / _f64 with the edges as the contraction;  g_scal (E, S) = dz W_s.
 * Same shape limits as the forward entry except the per-edge scalars: 80 in fp32, 40 in float64 (scalars and their gradients in LDS).
 * egnn_edge_exact_node_sums_*: d/d P_i[n] = sum of dz over the K edges leaving n, d/d P_j[n] = sum over the edges arriving at n (the CSR
 * lists of egnn_dest_lists_i32), in a fixed order, each in both layouts -- (nodes, H) and (H, nodes).  The caller bounds 2 H E elements
 * by cutting the batch into chunks of graphs. */
typedef struct egnn_edge_exact_bwd_args {
    int32_t B, N, K, m_dim, H, fourier, edge_dim, coor_dim, edges_by_k, reserved;
    /* data pointers: float for egnn_edge_exact_bwd_f32, double for egnn_edge_exact_bwd_f64 */
    const void* Pi;             /* as egnn_edge_exact_args: the projection table the forward used */
    const void* Pj;
    int64_t ldp;
    const void* Ws;
    int64_t ldws;
    const void* W2;             /* (m_dim, H) */
    const void* coors;
    const void* edges;
    const int32_t* idx;
    const void* gU;             /* (B*N*K, m_dim) d loss / d u */
    void* A_T;                  /* out (H, B*N*K) */
    void* DZ_T;                 /* out (H, B*N*K) */
    void* g_scal;               /* out (B*N*K, S), S = 2 fourier + 1 + edge_dim: [sin.., cos.., dist, edge features..] */
    /* training-mode dropout behind edge_mlp's first Linear, as in `egnn_edge_exact_args` -- ABI 34: z is re-evaluated with the forward's
     * mask, A_T = SiLU of the dropped pre-activation, DZ_T = the gradient with respect to the Linear's output */
    uint32_t drop_thr, drop_seed;
    float drop_inv_keep;
    int64_t drop_eid0;
} egnn_edge_exact_bwd_args;
int egnn_edge_exact_bwd_f32(const egnn_edge_exact_bwd_args* args, void* stream);
int egnn_edge_exact_bwd_f64(const egnn_edge_exact_bwd_args* args, void* stream);
int egnn_edge_exact_node_sums_f32(const void* DZ_T, int64_t E, int H, int64_t nodes, int K, const int64_t* csr_order, const int64_t* csr_seg,
                                  void* gPi, void* gPi_T, void* gPj, void* gPj_T, void* stream);
int egnn_edge_exact_node_sums_f64(const void* DZ_T, int64_t E, int H, int64_t nodes, int K, const int64_t* csr_order, const int64_t* csr_seg,
                                  void* gPi, void* gPi_T, void* gPj, void* gPj_T, void* stream);

/* The per-edge chain BEHIND u in closed form for any head width up to 64 channels and any coordinate dimension, fp32 or float64
 * (autograd of egnn_pytorch.py:287 second SiLU, :289-290 gate, :292-317 pair mask / coors_mlp / CoorsNorm / clamp / coordinate update,
 * :319-333 pooling): what egnn_edge_tail_bwd_f32 is for the standard layer (16 channels, 3-D coordinates, matrix cores).  One thread per
 * edge; egnn_pytorch_amd/autograd.py::tail_edge_backward is the specification.  Per edge e = (b, i, k), neighbour j:
 *     gU (E, m_dim) = d loss / d u;   g_rel (coor_dim, E) = d loss / d (x_i - x_j) without the distance path (a self pair: exact 0),
 *     transposed like the other per-edge outputs so that egnn_edge_exact_node_sums_* turns it into the per-node sums;
 *     the operands of the parameter gradients, TRANSPOSED (rows, E) so that the sums over all edges are C = X W^T products of
 *     egnn_linear_f32 / _f64:  ghid_t, a3_t (4 m_dim, E) = d/d (pre-activation of coors_mlp's SiLU) and its activation,
 *     mm_t (m_dim, E) = the (gated) messages, m0_t (m_dim, E) = SiLU(u) (gate only);  g_w, g_scale, g_gate (E) = d/d coors_mlp's output,
 *     the terms of d/d coors_norm.scale, d/d the gate's pre-activation.
 * Inputs: u (E, m_dim); g_coors_out (B N, coor_dim) = d loss / d coors_out; g_msum (B N, m_dim) = d loss / d (sum over k of the
 * pair-masked messages) or NULL; pair_mask (E) bytes or NULL (the reference applies masks only when a mask is passed, :292); W3 NULL =
 * no coors_mlp (update_coors = False); clamp < 0 = none.  With drop_thr != 0, a3_t holds SiLU of the DROPPED pre-activation and ghid_t
 * the gradient with respect to the Linear's output (mask and 1 / keep applied). */
typedef struct egnn_edge_tail_exact_args {
    int32_t B, N, K, m_dim, coor_dim, norm_coors;
    double eps, clamp;
    const void* u;
    const void* coors;
    const int32_t* idx;         /* (E) or NULL = dense (K == N, j = k) */
    const uint8_t* pair_mask;
    const void* g_coors_out;
    const void* g_msum;
    const void* W3;             /* coors_mlp.0.weight (4 m_dim, m_dim), .bias; coors_mlp.3.weight (4 m_dim), .bias (1) */
    const void* b3;
    const void* W4;
    const void* b4;
    const void* scale;          /* coors_norm.scale (1) when norm_coors */
    const void* gate_w;         /* edge_gate.0.weight (m_dim), .bias (1), or NULL */
    const void* gate_b;
    void* gU;
    void* g_rel;
    void* ghid_t;
    void* a3_t;
    void* mm_t;
    void* m0_t;                 /* optional (gate) */
    void* g_w;                  /* optional */
    void* g_scale;              /* optional */
    void* g_gate;               /* optional */
    /* training-mode dropout behind coors_mlp's first Linear (egnn_pytorch.py:203-208; ABI 34): the forward's hash mask (csrc/egnn_common.h:
     * site coors, row = drop_eid0 + edge, column = hidden unit) re-evaluated; drop_thr = 0: none */
    uint32_t drop_thr, drop_seed;
    float drop_inv_keep;
    int64_t drop_eid0;
} egnn_edge_tail_exact_args;
int egnn_edge_tail_exact_bwd_f32(const egnn_edge_tail_exact_args* args, void* stream);
int egnn_edge_tail_exact_bwd_f64(const egnn_edge_tail_exact_args* args, void* stream);

/* =============================================================================================
 * The float64 path: a float64 module in float64 arithmetic.
 *
 * The reference is dtype-generic and its own tests run in float64 (tests/test_equivariance.py:6); the fast kernels carry ~22
 * significant bits per product.  A binding routes a module whose parameters are float64 through the entries below (the shipped one
 * does: egnn_pytorch_amd/layer.py) -- the same plain kernels as the wide-range path, instantiated for double:
 *   egnn_knn_select_f64   as egnn_knn_select_f32 (squared distances in the same operation order, ranking edits, exact top-K, ties
 *                         towards the lowest index), coor_dim <= 64; rank_out in float64.  One workgroup per row, keys in LDS: N <= 20 000, K <= 1024
 *   egnn_linear_f64       as egnn_linear_f32, on v_mfma_f64_16x16x4_f64
 *   egnn_node_prep_f64    as egnn_node_prep_f32
 *   egnn_edge_exact_f64   as egnn_edge_exact_f32 with every data pointer of egnn_edge_exact_args a double*; workspace twice
 *                         egnn_edge_exact_workspace_bytes()
 * Correct and deterministic first: gfx950's float64 matrix rate is 1/32 of its fp16 rate.  Never what bench.py times.
 * ============================================================================================= */
int egnn_knn_select_f64(const double* coors, const uint8_t* mask, const uint8_t* adj, int64_t adj_batch_stride,
                        int B, int N, int K, int coor_dim, int32_t* idx_out, double* rank_out, void* stream);
int egnn_linear_f64(const double* A, int64_t lda, const double* W, int64_t ldw, const double* bias,
                    const double* residual, int64_t ldr, double* C, int64_t ldc,
                    int64_t M, int N, int K, int act, void* stream);
int egnn_node_prep_f64(const double* feats, const double* m_i, const double* gamma, const double* beta,
                       double eps, double* out, int64_t rows, int dim, int m_dim, void* stream);
int egnn_edge_exact_f64(const egnn_edge_exact_args* args, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* EGNN_HIP_H */
