/* A binding without torch: one EGNN layer forward through the C ABI of libegnn_hip.so, checked against a golden case.
 *
 *   gcc -O1 -std=c11 -D__HIP_PLATFORM_AMD__ -I include -I /opt/rocm/include tests/c_abi/layer_forward_test.c \
 *       -L egnn_pytorch_amd -legnn_hip -L /opt/rocm/lib -lamdhip64 -lm -Wl,-rpath,$PWD/egnn_pytorch_amd -o /tmp/layer_forward_test
 *   /tmp/layer_forward_test case.bin          (written by tests/test_c_abi.py from tests/golden/<name>.npz)
 *
 * Only include/egnn_hip.h and the HIP runtime API are used: egnn_packed_weights_bytes -> egnn_pack_weights_host (host) ->
 * hipMemcpy -> egnn_workspace_bytes -> egnn_layer_forward_f32, i.e. what the reference-side stub of INTEGRATION.md calls.
 * The golden outputs were produced by the reference itself (tests/golden/make_golden.py); tolerance 1e-4 (BASELINE.json).
 */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "egnn_hip.h"

#define NPARAM 17
#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %d at %s:%d\n", (int)e_, __FILE__, __LINE__); return 2; } } while (0)

static void* read_block(FILE* f, int64_t* count, size_t elem)
{
    if (fread(count, sizeof(int64_t), 1, f) != 1) { fprintf(stderr, "truncated case file\n"); exit(2); }
    if (*count == 0) return NULL;
    void* p = malloc((size_t)*count * elem);
    if (fread(p, elem, (size_t)*count, f) != (size_t)*count) { fprintf(stderr, "truncated case file\n"); exit(2); }
    return p;
}

static int to_device(void** dst, const void* src, size_t bytes)
{
    *dst = NULL;
    if (!src || !bytes) return 0;
    if (hipMalloc(dst, bytes) != hipSuccess) return 1;
    return hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice) != hipSuccess;
}

int main(int argc, char** argv)
{
    if (argc < 2) { fprintf(stderr, "usage: %s case.bin\n", argv[0]); return 2; }
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 2; }
    char magic[8];
    if (fread(magic, 1, 8, f) != 8 || memcmp(magic, "EGNNCASE", 8) != 0) { fprintf(stderr, "not a case file\n"); return 2; }
    egnn_layer_desc desc;
    int32_t shape[7];                                   /* B, N, K, coor_dim, adj_kind (0 none, 1 = (N,N), 2 = (B,N,N)), 0, 0 */
    if (fread(&desc, sizeof(desc), 1, f) != 1 || fread(shape, sizeof(shape), 1, f) != 1) { fprintf(stderr, "truncated header\n"); return 2; }
    const int B = shape[0], N = shape[1], K = shape[2], C = shape[3], adj_kind = shape[4];

    const float* host_params[NPARAM];
    int64_t cnt;
    for (int i = 0; i < NPARAM; ++i) host_params[i] = (const float*)read_block(f, &cnt, 4);
    egnn_layer_params params;
    memcpy(&params, host_params, sizeof(params));       /* the struct is exactly these 17 pointers, in order */

    int64_t n_feats, n_coors, n_edges, n_mask, n_adj, n_node_out, n_coors_out;
    float* feats = (float*)read_block(f, &n_feats, 4);
    float* coors = (float*)read_block(f, &n_coors, 4);
    float* edges = (float*)read_block(f, &n_edges, 4);
    uint8_t* mask = (uint8_t*)read_block(f, &n_mask, 1);
    uint8_t* adj = (uint8_t*)read_block(f, &n_adj, 1);
    float* want_node = (float*)read_block(f, &n_node_out, 4);
    float* want_coors = (float*)read_block(f, &n_coors_out, 4);
    fclose(f);

    /* 1. weights: host re-layout, one upload */
    const size_t blob_bytes = egnn_packed_weights_bytes(&desc);
    if (!blob_bytes) { fprintf(stderr, "descriptor rejected\n"); return 1; }
    void* blob = malloc(blob_bytes);
    egnn_packed_info info;
    int rc = egnn_pack_weights_host(&desc, &params, blob, &info);
    if (rc != EGNN_OK) { fprintf(stderr, "egnn_pack_weights_host: %s\n", egnn_error_string(rc)); return 1; }

    void *d_blob, *d_feats, *d_coors, *d_edges, *d_mask, *d_adj, *d_node_out, *d_coors_out, *d_ws, *d_status;
    if (to_device(&d_blob, blob, blob_bytes) || to_device(&d_feats, feats, (size_t)n_feats * 4) ||
        to_device(&d_coors, coors, (size_t)n_coors * 4) || to_device(&d_edges, edges, (size_t)n_edges * 4) ||
        to_device(&d_mask, mask, (size_t)n_mask) || to_device(&d_adj, adj, (size_t)n_adj)) { fprintf(stderr, "upload failed\n"); return 2; }
    CHECK_HIP(hipMalloc(&d_node_out, (size_t)n_feats * 4));
    CHECK_HIP(hipMalloc(&d_coors_out, (size_t)n_coors * 4));
    CHECK_HIP(hipMalloc(&d_status, 4));
    CHECK_HIP(hipMemset(d_status, 0, 4));

    /* 2. workspace, 3. the forward */
    const size_t ws_bytes = egnn_workspace_bytes(&desc, B, N, K);
    CHECK_HIP(hipMalloc(&d_ws, ws_bytes ? ws_bytes : 256));
    rc = egnn_layer_forward_f32(&desc, &info, d_blob, (const float*)d_feats, (const float*)d_coors, (const float*)d_edges,
                                (const uint8_t*)d_mask, (const uint8_t*)d_adj, adj_kind == 2 ? (int64_t)N * N : 0, B, N, K, C,
                                (float*)d_node_out, (float*)d_coors_out, d_ws, ws_bytes, (int32_t*)d_status, NULL);
    if (rc != EGNN_OK) { fprintf(stderr, "egnn_layer_forward_f32: %d (%s)\n", rc, egnn_error_string(rc)); return 1; }
    CHECK_HIP(hipDeviceSynchronize());

    float* got_node = (float*)malloc((size_t)n_feats * 4);
    float* got_coors = (float*)malloc((size_t)n_coors * 4);
    int32_t status = 0;
    CHECK_HIP(hipMemcpy(got_node, d_node_out, (size_t)n_feats * 4, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(got_coors, d_coors_out, (size_t)n_coors * 4, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(&status, d_status, 4, hipMemcpyDeviceToHost));

    double err_n = 0, err_c = 0;
    for (int64_t i = 0; i < n_feats; ++i) { double d = fabs((double)got_node[i] - want_node[i]); if (!(d <= err_n)) err_n = d; }
    for (int64_t i = 0; i < n_coors; ++i) { double d = fabs((double)got_coors[i] - want_coors[i]); if (!(d <= err_c)) err_c = d; }
    printf("B=%d N=%d K=%d dim=%d: blob %zu bytes, workspace %zu bytes, range status %d, max|d feats| = %.3e, max|d coors| = %.3e\n",
           B, N, K, desc.dim, blob_bytes, ws_bytes, status, err_n, err_c);
    const int ok = status == 0 && err_n <= 1e-4 && err_c <= 1e-4;
    puts(ok ? "OK" : "MISMATCH");
    return ok ? 0 : 1;
}
