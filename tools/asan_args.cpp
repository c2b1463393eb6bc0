// Host-side AddressSanitizer driver for libmagic_hip.so (built by `make -C magic_amd/csrc asan`).
// Calls the argument checking and the empty-input paths of the C ABI -- every call here must return before a
// kernel is launched or device memory is touched, so it runs on a machine without a GPU -- and checks the
// status codes.  What ASan watches is the host code of the wrappers: plan structs, argument copies, name tables.
#include <cstdio>
#include <cstring>
#include <vector>
#include "../include/magic_hip.h"

static int failures = 0;
#define EXPECT(call, want)                                                                  \
    do {                                                                                    \
        int got_ = (call);                                                                  \
        if (got_ != (want)) {                                                               \
            std::printf("FAIL %s -> %d (%s), wanted %d\n", #call, got_, magic_status_string(got_), (want)); \
            ++failures;                                                                     \
        }                                                                                   \
    } while (0)

int main() {
    if (magic_abi_version() < 3) return 2;
    for (int s = 0; s >= -7; --s) (void)std::strlen(magic_status_string(s));
    // profiler tables
    EXPECT(magic_prof_enable(0), MAGIC_OK);
    EXPECT(magic_prof_reset(), MAGIC_OK);
    std::vector<char> names(4 * 64);
    int64_t launches[4];
    double ms[4];
    EXPECT(magic_prof_read(4, names.data(), launches, ms), 0);
    EXPECT(magic_prof_read(4, nullptr, launches, ms), MAGIC_ERR_INVALID_ARG);
    EXPECT(magic_prof_sequence(4, names.data()), 0);
    EXPECT(magic_prof_sequence(-1, names.data()), MAGIC_ERR_INVALID_ARG);
    // CSR build
    size_t bytes = 0;
    EXPECT(magic_csr_build_workspace_bytes(10, 20, &bytes), MAGIC_OK);
    EXPECT(magic_csr_build_workspace_bytes(-1, 20, &bytes), MAGIC_ERR_INVALID_ARG);
    EXPECT(magic_csr_build(-1, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr),
           MAGIC_ERR_INVALID_ARG);
    EXPECT(magic_csr_build(4, 4, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr),
           MAGIC_ERR_INVALID_ARG);
    // projections
    float dummy[16] = {0};
    EXPECT(magic_proj_fwd(-1, 8, 8, 0, 0, dummy, 8, nullptr, 1.f, dummy, dummy, 8, nullptr, nullptr, 0, nullptr, 0, nullptr),
           MAGIC_ERR_INVALID_ARG);
    EXPECT(magic_proj_fwd(0, 8, 8, 0, 0, nullptr, 8, nullptr, 1.f, nullptr, nullptr, 8, nullptr, nullptr, 0, nullptr, 0, nullptr),
           MAGIC_OK);
    EXPECT(magic_proj_fwd(4, 8, 8, 2, 0, dummy, 4, nullptr, 1.f, dummy, dummy, 8, nullptr, nullptr, 0, nullptr, 0, nullptr),
           MAGIC_ERR_INVALID_ARG);  // ld_x < in_dim
    EXPECT(magic_proj_fwd_attn(-1, 8, 8, 0, 0, dummy, 8, nullptr, 1.f, dummy, dummy, 8, nullptr, nullptr, 0, dummy, dummy,
                               nullptr, 0, nullptr),
           MAGIC_ERR_INVALID_ARG);
    EXPECT(magic_proj_fwd_attn(4, 8, 8, 2, 0, dummy, 8, nullptr, 1.f, dummy, dummy, 8, nullptr, nullptr, 0, dummy, dummy,
                               nullptr, 0, nullptr),
           MAGIC_ERR_INVALID_ARG);  // score columns asked for, no score output
    (void)magic_proj_fwd_workspace_bytes(256, 264);
    (void)magic_proj_bwd_input_workspace_bytes(1000, 264, 256);
    (void)magic_proj_bwd_weight_workspace_bytes(1000, 264, 256);
    EXPECT(magic_proj_bwd_input(0, 8, 8, nullptr, 8, nullptr, nullptr, nullptr, 8, nullptr, 0, nullptr), MAGIC_OK);
    EXPECT(magic_proj_bwd_input(4, 8, 8, nullptr, 8, dummy, nullptr, dummy, 8, nullptr, 0, nullptr), MAGIC_ERR_INVALID_ARG);
    magic_proj_epilogue ep;
    std::memset(&ep, 0, sizeof ep);
    ep.add = dummy;
    ep.ld_add = 4;  // narrower than out_dim
    EXPECT(magic_proj_bwd_input(4, 8, 8, dummy, 8, dummy, &ep, dummy, 8, nullptr, 0, nullptr), MAGIC_ERR_INVALID_ARG);
    EXPECT(magic_proj_bwd_weight(4, 8, 8, dummy, 8, dummy, 8, nullptr, 1.f, nullptr, nullptr, 0, nullptr),
           MAGIC_ERR_INVALID_ARG);
    EXPECT(magic_proj_bwd_weight(4, 8, 8, nullptr, 8, dummy, 8, nullptr, 1.f, dummy, nullptr, 0, nullptr),
           MAGIC_ERR_INVALID_ARG);
    // sparse layer: sizes and plans
    (void)magic_gat_fwd_workspace_bytes(1000, 4, 4, 16);
    magic_plan plan;
    std::memset(&plan, 0, sizeof plan);
    EXPECT(magic_gat_fwd(-1, 0, 4, 16, &plan, nullptr, nullptr, nullptr, 64, nullptr, nullptr, 8, nullptr, nullptr, nullptr,
                         0.2f, nullptr, nullptr, 0, nullptr, 1.f, nullptr, nullptr, 64, nullptr, nullptr, nullptr, nullptr,
                         0, nullptr),
           MAGIC_ERR_INVALID_ARG);
    EXPECT(magic_gat_fwd(4, 4, 4, 16, nullptr, nullptr, nullptr, nullptr, 64, nullptr, nullptr, 8, nullptr, nullptr, nullptr,
                         0.2f, nullptr, nullptr, 0, nullptr, 1.f, nullptr, nullptr, 64, nullptr, nullptr, nullptr, nullptr,
                         0, nullptr),
           MAGIC_ERR_INVALID_ARG);
    // sweeps restricted to marked destinations: the attention output is refused, a missing plan too
    EXPECT(magic_gat_fwd_active(4, 4, 4, 16, nullptr, nullptr, nullptr, nullptr, 64, nullptr, nullptr, 8, nullptr, nullptr,
                                nullptr, 0.2f, nullptr, nullptr, 0, nullptr, 1.f, nullptr, nullptr, 64, nullptr, nullptr,
                                nullptr, (const uint8_t *)dummy, nullptr, 0, nullptr),
           MAGIC_ERR_INVALID_ARG);
    EXPECT(magic_gat_bwd_active(4, 4, 4, 16, nullptr, nullptr, nullptr, nullptr, nullptr, dummy, 64, dummy, 4, dummy, dummy,
                                nullptr, nullptr, 17, 0.2f, nullptr, nullptr, 0, nullptr, dummy, 64, dummy, 64, dummy, 72,
                                nullptr, nullptr, nullptr, (const uint8_t *)dummy, nullptr, 0, nullptr),
           MAGIC_ERR_INVALID_ARG);
    // coded first-layer input
    if (magic_coded_table_rows(5) != 5 + 1 + 32 || magic_coded_table_rows(MAGIC_CODED_MAX_DIM + 1) != 0) ++failures;
    EXPECT(magic_coded_table(0, 136, dummy, 5, dummy, 1.f, dummy, 136, nullptr), MAGIC_ERR_INVALID_ARG);
    EXPECT(magic_coded_table(5, 136, dummy, 4, dummy, 1.f, dummy, 136, nullptr), MAGIC_ERR_INVALID_ARG);  // ld_w < n_dim
    EXPECT(magic_node_code(10, MAGIC_CODED_MAX_DIM + 1, (const int64_t *)dummy, nullptr, 0, nullptr, (uint32_t *)dummy, nullptr),
           MAGIC_ERR_INVALID_ARG);
    EXPECT(magic_node_code(0, 5, nullptr, nullptr, 0, nullptr, nullptr, nullptr), MAGIC_OK);
    magic_coded_input ci;
    std::memset(&ci, 0, sizeof ci);
    EXPECT(magic_gat_fwd_coded(4, 4, 4, 16, &plan, nullptr, nullptr, dummy, nullptr, 0.2f, nullptr, nullptr, dummy, 64, nullptr,
                               nullptr, nullptr, 0, nullptr),
           MAGIC_ERR_INVALID_ARG);
    ci.code = (const uint32_t *)dummy, ci.table = dummy, ci.token = dummy, ci.n_dim = 5, ci.ld_table = 100;  // narrower than 2C+2H
    plan.ptr = (const int32_t *)dummy;
    EXPECT(magic_gat_fwd_coded(4, 4, 4, 16, &plan, nullptr, &ci, dummy, nullptr, 0.2f, nullptr, nullptr, dummy, 64, nullptr,
                               nullptr, nullptr, 0, nullptr),
           MAGIC_ERR_INVALID_ARG);
    ci.ld_table = 136;
    EXPECT(magic_gat_fwd_coded(4, 4, 1, 8, &plan, nullptr, &ci, dummy, nullptr, 0.2f, nullptr, nullptr, dummy, 8, nullptr, nullptr,
                               nullptr, 0, nullptr),
           MAGIC_ERR_UNSUPPORTED);  // compiled for four heads
    (void)magic_coded_weight_grad_workspace_bytes(136, 5);
    EXPECT(magic_coded_weight_grad(10, 136, nullptr, dummy, 136, dummy, 5, dummy, 5, nullptr, nullptr, 0, nullptr),
           MAGIC_ERR_INVALID_ARG);
    EXPECT(magic_coded_weight_grad(10, 136, &ci, dummy, 100, dummy, 5, dummy, 5, nullptr, nullptr, 0, nullptr),
           MAGIC_ERR_INVALID_ARG);  // ld_dy < m
    EXPECT(magic_coded_weight_grad(10, 136, &ci, dummy, 136, dummy, 5, dummy, 5, nullptr, nullptr, 0, nullptr),
           MAGIC_ERR_WORKSPACE);
    plan.ptr = nullptr;
    // loss, read-out, sampling helpers, scatter
    (void)magic_sce_workspace_bytes(100, 8);
    EXPECT(magic_sce_fwd_bwd(-1, 8, 0, nullptr, nullptr, 8, nullptr, 8, 3.f, 1.f, nullptr, nullptr, nullptr, nullptr, 0, nullptr),
           MAGIC_ERR_INVALID_ARG);
    EXPECT(magic_sce_fwd_bwd_static(10, MAGIC_SCE_STATIC_MAX_DIM + 1, 5, (const int64_t *)dummy, dummy, 17, dummy, 17, 3.f, 1.f,
                                    dummy, dummy, nullptr, nullptr),
           MAGIC_ERR_STATIC_LIMIT);
    (void)magic_readout_workspace_bytes(1000, 256, 70);
    EXPECT(magic_readout(-1, 8, dummy, 8, nullptr, 0, 1, dummy, nullptr, 0, nullptr), MAGIC_ERR_INVALID_ARG);
    EXPECT(magic_readout(4, 8, dummy, 4, nullptr, 0, 1, dummy, nullptr, 0, nullptr), MAGIC_ERR_INVALID_ARG);
    EXPECT(magic_readout(4, 8, dummy, 8, nullptr, 3, 1, dummy, nullptr, 0, nullptr), MAGIC_ERR_INVALID_ARG);
    (void)magic_rows_scatter_add_workspace_bytes(100, 50);
    EXPECT(magic_rows_scatter_add(-1, 4, 8, nullptr, nullptr, 8, nullptr, 8, nullptr, 0, nullptr), MAGIC_ERR_INVALID_ARG);
    EXPECT(magic_dropout_mask(-1, 0.9f, 1, nullptr, nullptr), MAGIC_ERR_INVALID_ARG);
    // optimizer: slot table checks
    magic_adam_slot slots[2];
    std::memset(slots, 0, sizeof slots);
    slots[0].n = 5;  // pointers missing
    EXPECT(magic_adam_step(1, slots, (int64_t *)dummy, 1e-3, 0.9, 0.999, 1e-8, 5e-4, nullptr), MAGIC_ERR_INVALID_ARG);
    EXPECT(magic_adam_step(1, slots, nullptr, 1e-3, 0.9, 0.999, 1e-8, 5e-4, nullptr), MAGIC_ERR_INVALID_ARG);
    EXPECT(magic_adam_step(-1, slots, (int64_t *)dummy, 1e-3, 0.9, 0.999, 1e-8, 5e-4, nullptr), MAGIC_ERR_INVALID_ARG);
    // KNN
    (void)magic_knn_workspace_bytes(2604530, 684107, 64, 10);
    (void)magic_knn_workspace_bytes(100000, 1000, 64, 200);
    EXPECT(magic_knn_topk(0, 4, 64, 10, dummy, 64, dummy, 64, 0, nullptr, nullptr, nullptr, 0, nullptr), MAGIC_ERR_INVALID_ARG);
    EXPECT(magic_knn_topk(100, 4, 64, 200, dummy, 64, dummy, 64, 0, (double *)dummy, (int64_t *)dummy, nullptr, 0, nullptr),
           MAGIC_ERR_INVALID_ARG);  // k > n_train
    EXPECT(magic_knn_merge(4, 0, 2, nullptr, nullptr, nullptr, nullptr, nullptr), MAGIC_ERR_INVALID_ARG);
    std::printf(failures ? "asan_args: %d failure(s)\n" : "asan_args: ok\n", failures);
    return failures ? 1 : 0;
}
