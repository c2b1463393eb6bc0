// CSC / CSR construction: stable sort of the COO edge list by destination and by source.
// rocPRIM's LSD radix sort (through hipcub) is stable, so edges of one segment keep
// ascending original edge id -- the order DGL's COO->CSR conversion and the CPU oracle use.
#include "common.h"
#include <hipcub/hipcub.hpp>

namespace {

__global__ void k_keys(int64_t E, const int64_t *__restrict__ key64, int32_t *__restrict__ key32,
                       int32_t *__restrict__ iota) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < E; i += stride) {
        key32[i] = (int32_t)key64[i];
        iota[i] = (int32_t)i;
    }
}

// other_sorted[i] = (int32) other64[eid[i]]
__global__ void k_gather(int64_t E, const int64_t *__restrict__ other64, const int32_t *__restrict__ eid,
                         int32_t *__restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < E; i += stride) out[i] = (int32_t)other64[eid[i]];
}

// ptr[v] = index of the first sorted edge whose key is >= v   (ptr[N] = E)
__global__ void k_ptr(int64_t N, int64_t E, const int32_t *__restrict__ sorted_key, int32_t *__restrict__ ptr) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    if (E == 0) {
        for (; i <= N; i += stride) ptr[i] = 0;
        return;
    }
    for (; i < E; i += stride) {
        int32_t k = sorted_key[i];
        int32_t kprev = i ? sorted_key[i - 1] : -1;
        for (int32_t v = kprev + 1; v <= k; ++v) ptr[v] = (int32_t)i;
        if (i == E - 1)
            for (int64_t v = (int64_t)k + 1; v <= N; ++v) ptr[v] = (int32_t)E;
    }
}

int bits_for(int64_t n) {
    int b = 1;
    while (b < 31 && ((int64_t)1 << b) < n) ++b;
    return b;
}

size_t sort_temp_bytes(int64_t E, int end_bit) {
    size_t bytes = 0;  // stays 0 if the query fails: the sort itself then reports the error
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, (const int32_t *)nullptr, (int32_t *)nullptr,
                                       (const int32_t *)nullptr, (int32_t *)nullptr, (int)E, 0, end_bit, nullptr);
    return bytes;
}

}  // namespace

extern "C" {

int magic_csr_build_workspace_bytes(int64_t N, int64_t E, size_t *bytes) {
    if (!bytes || N < 0 || E < 0 || N >= INT32_MAX || E >= INT32_MAX) return MAGIC_ERR_INVALID_ARG;
    size_t e = (size_t)(E > 0 ? E : 1);
    *bytes = 3 * magic::align_up(e * sizeof(int32_t)) + magic::align_up(sort_temp_bytes(e, bits_for(N + 1))) + 256;
    return MAGIC_OK;
}

int magic_csr_build(int64_t N, int64_t E, const int64_t *src, const int64_t *dst, int32_t *in_ptr,
                    int32_t *in_src, int32_t *in_eid, int32_t *out_ptr, int32_t *out_dst, int32_t *out_eid,
                    void *workspace, size_t workspace_bytes, void *stream_) {
    if (N < 0 || E < 0 || N >= INT32_MAX || E >= INT32_MAX) return MAGIC_ERR_INVALID_ARG;
    if (!in_ptr || !out_ptr) return MAGIC_ERR_INVALID_ARG;
    if (E > 0 && (!src || !dst || !in_src || !in_eid || !out_dst || !out_eid)) return MAGIC_ERR_INVALID_ARG;
    hipStream_t stream = (hipStream_t)stream_;
    int grid = magic::stride_grid(E > N ? E : N + 1, 256);
    if (E == 0) {
        MAGIC_LAUNCH("csr_ptr", k_ptr, grid, 256, 0, stream, N, E, (const int32_t *)nullptr, in_ptr);
        MAGIC_LAUNCH("csr_ptr", k_ptr, grid, 256, 0, stream, N, E, (const int32_t *)nullptr, out_ptr);
        return magic::launch_status();
    }
    magic::Arena ar(workspace, workspace_bytes);
    int32_t *key = ar.take<int32_t>(E);
    int32_t *key_sorted = ar.take<int32_t>(E);
    int32_t *iota = ar.take<int32_t>(E);
    int end_bit = bits_for(N + 1);
    size_t temp_bytes = sort_temp_bytes(E, end_bit);
    char *temp = ar.take<char>(temp_bytes);
    if (!ar.ok() || !key || !key_sorted || !iota || !temp) return MAGIC_ERR_WORKSPACE;

    for (int dir = 0; dir < 2; ++dir) {
        const int64_t *k64 = dir == 0 ? dst : src;
        const int64_t *o64 = dir == 0 ? src : dst;
        int32_t *ptr = dir == 0 ? in_ptr : out_ptr;
        int32_t *oth = dir == 0 ? in_src : out_dst;
        int32_t *eid = dir == 0 ? in_eid : out_eid;
        MAGIC_LAUNCH("csr_keys", k_keys, grid, 256, 0, stream, E, k64, key, iota);
        {
            magic::ProfScope ps("csr_radix_sort", stream);
            if (hipcub::DeviceRadixSort::SortPairs(temp, temp_bytes, key, key_sorted, iota, eid, (int)E, 0,
                                                   end_bit, stream) != hipSuccess)
                return MAGIC_ERR_RUNTIME;
        }
        MAGIC_LAUNCH("csr_gather", k_gather, grid, 256, 0, stream, E, o64, eid, oth);
        MAGIC_LAUNCH("csr_ptr", k_ptr, grid, 256, 0, stream, N, E, key_sorted, ptr);
    }
    return magic::launch_status();
}

}  // extern "C"
