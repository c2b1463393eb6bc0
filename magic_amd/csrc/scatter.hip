// dst[idx[s], :] += src[s, :] for a list of row indices that may repeat, reproducibly:
// (row, position) pairs go through a stable LSD radix sort over just the bits a row index needs,
// then the first position of every run of equal rows adds the run's contributions in position
// order.  No floating-point atomics.  Used for the gradient of the hidden rows gathered for the
// sampled edge reconstruction (model/autoencoder.py:257-258 of the reference:
// enc_rep[sample_src], enc_rep[sample_dst]).
#include "common.h"
#include <hipcub/hipcub.hpp>

namespace {

__global__ __launch_bounds__(256) void k_scatter_keys(int64_t n, const int64_t *idx, int32_t *key, int32_t *pos) {
    for (int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x; s < n; s += (int64_t)gridDim.x * 256) {
        key[s] = (int32_t)idx[s];
        pos[s] = (int32_t)s;
    }
}

// One 256-thread block per (position, 256-column slab).  The block of a run head adds the run's
// rows in position order, one column per thread, four loads in flight; every other block exits.
// The summation order depends only on the run.
__global__ __launch_bounds__(256) void k_scatter_add(int n, int width, const int32_t *key, const int32_t *pos,
                                                     const float *src, int64_t ld_src, float *dst, int64_t ld_dst) {
    const int i = blockIdx.x;
    const int r = key[i];
    if (i > 0 && key[i - 1] == r) return;  // not the head of its run
    const int lane = threadIdx.x & 63;
    int end = i + 1;  // end of the run: 64 candidates per step (every wave scans for itself)
    for (;;) {
        const int j = end + lane;
        const unsigned long long same = __ballot(j < n && key[j] == r);
        if (same == ~0ull) {
            end += 64;
            continue;
        }
        end += __ffsll((long long)~same) - 1;
        break;
    }
    const int c = blockIdx.y * 256 + threadIdx.x;
    if (c >= width) return;
    float acc = dst[(int64_t)r * ld_dst + c];
    int j = i;
    for (; j + 4 <= end; j += 4) {
        const float v0 = src[(int64_t)pos[j] * ld_src + c], v1 = src[(int64_t)pos[j + 1] * ld_src + c],
                    v2 = src[(int64_t)pos[j + 2] * ld_src + c], v3 = src[(int64_t)pos[j + 3] * ld_src + c];
        acc = (((acc + v0) + v1) + v2) + v3;
    }
    for (; j < end; ++j) acc += src[(int64_t)pos[j] * ld_src + c];
    dst[(int64_t)r * ld_dst + c] = acc;
}

// float4 form: one wave per position, the whole row width in passes of up to four 256-column slabs
// (lane l owns columns [4l, 4l+4) of every slab; the loads of a pass are issued together).
// Needs width % 4 == 0 and 16-byte aligned rows on both sides.
__global__ __launch_bounds__(256) void k_scatter_add4(int n, int width, const int32_t *key, const int32_t *pos,
                                                      const float *src, int64_t ld_src, float *dst, int64_t ld_dst) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    const int r = key[i];
    if (i > 0 && key[i - 1] == r) return;  // not the head of its run
    int end = i + 1;
    for (;;) {
        const int j = end + lane;
        const unsigned long long same = __ballot(j < n && key[j] == r);
        if (same == ~0ull) {
            end += 64;
            continue;
        }
        end += __ffsll((long long)~same) - 1;
        break;
    }
    float *drow = dst + (int64_t)r * ld_dst;
    for (int c0 = lane * 4; c0 < width; c0 += 1024) {
        float4 acc[4];
        bool on[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            on[q] = c0 + 256 * q < width;
            if (on[q]) acc[q] = *reinterpret_cast<const float4 *>(drow + c0 + 256 * q);
        }
        int j = i;
        for (; j + 4 <= end; j += 4) {  // long runs (hub rows): four rows in flight, added in position order
            const float *srow[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) srow[u] = src + (int64_t)pos[j + u] * ld_src + c0;
            float4 v[4][4];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (on[q]) v[u][q] = *reinterpret_cast<const float4 *>(srow[u] + 256 * q);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (on[q])
                        acc[q].x += v[u][q].x, acc[q].y += v[u][q].y, acc[q].z += v[u][q].z, acc[q].w += v[u][q].w;
        }
        for (; j < end; ++j) {
            const float *srow = src + (int64_t)pos[j] * ld_src + c0;
            float4 v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (on[q]) v[q] = *reinterpret_cast<const float4 *>(srow + 256 * q);
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (on[q]) acc[q].x += v[q].x, acc[q].y += v[q].y, acc[q].z += v[q].z, acc[q].w += v[q].w;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (on[q]) *reinterpret_cast<float4 *>(drow + c0 + 256 * q) = acc[q];
    }
}

int bits_for(int64_t n) {
    int b = 1;
    while (b < 31 && ((int64_t)1 << b) < n) ++b;
    return b;
}

size_t sort_temp_bytes(int64_t n, int end_bit) {
    size_t bytes = 0;  // stays 0 if the query fails: the sort itself then reports the error
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, (const int32_t *)nullptr, (int32_t *)nullptr,
                                       (const int32_t *)nullptr, (int32_t *)nullptr, (int)n, 0, end_bit);
    return bytes;
}

}  // namespace

extern "C" {

size_t magic_rows_scatter_add_workspace_bytes(int64_t n_src, int64_t n_rows) {
    const size_t n = (size_t)(n_src > 0 ? n_src : 1);
    return 4 * magic::align_up(n * sizeof(int32_t)) + magic::align_up(sort_temp_bytes((int64_t)n, bits_for(n_rows))) + 256;
}

int magic_rows_scatter_add(int64_t n_src, int64_t n_rows, int32_t width, const int64_t *idx, const float *src,
                           int64_t ld_src, float *dst, int64_t ld_dst, void *workspace, size_t workspace_bytes,
                           void *stream_) {
    if (n_src < 0 || n_rows < 0 || width <= 0) return MAGIC_ERR_INVALID_ARG;
    if (n_src == 0) return MAGIC_OK;
    if (!idx || !src || !dst || ld_src < width || ld_dst < width) return MAGIC_ERR_INVALID_ARG;
    if (n_src > 0x7fffffff || n_rows > 0x7fffffff) return MAGIC_ERR_UNSUPPORTED;
    hipStream_t stream = (hipStream_t)stream_;
    const int end_bit = bits_for(n_rows);
    magic::Arena ar(workspace, workspace_bytes);
    int32_t *key = ar.take<int32_t>((size_t)n_src), *pos = ar.take<int32_t>((size_t)n_src);
    int32_t *key_s = ar.take<int32_t>((size_t)n_src), *pos_s = ar.take<int32_t>((size_t)n_src);
    size_t temp_bytes = sort_temp_bytes(n_src, end_bit);
    char *temp = ar.take<char>(temp_bytes);
    if (!ar.ok() || !key || !pos || !key_s || !pos_s || !temp) return MAGIC_ERR_WORKSPACE;
    MAGIC_LAUNCH("rows_scatter_keys", k_scatter_keys, magic::stride_grid(n_src, 256, 1024), 256, 0, stream, n_src, idx,
                 key, pos);
    {
        magic::ProfScope ps("rows_scatter_sort", stream);
        if (hipcub::DeviceRadixSort::SortPairs(temp, temp_bytes, key, key_s, pos, pos_s, (int)n_src, 0, end_bit,
                                               stream) != hipSuccess)
            return MAGIC_ERR_RUNTIME;
    }
    // (narrow rows leave most of a wave idle in the float4 form: measured 74 us against 47 us at width 192)
    const bool vec = width >= 512 && (width % 4 == 0) && (ld_src % 4 == 0) && (ld_dst % 4 == 0) && ((uintptr_t)src % 16 == 0) &&
                     ((uintptr_t)dst % 16 == 0);
    if (vec)
        MAGIC_LAUNCH("rows_scatter_add", k_scatter_add4, (unsigned)((n_src + 3) / 4), 256, 0, stream, (int)n_src, width,
                     key_s, pos_s, src, ld_src, dst, ld_dst);
    else
        MAGIC_LAUNCH("rows_scatter_add", k_scatter_add, dim3((unsigned)n_src, (unsigned)((width + 255) / 256)), 256, 0,
                     stream, (int)n_src, width, key_s, pos_s, src, ld_src, dst, ld_dst);
    return magic::launch_status();
}

}  // extern "C"
