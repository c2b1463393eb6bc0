// Assembly of a union graph from device-resident component graphs (dgl.batch, model/train.py:39 of the
// reference) in a handful of launches: the loop that trains on batches of graphs is bound by its host work,
// and concatenating ~20 index arrays of 12 components with torch costs ~60 launches per batch.
//
//   magic_concat_offset   every (array, component) pair is one job {source, destination, length, addend}:
//                         dst[i] = src[i] + add.  The job table lives in device memory (one small
//                         asynchronous copy); one launch runs all jobs of one element width.
//   magic_walk_order      the walk order of the node-driven sweeps (nodes by XCD range, then by segment-length
//                         class) and the 8 range starts, for up to four segment arrays at once, without a host
//                         round trip: cost scan, 11-bit keys, one stable radix sort.
#include "common.h"
#include <hipcub/hipcub.hpp>

namespace {

struct ConcatJob {
    const void *src;
    void *dst;
    int64_t len;
    int64_t add;
};
static_assert(sizeof(ConcatJob) == 32, "job layout is part of the C ABI");

template <typename T>
__global__ __launch_bounds__(256) void k_concat(const ConcatJob *__restrict__ jobs, int n_jobs) {
    for (int j = blockIdx.y; j < n_jobs; j += gridDim.y) {
        const ConcatJob job = jobs[j];
        const T *s = static_cast<const T *>(job.src);
        T *d = static_cast<T *>(job.dst);
        const T add = (T)job.add;
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < job.len; i += (int64_t)gridDim.x * 256)
            d[i] = s ? (T)(s[i] + add) : add;  // a job without a source fills its range with the addend
    }
}

// ---- walk order ------------------------------------------------------------------------------------
// Costs, keys and values of all segment arrays sit back to back (array s from position base[s]) so that one
// scan and one sort serve them all; the array index is the top of the sort key.
constexpr int MAX_WALK = 4;
struct WalkArgs {
    const int32_t *ptr[MAX_WALK];
    int32_t *order[MAX_WALK];
    int32_t *xcd_start[MAX_WALK];
    int32_t n[MAX_WALK];
    int32_t base[MAX_WALK];
    int n_seg;
    int32_t total_n;
    int32_t light_max;
};

__device__ __forceinline__ int seg_of(const WalkArgs &a, int32_t i) {
    int s = 0;
#pragma unroll
    for (int k = 1; k < MAX_WALK; ++k)
        if (k < a.n_seg && i >= a.base[k]) s = k;
    return s;
}

// cost of node i: its edges when the segment is light, plus 2 for its own row traffic
__global__ __launch_bounds__(256) void k_walk_cost(WalkArgs a, int64_t *__restrict__ cost) {
    for (int32_t i = blockIdx.x * 256 + threadIdx.x; i < a.total_n; i += gridDim.x * 256) {
        const int s = seg_of(a, i);
        const int32_t v = i - a.base[s];
        const int32_t deg = a.ptr[s][v + 1] - a.ptr[s][v];
        cost[i] = (int64_t)(deg <= a.light_max ? deg : 0) + 2;
    }
}

// XCD range of the node at position i of array s: work before the node x 8 / total work of the array, at most 7
__device__ __forceinline__ int xcd_of(const WalkArgs &a, const int64_t *incl, const int64_t *cost, int s, int32_t i) {
    const int64_t before_seg = a.base[s] ? incl[a.base[s] - 1] : 0;
    int64_t total = incl[a.base[s] + a.n[s] - 1] - before_seg;
    if (total < 1) total = 1;
    const int64_t work = incl[i] - cost[i] - before_seg;
    const int64_t x = work * 8 / total;
    return (int)(x < 7 ? x : 7);
}

// length class: 0 (empty), 1 (1-4 edges), then ceil(log2(degree))
__device__ __forceinline__ int class_of(int32_t deg) {
    if (deg <= 0) return 0;
    if (deg <= 4) return 1;
    return 32 - __clz(deg - 1);
}

__global__ __launch_bounds__(256) void k_walk_keys(WalkArgs a, const int64_t *__restrict__ incl,
                                                   const int64_t *__restrict__ cost, int32_t *__restrict__ key,
                                                   int32_t *__restrict__ val) {
    for (int32_t i = blockIdx.x * 256 + threadIdx.x; i < a.total_n; i += gridDim.x * 256) {
        const int s = seg_of(a, i);
        const int32_t v = i - a.base[s];
        const int32_t deg = a.ptr[s][v + 1] - a.ptr[s][v];
        const int x = xcd_of(a, incl, cost, s, i);
        key[i] = (s * 8 + x) * 64 + class_of(deg);
        val[i] = v;
        // range starts: xcd_start[r] = number of nodes whose range is < r (the range is non-decreasing in v)
        int32_t *xs = a.xcd_start[s];
        const int xp = v ? xcd_of(a, incl, cost, s, i - 1) : -1;
        for (int r = xp + 1; r <= x; ++r) xs[r] = v;
        if (v == a.n[s] - 1)
            for (int r = x + 1; r <= 8; ++r) xs[r] = a.n[s];
    }
}

// order of array s = the values of its slice of the sorted pairs
__global__ __launch_bounds__(256) void k_walk_scatter(WalkArgs a, const int32_t *__restrict__ val_sorted) {
    for (int32_t i = blockIdx.x * 256 + threadIdx.x; i < a.total_n; i += gridDim.x * 256) {
        const int s = seg_of(a, i);
        a.order[s][i - a.base[s]] = val_sorted[i];
    }
}

constexpr int KEY_BITS = 11;  // 4 arrays x 8 ranges x 64 classes

struct WalkWs {
    int64_t *cost, *incl;
    int32_t *key_in, *key_out, *val_in, *val_out;
    void *cub;
    size_t cub_bytes, total_bytes;
};

bool carve_walk(void *ws, size_t bytes, int32_t total_n, WalkWs *w) {
    size_t scan_b = 0, sort_b = 0;
    (void)hipcub::DeviceScan::InclusiveSum(nullptr, scan_b, (const int64_t *)nullptr, (int64_t *)nullptr, (int)total_n,
                                           nullptr);
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, sort_b, (const int32_t *)nullptr, (int32_t *)nullptr,
                                       (const int32_t *)nullptr, (int32_t *)nullptr, (int)total_n, 0, KEY_BITS, nullptr);
    w->cub_bytes = (scan_b > sort_b ? scan_b : sort_b) + 256;
    size_t off = 0;
    auto take = [&](size_t b) {
        const size_t o = off;
        off += magic::align_up(b);
        return o;
    };
    const size_t n = (size_t)(total_n > 0 ? total_n : 1);
    const size_t o_cost = take(n * 8), o_incl = take(n * 8), o_ki = take(n * 4), o_ko = take(n * 4), o_vi = take(n * 4),
                 o_vo = take(n * 4), o_cub = take(w->cub_bytes);
    w->total_bytes = off;
    if (ws == nullptr) return true;  // size query
    const size_t skew = (256 - ((uintptr_t)ws & 255)) & 255;
    if (off + skew > bytes) return false;
    char *p = static_cast<char *>(ws) + skew;
    w->cost = reinterpret_cast<int64_t *>(p + o_cost);
    w->incl = reinterpret_cast<int64_t *>(p + o_incl);
    w->key_in = reinterpret_cast<int32_t *>(p + o_ki);
    w->key_out = reinterpret_cast<int32_t *>(p + o_ko);
    w->val_in = reinterpret_cast<int32_t *>(p + o_vi);
    w->val_out = reinterpret_cast<int32_t *>(p + o_vo);
    w->cub = p + o_cub;
    return true;
}

}  // namespace

extern "C" {

int magic_concat_offset(int32_t elem_bytes, int32_t n_jobs, const void *jobs_device, int64_t max_len, void *stream_) {
    if (n_jobs < 0 || (elem_bytes != 4 && elem_bytes != 8)) return MAGIC_ERR_INVALID_ARG;
    if (n_jobs == 0 || max_len <= 0) return MAGIC_OK;
    if (!jobs_device) return MAGIC_ERR_INVALID_ARG;
    hipStream_t stream = (hipStream_t)stream_;
    int64_t gx = (max_len + 256 * 8 - 1) / (256 * 8);
    if (gx < 1) gx = 1;
    if (gx > 256) gx = 256;
    dim3 grid((unsigned)gx, (unsigned)(n_jobs < 1024 ? n_jobs : 1024));
    const ConcatJob *jobs = static_cast<const ConcatJob *>(jobs_device);
    if (elem_bytes == 4)
        MAGIC_LAUNCH("concat_offset", k_concat<int32_t>, grid, 256, 0, stream, jobs, n_jobs);
    else
        MAGIC_LAUNCH("concat_offset", k_concat<int64_t>, grid, 256, 0, stream, jobs, n_jobs);
    return magic::launch_status();
}

size_t magic_walk_order_workspace_bytes(int64_t total_nodes) {
    if (total_nodes < 0 || total_nodes >= INT32_MAX) return 0;
    WalkWs w;
    carve_walk(nullptr, 0, (int32_t)total_nodes, &w);
    return w.total_bytes + 512;
}

int magic_walk_order(int32_t n_arrays, const int32_t *const *ptr, const int32_t *n_nodes, int32_t light_max,
                     int32_t *const *order, int32_t *const *xcd_start, void *workspace, size_t workspace_bytes,
                     void *stream_) {
    if (n_arrays < 1 || n_arrays > MAX_WALK || !ptr || !n_nodes || !order || !xcd_start) return MAGIC_ERR_INVALID_ARG;
    hipStream_t stream = (hipStream_t)stream_;
    WalkArgs a;
    a.n_seg = n_arrays;
    a.light_max = light_max;
    int64_t total = 0;
    for (int s = 0; s < n_arrays; ++s) {
        if (n_nodes[s] <= 0 || !xcd_start[s] || !ptr[s] || !order[s]) return MAGIC_ERR_INVALID_ARG;
        a.ptr[s] = ptr[s];
        a.order[s] = order[s];
        a.xcd_start[s] = xcd_start[s];
        a.n[s] = n_nodes[s];
        a.base[s] = (int32_t)total;
        total += n_nodes[s];
    }
    for (int s = n_arrays; s < MAX_WALK; ++s) {
        a.ptr[s] = a.ptr[0], a.order[s] = a.order[0], a.xcd_start[s] = a.xcd_start[0];
        a.n[s] = 0, a.base[s] = INT32_MAX;
    }
    if (total >= INT32_MAX) return MAGIC_ERR_UNSUPPORTED;
    a.total_n = (int32_t)total;
    WalkWs w;
    if (!workspace || !carve_walk(workspace, workspace_bytes, a.total_n, &w)) return MAGIC_ERR_WORKSPACE;
    const int grid = magic::stride_grid(a.total_n, 256, 2048);
    MAGIC_LAUNCH("walk_cost", k_walk_cost, grid, 256, 0, stream, a, w.cost);
    {
        magic::ProfScope ps("walk_scan", stream);
        size_t cb = w.cub_bytes;
        if (hipcub::DeviceScan::InclusiveSum(w.cub, cb, w.cost, w.incl, (int)a.total_n, stream) != hipSuccess)
            return MAGIC_ERR_RUNTIME;
    }
    MAGIC_LAUNCH("walk_keys", k_walk_keys, grid, 256, 0, stream, a, w.incl, w.cost, w.key_in, w.val_in);
    {
        magic::ProfScope ps("walk_sort", stream);
        size_t cb = w.cub_bytes;
        if (hipcub::DeviceRadixSort::SortPairs(w.cub, cb, w.key_in, w.key_out, w.val_in, w.val_out, (int)a.total_n, 0,
                                               KEY_BITS, stream) != hipSuccess)
            return MAGIC_ERR_RUNTIME;
    }
    MAGIC_LAUNCH("walk_scatter", k_walk_scatter, grid, 256, 0, stream, a, w.val_out);
    return magic::launch_status();
}

}  // extern "C"
