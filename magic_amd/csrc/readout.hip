// Graph read-out (utils/poolers.py:49-73 of the reference): mean or sum of the node embeddings over
// all nodes, or per node type with the per-type rows concatenated ([1, n_types * width], zeros for a
// type without nodes).  One pass over [N, width]: every block owns a contiguous range of rows and
// adds them, in row order, into a private LDS table [row sub-stream][type][column]; a second launch
// adds the block tables in block order (float64) and divides by the per-type row count for the mean.
// No atomics, no host synchronisation: bitwise reproducible.
#include "common.h"

namespace {

constexpr int kMaxTypes = 32;
constexpr int kRowsPerBlock = 128;

struct ReadoutArgs {
    int64_t n;
    int width, n_types, cw, sub;  // cw: columns per pass (power of two <= 256), sub = 256 / cw row sub-streams
    const float *feat;
    int64_t ld;
    const int64_t *types;  // or null: one type
    int64_t type0;         // first type of this launch's window (wider type sets take several launches)
    int grid;
    int64_t rows_per_block;
    float *ws_sum;    // [grid, sub, n_types, width]
    int32_t *ws_cnt;  // [grid, n_types]
    int mean;
    float *out;  // [n_types * width]
};

__global__ __launch_bounds__(256) void k_readout_partial(ReadoutArgs a) {
    extern __shared__ float tbl[];  // [sub][n_types][cw]
    __shared__ int cnt[kMaxTypes];
    const int c = threadIdx.x % a.cw, ss = threadIdx.x / a.cw;
    const int64_t r0 = (int64_t)blockIdx.x * a.rows_per_block;
    const int64_t r1 = r0 + a.rows_per_block < a.n ? r0 + a.rows_per_block : a.n;
    if (threadIdx.x < kMaxTypes) cnt[threadIdx.x] = 0;
    __syncthreads();
    for (int64_t r = r0 + threadIdx.x; r < r1; r += 256) {  // per-type row counts (integer adds: any order)
        const int64_t t = a.types ? a.types[r] - a.type0 : 0;
        if (t >= 0 && t < a.n_types) atomicAdd(&cnt[t], 1);
    }
    for (int c0 = 0; c0 < a.width; c0 += a.cw) {
        for (int t = 0; t < a.n_types; ++t) tbl[(ss * a.n_types + t) * a.cw + c] = 0.f;
        const int col = c0 + c;
        if (col < a.width) {
            for (int64_t r = r0 + ss; r < r1; r += a.sub) {  // this sub-stream's rows, ascending
                const int64_t t = a.types ? a.types[r] - a.type0 : 0;
                if (t >= 0 && t < a.n_types) tbl[(ss * a.n_types + (int)t) * a.cw + c] += a.feat[r * a.ld + col];
            }
            for (int t = 0; t < a.n_types; ++t)
                a.ws_sum[(((int64_t)blockIdx.x * a.sub + ss) * a.n_types + t) * a.width + col] =
                    tbl[(ss * a.n_types + t) * a.cw + c];
        }
    }
    __syncthreads();
    if (threadIdx.x < a.n_types) a.ws_cnt[(int64_t)blockIdx.x * a.n_types + threadIdx.x] = cnt[threadIdx.x];
}

// one thread per output element (type, column): block tables added in (block, sub-stream) order
__global__ __launch_bounds__(256) void k_readout_final(ReadoutArgs a) {
    const int o = blockIdx.x * 256 + threadIdx.x;
    if (o >= a.n_types * a.width) return;
    const int t = o / a.width, col = o % a.width;
    double s = 0.0;
    int64_t k = 0;
    for (int b = 0; b < a.grid; ++b) {
        for (int ss = 0; ss < a.sub; ++ss) s += (double)a.ws_sum[(((int64_t)b * a.sub + ss) * a.n_types + t) * a.width + col];
        k += a.ws_cnt[(int64_t)b * a.n_types + t];
    }
    a.out[o] = k == 0 ? 0.f : (float)(a.mean ? s / (double)k : s);
}

int readout_grid(int64_t n) { return (int)((n + kRowsPerBlock - 1) / kRowsPerBlock); }

int readout_cw(int width) {
    int cw = 16;
    while (cw < width && cw < 256) cw <<= 1;
    return cw;
}

}  // namespace

extern "C" {

size_t magic_readout_workspace_bytes(int64_t n_rows, int32_t width, int32_t n_types) {
    if (n_rows <= 0 || width <= 0) return 256;
    int T = n_types > 0 ? n_types : 1;
    if (T > kMaxTypes) T = kMaxTypes;  // wider type sets run in windows of 32 types
    const size_t g = readout_grid(n_rows), sub = 256 / readout_cw(width);
    return magic::align_up(g * sub * T * (size_t)width * sizeof(float)) + magic::align_up(g * T * sizeof(int32_t)) + 256;
}

int magic_readout(int64_t n_rows, int32_t width, const float *feat, int64_t ld_feat, const int64_t *node_type,
                  int32_t n_types, int32_t mean, float *out, void *workspace, size_t workspace_bytes, void *stream_) {
    if (n_rows < 0 || width <= 0 || n_types < 0 || !out || (n_rows > 0 && (!feat || ld_feat < width)))
        return MAGIC_ERR_INVALID_ARG;
    if (n_types > 0 && !node_type && n_rows > 0) return MAGIC_ERR_INVALID_ARG;
    hipStream_t stream = (hipStream_t)stream_;
    const int T = n_types > 0 ? n_types : 1;
    if (n_rows == 0) {
        // no rows: the mean over all nodes is NaN as torch's mean of an empty tensor is (utils/poolers.py:58),
        // the sum is 0; per-type read-outs start from zeros (utils/poolers.py:62) and stay zeros
        const bool nan = n_types == 0 && mean;
        if (hipMemsetD32Async((hipDeviceptr_t)out, nan ? 0x7fc00000 : 0, (size_t)T * width, stream) != hipSuccess)
            return MAGIC_ERR_RUNTIME;
        return MAGIC_OK;
    }
    ReadoutArgs a;
    a.n = n_rows;
    a.width = width;
    a.cw = readout_cw(width);
    a.sub = 256 / a.cw;
    a.feat = feat;
    a.ld = ld_feat;
    a.types = n_types > 0 ? node_type : nullptr;
    a.grid = readout_grid(n_rows);
    a.rows_per_block = kRowsPerBlock;
    a.mean = mean;
    const int Tw = T < kMaxTypes ? T : kMaxTypes;  // types per launch: the block's LDS table holds 32 of them
    magic::Arena ar(workspace, workspace_bytes);
    a.ws_sum = ar.take<float>((size_t)a.grid * a.sub * Tw * width);
    a.ws_cnt = ar.take<int32_t>((size_t)a.grid * Tw);
    if (!ar.ok() || !a.ws_sum || !a.ws_cnt) return MAGIC_ERR_WORKSPACE;
    for (int t0 = 0; t0 < T; t0 += kMaxTypes) {  // (stream order: a window's partial tables are consumed before the next)
        a.type0 = t0;
        a.n_types = T - t0 < kMaxTypes ? T - t0 : kMaxTypes;
        a.out = out + (size_t)t0 * width;
        const size_t sh = (size_t)256 * a.n_types * sizeof(float);  // sub * n_types * cw floats
        MAGIC_LAUNCH("readout_partial", k_readout_partial, a.grid, 256, sh, stream, a);
        MAGIC_LAUNCH("readout_final", k_readout_final, (a.n_types * width + 255) / 256, 256, 0, stream, a);
    }
    return magic::launch_status();
}

}  // extern "C"
