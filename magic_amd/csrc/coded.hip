// Coded input of the first encoder layer (gat_common.h: CodedIn): the table of the values an input row can
// take, the 4-byte code word per node, and the weight / mask-token gradients that belong to them.
//
// The reference feeds the first GAT layer one-hot node types (utils/loaddata.py:258) in which the masked rows
// have been overwritten with the mask token (model/autoencoder.py:201-210), through feature dropout
// (model/gat.py:283):  x[u, j] = keep_j(u) * scale * (masked(u) ? token[j] : [j == type(u)]).
// Code word of a node = the row of the table its input row maps to:
//   r in [0, n_dim)          unmasked node of type r whose one kept:      x = scale * e_r
//   r = n_dim                unmasked node whose one was dropped:         x = 0
//   r = n_dim + 1 + b        masked node with keep bits b (bit j = j-th column kept):  x = scale * (token . bits(b))
// Table row r = x(r) . W^T for the layer's augmented weight W [M, n_dim].
// Backward: dW[m, j] = sum_u d_y[u, m] x[u, j] is a sum of d_y rows grouped by code word,
//   A_j[m] = scale * sum over u with code j of d_y[u, m]            (unmasked, kept, type j)
//   S_j[m] = scale * sum over masked u with keep_j(u) of d_y[u, m]
//   dW[m, j] = A_j[m] + token[j] * S_j[m],     d_token[j] = sum_m W[m, j] * S_j[m]
// -- one pass over d_y with the code words, no [N, n_dim] input matrix and no input gradient.
#include <cstdlib>
#include "common.h"

namespace {

constexpr int ND_MAX = MAGIC_CODED_MAX_DIM;  // 13: the table has n_dim + 1 + 2^n_dim rows
constexpr int DW_ROWS = 16;                  // rows a block takes per trip (their loads are in flight together)
constexpr int DW_GRID = 2048;                // partial tables at most (a multiple of DW_FOLD)
constexpr int DW_FOLD = 32;                  // second-stage partial tables

// thread = (table row, 4 columns)
__global__ __launch_bounds__(256) void k_coded_table(int n_dim, int M, const float *__restrict__ w, int64_t ld_w,
                                                     const float *__restrict__ token, float scale, float *__restrict__ tab,
                                                     int64_t ld_tab) {
    const int rows = n_dim + 1 + (1 << n_dim);
    const int m4 = (M + 3) / 4;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)rows * m4) return;
    const int r = (int)(i / m4), c0 = (int)(i % m4) * 4;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (r < n_dim) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (c0 + k < M) v[k] = scale * w[(int64_t)(c0 + k) * ld_w + r];
    } else if (r > n_dim) {
        const uint32_t b = (uint32_t)(r - n_dim - 1);
        for (int j = 0; j < n_dim; ++j) {  // ascending j: the summation order of every row
            if (!((b >> j) & 1u)) continue;
            const float c = scale * token[j];
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (c0 + k < M) v[k] = fmaf(c, w[(int64_t)(c0 + k) * ld_w + j], v[k]);
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (c0 + k < ld_tab) tab[(int64_t)r * ld_tab + c0 + k] = v[k];
}

__device__ __forceinline__ uint32_t keep_bits(const uint8_t *keep, int64_t u, int n_dim) {
    if (!keep) return (1u << n_dim) - 1u;
    uint32_t bits = 0u;
    for (int j = 0; j < n_dim; ++j) bits |= (keep[u * n_dim + j] ? 1u : 0u) << j;
    return bits;
}

__global__ __launch_bounds__(256) void k_node_code(int64_t n, int n_dim, const int64_t *__restrict__ type,
                                                   const uint8_t *__restrict__ keep, uint32_t *__restrict__ code) {
    const int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (u >= n) return;
    int64_t t = type[u];
    t = t < 0 ? 0 : (t >= n_dim ? n_dim - 1 : t);
    code[u] = ((keep_bits(keep, u, n_dim) >> t) & 1u) ? (uint32_t)t : (uint32_t)n_dim;
}

__global__ __launch_bounds__(256) void k_node_code_mark(int64_t n_mask, int n_dim, const int64_t *__restrict__ rows,
                                                        const uint8_t *__restrict__ keep, uint32_t *__restrict__ code) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_mask) return;
    const int64_t u = rows[i];  // (row ids are distinct: a permutation prefix)
    code[u] = (uint32_t)n_dim + 1u + keep_bits(keep, u, n_dim);
}

struct DwArgs {
    int64_t n;
    int M, n_dim;
    const uint32_t *code;
    const float *dy;
    int64_t ld_dy;
    float scale;
    int grid;      // partial tables of this launch (a multiple of DW_FOLD)
    float *part;   // [grid][2][n_dim][M]
    float *part2;  // [DW_FOLD][2][n_dim][M]
    const float *w;  // [M, ld_w] the layer's augmented weight
    int64_t ld_w;
    const float *token;
    float *dw;  // [M, ld_dw]
    int64_t ld_dw;
    float *d_token;  // [n_dim]
};

// Rows of float4 (M and the row pitch multiples of 4, 16-byte aligned base): thread = (row slot, column quad) of the
// block's column window of `wq` quads, R = 256 / wq row slots.  A trip takes QK * R consecutive rows, QK float4 loads
// per thread in flight; a row's code word picks the accumulators through 0 / 1 factors (rows differ between the lanes
// of a wave: no branches).  Each slot adds its rows in ascending order, the slots are added in slot order.
constexpr int QK = 8;
template <int ND>
__global__ __launch_bounds__(256) void k_coded_dw_quads(DwArgs a, int wq, int R) {
    __shared__ float4 red[256];
    const int q = threadIdx.x % wq, rs = threadIdx.x / wq;
    const int c4 = (blockIdx.y * wq + q) * 4;
    const bool live = rs < R && c4 < a.M;
    const int cc = c4 < a.M ? c4 : 0;
    float4 A[ND], S[ND];
#pragma unroll
    for (int j = 0; j < ND; ++j) A[j] = S[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int64_t per_trip = (int64_t)QK * R;
    const int64_t n_trips = (a.n + per_trip - 1) / per_trip;
    const uint32_t nd = (uint32_t)a.n_dim;
    const int rsc = rs < R ? rs : 0;
    for (int64_t trip = blockIdx.x; trip < n_trips; trip += gridDim.x) {
        float4 d[QK];
        uint32_t c[QK];
#pragma unroll
        for (int k = 0; k < QK; ++k) {
            const int64_t u = trip * per_trip + (int64_t)k * R + rsc;
            const int64_t uc = u < a.n ? u : a.n - 1;
            d[k] = *reinterpret_cast<const float4 *>(a.dy + uc * a.ld_dy + cc);
            c[k] = u < a.n ? a.code[uc] : nd;  // (past the end: the dropped row, which contributes nothing)
        }
#pragma unroll
        for (int k = 0; k < QK; ++k) {
            const bool masked = c[k] > nd;
            const uint32_t b = masked ? c[k] - nd - 1u : 0u;
#pragma unroll
            for (int j = 0; j < ND; ++j) {
                const float fa = c[k] == (uint32_t)j ? 1.f : 0.f;
                const float fs = ((b >> j) & 1u) ? 1.f : 0.f;
                A[j].x = fmaf(d[k].x, fa, A[j].x), A[j].y = fmaf(d[k].y, fa, A[j].y);
                A[j].z = fmaf(d[k].z, fa, A[j].z), A[j].w = fmaf(d[k].w, fa, A[j].w);
                S[j].x = fmaf(d[k].x, fs, S[j].x), S[j].y = fmaf(d[k].y, fs, S[j].y);
                S[j].z = fmaf(d[k].z, fs, S[j].z), S[j].w = fmaf(d[k].w, fs, S[j].w);
            }
        }
    }
    float *p = a.part + (size_t)blockIdx.x * 2 * a.n_dim * a.M;
#pragma unroll
    for (int j = 0; j < 2 * ND; ++j) {
        if (j % ND >= a.n_dim) continue;
        float4 t = j < ND ? A[j % ND] : S[j % ND];
        __syncthreads();
        red[threadIdx.x] = t;
        __syncthreads();
        if (rs == 0 && live) {
            for (int k = 1; k < R; ++k) {
                const float4 o = red[k * wq + q];
                t.x += o.x, t.y += o.y, t.z += o.z, t.w += o.w;
            }
            t.x *= a.scale, t.y *= a.scale, t.z *= a.scale, t.w *= a.scale;
            *reinterpret_cast<float4 *>(p + ((size_t)(j < ND ? 0 : a.n_dim) + j % ND) * a.M + c4) = t;
        }
    }
}

// Any M / pitch: thread = column m (blockIdx.y selects the window of blockDim.x columns); the block walks its rows in a
// fixed order.  The code word is the same for every thread of the block: one scalar branch per row picks the
// accumulator set, the adds themselves are multiplications by 0 / 1 held in scalar registers (no per-column branches).
template <int ND>
__global__ __launch_bounds__(256) void k_coded_dw_partial(DwArgs a) {
    const int m = blockIdx.y * blockDim.x + threadIdx.x;
    const bool mok = m < a.M;
    const int mc = mok ? m : a.M - 1;
    float A[ND], S[ND];
#pragma unroll
    for (int j = 0; j < ND; ++j) A[j] = S[j] = 0.f;
    const int64_t n_trips = (a.n + DW_ROWS - 1) / DW_ROWS;
    const uint32_t nd = (uint32_t)a.n_dim;
    for (int64_t trip = blockIdx.x; trip < n_trips; trip += gridDim.x) {
        float d[DW_ROWS];
        uint32_t c[DW_ROWS];
#pragma unroll
        for (int r = 0; r < DW_ROWS; ++r) {
            const int64_t u = trip * DW_ROWS + r;
            const int64_t uc = u < a.n ? u : a.n - 1;
            d[r] = a.dy[uc * a.ld_dy + mc];
            c[r] = u < a.n ? a.code[uc] : nd;  // (past the end: the dropped row, which contributes nothing)
        }
#pragma unroll
        for (int r = 0; r < DW_ROWS; ++r) {
            const uint32_t cw = __builtin_amdgcn_readfirstlane(c[r]);
            if (cw > nd) {
                const uint32_t b = cw - nd - 1u;
#pragma unroll
                for (int j = 0; j < ND; ++j) S[j] = fmaf(d[r], ((b >> j) & 1u) ? 1.f : 0.f, S[j]);
            } else {
#pragma unroll
                for (int j = 0; j < ND; ++j) A[j] = fmaf(d[r], cw == (uint32_t)j ? 1.f : 0.f, A[j]);
            }
        }
    }
    if (mok) {
        float *p = a.part + (size_t)blockIdx.x * 2 * a.n_dim * a.M;
#pragma unroll
        for (int j = 0; j < ND; ++j)
            if (j < a.n_dim) {
                p[(size_t)j * a.M + m] = A[j] * a.scale;
                p[((size_t)a.n_dim + j) * a.M + m] = S[j] * a.scale;
            }
    }
}

// partial tables -> DW_FOLD (block y adds its run of tables in order)
__global__ __launch_bounds__(256) void k_coded_dw_fold(DwArgs a) {
    const int o = blockIdx.x * 256 + threadIdx.x;
    const int n_out = 2 * a.n_dim * a.M;
    if (o >= n_out) return;
    const int per = a.grid / DW_FOLD;
    float t = 0.f;
    for (int k = 0; k < per; ++k) t += a.part[(size_t)(blockIdx.y * per + k) * n_out + o];
    a.part2[(size_t)blockIdx.y * n_out + o] = t;
}

// block j: dW[:, j] and d_token[j]
__global__ __launch_bounds__(256) void k_coded_dw_final(DwArgs a) {
    __shared__ float red[256];
    const int j = blockIdx.x;
    const int n_out = 2 * a.n_dim * a.M;
    const float tok = a.token[j];
    float dt = 0.f;
    for (int m = threadIdx.x; m < a.M; m += 256) {
        float A = 0.f, S = 0.f;
        for (int k = 0; k < DW_FOLD; ++k) {
            A += a.part2[(size_t)k * n_out + (size_t)j * a.M + m];
            S += a.part2[(size_t)k * n_out + ((size_t)a.n_dim + j) * a.M + m];
        }
        a.dw[(size_t)m * a.ld_dw + j] = fmaf(tok, S, A);
        dt = fmaf(a.w[(size_t)m * a.ld_w + j], S, dt);
    }
    red[threadIdx.x] = dt;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0 && a.d_token) a.d_token[j] = red[0];
}

// a block should take several trips: its partial table costs as much traffic as a few rows of d_y
int partial_tables(int64_t n_rows, int64_t rows_per_trip) {
    const int64_t trips = (n_rows + rows_per_trip - 1) / rows_per_trip;
    int64_t g = trips / 4 / DW_FOLD * DW_FOLD;
    return (int)(g < DW_FOLD ? DW_FOLD : (g > DW_GRID ? DW_GRID : g));
}

}  // namespace

extern "C" {

int64_t magic_coded_table_rows(int32_t n_dim) {
    if (n_dim <= 0 || n_dim > ND_MAX) return 0;
    return (int64_t)n_dim + 1 + ((int64_t)1 << n_dim);
}

int magic_coded_table(int32_t n_dim, int32_t m, const float *w, int64_t ld_w, const float *token, float scale,
                      float *table, int64_t ld_table, void *stream_) {
    if (n_dim <= 0 || n_dim > ND_MAX || m <= 0 || !w || !token || !table || ld_w < n_dim || ld_table < m)
        return MAGIC_ERR_INVALID_ARG;
    const int64_t work = magic_coded_table_rows(n_dim) * ((m + 3) / 4);
    MAGIC_LAUNCH("coded_table", k_coded_table, (unsigned)((work + 255) / 256), 256, 0, (hipStream_t)stream_, (int)n_dim,
                 (int)m, w, ld_w, token, scale, table, ld_table);
    return magic::launch_status();
}

int magic_node_code(int64_t n_nodes, int32_t n_dim, const int64_t *node_type, const uint8_t *keep_mask, int64_t n_mask,
                    const int64_t *mask_rows, uint32_t *code, void *stream_) {
    if (n_nodes < 0 || n_dim <= 0 || n_dim > ND_MAX || n_mask < 0) return MAGIC_ERR_INVALID_ARG;
    if (n_nodes == 0) return MAGIC_OK;
    if (!node_type || !code || (n_mask > 0 && !mask_rows)) return MAGIC_ERR_INVALID_ARG;
    hipStream_t stream = (hipStream_t)stream_;
    MAGIC_LAUNCH("node_code", k_node_code, (unsigned)((n_nodes + 255) / 256), 256, 0, stream, n_nodes, (int)n_dim, node_type,
                 keep_mask, code);
    if (n_mask > 0)
        MAGIC_LAUNCH("node_code_mark", k_node_code_mark, (unsigned)((n_mask + 255) / 256), 256, 0, stream, n_mask,
                     (int)n_dim, mask_rows, keep_mask, code);
    return magic::launch_status();
}

size_t magic_coded_weight_grad_workspace_bytes(int32_t m, int32_t n_dim) {
    if (m <= 0 || n_dim <= 0) return 256;
    const size_t tab = (size_t)2 * n_dim * m * sizeof(float);
    return magic::align_up(tab * DW_GRID) + magic::align_up(tab * DW_FOLD) + 256;
}

int magic_coded_weight_grad(int64_t n_nodes, int32_t m, const magic_coded_input *input, const float *dy, int64_t ld_dy,
                            const float *w, int64_t ld_w, float *dw, int64_t ld_dw, float *d_token, void *workspace,
                            size_t workspace_bytes, void *stream_) {
    if (n_nodes < 0 || m <= 0 || !input || input->n_dim <= 0 || input->n_dim > ND_MAX) return MAGIC_ERR_INVALID_ARG;
    if (!input->code || !input->token || !dw || !w || ld_w < input->n_dim || ld_dw < input->n_dim)
        return MAGIC_ERR_INVALID_ARG;
    if (n_nodes > 0 && (!dy || ld_dy < m)) return MAGIC_ERR_INVALID_ARG;
    hipStream_t stream = (hipStream_t)stream_;
    DwArgs a;
    a.n = n_nodes, a.M = m, a.n_dim = input->n_dim;
    a.code = input->code, a.dy = dy, a.ld_dy = ld_dy, a.scale = input->scale;
    a.w = w, a.ld_w = ld_w, a.token = input->token;
    a.dw = dw, a.ld_dw = ld_dw, a.d_token = d_token;
    magic::Arena ar(workspace, workspace_bytes);
    const size_t tab = (size_t)2 * a.n_dim * m;
    a.part = ar.take<float>(tab * DW_GRID);
    a.part2 = ar.take<float>(tab * DW_FOLD);
    if (!ar.ok() || !a.part || !a.part2) return MAGIC_ERR_WORKSPACE;
    // column windows of equal width (a multiple of 64 threads, at most 256): 136 columns -> one block of 192
    // threads, 520 -> three of 192
    const int n_win = (m + 255) / 256;
    const int width = ((m + n_win - 1) / n_win + 63) / 64 * 64;
    static const bool force_cols = getenv("MAGIC_CODED_DW_COLS") != nullptr;  // (comparison runs: tools/ubench/coded_dw.py)
    const bool quads = (m & 3) == 0 && (ld_dy & 3) == 0 && ((uintptr_t)dy & 15) == 0 && !force_cols;
    if (quads) {
        // column windows of at most 64 quads, equal width: 136 columns -> 34 quads x 7 row slots, 520 -> 3 x (44 x 5)
        const int m4 = m / 4, qwin = (m4 + 63) / 64, wq = (m4 + qwin - 1) / qwin, R = 256 / wq;
        a.grid = partial_tables(n_nodes, (int64_t)QK * R);
        dim3 qgrid(a.grid, qwin);
        if (a.n_dim <= 5)
            MAGIC_LAUNCH("coded_dw_partial", k_coded_dw_quads<5>, qgrid, 256, 0, stream, a, wq, R);
        else if (a.n_dim <= 8)
            MAGIC_LAUNCH("coded_dw_partial", k_coded_dw_quads<8>, qgrid, 256, 0, stream, a, wq, R);
        else
            MAGIC_LAUNCH("coded_dw_partial", k_coded_dw_quads<ND_MAX>, qgrid, 256, 0, stream, a, wq, R);
    } else {
        a.grid = partial_tables(n_nodes, DW_ROWS);
        dim3 grid(a.grid, n_win);
        if (a.n_dim <= 5)
            MAGIC_LAUNCH("coded_dw_partial", k_coded_dw_partial<5>, grid, width, 0, stream, a);
        else if (a.n_dim <= 8)
            MAGIC_LAUNCH("coded_dw_partial", k_coded_dw_partial<8>, grid, width, 0, stream, a);
        else
            MAGIC_LAUNCH("coded_dw_partial", k_coded_dw_partial<ND_MAX>, grid, width, 0, stream, a);
    }
    dim3 fgrid((unsigned)((tab + 255) / 256), DW_FOLD);
    MAGIC_LAUNCH("coded_dw_fold", k_coded_dw_fold, fgrid, 256, 0, stream, a);
    MAGIC_LAUNCH("coded_dw_final", k_coded_dw_final, a.n_dim, 256, 0, stream, a);
    return magic::launch_status();
}

}  // extern "C"
