// Augmented projection weight of one GAT layer and its adjoint (model/gat.py:284-299, 323 of
// the reference collapsed into GEMM operands):
//   w_aug = [ W_fc (head-padded) ; fold(W_fc, attn_h) ; fold(W_fc, attn_t) ; W_res (head-padded) ]
//   table[t, h] = sum_d attn_e[h, d] * W_efc[h*D + d, t]
//   fold(W, a)[h, k] = sum_d a[h, d] * W[h*D + d, k]        (sums run over d in ascending order)
// The operands are tiny (<= a few hundred rows); one launch each way replaces a few dozen
// element-wise launches per layer.
#include "common.h"

namespace {

struct FoldArgs {
    int H, D, Dp, F, e_dim;
    bool has_res, has_bias;
    const float *w_fc, *attn_h, *attn_t, *w_res, *w_efc, *attn_e, *bias;
    float *w_aug, *table, *bias_p;
    // adjoint
    const float *d_w_aug, *d_table, *d_bias_p;
    float *d_w_fc, *d_attn_h, *d_attn_t, *d_w_res, *d_w_efc, *d_attn_e, *d_bias;
};

__device__ __forceinline__ void fold_fwd_body(const FoldArgs &a) {
    const int H = a.H, D = a.D, Dp = a.Dp, F = a.F;
    const int Cp = H * Dp;
    const int rows = Cp + 2 * H + (a.has_res ? Cp : 0);
    const int n_aug = rows * F, n_tab = a.e_dim * H, n_bias = a.has_bias ? Cp : 0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n_aug + n_tab + n_bias; i += gridDim.x * 256) {
        if (i < n_aug) {
            const int r = i / F, k = i % F;
            float v = 0.f;
            if (r < Cp) {
                const int h = r / Dp, d = r % Dp;
                if (d < D) v = a.w_fc[(h * D + d) * F + k];
            } else if (r < Cp + 2 * H) {
                const int h = (r - Cp) % H;
                const float *at = (r - Cp) < H ? a.attn_h : a.attn_t;
#pragma unroll 8
                for (int d = 0; d < D; ++d) v = fmaf(at[h * D + d], a.w_fc[(h * D + d) * F + k], v);
            } else {
                const int q = r - Cp - 2 * H, h = q / Dp, d = q % Dp;
                if (d < D) v = a.w_res[(h * D + d) * F + k];
            }
            a.w_aug[i] = v;
        } else if (i < n_aug + n_tab) {
            const int j = i - n_aug, t = j / H, h = j % H;
            float v = 0.f;
#pragma unroll 8
            for (int d = 0; d < D; ++d) v = fmaf(a.attn_e[h * D + d], a.w_efc[(h * D + d) * a.e_dim + t], v);
            a.table[j] = v;
        } else {
            const int j = i - n_aug - n_tab, h = j / Dp, d = j % Dp;
            a.bias_p[j] = d < D ? a.bias[h * D + d] : 0.f;
        }
    }
}

__device__ __forceinline__ void fold_bwd_body(const FoldArgs &a) {
    const int H = a.H, D = a.D, Dp = a.Dp, F = a.F, C = H * D, Cp = H * Dp;
    const float *dh = a.d_w_aug + (int64_t)Cp * F;        // [H,F] gradient of the attn_h fold rows
    const float *dt = a.d_w_aug + (int64_t)(Cp + H) * F;  // [H,F]
    const int n_fc = C * F, n_res = a.has_res ? C * F : 0, n_efc = C * a.e_dim;
    const int n_vec = C;  // attn_h, attn_t, attn_e, bias: C entries each
    const int total = n_fc + n_res + n_efc + (a.has_bias ? n_vec : 0);
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        int j = i;
        if (j < n_fc) {
            const int c = j / F, k = j % F, h = c / D, d = c % D;
            float v = a.d_w_aug ? a.d_w_aug[(h * Dp + d) * F + k] : 0.f;
            if (a.d_w_aug) v = fmaf(a.attn_h[c], dh[h * F + k], fmaf(a.attn_t[c], dt[h * F + k], v));
            a.d_w_fc[j] = v;
            continue;
        }
        j -= n_fc;
        if (j < n_res) {
            const int c = j / F, k = j % F, h = c / D, d = c % D;
            a.d_w_res[j] = a.d_w_aug ? a.d_w_aug[(Cp + 2 * H + h * Dp + d) * F + k] : 0.f;
            continue;
        }
        j -= n_res;
        if (j < n_efc) {
            const int c = j / a.e_dim, t = j % a.e_dim, h = c / D;
            a.d_w_efc[j] = a.d_table ? a.attn_e[c] * a.d_table[t * H + h] : 0.f;
            continue;
        }
        j -= n_efc;
        const int h = j / D, d = j % D;
        a.d_bias[j] = a.d_bias_p ? a.d_bias_p[h * Dp + d] : 0.f;
    }
    // the attention-vector gradients are dot products over F (or e_dim): one wave each, lanes
    // stride over the reduction index, fixed butterfly at the end
    const int lane = threadIdx.x & 63;
    for (int q = blockIdx.x * 4 + (threadIdx.x >> 6); q < 3 * n_vec; q += gridDim.x * 4) {
        const int which = q / n_vec, c = q % n_vec, h = c / D;
        float v = 0.f;
        if (which < 2) {
            const float *g = which == 0 ? dh : dt;
            if (a.d_w_aug)
                for (int k = lane; k < F; k += 64) v = fmaf(a.w_fc[c * F + k], g[h * F + k], v);
        } else if (a.d_table) {
            for (int t = lane; t < a.e_dim; t += 64) v = fmaf(a.w_efc[c * a.e_dim + t], a.d_table[t * H + h], v);
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
        if (lane == 0) (which == 0 ? a.d_attn_h : which == 1 ? a.d_attn_t : a.d_attn_e)[c] = v;
    }
}

__global__ __launch_bounds__(256) void k_fold_fwd(FoldArgs a) { fold_fwd_body(a); }
__global__ __launch_bounds__(256) void k_fold_bwd(FoldArgs a) { fold_bwd_body(a); }

// every layer of a model in one launch: blockIdx.y = layer (the arguments of up to 8 layers travel by value)
constexpr int FOLD_BATCH = 8;
struct FoldBatch {
    FoldArgs layer[FOLD_BATCH];
};
static_assert(sizeof(FoldBatch) <= 3584, "kernel arguments are limited to 4 KB");
__global__ __launch_bounds__(256) void k_fold_fwd_multi(FoldBatch b) { fold_fwd_body(b.layer[blockIdx.y]); }
__global__ __launch_bounds__(256) void k_fold_bwd_multi(FoldBatch b) { fold_bwd_body(b.layer[blockIdx.y]); }

bool dims_ok(int H, int D, int Dp, int F, int e_dim) { return H > 0 && D > 0 && Dp >= D && F > 0 && e_dim > 0; }

int64_t fwd_items(const FoldArgs &a) {
    const int Cp = a.H * a.Dp;
    return (int64_t)(Cp + 2 * a.H + (a.has_res ? Cp : 0)) * a.F + (int64_t)a.e_dim * a.H + (a.has_bias ? Cp : 0);
}

int bwd_grid(const FoldArgs &a) {
    const int C = a.H * a.D;
    const int64_t n = (int64_t)C * a.F * (a.has_res ? 2 : 1) + (int64_t)C * a.e_dim + (int64_t)C * (a.has_bias ? 4 : 3);
    int grid = magic::stride_grid(n, 256, 1024);
    const int dot_blocks = (3 * C + 3) / 4;  // one wave per dot product
    if (grid < dot_blocks) grid = dot_blocks < 1024 ? dot_blocks : 1024;
    return grid;
}

bool from_layer(const magic_fold_layer &l, bool backward, FoldArgs *a) {
    if (!dims_ok(l.H, l.D, l.Dp, l.F, l.e_dim)) return false;
    if (!l.w_fc || !l.attn_h || !l.attn_t || !l.w_efc || !l.attn_e) return false;
    *a = FoldArgs{};
    a->H = l.H, a->D = l.D, a->Dp = l.Dp, a->F = l.F, a->e_dim = l.e_dim;
    a->w_fc = l.w_fc, a->attn_h = l.attn_h, a->attn_t = l.attn_t, a->w_efc = l.w_efc, a->attn_e = l.attn_e;
    if (!backward) {
        if (!l.w_aug || !l.table || (l.bias && !l.bias_p)) return false;
        a->has_res = l.w_res != nullptr, a->has_bias = l.bias != nullptr;
        a->w_res = l.w_res, a->bias = l.bias, a->w_aug = l.w_aug, a->table = l.table, a->bias_p = l.bias_p;
    } else {
        if (!l.d_w_fc || !l.d_attn_h || !l.d_attn_t || !l.d_w_efc || !l.d_attn_e) return false;
        a->has_res = l.d_w_res != nullptr, a->has_bias = l.d_bias != nullptr;
        a->d_w_aug = l.d_w_aug, a->d_table = l.d_table, a->d_bias_p = l.d_bias_p;
        a->d_w_fc = l.d_w_fc, a->d_attn_h = l.d_attn_h, a->d_attn_t = l.d_attn_t, a->d_w_res = l.d_w_res;
        a->d_w_efc = l.d_w_efc, a->d_attn_e = l.d_attn_e, a->d_bias = l.d_bias;
    }
    return true;
}

int fold_multi(int32_t n_layers, const magic_fold_layer *layers, bool backward, hipStream_t stream) {
    if (n_layers < 0 || (n_layers > 0 && !layers)) return MAGIC_ERR_INVALID_ARG;
    for (int32_t l0 = 0; l0 < n_layers; l0 += FOLD_BATCH) {
        FoldBatch b;
        const int nb = n_layers - l0 < FOLD_BATCH ? n_layers - l0 : FOLD_BATCH;
        int grid = 1;
        for (int i = 0; i < nb; ++i) {
            if (!from_layer(layers[l0 + i], backward, &b.layer[i])) return MAGIC_ERR_INVALID_ARG;
            const int g = backward ? bwd_grid(b.layer[i]) : magic::stride_grid(fwd_items(b.layer[i]), 256, 1024);
            if (g > grid) grid = g;
        }
        for (int i = nb; i < FOLD_BATCH; ++i) b.layer[i] = b.layer[0];  // (never indexed: gridDim.y = nb)
        if (backward)
            MAGIC_LAUNCH("fold_bwd", k_fold_bwd_multi, dim3(grid, nb), 256, 0, stream, b);
        else
            MAGIC_LAUNCH("fold_fwd", k_fold_fwd_multi, dim3(grid, nb), 256, 0, stream, b);
    }
    return magic::launch_status();
}

}  // namespace

extern "C" {

int magic_fold_fwd(int32_t H, int32_t D, int32_t Dp, int32_t F, int32_t e_dim, const float *w_fc, const float *attn_h,
                   const float *attn_t, const float *w_res, const float *w_efc, const float *attn_e,
                   const float *bias, float *w_aug, float *table, float *bias_p, void *stream_) {
    if (!dims_ok(H, D, Dp, F, e_dim)) return MAGIC_ERR_INVALID_ARG;
    if (!w_fc || !attn_h || !attn_t || !w_efc || !attn_e || !w_aug || !table || (bias && !bias_p))
        return MAGIC_ERR_INVALID_ARG;
    FoldArgs a = {};
    a.H = H, a.D = D, a.Dp = Dp, a.F = F, a.e_dim = e_dim;
    a.has_res = w_res != nullptr, a.has_bias = bias != nullptr;
    a.w_fc = w_fc, a.attn_h = attn_h, a.attn_t = attn_t, a.w_res = w_res, a.w_efc = w_efc, a.attn_e = attn_e;
    a.bias = bias, a.w_aug = w_aug, a.table = table, a.bias_p = bias_p;
    const int Cp = H * Dp;
    const int64_t n = (int64_t)(Cp + 2 * H + (a.has_res ? Cp : 0)) * F + (int64_t)e_dim * H + (a.has_bias ? Cp : 0);
    MAGIC_LAUNCH("fold_fwd", k_fold_fwd, magic::stride_grid(n, 256, 1024), 256, 0, (hipStream_t)stream_, a);
    return magic::launch_status();
}

int magic_fold_bwd(int32_t H, int32_t D, int32_t Dp, int32_t F, int32_t e_dim, const float *w_fc, const float *attn_h,
                   const float *attn_t, const float *w_efc, const float *attn_e, const float *d_w_aug,
                   const float *d_table, const float *d_bias_p, float *d_w_fc, float *d_attn_h, float *d_attn_t,
                   float *d_w_res, float *d_w_efc, float *d_attn_e, float *d_bias, void *stream_) {
    if (!dims_ok(H, D, Dp, F, e_dim)) return MAGIC_ERR_INVALID_ARG;
    if (!w_fc || !attn_h || !attn_t || !w_efc || !attn_e || !d_w_fc || !d_attn_h || !d_attn_t || !d_w_efc || !d_attn_e)
        return MAGIC_ERR_INVALID_ARG;
    FoldArgs a = {};
    a.H = H, a.D = D, a.Dp = Dp, a.F = F, a.e_dim = e_dim;
    a.has_res = d_w_res != nullptr, a.has_bias = d_bias != nullptr;
    a.w_fc = w_fc, a.attn_h = attn_h, a.attn_t = attn_t, a.w_efc = w_efc, a.attn_e = attn_e;
    a.d_w_aug = d_w_aug, a.d_table = d_table, a.d_bias_p = d_bias_p;
    a.d_w_fc = d_w_fc, a.d_attn_h = d_attn_h, a.d_attn_t = d_attn_t, a.d_w_res = d_w_res, a.d_w_efc = d_w_efc;
    a.d_attn_e = d_attn_e, a.d_bias = d_bias;
    const int C = H * D;
    const int64_t n = (int64_t)C * F * (a.has_res ? 2 : 1) + (int64_t)C * e_dim + (int64_t)C * (a.has_bias ? 4 : 3);
    int grid = magic::stride_grid(n, 256, 1024);
    const int dot_blocks = (3 * C + 3) / 4;  // one wave per dot product
    if (grid < dot_blocks) grid = dot_blocks < 1024 ? dot_blocks : 1024;
    MAGIC_LAUNCH("fold_bwd", k_fold_bwd, grid, 256, 0, (hipStream_t)stream_, a);
    return magic::launch_status();
}

int magic_fold_fwd_multi(int32_t n_layers, const magic_fold_layer *layers, void *stream) {
    return fold_multi(n_layers, layers, false, (hipStream_t)stream);
}

int magic_fold_bwd_multi(int32_t n_layers, const magic_fold_layer *layers, void *stream) {
    return fold_multi(n_layers, layers, true, (hipStream_t)stream);
}

}  // extern "C"
