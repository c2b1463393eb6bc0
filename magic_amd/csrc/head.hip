// Tail of the sampled edge-reconstruction loss (model/autoencoder.py:257-266 of the reference through the head
// built at :114-119): LeakyReLU -> Linear(hidden, 1) -> Sigmoid -> binary cross entropy against 1 for the
// positive pairs and 0 for the negative ones, averaged over the pairs that count.
//
// As tensor operations that tail is ~25 launches of a few microseconds each, forward and backward, on
// [pairs, hidden] and [pairs] tensors (pairs ~ 20,000).  Here it is two passes over the hidden rows:
//   forward   wavefront per row: z = <leaky(h), w2> + b2, p = sigmoid(z), the row's loss term and the gradient
//             with respect to z (before the division by the weight sum); block sums of loss and weight in fixed
//             order, a last block adds them;
//   backward  d_h = g / W * d_z * w2 * leaky'(h) written row by row; the weight gradient sum_r d_z[r] leaky(h[r,:])
//             and the column sums of d_h (the first layer's bias gradient) accumulated per lane over a block's 64
//             rows (lane = column), block partials added by one wavefront per output in a fixed order.
// The first layer's bias may be handed in separately (b1): it is then added to h on the fly in both passes and
// its gradient comes out of the backward pass, which saves the caller an element-wise pass and a reduction.
// The binary cross entropy follows torch's BCELoss: logarithms clamped at -100, p (1 - p) at 1e-12 in the
// gradient.
#include "common.h"

namespace {

constexpr int HEAD_MAX_W = 1024;     // hidden width limit (4 float4 per lane)
constexpr int HEAD_BWD_ROWS = 64;    // rows per block in the backward pass

struct HeadArgs {
    int64_t n_rows, n_pos;
    int width;
    const float *h;
    int64_t ld_h;
    const float *b1;  // [width] bias of the first layer, added to h on the fly, or null
    const float *w2, *b2;
    const uint8_t *neg_valid;  // [n_rows - n_pos] or null
    float slope;
    float *loss, *d_logit, *weight_sum;
    float *part;  // [n_blocks, 2] block sums (loss, weight)
    int n_blocks;
    // backward
    const float *grad_out;
    float *d_h;
    int64_t ld_dh;
    float *d_w2, *d_b2, *d_b1;
    float *col_part;  // [n_blocks_bwd, 2 * width + 1]: d_w2 partials, d_b1 partials, d_b2 partial
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

__device__ __forceinline__ float row_weight(const HeadArgs &a, int64_t r) {
    if (r < a.n_pos || !a.neg_valid) return 1.f;
    return a.neg_valid[r - a.n_pos] ? 1.f : 0.f;
}

// forward: 4 rows per block (wavefront per row)
__global__ __launch_bounds__(256) void k_head_fwd(HeadArgs a) {
    __shared__ float red[4][2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t r = (int64_t)blockIdx.x * 4 + wave;
    float l = 0.f, wt = 0.f;
    if (r < a.n_rows) {
        float z = 0.f;
        for (int c = lane * 4; c < a.width; c += 256) {
            float4 h = *reinterpret_cast<const float4 *>(a.h + r * a.ld_h + c);
            if (a.b1) {
                const float4 b = *reinterpret_cast<const float4 *>(a.b1 + c);
                h.x += b.x, h.y += b.y, h.z += b.z, h.w += b.w;
            }
            const float4 w = *reinterpret_cast<const float4 *>(a.w2 + c);
            z = fmaf(h.x > 0.f ? h.x : a.slope * h.x, w.x, z);
            z = fmaf(h.y > 0.f ? h.y : a.slope * h.y, w.y, z);
            z = fmaf(h.z > 0.f ? h.z : a.slope * h.z, w.z, z);
            z = fmaf(h.w > 0.f ? h.w : a.slope * h.w, w.w, z);
        }
        z = wave_sum(z) + a.b2[0];
        const float p = 1.f / (1.f + expf(-z)), q = 1.f - p;
        const bool pos = r < a.n_pos;
        wt = row_weight(a, r);
        l = -wt * fmaxf(logf(pos ? p : q), -100.f);
        const float pq = p * q;
        if (lane == 0) a.d_logit[r] = wt * (p - (pos ? 1.f : 0.f)) * (pq / fmaxf(pq, 1e-12f));
    }
    if (lane == 0) red[wave][0] = l, red[wave][1] = wt;
    __syncthreads();
    if (threadIdx.x == 0) {
        a.part[(int64_t)blockIdx.x * 2] = (red[0][0] + red[1][0]) + (red[2][0] + red[3][0]);
        a.part[(int64_t)blockIdx.x * 2 + 1] = (red[0][1] + red[1][1]) + (red[2][1] + red[3][1]);
    }
}

// one block: the block sums in block order -> loss = sum(l) / sum(w)
__global__ __launch_bounds__(256) void k_head_final(HeadArgs a) {
    __shared__ float red[256][2];
    float l = 0.f, w = 0.f;
    for (int b = threadIdx.x; b < a.n_blocks; b += 256) l += a.part[(int64_t)b * 2], w += a.part[(int64_t)b * 2 + 1];
    red[threadIdx.x][0] = l, red[threadIdx.x][1] = w;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x][0] += red[threadIdx.x + s][0], red[threadIdx.x][1] += red[threadIdx.x + s][1];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        a.loss[0] = red[0][0] / red[0][1];
        a.weight_sum[0] = red[0][1];
    }
}

// backward: 64 rows per block, 16 per wavefront; lane = columns lane*4 + 256*k
__global__ __launch_bounds__(256) void k_head_bwd(HeadArgs a) {
    __shared__ float4 cols[2][4][HEAD_MAX_W / 4];
    __shared__ float dzs[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float scale = a.grad_out[0] / a.weight_sum[0];
    float4 acc[HEAD_MAX_W / 256], accb[HEAD_MAX_W / 256], w[HEAD_MAX_W / 256], b[HEAD_MAX_W / 256];
#pragma unroll
    for (int k = 0; k < HEAD_MAX_W / 256; ++k) {
        const int c = lane * 4 + 256 * k;
        acc[k] = accb[k] = b[k] = w[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c < a.width) {
            w[k] = *reinterpret_cast<const float4 *>(a.w2 + c);
            if (a.b1) b[k] = *reinterpret_cast<const float4 *>(a.b1 + c);
        }
    }
    float dz_sum = 0.f;
    const int64_t r0 = (int64_t)blockIdx.x * HEAD_BWD_ROWS + wave * (HEAD_BWD_ROWS / 4);
    for (int i = 0; i < HEAD_BWD_ROWS / 4; ++i) {
        const int64_t r = r0 + i;
        if (r >= a.n_rows) break;
        const float dz = a.d_logit[r] * scale;
        dz_sum += dz;
#pragma unroll
        for (int k = 0; k < HEAD_MAX_W / 256; ++k) {
            const int c = lane * 4 + 256 * k;
            if (c < a.width) {
                float4 h = *reinterpret_cast<const float4 *>(a.h + r * a.ld_h + c);
                h.x += b[k].x, h.y += b[k].y, h.z += b[k].z, h.w += b[k].w;
                float4 d;
                d.x = dz * w[k].x * (h.x > 0.f ? 1.f : a.slope);
                d.y = dz * w[k].y * (h.y > 0.f ? 1.f : a.slope);
                d.z = dz * w[k].z * (h.z > 0.f ? 1.f : a.slope);
                d.w = dz * w[k].w * (h.w > 0.f ? 1.f : a.slope);
                *reinterpret_cast<float4 *>(a.d_h + r * a.ld_dh + c) = d;
                accb[k].x += d.x, accb[k].y += d.y, accb[k].z += d.z, accb[k].w += d.w;
                acc[k].x = fmaf(dz, h.x > 0.f ? h.x : a.slope * h.x, acc[k].x);
                acc[k].y = fmaf(dz, h.y > 0.f ? h.y : a.slope * h.y, acc[k].y);
                acc[k].z = fmaf(dz, h.z > 0.f ? h.z : a.slope * h.z, acc[k].z);
                acc[k].w = fmaf(dz, h.w > 0.f ? h.w : a.slope * h.w, acc[k].w);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < HEAD_MAX_W / 256; ++k)
        if (lane * 4 + 256 * k < a.width) cols[0][wave][lane + 64 * k] = acc[k], cols[1][wave][lane + 64 * k] = accb[k];
    if (lane == 0) dzs[wave] = dz_sum;
    __syncthreads();
    float *out = a.col_part + (int64_t)blockIdx.x * (2 * a.width + 1);
    for (int q = threadIdx.x; q * 4 < 2 * a.width; q += 256) {
        const int which = q * 4 >= a.width, c4 = q - which * (a.width / 4);
        const float4 s0 = cols[which][0][c4], s1 = cols[which][1][c4], s2 = cols[which][2][c4], s3 = cols[which][3][c4];
        out[q * 4 + 0] = (s0.x + s1.x) + (s2.x + s3.x);
        out[q * 4 + 1] = (s0.y + s1.y) + (s2.y + s3.y);
        out[q * 4 + 2] = (s0.z + s1.z) + (s2.z + s3.z);
        out[q * 4 + 3] = (s0.w + s1.w) + (s2.w + s3.w);
    }
    if (threadIdx.x == 0) out[2 * a.width] = (dzs[0] + dzs[1]) + (dzs[2] + dzs[3]);
}

// d_w2[c], d_b1[c], d_b2: one wavefront per output; lanes take the block partials round robin, fixed butterfly
__global__ __launch_bounds__(64) void k_head_bwd_final(HeadArgs a, int n_blocks) {
    const int c = blockIdx.x;  // 0 .. 2 * width
    float s = 0.f;
    for (int b = threadIdx.x; b < n_blocks; b += 64) s += a.col_part[(int64_t)b * (2 * a.width + 1) + c];
    s = wave_sum(s);
    if (threadIdx.x == 0) {
        if (c < a.width)
            a.d_w2[c] = s;
        else if (c < 2 * a.width) {
            if (a.d_b1) a.d_b1[c - a.width] = s;
        } else
            a.d_b2[0] = s;
    }
}

inline int64_t fwd_blocks(int64_t n_rows) { return (n_rows + 3) / 4; }
inline int64_t bwd_blocks(int64_t n_rows) { return (n_rows + HEAD_BWD_ROWS - 1) / HEAD_BWD_ROWS; }

bool head_shape_ok(int64_t n_rows, int32_t width) {
    return n_rows > 0 && n_rows < (int64_t)1 << 30 && width > 0 && width <= HEAD_MAX_W && width % 4 == 0;
}

}  // namespace

extern "C" {

size_t magic_edge_head_workspace_bytes(int64_t n_rows, int32_t width) {
    if (!head_shape_ok(n_rows, width)) return 0;
    return magic::align_up((size_t)fwd_blocks(n_rows) * 2 * sizeof(float)) +
           magic::align_up((size_t)bwd_blocks(n_rows) * (2 * width + 1) * sizeof(float)) + 512;
}

int magic_edge_head_fwd(int64_t n_rows, int64_t n_pos, int32_t width, const float *h_pre, int64_t ld_h,
                        const float *b1, const float *w2, const float *b2, const uint8_t *neg_valid, float slope, float *loss,
                        float *d_logit, float *weight_sum, void *workspace, size_t workspace_bytes, void *stream_) {
    if (!head_shape_ok(n_rows, width)) return n_rows <= 0 || width <= 0 ? MAGIC_ERR_INVALID_ARG : MAGIC_ERR_UNSUPPORTED;
    if (!h_pre || !w2 || !b2 || !loss || !d_logit || !weight_sum || n_pos < 0 || n_pos > n_rows || ld_h < width)
        return MAGIC_ERR_INVALID_ARG;
    if (((uintptr_t)h_pre | (uintptr_t)w2 | (uintptr_t)b1) % 16 != 0 || ld_h % 4 != 0) return MAGIC_ERR_UNSUPPORTED;
    magic::Arena ar(workspace, workspace_bytes);
    HeadArgs a = {};
    a.b1 = b1;
    a.part = ar.take<float>((size_t)fwd_blocks(n_rows) * 2);
    if (!ar.ok()) return MAGIC_ERR_WORKSPACE;
    hipStream_t stream = (hipStream_t)stream_;
    a.n_rows = n_rows, a.n_pos = n_pos, a.width = width;
    a.h = h_pre, a.ld_h = ld_h, a.w2 = w2, a.b2 = b2, a.neg_valid = neg_valid, a.slope = slope;
    a.loss = loss, a.d_logit = d_logit, a.weight_sum = weight_sum;
    a.n_blocks = (int)fwd_blocks(n_rows);
    MAGIC_LAUNCH("edge_head_fwd", k_head_fwd, (unsigned)a.n_blocks, 256, 0, stream, a);
    MAGIC_LAUNCH("edge_head_final", k_head_final, 1, 256, 0, stream, a);
    return magic::launch_status();
}

int magic_edge_head_bwd(int64_t n_rows, int32_t width, const float *h_pre, int64_t ld_h, const float *b1,
                        const float *w2, const float *d_logit, const float *weight_sum, const float *grad_out,
                        float slope, float *d_h_pre, int64_t ld_dh, float *d_b1, float *d_w2, float *d_b2,
                        void *workspace, size_t workspace_bytes, void *stream_) {
    if (!head_shape_ok(n_rows, width)) return n_rows <= 0 || width <= 0 ? MAGIC_ERR_INVALID_ARG : MAGIC_ERR_UNSUPPORTED;
    if (!h_pre || !w2 || !d_logit || !weight_sum || !grad_out || !d_h_pre || !d_w2 || !d_b2 || ld_h < width ||
        ld_dh < width)
        return MAGIC_ERR_INVALID_ARG;
    if (((uintptr_t)h_pre | (uintptr_t)w2 | (uintptr_t)d_h_pre | (uintptr_t)b1) % 16 != 0 || ld_h % 4 != 0 ||
        ld_dh % 4 != 0)
        return MAGIC_ERR_UNSUPPORTED;
    magic::Arena ar(workspace, workspace_bytes);
    HeadArgs a = {};
    (void)ar.take<float>((size_t)fwd_blocks(n_rows) * 2);
    const int nb = (int)bwd_blocks(n_rows);
    a.col_part = ar.take<float>((size_t)nb * (2 * width + 1));
    if (!ar.ok()) return MAGIC_ERR_WORKSPACE;
    hipStream_t stream = (hipStream_t)stream_;
    a.n_rows = n_rows, a.width = width;
    a.h = h_pre, a.ld_h = ld_h, a.b1 = b1, a.w2 = w2, a.slope = slope;
    a.d_b1 = d_b1;
    a.d_logit = const_cast<float *>(d_logit), a.weight_sum = const_cast<float *>(weight_sum), a.grad_out = grad_out;
    a.d_h = d_h_pre, a.ld_dh = ld_dh, a.d_w2 = d_w2, a.d_b2 = d_b2;
    MAGIC_LAUNCH("edge_head_bwd", k_head_bwd, (unsigned)nb, 256, 0, stream, a);
    MAGIC_LAUNCH("edge_head_bwd_final", k_head_bwd_final, (unsigned)(2 * width + 1), 64, 0, stream, a, nb);
    return magic::launch_status();
}

}  // extern "C"
