// GAT layer, edge phase, forward: attention logits -> leaky ReLU -> per-destination
// softmax -> weighted aggregation of source rows -> bias + residual (+ PReLU).
// HBM-bound: per edge one gathered hs row (4*H*D bytes) dominates; everything else is
// streamed in CSC order.  No atomics; every reduction has a fixed order.
//
// Launch sequence per layer:
//   gat_edge_logits       one thread per edge: x[e,:] = (eh[src e] + ee[e]) + et[dst e]
//                         (the only place that gathers eh; fully edge-parallel)
//   gat_fwd_light         one lane group per destination with <= light_max in-edges:
//                         max / sum-exp over the streamed logits, then the row gather
//   gat_fwd_heavy_*       longer segments in chunks of one wave each (stats, aggregate,
//                         finish), combined in chunk order
#include "gat_common.h"

namespace magic {

struct FwdArgs {
    int N, E;
    PlanDev plan;
    const int32_t *src;  // [E] CSC order
    const int32_t *dst;  // [E] CSC order
    const float *hs;
    int64_t ld_hs;
    const float *eh;  // [N,ld_s] head scores of the node as a source
    const float *et;  // [N,ld_s] head scores of the node as a destination
    int64_t ld_s;
    const float *ee_table;  // [e_dim,H]
    const int32_t *etype;   // [E] CSC order
    const float *ee;        // [E,H] CSC order
    float slope;
    const float *bias;
    const float *res;
    int64_t ld_res;
    const uint8_t *res_mask;  // [N,C] keep mask of the residual input, or null
    float res_scale;
    const float *prelu_w;
    float *out;
    int64_t ld_out;
    float *pre;   // [N,C] or null
    float *attn;  // [E,H] or null
    float *stat;  // [N,H,2] per destination and head: (et, max + log sum exp of its in-edge logits), or null
    float *x;         // [E,H] raw logits (workspace)
    float *ws_stats;  // [n_chunks,H,2]
    float *ws_acc;    // [n_chunks,C]
};

template <int H, bool TABLE>
__global__ __launch_bounds__(256) void k_edge_logits(FwdArgs a) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < a.E; i += gridDim.x * 256) {
        const int u = a.src[i], v = a.dst[i];
        const float *ehp = a.eh + (int64_t)u * a.ld_s;
        const float *etp = a.et + (int64_t)v * a.ld_s;
        const float *eep = TABLE ? a.ee_table + a.etype[i] * H : a.ee + (int64_t)i * H;
        float r[H];
#pragma unroll
        for (int h = 0; h < H; ++h) r[h] = (ehp[h] + eep[h]) + etp[h];
#pragma unroll
        for (int h = 0; h < H; ++h) a.x[(int64_t)i * H + h] = r[h];
    }
}

template <int H, int LPH>
__device__ __forceinline__ void epilogue(const FwdArgs &a, int v, int g, float4 acc) {
    constexpr int C = H * LPH * 4;
    if (a.bias) add4(acc, ld4(a.bias + g * 4));
    if (a.res) {
        float4 r = ld4_nt(a.res + (int64_t)v * a.ld_res + g * 4);
        if (a.res_mask) {  // residual input = dropout(input): keep mask applied on the fly
            const uchar4 m = *reinterpret_cast<const uchar4 *>(a.res_mask + (int64_t)v * C + g * 4);
            r.x = m.x ? r.x * a.res_scale : 0.f;
            r.y = m.y ? r.y * a.res_scale : 0.f;
            r.z = m.z ? r.z * a.res_scale : 0.f;
            r.w = m.w ? r.w * a.res_scale : 0.f;
        }
        add4(acc, r);
    }
    if (a.prelu_w) {
        float w = a.prelu_w[0];
        // with a positive slope the backward pass recovers the pre-activation from the output
        if (a.pre && !(w > 0.f)) st4(a.pre + (int64_t)v * C + g * 4, acc);
        acc.x = acc.x >= 0.f ? acc.x : w * acc.x;
        acc.y = acc.y >= 0.f ? acc.y : w * acc.y;
        acc.z = acc.z >= 0.f ? acc.z : w * acc.z;
        acc.w = acc.w >= 0.f ? acc.w : w * acc.w;
    }
    st4(a.out + (int64_t)v * a.ld_out + g * 4, acc);
}

// One lane group (G lanes) per destination node.  The LPH lanes of a head split the
// segment for the max and the sum of exponentials (combined with DPP), then every lane
// walks all edges for the aggregation of its 4 channels.
template <int H, int LPH>
__global__ __launch_bounds__(256) void k_fwd_light(FwdArgs a) {
    constexpr int G = H * LPH;
    constexpr int NPB = 256 / G;
    const int g = threadIdx.x % G;
    const int h = g / LPH;
    const int q = g % LPH;
    const int slot = threadIdx.x / G;
    const bool writer = q == 0;
    const Walk wk = xcd_walk(a.plan, a.N, NPB, slot);
    for (int vi = wk.pos; vi < wk.end; vi += wk.stride) {
        // walk nodes by segment length when a node row is wide enough to stand alone in memory
        const int v = (G >= 8 && a.plan.order) ? a.plan.order[vi] : vi;
        const int b = a.plan.ptr[v], e = a.plan.ptr[v + 1];
        const int deg = e - b;
        if (deg > a.plan.light_max) continue;
        float4 acc = f4zero();
        float lse = 0.f;
        if (deg > 0 && deg <= 4) {
            // short segment (the common case): one batch of clamped loads, one dependent chain
            float x[4], l[4], p[4];
            float4 r[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int ii = min(b + j, e - 1);
                r[j] = ld4(a.hs + (int64_t)a.src[ii] * a.ld_hs + g * 4);
                x[j] = a.x[(int64_t)ii * H + h];
                l[j] = leaky(x[j], a.slope);
            }
            float m = l[0];
#pragma unroll
            for (int j = 1; j < 4; ++j) m = j < deg ? fmaxf(m, l[j]) : m;
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                p[j] = j < deg ? __expf(l[j] - m) : 0.f;
                s += p[j];
            }
            const float inv = 1.f / s;
            lse = m + logf(s);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (j < deg) {
                    const float w = p[j] * inv;
                    if (writer && a.attn)
                        __builtin_nontemporal_store(pack_attn(w, x[j]), a.attn + (int64_t)(b + j) * H + h);
                    fma4(acc, w, r[j]);
                }
            }
        } else if (deg > 4) {
            float m = -INFINITY;
            for (int i = b + q; i < e; i += LPH) m = fmaxf(m, leaky(a.x[(int64_t)i * H + h], a.slope));
            m = head_max<LPH>(m);
            float s = 0.f;
            for (int i = b + q; i < e; i += LPH) s += __expf(leaky(a.x[(int64_t)i * H + h], a.slope) - m);
            s = head_sum<LPH>(s);
            const float inv = 1.f / s;
            lse = m + logf(s);
            int i = b;
            for (; i + 4 <= e; i += 4) {
                float4 r[4];
                float w[4], x[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) r[j] = ld4(a.hs + (int64_t)a.src[i + j] * a.ld_hs + g * 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) x[j] = a.x[(int64_t)(i + j) * H + h];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    w[j] = __expf(leaky(x[j], a.slope) - m) * inv;
                    if (writer && a.attn) a.attn[(int64_t)(i + j) * H + h] = pack_attn(w[j], x[j]);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) fma4(acc, w[j], r[j]);
            }
            if (i < e) {  // tail of 1..3 edges: clamped batch so that the gathers overlap
                float4 r[3];
                float x[3];
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const int ii = min(i + j, e - 1);
                    r[j] = ld4(a.hs + (int64_t)a.src[ii] * a.ld_hs + g * 4);
                    x[j] = a.x[(int64_t)ii * H + h];
                }
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    if (i + j < e) {
                        const float w = __expf(leaky(x[j], a.slope) - m) * inv;
                        if (writer && a.attn) a.attn[(int64_t)(i + j) * H + h] = pack_attn(w, x[j]);
                        fma4(acc, w, r[j]);
                    }
                }
            }
        }
        if (writer && a.stat)
            *reinterpret_cast<float2 *>(a.stat + ((int64_t)v * H + h) * 2) =
                make_float2(a.et[(int64_t)v * a.ld_s + h], lse);
        epilogue<H, LPH>(a, v, g, acc);
    }
}

// Heavy segments, step 1: one wave per chunk, one lane per edge; per-chunk (max, sum exp)
// for every head.
template <int H>
__global__ __launch_bounds__(256) void k_fwd_heavy_stats(FwdArgs a) {
    const int lane = threadIdx.x & 63;
    const int c = xcd_chunk(threadIdx.x >> 6);
    if (c >= a.plan.n_chunks) return;
    const int beg = a.plan.chunk_beg[c], end = a.plan.chunk_end[c];
    float m[H], s[H];
#pragma unroll
    for (int h = 0; h < H; ++h) {
        m[h] = -INFINITY;
        s[h] = 0.f;
    }
    for (int i = beg + lane; i < end; i += 64) {
#pragma unroll
        for (int h = 0; h < H; ++h) ms_fold(m[h], s[h], leaky(a.x[(int64_t)i * H + h], a.slope), 1.f);
    }
#pragma unroll
    for (int h = 0; h < H; ++h) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            float m2 = __shfl_xor(m[h], off, 64), s2 = __shfl_xor(s[h], off, 64);
            ms_fold(m[h], s[h], m2, s2);
        }
        if (lane == 0) {
            a.ws_stats[((int64_t)c * H + h) * 2 + 0] = m[h];
            a.ws_stats[((int64_t)c * H + h) * 2 + 1] = s[h];
        }
    }
}

// Fold the per-chunk statistics of heavy node `slot` in chunk order (lanes take chunks
// round robin, then a fixed butterfly); every lane ends with the node's (max, sum).
template <int H>
__device__ __forceinline__ void node_stats(const FwdArgs &a, int slot, int lane, float (&m)[H], float (&s)[H]) {
    const int c0 = a.plan.heavy_cptr[slot], c1 = a.plan.heavy_cptr[slot + 1];
#pragma unroll
    for (int h = 0; h < H; ++h) {
        m[h] = -INFINITY;
        s[h] = 0.f;
    }
    for (int c = c0 + lane; c < c1; c += 64) {
#pragma unroll
        for (int h = 0; h < H; ++h)
            ms_fold(m[h], s[h], a.ws_stats[((int64_t)c * H + h) * 2], a.ws_stats[((int64_t)c * H + h) * 2 + 1]);
    }
#pragma unroll
    for (int h = 0; h < H; ++h) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            float m2 = __shfl_xor(m[h], off, 64), s2 = __shfl_xor(s[h], off, 64);
            ms_fold(m[h], s[h], m2, s2);
        }
    }
}

// Heavy segments, step 2: one wave per chunk; the 64/G lane groups take edges round robin,
// write the attention weights and a partial aggregate per chunk.
template <int H, int LPH>
__global__ __launch_bounds__(256) void k_fwd_heavy_agg(FwdArgs a) {
    constexpr int G = H * LPH;
    constexpr int NSG = 64 / G;
    constexpr int C = G * 4;
    const int lane = threadIdx.x & 63;
    const int c = xcd_chunk(threadIdx.x >> 6);
    if (c >= a.plan.n_chunks) return;
    const int slot = a.plan.chunk_node[c];
    const int beg = a.plan.chunk_beg[c], end = a.plan.chunk_end[c];
    float mm[H], ss[H];
    node_stats<H>(a, slot, lane, mm, ss);
    const int g = lane % G, sg = lane / G, h = g / LPH;
    float m = mm[0], s = ss[0];
#pragma unroll
    for (int k = 1; k < H; ++k)
        if (h == k) {
            m = mm[k];
            s = ss[k];
        }
    const float inv = 1.f / s;
    const bool writer = (g % LPH) == 0;
    if (a.stat && c == a.plan.heavy_cptr[slot] && sg == 0 && writer) {  // the node's first chunk records its statistics
        const int v = a.plan.heavy_node[slot];
        *reinterpret_cast<float2 *>(a.stat + ((int64_t)v * H + h) * 2) =
            make_float2(a.et[(int64_t)v * a.ld_s + h], m + logf(s));
    }
    float4 acc = f4zero();
    int i = beg + sg;
    for (; i + 3 * NSG < end; i += 4 * NSG) {
        float4 r[4];
        float w[4], x[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) r[j] = ld4(a.hs + (int64_t)a.src[i + j * NSG] * a.ld_hs + g * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) x[j] = a.x[(int64_t)(i + j * NSG) * H + h];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            w[j] = __expf(leaky(x[j], a.slope) - m) * inv;
            if (writer && a.attn) a.attn[(int64_t)(i + j * NSG) * H + h] = pack_attn(w[j], x[j]);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) fma4(acc, w[j], r[j]);
    }
    for (; i < end; i += NSG) {
        float4 r = ld4(a.hs + (int64_t)a.src[i] * a.ld_hs + g * 4);
        float x = a.x[(int64_t)i * H + h];
        float w = __expf(leaky(x, a.slope) - m) * inv;
        if (writer && a.attn) a.attn[(int64_t)i * H + h] = pack_attn(w, x);
        fma4(acc, w, r);
    }
    acc = substream_sum4<G>(acc);
    if (sg == 0) st4(a.ws_acc + (int64_t)c * C + g * 4, acc);
}

// Heavy segments, step 3: one block per heavy node sums its chunk partials in a fixed order
// (sub-groups take chunks round robin, LDS tree in sub-group order) and applies the epilogue.
template <int H, int LPH>
__global__ __launch_bounds__(256) void k_fwd_heavy_finish(FwdArgs a) {
    constexpr int G = H * LPH;
    constexpr int NSGB = 256 / G;
    constexpr int C = G * 4;
    __shared__ float4 part[256];
    const int slot = blockIdx.x;
    const int g = threadIdx.x % G, sg = threadIdx.x / G;
    const int c0 = a.plan.heavy_cptr[slot], c1 = a.plan.heavy_cptr[slot + 1];
    float4 acc = f4zero();
    for (int c = c0 + sg; c < c1; c += NSGB) add4(acc, ld4(a.ws_acc + (int64_t)c * C + g * 4));
    part[threadIdx.x] = acc;
    __syncthreads();
    if (sg == 0) {
        for (int k = 1; k < NSGB; ++k) add4(acc, part[k * G + g]);
        epilogue<H, LPH>(a, a.plan.heavy_node[slot], g, acc);
    }
}

template <int H, int LPH>
int launch_fwd(const FwdArgs &a, bool table, hipStream_t stream) {
    constexpr int G = H * LPH;
    constexpr int NPB = 256 / G;
    if (a.E > 0) {
        int grid = stride_grid(a.E, 256, 256 * 32);
        if (table)
            MAGIC_LAUNCH(prof_on() ? prof_name("gat_edge_logits", H, LPH) : "gat_edge_logits", (k_edge_logits<H, true>), grid, 256, 0, stream, a);
        else
            MAGIC_LAUNCH(prof_on() ? prof_name("gat_edge_logits", H, LPH) : "gat_edge_logits", (k_edge_logits<H, false>), grid, 256, 0, stream, a);
    }
    if (a.N > 0) {
        int grid = stride_grid(a.N, NPB, 256 * 32);
        MAGIC_LAUNCH(prof_on() ? prof_name("gat_fwd_light", H, LPH) : "gat_fwd_light", (k_fwd_light<H, LPH>), grid, 256, 0, stream, a);
    }
    if (a.plan.n_chunks > 0) {
        int cgrid = chunk_grid(a.plan.n_chunks);
        MAGIC_LAUNCH("gat_fwd_heavy_stats", (k_fwd_heavy_stats<H>), cgrid, 256, 0, stream, a);
        MAGIC_LAUNCH(prof_on() ? prof_name("gat_fwd_heavy_agg", H, LPH) : "gat_fwd_heavy_agg", (k_fwd_heavy_agg<H, LPH>), cgrid, 256, 0, stream, a);
        MAGIC_LAUNCH("gat_fwd_heavy_finish", (k_fwd_heavy_finish<H, LPH>), a.plan.n_heavy, 256, 0, stream, a);
    }
    return launch_status();
}

}  // namespace magic

extern "C" {

size_t magic_gat_fwd_workspace_bytes(int32_t n_edges, int32_t n_chunks, int32_t H, int32_t D) {
    size_t b = magic::align_up((size_t)(n_edges > 0 ? n_edges : 1) * H * sizeof(float));
    if (n_chunks > 0)
        b += magic::align_up((size_t)n_chunks * H * 2 * sizeof(float)) +
             magic::align_up((size_t)n_chunks * H * D * sizeof(float));
    return b + 256;
}

int magic_gat_fwd(int32_t N, int32_t E, int32_t H, int32_t D, const magic_plan *in_plan, const int32_t *in_src,
                  const int32_t *in_dst, const float *hs, int64_t ld_hs, const float *eh, const float *et,
                  int64_t ld_score, const float *ee_table, const int32_t *etype_csc, const float *ee_csc,
                  float slope, const float *bias, const float *res, int64_t ld_res, const uint8_t *res_mask,
                  float res_scale, const float *prelu_w,
                  float *out, int64_t ld_out, float *pre, float *attn, float *stat, void *workspace,
                  size_t workspace_bytes, void *stream_) {
    using namespace magic;
    if (N < 0 || E < 0 || H <= 0 || D <= 0) return MAGIC_ERR_INVALID_ARG;
    if (!shape_supported(H, D)) return MAGIC_ERR_UNSUPPORTED;
    if (N == 0) return MAGIC_OK;
    if (!plan_valid(in_plan) || !hs || !eh || !et || !out) return MAGIC_ERR_INVALID_ARG;
    if (E > 0 && (!in_src || !in_dst)) return MAGIC_ERR_INVALID_ARG;
    const bool table = ee_table != nullptr;
    if (E > 0 && table && !etype_csc) return MAGIC_ERR_INVALID_ARG;
    if (E > 0 && !table && !ee_csc) return MAGIC_ERR_INVALID_ARG;
    const int C = H * D;
    if (ld_hs < C || ld_out < C || (res && ld_res < C) || ld_score < H) return MAGIC_ERR_INVALID_ARG;
    if ((ld_hs | ld_out | (res ? ld_res : 0)) & 3) return MAGIC_ERR_INVALID_ARG;  // float4 rows
    FwdArgs a;
    a.N = N;
    a.E = E;
    a.plan = plan_dev(in_plan);
    a.src = in_src;
    a.dst = in_dst;
    a.hs = hs;
    a.ld_hs = ld_hs;
    a.eh = eh;
    a.et = et;
    a.ld_s = ld_score;
    a.ee_table = ee_table;
    a.etype = etype_csc;
    a.ee = ee_csc;
    a.slope = slope;
    a.bias = bias;
    a.res = res;
    a.ld_res = ld_res;
    a.res_mask = res ? res_mask : nullptr;
    a.res_scale = res_scale;
    a.prelu_w = prelu_w;
    a.out = out;
    a.ld_out = ld_out;
    a.pre = pre;
    a.attn = attn;
    a.stat = stat;
    a.ws_stats = a.ws_acc = nullptr;
    Arena ar(workspace, workspace_bytes);
    a.x = ar.take<float>((size_t)(E > 0 ? E : 1) * H);
    if (a.plan.n_chunks > 0) {
        a.ws_stats = ar.take<float>((size_t)a.plan.n_chunks * H * 2);
        a.ws_acc = ar.take<float>((size_t)a.plan.n_chunks * C);
    }
    if (!ar.ok() || !a.x) return MAGIC_ERR_WORKSPACE;
    hipStream_t stream = (hipStream_t)stream_;
    const int lph = D / 4;
#define X(h, l) \
    if (H == h && lph == l) return launch_fwd<h, l>(a, table, stream);
    MAGIC_GAT_SHAPES(X)
#undef X
    return MAGIC_ERR_UNSUPPORTED;
}

}  // extern "C"
