// GAT layer, edge phase, forward: attention logits -> leaky ReLU -> per-destination
// softmax -> weighted aggregation of source rows -> bias + residual (+ PReLU).
// HBM-bound: per edge one gathered hs row (4*H*D bytes) dominates, next to it the gather of the
// source's H head scores; everything else is streamed in CSC order.  No atomics; every reduction
// has a fixed order.  The logits x[e,:] = (eh[src e] + ee[e]) + et[dst e] are formed inside the
// sweeps; only the edges of long segments pass through memory.
//
// Launch sequence per layer:
//   gat_fwd_light         one lane group per destination with <= light_max in-edges: logits,
//                         max / sum-exp and the row gather in one pass (logits staged in LDS)
//   gat_fwd_heavy_*       longer segments in chunks of one wave each (logits + stats, aggregate,
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
    float *ws_acc;    // [n_chunks * agg_split,C]
    int agg_split;    // waves that share one chunk in the aggregation (graphs with few chunks)
    CodedIn ci;       // coded layer input (the CODED kernels): replaces hs, eh, et and res
    const uint8_t *active;  // [N] 1 = the output row of this destination is wanted; null = all of them
};

// A destination whose output nobody reads (the decoder's rows outside the masked set) is treated as a node without
// in-edges: none of its edges is visited, its output row is bias (+ residual) and its log-sum-exp +inf, which makes
// every attention weight recomputed from it in backward an exact zero.
__device__ __forceinline__ bool dst_on(const FwdArgs &a, int v) { return !a.active || a.active[v] != 0; }

// A chunk is 256 edges, taken four gathered rows at a time by one wave: a graph with few heavy edges leaves
// most of the chip idle behind 64 dependent round trips per wave.  Below ~1,500 chunks every chunk is cut
// into 2, 4 or 8 runs with a wave each; the finish step adds the runs in order.
inline int agg_split_for(int n_chunks) {
    if (n_chunks >= 1536 || n_chunks <= 0) return 1;
    if (n_chunks >= 768) return 2;
    if (n_chunks >= 384) return 4;
    return 8;
}

constexpr int SEG = 64;  // longest segment a lane group takes (the plan's light_max may not exceed it)

// Raw logit of CSC edge i (source u) under head h: (eh[u] + edge term) + et[v], the sum order of
// model/gat.py:302-304 (u_add_v after the edge term joined the source score).
template <int H, bool TABLE>
__device__ __forceinline__ float edge_logit_from(const FwdArgs &a, int i, float ehu, int h, float etv) {
    const float ee = TABLE ? a.ee_table[a.etype[i] * H + h] : a.ee[(int64_t)i * H + h];
    return (ehu + ee) + etv;
}
template <int H, bool TABLE>
__device__ __forceinline__ float edge_logit(const FwdArgs &a, int i, int u, int h, float etv) {
    return edge_logit_from<H, TABLE>(a, i, a.eh[(int64_t)u * a.ld_s + h], h, etv);
}
// the H source scores of a coded node (H == 4: one float4 of the table's eh block)
template <int H> __device__ __forceinline__ void coded_heads(const CodedIn &ci, uint32_t code, int col0, float (&o)[H]) {
    if (H == 4) {
        const float4 q = coded4(ci, code, col0);
        o[0] = q.x, o[1 % H] = q.y, o[2 % H] = q.z, o[3 % H] = q.w;
    } else {
#pragma unroll
        for (int h = 0; h < H; ++h) o[h] = coded1(ci, code, col0 + h);
    }
}

// sum / max over the G (aligned, all active) lanes of a node's group; every lane receives the same bits
template <int G> __device__ __forceinline__ float group_sum(float v) {
    v = head_sum<(G < 16 ? G : 16)>(v);
#pragma unroll
    for (int off = 16; off < G; off <<= 1) v += __shfl_xor(v, off, 64);
    return v;
}
template <int G> __device__ __forceinline__ float group_max(float v) {
    v = head_max<(G < 16 ? G : 16)>(v);
#pragma unroll
    for (int off = 16; off < G; off <<= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}

// attention weight w with the sign bit of the packed exponential (which carries "logit <= 0")
__device__ __forceinline__ float sign_of(float w, float packed) {
    return __uint_as_float(__float_as_uint(w) | (__float_as_uint(packed) & 0x80000000u));
}

template <int H, int LPH, bool CODED>
__device__ __forceinline__ void epilogue(const FwdArgs &a, int v, int g, float4 acc) {
    constexpr int C = H * LPH * 4;
    if (a.bias) add4(acc, ld4(a.bias + g * 4));
    if (CODED) {  // linear residual of the coded input row (the dropout is part of its coefficients)
        add4(acc, coded4(a.ci, a.ci.code[v], C + 2 * H + g * 4));
    } else if (a.res) {
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

// One group of W = max(G, 8) lanes per destination node; the logits are formed here and never touch HBM.
// A row needs G lanes, so a group holds W/G sub-streams that take the edges round robin in the aggregation
// (narrow decoder rows: 2 lanes per row, 4 edges at a time).
//   <= 4 in-edges with one sub-stream (the common case of the wide layers): every lane forms its head's logit
//     of each edge next to the row gather; one batch of clamped loads.
//   otherwise: the W lanes take the edges round robin, lane = edge, all H heads: source id, edge type, the
//     source's head scores -> logits into the group's LDS strip; group max; exponentials back into the strip
//     (sign bit = "logit <= 0"); group sum; then the sub-streams walk the strip (weights by LDS broadcast,
//     source ids from LDS) for the aggregation.  A logit is computed once.
template <int H, int LPH, bool TABLE, bool CODED>
__global__ __launch_bounds__(256) void k_fwd_light(FwdArgs a) {
    constexpr int G = H * LPH;
    constexpr int C = G * 4;
    constexpr int W = G < 8 ? 8 : G;
    constexpr int NSUB = W / G;
    constexpr int NPB = 256 / W;
    __shared__ float strip_w[NPB * SEG * H];
    __shared__ int strip_u[NPB * SEG];
    const int t = threadIdx.x % W;  // lane of the group: the edge it takes in the logit phase
    const int g = t % G;            // float4 of the row
    const int sub = t / G;
    const int h = g / LPH;
    const int slot = threadIdx.x / W;
    const bool writer = (g % LPH) == 0;
    float *ws = strip_w + slot * SEG * H;
    int *us = strip_u + slot * SEG;
    const Walk wk = xcd_walk(a.plan, a.N, NPB, slot);
    for (int vi = wk.pos; vi < wk.end; vi += wk.stride) {
        // walk nodes by segment length when a node row is wide enough to stand alone in memory
        const int v = (G >= 8 && a.plan.order) ? a.plan.order[vi] : vi;
        const int b = a.plan.ptr[v], e = a.plan.ptr[v + 1];
        if (e - b > a.plan.light_max) continue;
        const bool on = dst_on(a, v);
        const int deg = on ? e - b : 0;
        const uint32_t cv = CODED ? a.ci.code[v] : 0u;
        const float etv = CODED ? coded1(a.ci, cv, C + H + h) : a.et[(int64_t)v * a.ld_s + h];
        float4 acc = f4zero();
        float lse = on ? 0.f : INFINITY;
        if (NSUB == 1 && deg > 0 && deg <= 4) {
            float x[4], l[4], p[4];
            float4 r[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int ii = min(b + j, e - 1);
                const int u = a.src[ii];
                if (CODED) {
                    const uint32_t cu = a.ci.code[u];
                    r[j] = coded4(a.ci, cu, g * 4);
                    x[j] = edge_logit_from<H, TABLE>(a, ii, coded1(a.ci, cu, C + h), h, etv);
                } else {
                    r[j] = ld4(a.hs + (int64_t)u * a.ld_hs + g * 4);
                    x[j] = edge_logit<H, TABLE>(a, ii, u, h, etv);
                }
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
        } else if (deg > 0) {
            // lane = edge: logits of all heads into the strip, running max per head; two edges per trip so
            // that their id -> score chains overlap
            float et4[H], m4[H], s4[H];
            if (CODED) coded_heads<H>(a.ci, cv, C + H, et4);
#pragma unroll
            for (int k = 0; k < H; ++k) {
                if (!CODED) et4[k] = a.et[(int64_t)v * a.ld_s + k];
                m4[k] = -INFINITY;
                s4[k] = 0.f;
            }
            __builtin_amdgcn_wave_barrier();  // the previous node's reads of the strip are done
            for (int k = t; k < deg; k += 2 * W) {
                const int k2 = min(k + W, deg - 1);
                const bool two = k + W < deg;
                int u1 = a.src[b + k], u2 = a.src[b + k2];
                float x1[H], x2[H];
                if (CODED) {  // the strip keeps the code words: the row loop needs nothing else of a source
                    u1 = (int)a.ci.code[u1], u2 = (int)a.ci.code[u2];
                    float e1[H], e2[H];
                    coded_heads<H>(a.ci, (uint32_t)u1, C, e1);
                    coded_heads<H>(a.ci, (uint32_t)u2, C, e2);
#pragma unroll
                    for (int hh = 0; hh < H; ++hh) {
                        x1[hh] = edge_logit_from<H, TABLE>(a, b + k, e1[hh], hh, et4[hh]);
                        x2[hh] = edge_logit_from<H, TABLE>(a, b + k2, e2[hh], hh, et4[hh]);
                    }
                } else {
#pragma unroll
                    for (int hh = 0; hh < H; ++hh) {
                        x1[hh] = edge_logit<H, TABLE>(a, b + k, u1, hh, et4[hh]);
                        x2[hh] = edge_logit<H, TABLE>(a, b + k2, u2, hh, et4[hh]);
                    }
                }
                us[k] = u1;
#pragma unroll
                for (int hh = 0; hh < H; ++hh) {
                    ws[k * H + hh] = x1[hh];
                    m4[hh] = fmaxf(m4[hh], leaky(x1[hh], a.slope));
                }
                if (two) {
                    us[k2] = u2;
#pragma unroll
                    for (int hh = 0; hh < H; ++hh) {
                        ws[k2 * H + hh] = x2[hh];
                        m4[hh] = fmaxf(m4[hh], leaky(x2[hh], a.slope));
                    }
                }
            }
#pragma unroll
            for (int hh = 0; hh < H; ++hh) m4[hh] = group_max<W>(m4[hh]);
            for (int k = t; k < deg; k += W) {
#pragma unroll
                for (int hh = 0; hh < H; ++hh) {
                    const float x = ws[k * H + hh];
                    const float p = __expf(leaky(x, a.slope) - m4[hh]);
                    s4[hh] += p;
                    ws[k * H + hh] = pack_attn(p, x);
                }
            }
#pragma unroll
            for (int hh = 0; hh < H; ++hh) s4[hh] = group_sum<W>(s4[hh]);
            float m = m4[0], s = s4[0];
#pragma unroll
            for (int k = 1; k < H; ++k)
                if (h == k) {
                    m = m4[k];
                    s = s4[k];
                }
            const float inv = 1.f / s;
            lse = m + logf(s);
            __builtin_amdgcn_wave_barrier();  // the strip is complete before anyone walks it
            for (int k = sub; k < deg; k += 4 * NSUB) {
                float4 r[4];
                float pw[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int kk = min(k + j * NSUB, deg - 1);
                    r[j] = CODED ? coded4(a.ci, (uint32_t)us[kk], g * 4) : ld4(a.hs + (int64_t)us[kk] * a.ld_hs + g * 4);
                    pw[j] = ws[kk * H + h];
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (k + j * NSUB < deg) {
                        const float w = fabsf(pw[j]) * inv;
                        if (writer && a.attn) a.attn[(int64_t)(b + k + j * NSUB) * H + h] = sign_of(w, pw[j]);
                        fma4(acc, w, r[j]);
                    }
                }
            }
        }
        if (NSUB > 1) {  // sub-stream partials in a fixed order; sub-stream 0 keeps the sum
#pragma unroll
            for (int off = G; off < W; off <<= 1) {
                acc.x += __shfl_xor(acc.x, off, 64);
                acc.y += __shfl_xor(acc.y, off, 64);
                acc.z += __shfl_xor(acc.z, off, 64);
                acc.w += __shfl_xor(acc.w, off, 64);
            }
        }
        if (sub == 0) {
            if (writer && a.stat)
                *reinterpret_cast<float2 *>(a.stat + ((int64_t)v * H + h) * 2) = make_float2(etv, lse);
            epilogue<H, LPH, CODED>(a, v, g, acc);
        }
    }
}

// Heavy segments, step 1: one wave per chunk, one lane per edge; forms the logits of the chunk's edges
// (kept in x for step 2) and the per-chunk (max, sum exp) of every head.
template <int H, bool TABLE, bool CODED>
__global__ __launch_bounds__(256) void k_fwd_heavy_stats(FwdArgs a) {
    const int lane = threadIdx.x & 63;
    const int c = xcd_chunk(threadIdx.x >> 6);
    if (c >= a.plan.n_chunks) return;
    const int beg = a.plan.chunk_beg[c], end = a.plan.chunk_end[c];
    const int v = a.plan.heavy_node[a.plan.chunk_node[c]];
    if (!dst_on(a, v)) return;
    float m[H], s[H], etv[H];
    if (CODED) coded_heads<H>(a.ci, a.ci.code[v], a.ci.C + H, etv);
#pragma unroll
    for (int h = 0; h < H; ++h) {
        m[h] = -INFINITY;
        s[h] = 0.f;
        if (!CODED) etv[h] = a.et[(int64_t)v * a.ld_s + h];
    }
    for (int i0 = beg + lane; i0 < end; i0 += 256) {  // four edges per lane and trip: their id -> score chains overlap
        int ii[4], u[4];
        float x[4][H];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            ii[j] = min(i0 + 64 * j, end - 1);
            u[j] = a.src[ii[j]];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (CODED) {
                float eh[H];
                coded_heads<H>(a.ci, a.ci.code[u[j]], a.ci.C, eh);
#pragma unroll
                for (int h = 0; h < H; ++h) x[j][h] = edge_logit_from<H, TABLE>(a, ii[j], eh[h], h, etv[h]);
            } else {
#pragma unroll
                for (int h = 0; h < H; ++h) x[j][h] = edge_logit<H, TABLE>(a, ii[j], u[j], h, etv[h]);
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (i0 + 64 * j < end) {
#pragma unroll
                for (int h = 0; h < H; ++h) {
                    a.x[(int64_t)ii[j] * H + h] = x[j][h];
                    ms_fold(m[h], s[h], leaky(x[j][h], a.slope), 1.f);
                }
            }
        }
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
template <int H, int LPH, bool CODED>
__global__ __launch_bounds__(256) void k_fwd_heavy_agg(FwdArgs a) {
    constexpr int G = H * LPH;
    constexpr int NSG = 64 / G;
    constexpr int C = G * 4;
    const int lane = threadIdx.x & 63;
    const int unit = xcd_chunk(threadIdx.x >> 6);  // (chunk, run of the chunk)
    const int c = unit / a.agg_split, run = unit % a.agg_split;
    if (c >= a.plan.n_chunks) return;
    const int slot = a.plan.chunk_node[c];
    int beg = a.plan.chunk_beg[c], end = a.plan.chunk_end[c];
    if (a.agg_split > 1) {
        const int per = (end - beg + a.agg_split - 1) / a.agg_split;
        beg = min(beg + run * per, end);
        end = min(beg + per, end);
    }
    if (!dst_on(a, a.plan.heavy_node[slot])) {  // no statistics were formed: only the log-sum-exp is recorded
        if (a.stat && c == a.plan.heavy_cptr[slot] && run == 0 && lane < H)
            *reinterpret_cast<float2 *>(a.stat + ((int64_t)a.plan.heavy_node[slot] * H + lane) * 2) = make_float2(0.f, INFINITY);
        return;
    }
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
    if (a.stat && c == a.plan.heavy_cptr[slot] && run == 0 && sg == 0 && writer) {  // the node's first chunk records its statistics
        const int v = a.plan.heavy_node[slot];
        const float etv = CODED ? coded1(a.ci, a.ci.code[v], C + H + h) : a.et[(int64_t)v * a.ld_s + h];
        *reinterpret_cast<float2 *>(a.stat + ((int64_t)v * H + h) * 2) = make_float2(etv, m + logf(s));
    }
    float4 acc = f4zero();
    int i = beg + sg;
    for (; i + 3 * NSG < end; i += 4 * NSG) {
        float4 r[4];
        float w[4], x[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int u = a.src[i + j * NSG];
            r[j] = CODED ? coded4(a.ci, a.ci.code[u], g * 4) : ld4(a.hs + (int64_t)u * a.ld_hs + g * 4);
        }
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
        const int u = a.src[i];
        float4 r = CODED ? coded4(a.ci, a.ci.code[u], g * 4) : ld4(a.hs + (int64_t)u * a.ld_hs + g * 4);
        float x = a.x[(int64_t)i * H + h];
        float w = __expf(leaky(x, a.slope) - m) * inv;
        if (writer && a.attn) a.attn[(int64_t)i * H + h] = pack_attn(w, x);
        fma4(acc, w, r);
    }
    acc = substream_sum4<G>(acc);
    if (sg == 0) st4(a.ws_acc + (int64_t)unit * C + g * 4, acc);
}

// Heavy segments, step 3: one block per heavy node sums its chunk partials in a fixed order
// (sub-groups take chunks round robin, LDS tree in sub-group order) and applies the epilogue.
template <int H, int LPH, bool CODED>
__global__ __launch_bounds__(256) void k_fwd_heavy_finish(FwdArgs a) {
    constexpr int G = H * LPH;
    constexpr int NSGB = 256 / G;
    constexpr int C = G * 4;
    __shared__ float4 part[256];
    const int slot = blockIdx.x;
    const int g = threadIdx.x % G, sg = threadIdx.x / G;
    const int c0 = a.plan.heavy_cptr[slot] * a.agg_split, c1 = a.plan.heavy_cptr[slot + 1] * a.agg_split;
    float4 acc = f4zero();
    if (dst_on(a, a.plan.heavy_node[slot]))
        for (int c = c0 + sg; c < c1; c += NSGB) add4(acc, ld4(a.ws_acc + (int64_t)c * C + g * 4));
    part[threadIdx.x] = acc;
    __syncthreads();
    if (sg == 0) {
        for (int k = 1; k < NSGB; ++k) add4(acc, part[k * G + g]);
        epilogue<H, LPH, CODED>(a, a.plan.heavy_node[slot], g, acc);
    }
}

template <int H, int LPH, bool TABLE, bool CODED = false>
int launch_fwd(const FwdArgs &a, hipStream_t stream) {
    constexpr int G = H * LPH;
    constexpr int NPB = 256 / (G < 8 ? 8 : G);
    if (a.N > 0) {
        int grid = stride_grid(a.N, NPB, 256 * 32);
        MAGIC_LAUNCH(prof_on() ? prof_name(CODED ? "gat_fwd_light_coded" : "gat_fwd_light", H, LPH) : "gat_fwd_light", (k_fwd_light<H, LPH, TABLE, CODED>), grid, 256, 0, stream, a);
    }
    if (a.plan.n_chunks > 0) {
        int cgrid = chunk_grid(a.plan.n_chunks);
        MAGIC_LAUNCH(prof_on() ? prof_name("gat_fwd_heavy_stats", H, LPH) : "gat_fwd_heavy_stats", (k_fwd_heavy_stats<H, TABLE, CODED>), cgrid, 256, 0, stream, a);
        MAGIC_LAUNCH(prof_on() ? prof_name("gat_fwd_heavy_agg", H, LPH) : "gat_fwd_heavy_agg", (k_fwd_heavy_agg<H, LPH, CODED>),
                     chunk_grid(a.plan.n_chunks * a.agg_split), 256, 0, stream, a);
        MAGIC_LAUNCH("gat_fwd_heavy_finish", (k_fwd_heavy_finish<H, LPH, CODED>), a.plan.n_heavy, 256, 0, stream, a);
    }
    return launch_status();
}

}  // namespace magic

extern "C" {

size_t magic_gat_fwd_workspace_bytes(int32_t n_edges, int32_t n_chunks, int32_t H, int32_t D) {
    size_t b = magic::align_up((size_t)(n_edges > 0 ? n_edges : 1) * H * sizeof(float));
    if (n_chunks > 0)
        b += magic::align_up((size_t)n_chunks * H * 2 * sizeof(float)) +
             magic::align_up((size_t)n_chunks * magic::agg_split_for(n_chunks) * H * D * sizeof(float));
    return b + 256;
}

// both entry points: `ci` != NULL selects the coded-input kernels (hs, eh, et and res are then not read)
static int gat_fwd_run(int32_t N, int32_t E, int32_t H, int32_t D, const magic_plan *in_plan, const int32_t *in_src,
                       const int32_t *in_dst, const float *hs, int64_t ld_hs, const float *eh, const float *et,
                       int64_t ld_score, const float *ee_table, const int32_t *etype_csc, const float *ee_csc,
                       float slope, const float *bias, const float *res, int64_t ld_res, const uint8_t *res_mask,
                       float res_scale, const float *prelu_w, float *out, int64_t ld_out, float *pre, float *attn,
                       float *stat, void *workspace, size_t workspace_bytes, void *stream_,
                       const magic_coded_input *ci, const uint8_t *dst_active) {
    using namespace magic;
    if (N < 0 || E < 0 || H <= 0 || D <= 0) return MAGIC_ERR_INVALID_ARG;
    if (!shape_supported(H, D)) return MAGIC_ERR_UNSUPPORTED;
    if (N == 0) return MAGIC_OK;
    if (!plan_valid(in_plan) || !out) return MAGIC_ERR_INVALID_ARG;
    if (ci) {
        if (!ci->code || !ci->table || ci->n_dim <= 0 || ci->n_dim > MAGIC_CODED_MAX_DIM) return MAGIC_ERR_INVALID_ARG;
        if (ci->ld_table < 2 * H * D + 2 * H || (ci->ld_table & 3) || ((uintptr_t)ci->table & 15)) return MAGIC_ERR_INVALID_ARG;
        if (H != 4 || !ee_table) return MAGIC_ERR_UNSUPPORTED;  // compiled for the encoder's four heads, typed edges
        hs = eh = et = out;  // (placeholders: never read)
        ld_hs = ld_out;
        ld_score = H;
        res = nullptr;
    }
    if (!hs || !eh || !et) return MAGIC_ERR_INVALID_ARG;
    if (dst_active && attn) return MAGIC_ERR_UNSUPPORTED;  // (the edges of an unwanted destination are not visited)
    if (E > 0 && !in_src) return MAGIC_ERR_INVALID_ARG;
    if (in_plan->light_max > magic::SEG) return MAGIC_ERR_UNSUPPORTED;  // a lane group's strip holds SEG edges
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
    a.agg_split = agg_split_for(a.plan.n_chunks);
    a.active = dst_active;
    a.ci = CodedIn{nullptr, nullptr, 0, C};
    if (ci) a.ci = CodedIn{ci->code, ci->table, (int)ci->ld_table, C};
    Arena ar(workspace, workspace_bytes);
    a.x = ar.take<float>((size_t)(E > 0 ? E : 1) * H);
    if (a.plan.n_chunks > 0) {
        a.ws_stats = ar.take<float>((size_t)a.plan.n_chunks * H * 2);
        a.ws_acc = ar.take<float>((size_t)a.plan.n_chunks * a.agg_split * C);
    }
    if (!ar.ok() || !a.x) return MAGIC_ERR_WORKSPACE;
    hipStream_t stream = (hipStream_t)stream_;
    const int lph = D / 4;
    if (ci) {
        switch (lph) {
            case 2: return launch_fwd<4, 2, true, true>(a, stream);
            case 4: return launch_fwd<4, 4, true, true>(a, stream);
            case 8: return launch_fwd<4, 8, true, true>(a, stream);
            case 16: return launch_fwd<4, 16, true, true>(a, stream);
        }
        return MAGIC_ERR_UNSUPPORTED;
    }
#define X(h, l) \
    if (H == h && lph == l) return table ? launch_fwd<h, l, true>(a, stream) : launch_fwd<h, l, false>(a, stream);
    MAGIC_GAT_SHAPES(X)
#undef X
    return MAGIC_ERR_UNSUPPORTED;
}

int magic_gat_fwd(int32_t N, int32_t E, int32_t H, int32_t D, const magic_plan *in_plan, const int32_t *in_src,
                  const int32_t *in_dst, const float *hs, int64_t ld_hs, const float *eh, const float *et,
                  int64_t ld_score, const float *ee_table, const int32_t *etype_csc, const float *ee_csc,
                  float slope, const float *bias, const float *res, int64_t ld_res, const uint8_t *res_mask,
                  float res_scale, const float *prelu_w,
                  float *out, int64_t ld_out, float *pre, float *attn, float *stat, void *workspace,
                  size_t workspace_bytes, void *stream_) {
    return gat_fwd_run(N, E, H, D, in_plan, in_src, in_dst, hs, ld_hs, eh, et, ld_score, ee_table, etype_csc, ee_csc,
                       slope, bias, res, ld_res, res_mask, res_scale, prelu_w, out, ld_out, pre, attn, stat, workspace,
                       workspace_bytes, stream_, nullptr, nullptr);
}

int magic_gat_fwd_active(int32_t N, int32_t E, int32_t H, int32_t D, const magic_plan *in_plan, const int32_t *in_src,
                         const int32_t *in_dst, const float *hs, int64_t ld_hs, const float *eh, const float *et,
                         int64_t ld_score, const float *ee_table, const int32_t *etype_csc, const float *ee_csc,
                         float slope, const float *bias, const float *res, int64_t ld_res, const uint8_t *res_mask,
                         float res_scale, const float *prelu_w, float *out, int64_t ld_out, float *pre, float *attn,
                         float *stat, const uint8_t *dst_active, void *workspace, size_t workspace_bytes,
                         void *stream_) {
    return gat_fwd_run(N, E, H, D, in_plan, in_src, in_dst, hs, ld_hs, eh, et, ld_score, ee_table, etype_csc, ee_csc,
                       slope, bias, res, ld_res, res_mask, res_scale, prelu_w, out, ld_out, pre, attn, stat, workspace,
                       workspace_bytes, stream_, nullptr, dst_active);
}

int magic_gat_fwd_coded(int32_t N, int32_t E, int32_t H, int32_t D, const magic_plan *in_plan, const int32_t *in_src,
                        const magic_coded_input *input, const float *ee_table, const int32_t *etype_csc, float slope,
                        const float *bias, const float *prelu_w, float *out, int64_t ld_out, float *pre, float *stat,
                        void *workspace, size_t workspace_bytes, void *stream_) {
    if (!input) return MAGIC_ERR_INVALID_ARG;
    return gat_fwd_run(N, E, H, D, in_plan, in_src, nullptr, nullptr, 0, nullptr, nullptr, 0, ee_table, etype_csc,
                       nullptr, slope, bias, nullptr, 0, nullptr, 1.f, prelu_w, out, ld_out, pre, nullptr, stat,
                       workspace, workspace_bytes, stream_, input, nullptr);
}

}  // extern "C"
