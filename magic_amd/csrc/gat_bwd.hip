// GAT layer, edge phase, backward.  Segmented sweeps, no atomics, one gather of node rows and one
// transposition of per-edge data between source order (CSR) and destination order (CSC):
//   node adjoint      PReLU / bias adjoints of the incoming gradient (skipped when the caller
//                     already holds the pre-activation gradient);
//   source sweep      (CSR) per edge u->v gathers d_pre[v] once and uses it twice:
//                     d_hs[u] += a_e d_pre[v]   and   t_e = <hs[u], d_pre[v]> per head
//                     (hs[u] is the sweep's own row).  The attention weight is recomputed, not
//                     fetched: a_e = exp(leaky(eh[u] + ee_e + et[v]) - lse[v]) with (et[v], lse[v])
//                     from the forward's per-destination table [N,H,2] (small, hubs stay cached)
//                     and ee_e from the edge-type table in LDS.  (a_e, t_e) pairs leave in CSR
//                     order as streaming stores;
//   destination pass  (CSC) per destination v gathers the pairs of its in-edges through in_pos
//                     (the one random [E,H] access of the whole backward), once: delta_v = sum a_e t_e
//                     and d_et[v] = sum a_e (t_e - delta_v) leaky'_e from three segment sums;
//                     stores delta_v;
//   source pass 2     (CSR) streams the pairs: d_logit_e = a_e (t_e - delta[dst e]) leaky'_e with
//                     delta from the per-node table, d_eh[u] = sum over out-edges; d_logit leaves
//                     in CSR order (and is scattered to CSC order only when the caller wants it);
//   edge-type table   fixed-order reduction of d_logit per edge type (CSR order).
#include "gat_common.h"

namespace magic {

struct BwdArgs {
    int N, E;
    PlanDev pin;   // CSC plan
    PlanDev pout;  // CSR plan
    const int32_t *out_dst;
    const int32_t *out_pos;  // [E] CSC position of every CSR edge
    const int32_t *in_pos;   // [E] CSR position of every CSC edge
    const float *hs;
    int64_t ld_hs;
    const float *eh;  // [N,ld_s] head scores of the node as a source
    int64_t ld_s;
    const float *stat;      // [N,H,2] (et, log-sum-exp of the in-edge logits) per destination and head
    const float *ee_table;  // [e_dim,H] or null
    const int32_t *etype;   // [E] edge type, CSR order (with ee_table)
    const float *ee;        // [E,H] dense edge term, CSC order (without ee_table)
    int e_dim;
    float slope;
    const float *pre;      // [N,C] pre-activation (valid when the slope is <= 0)
    const float *act_out;  // [N,ld_act] forward output (pre-activation recovered from it when slope > 0)
    int64_t ld_act;
    const float *prelu_w;
    const float *d_out;
    int64_t ld_dout;
    float *d_pre;  // [N,ld_dpre]
    int64_t ld_dpre;
    float *d_y;
    int64_t ld_dy;
    float *d_table;  // [e_dim,H]
    float *dl;       // [E,H] d_logit, CSC order, or null when the caller does not need it
    float *d_bias;   // [C]
    float *d_prelu;  // [1]
    // workspace
    float *pairs;     // [E,H,2] (attention weight with the leaky-ReLU sign, <hs[src], d_pre[dst]>), CSR order
    float *dl_csr;    // [E,H] d_logit, CSR order
    float *delta;     // [N,H] sum over in-edges of a * t
    float *ws_bias;   // [grid_light, C]
    float *ws_prelu;  // [grid_light]
    int grid_light;
    float *ws_delta;  // [n_chunks_in, H]
    float *ws_det;    // [n_chunks_in, H, 2]
    float *ws_acc;    // [n_chunks_out, C]
    float *ws_deh;    // [n_chunks_out, H]
    float *ws_table;  // [grid_table, e_dim*H]
    int grid_table;
    CodedIn ci;  // coded layer input (the CODED source sweeps): replaces hs and eh
    const uint8_t *active;  // [N] 1 = gradient arrives at this destination; null = at all of them
};

// ---- destinations without incoming gradient ---------------------------------------------------
// When the caller marks the destinations whose d_out row can be non-zero (the decoder: the masked nodes), the edges
// into all the others contribute exact zeros: their attention weight is recomputed from the +inf log-sum-exp the
// forward sweep recorded for them.  The sweeps still stream those edges, but need not gather anything for them from
// 2 x N different rows: the CSR destination ids are rewritten so that every such edge points at ONE unmarked node
// (the first among nodes 0..255; without one the ids stay as they are), whose rows then sit in cache, and the
// destination pass skips the unmarked nodes.
__global__ __launch_bounds__(256) void k_active_dst(int N, int E, const int32_t *__restrict__ out_dst,
                                                    const uint8_t *__restrict__ active, int32_t *__restrict__ eff) {
    __shared__ int first[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool off = (int)threadIdx.x < N && active[threadIdx.x] == 0;
    const unsigned long long m = __ballot(off);
    if (lane == 0) first[wave] = m ? wave * 64 + __builtin_ctzll(m) : -1;
    __syncthreads();
    int park = -1;
#pragma unroll
    for (int w = 3; w >= 0; --w)
        if (first[w] >= 0) park = first[w];
    // four edges per thread and trip (both arrays are 16-byte aligned): the four flag lookups are in flight together
    const int E4 = E / 4;
    for (int j = blockIdx.x * 256 + threadIdx.x; j < E4; j += gridDim.x * 256) {
        const int4 v = reinterpret_cast<const int4 *>(out_dst)[j];
        const uint8_t f0 = active[v.x], f1 = active[v.y], f2 = active[v.z], f3 = active[v.w];
        int4 o;
        o.x = (f0 || park < 0) ? v.x : park, o.y = (f1 || park < 0) ? v.y : park;
        o.z = (f2 || park < 0) ? v.z : park, o.w = (f3 || park < 0) ? v.w : park;
        reinterpret_cast<int4 *>(eff)[j] = o;
    }
    if (blockIdx.x == 0 && (int)threadIdx.x < E - E4 * 4) {
        const int j = E4 * 4 + threadIdx.x;
        const int v = out_dst[j];
        eff[j] = (active[v] || park < 0) ? v : park;
    }
}
__device__ __forceinline__ bool dst_on(const BwdArgs &a, int v) { return !a.active || a.active[v] != 0; }

// ---- node adjoint: d_pre = d_out * PReLU'(pre), column sums (bias gradient), slope gradient ---
template <int H, int LPH>
__global__ __launch_bounds__(256) void k_bwd_node_adjoint(BwdArgs a) {
    constexpr int G = H * LPH;
    constexpr int NPB = 256 / G;
    constexpr int C = G * 4;
    __shared__ float4 red4[256];
    __shared__ float red1[256];
    const int g = threadIdx.x % G;
    const int slot = threadIdx.x / G;
    const float pw = a.prelu_w ? a.prelu_w[0] : 1.f;
    const float iw = pw > 0.f ? 1.f / pw : 0.f;
    float4 bsum = f4zero();
    float wsum = 0.f;
    for (int v = blockIdx.x * NPB + slot; v < a.N; v += gridDim.x * NPB) {
        float4 go = ld4(a.d_out + (int64_t)v * a.ld_dout + g * 4);
        if (a.prelu_w) {
            float4 p;
            if (pw > 0.f) {  // PReLU is invertible for a positive slope
                p = ld4(a.act_out + (int64_t)v * a.ld_act + g * 4);
                p.x = p.x > 0.f ? p.x : p.x * iw;
                p.y = p.y > 0.f ? p.y : p.y * iw;
                p.z = p.z > 0.f ? p.z : p.z * iw;
                p.w = p.w > 0.f ? p.w : p.w * iw;
            } else {
                p = ld4(a.pre + (int64_t)v * C + g * 4);
            }
            wsum += (p.x < 0.f ? go.x * p.x : 0.f) + (p.y < 0.f ? go.y * p.y : 0.f) +
                    (p.z < 0.f ? go.z * p.z : 0.f) + (p.w < 0.f ? go.w * p.w : 0.f);
            go.x = p.x > 0.f ? go.x : pw * go.x;
            go.y = p.y > 0.f ? go.y : pw * go.y;
            go.z = p.z > 0.f ? go.z : pw * go.z;
            go.w = p.w > 0.f ? go.w : pw * go.w;
        }
        if (a.d_pre != a.d_out) st4(a.d_pre + (int64_t)v * a.ld_dpre + g * 4, go);
        add4(bsum, go);
    }
    // per-block partials, summed in slot order
    red4[threadIdx.x] = bsum;
    red1[threadIdx.x] = wsum;
    __syncthreads();
    if (slot == 0) {
        for (int k = 1; k < NPB; ++k) add4(bsum, red4[k * G + g]);
        st4(a.ws_bias + (int64_t)blockIdx.x * C + g * 4, bsum);
    }
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int k = 0; k < 256; ++k) t += red1[k];
        a.ws_prelu[blockIdx.x] = t;
    }
}

// final fixed-order reduction of the per-block partials: block c < C reduces channel c,
// block C reduces the PReLU-slope partials
__global__ __launch_bounds__(256) void k_bwd_reduce_bias(BwdArgs a, int C) {
    __shared__ float red[256];
    const int c = blockIdx.x;
    float t = 0.f;
    if (c < C) {
        for (int k = threadIdx.x; k < a.grid_light; k += 256) t += a.ws_bias[(int64_t)k * C + c];
    } else {
        for (int k = threadIdx.x; k < a.grid_light; k += 256) t += a.ws_prelu[k];
    }
    red[threadIdx.x] = t;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if (c < C) {
            if (a.d_bias) a.d_bias[c] = red[0];
        } else if (a.d_prelu) {
            a.d_prelu[0] = red[0];
        }
    }
}

// ---- source sweep --------------------------------------------------------------------------
constexpr int MAX_E_DIM = 32;

// edge-type table into LDS (every thread of the block calls this once, before any early exit)
template <int H, bool TABLE> __device__ __forceinline__ void stage_table(const BwdArgs &a, float *tbl) {
    if (TABLE) {
        for (int i = threadIdx.x; i < a.e_dim * H; i += blockDim.x) tbl[i] = a.ee_table[i];
        __syncthreads();
    }
}

// edge term of CSR edge j, head h
template <int H, bool TABLE>
__device__ __forceinline__ float edge_term(const BwdArgs &a, const float *tbl, int j, int h) {
    return TABLE ? tbl[a.etype[j] * H + h] : a.ee[(int64_t)a.out_pos[j] * H + h];
}

// One lane group per source node u with <= light_max out-edges.  Per edge: the destination's
// gradient row and its (et, lse) pair are gathered once; the pair (a_e, t_e) is streamed out.
//
// The walk is software-pipelined so that one batch of SB edges costs one memory round trip instead
// of the chain walk order -> segment bounds -> destination ids -> rows: while the rows of the current
// batch are in flight, the destination ids / edge types of the NEXT batch (of this node or of the next
// node), the next node's own row, the segment bounds of the node after it and the walk-order entry of
// the one after that are requested as well.  Every address of an iteration is known when it starts.
// All loads are unconditional (clamped indices); only stores and accumulations are predicated.
template <int H, int LPH, bool TABLE, bool CODED>
__global__ __launch_bounds__(256) void k_bwd_src_light(BwdArgs a) {
    constexpr int G = H * LPH;
    constexpr int C = G * 4;
    constexpr int NPB = 256 / G;
    constexpr int SB = 4;  // edges per batch (8 for wide rows: fewer resident waves, slower)
    __shared__ float tbl[TABLE ? MAX_E_DIM * H : 1];
    stage_table<H, TABLE>(a, tbl);
    const int g = threadIdx.x % G;
    const int h = g / LPH;
    const bool writer = (g % LPH) == 0;
    const int slot = threadIdx.x / G;
    const Walk wk = xcd_walk(a.pout, a.N, NPB, slot);
    const int stride = wk.stride;
    const bool ordered = G >= 8 && a.pout.order != nullptr;
    const int last_pos = max(wk.end - 1, 0);
    int i0 = wk.pos;  // walk index of the current node
    auto node_at = [&](int i) {
        i = min(i, last_pos);
        return ordered ? a.pout.order[i] : i;
    };
    // edge slot of batch element k (clamped into the segment; into the edge array for empty segments)
    auto slot_of = [&](int j, int e, int k) { return max(min(j + k, e - 1), 0); };
    // ---- pipeline fill
    int u0 = node_at(i0), u1 = node_at(i0 + stride), u2 = node_at(i0 + 2 * stride);
    int b0 = a.pout.ptr[u0], e0 = a.pout.ptr[u0 + 1];
    int b1 = a.pout.ptr[u1], e1 = a.pout.ptr[u1 + 1];
    int j = b0;
    int v[SB], jj[SB];
    float ee[SB];
#pragma unroll
    for (int k = 0; k < SB; ++k) {
        jj[k] = slot_of(j, e0, k);
        v[k] = a.out_dst[jj[k]];
        ee[k] = edge_term<H, TABLE>(a, tbl, jj[k], h);
    }
    float4 hu;
    float ehu;
    if (CODED) {
        const uint32_t cu = a.ci.code[u0];
        hu = coded4(a.ci, cu, g * 4);
        ehu = coded1(a.ci, cu, C + h);
    } else {
        hu = ld4_nt(a.hs + (int64_t)u0 * a.ld_hs + g * 4);
        ehu = a.eh[(int64_t)u0 * a.ld_s + h];
    }
    float4 acc = f4zero();
    while (i0 < wk.end) {
        const bool light = e0 - b0 <= a.pout.light_max;
        const bool last = !light || j + SB >= e0;  // this batch ends the node
        // 1. gathers of the current batch
        float4 r[SB];
        float2 st[SB];
#pragma unroll
        for (int k = 0; k < SB; ++k) {
            r[k] = ld4(a.d_pre + (int64_t)v[k] * a.ld_dpre + g * 4);
            st[k] = *reinterpret_cast<const float2 *>(a.stat + ((int64_t)v[k] * H + h) * 2);
        }
        // 2. requests for what the following iterations need
        const int nj = last ? b1 : j + SB, ne = last ? e1 : e0;
        int nv[SB], njj[SB];
        float nee[SB];
#pragma unroll
        for (int k = 0; k < SB; ++k) {
            njj[k] = slot_of(nj, ne, k);
            nv[k] = a.out_dst[njj[k]];
            nee[k] = edge_term<H, TABLE>(a, tbl, njj[k], h);
        }
        float4 nhu = hu;
        float nehu = ehu;
        int b2 = 0, e2 = 0, u3 = 0;
        if (last) {
            if (CODED) {
                const uint32_t cu = a.ci.code[u1];
                nhu = coded4(a.ci, cu, g * 4);
                nehu = coded1(a.ci, cu, C + h);
            } else {
                nhu = ld4_nt(a.hs + (int64_t)u1 * a.ld_hs + g * 4);
                nehu = a.eh[(int64_t)u1 * a.ld_s + h];
            }
            b2 = a.pout.ptr[u2], e2 = a.pout.ptr[u2 + 1];
            u3 = node_at(i0 + 3 * stride);
        }
        // 3. the batch itself
#pragma unroll
        for (int k = 0; k < SB; ++k) {
            const float x = (ehu + ee[k]) + st[k].x;
            const float w = __expf(leaky(x, a.slope) - st[k].y);
            const float tt = head_sum<LPH>(dot4(r[k], hu));
            if (light && j + k < e0) {
                if (writer) st2_nt(a.pairs + ((int64_t)(j + k) * H + h) * 2, pack_attn(w, x), tt);
                fma4(acc, w, r[k]);
            }
        }
        // 4. advance
        if (last) {
            if (light) st4_nt(a.d_y + (int64_t)u0 * a.ld_dy + g * 4, acc);
            acc = f4zero();
            u0 = u1, b0 = b1, e0 = e1;
            u1 = u2, b1 = b2, e1 = e2;
            u2 = u3;
            hu = nhu, ehu = nehu;
            i0 += stride;
            j = b0;
        } else {
            j += SB;
        }
#pragma unroll
        for (int k = 0; k < SB; ++k) v[k] = nv[k], ee[k] = nee[k];
    }
}

// ---- source sweep, one wavefront per node (rows of 64 float4: H = 4, D = 64) ------------------------
// With a whole wavefront on one node the per-edge scalars are loaded ONCE per node, lane = edge (light
// segments have at most 64 edges): destination ids and edge types in one 256-byte load each, the
// destinations' (et, lse) pairs in two gathers, the four attention weights of an edge by its own lane (four
// exponentials per lane and node instead of one per lane and edge).  The weights go through a
// wave-private LDS strip; the row loop takes the destination of edge k with v_readlane (scalar base
// address) and double-buffers its batches of four rows, so nothing in it waits on anything but the rows
// themselves.  The ids / types of the NEXT node are requested while this one runs.
template <bool TABLE, bool CODED>
__global__ __launch_bounds__(256) void k_bwd_src_wave(BwdArgs a) {
    constexpr int H = 4, LPH = 16, SB = 4, C = 256;
    __shared__ float tbl[TABLE ? MAX_E_DIM * H : 1];
    __shared__ __align__(16) float wls[4][64 * H];
    stage_table<H, TABLE>(a, tbl);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int h = lane / LPH;
    const bool writer = (lane % LPH) == 0;
    float *wl = wls[wave];
    const Walk wk = xcd_walk(a.pout, a.N, 4, wave);
    const int stride = wk.stride;
    const bool ordered = a.pout.order != nullptr;
    const int last_pos = max(wk.end - 1, 0);
    auto node_at = [&](int i) {
        i = min(i, last_pos);
        return ordered ? a.pout.order[i] : i;
    };
    auto edge_slot = [&](int b, int e) { return max(min(b + lane, e - 1), 0); };
    int i0 = wk.pos;
    int u0 = node_at(i0), u1 = node_at(i0 + stride), u2 = node_at(i0 + 2 * stride);
    int b0 = a.pout.ptr[u0], e0 = a.pout.ptr[u0 + 1];
    int b1 = a.pout.ptr[u1], e1 = a.pout.ptr[u1 + 1];
    int idx = a.out_dst[edge_slot(b0, e0)];
    int ty = TABLE ? a.etype[edge_slot(b0, e0)] : a.out_pos[edge_slot(b0, e0)];  // dense path: the CSC slot of ee
    float4 hu, eh4;
    if (CODED) {
        const uint32_t cu = a.ci.code[u0];
        hu = coded4(a.ci, cu, lane * 4);
        eh4 = coded4(a.ci, cu, C);
    } else {
        hu = ld4_nt(a.hs + (int64_t)u0 * a.ld_hs + lane * 4);
        eh4 = ld4(a.eh + (int64_t)u0 * a.ld_s);
    }
    while (i0 < wk.end) {
        const int deg = e0 - b0;
        const bool light = deg <= a.pout.light_max;
        // 1. per-edge scalars of this node (lane = edge): the destination's statistics
        const float4 s01 = ld4(a.stat + (int64_t)idx * (2 * H)), s23 = ld4(a.stat + (int64_t)idx * (2 * H) + 4);
        float4 eev;
        if (TABLE)
            eev = make_float4(tbl[ty * H + 0], tbl[ty * H + 1], tbl[ty * H + 2], tbl[ty * H + 3]);
        else
            eev = ld4(a.ee + (int64_t)ty * H);
        // 2. first batch of rows
        float4 rA[SB], rB[SB];
        const int nb = light ? (deg + SB - 1) / SB : 0;  // batches of this node
#pragma unroll
        for (int k = 0; k < SB; ++k) {
            const int v = __builtin_amdgcn_readlane(idx, min(k, max(deg - 1, 0)));
            rA[k] = ld4(a.d_pre + (int64_t)v * a.ld_dpre + lane * 4);
        }
        // 3. requests for the following nodes
        const int nidx = a.out_dst[edge_slot(b1, e1)];
        const int nty = TABLE ? a.etype[edge_slot(b1, e1)] : a.out_pos[edge_slot(b1, e1)];
        float4 nhu, neh4;
        if (CODED) {
            const uint32_t cu = a.ci.code[u1];
            nhu = coded4(a.ci, cu, lane * 4);
            neh4 = coded4(a.ci, cu, C);
        } else {
            nhu = ld4_nt(a.hs + (int64_t)u1 * a.ld_hs + lane * 4);
            neh4 = ld4(a.eh + (int64_t)u1 * a.ld_s);
        }
        const int b2 = a.pout.ptr[u2], e2 = a.pout.ptr[u2 + 1];
        const int u3 = node_at(i0 + 3 * stride);
        // 4. attention weights of edge `lane`, all heads, into the strip
        {
            const float x0 = (eh4.x + eev.x) + s01.x, x1 = (eh4.y + eev.y) + s01.z;
            const float x2 = (eh4.z + eev.z) + s23.x, x3 = (eh4.w + eev.w) + s23.z;
            float4 pw;
            pw.x = pack_attn(__expf(leaky(x0, a.slope) - s01.y), x0);
            pw.y = pack_attn(__expf(leaky(x1, a.slope) - s01.w), x1);
            pw.z = pack_attn(__expf(leaky(x2, a.slope) - s23.y), x2);
            pw.w = pack_attn(__expf(leaky(x3, a.slope) - s23.w), x3);
            __builtin_amdgcn_wave_barrier();  // the previous node's reads of the strip are done
            *reinterpret_cast<float4 *>(wl + lane * H) = pw;
            __builtin_amdgcn_wave_barrier();
        }
        // 5. rows, two batches in flight
        float4 acc = f4zero();
        auto consume = [&](const float4 (&r)[SB], int j) {
#pragma unroll
            for (int k = 0; k < SB; ++k) {
                const float pw = wl[min(j + k, 63) * H + h];
                const float tt = head_sum<LPH>(dot4(r[k], hu));
                if (j + k < deg) {
                    if (writer) st2_nt(a.pairs + ((int64_t)(b0 + j + k) * H + h) * 2, pw, tt);
                    fma4(acc, fabsf(pw), r[k]);
                }
            }
        };
        auto request = [&](float4 (&r)[SB], int j) {
#pragma unroll
            for (int k = 0; k < SB; ++k) {
                const int v = __builtin_amdgcn_readlane(idx, min(j + k, max(deg - 1, 0)));
                r[k] = ld4(a.d_pre + (int64_t)v * a.ld_dpre + lane * 4);
            }
        };
        for (int bi = 0; bi < nb; bi += 2) {
            if (bi + 1 < nb) request(rB, (bi + 1) * SB);
            consume(rA, bi * SB);
            if (bi + 1 < nb) {
                if (bi + 2 < nb) request(rA, (bi + 2) * SB);
                consume(rB, (bi + 1) * SB);
            }
        }
        if (light) st4_nt(a.d_y + (int64_t)u0 * a.ld_dy + lane * 4, acc);
        // 6. advance
        u0 = u1, b0 = b1, e0 = e1;
        u1 = u2, b1 = b2, e1 = e2;
        u2 = u3;
        idx = nidx, ty = nty, hu = nhu, eh4 = neh4;
        i0 += stride;
    }
}

template <int H, int LPH, bool TABLE, bool CODED>
__global__ __launch_bounds__(256) void k_bwd_src_heavy1(BwdArgs a) {
    constexpr int G = H * LPH;
    constexpr int NSG = 64 / G;
    constexpr int C = G * 4;
    __shared__ float tbl[TABLE ? MAX_E_DIM * H : 1];
    stage_table<H, TABLE>(a, tbl);
    const int lane = threadIdx.x & 63;
    const int c = xcd_chunk(threadIdx.x >> 6);
    if (c >= a.pout.n_chunks) return;
    const int u = a.pout.heavy_node[a.pout.chunk_node[c]];
    const int beg = a.pout.chunk_beg[c], end = a.pout.chunk_end[c];
    const int g = lane % G, sg = lane / G, h = g / LPH;
    const bool writer = (g % LPH) == 0;
    float4 hu;
    float ehu;
    if (CODED) {
        const uint32_t cu = a.ci.code[u];
        hu = coded4(a.ci, cu, g * 4);
        ehu = coded1(a.ci, cu, C + h);
    } else {
        hu = ld4(a.hs + (int64_t)u * a.ld_hs + g * 4);
        ehu = a.eh[(int64_t)u * a.ld_s + h];
    }
    float4 acc = f4zero();
    for (int j = beg + sg; j < end; j += 4 * NSG) {
        float4 r[4];
        float2 st[4];
        float ee[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int jj = min(j + k * NSG, end - 1);
            const int v = a.out_dst[jj];
            r[k] = ld4(a.d_pre + (int64_t)v * a.ld_dpre + g * 4);
            st[k] = *reinterpret_cast<const float2 *>(a.stat + ((int64_t)v * H + h) * 2);
            ee[k] = edge_term<H, TABLE>(a, tbl, jj, h);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float x = (ehu + ee[k]) + st[k].x;
            const float w = __expf(leaky(x, a.slope) - st[k].y);
            const float tt = head_sum<LPH>(dot4(r[k], hu));
            if (j + k * NSG < end) {
                if (writer) st2_nt(a.pairs + ((int64_t)(j + k * NSG) * H + h) * 2, pack_attn(w, x), tt);
                fma4(acc, w, r[k]);
            }
        }
    }
    acc = substream_sum4<G>(acc);
    if (sg == 0) st4(a.ws_acc + (int64_t)c * C + g * 4, acc);
}

template <int H, int LPH>
__global__ __launch_bounds__(256) void k_bwd_src_heavy2(BwdArgs a) {
    constexpr int G = H * LPH;
    constexpr int NSGB = 256 / G;
    constexpr int C = G * 4;
    __shared__ float4 part[256];
    const int slot = blockIdx.x;
    const int g = threadIdx.x % G, sg = threadIdx.x / G;
    const int c0 = a.pout.heavy_cptr[slot], c1 = a.pout.heavy_cptr[slot + 1];
    const int u = a.pout.heavy_node[slot];
    float4 acc = f4zero();
    for (int c = c0 + sg; c < c1; c += NSGB) add4(acc, ld4(a.ws_acc + (int64_t)c * C + g * 4));
    part[threadIdx.x] = acc;
    __syncthreads();
    if (sg == 0) {
        for (int k = 1; k < NSGB; ++k) add4(acc, part[k * G + g]);
        st4(a.d_y + (int64_t)u * a.ld_dy + g * 4, acc);
    }
}

// ---- per-edge [H] vectors ------------------------------------------------------------------------
template <int H> struct HeadVec {
    float v[H];
};
template <int H> __device__ __forceinline__ HeadVec<H> ldh(const float *p) {
    HeadVec<H> r;
    if (H == 4) {
        const float4 q = *reinterpret_cast<const float4 *>(p);
        r.v[0] = q.x, r.v[1 % H] = q.y, r.v[2 % H] = q.z, r.v[3 % H] = q.w;
    } else {
#pragma unroll
        for (int h = 0; h < H; ++h) r.v[h] = p[h];
    }
    return r;
}
template <int H> __device__ __forceinline__ void sth(float *p, const HeadVec<H> &r) {
    if (H == 4) {
        *reinterpret_cast<float4 *>(p) = make_float4(r.v[0], r.v[1 % H], r.v[2 % H], r.v[3 % H]);
    } else {
#pragma unroll
        for (int h = 0; h < H; ++h) p[h] = r.v[h];
    }
}

// the (a, t) pairs of one edge: ap = packed attention weights, tt = SDDMM terms
template <int H> struct PairVec {
    HeadVec<H> ap, tt;
};
template <int H> __device__ __forceinline__ PairVec<H> ld_pairs(const float *pairs, int64_t slot) {
    PairVec<H> r;
    const float *p = pairs + slot * (2 * H);
    if (H == 4) {
        const float4 q0 = *reinterpret_cast<const float4 *>(p), q1 = *reinterpret_cast<const float4 *>(p + 4);
        r.ap.v[0] = q0.x, r.tt.v[0] = q0.y, r.ap.v[1 % H] = q0.z, r.tt.v[1 % H] = q0.w;
        r.ap.v[2 % H] = q1.x, r.tt.v[2 % H] = q1.y, r.ap.v[3 % H] = q1.z, r.tt.v[3 % H] = q1.w;
    } else {
#pragma unroll
        for (int h = 0; h < H; ++h) {
            const float2 q = *reinterpret_cast<const float2 *>(p + 2 * h);
            r.ap.v[h] = q.x, r.tt.v[h] = q.y;
        }
    }
    return r;
}

// ---- destination pass: delta_v and d_et[v] from the pairs of the in-edges ------------------------
// With lk_e = leaky'_e (1 or the slope) the segment sums
//     S_at = sum a_e t_e,   S_atl = sum a_e t_e lk_e,   S_al = sum a_e lk_e
// give delta_v = S_at and d_et[v] = sum a_e (t_e - delta_v) lk_e = S_atl - delta_v S_al: one pass over
// the gathered pairs, no second visit once delta_v is known.
template <int H> struct DstSums {
    float at[H], atl[H], al[H];
};
template <int H> __device__ __forceinline__ void dst_zero(DstSums<H> &s) {
#pragma unroll
    for (int h = 0; h < H; ++h) s.at[h] = s.atl[h] = s.al[h] = 0.f;
}
template <int H> __device__ __forceinline__ void dst_add(DstSums<H> &s, const PairVec<H> &p, float slope, bool on) {
#pragma unroll
    for (int h = 0; h < H; ++h) {
        const float w = on ? fabsf(p.ap.v[h]) : 0.f;
        const float wl = attn_neg(p.ap.v[h]) ? w * slope : w;
        s.at[h] = fmaf(w, p.tt.v[h], s.at[h]);
        s.atl[h] = fmaf(wl, p.tt.v[h], s.atl[h]);
        s.al[h] += wl;
    }
}

// W lanes per destination node with <= light_max in-edges; the lanes take the in-edges round robin
// (one gathered pair per lane and round: the loads of a segment are in flight together) and combine
// with a fixed DPP butterfly.  Nodes are walked by segment-length class.
constexpr int DST_W = 8;
template <int H>
__global__ __launch_bounds__(256) void k_bwd_dst_light(BwdArgs a, int C) {
    constexpr int W = DST_W, NPB = 256 / W;
    const int q = threadIdx.x % W, slot = threadIdx.x / W;
    const Walk wk = xcd_walk(a.pin, a.N, NPB, slot);
    for (int vi = wk.pos; vi < wk.end; vi += wk.stride) {
        const int v = a.pin.order ? a.pin.order[vi] : vi;
        const int b = a.pin.ptr[v], e = a.pin.ptr[v + 1];
        if (e - b > a.pin.light_max) continue;
        DstSums<H> s;
        dst_zero<H>(s);
        if (e - b > 0 && dst_on(a, v)) {
            int i = b + q;
            for (; i + W < e; i += 2 * W) {  // two rounds per trip: four independent gathers in flight
                const PairVec<H> p0 = ld_pairs<H>(a.pairs, a.in_pos[i]), p1 = ld_pairs<H>(a.pairs, a.in_pos[i + W]);
                dst_add<H>(s, p0, a.slope, true);
                dst_add<H>(s, p1, a.slope, true);
            }
            const PairVec<H> p = ld_pairs<H>(a.pairs, a.in_pos[min(i, e - 1)]);
            dst_add<H>(s, p, a.slope, i < e);
#pragma unroll
            for (int h = 0; h < H; ++h) {
                s.at[h] = head_sum<W>(s.at[h]);
                s.atl[h] = head_sum<W>(s.atl[h]);
                s.al[h] = head_sum<W>(s.al[h]);
            }
        }
        if (q == 0) {
            HeadVec<H> delta, det;
#pragma unroll
            for (int h = 0; h < H; ++h) delta.v[h] = s.at[h], det.v[h] = s.atl[h] - s.at[h] * s.al[h];
            sth<H>(a.delta + (int64_t)v * H, delta);
            sth<H>(a.d_y + (int64_t)v * a.ld_dy + C + H, det);
        }
    }
}

// heavy destinations, step 1: wave per chunk, lane per edge: the chunk's three sums
template <int H>
__global__ __launch_bounds__(256) void k_bwd_dst_heavy1(BwdArgs a) {
    const int lane = threadIdx.x & 63;
    const int c = xcd_chunk(threadIdx.x >> 6);
    if (c >= a.pin.n_chunks) return;
    const int beg = a.pin.chunk_beg[c];
    const int end = dst_on(a, a.pin.heavy_node[a.pin.chunk_node[c]]) ? a.pin.chunk_end[c] : beg;
    DstSums<H> s;
    dst_zero<H>(s);
    for (int i = beg + lane; i < end; i += 64) {
        const PairVec<H> p = ld_pairs<H>(a.pairs, a.in_pos[i]);
        dst_add<H>(s, p, a.slope, true);
    }
#pragma unroll
    for (int h = 0; h < H; ++h) {
        const float x = wave_sum(s.at[h]), y = wave_sum(s.atl[h]), z = wave_sum(s.al[h]);
        if (lane == 0) {
            a.ws_delta[(int64_t)c * H + h] = x;
            a.ws_det[((int64_t)c * H + h) * 2] = y;
            a.ws_det[((int64_t)c * H + h) * 2 + 1] = z;
        }
    }
}

// step 2: wave per heavy node: the chunk sums in chunk order -> delta_v, d_et[v]
template <int H>
__global__ __launch_bounds__(256) void k_bwd_dst_heavy2(BwdArgs a, int C) {
    const int lane = threadIdx.x & 63;
    const int slot = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (slot >= a.pin.n_heavy) return;
    const int c0 = a.pin.heavy_cptr[slot], c1 = a.pin.heavy_cptr[slot + 1];
    const int v = a.pin.heavy_node[slot];
#pragma unroll
    for (int h = 0; h < H; ++h) {
        float x = 0.f, y = 0.f, z = 0.f;
        for (int k = c0 + lane; k < c1; k += 64) {
            x += a.ws_delta[(int64_t)k * H + h];
            y += a.ws_det[((int64_t)k * H + h) * 2];
            z += a.ws_det[((int64_t)k * H + h) * 2 + 1];
        }
        x = wave_sum(x), y = wave_sum(y), z = wave_sum(z);
        if (lane == 0) {
            a.delta[(int64_t)v * H + h] = x;
            a.d_y[(int64_t)v * a.ld_dy + C + H + h] = y - x * z;
        }
    }
}

// ---- source pass 2: d_logit of every out-edge from the streamed (a, t) pairs and the destination's
// delta; d_eh[u] = sum over out-edges of d_logit ------------------------------------------------
template <int H>
__device__ __forceinline__ HeadVec<H> edge_dlogit(const BwdArgs &a, int j) {
    const PairVec<H> p = ld_pairs<H>(a.pairs, j);
    const HeadVec<H> dv = ldh<H>(a.delta + (int64_t)a.out_dst[j] * H);
    HeadVec<H> d;
#pragma unroll
    for (int h = 0; h < H; ++h) {
        float x = fabsf(p.ap.v[h]) * (p.tt.v[h] - dv.v[h]);
        if (attn_neg(p.ap.v[h])) x *= a.slope;
        d.v[h] = x;
    }
    sth<H>(a.dl_csr + (int64_t)j * H, d);
    if (a.dl) sth<H>(a.dl + (int64_t)a.out_pos[j] * H, d);
    return d;
}

// W lanes per source node with <= light_max out-edges, two edges per lane and round: their loads are
// issued before the first d_logit is stored
constexpr int DEH_W = 4;
template <int H>
__global__ __launch_bounds__(256) void k_bwd_deh_light(BwdArgs a, int C) {
    constexpr int W = DEH_W, NPB = 256 / W;
    const int q = threadIdx.x % W, slot = threadIdx.x / W;
    const Walk wk = xcd_walk(a.pout, a.N, NPB, slot);  // the ranges cut the node ids at the same positions
    for (int ui = wk.pos; ui < wk.end; ui += wk.stride) {
        const int u = ui;  // natural order: neighbouring groups stream neighbouring segments
        const int b = a.pout.ptr[u], e = a.pout.ptr[u + 1];
        if (e - b > a.pout.light_max) continue;
        HeadVec<H> s;
#pragma unroll
        for (int h = 0; h < H; ++h) s.v[h] = 0.f;
        for (int j = b + q; j < e; j += 2 * W) {
            const int j1 = min(j + W, e - 1);
            const PairVec<H> p0 = ld_pairs<H>(a.pairs, j), p1 = ld_pairs<H>(a.pairs, j1);
            const HeadVec<H> dv0 = ldh<H>(a.delta + (int64_t)a.out_dst[j] * H);
            const HeadVec<H> dv1 = ldh<H>(a.delta + (int64_t)a.out_dst[j1] * H);
            const int pos0 = a.dl ? a.out_pos[j] : 0, pos1 = a.dl ? a.out_pos[j1] : 0;
            HeadVec<H> d0, d1;
#pragma unroll
            for (int h = 0; h < H; ++h) {
                float x0 = fabsf(p0.ap.v[h]) * (p0.tt.v[h] - dv0.v[h]);
                float x1 = fabsf(p1.ap.v[h]) * (p1.tt.v[h] - dv1.v[h]);
                if (attn_neg(p0.ap.v[h])) x0 *= a.slope;
                if (attn_neg(p1.ap.v[h])) x1 *= a.slope;
                d0.v[h] = x0, d1.v[h] = x1;
                s.v[h] += x0 + (j + W < e ? x1 : 0.f);
            }
            sth<H>(a.dl_csr + (int64_t)j * H, d0);
            if (a.dl) sth<H>(a.dl + (int64_t)pos0 * H, d0);
            if (j + W < e) {
                sth<H>(a.dl_csr + (int64_t)(j + W) * H, d1);
                if (a.dl) sth<H>(a.dl + (int64_t)pos1 * H, d1);
            }
        }
#pragma unroll
        for (int h = 0; h < H; ++h) s.v[h] = head_sum<W>(s.v[h]);
        if (q == 0) sth<H>(a.d_y + (int64_t)u * a.ld_dy + C, s);
    }
}

template <int H>
__global__ __launch_bounds__(256) void k_bwd_deh_heavy1(BwdArgs a) {
    const int lane = threadIdx.x & 63;
    const int c = xcd_chunk(threadIdx.x >> 6);
    if (c >= a.pout.n_chunks) return;
    const int beg = a.pout.chunk_beg[c], end = a.pout.chunk_end[c];
    float s[H];
#pragma unroll
    for (int h = 0; h < H; ++h) s[h] = 0.f;
    for (int j = beg + lane; j < end; j += 64) {
        const HeadVec<H> d = edge_dlogit<H>(a, j);
#pragma unroll
        for (int h = 0; h < H; ++h) s[h] += d.v[h];
    }
#pragma unroll
    for (int h = 0; h < H; ++h) {
        const float t = wave_sum(s[h]);
        if (lane == 0) a.ws_deh[(int64_t)c * H + h] = t;
    }
}

template <int H>
__global__ __launch_bounds__(256) void k_bwd_deh_heavy2(BwdArgs a, int C) {
    const int lane = threadIdx.x & 63;
    const int slot = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (slot >= a.pout.n_heavy) return;
    const int c0 = a.pout.heavy_cptr[slot], c1 = a.pout.heavy_cptr[slot + 1];
    const int u = a.pout.heavy_node[slot];
#pragma unroll
    for (int h = 0; h < H; ++h) {
        float t = 0.f;
        for (int k = c0 + lane; k < c1; k += 64) t += a.ws_deh[(int64_t)k * H + h];
        t = wave_sum(t);
        if (lane == 0) a.d_y[(int64_t)u * a.ld_dy + C + h] = t;
    }
}

// ---- d_ee_table[t,h] = sum over edges of type t of d_logit[e,h] ---------------------------
// (edges are walked in CSR order: d_logit and the edge types both come in that order.)
// Every thread owns a private column of an LDS table [e_dim*H][128] (bank = thread: no conflicts,
// no atomics) and adds its edges there in edge order; the 128 columns of a row are then summed
// in a fixed tree and a last launch adds the block rows.
constexpr int TBL_THREADS = 128;

template <int H>
__global__ __launch_bounds__(TBL_THREADS) void k_bwd_table_partial(BwdArgs a) {
    extern __shared__ float tbl[];  // [e_dim*H][TBL_THREADS]
    const int tid = threadIdx.x;
    const int rows = a.e_dim * H;
    for (int r = 0; r < rows; ++r) tbl[r * TBL_THREADS + tid] = 0.f;
    const int stride = gridDim.x * TBL_THREADS;
    int i = blockIdx.x * TBL_THREADS + tid;
    for (; i + stride < a.E; i += 2 * stride) {  // two edges per round: their loads are in flight together
        const int t0 = a.etype[i], t1 = a.etype[i + stride];
        const HeadVec<H> d0 = ldh<H>(a.dl_csr + (int64_t)i * H), d1 = ldh<H>(a.dl_csr + (int64_t)(i + stride) * H);
#pragma unroll
        for (int h = 0; h < H; ++h) tbl[(t0 * H + h) * TBL_THREADS + tid] += d0.v[h];
#pragma unroll
        for (int h = 0; h < H; ++h) tbl[(t1 * H + h) * TBL_THREADS + tid] += d1.v[h];
    }
    if (i < a.E) {
        const int t0 = a.etype[i];
        const HeadVec<H> d0 = ldh<H>(a.dl_csr + (int64_t)i * H);
#pragma unroll
        for (int h = 0; h < H; ++h) tbl[(t0 * H + h) * TBL_THREADS + tid] += d0.v[h];
    }
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6;
    for (int r = wave; r < rows; r += TBL_THREADS / 64) {
        const float v = wave_sum(tbl[r * TBL_THREADS + lane] + tbl[r * TBL_THREADS + 64 + lane]);
        if (lane == 0) a.ws_table[(int64_t)blockIdx.x * rows + r] = v;
    }
}

// one block per table entry: 256 threads add the block rows round robin, then a fixed tree
__global__ __launch_bounds__(256) void k_bwd_table_final(BwdArgs a, int H) {
    __shared__ float red[256];
    const int n = a.e_dim * H;
    const int o = blockIdx.x;
    float t = 0.f;
    for (int k = threadIdx.x; k < a.grid_table; k += 256) t += a.ws_table[(int64_t)k * n + o];
    red[threadIdx.x] = t;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) a.d_table[o] = red[0];
}

template <int H, int LPH>
int launch_table(BwdArgs &a, hipStream_t stream) {
    if (!a.etype || !a.d_table) return MAGIC_OK;
    if (a.E == 0) {
        if (hipMemsetAsync(a.d_table, 0, sizeof(float) * a.e_dim * H, stream) != hipSuccess) return MAGIC_ERR_RUNTIME;
        return MAGIC_OK;
    }
    const size_t sh = (size_t)a.e_dim * H * TBL_THREADS * sizeof(float);
    if (magic::ensure_dynamic_lds((const void *)k_bwd_table_partial<H>, sh) != MAGIC_OK) return MAGIC_ERR_RUNTIME;
    MAGIC_LAUNCH(prof_on() ? prof_name("gat_bwd_table_partial", H, LPH) : "gat_bwd_table_partial", (k_bwd_table_partial<H>), a.grid_table, TBL_THREADS, sh, stream, a);
    MAGIC_LAUNCH("gat_bwd_table_final", k_bwd_table_final, a.e_dim * H, 256, 0, stream, a, H);
    return MAGIC_OK;
}

template <int H, int LPH, bool TABLE, bool CODED = false>
int launch_bwd(BwdArgs &a, hipStream_t stream) {
    constexpr int C = H * LPH * 4;
    constexpr int NPB = 256 / (H * LPH);
    if (a.prelu_w || a.d_bias || a.d_pre != a.d_out) {
        MAGIC_LAUNCH(prof_on() ? prof_name("gat_bwd_node_adjoint", H, LPH) : "gat_bwd_node_adjoint", (k_bwd_node_adjoint<H, LPH>), a.grid_light, 256, 0, stream, a);
        if (a.d_bias || a.d_prelu)
            MAGIC_LAUNCH("gat_bwd_reduce_bias", k_bwd_reduce_bias, C + 1, 256, 0, stream, a, C);
    }
    if (a.E == 0) {  // no edges: d_hs = d_eh = d_et = 0 and an all-zero edge-type table
        if (hipMemset2DAsync(a.d_y, sizeof(float) * a.ld_dy, 0, sizeof(float) * (C + 2 * H), a.N, stream) != hipSuccess)
            return MAGIC_ERR_RUNTIME;
        int st0 = launch_table<H, LPH>(a, stream);
        return st0 != MAGIC_OK ? st0 : launch_status();
    }
    if (H == 4 && LPH == 16)  // 1 KB rows: one wavefront per node
        MAGIC_LAUNCH(prof_on() ? prof_name(CODED ? "gat_bwd_src_light_coded" : "gat_bwd_src_light", H, LPH) : "gat_bwd_src_light", (k_bwd_src_wave<TABLE, CODED>), stride_grid(a.N, 4, 256 * 5), 256, 0, stream, a);  // resident workgroups only: the pipeline fill is paid once per wavefront
    else
        MAGIC_LAUNCH(prof_on() ? prof_name(CODED ? "gat_bwd_src_light_coded" : "gat_bwd_src_light", H, LPH) : "gat_bwd_src_light", (k_bwd_src_light<H, LPH, TABLE, CODED>), stride_grid(a.N, NPB, 256 * 16), 256, 0, stream, a);
    if (a.pout.n_chunks > 0) {
        const int cgrid = chunk_grid(a.pout.n_chunks);
        MAGIC_LAUNCH(prof_on() ? prof_name("gat_bwd_src_heavy1", H, LPH) : "gat_bwd_src_heavy1", (k_bwd_src_heavy1<H, LPH, TABLE, CODED>), cgrid, 256, 0, stream, a);
        MAGIC_LAUNCH("gat_bwd_src_heavy2", (k_bwd_src_heavy2<H, LPH>), a.pout.n_heavy, 256, 0, stream, a);
    }
    MAGIC_LAUNCH(prof_on() ? prof_name("gat_bwd_dst_light", H, LPH) : "gat_bwd_dst_light", (k_bwd_dst_light<H>), stride_grid(a.N, 256 / DST_W, 256 * 32), 256, 0, stream, a, C);
    if (a.pin.n_chunks > 0) {
        const int cgrid = chunk_grid(a.pin.n_chunks);
        MAGIC_LAUNCH("gat_bwd_dst_heavy1", (k_bwd_dst_heavy1<H>), cgrid, 256, 0, stream, a);
        MAGIC_LAUNCH("gat_bwd_dst_heavy2", (k_bwd_dst_heavy2<H>), (a.pin.n_heavy + 3) / 4, 256, 0, stream, a, C);
    }
    MAGIC_LAUNCH(prof_on() ? prof_name("gat_bwd_deh_light", H, LPH) : "gat_bwd_deh_light", (k_bwd_deh_light<H>), stride_grid(a.N, 256 / DEH_W, 256 * 32), 256, 0, stream, a, C);
    if (a.pout.n_chunks > 0) {
        MAGIC_LAUNCH("gat_bwd_deh_heavy1", (k_bwd_deh_heavy1<H>), chunk_grid(a.pout.n_chunks), 256, 0, stream, a);
        MAGIC_LAUNCH("gat_bwd_deh_heavy2", (k_bwd_deh_heavy2<H>), (a.pout.n_heavy + 3) / 4, 256, 0, stream, a, C);
    }
    int st = launch_table<H, LPH>(a, stream);
    if (st != MAGIC_OK) return st;
    return launch_status();
}
static int light_grid(int N, int H, int D) {
    int npb = 256 / (H * D / 4);
    return stride_grid(N, npb, 2048);
}
static int table_grid(int E) { return stride_grid(E, 128 * 8, 1024); }

}  // namespace magic

extern "C" {

size_t magic_gat_bwd_workspace_bytes(int32_t N, int32_t E, int32_t n_chunks_in, int32_t n_chunks_out, int32_t H,
                                     int32_t D, int32_t e_dim) {
    using namespace magic;
    if (N <= 0 || H <= 0 || D <= 0) return 256;
    const size_t C = (size_t)H * D, f = sizeof(float);
    const size_t En = E > 0 ? E : 1;
    size_t gl = light_grid(N, H, D), gt = table_grid(E);
    size_t b = align_up(gl * C * f) + align_up(gl * f);
    b += align_up((size_t)(n_chunks_in > 0 ? n_chunks_in : 1) * H * f) +
         align_up((size_t)(n_chunks_in > 0 ? n_chunks_in : 1) * H * 2 * f);
    b += align_up((size_t)(n_chunks_out > 0 ? n_chunks_out : 1) * C * f) +
         align_up((size_t)(n_chunks_out > 0 ? n_chunks_out : 1) * H * f);
    b += align_up(gt * (size_t)(e_dim > 0 ? e_dim : 1) * H * f);
    b += align_up(En * H * 2 * f) + align_up(En * H * f) + align_up((size_t)N * H * f);
    b += align_up(En * sizeof(int32_t));  // rewritten destination ids (magic_gat_bwd_active)
    return b + 256;
}

// both entry points: `ci` != NULL selects the coded-input source sweeps (hs and eh are then not read)
static int gat_bwd_run(int32_t N, int32_t E, int32_t H, int32_t D, const magic_plan *in_plan, const int32_t *in_pos,
                       const magic_plan *out_plan, const int32_t *out_dst, const int32_t *out_pos, const float *hs,
                       int64_t ld_hs, const float *eh, int64_t ld_score, const float *stat, const float *ee_table,
                       const int32_t *etype_csr, const float *ee_csc, int32_t e_dim, float slope, const float *pre,
                       const float *act_out, int64_t ld_act, const float *prelu_w, const float *d_out, int64_t ld_dout,
                       float *d_pre, int64_t ld_dpre, float *d_y, int64_t ld_dy, float *d_ee_table, float *d_logit,
                       float *d_bias, float *d_prelu_w, void *workspace, size_t workspace_bytes, void *stream_,
                       const magic_coded_input *ci, const uint8_t *dst_active) {
    using namespace magic;
    if (N < 0 || E < 0 || H <= 0 || D <= 0) return MAGIC_ERR_INVALID_ARG;
    if (!shape_supported(H, D)) return MAGIC_ERR_UNSUPPORTED;
    if (N == 0) return MAGIC_OK;
    const int C = H * D;
    if (ci) {
        if (!ci->code || !ci->table || ci->n_dim <= 0 || ci->n_dim > MAGIC_CODED_MAX_DIM) return MAGIC_ERR_INVALID_ARG;
        if (ci->ld_table < 2 * C + 2 * H || (ci->ld_table & 3) || ((uintptr_t)ci->table & 15)) return MAGIC_ERR_INVALID_ARG;
        if (H != 4 || !ee_table) return MAGIC_ERR_UNSUPPORTED;
        hs = eh = d_out;  // (placeholders: never read)
        ld_hs = ld_dout;
        ld_score = H;
    }
    if (!plan_valid(in_plan) || !plan_valid(out_plan) || !hs || !d_out || !d_pre || !d_y) return MAGIC_ERR_INVALID_ARG;
    if (out_plan->light_max > 64) return MAGIC_ERR_UNSUPPORTED;  // lane = edge in the wavefront-per-node sweep
    const bool table = ee_table != nullptr;
    if (E > 0 && (!in_pos || !out_dst || !out_pos || !eh || !stat || ld_score < H)) return MAGIC_ERR_INVALID_ARG;
    if (E > 0 && table && !etype_csr) return MAGIC_ERR_INVALID_ARG;
    if (E > 0 && !table && !ee_csc) return MAGIC_ERR_INVALID_ARG;
    if (table && (e_dim <= 0 || e_dim > MAX_E_DIM)) return MAGIC_ERR_UNSUPPORTED;
    if (prelu_w && (!pre || !act_out || ld_act < C || (ld_act & 3))) return MAGIC_ERR_INVALID_ARG;
    if (prelu_w && d_pre == d_out) return MAGIC_ERR_INVALID_ARG;  // in-place only without an activation
    if (ld_hs < C || ld_dout < C || ld_dy < C + 2 * H || ld_dpre < C) return MAGIC_ERR_INVALID_ARG;
    if ((ld_hs | ld_dout | ld_dy | ld_dpre) & 3) return MAGIC_ERR_INVALID_ARG;
    BwdArgs a;
    a.N = N;
    a.E = E;
    a.pin = plan_dev(in_plan);
    a.pout = plan_dev(out_plan);
    a.out_dst = out_dst;
    a.out_pos = out_pos;
    a.in_pos = in_pos;
    a.hs = hs;
    a.ld_hs = ld_hs;
    a.eh = eh;
    a.ld_s = ld_score;
    a.stat = stat;
    a.ee_table = ee_table;
    a.etype = table ? etype_csr : nullptr;
    a.ee = ee_csc;
    a.e_dim = e_dim;
    a.slope = slope;
    a.pre = pre;
    a.act_out = act_out;
    a.ld_act = ld_act;
    a.prelu_w = prelu_w;
    a.d_out = d_out;
    a.ld_dout = ld_dout;
    a.d_pre = d_pre;
    a.ld_dpre = ld_dpre;
    a.d_y = d_y;
    a.ld_dy = ld_dy;
    a.d_table = table ? d_ee_table : nullptr;
    a.dl = d_logit;
    a.d_bias = d_bias;
    a.d_prelu = d_prelu_w;
    a.grid_light = light_grid(N, H, D);
    a.grid_table = table_grid(E);
    const size_t En = E > 0 ? E : 1;
    Arena ar(workspace, workspace_bytes);
    a.ws_bias = ar.take<float>((size_t)a.grid_light * C);
    a.ws_prelu = ar.take<float>((size_t)a.grid_light);
    a.ws_delta = ar.take<float>((size_t)(a.pin.n_chunks > 0 ? a.pin.n_chunks : 1) * H);
    a.ws_det = ar.take<float>((size_t)(a.pin.n_chunks > 0 ? a.pin.n_chunks : 1) * H * 2);
    a.ws_acc = ar.take<float>((size_t)(a.pout.n_chunks > 0 ? a.pout.n_chunks : 1) * C);
    a.ws_deh = ar.take<float>((size_t)(a.pout.n_chunks > 0 ? a.pout.n_chunks : 1) * H);
    a.ws_table = ar.take<float>((size_t)a.grid_table * (e_dim > 0 ? e_dim : 1) * H);
    a.pairs = ar.take<float>(En * H * 2);
    a.dl_csr = ar.take<float>(En * H);
    a.delta = ar.take<float>((size_t)N * H);
    int32_t *dst_eff = dst_active ? ar.take<int32_t>(En) : nullptr;
    if (!ar.ok()) return MAGIC_ERR_WORKSPACE;
    hipStream_t stream = (hipStream_t)stream_;
    a.active = dst_active;
    if (dst_active && d_logit) return MAGIC_ERR_UNSUPPORTED;
    if (dst_active && E > 0) {
        if ((uintptr_t)out_dst & 15) return MAGIC_ERR_INVALID_ARG;
        MAGIC_LAUNCH("gat_bwd_active_dst", k_active_dst, stride_grid((E + 3) / 4, 256, 4096), 256, 0, stream, (int)N, (int)E,
                     out_dst, dst_active, dst_eff);
        a.out_dst = dst_eff;
    }
    const int lph = D / 4;
    a.ci = CodedIn{nullptr, nullptr, 0, C};
    if (ci) {
        a.ci = CodedIn{ci->code, ci->table, (int)ci->ld_table, C};
        switch (lph) {
            case 2: return launch_bwd<4, 2, true, true>(a, stream);
            case 4: return launch_bwd<4, 4, true, true>(a, stream);
            case 8: return launch_bwd<4, 8, true, true>(a, stream);
            case 16: return launch_bwd<4, 16, true, true>(a, stream);
        }
        return MAGIC_ERR_UNSUPPORTED;
    }
#define X(h, l)                                                \
    if (H == h && lph == l)                                    \
        return table ? launch_bwd<h, l, true>(a, stream) : launch_bwd<h, l, false>(a, stream);
    MAGIC_GAT_SHAPES(X)
#undef X
    return MAGIC_ERR_UNSUPPORTED;
}

int magic_gat_bwd(int32_t N, int32_t E, int32_t H, int32_t D, const magic_plan *in_plan, const int32_t *in_pos,
                  const magic_plan *out_plan, const int32_t *out_dst, const int32_t *out_pos, const float *hs,
                  int64_t ld_hs, const float *eh, int64_t ld_score, const float *stat, const float *ee_table,
                  const int32_t *etype_csr, const float *ee_csc, int32_t e_dim, float slope, const float *pre,
                  const float *act_out, int64_t ld_act, const float *prelu_w, const float *d_out, int64_t ld_dout,
                  float *d_pre, int64_t ld_dpre, float *d_y, int64_t ld_dy, float *d_ee_table, float *d_logit,
                  float *d_bias, float *d_prelu_w, void *workspace, size_t workspace_bytes, void *stream_) {
    return gat_bwd_run(N, E, H, D, in_plan, in_pos, out_plan, out_dst, out_pos, hs, ld_hs, eh, ld_score, stat, ee_table,
                       etype_csr, ee_csc, e_dim, slope, pre, act_out, ld_act, prelu_w, d_out, ld_dout, d_pre, ld_dpre,
                       d_y, ld_dy, d_ee_table, d_logit, d_bias, d_prelu_w, workspace, workspace_bytes, stream_, nullptr,
                       nullptr);
}

int magic_gat_bwd_active(int32_t N, int32_t E, int32_t H, int32_t D, const magic_plan *in_plan, const int32_t *in_pos,
                         const magic_plan *out_plan, const int32_t *out_dst, const int32_t *out_pos, const float *hs,
                         int64_t ld_hs, const float *eh, int64_t ld_score, const float *stat, const float *ee_table,
                         const int32_t *etype_csr, const float *ee_csc, int32_t e_dim, float slope, const float *pre,
                         const float *act_out, int64_t ld_act, const float *prelu_w, const float *d_out,
                         int64_t ld_dout, float *d_pre, int64_t ld_dpre, float *d_y, int64_t ld_dy, float *d_ee_table,
                         float *d_bias, float *d_prelu_w, const uint8_t *dst_active, void *workspace,
                         size_t workspace_bytes, void *stream_) {
    return gat_bwd_run(N, E, H, D, in_plan, in_pos, out_plan, out_dst, out_pos, hs, ld_hs, eh, ld_score, stat, ee_table,
                       etype_csr, ee_csc, e_dim, slope, pre, act_out, ld_act, prelu_w, d_out, ld_dout, d_pre, ld_dpre,
                       d_y, ld_dy, d_ee_table, nullptr, d_bias, d_prelu_w, workspace, workspace_bytes, stream_, nullptr,
                       dst_active);
}

int magic_gat_bwd_coded(int32_t N, int32_t E, int32_t H, int32_t D, const magic_plan *in_plan, const int32_t *in_pos,
                        const magic_plan *out_plan, const int32_t *out_dst, const int32_t *out_pos,
                        const magic_coded_input *input, const float *stat, const float *ee_table,
                        const int32_t *etype_csr, int32_t e_dim, float slope, const float *pre, const float *act_out,
                        int64_t ld_act, const float *prelu_w, const float *d_out, int64_t ld_dout, float *d_pre,
                        int64_t ld_dpre, float *d_y, int64_t ld_dy, float *d_ee_table, float *d_bias, float *d_prelu_w,
                        void *workspace, size_t workspace_bytes, void *stream_) {
    if (!input) return MAGIC_ERR_INVALID_ARG;
    return gat_bwd_run(N, E, H, D, in_plan, in_pos, out_plan, out_dst, out_pos, nullptr, 0, nullptr, 0, stat, ee_table,
                       etype_csr, nullptr, e_dim, slope, pre, act_out, ld_act, prelu_w, d_out, ld_dout, d_pre, ld_dpre,
                       d_y, ld_dy, d_ee_table, nullptr, d_bias, d_prelu_w, workspace, workspace_bytes, stream_, input,
                       nullptr);
}

}  // extern "C"
