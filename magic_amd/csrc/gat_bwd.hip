// GAT layer, edge phase, backward.  Segmented sweeps, no atomics, one gather of node rows:
//   node adjoint      PReLU / bias adjoints of the incoming gradient (skipped when the caller
//                     already holds the pre-activation gradient);
//   source sweep      (CSR) per edge u->v gathers d_pre[v] once and uses it twice:
//                     d_hs[u] += a_e d_pre[v]   and   t_e = <hs[u], d_pre[v]> per head
//                     (hs[u] is the sweep's own row), t_e stored at the edge's CSC position;
//   destination pass  (CSC, streams [E,H] arrays only) softmax backward
//                     d_logit_e = a_e (t_e - sum_seg a t) * leaky', d_et[v] = sum_seg d_logit;
//   source pass 2     d_eh[u] = sum over out-edges of d_logit (16-byte gathers);
//   then a fixed-order reduction of d_logit per edge type for the edge-score table.
#include "gat_common.h"

namespace magic {

struct BwdArgs {
    int N, E;
    PlanDev pin;   // CSC plan
    PlanDev pout;  // CSR plan
    const int32_t *out_dst;
    const int32_t *out_pos;
    const float *hs;
    int64_t ld_hs;
    const float *attn;  // [E,H] packed (sign bit = logit not positive), CSC order
    const int32_t *etype;
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
    float *dl;       // [E,H] d_logit, CSC order
    float *d_bias;   // [C]
    float *d_prelu;  // [1]
    // workspace
    float *t;         // [E,H] <hs[src], d_pre[dst]> per head, CSC order
    float *ws_bias;   // [grid_light, C]
    float *ws_prelu;  // [grid_light]
    int grid_light;
    float *ws_delta;  // [n_chunks_in, H]
    float *ws_det;    // [n_chunks_in, H]
    float *ws_acc;    // [n_chunks_out, C]
    float *ws_deh;    // [n_chunks_out, H]
    float *ws_table;  // [grid_table, e_dim*H]
    int grid_table;
};

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
// One lane group per source node u with <= light_max out-edges.  Per edge: the destination's
// gradient row is gathered once; a_e comes from the edge's CSC slot, t_e goes back to it.
template <int H, int LPH>
__global__ __launch_bounds__(256) void k_bwd_src_light(BwdArgs a) {
    constexpr int G = H * LPH;
    constexpr int NPB = 256 / G;
    constexpr int SB = 4;  // edges per batch (8 measured the same: the sweep is bandwidth-bound)
    const int g = threadIdx.x % G;
    const int h = g / LPH;
    const bool writer = (g % LPH) == 0;
    const int slot = threadIdx.x / G;
    for (int ui = blockIdx.x * NPB + slot; ui < a.N; ui += gridDim.x * NPB) {
        const int u = (G >= 8 && a.pout.order) ? a.pout.order[ui] : ui;
        const int b = a.pout.ptr[u], e = a.pout.ptr[u + 1];
        if (e - b > a.pout.light_max) continue;
        const float4 hu = ld4_nt(a.hs + (int64_t)u * a.ld_hs + g * 4);
        float4 acc = f4zero();
        for (int j = b; j < e; j += SB) {  // batches with clamped indices: loads stay unconditional
            int pos[SB];
            float4 r[SB];
            float w[SB];
#pragma unroll
            for (int k = 0; k < SB; ++k) {
                const int jj = min(j + k, e - 1);
                pos[k] = a.out_pos[jj];
                r[k] = ld4(a.d_pre + (int64_t)a.out_dst[jj] * a.ld_dpre + g * 4);
            }
#pragma unroll
            for (int k = 0; k < SB; ++k) w[k] = fabsf(a.attn[(int64_t)pos[k] * H + h]);
#pragma unroll
            for (int k = 0; k < SB; ++k) {
                const float tt = head_sum<LPH>(dot4(r[k], hu));
                if (j + k < e) {
                    if (writer) a.t[(int64_t)pos[k] * H + h] = tt;
                    fma4(acc, w[k], r[k]);
                }
            }
        }
        st4_nt(a.d_y + (int64_t)u * a.ld_dy + g * 4, acc);
    }
}

template <int H, int LPH>
__global__ __launch_bounds__(256) void k_bwd_src_heavy1(BwdArgs a) {
    constexpr int G = H * LPH;
    constexpr int NSG = 64 / G;
    constexpr int C = G * 4;
    const int lane = threadIdx.x & 63;
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= a.pout.n_chunks) return;
    const int u = a.pout.heavy_node[a.pout.chunk_node[c]];
    const int beg = a.pout.chunk_beg[c], end = a.pout.chunk_end[c];
    const int g = lane % G, sg = lane / G, h = g / LPH;
    const bool writer = (g % LPH) == 0;
    const float4 hu = ld4(a.hs + (int64_t)u * a.ld_hs + g * 4);
    float4 acc = f4zero();
    for (int j = beg + sg; j < end; j += 4 * NSG) {
        int pos[4];
        float4 r[4];
        float w[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int jj = min(j + k * NSG, end - 1);
            pos[k] = a.out_pos[jj];
            r[k] = ld4(a.d_pre + (int64_t)a.out_dst[jj] * a.ld_dpre + g * 4);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) w[k] = fabsf(a.attn[(int64_t)pos[k] * H + h]);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float tt = head_sum<LPH>(dot4(r[k], hu));
            if (j + k * NSG < end) {
                if (writer) a.t[(int64_t)pos[k] * H + h] = tt;
                fma4(acc, w[k], r[k]);
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

// ---- destination pass: softmax backward over the streamed [E,H] arrays -----------------------
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

// one thread per destination node with <= light_max in-edges
template <int H>
__global__ __launch_bounds__(256) void k_bwd_dst_light(BwdArgs a, int C) {
    for (int vi = blockIdx.x * 256 + threadIdx.x; vi < a.N; vi += gridDim.x * 256) {
        const int v = vi;  // one thread per node: neighbouring threads read neighbouring segments
        const int b = a.pin.ptr[v], e = a.pin.ptr[v + 1];
        if (e - b > a.pin.light_max) continue;
        HeadVec<H> delta, det;
#pragma unroll
        for (int h = 0; h < H; ++h) delta.v[h] = det.v[h] = 0.f;
        if (e == b) {
            sth<H>(a.d_y + (int64_t)v * a.ld_dy + C + H, det);
            continue;
        }
        if (e - b <= 4) {
            // short segment (the common case): one batch of clamped loads, everything in registers
            HeadVec<H> ap[4], tt[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int ii = min(b + k, e - 1);
                ap[k] = ldh<H>(a.attn + (int64_t)ii * H);
                tt[k] = ldh<H>(a.t + (int64_t)ii * H);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int h = 0; h < H; ++h)
                    delta.v[h] = b + k < e ? fmaf(fabsf(ap[k].v[h]), tt[k].v[h], delta.v[h]) : delta.v[h];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (b + k < e) {
                    HeadVec<H> dl;
#pragma unroll
                    for (int h = 0; h < H; ++h) {
                        float d = fabsf(ap[k].v[h]) * (tt[k].v[h] - delta.v[h]);
                        if (attn_neg(ap[k].v[h])) d *= a.slope;
                        dl.v[h] = d;
                        det.v[h] += d;
                    }
                    sth<H>(a.dl + (int64_t)(b + k) * H, dl);
                }
            }
        } else {
            for (int i = b; i < e; ++i) {
                const HeadVec<H> ap = ldh<H>(a.attn + (int64_t)i * H), tt = ldh<H>(a.t + (int64_t)i * H);
#pragma unroll
                for (int h = 0; h < H; ++h) delta.v[h] = fmaf(fabsf(ap.v[h]), tt.v[h], delta.v[h]);
            }
            for (int i = b; i < e; ++i) {
                const HeadVec<H> ap = ldh<H>(a.attn + (int64_t)i * H), tt = ldh<H>(a.t + (int64_t)i * H);
                HeadVec<H> dl;
#pragma unroll
                for (int h = 0; h < H; ++h) {
                    float d = fabsf(ap.v[h]) * (tt.v[h] - delta.v[h]);
                    if (attn_neg(ap.v[h])) d *= a.slope;
                    dl.v[h] = d;
                    det.v[h] += d;
                }
                sth<H>(a.dl + (int64_t)i * H, dl);
            }
        }
        sth<H>(a.d_y + (int64_t)v * a.ld_dy + C + H, det);
    }
}

// heavy destinations, step 1: wave per chunk, lane per edge: the chunk's sum of a * t
template <int H>
__global__ __launch_bounds__(256) void k_bwd_dst_heavy1(BwdArgs a) {
    const int lane = threadIdx.x & 63;
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= a.pin.n_chunks) return;
    const int beg = a.pin.chunk_beg[c], end = a.pin.chunk_end[c];
    float delta[H];
#pragma unroll
    for (int h = 0; h < H; ++h) delta[h] = 0.f;
    for (int i = beg + lane; i < end; i += 64) {
        const HeadVec<H> ap = ldh<H>(a.attn + (int64_t)i * H), tt = ldh<H>(a.t + (int64_t)i * H);
#pragma unroll
        for (int h = 0; h < H; ++h) delta[h] = fmaf(fabsf(ap.v[h]), tt.v[h], delta[h]);
    }
#pragma unroll
    for (int h = 0; h < H; ++h) {
        const float d = wave_sum(delta[h]);
        if (lane == 0) a.ws_delta[(int64_t)c * H + h] = d;
    }
}

// step 2: wave per chunk, lane per edge: node delta (chunk order), d_logit, chunk d_et
template <int H>
__global__ __launch_bounds__(256) void k_bwd_dst_heavy2(BwdArgs a) {
    const int lane = threadIdx.x & 63;
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= a.pin.n_chunks) return;
    const int slot = a.pin.chunk_node[c];
    const int c0 = a.pin.heavy_cptr[slot], c1 = a.pin.heavy_cptr[slot + 1];
    const int beg = a.pin.chunk_beg[c], end = a.pin.chunk_end[c];
    float delta[H], det[H];
#pragma unroll
    for (int h = 0; h < H; ++h) {
        float t = 0.f;
        for (int k = c0 + lane; k < c1; k += 64) t += a.ws_delta[(int64_t)k * H + h];
        delta[h] = wave_sum(t);
        det[h] = 0.f;
    }
    for (int i = beg + lane; i < end; i += 64) {
        const HeadVec<H> ap = ldh<H>(a.attn + (int64_t)i * H), tt = ldh<H>(a.t + (int64_t)i * H);
        HeadVec<H> dl;
#pragma unroll
        for (int h = 0; h < H; ++h) {
            float d = fabsf(ap.v[h]) * (tt.v[h] - delta[h]);
            if (attn_neg(ap.v[h])) d *= a.slope;
            dl.v[h] = d;
            det[h] += d;
        }
        sth<H>(a.dl + (int64_t)i * H, dl);
    }
#pragma unroll
    for (int h = 0; h < H; ++h) {
        float t = wave_sum(det[h]);
        if (lane == 0) a.ws_det[(int64_t)c * H + h] = t;
    }
}

// step 3: wave per heavy node: d_et = sum of chunk partials in chunk order
template <int H>
__global__ __launch_bounds__(256) void k_bwd_dst_heavy3(BwdArgs a, int C) {
    const int lane = threadIdx.x & 63;
    const int slot = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (slot >= a.pin.n_heavy) return;
    const int c0 = a.pin.heavy_cptr[slot], c1 = a.pin.heavy_cptr[slot + 1];
    const int v = a.pin.heavy_node[slot];
#pragma unroll
    for (int h = 0; h < H; ++h) {
        float t = 0.f;
        for (int k = c0 + lane; k < c1; k += 64) t += a.ws_det[(int64_t)k * H + h];
        t = wave_sum(t);
        if (lane == 0) a.d_y[(int64_t)v * a.ld_dy + C + H + h] = t;
    }
}

// ---- source pass 2: d_eh[u] = sum over out-edges of d_logit -----------------------------------
template <int H>
__global__ __launch_bounds__(256) void k_bwd_deh_light(BwdArgs a, int C) {
    for (int ui = blockIdx.x * 256 + threadIdx.x; ui < a.N; ui += gridDim.x * 256) {
        const int u = ui;
        const int b = a.pout.ptr[u], e = a.pout.ptr[u + 1];
        if (e - b > a.pout.light_max) continue;
        HeadVec<H> s;
#pragma unroll
        for (int h = 0; h < H; ++h) s.v[h] = 0.f;
        for (int j = b; j < e; j += 4) {
            HeadVec<H> d[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) d[k] = ldh<H>(a.dl + (int64_t)a.out_pos[min(j + k, e - 1)] * H);
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int h = 0; h < H; ++h) s.v[h] += j + k < e ? d[k].v[h] : 0.f;
        }
        sth<H>(a.d_y + (int64_t)u * a.ld_dy + C, s);
    }
}

template <int H>
__global__ __launch_bounds__(256) void k_bwd_deh_heavy1(BwdArgs a) {
    const int lane = threadIdx.x & 63;
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= a.pout.n_chunks) return;
    const int beg = a.pout.chunk_beg[c], end = a.pout.chunk_end[c];
    float s[H];
#pragma unroll
    for (int h = 0; h < H; ++h) s[h] = 0.f;
    for (int j = beg + lane; j < end; j += 64) {
        const HeadVec<H> d = ldh<H>(a.dl + (int64_t)a.out_pos[j] * H);
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
        const HeadVec<H> d0 = ldh<H>(a.dl + (int64_t)i * H), d1 = ldh<H>(a.dl + (int64_t)(i + stride) * H);
#pragma unroll
        for (int h = 0; h < H; ++h) tbl[(t0 * H + h) * TBL_THREADS + tid] += d0.v[h];
#pragma unroll
        for (int h = 0; h < H; ++h) tbl[(t1 * H + h) * TBL_THREADS + tid] += d1.v[h];
    }
    if (i < a.E) {
        const int t0 = a.etype[i];
        const HeadVec<H> d0 = ldh<H>(a.dl + (int64_t)i * H);
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
    static size_t attr_bytes = 0;
    if (sh > 48 * 1024 && sh > attr_bytes) {
        if (hipFuncSetAttribute((const void *)k_bwd_table_partial<H>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)sh) != hipSuccess)
            return MAGIC_ERR_RUNTIME;
        attr_bytes = sh;
    }
    MAGIC_LAUNCH(prof_on() ? prof_name("gat_bwd_table_partial", H, LPH) : "gat_bwd_table_partial", (k_bwd_table_partial<H>), a.grid_table, TBL_THREADS, sh, stream, a);
    MAGIC_LAUNCH("gat_bwd_table_final", k_bwd_table_final, a.e_dim * H, 256, 0, stream, a, H);
    return MAGIC_OK;
}

template <int H, int LPH>
int launch_bwd(BwdArgs &a, hipStream_t stream) {
    constexpr int C = H * LPH * 4;
    constexpr int NPB = 256 / (H * LPH);
    if (a.prelu_w || a.d_bias || a.d_pre != a.d_out) {
        MAGIC_LAUNCH(prof_on() ? prof_name("gat_bwd_node_adjoint", H, LPH) : "gat_bwd_node_adjoint", (k_bwd_node_adjoint<H, LPH>), a.grid_light, 256, 0, stream, a);
        if (a.d_bias || a.d_prelu)
            MAGIC_LAUNCH("gat_bwd_reduce_bias", k_bwd_reduce_bias, C + 1, 256, 0, stream, a, C);
    }
    MAGIC_LAUNCH(prof_on() ? prof_name("gat_bwd_src_light", H, LPH) : "gat_bwd_src_light", (k_bwd_src_light<H, LPH>), stride_grid(a.N, NPB, 256 * 16), 256, 0, stream, a);
    if (a.pout.n_chunks > 0) {
        const int cgrid = (a.pout.n_chunks + 3) / 4;
        MAGIC_LAUNCH(prof_on() ? prof_name("gat_bwd_src_heavy1", H, LPH) : "gat_bwd_src_heavy1", (k_bwd_src_heavy1<H, LPH>), cgrid, 256, 0, stream, a);
        MAGIC_LAUNCH("gat_bwd_src_heavy2", (k_bwd_src_heavy2<H, LPH>), a.pout.n_heavy, 256, 0, stream, a);
    }
    const int ngrid = stride_grid(a.N, 256, 256 * 16);
    MAGIC_LAUNCH(prof_on() ? prof_name("gat_bwd_dst_light", H, LPH) : "gat_bwd_dst_light", (k_bwd_dst_light<H>), ngrid, 256, 0, stream, a, C);
    if (a.pin.n_chunks > 0) {
        const int cgrid = (a.pin.n_chunks + 3) / 4;
        MAGIC_LAUNCH("gat_bwd_dst_heavy1", (k_bwd_dst_heavy1<H>), cgrid, 256, 0, stream, a);
        MAGIC_LAUNCH("gat_bwd_dst_heavy2", (k_bwd_dst_heavy2<H>), cgrid, 256, 0, stream, a);
        MAGIC_LAUNCH("gat_bwd_dst_heavy3", (k_bwd_dst_heavy3<H>), (a.pin.n_heavy + 3) / 4, 256, 0, stream, a, C);
    }
    MAGIC_LAUNCH(prof_on() ? prof_name("gat_bwd_deh_light", H, LPH) : "gat_bwd_deh_light", (k_bwd_deh_light<H>), ngrid, 256, 0, stream, a, C);
    if (a.pout.n_chunks > 0) {
        MAGIC_LAUNCH("gat_bwd_deh_heavy1", (k_bwd_deh_heavy1<H>), (a.pout.n_chunks + 3) / 4, 256, 0, stream, a);
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
    size_t gl = light_grid(N, H, D), gt = table_grid(E);
    size_t b = align_up(gl * C * f) + align_up(gl * f);
    b += 2 * align_up((size_t)(n_chunks_in > 0 ? n_chunks_in : 1) * H * f);
    b += align_up((size_t)(n_chunks_out > 0 ? n_chunks_out : 1) * C * f) +
         align_up((size_t)(n_chunks_out > 0 ? n_chunks_out : 1) * H * f);
    b += align_up(gt * (size_t)(e_dim > 0 ? e_dim : 1) * H * f);
    b += align_up((size_t)(E > 0 ? E : 1) * H * f);
    return b + 256;
}

int magic_gat_bwd(int32_t N, int32_t E, int32_t H, int32_t D, const magic_plan *in_plan, const int32_t *in_src,
                  const magic_plan *out_plan, const int32_t *out_dst, const int32_t *out_pos, const float *hs,
                  int64_t ld_hs, const float *attn, const int32_t *etype_csc, int32_t e_dim, float slope,
                  const float *pre, const float *act_out, int64_t ld_act, const float *prelu_w,
                  const float *d_out, int64_t ld_dout, float *d_pre,
                  int64_t ld_dpre, float *d_y, int64_t ld_dy, float *d_ee_table, float *d_logit, float *d_bias, float *d_prelu_w,
                  void *workspace, size_t workspace_bytes, void *stream_) {
    using namespace magic;
    if (N < 0 || E < 0 || H <= 0 || D <= 0) return MAGIC_ERR_INVALID_ARG;
    if (!shape_supported(H, D)) return MAGIC_ERR_UNSUPPORTED;
    if (N == 0) return MAGIC_OK;
    const int C = H * D;
    if (!plan_valid(in_plan) || !plan_valid(out_plan) || !hs || !d_out || !d_pre || !d_y) return MAGIC_ERR_INVALID_ARG;
    if (E > 0 && (!out_dst || !out_pos || !attn || !d_logit)) return MAGIC_ERR_INVALID_ARG;
    if (prelu_w && (!pre || !act_out || ld_act < C || (ld_act & 3))) return MAGIC_ERR_INVALID_ARG;
    if (prelu_w && d_pre == d_out) return MAGIC_ERR_INVALID_ARG;  // in-place only without an activation
    if (ld_hs < C || ld_dout < C || ld_dy < C + 2 * H || ld_dpre < C) return MAGIC_ERR_INVALID_ARG;
    if ((ld_hs | ld_dout | ld_dy | ld_dpre) & 3) return MAGIC_ERR_INVALID_ARG;
    if (etype_csc && d_ee_table && (e_dim <= 0 || e_dim > 32)) return MAGIC_ERR_UNSUPPORTED;
    BwdArgs a;
    a.N = N;
    a.E = E;
    a.pin = plan_dev(in_plan);
    a.pout = plan_dev(out_plan);
    a.out_dst = out_dst;
    a.out_pos = out_pos;
    a.hs = hs;
    a.ld_hs = ld_hs;
    a.attn = attn;
    a.etype = etype_csc;
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
    a.d_table = d_ee_table;
    a.dl = d_logit;
    a.d_bias = d_bias;
    a.d_prelu = d_prelu_w;
    a.grid_light = light_grid(N, H, D);
    a.grid_table = table_grid(E);
    Arena ar(workspace, workspace_bytes);
    a.ws_bias = ar.take<float>((size_t)a.grid_light * C);
    a.ws_prelu = ar.take<float>((size_t)a.grid_light);
    a.ws_delta = ar.take<float>((size_t)(a.pin.n_chunks > 0 ? a.pin.n_chunks : 1) * H);
    a.ws_det = ar.take<float>((size_t)(a.pin.n_chunks > 0 ? a.pin.n_chunks : 1) * H);
    a.ws_acc = ar.take<float>((size_t)(a.pout.n_chunks > 0 ? a.pout.n_chunks : 1) * C);
    a.ws_deh = ar.take<float>((size_t)(a.pout.n_chunks > 0 ? a.pout.n_chunks : 1) * H);
    a.ws_table = ar.take<float>((size_t)a.grid_table * (e_dim > 0 ? e_dim : 1) * H);
    a.t = ar.take<float>((size_t)(E > 0 ? E : 1) * H);
    if (!ar.ok()) return MAGIC_ERR_WORKSPACE;
    hipStream_t stream = (hipStream_t)stream_;
    const int lph = D / 4;
#define X(h, l) \
    if (H == h && lph == l) return launch_bwd<h, l>(a, stream);
    MAGIC_GAT_SHAPES(X)
#undef X
    return MAGIC_ERR_UNSUPPORTED;
}

}  // extern "C"
