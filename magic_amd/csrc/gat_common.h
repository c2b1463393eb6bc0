// Device helpers shared by the GAT forward / backward kernels.
//
// Thread mapping used everywhere: a node's H*D channels are spread over G = H*LPH lanes,
// 4 consecutive channels (one float4) per lane, LPH = D/4 lanes per head.  G is a power of
// two <= 64, so 64/G nodes (light kernels) or 64/G edge sub-streams (heavy kernels) share
// one wavefront.
#pragma once
#include "common.h"
#include <math.h>

namespace magic {

struct PlanDev {
    const int32_t *ptr;
    int32_t light_max;
    int32_t n_heavy;
    const int32_t *heavy_node;
    const int32_t *heavy_cptr;
    int32_t n_chunks;
    const int32_t *chunk_node;
    const int32_t *chunk_beg;
    const int32_t *chunk_end;
    const int32_t *order;      // node walk order, or null
    const int32_t *xcd_start;  // [9] cuts of the walk into 8 ranges, or null
};

inline PlanDev plan_dev(const magic_plan *p) {
    PlanDev d;
    d.ptr = p->ptr;
    d.light_max = p->light_max;
    d.n_heavy = p->n_heavy;
    d.heavy_node = p->heavy_node;
    d.heavy_cptr = p->heavy_cptr;
    d.n_chunks = p->n_chunks;
    d.chunk_node = p->chunk_node;
    d.chunk_beg = p->chunk_beg;
    d.chunk_end = p->chunk_end;
    d.order = p->node_order;
    d.xcd_start = p->xcd_start;
    return d;
}

inline bool plan_valid(const magic_plan *p) {
    if (!p || !p->ptr || p->light_max < 0 || p->n_heavy < 0 || p->n_chunks < 0) return false;
    if (p->n_heavy > 0 && (!p->heavy_node || !p->heavy_cptr)) return false;
    if (p->n_chunks > 0 && (!p->chunk_node || !p->chunk_beg || !p->chunk_end)) return false;
    if ((p->n_heavy == 0) != (p->n_chunks == 0)) return false;
    return true;
}

// ---- XCD-aware work distribution --------------------------------------------------------------
// Workgroups are dealt round robin over the 8 XCDs (b and b + 8 share one L2).  A node-driven kernel
// gives XCD range x = b % 8 of the walk to the workgroups b = x, x + 8, ...; `per_block` walk positions
// per workgroup and trip.
struct Walk {
    int pos, end, stride;
};
__device__ __forceinline__ Walk xcd_walk(const PlanDev &p, int N, int per_block, int slot) {
    Walk w;
    if (p.xcd_start == nullptr || gridDim.x < 8) {
        w.pos = blockIdx.x * per_block + slot;
        w.end = N;
        w.stride = gridDim.x * per_block;
    } else {
        const int x = blockIdx.x & 7, lb = blockIdx.x >> 3;
        const int nlb = (gridDim.x - x + 7) >> 3;  // workgroups that share this range
        w.pos = p.xcd_start[x] + lb * per_block + slot;
        w.end = p.xcd_start[x + 1];
        w.stride = nlb * per_block;
    }
    return w;
}
// Chunk of wave `wave` (0..3) of this workgroup in a one-wave-per-chunk kernel whose grid is a multiple
// of 8 workgroups (chunk_grid): every XCD takes a contiguous run of chunks, i.e. of heavy nodes.
__device__ __forceinline__ int xcd_chunk(int wave) {
    const int per = (gridDim.x >> 3) * 4;
    return (blockIdx.x & 7) * per + (blockIdx.x >> 3) * 4 + wave;
}
inline int chunk_grid(int n_chunks) { return ((n_chunks + 3) / 4 + 7) / 8 * 8; }

__device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ void st4(float *p, float4 v) { *reinterpret_cast<float4 *>(p) = v; }
// streaming (read-once / write-once) forms: keep the caches for the gathered node rows
typedef float f4v_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld4_nt(const float *p) {
    const f4v_t v = __builtin_nontemporal_load(reinterpret_cast<const f4v_t *>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void st4_nt(float *p, float4 v) {
    f4v_t t;
    t.x = v.x, t.y = v.y, t.z = v.z, t.w = v.w;
    __builtin_nontemporal_store(t, reinterpret_cast<f4v_t *>(p));
}
typedef float f2v_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void st2_nt(float *p, float a, float b) {
    f2v_t t;
    t.x = a, t.y = b;
    __builtin_nontemporal_store(t, reinterpret_cast<f2v_t *>(p));
}
__device__ __forceinline__ float4 f4zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ void fma4(float4 &acc, float a, float4 r) {
    acc.x = fmaf(a, r.x, acc.x);
    acc.y = fmaf(a, r.y, acc.y);
    acc.z = fmaf(a, r.z, acc.z);
    acc.w = fmaf(a, r.w, acc.w);
}
__device__ __forceinline__ void add4(float4 &acc, float4 r) {
    acc.x += r.x;
    acc.y += r.y;
    acc.z += r.z;
    acc.w += r.w;
}
__device__ __forceinline__ float dot4(float4 a, float4 b) {
    return fmaf(a.w, b.w, fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)));
}

// Cross-lane exchange inside a row of 16 lanes through DPP (no LDS round trip):
// 0xB1 = quad_perm[1,0,3,2], 0x4E = quad_perm[2,3,0,1], 0x141 = row_half_mirror (i <-> 7-i),
// 0x140 = row_mirror (i <-> 15-i).  All lanes of a head group are active together, so every
// source lane is valid.
template <int CTRL> __device__ __forceinline__ float dpp_f(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}

// sum over the LPH (<= 16, aligned) lanes of one head; every lane receives the same bits
template <int LPH> __device__ __forceinline__ float head_sum(float v) {
    if (LPH >= 2) v += dpp_f<0xB1>(v);
    if (LPH >= 4) v += dpp_f<0x4E>(v);
    if (LPH >= 8) v += dpp_f<0x141>(v);
    if (LPH >= 16) v += dpp_f<0x140>(v);
    return v;
}

template <int LPH> __device__ __forceinline__ float head_max(float v) {
    if (LPH >= 2) v = fmaxf(v, dpp_f<0xB1>(v));
    if (LPH >= 4) v = fmaxf(v, dpp_f<0x4E>(v));
    if (LPH >= 8) v = fmaxf(v, dpp_f<0x141>(v));
    if (LPH >= 16) v = fmaxf(v, dpp_f<0x140>(v));
    return v;
}

// sum over the 64/G sub-streams of a wave (lanes that differ in bits >= log2 G)
template <int G> __device__ __forceinline__ float substream_sum(float v) {
#pragma unroll
    for (int off = G; off < 64; off <<= 1) v += __shfl_xor(v, off, 64);
    return v;
}

template <int G> __device__ __forceinline__ float4 substream_sum4(float4 v) {
    v.x = substream_sum<G>(v.x);
    v.y = substream_sum<G>(v.y);
    v.z = substream_sum<G>(v.z);
    v.w = substream_sum<G>(v.w);
    return v;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// running (max, sum of exp) pairs: fold (m2, s2) into (m, s)
__device__ __forceinline__ void ms_fold(float &m, float &s, float m2, float s2) {
    float mn = fmaxf(m, m2);
    float f1 = (m == -INFINITY) ? 0.f : expf(m - mn);
    float f2 = (m2 == -INFINITY) ? 0.f : expf(m2 - mn);
    s = s * f1 + s2 * f2;
    m = mn;
}

__device__ __forceinline__ float leaky(float x, float slope) { return x > 0.f ? x : x * slope; }

// attention weight with the sign bit carrying "pre-activation logit was not positive"
__device__ __forceinline__ float pack_attn(float a, float x) {
    return x > 0.f ? a : __uint_as_float(__float_as_uint(a) | 0x80000000u);
}
__device__ __forceinline__ bool attn_neg(float packed) { return (__float_as_uint(packed) >> 31) != 0u; }

// ---- coded layer input ---------------------------------------------------------------------------
// The first encoder layer's input row is one-hot(node type) or, for a masked node, the mask token, under a
// dropout keep mask: x[u, j] = keep_j(u) * scale * (masked(u) ? token[j] : [j == type(u)]).  Such a row takes one
// of n_dim + 1 + 2^n_dim values (a kept type, the dropped row, the token under one of the 2^n_dim keep patterns),
// so everything the layer derives linearly from it -- its projected row hs[u], its head scores eh[u] / et[u], its
// linear residual -- is one row of a small table (coded.hip builds it from the transposed augmented weight: 20 KB
// at n_dim 5, 4.4 MB at n_dim 13).  The sweeps fetch that row by a 4-byte code word per node (the row index)
// from the caches instead of gathering 4*C-byte rows of a materialised [N, C] array from HBM.
//   table [rows, ld]: columns [0, C) hs | [C, C+H) eh | [C+H, C+2H) et | [C+2H, 2C+2H) residual.
struct CodedIn {
    const uint32_t *code;  // [N] table row of every node
    const float *tab;      // [rows, ld]
    int ld;
    int C;  // width of the hs block (column offsets of the others follow from it and H)
};

// columns [col, col + 4) of the node's table row (col % 4 == 0)
__device__ __forceinline__ float4 coded4(const CodedIn &ci, uint32_t code, int col) {
    return ld4(ci.tab + (size_t)code * ci.ld + col);
}
// one column of it
__device__ __forceinline__ float coded1(const CodedIn &ci, uint32_t code, int col) {
    return ci.tab[(size_t)code * ci.ld + col];
}

// dispatch over the compiled (H, LPH) shapes
#define MAGIC_GAT_SHAPES(X) \
    X(4, 4)                 \
    X(4, 16)                \
    X(4, 8)                 \
    X(4, 2)                 \
    X(1, 1)                 \
    X(1, 2)                 \
    X(1, 4)

inline bool shape_supported(int H, int D) {
    if (D % 4) return false;
    int lph = D / 4;
#define X(h, l) \
    if (H == h && lph == l) return true;
    MAGIC_GAT_SHAPES(X)
#undef X
    return false;
}

}  // namespace magic
