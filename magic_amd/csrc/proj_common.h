// Shared pieces of the dense-projection translation units (proj.hip, proj_rows.hip, proj_gemm.hip).
#pragma once
#include "common.h"

namespace magic {

struct ProjArgs {
    int64_t n;
    int F, M, C, NS, RC;
    const float *x;
    int64_t ld_x;
    const uint8_t *x_mask;  // [N,F] keep mask on X (dropout), or null
    float x_scale;
    const float *w;
    int64_t w_rs, w_cs;  // element (row r, k) of the weight = w[r * w_rs + k * w_cs]
    int col0;            // first output column of this launch (wide outputs take several)
    const char *name;
    float *hs;
    int64_t ld_hs;
    float *score;
    float *res;
    int64_t ld_res;
    // epilogue chain on the hs columns
    const float *add;
    int64_t ld_add;
    const uint8_t *out_mask;  // [N,C]
    float out_scale;
    const float *add2;
    int64_t ld_add2;
    const float *act_out;  // forward output of the layer whose PReLU is differentiated
    int64_t ld_act;
    const float *act_pre;  // [N,C] its pre-activation (read only when the slope is <= 0)
    const float *prelu_w;  // device scalar
    float *part_col;       // [n_part, C] per-wave column sums (or null)
    float *part_prelu;     // [n_part]
    // Head scores from the output row (model/gat.py:291-292: (feat * attn).sum(-1)) instead of from folded weight
    // rows: per output column the attention vector entries of the node as a source / as a destination, [C] each.
    // Set by the dispatcher only for the block-tiled kernel with 64-wide heads; M then excludes the score columns.
    const float *sc_src, *sc_dst;
};

}  // namespace magic

namespace {

using magic::ProjArgs;

typedef float float16_t __attribute__((ext_vector_type(16)));

constexpr int BM = 128;     // rows per block (block-tile variant)
constexpr int KC = 32;      // k per LDS chunk
constexpr int LDS_LD = 36;  // floats per staged row (144 B)


struct Sums {
    float col;    // running column sum of this lane's column (current column tile)
    float prelu;  // running slope gradient
};

// value -> epilogue chain for one element of an hs column; returns the stored value
__device__ __forceinline__ float chain(const ProjArgs &a, float v, int64_t row, int col, float w, float &dprelu) {
    if (a.add) v += a.add[row * a.ld_add + col];
    if (a.out_mask) v = a.out_mask[row * a.C + col] ? v * a.out_scale : 0.f;
    if (a.add2) v += a.add2[row * a.ld_add2 + col];
    if (a.act_out) {
        const float o = a.act_out[row * a.ld_act + col];
        float pre;
        if (w > 0.f)
            pre = o > 0.f ? o : o / w;  // PReLU is invertible for a positive slope
        else
            pre = a.act_pre[row * a.C + col];
        if (!(pre > 0.f)) {
            dprelu = fmaf(v, pre, dprelu);
            v *= w;
        }
    }
    return v;
}

// One float4 of the (masked) input row.  The loads themselves are unconditional -- row and
// column are clamped into the buffer and the lanes outside it zeroed afterwards -- so that a
// sequence of calls puts all its loads in flight together (no divergent control flow between them).
__device__ __forceinline__ float4 load_x4(const ProjArgs &a, int64_t row, int c, bool vec) {
    const bool rok = row < a.n;
    const int64_t rc = rok ? row : a.n - 1;
    float4 v;
    uint32_t mw = 0x01010101u;
    if (vec) {  // F % 4 == 0, 16-byte aligned rows
        const bool ok = rok && c < a.F;
        const int cc = c < a.F ? c : 0;
        v = *reinterpret_cast<const float4 *>(a.x + rc * a.ld_x + cc);
        if (a.x_mask) {
            if (((uintptr_t)a.x_mask & 3) == 0) {
                mw = *reinterpret_cast<const uint32_t *>(a.x_mask + rc * a.F + cc);
            } else {
                const uint8_t *m = a.x_mask + rc * a.F + cc;
                mw = m[0] | (m[1] << 8) | (m[2] << 16) | ((uint32_t)m[3] << 24);
            }
        }
        if (!ok) v = make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
        const int c0 = c < a.F ? c : 0, c1 = c + 1 < a.F ? c + 1 : 0, c2 = c + 2 < a.F ? c + 2 : 0,
                  c3 = c + 3 < a.F ? c + 3 : 0;
        const float *p = a.x + rc * a.ld_x;
        v.x = p[c0], v.y = p[c1], v.z = p[c2], v.w = p[c3];
        if (a.x_mask) {
            const uint8_t *m = a.x_mask + rc * a.F;
            mw = m[c0] | (m[c1] << 8) | (m[c2] << 16) | ((uint32_t)m[c3] << 24);
        }
        if (!(rok && c < a.F)) v.x = 0.f;
        if (!(rok && c + 1 < a.F)) v.y = 0.f;
        if (!(rok && c + 2 < a.F)) v.z = 0.f;
        if (!(rok && c + 3 < a.F)) v.w = 0.f;
    }
    if (a.x_mask) {
        v.x = (mw & 0xffu) ? v.x * a.x_scale : 0.f;
        v.y = (mw & 0xff00u) ? v.y * a.x_scale : 0.f;
        v.z = (mw & 0xff0000u) ? v.z * a.x_scale : 0.f;
        v.w = (mw & 0xff000000u) ? v.w * a.x_scale : 0.f;
    }
    return v;
}

__device__ __forceinline__ float4 load_w4(const ProjArgs &a, int r, int c, bool wvec) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a.col0 + r < a.M && c < a.F) {
        const float *p = a.w + (int64_t)(a.col0 + r) * a.w_rs + (int64_t)c * a.w_cs;
        if (wvec && c + 3 < a.F) {
            v = *reinterpret_cast<const float4 *>(p);
        } else {
            v.x = p[0];
            if (c + 1 < a.F) v.y = p[a.w_cs];
            if (c + 2 < a.F) v.z = p[2 * a.w_cs];
            if (c + 3 < a.F) v.w = p[3 * a.w_cs];
        }
    }
    return v;
}

// Row-major float4 form of the chain (streaming kernel, tile transposed through LDS).  `row` may
// lie past the last row (ragged last tile): operand loads then read the last row, the store and
// the sums are skipped.
template <bool EP>
__device__ __forceinline__ void epi_store4(const ProjArgs &a, float4 v, int64_t row, int c4, float w, float inv_w,
                                           float4 &cs, float &dprelu) {
    const bool rok = row < a.n;
    const int64_t rc = rok ? row : a.n - 1;
    if (EP) {
        float4 t1, t2, o, pre;
        uint32_t mw = 0;
        if (a.add) t1 = *reinterpret_cast<const float4 *>(a.add + rc * a.ld_add + c4);
        if (a.out_mask) mw = *reinterpret_cast<const uint32_t *>(a.out_mask + rc * a.C + c4);
        if (a.add2) t2 = *reinterpret_cast<const float4 *>(a.add2 + rc * a.ld_add2 + c4);
        if (a.act_out) {
            o = *reinterpret_cast<const float4 *>(a.act_out + rc * a.ld_act + c4);
            if (!(w > 0.f)) pre = *reinterpret_cast<const float4 *>(a.act_pre + rc * a.C + c4);
        }
        if (a.add) v.x += t1.x, v.y += t1.y, v.z += t1.z, v.w += t1.w;
        if (a.out_mask) {
            v.x = (mw & 0xffu) ? v.x * a.out_scale : 0.f;
            v.y = (mw & 0xff00u) ? v.y * a.out_scale : 0.f;
            v.z = (mw & 0xff0000u) ? v.z * a.out_scale : 0.f;
            v.w = (mw & 0xff000000u) ? v.w * a.out_scale : 0.f;
        }
        if (a.add2) v.x += t2.x, v.y += t2.y, v.z += t2.z, v.w += t2.w;
        if (a.act_out) {
            if (w > 0.f) {  // PReLU is invertible for a positive slope
                pre.x = o.x > 0.f ? o.x : o.x * inv_w;
                pre.y = o.y > 0.f ? o.y : o.y * inv_w;
                pre.z = o.z > 0.f ? o.z : o.z * inv_w;
                pre.w = o.w > 0.f ? o.w : o.w * inv_w;
            }
            if (!rok) v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (!(pre.x > 0.f)) dprelu = fmaf(v.x, pre.x, dprelu), v.x *= w;
            if (!(pre.y > 0.f)) dprelu = fmaf(v.y, pre.y, dprelu), v.y *= w;
            if (!(pre.z > 0.f)) dprelu = fmaf(v.z, pre.z, dprelu), v.z *= w;
            if (!(pre.w > 0.f)) dprelu = fmaf(v.w, pre.w, dprelu), v.w *= w;
        }
        if (rok) cs.x += v.x, cs.y += v.y, cs.z += v.z, cs.w += v.w;
    }
    if (rok) *reinterpret_cast<float4 *>(a.hs + row * a.ld_hs + c4) = v;
}

constexpr int SR = 16;  // rows per transposition strip

// where the four output columns [gc, gc+4) of a lane go: one float4 destination (hs | score | res
// all keep float4 alignment, see choose()), or nowhere (columns past M)
struct Dst {
    float *p;
    int64_t ld;
    int mode;  // 0 none, 1 float4, 2 element-wise (single-tile outputs only)
};

__device__ __forceinline__ Dst route(const ProjArgs &a, int gc) {
    Dst d;
    d.p = nullptr, d.ld = 0, d.mode = 0;
    const int s0 = a.C, r0 = a.C + a.NS;
    if (gc + 4 <= s0 && ((uintptr_t)a.hs & 15) == 0 && (a.ld_hs & 3) == 0) {
        d.p = a.hs + gc, d.ld = a.ld_hs, d.mode = 1;
    } else if (gc >= s0 && gc + 4 <= r0 && ((gc - s0) & 3) == 0 && (a.NS & 3) == 0 && ((uintptr_t)a.score & 15) == 0) {
        d.p = a.score + (gc - s0), d.ld = a.NS, d.mode = 1;
    } else if (gc >= r0 && gc + 4 <= a.M && ((gc - r0) & 3) == 0 && (a.ld_res & 3) == 0 &&
               ((uintptr_t)a.res & 15) == 0) {
        d.p = a.res + (gc - r0), d.ld = a.ld_res, d.mode = 1;
    } else if (gc < a.M) {
        d.mode = 2;
    }
    return d;
}

// element-wise routing of a float4 that straddles destinations or lacks float4 alignment
__device__ __forceinline__ void route_store_elems(const ProjArgs &a, float4 v, int64_t row, int c4) {
    const int s0 = a.C, r0 = a.C + a.NS;
    const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = c4 + j;
        if (c < s0)
            a.hs[row * a.ld_hs + c] = e[j];
        else if (c < r0)
            a.score[row * a.NS + (c - s0)] = e[j];
        else if (c < a.M)
            a.res[row * a.ld_res + (c - r0)] = e[j];
    }
}

inline bool has_epilogue(const ProjArgs &a) { return a.add || a.out_mask || a.add2 || a.act_out || a.part_col; }

inline bool aligned16(const void *p, int64_t ld) { return ((uintptr_t)p % 16 == 0) && (ld % 4 == 0); }


}  // namespace
