// Streaming variant of the dense node projection for the HBM-bound shapes (K <= a few hundred, at most 96
// outputs): every byte of X arrives by fully coalesced loads and the tile is turned into matrix-instruction
// operands inside the wavefront's own LDS strip.
//
//  * A wavefront owns 32-row tiles.  One K block of a tile (32 rows x KSB*16 columns) is loaded as whole rows:
//    consecutive lanes take consecutive 16-byte pieces of a row and then of the next row, so a wave-instruction
//    reads 1 KB of contiguous memory (whole 128-byte lines) instead of 64 pieces of 64 different rows.
//  * The pieces wait in registers while the wavefront works on the previous block: the loads of block u+1 are
//    issued before the matrix work on block u starts, so a computing wavefront keeps 8-10 KB in flight.  The
//    prefetch registers are free during the matrix work because its A operand comes from LDS, not from them.
//  * Block u is written to the strip (rows padded by one piece: conflict-free fragment reads), dropout mask and
//    zero padding applied on the way; lane (row r, half h) then reads the 8 consecutive k of each matrix step,
//    splits them into bf16 pieces (bf16_split.h: fp32-grade products, fixed summation order) and multiplies
//    them with the LDS-resident, pre-split weight.
//  * The accumulators leave through the same strip (the X block is dead by then) as whole rows of float4, so
//    the epilogue operands are read and the result written fully coalesced (proj_common.h: epi_store4).
//  * One block per CU (W is staged once per block), as many wavefronts as the LDS holds, at most 12
//    (168 VGPRs; 8 for the longest K blocks).
#include <cstdlib>
#include "proj_common.h"
#include "bf16_split.h"

namespace {

using magic::bf16x8_t;
using magic::bf16x2_t;
using magic::f32x2_t;
using magic::split2;
using magic::split8;

// waves per block (one block per CU) the register budget is sized for: 12 (168 VGPRs), 8 (256 VGPRs) when the
// prefetched unit alone takes 50 registers
constexpr int strip_max_waves(int ksb) { return ksb >= 5 ? 8 : 12; }

struct StripPlan {
    int nkb;   // K blocks per tile
    int ldw;   // bf16 elements per staged weight row (nkb*KSB*16 + 8)
    int wpb;   // waves per block
    int strip_floats;  // floats of one wave's strip
};

// The 2*KSB pieces a lane holds of one (tile, K block) unit: piece p = i*64 + lane of the 32 x (4*KSB) piece grid
// (row p / (4*KSB), piece p % (4*KSB) of the row), as loaded (mask bytes beside it).
template <int KSB> struct Unit {
    float4 v[2 * KSB];
    uint32_t m[2 * KSB];
};

template <int KSB>
__device__ __forceinline__ void load_unit(const ProjArgs &a, int64_t tile, int kb, int lane, Unit<KSB> &u) {
    constexpr int KCB = 4 * KSB;
#pragma unroll
    for (int i = 0; i < 2 * KSB; ++i) {
        const int p = i * 64 + lane;
        const int r = p / KCB, c = p % KCB;
        int64_t row = tile * 32 + r;
        if (row >= a.n) row = a.n - 1;
        int col = (kb * KCB + c) * 4;
        if (col >= a.F) col = 0;
        u.v[i] = *reinterpret_cast<const float4 *>(a.x + row * a.ld_x + col);
        if (a.x_mask) u.m[i] = *reinterpret_cast<const uint32_t *>(a.x_mask + row * a.F + col);
    }
}

// ... into the strip: zero past the last row / last column, dropout mask and scale applied
template <int KSB>
__device__ __forceinline__ void write_unit(const ProjArgs &a, int64_t tile, int kb, int lane, const Unit<KSB> &u, float *xs) {
    constexpr int KCB = 4 * KSB;
#pragma unroll
    for (int i = 0; i < 2 * KSB; ++i) {
        const int p = i * 64 + lane;
        const int r = p / KCB, c = p % KCB;
        float4 v = u.v[i];
        if (a.x_mask) {
            const uint32_t mw = u.m[i];
            v.x = (mw & 0xffu) ? v.x * a.x_scale : 0.f;
            v.y = (mw & 0xff00u) ? v.y * a.x_scale : 0.f;
            v.z = (mw & 0xff0000u) ? v.z * a.x_scale : 0.f;
            v.w = (mw & 0xff000000u) ? v.w * a.x_scale : 0.f;
        }
        if (tile * 32 + r >= a.n || (kb * KCB + c) * 4 >= a.F) v = make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4 *>(xs + (r * (KCB + 1) + c) * 4) = v;
    }
}

template <int KSB, int NCG, bool EP>
__global__ __launch_bounds__(strip_max_waves(KSB) * 64) void k_proj_strip(ProjArgs a, StripPlan p) {
    constexpr int KCB = 4 * KSB;       // 16-byte pieces per row and K block
    constexpr int LDO = NCG * 32 + 4;  // output strip row stride
    constexpr int C4 = NCG * 8;        // float4 per output strip row
    extern __shared__ __align__(16) float sm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 31, hf = lane >> 5;
    constexpr int mpad = NCG * 32;
    // three bf16 planes of the weight, wb[plane][col][k], then one strip per wave
    __bf16 *wb = reinterpret_cast<__bf16 *>(sm);
    const int plane = mpad * p.ldw;  // bf16 elements per plane (a multiple of 8)
    float *xs = sm + (size_t)3 * plane / 2 + (size_t)wave * p.strip_floats;
    float *os = xs;  // the output rows pass through the same memory once the X block is dead
    {                // stage the zero-padded weight
        const bool wvec = (a.F % 4 == 0) && a.w_cs == 1 && (a.w_rs % 4 == 0) && ((uintptr_t)a.w % 16 == 0);
        const int kq = (p.ldw - 8) / 4;
        for (int idx = tid; idx < mpad * kq; idx += blockDim.x) {
            const int r = idx / kq, c4 = (idx % kq) * 4;
            const float4 v = load_w4(a, r, c4, wvec);
            bf16x2_t h1[2], h2[2], h3[2];
            split2(f32x2_t{v.x, v.y}, h1[0], h2[0], h3[0]);
            split2(f32x2_t{v.z, v.w}, h1[1], h2[1], h3[1]);
            __bf16 *d = wb + r * p.ldw + c4;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                d[2 * j] = h1[j][0], d[2 * j + 1] = h1[j][1];
                d[plane + 2 * j] = h2[j][0], d[plane + 2 * j + 1] = h2[j][1];
                d[2 * plane + 2 * j] = h3[j][0], d[2 * plane + 2 * j + 1] = h3[j][1];
            }
        }
    }
    __syncthreads();
    // a block takes a contiguous run of tiles (equal shares to within one tile), its waves take them in turn
    const int64_t all_tiles = (a.n + 31) / 32;
    const int64_t n_tiles = all_tiles * (blockIdx.x + 1) / gridDim.x;   // one past this block's last tile
    const int64_t t_stride = p.wpb;
    const int64_t t_first = all_tiles * blockIdx.x / gridDim.x + wave;
    const float w = (EP && a.prelu_w) ? a.prelu_w[0] : 1.f;
    const float inv_w = w > 0.f ? 1.f / w : 0.f;
    float4 cs4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float dprelu = 0.f;

    if (t_first < n_tiles) {
        Unit<KSB> u;
        load_unit<KSB>(a, t_first, 0, lane, u);
        float16_t acc[NCG];
        int64_t t = t_first;
        int kb = 0;
        while (true) {
            // ---- this unit into the strip; the next one's loads go out before the matrix work starts
            __builtin_amdgcn_wave_barrier();
            write_unit<KSB>(a, t, kb, lane, u, xs);
            int64_t tn = t;
            int kbn = kb + 1;
            if (kbn == p.nkb) kbn = 0, tn = t + t_stride;
            const bool more = tn < n_tiles;
            if (more) load_unit<KSB>(a, tn, kbn, lane, u);  // (wave-uniform branch; the loads inside are unconditional)
            __builtin_amdgcn_wave_barrier();
            if (kb == 0) {
#pragma unroll
                for (int ct = 0; ct < NCG; ++ct)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[ct][r] = 0.f;
            }
#pragma unroll
            for (int s = 0; s < KSB; ++s) {
                const float *xp = xs + (lr * (KCB + 1) + 4 * s + 2 * hf) * 4;
                const float4 lo = *reinterpret_cast<const float4 *>(xp);
                const float4 hi = *reinterpret_cast<const float4 *>(xp + 4);
                bf16x8_t a1, a2, a3;
                split8(lo, hi, a1, a2, a3);
#pragma unroll
                for (int ct = 0; ct < NCG; ++ct) {
                    const __bf16 *wp = wb + (ct * 32 + lr) * p.ldw + (kb * KSB + s) * 16 + hf * 8;
                    const bf16x8_t b1 = *reinterpret_cast<const bf16x8_t *>(wp);
                    const bf16x8_t b2 = *reinterpret_cast<const bf16x8_t *>(wp + plane);
                    const bf16x8_t b3 = *reinterpret_cast<const bf16x8_t *>(wp + 2 * plane);
                    magic::mfma_split6(acc[ct], a1, a2, a3, b1, b2, b3);
                }
            }
            if (kb == p.nkb - 1) {
                // ---- accumulators through the strip, out as rows of float4
                // this lane's columns: the same in every strip row while a row's float4 count divides 64
                constexpr bool kFixedCols = (64 % C4) == 0;
                int gc = a.col0 + (lane % C4) * 4;
                Dst d;
                if (!EP && kFixedCols) d = route(a, gc);
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int ct = 0; ct < NCG; ++ct)
#pragma unroll
                        for (int r8 = 0; r8 < 8; ++r8) {
                            const int r = h * 8 + r8;  // rows (r&3) + 8*(r>>2) + 4*hf, minus 16*h
                            os[((r8 & 3) + 8 * (r8 >> 2) + 4 * hf) * LDO + ct * 32 + lr] = acc[ct][r];
                        }
                    __builtin_amdgcn_wave_barrier();
#pragma unroll 2
                    for (int i = 0; i < SR * C4 / 64; ++i) {
                        const int r = (lane + 64 * i) / C4;
                        const int cq = (lane + 64 * i) % C4;
                        if (!kFixedCols) {
                            gc = a.col0 + cq * 4;
                            if (!EP) d = route(a, gc);
                        }
                        const float4 q = *reinterpret_cast<const float4 *>(os + r * LDO + cq * 4);
                        const int64_t orow = t * 32 + h * SR + r;
                        if (EP) {
                            epi_store4<true>(a, q, orow, gc, w, inv_w, cs4, dprelu);
                        } else if (orow < a.n) {
                            if (d.mode == 1) *reinterpret_cast<float4 *>(d.p + orow * d.ld) = q;
                            if (d.mode == 2) route_store_elems(a, q, orow, gc);
                        }
                    }
                }
            }
            if (!more) break;
            t = tn, kb = kbn;
        }
    }

    if (EP && a.part_col) {
        // lanes l, l + C4, l + 2*C4, ... hold the same four columns
#pragma unroll
        for (int off = C4; off < 64; off <<= 1) {
            cs4.x += __shfl_xor(cs4.x, off, 64);
            cs4.y += __shfl_xor(cs4.y, off, 64);
            cs4.z += __shfl_xor(cs4.z, off, 64);
            cs4.w += __shfl_xor(cs4.w, off, 64);
        }
        const int64_t prow = (int64_t)blockIdx.x * p.wpb + wave;
        if (lane < C4 && a.col0 + lane * 4 < a.C)
            *reinterpret_cast<float4 *>(a.part_col + prow * a.C + a.col0 + lane * 4) = cs4;
        if (a.part_prelu) {
            float s = dprelu;
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
            if (lane == 0) a.part_prelu[prow] = s;
        }
    }
}

struct StripChoice {
    int ksb, ncg;
    StripPlan plan;
    size_t lds;
    int grid;
};

bool choose_strip(const ProjArgs &a, StripChoice &c) {
    if (a.n <= 0 || a.col0 != 0) return false;
    if ((a.F & 3) || !aligned16(a.x, a.ld_x)) return false;
    if (a.x_mask && ((uintptr_t)a.x_mask & 3)) return false;
    const int tiles = (a.M + 31) / 32;
    if (tiles > 3) return false;
    const bool ep = has_epilogue(a);
    if (ep) {
        // the vectorised epilogue: whole 32-column tiles of hs, float4-aligned operands
        if (a.M != a.C || (a.M != 32 && a.M != 64)) return false;   // (per-lane column sums: fixed columns per lane)
        if (!aligned16(a.hs, a.ld_hs)) return false;
        if (a.add && !aligned16(a.add, a.ld_add)) return false;
        if (a.add2 && !aligned16(a.add2, a.ld_add2)) return false;
        if (a.act_out && (!aligned16(a.act_out, a.ld_act) || !aligned16(a.act_pre, a.C))) return false;
        if (a.out_mask && (uintptr_t)a.out_mask % 4 != 0) return false;
        if (a.part_col && (uintptr_t)a.part_col % 16 != 0) return false;
    } else if (a.M > 32) {
        // column groups of several tiles route whole float4s only: destinations must keep float4 alignment
        if ((a.C & 3) || (a.NS & 3) || (a.RC & 3) || !aligned16(a.hs, a.ld_hs)) return false;
        if (a.NS && ((uintptr_t)a.score % 16 != 0)) return false;
        if (a.RC && !aligned16(a.res, a.ld_res)) return false;
    }
    const int ksteps = (a.F + 15) / 16;
    const int nkb = (ksteps + 4) / 5;
    const int ksb = (ksteps + nkb - 1) / nkb;
    c.ksb = ksb;
    c.ncg = tiles;
    c.plan.nkb = nkb;
    c.plan.ldw = nkb * ksb * 16 + 8;  // + 16 bytes: rows start in different banks
    const size_t w_bytes = (size_t)3 * tiles * 32 * c.plan.ldw * 2;
    const size_t x_strip = (size_t)32 * (4 * ksb + 1) * 16;
    const size_t o_strip = (size_t)SR * (tiles * 32 + 4) * sizeof(float);
    const size_t strip = magic::align_up(x_strip > o_strip ? x_strip : o_strip, 16);
    c.plan.strip_floats = (int)(strip / 4);
    int wpb = (int)((160 * 1024 - w_bytes) / strip);
    if (w_bytes >= 160 * 1024 || wpb < 6) return false;
    if (wpb > strip_max_waves(ksb)) wpb = strip_max_waves(ksb);
    c.plan.wpb = wpb;
    c.lds = w_bytes + wpb * strip;
    const int64_t tiles_n = (a.n + 31) / 32;
    const int64_t want = (tiles_n + wpb - 1) / wpb;
    c.grid = (int)(want < 256 ? want : 256);
    return true;
}

template <int KSB, int NCG, bool EP> int launch_strip(const ProjArgs &a, const StripChoice &c, hipStream_t stream) {
    if (magic::ensure_dynamic_lds((const void *)k_proj_strip<KSB, NCG, EP>, c.lds) != MAGIC_OK) return MAGIC_ERR_RUNTIME;
    MAGIC_LAUNCH(a.name, (k_proj_strip<KSB, NCG, EP>), c.grid, c.plan.wpb * 64, c.lds, stream, a, c.plan);
    return magic::launch_status();
}

template <int KSB> int launch_ksb(const ProjArgs &a, const StripChoice &c, bool ep, hipStream_t stream) {
    switch (c.ncg) {
        case 1: return ep ? launch_strip<KSB, 1, true>(a, c, stream) : launch_strip<KSB, 1, false>(a, c, stream);
        case 2: return ep ? launch_strip<KSB, 2, true>(a, c, stream) : launch_strip<KSB, 2, false>(a, c, stream);
        case 3: return ep ? launch_strip<KSB, 3, true>(a, c, stream) : launch_strip<KSB, 3, false>(a, c, stream);
    }
    return MAGIC_ERR_UNSUPPORTED;
}

}  // namespace

namespace magic {

// Runs the projection on the streaming kernel when the shape (at most 96 outputs, K % 4 == 0) and the
// operand alignment allow it; MAGIC_ERR_UNSUPPORTED otherwise (the caller goes on to the row-owner kernel).
// `n_part` receives the number of partial-sum rows an epilogue launch writes (one per wave of the grid),
// `n_win` the number of column windows (always 1 here).  With `dry_run` nothing is launched.
// MAGIC_PROJ_STRIP=0 in the environment switches the kernel off (A/B timing in one process).
int proj_strip_try(const ProjArgs &a0, int64_t *n_part, int *n_win, bool dry_run, hipStream_t stream) {
    const char *off = std::getenv("MAGIC_PROJ_STRIP");
    if (off && off[0] == '0') return MAGIC_ERR_UNSUPPORTED;
    ProjArgs a = a0;
    a.col0 = 0;
    StripChoice c;
    if (!choose_strip(a, c)) return MAGIC_ERR_UNSUPPORTED;
    if (n_part) *n_part = (int64_t)c.grid * c.plan.wpb;
    if (n_win) *n_win = 1;
    if (dry_run) return MAGIC_OK;
    const bool ep = has_epilogue(a);
    switch (c.ksb) {
        case 1: return launch_ksb<1>(a, c, ep, stream);
        case 2: return launch_ksb<2>(a, c, ep, stream);
        case 3: return launch_ksb<3>(a, c, ep, stream);
        case 4: return launch_ksb<4>(a, c, ep, stream);
        case 5: return launch_ksb<5>(a, c, ep, stream);
    }
    return MAGIC_ERR_UNSUPPORTED;
}

}  // namespace magic
