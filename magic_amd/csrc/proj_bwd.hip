// Weight gradient of the node projection:  dW[M,F] = dY[N,M]^T . X[N,F]  on the matrix cores with
// split-bf16 products (bf16_split.h: fp32-grade results, fp32 accumulation).  The reduction runs
// over the N nodes (hundreds of thousands), the output is tiny, so the kernel streams both inputs
// once: every operand element is part of one coalesced 128-byte row segment read straight from
// HBM (lane l supplies dY[row][m0 + (l&31)] and X[row][f0 + (l&31)] for the eight rows
// r + 8*(l>>5) .. + 7 of a 16-row step, which are the lane's eight k of one 32x32x16 MFMA), no LDS
// staging.  Each wave owns a strip of rows and a register tile of the output; the waves of a
// block are summed through LDS in wave order and the block partials by a second launch in
// block order (bitwise reproducible, no atomics).
#include "common.h"
#include "bf16_split.h"

namespace {

typedef float float16_t __attribute__((ext_vector_type(16)));

struct DwArgs {
    int64_t n;
    int M, F;
    const float *dy;
    int64_t ld_dy;
    const float *x;
    int64_t ld_x;
    const uint8_t *x_mask;  // [N,F] keep mask on X (dropout), or null
    float x_scale;
    float *partial;  // [n_strips][n_groups][MT*FT][16][64]
    float *dw;       // [M,F]
    int n_strips;
    int64_t rows_per_wave;  // even
    int mg, fg;             // output tile groups along M and F
};

template <int MT, int FT>
__global__ __launch_bounds__(256) void k_proj_dw(DwArgs a) {
    extern __shared__ float lds[];  // MT*FT*16*64 floats
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int hf = lane >> 5, li = lane & 31;
    const int strip = blockIdx.x;
    const int grp = blockIdx.y;
    const int m0 = (grp / a.fg) * MT * 32, f0 = (grp % a.fg) * FT * 32;
    float16_t acc[MT][FT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int ft = 0; ft < FT; ++ft)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][ft][r] = 0.f;
    const int64_t r_beg = ((int64_t)strip * 4 + wave) * a.rows_per_wave;
    int64_t r_end = r_beg + a.rows_per_wave;
    if (r_end > a.n) r_end = a.n;
    // Loads are unconditional (clamped addresses, values zeroed afterwards): no divergent control
    // flow around them, so one iteration's U*(MT+FT) loads are all in flight together.
    bool mok[MT], fok[FT];
    int mcol[MT], fcol[FT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        mok[mt] = m0 + mt * 32 + li < a.M;
        mcol[mt] = mok[mt] ? m0 + mt * 32 + li : 0;
    }
#pragma unroll
    for (int ft = 0; ft < FT; ++ft) {
        fok[ft] = f0 + ft * 32 + li < a.F;
        fcol[ft] = fok[ft] ? f0 + ft * 32 + li : 0;
    }
#ifndef MAGIC_DW_U
#define MAGIC_DW_U 8
#endif
    constexpr int U = MAGIC_DW_U;  // rows per lane and step: the eight k of one 32x32x16 MFMA
    static_assert(U == 8, "one MFMA step takes eight rows from each lane half");
    const int64_t last = a.n - 1;
    for (int64_t r = r_beg; r < r_end; r += 2 * U) {
        float av[MT][U], bv[FT][U];
        uint8_t mv[FT][U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t row = r + U * hf + u;
            const int64_t rc = row < last ? row : last;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) av[mt][u] = a.dy[rc * a.ld_dy + mcol[mt]];
#pragma unroll
            for (int ft = 0; ft < FT; ++ft) bv[ft][u] = a.x[rc * a.ld_x + fcol[ft]];
            if (a.x_mask) {
#pragma unroll
                for (int ft = 0; ft < FT; ++ft) mv[ft][u] = a.x_mask[rc * a.F + fcol[ft]];
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool rok = r + U * hf + u < r_end;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) av[mt][u] = (rok && mok[mt]) ? av[mt][u] : 0.f;
#pragma unroll
            for (int ft = 0; ft < FT; ++ft) {
                float b = (rok && fok[ft]) ? bv[ft][u] : 0.f;
                if (a.x_mask) b = mv[ft][u] ? b * a.x_scale : 0.f;
                bv[ft][u] = b;
            }
        }
        magic::bf16x8_t a1[MT], a2[MT], a3[MT], b1[FT], b2[FT], b3[FT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
            magic::split8(make_float4(av[mt][0], av[mt][1], av[mt][2], av[mt][3]),
                          make_float4(av[mt][4], av[mt][5], av[mt][6], av[mt][7]), a1[mt], a2[mt], a3[mt]);
#pragma unroll
        for (int ft = 0; ft < FT; ++ft)
            magic::split8(make_float4(bv[ft][0], bv[ft][1], bv[ft][2], bv[ft][3]),
                          make_float4(bv[ft][4], bv[ft][5], bv[ft][6], bv[ft][7]), b1[ft], b2[ft], b3[ft]);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int ft = 0; ft < FT; ++ft)
                magic::mfma_split6(acc[mt][ft], a1[mt], a2[mt], a3[mt], b1[ft], b2[ft], b3[ft]);
    }
    // waves 1..3 park their tiles in LDS; wave 0 adds them in wave order and writes the block partial
    constexpr int TILE = MT * FT * 16 * 64;
    for (int w = 1; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int ft = 0; ft < FT; ++ft)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int o = ((mt * FT + ft) * 16 + r) * 64 + lane;
                        lds[o] = w == 1 ? acc[mt][ft][r] : lds[o] + acc[mt][ft][r];
                    }
        }
        __syncthreads();
    }
    if (wave == 0) {
        float *out = a.partial + ((int64_t)strip * gridDim.y + grp) * TILE;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int ft = 0; ft < FT; ++ft)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int o = ((mt * FT + ft) * 16 + r) * 64 + lane;
                    out[o] = acc[mt][ft][r] + lds[o];
                }
    }
}

// dW[i][j] = sum over strips of the partial tiles, in a fixed tree: a block takes 32 consecutive
// tile elements; eight thread groups sum one contiguous eighth of the strips each (strip order,
// four interleaved running sums), thread group 0 adds the eight results in order.
template <int MT, int FT>
__global__ __launch_bounds__(256) void k_proj_dw_reduce(DwArgs a) {
    constexpr int TILE = MT * FT * 16 * 64;
    __shared__ float part[8][32];
    const int grp = blockIdx.y;
    const int e = threadIdx.x & 31, sl = threadIdx.x >> 5;
    const int o = blockIdx.x * 32 + e;  // TILE is a multiple of 32
    const float *p = a.partial + (int64_t)grp * TILE + o;
    const int64_t stride = (int64_t)gridDim.y * TILE;
    const int per = (a.n_strips + 7) / 8;
    const int k0 = sl * per, k1 = min(k0 + per, a.n_strips);
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    int k = k0;
    for (; k + 4 <= k1; k += 4)
#pragma unroll
        for (int u = 0; u < 4; ++u) s[u] += p[(int64_t)(k + u) * stride];
    for (; k < k1; ++k) s[0] += p[(int64_t)k * stride];
    part[sl][e] = (s[0] + s[1]) + (s[2] + s[3]);
    __syncthreads();
    if (sl != 0) return;
    float t = part[0][e];
#pragma unroll
    for (int q = 1; q < 8; ++q) t += part[q][e];
    const int lane = o & 63, r = (o >> 6) & 15, tl = o >> 10;
    const int mt = tl / FT, ft = tl % FT;
    const int m0 = (grp / a.fg) * MT * 32, f0 = (grp % a.fg) * FT * 32;
    const int i = m0 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    const int j = f0 + ft * 32 + (lane & 31);
    if (i < a.M && j < a.F) a.dw[(int64_t)i * a.F + j] = t;
}

struct DwPlan {
    int mt, ft, mg, fg, n_strips;
    int64_t rows_per_wave;
    size_t partial_floats;
};

DwPlan plan_dw(int64_t n, int M, int F) {
    DwPlan p;
    const int mtiles = (M + 31) / 32, ftiles = (F + 31) / 32;
    // register tile (at most 6 accumulator tiles per wave) that re-reads the least: every M group
    // streams X once more, every F group streams dY once more
    static const int cand[][2] = {{3, 2}, {2, 3}, {3, 1}, {1, 3}, {2, 2}, {2, 1}, {1, 2}, {1, 1}};
    double best = -1.0;
    p.mt = p.ft = 1;
    for (auto &c : cand) {
        const int mt = c[0] < mtiles ? c[0] : mtiles, ft = c[1] < ftiles ? c[1] : ftiles;
        const int mg = (mtiles + mt - 1) / mt, fg = (ftiles + ft - 1) / ft;
        const double cost = (double)fg * M + (double)mg * F;
        if (best < 0 || cost < best - 1e-9) {
            best = cost;
            p.mt = mt;
            p.ft = ft;
        }
    }
    p.mg = (mtiles + p.mt - 1) / p.mt;
    p.fg = (ftiles + p.ft - 1) / p.ft;
    int groups = p.mg * p.fg;
    int strips = 512 / groups;  // two resident blocks per CU: the whole grid runs as one wave of blocks
    if (strips < 8) strips = 8;
    int64_t waves = (int64_t)strips * 4;
    int64_t rpw = (n + waves - 1) / waves;
    rpw = (rpw + 2 * MAGIC_DW_U - 1) / (2 * MAGIC_DW_U) * (2 * MAGIC_DW_U);  // whole unrolled iterations
    if (rpw < 2 * MAGIC_DW_U) rpw = 2 * MAGIC_DW_U;
    p.rows_per_wave = rpw;
    p.n_strips = (int)((n + rpw * 4 - 1) / (rpw * 4));
    if (p.n_strips < 1) p.n_strips = 1;
    p.partial_floats = (size_t)p.n_strips * groups * p.mt * p.ft * 16 * 64;
    return p;
}

template <int MT, int FT> int run_dw(DwArgs &a, const DwPlan &p, hipStream_t stream) {
    constexpr int TILE = MT * FT * 16 * 64;
    size_t sh = (size_t)TILE * sizeof(float);
    if (magic::ensure_dynamic_lds((const void *)k_proj_dw<MT, FT>, sh) != MAGIC_OK) return MAGIC_ERR_RUNTIME;
    dim3 grid(p.n_strips, p.mg * p.fg);
    MAGIC_LAUNCH(magic::prof_on() ? magic::prof_name("proj_bwd_weight", a.M, a.F) : "proj_bwd_weight", (k_proj_dw<MT, FT>), grid, 256, sh, stream, a);
    dim3 rgrid(TILE / 32, p.mg * p.fg);
    MAGIC_LAUNCH("proj_bwd_weight_reduce", (k_proj_dw_reduce<MT, FT>), rgrid, 256, 0, stream, a);
    return magic::launch_status();
}

// ---- wide layers (M >= 128 and F >= 128): block-tiled variant ---------------------------------
// The 64 x 96 register tile above re-reads both operands once per tile column / row (3-5 times at
// hidden 256: 790 MB per launch).  Here a block of four waves shares a 128 x 128 output tile.
// Both operands are row-major with the reduction index (the row) as the slow one, while an MFMA
// fragment wants eight consecutive rows of one column; the transposition happens in how LDS is
// laid out, not in how memory is read:
//  * a thread loads two consecutive rows of four columns as two float4 (coalesced 512-byte row
//    segments), splits the four (row, row+1) pairs into bf16 planes and stores, per plane, four
//    packed pairs with one 16-byte write into  tile[plane][row pair][column]  (conflict free);
//  * a lane of v_mfma_f32_16x16x32_bf16 (one 32-row step = one matrix instruction deep; the chip holds a
//    higher clock on this shape than on 32x32x16) assembles its fragment (column c, rows 8q .. 8q+7,
//    q = lane / 16) from the four row pairs 4q .. 4q+3 at column c: four 4-byte reads, 132 words apart,
//    so the four lane quarters sit 16 banks apart and a wave covers all 64 banks once.
// Row strips go to different blocks; the strip partials are added by a second launch in strip
// order (bitwise reproducible).
constexpr int WT = 128;            // tile edge
constexpr int WK = 32;             // rows per step
constexpr int WS = WT + 4;         // words per row pair of a plane
constexpr int WPLANE = (WK / 2) * WS;            // words per plane
constexpr size_t WIDE_LDS = (size_t)2 * 3 * WPLANE * 4;

struct WideArgs {
    DwArgs a;
    int m_blocks, f_blocks, n_strips;
    int64_t rows_per_strip;  // multiple of WK
};

typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

union PairBits {
    magic::bf16x2_t h;
    unsigned int u;
};

__global__ __launch_bounds__(256, 2) void k_proj_dw_wide(WideArgs w) {
    extern __shared__ __align__(16) unsigned int wsm[];
    unsigned int *sA = wsm;                // dY tile: [3][16][WS]
    unsigned int *sB = wsm + 3 * WPLANE;   // X tile
    const DwArgs &a = w.a;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l16 = lane & 15, quad = lane >> 4;
    const int wr = wave >> 1, wc = wave & 1;
    const int strip = blockIdx.x, tile = blockIdx.y;
    const int m0 = (tile / w.f_blocks) * WT, f0 = (tile % w.f_blocks) * WT;
    const int64_t r_beg = (int64_t)strip * w.rows_per_strip;
    int64_t r_end = r_beg + w.rows_per_strip;
    if (r_end > a.n) r_end = a.n;

    // lane l: A[i = l&15][k = 8*(l>>4) ..+7], B[k = 8*(l>>4) ..+7][j = l&15], D[i = 4*(l>>4) + r][j = l&15]
    f32x4_t acc[4][4];
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) acc[rt][ct] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    bool ok[4][4];  // MFMA tiles of this wave that hold output elements at all (uniform per wave)
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
            ok[rt][ct] = m0 + wr * 64 + rt * 16 < a.M && f0 + wc * 64 + ct * 16 < a.F;

    // staging role: four columns (c4) of both tiles, row pairs rp and rp + 8 of the step
    const int c4 = (tid & 31) * 4, rp = tid >> 5;
    const bool mok = m0 + c4 < a.M, fok = f0 + c4 < a.F;   // M % 4 == F % 4 == 0: whole float4s
    const float *pdy = a.dy + (mok ? m0 + c4 : 0);
    const float *px = a.x + (fok ? f0 + c4 : 0);
    const uint8_t *pmk = a.x_mask ? a.x_mask + (fok ? f0 + c4 : 0) : nullptr;
    const int64_t last = a.n - 1;
    f32x4_t va[2][2], vb[2][2];
    unsigned int vm[2][2];
    auto load_step = [&](int64_t r) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int64_t row = r + 2 * (rp + 8 * j) + e;
                const int64_t rc = row < last ? row : last;
                va[j][e] = *reinterpret_cast<const f32x4_t *>(pdy + rc * a.ld_dy);
                vb[j][e] = *reinterpret_cast<const f32x4_t *>(px + rc * a.ld_x);
                if (pmk) vm[j][e] = *reinterpret_cast<const unsigned int *>(pmk + rc * a.F);
            }
    };
    auto stage = [&](unsigned int *tile_base, int j, f32x4_t v0, f32x4_t v1) {
        u32x4_t p1, p2, p3;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            PairBits h1, h2, h3;
            magic::split2(magic::f32x2_t{v0[i], v1[i]}, h1.h, h2.h, h3.h);
            p1[i] = h1.u, p2[i] = h2.u, p3[i] = h3.u;
        }
        unsigned int *d = tile_base + (rp + 8 * j) * WS + c4;
        *reinterpret_cast<u32x4_t *>(d) = p1;
        *reinterpret_cast<u32x4_t *>(d + WPLANE) = p2;
        *reinterpret_cast<u32x4_t *>(d + 2 * WPLANE) = p3;
    };
    auto store_step = [&](int64_t r) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            f32x4_t ea[2], eb[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const bool rok = r + 2 * (rp + 8 * j) + e < r_end;
                ea[e] = (rok && mok) ? va[j][e] : f32x4_t{0.f, 0.f, 0.f, 0.f};
                f32x4_t b = (rok && fok) ? vb[j][e] : f32x4_t{0.f, 0.f, 0.f, 0.f};
                if (pmk) {
                    const unsigned int mw = vm[j][e];
                    b[0] = (mw & 0xffu) ? b[0] * a.x_scale : 0.f;
                    b[1] = (mw & 0xff00u) ? b[1] * a.x_scale : 0.f;
                    b[2] = (mw & 0xff0000u) ? b[2] * a.x_scale : 0.f;
                    b[3] = (mw & 0xff000000u) ? b[3] * a.x_scale : 0.f;
                }
                eb[e] = b;
            }
            stage(sA, j, ea[0], ea[1]);
            stage(sB, j, eb[0], eb[1]);
        }
    };
    // fragment of column `col` (tile-relative), rows 8 quad .. + 7 of the step: row pairs 4 quad .. + 3
    auto frag = [&](const unsigned int *plane, int col) {
        const unsigned int *q = plane + (4 * quad) * WS + col;
        union {
            u32x4_t u;
            magic::bf16x8_t h;
        } f;
        f.u = u32x4_t{q[0], q[WS], q[2 * WS], q[3 * WS]};
        return f.h;
    };

    if (r_beg < r_end) load_step(r_beg);
    for (int64_t r = r_beg; r < r_end; r += WK) {
        __syncthreads();  // the previous step's operand reads are done
        store_step(r);
        __syncthreads();
        if (r + WK < r_end) load_step(r + WK);
        magic::bf16x8_t ap[4][3];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) ap[t][pl] = frag(sA + pl * WPLANE, wr * 64 + t * 16 + l16);
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
            if (!ok[0][ct]) continue;
            magic::bf16x8_t bp[3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) bp[pl] = frag(sB + pl * WPLANE, wc * 64 + ct * 16 + l16);
            // the six kept products (bf16_split.h), smallest terms first; row tiles side by side so that
            // consecutive matrix instructions never accumulate into the result of the one before
            constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
            for (int p6 = 0; p6 < 6; ++p6)
#pragma unroll
                for (int rt = 0; rt < 4; ++rt)
                    if (ok[rt][ct])
                        acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ap[rt][PA[p6]], bp[PB[p6]], acc[rt][ct], 0, 0, 0);
        }
    }
    // strip partial of this tile, row-major [128][128]
    float *out = a.partial + ((int64_t)strip * gridDim.y + tile) * (WT * WT);
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = wr * 64 + rt * 16 + 4 * quad + r;
                out[i * WT + wc * 64 + ct * 16 + l16] = acc[rt][ct][r];
            }
}

// dW = sum over strips of the tile partials: one thread per element, strips in order (four
// interleaved running sums, combined in a fixed tree)
__global__ __launch_bounds__(256) void k_proj_dw_wide_reduce(WideArgs w) {
    const DwArgs &a = w.a;
    const int tile = blockIdx.y;
    const int e = blockIdx.x * 256 + threadIdx.x;  // element of the 128 x 128 tile
    const int i = (tile / w.f_blocks) * WT + (e >> 7), j = (tile % w.f_blocks) * WT + (e & 127);
    if (i >= a.M || j >= a.F) return;
    const float *p = a.partial + (int64_t)tile * (WT * WT) + e;
    const int64_t stride = (int64_t)gridDim.y * (WT * WT);
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    int k = 0;
    for (; k + 4 <= w.n_strips; k += 4)
#pragma unroll
        for (int u = 0; u < 4; ++u) s[u] += p[(int64_t)(k + u) * stride];
    for (; k < w.n_strips; ++k) s[0] += p[(int64_t)k * stride];
    a.dw[(int64_t)i * a.F + j] = (s[0] + s[1]) + (s[2] + s[3]);
}

// shapes and alignments the block-tiled kernel serves (whole float4s of both operands)
bool wide_ok(int M, int F, const float *dy, int64_t ld_dy, const float *x, int64_t ld_x, const uint8_t *mask) {
    if (!(M >= 128 && F >= 128 && M % 4 == 0 && F % 4 == 0 && ld_dy % 4 == 0 && ld_x % 4 == 0 &&
          (uintptr_t)dy % 16 == 0 && (uintptr_t)x % 16 == 0 && (uintptr_t)mask % 4 == 0))
        return false;
    // a ragged last tile row / column still stages whole operand tiles: with much padding (M = 264 is
    // 2 tiles + 8 columns) the register-tile kernel is the faster one (measured 134 us against 147 us)
    const int64_t padded = (int64_t)((M + WT - 1) / WT * WT) * ((F + WT - 1) / WT * WT);
    return padded * 100 <= (int64_t)M * F * 115;
}

struct WidePlan {
    int m_blocks, f_blocks, n_strips;
    int64_t rows_per_strip;
    size_t partial_floats;
};

WidePlan plan_wide(int64_t n, int M, int F) {
    WidePlan p;
    p.m_blocks = (M + WT - 1) / WT;
    p.f_blocks = (F + WT - 1) / WT;
    const int tiles = p.m_blocks * p.f_blocks;
    int strips = 512 / tiles;  // two resident blocks per CU: the whole grid runs as one wave of blocks
    if (strips < 1) strips = 1;
    int64_t rps = (n + strips - 1) / strips;
    rps = (rps + WK - 1) / WK * WK;
    if (rps < WK) rps = WK;
    p.rows_per_strip = rps;
    p.n_strips = (int)((n + rps - 1) / rps);
    if (p.n_strips < 1) p.n_strips = 1;
    p.partial_floats = (size_t)p.n_strips * tiles * WT * WT;
    return p;
}

int run_wide(DwArgs &a, const WidePlan &p, hipStream_t stream) {
    if (magic::ensure_dynamic_lds((const void *)k_proj_dw_wide, WIDE_LDS) != MAGIC_OK) return MAGIC_ERR_RUNTIME;
    WideArgs w;
    w.a = a;
    w.m_blocks = p.m_blocks, w.f_blocks = p.f_blocks, w.n_strips = p.n_strips;
    w.rows_per_strip = p.rows_per_strip;
    dim3 grid(p.n_strips, p.m_blocks * p.f_blocks);
    MAGIC_LAUNCH(magic::prof_on() ? magic::prof_name("proj_bwd_weight", a.M, a.F) : "proj_bwd_weight", k_proj_dw_wide, grid, 256, WIDE_LDS, stream, w);
    dim3 rgrid(WT * WT / 256, p.m_blocks * p.f_blocks);
    MAGIC_LAUNCH("proj_bwd_weight_reduce", k_proj_dw_wide_reduce, rgrid, 256, 0, stream, w);
    return magic::launch_status();
}

}  // namespace

extern "C" {

size_t magic_proj_bwd_weight_workspace_bytes(int64_t n_rows, int32_t m, int32_t f) {
    if (n_rows <= 0 || m <= 0 || f <= 0) return 256;
    size_t floats = plan_dw(n_rows, m, f).partial_floats;
    if (m >= 128 && f >= 128) {  // either kernel may run (operand alignment decides at the call)
        const size_t wide = plan_wide(n_rows, m, f).partial_floats;
        if (wide > floats) floats = wide;
    }
    return magic::align_up(floats * sizeof(float)) + 256;
}

int magic_proj_bwd_weight(int64_t n_rows, int32_t m, int32_t f, const float *dy, int64_t ld_dy, const float *x,
                          int64_t ld_x, const uint8_t *x_mask, float x_scale, float *dw, void *workspace,
                          size_t workspace_bytes, void *stream_) {
    if (n_rows < 0 || m <= 0 || f <= 0) return MAGIC_ERR_INVALID_ARG;
    if (!dw) return MAGIC_ERR_INVALID_ARG;
    hipStream_t stream = (hipStream_t)stream_;
    if (n_rows == 0) {
        if (hipMemsetAsync(dw, 0, sizeof(float) * (size_t)m * f, stream) != hipSuccess) return MAGIC_ERR_RUNTIME;
        return MAGIC_OK;
    }
    if (!dy || !x || ld_dy < m || ld_x < f) return MAGIC_ERR_INVALID_ARG;
    magic::Arena ar(workspace, workspace_bytes);
    DwArgs a;
    if (wide_ok(m, f, dy, ld_dy, x, ld_x, x_mask)) {
        const WidePlan wp = plan_wide(n_rows, m, f);
        a.n = n_rows, a.M = m, a.F = f;
        a.dy = dy, a.ld_dy = ld_dy, a.x = x, a.ld_x = ld_x;
        a.x_mask = x_mask, a.x_scale = x_scale;
        a.partial = ar.take<float>(wp.partial_floats);
        if (!ar.ok() || !a.partial) return MAGIC_ERR_WORKSPACE;
        a.dw = dw;
        a.n_strips = wp.n_strips, a.rows_per_wave = 0, a.mg = wp.m_blocks, a.fg = wp.f_blocks;
        return run_wide(a, wp, stream);
    }
    DwPlan p = plan_dw(n_rows, m, f);
    a.n = n_rows;
    a.M = m;
    a.F = f;
    a.dy = dy;
    a.ld_dy = ld_dy;
    a.x = x;
    a.ld_x = ld_x;
    a.x_mask = x_mask;
    a.x_scale = x_scale;
    a.partial = ar.take<float>(p.partial_floats);
    if (!ar.ok() || !a.partial) return MAGIC_ERR_WORKSPACE;
    a.dw = dw;
    a.n_strips = p.n_strips;
    a.rows_per_wave = p.rows_per_wave;
    a.mg = p.mg;
    a.fg = p.fg;
    if (p.mt == 3 && p.ft == 2) return run_dw<3, 2>(a, p, stream);
    if (p.mt == 2 && p.ft == 3) return run_dw<2, 3>(a, p, stream);
    if (p.mt == 1 && p.ft == 3) return run_dw<1, 3>(a, p, stream);
    if (p.mt == 3 && p.ft == 1) return run_dw<3, 1>(a, p, stream);
    if (p.mt == 2 && p.ft == 2) return run_dw<2, 2>(a, p, stream);
    if (p.mt == 2 && p.ft == 1) return run_dw<2, 1>(a, p, stream);
    if (p.mt == 1 && p.ft == 2) return run_dw<1, 2>(a, p, stream);
    return run_dw<1, 1>(a, p, stream);
}

}  // extern "C"
