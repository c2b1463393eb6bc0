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
    static bool attr_set = false;
    if (sh > 48 * 1024 && !attr_set) {
        (void)hipFuncSetAttribute((const void *)k_proj_dw<MT, FT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
        attr_set = true;
    }
    dim3 grid(p.n_strips, p.mg * p.fg);
    MAGIC_LAUNCH("proj_bwd_weight_mfma_f32", (k_proj_dw<MT, FT>), grid, 256, sh, stream, a);
    dim3 rgrid(TILE / 32, p.mg * p.fg);
    MAGIC_LAUNCH("proj_bwd_weight_reduce", (k_proj_dw_reduce<MT, FT>), rgrid, 256, 0, stream, a);
    return magic::launch_status();
}

}  // namespace

extern "C" {

size_t magic_proj_bwd_weight_workspace_bytes(int64_t n_rows, int32_t m, int32_t f) {
    if (n_rows <= 0 || m <= 0 || f <= 0) return 256;
    return magic::align_up(plan_dw(n_rows, m, f).partial_floats * sizeof(float)) + 256;
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
    DwPlan p = plan_dw(n_rows, m, f);
    magic::Arena ar(workspace, workspace_bytes);
    DwArgs a;
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
