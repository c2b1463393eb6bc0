// Block-tiled variant of the dense node projection for the wide layers (hidden 256 and its
// 1024-wide concatenation), where the projection is a GEMM bound by the matrix pipe rather than
// a stream: Y[N,M] = (X (.) mask)[N,K] . op(W), K >= 96, M >= 128.
//
//  * 128 x 128 output tile per block of four waves (2 x 2, 64 x 64 per wave = 4 x 4 tiles of
//    v_mfma_f32_16x16x32_bf16: on random operands the chip holds a higher clock on this shape than on
//    32x32x16 at the same cycles per product), K in steps of 32 = one matrix instruction deep.
//  * split-bf16 products (bf16_split.h).  The weight is split once per call by a small pre-pass
//    into its three bf16 planes, stored tile by tile ([column block][K step][plane][128][32], zero
//    padded) in the caller's workspace; the main kernel copies one 24 KB tile image per K step
//    into LDS with 16-byte loads and stores, no arithmetic.  The X tile is split when it is
//    staged (coalesced 128-byte row segments), once per column block instead of once per MFMA
//    tile.  (A variant with the lane-owns-row loads of proj_rows.hip, X never in LDS, measured
//    the same within 3 % at 50 more registers.)
//  * one LDS buffer, register prefetch: the X loads of step k+2 and the W loads of step k+1 are
//    issued before the MFMAs of step k and land in registers while those run; two resident blocks
//    per CU overlap one block's staging with the other's MFMAs.
//  * rows of 40 bf16 (80 bytes): the 16-byte operand reads of 16 consecutive lanes cover all 64
//    banks exactly once.
//  * blockIdx is remapped so that the column blocks of one row block run on the same XCD, next to
//    each other in time: the X tile comes out of that XCD's L2 for all but the first of them.
//  * the accumulators leave through a 16-row LDS strip per wave as whole rows of float4 (the
//    epilogue of proj_rows.hip: routed outputs, or the fused input-gradient chain with its sums).
#include "proj_common.h"
#include "bf16_split.h"

namespace {

using magic::bf16x2_t;
using magic::bf16x8_t;
using magic::f32x2_t;
using magic::split2;

constexpr int GBM = 128, GBN = 128, GBK = 32;
constexpr int GLD = GBK + 8;       // bf16 per staged row
constexpr int PLANE = 128 * GLD;   // bf16 per plane of one operand tile
constexpr size_t GEMM_LDS = (size_t)2 * 3 * PLANE * 2;
constexpr int LDO = 64 + 4;        // strip row stride (floats)
constexpr int TILE_ELEMS = 3 * GBN * GBK;  // bf16 per weight tile image in the workspace

typedef float vec4_t __attribute__((ext_vector_type(4)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));  // 16-byte register quad (plain vector: stays in VGPRs)

struct GemmPlan {
    int col_blocks;
    int64_t row_blocks;
    int64_t per_xcd;  // blocks per XCD (grid = 8 * per_xcd)
    int n_win;        // 64-column windows that hold output columns
    int nk;
    const __bf16 *planes;  // [col_blocks][nk][3][128][32]
};

union Pack4 {
    bf16x2_t h[2];
    uint2 u;
};

// four consecutive k of one tile row -> the three planes (8 bytes each)
__device__ __forceinline__ void stage4(__bf16 *dst, float4 v) {
    Pack4 p1, p2, p3;
    split2(f32x2_t{v.x, v.y}, p1.h[0], p2.h[0], p3.h[0]);
    split2(f32x2_t{v.z, v.w}, p1.h[1], p2.h[1], p3.h[1]);
    *reinterpret_cast<uint2 *>(dst) = p1.u;
    *reinterpret_cast<uint2 *>(dst + PLANE) = p2.u;
    *reinterpret_cast<uint2 *>(dst + 2 * PLANE) = p3.u;
}

// weight -> tile images of its three bf16 planes (two consecutive k per thread)
__global__ __launch_bounds__(256) void k_gemm_split_w(ProjArgs a, int col_blocks, int nk, __bf16 *planes) {
    const int64_t total = (int64_t)col_blocks * nk * GBN * (GBK / 2);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int k2 = (int)(i % (GBK / 2));
        const int row = (int)((i / (GBK / 2)) % GBN);
        const int64_t tile = i / ((GBK / 2) * GBN);   // cb * nk + kb
        const int kb = (int)(tile % nk), cb = (int)(tile / nk);
        const int col = cb * GBN + row, k = kb * GBK + 2 * k2;
        f32x2_t v = {0.f, 0.f};
        if (col < a.M) {
            if (k < a.F) v[0] = a.w[(int64_t)col * a.w_rs + (int64_t)k * a.w_cs];
            if (k + 1 < a.F) v[1] = a.w[(int64_t)col * a.w_rs + (int64_t)(k + 1) * a.w_cs];
        }
        bf16x2_t h1, h2, h3;
        split2(v, h1, h2, h3);
        __bf16 *d = planes + tile * TILE_ELEMS + row * GBK + 2 * k2;
        *reinterpret_cast<bf16x2_t *>(d) = h1;
        *reinterpret_cast<bf16x2_t *>(d + GBN * GBK) = h2;
        *reinterpret_cast<bf16x2_t *>(d + 2 * GBN * GBK) = h3;
    }
}

template <bool EP>
__global__ __launch_bounds__(256, 2) void k_proj_gemm(ProjArgs a, GemmPlan p) {
    extern __shared__ __align__(16) unsigned char smem[];
    __bf16 *sA = reinterpret_cast<__bf16 *>(smem);
    __bf16 *sB = sA + 3 * PLANE;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l16 = lane & 15, quad = lane >> 4;
    const int wr = wave >> 1, wc = wave & 1;
    // consecutive block ids alternate over the 8 XCDs: give every XCD one contiguous range of tiles
    const int64_t pid = (int64_t)(blockIdx.x & 7) * p.per_xcd + (blockIdx.x >> 3);
    if (pid >= p.row_blocks * p.col_blocks) return;
    const int cb = (int)(pid % p.col_blocks);
    const int64_t rb = pid / p.col_blocks;
    const int64_t row0 = rb * GBM;
    const int colb = cb * GBN;

    // lane l of a 16x16x32 step: A[i = l&15][k = 8*(l>>4) ..+7], B[k = 8*(l>>4) ..+7][j = l&15],
    // D[i = 4*(l>>4) + r][j = l&15], r = 0..3
    f32x4_t acc[4][4];
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) acc[rt][ct] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    // column tiles of this wave that hold output columns at all (uniform per wave)
    bool tile_ok[4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) tile_ok[ct] = colb + wc * 64 + ct * 16 < a.M;

    // staging roles: 8 threads per tile row (4 k each), 32 tile rows per pass, 4 passes
    const int sr = tid >> 3, sk = (tid & 7) * 4;
    const int64_t last = a.n - 1;
    auto load_a = [&](int kb, float4 (&ra)[4], uint32_t (&rm)[4]) {
        const int k = kb * GBK + sk;
        const int kc = k < a.F ? k : 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t row = row0 + sr + 32 * j;
            const int64_t rc = row < last ? row : last;
            ra[j] = *reinterpret_cast<const float4 *>(a.x + rc * a.ld_x + kc);
            if (a.x_mask) rm[j] = *reinterpret_cast<const uint32_t *>(a.x_mask + rc * a.F + kc);
        }
    };
    // weight tile image: 1536 16-byte pieces, six per thread; piece c = (plane, row, quarter row)
    const __bf16 *tiles = p.planes + (int64_t)cb * p.nk * TILE_ELEMS;
    auto load_w = [&](int kb, vec4_t (&rw)[6]) {
        const vec4_t *src = reinterpret_cast<const vec4_t *>(tiles + (int64_t)kb * TILE_ELEMS);
#pragma unroll
        for (int j = 0; j < 6; ++j) rw[j] = src[tid + 256 * j];
    };
    auto store_tiles = [&](int kb, const float4 (&ra)[4], const uint32_t (&rm)[4], const vec4_t (&rw)[6]) {
        const int k = kb * GBK + sk;
        const bool kok = k < a.F;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float4 v = ra[j];
            if (a.x_mask) {
                v.x = (rm[j] & 0xffu) ? v.x * a.x_scale : 0.f;
                v.y = (rm[j] & 0xff00u) ? v.y * a.x_scale : 0.f;
                v.z = (rm[j] & 0xff0000u) ? v.z * a.x_scale : 0.f;
                v.w = (rm[j] & 0xff000000u) ? v.w * a.x_scale : 0.f;
            }
            if (!kok) v = make_float4(0.f, 0.f, 0.f, 0.f);
            stage4(sA + (sr + 32 * j) * GLD + sk, v);
        }
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int c = tid + 256 * j;
            const int pr = c >> 2, part = c & 3;  // pr = plane * 128 + row
            *reinterpret_cast<vec4_t *>(sB + pr * GLD + part * 8) = rw[j];
        }
    };
    auto mfma_step = [&]() {
        bf16x8_t ap[4][3];
#pragma unroll
        for (int rt = 0; rt < 4; ++rt)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
                ap[rt][pl] = *reinterpret_cast<const bf16x8_t *>(sA + pl * PLANE + (wr * 64 + rt * 16 + l16) * GLD + quad * 8);
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
            if (tile_ok[ct]) {
                bf16x8_t bp[3];
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
                    bp[pl] = *reinterpret_cast<const bf16x8_t *>(sB + pl * PLANE + (wc * 64 + ct * 16 + l16) * GLD + quad * 8);
                // the six kept products (bf16_split.h), smallest terms first; the four row tiles side by side so
                // that consecutive matrix instructions never accumulate into the result of the one before
                constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
                for (int p6 = 0; p6 < 6; ++p6)
#pragma unroll
                    for (int rt = 0; rt < 4; ++rt)
                        acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ap[rt][PA[p6]], bp[PB[p6]], acc[rt][ct], 0, 0, 0);
            }
    };

    // X tiles are prefetched two K steps ahead (HBM latency), W tiles one (they come out of L2)
    const int nk = p.nk;
    float4 ra0[4], ra1[4];
    vec4_t rw[6];
    uint32_t rm0[4], rm1[4];
    load_a(0, ra0, rm0);
    load_w(0, rw);
    if (nk > 1) load_a(1, ra1, rm1);
    for (int kb = 0; kb < nk; kb += 2) {
        __syncthreads();  // the previous step's operand reads are done
        store_tiles(kb, ra0, rm0, rw);
        __syncthreads();
        if (kb + 1 < nk) load_w(kb + 1, rw);
        if (kb + 2 < nk) load_a(kb + 2, ra0, rm0);
        mfma_step();
        if (kb + 1 >= nk) break;
        __syncthreads();
        store_tiles(kb + 1, ra1, rm1, rw);
        __syncthreads();
        if (kb + 2 < nk) load_w(kb + 2, rw);
        if (kb + 3 < nk) load_a(kb + 3, ra1, rm1);
        mfma_step();
    }
    __syncthreads();  // LDS becomes the transposition strips

    float *os = reinterpret_cast<float *>(smem) + wave * (SR * LDO);
    const int gc = colb + wc * 64 + (lane & 15) * 4;  // this lane's four columns in every strip row
    const float w = (EP && a.prelu_w) ? a.prelu_w[0] : 1.f;
    const float inv_w = w > 0.f ? 1.f / w : 0.f;
    float4 cs4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float dprelu = 0.f;
    Dst d;
    d.p = nullptr, d.ld = 0, d.mode = 0;
    if (!EP) d = route(a, gc);
    // head scores from the output rows: this wave's 64 columns are one head (64-wide heads only)
    const bool scores = !EP && a.sc_src != nullptr;
    float4 sv = make_float4(0.f, 0.f, 0.f, 0.f), dv = sv;
    if (scores && gc + 4 <= a.C) {
        sv = *reinterpret_cast<const float4 *>(a.sc_src + gc);
        dv = *reinterpret_cast<const float4 *>(a.sc_dst + gc);
    }
    const int head = (colb + wc * 64) >> 6, n_heads = a.NS >> 1;
    if (tile_ok[0]) {
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) {  // 16 rows x 64 columns through the strip
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) os[(4 * quad + r) * LDO + ct * 16 + l16] = acc[rt][ct][r];
            __builtin_amdgcn_wave_barrier();
#pragma unroll 2
            for (int i = 0; i < 4; ++i) {
                const int r = (lane >> 4) + 4 * i;
                const float4 q = *reinterpret_cast<const float4 *>(os + r * LDO + (lane & 15) * 4);
                const int64_t orow = row0 + wr * 64 + rt * SR + r;
                if (EP) {
                    if (gc < a.C) epi_store4<true>(a, q, orow, gc, w, inv_w, cs4, dprelu);
                } else {
                    if (orow < a.n) {
                        if (d.mode == 1) *reinterpret_cast<float4 *>(d.p + orow * d.ld) = q;
                        if (d.mode == 2) route_store_elems(a, q, orow, gc);
                    }
                    if (scores) {  // (all lanes take part in the shuffles; 16 lanes hold one row)
                        float ps = fmaf(q.w, sv.w, fmaf(q.z, sv.z, fmaf(q.y, sv.y, q.x * sv.x)));
                        float pd = fmaf(q.w, dv.w, fmaf(q.z, dv.z, fmaf(q.y, dv.y, q.x * dv.x)));
#pragma unroll
                        for (int off = 1; off < 16; off <<= 1) {
                            ps += __shfl_xor(ps, off, 64);
                            pd += __shfl_xor(pd, off, 64);
                        }
                        if ((lane & 15) == 0 && orow < a.n && head < n_heads) {
                            a.score[orow * a.NS + head] = ps;
                            a.score[orow * a.NS + n_heads + head] = pd;
                        }
                    }
                }
            }
        }
    }
    if (EP && a.part_col) {
        // lanes l, l + 16, l + 32, l + 48 hold the same four columns
        cs4.x += __shfl_xor(cs4.x, 16, 64), cs4.y += __shfl_xor(cs4.y, 16, 64);
        cs4.z += __shfl_xor(cs4.z, 16, 64), cs4.w += __shfl_xor(cs4.w, 16, 64);
        cs4.x += __shfl_xor(cs4.x, 32, 64), cs4.y += __shfl_xor(cs4.y, 32, 64);
        cs4.z += __shfl_xor(cs4.z, 32, 64), cs4.w += __shfl_xor(cs4.w, 32, 64);
        const int64_t prow = rb * 2 + wr;           // part_col [2*row_blocks, C]
        if (lane < 16 && gc < a.C) *reinterpret_cast<float4 *>(a.part_col + prow * a.C + gc) = cs4;
        const int win = cb * 2 + wc;                // part_prelu [n_win, 2*row_blocks]
        if (a.part_prelu && win < p.n_win) {
            float s = dprelu;
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
            if (lane == 0) a.part_prelu[(int64_t)win * (p.row_blocks * 2) + prow] = s;
        }
    }
}

template <bool EP> int launch(const ProjArgs &a, const GemmPlan &p, hipStream_t stream) {
    if (magic::ensure_dynamic_lds((const void *)k_proj_gemm<EP>, GEMM_LDS) != MAGIC_OK) return MAGIC_ERR_RUNTIME;
    MAGIC_LAUNCH(a.name, (k_proj_gemm<EP>), (unsigned)(p.per_xcd * 8), 256, GEMM_LDS, stream, a, p);
    return magic::launch_status();
}

bool gemm_shape(int F, int M) { return F >= 96 && M >= 128; }

// Shapes the row-owner kernel cannot hold (its LDS-resident weight outgrows a CU: a long K against 64 outputs, or
// many outputs from a short K) still run faster here, with part of the 128-column block idle, than on the
// LDS-staged fallback; they are offered to this kernel after the row-owner one has declined.
bool gemm_shape_relaxed(int F, int M) { return gemm_shape(F, M) || (F >= 256 && M >= 64) || (F >= 64 && M >= 256); }

}  // namespace

namespace magic {

// bytes of workspace the block-tiled kernel needs for the split weight (0: shape not served)
size_t proj_gemm_workspace_bytes(int F, int M) {
    if (!gemm_shape_relaxed(F, M)) return 0;
    const size_t cbs = (M + GBN - 1) / GBN, nk = (F + GBK - 1) / GBK;
    return align_up(cbs * nk * TILE_ELEMS * 2);
}

// Runs the projection on the block-tiled kernel when it is a GEMM (wide K and M), the operand
// alignment allows it and `planes_ws` (proj_gemm_workspace_bytes) is given; MAGIC_ERR_UNSUPPORTED
// otherwise.  `n_part` / `n_win` describe the partial sums an epilogue launch writes:
// part_col [n_part, C], part_prelu [n_win, n_part].  `relaxed`: also take the shapes of gemm_shape_relaxed
// (launches without an epilogue only).
int proj_gemm_try(const ProjArgs &a, void *planes_ws, int64_t *n_part, int *n_win, bool dry_run, hipStream_t stream,
                  bool relaxed) {
    if (a.n <= 0 || a.col0 != 0) return MAGIC_ERR_UNSUPPORTED;
    if (!(relaxed && !has_epilogue(a) ? gemm_shape_relaxed(a.F, a.M) : gemm_shape(a.F, a.M))) return MAGIC_ERR_UNSUPPORTED;
    if ((a.F & 3) || !aligned16(a.x, a.ld_x)) return MAGIC_ERR_UNSUPPORTED;
    if (a.x_mask && ((uintptr_t)a.x_mask & 3)) return MAGIC_ERR_UNSUPPORTED;
    const bool ep = has_epilogue(a);
    if (ep) {
        if (a.M != a.C || (a.C & 3) || !aligned16(a.hs, a.ld_hs)) return MAGIC_ERR_UNSUPPORTED;
        if (a.add && !aligned16(a.add, a.ld_add)) return MAGIC_ERR_UNSUPPORTED;
        if (a.add2 && !aligned16(a.add2, a.ld_add2)) return MAGIC_ERR_UNSUPPORTED;
        if (a.act_out && (!aligned16(a.act_out, a.ld_act) || !aligned16(a.act_pre, a.C))) return MAGIC_ERR_UNSUPPORTED;
        if (a.out_mask && ((uintptr_t)a.out_mask & 3)) return MAGIC_ERR_UNSUPPORTED;
        if (a.part_col && ((uintptr_t)a.part_col & 15)) return MAGIC_ERR_UNSUPPORTED;
    }
    GemmPlan p;
    p.col_blocks = (a.M + GBN - 1) / GBN;
    p.nk = (a.F + GBK - 1) / GBK;
    p.row_blocks = (a.n + GBM - 1) / GBM;
    const int64_t blocks = p.row_blocks * p.col_blocks;
    p.per_xcd = (blocks + 7) / 8;
    p.n_win = (a.M + 63) / 64;
    if (p.per_xcd * 8 > 0x7fffffffLL) return MAGIC_ERR_UNSUPPORTED;
    if (n_part) *n_part = p.row_blocks * 2;
    if (n_win) *n_win = p.n_win;
    if (dry_run) return MAGIC_OK;
    if (!planes_ws || ((uintptr_t)planes_ws & 15)) return MAGIC_ERR_UNSUPPORTED;
    p.planes = static_cast<const __bf16 *>(planes_ws);
    const int64_t pairs = (int64_t)p.col_blocks * p.nk * GBN * (GBK / 2);
    MAGIC_LAUNCH("proj_split_weight", k_gemm_split_w, stride_grid(pairs, 256, 1024), 256, 0, stream, a, p.col_blocks,
                 p.nk, static_cast<__bf16 *>(planes_ws));
    return ep ? launch<true>(a, p, stream) : launch<false>(a, p, stream);
}

}  // namespace magic
