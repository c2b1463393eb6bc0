// Row-owner variant of the dense node projection, laid out for memory-level parallelism.
//
// Arithmetic: split-bf16 products (bf16_split.h): fp32-grade results with the matrix pipe busy 2.7x
// less than with the fp32 MFMA of the LDS-staged kernels in proj.hip.  Fixed summation order.
//
//  * A operand straight from HBM: lane l owns row (l & 31) of the wave's 32-row tile and the
//    half (l >> 5) of every K block -- KQB consecutive float4 of its own row, which are exactly the
//    per-lane A fragments of KQB/2 MFMA steps of 16 k (the K index inside a block is permuted so
//    that a lane half owns contiguous k; the weight fragments use the same permutation).  No
//    staging of X, no barrier, K padded to a multiple of 16 only.
//  * W (zero padded, already split into its three bf16 planes) stays in LDS for the life of the block.
//  * The 32 x 64 accumulator group goes through a 16-row LDS strip per wave and leaves as whole
//    rows of float4, so the epilogue operands (residual gradient, dropout mask, concatenation
//    gradient, activation output) are read and the result written fully coalesced.
//  * <= 128 VGPRs (168 for long row segments): 16 (12) waves per CU keep enough loads in flight
//    without software prefetch.
#include "proj_common.h"
#include "bf16_split.h"

namespace {

using magic::bf16x8_t;
using magic::bf16x2_t;
using magic::f32x2_t;
using magic::split2;
using magic::split8;

struct RowsPlan {
    int nkb;      // K blocks of KQB*8
    int ngroups;  // column groups of NCG*32
    int ldw;      // bf16 elements per staged weight row (nkb*KQB*8 + 8)
    int wpb;      // waves per block
};

// The lane's KQB float4 of one K block of its own row (k in [k0, k0 + 4*KQB)), dropout mask applied.
// Loads are unconditional (clamped row `rc`, clamped column): rows past the end and columns past F
// come back as zeros.  `vec`: rows are 16-byte aligned and F % 4 == 0; the element-wise form only
// exists for short rows (KQB <= 4).
template <int KQB>
__device__ __forceinline__ void load_a(const ProjArgs &a, int64_t rc, bool rok, int k0, bool vec, float4 (&af)[KQB]) {
    uint32_t mw[KQB];
    if (KQB > 4 || vec) {
        const float *px = a.x + rc * a.ld_x;
#pragma unroll
        for (int q = 0; q < KQB; ++q) {
            const int c = k0 + 4 * q;
            af[q] = *reinterpret_cast<const float4 *>(px + (c < a.F ? c : 0));
        }
        if (a.x_mask) {
            const uint8_t *pm = a.x_mask + rc * a.F;
            if (KQB % 4 == 0 && (a.F & 15) == 0 && ((uintptr_t)a.x_mask & 15) == 0) {
#pragma unroll
                for (int q = 0; q < KQB; q += 4) {  // 16 mask bytes per load
                    const int c = k0 + 4 * q;
                    const uint4 m = *reinterpret_cast<const uint4 *>(pm + (c < a.F ? c : 0));
                    mw[q] = m.x, mw[q + 1 < KQB ? q + 1 : q] = m.y, mw[q + 2 < KQB ? q + 2 : q] = m.z,
                    mw[q + 3 < KQB ? q + 3 : q] = m.w;
                }
            } else {
#pragma unroll
                for (int q = 0; q < KQB; ++q) {
                    const int c = k0 + 4 * q;
                    mw[q] = *reinterpret_cast<const uint32_t *>(pm + (c < a.F ? c : 0));
                }
            }
        }
#pragma unroll
        for (int q = 0; q < KQB; ++q)
            if (!(rok && k0 + 4 * q < a.F)) af[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
        const float *px = a.x + rc * a.ld_x;
        const uint8_t *pm = a.x_mask ? a.x_mask + rc * a.F : nullptr;
#pragma unroll
        for (int q = 0; q < KQB; ++q) {
            float e[4];
            uint32_t m = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c = k0 + 4 * q + j;
                const int cc = c < a.F ? c : 0;
                e[j] = px[cc];
                if (pm) m |= (uint32_t)(pm[cc] != 0) << (8 * j);
                if (!(rok && c < a.F)) e[j] = 0.f;
            }
            af[q] = make_float4(e[0], e[1], e[2], e[3]);
            mw[q] = m;
        }
    }
    if (a.x_mask) {
#pragma unroll
        for (int q = 0; q < KQB; ++q) {
            af[q].x = (mw[q] & 0xffu) ? af[q].x * a.x_scale : 0.f;
            af[q].y = (mw[q] & 0xff00u) ? af[q].y * a.x_scale : 0.f;
            af[q].z = (mw[q] & 0xff0000u) ? af[q].z * a.x_scale : 0.f;
            af[q].w = (mw[q] & 0xff000000u) ? af[q].w * a.x_scale : 0.f;
        }
    }
}

// waves per CU the register budget is sized for: 16 (128 VGPRs) for short rows, 12 (168 VGPRs)
// when the lane's row segment alone takes 32+ registers
constexpr int rows_max_waves(int kqb) { return kqb >= 8 ? 12 : 16; }

template <int KQB, int NCG, bool EP>
__global__ __launch_bounds__(rows_max_waves(KQB) * 64) void k_proj_rows(ProjArgs a, RowsPlan p) {
    constexpr int KBH = KQB * 4;       // k per lane half and block
    constexpr int KB = KQB * 8;        // k per block
    constexpr int LDO = NCG * 32 + 4;  // strip row stride
    constexpr int C4 = NCG * 8;        // float4 per strip row
    extern __shared__ __align__(16) float sm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 31, hf = lane >> 5;
    const int mpad = p.ngroups * NCG * 32;
    // three bf16 planes of the weight, wb[plane][col][k], then one fp32 strip per wave
    __bf16 *wb = reinterpret_cast<__bf16 *>(sm);
    const int plane = mpad * p.ldw;  // bf16 elements per plane (a multiple of 8)
    float *os = sm + (size_t)3 * plane / 2 + wave * (SR * LDO);
    {  // stage the zero-padded weight
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
    const bool vec = (a.F % 4 == 0) && (a.ld_x % 4 == 0) && ((uintptr_t)a.x % 16 == 0) &&
                     (!a.x_mask || ((uintptr_t)a.x_mask & 3) == 0);
    const int64_t n_tiles = (a.n + 31) / 32;
    const int64_t t_stride = (int64_t)gridDim.x * p.wpb;
    const float w = (EP && a.prelu_w) ? a.prelu_w[0] : 1.f;
    const float inv_w = w > 0.f ? 1.f / w : 0.f;
    float4 cs4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float dprelu = 0.f;

    for (int64_t t = (int64_t)blockIdx.x * p.wpb + wave; t < n_tiles; t += t_stride) {
        const int64_t row = t * 32 + lr;
        const bool rok = row < a.n;
        const int64_t rc = rok ? row : a.n - 1;
        float4 af[KQB];
        float16_t acc[NCG];

        // one column group's accumulators through the strip, out as rows of float4
        auto store_group = [&](int g) {
            const int gc = a.col0 + g * NCG * 32 + (lane % C4) * 4;  // this lane's columns in every strip row
            Dst d;
            if (!EP) d = route(a, gc);
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
                    const float4 q = *reinterpret_cast<const float4 *>(os + r * LDO + (lane % C4) * 4);
                    const int64_t orow = t * 32 + h * SR + r;
                    if (EP) {
                        epi_store4<true>(a, q, orow, gc, w, inv_w, cs4, dprelu);
                    } else if (orow < a.n) {
                        if (d.mode == 1) *reinterpret_cast<float4 *>(d.p + orow * d.ld) = q;
                        if (NCG == 1 && d.mode == 2) route_store_elems(a, q, orow, gc);
                    }
                }
            }
        };

        auto mfma_block = [&](int g, int b) {
#pragma unroll
            for (int m = 0; m < KQB / 2; ++m) {
                bf16x8_t a1, a2, a3;
                split8(af[2 * m], af[2 * m + 1], a1, a2, a3);
#pragma unroll
                for (int ct = 0; ct < NCG; ++ct) {
                    const __bf16 *wp = wb + ((g * NCG + ct) * 32 + lr) * p.ldw + b * KB + hf * KBH + 8 * m;
                    const bf16x8_t b1 = *reinterpret_cast<const bf16x8_t *>(wp);
                    const bf16x8_t b2 = *reinterpret_cast<const bf16x8_t *>(wp + plane);
                    const bf16x8_t b3 = *reinterpret_cast<const bf16x8_t *>(wp + 2 * plane);
                    magic::mfma_split6(acc[ct], a1, a2, a3, b1, b2, b3);
                }
            }
        };

        auto zero_acc = [&]() {
#pragma unroll
            for (int ct = 0; ct < NCG; ++ct)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[ct][r] = 0.f;
        };

        if (p.nkb == 1) {  // the row's K fits one block: load it once, sweep the column groups
            load_a<KQB>(a, rc, rok, hf * KBH, vec, af);
            if (EP) {  // the epilogue form has a single column group: the row's registers die here
                zero_acc();
                mfma_block(0, 0);
                store_group(0);
            } else {
                for (int g = 0; g < p.ngroups; ++g) {
                    zero_acc();
                    mfma_block(g, 0);
                    store_group(g);
                }
            }
        } else {  // one column group, K in blocks
            zero_acc();
            for (int b = 0; b < p.nkb; ++b) {
                load_a<KQB>(a, rc, rok, b * KB + hf * KBH, vec, af);
                mfma_block(0, b);
            }
            store_group(0);
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

struct Choice {
    int kqb, ncg;
    RowsPlan plan;
    size_t lds;
    int grid;
};

// float4 per lane half and K block the kernel is instantiated for
const int kKqb[] = {2, 4, 6, 8, 10, 12};  // even: one MFMA step takes 8 k from each lane half

// `mw`: number of output columns this launch covers, starting at a.col0 (the whole M, or one
// 64-column window of a wide epilogue launch)
bool choose(const ProjArgs &a, int mw, Choice &c) {
    if (a.n <= 0) return false;
    const bool vec = (a.F & 3) == 0 && aligned16(a.x, a.ld_x) && (!a.x_mask || (uintptr_t)a.x_mask % 4 == 0);
    if (!vec && a.F > 32) return false;  // element-wise operand loads exist for short rows only
    const bool ep = has_epilogue(a);
    if (ep) {
        // the vectorised epilogue: one column group of whole tiles, float4-aligned operands
        if (a.M != a.C || (mw != 32 && mw != 64) || a.col0 + mw > a.C) return false;
        if (!aligned16(a.hs, a.ld_hs)) return false;
        if (a.add && !aligned16(a.add, a.ld_add)) return false;
        if (a.add2 && !aligned16(a.add2, a.ld_add2)) return false;
        if (a.act_out && (!aligned16(a.act_out, a.ld_act) || !aligned16(a.act_pre, a.C))) return false;
        if (a.out_mask && (uintptr_t)a.out_mask % 4 != 0) return false;
        if (a.part_col && (uintptr_t)a.part_col % 16 != 0) return false;
    } else if (a.M > 32) {
        // two-tile column groups route whole float4s only: destinations must keep float4 alignment
        if ((a.C & 3) || (a.NS & 3) || (a.RC & 3) || !aligned16(a.hs, a.ld_hs)) return false;
        if (a.NS && ((uintptr_t)a.score % 16 != 0)) return false;
        if (a.RC && !aligned16(a.res, a.ld_res)) return false;
    }
    const int k8 = (a.F + 7) / 8;  // float4 per lane half over the whole K
    const int nkb = (k8 + 11) / 12;
    int kq = (k8 + nkb - 1) / nkb;
    int kqb = 0;
    for (int v : kKqb)
        if (v >= kq) {
            kqb = v;
            break;
        }
    if (!kqb) return false;
    const int tiles = (mw + 31) / 32;
    const int ncg = tiles >= 2 ? 2 : 1;
    const int ngroups = (tiles + ncg - 1) / ncg;
    if (nkb > 1 && ngroups > 1) return false;
    c.kqb = kqb;
    c.ncg = ncg;
    c.plan.nkb = nkb;
    c.plan.ngroups = ngroups;
    c.plan.ldw = nkb * kqb * 8 + 8;  // + 16 bytes: rows start in different banks
    const size_t w_bytes = (size_t)3 * ngroups * ncg * 32 * c.plan.ldw * 2;  // three bf16 planes
    const size_t strip = (size_t)SR * (ncg * 32 + 4) * sizeof(float);
    // block size that puts the most waves on a CU (160 KB of LDS, 16 waves at <= 128 VGPRs)
    const int max_waves = rows_max_waves(kqb);
    int best_wpb = 0, best_waves = 0;
    for (int wpb : {4, 8, 12, 16}) {
        if (wpb > max_waves) continue;
        const size_t lds = w_bytes + wpb * strip;
        if (lds > 160 * 1024) continue;
        int blocks = (int)((160 * 1024) / lds);
        if (blocks * wpb > max_waves) blocks = max_waves / wpb;
        if (blocks * wpb > best_waves) {
            best_waves = blocks * wpb;
            best_wpb = wpb;
        }
    }
    if (best_waves < 8) return false;
    c.plan.wpb = best_wpb;
    c.lds = w_bytes + best_wpb * strip;
    const int64_t tiles_n = (a.n + 31) / 32;
    const int64_t want = (tiles_n + best_wpb - 1) / best_wpb;
    const int64_t cap = 256LL * (best_waves / best_wpb);
    c.grid = (int)(want < cap ? want : cap);
    return true;
}

template <int KQB, int NCG, bool EP> int launch(const ProjArgs &a, const Choice &c, hipStream_t stream) {
    if (magic::ensure_dynamic_lds((const void *)k_proj_rows<KQB, NCG, EP>, c.lds) != MAGIC_OK) return MAGIC_ERR_RUNTIME;
    MAGIC_LAUNCH(a.name, (k_proj_rows<KQB, NCG, EP>), c.grid, c.plan.wpb * 64, c.lds, stream, a, c.plan);
    return magic::launch_status();
}

template <int KQB> int launch_kq(const ProjArgs &a, const Choice &c, bool ep, hipStream_t stream) {
    if (c.ncg == 2) return ep ? launch<KQB, 2, true>(a, c, stream) : launch<KQB, 2, false>(a, c, stream);
    return ep ? launch<KQB, 1, true>(a, c, stream) : launch<KQB, 1, false>(a, c, stream);
}

}  // namespace

namespace magic {

// Runs the projection on the row-owner kernel when the shape and the operand alignment allow it.
// Returns MAGIC_ERR_UNSUPPORTED otherwise (the caller falls back to the LDS-staged kernels).
// An epilogue launch wider than 64 columns runs as 64-column windows (each streams dY again:
// those shapes are MFMA-bound, not HBM-bound).  `n_part` receives the number of partial-sum rows
// one window writes (one per wave of the grid), `n_win` the number of windows: part_col is
// [n_part, C], part_prelu [n_win, n_part].  With `dry_run` nothing is launched.
int proj_strip_try(const ProjArgs &a0, int64_t *n_part, int *n_win, bool dry_run, hipStream_t stream);

int proj_rows_try(const ProjArgs &a0, int64_t *n_part, int *n_win, bool dry_run, hipStream_t stream) {
    // the streaming kernel (coalesced loads, tile turned into operands in LDS) takes the shapes it covers
    const int st_strip = proj_strip_try(a0, n_part, n_win, dry_run, stream);
    if (st_strip != MAGIC_ERR_UNSUPPORTED) return st_strip;
    ProjArgs a = a0;
    a.col0 = 0;
    const bool ep = has_epilogue(a);
    int windows = 1, mw = a.M;
    if (ep && a.M == a.C && a.C > 64) {
        if (a.C % 64) return MAGIC_ERR_UNSUPPORTED;
        windows = a.C / 64;
        mw = 64;
    }
    Choice c;
    if (!choose(a, mw, c)) return MAGIC_ERR_UNSUPPORTED;
    if (n_part) *n_part = (int64_t)c.grid * c.plan.wpb;
    if (n_win) *n_win = windows;
    if (dry_run) return MAGIC_OK;
    for (int wdx = 0; wdx < windows; ++wdx) {
        a.col0 = wdx * mw;
        if (a0.part_prelu) a.part_prelu = a0.part_prelu + (int64_t)wdx * c.grid * c.plan.wpb;
        int st;
        switch (c.kqb) {
            case 2: st = launch_kq<2>(a, c, ep, stream); break;
            case 4: st = launch_kq<4>(a, c, ep, stream); break;
            case 6: st = launch_kq<6>(a, c, ep, stream); break;
            case 8: st = launch_kq<8>(a, c, ep, stream); break;
            case 10: st = launch_kq<10>(a, c, ep, stream); break;
            case 12: st = launch_kq<12>(a, c, ep, stream); break;
            default: return MAGIC_ERR_UNSUPPORTED;
        }
        if (st != MAGIC_OK) return st;
    }
    return MAGIC_OK;
}

}  // namespace magic
