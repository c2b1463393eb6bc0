// Dense node projection  Y = (X (.) mask) . op(W)  in exact fp32 on the matrix cores
// (v_mfma_f32_32x32x2_f32: bit-equal to an fmaf chain over k), with a fused epilogue.
// These LDS-staged kernels are the fallback: shapes and alignments the row-owner kernel of
// proj_rows.hip (split-bf16 MFMA) accepts run there.
//
// Forward use: output columns are routed to three destinations -- the gathered operand hs
// (own buffer, 16-byte aligned rows), the 2H attention score columns, the residual projection.
// Backward use (dX = dY . W): the epilogue chains, per element of the hs columns,
//     v = acc  (+ add)  (* out_mask * out_scale)  (+ add2)  (* PReLU'(pre))  -> store
// and accumulates the column sums of the stored values (bias gradient) and the PReLU slope
// gradient, so that the gradient w.r.t. the previous layer's pre-activation leaves this kernel
// ready for the sparse backward sweeps.
//
// Operand map of the 32x32x2 MFMA (lane l): A[i = l&31][k = l>>5], B[k = l>>5][j = l&31],
// D[i = (r&3) + 8*(r>>2) + 4*(l>>5)][j = l&31].  Inside a 32-wide K chunk the K index is
// permuted so that lane half (l>>5) owns 16 contiguous k: A and B use the same permutation,
// which only reorders the k-sum.  All variants are HBM-bound (4N(F+M) bytes against 2NFM
// flops with F, M <= a few hundred): tiles are sized for streaming, not for reuse.
#include "proj_common.h"

namespace magic {
int proj_rows_try(const ProjArgs &a, int64_t *n_part, int *n_win, bool dry_run, hipStream_t stream);
int proj_gemm_try(const ProjArgs &a, void *planes_ws, int64_t *n_part, int *n_win, bool dry_run, hipStream_t stream,
                  bool relaxed = false);
size_t proj_gemm_workspace_bytes(int F, int M);
}

namespace {

// Store one 32x32 accumulator tile (rows t_row0 + map(r, hf), column c_lo + lr) through the
// epilogue; `csum` collects the column sum of the stored hs values, `dprelu` the slope gradient.
template <bool EP>
__device__ __forceinline__ void store_tile(const ProjArgs &a, const float16_t &acc, int64_t row0, int c_lo, int lr,
                                           int hf, float w, float &csum, float &dprelu) {
    const int col = c_lo + lr;
    const bool full_rows = row0 + 32 <= a.n;  // wave-uniform
    constexpr bool plain = !EP;
    if (full_rows && c_lo + 32 <= a.C) {  // whole tile in hs
        float *p = a.hs + (row0 + 4 * hf) * a.ld_hs + col;
        if (plain) {
#pragma unroll
            for (int r = 0; r < 16; ++r) p[((r & 3) + 8 * (r >> 2)) * a.ld_hs] = acc[r];
        } else {
            // four rows at a time: keeps the operand loads of one group in flight without
            // holding all 16 x 4 of them in registers
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = chain(a, acc[4 * g + q], row0 + 4 * hf + q + 8 * g, col, w, dprelu);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    p[(q + 8 * g) * a.ld_hs] = v[q];
                    csum += v[q];
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        return;
    }
    if (full_rows && c_lo >= a.C + a.NS && c_lo + 32 <= a.M) {  // whole tile in res
        float *p = a.res + (row0 + 4 * hf) * a.ld_res + (col - a.C - a.NS);
#pragma unroll
        for (int r = 0; r < 16; ++r) p[((r & 3) + 8 * (r >> 2)) * a.ld_res] = acc[r];
        return;
    }
    const int c_hi = c_lo + 32 < a.M ? c_lo + 32 : a.M;
    if (full_rows && c_lo >= a.C && c_hi <= a.C + a.NS) {  // tile holds score columns only
        if (col < c_hi) {
            float *p = a.score + (row0 + 4 * hf) * a.NS + (col - a.C);
#pragma unroll
            for (int r = 0; r < 16; ++r) p[((r & 3) + 8 * (r >> 2)) * a.NS] = acc[r];
        }
        return;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int64_t row = row0 + (r & 3) + 8 * (r >> 2) + 4 * hf;
        if (row >= a.n) continue;
        if (col < a.C) {
            const float v = plain ? acc[r] : chain(a, acc[r], row, col, w, dprelu);
            a.hs[row * a.ld_hs + col] = v;
            csum += v;
        } else if (col < a.C + a.NS) {
            a.score[row * a.NS + (col - a.C)] = acc[r];
        } else if (col < a.M) {
            a.res[row * a.ld_res + (col - a.C - a.NS)] = acc[r];
        }
    }
}

// per-wave partial sums -> workspace rows (one row per wave of the grid)
template <int NCT>
__device__ __forceinline__ void flush_sums(const ProjArgs &a, const float (&csum)[NCT], float dprelu, int lane) {
    if (!a.part_col) return;
    const int64_t prow = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
        float v = csum[ct] + __shfl_xor(csum[ct], 32, 64);  // the two lane halves hold the same column
        const int col = a.col0 + ct * 32 + (lane & 31);
        if (lane < 32 && col < a.C) a.part_col[prow * a.C + col] = v;
    }
    if (a.part_prelu) {
        float v = dprelu;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
        if (lane == 0) a.part_prelu[prow] = v;
    }
}

// ---- block-tile variant: 4 waves = 128 rows x all columns, K chunks staged through LDS -------
template <int NCT, bool EP>
__global__ __launch_bounds__(256) void k_proj(ProjArgs a) {
    extern __shared__ __align__(16) float sm[];
    float *xs = sm;                // [BM][LDS_LD]
    float *ws = sm + BM * LDS_LD;  // [NCT*32][LDS_LD]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hf = lane >> 5, lr = lane & 31;
    const int64_t row0 = (int64_t)blockIdx.x * BM;
    float16_t acc[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[ct][r] = 0.f;
    const bool vec = (a.F % 4 == 0) && (a.ld_x % 4 == 0) && ((uintptr_t)a.x % 16 == 0);
    const bool wvec = (a.F % 4 == 0) && a.w_cs == 1 && (a.w_rs % 4 == 0) && ((uintptr_t)a.w % 16 == 0);
    for (int k0 = 0; k0 < a.F; k0 += KC) {
        __syncthreads();
        for (int idx = tid; idx < BM * (KC / 4); idx += 256) {
            const int r = idx / (KC / 4), c4 = (idx % (KC / 4)) * 4;
            *reinterpret_cast<float4 *>(xs + r * LDS_LD + c4) = load_x4(a, row0 + r, k0 + c4, vec);
        }
        for (int idx = tid; idx < NCT * 32 * (KC / 4); idx += 256) {
            const int r = idx / (KC / 4), c4 = (idx % (KC / 4)) * 4;
            *reinterpret_cast<float4 *>(ws + r * LDS_LD + c4) = load_w4(a, r, k0 + c4, wvec);
        }
        __syncthreads();
        float af[16];
        {
            const float *p = xs + (wave * 32 + lr) * LDS_LD + hf * 16;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float4 v = *reinterpret_cast<const float4 *>(p + 4 * j);
                af[4 * j] = v.x;
                af[4 * j + 1] = v.y;
                af[4 * j + 2] = v.z;
                af[4 * j + 3] = v.w;
            }
        }
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
            float bf[16];
            const float *p = ws + (ct * 32 + lr) * LDS_LD + hf * 16;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float4 v = *reinterpret_cast<const float4 *>(p + 4 * j);
                bf[4 * j] = v.x;
                bf[4 * j + 1] = v.y;
                bf[4 * j + 2] = v.z;
                bf[4 * j + 3] = v.w;
            }
#pragma unroll
            for (int s = 0; s < 16; ++s) acc[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[s], bf[s], acc[ct], 0, 0, 0);
        }
    }
    const float w = a.prelu_w ? a.prelu_w[0] : 1.f;
    float csum[NCT];
    float dprelu = 0.f;
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
        csum[ct] = 0.f;
        store_tile<EP>(a, acc[ct], row0 + wave * 32, a.col0 + ct * 32, lr, hf, w, csum[ct], dprelu);
    }
    if (EP) flush_sums<NCT>(a, csum, dprelu, lane);
}

// ---- streaming variant: the whole weight block stays in LDS; the four waves are independent,
// each walks 32-row tiles, copies its tile through a private LDS strip with coalesced 16-byte
// loads (the next tile's loads are issued before the current tile's MFMAs) and never meets a
// block barrier after the weight staging.
// LDS floats per wave strip: the staged input tile, reused for the transposed output tile
constexpr int stream_strip(int nct, int kch, int mode) {
    return 32 * ((mode != 0 && nct * 32 > kch * 32 ? nct * 32 : kch * 32) + 4);
}
constexpr size_t stream_lds_c(int nct, int kch, int mode) {
    return ((size_t)nct * 32 * (kch * 32 + 4) + 4 * (size_t)stream_strip(nct, kch, mode)) * sizeof(float);
}
// waves per SIMD the register allocator must leave room for: as many blocks as the LDS footprint allows
constexpr int stream_waves(int nct, int kch, int mode) { return stream_lds_c(nct, kch, mode) <= 80 * 1024 ? 2 : 1; }

// MODE 0: accumulator tiles stored straight from the MFMA layout, columns routed to hs / score /
//         res (forward projections).
// MODE 1: all columns go to hs (M == C, C = NCT*32 ... see dispatch): the tile is transposed through the
//         wave's LDS strip and stored as whole rows of float4.
// MODE 2: MODE 1 + the epilogue chain and its column / slope sums.
template <int NCT, int KCH, int MODE>
__global__ __launch_bounds__(256, stream_waves(NCT, KCH, MODE)) void k_proj_stream(ProjArgs a) {
    constexpr int KP = KCH * 32;
    constexpr int LDW = KP + 4;  // 16-B aligned rows, 4-bank skew between rows
    constexpr int NV = KP / 8;   // float4 per lane per tile
    constexpr int STRIP = stream_strip(NCT, KCH, MODE);
    extern __shared__ __align__(16) float sm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hf = lane >> 5, lr = lane & 31;
    float *ws = sm;
    float *xs = sm + NCT * 32 * LDW + wave * STRIP;
    const bool wvec = (a.F % 4 == 0) && a.w_cs == 1 && (a.w_rs % 4 == 0) && ((uintptr_t)a.w % 16 == 0);
    for (int idx = tid; idx < NCT * 32 * (KP / 4); idx += 256) {
        const int r = idx / (KP / 4), c4 = (idx % (KP / 4)) * 4;
        *reinterpret_cast<float4 *>(ws + r * LDW + c4) = load_w4(a, r, c4, wvec);
    }
    __syncthreads();
    const bool vec = (a.F % 4 == 0) && (a.ld_x % 4 == 0) && ((uintptr_t)a.x % 16 == 0);
    const int64_t n_tiles = (a.n + 31) / 32;
    const int64_t t_stride = (int64_t)gridDim.x * 4;
    const float w = a.prelu_w ? a.prelu_w[0] : 1.f;
    const float inv_w = w > 0.f ? 1.f / w : 0.f;
    float csum[NCT];  // MODE 0 only (unused there, kept for store_tile's signature)
    float4 cs4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float dprelu = 0.f;
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) csum[ct] = 0.f;
    float4 v[NV];
    auto load_tile = [&](int64_t t) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = lane + 64 * i;
            const int r = idx / (KP / 4), c4 = (idx % (KP / 4)) * 4;
            v[i] = load_x4(a, t * 32 + r, c4, vec);
        }
    };
    int64_t t = (int64_t)blockIdx.x * 4 + wave;
    if (t < n_tiles) load_tile(t);
    for (; t < n_tiles; t += t_stride) {
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = lane + 64 * i;
            const int r = idx / (KP / 4), c4 = (idx % (KP / 4)) * 4;
            *reinterpret_cast<float4 *>(xs + r * LDW + c4) = v[i];
        }
        __builtin_amdgcn_wave_barrier();
        if (t + t_stride < n_tiles) load_tile(t + t_stride);  // in flight behind the MFMAs below
        float16_t acc[NCT];
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ct][r] = 0.f;
#pragma unroll
        for (int ch = 0; ch < KCH; ++ch) {
            float af[16];
            {
                const float *p = xs + lr * LDW + ch * 32 + hf * 16;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float4 q = *reinterpret_cast<const float4 *>(p + 4 * j);
                    af[4 * j] = q.x;
                    af[4 * j + 1] = q.y;
                    af[4 * j + 2] = q.z;
                    af[4 * j + 3] = q.w;
                }
            }
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                float bf[16];
                const float *p = ws + (ct * 32 + lr) * LDW + ch * 32 + hf * 16;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float4 q = *reinterpret_cast<const float4 *>(p + 4 * j);
                    bf[4 * j] = q.x;
                    bf[4 * j + 1] = q.y;
                    bf[4 * j + 2] = q.z;
                    bf[4 * j + 3] = q.w;
                }
#pragma unroll
                for (int s = 0; s < 16; ++s)
                    acc[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[s], bf[s], acc[ct], 0, 0, 0);
            }
        }
        if (MODE == 0) {
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct)
                store_tile<false>(a, acc[ct], t * 32, a.col0 + ct * 32, lr, hf, w, csum[ct], dprelu);
        } else {
            // the strip's operands are consumed (LDS operations of a wave complete in order)
            constexpr int LDO = NCT * 32 + 4;
            constexpr int C4 = NCT * 8;  // float4 per output row
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    xs[((r & 3) + 8 * (r >> 2) + 4 * hf) * LDO + ct * 32 + lr] = acc[ct][r];
            __builtin_amdgcn_wave_barrier();
#pragma unroll 4
            for (int i = 0; i < 32 * C4 / 64; ++i) {
                const int idx = lane + 64 * i;
                const int r = idx / C4, c4 = (idx % C4) * 4;  // c4 is the same for every i (C4 divides 64)
                const float4 q = *reinterpret_cast<const float4 *>(xs + r * LDO + c4);
                epi_store4<MODE == 2>(a, q, t * 32 + r, a.col0 + c4, w, inv_w, cs4, dprelu);
            }
        }
    }
    if (MODE == 2 && a.part_col) {
        constexpr int C4 = NCT * 8;
        // lanes l, l + C4, l + 2*C4, ... hold the same four columns
#pragma unroll
        for (int off = C4; off < 64; off <<= 1) {
            cs4.x += __shfl_xor(cs4.x, off, 64);
            cs4.y += __shfl_xor(cs4.y, off, 64);
            cs4.z += __shfl_xor(cs4.z, off, 64);
            cs4.w += __shfl_xor(cs4.w, off, 64);
        }
        const int64_t prow = (int64_t)blockIdx.x * 4 + wave;
        if (lane < C4) *reinterpret_cast<float4 *>(a.part_col + prow * a.C + lane * 4) = cs4;
        if (a.part_prelu) {
            float s = dprelu;
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
            if (lane == 0) a.part_prelu[prow] = s;
        }
    }
}

// fixed-order reduction of the per-wave partial rows: block c < C sums column c, block C the
// PReLU partials (256 threads take rows round robin, then a tree)
__global__ __launch_bounds__(256) void k_proj_sums(const float *part_col, const float *part_prelu, int n_part,
                                                   int n_prelu, int C, float *col_sum, float *prelu_grad) {
    __shared__ float red[256];
    const int c = blockIdx.x;
    float t = 0.f;
    if (c < C) {
        for (int k = threadIdx.x; k < n_part; k += 256) t += part_col[(int64_t)k * C + c];
    } else {
        for (int k = threadIdx.x; k < n_prelu; k += 256) t += part_prelu[k];
    }
    red[threadIdx.x] = t;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if (c < C) {
            if (col_sum) col_sum[c] = red[0];
        } else if (prelu_grad) {
            prelu_grad[0] = red[0];
        }
    }
}

// row-major float4 stores (and epilogue operand loads) are possible
bool rows_vectorizable(const ProjArgs &a, int nct) {
    if (a.M != a.C || a.C != nct * 32 || a.col0 != 0) return false;
    if (nct != 1 && nct != 2 && nct != 4 && nct != 8) return false;
    if (!aligned16(a.hs, a.ld_hs)) return false;
    if (a.add && !aligned16(a.add, a.ld_add)) return false;
    if (a.add2 && !aligned16(a.add2, a.ld_add2)) return false;
    if (a.act_out && (!aligned16(a.act_out, a.ld_act) || !aligned16(a.act_pre, a.C))) return false;
    if (a.out_mask && (uintptr_t)a.out_mask % 4 != 0) return false;
    if (a.part_col && (uintptr_t)a.part_col % 16 != 0) return false;
    return true;
}

size_t stream_lds(int nct, int kch, int mode) { return stream_lds_c(nct, kch, mode); }

int stream_mode(const ProjArgs &a, int nct) {
    const int kch = (a.F + 31) / 32;
    if (rows_vectorizable(a, nct) && stream_lds(nct, kch, 2) <= 160 * 1024) return has_epilogue(a) ? 2 : 1;
    return has_epilogue(a) ? -1 : 0;  // -1: the block-tile kernel carries the element-wise epilogue
}

int stream_grid(const ProjArgs &a, int nct, int kch, int mode) {
    size_t sh = stream_lds(nct, kch, mode);
    int64_t tiles = (a.n + 31) / 32;
    int blocks_per_cu = sh <= 40 * 1024 ? 4 : (sh <= 80 * 1024 ? 2 : 1);
    int64_t want = (tiles + 3) / 4;
    int64_t cap = 256 * blocks_per_cu;
    return (int)(want < cap ? want : cap);
}

template <int NCT, int KCH, int MODE> int launch_stream_mode(const ProjArgs &a, hipStream_t stream) {
    size_t sh = stream_lds(NCT, KCH, MODE);
    if (magic::ensure_dynamic_lds((const void *)k_proj_stream<NCT, KCH, MODE>, sh) != MAGIC_OK) return MAGIC_ERR_RUNTIME;
    MAGIC_LAUNCH(a.name, (k_proj_stream<NCT, KCH, MODE>), stream_grid(a, NCT, KCH, MODE), 256, sh, stream, a);
    return magic::launch_status();
}

template <int NCT, int KCH> int launch_stream(const ProjArgs &a, int mode, hipStream_t stream) {
    if constexpr ((NCT == 1 || NCT == 2 || NCT == 4 || NCT == 8) && stream_lds_c(NCT, KCH, 2) <= 160 * 1024) {
        if (mode == 2) return launch_stream_mode<NCT, KCH, 2>(a, stream);
        if (mode == 1) return launch_stream_mode<NCT, KCH, 1>(a, stream);
    }
    return launch_stream_mode<NCT, KCH, 0>(a, stream);
}

// (column tiles, 32-wide K chunks) pairs with a streaming instantiation
#define MAGIC_PROJ_STREAM_SHAPES(X) \
    X(5, 1) X(3, 2) X(2, 6) X(1, 2) X(2, 1) X(6, 2) X(2, 3) X(1, 5) X(1, 1) X(2, 2) X(3, 1) X(4, 1) X(8, 1)

bool has_stream(int nct, int kch) {
#define X(n, k) \
    if (nct == n && kch == k) return true;
    MAGIC_PROJ_STREAM_SHAPES(X)
#undef X
    return false;
}

template <int NCT, bool EP> int launch_block_ep(const ProjArgs &a, hipStream_t stream) {
    size_t sh = (size_t)(BM + NCT * 32) * LDS_LD * sizeof(float);
    unsigned grid = (unsigned)((a.n + BM - 1) / BM);
    if (magic::ensure_dynamic_lds((const void *)k_proj<NCT, EP>, sh) != MAGIC_OK) return MAGIC_ERR_RUNTIME;
    MAGIC_LAUNCH(a.name, (k_proj<NCT, EP>), grid, 256, sh, stream, a);
    return magic::launch_status();
}

template <int NCT> int launch_block(const ProjArgs &a, hipStream_t stream) {
    return has_epilogue(a) ? launch_block_ep<NCT, true>(a, stream) : launch_block_ep<NCT, false>(a, stream);
}

int dispatch_one(const ProjArgs &a, int nct, hipStream_t stream) {
    const int kch = (a.F + 31) / 32;
    const int mode = stream_mode(a, nct);
    if (mode >= 0) {
#define X(n, k) \
    if (nct == n && kch == k) return launch_stream<n, k>(a, mode, stream);
        MAGIC_PROJ_STREAM_SHAPES(X)
#undef X
    }
    switch (nct) {
        case 1: return launch_block<1>(a, stream);
        case 2: return launch_block<2>(a, stream);
        case 3: return launch_block<3>(a, stream);
        case 4: return launch_block<4>(a, stream);
        case 5: return launch_block<5>(a, stream);
        case 6: return launch_block<6>(a, stream);
        case 7: return launch_block<7>(a, stream);
        case 8: return launch_block<8>(a, stream);
        case 9: return launch_block<9>(a, stream);
        case 10: return launch_block<10>(a, stream);
        default: return MAGIC_ERR_UNSUPPORTED;
    }
}

// number of partial rows (= waves of the grid) one launch over `nct` column tiles writes
int64_t n_partials(const ProjArgs &a, int nct) {
    const int kch = (a.F + 31) / 32;
    const int mode = stream_mode(a, nct);
    if (mode >= 0 && has_stream(nct, kch)) return (int64_t)stream_grid(a, nct, kch, mode) * 4;
    return ((a.n + BM - 1) / BM) * 4;
}

// up to 10 column tiles (320 outputs) per launch; wider outputs take several launches of 8
int dispatch(const ProjArgs &a0, void *gemm_ws, hipStream_t stream) {
    ProjArgs a = a0;
    a.col0 = 0;
    const int st_gemm = magic::proj_gemm_try(a, gemm_ws, nullptr, nullptr, false, stream);
    if (st_gemm != MAGIC_ERR_UNSUPPORTED) return st_gemm;
    const int st_rows = magic::proj_rows_try(a, nullptr, nullptr, false, stream);
    if (st_rows != MAGIC_ERR_UNSUPPORTED) return st_rows;
    const int st_relaxed = magic::proj_gemm_try(a, gemm_ws, nullptr, nullptr, false, stream, true);
    if (st_relaxed != MAGIC_ERR_UNSUPPORTED) return st_relaxed;
    const int tiles = (a.M + 31) / 32;
    if (tiles <= 10) {
        a.col0 = 0;
        return dispatch_one(a, tiles, stream);
    }
    for (int t0 = 0; t0 < tiles; t0 += 8) {
        a.col0 = t0 * 32;
        int st = dispatch_one(a, tiles - t0 < 8 ? tiles - t0 : 8, stream);
        if (st != MAGIC_OK) return st;
    }
    return MAGIC_OK;
}

void init_args(ProjArgs &a) {
    a.x_mask = nullptr;
    a.x_scale = 1.f;
    a.col0 = 0;
    a.score = nullptr;
    a.res = nullptr;
    a.ld_res = 0;
    a.NS = 0;
    a.RC = 0;
    a.add = nullptr;
    a.ld_add = 0;
    a.out_mask = nullptr;
    a.out_scale = 1.f;
    a.add2 = nullptr;
    a.ld_add2 = 0;
    a.act_out = nullptr;
    a.ld_act = 0;
    a.act_pre = nullptr;
    a.prelu_w = nullptr;
    a.part_col = nullptr;
    a.part_prelu = nullptr;
    a.sc_src = a.sc_dst = nullptr;
}

}  // namespace

extern "C" {

static int proj_fwd_run(int64_t n_rows, int32_t in_dim, int32_t out_c, int32_t n_score, int32_t res_c, const float *x,
                        int64_t ld_x, const uint8_t *x_mask, float x_scale, const float *w, float *hs, int64_t ld_hs,
                        float *score, float *res, int64_t ld_res, const float *attn_src, const float *attn_dst,
                        void *workspace, size_t workspace_bytes, void *stream_) {
    if (n_rows < 0 || in_dim <= 0 || out_c <= 0 || n_score < 0 || res_c < 0) return MAGIC_ERR_INVALID_ARG;
    if (n_rows == 0) return MAGIC_OK;
    if (!x || !w || !hs || ld_x < in_dim || ld_hs < out_c) return MAGIC_ERR_INVALID_ARG;
    if ((n_score > 0 && !score) || (res_c > 0 && (!res || ld_res < res_c))) return MAGIC_ERR_INVALID_ARG;
    ProjArgs a;
    init_args(a);
    a.n = n_rows;
    a.F = in_dim;
    a.C = out_c;
    a.NS = n_score;
    a.RC = res_c;
    a.M = out_c + n_score + res_c;
    a.x = x;
    a.ld_x = ld_x;
    a.x_mask = x_mask;
    a.x_scale = x_scale;
    a.w = w;
    a.w_rs = in_dim;
    a.w_cs = 1;
    a.name = magic::prof_on() ? magic::prof_name("proj_fwd", in_dim, a.M) : "proj_fwd";
    a.hs = hs;
    a.ld_hs = ld_hs;
    a.score = score;
    a.res = res;
    a.ld_res = ld_res;
    // the block-tiled kernel keeps the split weight in the workspace; without one the other kernels run
    const size_t need = magic::proj_gemm_workspace_bytes(a.F, a.M);
    void *gemm_ws = (need && workspace && workspace_bytes >= need) ? workspace : nullptr;
    // Wide layers with 64-wide heads and no residual projection: the 2H score columns would cost the block-tiled
    // kernel a third, nearly idle column block (264 outputs: 122 us against 85 us for 256 at 96,000 rows); with the
    // attention vectors given, its epilogue forms them from the 64 outputs of a head, which one wavefront holds.
    if (attn_src && attn_dst && gemm_ws && res_c == 0 && n_score > 0 && out_c == 32 * n_score &&
        (((uintptr_t)attn_src | (uintptr_t)attn_dst) & 15) == 0 && (ld_hs & 3) == 0 && ((uintptr_t)hs & 15) == 0) {
        ProjArgs b = a;
        b.M = out_c;
        b.sc_src = attn_src;
        b.sc_dst = attn_dst;
        const int st = magic::proj_gemm_try(b, gemm_ws, nullptr, nullptr, false, (hipStream_t)stream_);
        if (st != MAGIC_ERR_UNSUPPORTED) return st;
    }
    return dispatch(a, gemm_ws, (hipStream_t)stream_);
}

int magic_proj_fwd(int64_t n_rows, int32_t in_dim, int32_t out_c, int32_t n_score, int32_t res_c, const float *x,
                   int64_t ld_x, const uint8_t *x_mask, float x_scale, const float *w, float *hs, int64_t ld_hs,
                   float *score, float *res, int64_t ld_res, void *workspace, size_t workspace_bytes, void *stream_) {
    return proj_fwd_run(n_rows, in_dim, out_c, n_score, res_c, x, ld_x, x_mask, x_scale, w, hs, ld_hs, score, res, ld_res,
                        nullptr, nullptr, workspace, workspace_bytes, stream_);
}

int magic_proj_fwd_attn(int64_t n_rows, int32_t in_dim, int32_t out_c, int32_t n_score, int32_t res_c, const float *x,
                        int64_t ld_x, const uint8_t *x_mask, float x_scale, const float *w, float *hs, int64_t ld_hs,
                        float *score, float *res, int64_t ld_res, const float *attn_src, const float *attn_dst,
                        void *workspace, size_t workspace_bytes, void *stream_) {
    return proj_fwd_run(n_rows, in_dim, out_c, n_score, res_c, x, ld_x, x_mask, x_scale, w, hs, ld_hs, score, res, ld_res,
                        attn_src, attn_dst, workspace, workspace_bytes, stream_);
}

size_t magic_proj_fwd_workspace_bytes(int32_t in_dim, int32_t out_total) {
    if (in_dim <= 0 || out_total <= 0) return 256;
    return magic::proj_gemm_workspace_bytes(in_dim, out_total) + 256;
}

size_t magic_proj_bwd_input_workspace_bytes(int64_t n_rows, int32_t k_dim, int32_t out_dim) {
    if (n_rows <= 0 || k_dim <= 0 || out_dim <= 0) return 256;
    // upper bound over the kernel variants: one partial row per wave of the grid
    int64_t np = ((n_rows + BM - 1) / BM) * 4;
    if (np < 4096) np = 4096;
    const size_t windows = (size_t)(out_dim + 63) / 64;  // part_prelu holds one row per 64-column window
    return magic::align_up((size_t)np * out_dim * sizeof(float)) + magic::align_up((size_t)np * windows * sizeof(float)) +
           magic::proj_gemm_workspace_bytes(k_dim, out_dim) + 256;
}

int magic_proj_bwd_input(int64_t n_rows, int32_t k_dim, int32_t out_dim, const float *dy, int64_t ld_dy,
                         const float *w, const magic_proj_epilogue *ep, float *dx, int64_t ld_dx, void *workspace,
                         size_t workspace_bytes, void *stream_) {
    if (n_rows < 0 || k_dim <= 0 || out_dim <= 0) return MAGIC_ERR_INVALID_ARG;
    if (n_rows == 0) return MAGIC_OK;
    if (!dy || !w || !dx || ld_dy < k_dim || ld_dx < out_dim) return MAGIC_ERR_INVALID_ARG;
    hipStream_t stream = (hipStream_t)stream_;
    ProjArgs a;
    init_args(a);
    a.n = n_rows;
    a.F = k_dim;  // the reduction runs over the columns of dY
    a.C = a.M = out_dim;
    a.x = dy;
    a.ld_x = ld_dy;
    a.w = w;  // [k_dim, out_dim] row-major, read transposed
    a.w_rs = 1;
    a.w_cs = out_dim;
    a.name = magic::prof_on() ? magic::prof_name("proj_bwd_input", k_dim, out_dim) : "proj_bwd_input";
    a.hs = dx;
    a.ld_hs = ld_dx;
    bool sums = false;
    if (ep) {
        if ((ep->add && ep->ld_add < out_dim) || (ep->add2 && ep->ld_add2 < out_dim)) return MAGIC_ERR_INVALID_ARG;
        if (ep->act_out && (!ep->prelu_w || !ep->act_pre || ep->ld_act < out_dim)) return MAGIC_ERR_INVALID_ARG;
        a.add = ep->add;
        a.ld_add = ep->ld_add;
        a.out_mask = ep->out_mask;
        a.out_scale = ep->out_scale;
        a.add2 = ep->add2;
        a.ld_add2 = ep->ld_add2;
        a.act_out = ep->act_out;
        a.ld_act = ep->ld_act;
        a.act_pre = ep->act_pre;
        a.prelu_w = ep->act_out ? ep->prelu_w : nullptr;
        sums = ep->col_sum != nullptr || ep->prelu_grad != nullptr;
    }
    int64_t np = 0;
    int n_win = 1;
    magic::Arena ar(workspace, workspace_bytes);
    void *gemm_ws = nullptr;
    const size_t need = magic::proj_gemm_workspace_bytes(a.F, a.M);
    if (need && ar.fits(need)) gemm_ws = ar.take<unsigned char>(need);
    if (sums) {
        const int tiles = (out_dim + 31) / 32;
        a.part_col = static_cast<float *>(workspace);  // provisional: the kernel choice looks at it
        if ((!gemm_ws || magic::proj_gemm_try(a, gemm_ws, &np, &n_win, true, stream) != MAGIC_OK) &&
            magic::proj_rows_try(a, &np, &n_win, true, stream) != MAGIC_OK) {
            if (tiles > 10) return MAGIC_ERR_UNSUPPORTED;  // column sums need a single launch there
            np = n_partials(a, tiles);
            n_win = 1;
        }
        a.part_col = ar.take<float>((size_t)np * out_dim);
        a.part_prelu = ar.take<float>((size_t)np * n_win);
        if (!ar.ok() || !a.part_col || !a.part_prelu) return MAGIC_ERR_WORKSPACE;
    }
    int st = dispatch(a, gemm_ws, stream);
    if (st != MAGIC_OK) return st;
    if (sums) {
        MAGIC_LAUNCH("proj_bwd_input_sums", k_proj_sums, out_dim + 1, 256, 0, stream, a.part_col, a.part_prelu, (int)np,
                     (int)np * n_win, out_dim, ep->col_sum, ep->prelu_grad);
    }
    return magic::launch_status();
}

}  // extern "C"
