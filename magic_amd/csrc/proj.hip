// Dense node projection  Y = X . W^T  in exact fp32 on the matrix cores
// (v_mfma_f32_32x32x2_f32: bit-equal to an fmaf chain over k), with the output columns
// routed to three destinations: the gathered operand hs (kept in its own 128-byte aligned
// rows), the 2H attention score columns, and the residual projection.
//
// Block = 4 waves = 128 input rows x all M output columns; K is walked in chunks of 32
// staged through LDS.  Operand map of the 32x32x2 MFMA (lane l): A[i = l&31][k = l>>5],
// B[k = l>>5][j = l&31], D[i = (r&3) + 8*(r>>2) + 4*(l>>5)][j = l&31].  Inside a chunk the
// K index is permuted so that lane half (l>>5) owns 16 contiguous k: A and B use the same
// permutation, which only reorders the k-sum.  The kernel is HBM-bound (N*(F+M)*4 bytes
// against 2*N*F*M flops with F, M <= 264), so the tile is sized for streaming, not reuse.
#include "common.h"

namespace {

typedef float float16_t __attribute__((ext_vector_type(16)));

constexpr int BM = 128;   // rows per block
constexpr int KC = 32;    // k per LDS chunk
constexpr int LDS_LD = 36;  // floats per staged row (144 B)

struct ProjArgs {
    int64_t n;
    int F, M, C, NS, RC;
    const float *x;
    int64_t ld_x;
    const float *w;
    int64_t w_rs, w_cs;  // element (row r, k) of the weight = w[r * w_rs + k * w_cs]
    int col0;            // first output column of this launch (wide outputs take several)
    const float *add;    // optional [N, ld_add] added to the hs columns
    int64_t ld_add;
    const char *name;
    float *hs;
    int64_t ld_hs;
    float *score;
    float *res;
    int64_t ld_res;
};

template <int NCT>  // number of 32-column tiles
__global__ __launch_bounds__(256) void k_proj(ProjArgs a) {
    extern __shared__ __align__(16) float sm[];
    float *xs = sm;                  // [BM][LDS_LD]
    float *ws = sm + BM * LDS_LD;    // [NCT*32][LDS_LD]
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
        // stage X chunk: BM rows x KC floats
        for (int idx = tid; idx < BM * (KC / 4); idx += 256) {
            int r = idx / (KC / 4), c4 = (idx % (KC / 4)) * 4;
            int64_t row = row0 + r;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < a.n) {
                const float *p = a.x + row * a.ld_x + k0 + c4;
                if (vec && k0 + c4 + 3 < a.F) {
                    v = *reinterpret_cast<const float4 *>(p);
                } else {
                    if (k0 + c4 + 0 < a.F) v.x = p[0];
                    if (k0 + c4 + 1 < a.F) v.y = p[1];
                    if (k0 + c4 + 2 < a.F) v.z = p[2];
                    if (k0 + c4 + 3 < a.F) v.w = p[3];
                }
            }
            *reinterpret_cast<float4 *>(xs + r * LDS_LD + c4) = v;
        }
        // stage W chunk: NCT*32 output channels x KC floats
        for (int idx = tid; idx < NCT * 32 * (KC / 4); idx += 256) {
            int r = idx / (KC / 4), c4 = (idx % (KC / 4)) * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (a.col0 + r < a.M) {
                const float *p = a.w + (int64_t)(a.col0 + r) * a.w_rs + (int64_t)(k0 + c4) * a.w_cs;
                if (wvec && k0 + c4 + 3 < a.F) {
                    v = *reinterpret_cast<const float4 *>(p);
                } else {
                    if (k0 + c4 + 0 < a.F) v.x = p[0];
                    if (k0 + c4 + 1 < a.F) v.y = p[a.w_cs];
                    if (k0 + c4 + 2 < a.F) v.z = p[2 * a.w_cs];
                    if (k0 + c4 + 3 < a.F) v.w = p[3 * a.w_cs];
                }
            }
            *reinterpret_cast<float4 *>(ws + r * LDS_LD + c4) = v;
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
    // epilogue: route columns
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
        const int col = a.col0 + ct * 32 + lr;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int64_t row = row0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * hf;
            if (row >= a.n) continue;
            float v = acc[ct][r];
            if (col < a.C) {
                if (a.add) v += a.add[row * a.ld_add + col];
                a.hs[row * a.ld_hs + col] = v;
            } else if (col < a.C + a.NS) {
                a.score[row * a.NS + (col - a.C)] = v;
            } else if (col < a.M) {
                a.res[row * a.ld_res + (col - a.C - a.NS)] = v;
            }
        }
    }
}

// Streaming variant for K <= 32*KCH small enough that the whole weight block fits in LDS:
// W is staged once per block; afterwards the four waves are independent -- each walks 32-row
// tiles, copies its tile through a private LDS strip with coalesced 16-byte loads (the next
// tile's loads are issued before the current tile's MFMAs), and never meets a block barrier.
template <int NCT, int KCH>
__global__ __launch_bounds__(256) void k_proj_stream(ProjArgs a) {
    constexpr int KP = KCH * 32;
    constexpr int LDW = KP + 4;          // 16-B aligned rows, 4-bank skew between rows
    constexpr int NV = KP / 8;           // float4 per lane per tile
    extern __shared__ __align__(16) float sm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hf = lane >> 5, lr = lane & 31;
    float *ws = sm;
    float *xs = sm + NCT * 32 * LDW + wave * 32 * LDW;
    const bool wvec = (a.F % 4 == 0) && a.w_cs == 1 && (a.w_rs % 4 == 0) && ((uintptr_t)a.w % 16 == 0);
    for (int idx = tid; idx < NCT * 32 * (KP / 4); idx += 256) {
        const int r = idx / (KP / 4), c4 = (idx % (KP / 4)) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (a.col0 + r < a.M) {
            const float *p = a.w + (int64_t)(a.col0 + r) * a.w_rs + (int64_t)c4 * a.w_cs;
            if (wvec && c4 + 3 < a.F) {
                v = *reinterpret_cast<const float4 *>(p);
            } else {
                if (c4 + 0 < a.F) v.x = p[0];
                if (c4 + 1 < a.F) v.y = p[a.w_cs];
                if (c4 + 2 < a.F) v.z = p[2 * a.w_cs];
                if (c4 + 3 < a.F) v.w = p[3 * a.w_cs];
            }
        }
        *reinterpret_cast<float4 *>(ws + r * LDW + c4) = v;
    }
    __syncthreads();
    const bool vec = (a.F % 4 == 0) && (a.ld_x % 4 == 0) && ((uintptr_t)a.x % 16 == 0);
    const int64_t n_tiles = (a.n + 31) / 32;
    const int64_t t_stride = (int64_t)gridDim.x * 4;
    float4 v[NV];
    auto load_tile = [&](int64_t t) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = lane + 64 * i;
            const int r = idx / (KP / 4), c4 = (idx % (KP / 4)) * 4;
            const int64_t row = t * 32 + r;
            float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < a.n && c4 < a.F) {
                const float *p = a.x + row * a.ld_x + c4;
                if (vec && c4 + 3 < a.F) {
                    q = *reinterpret_cast<const float4 *>(p);
                } else {
                    q.x = p[0];
                    if (c4 + 1 < a.F) q.y = p[1];
                    if (c4 + 2 < a.F) q.z = p[2];
                    if (c4 + 3 < a.F) q.w = p[3];
                }
            }
            v[i] = q;
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
        if (t + t_stride < n_tiles) load_tile(t + t_stride);   // in flight behind the MFMAs below
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
        const bool full_rows = t * 32 + 32 <= a.n;   // wave-uniform
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
            const int c_lo = a.col0 + ct * 32;
            const int col = c_lo + lr;
            // whole column tile inside one destination and all 32 rows valid: plain strided stores
            float *dst = nullptr;
            int64_t ld = 0;
            if (full_rows) {
                if (c_lo + 32 <= a.C) {
                    dst = a.hs + col;
                    ld = a.ld_hs;
                } else if (c_lo >= a.C + a.NS && c_lo + 32 <= a.M) {
                    dst = a.res + (col - a.C - a.NS);
                    ld = a.ld_res;
                }
            }
            if (dst) {
                float *p = dst + (t * 32 + 4 * hf) * ld;
                if (a.add && c_lo + 32 <= a.C) {
                    const float *q = a.add + (t * 32 + 4 * hf) * a.ld_add + col;
                    float ad[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) ad[r] = q[((r & 3) + 8 * (r >> 2)) * a.ld_add];
#pragma unroll
                    for (int r = 0; r < 16; ++r) p[((r & 3) + 8 * (r >> 2)) * ld] = acc[ct][r] + ad[r];
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) p[((r & 3) + 8 * (r >> 2)) * ld] = acc[ct][r];
                }
                continue;
            }
            const int c_hi = c_lo + 32 < a.M ? c_lo + 32 : a.M;
            if (full_rows && c_lo >= a.C && c_hi <= a.C + a.NS) {  // tile holds score columns only
                if (col < c_hi) {
                    float *p = a.score + (t * 32 + 4 * hf) * a.NS + (col - a.C);
#pragma unroll
                    for (int r = 0; r < 16; ++r) p[((r & 3) + 8 * (r >> 2)) * a.NS] = acc[ct][r];
                }
                continue;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t row = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * hf;
                if (row >= a.n) continue;
                float val = acc[ct][r];
                if (col < a.C) {
                    if (a.add) val += a.add[row * a.ld_add + col];
                    a.hs[row * a.ld_hs + col] = val;
                } else if (col < a.C + a.NS) {
                    a.score[row * a.NS + (col - a.C)] = val;
                } else if (col < a.M) {
                    a.res[row * a.ld_res + (col - a.C - a.NS)] = val;
                }
            }
        }
    }
}

template <int NCT, int KCH> int launch_stream(const ProjArgs &a, hipStream_t stream) {
    constexpr int LDW = KCH * 32 + 4;
    size_t sh = (size_t)(NCT * 32 + 4 * 32) * LDW * sizeof(float);
    static bool attr_set = false;
    if (sh > 48 * 1024 && !attr_set) {
        (void)hipFuncSetAttribute((const void *)k_proj_stream<NCT, KCH>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)sh);
        attr_set = true;
    }
    int64_t tiles = (a.n + 31) / 32;
    int blocks_per_cu = sh <= 40 * 1024 ? 4 : (sh <= 80 * 1024 ? 2 : 1);
    int64_t want = (tiles + 3) / 4;
    int64_t cap = 256 * blocks_per_cu;
    unsigned grid = (unsigned)(want < cap ? want : cap);
    MAGIC_LAUNCH(a.name, (k_proj_stream<NCT, KCH>), grid, 256, sh, stream, a);
    return magic::launch_status();
}

// (column tiles, 32-wide K chunks) pairs with a streaming instantiation
#define MAGIC_PROJ_STREAM_SHAPES(X) \
    X(5, 1) X(3, 2) X(2, 6) X(1, 2) X(2, 1) X(6, 2) X(2, 3) X(1, 5) X(1, 1) X(2, 2) X(3, 1) X(4, 1) X(8, 1)

template <int NCT> int launch(const ProjArgs &a, hipStream_t stream) {
    size_t sh = (size_t)(BM + NCT * 32) * LDS_LD * sizeof(float);
    unsigned grid = (unsigned)((a.n + BM - 1) / BM);
    static bool attr_set = false;
    if (sh > 48 * 1024 && !attr_set) {
        (void)hipFuncSetAttribute((const void *)k_proj<NCT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
        attr_set = true;
    }
    MAGIC_LAUNCH(a.name, (k_proj<NCT>), grid, 256, sh, stream, a);
    return magic::launch_status();
}

int dispatch(const ProjArgs &a, hipStream_t stream);

}  // namespace

extern "C" {

int magic_proj_fwd(int64_t n_rows, int32_t in_dim, int32_t out_c, int32_t n_score, int32_t res_c, const float *x,
                   int64_t ld_x, const float *w, float *hs, int64_t ld_hs, float *score, float *res, int64_t ld_res,
                   void *stream_) {
    if (n_rows < 0 || in_dim <= 0 || out_c <= 0 || n_score < 0 || res_c < 0) return MAGIC_ERR_INVALID_ARG;
    if (n_rows == 0) return MAGIC_OK;
    if (!x || !w || !hs || ld_x < in_dim || ld_hs < out_c) return MAGIC_ERR_INVALID_ARG;
    if ((n_score > 0 && !score) || (res_c > 0 && (!res || ld_res < res_c))) return MAGIC_ERR_INVALID_ARG;
    ProjArgs a;
    a.n = n_rows;
    a.F = in_dim;
    a.C = out_c;
    a.NS = n_score;
    a.RC = res_c;
    a.M = out_c + n_score + res_c;
    a.x = x;
    a.ld_x = ld_x;
    a.w = w;
    a.w_rs = in_dim;
    a.w_cs = 1;
    a.add = nullptr;
    a.ld_add = 0;
    a.name = "proj_fwd_mfma_f32";
    a.hs = hs;
    a.ld_hs = ld_hs;
    a.score = score;
    a.res = res;
    a.ld_res = ld_res;
    hipStream_t stream = (hipStream_t)stream_;
    return dispatch(a, stream);
}

int magic_proj_bwd_input(int64_t n_rows, int32_t k_dim, int32_t out_dim, const float *dy, int64_t ld_dy,
                         const float *w, const float *add, int64_t ld_add, float *dx, int64_t ld_dx,
                         void *stream_) {
    if (n_rows < 0 || k_dim <= 0 || out_dim <= 0) return MAGIC_ERR_INVALID_ARG;
    if (n_rows == 0) return MAGIC_OK;
    if (!dy || !w || !dx || ld_dy < k_dim || ld_dx < out_dim || (add && ld_add < out_dim)) return MAGIC_ERR_INVALID_ARG;
    ProjArgs a;
    a.n = n_rows;
    a.F = k_dim;       // reduction runs over the columns of dY
    a.C = out_dim;
    a.NS = 0;
    a.RC = 0;
    a.M = out_dim;
    a.x = dy;
    a.ld_x = ld_dy;
    a.w = w;           // [k_dim, out_dim] row-major, read transposed
    a.w_rs = 1;
    a.w_cs = out_dim;
    a.add = add;
    a.ld_add = ld_add;
    a.name = "proj_bwd_input_mfma_f32";
    a.hs = dx;
    a.ld_hs = ld_dx;
    a.score = nullptr;
    a.res = nullptr;
    a.ld_res = 0;
    return dispatch(a, (hipStream_t)stream_);
}

}  // extern "C"

namespace {
int dispatch_one(const ProjArgs &a, int nct, hipStream_t stream) {
    const int kch = (a.F + 31) / 32;
#define X(n, k) \
    if (nct == n && kch == k) return launch_stream<n, k>(a, stream);
    MAGIC_PROJ_STREAM_SHAPES(X)
#undef X
    switch (nct) {
        case 1: return launch<1>(a, stream);
        case 2: return launch<2>(a, stream);
        case 3: return launch<3>(a, stream);
        case 4: return launch<4>(a, stream);
        case 5: return launch<5>(a, stream);
        case 6: return launch<6>(a, stream);
        case 7: return launch<7>(a, stream);
        case 8: return launch<8>(a, stream);
        case 9: return launch<9>(a, stream);
        case 10: return launch<10>(a, stream);
        default: return MAGIC_ERR_UNSUPPORTED;
    }
}

// up to 10 column tiles (320 outputs) per launch; wider outputs take several launches of 8
int dispatch(const ProjArgs &a0, hipStream_t stream) {
    ProjArgs a = a0;
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
}  // namespace
