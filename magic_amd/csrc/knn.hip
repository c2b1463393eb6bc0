// Brute-force k-nearest-neighbour search: exact float64 results, found in two stages when k is
// small.
//
// Stage 1 (filter) runs the whole Q x N sweep on the bf16 matrix cores with split float32 operands
// (float32-grade products at more than five times the float64 matrix rate) and keeps, per query, the
// k' > k candidates with the smallest APPROXIMATE keys.  Stage 2 (refine)
// recomputes the float64 key of those k' candidates, orders them by (key, index) and proves the
// answer: with eps an upper bound of the stage-1 key error, every row that stage 1 dropped has an
// exact key >= tau - eps (tau = the k'-th approximate key), so the k best candidates are the k
// nearest neighbours whenever the k-th exact key is < tau - eps.  Queries that fail the test (more
// than k' - k rows within 2 eps of the k-th neighbour: heavy near-ties) are compacted on the device and
// searched again by the float64 kernel below.  The result is the float64 answer in every case.
//
// Float64 kernel:
// key(q, t) = |t|^2 - 2 q.t  orders the train rows of one query exactly like the squared
// Euclidean distance does (|q|^2 is constant per query and added at the end).  The q.t
// products come from v_mfma_f64_16x16x4_f64 with float32 inputs widened in registers (a
// float32 x float32 product is exact in float64).  Each lane keeps the current k-th best
// key of "its" query and only candidates below it enter a sorted list (in LDS for small k,
// in global memory otherwise); after the first few tiles insertions are rare
// (~k ln(N/k) per query), so the kernel runs at the matrix-core rate.
//
// MFMA f64 16x16x4 operand map (lane l): A[i = l&15][k = l>>4], B[k = l>>4][j = l&15],
// D[i = (l>>4) + 4*reg][j = l&15].  The K dimension is permuted so that lane group (l>>4)
// owns 16 contiguous input dims: A and B use the same permutation, the sum is unchanged.
#include "common.h"
#include "bf16_split.h"
#include <hipcub/hipcub.hpp>
#include <math.h>

namespace {

typedef double double4_t __attribute__((ext_vector_type(4)));

constexpr int TQ = 256;   // queries per block (4 waves x 4 blocks of 16)
constexpr int TT = 64;    // train rows per LDS tile
constexpr int DP = 64;    // padded feature width of the MFMA path
constexpr int LDT = 68;   // LDS row stride in floats (272 B: 16-B aligned, spreads banks)
constexpr int kIdxNone = 0x7fffffff;

struct KnnArgs {
    int64_t n_train, n_query;
    int d, k;
    const float *train;
    int64_t ld_train;
    const float *query;
    int64_t ld_query;
    const double *tnorm;  // [n_train]
    int n_split;
    int64_t rows_per_split;  // multiple of TT
    double *part_key;        // [n_split, Qpad, k]
    int *part_idx;           // [n_split, Qpad, k]
    int64_t q_pad;
    int lds_lists;
    // search only the queries qmap[0 .. *qcount) (both device memory), or all when qmap is null
    const int *qmap;
    const int *qcount;
};

// number of queries of this launch and the query row behind position q
__device__ __forceinline__ int64_t active_queries(const KnnArgs &a) { return a.qmap ? (int64_t)*a.qcount : a.n_query; }
__device__ __forceinline__ int64_t query_row(const KnnArgs &a, int64_t q) { return a.qmap ? (int64_t)a.qmap[q] : q; }

__global__ __launch_bounds__(256) void k_sqnorm(int64_t n, int d, const float *__restrict__ x, int64_t ld,
                                                double *__restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    for (; i < n; i += (int64_t)gridDim.x * 256) {
        double s = 0.0;
        for (int j = 0; j < d; ++j) {
            double v = (double)x[i * ld + j];
            s = fma(v, v, s);
        }
        out[i] = s;
    }
}

// The same sums (same order per row: bit-identical) with coalesced loads: a wave stages 64 rows in LDS with
// 16-byte loads (consecutive lanes read consecutive pieces), then lane r adds row r.  d % 4 == 0, d <= 64 (two
// waves per block: 33 KB of LDS), 16-byte aligned rows.
constexpr int SQ_MAXD = 64;
__global__ __launch_bounds__(128) void k_sqnorm_rows(int64_t n, int d, const float *__restrict__ x, int64_t ld,
                                                     double *__restrict__ out) {
    extern __shared__ float sq_tile[];  // 2 waves x 64 rows x (d + 1)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pitch = d + 1, d4 = d >> 2;
    float *t = sq_tile + (size_t)wave * 64 * pitch;
    const int64_t groups = (n + 63) / 64;
    for (int64_t gidx = (int64_t)blockIdx.x * 2 + wave; gidx < groups; gidx += (int64_t)gridDim.x * 2) {
        const int64_t r0 = gidx * 64;
        __builtin_amdgcn_wave_barrier();
        for (int p = lane; p < 64 * d4; p += 64) {
            const int r = p / d4, c = (p % d4) * 4;
            const int64_t row = r0 + r < n ? r0 + r : n - 1;
            const float4 v = *reinterpret_cast<const float4 *>(x + row * ld + c);
            float *q = t + r * pitch + c;
            q[0] = v.x, q[1] = v.y, q[2] = v.z, q[3] = v.w;
        }
        __builtin_amdgcn_wave_barrier();
        double s = 0.0;
        const float *q = t + lane * pitch;
        for (int j = 0; j < d; ++j) {
            const double v = (double)q[j];
            s = fma(v, v, s);
        }
        if (r0 + lane < n) out[r0 + lane] = s;
    }
}

// list access: LDS lists are plain; global lists go through L2 (agent scope) so that every
// lane of the wave sees the wave's earlier insertions
template <bool GLOBAL, typename T> __device__ __forceinline__ T ld_rel(const T *p) {
    if (GLOBAL) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return *p;
}
template <bool GLOBAL, typename T> __device__ __forceinline__ void st_rel(T *p, T v) {
    if (GLOBAL)
        __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else
        *p = v;
}
template <bool GLOBAL> __device__ __forceinline__ double ld_key(const double *p) { return ld_rel<GLOBAL>(p); }
template <bool GLOBAL> __device__ __forceinline__ int ld_idx(const int *p) { return ld_rel<GLOBAL>(p); }
template <bool GLOBAL> __device__ __forceinline__ void st_key(double *p, double v) { st_rel<GLOBAL>(p, v); }
template <bool GLOBAL> __device__ __forceinline__ void st_idx(int *p, int v) { st_rel<GLOBAL>(p, v); }

// Whole-wave sorted insertion of (kv, ti) into the k-entry list of one query (wave-uniform
// arguments).  Order: ascending (key, index).  Returns the new k-th key.
template <bool GLOBAL, typename KT>
__device__ __forceinline__ KT wave_insert(KT *L, int *I, int k, KT kv, int ti, int lane) {
    int pos = 0;
    for (int base = 0; base < k; base += 64) {
        int j = base + lane;
        bool before = false;
        if (j < k) {
            KT v = ld_rel<GLOBAL>(L + j);
            int vi = ld_rel<GLOBAL>(I + j);
            before = (v < kv) || (v == kv && vi < ti);
        }
        pos += __popcll(__ballot(before));
    }
    if (pos >= k) return ld_rel<GLOBAL>(L + k - 1);
    for (int base = ((k - 1) / 64) * 64; base >= 0; base -= 64) {
        int j = base + lane;
        KT v = 0;
        int vi = 0;
        bool mv = j >= pos && j + 1 < k;
        if (mv) {
            v = ld_rel<GLOBAL>(L + j);
            vi = ld_rel<GLOBAL>(I + j);
        }
        if (GLOBAL) __builtin_amdgcn_s_waitcnt(0);
        if (mv) {
            st_rel<GLOBAL>(L + j + 1, v);
            st_rel<GLOBAL>(I + j + 1, vi);
        }
        if (GLOBAL) __builtin_amdgcn_s_waitcnt(0);
    }
    if (lane == 0) {
        st_rel<GLOBAL>(L + pos, kv);
        st_rel<GLOBAL>(I + pos, ti);
    }
    if (GLOBAL) __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    return ld_rel<GLOBAL>(L + k - 1);
}

// value of lane `src` (wave-uniform) without a trip through the LDS crossbar
__device__ __forceinline__ int lane_value(int v, int src) { return __builtin_amdgcn_readlane(v, src); }
__device__ __forceinline__ float lane_value(float v, int src) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src));
}

// The same insertion for a float-keyed list of at most 64 entries in LDS: lane j holds entry j, so the
// entries are read once, the lanes behind the insertion point store theirs one slot further, and the new
// k-th key comes out of a lane read instead of another LDS read: one LDS round trip per insertion.
__device__ __forceinline__ float wave_insert_lds(float *L, int *I, int k, float kv, int ti, int lane) {
    const bool in = lane < k;
    float v = kv;
    int vi = ti;
    if (in) {
        v = L[lane];
        vi = I[lane];
    }
    const bool before = in && ((v < kv) || (v == kv && vi < ti));
    const int pos = __popcll(__ballot(before));
    if (pos >= k) return lane_value(v, k - 1);
    const float prev = lane_value(v, k >= 2 ? k - 2 : 0);
    if (lane >= pos && lane + 1 < k) {
        L[lane + 1] = v;
        I[lane + 1] = vi;
    }
    if (lane == pos) {
        L[pos] = kv;
        I[pos] = ti;
    }
    __builtin_amdgcn_wave_barrier();
    return pos == k - 1 ? kv : prev;
}

template <bool GLOBAL>
__global__ __launch_bounds__(256, 2) void k_knn_mfma(KnnArgs a) {
    extern __shared__ __align__(16) char smem[];
    float *tile = reinterpret_cast<float *>(smem);                         // TT*LDT floats
    double *tn = reinterpret_cast<double *>(smem + TT * LDT * 4);          // TT
    double *thr = tn + TT;                                                 // TQ
    double *lkey = thr + TQ;                                               // TQ*k (LDS lists)
    int *lidx = reinterpret_cast<int *>(lkey + (GLOBAL ? 0 : (size_t)TQ * a.k));

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int split = blockIdx.y;
    const int64_t q_base = (int64_t)blockIdx.x * TQ;
    const int64_t n_active = active_queries(a);
    if (q_base >= n_active) return;  // uniform: the whole block leaves
    const int64_t t_lo = (int64_t)split * a.rows_per_split;
    int64_t t_hi = t_lo + a.rows_per_split;
    if (t_hi > a.n_train) t_hi = a.n_train;

    double *L0;
    int *I0;
    if (GLOBAL) {
        L0 = a.part_key + ((int64_t)split * a.q_pad + q_base) * a.k;
        I0 = a.part_idx + ((int64_t)split * a.q_pad + q_base) * a.k;
    } else {
        L0 = lkey;
        I0 = lidx;
    }
    for (int i = tid; i < TQ * a.k; i += 256) {
        st_key<GLOBAL>(L0 + i, INFINITY);
        st_idx<GLOBAL>(I0 + i, kIdxNone);
    }
    thr[tid] = INFINITY;

    // query fragments: lane holds dims [16*(lane>>4), +16) of query (wave*64 + qb*16 + lane&15)
    double bq[4][16];
#pragma unroll
    for (int qb = 0; qb < 4; ++qb) {
        int64_t q = q_base + wave * 64 + qb * 16 + (lane & 15);
        const int d0 = 16 * (lane >> 4);
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            float v = 0.f;
            if (q < n_active && d0 + j < a.d) v = a.query[query_row(a, q) * a.ld_query + d0 + j];
            bq[qb][j] = (double)v;
        }
    }
    __syncthreads();

    // 16-query tiles of this wave that hold a query at all: the search of the few queries the proof rejected
    // fills one or two tiles of a block, and the other waves only help staging the train tiles
    const int64_t q_left = n_active - q_base - wave * 64;
    const int nqb = q_left <= 0 ? 0 : (q_left >= 64 ? 4 : (int)((q_left + 15) / 16));

    // Software pipeline over the train tiles: the global loads of tile i+1 are issued into registers
    // before the matrix work on tile i and written to LDS after it, so their latency hides behind the
    // MFMAs instead of standing between two barriers.
    float4 pre[TT / 16];
    double pre_tn = (double)INFINITY;
    auto fetch = [&](int64_t t0) {
#pragma unroll
        for (int i = 0; i < TT / 16; ++i) {
            const int r = (tid >> 4) + 16 * i;
            const int c4 = (tid & 15) * 4;
            const int64_t row = t0 + r;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < t_hi) {
                const float *p = a.train + row * a.ld_train + c4;
                if (c4 + 3 < a.d) {
                    v = *reinterpret_cast<const float4 *>(p);
                } else {
                    if (c4 + 0 < a.d) v.x = p[0];
                    if (c4 + 1 < a.d) v.y = p[1];
                    if (c4 + 2 < a.d) v.z = p[2];
                }
            }
            pre[i] = v;
        }
        if (tid < TT) pre_tn = (t0 + tid < t_hi) ? a.tnorm[t0 + tid] : (double)INFINITY;
    };
    if (t_lo < t_hi) fetch(t_lo);
    for (int64_t t0 = t_lo; t0 < t_hi; t0 += TT) {
        // stage the prefetched TT train rows (zero padded) + their squared norms
        __syncthreads();
#pragma unroll
        for (int i = 0; i < TT / 16; ++i)
            *reinterpret_cast<float4 *>(tile + ((tid >> 4) + 16 * i) * LDT + (tid & 15) * 4) = pre[i];
        if (tid < TT) tn[tid] = pre_tn;
        __syncthreads();
        if (t0 + TT < t_hi) fetch(t0 + TT);

#pragma unroll 1
        for (int tb = 0; tb < (nqb > 0 ? TT / 16 : 0); ++tb) {
            double af[16];
            {
                const float *p = tile + (tb * 16 + (lane & 15)) * LDT + 16 * (lane >> 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float4 v = *reinterpret_cast<const float4 *>(p + 4 * j);
                    af[4 * j + 0] = (double)v.x;
                    af[4 * j + 1] = (double)v.y;
                    af[4 * j + 2] = (double)v.z;
                    af[4 * j + 3] = (double)v.w;
                }
            }
            double tnr[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) tnr[r] = tn[tb * 16 + (lane >> 4) + 4 * r];
#pragma unroll
            for (int qb = 0; qb < 4; ++qb) {
                if (qb >= nqb) continue;  // uniform per wave: no query in this tile
                double4_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int ks = 0; ks < 16; ++ks)
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(af[ks], bq[qb][ks], acc, 0, 0, 0);
                const int ql = wave * 64 + qb * 16 + (lane & 15);
                const double th = thr[ql];
                double key[4];
                bool any = false;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    key[r] = fma(-2.0, acc[r], tnr[r]);
                    any |= key[r] < th;
                }
                if (__any(any)) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        unsigned long long mask = __ballot(key[r] < th);
                        const int ti_mine = (int)(t0 + tb * 16 + (lane >> 4) + 4 * r);
                        while (mask) {
                            const int srcl = __ffsll((long long)mask) - 1;
                            mask &= mask - 1;
                            const int q_ins = __shfl(ql, srcl, 64);
                            const double kv = __shfl(key[r], srcl, 64);
                            const int ti = __shfl(ti_mine, srcl, 64);
                            if (kv < thr[q_ins]) {
                                double nt = wave_insert<GLOBAL, double>(L0 + (int64_t)q_ins * a.k,
                                                                        I0 + (int64_t)q_ins * a.k, a.k, kv, ti, lane);
                                if (lane == 0) thr[q_ins] = nt;
                                __builtin_amdgcn_wave_barrier();
                            }
                        }
                    }
                }
            }
        }
    }
    if (!GLOBAL) {
        __syncthreads();
        double *gk = a.part_key + ((int64_t)split * a.q_pad + q_base) * a.k;
        int *gi = a.part_idx + ((int64_t)split * a.q_pad + q_base) * a.k;
        for (int i = tid; i < TQ * a.k; i += 256) {
            gk[i] = lkey[i];
            gi[i] = lidx[i];
        }
    }
}

// ---- float64 sweep for long lists (k beyond what LDS holds) -------------------------------------------
// A sorted list in global memory costs a wave four dependent memory round trips per insertion, and a list of k
// entries takes k ln(N/k) insertions: 1,500 for k = 200.  Here every query owns an UNSORTED buffer of
// cap = 64 * NI >= 2k candidates in global memory: a lane whose key is under its query's threshold appends it
// (the slot comes from a per-query count held in registers and a ballot; the lanes of a tile append side by side,
// nothing is shifted and nothing is waited for).  When a
// buffer may overflow in the next train tile (64 rows) its wave sorts it -- entries in registers, NI per lane, a
// bitonic network over (key, row) with lane shuffles -- keeps the k smallest and tightens the threshold to the
// k-th key (checked once per 64-row train tile, where few registers are live).  The number of rows between two sorts grows geometrically (the threshold is exact when it is set, so
// a later row passes with probability k / rows seen): about log(N / cap) / log(cap / k) sorts per query.  What a
// buffer holds is a set, so the result does not depend on the order of the appends.
// (bitwise on purpose: the sorting network below is straight-line code of compares and selects)
__device__ __forceinline__ bool pair_less(double ka, int ia, double kb, int ib) {
    return (ka < kb) | ((ka == kb) & (ia < ib));
}

template <int NI>
__device__ __forceinline__ void wave_sort_pairs(double (&K)[NI], int (&I)[NI], int lane) {
    constexpr int N = 64 * NI;  // element g = r * 64 + lane
#pragma unroll
    for (int size = 2; size <= N; size <<= 1) {
#pragma unroll
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            if (stride >= 64) {  // partner in the same lane, register r ^ (stride / 64)
                const int rs = stride >> 6;
#pragma unroll
                for (int r = 0; r < NI; ++r) {
                    if ((r & rs) == 0) {
                        const bool asc = ((r * 64) & size) == 0;
                        const double k0 = K[r], k1 = K[r | rs];
                        const int i0 = I[r], i1 = I[r | rs];
                        const bool sw = pair_less(k1, i1, k0, i0) == asc;
                        K[r] = sw ? k1 : k0, I[r] = sw ? i1 : i0;
                        K[r | rs] = sw ? k0 : k1, I[r | rs] = sw ? i0 : i1;
                    }
                }
            } else {  // partner in lane ^ stride, same register
#pragma unroll
                for (int r = 0; r < NI; ++r) {
                    const double ko = __shfl_xor(K[r], stride, 64);
                    const int io = __shfl_xor(I[r], stride, 64);
                    const bool keep_min = ((lane & stride) == 0) == ((((r * 64) | lane) & size) == 0);
                    const bool take = pair_less(ko, io, K[r], I[r]) == keep_min;
                    K[r] = take ? ko : K[r];
                    I[r] = take ? io : I[r];
                }
            }
        }
    }
}

// Sorts the first n entries of one query's buffer (whole wave, wave-uniform arguments), writes the k smallest
// back in ascending (key, row) order; returns through `cnt` / `thr` what the query continues with.
template <int NI>
__device__ __forceinline__ void buffer_compact(double *bk, int *bi, int n, int k, int lane, int &cnt, double &thr) {
    double K[NI];
    int I[NI];
    __builtin_amdgcn_s_waitcnt(0);  // the appends of this wave have left
#pragma unroll
    for (int r = 0; r < NI; ++r) {
        const int g = r * 64 + lane;
        K[r] = g < n ? ld_rel<true>(bk + g) : (double)INFINITY;
        I[r] = g < n ? ld_rel<true>(bi + g) : kIdxNone;
    }
    wave_sort_pairs<NI>(K, I, lane);
    cnt = n < k ? n : k;
    double t = INFINITY;
#pragma unroll
    for (int r = 0; r < NI; ++r) {
        const int g = r * 64 + lane;
        if (g < cnt) {
            st_rel<true>(bk + g, K[r]);
            st_rel<true>(bi + g, I[r]);
        }
        if (r == ((k - 1) >> 6)) t = K[r];
    }
    t = __shfl(t, (k - 1) & 63, 64);
    thr = n >= k ? t : (double)INFINITY;
    __builtin_amdgcn_s_waitcnt(0);
}

template <int NI>
__global__ __launch_bounds__(256, 2) void k_knn_mfma_buf(KnnArgs a, double *buf_key, int *buf_idx) {
    constexpr int CAP = 64 * NI;
    extern __shared__ __align__(16) char smem[];
    float *tile = reinterpret_cast<float *>(smem);                 // TT*LDT floats
    double *tn = reinterpret_cast<double *>(smem + TT * LDT * 4);  // TT
    double *thr = tn + TT;                                         // TQ
    int *cnt = reinterpret_cast<int *>(thr + TQ);                  // TQ

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int split = blockIdx.y;
    const int64_t q_base = (int64_t)blockIdx.x * TQ;
    const int64_t n_active = active_queries(a);
    if (q_base >= n_active) return;  // uniform: the whole block leaves
    const int64_t t_lo = (int64_t)split * a.rows_per_split;
    int64_t t_hi = t_lo + a.rows_per_split;
    if (t_hi > a.n_train) t_hi = a.n_train;
    double *B0 = buf_key + ((int64_t)split * a.q_pad + q_base) * CAP;
    int *J0 = buf_idx + ((int64_t)split * a.q_pad + q_base) * CAP;
    thr[tid] = INFINITY;
    cnt[tid] = 0;

    double bq[4][16];
#pragma unroll
    for (int qb = 0; qb < 4; ++qb) {
        int64_t q = q_base + wave * 64 + qb * 16 + (lane & 15);
        const int d0 = 16 * (lane >> 4);
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            float v = 0.f;
            if (q < n_active && d0 + j < a.d) v = a.query[query_row(a, q) * a.ld_query + d0 + j];
            bq[qb][j] = (double)v;
        }
    }
    __syncthreads();
    const int64_t q_left = n_active - q_base - wave * 64;  // queries of this wave
    const int nqb = q_left <= 0 ? 0 : (q_left >= 64 ? 4 : (int)((q_left + 15) / 16));

    float4 pre[TT / 16];
    double pre_tn = (double)INFINITY;
    auto fetch = [&](int64_t t0) {
#pragma unroll
        for (int i = 0; i < TT / 16; ++i) {
            const int r = (tid >> 4) + 16 * i;
            const int c4 = (tid & 15) * 4;
            const int64_t row = t0 + r;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < t_hi) {
                const float *p = a.train + row * a.ld_train + c4;
                if (c4 + 3 < a.d) {
                    v = *reinterpret_cast<const float4 *>(p);
                } else {
                    if (c4 + 0 < a.d) v.x = p[0];
                    if (c4 + 1 < a.d) v.y = p[1];
                    if (c4 + 2 < a.d) v.z = p[2];
                }
            }
            pre[i] = v;
        }
        if (tid < TT) pre_tn = (t0 + tid < t_hi) ? a.tnorm[t0 + tid] : (double)INFINITY;
    };
    // sort the buffer of the wave's query `ql` (wave-uniform) and publish its new count / threshold
    auto compact = [&](int ql) {
        int c;
        double t;
        buffer_compact<NI>(B0 + (int64_t)ql * CAP, J0 + (int64_t)ql * CAP, cnt[ql], a.k, lane, c, t);
        if (lane == 0) {
            cnt[ql] = c;
            thr[ql] = t;
        }
        __builtin_amdgcn_wave_barrier();
    };
    // count and threshold of the lane's query in each of the wave's four query tiles, in registers (the same value
    // in the four lanes of a query); the LDS copies are current at tile boundaries
    int cnq[4];
    double thq[4];
    auto load_state = [&]() {
#pragma unroll
        for (int qb = 0; qb < 4; ++qb) {
            cnq[qb] = cnt[wave * 64 + qb * 16 + (lane & 15)];
            thq[qb] = thr[wave * 64 + qb * 16 + (lane & 15)];
        }
    };
    load_state();
    if (t_lo < t_hi) fetch(t_lo);
    for (int64_t t0 = t_lo; t0 < t_hi; t0 += TT) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < TT / 16; ++i)
            *reinterpret_cast<float4 *>(tile + ((tid >> 4) + 16 * i) * LDT + (tid & 15) * 4) = pre[i];
        if (tid < TT) tn[tid] = pre_tn;
        __syncthreads();
        if (t0 + TT < t_hi) fetch(t0 + TT);
        // Room for the TT rows of this tile in every buffer.  A wave that sorts keeps its whole block waiting at the
        // next barrier, so the block sorts together: when any of its 256 buffers is about to overflow, every wave
        // sorts those of its own buffers that are more than half way there (lane = query).
        const bool block_full = __any(cnt[lane] > CAP - TT || cnt[64 + lane] > CAP - TT || cnt[128 + lane] > CAP - TT ||
                                      cnt[192 + lane] > CAP - TT);
        if (nqb > 0 && block_full) {
            unsigned long long full = __ballot(cnt[wave * 64 + lane] > a.k + (CAP - TT - a.k) / 2);
            if (full) {
                while (full) {
                    const int l = __ffsll((long long)full) - 1;
                    full &= full - 1;
                    compact(wave * 64 + l);
                }
                load_state();
            }
        }

#pragma unroll 1
        for (int tb = 0; tb < (nqb > 0 ? TT / 16 : 0); ++tb) {
            double af[16];
            {
                const float *p = tile + (tb * 16 + (lane & 15)) * LDT + 16 * (lane >> 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float4 v = *reinterpret_cast<const float4 *>(p + 4 * j);
                    af[4 * j + 0] = (double)v.x;
                    af[4 * j + 1] = (double)v.y;
                    af[4 * j + 2] = (double)v.z;
                    af[4 * j + 3] = (double)v.w;
                }
            }
            double tnr[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) tnr[r] = tn[tb * 16 + (lane >> 4) + 4 * r];
#pragma unroll
            for (int qb = 0; qb < 4; ++qb) {
                if (qb >= nqb) continue;  // uniform per wave: no query in this tile
                double4_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int ks = 0; ks < 16; ++ks)
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(af[ks], bq[qb][ks], acc, 0, 0, 0);
                const int ql = wave * 64 + qb * 16 + (lane & 15);
                const bool valid = q_base + ql < n_active;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const double key = fma(-2.0, acc[r], tnr[r]);
                    const bool cand = valid && key < thq[qb];
                    // slots without a round trip: the four lanes of a query count each other's candidates
                    const unsigned long long m = __ballot(cand);
                    if (m) {
                        const unsigned long long col = (m >> (lane & 15)) & 0x0001000100010001ull;
                        const int below = __popcll(col & ((1ull << (16 * (lane >> 4))) - 1ull));
                        if (cand) {
                            const int slot = cnq[qb] + below;
                            st_rel<true>(B0 + (int64_t)ql * CAP + slot, key);
                            st_rel<true>(J0 + (int64_t)ql * CAP + slot, (int)(t0 + tb * 16 + (lane >> 4) + 4 * r));
                        }
                        cnq[qb] += __popcll(col);
                    }
                }
            }
        }
        if (nqb > 0) {  // publish the counts of this tile
#pragma unroll
            for (int qb = 0; qb < 4; ++qb)
                if (lane < 16) cnt[wave * 64 + qb * 16 + lane] = cnq[qb];
            __builtin_amdgcn_wave_barrier();
        }
    }
    // the k best of every query, ascending, in the layout the finalize kernels merge
    double *gk = a.part_key + ((int64_t)split * a.q_pad + q_base) * a.k;
    int *gi = a.part_idx + ((int64_t)split * a.q_pad + q_base) * a.k;
    for (int j = 0; j < 64; ++j) {
        const int ql = wave * 64 + j;
        if (q_base + ql >= n_active) break;  // uniform
        compact(ql);
        const int c = cnt[ql];
        for (int o = lane; o < a.k; o += 64) {
            gk[(int64_t)ql * a.k + o] = o < c ? ld_rel<true>(B0 + (int64_t)ql * CAP + o) : (double)INFINITY;
            gi[(int64_t)ql * a.k + o] = o < c ? ld_rel<true>(J0 + (int64_t)ql * CAP + o) : kIdxNone;
        }
    }
}

// ---- stage 1: approximate keys on the bf16 matrix cores, float32-grade from split operands ---------
// Every float32 operand is the sum of three bf16 pieces and the six leading cross products are
// accumulated in float32 (bf16_split.h): 6 x 32 cycles per 32x32x16 step against 4 x 4 x 32 cycles of
// v_mfma_f32_16x16x4_f32 for the same products, 2.7 times the float32 matrix rate.  What is dropped
// (2^-23 |q||t| per product) goes into the error bound of the proof in stage 2.
// Block: 256 queries, wave w owns the query tiles 4w .. 4w+3 (16 queries each) with their split fragments
// in registers for the whole sweep; the 64-row train tile is split while it is staged into LDS (three
// planes of [64][72] bf16: the 16-byte fragment reads of 16 consecutive lanes cover all 64 banks once).
// The matrix instruction is v_mfma_f32_16x16x32_bf16: on random operands the chip holds a higher clock on
// this shape than on 32x32x16 at the same cycles per product (measured: the sweep runs faster by the
// clock ratio).  Lane l: A[i = l&15][k = 8*(l>>4) ..+7], B[k = 8*(l>>4) ..+7][j = l&15],
// D[i = 4*(l>>4) + r][j = l&15], r = 0..3.
constexpr int BLD = DP + 8;             // bf16 per staged train row
constexpr int BPLANE = TT * BLD;        // bf16 per plane
typedef float float16_t __attribute__((ext_vector_type(16)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256, 2) void k_knn_filter_bf16(KnnArgs a, int kf, float *fkey, int *fidx) {
    using magic::bf16x2_t;
    using magic::bf16x8_t;
    using magic::f32x2_t;
    extern __shared__ __align__(16) char smem[];
    __bf16 *tile = reinterpret_cast<__bf16 *>(smem);                 // 3 * BPLANE
    float *tn = reinterpret_cast<float *>(smem + 3 * BPLANE * 2);    // TT
    float *thr = tn + TT;                                            // TQ
    float *lkey = thr + TQ;                                          // TQ*kf
    int *lidx = reinterpret_cast<int *>(lkey + (size_t)TQ * kf);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l16 = lane & 15, quad = lane >> 4;
    const int split = blockIdx.y;
    const int64_t q_base = (int64_t)blockIdx.x * TQ;
    const int64_t t_lo = (int64_t)split * a.rows_per_split;
    int64_t t_hi = t_lo + a.rows_per_split;
    if (t_hi > a.n_train) t_hi = a.n_train;
    for (int i = tid; i < TQ * kf; i += 256) {
        lkey[i] = INFINITY;
        lidx[i] = kIdxNone;
    }
    thr[tid] = INFINITY;

    // query fragments: tile qt, K step s -> dims [32 s + 8 quad, +8) of query (4 wave + qt) * 16 + l16
    bf16x8_t bq[4][2][3];
#pragma unroll
    for (int qt = 0; qt < 4; ++qt) {
        const int64_t q = q_base + (4 * wave + qt) * 16 + l16;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int dj = 32 * s + 8 * quad + j;
                v[j] = (q < a.n_query && dj < a.d) ? a.query[q * a.ld_query + dj] : 0.f;
            }
            magic::split8(make_float4(v[0], v[1], v[2], v[3]), make_float4(v[4], v[5], v[6], v[7]), bq[qt][s][0],
                          bq[qt][s][1], bq[qt][s][2]);
        }
    }
    __syncthreads();

    // train tiles are requested two tiles ahead (two register buffers, used alternately): one tile of matrix
    // work is shorter than a loaded memory system's latency
    float4 preA[TT / 16], preB[TT / 16];
    float tnA = INFINITY, tnB = INFINITY;
    auto fetch = [&](int64_t t0, float4 (&pre)[TT / 16], float &pre_tn) {
        if (t0 >= t_hi) return;
#pragma unroll
        for (int i = 0; i < TT / 16; ++i) {
            const int r = (tid >> 4) + 16 * i;
            const int c4 = (tid & 15) * 4;
            const int64_t row = t0 + r;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < t_hi) {
                const float *p = a.train + row * a.ld_train + c4;
                if (c4 + 3 < a.d) {
                    v = *reinterpret_cast<const float4 *>(p);
                } else {
                    if (c4 + 0 < a.d) v.x = p[0];
                    if (c4 + 1 < a.d) v.y = p[1];
                    if (c4 + 2 < a.d) v.z = p[2];
                }
            }
            pre[i] = v;
        }
        if (tid < TT) pre_tn = (t0 + tid < t_hi) ? (float)a.tnorm[t0 + tid] : INFINITY;
    };
    union Pack4 {
        bf16x2_t h[2];
        uint2 u;
    };
    fetch(t_lo, preA, tnA);
    fetch(t_lo + TT, preB, tnB);
    // thresholds (the k'-th key so far) of this lane's four queries and their gate bars: registers, replicated
    // over the four lanes that hold rows of the same query; only this wave inserts for its queries
    float thq[4], bar[4];
#pragma unroll
    for (int qt = 0; qt < 4; ++qt) {
        thq[qt] = INFINITY;
        bar[qt] = -INFINITY;
    }
    auto tile_step = [&](int64_t t0, float4 (&pre)[TT / 16], float &pre_tn) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < TT / 16; ++i) {  // four consecutive k of one row -> 8 bytes in each plane
            Pack4 p1, p2, p3;
            magic::split2(f32x2_t{pre[i].x, pre[i].y}, p1.h[0], p2.h[0], p3.h[0]);
            magic::split2(f32x2_t{pre[i].z, pre[i].w}, p1.h[1], p2.h[1], p3.h[1]);
            __bf16 *dst = tile + ((tid >> 4) + 16 * i) * BLD + (tid & 15) * 4;
            *reinterpret_cast<uint2 *>(dst) = p1.u;
            *reinterpret_cast<uint2 *>(dst + BPLANE) = p2.u;
            *reinterpret_cast<uint2 *>(dst + 2 * BPLANE) = p3.u;
        }
        if (tid < TT) tn[tid] = pre_tn;
        __syncthreads();
        fetch(t0 + 2 * TT, pre, pre_tn);

        // matrix work of 16 train rows against the wave's four query tiles (side by side: consecutive matrix
        // instructions never accumulate into the result of the one right before them)
        auto issue = [&](int rt, f32x4_t (&acc)[4], float (&htn)[4]) {
            bf16x8_t af[2][3];
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
                    af[s][pl] = *reinterpret_cast<const bf16x8_t *>(tile + pl * BPLANE + (rt * 16 + l16) * BLD + 32 * s + 8 * quad);
            // htn[r] = -|t|^2 / 2 of the 4 train rows this lane sees: key < th  <=>  acc + htn > -th / 2
            const float4 t4 = *reinterpret_cast<const float4 *>(tn + rt * 16 + 4 * quad);
            htn[0] = -0.5f * t4.x, htn[1] = -0.5f * t4.y, htn[2] = -0.5f * t4.z, htn[3] = -0.5f * t4.w;
#pragma unroll
            for (int qt = 0; qt < 4; ++qt) acc[qt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};  // smallest terms first
#pragma unroll
                for (int p6 = 0; p6 < 6; ++p6)
#pragma unroll
                    for (int qt = 0; qt < 4; ++qt)
                        acc[qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[s][PA[p6]], bq[qt][s][PB[p6]], acc[qt], 0, 0, 0);
            }
        };
        // threshold test of 16 train rows and the insertions that pass it
        auto gate = [&](int rt, const f32x4_t (&acc)[4], const float (&htn)[4]) {
            // cheap gate (4 adds, 2 maxima per tile): can any of the keys be below its query's threshold?  The
            // bar sits a little low (1e-6 |th|: more than the roundings that separate this form from the key
            // below), so the gate never hides a candidate; the exact test follows inside.
            bool hit = false;
#pragma unroll
            for (int qt = 0; qt < 4; ++qt) {
                const float m = fmaxf(fmaxf(acc[qt][0] + htn[0], acc[qt][1] + htn[1]),
                                      fmaxf(acc[qt][2] + htn[2], acc[qt][3] + htn[3]));
                hit |= m > bar[qt];
            }
            // ask the scheduler to thread these vector instructions between the matrix instructions issued above
#pragma unroll
            for (int i = 0; i < 36; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
            }
            if (__any(hit)) {
#pragma unroll
                for (int qt = 0; qt < 4; ++qt) {
                    const float mq = fmaxf(fmaxf(acc[qt][0] + htn[0], acc[qt][1] + htn[1]),
                                           fmaxf(acc[qt][2] + htn[2], acc[qt][3] + htn[3]));
                    if (!__any(mq > bar[qt])) continue;  // this query tile holds no candidate
                    float th = thq[qt];  // threshold of query (qt, l16), kept current through the insertions below
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float key_r = fmaf(-2.f, acc[qt][r], -2.f * htn[r]);
                        unsigned long long mask = __ballot(key_r < th);
                        const int ti_mine = (int)(t0 + rt * 16 + 4 * quad + r);
                        while (mask) {
                            const int srcl = __ffsll((long long)mask) - 1;
                            mask &= mask - 1;
                            const float kv = lane_value(key_r, srcl);
                            if (kv < lane_value(th, srcl)) {  // an earlier insertion may have tightened it
                                const int ti = lane_value(ti_mine, srcl);
                                const int q_ins = (4 * wave + qt) * 16 + (srcl & 15);
                                const float nt = wave_insert_lds(lkey + (size_t)q_ins * kf, lidx + (size_t)q_ins * kf, kf, kv,
                                                                 ti, lane);
                                if (l16 == (srcl & 15)) th = nt;
                            }
                        }
                    }
                    thq[qt] = th;
                    bar[qt] = fmaf(-0.5f, th, -fabsf(th) * 1e-6f - 1e-30f);
                }
            }
        };
        // the test of row tile i runs behind the matrix instructions of row tile i + 1: its vector
        // instructions read results that are complete instead of waiting for the matrix pipe to drain
        f32x4_t accC[4], accN[4];
        float htnC[4], htnN[4];
        issue(0, accC, htnC);
#pragma unroll 1
        for (int rt = 0; rt + 1 < TT / 16; ++rt) {
            issue(rt + 1, accN, htnN);
            gate(rt, accC, htnC);
#pragma unroll
            for (int qt = 0; qt < 4; ++qt) accC[qt] = accN[qt];
#pragma unroll
            for (int r = 0; r < 4; ++r) htnC[r] = htnN[r];
        }
        gate(TT / 16 - 1, accC, htnC);
    };
    for (int64_t t0 = t_lo; t0 < t_hi; t0 += 2 * TT) {
        tile_step(t0, preA, tnA);
        if (t0 + TT < t_hi) tile_step(t0 + TT, preB, tnB);
    }
    __syncthreads();
    float *gk = fkey + ((int64_t)split * a.q_pad + q_base) * kf;
    int *gi = fidx + ((int64_t)split * a.q_pad + q_base) * kf;
    for (int i = tid; i < TQ * kf; i += 256) {
        gk[i] = lkey[i];
        gi[i] = lidx[i];
    }
}

// largest squared train norm (bit pattern of a non-negative double orders like the value)
__global__ __launch_bounds__(256) void k_tmax(int64_t n, const double *tnorm, unsigned long long *out) {
    double m = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) m = fmax(m, tnorm[i]);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = fmax(m, __shfl_xor(m, off, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(out, (unsigned long long)__double_as_longlong(m));
}

// exact float64 key of (query row, train row): the arithmetic every reported distance comes from
__device__ __forceinline__ double exact_key(const KnnArgs &a, const float *qr, int ti) {
    const float *tr = a.train + (int64_t)ti * a.ld_train;
    double dot = 0.0;
    for (int j = 0; j < a.d; ++j) dot = fma((double)qr[j], (double)tr[j], dot);
    return fma(-2.0, dot, a.tnorm[ti]);
}

constexpr int KF_MAX = 32;

// sort n (key, index) pairs ascending by (key, index); n <= KF_MAX
__device__ __forceinline__ void sort_pairs(double *K, int *I, int n) {
    for (int i = 1; i < n; ++i) {
        const double kv = K[i];
        const int iv = I[i];
        int j = i - 1;
        while (j >= 0 && (K[j] > kv || (K[j] == kv && I[j] > iv))) {
            K[j + 1] = K[j], I[j + 1] = I[j];
            --j;
        }
        K[j + 1] = kv, I[j + 1] = iv;
    }
}

// ---- stage 2: exact keys of the kf candidates of every query, proof, output -------------------------
// One thread per query.  flag[q] = 1 when the proof fails (the query goes to the float64 search).
__global__ __launch_bounds__(128) void k_knn_refine(KnnArgs a, int kf, const float *fkey, const int *fidx,
                                                    const unsigned long long *tmax_bits, double gamma,
                                                    int64_t index_offset, double *dist, int64_t *idx, int *flag) {
    const int64_t q = (int64_t)blockIdx.x * 128 + threadIdx.x;
    if (q >= a.n_query) return;
    const float *qr = a.query + q * a.ld_query;
    double qn = 0.0;
    for (int j = 0; j < a.d; ++j) qn = fma((double)qr[j], (double)qr[j], qn);
    // the kf smallest approximate keys over the per-split lists, ascending (key, row)
    double K[KF_MAX];
    int I[KF_MAX];
    int cur[64];
    for (int s = 0; s < a.n_split; ++s) cur[s] = 0;
    int n = 0;
    float tau = INFINITY;  // approximate key of the kf-th candidate: no dropped row lies below it
    for (int o = 0; o < kf; ++o) {
        float best = INFINITY;
        int bi = kIdxNone, bs = -1;
        for (int s = 0; s < a.n_split; ++s) {
            if (cur[s] >= kf) continue;
            const int64_t off = ((int64_t)s * a.q_pad + q) * kf + cur[s];
            const float v = fkey[off];
            const int vi = fidx[off];
            if (vi != kIdxNone && (bs < 0 || v < best || (v == best && vi < bi))) best = v, bi = vi, bs = s;
        }
        if (bs < 0) break;
        cur[bs]++;
        I[n++] = bi;
        if (o == kf - 1) tau = best;
    }
    for (int c = 0; c < n; ++c) K[c] = exact_key(a, qr, I[c]);
    sort_pairs(K, I, n);
    // error bound of a stage-1 key = |t|^2 - 2 q.t: |error of q.t| <= gamma |q||t| (gamma from the host: the
    // roundings of the accumulation, and the dropped cross products of the split operands), the norm
    // rounded to float32 and the final fma; a quarter more for margin
    const double tmax = sqrt(__longlong_as_double((long long)*tmax_bits)), qs = sqrt(qn);
    const double u = 5.9604644775390625e-08;  // 2^-24
    const double eps = 1.25 * (2.0 * gamma * qs * tmax + 2.0 * u * (tmax * tmax + 2.0 * qs * tmax));
    const bool complete = n < kf;  // fewer candidates than list entries: every train row is a candidate
    const bool proven = complete || (a.k <= n && K[a.k - 1] < (double)tau - eps);
    flag[q] = proven ? 0 : 1;
    for (int o = 0; o < a.k; ++o) {
        const bool ok = o < n;
        const double d2 = ok ? K[o] + qn : 0.0;
        dist[q * a.k + o] = ok ? sqrt(d2 > 0.0 ? d2 : 0.0) : (double)INFINITY;
        idx[q * a.k + o] = ok ? (int64_t)I[o] + index_offset : (int64_t)-1;
    }
}

// Stage 2 for unsplit train rows (one list per query, kf <= 32): half a wavefront per query, lane = candidate.
// Every lane forms the exact key of its candidate with the arithmetic of exact_key, the 32 (key, row) pairs
// are ordered by a bitonic network of lane exchanges, and the proof and the outputs are those of
// k_knn_refine (same tau, same eps, same order): the two kernels give identical results.
__global__ __launch_bounds__(256) void k_knn_refine_one(KnnArgs a, int kf, const float *fkey, const int *fidx,
                                                        const unsigned long long *tmax_bits, double gamma,
                                                        int64_t index_offset, double *dist, int64_t *idx, int *flag) {
    const int lane = threadIdx.x & 63, c = lane & 31;
    const int64_t q = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 5;
    const bool qok = q < a.n_query;
    const int64_t qc = qok ? q : a.n_query - 1;  // all lanes stay for the exchanges
    const float *qr = a.query + qc * a.ld_query;
    double qn = 0.0;
    for (int j = 0; j < a.d; ++j) qn = fma((double)qr[j], (double)qr[j], qn);
    const bool slot = c < kf;
    const int ti = slot ? fidx[qc * kf + c] : kIdxNone;
    const float approx = slot ? fkey[qc * kf + c] : INFINITY;
    const bool valid = ti != kIdxNone;
    const unsigned long long half = __ballot(valid) >> (lane & 32) & 0xffffffffull;
    const int n = __popcll(half);
    // the list is ascending: its last entry is the kf-th smallest approximate key
    const float tau_full = __shfl(approx, (lane & 32) + kf - 1, 64);
    const float tau = n == kf ? tau_full : INFINITY;
    double K = valid ? exact_key(a, qr, ti) : (double)INFINITY;
    int I = valid ? ti : INT32_MAX;
#pragma unroll
    for (int k2 = 2; k2 <= 32; k2 <<= 1) {
#pragma unroll
        for (int j = k2 >> 1; j > 0; j >>= 1) {
            const double oK = __shfl_xor(K, j, 64);
            const int oI = __shfl_xor(I, j, 64);
            const bool up = (c & k2) == 0;           // this block of the network sorts ascending
            const bool keep_min = ((c & j) == 0) == up;
            const bool mine_less = K < oK || (K == oK && I < oI);
            if (keep_min != mine_less) K = oK, I = oI;
        }
    }
    const double tmax = sqrt(__longlong_as_double((long long)*tmax_bits)), qs = sqrt(qn);
    const double u = 5.9604644775390625e-08;  // 2^-24
    const double eps = 1.25 * (2.0 * gamma * qs * tmax + 2.0 * u * (tmax * tmax + 2.0 * qs * tmax));
    const double Kk = __shfl(K, (lane & 32) + a.k - 1, 64);
    const bool complete = n < kf;
    const bool proven = complete || (a.k <= n && Kk < (double)tau - eps);
    if (!qok) return;
    if (c == 0) flag[q] = proven ? 0 : 1;
    if (c < a.k) {
        const bool ok = c < n;
        const double d2 = ok ? K + qn : 0.0;
        dist[q * a.k + c] = ok ? sqrt(d2 > 0.0 ? d2 : 0.0) : (double)INFINITY;
        idx[q * a.k + c] = ok ? (int64_t)I + index_offset : (int64_t)-1;
    }
}

// float64 search results of the compacted queries: merge the per-split lists, recompute the keys of the
// k rows with the arithmetic of exact_key (so that both stages report bit-identical distances for the
// same pair), order by (key, row) and write to the query's own row
__global__ __launch_bounds__(128) void k_knn_finalize_compact(KnnArgs a, int64_t index_offset, double *dist, int64_t *idx) {
    const int64_t p = (int64_t)blockIdx.x * 128 + threadIdx.x;
    if (p >= (int64_t)*a.qcount) return;
    const int64_t q = a.qmap[p];
    const float *qr = a.query + q * a.ld_query;
    double qn = 0.0;
    for (int j = 0; j < a.d; ++j) qn = fma((double)qr[j], (double)qr[j], qn);
    double K[KF_MAX];
    int I[KF_MAX];
    int cur[64];
    for (int s = 0; s < a.n_split; ++s) cur[s] = 0;
    int n = 0;
    for (int o = 0; o < a.k; ++o) {
        double best = INFINITY;
        int bi = kIdxNone, bs = -1;
        for (int s = 0; s < a.n_split; ++s) {
            if (cur[s] >= a.k) continue;
            const int64_t off = ((int64_t)s * a.q_pad + p) * a.k + cur[s];
            const double v = a.part_key[off];
            const int vi = a.part_idx[off];
            if (vi != kIdxNone && (bs < 0 || v < best || (v == best && vi < bi))) best = v, bi = vi, bs = s;
        }
        if (bs < 0) break;
        cur[bs]++;
        I[n++] = bi;
    }
    for (int c = 0; c < n; ++c) K[c] = exact_key(a, qr, I[c]);
    sort_pairs(K, I, n);
    for (int o = 0; o < a.k; ++o) {
        const bool ok = o < n;
        const double d2 = ok ? K[o] + qn : 0.0;
        dist[q * a.k + o] = ok ? sqrt(d2 > 0.0 ? d2 : 0.0) : (double)INFINITY;
        idx[q * a.k + o] = ok ? (int64_t)I[o] + index_offset : (int64_t)-1;
    }
}

// merge the per-split lists of one query; recompute the keys of the k rows with the arithmetic of
// exact_key (every search path reports the same bits for the same pair, so per-shard results merge
// into the single-search result whatever path each shard took), order by (key, row), add |q|^2,
// sqrt, apply the index offset
__global__ __launch_bounds__(256) void k_knn_finalize(KnnArgs a, int64_t index_offset, double *dist, int64_t *idx) {
    int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (q >= a.n_query) return;
    const float *qr = a.query + q * a.ld_query;
    double qn = 0.0;
    for (int j = 0; j < a.d; ++j) {
        double v = (double)qr[j];
        qn = fma(v, v, qn);
    }
    double *K = dist + q * a.k;
    int64_t *I = idx + q * a.k;
    int cur[64];
    for (int s = 0; s < a.n_split; ++s) cur[s] = 0;
    int n = 0;
    for (int o = 0; o < a.k; ++o) {
        double best = INFINITY;
        int bi = kIdxNone, bs = -1;
        for (int s = 0; s < a.n_split; ++s) {
            if (cur[s] >= a.k) continue;
            int64_t off = ((int64_t)s * a.q_pad + q) * a.k + cur[s];
            double v = a.part_key[off];
            int vi = a.part_idx[off];
            if (vi != kIdxNone && (bs < 0 || v < best || (v == best && vi < bi))) {
                best = v;
                bi = vi;
                bs = s;
            }
        }
        if (bs < 0) break;
        cur[bs]++;
        I[n] = bi;
        K[n] = exact_key(a, qr, bi);
        ++n;
    }
    for (int i = 1; i < n; ++i) {  // nearly sorted already: the two arithmetics differ in the last bits only
        const double kv = K[i];
        const int64_t iv = I[i];
        int j = i - 1;
        while (j >= 0 && (K[j] > kv || (K[j] == kv && I[j] > iv))) {
            K[j + 1] = K[j], I[j + 1] = I[j];
            --j;
        }
        K[j + 1] = kv, I[j + 1] = iv;
    }
    for (int o = 0; o < a.k; ++o) {
        const bool ok = o < n;
        const double d2 = ok ? K[o] + qn : 0.0;
        K[o] = ok ? sqrt(d2 > 0.0 ? d2 : 0.0) : (double)INFINITY;
        I[o] = ok ? I[o] + index_offset : (int64_t)-1;
    }
}

// ---- generic path (any feature width, small problems): exact float64 keys, one block per
// query, k rounds of block-wide arg-min in (key, index) order -------------------------------
__global__ __launch_bounds__(256) void k_knn_generic(KnnArgs a, int64_t index_offset, double *keys /* [Q,N] */,
                                                     double *dist, int64_t *idx) {
    __shared__ double rk[4];
    __shared__ int ri[4];
    const int64_t q = blockIdx.x;
    const float *qr = a.query + q * a.ld_query;
    double qn = 0.0;
    for (int j = 0; j < a.d; ++j) qn = fma((double)qr[j], (double)qr[j], qn);
    double *row = keys + q * a.n_train;
    for (int64_t t = threadIdx.x; t < a.n_train; t += 256) {
        const float *tr = a.train + t * a.ld_train;
        double dot = 0.0;
        for (int j = 0; j < a.d; ++j) dot = fma((double)qr[j], (double)tr[j], dot);
        row[t] = fma(-2.0, dot, a.tnorm[t]);
    }
    __syncthreads();
    for (int o = 0; o < a.k; ++o) {
        double best = INFINITY;
        int bi = kIdxNone;
        for (int64_t t = threadIdx.x; t < a.n_train; t += 256) {
            double v = row[t];
            if (v < best || (v == best && (int)t < bi)) {
                best = v;
                bi = (int)t;
            }
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            double ov = __shfl_xor(best, off, 64);
            int oi = __shfl_xor(bi, off, 64);
            if (ov < best || (ov == best && oi < bi)) {
                best = ov;
                bi = oi;
            }
        }
        if ((threadIdx.x & 63) == 0) {
            rk[threadIdx.x >> 6] = best;
            ri[threadIdx.x >> 6] = bi;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int w = 1; w < 4; ++w)
                if (rk[w] < best || (rk[w] == best && ri[w] < bi)) {
                    best = rk[w];
                    bi = ri[w];
                }
            if (bi != kIdxNone) {
                double d2 = best + qn;
                dist[q * a.k + o] = sqrt(d2 > 0.0 ? d2 : 0.0);
                idx[q * a.k + o] = (int64_t)bi + index_offset;
                row[bi] = INFINITY;  // taken
            } else {
                dist[q * a.k + o] = INFINITY;
                idx[q * a.k + o] = -1;
            }
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void k_knn_merge(int64_t n_query, int k, int n_parts, const double *pd,
                                                   const int64_t *pi, double *dist, int64_t *idx) {
    int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (q >= n_query) return;
    int cur[64];
    for (int s = 0; s < n_parts; ++s) cur[s] = 0;
    for (int o = 0; o < k; ++o) {
        double best = INFINITY;
        int64_t bi = INT64_MAX;
        int bs = -1;
        for (int s = 0; s < n_parts; ++s) {
            if (cur[s] >= k) continue;
            int64_t off = ((int64_t)s * n_query + q) * k + cur[s];
            double v = pd[off];
            int64_t vi = pi[off];
            if (vi < 0) continue;
            if (bs < 0 || v < best || (v == best && vi < bi)) {
                best = v;
                bi = vi;
                bs = s;
            }
        }
        if (bs >= 0) cur[bs]++;
        dist[q * k + o] = bs >= 0 ? best : (double)INFINITY;
        idx[q * k + o] = bs >= 0 ? bi : (int64_t)-1;
    }
}

bool mfma_path(int d) { return d >= 1 && d <= DP; }

// The grid is (query blocks) x (train splits); 512 blocks are resident at once (256 CUs x 2) and run in
// batches of that size.  With s splits a block sweeps 1/s of the train rows, but every split starts each of
// its queries' lists afresh, and a fresh list of L entries takes about L (1 + ln(rows / L)) insertions before
// its threshold has settled.  Cost of a batch in swept-row equivalents: rows / s + g L (1 + ln(rows / (s L))),
// g = what one insertion per query costs a block against sweeping one row (measured: 29 us against 68 ns in the
// bf16 filter, about 430; about 150 in the slower float64 sweep); the split factor (at most 64: every split
// adds a list per query to merge) minimises batches x that.  (196 query blocks against 2.6 M rows: 5 splits,
// two batches of 1/5; against a 325,566-row shard: 2 splits, one batch of 1/2 -- thirteen splits would make
// five batches of mostly list building.)  Used for the float64 search and for the query blocks of the
// two-stage search that do not fill a whole batch.
void plan_split(int64_t n_train, int64_t n_query, int list_len, double insert_cost, int *n_split,
                int64_t *rows_per_split) {
    const int64_t resident = 512;
    const int64_t qblocks = (n_query + TQ - 1) / TQ;
    const int64_t tiles = (n_train + TT - 1) / TT;
    const int64_t max_s = tiles / 8 > 0 ? tiles / 8 : 1;  // at least 8 tiles per split
    const double L = (double)(list_len > 0 ? list_len : 1);
    int64_t s = 1;
    double best = 1e300;
    for (int64_t c = 1; c <= 64 && c <= max_s; ++c) {
        const double rows = (double)n_train / (double)c;
        const double fill = rows > L ? L * (1.0 + log(rows / L)) : rows;
        const double cost = (double)((qblocks * c + resident - 1) / resident) * (rows + insert_cost * fill);
        if (cost < best * 0.99) {  // a finer split has to buy at least 1 %
            best = cost;
            s = c;
        }
    }
    int64_t tps = (tiles + s - 1) / s;
    s = (tiles + tps - 1) / tps;
    *n_split = (int)s;
    *rows_per_split = tps * TT;
}
constexpr double kInsertCostFilter = 430.0, kInsertCostF64 = 150.0;

// ---- two-stage search: sizes ---------------------------------------------------------------------
// candidates kept per query by the filter: k plus a margin for near-ties
int filter_kf(int k) { return k + (k > 6 ? k : 6); }
size_t filter_lds_bytes(int kf) { return (size_t)3 * TT * (DP + 8) * 2 + (size_t)TT * 4 + (size_t)TQ * 4 + (size_t)TQ * kf * 8; }
// the filter pays off for small k (lists in LDS, two blocks per CU) and problems that fill the chip
bool filter_eligible(int64_t n_train, int64_t n_query, int d, int k) {
    const int kf = filter_kf(k);
    return d >= 1 && d <= DP && kf <= KF_MAX && kf <= n_train && filter_lds_bytes(kf) <= 80 * 1024 && n_train >= 4096 &&
           n_query >= 64;
}
// split of the float64 search over the queries the proof rejects
void plan_fallback(int64_t n_train, int64_t n_query, int *n_split, int64_t *rows_per_split);

struct FilterWs {
    float *fkey;
    int *fidx;
    unsigned long long *tmax;
    int *flag, *qmap, *qcount;
    void *cub;
    size_t cub_bytes;
    double *part_key;
    int *part_idx;
    int ns_x;
    int64_t rps_x, q_pad;
    // stage 1 runs in up to two parts: whole rounds of resident blocks sweep all train rows unsplit (a split
    // would build every query's list once per split: each list start inserts a few hundred candidates before
    // its threshold bites), the remaining query blocks are split so that they fill the chip once more
    int64_t q_main;  // queries of the unsplit part (a multiple of the resident blocks' queries, may be 0)
    int ns_tail;
    int64_t rps_tail;
};

size_t lds_bytes(int k, bool lds_lists) {
    return (size_t)TT * LDT * 4 + (size_t)TT * 8 + (size_t)TQ * 8 + (lds_lists ? (size_t)TQ * k * 12 : 0);
}
// long lists: registers per lane of the buffered sweep (buffer capacity 64 * NI >= 2k + one train tile), 0 = not
// served
int buf_ni(int k) {
    for (int ni = 2; ni <= 16; ni *= 2)
        if (64 * ni >= 2 * k + TT) return ni;
    return 0;
}
size_t buf_lds_bytes() { return (size_t)TT * LDT * 4 + (size_t)TT * 8 + (size_t)TQ * 8 + (size_t)TQ * 4; }

void plan_fallback(int64_t n_train, int64_t n_query, int *n_split, int64_t *rows_per_split) {
    // Few queries are expected here, so the train rows are cut finely (up to 16 ways: a block's sweep, the
    // least this stage can take, shrinks with the cut); the lists are sized for every query, which bounds
    // the cut (16 x Q x k x 12 bytes).
    const int64_t tiles = (n_train + TT - 1) / TT;
    int64_t s = tiles / 8 > 0 ? tiles / 8 : 1;  // at least 8 tiles per split
    if (s > 16) s = 16;
    const int64_t tps = (tiles + s - 1) / s;
    *n_split = (int)((tiles + tps - 1) / tps);
    *rows_per_split = tps * TT;
    (void)n_query;
}

// carve the two-stage workspace (after the train norms) out of `ar`; false when it does not fit
bool filter_workspace(magic::Arena &ar, int64_t n_train, int64_t n_query, int k, FilterWs *w) {
    const int kf = filter_kf(k);
    const int64_t resident = 512;  // 256 CUs x 2 blocks
    const int64_t qblocks = (n_query + TQ - 1) / TQ;
    w->q_main = qblocks / resident * resident * TQ;
    w->ns_tail = 1;
    w->rps_tail = 0;
    size_t lists = (size_t)w->q_main;  // list slots (x kf) the larger part needs
    if (n_query > w->q_main) {
        plan_split(n_train, n_query - w->q_main, kf, kInsertCostFilter, &w->ns_tail, &w->rps_tail);
        const size_t tail = (size_t)w->ns_tail * (size_t)((n_query - w->q_main + TQ - 1) / TQ * TQ);
        if (tail > lists) lists = tail;
    }
    plan_fallback(n_train, n_query, &w->ns_x, &w->rps_x);
    w->q_pad = qblocks * TQ;
    w->fkey = ar.take<float>(lists * kf);
    w->fidx = ar.take<int>(lists * kf);
    w->tmax = ar.take<unsigned long long>(1);
    w->flag = ar.take<int>((size_t)w->q_pad);
    w->qmap = ar.take<int>((size_t)w->q_pad);
    w->qcount = ar.take<int>(1);
    w->cub_bytes = 0;
    if (hipcub::DeviceSelect::Flagged(nullptr, w->cub_bytes, hipcub::CountingInputIterator<int>(0), (const int *)nullptr,
                                      (int *)nullptr, (int *)nullptr, (int)n_query) != hipSuccess)
        return false;
    w->cub = ar.take<char>(w->cub_bytes + 256);
    w->part_key = ar.take<double>((size_t)w->ns_x * w->q_pad * k);
    w->part_idx = ar.take<int>((size_t)w->ns_x * w->q_pad * k);
    return ar.ok();
}

}  // namespace

extern "C" {

size_t magic_knn_workspace_bytes(int64_t n_train, int64_t n_query, int32_t d, int32_t k) {
    if (n_train <= 0 || n_query <= 0 || d <= 0 || k <= 0) return 256;
    size_t b = magic::align_up((size_t)n_train * 8);
    if (filter_eligible(n_train, n_query, d, k)) {
        // measure the carve-up on a huge pretend buffer
        magic::Arena ar(reinterpret_cast<void *>(256), (size_t)1 << 60);
        FilterWs w;
        filter_workspace(ar, n_train, n_query, k, &w);
        return b + ar.off + 256;
    }
    if (mfma_path(d)) {
        int ns;
        int64_t rps;
        plan_split(n_train, n_query, k, kInsertCostF64, &ns, &rps);
        size_t qpad = (size_t)((n_query + TQ - 1) / TQ) * TQ;
        b += magic::align_up((size_t)ns * qpad * k * 8) + magic::align_up((size_t)ns * qpad * k * 4);
        if (lds_bytes(k, true) > 64 * 1024 && buf_ni(k) > 0) {
            const size_t cap = (size_t)64 * buf_ni(k);
            b += magic::align_up((size_t)ns * qpad * cap * 8) + magic::align_up((size_t)ns * qpad * cap * 4);
        }
    } else {
        b += magic::align_up((size_t)n_query * (size_t)n_train * 8);  // generic path key matrix
    }
    return b + 256;
}

int magic_knn_topk(int64_t n_train, int64_t n_query, int32_t d, int32_t k, const float *train, int64_t ld_train,
                   const float *query, int64_t ld_query, int64_t index_offset, double *dist, int64_t *idx,
                   void *workspace, size_t workspace_bytes, void *stream_) {
    if (n_train <= 0 || n_query < 0 || d <= 0 || k <= 0 || k > n_train) return MAGIC_ERR_INVALID_ARG;
    if (n_train >= INT32_MAX) return MAGIC_ERR_UNSUPPORTED;
    if (n_query == 0) return MAGIC_OK;
    if (!train || !query || !dist || !idx || ld_train < d || ld_query < d) return MAGIC_ERR_INVALID_ARG;
    hipStream_t stream = (hipStream_t)stream_;
    magic::Arena ar(workspace, workspace_bytes);
    KnnArgs a;
    a.n_train = n_train;
    a.n_query = n_query;
    a.d = d;
    a.k = k;
    a.train = train;
    a.ld_train = ld_train;
    a.query = query;
    a.ld_query = ld_query;
    double *tnorm = ar.take<double>(n_train);
    if (!tnorm) return MAGIC_ERR_WORKSPACE;
    a.tnorm = tnorm;
    if ((d & 3) == 0 && d <= SQ_MAXD && (ld_train & 3) == 0 && ((uintptr_t)train & 15) == 0)
        MAGIC_LAUNCH("knn_sqnorm", k_sqnorm_rows, magic::stride_grid(n_train, 128, 4096), 128,
                     (size_t)2 * 64 * (d + 1) * sizeof(float), stream, n_train, d, train, ld_train, tnorm);
    else
        MAGIC_LAUNCH("knn_sqnorm", k_sqnorm, magic::stride_grid(n_train, 256), 256, 0, stream, n_train, d, train, ld_train,
                     tnorm);
    a.qmap = nullptr;
    a.qcount = nullptr;
    if (filter_eligible(n_train, n_query, d, k)) {
        if ((ld_train % 4) != 0 || ((uintptr_t)train % 16) != 0) return MAGIC_ERR_INVALID_ARG;  // float4 row loads
        const int kf = filter_kf(k);
        FilterWs w;
        if (!filter_workspace(ar, n_train, n_query, k, &w)) return MAGIC_ERR_WORKSPACE;
        if (magic::ensure_dynamic_lds((const void *)k_knn_filter_bf16, 80 * 1024) != MAGIC_OK) return MAGIC_ERR_RUNTIME;
        if (hipMemsetAsync(w.tmax, 0, sizeof(unsigned long long), stream) != hipSuccess) return MAGIC_ERR_RUNTIME;
        MAGIC_LAUNCH("knn_tmax", k_tmax, magic::stride_grid(n_train, 256 * 16, 1024), 256, 0, stream, n_train, tnorm, w.tmax);
        // float32 accumulation of d products in 6 * d/16 matrix steps + the dropped cross products (2^-23 each)
        const double gamma = ((double)d + 28.0) * 5.9604644775390625e-08 + 1.1920928955078125e-07;
        a.part_key = nullptr;
        a.part_idx = nullptr;
        a.lds_lists = 1;
        // stage 1 (approximate candidates) and stage 2 (exact keys of the candidates, proof, output; rejected
        // queries flagged) of the queries [q0, q0 + nq), train rows cut `ns` ways
        auto two_stages = [&](int64_t q0, int64_t nq, int ns, int64_t rps) -> int {
            KnnArgs b = a;
            b.query = query + q0 * ld_query;
            b.n_query = nq;
            b.q_pad = (nq + TQ - 1) / TQ * TQ;
            b.n_split = ns;
            b.rows_per_split = rps;
            dim3 fgrid((unsigned)(b.q_pad / TQ), (unsigned)ns);
            MAGIC_LAUNCH("knn_filter_bf16x6_mfma", k_knn_filter_bf16, fgrid, 256, filter_lds_bytes(kf), stream, b, kf, w.fkey,
                         w.fidx);
            if (ns == 1 && kf <= 32 && k <= 32)
                MAGIC_LAUNCH("knn_refine", k_knn_refine_one, (unsigned)((nq * 32 + 255) / 256), 256, 0, stream, b, kf, w.fkey,
                             w.fidx, w.tmax, gamma, index_offset, dist + q0 * k, idx + q0 * k, w.flag + q0);
            else
                MAGIC_LAUNCH("knn_refine", k_knn_refine, (unsigned)((nq + 127) / 128), 128, 0, stream, b, kf, w.fkey, w.fidx,
                             w.tmax, gamma, index_offset, dist + q0 * k, idx + q0 * k, w.flag + q0);
            return MAGIC_OK;
        };
        if (w.q_main > 0) two_stages(0, w.q_main, 1, (n_train + TT - 1) / TT * TT);
        if (n_query > w.q_main) two_stages(w.q_main, n_query - w.q_main, w.ns_tail, w.rps_tail);
        a.q_pad = w.q_pad;
        // rejected queries -> compact list, searched in float64
        {
            magic::ProfScope ps("knn_compact", stream);
            size_t cb = w.cub_bytes;
            if (hipcub::DeviceSelect::Flagged(w.cub, cb, hipcub::CountingInputIterator<int>(0), w.flag, w.qmap, w.qcount,
                                              (int)n_query, stream) != hipSuccess)
                return MAGIC_ERR_RUNTIME;
        }
        a.qmap = w.qmap;
        a.qcount = w.qcount;
        a.n_split = w.ns_x;
        a.rows_per_split = w.rps_x;
        a.part_key = w.part_key;
        a.part_idx = w.part_idx;
        a.lds_lists = lds_bytes(k, true) <= 64 * 1024;
        dim3 xgrid((unsigned)(a.q_pad / TQ), (unsigned)a.n_split);
        size_t sh = lds_bytes(k, a.lds_lists);
        if (a.lds_lists)
            MAGIC_LAUNCH("knn_topk_f64_mfma", (k_knn_mfma<false>), xgrid, 256, sh, stream, a);
        else
            MAGIC_LAUNCH("knn_topk_f64_mfma", (k_knn_mfma<true>), xgrid, 256, sh, stream, a);
        MAGIC_LAUNCH("knn_finalize", k_knn_finalize_compact, (unsigned)((n_query + 127) / 128), 128, 0, stream, a,
                     index_offset, dist, idx);
    } else if (mfma_path(d)) {
        if ((ld_train % 4) != 0 || ((uintptr_t)train % 16) != 0) return MAGIC_ERR_INVALID_ARG;  // float4 row loads
        plan_split(n_train, n_query, k, kInsertCostF64, &a.n_split, &a.rows_per_split);
        a.q_pad = (n_query + TQ - 1) / TQ * TQ;
        a.part_key = ar.take<double>((size_t)a.n_split * a.q_pad * k);
        a.part_idx = ar.take<int>((size_t)a.n_split * a.q_pad * k);
        if (!ar.ok() || !a.part_key || !a.part_idx) return MAGIC_ERR_WORKSPACE;
        a.lds_lists = lds_bytes(k, true) <= 64 * 1024;
        dim3 grid((unsigned)(a.q_pad / TQ), (unsigned)a.n_split);
        size_t sh = lds_bytes(k, a.lds_lists);
        const int ni = a.lds_lists ? 0 : buf_ni(k);
        if (a.lds_lists) {
            MAGIC_LAUNCH("knn_topk_f64_mfma", (k_knn_mfma<false>), grid, 256, sh, stream, a);
        } else if (ni > 0) {  // long lists: unsorted candidate buffers, sorted by their wave when they fill
            const size_t cap = (size_t)64 * ni;
            double *bkey = ar.take<double>((size_t)a.n_split * a.q_pad * cap);
            int *bidx = ar.take<int>((size_t)a.n_split * a.q_pad * cap);
            if (!ar.ok() || !bkey || !bidx) return MAGIC_ERR_WORKSPACE;
            switch (ni) {
                case 2: MAGIC_LAUNCH("knn_topk_f64_mfma", (k_knn_mfma_buf<2>), grid, 256, buf_lds_bytes(), stream, a, bkey, bidx); break;
                case 4: MAGIC_LAUNCH("knn_topk_f64_mfma", (k_knn_mfma_buf<4>), grid, 256, buf_lds_bytes(), stream, a, bkey, bidx); break;
                case 8: MAGIC_LAUNCH("knn_topk_f64_mfma", (k_knn_mfma_buf<8>), grid, 256, buf_lds_bytes(), stream, a, bkey, bidx); break;
                default: MAGIC_LAUNCH("knn_topk_f64_mfma", (k_knn_mfma_buf<16>), grid, 256, buf_lds_bytes(), stream, a, bkey, bidx); break;
            }
        } else {
            MAGIC_LAUNCH("knn_topk_f64_mfma", (k_knn_mfma<true>), grid, 256, sh, stream, a);
        }
        MAGIC_LAUNCH("knn_finalize", k_knn_finalize, (unsigned)((n_query + 255) / 256), 256, 0, stream, a, index_offset,
                     dist, idx);
    } else {
        double *keys = ar.take<double>((size_t)n_query * n_train);
        if (!ar.ok() || !keys) return MAGIC_ERR_WORKSPACE;
        MAGIC_LAUNCH("knn_generic_f64", k_knn_generic, (unsigned)n_query, 256, 0, stream, a, index_offset, keys, dist,
                     idx);
    }
    return magic::launch_status();
}

int magic_knn_merge(int64_t n_query, int32_t k, int32_t n_parts, const double *part_dist, const int64_t *part_idx,
                    double *dist, int64_t *idx, void *stream_) {
    if (n_query < 0 || k <= 0 || n_parts <= 0 || n_parts > 64) return MAGIC_ERR_INVALID_ARG;
    if (n_query == 0) return MAGIC_OK;
    if (!part_dist || !part_idx || !dist || !idx) return MAGIC_ERR_INVALID_ARG;
    hipStream_t stream = (hipStream_t)stream_;
    MAGIC_LAUNCH("knn_merge", k_knn_merge, (unsigned)((n_query + 255) / 256), 256, 0, stream, n_query, k, n_parts,
                 part_dist, part_idx, dist, idx);
    return magic::launch_status();
}

}  // extern "C"
