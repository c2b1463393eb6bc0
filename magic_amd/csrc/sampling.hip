// Acceptance test of uniformly drawn negative pairs (dgl.sampling.global_uniform_negative_sampling at
// model/autoencoder.py:251 of the reference): of m candidate (src, dst) pairs, in draw order, the first
// `num_samples` that are no edge of the graph, no self-loop and no repeat of an earlier candidate.
//
// Two launches:
//   edge test    one thread per candidate over the whole chip: the shorter of the source's out-segment and the
//                destination's in-segment is scanned (one workgroup doing this for every candidate spends its time
//                waiting on dependent loads: 250 us against 10);
//   repeats and  one workgroup: an open-addressing table in LDS keyed by src * N + dst keeps, per key, the smallest
//   the count    candidate index (compare-and-swap for the empty slot, atomic minimum afterwards): whoever is not the
//                index stored for its key is a repeat -- the result does not depend on the insertion order; then
//                ballot / population count per wavefront, wavefront totals through LDS, 1024 candidates a round.
#include "common.h"

namespace {

constexpr int NEG_THREADS = 1024;
constexpr int NEG_MAX = MAGIC_NEGATIVE_FILTER_MAX;  // candidates per call
constexpr int NEG_PER = NEG_MAX / NEG_THREADS;      // rounds
constexpr int NEG_SLOTS = 2 * NEG_MAX;              // table slots: load factor at most 1/2
static_assert(NEG_MAX % NEG_THREADS == 0 && (NEG_SLOTS & (NEG_SLOTS - 1)) == 0, "table size must be a power of two");

struct NegArgs {
    int64_t n;
    const int32_t *out_ptr, *out_dst, *in_ptr, *in_src;
    int m;
    const int64_t *u, *v;
    int num_samples;
    uint8_t *ok;
};

static_assert(NEG_MAX <= 65536, "a candidate index must fit the low half of a table slot");

// pass[i] = 1 when candidate i names two different nodes of the graph that no edge joins
__global__ __launch_bounds__(256) void k_negative_edges(NegArgs a) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= a.m) return;
    const int64_t u = a.u[i], v = a.v[i];
    bool pass = false;
    if (u >= 0 && u < a.n && v >= 0 && v < a.n && u != v) {
        const int ob = a.out_ptr[u], oe = a.out_ptr[u + 1], ib = a.in_ptr[v], ie = a.in_ptr[v + 1];
        bool is_edge = false;
        if (oe - ob <= ie - ib) {
            for (int e = ob; e < oe; ++e) is_edge |= a.out_dst[e] == (int32_t)v;
        } else {
            for (int e = ib; e < ie; ++e) is_edge |= a.in_src[e] == (int32_t)u;
        }
        pass = !is_edge;
    }
    a.ok[i] = pass ? 1 : 0;
}

// in: ok[i] = the edge test's verdict; out: ok[i] = accepted
__global__ __launch_bounds__(NEG_THREADS) void k_negative_filter(NegArgs a) {
    extern __shared__ int slots[];
    __shared__ int wave_cnt[NEG_THREADS / 64];
    __shared__ int accepted;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // this thread's candidates (all loads in flight together)
    int64_t key[NEG_PER];
    unsigned good = 0;
#pragma unroll
    for (int j = 0; j < NEG_PER; ++j) {
        const int i = min(j * NEG_THREADS + tid, a.m - 1);
        const int64_t u = a.u[i], v = a.v[i];
        key[j] = (u >= 0 && u < a.n && v >= 0 && v < a.n && j * NEG_THREADS + tid < a.m) ? u * a.n + v : -1;
        if (a.ok[i]) good |= 1u << j;
    }
    // a slot holds (15-bit tag of the key) << 16 | candidate index: a probe that meets another key sees another tag
    // (all but one time in 32,768) and moves on without fetching that candidate's pair from memory
    for (int i = tid; i < NEG_SLOTS; i += NEG_THREADS) slots[i] = -1;
    if (tid == 0) accepted = 0;
    __syncthreads();
    volatile int *vslots = slots;
    int where[NEG_PER];
#pragma unroll
    for (int j = 0; j < NEG_PER; ++j) {
        const int i = j * NEG_THREADS + tid;
        where[j] = 0;
        if (key[j] < 0) continue;  // no node pair at all
        const uint64_t mix = (uint64_t)key[j] * 0x9E3779B97F4A7C15ull;
        int h = (int)(mix >> 40) & (NEG_SLOTS - 1);
        const int tag = (int)((mix >> 20) & 0x7fffu);  // 15 bits: a packed slot stays non-negative
        const int mine = (tag << 16) | i;
        while (true) {
            int cur = vslots[h];
            if (cur < 0) {
                cur = atomicCAS(&slots[h], -1, mine);
                if (cur < 0) break;  // claimed the empty slot
            }
            if ((cur >> 16) == tag) {
                const int c = cur & 0xffff;
                if (a.u[c] * a.n + a.v[c] == key[j]) {  // a slot never changes its key, only to a smaller index
                    atomicMin(&slots[h], mine);
                    break;
                }
            }
            h = (h + 1) & (NEG_SLOTS - 1);
        }
        where[j] = h;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NEG_PER; ++j) {
        const int i = j * NEG_THREADS + tid;
        const bool ok = ((good >> j) & 1u) && key[j] >= 0 && (slots[where[j]] & 0xffff) == i;
        const unsigned long long b = __ballot(ok);
        if (lane == 0) wave_cnt[wave] = __popcll(b);
        __syncthreads();
        int before = accepted, all = 0;
        for (int w = 0; w < NEG_THREADS / 64; ++w) {
            const int c = wave_cnt[w];
            if (w < wave) before += c;
            all += c;
        }
        const int rank = before + __popcll(b & ((1ull << lane) - 1ull)) + 1;  // this candidate's place among the passing
        if (i < a.m) a.ok[i] = (ok && rank <= a.num_samples) ? 1 : 0;
        __syncthreads();
        if (tid == 0) accepted += all;
        __syncthreads();
    }
}

}  // namespace

extern "C" int magic_negative_filter(int64_t n_nodes, const int32_t *out_ptr, const int32_t *out_dst,
                                     const int32_t *in_ptr, const int32_t *in_src, int32_t n_candidates,
                                     const int64_t *cand_src, const int64_t *cand_dst, int32_t num_samples,
                                     uint8_t *accept, void *stream_) {
    if (n_nodes < 0 || n_candidates < 0 || num_samples < 0) return MAGIC_ERR_INVALID_ARG;
    if (n_candidates == 0) return MAGIC_OK;
    if (n_candidates > NEG_MAX) return MAGIC_ERR_UNSUPPORTED;
    if (!out_ptr || !in_ptr || !cand_src || !cand_dst || !accept) return MAGIC_ERR_INVALID_ARG;
    hipStream_t stream = (hipStream_t)stream_;
    constexpr size_t lds = (size_t)NEG_SLOTS * sizeof(int);
    if (magic::ensure_dynamic_lds((const void *)k_negative_filter, lds) != MAGIC_OK) return MAGIC_ERR_RUNTIME;
    NegArgs a;
    a.n = n_nodes;
    a.out_ptr = out_ptr, a.out_dst = out_dst, a.in_ptr = in_ptr, a.in_src = in_src;
    a.m = n_candidates;
    a.u = cand_src, a.v = cand_dst;
    a.num_samples = num_samples;
    a.ok = accept;
    MAGIC_LAUNCH("negative_edges", k_negative_edges, (n_candidates + 255) / 256, 256, 0, stream, a);
    MAGIC_LAUNCH("negative_filter", k_negative_filter, 1, NEG_THREADS, lds, stream, a);
    return magic::launch_status();
}
