// Keep masks for feature dropout (model/gat.py:283 of the reference: nn.Dropout(feat_drop)):
// one byte per element, 1 = keep, drawn with Philox4x32-10 keyed by (seed, element block) so
// that a mask depends only on the seed and the element index, never on the launch geometry.
#include "common.h"

namespace {

struct Philox {
    uint32_t c[4];
    uint32_t k[2];
};

__device__ __forceinline__ void philox_round(Philox &p) {
    // one 32x32->64 multiply per product (v_mad_u64_u32) instead of a high and a low multiply
    const uint64_t m0 = (uint64_t)0xD2511F53u * p.c[0], m1 = (uint64_t)0xCD9E8D57u * p.c[2];
    const uint32_t hi0 = (uint32_t)(m0 >> 32), lo0 = (uint32_t)m0;
    const uint32_t hi1 = (uint32_t)(m1 >> 32), lo1 = (uint32_t)m1;
    const uint32_t n0 = hi1 ^ p.c[1] ^ p.k[0], n2 = hi0 ^ p.c[3] ^ p.k[1];
    p.c[0] = n0, p.c[1] = lo1, p.c[2] = n2, p.c[3] = lo0;
    p.k[0] += 0x9E3779B9u, p.k[1] += 0xBB67AE85u;
}

__device__ __forceinline__ void philox10(uint64_t seed, uint64_t ctr, uint32_t (&out)[4]) {
    Philox p;
    p.c[0] = (uint32_t)ctr, p.c[1] = (uint32_t)(ctr >> 32), p.c[2] = 0x6d616769u, p.c[3] = 0u;
    p.k[0] = (uint32_t)seed, p.k[1] = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) philox_round(p);
    out[0] = p.c[0], out[1] = p.c[1], out[2] = p.c[2], out[3] = p.c[3];
}

// every thread fills 16 consecutive bytes: two Philox blocks, 16 random bits per element
// (the seed comes by value, or from device memory when `seed_dev` is given: a launch recorded in a
// HIP graph then draws a fresh mask at every replay once the caller has rewritten that word)
__global__ __launch_bounds__(256) void k_dropout_mask(int64_t n, uint32_t thr16, uint64_t seed, const uint64_t *seed_dev,
                                                      uint8_t *out) {
    if (seed_dev) seed = seed_dev[0];
    const int64_t n16 = (n + 15) / 16;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (int64_t)gridDim.x * 256) {
        uint32_t r[8];
        {
            uint32_t t[4];
            philox10(seed, 2 * (uint64_t)i, t);
            r[0] = t[0], r[1] = t[1], r[2] = t[2], r[3] = t[3];
            philox10(seed, 2 * (uint64_t)i + 1, t);
            r[4] = t[0], r[5] = t[1], r[6] = t[2], r[7] = t[3];
        }
        uint32_t w[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t a = r[2 * q], b = r[2 * q + 1];
            w[q] = ((a & 0xffffu) < thr16 ? 1u : 0u) | ((a >> 16) < thr16 ? 0x100u : 0u) |
                   ((b & 0xffffu) < thr16 ? 0x10000u : 0u) | ((b >> 16) < thr16 ? 0x1000000u : 0u);
        }
        if ((i + 1) * 16 <= n) {
            *reinterpret_cast<uint4 *>(out + i * 16) = make_uint4(w[0], w[1], w[2], w[3]);
        } else {
            for (int64_t j = i * 16; j < n; ++j) out[j] = (uint8_t)((w[(j & 15) >> 2] >> (8 * (j & 3))) & 1u);
        }
    }
}

}  // namespace

extern "C" {

static int dropout_launch(int64_t n_elems, float keep_prob, uint64_t seed, const uint64_t *seed_dev, uint8_t *mask,
                          void *stream_) {
    if (n_elems < 0 || !(keep_prob >= 0.f && keep_prob <= 1.f)) return MAGIC_ERR_INVALID_ARG;
    if (n_elems == 0) return MAGIC_OK;
    if (!mask || ((uintptr_t)mask & 15)) return MAGIC_ERR_INVALID_ARG;
    // keep iff a 16-bit draw is below keep_prob * 2^16 (probability resolution 1.5e-5)
    uint32_t thr = (uint32_t)(keep_prob * 65536.f + 0.5f);
    if (thr > 65536u) thr = 65536u;
    MAGIC_LAUNCH("dropout_mask", k_dropout_mask, magic::stride_grid((n_elems + 15) / 16, 256, 4096), 256, 0,
                 (hipStream_t)stream_, n_elems, thr, seed, seed_dev, mask);
    return magic::launch_status();
}

int magic_dropout_mask(int64_t n_elems, float keep_prob, uint64_t seed, uint8_t *mask, void *stream) {
    return dropout_launch(n_elems, keep_prob, seed, nullptr, mask, stream);
}

int magic_dropout_mask_dev(int64_t n_elems, float keep_prob, const uint64_t *seed_dev, uint8_t *mask, void *stream) {
    if (!seed_dev) return MAGIC_ERR_INVALID_ARG;
    return dropout_launch(n_elems, keep_prob, 0, seed_dev, mask, stream);
}

}  // extern "C"
