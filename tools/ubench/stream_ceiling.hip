// Memory-system ceiling for the fused projection-backward access pattern: per 32-row tile a wave
// reads a [32,72] tile of dY, three [32,64] operand tiles (two of them column blocks of 192-wide
// buffers), a [32,64] byte mask, and writes a [32,64] tile.  No LDS, no MFMA: what the loads and
// stores alone cost.   hipcc -O3 --offload-arch=gfx950 stream_ceiling.hip -o stream_ceiling
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x)                                                                   \
    do {                                                                        \
        hipError_t e_ = (x);                                                    \
        if (e_ != hipSuccess) {                                                 \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            exit(1);                                                            \
        }                                                                       \
    } while (0)

struct Args {
    int64_t n;
    const float *dy;   // [N,72]
    const float *add;  // [N,64]
    const float *add2; // [N,192] block 0
    const float *act;  // [N,192] block 0
    const uint8_t *mask;
    float *out;        // [N,64]
    int what;          // bit0: dy, bit1: add, bit2: add2, bit3: act, bit4: mask
};

template <int TPW>  // tiles in flight per wave
__global__ __launch_bounds__(256) void k_ceiling(Args a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t n_tiles = (a.n + 31) / 32;
    const int64_t stride = (int64_t)gridDim.x * 4;
    for (int64_t t = (int64_t)blockIdx.x * 4 + wave; t < n_tiles; t += stride) {
        float4 acc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (a.what & 1) {
            // 32 rows x 72 floats = 576 float4 = 9 per lane, contiguous
            const float4 *p = reinterpret_cast<const float4 *>(a.dy + t * 32 * 72);
#pragma unroll
            for (int i = 0; i < 9; ++i) {
                const int64_t idx = lane + 64 * i;
                if (t * 32 * 18 + idx < a.n * 18) {
                    float4 v = p[idx];
                    acc[i & 7].x += v.x, acc[i & 7].y += v.y, acc[i & 7].z += v.z, acc[i & 7].w += v.w;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int idx = lane + 64 * i;
            const int r = idx >> 4, c4 = (idx & 15) * 4;
            const int64_t row = t * 32 + r;
            if (row >= a.n) continue;
            float4 v = acc[i];
            if (a.what & 2) {
                float4 q = *reinterpret_cast<const float4 *>(a.add + row * 64 + c4);
                v.x += q.x, v.y += q.y, v.z += q.z, v.w += q.w;
            }
            if (a.what & 16) {
                uint32_t m = *reinterpret_cast<const uint32_t *>(a.mask + row * 64 + c4);
                v.x = (m & 0xff) ? v.x : 0.f, v.y = (m & 0xff00) ? v.y : 0.f, v.z = (m & 0xff0000) ? v.z : 0.f,
                v.w = (m & 0xff000000) ? v.w : 0.f;
            }
            if (a.what & 4) {
                float4 q = *reinterpret_cast<const float4 *>(a.add2 + row * 192 + c4);
                v.x += q.x, v.y += q.y, v.z += q.z, v.w += q.w;
            }
            if (a.what & 8) {
                float4 q = *reinterpret_cast<const float4 *>(a.act + row * 192 + c4);
                v.x *= q.x > 0.f ? 1.f : 0.25f, v.y *= q.y > 0.f ? 1.f : 0.25f, v.z *= q.z > 0.f ? 1.f : 0.25f,
                    v.w *= q.w > 0.f ? 1.f : 0.25f;
            }
            *reinterpret_cast<float4 *>(a.out + row * 64 + c4) = v;
        }
    }
}

// Same bytes, but every lane owns one row of the tile (lane & 31) and one half of its columns
// (lane >> 5): the access shape of MFMA operands / results taken straight from global memory
// (16-byte pieces, 32 rows per instruction).
__global__ __launch_bounds__(256) void k_lane_row(Args a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lr = lane & 31, hf = lane >> 5;
    const int64_t n_tiles = (a.n + 31) / 32;
    const int64_t stride = (int64_t)gridDim.x * 4;
    for (int64_t t = (int64_t)blockIdx.x * 4 + wave; t < n_tiles; t += stride) {
        const int64_t row = t * 32 + lr;
        if (row >= a.n) continue;
        float4 acc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (a.what & 1) {
            const float4 *p = reinterpret_cast<const float4 *>(a.dy + row * 72 + hf * 36);
#pragma unroll
            for (int i = 0; i < 9; ++i) {
                float4 v = p[i];
                acc[i & 7].x += v.x, acc[i & 7].y += v.y, acc[i & 7].z += v.z, acc[i & 7].w += v.w;
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int c4 = (i >> 2) * 32 + (i & 3) * 8 + hf * 4;
            float4 v = acc[i];
            if (a.what & 2) {
                float4 q = *reinterpret_cast<const float4 *>(a.add + row * 64 + c4);
                v.x += q.x, v.y += q.y, v.z += q.z, v.w += q.w;
            }
            if (a.what & 16) {
                uint32_t m = *reinterpret_cast<const uint32_t *>(a.mask + row * 64 + c4);
                v.x = (m & 0xff) ? v.x : 0.f, v.y = (m & 0xff00) ? v.y : 0.f, v.z = (m & 0xff0000) ? v.z : 0.f,
                v.w = (m & 0xff000000) ? v.w : 0.f;
            }
            if (a.what & 4) {
                float4 q = *reinterpret_cast<const float4 *>(a.add2 + row * 192 + c4);
                v.x += q.x, v.y += q.y, v.z += q.z, v.w += q.w;
            }
            if (a.what & 8) {
                float4 q = *reinterpret_cast<const float4 *>(a.act + row * 192 + c4);
                v.x *= q.x > 0.f ? 1.f : 0.25f, v.y *= q.y > 0.f ? 1.f : 0.25f, v.z *= q.z > 0.f ? 1.f : 0.25f,
                    v.w *= q.w > 0.f ? 1.f : 0.25f;
            }
            *reinterpret_cast<float4 *>(a.out + row * 64 + c4) = v;
        }
    }
}

int main(int argc, char **argv) {
    const int64_t n = argc > 1 ? atoll(argv[1]) : 700000;
    Args a;
    a.n = n;
    float *dy, *add, *add2, *act, *out;
    uint8_t *mask;
    CK(hipMalloc(&dy, n * 72 * 4));
    CK(hipMalloc(&add, n * 64 * 4));
    CK(hipMalloc(&add2, n * 192 * 4));
    CK(hipMalloc(&act, n * 192 * 4));
    CK(hipMalloc(&out, n * 64 * 4));
    CK(hipMalloc(&mask, n * 64));
    CK(hipMemset(dy, 0, n * 72 * 4));
    CK(hipMemset(add, 0, n * 64 * 4));
    CK(hipMemset(add2, 0, n * 192 * 4));
    CK(hipMemset(act, 0, n * 192 * 4));
    CK(hipMemset(mask, 1, n * 64));
    a.dy = dy, a.add = add, a.add2 = add2, a.act = act, a.mask = mask, a.out = out;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int whats[] = {0, 1, 3, 7, 15, 31};
    const double bytes_of[] = {256, 256 + 288, 256 + 288 + 256, 256 + 288 + 512, 256 + 288 + 768, 256 + 288 + 768 + 64};
    for (int grid : {512, 1024}) {
        for (int w = 0; w < 6; ++w) {
            a.what = whats[w];
            for (int i = 0; i < 3; ++i) k_ceiling<1><<<grid, 256>>>(a);
            CK(hipEventRecord(e0));
            for (int i = 0; i < 20; ++i) k_ceiling<1><<<grid, 256>>>(a);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            const double us = ms / 20 * 1e3;
            printf("grid %5d what %2d: %7.1f us  %6.2f TB/s\n", grid, a.what, us, bytes_of[w] * n / us / 1e6);
        }
    }
    for (int grid : {512, 1024, 2048, 4096}) {
        for (int w = 0; w < 6; ++w) {
            a.what = whats[w];
            for (int i = 0; i < 3; ++i) k_lane_row<<<grid, 256>>>(a);
            CK(hipEventRecord(e0));
            for (int i = 0; i < 20; ++i) k_lane_row<<<grid, 256>>>(a);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            const double us = ms / 20 * 1e3;
            printf("lane-row grid %5d what %2d: %7.1f us  %6.2f TB/s\n", grid, a.what, us, bytes_of[w] * n / us / 1e6);
        }
    }
    return 0;
}
