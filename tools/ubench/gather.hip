// Microbenchmark: random gather of 256-byte rows (float4 per lane, 16 lanes per row) from a
// table of N rows, U rows in flight per lane group; reports effective TB/s of gathered bytes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>

template <int U>
__global__ __launch_bounds__(256) void k_gather(const float* __restrict__ tab, const int* __restrict__ idx, long n_idx,
                                                float* __restrict__ out) {
    const int g = threadIdx.x & 15;
    long grp = ((long)blockIdx.x * 256 + threadIdx.x) >> 4;
    long ngrp = ((long)gridDim.x * 256) >> 4;
    float4 acc = make_float4(0, 0, 0, 0);
    for (long i = grp * U; i + U <= n_idx; i += ngrp * U) {
        int u[U];
        float4 r[U];
#pragma unroll
        for (int j = 0; j < U; ++j) u[j] = idx[i + j];
#pragma unroll
        for (int j = 0; j < U; ++j) r[j] = *reinterpret_cast<const float4*>(tab + (long)u[j] * 64 + g * 4);
#pragma unroll
        for (int j = 0; j < U; ++j) { acc.x += r[j].x; acc.y += r[j].y; acc.z += r[j].z; acc.w += r[j].w; }
    }
    *reinterpret_cast<float4*>(out + (((long)blockIdx.x * 256 + threadIdx.x) * 4)) = acc;
}

int main(int argc, char** argv) {
    long N = argc > 1 ? atol(argv[1]) : 700000;
    long E = argc > 2 ? atol(argv[2]) : 3000000;
    int sorted = argc > 3 ? atoi(argv[3]) : 0;
    float* tab; int* idx; float* out;
    hipMalloc(&tab, N * 64 * 4); hipMalloc(&idx, E * 4); hipMalloc(&out, 8192L * 256 * 16);
    hipMemset(tab, 0, N * 64 * 4);
    std::vector<int> h(E); std::mt19937 rng(1);
    for (long i = 0; i < E; ++i) h[i] = rng() % N;
    if (sorted) { /* windowed locality: indices within +-4096 of position */ for (long i = 0; i < E; ++i) h[i] = (int)(((i * N) / E + (rng() % 8192)) % N); }
    hipMemcpy(idx, h.data(), E * 4, hipMemcpyHostToDevice);
    float* big; hipMalloc(&big, 1L << 30);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    auto run = [&](auto kern, const char* name, int grid) {
        float best = 1e9;
        for (int it = 0; it < 5; ++it) {
            hipMemsetAsync(big, it, 1L << 30, 0);  // flush caches
            hipEventRecord(a, 0);
            hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, tab, idx, E, out);
            hipEventRecord(b, 0); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
        }
        printf("%-10s grid %5d: %.1f us  %.2f TB/s (rows) \n", name, grid, best * 1e3, E * 256.0 / (best * 1e-3) / 1e12);
    };
    for (int grid : {2048, 4096, 8192}) {
        run(k_gather<1>, "U=1", grid); run(k_gather<2>, "U=2", grid); run(k_gather<4>, "U=4", grid); run(k_gather<8>, "U=8", grid);
    }
    return 0;
}
