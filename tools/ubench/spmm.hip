// Microbenchmark: node-parallel segmented row aggregation (the skeleton of gat_fwd_light) with
// pieces switched on one at a time, to see which one costs the bandwidth.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>

struct Args { int N; const int* ptr; const int* src; const float* hs; const float* x; const float* res; float* out; float* pre; float* attn; };

template <int MODE>  // bit0: write out, bit1: read x + write attn, bit2: read res, bit3: write pre
__global__ __launch_bounds__(256) void k_spmm(Args a) {
    const int g = threadIdx.x & 15, slot = threadIdx.x >> 4, h = g >> 2;
    for (int v = blockIdx.x * 16 + slot; v < a.N; v += gridDim.x * 16) {
        const int b = a.ptr[v], e = a.ptr[v + 1];
        float4 acc = make_float4(0, 0, 0, 0);
        for (int i = b; i < e; i += 4) {
            float4 r[4]; float w[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) { int ii = min(i + j, e - 1); r[j] = *reinterpret_cast<const float4*>(a.hs + (long)a.src[ii] * 64 + g * 4);
                w[j] = (MODE & 2) ? a.x[(long)ii * 4 + h] : 0.25f; }
#pragma unroll
            for (int j = 0; j < 4; ++j) if (i + j < e) { acc.x += w[j] * r[j].x; acc.y += w[j] * r[j].y; acc.z += w[j] * r[j].z; acc.w += w[j] * r[j].w;
                if ((MODE & 2) && (g & 3) == 0) a.attn[(long)(i + j) * 4 + h] = w[j]; }
        }
        if (MODE & 4) { float4 q = *reinterpret_cast<const float4*>(a.res + (long)v * 64 + g * 4); acc.x += q.x; acc.y += q.y; acc.z += q.z; acc.w += q.w; }
        if (MODE & 8) *reinterpret_cast<float4*>(a.pre + (long)v * 64 + g * 4) = acc;
        if (MODE & 1) *reinterpret_cast<float4*>(a.out + (long)v * 64 + g * 4) = acc;
        else if (acc.x == 12345.f) a.out[0] = 1.f;
    }
}

int main(int argc, char** argv) {
    int N = 700000; long E = 0;
    std::mt19937 rng(1);
    std::vector<int> ptr(N + 1, 0);
    for (int v = 0; v < N; ++v) { int d = (rng() % 100 < 40) ? 0 : 1 + (int)(rng() % 13); ptr[v + 1] = ptr[v] + d; }
    E = ptr[N];
    std::vector<int> src(E); for (long i = 0; i < E; ++i) src[i] = rng() % N;
    printf("N %d E %ld\n", N, E);
    int *dptr, *dsrc; float *hs, *x, *res, *out, *pre, *attn, *big;
    hipMalloc(&dptr, (N + 1) * 4); hipMalloc(&dsrc, E * 4); hipMalloc(&hs, (long)N * 256); hipMalloc(&x, E * 16); hipMalloc(&res, (long)N * 256);
    hipMalloc(&out, (long)N * 256); hipMalloc(&pre, (long)N * 256); hipMalloc(&attn, E * 16); hipMalloc(&big, 1L << 30);
    hipMemcpy(dptr, ptr.data(), (N + 1) * 4, hipMemcpyHostToDevice); hipMemcpy(dsrc, src.data(), E * 4, hipMemcpyHostToDevice);
    hipMemset(hs, 0, (long)N * 256); hipMemset(x, 0, E * 16); hipMemset(res, 0, (long)N * 256);
    Args a{N, dptr, dsrc, hs, x, res, out, pre, attn};
    hipEvent_t ea, eb; hipEventCreate(&ea); hipEventCreate(&eb);
    auto run = [&](auto kern, const char* name, int grid, double bytes) {
        float best = 1e9;
        for (int it = 0; it < 4; ++it) {
            hipMemsetAsync(big, it, 1L << 30, 0);
            hipEventRecord(ea, 0); hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, a); hipEventRecord(eb, 0); hipEventSynchronize(eb);
            float ms; hipEventElapsedTime(&ms, ea, eb); if (ms < best) best = ms;
        }
        printf("%-28s grid %5d: %7.1f us  %.2f TB/s\n", name, grid, best * 1e3, bytes / (best * 1e-3) / 1e12);
    };
    double gb = E * 256.0 + N * 8.0 + E * 4.0;
    for (int grid : {2048, 8192, 43750}) {
        run(k_spmm<0>, "gather only", grid, gb);
        run(k_spmm<1>, "+out", grid, gb + N * 256.0);
        run(k_spmm<3>, "+out +x/attn", grid, gb + N * 256.0 + E * 32.0);
        run(k_spmm<7>, "+out +x/attn +res", grid, gb + N * 512.0 + E * 32.0);
        run(k_spmm<15>, "+out +x/attn +res +pre", grid, gb + N * 768.0 + E * 32.0);
    }
    return 0;
}
