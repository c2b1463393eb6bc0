// Shared helpers for the libmagic_hip.so translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/magic_hip.h"

#define MAGIC_WAVE 64

namespace magic {

// ---- per-kernel event timing (prof.hip) ----------------------------------------------
bool prof_on();
void prof_begin(const char *name, hipStream_t s);
void prof_end(hipStream_t s);
const char *prof_name(const char *base, int a, int b);  // "base<a,b>", static storage

struct ProfScope {
    hipStream_t s;
    bool on;
    ProfScope(const char *name, hipStream_t st) : s(st), on(prof_on()) {
        if (on) prof_begin(name, s);
    }
    ~ProfScope() {
        if (on) prof_end(s);
    }
};

// Raise a kernel's dynamic-LDS limit to at least `bytes` on the CURRENT device (the attribute is kept per
// device, so the largest request is remembered per (kernel, device)).  Returns MAGIC_OK or MAGIC_ERR_RUNTIME.
int ensure_dynamic_lds(const void *kernel, size_t bytes);

inline int launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? MAGIC_OK : MAGIC_ERR_LAUNCH;
}

inline size_t align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

// Bump allocator over the caller's workspace.
struct Arena {
    char *base;
    size_t cap, off;
    Arena(void *p, size_t n) : base((char *)p), cap(n), off(0) {}
    template <typename T> T *take(size_t count) {
        size_t bytes = align_up(count * sizeof(T));
        if (base == nullptr || off + bytes > cap) {
            off = cap + 1;
            return nullptr;
        }
        T *r = (T *)(base + off);
        off += bytes;
        return r;
    }
    bool ok() const { return off <= cap; }
    bool fits(size_t bytes) const { return base != nullptr && off <= cap && off + align_up(bytes) <= cap; }
};

// grid for a grid-stride kernel: enough blocks to fill 256 CUs x 8, never more than needed
inline int stride_grid(int64_t work_items, int items_per_block, int max_blocks = 2048) {
    int64_t need = (work_items + items_per_block - 1) / items_per_block;
    if (need < 1) need = 1;
    return (int)(need < max_blocks ? need : max_blocks);
}

}  // namespace magic

// launch a kernel under an optional event bracket named after it
#define MAGIC_LAUNCH(name, kernel, grid, block, shmem, stream, ...)                      \
    do {                                                                                 \
        magic::ProfScope _ps(name, stream);                                              \
        hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), shmem, stream, __VA_ARGS__); \
    } while (0)
