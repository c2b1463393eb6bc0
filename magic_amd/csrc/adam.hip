// Adam as the reference configures it (utils/utils.py:24-31 of the reference: torch.optim.Adam with
// weight_decay, i.e. the L2 term added to the gradient) for every parameter tensor of the model at once.
//
// The parameter / gradient / moment pointers travel BY VALUE in the kernel arguments (up to 64 tensors per
// launch: nothing to upload per step, and a launch recorded into a HIP graph carries them with it); the
// step count lives in device memory (read by every block, advanced by a one-thread launch afterwards), so a
// recorded step counts on its own.  A block takes 2048 consecutive elements of one tensor: the ~40 tensors
// of the model (1.2 M elements, most of them in three matrices) make ~640 blocks instead of the ~60 chunks
// a chunk-per-64k-elements scheme gives, which is what bounds a multi-tensor launch of this size.
#include "common.h"

namespace {

constexpr int ADAM_SLOTS = 64;
constexpr int ADAM_CHUNK = 2048;

struct AdamBatch {
    magic_adam_slot slot[ADAM_SLOTS];
    int32_t first_block[ADAM_SLOTS + 1];  // block range of every slot
    int n_slots;
};
static_assert(sizeof(AdamBatch) <= 3584, "kernel arguments are limited to 4 KB");

__global__ __launch_bounds__(256) void k_adam(AdamBatch b, const int64_t *__restrict__ step, double lr_d, double beta1_d,
                                              double beta2_d, float eps, float weight_decay) {
    // slot of this block: the ranges are ascending, a linear walk over at most 64 entries
    int s = 0;
    while (s + 1 < b.n_slots && (int)blockIdx.x >= b.first_block[s + 1]) ++s;
    const magic_adam_slot sl = b.slot[s];
    const int64_t base = (int64_t)((int)blockIdx.x - b.first_block[s]) * ADAM_CHUNK;
    // bias corrections and step size in double (two pow per block), rounded once
    const double t = (double)(step[0] + 1);
    const double bc1 = 1.0 - pow(beta1_d, t), bc2 = 1.0 - pow(beta2_d, t);
    const float step_size = (float)(lr_d / bc1), bc2_sqrt = (float)sqrt(bc2);
    const float beta1 = (float)beta1_d, beta2 = (float)beta2_d;
#pragma unroll
    for (int j = 0; j < ADAM_CHUNK / 256; ++j) {
        const int64_t i = base + j * 256 + threadIdx.x;
        if (i < sl.n) {
            float p = sl.param[i];
            const float g = fmaf(weight_decay, p, sl.grad[i]);
            float m = sl.exp_avg[i], v = sl.exp_avg_sq[i];
            m = fmaf(1.f - beta1, g - m, m);
            v = fmaf(1.f - beta2, g * g, beta2 * v);
            const float denom = sqrtf(v) / bc2_sqrt + eps;
            p -= step_size * (m / denom);
            sl.param[i] = p;
            sl.exp_avg[i] = m;
            sl.exp_avg_sq[i] = v;
        }
    }
}

__global__ void k_adam_tick(int64_t *step) {
    if (threadIdx.x == 0 && blockIdx.x == 0) step[0] += 1;
}

}  // namespace

extern "C" int magic_adam_step(int32_t n_tensors, const magic_adam_slot *slots, int64_t *step_device, double lr,
                               double beta1, double beta2, double eps, double weight_decay, void *stream_) {
    if (n_tensors < 0 || !step_device || (n_tensors > 0 && !slots)) return MAGIC_ERR_INVALID_ARG;
    hipStream_t stream = (hipStream_t)stream_;
    for (int32_t s0 = 0; s0 < n_tensors; s0 += ADAM_SLOTS) {
        AdamBatch b;
        b.n_slots = 0;
        int64_t blocks = 0;
        for (int32_t s = s0; s < n_tensors && s < s0 + ADAM_SLOTS; ++s) {
            const magic_adam_slot &sl = slots[s];
            if (sl.n < 0 || (sl.n > 0 && (!sl.param || !sl.grad || !sl.exp_avg || !sl.exp_avg_sq)))
                return MAGIC_ERR_INVALID_ARG;
            if (sl.n == 0) continue;
            b.slot[b.n_slots] = sl;
            b.first_block[b.n_slots] = (int32_t)blocks;
            blocks += (sl.n + ADAM_CHUNK - 1) / ADAM_CHUNK;
            if (blocks >= INT32_MAX) return MAGIC_ERR_UNSUPPORTED;
            ++b.n_slots;
        }
        for (int s = b.n_slots; s <= ADAM_SLOTS; ++s) b.first_block[s] = (int32_t)blocks;
        for (int s = b.n_slots; s < ADAM_SLOTS; ++s) b.slot[s] = magic_adam_slot{nullptr, nullptr, nullptr, nullptr, 0};
        if (b.n_slots == 0) continue;
        MAGIC_LAUNCH("adam_step", k_adam, (unsigned)blocks, 256, 0, stream, b, step_device, lr, beta1, beta2, (float)eps,
                     (float)weight_decay);
    }
    MAGIC_LAUNCH("adam_tick", k_adam_tick, 1, 64, 0, stream, step_device);
    return magic::launch_status();
}
