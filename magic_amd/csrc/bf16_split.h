// fp32 values as sums of three bf16 pieces, the operands of the split-bf16 matrix products:
// with a = a1 + a2 + a3 and b = b1 + b2 + b3 (8 significant bits each, round-to-nearest at every
// step, so the sums are exact for normal values), the six leading cross products a1b1, a1b2, a2b1,
// a2b2, a1b3, a3b1 accumulated in fp32 by v_mfma_f32_32x32x16_bf16 reproduce the fp32 product to
// within 2^-23 |a||b| (the dropped a2b3, a3b2, a3b3), one fp32 rounding.  Six bf16 MFMAs cover
// 16 k in 6 x 32 cycles where v_mfma_f32_32x32x2_f32 needs 8 x 64.
#pragma once
#include <hip/hip_runtime.h>

namespace magic {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

// v = h1 + h2 + h3 exactly (round-to-nearest pieces, v_cvt_pk_bf16_f32), two values at a time
__device__ __forceinline__ void split2(f32x2_t v, bf16x2_t &h1, bf16x2_t &h2, bf16x2_t &h3) {
    h1 = __builtin_convertvector(v, bf16x2_t);
    const f32x2_t r1 = v - __builtin_convertvector(h1, f32x2_t);
    h2 = __builtin_convertvector(r1, bf16x2_t);
    const f32x2_t r2 = r1 - __builtin_convertvector(h2, f32x2_t);
    h3 = __builtin_convertvector(r2, bf16x2_t);
}

// eight consecutive k of a lane's row -> the three bf16 operand vectors of one MFMA step
__device__ __forceinline__ void split8(float4 lo, float4 hi, bf16x8_t &p1, bf16x8_t &p2, bf16x8_t &p3) {
    const f32x2_t v[4] = {{lo.x, lo.y}, {lo.z, lo.w}, {hi.x, hi.y}, {hi.z, hi.w}};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        bf16x2_t h1, h2, h3;
        split2(v[j], h1, h2, h3);
        p1[2 * j] = h1[0], p1[2 * j + 1] = h1[1];
        p2[2 * j] = h2[0], p2[2 * j + 1] = h2[1];
        p3[2 * j] = h3[0], p3[2 * j + 1] = h3[1];
    }
}

// the six kept products of one 32x32x16 step, smallest terms first
__device__ __forceinline__ void mfma_split6(float __attribute__((ext_vector_type(16))) &acc, bf16x8_t a1, bf16x8_t a2,
                                            bf16x8_t a3, bf16x8_t b1, bf16x8_t b2, bf16x8_t b3) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, b1, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b3, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b2, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b1, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b2, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc, 0, 0, 0);
}

}  // namespace magic
