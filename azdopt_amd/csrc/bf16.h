// bf16.h -- round-to-nearest-even f32 -> bf16, shared by the evaluator kernels and their tests' CPU
// restatement (no NaN handling: activations and weights are finite)
#pragma once
#include <stdint.h>

namespace azd {
__host__ __device__ inline uint32_t bf16_bits(float f) {
    union { float f; uint32_t u; } c;
    c.f = f;
    uint32_t u = c.u;
    u += 0x7FFFu + ((u >> 16) & 1u);
    return u >> 16;
}
__host__ __device__ inline float bf16_round(float f) {
    union { float f; uint32_t u; } c;
    c.u = bf16_bits(f) << 16;
    return c.f;
}
} // namespace azd
