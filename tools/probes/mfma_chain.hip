// Is v_mfma_f32_16x16x4_f32 bit for bit a chain of fmaf over its four k values (in which order)?  The evaluator's few-row path
// (VALU) must reproduce the MFMA path's prediction rows exactly.  hipcc --offload-arch=gfx950 -O2 -ffp-contract=off
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k_mfma(const float *x, const float *w, float *y, int K) { // x[16][K], w[16][K] -> y[16][16] = x w^T, k order as mlp_tile_task
    const int lane = threadIdx.x, r = lane & 15, g = lane >> 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < K / 16; ++s) {
        const float4 a = *reinterpret_cast<const float4 *>(x + r * K + 16 * s + 4 * g);
        const float4 b = *reinterpret_cast<const float4 *>(w + r * K + 16 * s + 4 * g);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, acc, 0, 0, 0);
    }
    for (int i = 0; i < 4; ++i) y[(4 * g + i) * 16 + r] = acc[i];
}
__global__ void k_chain(const float *x, const float *w, float *y, int K, int mode) {
    const int row = threadIdx.x >> 4, col = threadIdx.x & 15;
    float acc = 0.f;
    for (int s = 0; s < K / 16; ++s)
        for (int j = 0; j < 4; ++j) {
            if (mode == 0) { // sequential chain over g
                for (int g = 0; g < 4; ++g) acc = fmaf(x[row * K + 16 * s + 4 * g + j], w[col * K + 16 * s + 4 * g + j], acc);
            } else if (mode == 1) { // reversed
                for (int g = 3; g >= 0; --g) acc = fmaf(x[row * K + 16 * s + 4 * g + j], w[col * K + 16 * s + 4 * g + j], acc);
            } else { // products summed first (exact products, tree), then added
                float p[4];
                for (int g = 0; g < 4; ++g) p[g] = x[row * K + 16 * s + 4 * g + j] * w[col * K + 16 * s + 4 * g + j];
                acc = acc + ((p[0] + p[1]) + (p[2] + p[3]));
            }
        }
    y[row * 16 + col] = acc;
}
int main() {
    const int K = 304;
    std::vector<float> x(16 * K), w(16 * K), y0(256), y1(256);
    float *dx, *dw, *dy;
    hipMalloc(&dx, x.size() * 4); hipMalloc(&dw, w.size() * 4); hipMalloc(&dy, 1024);
    int bad[3] = {0, 0, 0};
    for (int trial = 0; trial < 200; ++trial) {
        srand(trial + 1);
        for (auto &v : x) v = (trial & 1) ? (float)(rand() & 1) : (float)rand() / RAND_MAX * 2.f - 1.f;
        for (auto &v : w) v = ((float)rand() / RAND_MAX * 2.f - 1.f) * ((trial % 3 == 0) ? 1e-3f : 1.f);
        hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(dw, w.data(), w.size() * 4, hipMemcpyHostToDevice);
        k_mfma<<<1, 64>>>(dx, dw, dy, K);
        hipMemcpy(y0.data(), dy, 1024, hipMemcpyDeviceToHost);
        for (int mode = 0; mode < 3; ++mode) {
            k_chain<<<1, 256>>>(dx, dw, dy, K, mode);
            hipMemcpy(y1.data(), dy, 1024, hipMemcpyDeviceToHost);
            if (memcmp(y0.data(), y1.data(), 1024) != 0) bad[mode]++;
        }
    }
    printf("trials 200: mismatching trials  chain g=0..3: %d   chain g=3..0: %d   tree of products: %d\n", bad[0], bad[1], bad[2]);
    return 0;
}
