// mlp_kernels.hip -- the evaluator: ActionModel (az-discrete-opt/src/nabla/model/dfdx.rs) as an
// fp32 MLP on the gfx950 matrix cores.
//
//   forward   y = act(x W^T + b) per layer     -> one MFMA GEMM launch per layer, bias+activation
//                                                  fused into the accumulator epilogue
//   update    w /= sum(w); L = sum w (p - o)^2  -> loss/delta kernel (deterministic two-stage sums),
//             backward                             dX / dW GEMMs on the same MFMA kernel (ReLU mask fused),
//             Adam with L2 (04-c21-tree.rs:87-92)  one elementwise kernel over the flat parameter vector
//
// v_mfma_f32_32x32x2_f32 computes an exact f32 FMA chain in k order (no reduced precision), so the
// forward differs from a CPU fp32 reference only by summation order (tests: 2e-5 abs on sigmoid outputs).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>
#include <new>
#include <vector>

#include "c21_host.h"
#include "bf16.h"
#include "evaluator.h"

namespace azd {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 64, BN = 64, BK = 16;
constexpr int LDS_LD = BM + 4; // k-major tiles: s[k][i]; +4 keeps the transposing stores off a 4-way conflict

enum { EPI_NONE = 0, EPI_BIAS_ACT = 1, EPI_RELU_MASK = 2, EPI_BIAS_ACT_BF16 = 3 };

// C[M,N] = epi( sum_k A(i,k) B(k,j) ).
//   A_KC: A(i,k) = A[i*lda + k] (k contiguous) else A[k*lda + i]
//   B_KC: B(k,j) = B[j*ldb + k] (k contiguous) else B[k*ldb + j]
// Tiles are staged through LDS k-major so every MFMA operand read is a conflict-free ds_read_b32;
// 256 threads = 4 waves in a 2x2 grid of 32x32 accumulators (v_mfma_f32_32x32x2_f32).
template <bool A_KC, bool B_KC>
__global__ __launch_bounds__(256) void k_gemm(const float *__restrict__ A, int lda, const float *__restrict__ B, int ldb,
                                              float *__restrict__ C, int ldc, int M, int N, int K_all, int epi, int act,
                                              const float *__restrict__ bias, const float *__restrict__ aux, int ldaux,
                                              int k_split, size_t c_split_stride) {
    // split K (gridDim.z > 1): block z accumulates k in [z * k_split, (z + 1) * k_split) into its own copy of C
    // (k_split is a multiple of BK); the copies are added up in a fixed order afterwards (k_sum_splits)
    const int k_begin = (int)blockIdx.z * k_split;
    const int K = (k_begin + k_split < K_all) ? k_begin + k_split : K_all;
    C += (size_t)blockIdx.z * c_split_stride;
    __shared__ float sA[BK][LDS_LD];
    __shared__ float sB[BK][LDS_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    for (int k0 = k_begin; k0 < K; k0 += BK) {
        // ---- stage A tile: BM x BK
        if (A_KC) {
            int i = tid >> 2, kq = (tid & 3) * 4; // 64 rows x 4 quads
            int gi = m0 + i, gk = k0 + kq;
            float v[4] = {0.f, 0.f, 0.f, 0.f};
            if (gi < M) {
                const float *p = A + (size_t)gi * lda + gk;
                if (gk + 3 < K && ((reinterpret_cast<uintptr_t>(p) & 15) == 0)) {
                    float4 q = *reinterpret_cast<const float4 *>(p);
                    v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (gk + j < K) v[j] = p[j];
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) sA[kq + j][i] = v[j];
        } else {
            int kk = tid >> 4, iq = (tid & 15) * 4; // 16 k-rows x 16 quads of i
            int gk = k0 + kk, gi = m0 + iq;
            float v[4] = {0.f, 0.f, 0.f, 0.f};
            if (gk < K) {
                const float *p = A + (size_t)gk * lda + gi;
                if (gi + 3 < M && ((reinterpret_cast<uintptr_t>(p) & 15) == 0)) {
                    float4 q = *reinterpret_cast<const float4 *>(p);
                    v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (gi + j < M) v[j] = p[j];
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) sA[kk][iq + j] = v[j];
        }
        // ---- stage B tile: BK x BN
        if (B_KC) {
            int jn = tid >> 2, kq = (tid & 3) * 4;
            int gj = n0 + jn, gk = k0 + kq;
            float v[4] = {0.f, 0.f, 0.f, 0.f};
            if (gj < N) {
                const float *p = B + (size_t)gj * ldb + gk;
                if (gk + 3 < K && ((reinterpret_cast<uintptr_t>(p) & 15) == 0)) {
                    float4 q = *reinterpret_cast<const float4 *>(p);
                    v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (gk + j < K) v[j] = p[j];
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) sB[kq + j][jn] = v[j];
        } else {
            int kk = tid >> 4, jq = (tid & 15) * 4;
            int gk = k0 + kk, gj = n0 + jq;
            float v[4] = {0.f, 0.f, 0.f, 0.f};
            if (gk < K) {
                const float *p = B + (size_t)gk * ldb + gj;
                if (gj + 3 < N && ((reinterpret_cast<uintptr_t>(p) & 15) == 0)) {
                    float4 q = *reinterpret_cast<const float4 *>(p);
                    v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (gj + j < N) v[j] = p[j];
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) sB[kk][jq + j] = v[j];
        }
        __syncthreads();
        // ---- 8 MFMAs: lane l feeds A[i = l&31][k = l>>5], B[k = l>>5][j = l&31]
        const int li = lane & 31, lk = lane >> 5;
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            float a = sA[kk + lk][wm * 32 + li];
            float b = sB[kk + lk][wn * 32 + li];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
        __syncthreads();
    }
    // ---- epilogue: C/D layout col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    const int col = n0 + wn * 32 + (lane & 31);
    if (col < N) {
        float bj = ((epi == EPI_BIAS_ACT || epi == EPI_BIAS_ACT_BF16) && bias) ? bias[col] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            int row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (row < M) {
                float v = acc[r];
                if (epi == EPI_BIAS_ACT || epi == EPI_BIAS_ACT_BF16) {
                    v += bj;
                    if (act == AZD_ACT_RELU) v = v > 0.f ? v : 0.f;
                    else if (act == AZD_ACT_SIGMOID) v = 1.0f / (1.0f + expf(-v));
                    if (epi == EPI_BIAS_ACT_BF16) v = bf16_round(v); // hidden activation stored at bf16 precision
                } else if (epi == EPI_RELU_MASK) {
                    v = aux[(size_t)row * ldaux + col] > 0.f ? v : 0.f;
                }
                C[(size_t)row * ldc + col] = v;
            }
        }
    }
}

// ---- batched bf16 GEMM for inference under AZD_STORAGE_BF16 (write_predictions_dev, par_new / par_reset_trees rows,
// the launch-per-phase step): Y[M,N] = act(bf16(X)[M,K] . W16[N,K]^T + b), products exact, f32 accumulation on
// v_mfma_f32_32x32x16_bf16 (the dfdx forward of model/dfdx.rs:82,116 with bf16 storage).
//   block = 256 threads = 4 waves (2 x 2), tile 128 x 128 x 32; a wave owns 64 x 64 = 2 x 2 MFMA tiles (64 accumulator
//   VGPRs); X rows are read as f32 and rounded to bf16 (RNE, bf16.h) on their way into LDS, W16 rows are bf16 already;
//   both tiles sit in LDS k-contiguous with an 80-byte row pitch, so that an operand fragment (8 consecutive k of one
//   row: lane l holds A[l & 31][8 (l >> 5) + j], B[8 (l >> 5) + j][l & 31]) is one 16-byte ds_read; the next k tile's
//   global loads are in flight while the current one is multiplied.
// K and the row pitches must be multiples of 4 (16-byte f32 / 8-byte bf16 vector loads); other shapes take k_gemm.
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
// MI = 32-row MFMA tiles per wave along M: the block tile is 64 MI x 128.  MI = 2 (128 x 128) reuses each W tile over more
// rows; MI = 1 (64 x 128) gives twice the blocks -- at the evaluator's shapes (M = 8192, N = 512: 256 blocks of 128 x 128,
// four waves per CU) the k loop is bound by the latency of its global loads, and more resident waves hide it better than
// a bigger tile saves traffic (profiles/r02_gemm.txt).
constexpr int GB_N = 128, GB_K = 32, GB_LD = 40; // LDS row pitch in bf16 elements (80 B)
template <int MI>
__global__ __launch_bounds__(256, MI == 1 ? 4 : 2) void k_gemm_bf16(const float *__restrict__ X, int ldx, const uint16_t *__restrict__ W, int ldw,
                                                      float *__restrict__ Y, int ldy, int M, int N, int K, int act, int round_out,
                                                      const float *__restrict__ bias) {
    constexpr int GB_M = 64 * MI;
    __shared__ __attribute__((aligned(16))) uint16_t sA[GB_M * GB_LD];
    __shared__ __attribute__((aligned(16))) uint16_t sB[GB_N * GB_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * GB_M, n0 = blockIdx.x * GB_N;
    // staging: A: thread t moves 8 MI consecutive k of row t / (4 / MI) (2 MI float4); B: 16 consecutive k of row t >> 1
    constexpr int AQ = 2 * MI;                       // float4 groups per thread
    const int arow = tid / (4 / MI), akof = (tid % (4 / MI)) * (8 * MI);
    const int srow = tid >> 1, skof = (tid & 1) * 16;
    const float *xa = X + (size_t)(m0 + arow) * ldx + akof;
    const uint16_t *wb = W + (size_t)(n0 + srow) * ldw + skof;
    const bool a_ok = m0 + arow < M, b_ok = n0 + srow < N;
    // one register stage: tile k0 + 32 is on its way while tile k0 is multiplied.  (Measured and discarded: a second stage,
    // tiles k0 + 32 and k0 + 64 in flight -- 217 -> 165 TFLOP/s on config E's shapes, 405 -> 338 at 8192 x 4096 x 4096: the
    // rotation copies and the lost occupancy cost more than the extra tile in flight hides.  Nor two LDS stages with one barrier
    // per k tile, the next tile converted and written behind this tile's MFMAs: 202 -> 183 and 407 -> 353.)
    float4 ra[AQ];
    uint2 rb[4];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int q = 0; q < AQ; ++q)
            ra[q] = (a_ok && k0 + akof + 4 * q < K) ? *reinterpret_cast<const float4 *>(xa + k0 + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int q = 0; q < 4; ++q)
            rb[q] = (b_ok && k0 + skof + 4 * q < K) ? *reinterpret_cast<const uint2 *>(wb + k0 + 4 * q) : make_uint2(0u, 0u);
    };
    auto stash = [&]() {
        uint32_t pa[2 * AQ];
#pragma unroll
        for (int q = 0; q < AQ; ++q) {
            pa[2 * q] = bf16_bits(ra[q].x) | (bf16_bits(ra[q].y) << 16);
            pa[2 * q + 1] = bf16_bits(ra[q].z) | (bf16_bits(ra[q].w) << 16);
        }
        uint4 *da = reinterpret_cast<uint4 *>(sA + arow * GB_LD + akof);
#pragma unroll
        for (int q = 0; q < AQ / 2; ++q) da[q] = make_uint4(pa[4 * q], pa[4 * q + 1], pa[4 * q + 2], pa[4 * q + 3]);
        uint4 *db = reinterpret_cast<uint4 *>(sB + srow * GB_LD + skof);
        db[0] = make_uint4(rb[0].x, rb[0].y, rb[1].x, rb[1].y);
        db[1] = make_uint4(rb[2].x, rb[2].y, rb[3].x, rb[3].y);
    };
    f32x16 acc[MI][2];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    fetch(0);
    for (int k0 = 0; k0 < K; k0 += GB_K) {
        stash();
        __syncthreads();
        if (k0 + GB_K < K) fetch(k0 + GB_K);
        const int fr = lane & 31, fh = (lane >> 5) * 8;
#pragma unroll
        for (int ks = 0; ks < GB_K; ks += 16) {
            bf16x8_t fa[MI], fb[2];
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const uint4 va = *reinterpret_cast<const uint4 *>(sA + (wm * 32 * MI + i * 32 + fr) * GB_LD + ks + fh);
                __builtin_memcpy(&fa[i], &va, 16);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const uint4 vb = *reinterpret_cast<const uint4 *>(sB + (wn * 64 + j * 32 + fr) * GB_LD + ks + fh);
                __builtin_memcpy(&fb[j], &vb, 16);
            }
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }
    // epilogue: C/D layout col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = n0 + wn * 64 + j * 32 + (lane & 31);
        if (col >= N) continue;
        const float bj = bias ? bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 32 * MI + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (row < M) {
                    float v = acc[i][j][r] + bj;
                    if (act == AZD_ACT_RELU) v = v > 0.f ? v : 0.f;
                    else if (act == AZD_ACT_SIGMOID) v = 1.0f / (1.0f + expf(-v));
                    if (round_out) v = bf16_round(v); // a hidden activation is the next layer's bf16 input
                    Y[(size_t)row * ldy + col] = v;
                }
            }
    }
}

#include "gemm_bf16_glds.inc"
#include "gemm_bf16_hidden2.inc"

// ---- bf16 weight storage: w16 = RNE(params) in 16-bit words, params_q = the same values widened back
// to f32 (what the f32 GEMM path multiplies with, so that it computes what the bf16 MFMA path computes)
__global__ void k_quantize_bf16(const float *__restrict__ p, size_t n, uint16_t *__restrict__ w16, float *__restrict__ q) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t b = bf16_bits(p[i]);
    w16[i] = (uint16_t)b;
    q[i] = __uint_as_float(b << 16);
}
// fragment-major copy of one layer's weights for the asynchronous step (engine_types.h:FusedEval::wpk)
__global__ void k_pack_weights(const float *__restrict__ W, int K, int N, float *__restrict__ dst32, uint16_t *__restrict__ dst16) {
    const int steps = (K + 15) >> 4, tiles = (N + 15) >> 4;
    const size_t n = (size_t)tiles * steps * 256;
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const int i = (int)(e & 3), lane = (int)((e >> 2) & 63);
    const size_t js = e >> 8;
    const int s = (int)(js % (size_t)steps), j = (int)(js / (size_t)steps);
    const int row = 16 * j + (lane & 15), k = 16 * s + 4 * (lane >> 4) + i;
    const float v = (row < N && k < K) ? W[(size_t)row * K + k] : 0.f;
    if (dst16) dst16[e] = (uint16_t)bf16_bits(v);
    else dst32[e] = v;
}
__global__ void k_round_bf16(const float *__restrict__ in, size_t n, float *__restrict__ out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = bf16_round(in[i]);
}

// ---- loss: weight_sum (dfdx.rs:106), then delta = dL/dz of the head and per-block loss partials
__global__ void k_block_sum(const float *__restrict__ x, size_t n, float *__restrict__ partial) {
    __shared__ float s[256];
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += x[i];
    s[threadIdx.x] = acc;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) s[threadIdx.x] += s[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = s[0];
}
__global__ void k_final_sum(const float *__restrict__ partial, int n, float *__restrict__ out) {
    __shared__ float s[256];
    float acc = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) acc += partial[i];
    s[threadIdx.x] = acc;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) s[threadIdx.x] += s[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) *out = s[0];
}
// dfdx.rs:110,118-123: w~ = w / weight_sum; L = sum w~ (p - o)^2; d = dL/dz through the head activation
__global__ void k_loss_delta(const float *__restrict__ pred, const float *__restrict__ obs, const float *__restrict__ w,
                             const float *__restrict__ wsum, size_t n, int act, float *__restrict__ delta,
                             float *__restrict__ partial) {
    __shared__ float s[256];
    const float ws = *wsum;
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float wt = w[i] / ws;
        float p = pred[i];
        float d = p - obs[i];
        acc += wt * d * d;
        float dp = 2.0f * wt * d;
        if (act == AZD_ACT_SIGMOID) dp *= p * (1.0f - p);
        else if (act == AZD_ACT_RELU) dp = p > 0.f ? dp : 0.f;
        delta[i] = dp;
    }
    s[threadIdx.x] = acc;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) s[threadIdx.x] += s[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = s[0];
}
// db[j] = sum_i dZ[i][j]: 64 columns x 16 row-strides per block, fixed-order LDS tree over the strides
// (gridDim.y > 1: block y sums rows [y * rows_per, (y + 1) * rows_per) into part y of `out`; k_sum_splits adds
// the parts in order)
__global__ __launch_bounds__(1024) void k_col_sum(const float *__restrict__ dz, int M, int N, float *__restrict__ out, int rows_per) {
    __shared__ float s[16][65];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int j = blockIdx.x * 64 + tx;
    const int r0 = (int)blockIdx.y * rows_per;
    const int r1 = (r0 + rows_per < M) ? r0 + rows_per : M;
    out += (size_t)blockIdx.y * N;
    float acc = 0.f;
    if (j < N)
        for (int i = r0 + ty; i < r1; i += 16) acc += dz[(size_t)i * N + j];
    s[ty][tx] = acc;
    __syncthreads();
    for (int off = 8; off > 0; off >>= 1) {
        if (ty < off) s[ty][tx] += s[ty + off][tx];
        __syncthreads();
    }
    if (ty == 0 && j < N) out[j] = s[0][tx];
}
// dfdx Adam with WeightDecay::L2: g += wd p; m, v; bias-corrected; p -= lr m^ / (sqrt(v^) + eps)
__global__ void k_adam(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m, float *__restrict__ v,
                       size_t n, float lr, float b1, float b2, float eps, float l2, float bc1, float bc2) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float gi = g[i] + l2 * p[i];
    float mi = b1 * m[i] + (1.0f - b1) * gi;
    float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    float mh = mi / bc1, vh = vi / bc2;
    p[i] -= lr * mh / (sqrtf(vh) + eps);
}

template <bool A_KC, bool B_KC>
static void gemm(hipStream_t st, const float *A, int lda, const float *B, int ldb, float *C, int ldc, int M, int N, int K,
                 int epi, int act, const float *bias, const float *aux, int ldaux) {
    dim3 grid((N + BN - 1) / BN, (M + BM - 1) / BM);
    k_gemm<A_KC, B_KC><<<grid, dim3(256), 0, st>>>(A, lda, B, ldb, C, ldc, M, N, K, epi, act, bias, aux, ldaux, K, 0);
}

// C = sum over splits, in split order: the same result whatever the schedule (every rank of a multi-GPU run
// must take the identical optimiser step)
__global__ void k_sum_splits(const float *__restrict__ parts, size_t n, int splits, float *__restrict__ C) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float acc = parts[i];
    for (int z = 1; z < splits; ++z) acc += parts[(size_t)z * n + i];
    C[i] = acc;
}
// dW[out][in] = dZ^T . X with the batch as K: a 64x64-tiled grid of (out/64) x (in/64) workgroups is 20 for a
// 256x304 layer, each walking the whole batch; K is split over gridDim.z instead and the parts are summed
// in order (C contiguous, ldc == N)
constexpr int GEMM_SPLIT_K = 512;
static void gemm_dw(hipStream_t st, const float *dz, int out, const float *x, int in, float *dW, int batch, float *parts, int max_splits) {
    int splits = (batch + GEMM_SPLIT_K - 1) / GEMM_SPLIT_K;
    if (splits > max_splits) splits = max_splits;
    if (splits <= 1 || !parts) {
        gemm<false, false>(st, dz, out, x, in, dW, in, out, in, batch, EPI_NONE, 0, nullptr, nullptr, 0);
        return;
    }
    int k_split = (batch + splits - 1) / splits;
    k_split = (k_split + BK - 1) / BK * BK;
    splits = (batch + k_split - 1) / k_split;
    const size_t n = (size_t)out * in;
    dim3 grid((in + BN - 1) / BN, (out + BM - 1) / BM, splits);
    k_gemm<false, false><<<grid, dim3(256), 0, st>>>(dz, out, x, in, parts, in, out, in, batch, EPI_NONE, 0, nullptr, nullptr, 0, k_split, n);
    k_sum_splits<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(parts, n, splits, dW);
}

struct MlpEvaluator : azd_evaluator {
    int L = 0;
    std::vector<int> dims; // L+1
    int final_act = AZD_ACT_SIGMOID;
    azd_adam_config adam{};
    int max_batch = 0;
    int t = 0;
    int64_t n_params = 0;
    std::vector<int64_t> w_off, b_off;
    float *d_params = nullptr, *d_grads = nullptr, *d_m = nullptr, *d_v = nullptr;
    std::vector<float *> d_act; // d_act[l], l = 1..L-1 hidden outputs (max_batch x dims[l]); [0] and [L] are caller buffers
    float *d_pred_train = nullptr;
    float *d_delta_a = nullptr, *d_delta_b = nullptr; // ping-pong deltas (max_batch x max_dim)
    float *d_partial = nullptr, *d_scalars = nullptr; // [0] wsum [1] loss
    float *h_scalars = nullptr;
    int cap_batch = 0;
    bool bf16 = false;             // AZD_STORAGE_BF16
    uint16_t *d_w16 = nullptr;     // bf16 copy of d_params (same offsets)
    // the bf16 forward GEMM's operands (gemm_bf16_glds.inc): rows of pitch kp[l] = dims[l] rounded up to 64, zero beyond dims[l]
    uint16_t *d_w16p = nullptr;    // weights, layer l at wp_off[l]: [dims[l + 1]][kp[l]]
    std::vector<int64_t> wp_off;
    std::vector<int> kp;
    std::vector<uint16_t *> d_act16; // d_act16[l], l = 0 .. L-1: the input rows of layer l, [cap_batch][kp[l]] ([0]: the converted state vectors)
    int n_cus = 256;
    float *d_params_q = nullptr;   // the bf16 values widened to f32 (GEMM path)
    float *d_xq = nullptr;         // input rows rounded to bf16 precision (GEMM path)
    float *d_split = nullptr;      // split-K parts of the largest weight gradient (gemm_dw)
    static constexpr int MAX_SPLITS = 64;
    float *d_wpk = nullptr;        // fragment-major weights for the asynchronous step (f32 words / bf16 halves)
    float *d_part = nullptr;       // split-k partial sums of the bf16 forward (gemm_bf16_glds.inc)
    size_t part_stride = 0;        // elements per slice
    std::vector<int64_t> p_off;
    int64_t n_packed = 0;
    bool fuse_hidden2 = true; // two consecutive 512-wide hidden layers in one launch (gemm_bf16_hidden2.inc; AZD_MLP_FUSE_HIDDEN=0: layer by layer)
    int gemm_small_below = 1024; // bf16 GEMM: grids of fewer 128 x 128 tiles than this use 64 x 128 tiles (AZD_GEMM_SMALL_BELOW overrides: experiments)

    ~MlpEvaluator() override {
        (void)hipSetDevice(device);
        if (d_w16) (void)hipFree(d_w16);
        if (d_w16p) (void)hipFree(d_w16p);
        for (uint16_t *p : d_act16)
            if (p) (void)hipFree(p);
        for (float *p : {d_params, d_grads, d_m, d_v, d_pred_train, d_delta_a, d_delta_b, d_partial, d_scalars, d_params_q, d_xq, d_wpk, d_split, d_part})
            if (p) (void)hipFree(p);
        for (float *p : d_act)
            if (p) (void)hipFree(p);
        if (h_scalars) (void)hipHostFree(h_scalars);
    }

    int ensure_batch(int batch) {
        if (batch <= cap_batch) return AZD_OK;
        layout_version += 1;
        for (float *&p : d_act) {
            if (p) (void)hipFree(p);
            p = nullptr;
        }
        for (float **pp : {&d_pred_train, &d_delta_a, &d_delta_b}) {
            if (*pp) (void)hipFree(*pp);
            *pp = nullptr;
        }
        int maxd = 0;
        for (int d : dims) maxd = d > maxd ? d : maxd;
        d_act.assign((size_t)L + 1, nullptr);
        for (int l = 1; l < L; ++l) AZD_HIP(hipMalloc(&d_act[(size_t)l], (size_t)batch * dims[(size_t)l] * 4));
        AZD_HIP(hipMalloc(&d_pred_train, (size_t)batch * dims[(size_t)L] * 4));
        AZD_HIP(hipMalloc(&d_delta_a, (size_t)batch * maxd * 4));
        AZD_HIP(hipMalloc(&d_delta_b, (size_t)batch * maxd * 4));
        if (d_xq) {
            (void)hipFree(d_xq);
            d_xq = nullptr;
        }
        if (bf16 || d_w16) AZD_HIP(hipMalloc(&d_xq, (size_t)batch * dims[0] * 4));
        cap_batch = batch;
        if (d_w16p) {
            int st16 = alloc_act16();
            if (st16) return st16;
        }
        return AZD_OK;
    }

    int alloc_act16() { // zeroed once: the kernels write columns < dims[l] only, the padding stays zero
        if (d_part) (void)hipFree(d_part);
        d_part = nullptr;
        part_stride = 0;
        for (int l = 0; l + 1 < L; ++l) // split-k layers (by shape: gemm16_ksplit): f32 partial sums, 8 slices at most, in batch row order
            if (gemm16_ksplit(dims[(size_t)l + 1], kp[(size_t)l]) > 1 || getenv("AZD_GEMM16_KSPLIT")) part_stride = std::max(part_stride, (size_t)cap_batch * kp[(size_t)l + 1]);
        if (part_stride) AZD_HIP(hipMalloc(&d_part, part_stride * 8 * sizeof(float)));
        for (uint16_t *&p : d_act16) {
            if (p) (void)hipFree(p);
            p = nullptr;
        }
        d_act16.assign((size_t)L, nullptr);
        for (int l = 0; l < L; ++l) {
            AZD_HIP(hipMalloc(&d_act16[(size_t)l], (size_t)cap_batch * kp[(size_t)l] * 2));
            AZD_HIP(hipMemset(d_act16[(size_t)l], 0, (size_t)cap_batch * kp[(size_t)l] * 2));
        }
        return AZD_OK;
    }
    // layers l and l + 1 are both 512 -> 512 hidden layers whose input is an activation buffer in batch order: one fused launch
    bool hidden_pair(int l) const {
        return fuse_hidden2 && l >= 1 && l + 2 < L && dims[(size_t)l] == HF_H && dims[(size_t)l + 1] == HF_H && dims[(size_t)l + 2] == HF_H;
    }
    // The bf16 forward on the LDS-DMA GEMM (gemm_bf16_glds.inc): x16 = the input rows as bf16 with pitch kp[0] (the producer's own
    // copy: azd_evaluator::write_predictions_dev16), or null: converted here from the f32 rows.  Same sums as k_gemm_bf16.
    // act_row0: the rows of the activation buffers this call may use (write_predictions_rows: disjoint ranges run concurrently)
    int forward16(int batch, const float *d_s, const uint16_t *x16, float *d_p, hipStream_t st, int act_row0 = 0) {
        if (!x16) {
            const int per = kp[0] / 4;
            const size_t n = (size_t)batch * per;
            k_rows_to_bf16<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(d_s, dims[0], batch, dims[0], d_act16[0] + (size_t)act_row0 * kp[0], kp[0]);
            x16 = d_act16[0] + (size_t)act_row0 * kp[0];
        }
        for (int l = 0; l < L; ++l) {
            const bool last = l == L - 1;
            if (hidden_pair(l)) {
                uint16_t *y2 = d_act16[(size_t)l + 2] + (size_t)act_row0 * HF_H;
                launch_hidden2<false>(st, x16, HF_H, d_w16p + wp_off[(size_t)l], d_w16p + wp_off[(size_t)l + 1], HF_H, d_params + b_off[(size_t)l],
                                      d_params + b_off[(size_t)l + 1], y2, HF_H, batch, nullptr);
                x16 = y2;
                l += 1;
                continue;
            }
            void *y = last ? (void *)d_p : (void *)(d_act16[(size_t)l + 1] + (size_t)act_row0 * kp[(size_t)l + 1]);
            launch_gemm16(st, x16, kp[(size_t)l], d_w16p + wp_off[(size_t)l], kp[(size_t)l], y, last ? dims[(size_t)L] : kp[(size_t)l + 1], batch,
                          dims[(size_t)l + 1], kp[(size_t)l], last ? final_act : AZD_ACT_RELU, last ? 0 : 1, d_params + b_off[(size_t)l], n_cus, 0,
                          (!last && d_part) ? d_part + (size_t)act_row0 * kp[(size_t)l + 1] : nullptr, part_stride);
            if (!last) x16 = d_act16[(size_t)l + 1] + (size_t)act_row0 * kp[(size_t)l + 1];
        }
        AZD_HIP(hipGetLastError());
        return AZD_OK;
    }
    // quant: inference with bf16-stored weights and activations (f32 accumulate); training always runs on
    // the f32 master weights
    int forward(int batch, const float *d_s, float *d_p, hipStream_t st, bool quant = false) {
        const float *x = d_s;
        const float *P = quant ? d_params_q : d_params;
        if (quant && d_w16p && !getenv("AZD_GEMM_OLD")) return forward16(batch, d_s, nullptr, d_p, st);
        bool mfma16 = quant; // the bf16 MFMA GEMM takes row pitches in multiples of 4
        for (int l = 0; l < L && mfma16; ++l) mfma16 = dims[(size_t)l] % 4 == 0 && w_off[(size_t)l] % 4 == 0;
        if (mfma16) {
            for (int l = 0; l < L; ++l) {
                float *y = (l == L - 1) ? d_p : d_act[(size_t)l + 1];
                const int K = dims[(size_t)l], N = dims[(size_t)l + 1];
                const int nb = (N + GB_N - 1) / GB_N;
                const bool small = (size_t)nb * ((batch + 127) / 128) < (size_t)gemm_small_below; // 128 x 128 tiles would leave CUs short of waves
                if (small)
                    k_gemm_bf16<1><<<dim3(nb, (batch + 63) / 64), dim3(256), 0, st>>>(x, K, d_w16 + w_off[(size_t)l], K, y, N, batch, N, K,
                                                                                     (l == L - 1) ? final_act : AZD_ACT_RELU, l < L - 1 ? 1 : 0,
                                                                                     d_params + b_off[(size_t)l]);
                else
                    k_gemm_bf16<2><<<dim3(nb, (batch + 127) / 128), dim3(256), 0, st>>>(x, K, d_w16 + w_off[(size_t)l], K, y, N, batch, N, K,
                                                                                       (l == L - 1) ? final_act : AZD_ACT_RELU, l < L - 1 ? 1 : 0,
                                                                                       d_params + b_off[(size_t)l]);
                x = y;
            }
            AZD_HIP(hipGetLastError());
            return AZD_OK;
        }
        if (quant) {
            const size_t n = (size_t)batch * dims[0];
            k_round_bf16<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(d_s, n, d_xq);
            x = d_xq;
        }
        for (int l = 0; l < L; ++l) {
            float *y = (l == L - 1) ? d_p : d_act[(size_t)l + 1];
            int act = (l == L - 1) ? final_act : AZD_ACT_RELU;
            const int epi = (quant && l < L - 1) ? EPI_BIAS_ACT_BF16 : EPI_BIAS_ACT;
            gemm<true, true>(st, x, dims[(size_t)l], P + w_off[(size_t)l], dims[(size_t)l], y, dims[(size_t)l + 1], batch,
                             dims[(size_t)l + 1], dims[(size_t)l], epi, act, d_params + b_off[(size_t)l], nullptr, 0);
            x = y;
        }
        AZD_HIP(hipGetLastError());
        return AZD_OK;
    }

    // dfdx.rs:69-84
    int write_predictions_dev(int batch, const float *d_s, float *d_p, hipStream_t st) override {
        AZD_HIP(hipSetDevice(device));
        int s = ensure_batch(batch);
        if (s) return s;
        calls += 1;
        return forward(batch, d_s, d_p, st, bf16);
    }

    int write_predictions_dev16(int batch, const float *d_s, const uint16_t *d_s16, int pitch16, float *d_p, hipStream_t st) override {
        if (!bf16 || !d_w16p || !d_s16 || pitch16 != kp[0] || getenv("AZD_GEMM_OLD")) return write_predictions_dev(batch, d_s, d_p, st);
        AZD_HIP(hipSetDevice(device));
        int s = ensure_batch(batch);
        if (s) return s;
        calls += 1;
        return forward16(batch, d_s, d_s16, d_p, st);
    }
    int input16_pitch() override { return (bf16 && d_w16p) ? kp[0] : 0; }
    int write_predictions_rows(int row0, int count, const float *d_s, const uint16_t *d_s16, int pitch16, float *d_p, hipStream_t st) override {
        if (!rows_concurrent() || row0 + count > cap_batch) return azd_evaluator::write_predictions_rows(row0, count, d_s, d_s16, pitch16, d_p, st);
        AZD_HIP(hipSetDevice(device));
        const uint16_t *x16 = (d_s16 && pitch16 == kp[0]) ? d_s16 + (size_t)row0 * pitch16 : nullptr;
        return forward16(count, d_s + (size_t)row0 * state_dim, x16, d_p + (size_t)row0 * action_dim, st, row0);
    }
    bool rows_concurrent() override { return bf16 && d_w16p && !getenv("AZD_GEMM_OLD"); }
    int ensure_rows(int rows) override {
        AZD_HIP(hipSetDevice(device));
        return ensure_batch(rows);
    }
    int write_predictions_gathered(const uint32_t *d_rows, const uint32_t *d_count, int max_rows, const uint16_t *d_s16, int pitch16, float *d_p,
                                   hipStream_t st, int act_row0) override {
        if (!rows_concurrent() || !d_s16 || pitch16 != kp[0]) return AZD_ERR_UNSUPPORTED;
        if (act_row0 + max_rows > cap_batch) return AZD_ERR_CAPACITY; // (the engine sizes the evaluator first: ensure_rows)
        AZD_HIP(hipSetDevice(device));
        const uint16_t *x16 = d_s16;
        for (int l = 0; l < L; ++l) {
            const bool last = l == L - 1;
            if (hidden_pair(l)) {
                uint16_t *y2 = d_act16[(size_t)l + 2] + (size_t)act_row0 * HF_H;
                launch_hidden2<true>(st, x16, HF_H, d_w16p + wp_off[(size_t)l], d_w16p + wp_off[(size_t)l + 1], HF_H, d_params + b_off[(size_t)l],
                                     d_params + b_off[(size_t)l + 1], y2, HF_H, max_rows, d_count);
                x16 = y2;
                l += 1;
                continue;
            }
            void *y = last ? (void *)d_p : (void *)(d_act16[(size_t)l + 1] + (size_t)act_row0 * kp[(size_t)l + 1]);
            launch_gemm16_ext(st, x16, kp[(size_t)l], d_w16p + wp_off[(size_t)l], kp[(size_t)l], y, last ? dims[(size_t)L] : kp[(size_t)l + 1], max_rows,
                              dims[(size_t)l + 1], kp[(size_t)l], last ? final_act : AZD_ACT_RELU, last ? 0 : 1, d_params + b_off[(size_t)l], n_cus, d_count,
                              l == 0 ? d_rows : nullptr, last ? d_rows : nullptr,
                              (!last && d_part) ? d_part + (size_t)act_row0 * kp[(size_t)l + 1] : nullptr, part_stride);
            if (!last) x16 = d_act16[(size_t)l + 1] + (size_t)act_row0 * kp[(size_t)l + 1];
        }
        AZD_HIP(hipGetLastError());
        return AZD_OK;
    }

    // dfdx.rs:86-131
    int update_model_dev(int batch, const float *d_s, const float *d_o, const float *d_w, float *loss, hipStream_t st) override {
        AZD_HIP(hipSetDevice(device));
        int s = ensure_batch(batch);
        if (s) return s;
        s = forward(batch, d_s, d_pred_train, st);
        if (s) return s;
        const int A = dims[(size_t)L];
        const size_t n = (size_t)batch * A;
        const int nb = 256;
        k_block_sum<<<nb, 256, 0, st>>>(d_w, n, d_partial);
        k_final_sum<<<1, 256, 0, st>>>(d_partial, nb, d_scalars);
        k_loss_delta<<<nb, 256, 0, st>>>(d_pred_train, d_o, d_w, d_scalars, n, final_act, d_delta_a, d_partial);
        k_final_sum<<<1, 256, 0, st>>>(d_partial, nb, d_scalars + 1);
        float *dz = d_delta_a, *dx = d_delta_b;
        for (int l = L - 1; l >= 0; --l) {
            const int in = dims[(size_t)l], out = dims[(size_t)l + 1];
            const float *x = (l == 0) ? d_s : d_act[(size_t)l];
            // dW[out][in] = dZ^T[out][batch] . X[batch][in]
            gemm_dw(st, dz, out, x, in, d_grads + w_off[(size_t)l], batch, d_split, MAX_SPLITS);
            {   // db[out] = column sums of dZ, the batch split like the weight gradient's K
                int parts = (batch + GEMM_SPLIT_K - 1) / GEMM_SPLIT_K;
                parts = parts > MAX_SPLITS ? MAX_SPLITS : parts;
                if (parts <= 1) k_col_sum<<<(out + 63) / 64, 1024, 0, st>>>(dz, batch, out, d_grads + b_off[(size_t)l], batch);
                else {
                    const int rows_per = (batch + parts - 1) / parts;
                    parts = (batch + rows_per - 1) / rows_per;
                    k_col_sum<<<dim3((out + 63) / 64, parts), 1024, 0, st>>>(dz, batch, out, d_split, rows_per);
                    k_sum_splits<<<(out + 255) / 256, 256, 0, st>>>(d_split, (size_t)out, parts, d_grads + b_off[(size_t)l]);
                }
            }
            if (l > 0) {
                // dX[batch][in] = (dZ[batch][out] . W[out][in]) masked by ReLU'(x)
                gemm<true, false>(st, dz, out, d_params + w_off[(size_t)l], in, dx, in, batch, in, out, EPI_RELU_MASK, 0, nullptr, x, in);
                float *tmp = dz;
                dz = dx;
                dx = tmp;
            }
        }
        t += 1;
        float bc1 = 1.0f - std::pow(adam.beta1, (float)t), bc2 = 1.0f - std::pow(adam.beta2, (float)t);
        k_adam<<<(unsigned)((n_params + 255) / 256), 256, 0, st>>>(d_params, d_grads, d_m, d_v, (size_t)n_params, adam.lr, adam.beta1,
                                                                  adam.beta2, adam.eps, adam.l2, bc1, bc2);
        if (bf16) requantize(st);
        repack(st);
        AZD_HIP(hipGetLastError());
        AZD_HIP(hipMemcpyAsync(h_scalars, d_scalars, 2 * sizeof(float), hipMemcpyDeviceToHost, st));
        AZD_HIP(hipStreamSynchronize(st));
        if (loss) *loss = h_scalars[1];
        return AZD_OK;
    }

    bool replayable(int batch) override { return batch <= cap_batch; }
    bool fused_desc(FusedEval *f) override {
        memset(f, 0, sizeof(*f));
        if (L > 7) return false;
        f->kind = 3;
        f->params = d_params;
        f->n_layers = L;
        f->final_act = final_act;
        int mh = 4;
        for (int l = 0; l <= L; ++l) f->dims[l] = dims[(size_t)l];
        for (int l = 0; l < L; ++l) {
            // the in-kernel GEMM streams float4 along K: inputs in multiples of 4, rows 16-B aligned
            // (the asynchronous step additionally needs multiples of 16: async_plan)
            if (dims[(size_t)l] % 4 != 0 || w_off[(size_t)l] % 4 != 0) return false;
            f->w_off[l] = w_off[(size_t)l];
            f->b_off[l] = b_off[(size_t)l];
            if (l >= 1 && dims[(size_t)l] > mh) mh = dims[(size_t)l];
        }
        f->max_hidden = mh;
        f->hid[0] = f->hid[1] = 16;
        for (int l = 0; l + 1 < L; ++l)
            if (dims[(size_t)l + 1] > f->hid[l & 1]) f->hid[l & 1] = dims[(size_t)l + 1];
        f->bf16 = bf16 ? 1 : 0;
        f->w16 = d_w16;
        f->wpk = d_wpk;
        for (int l = 0; l < L; ++l) f->p_off[l] = p_off[(size_t)l];
        return true;
    }
    // the asynchronous step's copy follows every change of the parameters or of the storage type
    void repack(hipStream_t st) {
        for (int l = 0; l < L; ++l) {
            const int K = dims[(size_t)l], N = dims[(size_t)l + 1];
            const size_t n = (size_t)((N + 15) >> 4) * ((K + 15) >> 4) * 256;
            float *d32 = bf16 ? nullptr : d_wpk + p_off[(size_t)l];
            uint16_t *d16 = bf16 ? reinterpret_cast<uint16_t *>(d_wpk) + p_off[(size_t)l] : nullptr;
            k_pack_weights<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(d_params + w_off[(size_t)l], K, N, d32, d16);
        }
    }
    void requantize(hipStream_t st) {
        k_quantize_bf16<<<(unsigned)((n_params + 255) / 256), 256, 0, st>>>(d_params, (size_t)n_params, d_w16, d_params_q);
        for (int l = 0; l < L; ++l) { // the padded rows of the LDS-DMA GEMM
            const size_t n = (size_t)dims[(size_t)l + 1] * (kp[(size_t)l] / 4);
            k_rows_to_bf16<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(d_params + w_off[(size_t)l], dims[(size_t)l], dims[(size_t)l + 1], dims[(size_t)l],
                                                                         d_w16p + wp_off[(size_t)l], kp[(size_t)l]);
        }
    }
    // AZD_STORAGE_BF16: inference reads bf16 copies of the weights (refreshed after every optimiser step),
    // the optimiser keeps f32 master weights
    int set_weight_storage(int dtype) override {
        AZD_HIP(hipSetDevice(device));
        if (dtype != AZD_STORAGE_F32 && dtype != AZD_STORAGE_BF16) return AZD_ERR_INVALID_ARGUMENT;
        layout_version += 1;
        AZD_HIP(hipDeviceSynchronize());
        if (dtype == AZD_STORAGE_BF16 && !d_w16) {
            AZD_HIP(hipMalloc(&d_w16, (size_t)n_params * 2));
            AZD_HIP(hipMalloc(&d_params_q, (size_t)n_params * 4));
            {
                kp.clear();
                wp_off.clear();
                int64_t off = 0;
                for (int l = 0; l < L; ++l) {
                    kp.push_back((dims[(size_t)l] + 63) / 64 * 64);
                    wp_off.push_back(off);
                    off += (int64_t)dims[(size_t)l + 1] * kp[(size_t)l];
                }
                AZD_HIP(hipMalloc(&d_w16p, (size_t)off * 2));
                hipDeviceProp_t prop;
                if (hipGetDeviceProperties(&prop, device) == hipSuccess) n_cus = prop.multiProcessorCount;
                int st16 = alloc_act16();
                if (st16) return st16;
            }
            AZD_HIP(hipMalloc(&d_xq, (size_t)cap_batch * dims[0] * 4));
        }
        bf16 = dtype == AZD_STORAGE_BF16;
        if (bf16) requantize(nullptr);
        repack(nullptr);
        AZD_HIP(hipDeviceSynchronize());
        return AZD_OK;
    }
    int64_t num_params() override { return n_params; }
    int get_params(float *out) override {
        AZD_HIP(hipSetDevice(device));
        AZD_HIP(hipDeviceSynchronize());
        AZD_HIP(hipMemcpy(out, d_params, (size_t)n_params * 4, hipMemcpyDeviceToHost));
        return AZD_OK;
    }
    int set_params(const float *in) override {
        AZD_HIP(hipSetDevice(device));
        AZD_HIP(hipDeviceSynchronize());
        AZD_HIP(hipMemcpy(d_params, in, (size_t)n_params * 4, hipMemcpyHostToDevice));
        if (bf16) requantize(nullptr);
        repack(nullptr);
        AZD_HIP(hipDeviceSynchronize());
        return AZD_OK;
    }
};

static int mlp_init(MlpEvaluator *m, uint64_t seed) {
    AZD_HIP(hipSetDevice(m->device));
    AZD_HIP(hipMalloc(&m->d_params, (size_t)m->n_params * 4));
    AZD_HIP(hipMalloc(&m->d_grads, (size_t)m->n_params * 4));
    AZD_HIP(hipMalloc(&m->d_m, (size_t)m->n_params * 4));
    AZD_HIP(hipMalloc(&m->d_v, (size_t)m->n_params * 4));
    AZD_HIP(hipMalloc(&m->d_partial, 256 * 4));
    AZD_HIP(hipMalloc(&m->d_scalars, 4 * 4));
    AZD_HIP(hipHostMalloc((void **)&m->h_scalars, 4 * 4));
    AZD_HIP(hipMemset(m->d_grads, 0, (size_t)m->n_params * 4));
    AZD_HIP(hipMemset(m->d_m, 0, (size_t)m->n_params * 4));
    AZD_HIP(hipMemset(m->d_v, 0, (size_t)m->n_params * 4));
    // dfdx Linear init: weight and bias ~ U(-1/sqrt(in), 1/sqrt(in)) (counter-based draws)
    std::vector<float> h((size_t)m->n_params);
    for (int l = 0; l < m->L; ++l) {
        int in = m->dims[(size_t)l], out = m->dims[(size_t)l + 1];
        float bound = 1.0f / std::sqrt((float)in);
        for (int64_t i = 0; i < (int64_t)in * out; ++i) {
            uint64_t r = stream_key(seed, 0x6d6c7057ull, (uint64_t)l, (uint64_t)i);
            float u = (float)(r >> 40) * (1.0f / 16777216.0f);
            h[(size_t)(m->w_off[(size_t)l] + i)] = (2.0f * u - 1.0f) * bound;
        }
        for (int i = 0; i < out; ++i) {
            uint64_t r = stream_key(seed, 0x6d6c7062ull, (uint64_t)l, (uint64_t)i);
            float u = (float)(r >> 40) * (1.0f / 16777216.0f);
            h[(size_t)(m->b_off[(size_t)l] + i)] = (2.0f * u - 1.0f) * bound;
        }
    }
    AZD_HIP(hipMemcpy(m->d_params, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    m->p_off.assign((size_t)m->L, 0);
    m->n_packed = 0;
    for (int l = 0; l < m->L; ++l) {
        m->p_off[(size_t)l] = m->n_packed;
        m->n_packed += (int64_t)((m->dims[(size_t)l + 1] + 15) / 16) * ((m->dims[(size_t)l] + 15) / 16) * 256;
    }
    // + 8 KB: a ragged last group of the pool step's tile task requests up to 7 k-steps past its tile (tile_task_asm.inc); never used
    AZD_HIP(hipMalloc(&m->d_wpk, (size_t)m->n_packed * 4 + 8192));
    AZD_HIP(hipMemset(m->d_wpk, 0, (size_t)m->n_packed * 4 + 8192));
    {
        size_t big = 0;
        for (int l = 0; l < m->L; ++l) {
            const size_t n = (size_t)m->dims[(size_t)l] * m->dims[(size_t)l + 1];
            big = n > big ? n : big;
        }
        AZD_HIP(hipMalloc(&m->d_split, big * MlpEvaluator::MAX_SPLITS * 4));
    }
    m->repack(nullptr);
    AZD_HIP(hipDeviceSynchronize());
    return m->ensure_batch(m->max_batch);
}

azd_evaluator *make_mlp_evaluator(int device, int max_batch, int state_dim, int action_dim, const int *hidden, int n_hidden,
                                  int final_act, const azd_adam_config *adam, uint64_t seed, int *status) {
    MlpEvaluator *m = new (std::nothrow) MlpEvaluator();
    if (!m) {
        *status = AZD_ERR_OUT_OF_MEMORY;
        return nullptr;
    }
    m->device = device;
    m->state_dim = state_dim;
    m->action_dim = action_dim;
    m->L = n_hidden + 1;
    m->dims.push_back(state_dim);
    for (int i = 0; i < n_hidden; ++i) m->dims.push_back(hidden[i]);
    m->dims.push_back(action_dim);
    m->final_act = final_act;
    if (const char *env = getenv("AZD_GEMM_SMALL_BELOW")) m->gemm_small_below = atoi(env);
    if (const char *env = getenv("AZD_MLP_FUSE_HIDDEN")) m->fuse_hidden2 = atoi(env) != 0;
    m->adam = *adam;
    m->max_batch = max_batch;
    int64_t off = 0;
    for (int l = 0; l < m->L; ++l) {
        m->w_off.push_back(off);
        off += (int64_t)m->dims[(size_t)l] * m->dims[(size_t)l + 1];
        m->b_off.push_back(off);
        off += m->dims[(size_t)l + 1];
    }
    m->n_params = off;
    int st = mlp_init(m, seed);
    if (st) {
        delete m;
        *status = st;
        return nullptr;
    }
    *status = AZD_OK;
    return m;
}

} // namespace azd

// the bf16 forward GEMM in isolation (tools/time_gemm.py, tests): device pointers, A [M][Kp] and W [N][Kp] bf16 with Kp a
// multiple of 64 (zero beyond the real K), Y f32 or bf16 [M][ldy]; *ms = mean GPU time of `reps` launches after one warm-up
extern "C" int azd_debug_gemm_bf16(int device, int M, int N, int Kp, const void *d_a, const void *d_w, const float *d_bias, void *d_y, int ldy,
                                   int out_bf16, int act, int reps, float *ms) {
    if (M <= 0 || N <= 0 || Kp <= 0 || Kp % 64 != 0 || !d_a || !d_w || !d_y || reps < 1) return AZD_ERR_INVALID_ARGUMENT;
    AZD_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    AZD_HIP(hipGetDeviceProperties(&prop, device));
    hipStream_t st;
    AZD_HIP(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    AZD_HIP(hipEventCreate(&e0));
    AZD_HIP(hipEventCreate(&e1));
    const int force_bn = getenv("AZD_GEMM16_BN") ? atoi(getenv("AZD_GEMM16_BN")) : 0; // (this entry only: tile experiments)
    azd::launch_gemm16(st, (const uint16_t *)d_a, Kp, (const uint16_t *)d_w, Kp, d_y, ldy, M, N, Kp, act, out_bf16, d_bias, prop.multiProcessorCount, force_bn);
    AZD_HIP(hipEventRecord(e0, st));
    for (int r = 0; r < reps; ++r)
        azd::launch_gemm16(st, (const uint16_t *)d_a, Kp, (const uint16_t *)d_w, Kp, d_y, ldy, M, N, Kp, act, out_bf16, d_bias, prop.multiProcessorCount, force_bn);
    AZD_HIP(hipEventRecord(e1, st));
    AZD_HIP(hipStreamSynchronize(st));
    AZD_HIP(hipGetLastError());
    float t = 0.f;
    AZD_HIP(hipEventElapsedTime(&t, e0, e1));
    if (ms) *ms = t / (float)reps;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipStreamDestroy(st);
    return AZD_OK;
}
