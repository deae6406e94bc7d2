// Microbenchmark of the evaluator's tile task (async_step.inc:mlp_tile_task) in isolation:
// one workgroup of 16 waves per CU, `servers` of them run 16x16 tile tasks of a 256-wide layer
// back to back, the others sleep.  Variants isolate the weight stream, the LDS reads and the MFMAs.
//   hipcc --offload-arch=gfx950 -O3 tile_probe.hip -o tile_probe && ./tile_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define K 256
#define NCOL 256
constexpr int PF = 8;

template <int VAR>
__global__ __launch_bounds__(1024) void k_probe(const float *__restrict__ W, const float *__restrict__ Wp, float *out,
                                                unsigned long long *ticks, int servers, int reps,
                                                const uint4 *__restrict__ heap, unsigned heap_mask, int chasers) {
    extern __shared__ __align__(16) float lds[]; // 16 rows x (K + 4 skew); 16-B aligned: a misaligned ds_read_b128 is split and halves the loop's speed
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 16 * (K + 4); i += 1024) lds[i] = (float)(i & 7) * 0.125f;
    __syncthreads();
#ifdef V1
    if (wave >= servers) return;
#else
    __shared__ __align__(16) unsigned done_cnt[4];
    if (threadIdx.x == 0) done_cnt[0] = 0;
    __syncthreads();
    if (wave >= servers) {
        if (wave - servers >= chasers) return;
        // a searcher's memory behaviour: dependent 64-lane gathers of 16 B from a heap far larger than the caches
        unsigned idx = (blockIdx.x * 1024 + threadIdx.x) * 2654435761u;
        unsigned acc = 0;
        while (__hip_atomic_load(&done_cnt[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < (unsigned)servers) {
            for (int i = 0; i < 4; ++i) {
                const uint4 v = heap[idx & heap_mask];
                idx = idx * 1664525u + 1013904223u + (v.x & 1u);
                acc += v.y;
            }
        }
        out[(size_t)blockIdx.x * 1024 + threadIdx.x] = (float)acc;
        return;
    }
#endif
    const int r = lane & 15;
    const float *ap = lds + r * (K + 4) + 4 * (lane >> 4);
    f32x4 tot = {0.f, 0.f, 0.f, 0.f};
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int rep = 0; rep < reps; ++rep) {
        const int tile = (wave + rep * 5 + blockIdx.x) & 15;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        if (VAR == 2) { // fragment-major packed weights: one contiguous 1 KB per k-step
            const float *wp = Wp + (size_t)tile * 16 * K + lane * 4;
            for (int s0 = 0; s0 < 16; s0 += PF) {
                float4 bq[PF];
#pragma unroll
                for (int j = 0; j < PF; ++j) bq[j] = *reinterpret_cast<const float4 *>(wp + 256 * (s0 + j));
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < PF; ++j) {
                    const float4 av = *reinterpret_cast<const float4 *>(ap + 16 * (s0 + j));
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bq[j].x, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bq[j].y, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, bq[j].z, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bq[j].w, acc, 0, 0, 0);
                }
            }
        } else {
            const float *wp = W + (size_t)(tile * 16 + r) * K + 4 * (lane >> 4);
            constexpr int P = VAR == 1 ? 16 : PF;
            for (int s0 = 0; s0 < 16; s0 += P) {
                float4 bq[P];
#pragma unroll
                for (int j = 0; j < P; ++j) {
                    if (VAR == 4) bq[j] = make_float4(1.f, 2.f, 3.f, (float)j);
                    else bq[j] = *reinterpret_cast<const float4 *>(wp + 16 * (s0 + j));
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < P; ++j) {
                    const float4 av = *reinterpret_cast<const float4 *>(ap + 16 * (s0 + j));
                    if (VAR == 3) {
                        acc[0] += av.x * bq[j].x; acc[1] += av.y * bq[j].y; acc[2] += av.z * bq[j].z; acc[3] += av.w * bq[j].w;
                    } else {
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bq[j].x, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bq[j].y, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, bq[j].z, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bq[j].w, acc, 0, 0, 0);
                    }
                }
            }
        }
        tot += acc;
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    #ifdef V1
    if (lane == 0) ticks[blockIdx.x * 16 + wave] = t1 - t0;
#else
    if (lane == 0) { ticks[blockIdx.x * 16 + wave] = t1 - t0; atomicAdd(&done_cnt[0], 1u); }
#endif
    out[(size_t)blockIdx.x * 1024 + threadIdx.x] = tot[0] + tot[1] + tot[2] + tot[3];
}

static int g_reps = 3000;
template <int VAR>
static void run(const char *name, const float *W, const float *Wp, float *out, unsigned long long *ticks, int nblk, const uint4 *heap, unsigned heap_mask, int chasers) {
    const int reps = g_reps;
    const size_t lds = 16 * (K + 4) * sizeof(float);
    for (int servers : {1, 4, 6, 8, 16}) {
        if (servers + chasers > 16) continue;
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        k_probe<VAR><<<nblk, 1024, lds>>>(W, Wp, out, ticks, servers, 1000, heap, heap_mask, chasers); // warm-up: clocks up, weights in L2
        hipEventRecord(e0);
        k_probe<VAR><<<nblk, 1024, lds>>>(W, Wp, out, ticks, servers, reps, heap, heap_mask, chasers);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("%-34s servers %2d: %.2f us per tile task (kernel %.2f ms; %.1f us per round of 16 tiles/CU)\n", name, servers,
               ms * 1e3 / reps, ms, ms * 1e3 / reps * 16 / servers);
    }
}

int main(int argc, char **argv) {
    if (argc > 1) g_reps = atoi(argv[1]);
    const bool quick = argc > 2;
    const int nblk = 256;
    std::vector<float> h((size_t)NCOL * K), hp((size_t)NCOL * K);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) >> 20 & 255) / 256.f - 0.5f;
    for (int tile = 0; tile < 16; ++tile)
        for (int s = 0; s < 16; ++s)
            for (int lane = 0; lane < 64; ++lane)
                for (int i = 0; i < 4; ++i)
                    hp[(size_t)tile * 16 * K + s * 256 + lane * 4 + i] = h[(size_t)(tile * 16 + (lane & 15)) * K + 16 * s + 4 * (lane >> 4) + i];
    float *W, *Wp, *out;
    unsigned long long *ticks;
    hipMalloc(&W, h.size() * 4); hipMalloc(&Wp, h.size() * 4); hipMalloc(&out, (size_t)nblk * 1024 * 4); hipMalloc(&ticks, nblk * 16 * 8);
    hipMemcpy(W, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(Wp, hp.data(), h.size() * 4, hipMemcpyHostToDevice);
    uint4 *heap = nullptr;
    const unsigned heap_mask = (1u << 27) - 1; // 2^27 x 16 B = 2 GiB
    if (quick) { // the three variants that matter, nothing else on the device before them
        run<4>("MFMA + LDS, no weight loads", W, Wp, out, ticks, nblk, nullptr, 0, 0);
        run<0>("row-major W, PF 8", W, Wp, out, ticks, nblk, nullptr, 0, 0);
        run<2>("fragment-major W, PF 8 (product)", W, Wp, out, ticks, nblk, nullptr, 0, 0);
        return 0;
    }
    hipMalloc(&heap, ((size_t)heap_mask + 1) * 16);
    hipMemset(heap, 0, ((size_t)heap_mask + 1) * 16);
    run<0>("row-major W, PF 8", W, Wp, out, ticks, nblk, heap, heap_mask, 0);
    run<1>("row-major W, PF 16", W, Wp, out, ticks, nblk, heap, heap_mask, 0);
    run<2>("fragment-major W, PF 8 (product)", W, Wp, out, ticks, nblk, heap, heap_mask, 0);
    run<3>("loads + LDS, VALU instead of MFMA", W, Wp, out, ticks, nblk, heap, heap_mask, 0);
    run<4>("MFMA + LDS, no weight loads", W, Wp, out, ticks, nblk, heap, heap_mask, 0);
    // the same with other waves of the workgroup chasing gathers through HBM, as searching agents do
    run<2>("fragment-major + 4 gather waves", W, Wp, out, ticks, nblk, heap, heap_mask, 4);
    run<2>("fragment-major + 8 gather waves", W, Wp, out, ticks, nblk, heap, heap_mask, 8);
    run<4>("no weight loads + 8 gather waves", W, Wp, out, ticks, nblk, heap, heap_mask, 8);
    return 0;
}
