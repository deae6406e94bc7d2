// chase_latency.hip -- diagnostic: round-trip time of the tree search's dependent gathers.
// Every wave chases through a footprint of F bytes: 64 lanes read 16 B each of one random 1 KB block (a node's prediction
// records), the next block depends on what came back.  Prints ns per round trip for footprints from L2-sized to
// population-sized, with `waves` waves per CU chasing at once (the pool step's searchers: 16 per CU on 168 CUs).
//   hipcc --offload-arch=gfx950 -O3 -o chase_latency chase_latency.hip && ./chase_latency
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t x) {
    x += 0x9e3779b97f4a7c15ull;
    x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ull;
    x = (x ^ (x >> 27)) * 0x94d049bb133111ebull;
    return x ^ (x >> 31);
}

__global__ __launch_bounds__(1024) void k_fill(uint32_t *p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (uint32_t)mix(i);
}

// block_bytes: 1024 (64 lanes x 16 B); per-wave region: each wave chases inside its own slice of the footprint when
// `sliced` (an agent's tree), else anywhere
__global__ __launch_bounds__(1024) void k_chase(const uint4 *base, uint64_t n_blocks, int iters, int active_waves, unsigned long long *out) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wave >= active_waves) return;
    uint64_t idx = mix((uint64_t)blockIdx.x * 64 + wave);
    uint32_t acc = 0;
    const unsigned long long t0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
        const uint64_t b = idx % n_blocks;
        const uint4 v = base[b * 64 + lane];
        acc ^= v.x;
        const uint32_t first = (uint32_t)__builtin_amdgcn_readfirstlane((int)v.y);
        idx = mix(idx ^ first);
    }
    const unsigned long long t1 = wall_clock64();
    if (lane == 0) {
        atomicAdd(out, t1 - t0);
        atomicAdd(out + 1, (unsigned long long)(acc & 1u));
    }
}

int main() {
    const size_t max_bytes = (size_t)24 << 30;
    uint4 *buf = nullptr;
    CHECK(hipMalloc(&buf, max_bytes));
    k_fill<<<4096, 1024>>>((uint32_t *)buf, max_bytes / 4);
    CHECK(hipDeviceSynchronize());
    unsigned long long *out = nullptr;
    CHECK(hipMalloc(&out, 16));
    const int iters = 2000;
    printf("%-12s %6s %6s %10s\n", "footprint", "WGs", "waves", "ns/trip");
    const size_t fps[] = {(size_t)16 << 20, (size_t)128 << 20, (size_t)1 << 30, (size_t)3 << 30, (size_t)6 << 30, (size_t)12 << 30, (size_t)24 << 30};
    const int cfgs[][2] = {{1, 1}, {168, 4}, {168, 16}};
    for (size_t fp : fps)
        for (auto &c : cfgs) {
            CHECK(hipMemset(out, 0, 16));
            k_chase<<<c[0], 1024>>>(buf, fp / 1024, iters, c[1], out);
            CHECK(hipDeviceSynchronize());
            unsigned long long h[2];
            CHECK(hipMemcpy(h, out, 16, hipMemcpyDeviceToHost));
            const double ns = (double)h[0] * 10.0 / ((double)c[0] * c[1] * iters); // wall_clock64: 100 MHz
            printf("%8.0f MB %6d %6d %10.0f\n", (double)fp / (1 << 20), c[0], c[1], ns);
        }
    return 0;
}
