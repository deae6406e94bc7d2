// gather_calib.hip -- calibrates rocprofv3's FETCH_SIZE for THIS build's access patterns (MI355X_MICROARCH.md: the
// counter is calibrated for wide coalesced streaming reads only, where it reports half the bytes; "other access widths
// are uncalibrated: calibrate on a known byte count in your own access pattern").  Three kernels over a 2 GiB table
// (far beyond the 256 MiB Infinity Cache), each requesting 1 GiB in all:
//   k_stream16   coalesced 16-B-per-lane streaming read                (the guide's calibrated case)
//   k_gather16   one random 16-B record per lane  (PredRec / ArcRec gathers of the tree search)
//   k_gather32   one random 32-B record per lane  (NodeRec gathers)
// run:  hipcc --offload-arch=gfx950 -O3 tools/probes/gather_calib.hip -o /tmp/gather_calib
//       rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -- /tmp/gather_calib
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

__device__ __forceinline__ uint64_t mix(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
__global__ void k_stream16(const uint4 *t, size_t n, uint32_t *out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint4 v = t[i % n];
    if ((v.x ^ v.y ^ v.z ^ v.w) == 0x12345u) out[0] = 1;
}
__global__ void k_gather16(const uint4 *t, size_t n, uint32_t *out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint4 v = t[mix(i) % n];
    if ((v.x ^ v.y ^ v.z ^ v.w) == 0x12345u) out[0] = 1;
}
__global__ void k_gather32(const uint4 *t, size_t n, uint32_t *out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t r = (mix(i) % (n / 2)) * 2;
    uint4 v = t[r], w = t[r + 1];
    if ((v.x ^ v.y ^ v.z ^ v.w ^ w.x ^ w.y ^ w.z ^ w.w) == 0x12345u) out[0] = 1;
}
int main() {
    const size_t bytes = 2ull << 30, n = bytes / 16;
    uint4 *t;
    uint32_t *out;
    if (hipMalloc(&t, bytes) != hipSuccess || hipMalloc(&out, 4) != hipSuccess) return 1;
    hipMemset(t, 1, bytes);
    const size_t req = 1ull << 30;
    const int bs = 256;
    k_stream16<<<dim3((unsigned)(req / 16 / bs)), dim3(bs)>>>(t, n, out);
    k_gather16<<<dim3((unsigned)(req / 16 / bs)), dim3(bs)>>>(t, n, out);
    k_gather32<<<dim3((unsigned)(req / 32 / bs)), dim3(bs)>>>(t, n, out);
    hipError_t e = hipDeviceSynchronize();
    printf("requested bytes per kernel: %zu (%s)\n", req, hipGetErrorString(e));
    return e == hipSuccess ? 0 : 1;
}
