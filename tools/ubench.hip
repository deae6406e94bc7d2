// ubench.hip -- single-wave latency micro-benchmarks (gfx950) that inform the kernel structure.
// Build & run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/ubench.hip -o /tmp/ubench && /tmp/ubench
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#define N_IT 4096

__global__ void k_f64_chain(double *out, double a, double b, unsigned long long *cyc) {
    double x = a + threadIdx.x;
    unsigned long long t0 = clock64();
    for (int i = 0; i < N_IT; ++i) {
        x = x * b;
        x = x + a;
    }
    unsigned long long t1 = clock64();
    out[threadIdx.x] = x;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_f32_chain(float *out, float a, float b, unsigned long long *cyc) {
    float x = a + threadIdx.x;
    unsigned long long t0 = clock64();
    for (int i = 0; i < N_IT; ++i) {
        x = x * b;
        x = x + a;
    }
    unsigned long long t1 = clock64();
    out[threadIdx.x] = x;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_f64_indep(double *out, double a, double b, unsigned long long *cyc) {
    double x0 = a + threadIdx.x, x1 = a - threadIdx.x, x2 = a * 2, x3 = a * 3;
    unsigned long long t0 = clock64();
    for (int i = 0; i < N_IT; ++i) {
        x0 = x0 * b; x1 = x1 * b; x2 = x2 * b; x3 = x3 * b;
        x0 = x0 + a; x1 = x1 + a; x2 = x2 + a; x3 = x3 + a;
    }
    unsigned long long t1 = clock64();
    out[threadIdx.x] = x0 + x1 + x2 + x3;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
// LDS dependent round trip: write then read back through a lane-private column, b64
__global__ void k_lds_rt(double *out, double a, unsigned long long *cyc) {
    __shared__ double buf[64 * 8];
    double x = a + threadIdx.x;
    buf[threadIdx.x] = x;
    __syncthreads();
    unsigned long long t0 = clock64();
    for (int i = 0; i < N_IT; ++i) {
        double y = buf[(i & 7) * 64 + threadIdx.x];
        x = y + x;
        buf[((i + 1) & 7) * 64 + threadIdx.x] = x;
    }
    unsigned long long t1 = clock64();
    out[threadIdx.x] = x;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
// uniform taken branches: data-dependent scalar condition the compiler cannot remove
__global__ void k_branch(int *out, const int *pat, unsigned long long *cyc) {
    int acc = threadIdx.x;
    unsigned long long t0 = clock64();
    for (int i = 0; i < N_IT; ++i) {
        int p = __builtin_amdgcn_readfirstlane(pat[i & 63]);
        if (p & 1) acc += 3; else acc ^= 5;
        if (p & 2) acc *= 7; else acc -= 1;
        if (p & 4) acc += 11; else acc ^= 9;
        if (p & 8) acc -= 13; else acc += 2;
    }
    unsigned long long t1 = clock64();
    out[threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_branch_base(int *out, const int *pat, unsigned long long *cyc) {
    int acc = threadIdx.x;
    unsigned long long t0 = clock64();
    for (int i = 0; i < N_IT; ++i) {
        int p = __builtin_amdgcn_readfirstlane(pat[i & 63]);
        acc += p;
    }
    unsigned long long t1 = clock64();
    out[threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
// global memory dependent pointer chase (uniform address), 32 MB footprint
__global__ void k_chase(const unsigned *next, unsigned start, int steps, unsigned *out, unsigned long long *cyc) {
    unsigned p = start;
    unsigned long long t0 = clock64();
    for (int i = 0; i < steps; ++i) p = next[p];
    unsigned long long t1 = clock64();
    out[0] = p;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
// f64 division chain
__global__ void k_div_chain(double *out, double a, unsigned long long *cyc) {
    double x = a + threadIdx.x;
    unsigned long long t0 = clock64();
    for (int i = 0; i < N_IT; ++i) x = a / x + 1.0;
    unsigned long long t1 = clock64();
    out[threadIdx.x] = x;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

int main() {
    unsigned long long *d_cyc, h;
    double *d_out;
    int *d_pat;
    hipMalloc(&d_cyc, 8);
    hipMalloc(&d_out, 64 * 8);
    hipMalloc(&d_pat, 64 * 4);
    std::vector<int> pat(64);
    for (int i = 0; i < 64; ++i) pat[i] = (i * 2654435761u) >> 28;
    hipMemcpy(d_pat, pat.data(), 256, hipMemcpyHostToDevice);
#define RUN(name, launch, per)                                                  \
    for (int rep = 0; rep < 2; ++rep) { launch; hipDeviceSynchronize(); }      \
    hipMemcpy(&h, d_cyc, 8, hipMemcpyDeviceToHost);                             \
    printf("%-28s %8.1f cycles per %s\n", name, (double)h / N_IT, per);
    RUN("f64 mul+add dependent", (k_f64_chain<<<1, 64>>>(d_out, 1.0000001, 0.9999999, d_cyc)), "mul+add pair");
    RUN("f32 mul+add dependent", (k_f32_chain<<<1, 64>>>((float *)d_out, 1.0000001f, 0.9999999f, d_cyc)), "mul+add pair");
    RUN("f64 4 indep mul+add", (k_f64_indep<<<1, 64>>>(d_out, 1.0000001, 0.9999999, d_cyc)), "4 pairs");
    RUN("LDS b64 read->add->write", (k_lds_rt<<<1, 64>>>(d_out, 1.0, d_cyc)), "round trip");
    RUN("4 uniform branches", (k_branch<<<1, 64>>>((int *)d_out, d_pat, d_cyc)), "iteration");
    RUN("branch loop baseline", (k_branch_base<<<1, 64>>>((int *)d_out, d_pat, d_cyc)), "iteration");
    RUN("f64 div dependent", (k_div_chain<<<1, 64>>>(d_out, 1.5, d_cyc)), "div+add");
    // pointer chase
    {
        const unsigned n = 8u << 20; // 32 MB of u32
        std::vector<unsigned> nx(n);
        unsigned long long st = 88172645463325252ull;
        std::vector<unsigned> perm(n);
        for (unsigned i = 0; i < n; ++i) perm[i] = i;
        for (unsigned i = n - 1; i > 0; --i) {
            st ^= st << 13; st ^= st >> 7; st ^= st << 17;
            unsigned j = (unsigned)(st % (i + 1));
            unsigned t = perm[i]; perm[i] = perm[j]; perm[j] = t;
        }
        for (unsigned i = 0; i < n; ++i) nx[perm[i]] = perm[(i + 1) % n];
        unsigned *d_nx, *d_o;
        hipMalloc(&d_nx, (size_t)n * 4);
        hipMalloc(&d_o, 4);
        hipMemcpy(d_nx, nx.data(), (size_t)n * 4, hipMemcpyHostToDevice);
        for (int rep = 0; rep < 2; ++rep) { k_chase<<<1, 64>>>(d_nx, 0, 2048, d_o, d_cyc); hipDeviceSynchronize(); }
        hipMemcpy(&h, d_cyc, 8, hipMemcpyDeviceToHost);
        printf("%-28s %8.1f cycles per %s\n", "global chase 32MB (1 wave)", (double)h / 2048, "load");
        // same with 4096 waves in flight
        for (int rep = 0; rep < 2; ++rep) { k_chase<<<4096, 64>>>(d_nx, 0, 2048, d_o, d_cyc); hipDeviceSynchronize(); }
        hipMemcpy(&h, d_cyc, 8, hipMemcpyDeviceToHost);
        printf("%-28s %8.1f cycles per %s\n", "global chase, 4096 waves", (double)h / 2048, "load");
    }
    int clk = 0;
    hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
    printf("clock rate attr %d kHz; clock64 ticks are shader cycles\n", clk);
    return 0;
}
