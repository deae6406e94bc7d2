// group_exchange.hip -- probe for the column-split evaluator of the pool step (round-4 verdict, item 1; DESIGN.md section 4 "k_pool").
//
// A GROUP of g workgroups (one per CU) serves evaluator batches together: member j keeps the weight fragments of its column
// tiles of EVERY layer in LDS for the whole launch, a batch's activations travel between the layers through a fragment-major
// exchange buffer in device memory (sc1 write-through stores, the storing wave's vmcnt(0), one agent-scope add on the layer's
// arrival counter; consumers poll the counter with sc1 loads and read their A operands with sc1 buffer loads straight into the
// MFMA registers).  A workgroup's 16 waves are NS = 16 / W batch slots of W waves (W = column tiles per member and layer): a
// slot of a group = W waves on each of its g CUs, walking a batch through the layers; slots run batches independently.
// What the probe measures (us per batch and per layer, us of MFMA chain alone, us of polls), for the config B model
// 304-256-256-256-152 f32, 16-row batches:
//   * group on ONE XCD (members = blocks with equal blockIdx % 8) against members dealt over the 8 XCDs;
//   * 1 .. NS slots running at once;
//   * with and without the rest of the chip chasing dependent gathers through HBM (what the searchers do);
//   * against the classic form: ONE workgroup streaming the whole model from L2 per batch (pool_eval today).
// Every output row is checked against a host f64 forward of the same rows (stale or torn exchange data would show).
//   hipcc --offload-arch=gfx950 -O3 -o group_exchange group_exchange.hip && ./group_exchange
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int NL = 4;
constexpr int MAXG = 16;
struct Model {
    int dims[NL + 1];   // 304 256 256 256 152
    int steps[NL];      // k-steps of 16 per layer
    int tiles[NL];      // column tiles of 16 per layer
    size_t w_off[NL];   // float offset of the layer's fragment-major weights: [tile][step][64 lanes][4]
    size_t b_off[NL];   // float offset of the layer's bias in `bias`
};
struct Args {
    Model m;
    const float *wfrag;  // fragment-major weights
    const float *bias;
    const float *states; // [n_rows_total][S]
    float *out;          // [n_rows_total][A]
    float *xbuf;         // [groups][NS][2][16 steps][64][4]  (fragment-major activations, 16 KB per buffer at width 256)
    uint32_t *desc;      // [groups][NS][64]: word 0 = gen, words 16.. = row ids
    uint32_t *cnt;       // [groups][NS][NL][32]: arrival counters on lines of their own
    unsigned long long *stat; // [groups][NS][8]: leader's ticks: total, per layer boundary
    uint32_t *bg_stop;
    int g, W, NS, active_slots, n_batches, spread, n_group_blocks, n_groups, bg_iters;
    const uint4 *bg_buf;
    uint64_t bg_blocks;
};

extern __shared__ __attribute__((aligned(16))) char dyn_lds[];

__device__ __forceinline__ uint32_t ld_sc1(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_sc1(uint32_t *p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
#define VM_DRAIN() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
// every poll gives up after 2 s of wall clock (100 MHz ticks) and raises the probe's abort flag, which every poll also tests
#define POLL_UNTIL(cond, sleep)                                                                   \
    do {                                                                                          \
        const unsigned long long pw0_ = wall_clock64();                                           \
        uint32_t it_ = 0;                                                                         \
        while (!(cond)) {                                                                         \
            __builtin_amdgcn_s_sleep(sleep);                                                      \
            if ((++it_ & 63u) == 0u && (ld_sc1(a.bg_stop + 1) != 0u || wall_clock64() - pw0_ > 200000000ull)) { \
                st_sc1(a.bg_stop + 1, 1u);                                                        \
                aborted = true;                                                                   \
                break;                                                                            \
            }                                                                                     \
        }                                                                                         \
    } while (0)
__device__ __forceinline__ uint64_t mix(uint64_t x) {
    x += 0x9e3779b97f4a7c15ull;
    x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ull;
    x = (x ^ (x >> 27)) * 0x94d049bb133111ebull;
    return x ^ (x >> 31);
}
__device__ __forceinline__ uint32_t xcc_id() {
    uint32_t v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 7u;
}
__device__ __forceinline__ f32x4 buf_ld(__amdgpu_buffer_rsrc_t r, uint32_t byte_off) {
    u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 16); // aux 16 = sc1
    f32x4 f;
    __builtin_memcpy(&f, &v, 16);
    return f;
}

// the tile's LDS slot: tiles of member j, in (layer, w) order
__device__ __forceinline__ uint32_t lds_tile_off(const Model &m, int g, int member, int l, int w) {
    uint32_t off = 0;
    for (int ll = 0; ll < NL; ++ll)
        for (int ww = 0; ww * g + member < m.tiles[ll]; ++ww) {
            if (ll == l && ww == w) return off;
            off += (uint32_t)m.steps[ll] * 1024u;
        }
    return off;
}

// MODE 0: group exchange; MODE 1: classic (one workgroup per batch, weights streamed from L2, activations in LDS)
__global__ __launch_bounds__(1024) void k_probe(Args a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const Model &m = a.m;
    if ((int)blockIdx.x >= a.n_group_blocks) { // background: dependent gathers through HBM until the groups are through
        uint64_t idx = mix((uint64_t)blockIdx.x * 64 + wave);
        uint32_t acc = 0;
        for (int i = 0; i < a.bg_iters; ++i) {
            const uint64_t b = idx % a.bg_blocks;
            const uint4 v = a.bg_buf[b * 64 + lane];
            acc ^= v.x;
            idx = mix(idx ^ (uint32_t)__builtin_amdgcn_readfirstlane((int)v.y));
            if ((i & 31) == 31 && ld_sc1(a.bg_stop) != 0u) break;
        }
        if (acc == 0x12345u && lane == 0) a.out[0] = 1.f;
        return;
    }
    int group, member;
    if (a.spread) { // members of a group on consecutive blocks = 8 different XCDs
        group = blockIdx.x / a.g;
        member = blockIdx.x % a.g;
    } else {        // members of a group on blocks of equal blockIdx % 8 = one XCD (observed placement; speed only)
        const int lane8 = blockIdx.x & 7, idx = blockIdx.x >> 3;
        group = lane8 + 8 * (idx / a.g);
        member = idx % a.g;
    }
    if (group >= a.n_groups) return;
    // ---- weights of this member's tiles into LDS, once
    for (int l = 0; l < NL; ++l)
        for (int w = 0; w * a.g + member < m.tiles[l]; ++w) {
            const int tile = w * a.g + member;
            const uint32_t off = lds_tile_off(m, a.g, member, l, w);
            const float *src = a.wfrag + m.w_off[l] + (size_t)tile * m.steps[l] * 256;
            for (int i = threadIdx.x; i < m.steps[l] * 64; i += 1024)
                *reinterpret_cast<f32x4 *>(dyn_lds + off + (size_t)i * 16) = *reinterpret_cast<const f32x4 *>(src + (size_t)i * 4);
        }
    __syncthreads();
    const int slot = wave / a.W, tw = wave % a.W;
    if (slot >= a.active_slots) return;
    const size_t si = (size_t)group * a.NS + slot;
    uint32_t *desc = a.desc + si * 64;
    uint32_t *cnt = a.cnt + si * NL * 32;
    float *xb = a.xbuf + si * 2 * 16 * 256;
    const bool leader = member == 0 && tw == 0;
    const __amdgpu_buffer_rsrc_t r_states = __builtin_amdgcn_make_buffer_rsrc((void *)a.states, 0, 0x7FFFFFFF, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_x = __builtin_amdgcn_make_buffer_rsrc((void *)xb, 0, 0x7FFFFFFF, 0x00020000);
    unsigned long long t_total = 0, t_layer[NL] = {0, 0, 0, 0}, t_poll = 0, t_mfma = 0;
    const int r = lane & 15, q = lane >> 4;
    bool aborted = false;
    for (int gen = 1; gen <= a.n_batches && !aborted; ++gen) {
        unsigned long long t0 = 0;
        if (leader) { // "form" a batch: 16 rows of this slot's own share of the input
            if (lane < 16) st_sc1(desc + 16 + lane, (uint32_t)((si * a.n_batches + (gen - 1)) * 16 + lane));
            VM_DRAIN();
            t0 = wall_clock64();
            if (lane == 0) st_sc1(desc, (uint32_t)gen);
        } else {
            const unsigned long long p0 = wall_clock64();
            POLL_UNTIL(ld_sc1(desc) >= (uint32_t)gen, 2);
            t_poll += wall_clock64() - p0;
        }
        const uint32_t row_id = ld_sc1(desc + 16 + r);
        unsigned long long tl = wall_clock64();
        for (int l = 0; l < NL && !aborted; ++l) {
            const int tile = tw * a.g + member;
            const bool has = tile < m.tiles[l];
            const int steps = m.steps[l];
            if (l > 0) { // every tile of the layer before is in memory
                const unsigned long long p0 = wall_clock64();
                const uint32_t want = (uint32_t)gen * (uint32_t)m.tiles[l - 1];
                POLL_UNTIL(ld_sc1(cnt + (l - 1) * 32) >= want, 1);
                t_poll += wall_clock64() - p0;
            }
            if (leader) {
                const unsigned long long now = wall_clock64();
                if (l > 0) t_layer[l - 1] += now - tl;
                tl = now;
            }
            if (!has) continue;
            const unsigned long long m0 = wall_clock64();
            const uint32_t woff = lds_tile_off(m, a.g, member, l, tw);
            const f32x4 *wl = reinterpret_cast<const f32x4 *>(dyn_lds + woff) + lane;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            const float *xin = nullptr;
            const uint32_t a_base = l == 0 ? (row_id * (uint32_t)m.dims[0] + 4u * (uint32_t)q) * 4u
                                           : ((uint32_t)(((l - 1) & 1) * 16 * 256) + (uint32_t)lane * 4u) * 4u;
            const uint32_t a_step = l == 0 ? 64u : 1024u; // bytes per k-step
            (void)xin;
            for (int s0 = 0; s0 < steps; s0 += 8) {
                f32x4 av[8], bv[8];
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (s0 + j < steps) av[j] = buf_ld(l == 0 ? r_states : r_x, a_base + (uint32_t)(s0 + j) * a_step);
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (s0 + j < steps) bv[j] = wl[(s0 + j) * 64];
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (s0 + j < steps) {
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j].x, bv[j].x, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j].y, bv[j].y, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j].z, bv[j].z, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j].w, bv[j].w, acc, 0, 0, 0);
                    }
            }
            const int col = tile * 16 + r;
            const float bj = col < m.dims[l + 1] ? a.bias[m.b_off[l] + col] : 0.f;
            if (l == NL - 1) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float v = 1.0f / (1.0f + __expf(-(acc[i] + bj)));
                    const uint32_t rid = __shfl(row_id, 4 * q + i); // (every lane takes part: the last tile's upper columns are masked below)
                    if (col < m.dims[l + 1]) __hip_atomic_store(a.out + (size_t)rid * m.dims[NL] + col, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            } else { // fragment-major for the next layer: element (row R, k = col) at ((tile * 64 + (k & 15) / 4 * 16 + R) * 4 + (k & 3))
                float *xo = xb + (size_t)(l & 1) * 16 * 256;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float v = fmaxf(acc[i] + bj, 0.f);
                    const int R = 4 * q + i;
                    __hip_atomic_store(xo + ((size_t)tile * 64 + (size_t)(r >> 2) * 16 + R) * 4 + (r & 3), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            VM_DRAIN();
            if (lane == 0) __hip_atomic_fetch_add(cnt + l * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            t_mfma += wall_clock64() - m0;
        }
        if (leader) { // the batch is through when the head's tiles are all in
            const uint32_t want = (uint32_t)gen * (uint32_t)m.tiles[NL - 1];
            POLL_UNTIL(ld_sc1(cnt + (NL - 1) * 32) >= want, 1);
            const unsigned long long now = wall_clock64();
            t_layer[NL - 1] += now - tl;
            t_total += now - t0;
        }
    }
    if (lane == 0) {
        unsigned long long *st = a.stat + si * 8;
        if (leader) {
            st[0] = t_total;
            for (int l = 0; l < NL; ++l) st[1 + l] = t_layer[l];
            st[7] = xcc_id();
        }
        atomicAdd(&st[5], t_poll);
        atomicAdd(&st[6], t_mfma);
        if (member != 0 && tw == 0 && xcc_id() != (uint32_t)st[7] && false) st[7] |= 0x100; // (placement is reported by k_xcc below)
    }
    if (threadIdx.x == 0 && blockIdx.x == 0) { /* the last group to finish stops the background: approximated by block 0 */
    }
}

// classic form: one workgroup per batch; weights streamed from L2 in fragment order, activations in LDS
__global__ __launch_bounds__(1024) void k_classic(Args a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const Model &m = a.m;
    if ((int)blockIdx.x >= a.n_group_blocks) {
        uint64_t idx = mix((uint64_t)blockIdx.x * 64 + wave);
        uint32_t acc = 0;
        for (int i = 0; i < a.bg_iters; ++i) {
            const uint64_t b = idx % a.bg_blocks;
            const uint4 v = a.bg_buf[b * 64 + lane];
            acc ^= v.x;
            idx = mix(idx ^ (uint32_t)__builtin_amdgcn_readfirstlane((int)v.y));
            if ((i & 31) == 31 && ld_sc1(a.bg_stop) != 0u) break;
        }
        if (acc == 0x12345u && lane == 0) a.out[0] = 1.f;
        return;
    }
    float *x = reinterpret_cast<float *>(dyn_lds); // [16][304 + 256 + 256]
    const int stride = 304 + 256 + 256 + 4;
    unsigned long long t_total = 0;
    const int r = lane & 15, q = lane >> 4;
    for (int gen = 1; gen <= a.n_batches; ++gen) {
        const unsigned long long t0 = wall_clock64();
        const size_t row0 = ((size_t)blockIdx.x * a.n_batches + (gen - 1)) * 16;
        for (int i = threadIdx.x; i < 16 * 76; i += 1024) {
            const int rr = i / 76, c4 = i % 76;
            *reinterpret_cast<f32x4 *>(x + rr * stride + 4 * c4) = *reinterpret_cast<const f32x4 *>(a.states + (row0 + rr) * 304 + 4 * c4);
        }
        __syncthreads();
        for (int l = 0; l < NL; ++l) {
            const int tile = wave;
            if (tile < m.tiles[l]) {
                const int steps = m.steps[l];
                const float *ap = x + r * stride + (l == 0 ? 0 : 304 + ((l - 1) & 1) * 256) + 4 * q;
                const float *wp = a.wfrag + m.w_off[l] + ((size_t)tile * steps * 64 + lane) * 4;
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                for (int s0 = 0; s0 < steps; s0 += 8) {
                    f32x4 bv[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        if (s0 + j < steps) bv[j] = *reinterpret_cast<const f32x4 *>(wp + 256 * (s0 + j));
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        if (s0 + j < steps) {
                            const f32x4 av = *reinterpret_cast<const f32x4 *>(ap + 16 * (s0 + j));
                            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bv[j].x, acc, 0, 0, 0);
                            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bv[j].y, acc, 0, 0, 0);
                            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, bv[j].z, acc, 0, 0, 0);
                            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bv[j].w, acc, 0, 0, 0);
                        }
                }
                const int col = tile * 16 + r;
                if (col < m.dims[l + 1]) {
                    const float bj = a.bias[m.b_off[l] + col];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int R = 4 * q + i;
                        if (l == NL - 1) a.out[(row0 + R) * m.dims[NL] + col] = 1.0f / (1.0f + __expf(-(acc[i] + bj)));
                        else x[R * stride + 304 + (l & 1) * 256 + col] = fmaxf(acc[i] + bj, 0.f);
                    }
                }
            }
            __syncthreads();
        }
        t_total += wall_clock64() - t0;
    }
    if (threadIdx.x == 0) a.stat[(size_t)blockIdx.x * 8] = t_total;
}

// classic form, software-pipelined: the weight quads of the NEXT group of 8 k-steps are requested before the MFMAs of the current
// one, and a layer's first group before the barrier of the layer before (weights do not depend on activations) -- the same MFMA
// sequence per output element.  In k_classic the four waves of a SIMD leave the layer barrier together, request together and
// compute together: per layer 2 x (a request's latency + 4 waves' MFMAs), 23 us per batch against 13 us of MFMA issue.
// Every group issues exactly 8 requests (a ragged last group repeats its last step's address), so that "the group before is in"
// is always s_waitcnt vmcnt(8); the waits are explicit (the compiler's own insertion drains to vmcnt(0) at every merge).
#define WAIT_VM(n) __builtin_amdgcn_s_waitcnt((((n) & 15) | (((n) >> 4) << 14)) | 0x0F70) // vmcnt = n, expcnt / lgkmcnt left alone
#define LDS_BARRIER()                                                   \
    do {                                                                \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local"); \
        __builtin_amdgcn_s_barrier();                                   \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local"); \
    } while (0)
// the requests and their waits are inline asm: the compiler's wait insertion does not see them (it drains to vmcnt(0) at every
// merge of control flow, and throttles requests into registers it believes in flight); the waits below name the registers they
// release, so that no use can move above them
__device__ __forceinline__ void ld_group8(f32x4 (&b)[8], const float *wp, const int g, const int steps) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int st = 8 * g + j < steps ? 8 * g + j : steps - 1;
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(b[j]) : "v"(wp + 256 * st) : "memory");
    }
}
__device__ __forceinline__ void wait_group8(f32x4 (&b)[8], const bool younger_in_flight) {
    if (younger_in_flight)
        asm volatile("s_waitcnt vmcnt(8)" : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]), "+v"(b[7])::"memory");
    else
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]), "+v"(b[7])::"memory");
}
__device__ __forceinline__ void mfma_group8(f32x4 &acc, const f32x4 (&b)[8], const float *ap, const int g, const int steps) {
    if (8 * g + 8 <= steps) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const f32x4 av = *reinterpret_cast<const f32x4 *>(ap + 16 * (8 * g + j));
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, b[j].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, b[j].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, b[j].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, b[j].w, acc, 0, 0, 0);
        }
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (8 * g + j < steps) {
                const f32x4 av = *reinterpret_cast<const f32x4 *>(ap + 16 * (8 * g + j));
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, b[j].x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, b[j].y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, b[j].z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, b[j].w, acc, 0, 0, 0);
            }
    }
}
__global__ __launch_bounds__(1024) void k_classic2(Args a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const Model &m = a.m;
    if ((int)blockIdx.x >= a.n_group_blocks) {
        uint64_t idx = mix((uint64_t)blockIdx.x * 64 + wave);
        uint32_t acc = 0;
        for (int i = 0; i < a.bg_iters; ++i) {
            const uint64_t b = idx % a.bg_blocks;
            const uint4 v = a.bg_buf[b * 64 + lane];
            acc ^= v.x;
            idx = mix(idx ^ (uint32_t)__builtin_amdgcn_readfirstlane((int)v.y));
            if ((i & 31) == 31 && ld_sc1(a.bg_stop) != 0u) break;
        }
        if (acc == 0x12345u && lane == 0) a.out[0] = 1.f;
        return;
    }
    float *x = reinterpret_cast<float *>(dyn_lds); // [16][304 + 256 + 256]
    const int stride = 304 + 256 + 256 + 4;
    unsigned long long t_total = 0;
    const int r = lane & 15, q = lane >> 4;
    const int tile = wave;
    f32x4 bA[8], bB[8];
    // (tiles of a layer this wave has none of: it still requests -- tile 0's -- so that every wave's counts are the same)
    ld_group8(bA, a.wfrag + m.w_off[0] + ((size_t)(tile < m.tiles[0] ? tile : 0) * m.steps[0] * 64 + lane) * 4, 0, m.steps[0]);
    for (int gen = 1; gen <= a.n_batches; ++gen) {
        const unsigned long long t0 = wall_clock64();
        const size_t row0 = ((size_t)blockIdx.x * a.n_batches + (gen - 1)) * 16;
        for (int i = threadIdx.x; i < 16 * 76; i += 1024) {
            const int rr = i / 76, c4 = i % 76;
            const f32x4 v = *reinterpret_cast<const f32x4 *>(a.states + (row0 + rr) * 304 + 4 * c4);
            *reinterpret_cast<f32x4 *>(x + rr * stride + 4 * c4) = v;
        }
        LDS_BARRIER();
        for (int l = 0; l < NL; ++l) {
            const int steps = m.steps[l], ng = (steps + 7) >> 3;
            const bool has = tile < m.tiles[l];
            const float *ap = x + r * stride + (l == 0 ? 0 : 304 + ((l - 1) & 1) * 256) + 4 * q;
            const float *wp = a.wfrag + m.w_off[l] + ((size_t)(has ? tile : 0) * steps * 64 + lane) * 4;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            for (int g = 0;; g += 2) {
                if (g + 1 < ng) {
                    ld_group8(bB, wp, g + 1, steps);
                    wait_group8(bA, true);
                } else wait_group8(bA, false);
                mfma_group8(acc, bA, ap, g, steps);
                if (g + 1 >= ng) break;
                if (g + 2 < ng) {
                    ld_group8(bA, wp, g + 2, steps);
                    wait_group8(bB, true);
                } else wait_group8(bB, false);
                mfma_group8(acc, bB, ap, g + 1, steps);
                if (g + 2 >= ng) break;
            }
            const int col = tile * 16 + r;
            if (has && col < m.dims[l + 1]) {
                const float bj = a.bias[m.b_off[l] + col];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int R = 4 * q + i;
                    if (l == NL - 1) a.out[(row0 + R) * m.dims[NL] + col] = 1.0f / (1.0f + __expf(-(acc[i] + bj)));
                    else x[R * stride + 304 + (l & 1) * 256 + col] = fmaxf(acc[i] + bj, 0.f);
                }
            }
            { // the next layer's first group (the next batch's, after the head) before the barrier
                const int ln = l + 1 < NL ? l + 1 : 0;
                ld_group8(bA, a.wfrag + m.w_off[ln] + ((size_t)(tile < m.tiles[ln] ? tile : 0) * m.steps[ln] * 64 + lane) * 4, 0, m.steps[ln]);
            }
            LDS_BARRIER();
        }
        t_total += wall_clock64() - t0;
    }
    if (threadIdx.x == 0) a.stat[(size_t)blockIdx.x * 8] = t_total;
}

// classic form with the generated asm k loop (tools/gen_tile_asm.py -> azdopt_amd/csrc/tile_task_asm.inc): within a tile task
// the next group of 8 k-steps is requested before the current group's MFMAs
#include "../../azdopt_amd/csrc/tile_task_asm.inc"
template <int XPF> // XPF = 1: the k loop only (every layer begins with one exposed request)
__global__ __launch_bounds__(1024) void k_classic3(Args a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const Model &m = a.m;
    if ((int)blockIdx.x >= a.n_group_blocks) {
        uint64_t idx = mix((uint64_t)blockIdx.x * 64 + wave);
        uint32_t acc = 0;
        for (int i = 0; i < a.bg_iters; ++i) {
            const uint64_t b = idx % a.bg_blocks;
            const uint4 v = a.bg_buf[b * 64 + lane];
            acc ^= v.x;
            idx = mix(idx ^ (uint32_t)__builtin_amdgcn_readfirstlane((int)v.y));
            if ((i & 31) == 31 && ld_sc1(a.bg_stop) != 0u) break;
        }
        if (acc == 0x12345u && lane == 0) a.out[0] = 1.f;
        return;
    }
    float *x = reinterpret_cast<float *>(dyn_lds); // [16][304 + 256 + 256]
    const int stride = 304 + 256 + 256 + 4;
    unsigned long long t_total = 0;
    const int r = lane & 15, q = lane >> 4;
    const int tile = wave;
    azd_tile_ring32 ring;
    int chain_state = 0;
    for (int gen = 1; gen <= a.n_batches; ++gen) {
        const unsigned long long t0 = wall_clock64();
        const size_t row0 = ((size_t)blockIdx.x * a.n_batches + (gen - 1)) * 16;
        for (int i = threadIdx.x; i < 16 * 76; i += 1024) {
            const int rr = i / 76, c4 = i % 76;
            *reinterpret_cast<f32x4 *>(x + rr * stride + 4 * c4) = *reinterpret_cast<const f32x4 *>(a.states + (row0 + rr) * 304 + 4 * c4);
        }
        __syncthreads();
        for (int l = 0; l < NL; ++l) {
            if (tile < m.tiles[l]) {
                const int steps = m.steps[l];
                const float *ap = x + r * stride + (l == 0 ? 0 : 304 + ((l - 1) & 1) * 256) + 4 * q;
                azd_tile_acc acc = {0.f, 0.f, 0.f, 0.f};
                if (XPF == 2) { // chained: the next layer's first group is requested during this layer's last group
                    const float *nb = (l + 1 < NL && tile < m.tiles[l + 1]) ? a.wfrag + m.w_off[l + 1] + (size_t)tile * m.steps[l + 1] * 256 : nullptr;
                    tile_k_chain_f32(acc, ring, chain_state, a.wfrag + m.w_off[l] + (size_t)tile * steps * 256, nb, (uint32_t)lane * 16u, (uint32_t)(uintptr_t)ap, steps);
                } else
                    tile_k_loop_f32(acc, a.wfrag + m.w_off[l] + (size_t)tile * steps * 256, (uint32_t)lane * 16u, (uint32_t)(uintptr_t)ap, steps);
                const int col = tile * 16 + r;
                if (col < m.dims[l + 1]) {
                    const float bj = a.bias[m.b_off[l] + col];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int R = 4 * q + i;
                        if (l == NL - 1) a.out[(row0 + R) * m.dims[NL] + col] = 1.0f / (1.0f + __expf(-(acc[i] + bj)));
                        else x[R * stride + 304 + (l & 1) * 256 + col] = fmaxf(acc[i] + bj, 0.f);
                    }
                }
            }
            __syncthreads();
        }
        t_total += wall_clock64() - t0;
    }
    if (threadIdx.x == 0) a.stat[(size_t)blockIdx.x * 8] = t_total;
}

__global__ void k_xcc(uint32_t *out) {
    if (threadIdx.x == 0) out[blockIdx.x] = xcc_id();
}

static double urand(uint64_t &s) {
    s = s * 6364136223846793005ull + 1442695040888963407ull;
    return (double)(s >> 11) / 9007199254740992.0;
}

int main(int argc, char **argv) {
    const bool debug = argc > 1;
    Model m;
    const int dims[NL + 1] = {304, 256, 256, 256, 152};
    size_t wtot = 0, btot = 0;
    for (int l = 0; l <= NL; ++l) m.dims[l] = dims[l];
    for (int l = 0; l < NL; ++l) {
        m.steps[l] = (dims[l] + 15) / 16;
        m.tiles[l] = (dims[l + 1] + 15) / 16;
        m.w_off[l] = wtot;
        m.b_off[l] = btot;
        wtot += (size_t)m.tiles[l] * m.steps[l] * 256;
        btot += dims[l + 1];
    }
    // weights W[l][out][in], fragment-major copy: tile t, step s, lane: cols t*16 + (lane & 15), k = 16 s + 4 (lane >> 4) + i
    uint64_t seed = 1;
    std::vector<std::vector<double>> W(NL), Bv(NL);
    std::vector<float> wfrag(wtot, 0.f), bias(btot);
    for (int l = 0; l < NL; ++l) {
        const int K = dims[l], N = dims[l + 1];
        W[l].resize((size_t)N * K);
        Bv[l].resize(N);
        const double sc = 1.0 / sqrt((double)K);
        for (auto &v : W[l]) v = (double)(float)((2 * urand(seed) - 1) * sc);
        for (int n = 0; n < N; ++n) bias[m.b_off[l] + n] = (float)(Bv[l][n] = (double)(float)((2 * urand(seed) - 1) * sc));
        for (int t = 0; t < m.tiles[l]; ++t)
            for (int s = 0; s < m.steps[l]; ++s)
                for (int ln = 0; ln < 64; ++ln)
                    for (int i = 0; i < 4; ++i) {
                        const int col = t * 16 + (ln & 15), k = 16 * s + 4 * (ln >> 4) + i;
                        wfrag[m.w_off[l] + (((size_t)t * m.steps[l] + s) * 64 + ln) * 4 + i] = (col < N && k < K) ? (float)W[l][(size_t)col * K + k] : 0.f;
                    }
    }
    const int n_batches = 200;
    const int max_slots_total = 32 * 16; // groups x NS upper bound
    const size_t n_rows = (size_t)max_slots_total * n_batches * 16;
    std::vector<float> states(n_rows * 304);
    for (auto &v : states) v = urand(seed) < 0.15 ? 1.f : 0.f; // 0/1 state vectors, as the c21 space writes them
    Args a;
    a.m = m;
    float *d_w, *d_b, *d_s, *d_o, *d_x;
    uint32_t *d_desc, *d_cnt, *d_stop, *d_xcc;
    unsigned long long *d_stat;
    uint4 *d_bg;
    const size_t bg_bytes = (size_t)4 << 30;
    CHECK(hipMalloc(&d_w, wtot * 4 + 8192)); // (a ragged last group of a tile task requests up to 7 KB past its tile)
    CHECK(hipMalloc(&d_b, btot * 4));
    CHECK(hipMalloc(&d_s, states.size() * 4));
    CHECK(hipMalloc(&d_o, n_rows * 152 * 4));
    CHECK(hipMalloc(&d_x, (size_t)max_slots_total * 2 * 16 * 256 * 4));
    CHECK(hipMalloc(&d_desc, (size_t)max_slots_total * 64 * 4));
    CHECK(hipMalloc(&d_cnt, (size_t)max_slots_total * NL * 32 * 4));
    CHECK(hipMalloc(&d_stat, (size_t)max_slots_total * 8 * 8));
    CHECK(hipMalloc(&d_stop, 8));
    CHECK(hipMalloc(&d_xcc, 256 * 4));
    CHECK(hipMalloc(&d_bg, bg_bytes));
    CHECK(hipMemset(d_bg, 0x5a, bg_bytes));
    CHECK(hipMemcpy(d_w, wfrag.data(), wtot * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_b, bias.data(), btot * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_s, states.data(), states.size() * 4, hipMemcpyHostToDevice));
    a.wfrag = d_w;
    a.bias = d_b;
    a.states = d_s;
    a.out = d_o;
    a.xbuf = d_x;
    a.desc = d_desc;
    a.cnt = d_cnt;
    a.stat = d_stat;
    a.bg_stop = d_stop;
    a.bg_buf = d_bg;
    a.bg_blocks = bg_bytes / 1024;
    a.n_batches = n_batches;
    k_xcc<<<256, 64>>>(d_xcc);
    std::vector<uint32_t> xcc(256);
    CHECK(hipMemcpy(xcc.data(), d_xcc, 256 * 4, hipMemcpyDeviceToHost));
    int rr_ok = 1;
    for (int b = 8; b < 256; ++b) rr_ok &= xcc[b] == xcc[b - 8];
    printf("placement of a 256-block grid: blocks b and b + 8 on one XCD: %s\n", rr_ok ? "yes" : "NO");
    CHECK(hipFuncSetAttribute((const void *)k_probe, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64));
    CHECK(hipFuncSetAttribute((const void *)k_classic, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64));
    CHECK(hipFuncSetAttribute((const void *)k_classic2, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64));
    CHECK(hipFuncSetAttribute((const void *)k_classic3<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64));
    CHECK(hipFuncSetAttribute((const void *)k_classic3<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64));

    auto host_check = [&](size_t rows_used, const char *tag, int NS, int active) {
        std::vector<float> out(rows_used * 152);
        CHECK(hipMemcpy(out.data(), d_o, out.size() * 4, hipMemcpyDeviceToHost));
        double worst = 0;
        for (size_t rix = 0; rix < rows_used; rix += 97) {
            if ((int)((rix / ((size_t)n_batches * 16)) % NS) >= active) continue; // a slot that did not run
            std::vector<double> x(states.begin() + rix * 304, states.begin() + (rix + 1) * 304), y;
            for (int l = 0; l < NL; ++l) {
                y.assign(dims[l + 1], 0.0);
                for (int n = 0; n < dims[l + 1]; ++n) {
                    double s = Bv[l][n];
                    for (int k = 0; k < dims[l]; ++k) s += W[l][(size_t)n * dims[l] + k] * x[k];
                    y[n] = l == NL - 1 ? 1.0 / (1.0 + exp(-s)) : (s > 0 ? s : 0);
                }
                x = y;
            }
            for (int n = 0; n < 152; ++n) worst = fmax(worst, fabs(x[n] - (double)out[rix * 152 + n]));
        }
        printf("    %s: max |out - f64 forward| over sampled rows = %.2e %s\n", tag, worst, worst < 1e-5 ? "ok" : "MISMATCH");
        if (debug) {
            std::vector<double> x(states.begin(), states.begin() + 304), y;
            for (int l = 0; l < NL; ++l) {
                y.assign(dims[l + 1], 0.0);
                for (int n = 0; n < dims[l + 1]; ++n) {
                    double sacc = Bv[l][n];
                    for (int k = 0; k < dims[l]; ++k) sacc += W[l][(size_t)n * dims[l] + k] * x[k];
                    y[n] = l == NL - 1 ? 1.0 / (1.0 + exp(-sacc)) : (sacc > 0 ? sacc : 0);
                }
                printf("    host layer %d:", l);
                for (int n = 0; n < 6; ++n) printf(" %.6f", y[n]);
                printf("\n");
                x = y;
            }
            printf("    device row 0:");
            for (int n = 0; n < 6; ++n) printf(" %.6f", out[n]);
            printf("\n");
        }
        return worst;
    };

    printf("%-8s %2s %2s %3s %6s %5s %3s | %9s | %7s %7s %7s %7s | %8s %8s\n", "form", "g", "W", "NS", "groups", "slots", "bg", "us/batch", "L0", "L1", "L2", "L3", "poll us", "tile us");
    struct Cfg { int g, n_groups, active, spread, bg; };
    const Cfg cfgs[] = {
        {8, 1, 1, 0, 0}, {8, 1, 1, 1, 0}, {8, 1, 8, 0, 0}, {8, 8, 8, 0, 0}, {8, 8, 8, 1, 0},
        {8, 8, 1, 0, 1}, {8, 8, 4, 0, 1}, {8, 8, 8, 0, 1}, {8, 8, 8, 1, 1}, {8, 12, 8, 0, 1},
        {16, 4, 16, 0, 1}, {16, 6, 16, 0, 1},
    };
    int cfg_i = 0;
    for (const Cfg &c : cfgs) {
        if (debug && cfg_i++ > 0) break;
        a.g = c.g;
        a.W = 16 / c.g > 0 ? 16 / c.g : 1; // hidden layers: 16 tiles over g members
        a.NS = 16 / a.W;
        a.active_slots = c.active < a.NS ? c.active : a.NS;
        a.spread = c.spread;
        a.n_groups = c.n_groups;
        a.n_group_blocks = c.spread ? c.n_groups * c.g : 8 * c.g * ((c.n_groups + 7) / 8); // one-xcd: groups come eight at a time, one per blockIdx % 8
        a.bg_iters = c.bg ? 4000000 : 0;
        const int grid = c.bg ? (a.n_group_blocks > 256 ? a.n_group_blocks : 256) : a.n_group_blocks;
        CHECK(hipMemset(d_desc, 0, (size_t)max_slots_total * 64 * 4));
        CHECK(hipMemset(d_cnt, 0, (size_t)max_slots_total * NL * 32 * 4));
        CHECK(hipMemset(d_stat, 0, (size_t)max_slots_total * 8 * 8));
        CHECK(hipMemset(d_stop, 0, 8));
        CHECK(hipMemset(d_o, 0, n_rows * 152 * 4));
        hipStream_t s2;
        CHECK(hipStreamCreate(&s2));
        k_probe<<<grid, 1024, 160 * 1024 - 64>>>(a);
        if (c.bg) { // stop the background once the group blocks' leaders are through: poll the stats from the host
            std::vector<unsigned long long> st((size_t)max_slots_total * 8);
            for (int spin = 0; spin < 20000; ++spin) {
                CHECK(hipMemcpyAsync(st.data(), d_stat, st.size() * 8, hipMemcpyDeviceToHost, s2));
                CHECK(hipStreamSynchronize(s2));
                int done = 0;
                for (int gi = 0; gi < c.n_groups; ++gi)
                    for (int sl = 0; sl < a.active_slots; ++sl) done += st[((size_t)gi * a.NS + sl) * 8] != 0;
                if (done == c.n_groups * a.active_slots) break;
            }
            uint32_t one = 1;
            CHECK(hipMemcpyAsync(d_stop, &one, 4, hipMemcpyHostToDevice, s2));
            CHECK(hipStreamSynchronize(s2));
        }
        CHECK(hipDeviceSynchronize());
        CHECK(hipStreamDestroy(s2));
        std::vector<unsigned long long> st((size_t)max_slots_total * 8);
        CHECK(hipMemcpy(st.data(), d_stat, st.size() * 8, hipMemcpyDeviceToHost));
        double tot = 0, lay[NL] = {0, 0, 0, 0}, poll = 0, tile = 0;
        int n = 0;
        for (int gi = 0; gi < c.n_groups; ++gi)
            for (int sl = 0; sl < a.active_slots; ++sl) {
                const unsigned long long *p = &st[((size_t)gi * a.NS + sl) * 8];
                tot += (double)p[0];
                for (int l = 0; l < NL; ++l) lay[l] += (double)p[1 + l];
                poll += (double)p[5];
                tile += (double)p[6];
                ++n;
            }
        const double k = 0.01 / ((double)n * n_batches); // 100 MHz ticks -> us per batch
        const double waves_per_slot = (double)a.W * c.g;
        {
            uint32_t ab[2];
            CHECK(hipMemcpy(ab, d_stop, 8, hipMemcpyDeviceToHost));
            if (ab[1]) printf("  (ABORTED: a poll ran into its 2-s bound)\n");
        }
        printf("%-8s %2d %2d %3d %6d %5d %3d | %9.2f | %7.2f %7.2f %7.2f %7.2f | %8.2f %8.2f\n", c.spread ? "spread" : "one-xcd", c.g, a.W, a.NS, c.n_groups, a.active_slots,
               c.bg, tot * k, lay[0] * k, lay[1] * k, lay[2] * k, lay[3] * k, poll * k / waves_per_slot, tile * k / waves_per_slot);
        fflush(stdout);
        host_check((size_t)c.n_groups * a.NS * n_batches * 16 < n_rows ? (size_t)c.n_groups * a.NS * n_batches * 16 : n_rows, "check", a.NS, a.active_slots);
        fflush(stdout);
    }
    // classic: n_wg workgroups, each a batch after the other
    for (int variant = 0; variant < 4; ++variant)
    for (int bg = 0; bg < 2; ++bg)
        for (int n_wg : {1, 64, 104}) {
            if (debug && (n_wg != 1 || bg)) continue;
            if (variant == 1) continue; // k_classic2 (compiler-scheduled pipelining): kept as source for the record, slower and not exact
            a.n_group_blocks = n_wg;
            a.bg_iters = bg ? 4000000 : 0;
            CHECK(hipMemset(d_stat, 0, (size_t)max_slots_total * 8 * 8));
            CHECK(hipMemset(d_stop, 0, 8));
            hipStream_t s2;
            CHECK(hipStreamCreate(&s2));
            CHECK(hipMemset(d_o, 0, n_rows * 152 * 4));
            if (variant == 3) k_classic3<2><<<bg ? 256 : n_wg, 1024, 160 * 1024 - 64>>>(a);
            else if (variant == 2) k_classic3<1><<<bg ? 256 : n_wg, 1024, 160 * 1024 - 64>>>(a);
            else if (variant) k_classic2<<<bg ? 256 : n_wg, 1024, 160 * 1024 - 64>>>(a);
            else k_classic<<<bg ? 256 : n_wg, 1024, 160 * 1024 - 64>>>(a);
            if (bg) {
                std::vector<unsigned long long> st((size_t)n_wg * 8);
                for (int spin = 0; spin < 20000; ++spin) {
                    CHECK(hipMemcpyAsync(st.data(), d_stat, st.size() * 8, hipMemcpyDeviceToHost, s2));
                    CHECK(hipStreamSynchronize(s2));
                    int done = 0;
                    for (int i = 0; i < n_wg; ++i) done += st[(size_t)i * 8] != 0;
                    if (done == n_wg) break;
                }
                uint32_t one = 1;
                CHECK(hipMemcpyAsync(d_stop, &one, 4, hipMemcpyHostToDevice, s2));
                CHECK(hipStreamSynchronize(s2));
            }
            CHECK(hipDeviceSynchronize());
            CHECK(hipStreamDestroy(s2));
            std::vector<unsigned long long> st((size_t)n_wg * 8);
            CHECK(hipMemcpy(st.data(), d_stat, st.size() * 8, hipMemcpyDeviceToHost));
            double tot = 0;
            for (int i = 0; i < n_wg; ++i) tot += (double)st[(size_t)i * 8];
            fflush(stdout);
            if (n_wg == 1 && !bg) host_check(16, "classic check", 1, 1);
            printf("%-8s %2s %2s %3s %6d %5s %3d | %9.2f |\n", variant == 3 ? "chained" : variant == 2 ? "classic3" : variant ? "classic2" : "classic", "-", "-", "-", n_wg, "-", bg, tot * 0.01 / ((double)n_wg * n_batches));
        }
    return 0;
}
