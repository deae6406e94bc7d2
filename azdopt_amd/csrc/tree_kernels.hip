// tree_kernels.hip -- c21 translation unit of the data-parallel tree-search step on gfx950:
// tree_core.inc (space-independent search) instantiated with the C21Space policy (space_c21.inc),
// the CU-resident persistent step, the device root policy and the launchers.
// (async_kernels.hip re-includes this file with AZD_TU_ASYNC for the asynchronous step;
// ramsey_kernels.hip is the same core with the Ramsey policy.)
#include <hip/hip_runtime.h>

#include "bf16.h"
#include "c21_host.h"
#include "engine_types.h"

namespace azd {

#include "tree_core.inc"
#include "space_c21.inc"

#ifndef AZD_TU_ASYNC
__global__ void k_hash_predictions(float *out, int batch, int action_dim, uint64_t seed, uint64_t first_agent,
                                   uint64_t call) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t total = (size_t)batch * action_dim;
    if (i >= total) return;
    uint64_t agent = first_agent + i / action_dim, act = i % action_dim;
    uint64_t r = splitmix(splitmix(splitmix(splitmix(seed ^ 0x70726564ull) ^ agent) ^ call) ^ act);
    out[i] = (float)(r >> 40) * (1.0f / 16777216.0f);
}
#endif // !AZD_TU_ASYNC

// BIG is a spare specialisation flag (n > 19); the kernels no longer depend on it
#define DISPATCH_KW(A, FN, ...)                                   \
    switch ((A).KW) {                                             \
    case 1: FN<C21Space<1>>(__VA_ARGS__); break;                  \
    case 2: FN<C21Space<2>>(__VA_ARGS__); break;                  \
    case 3: FN<C21Space<3>>(__VA_ARGS__); break;                  \
    default: FN<C21Space<4>>(__VA_ARGS__); break;                 \
    }

#include "persistent_step.inc"
#ifdef AZD_TU_POOL
#include "async_step.inc"
#include "pool_step.inc"
template <class SP>
static void l_pool(const Arenas &a, const PersistArgs *d_args, const StepLaunch &sl, const float *params,
                   const void *wpk, int n_blocks, uint32_t dyn_stride, size_t dyn_bytes, hipStream_t st) {
    // the dynamic-LDS attribute is per device: set on every launch (see l_async)
    const int mode = (sl.hashed ? 1 : 0) | (sl.window ? 2 : 0) | ((sl.groups && !sl.hashed) ? 4 : 0);
#define AZD_LAUNCH_POOL(M)                                                                                                          \
    case M:                                                                                                                         \
        if (hipFuncSetAttribute((const void *)k_pool<SP, M>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn_bytes) != hipSuccess) return; \
        k_pool<SP, M><<<dim3(n_blocks), dim3(PERSIST_WAVES * 64), dyn_bytes, st>>>(d_args, sl.n_calls, sl.log_key, dyn_stride, params, a.state_vecs, a.h_theta, wpk); \
        break;
    switch (mode) {
        AZD_LAUNCH_POOL(0)
        AZD_LAUNCH_POOL(1)
        AZD_LAUNCH_POOL(2)
        AZD_LAUNCH_POOL(3)
        AZD_LAUNCH_POOL(4) // evaluator groups (pool_eval_group) ...
        AZD_LAUNCH_POOL(6) // ... and inside a run-ahead window
    }
#undef AZD_LAUNCH_POOL
    k_argmin_log1<SP><<<dim3(1), dim3(64), SP::dyn_bytes(a), st>>>(a, sl.n_calls, sl.log_key, sl.ctl);
}
void launch_pool(const Arenas &a, const PersistArgs *d_args, const StepLaunch &sl, const float *params,
                 const void *wpk, int n_blocks, uint32_t dyn_stride, size_t dyn_bytes, void *stream) {
    if (a.space == SPACE_RAMSEY) return ramsey_launch_pool(a, d_args, sl, params, wpk, n_blocks, dyn_stride, dyn_bytes, stream);
    DISPATCH_KW(a, l_pool, a, d_args, sl, params, wpk, n_blocks, dyn_stride, dyn_bytes, (hipStream_t)stream);
}
// Test entry (azd_engine_debug_tile_forward): the forward of the IN-KERNEL evaluator -- pool_eval's staging and mlp_tile_task's
// sums, 16 rows per workgroup -- for rows the host hands over.  A prediction row does not depend on the batch it travels in
// (every output element is its own chain of sums), so these are the rows k_pool's evaluator workgroups hand their agents: the
// oracle is fed with them, call by call, to check a whole launch of the PRODUCT kernel k_pool<SP, 0> with the real model
// (tests/test_gpu_pool.py).  f32 or bf16 storage, as the engine's evaluator has it.
__global__ __launch_bounds__(PERSIST_WAVES * 64) void k_tile_forward(const FusedEval ev, const float *__restrict__ params, const void *__restrict__ wpk,
                                                                     const uint32_t stride, const uint32_t out_off, const int n_rows,
                                                                     const float *__restrict__ states, float *__restrict__ out) {
    __shared__ uint32_t agents[PERSIST_WAVES];
    const int tid = threadIdx.x, wave = tid >> 6, first = blockIdx.x * PERSIST_WAVES;
    const int n = n_rows - first < PERSIST_WAVES ? n_rows - first : PERSIST_WAVES;
    const int S = ev.dims[0], S16 = (S + 15) & ~15, L = ev.n_layers, A = ev.dims[L];
    if (tid < PERSIST_WAVES) agents[tid] = (uint32_t)(first + tid);
    PoolRows rows;
    rows.stride = stride;
    rows.out_off = out_off;
    rows.agents = agents;
    rows.n = n;
    EvalPtrs gp;
    gp.params = params;
    gp.wpk = wpk;
    gp.state_vecs = states;
    gp.h_theta = out;
    for (int idx = tid; idx < n * S16; idx += PERSIST_WAVES * 64) {
        const int r = idx / S16, c = idx - r * S16;
        const float v = c < S ? states[(size_t)(first + r) * S + c] : 0.f;
        if (ev.bf16) reinterpret_cast<uint16_t *>(rows.row(r))[c] = (uint16_t)bf16_bits(v);
        else rows.row(r)[c] = v;
    }
    __syncthreads();
    for (int l = 0; l < L; ++l) {
        const int nt = (ev.dims[l + 1] + 15) >> 4;
        for (int tile = wave; tile < nt; tile += PERSIST_WAVES) {
            unsigned long long ph[3];
            mlp_tile_task<PoolRows, true>(ev, rows, gp, l, tile, ph);
        }
        __syncthreads();
    }
    for (int idx = tid; idx < n * A; idx += PERSIST_WAVES * 64) {
        const int r = idx / A, c = idx - r * A;
        out[(size_t)(first + r) * A + c] = rows.out(r)[c];
    }
}
hipError_t launch_tile_forward(const FusedEval &ev, const PoolArgs &pool, int n_rows, const float *states, float *out, void *stream) {
    const size_t dyn_bytes = (size_t)pool.eval_stride * sizeof(float) * PERSIST_WAVES;
    hipError_t he = hipFuncSetAttribute((const void *)k_tile_forward, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn_bytes);
    if (he != hipSuccess) return he;
    k_tile_forward<<<dim3((n_rows + PERSIST_WAVES - 1) / PERSIST_WAVES), dim3(PERSIST_WAVES * 64), dyn_bytes, (hipStream_t)stream>>>(
        ev, ev.params, ev.wpk, pool.eval_stride, pool.eval_out_off, n_rows, states, out);
    return hipGetLastError(); // (a launch that was refused -- LDS, grid -- must not leave the caller copying an unwritten buffer back)
}
template <class SP>
static void q_pool_resident(int *out, size_t dyn_bytes) {
    int nb = 0;
    if (hipFuncSetAttribute((const void *)k_pool<SP, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn_bytes) != hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void *)k_pool<SP, 0>, PERSIST_WAVES * 64, dyn_bytes) != hipSuccess) {
        (void)hipGetLastError();
        nb = 0;
    }
    *out = nb;
}
int pool_max_resident(const Arenas &a, size_t dyn_bytes, int n_cus) {
    if (a.space == SPACE_RAMSEY) return ramsey_pool_max_resident(a, dyn_bytes, n_cus);
    int nb = 0;
    DISPATCH_KW(a, q_pool_resident, &nb, dyn_bytes);
    return nb * n_cus;
}
// After an aborted pool launch (PoolCtl::abort: a wait ran into its bound): where every agent stands, for the asynchronous
// step that takes over (StepLaunch::resume).  A wave never leaves an agent inside a call, so an agent is in one of three
// states: never taken (index >= claimed: no call made), waiting for the prediction row of its last call's new node (the node
// it stands on has no actions yet; PendRec::call = calls completed), or through all its calls.
__global__ void k_pool_resume_scan(Arenas a, const PendRec *__restrict__ pend, const uint32_t *__restrict__ claim_next, const int n_calls,
                                   uint32_t *__restrict__ resume) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= a.B) return;
    const uint32_t claimed = *claim_next; // claims handed out (it runs past B once every agent is taken)
    uint32_t r = 0u;
    if ((uint32_t)t < claimed) {
        const NodeRec nd = a.nodes[(size_t)t * a.node_cap + a.state_pos[t]];
        const bool pending = a.flags[t] == 0u && nd.act_end == 0u;
        r = pending ? (pend[t].call | 0x80000000u) : (uint32_t)n_calls;
    }
    resume[t] = r;
}
void launch_pool_resume_scan(const Arenas &a, const PoolArgs &pool, int n_calls, uint32_t *resume, void *stream) {
    k_pool_resume_scan<<<dim3((a.B + 255) / 256), dim3(256), 0, (hipStream_t)stream>>>(a, pool.pend, &pool.ctl->claim_next, n_calls, resume);
}
__global__ void k_probe_xcc(uint32_t *out) {
    if (threadIdx.x == 0) out[blockIdx.x] = pool_xcc_id();
}
void launch_probe_xcc(uint32_t *d_out, int n_blocks, void *stream) {
    k_probe_xcc<<<dim3(n_blocks), dim3(64), 0, (hipStream_t)stream>>>(d_out);
}
// LDS plan of the pool step: a searcher wave's scratch (with room to build its state-vector row) or an evaluator's
// batch of 16 rows [x][h0][h1][out], whichever is larger
bool pool_plan(const Arenas &a, const FusedEval &ev, PoolArgs *pool, uint32_t *dyn_stride, size_t *dyn_bytes, const char **why) {
    const char *dummy;
    if (!why) why = &dummy;
    if (a.space == SPACE_DENSE) {
        *why = "dense-graph space: its CU-resident form is the pool searchers with the evaluator outside the kernel (engine.hip: dense_pool_run)";
        return false;
    }
    if (a.space == SPACE_RAMSEY) return ramsey_pool_plan(a, ev, pool, dyn_stride, dyn_bytes, why);
    return pool_plan_common(a, ev, pool, dyn_stride, dyn_bytes, why, C21Space<1>::pool_dyn_bytes(a), sizeof(WaveLds));
}
#elif defined(AZD_TU_ASYNC)
#include "async_step.inc"
template <class SP>
static void l_async(const Arenas &a, const PersistArgs *d_args, const StepLaunch &sl,
                    const float *params, const void *wpk, uint32_t dyn_stride, size_t dyn_bytes, hipStream_t st) {
    // dynamic LDS beyond the default 64 KB needs the attribute, which is per DEVICE (the current one): set on every
    // launch -- a host-side call, once per <= 1024 search calls -- so that engines on several devices in one process
    // all get it; the plans have already checked that the request fits beside the kernel's static LDS
    if (hipFuncSetAttribute((const void *)k_async<SP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn_bytes) != hipSuccess) return; // sticky: the caller's hipGetLastError reports it
    const int n_wg = (a.B + PERSIST_WAVES - 1) / PERSIST_WAVES;
    k_async<SP><<<dim3(n_wg), dim3(PERSIST_WAVES * 64), dyn_bytes, st>>>(d_args, sl.n_calls, sl.log_key, dyn_stride, params, a.state_vecs, a.h_theta, wpk, sl.resume);
    k_argmin_log1<SP><<<dim3(1), dim3(64), SP::dyn_bytes(a), st>>>(a, sl.n_calls, sl.log_key, nullptr);
}
void launch_async(const Arenas &a, const PersistArgs *d_args, const StepLaunch &sl,
                  const float *params, const void *wpk, uint32_t dyn_stride, size_t dyn_bytes, void *stream) {
    if (a.space == SPACE_RAMSEY) return ramsey_launch_async(a, d_args, sl, params, wpk, dyn_stride, dyn_bytes, stream);
    DISPATCH_KW(a, l_async, a, d_args, sl, params, wpk, dyn_stride, dyn_bytes, (hipStream_t)stream);
}
// LDS plan of the asynchronous step (no evaluator buffers in LDS)
bool async_plan(const Arenas &a, const FusedEval &ev, uint32_t *dyn_stride, size_t *dyn_bytes, const char **why) {
    const char *dummy;
    if (!why) why = &dummy;
    if (a.space == SPACE_DENSE) {
        *why = "dense-graph space: its CU-resident form is the pool searchers with the evaluator outside the kernel (engine.hip: dense_pool_run)";
        return false;
    }
    if (a.space == SPACE_RAMSEY) return ramsey_async_plan(a, ev, dyn_stride, dyn_bytes, why);
    if (ev.kind == 3)
        for (int l = 0; l < ev.n_layers; ++l)
            if (ev.dims[l] % (l == 0 ? 4 : 16) != 0) { // tile tasks walk K in steps of 16; x is zero-padded
                *why = "asynchronous step: hidden widths must be multiples of 16 and the input width a multiple of 4";
                return false;
            }
    if (a.B > 65536 || a.node_cap > 65536) { // (agent, node) are packed 16 + 16 bits in the argmin log
        *why = "asynchronous step: more than 65536 agents or nodes per tree";
        return false;
    }
    size_t stride = (dyn_lds_bytes(a.n) + 15) & ~(size_t)15;
    if (ev.kind == 3) { // a waiting agent's region holds its row's activations: [x][h0][h1] + the 16-B-per-wave skew
        size_t rows = ((size_t)((ev.dims[0] + 15) & ~15) + (size_t)ev.hid[0] + (size_t)ev.hid[1]) * sizeof(float) + 16 * PERSIST_WAVES;
        if (rows > stride) stride = (rows + 15) & ~(size_t)15;
    }
    size_t total = stride * PERSIST_WAVES;
    const size_t static_lds = PERSIST_WAVES * (sizeof(WaveLds) + 16) + sizeof(AsyncCtl) + 256;
    if (total + static_lds > 160 * 1024) {
        *why = "asynchronous step: 16 rows of activations do not fit the CU's 160 KB of LDS";
        return false;
    }
    *dyn_stride = (uint32_t)stride;
    *dyn_bytes = total;
    return true;
}
#else
#include "root_policy.inc"

// parity probe for the f32 primitives the selection rule depends on; four outputs per input pair:
//   [0] the kernel's own sqrt(|x - y|) (azd_sqrt)   [1] sqrtf   [2] __fsqrt_rn   [3] x - (x - y)
__global__ void k_probe_math(const float *in, float *out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float x = in[2 * i], y = in[2 * i + 1];
    float d = fabsf(x - y);
    out[4 * i] = azd_sqrt(d);
    float g = x - y;
    out[4 * i + 3] = x - g;
}

// the alternatives, in a kernel of their own so that nothing is shared with azd_sqrt above
__global__ void k_probe_sqrt_alt(const float *in, float *out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float d = fabsf(in[2 * i] - in[2 * i + 1]);
    out[4 * i + 1] = sqrtf(d);
    out[4 * i + 2] = (float)sqrt((double)d);
}

// lambda_1 / matching probe: one wave per tree, `reps` repetitions (timing), result of the last one
__global__ __launch_bounds__(64) void k_probe_cost(const uint8_t *__restrict__ parents, int n, int count, int reps,
                                                   int full, double *__restrict__ lam_out, int *__restrict__ mu_out, double lo0, double hi0) {
    __shared__ WaveLds s;
    const uint32_t dyn = 0;
    const int t = blockIdx.x;
    if (t >= count) return;
    if (LANE < PARENTS_STRIDE) s.par[LANE] = LANE < n ? parents[(size_t)t * n + LANE] : 0;
    WAVE_SYNC();
    double lam = 0.0;
    int mu = 0;
    for (int r = 0; r < reps; ++r) {
        const PackedTree pt = pack_tree(s, n);
        lam = full ? lambda1_wave<true>(pt, n, dyn, lo0, hi0) : lambda1_wave<false>(pt, n, dyn, lo0, hi0);
        mu = full ? matching_wave(s, n, nullptr) : matching_size_wave(pt, n);
        WAVE_SYNC();
    }
    if (LANE == 0) {
        lam_out[t] = lam;
        mu_out[t] = mu;
    }
}

// ---------------------------------------------------------------- launchers
template <class SP>
static void l_init_roots(const Arenas &a, const uint8_t *p, const uint64_t *m, hipStream_t st) {
    k_init_roots<SP><<<dim3(a.B), dim3(64), SP::dyn_bytes(a), st>>>(a, p, m);
}
template <class SP>
static void l_add_actions(const Arenas &a, int root_mode, hipStream_t st) {
    k_add_actions<SP><<<dim3(a.tn ? a.tn : a.B), dim3(64), SP::dyn_bytes(a), st>>>(a, root_mode);
}
template <class SP>
static void l_rollout(const Arenas &a, const TolTable &tol, hipStream_t st) {
    k_rollout<SP><<<dim3(a.tn ? a.tn : a.B), dim3(64), SP::dyn_bytes(a), st>>>(a, tol);
}
template <class SP>
static void l_argmin(const Arenas &a, int init_mode, hipStream_t st) {
    k_argmin<SP><<<dim3(1), dim3(1024), SP::dyn_bytes(a), st>>>(a, init_mode);
}
template <class SP>
static void l_observe(const Arenas &a, uint32_t tol, hipStream_t st) {
    k_observe<SP><<<dim3(a.B), dim3(64), SP::dyn_bytes(a), st>>>(a, tol);
}

void launch_init_roots(const Arenas &a, const uint8_t *d_parents, const uint64_t *d_permitted, void *stream) {
    if (a.space == SPACE_RAMSEY) return ramsey_launch_init_roots(a, d_parents, d_permitted, stream);
    if (a.space == SPACE_DENSE) return dense_launch_init_roots(a, d_parents, d_permitted, stream);
    DISPATCH_KW(a, l_init_roots, a, d_parents, d_permitted, (hipStream_t)stream);
}
void launch_add_actions(const Arenas &a, int root_mode, void *stream) {
    if (a.space == SPACE_RAMSEY) return ramsey_launch_add_actions(a, root_mode, stream);
    if (a.space == SPACE_DENSE) return dense_launch_add_actions(a, root_mode, stream);
    DISPATCH_KW(a, l_add_actions, a, root_mode, (hipStream_t)stream);
}
void launch_rollout(const Arenas &a, const TolTable &tol, void *stream) {
    if (a.space == SPACE_RAMSEY) return ramsey_launch_rollout(a, tol, stream);
    if (a.space == SPACE_DENSE) return dense_launch_rollout(a, tol, stream);
    DISPATCH_KW(a, l_rollout, a, tol, (hipStream_t)stream);
}
void launch_argmin(const Arenas &a, int init_mode, void *stream) {
    if (a.space == SPACE_RAMSEY) return ramsey_launch_argmin(a, init_mode, stream);
    if (a.space == SPACE_DENSE) return dense_launch_argmin(a, init_mode, stream);
    DISPATCH_KW(a, l_argmin, a, init_mode, (hipStream_t)stream);
}
template <class SP>
static void l_argmin_log(const Arenas &a, int n_calls, unsigned long long *log_key, hipStream_t st) {
    k_argmin_log1<SP><<<dim3(1), dim3(64), SP::dyn_bytes(a), st>>>(a, n_calls, log_key, nullptr);
}
template <class SP>
static void l_argmin_one(const Arenas &a, int agent, uint32_t node, hipStream_t st) {
    k_argmin_one<SP><<<dim3(1), dim3(64), SP::dyn_bytes(a), st>>>(a, agent, node);
}
void launch_argmin_one(const Arenas &a, int agent, uint32_t node, void *stream) {
    if (a.space == SPACE_RAMSEY) return ramsey_launch_argmin_one(a, agent, node, stream);
    if (a.space == SPACE_DENSE) return; // (the dense space has no CU-resident form, hence no window)
    DISPATCH_KW(a, l_argmin_one, a, agent, node, (hipStream_t)stream);
}
void launch_argmin_log(const Arenas &a, int n_calls, unsigned long long *log_key, void *stream) {
    if (a.space == SPACE_RAMSEY) return ramsey_launch_argmin_log(a, n_calls, log_key, stream);
    if (a.space == SPACE_DENSE) return dense_launch_argmin_log(a, n_calls, log_key, stream);
    DISPATCH_KW(a, l_argmin_log, a, n_calls, log_key, (hipStream_t)stream);
}
void launch_log_candidates(const Arenas &a, unsigned long long *log_key, uint32_t *call_ctr, void *stream) {
    k_log_candidates<<<dim3(1), dim3(1024), 0, (hipStream_t)stream>>>(a, log_key, call_ctr);
}
void launch_observe(const Arenas &a, uint32_t n_obs_tol, void *stream) {
    if (a.space == SPACE_RAMSEY) return ramsey_launch_observe(a, n_obs_tol, stream);
    if (a.space == SPACE_DENSE) return dense_launch_observe(a, n_obs_tol, stream);
    DISPATCH_KW(a, l_observe, a, n_obs_tol, (hipStream_t)stream);
}
template <class SP>
static void l_persist(const Arenas &a, const PersistArgs *d_args, const StepLaunch &sl,
                      uint32_t *log_node, uint32_t dyn_stride, size_t dyn_bytes, hipStream_t st) {
    // dynamic LDS beyond the default 64 KB needs the attribute, which is per DEVICE (the current one): set on every
    // launch -- a host-side call, once per <= 1024 search calls -- so that engines on several devices in one process
    // all get it; the plans have already checked that the request fits beside the kernel's static LDS
    if (hipFuncSetAttribute((const void *)k_persist<SP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn_bytes) != hipSuccess) return; // sticky: the caller's hipGetLastError reports it
    const int n_wg = (a.B + PERSIST_WAVES - 1) / PERSIST_WAVES;
    k_persist<SP><<<dim3(n_wg), dim3(PERSIST_WAVES * 64), dyn_bytes, st>>>(d_args, sl.n_calls, sl.log_key, log_node, dyn_stride);
    k_argmin_log<SP><<<dim3(1), dim3(64), SP::dyn_bytes(a), st>>>(a, sl.n_calls, n_wg, sl.log_key, log_node);
}
// LDS plan of the persistent step; returns false when the workgroup does not fit a CU
bool persist_plan(const Arenas &a, const FusedEval &ev, uint32_t *dyn_stride, size_t *dyn_bytes, const char **why) {
    const char *dummy;
    if (!why) why = &dummy;
    size_t stride = (dyn_lds_bytes(a.n) + 15) & ~(size_t)15;
    size_t total = stride * PERSIST_WAVES;
    if (a.space == SPACE_DENSE) {
        *why = "dense-graph space: its CU-resident form is the pool searchers with the evaluator outside the kernel (engine.hip: dense_pool_run)";
        return false;
    }
    if (a.space == SPACE_RAMSEY) return ramsey_persist_plan(a, ev, dyn_stride, dyn_bytes, why);
    if (ev.kind == 3) {
        if (ev.bf16) { // bf16 weight storage is built into the asynchronous step only
            *why = "barrier step: bf16 weight storage is not built into it";
            return false;
        }
        for (int l = 0; l < ev.n_layers; ++l)
            if (ev.dims[l] % 4 != 0) { // the in-kernel MLP loads rows as float4
                *why = "barrier step: layer widths must be multiples of 4";
                return false;
            }
        size_t mlp = (size_t)PERSIST_WAVES * ((size_t)(ev.dims[0] + 4) + (size_t)(ev.hid[0] + 4) + (size_t)(ev.hid[1] + 4)) * sizeof(float);
        if (mlp > total) total = mlp;
    }
    const size_t static_lds = PERSIST_WAVES * (sizeof(WaveLds) + 16) + 256;
    if (total + static_lds > 160 * 1024) {
        *why = "barrier step: 16 rows of activations do not fit the CU's 160 KB of LDS";
        return false;
    }
    *dyn_stride = (uint32_t)stride;
    *dyn_bytes = total;
    return true;
}
void launch_persist(const Arenas &a, const PersistArgs *d_args, const StepLaunch &sl,
                    uint32_t *log_node, uint32_t dyn_stride, size_t dyn_bytes, void *stream) {
    if (a.space == SPACE_RAMSEY) return ramsey_launch_persist(a, d_args, sl, log_node, dyn_stride, dyn_bytes, stream);
    DISPATCH_KW(a, l_persist, a, d_args, sl, log_node, dyn_stride, dyn_bytes, (hipStream_t)stream);
}
template <class SP>
static void l_modify_roots(const Arenas &a, uint64_t seed, uint64_t epoch, uint64_t first_agent, int kmin, int kmax,
                           uint8_t *d_parents, uint64_t *d_perm, hipStream_t st) {
    k_modify_roots<SP><<<dim3(a.B), dim3(64), SP::dyn_bytes(a), st>>>(a, seed, epoch, first_agent, kmin, kmax, d_parents, d_perm, d_perm);
}
void launch_c21_modify_roots(const Arenas &a, uint64_t seed, uint64_t epoch, uint64_t first_agent, int kmin, int kmax,
                             uint8_t *d_parents, uint64_t *d_perm, void *stream) {
    if (a.space == SPACE_RAMSEY) return ramsey_launch_modify_roots(a, seed, epoch, first_agent, kmin, kmax, d_parents, d_perm, stream);
    DISPATCH_KW(a, l_modify_roots, a, seed, epoch, first_agent, kmin, kmax, d_parents, d_perm, (hipStream_t)stream);
}
void launch_hash_predictions(float *d_out, int batch, int action_dim, uint64_t seed, uint64_t first_agent,
                             uint64_t call, void *stream) {
    size_t total = (size_t)batch * action_dim;
    int threads = 256;
    int blocks = (int)((total + threads - 1) / threads);
    hipLaunchKernelGGL(k_hash_predictions, dim3(blocks), dim3(threads), 0, (hipStream_t)stream, d_out, batch, action_dim,
                       seed, first_agent, call);
}
void launch_probe_cost(const uint8_t *d_parents, int n, int count, int reps, int full, double *d_lam, int *d_mu,
                       void *stream) {
    double lo0, hi0;
    c21_lambda_bracket(n, &lo0, &hi0);
    k_probe_cost<<<dim3(count), dim3(64), dyn_lds_bytes(n), (hipStream_t)stream>>>(d_parents, n, count, reps, full, d_lam, d_mu, lo0, hi0);
}
void launch_probe_math(const float *d_in, float *d_out, int n, void *stream) {
    hipLaunchKernelGGL(k_probe_math, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, d_in, d_out, n);
    hipLaunchKernelGGL(k_probe_sqrt_alt, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, d_in, d_out, n);
}

#endif // AZD_TU_ASYNC

} // namespace azd
