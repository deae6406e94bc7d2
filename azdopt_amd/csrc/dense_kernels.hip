// dense_kernels.hip -- dense-graph translation unit of the data-parallel tree-search step on gfx950: tree_core.inc
// instantiated with the DenseSpace policy (space_dense.inc) for key widths 2 / 4 / 10 / 16: launch-per-phase kernels, the
// device root policy, and the pool step's searchers (pool_step.inc: k_pool_search) with the evaluator OUTSIDE the kernel --
// the space's model (3676-512-512-512-2450 at N = 50) does not fit an evaluator workgroup's LDS, its rows are served by
// batched GEMM launches over what the searchers have posted (k_ext_take / k_ext_deliver).
// Built with -ffp-contract=off like the other tree units.
#include <hip/hip_runtime.h>

#include "bf16.h"
#include "engine_types.h"

namespace azd {

#include "tree_core.inc"
#include "space_dense.inc"

#include "root_policy.inc"

#define AZD_TU_POOL_EXT 1
#include "persistent_step.inc"
#include "async_step.inc"
#include "pool_step.inc"

// the key width (words of a rank set) follows the engine's max_slots: 2, 4, 10 or 16 (engine.hip)
#define DISPATCH_DKW(A, FN, ...)                                  \
    switch ((A).KW) {                                             \
    case 2: FN<DenseSpace<2>>(__VA_ARGS__); break;                \
    case 4: FN<DenseSpace<4>>(__VA_ARGS__); break;                \
    case 10: FN<DenseSpace<10>>(__VA_ARGS__); break;              \
    default: FN<DenseSpace<16>>(__VA_ARGS__); break;              \
    }

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
template <class SP>
static void l_modify_roots(const Arenas &a, uint64_t seed, uint64_t epoch, uint64_t first_agent, int kmin, int kmax, uint8_t *d_adj,
                           uint64_t *d_packed, uint64_t *d_slots, hipStream_t st) {
    k_modify_roots<SP><<<dim3(a.B), dim3(64), SP::dyn_bytes(a), st>>>(a, seed, epoch, first_agent, kmin, kmax, d_adj, d_packed, d_slots);
}

void dense_launch_init_roots(const Arenas &a, const uint8_t *d_adj, const uint64_t *d_packed, void *stream) {
    DISPATCH_DKW(a, l_init_roots, a, d_adj, d_packed, (hipStream_t)stream);
}
void dense_launch_add_actions(const Arenas &a, int root_mode, void *stream) { DISPATCH_DKW(a, l_add_actions, a, root_mode, (hipStream_t)stream); }
void dense_launch_rollout(const Arenas &a, const TolTable &tol, void *stream) { DISPATCH_DKW(a, l_rollout, a, tol, (hipStream_t)stream); }
void dense_launch_argmin(const Arenas &a, int init_mode, void *stream) { DISPATCH_DKW(a, l_argmin, a, init_mode, (hipStream_t)stream); }
template <class SP>
static void l_argmin_log(const Arenas &a, int n_calls, unsigned long long *log_key, hipStream_t st) {
    k_argmin_log1<SP><<<dim3(1), dim3(64), SP::dyn_bytes(a), st>>>(a, n_calls, log_key, nullptr);
}
void dense_launch_argmin_log(const Arenas &a, int n_calls, unsigned long long *log_key, void *stream) {
    DISPATCH_DKW(a, l_argmin_log, a, n_calls, log_key, (hipStream_t)stream);
}
void dense_launch_observe(const Arenas &a, uint32_t n_obs_tol, void *stream) { DISPATCH_DKW(a, l_observe, a, n_obs_tol, (hipStream_t)stream); }
void dense_launch_modify_roots(const Arenas &a, uint64_t seed, uint64_t epoch, uint64_t first_agent, int kmin, int kmax,
                               uint8_t *d_adj, uint64_t *d_packed, uint64_t *d_slots, void *stream) {
    DISPATCH_DKW(a, l_modify_roots, a, seed, epoch, first_agent, kmin, kmax, d_adj, d_packed, d_slots, (hipStream_t)stream);
}


// ---------------------------------------------------------------- pool step, searchers only
// LDS of a searcher workgroup: 16 waves' blocks and scratch regions (no row is staged: SP::write_rows_direct)
template <class SP>
static void q_pool_plan(const Arenas &a, int waves, uint32_t *dyn_stride, size_t *dyn_bytes) {
    const size_t stride = (SP::dyn_bytes(a) + 15) & ~(size_t)15;
    const size_t sw_bytes = (PERSIST_WAVES * sizeof(typename SP::Lds) + 15) & ~(size_t)15; // (k_pool_search's SW_BYTES: the layout of 16 waves' blocks)
    *dyn_stride = (uint32_t)stride;
    *dyn_bytes = sw_bytes + stride * (size_t)waves;
}
// waves: wavefronts per searcher workgroup (1 .. 16): fewer leave registers and LDS on the CU for the evaluator's GEMM blocks
bool dense_pool_plan(const Arenas &a, int waves, uint32_t *dyn_stride, size_t *dyn_bytes, const char **why) {
    if (a.B > 65536 || a.node_cap > 65536) {
        *why = "pool step: more than 65536 agents or nodes per tree";
        return false;
    }
    DISPATCH_DKW(a, q_pool_plan, a, waves, dyn_stride, dyn_bytes);
    if (*dyn_bytes + sizeof(PoolIdle) + 256 > 160 * 1024) {
        *why = "pool step: 16 searcher waves' blocks and scratch do not fit the CU's 160 KB of LDS (more than 640 slots per root)";
        return false;
    }
    return true;
}
template <class SP>
static void l_pool_search(const Arenas &a, const PersistArgs *d_args, const StepLaunch &sl, int n_blocks, int waves, uint32_t dyn_stride,
                          size_t dyn_bytes, hipStream_t st) {
    if (sl.hashed) { // the test harness' evaluator (FusedEval kind 4): the searchers note the call of every row they post
        if (hipFuncSetAttribute((const void *)k_pool_search<SP, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn_bytes) != hipSuccess) return;
        k_pool_search<SP, 1><<<dim3(n_blocks), dim3(waves * 64), dyn_bytes, st>>>(d_args, sl.n_calls, sl.log_key, dyn_stride);
    } else {
        if (hipFuncSetAttribute((const void *)k_pool_search<SP, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn_bytes) != hipSuccess) return;
        k_pool_search<SP, 0><<<dim3(n_blocks), dim3(waves * 64), dyn_bytes, st>>>(d_args, sl.n_calls, sl.log_key, dyn_stride);
    }
    k_argmin_log1<SP><<<dim3(1), dim3(64), SP::dyn_bytes(a), st>>>(a, sl.n_calls, sl.log_key, sl.ctl);
}
void dense_launch_pool_search(const Arenas &a, const PersistArgs *d_args, const StepLaunch &sl, int n_blocks, int waves, uint32_t dyn_stride,
                              size_t dyn_bytes, void *stream) {
    DISPATCH_DKW(a, l_pool_search, a, d_args, sl, n_blocks, waves, dyn_stride, dyn_bytes, (hipStream_t)stream);
}
template <class SP>
static void q_pool_search_resident(int *out, int waves, size_t dyn_bytes) {
    int nb = 0;
    if (hipFuncSetAttribute((const void *)k_pool_search<SP, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn_bytes) != hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void *)k_pool_search<SP, 0>, waves * 64, dyn_bytes) != hipSuccess) {
        (void)hipGetLastError();
        nb = 0;
    }
    *out = nb;
}
int dense_pool_search_resident(const Arenas &a, int waves, size_t dyn_bytes) { // workgroups of k_pool_search one CU holds
    int nb = 0;
    DISPATCH_DKW(a, q_pool_search_resident, &nb, waves, dyn_bytes);
    return nb;
}
void launch_ext_take(const PoolArgs &pool, uint32_t *rows, uint32_t *home, uint32_t *n, unsigned long long *t0, void *stream) {
    k_ext_take<<<dim3(1), dim3(POOL_XCDS * 64), 0, (hipStream_t)stream>>>(pool, rows, home, n, t0);
}
// ---------------------------------------------------------------- recovery of an aborted pool launch by the launch-per-phase kernels
// resume[t] = calls agent t has completed | 1u << 31 if its last call ended on a node whose prediction row is still due (k_pool_resume_scan)
__global__ void k_park(Arenas a, const uint32_t *__restrict__ resume, const int n_calls, const int round, const int park) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= a.B) return;
    uint32_t f = a.flags[t] & ~(uint32_t)FLAG_PARKED;
    if (park) {
        const uint32_t r = resume[t];
        const bool sits_out = round < 0 ? (r >> 31) == 0u : (int)(r & 0x7FFFFFFFu) + round >= n_calls;
        if (sits_out) f |= (uint32_t)FLAG_PARKED;
    }
    a.flags[t] = f;
}
void launch_park(const Arenas &a, const uint32_t *resume, int n_calls, int round, int park, void *stream) {
    k_park<<<dim3((a.B + 255) / 256), dim3(256), 0, (hipStream_t)stream>>>(a, resume, n_calls, round, park);
}
// the candidates of the agents that took part in round r, each under the call it really was (optimizer/mod.rs:194-246 up to the choice among trees)
__global__ void k_log_candidates_resume(Arenas a, unsigned long long *__restrict__ log_key, const uint32_t *__restrict__ resume, const int n_calls,
                                        const int round) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= a.B) return;
    const uint32_t node = a.cand_node[t];
    const int call = (int)(resume[t] & 0x7FFFFFFFu) + round;
    if (node != NONE && a.flags[t] == 0 && call < n_calls) {
        const unsigned long long key = ((unsigned long long)ordf(a.cand_c[t]) << 32) | ((unsigned long long)((uint32_t)t & 0xFFFFu) << 16) |
                                       (unsigned long long)(node & 0xFFFFu);
        atomicMin(&log_key[call], key);
        a.cand_node[t] = NONE; // num_inspected_nodes = nodes.len()
    }
}
void launch_log_candidates_resume(const Arenas &a, unsigned long long *log_key, const uint32_t *resume, int n_calls, int round, void *stream) {
    k_log_candidates_resume<<<dim3((a.B + 255) / 256), dim3(256), 0, (hipStream_t)stream>>>(a, log_key, resume, n_calls, round);
}
void launch_ext_hash_rows(const PersistArgs *d_args, const uint32_t *rows, const uint32_t *n, uint32_t cap, float *h_theta, void *stream) {
    k_ext_hash_rows<<<dim3(cap), dim3(256), 0, (hipStream_t)stream>>>(d_args, rows, n, h_theta);
}
void launch_ext_deliver(const PoolArgs &pool, const Arenas &a, const uint32_t *rows, const uint32_t *home, const uint32_t *n, uint32_t cap,
                        const unsigned long long *t0, void *stream) {
    k_ext_deliver<<<dim3((cap + 255) / 256), dim3(256), 0, (hipStream_t)stream>>>(pool, a, rows, home, n, t0);
}

} // namespace azd
