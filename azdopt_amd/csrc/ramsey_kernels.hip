// ramsey_kernels.hip -- Ramsey translation unit of the data-parallel tree-search step on gfx950:
// tree_core.inc instantiated with the RamseySpace policy (space_ramsey.inc), the CU-resident
// persistent step and the launchers the c21 entry points forward to for SPACE_RAMSEY.
// Built with -ffp-contract=off like the c21 unit.
#include <hip/hip_runtime.h>

#include "engine_types.h"

namespace azd {

#include "tree_core.inc"
#include "space_ramsey.inc"

#define DISPATCH_RKW(A, FN, ...)                                  \
    switch ((A).KW) {                                             \
    case 1: FN<RamseySpace<1>>(__VA_ARGS__); break;               \
    case 2: FN<RamseySpace<2>>(__VA_ARGS__); break;               \
    case 3: FN<RamseySpace<3>>(__VA_ARGS__); break;               \
    case 4: FN<RamseySpace<4>>(__VA_ARGS__); break;               \
    case 5: FN<RamseySpace<5>>(__VA_ARGS__); break;               \
    default: FN<RamseySpace<6>>(__VA_ARGS__); break;              \
    }

#include "persistent_step.inc"
#include "root_policy.inc"

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

template <class SP>
static void l_modify_roots(const Arenas &a, uint64_t seed, uint64_t epoch, uint64_t first_agent, int kmin, int kmax,
                           uint8_t *d_colors, uint64_t *d_perm, hipStream_t st) {
    k_modify_roots<SP><<<dim3(a.B), dim3(64), SP::dyn_bytes(a), st>>>(a, seed, epoch, first_agent, kmin, kmax, d_colors, d_perm, d_perm);
}
void ramsey_launch_modify_roots(const Arenas &a, uint64_t seed, uint64_t epoch, uint64_t first_agent, int kmin, int kmax,
                                uint8_t *d_colors, uint64_t *d_perm, void *stream) {
    DISPATCH_RKW(a, l_modify_roots, a, seed, epoch, first_agent, kmin, kmax, d_colors, d_perm, (hipStream_t)stream);
}
void ramsey_launch_init_roots(const Arenas &a, const uint8_t *d_colors, const uint64_t *d_permitted, void *stream) {
    DISPATCH_RKW(a, l_init_roots, a, d_colors, d_permitted, (hipStream_t)stream);
}
void ramsey_launch_add_actions(const Arenas &a, int root_mode, void *stream) {
    DISPATCH_RKW(a, l_add_actions, a, root_mode, (hipStream_t)stream);
}
void ramsey_launch_rollout(const Arenas &a, const TolTable &tol, void *stream) {
    DISPATCH_RKW(a, l_rollout, a, tol, (hipStream_t)stream);
}
template <class SP>
static void l_argmin_log(const Arenas &a, int n_calls, unsigned long long *log_key, hipStream_t st) {
    k_argmin_log1<SP><<<dim3(1), dim3(64), SP::dyn_bytes(a), st>>>(a, n_calls, log_key, nullptr);
}
template <class SP>
static void l_argmin_one(const Arenas &a, int agent, uint32_t node, hipStream_t st) {
    k_argmin_one<SP><<<dim3(1), dim3(64), SP::dyn_bytes(a), st>>>(a, agent, node);
}
void ramsey_launch_argmin_one(const Arenas &a, int agent, uint32_t node, void *stream) {
    DISPATCH_RKW(a, l_argmin_one, a, agent, node, (hipStream_t)stream);
}
void ramsey_launch_argmin_log(const Arenas &a, int n_calls, unsigned long long *log_key, void *stream) {
    DISPATCH_RKW(a, l_argmin_log, a, n_calls, log_key, (hipStream_t)stream);
}
void ramsey_launch_argmin(const Arenas &a, int init_mode, void *stream) {
    DISPATCH_RKW(a, l_argmin, a, init_mode, (hipStream_t)stream);
}
void ramsey_launch_observe(const Arenas &a, uint32_t n_obs_tol, void *stream) {
    DISPATCH_RKW(a, l_observe, a, n_obs_tol, (hipStream_t)stream);
}
// LDS plan of the persistent step; false when the workgroup does not fit a CU or the in-kernel MLP
// cannot take the layer widths (it loads rows as float4)
bool ramsey_persist_plan(const Arenas &a, const FusedEval &ev, uint32_t *dyn_stride, size_t *dyn_bytes, const char **why) {
    const char *dummy;
    if (!why) why = &dummy;
    size_t per = CORE_DYN_BYTES + (size_t)a.C * a.E * sizeof(int32_t);
    size_t stride = (per + 15) & ~(size_t)15;
    size_t total = stride * PERSIST_WAVES;
    if (ev.kind == 3) {
        if (ev.bf16) { // bf16 weight storage is built into the asynchronous step only
            *why = "barrier step: bf16 weight storage is not built into it";
            return false;
        }
        for (int l = 0; l < ev.n_layers; ++l)
            if (ev.dims[l] % 4 != 0) {
                *why = "barrier step: layer widths must be multiples of 4";
                return false;
            }
        size_t mlp = (size_t)PERSIST_WAVES * ((size_t)(ev.dims[0] + 4) + (size_t)(ev.hid[0] + 4) + (size_t)(ev.hid[1] + 4)) * sizeof(float);
        if (mlp > total) total = mlp;
    }
    const size_t static_lds = PERSIST_WAVES * (sizeof(RamseyLds) + 16) + 256;
    if (total + static_lds > 160 * 1024) {
        *why = "barrier step: 16 rows of activations do not fit the CU's 160 KB of LDS";
        return false;
    }
    *dyn_stride = (uint32_t)stride;
    *dyn_bytes = total;
    return true;
}
void ramsey_launch_persist(const Arenas &a, const PersistArgs *d_args, const StepLaunch &sl,
                           uint32_t *log_node, uint32_t dyn_stride, size_t dyn_bytes, void *stream) {
    DISPATCH_RKW(a, l_persist, a, d_args, sl, log_node, dyn_stride, dyn_bytes, (hipStream_t)stream);
}

} // namespace azd
