// ramsey_async_kernels.hip -- the asynchronous CU-resident step (async_step.inc) for the Ramsey space, in
// its own translation unit like the c21 one (async_kernels.hip): co-compiled with k_persist the shared
// device functions are register-allocated worse.
#define AZD_TU_ASYNC 1
#include <hip/hip_runtime.h>

#include "bf16.h"
#include "engine_types.h"

namespace azd {

#include "tree_core.inc"
#include "space_ramsey.inc"
#include "persistent_step.inc"
#include "async_step.inc"

#define DISPATCH_RKW(A, FN, ...)                                  \
    switch ((A).KW) {                                             \
    case 1: FN<RamseySpace<1>>(__VA_ARGS__); break;               \
    case 2: FN<RamseySpace<2>>(__VA_ARGS__); break;               \
    case 3: FN<RamseySpace<3>>(__VA_ARGS__); break;               \
    case 4: FN<RamseySpace<4>>(__VA_ARGS__); break;               \
    case 5: FN<RamseySpace<5>>(__VA_ARGS__); break;               \
    default: FN<RamseySpace<6>>(__VA_ARGS__); break;              \
    }

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
void ramsey_launch_async(const Arenas &a, const PersistArgs *d_args, const StepLaunch &sl,
                         const float *params, const void *wpk, uint32_t dyn_stride, size_t dyn_bytes, void *stream) {
    DISPATCH_RKW(a, l_async, a, d_args, sl, params, wpk, dyn_stride, dyn_bytes, (hipStream_t)stream);
}
// LDS plan of the asynchronous step: the per-wave region holds the search scratch + the clique counts
// during a call, and the row's activations [x][h0][h1] while the agent waits
bool ramsey_async_plan(const Arenas &a, const FusedEval &ev, uint32_t *dyn_stride, size_t *dyn_bytes, const char **why) {
    const char *dummy;
    if (!why) why = &dummy;
    if (a.B > 65536 || a.node_cap > 65536) { // (agent, node) are packed 16 + 16 bits in the argmin log
        *why = "asynchronous step: more than 65536 agents or nodes per tree";
        return false;
    }
    size_t stride = (RamseySpace<1>::dyn_bytes(a) + 15) & ~(size_t)15;
    if (ev.kind == 3) {
        for (int l = 0; l < ev.n_layers; ++l)
            if (ev.dims[l] % (l == 0 ? 4 : 16) != 0) {
                *why = "asynchronous step: hidden widths must be multiples of 16 and the input width a multiple of 4";
                return false;
            }
        size_t rows = ((size_t)((ev.dims[0] + 15) & ~15) + (size_t)ev.hid[0] + (size_t)ev.hid[1]) * sizeof(float) + 16 * PERSIST_WAVES;
        if (rows > stride) stride = (rows + 15) & ~(size_t)15;
    }
    const size_t total = stride * PERSIST_WAVES;
    const size_t static_lds = PERSIST_WAVES * (sizeof(RamseyLds) + 16) + sizeof(AsyncCtl) + 256;
    if (total + static_lds > 160 * 1024) {
        *why = "asynchronous step: 16 rows of activations do not fit the CU's 160 KB of LDS";
        return false;
    }
    *dyn_stride = (uint32_t)stride;
    *dyn_bytes = total;
    return true;
}

} // namespace azd
