// ramsey_pool_kernels.hip -- the pool step (pool_step.inc) for the Ramsey space, in its own translation unit.
#define AZD_TU_ASYNC 1
#define AZD_TU_POOL 1
#include <hip/hip_runtime.h>

#include "bf16.h"
#include "engine_types.h"

namespace azd {

#include "tree_core.inc"
#include "space_ramsey.inc"
#include "persistent_step.inc"
#include "async_step.inc"
#include "pool_step.inc"

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
static void l_pool(const Arenas &a, const PersistArgs *d_args, const StepLaunch &sl, const float *params,
                   const void *wpk, int n_blocks, uint32_t dyn_stride, size_t dyn_bytes, hipStream_t st) {
    const int mode = (sl.hashed ? 1 : 0) | (sl.window ? 2 : 0);
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
    }
#undef AZD_LAUNCH_POOL
    k_argmin_log1<SP><<<dim3(1), dim3(64), SP::dyn_bytes(a), st>>>(a, sl.n_calls, sl.log_key, sl.ctl);
}
void ramsey_launch_pool(const Arenas &a, const PersistArgs *d_args, const StepLaunch &sl, const float *params,
                        const void *wpk, int n_blocks, uint32_t dyn_stride, size_t dyn_bytes, void *stream) {
    DISPATCH_RKW(a, l_pool, a, d_args, sl, params, wpk, n_blocks, dyn_stride, dyn_bytes, (hipStream_t)stream);
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
int ramsey_pool_max_resident(const Arenas &a, size_t dyn_bytes, int n_cus) {
    int nb = 0;
    DISPATCH_RKW(a, q_pool_resident, &nb, dyn_bytes);
    return nb * n_cus;
}
bool ramsey_pool_plan(const Arenas &a, const FusedEval &ev, PoolArgs *pool, uint32_t *dyn_stride, size_t *dyn_bytes, const char **why) {
    const char *dummy;
    if (!why) why = &dummy;
    return pool_plan_common(a, ev, pool, dyn_stride, dyn_bytes, why, RamseySpace<1>::pool_dyn_bytes(a), sizeof(RamseyLds));
}

} // namespace azd
