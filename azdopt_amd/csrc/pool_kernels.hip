// pool_kernels.hip -- the pool step (pool_step.inc: agents multiplexed over searcher waves, evaluator workgroups on
// CUs of their own) for the c21 space, in its own translation unit like the asynchronous step (async_kernels.hip).
#define AZD_TU_ASYNC 1
#define AZD_TU_POOL 1
#include "tree_kernels.hip"
