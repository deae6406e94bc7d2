// async_kernels.hip -- the asynchronous CU-resident step (async_step.inc) in its own translation
// unit.  Co-compiled with k_persist in tree_kernels.hip, the shared device functions
// (rollout_agent, add_actions_agent) got different inlining/register allocation and k_persist's
// scratch use rose from 128 to 392 B/lane (-12 % end to end); separate TUs keep both at their own optimum.
#define AZD_TU_ASYNC 1
#include "tree_kernels.hip"
