"""GPU tests of the multi-GPU plumbing that one GPU can exercise: the native (C ABI + RCCL) epoch exchange with a
one-rank communicator, the ShardedOptimizer class at world size 1, and engines on two device ordinals in one
process when two are visible."""
import ctypes as C
import os

import numpy as np
import pytest

from test_gpu_parity import MAIN_CTRS, TOL_REF

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def az():
    import azdopt_amd
    assert azdopt_amd.device_count() > 0, "no MI355X visible"
    return azdopt_amd


def _load_rccl():
    for name in ("librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"):
        try:
            return C.CDLL(name, mode=C.RTLD_GLOBAL)
        except OSError:
            continue
    pytest.skip("librccl.so not found")


class _Uid(C.Structure):
    _fields_ = [("internal", C.c_char * 128)]


def _twin(az, B, seed, hidden=(64, 32)):
    space = az.ROTModifyParentsOnce(13)
    model = az.ActionModel(B, space.STATE_DIM, space.ACTION_DIM, hidden=hidden, seed=seed)
    opt = az.NablaOptimizer.par_new(space, space.generate_roots(seed, B), model, B)
    opt.par_roll_out_episodes(TOL_REF, n_calls=40)
    return opt, model


def test_native_sharded_update_with_one_rank_communicator(az):
    """azd_engine_par_update_model_sharded (observe -> ncclAllGather x 3 on the engine's stream -> the optimiser
    step on the pooled rows) with an ncclComm_t of one rank equals azd_engine_par_update_model: same loss, same
    parameters bit for bit.  This is the sequence a C / Rust host runs on every rank."""
    rccl = _load_rccl()
    uid = _Uid()
    assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
    comm = C.c_void_p()
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, _Uid, C.c_int]
    assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
    try:
        B = 96
        o1, m1 = _twin(az, B, 5)
        o2, m2 = _twin(az, B, 5)
        for it in range(2):
            l1 = o1.par_update_model(2)
            l2 = o2.par_update_model_sharded(2, comm)
            assert l1 == l2 and l1 > 0, it
            assert np.array_equal(m1.get_params().view(np.uint32), m2.get_params().view(np.uint32)), it
    finally:
        rccl.ncclCommDestroy.argtypes = [C.c_void_p]
        rccl.ncclCommDestroy(comm)


def test_sharded_optimizer_world_size_one_equals_plain_optimizer(az):
    """ShardedOptimizer without a process group is the plain NablaOptimizer: same counters, loss and argmin."""
    import torch
    from azdopt_amd.parallel import ShardedOptimizer
    space = az.ROTModifyParentsOnce(13)
    B, seed = 80, 2
    mk = lambda total: az.ActionModel(total, space.STATE_DIM, space.ACTION_DIM, hidden=(64, 32), seed=seed)
    sopt = ShardedOptimizer.par_new(space, mk, B, dist=None, torch=torch, seed=seed)
    model = mk(B)
    opt = az.NablaOptimizer.par_new(space, space.generate_roots(seed, B), model, B)
    for o in (sopt, opt):
        o.par_roll_out_episodes(TOL_REF, n_calls=50)
    assert sopt.par_update_model(2) == opt.par_update_model(2)
    sopt.par_reset_trees_policy(seed, 0)
    opt.par_reset_trees_policy(seed, 0)
    for o in (sopt, opt):
        o.par_roll_out_episodes(TOL_REF, n_calls=20)
    c0, c1 = sopt.shard.opt.counters(), opt.counters()
    for k in MAIN_CTRS:
        assert c0[k] == c1[k], k
    assert sopt.global_argmin()[0] == float(opt.argmin_data().eval)
    assert sopt.total_expansions() == c1["EXPANSIONS"]


def test_engines_on_two_devices_in_one_process(az):
    """The dynamic-LDS attribute of the CU-resident kernels is per device: an engine on a second device ordinal must
    run the asynchronous step too (it used to be set once per process)."""
    if az.device_count() < 2:
        pytest.skip("one device visible")
    space = az.ROTModifyParentsOnce(19)
    B = 64
    roots = space.generate_roots(0, B)
    res = []
    for dev in (0, 1):
        model = az.ActionModel(B, space.STATE_DIM, space.ACTION_DIM, hidden=(512, 1024, 512), seed=1, device=dev)
        o = az.NablaOptimizer.par_new(space, roots, model, B, device=dev)
        o.par_roll_out_episodes(TOL_REF, n_calls=30)
        assert o.step_form()[0] == "async"  # 64 agents: below the pool step's threshold
        res.append(o.counters())
    for k in MAIN_CTRS:
        assert res[0][k] == res[1][k], k
