import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import azdopt_amd as az
B = 512
space = az.ROTModifyParentsOnce(19)
model = az.ActionModel(B, space.STATE_DIM, space.ACTION_DIM, hidden=(512, 1024, 512), seed=0)
nc, ac, pc = az.tree_capacities(800, 76)
opt = az.NablaOptimizer.par_new(space, space.generate_roots(0, B), model, B, pool_step=True)
tol = ([200, 50, 50], 25)
for ep in range(5):
    opt.set_timing(True)
    t0 = time.perf_counter(); opt.par_roll_out_episodes(tol, n_calls=800); t1 = time.perf_counter()
    loss = opt.par_update_model(200); t2 = time.perf_counter()
    opt.par_reset_trees_policy(0, ep + 1); t3 = time.perf_counter()
    print("epoch %d: rollout %.1f ms (kernel %.1f), update %.1f ms, reset %.1f ms, groups %s split %s form %s" % (ep, (t1 - t0) * 1e3, opt.timing()["rollout_ms"], (t2 - t1) * 1e3, (t3 - t2) * 1e3, opt.pool_groups(), opt.pool_split(), opt.step_form()))
