import os, sys, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import azdopt_amd as az
from oracle import orc
n, B, seed = 19, 512, 9
space = az.ROTModifyParentsOnce(n)
roots = space.generate_roots(seed, B)
TOL = ([200, 50, 50], 25)
for mode in ("0", "auto"):
    if mode == "auto":
        os.environ.pop("AZD_POOL_EVAL_GROUP", None)
    else:
        os.environ["AZD_POOL_EVAL_GROUP"] = mode
    model = az.ActionModel(B, space.STATE_DIM, space.ACTION_DIM, hidden=(512, 1024, 512), seed=seed)
    opt = az.NablaOptimizer.par_new(space, roots, model, B, pool_step=True)
    oe = orc.Engine(n, B, threads=8)
    oe.new_begin(*roots)
    oe.new_end(opt.predictions())
    for chunk in (300, 100):
        for _ in range(chunk):
            oe.rollout_begin(*TOL)
            oe.rollout_end(opt.debug_tile_forward(oe.state_vecs()))
        opt.par_roll_out_episodes(TOL, n_calls=chunk)
        sv_ok = np.array_equal(opt.state_vecs(), oe.state_vecs())
        bad = np.where((opt.state_vecs() != oe.state_vecs()).any(axis=1))[0]
        cg, co = opt.counters(), oe.counters()
        if chunk == 300:
            print("  loss", opt.par_update_model(200))
        print(mode, opt.pool_groups(), "after +%d calls: state vecs equal %s, agents differing %d %s, EXP %d/%d" % (chunk, sv_ok, len(bad), bad[:8], cg["EXPANSIONS"], co["EXPANSIONS"]), flush=True)
