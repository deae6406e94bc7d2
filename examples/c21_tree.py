#!/usr/bin/env python3
"""The reference's c21 driver (graph-state/examples/04-c21-tree.rs) over the MI355X engine: same loop,
same hyper-parameters, same console lines and tensorboard scalars.

    python examples/c21_tree.py [--epochs 250] [--episodes 800] [--batch 512] [--hidden 512 1024 512]

Differences forced by the boundary: the `init_states` / `modify_root` closures are the seeded
built-ins (a Rust closure cannot cross to the device), and a step of `--stride` calls runs on the device
between two looks at `ArgminImprovement` (stride 1 = the reference's call-by-call loop)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import azdopt_amd as az  # noqa: E402
from azdopt_amd import sinks  # noqa: E402

N = 19                                    # 04-c21-tree.rs:36
C_LOWER, C_UPPER = 2, 5 + 10              # :58-67 (ceil(sqrt(18)) + (19 + 1) / 2)
GOAL = (5.2 - C_LOWER) / (C_UPPER - C_LOWER)  # squish(5.2), :116


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--epochs", type=int, default=250)      # :131
    ap.add_argument("--episodes", type=int, default=800)    # :132
    ap.add_argument("--batch", type=int, default=512)       # :54
    ap.add_argument("--hidden", type=int, nargs="*", default=[512, 1024, 512])  # :42-52
    ap.add_argument("--stride", type=int, default=1)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--no-run-ahead", action="store_true", help="one launch per look at ArgminImprovement")
    ap.add_argument("--out", default=None, help="event-file directory (default: tf_path()/04-c21-tree/<time>)")
    args = ap.parse_args()

    space = az.ROTModifyParentsOnce(N)
    model = az.ActionModel(args.batch, space.STATE_DIM, space.ACTION_DIM, hidden=args.hidden,
                           lr=1e-4, betas=(0.9, 0.999), eps=1e-8, l2=1e-6, seed=args.seed)  # :86-92
    if args.out:
        os.makedirs(args.out, exist_ok=True)
        writer = sinks.TensorboardWriter(open(os.path.join(args.out, "tfevents-losses"), "wb"))
        writer.write_file_version()
    else:
        writer = sinks.TensorboardWriter.create("04-c21-tree")
    kmin, kmax = 5, space.ACTION_DIM // 2                    # :85
    roots = space.generate_roots(args.seed, args.batch, kmin=kmin, kmax=kmax)
    # arenas for one epoch's tree at its largest: a node per episode, at most kmax predictions per node (the reference's Vecs grow)
    # (tree_capacities names the limits of the packed records: 65536 nodes, 65535 arcs, 2^20 predictions per tree)
    opt = az.NablaOptimizer.par_new(space, roots, model, args.batch, **az.tree_capacities(args.episodes, kmax))

    def process_argmin(argmin, step):
        cost = argmin.cost
        print("%12s\tConjecture2Dot1Cost { matching: %s, lambda_1: %s }" % (argmin.eval, cost["matching"], cost["lambda_1"]))
        writer.write_summary(None, step, sinks.argmin_summary(argmin))
        writer.flush()
        if argmin.eval < GOAL:
            raise SystemExit("state is optimal:\n%s" % (argmin.state,))

    process_argmin(opt.argmin_data(), 0)
    n_as_tol = ([200, 50, 50], 25)                           # :134-136
    n_obs_tol = 200
    for epoch in range(1, args.epochs + 1):
        print("==== EPOCH: %d ====" % epoch)
        done = 0
        # the reference asks for its episodes one call at a time (:139-160); the epoch's calls are started in one launch and
        # the loop below is answered as they complete (NablaOptimizer.run_ahead; a no-op where the engine cannot do that)
        if args.stride < args.episodes and not args.no_run_ahead:
            opt.run_ahead(n_as_tol, args.episodes)
        while done < args.episodes:
            k = min(args.stride, args.episodes - done)
            improved = opt.par_roll_out_episodes(n_as_tol, n_calls=k)
            done += k
            if improved:
                process_argmin(opt.argmin_data(), args.episodes * (epoch - 1) + done)
        print("==== EPISODE: %d ====" % args.episodes)
        print("sizes:", sinks.sizes(opt.get_tree(0)))
        open("tree.dot", "w").write(sinks.tree_dot(opt.get_tree(0)))
        open("tree2.dot", "w").write(sinks.tree_dot(opt.get_tree(args.batch - 1)))
        loss = opt.par_update_model(n_obs_tol)
        step = args.episodes * epoch
        writer.write_summary(None, step, sinks.loss_summary(loss))
        writer.write_summary(None, step, sinks.argmin_summary(opt.argmin_data()))
        writer.flush()
        opt.par_reset_trees_policy(args.seed, epoch, kmin, kmax)


if __name__ == "__main__":
    main()
