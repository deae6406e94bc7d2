#!/usr/bin/env python3
"""The reference's Ramsey drivers (graph-state/examples/01-r333.rs, 02-r44.rs) over the MI355X engine.

    python examples/ramsey.py r333|r44 [--epochs 250] [--episodes N] [--batch B]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import azdopt_amd as az  # noqa: E402
from azdopt_amd import sinks  # noqa: E402

DRIVERS = {  # N, SIZES, BATCH, episodes, n_as_tol, num_permitted_edges_range.start
    "r333": dict(n=16, sizes=[3, 3, 3], batch=256, episodes=6400, kmin=10,
                 tol=([200, 200, 200, 100, 100, 100, 50, 50, 50, 25, 25, 25], 10), tag="01-r333-grad"),   # 01-r333.rs:35-38,61,83,126-130
    "r44": dict(n=17, sizes=[4, 4], batch=512, episodes=3200, kmin=12,
                tol=([200, 200, 100, 100, 50, 50, 25, 25], 10), tag="01-r333-grad"),                      # 02-r44.rs:35-38,61,83,126-130
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("driver", choices=sorted(DRIVERS))
    ap.add_argument("--epochs", type=int, default=250)
    ap.add_argument("--episodes", type=int, default=0)
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--hidden", type=int, nargs="*", default=[512, 1024, 512])
    ap.add_argument("--stride", type=int, default=1)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    d = DRIVERS[args.driver]
    batch = args.batch or d["batch"]
    episodes = args.episodes or d["episodes"]

    space = az.RamseySpaceNoEdgeRecolor(d["n"], d["sizes"], [1.0] * len(d["sizes"]))
    model = az.ActionModel(batch, space.STATE_DIM, space.ACTION_DIM, hidden=args.hidden, lr=1e-4, l2=1e-6, seed=args.seed)
    if args.out:
        os.makedirs(args.out, exist_ok=True)
        writer = sinks.TensorboardWriter(open(os.path.join(args.out, "tfevents-losses"), "wb"))
        writer.write_file_version()
    else:
        writer = sinks.TensorboardWriter.create(d["tag"])
    kmin, kmax = d["kmin"], space.default_permitted_range()[1]   # ..=(E / 2), capped by what a node holds
    C = len(d["sizes"])
    caps = az.tree_capacities(episodes, kmax * (C - 1))  # (limits of the packed records: 65536 nodes, 65535 arcs, 2^20 predictions)
    opt = az.NablaOptimizer.par_new(space, space.generate_roots(args.seed, batch, kmin=kmin, kmax=kmax), model, batch, **caps)

    def process_argmin(argmin, step):
        print("%s\tTotalCounts(%s)" % (argmin.eval, argmin.cost["clique_counts"]))
        writer.write_summary(None, step, sinks.clique_counts_summary(argmin.cost["clique_counts"]))
        writer.flush()
        if argmin.eval == 0:
            raise SystemExit("state is optimal:\n%s" % (argmin.state["colors"],))

    process_argmin(opt.argmin_data(), 0)
    for epoch in range(1, args.epochs + 1):
        print("==== EPOCH: %d ====" % epoch)
        done = 0
        if args.stride < episodes:  # the epoch's calls in one launch, handed out call by call (NablaOptimizer.run_ahead)
            opt.run_ahead(d["tol"], episodes)
        while done < episodes:
            k = min(args.stride, episodes - done)
            if opt.par_roll_out_episodes(d["tol"], n_calls=k):
                process_argmin(opt.argmin_data(), episodes * (epoch - 1) + done + k)
            done += k
        print("==== EPISODE: %d ====" % episodes)
        print("sizes:", sinks.sizes(opt.get_tree(0)))
        loss = opt.par_update_model(200)
        writer.write_summary(None, episodes * epoch, sinks.loss_summary(loss))
        writer.write_summary(None, episodes * epoch, sinks.clique_counts_summary(opt.argmin_data().cost["clique_counts"]))
        writer.flush()
        opt.par_reset_trees_policy(args.seed, epoch, kmin, kmax)


if __name__ == "__main__":
    main()
