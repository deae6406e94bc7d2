#!/usr/bin/env python3
"""Best cost found outside c21 (round-4 verdict, item 6): the drivers' loop -- episodes, ONE optimiser step, the device root policy --
with the model trained and, as the control, never updated, per epoch:
  r44      graph-state/examples/02-r44.rs:128-228 as it is (512 agents, 3200 episodes, 512-1024-512, its tolerance table), and at the
           bench's config D shape (8192 agents, 800 episodes, 3 x 256)
  dense    the build-defined dense-graph space at config E's shape (N = 50, G(50, 0.1) roots, 512-wide bf16 model, 8192 agents, 800
           episodes), roots of up to 128 and of up to 612 modifiable slots
Per epoch: the optimiser step's loss, the best cost so far (Ramsey: total monochromatic cliques, 0 = a Ramsey colouring; dense:
lambda_1 + mu), the mean over agents of their roots' evaluation (what the root policy does to the population) and the share of agents
whose tree holds a node better than its root.
    python tools/best_cost_spaces.py r44|r44D|dense128|dense612 [--epochs 40]"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import azdopt_amd as az  # noqa: E402

CASES = {
    "r44": dict(kind="ramsey", n=17, sizes=[4, 4], batch=512, episodes=3200, hidden=(512, 1024, 512), kmin=12, dtype="f32",
                tol=([200, 200, 100, 100, 50, 50, 25, 25], 10)),
    "r44D": dict(kind="ramsey", n=17, sizes=[4, 4], batch=8192, episodes=800, hidden=(256, 256, 256), kmin=12, dtype="f32",
                 tol=([200, 200, 100, 100, 50, 50, 25, 25], 10)),
    "dense128": dict(kind="dense", n=50, p=0.1, max_slots=128, batch=8192, episodes=800, hidden=(512, 512, 512), dtype="bf16", tol=([200, 50, 50], 25)),
    "dense612": dict(kind="dense", n=50, p=0.1, max_slots=612, batch=4096, episodes=800, hidden=(512, 512, 512), dtype="bf16", tol=([200, 50, 50], 25)),
}


def arm(c, args, train, lr):
    if c["kind"] == "ramsey":
        space = az.RamseySpaceNoEdgeRecolor(c["n"], c["sizes"], [1.0] * len(c["sizes"]))
        kmin, kmax = c["kmin"], space.default_permitted_range()[1]
        per_node = kmax * (len(c["sizes"]) - 1)
    else:
        space = az.DenseGraphSpace(c["n"], c["p"], max_slots=c["max_slots"])
        kmin, kmax = space.default_permitted_range()
        per_node = kmax
    B = args.batch or c["batch"]
    model = az.ActionModel(B, space.STATE_DIM, space.ACTION_DIM, hidden=c["hidden"], lr=lr, betas=(0.9, 0.999), eps=1e-8, l2=1e-6, seed=args.seed, dtype=c["dtype"])
    opt = az.NablaOptimizer.par_new(space, space.generate_roots(args.seed, B, kmin=kmin, kmax=kmax), model, B, **az.tree_capacities(c["episodes"], per_node))
    rows, t0, e0 = [], time.perf_counter(), opt.counters()["EXPANSIONS"]

    def best_of(am):
        return float(sum(am.cost["clique_counts"])) if c["kind"] == "ramsey" else len(am.cost["matching"]) + am.cost["lambda_1"]

    for epoch in range(1, args.epochs + 1):
        opt.par_roll_out_episodes(c["tol"], n_calls=c["episodes"])
        am = opt.argmin_data()
        # the population: every 16th agent's tree -- its root's evaluation, and whether the search found a node below it
        roots_c, better = [], 0
        for i in range(0, B, max(1, B // 64)):
            t = opt.get_tree(i)
            roots_c.append(float(t.c[0]))
            better += int((t.c < t.c[0]).any())
        loss = opt.par_update_model(200) if train else float("nan")
        rows.append((epoch, loss, best_of(am), float(am.eval), float(np.mean(roots_c)), better / len(roots_c)))
        if c["kind"] == "ramsey" and am.eval == 0:
            break
        opt.par_reset_trees_policy(args.seed, epoch, kmin, kmax)
    dt = time.perf_counter() - t0
    return rows, (opt.counters()["EXPANSIONS"] - e0) / dt, opt.step_form()[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("case", choices=sorted(CASES))
    ap.add_argument("--epochs", type=int, default=40)
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--lr", type=float, nargs="*", default=[1e-4])
    args = ap.parse_args()
    c = CASES[args.case]
    print("# %s: %s, %d agents, model %s (%s), %d epochs x %d episodes, tol %s, seed %d" % (
        args.case, c["kind"], args.batch or c["batch"], "-".join(map(str, c["hidden"])), c["dtype"], args.epochs, c["episodes"], c["tol"], args.seed))
    arms = {}
    for name, train, lr in [("trained lr %g" % lr, True, lr) for lr in args.lr] + [("frozen", False, 1e-4)]:
        rows, rate, form = arm(c, args, train, lr)
        arms[name] = rows
        print("# %-16s %d epochs, %.2f M expansions/s over the whole loop (%s): loss %.5f -> %.5f, best %.4f -> %.4f, mean root eval %.4f -> %.4f" % (
            name, len(rows), rate / 1e6, form, rows[0][1], rows[-1][1], rows[0][2], rows[-1][2], rows[0][4], rows[-1][4]), flush=True)
    names = list(arms)
    print("epoch  " + "  ".join("%-44s" % (n + ": loss / best / mean root eval / better") for n in names))
    ne = max(len(r) for r in arms.values())
    for e in list(range(0, min(ne, 10))) + list(range(10, ne, 5)) + ([ne - 1] if ne > 10 and (ne - 1 - 10) % 5 else []):
        cells = []
        for n in names:
            r = arms[n]
            cells.append("%-44s" % ("%.5f / %.4f / %.4f / %.2f" % (r[e][1], r[e][2], r[e][4], r[e][5]) if e < len(r) else ""))
        print("%5d  " % (e + 1) + "  ".join(cells))


if __name__ == "__main__":
    main()
