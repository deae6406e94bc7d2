#!/usr/bin/env python3
"""The other half of the metric: best cost found (lambda_1 + mu of the argmin) over the reference driver's run length
(graph-state/examples/04-c21-tree.rs:133-208: 250 epochs x 800 episodes, one optimiser step per epoch, the root policy between
epochs), per epoch, for
  trained   the driver's loop as it is (Adam, lr 1e-4, L2 1e-6: 04-c21-tree.rs:86-92)
  frozen    the same loop with the model never updated (control: what the search and the root policy find by themselves)
  lr x100   the same loop with lr 1e-2 (does a model that visibly learns change the search?)
  cpu       the CPU restatement (oracle/: tree search + its own MLP and Adam) on the same roots, for as many epochs as
            --cpu-seconds buys (same seeds; rows differ from the GPU's in the last bits, so it is a second sample, not a replay)
    python tools/best_cost.py [--batch 512] [--hidden 512 1024 512] [--epochs 250] [--cpu-seconds 60]"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import azdopt_amd as az  # noqa: E402

N, TOL, N_OBS_TOL, GOAL = 19, ([200, 50, 50], 25), 200, 5.2


def gpu_arm(args, lr, train):
    space = az.ROTModifyParentsOnce(N)
    model = az.ActionModel(args.batch, space.STATE_DIM, space.ACTION_DIM, hidden=args.hidden, lr=lr, betas=(0.9, 0.999), eps=1e-8, l2=1e-6, seed=args.seed)
    kmin, kmax = 5, space.ACTION_DIM // 2
    roots = space.generate_roots(args.seed, args.batch, kmin=kmin, kmax=kmax)
    opt = az.NablaOptimizer.par_new(space, roots, model, args.batch, node_capacity=4096, arc_capacity=8192, prediction_capacity=(args.episodes + 1) * kmax + 128)
    rows, t0, exp0 = [], time.perf_counter(), opt.counters()["EXPANSIONS"]
    for epoch in range(1, args.epochs + 1):
        opt.par_roll_out_episodes(TOL, n_calls=args.episodes)
        am = opt.argmin_data()
        best = len(am.cost["matching"]) + am.cost["lambda_1"]
        if train:
            loss = opt.par_update_model(N_OBS_TOL)
        else:  # the loss the optimiser step would have seen (dfdx.rs:103-113), without taking the step
            sv, obs, w = opt.observe(N_OBS_TOL)
            pred = np.zeros_like(obs)
            model.write_predictions(sv, pred)
            tot = float(w.sum())
            loss = float((w / tot * (pred - obs) ** 2).sum()) if tot > 0 else 0.0
        rows.append((epoch, loss, best))
        if best < GOAL:
            break
        opt.par_reset_trees_policy(args.seed, epoch, kmin, kmax)
    dt = time.perf_counter() - t0
    return rows, (opt.counters()["EXPANSIONS"] - exp0) / dt, opt.step_form()


def cpu_arm(args):
    from oracle import orc
    space = az.ROTModifyParentsOnce(N)
    kmin, kmax = 5, space.ACTION_DIM // 2
    roots = space.generate_roots(args.seed, args.batch, kmin=kmin, kmax=kmax)
    threads = min(os.cpu_count() or 8, 64)
    mlp = orc.Mlp([space.STATE_DIM] + list(args.hidden) + [space.ACTION_DIM], lr=1e-4, l2=1e-6, seed=args.seed, threads=threads)
    oe = orc.Engine(N, args.batch, threads=threads)
    oe.new_begin(*roots)
    oe.new_end(mlp.forward_fast(oe.state_vecs()))
    rows, t0, epoch = [], time.perf_counter(), 0
    while time.perf_counter() - t0 < args.cpu_seconds and epoch < args.epochs:
        epoch += 1
        for _ in range(args.episodes):
            oe.rollout_begin(*TOL)
            oe.rollout_end(mlp.forward_fast(oe.state_vecs()))
        print("# cpu epoch %d done at %.0f s" % (epoch, time.perf_counter() - t0), flush=True)
        am = oe.argmin()
        best = am["matching"] + am["lambda1"]
        obs, w = oe.observe(N_OBS_TOL)
        loss = mlp.update(oe.state_vecs(), obs, w)
        rows.append((epoch, loss, best))
        new_roots = oe.modify_roots(args.seed, epoch, 0, kmin, kmax)
        oe.reset_begin(*new_roots)
        oe.reset_end(mlp.forward_fast(oe.state_vecs()))
    dt = time.perf_counter() - t0
    return rows, oe.counters()["EXPANSIONS"] / dt, threads


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=512)
    ap.add_argument("--hidden", type=int, nargs="*", default=[512, 1024, 512])
    ap.add_argument("--epochs", type=int, default=250)
    ap.add_argument("--episodes", type=int, default=800)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--cpu-seconds", type=float, default=60.0)
    args = ap.parse_args()
    print("# c21 N = 19, %d agents, model %s, %d epochs x %d episodes, tol %s, seed %d" % (args.batch, "-".join(map(str, args.hidden)), args.epochs, args.episodes, TOL, args.seed))
    arms = {}
    for name, lr, train in (("trained", 1e-4, True), ("frozen", 1e-4, False), ("lr_x100", 1e-2, True)):
        rows, rate, form = gpu_arm(args, lr, train)
        arms[name] = rows
        print("# %-8s %d epochs%s, %.2f M expansions/s over the whole loop (%s), loss %.5f -> %.5f, best cost %.4f -> %.4f" % (
            name, len(rows), " (goal %.1f reached: the driver stops, 04-c21-tree.rs:117,125)" % GOAL if rows[-1][2] < GOAL else "", rate / 1e6, form[0],
            rows[0][1], rows[-1][1], rows[0][2], rows[-1][2]), flush=True)
    if args.cpu_seconds > 0:
        rows, rate, threads = cpu_arm(args)
        arms["cpu"] = rows
        print("# %-8s %d epochs in %.0f s on %d host threads, %.3f M expansions/s, loss %.5f -> %.5f, best cost %.4f -> %.4f" % (
            "cpu", len(rows), args.cpu_seconds, threads, rate / 1e6, rows[0][1], rows[-1][1], rows[0][2], rows[-1][2]))
    names = list(arms)
    print("epoch  " + "  ".join("%-22s" % (n + " loss / best") for n in names))
    ne = max(len(r) for r in arms.values())
    for e in list(range(0, min(ne, 10))) + list(range(10, ne, 10)) + ([ne - 1] if (ne - 1) % 10 else []):
        cells = []
        for n in names:
            r = arms[n]
            cells.append("%.5f / %.4f      " % (r[e][1], r[e][2]) if e < len(r) else " " * 22)
        print("%5d  " % (e + 1) + "  ".join(cells))


if __name__ == "__main__":
    main()
