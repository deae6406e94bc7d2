"""The drivers under examples/ (the reference's 04-c21-tree.rs / 02-r44.rs loops over the engine) run end to
end on the GPU and leave the event file the reference would: version record, cost scalars, loss scalars."""
import os
import subprocess
import sys

import pytest

from azdopt_amd import sinks

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(script, *args, cwd):
    subprocess.run([sys.executable, os.path.join(ROOT, "examples", script), *args], check=True, cwd=cwd, timeout=300)


def test_c21_driver_writes_the_reference_scalars(tmp_path):
    out = tmp_path / "ev"
    run("c21_tree.py", "--epochs", "2", "--episodes", "60", "--batch", "64", "--hidden", "64", "64", "--stride", "20", "--out", str(out),
        cwd=tmp_path)
    ev = sinks.read_events(out / "tfevents-losses")
    assert ev[0][2] == "brain.Event:2"
    tags = [t for e in ev for t, _ in e[3]]
    assert tags.count("loss") == 2 and {"cost/cost", "cost/lambda_1", "cost/mu"} <= set(tags)
    steps = [e[1] for e in ev if any(t == "loss" for t, _ in e[3])]
    assert steps == [60, 120]
    assert (tmp_path / "tree.dot").read_text().startswith("graph search_tree {")


def test_ramsey_driver_writes_the_reference_scalars(tmp_path):
    out = tmp_path / "ev"
    run("ramsey.py", "r44", "--epochs", "2", "--episodes", "40", "--batch", "32", "--hidden", "64", "--stride", "10", "--out", str(out),
        cwd=tmp_path)
    ev = sinks.read_events(out / "tfevents-losses")
    tags = [t for e in ev for t, _ in e[3]]
    assert tags.count("loss") == 2 and {"clique_counts/0", "clique_counts/1"} <= set(tags)


@pytest.mark.parametrize("batch,stride", [(64, 20), (256, 1)])
def test_cpp_host_drives_the_same_engine(tmp_path, batch, stride):
    """examples/c21_tree.cpp over include/azdopt_amd.hpp (a compiled host on the C ABI) and the Python host run the same
    two epochs: same losses, same best evaluation and lambda_1 line by line.  256 agents at stride 1 = the reference's
    call-by-call loop on the pool step, which the compiled host answers from a run-ahead window (one launch per epoch); the
    Python host beside it makes a launch per call."""
    import azdopt_amd as az
    exe = tmp_path / "c21_tree"
    subprocess.run(["g++", "-O1", "-std=c++17", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "c21_tree.cpp"), "-o", str(exe),
                    "-L" + os.path.join(ROOT, "azdopt_amd"), "-lazdopt_amd", "-Wl,-rpath," + os.path.join(ROOT, "azdopt_amd")], check=True, timeout=300)
    epochs, episodes, seed, hidden = 2, 60, 3, [64, 64]
    r = subprocess.run([str(exe), str(epochs), str(episodes), str(batch), str(stride), str(seed)] + [str(h) for h in hidden],
                       capture_output=True, text=True, timeout=300, check=True)
    lines = r.stdout.splitlines()
    # the same loop through the Python host
    space = az.ROTModifyParentsOnce(19)
    model = az.ActionModel(batch, space.STATE_DIM, space.ACTION_DIM, hidden=hidden, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, l2=1e-6, seed=seed)
    kmin, kmax = 5, space.ACTION_DIM // 2
    opt = az.NablaOptimizer.par_new(space, space.generate_roots(seed, batch, kmin=kmin, kmax=kmax), model, batch)
    want = []

    def show(a):
        want.append("%12.9g\tConjecture2Dot1Cost { matching: [%s], lambda_1: %.17g }" % (
            a.eval, ", ".join("(%d, %d)" % tuple(p) for p in a.cost["matching"]), a.cost["lambda_1"]))

    show(opt.argmin_data())
    for epoch in range(1, epochs + 1):
        want.append("==== EPOCH: %d ====" % epoch)
        done = 0
        while done < episodes:
            improved = opt.par_roll_out_episodes(([200, 50, 50], 25), n_calls=stride)
            done += stride
            if improved:
                show(opt.argmin_data())
        if batch >= 256:
            assert opt.step_form()[0] == "pool"
        want.append("==== EPISODE: %d ====" % episodes)
        want.append("loss: %.9g" % opt.par_update_model(200))
        opt.par_reset_trees_policy(seed, epoch, kmin, kmax)
    assert lines[:len(want)] == want, "\n".join(lines[:len(want)]) + "\n---\n" + "\n".join(want)
    assert lines[len(want)].startswith("step form ")


def test_cpp_host_ramsey_driver_matches_the_python_host(tmp_path):
    """examples/ramsey.cpp (02-r44.rs over include/azdopt_amd.hpp) against the same loop through the Python host"""
    import azdopt_amd as az
    exe = tmp_path / "ramsey"
    subprocess.run(["g++", "-O1", "-std=c++17", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "ramsey.cpp"), "-o", str(exe),
                    "-L" + os.path.join(ROOT, "azdopt_amd"), "-lazdopt_amd", "-Wl,-rpath," + os.path.join(ROOT, "azdopt_amd")], check=True, timeout=300)
    epochs, episodes, batch, stride, seed, hidden = 2, 40, 32, 10, 4, [64]
    r = subprocess.run([str(exe), "r44", str(epochs), str(episodes), str(batch), str(stride), str(seed)] + [str(h) for h in hidden],
                       capture_output=True, text=True, timeout=300, check=True)
    lines = r.stdout.splitlines()
    space = az.RamseySpaceNoEdgeRecolor(17, [4, 4], [1.0, 1.0])
    model = az.ActionModel(batch, space.STATE_DIM, space.ACTION_DIM, hidden=hidden, lr=1e-4, l2=1e-6, seed=seed)
    kmin, kmax = 12, space.default_permitted_range()[1]
    caps = dict(node_capacity=episodes * 2 + 64, arc_capacity=episodes * 3 + 64, prediction_capacity=(episodes + 1) * kmax + 128)
    opt = az.NablaOptimizer.par_new(space, space.generate_roots(seed, batch, kmin=kmin, kmax=kmax), model, batch, **caps)
    tol = ([200, 200, 100, 100, 50, 50, 25, 25], 10)
    want = []

    def show(a):
        want.append("%.9g\tTotalCounts([%s])" % (a.eval, ", ".join(str(int(x)) for x in a.cost["clique_counts"])))

    show(opt.argmin_data())
    for epoch in range(1, epochs + 1):
        want.append("==== EPOCH: %d ====" % epoch)
        done = 0
        while done < episodes:
            if opt.par_roll_out_episodes(tol, n_calls=stride):
                show(opt.argmin_data())
            done += stride
        want.append("==== EPISODE: %d ====" % episodes)
        want.append("loss: %.9g" % opt.par_update_model(200))
        opt.par_reset_trees_policy(seed, epoch, kmin, kmax)
    # (a root whose colouring has no monochromatic clique makes h_sa = 1 - c*/c a NaN, and with it the loss, in the reference
    # too -- its drivers stop there; printf spells the sign of a NaN, Python does not)
    lines = [x.replace("-nan", "nan") for x in lines]
    assert lines == want, "\n".join(lines) + "\n---\n" + "\n".join(want)


def test_cpp_host_model_seam_matches_the_python_host(tmp_path):
    """a NablaModel on the host side of the boundary (azdopt::HostModel in include/azdopt_amd.hpp; tests/cpp/host_model.cpp)
    against the same model fed through the Python host's *_begin / *_end calls"""
    import numpy as np

    import azdopt_amd as az
    exe = tmp_path / "host_model"
    subprocess.run(["g++", "-O1", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "cpp", "host_model.cpp"), "-o", str(exe), "-L" + os.path.join(ROOT, "azdopt_amd"), "-lazdopt_amd",
                    "-Wl,-rpath," + os.path.join(ROOT, "azdopt_amd")], check=True, timeout=300)
    batch, calls, seed = 24, 30, 2
    r = subprocess.run([str(exe), str(batch), str(calls)], capture_output=True, text=True, timeout=300, check=True)
    space = az.ROTModifyParentsOnce(13)
    S, A = space.STATE_DIM, space.ACTION_DIM
    kmin, kmax = 3, A // 2

    def predict(sv):
        nz = (sv != 0).sum(axis=1).astype(np.int64)
        a = np.arange(A, dtype=np.int64)
        return (((a[None, :] * 7 + nz[:, None] * 13) % 97).astype(np.float32) / np.float32(97.0)).astype(np.float32)

    opt = az.NablaOptimizer(space, None, batch)
    opt.par_new_begin(*space.generate_roots(seed, batch, kmin=kmin, kmax=kmax))
    opt.par_new_end(predict(opt.state_vecs()))
    want = []
    for epoch in range(2):
        improved = 0
        for _ in range(calls):
            opt.roll_out_begin(([8, 4, 2], 1))
            improved += opt.roll_out_end(predict(opt.state_vecs()))
        a = opt.argmin_data()
        want.append("improved %d eval %.9g lambda_1 %.17g matching %d" % (improved, a.eval, a.cost["lambda_1"], len(a.cost["matching"])))
        sv, obs, w = opt.observe(3)
        want.append("loss %.9g" % np.float32((w != 0).sum()))
        roots = opt.modify_roots(seed, epoch, kmin, kmax, device=True)
        opt.reset_begin(*roots)
        opt.reset_end(predict(opt.state_vecs()))
    c = opt.counters()
    want.append("expansions %d transpositions %d terminals %d" % (c["EXPANSIONS"], c["TRANSPOSITIONS"], c["TERMINALS"]))
    dense = az.DenseGraphSpace(12, 0.3)
    mlp = az.ActionModel(batch, dense.STATE_DIM, dense.ACTION_DIM, hidden=[64], seed=seed)
    dopt = az.NablaOptimizer.par_new(dense, dense.generate_roots(seed, batch, kmin=4, kmax=20), mlp, batch)
    dimp = dopt.par_roll_out_episodes(([20, 10, 5], 3), n_calls=calls)
    da = dopt.argmin_data()
    want.append("dense improved %d eval %.9g lambda_1 %.17g matching %d loss %.9g" % (
        dimp, da.eval, da.cost["lambda_1"], len(da.cost["matching"]), dopt.par_update_model(3)))
    big = az.DenseGraphSpace(64, 0.15, max_slots=200)
    bopt = az.NablaOptimizer.par_new(big, big.generate_roots(seed, 4, kmin=150, kmax=200), az.TrivialModel(big.STATE_DIM, big.ACTION_DIM), 4)
    bopt.par_roll_out_episodes(([4, 2], 1), n_calls=6)
    ba = bopt.argmin_data()
    want.append("dense64 slot words %d open slots %d eval %.9g" % ((big.E + 63) // 64, sum(bin(int(w)).count("1") for w in ba.state["permitted"]), ba.eval))
    bopt.par_reset_trees_policy(seed, 1, 150, 200)
    bopt.par_roll_out_episodes(([4, 2], 1), n_calls=3)
    want.append("dense64 after policy eval %.9g" % bopt.argmin_data().eval)
    assert r.stdout.splitlines() == want, r.stdout + "\n---\n" + "\n".join(want)
