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
