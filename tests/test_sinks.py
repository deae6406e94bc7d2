"""Observability sinks: event-file framing, summaries, tree dump, sizes()."""
import io
import struct

import numpy as np

from azdopt_amd import sinks
from oracle import py_oracle as po

from test_oracle_cross import unpack_roots


def test_crc32c_known_answers():
    # RFC 3720 B.4 check values for CRC-32C
    assert sinks.crc32c(b"123456789") == 0xE3069283
    assert sinks.crc32c(bytes(32)) == 0x8A9136AA
    assert sinks.crc32c(bytes([0xFF] * 32)) == 0x62A8AB43
    assert sinks.crc32c(bytes(range(32))) == 0x46DD794E


def test_event_file_round_trip(tmp_path):
    p = tmp_path / "tfevents-losses"
    w = sinks.TensorboardWriter(open(p, "wb"))
    w.write_file_version(wall_time=1.5)
    w.write_summary(2.5, 0, sinks.c21_cost_summary(3.25, 8))
    w.write_summary(3.5, 7, sinks.loss_summary(np.float32(0.125)))
    w.write_summary(4.5, 300, sinks.clique_counts_summary([4, 0, 9]))
    w.flush()
    w.close()
    ev = sinks.read_events(p)
    assert ev[0] == (1.5, 0, "brain.Event:2", [])
    assert ev[1] == (2.5, 0, None, [("cost/cost", 11.25), ("cost/lambda_1", 3.25), ("cost/mu", 8.0)])
    assert ev[2] == (3.5, 7, None, [("loss", 0.125)])
    assert ev[3] == (4.5, 300, None, [("clique_counts/0", 4.0), ("clique_counts/1", 0.0), ("clique_counts/2", 9.0)])


def test_event_bytes_known_answer():
    """One record spelled out by hand: Event{wall_time=1.0, step=2, summary{value{tag "loss", simple_value 0.5}}}."""
    buf = io.BytesIO()
    sinks.TensorboardWriter(buf).write_summary(1.0, 2, sinks.loss_summary(0.5))
    value = b"\x0a\x04loss" + b"\x15" + struct.pack("<f", 0.5)
    summary = b"\x0a" + bytes([len(value)]) + value
    event = b"\x09" + struct.pack("<d", 1.0) + b"\x10\x02" + b"\x2a" + bytes([len(summary)]) + summary
    head = struct.pack("<Q", len(event))
    want = head + struct.pack("<I", sinks.masked_crc32c(head)) + event + struct.pack("<I", sinks.masked_crc32c(event))
    assert buf.getvalue() == want


def test_tf_path(monkeypatch):
    monkeypatch.setenv("OUT_DIR", "/x/out")
    monkeypatch.setenv("CARGO_MANIFEST_DIR", "/x/crate")
    assert sinks.tf_path() == "/x/out/tensorboard"
    monkeypatch.delenv("OUT_DIR")
    assert sinks.tf_path() == "/x/crate/target/tensorboard"
    monkeypatch.delenv("CARGO_MANIFEST_DIR")
    assert sinks.tf_path() == "/home/target/tensorboard"


def _grown_engines(orc, n=8, B=3, steps=60):
    ce = orc.Engine(n, B, threads=1)
    pe = po.PyEngine(n, B)
    parents, permitted = orc.gen_roots(5, 0, 0, B, n, 3, 9)
    ce.new_begin(parents, permitted)
    pe.new_begin(unpack_roots(parents, permitted, ce.A))
    h = orc.hash_predictions(5, 0, B, ce.A, 0)
    ce.new_end(h)
    pe.new_end(h)
    for call in range(1, steps + 1):
        ce.rollout_begin([6, 3], 2)
        pe.rollout_begin([6, 3], 2)
        h = orc.hash_predictions(5, 0, B, ce.A, call)
        ce.rollout_end(h)
        pe.rollout_end(h)
    return ce, pe


def test_sizes_and_dot_follow_the_tree(orc):
    ce, pe = _grown_engines(orc)
    for i in range(ce.B):
        t = ce.export_tree(i)
        pt = pe.trees[i]
        # sizes(): recount from the Python restatement's `pos` map (tree/mod.rs:51-68)
        want = [(0, 0)]
        for key, u in pt.pos.items():
            if u == 0:
                continue
            ln = len(key)
            if ln >= len(want):
                want.extend([(0, 0)] * (ln + 1 - len(want)))
            want[ln] = (want[ln][0] + 1, want[ln][1] + (1 if pt.active(u) else 0))
        got = sinks.sizes(t)
        assert got == want
        assert sum(a for a, _ in got) == len(t.c) - 1
        # graphviz statements: nodes in index order, arcs newest-first per source
        lab = ["s%dn%dx%d" % (u, nd["n"], nd["x"]) for u, nd in enumerate(pt.node)]
        lines = ["graph search_tree {"]
        for u in range(len(pt.node)):
            lines.append("  " + lab[u] + ("" if pt.active(u) else "[shape=doublecircle]"))
        for u in range(len(pt.node)):
            for e in reversed(pt.out[u]):
                v = pt.edge[e][1]
                lines.append("  %s -- %s" % (lab[u], lab[v]) + (" [dir=forward]" if pt.active(v) else ""))
        lines.append("}")
        assert sinks.tree_dot(t) == "\n".join(lines) + "\n"
