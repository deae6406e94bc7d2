"""Observability sinks: the outputs a user of the reference sees.

* `tf_path`            - az-discrete-opt/src/tensorboard/mod.rs:5-16
* `TensorboardWriter`  - the `tensorboard-writer` crate as the examples drive it
                         (graph-state/examples/04-c21-tree.rs:77-83,121-124,150-153):
                         write_file_version, write_summary(wall_time, step, summary)
* `loss_summary`       - az-discrete-opt/src/nabla/model/mod.rs:25-32
* `c21_cost_summary`   - graph-state/src/simple_graph/connected_bitset_graph/mod.rs:355-365
* `clique_counts_summary` - graph-state/src/ramsey_counts/mod.rs:195-205
* `tree_dot`           - az-discrete-opt/src/nabla/tree/graphviz.rs:9-90 (statement text; the
                         reference pipes the same statements through `dot -Tpng`)
* `sizes`              - az-discrete-opt/src/nabla/tree/mod.rs:51-68

Everything here is host-side byte formatting over arrays the engine exports; nothing touches
the device.  The event file framing is TFRecord (length, masked crc32c, payload, masked crc32c)
around `tensorflow.Event` protobuf messages, written with a hand-rolled encoder (no protobuf
dependency).
"""
import os
import shutil
import struct
import subprocess
import time

import numpy as np


def tf_path():
    """$OUT_DIR, else $CARGO_MANIFEST_DIR/target, else /home/target; + "tensorboard"."""
    out = os.environ.get("OUT_DIR")
    if out is None:
        manifest = os.environ.get("CARGO_MANIFEST_DIR")
        out = os.path.join(manifest, "target") if manifest is not None else "/home/target"
    return os.path.join(out, "tensorboard")


# ---------------------------------------------------------------- crc32c (Castagnoli), masked
def _crc_table():
    t = []
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
        t.append(c)
    return t


_CRC = _crc_table()


def crc32c(data):
    c = 0xFFFFFFFF
    for b in data:
        c = _CRC[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def masked_crc32c(data):
    c = crc32c(data)
    return ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


# ---------------------------------------------------------------- protobuf wire encoding
def _varint(v):
    v &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _key(field, wire):
    return _varint(field << 3 | wire)


def _bytes_field(field, payload):
    return _key(field, 2) + _varint(len(payload)) + payload


class Summary:
    """tensorflow.Summary: repeated Value{tag = 1, simple_value = 2 (float)}."""

    def __init__(self):
        self.values = []

    def scalar(self, tag, value):
        self.values.append((str(tag), float(np.float32(value))))
        return self

    def build(self):
        return self

    def encode(self):
        out = b""
        for tag, v in self.values:
            val = _bytes_field(1, tag.encode()) + _key(2, 5) + struct.pack("<f", v)
            out += _bytes_field(1, val)
        return out


def SummaryBuilder():
    return Summary()


def loss_summary(loss):
    return Summary().scalar("loss", loss)


def c21_cost_summary(lambda_1, matching_size):
    """tags and values of Conjecture2Dot1Cost::summary: the sum is taken in f64, then narrowed."""
    cost = float(lambda_1) + float(matching_size)
    return Summary().scalar("cost/cost", cost).scalar("cost/lambda_1", lambda_1).scalar("cost/mu", matching_size)


def clique_counts_summary(counts):
    s = Summary()
    for c, n in enumerate(counts):
        s.scalar("clique_counts/%d" % c, n)
    return s


def argmin_summary(argmin):
    """summary of an optimizer.ArgminData (c21 cost)."""
    return c21_cost_summary(argmin.cost["lambda_1"], len(argmin.cost["matching"]))


class TensorboardWriter:
    """Event-file writer over any binary file object."""

    def __init__(self, fileobj):
        self._f = fileobj

    @classmethod
    def create(cls, example, stamp=None, name="tfevents-losses"):
        """<tf_path()>/<example>/<rfc3339 stamp>/tfevents-losses, as the examples lay it out."""
        stamp = stamp or time.strftime("%Y-%m-%dT%H:%M:%S+00:00", time.gmtime())
        d = os.path.join(tf_path(), example, stamp)
        os.makedirs(d, exist_ok=True)
        w = cls(open(os.path.join(d, name), "wb"))
        w.write_file_version()
        return w

    def get_mut(self):
        return self._f

    def _record(self, event):
        head = struct.pack("<Q", len(event))
        self._f.write(head + struct.pack("<I", masked_crc32c(head)) + event + struct.pack("<I", masked_crc32c(event)))

    @staticmethod
    def _event(wall_time, step, tail):
        ev = _key(1, 1) + struct.pack("<d", float(wall_time))
        if step:
            ev += _key(2, 0) + _varint(int(step))
        return ev + tail

    def write_file_version(self, wall_time=None):
        wall_time = time.time() if wall_time is None else wall_time
        self._record(self._event(wall_time, 0, _bytes_field(3, b"brain.Event:2")))

    def write_summary(self, wall_time, step, summary):
        wall_time = time.time() if wall_time is None else wall_time
        self._record(self._event(wall_time, step, _bytes_field(5, summary.encode())))

    def flush(self):
        self._f.flush()

    def close(self):
        self._f.close()


def read_events(path):
    """Decode an event file back to [(wall_time, step, file_version | None, [(tag, value)])];
    verifies both checksums of every record (used by the tests and for inspection)."""
    out = []
    with open(path, "rb") as f:
        blob = f.read()
    pos = 0
    while pos < len(blob):
        head = blob[pos:pos + 8]
        (n,) = struct.unpack("<Q", head)
        (hc,) = struct.unpack("<I", blob[pos + 8:pos + 12])
        body = blob[pos + 12:pos + 12 + n]
        (bc,) = struct.unpack("<I", blob[pos + 12 + n:pos + 16 + n])
        if hc != masked_crc32c(head) or bc != masked_crc32c(body):
            raise ValueError("event file checksum mismatch at byte %d" % pos)
        pos += 16 + n
        out.append(_decode_event(body))
    return out


def _fields(buf):
    pos = 0
    while pos < len(buf):
        k, pos = _read_varint(buf, pos)
        field, wire = k >> 3, k & 7
        if wire == 0:
            v, pos = _read_varint(buf, pos)
        elif wire == 1:
            v, pos = buf[pos:pos + 8], pos + 8
        elif wire == 5:
            v, pos = buf[pos:pos + 4], pos + 4
        elif wire == 2:
            n, pos = _read_varint(buf, pos)
            v, pos = buf[pos:pos + n], pos + n
        else:
            raise ValueError("unsupported wire type %d" % wire)
        yield field, wire, v


def _read_varint(buf, pos):
    v = shift = 0
    while True:
        b = buf[pos]
        pos += 1
        v |= (b & 0x7F) << shift
        shift += 7
        if not b & 0x80:
            return v, pos


def _decode_event(body):
    wall, step, version, scalars = 0.0, 0, None, []
    for field, _, v in _fields(body):
        if field == 1:
            (wall,) = struct.unpack("<d", v)
        elif field == 2:
            step = v
        elif field == 3:
            version = bytes(v).decode()
        elif field == 5:
            for f2, _, val in _fields(v):
                if f2 != 1:
                    continue
                tag, x = None, None
                for f3, _, y in _fields(val):
                    if f3 == 1:
                        tag = bytes(y).decode()
                    elif f3 == 2:
                        (x,) = struct.unpack("<f", y)
                scalars.append((tag, x))
    return wall, step, version, scalars


# ---------------------------------------------------------------- tree dumps
def _active(t):
    """state_weight.rs:31-33"""
    return t.act_begin + t.exhausted < t.act_end


def _out_neighbours(t):
    """children per node in petgraph's iteration order: newest arc first."""
    kids = [[] for _ in range(len(t.c))]
    for e in range(len(t.e_src) - 1, -1, -1):
        kids[int(t.e_src[e])].append(int(t.e_dst[e]))
    return kids


def tree_dot(t):
    """The statements SearchTree::graphviz builds, as DOT text: one node per tree node labelled
    s<index>n<n_t>x<exhausted_children> (inactive: shape=doublecircle), one edge per arc in
    out-neighbour order (active target: dir=forward)."""
    act = _active(t)
    label = ["s%dn%dx%d" % (u, int(t.n_t[u]), int(t.exhausted[u])) for u in range(len(t.c))]
    lines = ["graph search_tree {"]
    for u, name in enumerate(label):
        lines.append("  %s" % name if act[u] else "  %s[shape=doublecircle]" % name)
    for u, kids in enumerate(_out_neighbours(t)):
        for v in kids:
            lines.append("  %s -- %s" % (label[u], label[v]) + (" [dir=forward]" if act[v] else ""))
    lines.append("}")
    return "\n".join(lines) + "\n"


def tree_png(t):
    """graphviz.rs:88: render through the `dot` executable (raises if it is not installed)."""
    exe = shutil.which("dot")
    if exe is None:
        raise FileNotFoundError("graphviz `dot` is not on PATH; use tree_dot() for the statement text")
    return subprocess.run([exe, "-Tpng"], input=tree_dot(t).encode(), stdout=subprocess.PIPE, check=True).stdout


def sizes(t):
    """SearchTree::sizes: per path length, (nodes, active nodes) over the transposition table.
    The root's empty path is not a key of `positions`, so row 0 stays (0, 0)."""
    act = _active(t)
    keys = np.asarray(t.keys, np.uint64).reshape(len(t.c), -1)
    out = [(0, 0)]
    for u in range(1, len(t.c)):
        ln = sum(bin(int(w)).count("1") for w in keys[u])
        if ln >= len(out):
            out.extend([(0, 0)] * (ln + 1 - len(out)))
        out[ln] = (out[ln][0] + 1, out[ln][1] + (1 if act[u] else 0))
    return out
