"""The C-ABI library loads without a GPU and exports every entry point include/azdopt_amd.h declares;
host-side logic of the boundary (dimensions, seeded root generators, argument checks) agrees with the
oracle; compute entry points fail loudly (no CPU fallback) when no gfx950 device is present."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "azdopt_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(azd_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    import azdopt_amd
    L = C.CDLL(azdopt_amd._lib.LIB_PATH)
    names = declared_symbols()
    assert len(names) > 50
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing


def test_python_binding_covers_the_header():
    import azdopt_amd
    L = azdopt_amd.lib()  # sets argtypes for every bound function; raises on an undefined symbol
    assert L.azd_version() >= 100
    assert isinstance(L.azd_device_count(), int)


def test_dimensions_match_the_oracle(orc):
    import azdopt_amd as az
    L = az.lib()
    for n in range(4, 25):
        assert L.azd_c21_state_dim(n) == orc.lib().orc_state_dim(n)
        assert L.azd_c21_action_dim(n) == orc.lib().orc_action_dim(n)
        assert L.azd_c21_key_words(n) == orc.lib().orc_key_words(n)
    for n, c in ((16, 3), (17, 2), (6, 2), (10, 4)):
        e = orc.Engine(n, 1, ramsey=([3] * c, [1.0] * c))
        assert (L.azd_ramsey_state_dim(n, c), L.azd_ramsey_action_dim(n, c), L.azd_ramsey_key_words(n, c)) == (e.S, e.A, e.KW)


def test_seeded_root_generators_match_the_oracle(orc):
    import azdopt_amd as az
    for n, kmin, kmax in ((5, 1, 2), (8, 2, 9), (19, 5, 76), (22, 5, 100)):
        sp = az.ROTModifyParentsOnce(n)
        for seed, epoch, first in ((0, 0, 0), (3, 2, 1000)):
            p, m = sp.generate_roots(seed, 33, first_agent=first, epoch=epoch, kmin=kmin, kmax=kmax)
            po, mo = orc.gen_roots(seed, epoch, first, 33, n, kmin, kmax)
            assert np.array_equal(p, po) and np.array_equal(m, mo)
    for n, sizes, kmin, kmax in ((6, [3, 3], 3, 7), (16, [3, 3, 3], 10, 60), (17, [4, 4], 12, 68)):
        sp = az.RamseySpaceNoEdgeRecolor(n, sizes)
        for seed, epoch, first in ((0, 0, 0), (5, 1, 77)):
            c, m = sp.generate_roots(seed, 21, first_agent=first, epoch=epoch, kmin=kmin, kmax=kmax)
            co, mo = orc.gen_ramsey_roots(seed, epoch, first, 21, n, len(sizes), kmin, kmax)
            assert np.array_equal(c, co) and np.array_equal(m, mo)
            assert c.max() < len(sizes)
            assert all(kmin <= sum(bin(int(w)).count("1") for w in row) <= kmax for row in m)


def test_no_cpu_fallback_without_a_device():
    import azdopt_amd as az
    if az.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(az.AzdError) as ei:
        az.HashStreamModel(10, 5, 0, 0)
    assert "device" in str(ei.value).lower()
    sp = az.ROTModifyParentsOnce(8)
    with pytest.raises(az.AzdError):
        az.NablaOptimizer(sp, None, 4)


def test_engine_config_validation_needs_no_device():
    """argument checks come before the device check: bad configurations are INVALID_ARGUMENT everywhere"""
    import ctypes as C
    import azdopt_amd as az
    from azdopt_amd import _lib
    L = az.lib()

    def create(**kw):
        cfg = _lib.EngineConfig()
        cfg.space_id, cfg.n, cfg.batch = _lib.SPACE_C21, 19, 8
        for k, v in kw.items():
            if isinstance(v, (list, tuple)):
                arr = getattr(cfg, k)
                for i, x in enumerate(v):
                    arr[i] = x
            else:
                setattr(cfg, k, v)
        h = C.c_void_p()
        st = L.azd_engine_create(C.byref(h), C.byref(cfg), None)
        if st == 0:
            L.azd_engine_destroy(h)
        return st

    INVALID = 1
    assert L.azd_status_string(INVALID).decode() == "invalid argument"
    ok = (0, 2)  # created, or "no gfx950 device" on a CPU-only box
    assert create() in ok
    for bad in (dict(n=3), dict(n=25), dict(batch=0), dict(space_id=7), dict(path_kind=2), dict(layers=9), dict(layers=-1)):
        assert create(**bad) == INVALID, bad
    ramsey = dict(space_id=_lib.SPACE_RAMSEY, n=16, n_colors=3, clique_sizes=[3, 3, 3], color_weights=[1.0, 1.0, 1.0])
    assert create(**ramsey) in ok
    for bad in (dict(n_colors=1), dict(n_colors=5), dict(clique_sizes=[3, 6, 3]), dict(clique_sizes=[1, 3, 3]), dict(n=24),
                dict(n=2), dict(n=20, n_colors=4)):  # n=24: E = 276 > 256; n=20, C=4: E*C = 760 > 384
        assert create(**{**ramsey, **bad}) == INVALID, bad
    assert L.azd_engine_create(None, None, None) == INVALID
    assert L.azd_ramsey_generate_roots(0, 0, 0, 1, 16, 3, 5, 200, None, None) == INVALID


def test_cpp_host_header_compiles_and_has_no_cpu_fallback(tmp_path):
    """include/azdopt_amd.hpp (the compiled-host mirror of NablaOptimizer / ActionModel) compiles warning-free against the
    C ABI, links with the library, and its example driver fails loudly where there is no device"""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for src in (("tests", "cpp", "host_model.cpp"), ("examples", "ramsey.cpp"), ("examples", "c21_tree.cpp")):
        exe = tmp_path / src[-1][:-4]
        subprocess.run(["g++", "-O1", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(root, "include"),
                        os.path.join(root, *src), "-o", str(exe), "-L" + os.path.join(root, "azdopt_amd"), "-lazdopt_amd",
                        "-Wl,-rpath," + os.path.join(root, "azdopt_amd")], check=True, timeout=300)
    import azdopt_amd as az
    if az.device_count() > 0:
        pytest.skip("a GPU is present: tests/test_gpu_examples.py runs the driver")
    r = subprocess.run([str(exe), "1", "1", "16"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 1 and "no gfx950 device" in r.stderr, (r.returncode, r.stderr)


def test_tree_capacities_names_the_record_format_limits():
    """examples/c21_tree.py computed arc_capacity = 3 * episodes + 64, which azd_engine_create refuses above ~21.8 k episodes
    (round-4 advisor finding): the helper clamps what is an estimate (arcs) and refuses, naming the argument, what is a need."""
    import azdopt_amd as az
    from azdopt_amd.optimizer import MAX_ARC_CAPACITY, MAX_NODE_CAPACITY, MAX_PREDICTION_CAPACITY
    assert az.tree_capacities(800, 76) == dict(node_capacity=4096, arc_capacity=8192, prediction_capacity=(801 * 76 + 128))
    assert az.tree_capacities(3200, 68)["arc_capacity"] == 8 * 3200 + 64  # the r44 driver's epoch (02-r44.rs:126)
    big = az.tree_capacities(30000, 20)
    assert big["arc_capacity"] == MAX_ARC_CAPACITY and big["node_capacity"] == 60064 <= MAX_NODE_CAPACITY
    assert big["prediction_capacity"] <= MAX_PREDICTION_CAPACITY
    with pytest.raises(ValueError, match="node_capacity"):
        az.tree_capacities(40000, 8)
    with pytest.raises(ValueError, match="prediction_capacity"):
        az.tree_capacities(20000, 76)
    hdr = open(os.path.join(ROOT, "include", "azdopt_amd.h")).read()
    assert "#define AZD_MAX_NODE_CAPACITY 65536" in hdr and "#define AZD_MAX_ARC_CAPACITY 65535" in hdr
