"""lambda_1 / matching kernel in isolation: parity vs the oracle and time per call."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

import azdopt_amd as az
from azdopt_amd import _lib
from oracle import orc

n = int(sys.argv[1]) if len(sys.argv) > 1 else 19
count = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
rng = np.random.default_rng(0)
parents = np.zeros((count, n), np.uint8)
for v in range(2, n - 1):
    parents[:, v] = rng.integers(0, v, size=count)
L = az.lib()
OL = orc.lib()
for full in (0, 1):
    lam = np.zeros(count, np.float64)
    mu = np.zeros(count, np.int32)
    ms = C.c_float()
    _lib.check(L.azd_debug_probe_cost(0, _lib.ptr(parents), n, count, reps, full, _lib.ptr(lam), _lib.ptr(mu), C.byref(ms)), "probe_cost")
    f = OL.orc_lambda1_sturm if full else OL.orc_lambda1_node
    bad = 0
    for i in range(min(count, 512)):
        p = np.ascontiguousarray(parents[i])
        bad += int(f(p.ctypes.data_as(C.c_void_p), n) != lam[i]) + int(OL.orc_maximum_matching(p.ctypes.data_as(C.c_void_p), n, None) != mu[i])
    print(f"n={n} count={count} full={full}: {ms.value * 1e3 / reps:.1f} us per (lambda1+matching) pass over all trees, mismatches vs oracle: {bad}")
