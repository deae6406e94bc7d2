"""The CPU baseline's blocked / vectorised MLP forward equals the scalar restatement bit for bit."""
import time

import numpy as np
import pytest


@pytest.mark.parametrize("dims,B", [((304, 256, 256, 256, 152), 1000), ((88, 48, 32, 44), 37), ((304, 512, 1024, 512, 152), 130)])
def test_forward_fast_is_bit_identical(orc, dims, B):
    rng = np.random.default_rng(1)
    x = (rng.random((B, dims[0])) < 0.3).astype(np.float32)
    m = orc.Mlp(dims, seed=3, threads=4)
    a = m.forward(x)
    b = m.forward_fast(x)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_forward_fast_is_faster(orc):
    dims, B = (304, 256, 256, 256, 152), 4096
    x = (np.random.default_rng(0).random((B, dims[0])) < 0.3).astype(np.float32)
    m = orc.Mlp(dims, seed=0, threads=4)
    m.forward_fast(x)
    if not m.vectorised:
        pytest.skip("no AVX2 on this host")
    t0 = time.perf_counter(); m.forward(x); t1 = time.perf_counter(); m.forward_fast(x); t2 = time.perf_counter()
    assert (t2 - t1) < (t1 - t0)
