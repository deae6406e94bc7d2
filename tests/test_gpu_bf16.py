"""bf16 weight storage of the evaluator (BASELINE config C: "bf16 storage / fp32 accumulate"):
weights and layer inputs rounded to bf16 (RNE), products exact, f32 accumulation, f32 biases/outputs.
Checked against a numpy restatement with the same rounding points; the search stays bit-exact
against the oracle when the oracle is fed the device's predictions."""
import numpy as np
import pytest

from test_gpu_ramsey import assert_tree_equal, az  # noqa: F401

pytestmark = pytest.mark.gpu
F = np.float32
# Tolerance of the bf16 mode: the f32 accumulation order of the device differs from the restatement's,
# and a hidden activation that lands next to a bf16 rounding boundary can round the other way (one bf16
# ulp = 2^-8 relative on that one activation), which moves an output by ~1e-4 at most; everything else
# agrees to f32 accumulation noise.
BF16_ATOL = 1e-3


def bf16_round(x):
    u = np.ascontiguousarray(x, F).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) >> 16
    return (u << 16).astype(np.uint32).view(F).reshape(np.shape(x))


def reference_forward(params, dims, x):
    """y = act(bf16(x) @ bf16(W)^T + b) per layer, f32 accumulation, hidden ReLU, Sigmoid head"""
    off, t = 0, np.asarray(x, F)
    for l in range(len(dims) - 1):
        W = params[off:off + dims[l] * dims[l + 1]].reshape(dims[l + 1], dims[l]); off += W.size
        b = params[off:off + dims[l + 1]]; off += b.size
        t = bf16_round(t).astype(np.float64) @ bf16_round(W).astype(np.float64).T  # exact products; f64 sum ~ exact
        t = (t + b).astype(F)
        t = np.maximum(t, 0) if l < len(dims) - 2 else (1.0 / (1.0 + np.exp(-t.astype(np.float64)))).astype(F)
    return t


# the last two: BASELINE config E's model (K = 3676 = 114.9 k tiles of 32, N = 2450 = 19.1 column tiles of 128: ragged
# on every side of the bf16 MFMA GEMM's 128 x 128 x 32 tile) and a batch that is not a multiple of the row tile
@pytest.mark.parametrize("dims,B", [((304, 256, 256, 256, 152), 512), ((304, 512, 1024, 512, 152), 96), ((840, 256, 128, 360), 64),
                                    ((3676, 512, 512, 512, 2450), 200), ((304, 256, 256, 256, 152), 1300)])
def test_bf16_gemm_path_matches_rounded_reference(az, dims, B):
    m = az.ActionModel(B, dims[0], dims[-1], hidden=dims[1:-1], seed=5, dtype="bf16")
    rng = np.random.default_rng(1)
    x = rng.integers(0, 6, (B, dims[0])).astype(F) * (rng.random((B, dims[0])) < 0.4)
    x[:, :7] += 0.3  # not bf16-representable: exercises the input rounding
    y = np.zeros((B, dims[-1]), F)
    m.write_predictions(x, y)
    want = reference_forward(m.get_params(), dims, x)
    assert np.max(np.abs(y - want)) < BF16_ATOL and np.mean(np.abs(y - want)) < 2e-6
    # against the f32 model the storage rounding is visible (so the option is really on) but small
    m32 = az.ActionModel(B, dims[0], dims[-1], hidden=dims[1:-1], seed=5)
    y32 = np.zeros_like(y)
    m32.write_predictions(x, y32)
    d = np.max(np.abs(y - y32))
    assert 1e-6 < d < 3e-2, d


def test_bf16_training_runs_on_f32_master_weights(az):
    dims, B = (304, 256, 256, 256, 152), 256
    rng = np.random.default_rng(2)
    x = (rng.random((B, dims[0])) < 0.3).astype(F)
    obs = rng.random((B, dims[-1])).astype(F)
    w = (rng.random((B, dims[-1])) < 0.2).astype(F)
    a = az.ActionModel(B, dims[0], dims[-1], hidden=dims[1:-1], seed=3, dtype="bf16")
    b = az.ActionModel(B, dims[0], dims[-1], hidden=dims[1:-1], seed=3)
    for _ in range(3):
        la, lb = a.update_model(x, obs, w), b.update_model(x, obs, w)
        assert la == lb  # identical f32 optimiser step
    assert np.array_equal(a.get_params(), b.get_params())
    y = np.zeros((B, dims[-1]), F)
    a.write_predictions(x, y)  # inference sees the refreshed bf16 copy of the updated weights
    assert np.max(np.abs(y - reference_forward(a.get_params(), dims, x))) < BF16_ATOL


@pytest.mark.parametrize("persistent", [True, False])
def test_bf16_search_parity_and_in_kernel_predictions(az, orc, persistent):
    n, B, seed = 19, 72, 4
    tol = ([200, 50, 50], 25)
    space = az.ROTModifyParentsOnce(n)
    dims = (space.STATE_DIM, 256, 256, 256, space.ACTION_DIM)
    model = az.ActionModel(B, dims[0], dims[-1], hidden=dims[1:-1], seed=seed, dtype="bf16")
    params = model.get_params()
    roots = space.generate_roots(seed, B)
    opt = az.NablaOptimizer.par_new(space, roots, model, B, persistent=persistent)
    oe = orc.Engine(n, B, threads=8)
    oe.new_begin(*roots)
    oe.new_end(opt.predictions())
    worst = 0.0
    for s in range(60):
        opt.par_roll_out_episodes(tol)
        oe.rollout_begin(*tol)
        sv = oe.state_vecs()
        assert np.array_equal(opt.state_vecs(), sv)
        h = opt.predictions()
        fresh = [i for i in range(B) if oe.agent_state(i)["path"].any()]
        if fresh:
            worst = max(worst, float(np.max(np.abs(h[fresh] - reference_forward(params, dims, sv[fresh])))))
        oe.rollout_end(h)
    assert worst < BF16_ATOL, worst
    for i in range(B):
        assert_tree_equal(opt.get_tree(i), oe.export_tree(i), f"agent {i}")
    assert np.isfinite(opt.par_update_model(5))


@pytest.mark.parametrize("B", [200, 77, 1])
def test_fused_hidden_layers_equal_the_layer_by_layer_forward(az, monkeypatch, B):
    """two 512-wide hidden layers in one launch (k_hidden2_fused: 32-row panels, activations in LDS, both weight matrices
    streamed through one ring) give every prediction the same bits as one k_gemm16 launch per layer; batches that are not
    multiples of the 32-row panel"""
    dims = (3676, 512, 512, 512, 2450)
    rng = np.random.default_rng(7)
    x = rng.integers(0, 3, (B, dims[0])).astype(F) * (rng.random((B, dims[0])) < 0.3)
    ys = []
    for fuse in ("1", "0"):
        monkeypatch.setenv("AZD_MLP_FUSE_HIDDEN", fuse)
        m = az.ActionModel(B, dims[0], dims[-1], hidden=dims[1:-1], seed=11, dtype="bf16")
        y = np.zeros((B, dims[-1]), F)
        m.write_predictions(x, y)
        ys.append(y)
    assert np.array_equal(ys[0].view(np.uint32), ys[1].view(np.uint32))
    assert np.max(np.abs(ys[0] - reference_forward(m.get_params(), dims, x))) < BF16_ATOL


@pytest.mark.parametrize("slices,form", [("4", None), ("2", "128"), ("8", "1064")])
def test_experiment_knobs_of_the_long_k_gemm_keep_the_forward_within_tolerance(az, monkeypatch, slices, form):
    """AZD_GEMM16_KSPLIT (the first layer's k loop dealt to 2 / 4 / 8 blocks per tile, partial sums added in slice order) and
    AZD_GEMM16_LONGK_FORM (another tile form): measured and left off -- but a knob that is documented runs:
    the forward stays within the bf16 mode's tolerance of the restatement, and two runs of the split give the same bits (the
    slices are added in a fixed order, whichever block finishes first)."""
    dims, B = (3676, 512, 512, 512, 2450), 200
    rng = np.random.default_rng(9)
    x = rng.integers(0, 3, (B, dims[0])).astype(F) * (rng.random((B, dims[0])) < 0.3)
    monkeypatch.setenv("AZD_GEMM16_KSPLIT", slices)
    if form:
        monkeypatch.setenv("AZD_GEMM16_LONGK_FORM", form)
    m = az.ActionModel(B, dims[0], dims[-1], hidden=dims[1:-1], seed=11, dtype="bf16")
    ys = []
    for _ in range(2):
        y = np.zeros((B, dims[-1]), F)
        m.write_predictions(x, y)
        ys.append(y)
    assert np.array_equal(ys[0].view(np.uint32), ys[1].view(np.uint32))
    assert np.max(np.abs(ys[0] - reference_forward(m.get_params(), dims, x))) < BF16_ATOL
