"""Evaluators behind the NablaModel seam (az-discrete-opt/src/nabla/model/mod.rs:4-8)."""
import ctypes as C

import numpy as np

from . import _lib


class NablaModel:
    """write_predictions(states, predictions) / update_model(states, observations, action_weights) -> loss
    over flat row-major f32 arrays, exactly the trait's slices."""

    def __init__(self):
        self._h = C.c_void_p()
        self.state_dim = self.action_dim = 0

    def __del__(self):
        if getattr(self, "_h", None) and self._h.value and _lib is not None and getattr(_lib, "lib", None):
            _lib.lib().azd_evaluator_destroy(self._h)  # (module globals may already be gone at interpreter exit)
            self._h = C.c_void_p()

    def write_predictions(self, states, predictions):
        states = np.ascontiguousarray(states, np.float32).reshape(-1, self.state_dim)
        assert predictions.dtype == np.float32 and predictions.flags.c_contiguous
        assert predictions.size == states.shape[0] * self.action_dim
        _lib.check(_lib.lib().azd_evaluator_write_predictions(self._h, states.shape[0], _lib.ptr(states),
                                                              _lib.ptr(predictions)), "write_predictions")

    def update_model(self, states, observations, action_weights):
        states = np.ascontiguousarray(states, np.float32).reshape(-1, self.state_dim)
        obs = np.ascontiguousarray(observations, np.float32)
        w = np.ascontiguousarray(action_weights, np.float32)
        loss = C.c_float()
        _lib.check(_lib.lib().azd_evaluator_update_model(self._h, states.shape[0], _lib.ptr(states), _lib.ptr(obs),
                                                         _lib.ptr(w), C.byref(loss)), "update_model")
        return loss.value

    # device-pointer forms (multi-GPU host code calls these after its all-gather)
    def write_predictions_dev(self, batch, d_states, d_predictions, stream=None):
        _lib.check(_lib.lib().azd_evaluator_write_predictions_dev(self._h, batch, d_states, d_predictions, stream),
                   "write_predictions_dev")

    def update_model_dev(self, batch, d_states, d_observations, d_weights, stream=None):
        loss = C.c_float()
        _lib.check(_lib.lib().azd_evaluator_update_model_dev(self._h, batch, d_states, d_observations, d_weights,
                                                             C.byref(loss), stream), "update_model_dev")
        return loss.value

    def calls(self):
        return _lib.lib().azd_evaluator_calls(self._h)


class TrivialModel(NablaModel):
    """model/mod.rs:10-23: predictions untouched, loss 0."""

    def __init__(self, state_dim, action_dim, device=0):
        super().__init__()
        self.state_dim, self.action_dim = state_dim, action_dim
        _lib.check(_lib.lib().azd_evaluator_create_trivial(C.byref(self._h), device, state_dim, action_dim),
                   "azd_evaluator_create_trivial")


class HashStreamModel(NablaModel):
    """Fixed prediction stream h(agent, call, a): the parity harness' model stand-in."""

    def __init__(self, state_dim, action_dim, seed, first_agent=0, device=0):
        super().__init__()
        self.state_dim, self.action_dim = state_dim, action_dim
        _lib.check(_lib.lib().azd_evaluator_create_hash_stream(C.byref(self._h), device, state_dim, action_dim, seed,
                                                               first_agent), "azd_evaluator_create_hash_stream")

    def serve_from_pool_evaluators(self, on=True):
        """test harness: the pool step's evaluator workgroups serve the rows (queues, early post, join) instead of the
        searching wave itself; same values"""
        _lib.check(_lib.lib().azd_debug_hash_stream_via_evaluators(self._h, int(on)), "azd_debug_hash_stream_via_evaluators")
        return self


class ActionModel(NablaModel):
    """ActionModel<M, BATCH, STATE, ACTION> (model/dfdx.rs:18-53): fp32 MLP
    STATE -> hidden... -> ACTION, ReLU between layers, `final_act` on the head, Adam with L2."""

    def __init__(self, batch, state_dim, action_dim, hidden=(512, 1024, 512), final_act=_lib.ACT_SIGMOID,
                 lr=1e-4, betas=(0.9, 0.999), eps=1e-8, l2=1e-6, seed=0, device=0, dtype="f32"):
        super().__init__()
        self.state_dim, self.action_dim, self.hidden = state_dim, action_dim, tuple(hidden)
        cfg = _lib.AdamConfig(lr, betas[0], betas[1], eps, l2)
        hid = np.asarray(hidden, np.int32)
        _lib.check(_lib.lib().azd_evaluator_create_mlp(C.byref(self._h), device, batch, state_dim, action_dim,
                                                       _lib.ptr(hid), len(hidden), final_act, C.byref(cfg), seed),
                   "azd_evaluator_create_mlp")
        self.dtype = dtype
        if dtype == "bf16":  # bf16 weight/activation storage for inference, f32 accumulate, f32 master weights
            _lib.check(_lib.lib().azd_evaluator_set_weight_storage(self._h, 1), "azd_evaluator_set_weight_storage")
        elif dtype != "f32":
            raise ValueError("dtype must be 'f32' or 'bf16'")

    def num_params(self):
        return _lib.lib().azd_evaluator_num_params(self._h)

    def get_params(self):
        out = np.zeros(self.num_params(), np.float32)
        _lib.check(_lib.lib().azd_evaluator_get_params(self._h, _lib.ptr(out)), "get_params")
        return out

    def set_params(self, p):
        p = np.ascontiguousarray(p, np.float32)
        assert p.size == self.num_params()
        _lib.check(_lib.lib().azd_evaluator_set_params(self._h, _lib.ptr(p)), "set_params")
