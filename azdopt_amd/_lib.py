"""ctypes loader of libazdopt_amd.so (the C ABI declared in include/azdopt_amd.h).

There is no Python or CPU fallback: if the HIP library is missing or no gfx950
device is visible, the product fails loudly."""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("AZD_LIB", os.path.join(_HERE, "libazdopt_amd.so"))
_LIB = None

CTR = dict(EXPANSIONS=0, TERMINALS=1, TRANSPOSITIONS=2, VISITED_STEPS=3, SELECT_CALLS=4, SUM_DEG=5,
           SUM_ACTIONS=6, CASCADE_NODES=7, NEW_PREDS=8, ROOT_EXHAUSTED=9, MAX_FRONTIER=10, MAX_DEPTH=11,
           CURIOSITY_PAIRS=12, EVAL_LAYER_CLOCKS=13, TICKS_ADD_ACTIONS=14, FAILED=15, TICKS_TOTAL=16, TICKS_SELECT=17, TICKS_LOOKUP=18, TICKS_NEWNODE=19,
           TICKS_CASCADE=20, TICKS_MAX_CALL=21, TICKS_LAMBDA=22, TICKS_MATCHING=23, TICKS_WAIT=24, EVAL_BATCHES=25, EVAL_ROWS=26, EVAL_TILES=27, TICKS_TILES=28, TICKS_BATCH=29, TICKS_TILE_SETUP=30, TICKS_TILE_KLOOP=31)
CTR_COUNT = 32

ACT_NONE, ACT_RELU, ACT_SIGMOID = 0, 1, 2
ENGINE_NO_PERSISTENT_STEP = 1
ENGINE_ASYNC_STEP = 2
ENGINE_BARRIER_STEP = 4
ENGINE_POOL_STEP = 8
SPACE_C21 = 1
SPACE_RAMSEY = 2
SPACE_DENSE = 3
PATH_SET, PATH_SEQUENCE = 0, 1


class AzdError(RuntimeError):
    def __init__(self, status, where):
        L = lib()
        msg = L.azd_status_string(status).decode()
        detail = L.azd_last_error().decode()
        super().__init__(f"{where}: {msg} (status {status}){': ' + detail if detail else ''}")
        self.status = status


class AdamConfig(C.Structure):  # dfdx AdamConfig as set at 04-c21-tree.rs:87-92
    _fields_ = [("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float), ("l2", C.c_float)]


class EngineConfig(C.Structure):
    _fields_ = [("space_id", C.c_int), ("n", C.c_int), ("batch", C.c_int), ("device", C.c_int),
                ("node_capacity", C.c_int), ("arc_capacity", C.c_int), ("prediction_capacity", C.c_int),
                ("first_agent", C.c_uint64), ("flags", C.c_uint32),
                ("n_colors", C.c_int), ("clique_sizes", C.c_int * 4), ("color_weights", C.c_float * 4),
                ("path_kind", C.c_int), ("layers", C.c_int), ("max_slots", C.c_int), ("dense_p", C.c_float)]


class RamseyArgmin(C.Structure):  # ArgminData<RamseyCountsNoRecolor, TotalCounts<C>>
    _fields_ = [("colors", C.c_uint8 * 256), ("permitted", C.c_uint64 * 4), ("totals", C.c_int32 * 4),
                ("eval", C.c_float), ("agent", C.c_int32), ("node", C.c_uint32)]


class DenseArgmin(C.Structure):  # ArgminData of the dense-graph space
    _fields_ = [("adj", C.c_uint64 * 64), ("permitted", C.c_uint64 * 40), ("lambda_1", C.c_double),
                ("matching_size", C.c_int32), ("eval", C.c_float), ("agent", C.c_int32), ("node", C.c_uint32)]


class Argmin(C.Structure):  # ArgminData<State, Cost>, az-discrete-opt/src/log.rs:1-11
    _fields_ = [("parents", C.c_uint8 * 32), ("permitted", C.c_uint64 * 4), ("lambda_1", C.c_double),
                ("matching_size", C.c_int32), ("matching", C.c_int32 * 32), ("eval", C.c_float),
                ("agent", C.c_int32), ("node", C.c_uint32)]


def build(force=False):
    """Compile the HIP library for gfx950 (cross-compiles without a GPU)."""
    if force or not os.path.exists(LIB_PATH):
        subprocess.check_call(["make", "-s", "-j4", "-C", os.path.join(_HERE, "csrc")])
    else:
        subprocess.check_call(["make", "-s", "-j4", "-C", os.path.join(_HERE, "csrc")])
    return LIB_PATH


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: run __graft_entry__.build() (hipcc --offload-arch=gfx950); "
                          "azdopt_amd has no CPU fallback")
    L = C.CDLL(LIB_PATH)
    vp, i32p, u64p, f32p = C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_uint64), C.POINTER(C.c_float)

    def sig(name, res, *args):
        f = getattr(L, name, None)
        if f is None:
            if "AZD_LIB" in os.environ:  # an older / experiment build selected by hand may lack the newer entry points
                return
            raise AttributeError("%s does not export %s" % (LIB_PATH, name))
        f.restype = res
        f.argtypes = list(args)

    sig("azd_status_string", C.c_char_p, C.c_int)
    sig("azd_last_error", C.c_char_p)
    sig("azd_version", C.c_int)
    sig("azd_device_count", C.c_int)
    sig("azd_c21_state_dim", C.c_int, C.c_int)
    sig("azd_c21_action_dim", C.c_int, C.c_int)
    sig("azd_c21_key_words", C.c_int, C.c_int)
    sig("azd_c21_generate_roots", C.c_int, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp)
    sig("azd_evaluator_create_mlp", C.c_int, C.POINTER(vp), C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_int, C.c_int,
        C.POINTER(AdamConfig), C.c_uint64)
    sig("azd_evaluator_create_trivial", C.c_int, C.POINTER(vp), C.c_int, C.c_int, C.c_int)
    sig("azd_evaluator_create_hash_stream", C.c_int, C.POINTER(vp), C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_uint64)
    sig("azd_evaluator_destroy", C.c_int, vp)
    sig("azd_evaluator_write_predictions", C.c_int, vp, C.c_int, vp, vp)
    sig("azd_evaluator_update_model", C.c_int, vp, C.c_int, vp, vp, vp, f32p)
    sig("azd_evaluator_write_predictions_dev", C.c_int, vp, C.c_int, vp, vp, vp)
    sig("azd_evaluator_update_model_dev", C.c_int, vp, C.c_int, vp, vp, vp, f32p, vp)
    sig("azd_evaluator_set_weight_storage", C.c_int, vp, C.c_int)
    sig("azd_evaluator_num_params", C.c_int64, vp)
    sig("azd_evaluator_get_params", C.c_int, vp, vp)
    sig("azd_evaluator_set_params", C.c_int, vp, vp)
    sig("azd_evaluator_calls", C.c_uint64, vp)
    sig("azd_engine_create", C.c_int, C.POINTER(vp), C.POINTER(EngineConfig), vp)
    sig("azd_engine_destroy", C.c_int, vp)
    sig("azd_engine_par_new", C.c_int, vp, vp, vp)
    sig("azd_engine_par_roll_out_episodes", C.c_int, vp, vp, C.c_int, C.c_uint32, C.c_int, i32p)
    sig("azd_engine_run_ahead", C.c_int, vp, vp, C.c_int, C.c_uint32, C.c_int, i32p)
    sig("azd_engine_par_update_model", C.c_int, vp, C.c_uint32, f32p)
    sig("azd_engine_par_update_model_sharded", C.c_int, vp, C.c_uint32, vp, f32p)
    sig("azd_engine_par_reset_trees", C.c_int, vp, vp, vp)
    sig("azd_engine_argmin_data", C.c_int, vp, C.POINTER(Argmin))
    sig("azd_engine_agent_counters", C.c_int, vp, vp)
    sig("azd_engine_agent_counters_per_agent", C.c_int, vp)
    sig("azd_engine_pool_agent_finish", C.c_int, vp, vp)
    sig("azd_engine_pool_groups", C.c_int, vp, vp, vp, vp)
    sig("azd_engine_ramsey_argmin_data", C.c_int, vp, C.POINTER(RamseyArgmin))
    sig("azd_engine_ramsey_agent_counts", C.c_int, vp, C.c_int, vp, vp)
    sig("azd_ramsey_state_dim", C.c_int, C.c_int, C.c_int)
    sig("azd_ramsey_action_dim", C.c_int, C.c_int, C.c_int)
    sig("azd_ramsey_key_words", C.c_int, C.c_int, C.c_int)
    sig("azd_ramsey_generate_roots", C.c_int, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_int,
        C.c_int, vp, vp)
    sig("azd_dense_state_dim", C.c_int, C.c_int)
    sig("azd_dense_action_dim", C.c_int, C.c_int)
    sig("azd_dense_key_words", C.c_int, C.c_int)
    sig("azd_dense_generate_roots", C.c_int, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, vp, vp)
    sig("azd_engine_dense_argmin_data", C.c_int, vp, C.POINTER(DenseArgmin))
    sig("azd_c21_modify_roots", C.c_int, vp, C.c_uint64, C.c_uint64, C.c_int, C.c_int, vp, vp)
    sig("azd_c21_modify_roots_dev", C.c_int, vp, C.c_uint64, C.c_uint64, C.c_int, C.c_int, vp, vp)
    sig("azd_engine_par_reset_trees_c21", C.c_int, vp, C.c_uint64, C.c_uint64, C.c_int, C.c_int)
    sig("azd_engine_modify_roots_dev", C.c_int, vp, C.c_uint64, C.c_uint64, C.c_int, C.c_int, vp, vp)
    sig("azd_engine_par_reset_trees_policy", C.c_int, vp, C.c_uint64, C.c_uint64, C.c_int, C.c_int)
    sig("azd_engine_par_new_begin", C.c_int, vp, vp, vp)
    sig("azd_engine_par_new_end", C.c_int, vp, vp)
    sig("azd_engine_roll_out_begin", C.c_int, vp, vp, C.c_int, C.c_uint32)
    sig("azd_engine_roll_out_end", C.c_int, vp, vp, i32p)
    sig("azd_engine_reset_begin", C.c_int, vp, vp, vp)
    sig("azd_engine_reset_end", C.c_int, vp, vp)
    sig("azd_engine_observe", C.c_int, vp, C.c_uint32, vp, vp, vp)
    sig("azd_engine_observe_dev", C.c_int, vp, C.c_uint32, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp))
    sig("azd_engine_read_state_vecs", C.c_int, vp, vp)
    sig("azd_engine_read_predictions", C.c_int, vp, vp)
    sig("azd_engine_debug_tile_forward", C.c_int, vp, vp, vp, C.c_int)
    sig("azd_engine_tree_sizes", C.c_int, vp, C.c_int, i32p, i32p, i32p)
    sig("azd_engine_export_tree", C.c_int, vp, C.c_int, *([vp] * 13))
    sig("azd_engine_agent_state", C.c_int, vp, C.c_int, vp, vp, vp, C.POINTER(C.c_uint32), C.POINTER(C.c_double), i32p)
    sig("azd_engine_counters", C.c_int, vp, vp)
    sig("azd_engine_set_timing", C.c_int, vp, C.c_int)
    sig("azd_engine_timing", C.c_int, vp, C.POINTER(C.c_double), C.POINTER(C.c_double), u64p)
    sig("azd_engine_stream", vp, vp)
    sig("azd_engine_step_form", C.c_int, vp, i32p, C.POINTER(C.c_char_p))
    sig("azd_engine_pool_split", C.c_int, vp, i32p, i32p)
    sig("azd_engine_pool_utilisation", C.c_int, vp, C.POINTER(C.c_double), C.POINTER(C.c_double))
    sig("azd_debug_probe_xcc", C.c_int, C.c_int, vp, C.c_int)
    sig("azd_debug_hash_stream_via_evaluators", C.c_int, vp, C.c_int)
    sig("azd_debug_probe_math", C.c_int, C.c_int, vp, vp, C.c_int)
    sig("azd_debug_gemm_bf16", C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, f32p)
    sig("azd_debug_probe_cost", C.c_int, C.c_int, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, f32p)
    _LIB = L
    return L


def check(status, where):
    if status != 0:
        raise AzdError(status, where)


def ptr(a):
    return a.ctypes.data_as(C.c_void_p)
