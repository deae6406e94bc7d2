// evaluator.h -- device-side NablaModel (az-discrete-opt/src/nabla/model/mod.rs:4-8)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>

#include "../../include/azdopt_amd.h"
#include "engine_types.h"

namespace azd {

extern thread_local std::string g_last_error;
int hip_fail(hipError_t e, const char *what);

#define AZD_HIP(call)                                        \
    do {                                                     \
        hipError_t _e = (call);                              \
        if (_e != hipSuccess) return azd::hip_fail(_e, #call); \
    } while (0)

} // namespace azd

struct azd_evaluator {
    int device = 0;
    int state_dim = 0;
    int action_dim = 0;
    uint64_t calls = 0;
    uint64_t layout_version = 0; // bumped whenever buffers are reallocated or the storage type changes: a captured graph of write_predictions_dev is stale then
    // staging for the host-pointer entry points
    float *d_states = nullptr, *d_preds = nullptr, *d_obs = nullptr, *d_w = nullptr;
    int staged_batch = 0;
    hipStream_t own_stream = nullptr;

    virtual ~azd_evaluator();
    // NablaModel::write_predictions on device pointers
    virtual int write_predictions_dev(int batch, const float *d_s, float *d_p, hipStream_t st) = 0;
    // NablaModel::update_model on device pointers; *loss is a host float, valid on return
    virtual int update_model_dev(int batch, const float *d_s, const float *d_o, const float *d_w, float *loss,
                                 hipStream_t st) = 0;
    // The same with the input rows ALSO available as bf16 (RNE of the f32 rows, pitch `pitch16` elements, zero beyond state_dim):
    // an evaluator with bf16 storage takes them as they are instead of converting the f32 rows.  input16_pitch(): the pitch it
    // wants (0: it has no use for such rows).
    virtual int write_predictions_dev16(int batch, const float *d_s, const uint16_t * /*d_s16*/, int /*pitch16*/, float *d_p, hipStream_t st) {
        return write_predictions_dev(batch, d_s, d_p, st);
    }
    virtual int input16_pitch() { return 0; }
    // The rows row0 .. row0 + count - 1 of a batch whose buffers start at d_s / d_s16 / d_p.  rows_concurrent(): calls on DISJOINT
    // row ranges may run at the same time on different streams (the engine's launch-per-phase form over sub-populations).
    virtual int write_predictions_rows(int row0, int count, const float *d_s, const uint16_t *d_s16, int pitch16, float *d_p, hipStream_t st) {
        return write_predictions_dev16(count, d_s + (size_t)row0 * state_dim, d_s16 ? d_s16 + (size_t)row0 * pitch16 : nullptr, pitch16,
                                       d_p + (size_t)row0 * action_dim, st);
    }
    virtual bool rows_concurrent() { return false; }
    virtual int ensure_rows(int /*rows*/) { return AZD_OK; } // the evaluator's own buffers hold this many rows from here on (may bump layout_version)
    // The pool step's evaluator outside the kernel (engine.hip, dense-graph space): predictions for the *d_count (<= max_rows) rows
    // whose indices stand in d_rows -- inputs d_s16[row], outputs d_p[row] -- with everything the launches need in device memory,
    // so that the sequence can be captured once and replayed.  AZD_ERR_UNSUPPORTED: this evaluator cannot (the engine asks once).
    // act_row0: first row of the evaluator's own buffers this call may use (calls on disjoint ranges [act_row0, act_row0 + max_rows) may
    // run at the same time on different streams).
    virtual int write_predictions_gathered(const uint32_t * /*d_rows*/, const uint32_t * /*d_count*/, int /*max_rows*/, const uint16_t * /*d_s16*/,
                                           int /*pitch16*/, float * /*d_p*/, hipStream_t /*st*/, int /*act_row0*/ = 0) {
        return AZD_ERR_UNSUPPORTED;
    }
    // description for the persistent step (evaluator inside the kernel); false = not fusable
    virtual bool fused_desc(azd::FusedEval *) { return false; }
    // write_predictions_dev(batch) launches the same kernels with the same arguments on every call and allocates nothing
    // once a batch of that size has run: the engine may capture it into a hipGraph and replay it
    virtual bool replayable(int /*batch*/) { return false; }
    virtual int64_t num_params() { return 0; }
    virtual int get_params(float *) { return AZD_ERR_UNSUPPORTED; }
    virtual int set_params(const float *) { return AZD_ERR_UNSUPPORTED; }
    virtual int set_weight_storage(int) { return AZD_ERR_UNSUPPORTED; }
    virtual int debug_serve_from_pool(int) { return AZD_ERR_UNSUPPORTED; } // hash stream only (azd_debug_hash_stream_via_evaluators)
    int ensure_staging(int batch);
};

namespace azd {
azd_evaluator *make_mlp_evaluator(int device, int max_batch, int state_dim, int action_dim, const int *hidden,
                                  int n_hidden, int final_act, const azd_adam_config *adam, uint64_t seed, int *status);
}
