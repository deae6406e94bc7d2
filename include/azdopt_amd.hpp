// azdopt_amd.hpp -- C++ host side of the drop-in boundary (header only, over the C ABI of azdopt_amd.h).
//
// The reference is compiled code (Rust); a host that is compiled code drives the MI355X engine through these classes,
// which keep the reference's names and argument meaning:
//
//   reference (file:line relative to /root/reference/az-discrete-opt/src)      here
//   ------------------------------------------------------------------------   ------------------------------------------
//   nabla/model/mod.rs:4-8      trait NablaModel { write_predictions,          azdopt::NablaModel (ActionModel, TrivialModel,
//                               update_model }                                  HashStreamModel)
//                               impl NablaModel for a host's own type          azdopt::HostModel (called across the boundary at
//                                                                               optimizer/mod.rs:72, :175-176, :270-277, :348)
//   nabla/model/dfdx.rs:36-53   ActionModel::new(model, cfg)                   azdopt::ActionModel(max_batch, S, A, hidden, adam)
//   nabla/optimizer/mod.rs:39   NablaOptimizer::par_new(space, init_states,    azdopt::NablaOptimizer::par_new(space, roots,
//                               model, batch, ...)                              model, batch)
//   nabla/optimizer/mod.rs:121  par_roll_out_episodes(n_as_tol)                par_roll_out_episodes(n_as_tol, n_calls)
//   nabla/optimizer/mod.rs:249  par_update_model(n_obs_tol)                    par_update_model(n_obs_tol)
//   nabla/optimizer/mod.rs:284  par_reset_trees(modify_root, ...)              par_reset_trees(roots) / par_reset_trees_policy(...)
//   nabla/optimizer/mod.rs:361  argmin_data()                                  argmin_data()
//   log.rs:1-11                 ArgminData { state, cost, eval }               azdopt::ArgminData
//
// What the boundary forces: closures cannot cross to the device, so `init_states` / `modify_root` are the seeded built-ins
// (Space::generate_roots, par_reset_trees_policy) or packed roots the caller builds, and `n_as_tol` is a table + default
// (04-c21-tree.rs:136-138).  Error behaviour: where the reference panics (unreachable!(), index out of bounds on an
// over-full arena, a CUDA error), these throw azdopt::Error carrying the AZD_ERR_* status; nothing falls back to the CPU.
#pragma once
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "azdopt_amd.h"

namespace azdopt {

class Error : public std::runtime_error {
public:
    Error(int status, const std::string &where)
        : std::runtime_error(where + ": " + azd_status_string(status) + " (" + azd_last_error() + ")"), status_(status) {}
    int status() const { return status_; }

private:
    int status_;
};
inline void check(int status, const char *where) {
    if (status != AZD_OK) throw Error(status, where);
}

// n_as_tol: |path| -> visits after which a node is no longer re-entered (04-c21-tree.rs:136-138), as table + default
struct Tolerance {
    std::vector<uint32_t> table;
    uint32_t otherwise;
};

// packed roots of a space: the `init_states` / `modify_root` results in the layout azdopt_amd.h documents per space
struct Roots {
    std::vector<uint8_t> state;      // c21: parents [count][n]; Ramsey: colours [count][E]
    std::vector<uint64_t> permitted; // [count][KEY_WORDS]
};

// ---------------------------------------------------------------- spaces (NablaStateActionSpace, nabla/space/mod.rs:5-38)
// ROTModifyParentsOnce<N, Conjecture2Dot1Cost> (graph-state/src/rooted_tree/space.rs:14-125)
class ROTModifyParentsOnce {
public:
    explicit ROTModifyParentsOnce(int n) : n_(n) {}
    int n() const { return n_; }
    int STATE_DIM() const { return azd_c21_state_dim(n_); }   // space.rs:46
    int ACTION_DIM() const { return azd_c21_action_dim(n_); } // space.rs:48
    int KEY_WORDS() const { return azd_c21_key_words(n_); }
    // the drivers' init_states (04-c21-tree.rs:108-112), seeded; k in [kmin, kmax] permitted actions per root
    Roots generate_roots(uint64_t seed, int count, int kmin, int kmax, uint64_t first_agent = 0, uint64_t epoch = 0) const {
        Roots r;
        r.state.resize((size_t)count * n_);
        r.permitted.resize((size_t)count * KEY_WORDS());
        check(azd_c21_generate_roots(seed, epoch, first_agent, count, n_, kmin, kmax, r.state.data(), r.permitted.data()), "generate_roots");
        return r;
    }
    void configure(azd_engine_config &cfg) const {
        cfg.space_id = AZD_SPACE_C21;
        cfg.n = n_;
    }

private:
    int n_;
};

// RamseySpaceNoEdgeRecolor<B32, N, E, C> (graph-state/src/ramsey_counts/space.rs:10-176)
class RamseySpaceNoEdgeRecolor {
public:
    RamseySpaceNoEdgeRecolor(int n, std::vector<int> sizes, std::vector<float> weights = {})
        : n_(n), sizes_(std::move(sizes)), weights_(std::move(weights)) {
        if (weights_.empty()) weights_.assign(sizes_.size(), 1.0f);
        if (sizes_.size() < 2 || sizes_.size() > 4 || weights_.size() != sizes_.size()) throw Error(AZD_ERR_INVALID_ARGUMENT, "RamseySpaceNoEdgeRecolor");
    }
    int n() const { return n_; }
    int C() const { return (int)sizes_.size(); }
    int E() const { return n_ * (n_ - 1) / 2; }
    int STATE_DIM() const { return azd_ramsey_state_dim(n_, C()); }   // space.rs:40
    int ACTION_DIM() const { return azd_ramsey_action_dim(n_, C()); } // space.rs:42
    int KEY_WORDS() const { return azd_ramsey_key_words(n_, C()); }
    Roots generate_roots(uint64_t seed, int count, int kmin, int kmax, uint64_t first_agent = 0, uint64_t epoch = 0) const {
        Roots r;
        r.state.resize((size_t)count * E());
        r.permitted.resize((size_t)count * KEY_WORDS());
        check(azd_ramsey_generate_roots(seed, epoch, first_agent, count, n_, C(), kmin, kmax, r.state.data(), r.permitted.data()), "generate_roots");
        return r;
    }
    void configure(azd_engine_config &cfg) const {
        cfg.space_id = AZD_SPACE_RAMSEY;
        cfg.n = n_;
        cfg.n_colors = C();
        for (int c = 0; c < C(); ++c) {
            cfg.clique_sizes[c] = sizes_[(size_t)c];
            cfg.color_weights[c] = weights_[(size_t)c];
        }
    }

private:
    int n_;
    std::vector<int> sizes_;
    std::vector<float> weights_;
};

// Connected graphs on N <= 64 vertices, AddOrDeleteEdge actions, every edge slot modified at most once (BASELINE configs[4],
// N = 50).  BUILD-DEFINED: the reference has the pieces (connected_bitset_graph/mod.rs, bitset_graph/space/action.rs:4-27,
// STATE = E + ACTION + 1 at 05-ah.rs:39-40) but no live NablaStateActionSpace over general graphs; see azdopt_amd.h.
class DenseGraphSpace {
public:
    // max_slots: the most modifiable edge slots a root may bring (azd_engine_config::max_slots; up to E / 2 in the drivers' image)
    explicit DenseGraphSpace(int n, double p = 0.2, int max_slots = 128) : n_(n), p_(p), max_slots_(max_slots) {}
    int max_slots() const { return max_slots_; }
    int n() const { return n_; }
    int E() const { return n_ * (n_ - 1) / 2; }
    int STATE_DIM() const { return azd_dense_state_dim(n_); }
    int ACTION_DIM() const { return azd_dense_action_dim(n_); }
    int KEY_WORDS() const { return azd_dense_key_words(n_); }
    // a connected G(n, p) per root and k in [kmin, kmax] modifiable edge slots; `state` holds the neighbourhood bitsets
    // (8 n bytes per root)
    Roots generate_roots(uint64_t seed, int count, int kmin, int kmax, uint64_t first_agent = 0, uint64_t epoch = 0) const {
        Roots r;
        std::vector<uint64_t> adj((size_t)count * n_);
        r.permitted.resize((size_t)count * KEY_WORDS());
        check(azd_dense_generate_roots(seed, epoch, first_agent, count, n_, kmin, kmax, p_, adj.data(), r.permitted.data()), "generate_roots");
        r.state.resize(adj.size() * 8);
        std::memcpy(r.state.data(), adj.data(), r.state.size());
        return r;
    }
    void configure(azd_engine_config &cfg) const {
        cfg.space_id = AZD_SPACE_DENSE;
        cfg.n = n_;
        cfg.max_slots = max_slots_;
        cfg.dense_p = (float)p_;
    }

private:
    int n_;
    double p_;
    int max_slots_;
};

// ---------------------------------------------------------------- models (NablaModel, nabla/model/mod.rs:4-8)
class NablaModel {
public:
    NablaModel(const NablaModel &) = delete;
    NablaModel &operator=(const NablaModel &) = delete;
    virtual ~NablaModel() {
        if (ev_) azd_evaluator_destroy(ev_);
    }
    // write_predictions(&mut self, x: &[f32], predictions: &mut [f32]) -- host slices, batch rows
    void write_predictions(int batch, const float *states, float *predictions) {
        check(azd_evaluator_write_predictions(ev_, batch, states, predictions), "write_predictions");
    }
    // update_model(&mut self, x, observations, action_weights) -> loss
    float update_model(int batch, const float *states, const float *observations, const float *action_weights) {
        float loss = 0.f;
        check(azd_evaluator_update_model(ev_, batch, states, observations, action_weights, &loss), "update_model");
        return loss;
    }
    azd_evaluator *handle() const { return ev_; }

protected:
    NablaModel() = default;
    azd_evaluator *ev_ = nullptr;
};

struct AdamConfig { // dfdx AdamConfig as the drivers set it (04-c21-tree.rs:87-92)
    float lr = 1e-4f, beta1 = 0.9f, beta2 = 0.999f, eps = 1e-8f, l2 = 1e-6f;
};

// ActionModel (nabla/model/dfdx.rs:16-53): MLP state_dim -> hidden... -> action_dim, ReLU between, `final_act` on the head
class ActionModel : public NablaModel {
public:
    ActionModel(int max_batch, int state_dim, int action_dim, const std::vector<int> &hidden, const AdamConfig &adam = AdamConfig(),
                uint64_t seed = 0, int final_act = AZD_ACT_SIGMOID /* the drivers' head, 04-c21-tree.rs:46-52 */, int device = 0,
                bool bf16_storage = false) {
        const azd_adam_config cfg = {adam.lr, adam.beta1, adam.beta2, adam.eps, adam.l2};
        check(azd_evaluator_create_mlp(&ev_, device, max_batch, state_dim, action_dim, hidden.data(), (int)hidden.size(), final_act, &cfg, seed), "ActionModel::new");
        if (bf16_storage) check(azd_evaluator_set_weight_storage(ev_, AZD_STORAGE_BF16), "set_weight_storage");
    }
    std::vector<float> get_params() {
        std::vector<float> p((size_t)azd_evaluator_num_params(ev_));
        check(azd_evaluator_get_params(ev_, p.data()), "get_params");
        return p;
    }
    void set_params(const std::vector<float> &p) { check(azd_evaluator_set_params(ev_, p.data()), "set_params"); }
};

// TrivialModel (nabla/model/mod.rs:10-23)
class TrivialModel : public NablaModel {
public:
    TrivialModel(int state_dim, int action_dim, int device = 0) { check(azd_evaluator_create_trivial(&ev_, device, state_dim, action_dim), "TrivialModel"); }
};

// the parity harness' fixed prediction stream (no counterpart in the reference)
class HashStreamModel : public NablaModel {
public:
    HashStreamModel(int state_dim, int action_dim, uint64_t seed, uint64_t first_agent = 0, int device = 0) {
        check(azd_evaluator_create_hash_stream(&ev_, device, state_dim, action_dim, seed, first_agent), "HashStreamModel");
    }
};

// A NablaModel that lives on the HOST side of the boundary: any C++ type with the trait's two methods over host slices.  The
// optimizer then runs every call split at the model call (the C ABI's *_begin / *_end pairs, cut at optimizer/mod.rs:72,
// :175-176, :348): state vectors come out, the host's predictions go back in.  One launch per phase and a host round trip
// per call -- the device-resident evaluators above are the fast path; this is the seam for a model the library does not have.
class HostModel {
public:
    virtual ~HostModel() = default;
    // write_predictions(&mut self, x: &[f32], predictions: &mut [f32]): batch rows of STATE / ACTION floats; `predictions`
    // holds the previous call's rows on entry (TrivialModel, model/mod.rs:10-23, leaves them as they are)
    virtual void write_predictions(int batch, const float *x, float *predictions) = 0;
    // update_model(&mut self, x, observations, action_weights) -> loss (model/mod.rs:6-7)
    virtual float update_model(int batch, const float *x, const float *observations, const float *action_weights) = 0;
};

// ---------------------------------------------------------------- ArgminData (log.rs:1-11)
struct C21Argmin {
    std::vector<uint8_t> parents;                  // state: the rooted tree
    std::vector<uint64_t> permitted;               // ... and its permitted actions
    double lambda_1;                               // cost: Conjecture2Dot1Cost { matching, lambda_1 }
    std::vector<std::pair<int, int>> matching;     // (parent, child) pairs
    float eval;
    int agent;
    uint32_t node;
};
struct RamseyArgmin {
    std::vector<uint8_t> colors;
    std::vector<uint64_t> permitted;
    std::vector<int> clique_counts; // cost: TotalCounts
    float eval;
    int agent;
    uint32_t node;
};

struct DenseArgmin {
    std::vector<uint64_t> adj;       // state: neighbourhood bitsets
    std::vector<uint64_t> permitted; // ... and the edge slots still modifiable
    double lambda_1;                 // cost: Conjecture2Dot1Cost { lambda_1, matching number }
    int matching_number;
    float eval;
    int agent;
    uint32_t node;
};

// ---------------------------------------------------------------- NablaOptimizer<Space, M, P> (nabla/optimizer/mod.rs)
template <class Space>
class NablaOptimizer {
public:
    NablaOptimizer(const NablaOptimizer &) = delete;
    NablaOptimizer &operator=(const NablaOptimizer &) = delete;
    NablaOptimizer(NablaOptimizer &&o) noexcept
        : space_(o.space_), model_(o.model_), host_(o.host_), h_(o.h_), batch_(o.batch_), sv_(std::move(o.sv_)), pred_(std::move(o.pred_)) {
        o.h_ = nullptr;
    }
    ~NablaOptimizer() {
        if (h_) azd_engine_destroy(h_);
    }

    // optimizer/mod.rs:39-118: trees from the roots, costs, first predictions, the roots' actions.  `model` must outlive the
    // optimizer (the reference moves it in; here the engine borrows the evaluator).  Capacities 0 = sized for 800 calls per epoch.
    static NablaOptimizer par_new(const Space &space, const Roots &roots, NablaModel &model, int batch, int device = 0, uint64_t first_agent = 0,
                                  uint32_t flags = 0, int node_capacity = 0, int arc_capacity = 0, int prediction_capacity = 0, int layers = 0,
                                  int path_kind = AZD_PATH_SET) {
        azd_engine_config cfg = {};
        space.configure(cfg);
        cfg.batch = batch;
        cfg.device = device;
        cfg.node_capacity = node_capacity;
        cfg.arc_capacity = arc_capacity;
        cfg.prediction_capacity = prediction_capacity;
        cfg.first_agent = first_agent;
        cfg.flags = flags;
        cfg.path_kind = path_kind;
        cfg.layers = layers;
        azd_engine *h = nullptr;
        check(azd_engine_create(&h, &cfg, model.handle()), "NablaOptimizer::par_new (engine)");
        NablaOptimizer opt(space, model, h, batch);
        check(azd_engine_par_new(h, roots.state.data(), roots.permitted.data()), "NablaOptimizer::par_new");
        return opt;
    }

    // the same with a model on the host side of the boundary (HostModel): the engine has no evaluator and every call is split
    // at the model call
    static NablaOptimizer par_new(const Space &space, const Roots &roots, HostModel &model, int batch, int device = 0, uint64_t first_agent = 0,
                                  int node_capacity = 0, int arc_capacity = 0, int prediction_capacity = 0) {
        azd_engine_config cfg = {};
        space.configure(cfg);
        cfg.batch = batch;
        cfg.device = device;
        cfg.node_capacity = node_capacity;
        cfg.arc_capacity = arc_capacity;
        cfg.prediction_capacity = prediction_capacity;
        cfg.first_agent = first_agent;
        azd_engine *h = nullptr;
        check(azd_engine_create(&h, &cfg, nullptr), "NablaOptimizer::par_new (engine)");
        NablaOptimizer opt(space, h, batch);
        opt.host_ = &model;
        opt.sv_.assign((size_t)batch * space.STATE_DIM(), 0.f);
        opt.pred_.assign((size_t)batch * space.ACTION_DIM(), 0.f);
        check(azd_engine_par_new_begin(h, roots.state.data(), roots.permitted.data()), "par_new_begin");
        opt.host_predictions();
        check(azd_engine_par_new_end(h, opt.pred_.data()), "par_new_end");
        return opt;
    }

    // optimizer/mod.rs:121-191, n_calls times back to back on the device (a HostModel: call by call through the host); returns
    // how many calls improved the argmin
    int par_roll_out_episodes(const Tolerance &n_as_tol, int n_calls = 1) {
        int improved = 0;
        if (!host_) {
            check(azd_engine_par_roll_out_episodes(h_, n_as_tol.table.data(), (int)n_as_tol.table.size(), n_as_tol.otherwise, n_calls, &improved), "par_roll_out_episodes");
            return improved;
        }
        for (int i = 0; i < n_calls; ++i) {
            int one = 0;
            check(azd_engine_roll_out_begin(h_, n_as_tol.table.data(), (int)n_as_tol.table.size(), n_as_tol.otherwise), "roll_out_begin");
            host_predictions();
            check(azd_engine_roll_out_end(h_, pred_.data(), &one), "roll_out_end");
            improved += one;
        }
        return improved;
    }
    // Starts the next n_calls calls of par_roll_out_episodes(n_as_tol, ...) now, in one launch; the calls that ask for them (one at a
    // time, as in 04-c21-tree.rs:142-150, or in chunks) are answered as the kernel completes them (azd_engine_run_ahead).  False:
    // this engine's step cannot do that (or the model lives on the host), and the calls run when they are asked for.
    bool run_ahead(const Tolerance &n_as_tol, int n_calls) {
        if (host_) return false;
        int ok = 0;
        check(azd_engine_run_ahead(h_, n_as_tol.table.data(), (int)n_as_tol.table.size(), n_as_tol.otherwise, n_calls, &ok), "run_ahead");
        return ok != 0;
    }
    // optimizer/mod.rs:249-281; returns the loss
    float par_update_model(uint32_t n_obs_tol) {
        float loss = 0.f;
        if (!host_) {
            check(azd_engine_par_update_model(h_, n_obs_tol, &loss), "par_update_model");
            return loss;
        }
        std::vector<float> obs(pred_.size()), w(pred_.size());
        check(azd_engine_observe(h_, n_obs_tol, sv_.data(), obs.data(), w.data()), "observe");
        return host_->update_model(batch_, sv_.data(), obs.data(), w.data());
    }
    // par_update_model for a population sharded over several GPUs, one process and engine per GPU: the
    // training triples are all-gathered over RCCL (`nccl_comm` = the caller's ncclComm_t) and every rank takes the same
    // optimiser step on the pooled rows; the model must have been created with max_batch >= batch * world size
    float par_update_model_sharded(uint32_t n_obs_tol, void *nccl_comm) {
        float loss = 0.f;
        check(azd_engine_par_update_model_sharded(h_, n_obs_tol, nccl_comm, &loss), "par_update_model_sharded");
        return loss;
    }
    // optimizer/mod.rs:284-360 with the caller's modify_root results ...
    void par_reset_trees(const Roots &roots) {
        if (!host_) {
            check(azd_engine_par_reset_trees(h_, roots.state.data(), roots.permitted.data()), "par_reset_trees");
            return;
        }
        check(azd_engine_reset_begin(h_, roots.state.data(), roots.permitted.data()), "reset_begin");
        host_predictions();
        check(azd_engine_reset_end(h_, pred_.data()), "reset_end");
    }
    // ... or with the drivers' modify_root policy (04-c21-tree.rs:172-206, 02-r44.rs:196-228) evaluated on the device
    void par_reset_trees_policy(uint64_t seed, uint64_t epoch, int kmin, int kmax) {
        if (!host_) {
            check(azd_engine_par_reset_trees_policy(h_, seed, epoch, kmin, kmax), "par_reset_trees_policy");
            return;
        }
        Roots r; // the policy on the device, the reset through the host's model
        r.state.resize((size_t)batch_ * root_bytes(space_));
        r.permitted.resize((size_t)batch_ * space_.KEY_WORDS());
        check(azd_engine_modify_roots_dev(h_, seed, epoch, kmin, kmax, r.state.data(), r.permitted.data()), "modify_roots");
        par_reset_trees(r);
    }
    // optimizer/mod.rs:361
    auto argmin_data() { return argmin_of(space_); }

    std::vector<uint64_t> counters() {
        std::vector<uint64_t> c(AZD_CTR_COUNT);
        check(azd_engine_counters(h_, c.data()), "counters");
        return c;
    }
    // which form of the step ran last (AZD_STEP_*) and why a faster one was not taken
    std::pair<int, std::string> step_form() {
        int form = 0;
        const char *why = "";
        check(azd_engine_step_form(h_, &form, &why), "step_form");
        return {form, why ? why : ""};
    }
    int batch() const { return batch_; }
    azd_engine *handle() const { return h_; }

private:
    NablaOptimizer(const Space &space, NablaModel &model, azd_engine *h, int batch) : space_(space), model_(&model), h_(h), batch_(batch) {}
    NablaOptimizer(const Space &space, azd_engine *h, int batch) : space_(space), h_(h), batch_(batch) {}
    static int root_bytes(const ROTModifyParentsOnce &sp) { return sp.n(); }
    static int root_bytes(const RamseySpaceNoEdgeRecolor &sp) { return sp.E(); }
    static int root_bytes(const DenseGraphSpace &sp) { return 8 * sp.n(); }
    void host_predictions() { // the model call of optimizer/mod.rs:72 / :175-176 / :348 on the host side
        check(azd_engine_read_state_vecs(h_, sv_.data()), "read_state_vecs");
        host_->write_predictions(batch_, sv_.data(), pred_.data());
    }
    C21Argmin argmin_of(const ROTModifyParentsOnce &sp) {
        azd_argmin a;
        check(azd_engine_argmin_data(h_, &a), "argmin_data");
        C21Argmin r;
        r.parents.assign(a.parents, a.parents + sp.n());
        r.permitted.assign(a.permitted, a.permitted + sp.KEY_WORDS());
        r.lambda_1 = a.lambda_1;
        for (int i = 0; i < a.matching_size; ++i) r.matching.emplace_back(a.matching[2 * i], a.matching[2 * i + 1]);
        r.eval = a.eval;
        r.agent = a.agent;
        r.node = a.node;
        return r;
    }
    RamseyArgmin argmin_of(const RamseySpaceNoEdgeRecolor &sp) {
        azd_ramsey_argmin a;
        check(azd_engine_ramsey_argmin_data(h_, &a), "argmin_data");
        RamseyArgmin r;
        r.colors.assign(a.colors, a.colors + sp.E());
        r.permitted.assign(a.permitted, a.permitted + (sp.E() + 63) / 64);
        r.clique_counts.assign(a.totals, a.totals + sp.C());
        r.eval = a.eval;
        r.agent = a.agent;
        r.node = a.node;
        return r;
    }
    DenseArgmin argmin_of(const DenseGraphSpace &sp) {
        azd_dense_argmin a;
        check(azd_engine_dense_argmin_data(h_, &a), "argmin_data");
        DenseArgmin r;
        r.adj.assign(a.adj, a.adj + sp.n());
        // `permitted` is a bitmap over the E edge SLOTS (colex positions) still open, not an action-id set: (E + 63) / 64 words
        const int slot_words = (sp.E() + 63) / 64;
        static_assert(sizeof(a.permitted) / sizeof(a.permitted[0]) >= (AZD_DENSE_MAX_N * (AZD_DENSE_MAX_N - 1) / 2 + 63) / 64,
                      "azd_dense_argmin::permitted must hold a slot bitmap at AZD_DENSE_MAX_N");
        r.permitted.assign(a.permitted, a.permitted + slot_words);
        r.lambda_1 = a.lambda_1;
        r.matching_number = a.matching_size;
        r.eval = a.eval;
        r.agent = a.agent;
        r.node = a.node;
        return r;
    }
    Space space_;
    NablaModel *model_ = nullptr;
    HostModel *host_ = nullptr;
    azd_engine *h_;
    int batch_;
    std::vector<float> sv_, pred_; // HostModel: state vectors out, predictions in (they persist from call to call)
};

} // namespace azdopt
