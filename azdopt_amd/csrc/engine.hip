// engine.hip -- host side of the C ABI (include/azdopt_amd.h): owns the device arenas,
// sequences the kernels of one NablaOptimizer call on a HIP stream, copies results out.
// No CPU fallback: every compute entry point needs a gfx950 device.
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/azdopt_amd.h"
#include "c21_host.h"
#include "engine_types.h"
#include "evaluator.h"

namespace azd {

thread_local std::string g_last_error;

int hip_fail(hipError_t e, const char *what) {
    g_last_error = std::string(what) + ": " + hipGetErrorString(e);
    if (e == hipErrorNoDevice || e == hipErrorInvalidDevice) return AZD_ERR_NO_DEVICE;
    if (e == hipErrorOutOfMemory) return AZD_ERR_OUT_OF_MEMORY;
    return AZD_ERR_HIP;
}

static int device_ok(int device) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        g_last_error = "no HIP device visible (this library has no CPU fallback)";
        return AZD_ERR_NO_DEVICE;
    }
    if (device < 0 || device >= n) {
        g_last_error = "device ordinal out of range";
        return AZD_ERR_INVALID_ARGUMENT;
    }
    return AZD_OK;
}

// ------------------------------------------------------------------ simple evaluators
struct TrivialEvaluator : azd_evaluator { // TrivialModel, model/mod.rs:10-23
    int write_predictions_dev(int, const float *, float *, hipStream_t) override {
        calls += 1;
        return AZD_OK;
    }
    int update_model_dev(int, const float *, const float *, const float *, float *loss, hipStream_t) override {
        if (loss) *loss = 0.f;
        return AZD_OK;
    }
    bool fused_desc(FusedEval *f) override {
        memset(f, 0, sizeof(*f));
        f->kind = 1;
        return true;
    }
};

struct HashStreamEvaluator : azd_evaluator {
    uint64_t seed = 0, first_agent = 0;
    bool via_pool = false; // the pool step serves the rows from its evaluator workgroups (FusedEval kind 4: test harness)
    int debug_serve_from_pool(int on) override {
        via_pool = on != 0;
        return AZD_OK;
    }
    int write_predictions_dev(int batch, const float *, float *d_p, hipStream_t st) override {
        launch_hash_predictions(d_p, batch, action_dim, seed, first_agent, calls, st);
        calls += 1;
        AZD_HIP(hipGetLastError());
        return AZD_OK;
    }
    bool fused_desc(FusedEval *f) override {
        memset(f, 0, sizeof(*f));
        f->kind = via_pool ? 4 : 2;
        f->seed = seed;
        f->first_agent = first_agent;
        return true;
    }
    int update_model_dev(int, const float *, const float *, const float *, float *loss, hipStream_t) override {
        if (loss) *loss = 0.f;
        return AZD_OK;
    }
};

} // namespace azd

azd_evaluator::~azd_evaluator() {
    if (d_states) (void)hipFree(d_states);
    if (d_preds) (void)hipFree(d_preds);
    if (d_obs) (void)hipFree(d_obs);
    if (d_w) (void)hipFree(d_w);
    if (own_stream) (void)hipStreamDestroy(own_stream);
}

int azd_evaluator::ensure_staging(int batch) {
    AZD_HIP(hipSetDevice(device));
    if (!own_stream) AZD_HIP(hipStreamCreate(&own_stream));
    if (batch <= staged_batch) return AZD_OK;
    if (d_states) (void)hipFree(d_states);
    if (d_preds) (void)hipFree(d_preds);
    if (d_obs) (void)hipFree(d_obs);
    if (d_w) (void)hipFree(d_w);
    d_states = d_preds = d_obs = d_w = nullptr;
    staged_batch = 0;
    AZD_HIP(hipMalloc(&d_states, (size_t)batch * state_dim * sizeof(float)));
    AZD_HIP(hipMalloc(&d_preds, (size_t)batch * action_dim * sizeof(float)));
    AZD_HIP(hipMalloc(&d_obs, (size_t)batch * action_dim * sizeof(float)));
    AZD_HIP(hipMalloc(&d_w, (size_t)batch * action_dim * sizeof(float)));
    staged_batch = batch;
    return AZD_OK;
}

// ------------------------------------------------------------------ engine
struct azd_engine {
    azd_engine_config cfg;
    azd::Arenas a;
    azd_evaluator *ev = nullptr;
    hipStream_t stream = nullptr;
    std::vector<void *> allocs;
    uint8_t *d_stage_parents = nullptr;
    uint64_t *d_stage_perm = nullptr;
    azd::StatusRec *h_status = nullptr; // pinned
    azd::ArgminRec *h_argmin = nullptr; // pinned
    unsigned long long seen_improved = 0;
    bool timing = false;
    double rollout_ms = 0, evaluator_ms = 0;
    uint64_t rollout_launches = 0;
    std::vector<hipEvent_t> ev_pool;
    std::vector<std::pair<int, std::pair<hipEvent_t, hipEvent_t>>> ev_inflight; // kind 0 rollout, 1 evaluator
    bool initialised = false;
    // persistent (CU-resident) step
    bool persist_enabled = true;
    bool barrier_step = false;
    azd::PersistArgs *d_pargs = nullptr;
    azd::PersistArgs *h_pargs = nullptr; // pinned
    azd::PersistArgs pargs_sent{};       // what d_pargs holds: the block is re-sent only when it changed
    bool pargs_valid = false;
    bool log_clean = false;              // d_log_key is all ones (k_argmin_log1 leaves it so; the barrier step and an abort do not)
    bool pool_clean = false;             // PoolCtl, the queue slots and the join counters are zero (a completed pool launch leaves them so)
    bool pool_failed = false;            // a pool launch aborted: this engine takes the asynchronous step from then on
    bool counters_by_wave = false;       // a pool launch of the product build has run since the counters were cleared: the blocks of
                                         // azd_engine_agent_counters then hold what searcher WAVES counted, not what agents did
    uint32_t *d_resume = nullptr;        // [B] take-over of an aborted pool launch by k_async (StepLaunch::resume)
    // evaluator groups of the pool step (PoolArgs::grp_*): device tables and per-slot exchange state, (re)built when the plan changes
    int16_t *d_grp_tile = nullptr;
    uint32_t *d_grp_lds = nullptr, *d_grp_desc = nullptr, *d_grp_cnt = nullptr, *d_grp_flag = nullptr;
    size_t grp_flag_words = 0;
    float *d_grp_x = nullptr;
    size_t grp_x_floats = 0, grp_slots_alloc = 0;
    std::vector<int16_t> grp_tile_host;  // what d_grp_tile holds (the plan is re-sent only when it changes)
    std::vector<uint32_t> grp_lds_host;
    unsigned long long *d_log_key = nullptr;
    uint32_t *d_log_node = nullptr;
    int log_calls = 0;
    // launch-per-phase form: the five launches of a call captured once in a hipGraph and replayed per call
    hipGraphExec_t call_graph = nullptr;
    azd::TolTable call_graph_tol{};
    uint64_t call_graph_layout = 0;
    bool graph_enabled = true;
    // launch-per-phase form over sub-populations, each on a stream of its own with its call captured in a graph: the kernels of
    // one sub-population's call overlap the slowest agents of another's (a roll-out launch lasts as long as its slowest agent)
    static constexpr int MAX_SUBS = 8;
    hipStream_t sub_stream[MAX_SUBS] = {};
    hipEvent_t sub_fork = nullptr, sub_join[MAX_SUBS] = {};
    hipGraphExec_t sub_graph[MAX_SUBS] = {};
    int sub_graph_n = 0;
    uint32_t *d_call_ctr = nullptr; // [MAX_SUBS] calls a sub-population has logged in the current run
    // pool step with the evaluator outside the kernel (dense-graph space): the searchers run on `stream`, the host replays a graph of
    // [collect the posted rows, the model's GEMMs over them, hand the agents back] on ext_stream until the searchers are through
    static constexpr int EXT_STREAMS = 2;    // evaluator streams: one collects and runs its first layer while the other is in its later ones
    static constexpr int EXT_IN_FLIGHT = 4;  // ring size per stream; replays queued at a time per stream: ext_depth
    hipStream_t ext_stream[EXT_STREAMS] = {};
    hipGraphExec_t ext_graph[EXT_STREAMS] = {};
    uint64_t ext_graph_layout = 0;
    int ext_graph_n = 0;
    bool ext_unsupported = false;     // the evaluator cannot serve gathered rows from device-side lists (asked once)
    uint32_t *d_ext_rows = nullptr, *d_ext_home = nullptr, *d_ext_n = nullptr; // [EXT_STREAMS][B], [EXT_STREAMS][B], [EXT_STREAMS]
    unsigned long long *d_ext_t0 = nullptr; // [EXT_STREAMS] when the batch in hand was collected (PoolCtl::eval_busy)
    int ext_depth = 2;
    hipEvent_t ext_done = nullptr, ext_fork = nullptr, ext_ring[EXT_STREAMS][EXT_IN_FLIGHT] = {};
    unsigned long long ext_iterations = 0; // evaluator graph replays of the last dense pool launch (diagnostics)
    // dense-graph space: host-visible key width (action-id sets) and the packed roots as the device wants them
    int kw_host = 0;
    int dense_slots = 0;              // 64 * a.KW: the most modifiable slots a root may bring
    uint64_t *d_stage_slots = nullptr; // device root policy: the slot masks it drew, [B][(E + 63) / 64]
    std::vector<uint64_t> dense_packed;
    // pool step (agents multiplexed over searcher waves, evaluator workgroups on CUs of their own)
    bool pool_step = false;
    azd::PoolArgs pool{};         // device pointers of the queues
    size_t pool_slot_words = 0;
    int n_cus = 0;
    int pool_eval_wgs = 0, pool_search_wgs = 0; // of the last launch (diagnostics)
    int pool_grp_g = 0, pool_grp_groups = 0, pool_grp_w = 0; // evaluator groups of the last pool launch (0: none)
    int pool_search_waves = 0;                  // searching waves of the last launch
    double pool_util_eval = 0, pool_util_search = 0; // busy share of the two sides in the last completed pool launch
    // Measured feedback on the split (the fitted constants give the first guess only).  Every pool launch reports how busy its two
    // sides were (PoolCtl::eval_busy / search_busy over the launch's span): the share of its time an evaluator workgroup spent
    // on batches, u_e, and the share a searcher wave spent with an agent in hand, u_s.  Across splits u_e falls and u_s rises
    // with the evaluators' share, and the best split of every workload measured sits where u_e - u_s is +0.00 .. +0.09
    // (profiles/r03_pool_split.txt: configs A-D, 8192 agents fp32, a 384 x 384 model no sweep had seen), so the next launch
    // moves the evaluators' share by 100 workgroups per unit of (u_e - u_s - target), target 0 .. 0.06 by how crowded the waves are.
    struct {
        int n_eval = 0; // what the next launch takes (0: the first guess)
        int updates = 0; // launches the controller has acted on
    } pool_fb;
    // Run-ahead window (azd_engine_run_ahead): one pool launch of n calls is under way, or over, and par_roll_out_episodes hands
    // its calls out one by one from what the kernel publishes (PoolArgs::win_*) instead of launching anything.
    struct Window {
        bool open = false;    // calls run ahead of the host are still to be handed out
        bool drained = false; // the launch behind them is over and its status is in
        bool lumped = false;  // the launch aborted: the calls the take-over completed were reported together
        int n = 0, consumed = 0;
        azd::TolTable tol{};
        uint32_t best_ord = 0;                // order key of the best cost as of `consumed` calls
        unsigned long long best_key = ~0ull;  // its log entry (all ones: still the record from before the window)
        unsigned long long side_key = ~0ull;  // what the side argmin record holds
        unsigned long long base_improved = 0; // StatusRec::improved when the window opened
        unsigned long long reported = 0;      // improvements handed out from this window
        azd::FusedEval fe{};
        azd::PoolArgs pool{};
    } win;
    unsigned long long *h_win_log = nullptr; // pinned, host-coherent [log_calls]
    uint32_t *h_win_flag = nullptr;          // pinned, host-coherent [log_calls]
    azd::ArgminRec *d_argmin_side = nullptr; // the argmin record as of a call inside the window (k_argmin_one)
    azd::RamseyArgminRec *d_argmin_r_side = nullptr;
    float *d_pool = nullptr;      // pooled training triple of all ranks (azd_engine_par_update_model_sharded)
    size_t pool_rows = 0;
    int step_form = 0;            // AZD_STEP_* chosen by the last par_roll_out_episodes
    std::string step_reason;      // why the faster forms were not taken ("" if the asynchronous step ran)

    template <typename T>
    int alloc(T **p, size_t count) {
        void *q = nullptr;
        hipError_t e = hipMalloc(&q, count * sizeof(T));
        if (e != hipSuccess) return azd::hip_fail(e, "hipMalloc");
        allocs.push_back(q);
        *p = (T *)q;
        return AZD_OK;
    }
    hipEvent_t get_event() {
        if (!ev_pool.empty()) {
            hipEvent_t e = ev_pool.back();
            ev_pool.pop_back();
            return e;
        }
        hipEvent_t e;
        (void)hipEventCreate(&e);
        return e;
    }
    void time_begin(int kind) {
        if (!timing) return;
        hipEvent_t b = get_event(), en = get_event();
        (void)hipEventRecord(b, stream);
        ev_inflight.push_back({kind, {b, en}});
    }
    void time_end() {
        if (!timing) return;
        (void)hipEventRecord(ev_inflight.back().second.second, stream);
    }
    void time_collect() { // after a stream sync
        for (auto &it : ev_inflight) {
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, it.second.first, it.second.second);
            if (it.first == 0) {
                rollout_ms += ms;
                rollout_launches += 1;
            } else evaluator_ms += ms;
            ev_pool.push_back(it.second.first);
            ev_pool.push_back(it.second.second);
        }
        ev_inflight.clear();
    }
};

static int window_close(azd_engine *e);
// Entry of every call that changes or reads what a running launch works on: the device is selected and a run-ahead window, if
// one is open, is closed first (its launch is waited for; calls not yet handed out have run and stay run).
#define AZD_ENTER(e)                                \
    do {                                            \
        AZD_HIP(hipSetDevice((e)->cfg.device));     \
        if ((e)->win.open) {                        \
            const int st_w_ = window_close(e);      \
            if (st_w_) return st_w_;                \
        }                                           \
    } while (0)

namespace {

using namespace azd;

uint32_t host_ordf(float f) { // tree_core.inc ordf
    f = f + 0.0f;
    uint32_t u;
    memcpy(&u, &f, 4);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

bool pool_feedback_on() {
    const char *env = getenv("AZD_POOL_FEEDBACK");
    return !env || atoi(env) != 0;
}

int next_pow2(int v) {
    int p = 128;
    while (p < v) p <<= 1;
    return p;
}

int fetch_status(azd_engine *e) { // the device's status block -> e->h_status, behind everything queued on the stream
    AZD_HIP(hipMemcpyAsync(e->h_status, e->a.status, sizeof(StatusRec), hipMemcpyDeviceToHost, e->stream));
    AZD_HIP(hipStreamSynchronize(e->stream));
    e->time_collect();
    AZD_HIP(hipGetLastError());
    return AZD_OK;
}
int check_status(azd_engine *e) {
    if (e->h_status->failed != 0) {
        // distinguish capacity from unreachable by reading the flags
        std::vector<uint32_t> fl((size_t)e->a.B);
        AZD_HIP(hipMemcpy(fl.data(), e->a.flags, fl.size() * 4, hipMemcpyDeviceToHost));
        uint32_t all = 0;
        for (uint32_t f : fl) all |= f;
        char buf[128];
        snprintf(buf, sizeof(buf), "%llu agent(s) stopped, flag union 0x%x", e->h_status->failed, all);
        g_last_error = buf;
        return (all & (FLAG_UNREACHABLE | FLAG_LOOP_GUARD)) ? AZD_ERR_UNREACHABLE : AZD_ERR_CAPACITY;
    }
    return AZD_OK;
}
int sync_status(azd_engine *e) {
    const int st = fetch_status(e);
    return st ? st : check_status(e);
}

// dense-graph space: check the roots and hand the device what it works with -- the modifiable slots ranked by ACTION ID
// (Add(e) = e for an absent edge, Delete(e) = E + e for a present one: adds first, each group ascending), as a
// rank -> action id table, and the mask of ranks still open (all k of them)
int upload_dense_roots(azd_engine *e, const uint8_t *adj_bytes, const uint64_t *slots) {
    const Arenas &a = e->a;
    const int n = a.n, E = a.E, KWH = e->kw_host, KW = a.KW, MAXS = e->dense_slots;
    const size_t per = (size_t)17 * KW; // [rank mask: KW words][rank -> action id: 64 KW x u16]
    e->dense_packed.assign((size_t)a.B * per, 0ull);
    for (int i = 0; i < a.B; ++i) {
        uint64_t adj[64];
        memcpy(adj, adj_bytes + (size_t)i * n * 8, (size_t)n * 8);
        const uint64_t all = n >= 64 ? ~0ull : ((1ull << n) - 1ull);
        for (int v = 0; v < n; ++v) {
            if ((adj[v] & ~all) || ((adj[v] >> v) & 1ull)) {
                g_last_error = "root graph: neighbour out of range or a loop";
                return AZD_ERR_INVALID_ARGUMENT;
            }
            for (int u = 0; u < v; ++u)
                if (((adj[v] >> u) & 1ull) != ((adj[u] >> v) & 1ull)) {
                    g_last_error = "root graph: neighbourhoods are not symmetric";
                    return AZD_ERR_INVALID_ARGUMENT;
                }
        }
        if (!dense_connected(adj, n)) {
            g_last_error = "root graph must be connected";
            return AZD_ERR_INVALID_ARGUMENT;
        }
        const uint64_t *sl = slots + (size_t)i * KWH;
        uint64_t *pk = &e->dense_packed[(size_t)i * per];
        uint16_t *tab = reinterpret_cast<uint16_t *>(pk + KW);
        for (int r = 0; r < MAXS; ++r) tab[r] = 0xFFFFu;
        int k = 0;
        for (int pass = 0; pass < 2; ++pass) { // adds (absent edges), then deletes (present edges)
            int slot = 0;
            for (int v = 1; v < n; ++v)
                for (int u = 0; u < v; ++u, ++slot) {
                    if (!((sl[slot >> 6] >> (slot & 63)) & 1ull)) continue;
                    const bool present = (adj[v] >> u) & 1ull;
                    if (present != (pass == 1)) continue;
                    if (k >= MAXS) {
                        char buf[160];
                        snprintf(buf, sizeof(buf), "a root brings more than %d modifiable slots (azd_engine_config::max_slots sizes the keys)", MAXS);
                        g_last_error = buf;
                        return AZD_ERR_INVALID_ARGUMENT;
                    }
                    tab[k++] = (uint16_t)(pass == 0 ? slot : E + slot);
                }
        }
        for (int w = 0; w < KWH; ++w) {
            const int hi = E - 64 * w;
            if ((hi <= 0 && sl[w] != 0) || (hi > 0 && hi < 64 && (sl[w] >> hi) != 0)) {
                g_last_error = "slot mask has bits beyond the edge count";
                return AZD_ERR_INVALID_ARGUMENT;
            }
        }
        for (int w = 0; w < KW; ++w) {
            const int hi = k - 64 * w;
            pk[w] = hi <= 0 ? 0ull : (hi >= 64 ? ~0ull : ((1ull << hi) - 1ull));
        }
    }
    AZD_HIP(hipMemcpyAsync(e->d_stage_parents, adj_bytes, (size_t)a.B * n * 8, hipMemcpyHostToDevice, e->stream));
    AZD_HIP(hipMemcpyAsync(e->d_stage_perm, e->dense_packed.data(), e->dense_packed.size() * 8, hipMemcpyHostToDevice, e->stream));
    AZD_HIP(hipStreamSynchronize(e->stream)); // the packed block is pageable host memory
    return AZD_OK;
}

int upload_roots(azd_engine *e, const uint8_t *parents, const uint64_t *permitted) {
    const Arenas &a = e->a;
    if (a.space == SPACE_DENSE) return upload_dense_roots(e, parents, permitted);
    if (a.space == SPACE_RAMSEY) {
        // packed roots: colour of every edge in colex order (E bytes) + permitted edge positions
        for (int i = 0; i < a.B; ++i) {
            const uint8_t *col = parents + (size_t)i * a.E;
            for (int x = 0; x < a.E; ++x)
                if (col[x] >= a.C) {
                    g_last_error = "root edge colour out of range";
                    return AZD_ERR_INVALID_ARGUMENT;
                }
            int cnt = 0;
            for (int w = 0; w < a.KW; ++w) {
                uint64_t m = permitted[(size_t)i * a.KW + w];
                int hi = a.E - 64 * w;
                if ((hi <= 0 && m != 0) || (hi > 0 && hi < 64 && (m >> hi) != 0)) {
                    g_last_error = "permitted mask has bits beyond the edge count";
                    return AZD_ERR_INVALID_ARGUMENT;
                }
                cnt += __builtin_popcountll(m);
            }
            if (cnt * (a.C - 1) > MAX_NODE_ACTIONS) {
                g_last_error = "more permitted actions than a node can hold";
                return AZD_ERR_INVALID_ARGUMENT;
            }
        }
        AZD_HIP(hipMemcpyAsync(e->d_stage_parents, parents, (size_t)a.B * a.E, hipMemcpyHostToDevice, e->stream));
        AZD_HIP(hipMemcpyAsync(e->d_stage_perm, permitted, (size_t)a.B * a.KW * 8, hipMemcpyHostToDevice, e->stream));
        return AZD_OK;
    }
    // validate on the host what the kernels assume (parents[v] < v; at most MAX_NODE_ACTIONS permitted)
    for (int i = 0; i < a.B; ++i) {
        const uint8_t *p = parents + (size_t)i * a.n;
        // rooted_tree/mod.rs:14-20: parents[0] = parents[1] = parents[N-1] = 0, and no action ever
        // re-parents vertex N-1 (action children are 2..N-2), so N-1 and N-2 stay leaves -- the
        // lambda_1 kernel gives them no accumulator slot
        if (p[0] != 0 || p[a.n - 1] != 0) {
            g_last_error = "root parents[0] and parents[N-1] must be 0";
            return AZD_ERR_INVALID_ARGUMENT;
        }
        for (int v = 1; v < a.n; ++v)
            if (p[v] >= v) {
                g_last_error = "root parents[v] must be < v";
                return AZD_ERR_INVALID_ARGUMENT;
            }
        int cnt = 0;
        for (int w = 0; w < a.KW; ++w) {
            uint64_t m = permitted[(size_t)i * a.KW + w];
            int hi = a.A - 64 * w;
            if (hi < 64 && hi > 0 && (m >> hi) != 0) {
                g_last_error = "permitted mask has bits beyond ACTION_DIM";
                return AZD_ERR_INVALID_ARGUMENT;
            }
            cnt += __builtin_popcountll(m);
        }
        if (cnt > MAX_NODE_ACTIONS) {
            g_last_error = "more permitted actions than a node can hold";
            return AZD_ERR_INVALID_ARGUMENT;
        }
    }
    AZD_HIP(hipMemcpyAsync(e->d_stage_parents, parents, (size_t)a.B * a.n, hipMemcpyHostToDevice, e->stream));
    AZD_HIP(hipMemcpyAsync(e->d_stage_perm, permitted, (size_t)a.B * a.KW * 8, hipMemcpyHostToDevice, e->stream));
    return AZD_OK;
}

int run_evaluator(azd_engine *e) {
    e->time_begin(1);
    int st = e->ev->write_predictions_dev16(e->a.B, e->a.state_vecs, e->a.state_vecs16, e->a.S16, e->a.h_theta, e->stream);
    e->time_end();
    return st;
}

int fill_tol(TolTable &t, const uint32_t *tol, int n_tol, uint32_t dflt) {
    if (n_tol < 0 || n_tol > MAX_TOL || (n_tol > 0 && !tol)) return AZD_ERR_INVALID_ARGUMENT;
    for (int i = 0; i < MAX_TOL; ++i) t.tol[i] = i < n_tol ? tol[i] : dflt;
    t.n_tol = n_tol;
    t.tol_default = dflt;
    return AZD_OK;
}

} // namespace

extern "C" {

const char *azd_status_string(int s) {
    switch (s) {
    case AZD_OK: return "ok";
    case AZD_ERR_INVALID_ARGUMENT: return "invalid argument";
    case AZD_ERR_NO_DEVICE: return "no gfx950 device (no CPU fallback)";
    case AZD_ERR_HIP: return "HIP runtime error";
    case AZD_ERR_CAPACITY: return "tree arena capacity exceeded";
    case AZD_ERR_UNREACHABLE: return "unreachable selection state";
    case AZD_ERR_NO_EVALUATOR: return "engine has no evaluator";
    case AZD_ERR_OUT_OF_MEMORY: return "out of device memory";
    case AZD_ERR_UNSUPPORTED: return "unsupported";
    default: return "unknown status";
    }
}
const char *azd_last_error(void) { return azd::g_last_error.c_str(); }
int azd_version(void) { return 100; }
int azd_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int azd_ramsey_state_dim(int n, int n_colors) { return azd::ramsey_state_dim(n, n_colors); }
int azd_ramsey_action_dim(int n, int n_colors) { return azd::ramsey_action_dim(n, n_colors); }
int azd_ramsey_key_words(int n, int n_colors) { return azd::ramsey_key_words(n, n_colors); }
int azd_ramsey_generate_roots(uint64_t seed, uint64_t epoch, uint64_t first_agent, int count, int n, int n_colors, int kmin,
                              int kmax, uint8_t *colors, uint64_t *permitted) {
    if (!colors || !permitted || count < 0 || n < 3 || n > AZD_RAMSEY_MAX_N || n_colors < 2 || n_colors > 4) return AZD_ERR_INVALID_ARGUMENT;
    if (kmin < 0 || kmax < kmin || kmax > azd::ramsey_edges(n)) return AZD_ERR_INVALID_ARGUMENT;
    azd::ramsey_generate_roots(seed, epoch, first_agent, count, n, n_colors, kmin, kmax, colors, permitted);
    return AZD_OK;
}
int azd_dense_state_dim(int n) { return azd::dense_state_dim(n); }
int azd_dense_action_dim(int n) { return azd::dense_action_dim(n); }
int azd_dense_key_words(int n) { return azd::dense_key_words(n); }
int azd_dense_generate_roots(uint64_t seed, uint64_t epoch, uint64_t first_agent, int count, int n, int kmin, int kmax, double p,
                             uint64_t *adj, uint64_t *slots) {
    if (!adj || !slots || count < 0 || n < 4 || n > AZD_DENSE_MAX_N || !(p > 0.0 && p <= 1.0)) return AZD_ERR_INVALID_ARGUMENT;
    if (kmin < 1 || kmax < kmin || kmax > AZD_DENSE_MAX_SLOTS || kmax > azd::dense_edges(n)) return AZD_ERR_INVALID_ARGUMENT;
    azd::dense_generate_roots(seed, epoch, first_agent, count, n, kmin, kmax, (uint32_t)(p * 16777216.0 + 0.5), adj, slots);
    return AZD_OK;
}
int azd_c21_state_dim(int n) { return azd::c21_state_dim(n); }
int azd_c21_action_dim(int n) { return azd::c21_action_dim(n); }
int azd_c21_key_words(int n) { return azd::c21_key_words(n); }
int azd_c21_generate_roots(uint64_t seed, uint64_t epoch, uint64_t first_agent, int count, int n, int kmin,
                           int kmax, uint8_t *parents, uint64_t *permitted) {
    if (n < 4 || n > AZD_C21_MAX_N || count < 0 || !parents || !permitted) return AZD_ERR_INVALID_ARGUMENT;
    int A = azd::c21_action_dim(n);
    if (kmin < 0 || kmax < kmin || kmax > A || kmax > azd::MAX_NODE_ACTIONS) return AZD_ERR_INVALID_ARGUMENT;
    azd::c21_generate_roots(seed, epoch, first_agent, count, n, kmin, kmax, parents, permitted);
    return AZD_OK;
}

// ------------------------------------------------------------------ evaluator ABI
int azd_evaluator_create_mlp(azd_evaluator **out, int device, int max_batch, int state_dim, int action_dim,
                             const int *hidden, int n_hidden, int final_act, const azd_adam_config *adam,
                             uint64_t seed) {
    if (!out || max_batch <= 0 || state_dim <= 0 || action_dim <= 0 || n_hidden < 0 || (n_hidden && !hidden) || !adam)
        return AZD_ERR_INVALID_ARGUMENT;
    int st = azd::device_ok(device);
    if (st) return st;
    azd_evaluator *ev = azd::make_mlp_evaluator(device, max_batch, state_dim, action_dim, hidden, n_hidden, final_act, adam, seed, &st);
    if (!ev) return st;
    *out = ev;
    return AZD_OK;
}
int azd_evaluator_create_trivial(azd_evaluator **out, int device, int state_dim, int action_dim) {
    if (!out) return AZD_ERR_INVALID_ARGUMENT;
    int st = azd::device_ok(device);
    if (st) return st;
    auto *ev = new (std::nothrow) azd::TrivialEvaluator();
    if (!ev) return AZD_ERR_OUT_OF_MEMORY;
    ev->device = device;
    ev->state_dim = state_dim;
    ev->action_dim = action_dim;
    *out = ev;
    return AZD_OK;
}
int azd_evaluator_create_hash_stream(azd_evaluator **out, int device, int state_dim, int action_dim, uint64_t seed,
                                     uint64_t first_agent) {
    if (!out) return AZD_ERR_INVALID_ARGUMENT;
    int st = azd::device_ok(device);
    if (st) return st;
    auto *ev = new (std::nothrow) azd::HashStreamEvaluator();
    if (!ev) return AZD_ERR_OUT_OF_MEMORY;
    ev->device = device;
    ev->state_dim = state_dim;
    ev->action_dim = action_dim;
    ev->seed = seed;
    ev->first_agent = first_agent;
    *out = ev;
    return AZD_OK;
}
int azd_evaluator_destroy(azd_evaluator *ev) {
    delete ev;
    return AZD_OK;
}
int azd_evaluator_write_predictions(azd_evaluator *ev, int batch, const float *states, float *predictions) {
    if (!ev || batch <= 0 || !states || !predictions) return AZD_ERR_INVALID_ARGUMENT;
    int st = ev->ensure_staging(batch);
    if (st) return st;
    size_t sb = (size_t)batch * ev->state_dim * 4, pb = (size_t)batch * ev->action_dim * 4;
    AZD_HIP(hipMemcpyAsync(ev->d_states, states, sb, hipMemcpyHostToDevice, ev->own_stream));
    // TrivialModel must leave the caller's buffer untouched: seed the staging copy with it
    AZD_HIP(hipMemcpyAsync(ev->d_preds, predictions, pb, hipMemcpyHostToDevice, ev->own_stream));
    st = ev->write_predictions_dev(batch, ev->d_states, ev->d_preds, ev->own_stream);
    if (st) return st;
    AZD_HIP(hipMemcpyAsync(predictions, ev->d_preds, pb, hipMemcpyDeviceToHost, ev->own_stream));
    AZD_HIP(hipStreamSynchronize(ev->own_stream));
    return AZD_OK;
}
int azd_evaluator_update_model(azd_evaluator *ev, int batch, const float *states, const float *observations,
                               const float *action_weights, float *loss) {
    if (!ev || batch <= 0 || !states || !observations || !action_weights) return AZD_ERR_INVALID_ARGUMENT;
    int st = ev->ensure_staging(batch);
    if (st) return st;
    size_t sb = (size_t)batch * ev->state_dim * 4, pb = (size_t)batch * ev->action_dim * 4;
    AZD_HIP(hipMemcpyAsync(ev->d_states, states, sb, hipMemcpyHostToDevice, ev->own_stream));
    AZD_HIP(hipMemcpyAsync(ev->d_obs, observations, pb, hipMemcpyHostToDevice, ev->own_stream));
    AZD_HIP(hipMemcpyAsync(ev->d_w, action_weights, pb, hipMemcpyHostToDevice, ev->own_stream));
    return ev->update_model_dev(batch, ev->d_states, ev->d_obs, ev->d_w, loss, ev->own_stream);
}
int azd_evaluator_write_predictions_dev(azd_evaluator *ev, int batch, const float *d_states, float *d_predictions,
                                        void *stream) {
    if (!ev || batch <= 0) return AZD_ERR_INVALID_ARGUMENT;
    AZD_HIP(hipSetDevice(ev->device));
    return ev->write_predictions_dev(batch, d_states, d_predictions, (hipStream_t)stream);
}
int azd_evaluator_update_model_dev(azd_evaluator *ev, int batch, const float *d_states, const float *d_observations,
                                   const float *d_action_weights, float *loss, void *stream) {
    if (!ev || batch <= 0) return AZD_ERR_INVALID_ARGUMENT;
    AZD_HIP(hipSetDevice(ev->device));
    return ev->update_model_dev(batch, d_states, d_observations, d_action_weights, loss, (hipStream_t)stream);
}
int azd_debug_hash_stream_via_evaluators(azd_evaluator *ev, int on) { return ev ? ev->debug_serve_from_pool(on) : AZD_ERR_INVALID_ARGUMENT; }
int azd_evaluator_set_weight_storage(azd_evaluator *ev, int dtype) { return ev ? ev->set_weight_storage(dtype) : AZD_ERR_INVALID_ARGUMENT; }
int64_t azd_evaluator_num_params(azd_evaluator *ev) { return ev ? ev->num_params() : 0; }
int azd_evaluator_get_params(azd_evaluator *ev, float *out) { return ev && out ? ev->get_params(out) : AZD_ERR_INVALID_ARGUMENT; }
int azd_evaluator_set_params(azd_evaluator *ev, const float *in) { return ev && in ? ev->set_params(in) : AZD_ERR_INVALID_ARGUMENT; }
uint64_t azd_evaluator_calls(azd_evaluator *ev) { return ev ? ev->calls : 0; }

// ------------------------------------------------------------------ engine ABI
int azd_engine_create(azd_engine **out, const azd_engine_config *cfg, azd_evaluator *ev) {
    if (!out || !cfg) return AZD_ERR_INVALID_ARGUMENT;
    const bool ramsey = cfg->space_id == AZD_SPACE_RAMSEY;
    const bool dense = cfg->space_id == AZD_SPACE_DENSE;
    if (dense) {
        if (cfg->n < 4 || cfg->n > AZD_DENSE_MAX_N || cfg->batch <= 0 || cfg->layers > 1 || cfg->max_slots < 0 || cfg->max_slots > AZD_DENSE_MAX_SLOTS ||
            !(cfg->dense_p >= 0.f && cfg->dense_p <= 1.f)) {
            azd::g_last_error = "unsupported dense-graph space (need 4 <= n <= 64, no Layered wrapper, max_slots <= 1024, 0 <= dense_p <= 1)";
            return AZD_ERR_INVALID_ARGUMENT;
        }
    } else if (ramsey) {
        bool ok = cfg->batch > 0 && cfg->n >= 3 && cfg->n <= AZD_RAMSEY_MAX_N && cfg->n_colors >= 2 && cfg->n_colors <= 4;
        if (ok) {
            for (int c = 0; c < cfg->n_colors; ++c) ok = ok && cfg->clique_sizes[c] >= 2 && cfg->clique_sizes[c] <= 5;
            ok = ok && azd::ramsey_edges(cfg->n) <= 256 && azd::ramsey_key_words(cfg->n, cfg->n_colors) <= azd::MAX_KW;
        }
        if (!ok) {
            azd::g_last_error = "unsupported Ramsey space (need 3 <= n, E <= 256, 2..4 colours, clique sizes 2..5, E*C <= 384)";
            return AZD_ERR_INVALID_ARGUMENT;
        }
    } else if (cfg->space_id != AZD_SPACE_C21 || cfg->n < 4 || cfg->n > AZD_C21_MAX_N || cfg->batch <= 0) {
        azd::g_last_error = "unsupported space / n / batch";
        return AZD_ERR_INVALID_ARGUMENT;
    }
    if (cfg->layers < 0 || cfg->layers > 8) {
        azd::g_last_error = "layers must be 0/1 (plain space) or 2..8";
        return AZD_ERR_INVALID_ARGUMENT;
    }
    if (cfg->path_kind != AZD_PATH_SET && cfg->path_kind != AZD_PATH_SEQUENCE) {
        azd::g_last_error = "unknown path encoding";
        return AZD_ERR_INVALID_ARGUMENT;
    }
    int st = azd::device_ok(cfg->device);
    if (st) return st;
    AZD_HIP(hipSetDevice(cfg->device));
    azd_engine *e = new (std::nothrow) azd_engine();
    if (!e) return AZD_ERR_OUT_OF_MEMORY;
    e->cfg = *cfg;
    e->ev = ev;
    azd::Arenas &a = e->a;
    memset(&a, 0, sizeof(a));
    a.n = cfg->n;
    a.B = cfg->batch;
    a.space = dense ? azd::SPACE_DENSE : ramsey ? azd::SPACE_RAMSEY : azd::SPACE_C21;
    a.path_kind = cfg->path_kind;
    a.layers = cfg->layers > 1 ? cfg->layers : 1;
    if (dense) {
        a.E = azd::dense_edges(cfg->n);
        a.A = azd::dense_action_dim(cfg->n);
        a.S = azd::dense_state_dim(cfg->n);
        {   // device keys: ranks of the root's modifiable slots, in 2 / 4 / 10 / 16 words (the widths dense_kernels.hip is built for)
            const int ms = cfg->max_slots > 0 ? cfg->max_slots : 128;
            a.KW = ms <= 128 ? 2 : ms <= 256 ? 4 : ms <= 640 ? 10 : 16;
            e->dense_slots = 64 * a.KW;
        }
        a.dense_p24 = (uint32_t)((cfg->dense_p > 0.f ? (double)cfg->dense_p : 0.2) * 16777216.0 + 0.5);
        a.eval_slope = azd::c21_eval_slope(cfg->n);
        e->kw_host = azd::dense_key_words(cfg->n);
    } else if (ramsey) {
        a.C = cfg->n_colors;
        a.E = azd::ramsey_edges(cfg->n);
        a.A = azd::ramsey_action_dim(cfg->n, a.C);
        a.S = azd::ramsey_state_dim(cfg->n, a.C);
        a.KW = azd::ramsey_key_words(cfg->n, a.C);
        for (int c = 0; c < a.C; ++c) {
            a.sizes[c] = cfg->clique_sizes[c];
            a.cweights[c] = cfg->color_weights[c];
        }
    } else {
        a.A = azd::c21_action_dim(cfg->n);
        a.S = azd::c21_state_dim(cfg->n);
        a.KW = azd::c21_key_words(cfg->n);
        a.eval_slope = azd::c21_eval_slope(cfg->n);
        azd::c21_lambda_bracket(cfg->n, &a.lam_lo, &a.lam_hi);
    }
    a.S_inner = a.S;
    a.S = a.S_inner * a.layers; // Layered<L, Space>::STATE_DIM (nabla/space/mod.rs:53)
    if (ev && (ev->state_dim != a.S || ev->action_dim != a.A)) {
        delete e;
        azd::g_last_error = "evaluator dimensions do not match the space";
        return AZD_ERR_INVALID_ARGUMENT;
    }
    a.node_cap = cfg->node_capacity > 0 ? (uint32_t)cfg->node_capacity : 4096u;
    a.arc_cap = cfg->arc_capacity > 0 ? (uint32_t)cfg->arc_capacity : 8192u;
    a.pred_cap = cfg->prediction_capacity > 0 ? (uint32_t)cfg->prediction_capacity : 32768u;
    a.ht_cap = (uint32_t)next_pow2((int)(2 * a.node_cap));
    // what the packed records hold (engine_types.h: PredRec / node_pack): child node 16 bits, arc id 16 (0xFFFF = none), a child's first
    // prediction 20, action id 12, actions per node 11, n_t 20 (one per arc at most), exhausted 12 (one per action at most).  The
    // reference's u32 indices (petgraph NodeIndex / EdgeIndex, tree/mod.rs:28-32) have no such limits: a configuration beyond them is
    // refused here, naming the argument, instead of corrupting a neighbouring bit field later.
    static_assert(AZD_DENSE_MAX_SLOTS <= 2047 && azd::MAX_NODE_ACTIONS <= 2047, "actions per node must fit PredRec's 11-bit count");
    static_assert(AZD_MAX_NODE_CAPACITY == 65536 && AZD_MAX_ARC_CAPACITY == 65535 && AZD_MAX_PREDICTION_CAPACITY == (1 << 20), "limits of the packed records");
    {
        const char *bad = nullptr;
        static thread_local char msg[256];
        const long per_node = dense ? (long)e->dense_slots : ramsey ? (long)a.E * (a.C - 1) : (long)a.A;
        if (a.node_cap > (uint32_t)AZD_MAX_NODE_CAPACITY) snprintf(msg, sizeof msg, "node_capacity %u is beyond the record format (<= %d)", a.node_cap, AZD_MAX_NODE_CAPACITY), bad = msg;
        else if (a.arc_cap > (uint32_t)AZD_MAX_ARC_CAPACITY) snprintf(msg, sizeof msg, "arc_capacity %u is beyond the record format (<= %d)", a.arc_cap, AZD_MAX_ARC_CAPACITY), bad = msg;
        else if (a.pred_cap > (uint32_t)AZD_MAX_PREDICTION_CAPACITY) snprintf(msg, sizeof msg, "prediction_capacity %u is beyond the record format (<= %d)", a.pred_cap, AZD_MAX_PREDICTION_CAPACITY), bad = msg;
        else if (a.A > 4096) snprintf(msg, sizeof msg, "ACTION_DIM %d is beyond the record format (<= 4096)", a.A), bad = msg;
        // (an upper bound of what a node can hold: c21 / Ramsey nodes are further limited to MAX_NODE_ACTIONS at run time, FLAG_NODE_ACTIONS)
        else if (per_node > 2047) snprintf(msg, sizeof msg, "%ld legal actions per node are beyond the record format (<= 2047)", per_node), bad = msg;
        if (bad) {
            azd::g_last_error = bad;
            delete e;
            return AZD_ERR_INVALID_ARGUMENT;
        }
    }
    const size_t B = (size_t)a.B;
#define TRY(x)              \
    do {                    \
        st = (x);           \
        if (st) {           \
            azd_engine_destroy(e); \
            return st;      \
        }                   \
    } while (0)
    {
        hipError_t he = hipStreamCreate(&e->stream);
        if (he != hipSuccess) {
            st = azd::hip_fail(he, "hipStreamCreate");
            delete e;
            return st;
        }
    }
    TRY(e->alloc(&a.nodes, B * a.node_cap));
    TRY(e->alloc(&a.keys, B * a.node_cap * a.KW));
    TRY(e->alloc(&a.arcs, B * a.arc_cap));
    TRY(e->alloc(&a.preds, B * a.pred_cap));
    TRY(e->alloc(&a.pred_g, B * a.pred_cap));
    TRY(e->alloc(&a.ht, B * a.ht_cap));
    TRY(e->alloc(&a.root_parents, B * azd::PARENTS_STRIDE));
    TRY(e->alloc(&a.cur_parents, B * azd::PARENTS_STRIDE));
    TRY(e->alloc(&a.root_perm, B * a.KW));
    TRY(e->alloc(&a.cur_perm, B * a.KW));
    TRY(e->alloc(&a.cur_path, B * a.KW));
    TRY(e->alloc(&a.cur_lambda, B));
    TRY(e->alloc(&a.cur_mu, B));
    TRY(e->alloc(&a.state_pos, B));
    TRY(e->alloc(&a.n_nodes, B));
    TRY(e->alloc(&a.n_arcs, B));
    TRY(e->alloc(&a.n_preds, B));
    TRY(e->alloc(&a.flags, B));
    a.fr_lds = (uint32_t)azd::FRONTIER_CAP;
    if (const char *v = getenv("AZD_DEBUG_FRONTIER_LDS")) { // test hook: a smaller LDS share, so that small trees reach the spill arena
        const int k = atoi(v);
        if (k >= 64 && k <= azd::FRONTIER_CAP && k % 64 == 0) a.fr_lds = (uint32_t)k;
    }
    if (ramsey) TRY(e->alloc(&a.fr_spill, B * 4 * a.node_cap)); // cascade frontier levels beyond the LDS's FRONTIER_CAP entries: (node, x) x node_cap x 2 levels (RamseySpace::FRONTIER_SPILL)
    TRY(e->alloc(&a.cand_c, B));
    TRY(e->alloc(&a.cand_node, B));
    TRY(e->alloc(&a.counters, B * azd::NUM_COUNTERS));
    TRY(e->alloc(&a.state_vecs, B * a.S));
    TRY(e->alloc(&a.h_theta, B * a.A));
    a.obs = a.h_theta; // the reference reuses h_theta_host as the observation buffer (optimizer/mod.rs:270-277)
    TRY(e->alloc(&a.weights, B * a.A));
    TRY(e->alloc(&a.argmin, 1));
    TRY(e->alloc(&a.status, 1));
    if (a.layers > 1) TRY(e->alloc(&a.cur_seq, B * azd::MAX_NODE_ACTIONS));
    TRY(e->alloc(&a.cur_stack, B * azd::PATH_STACK));
    if (dense) {
        TRY(e->alloc(&a.root_adj, B * 64));
        TRY(e->alloc(&a.cur_adj, B * 64));
        TRY(e->alloc(&a.root_aid, B * (size_t)e->dense_slots));
        TRY(e->alloc(&e->d_stage_slots, B * (size_t)((a.E + 63) / 64)));
        TRY(e->alloc(&a.argmin_d, 1));
        TRY(e->alloc(&a.node_mate, B * (size_t)a.node_cap * 64));
        // a bf16 evaluator takes the state vectors as bf16 rows: this space's kernels write them beside the f32 rows (write_vec16)
        if (ev && ev->input16_pitch() >= a.S) {
            a.S16 = ev->input16_pitch();
            TRY(e->alloc(&a.state_vecs16, B * (size_t)a.S16));
            if (hipMemset(a.state_vecs16, 0, B * (size_t)a.S16 * 2) != hipSuccess) return AZD_ERR_HIP;
        }
    }
    if (ramsey) {
        TRY(e->alloc(&a.root_nbr, B * 128));
        TRY(e->alloc(&a.cur_nbr, B * 128));
        TRY(e->alloc(&a.root_counts, B * (size_t)a.C * a.E));
        TRY(e->alloc(&a.cur_counts, B * (size_t)a.C * a.E));
        TRY(e->alloc(&a.root_tot, B * 4));
        TRY(e->alloc(&a.cur_tot, B * 4));
        TRY(e->alloc(&a.argmin_r, 1));
    }
    e->persist_enabled = (cfg->flags & AZD_ENGINE_NO_PERSISTENT_STEP) == 0;
    e->barrier_step = (cfg->flags & AZD_ENGINE_BARRIER_STEP) != 0;
    // default: the pool step for populations of 256 agents and more (below that a row's trip to an evaluator CU and
    // back costs more than the asynchronous step's in-workgroup evaluator: 0.36 against 0.49 M expansions/s at 64 agents,
    // 4.7 against 3.7 at 512); AZD_ENGINE_POOL_STEP / AZD_ENGINE_ASYNC_STEP / AZD_ENGINE_BARRIER_STEP force a form
    e->graph_enabled = getenv("AZD_NO_CALL_GRAPH") == nullptr;
    e->pool_step = (cfg->flags & AZD_ENGINE_POOL_STEP) != 0 ||
                   ((cfg->flags & (AZD_ENGINE_ASYNC_STEP | AZD_ENGINE_BARRIER_STEP)) == 0 && cfg->batch >= 256);
    if (const char *env = getenv("AZD_STEP_FORM")) { // experiments: override the configured form
        if (!strcmp(env, "pool")) e->pool_step = true, e->barrier_step = false;
        else if (!strcmp(env, "async")) e->pool_step = false, e->barrier_step = false;
        else if (!strcmp(env, "barrier")) e->pool_step = false, e->barrier_step = true;
    }
    {
        hipDeviceProp_t prop;
        e->n_cus = hipGetDeviceProperties(&prop, cfg->device) == hipSuccess ? prop.multiProcessorCount : 256;
        uint32_t qcap = 128;
        while (qcap < 2u * (uint32_t)a.B) qcap <<= 1;
        e->pool.qcap = qcap;
        e->pool_slot_words = (size_t)4 * azd::POOL_XCDS * qcap; // two ready lanes + two evaluator lanes
        TRY(e->alloc(&e->pool.ctl, 1));
        TRY(e->alloc(&e->pool.ready_slots, e->pool_slot_words));
        e->pool.eval_slots = e->pool.ready_slots + (size_t)2 * azd::POOL_XCDS * qcap;
        TRY(e->alloc(&e->pool.join, B));
        TRY(e->alloc(&e->pool.pend, B));
        TRY(e->alloc(&e->pool.stamp, B));
        TRY(e->alloc(&e->pool.post_call, B));
        TRY(e->alloc(&e->d_resume, B));
        TRY(e->alloc(&e->d_call_ctr, (size_t)azd_engine::MAX_SUBS));
        if (dense) {
            TRY(e->alloc(&e->d_ext_rows, B * azd_engine::EXT_STREAMS));
            TRY(e->alloc(&e->d_ext_home, B * azd_engine::EXT_STREAMS));
            TRY(e->alloc(&e->d_ext_n, (size_t)azd_engine::EXT_STREAMS));
            TRY(e->alloc(&e->d_ext_t0, (size_t)azd_engine::EXT_STREAMS));
        }
    }
    e->log_calls = 1024;
    {
        const size_t n_wg = (B + 15) / 16;
        TRY(e->alloc(&e->d_pargs, 1));
        hipError_t he2 = hipHostMalloc((void **)&e->h_pargs, sizeof(azd::PersistArgs));
        if (he2 != hipSuccess) {
            st = azd::hip_fail(he2, "hipHostMalloc");
            azd_engine_destroy(e);
            return st;
        }
        TRY(e->alloc(&e->d_log_key, (size_t)e->log_calls * n_wg));
        TRY(e->alloc(&e->d_log_node, (size_t)e->log_calls * n_wg));
        TRY(e->alloc(&e->pool.win_count, (size_t)e->log_calls));
        TRY(e->alloc(&e->d_argmin_side, 1));
        if (ramsey) TRY(e->alloc(&e->d_argmin_r_side, 1));
        // what the kernel hands the host in the middle of a launch: fine-grained (host-coherent) pinned memory
        hipError_t he3 = hipHostMalloc((void **)&e->h_win_log, (size_t)e->log_calls * sizeof(unsigned long long), hipHostMallocCoherent);
        if (he3 == hipSuccess) he3 = hipHostMalloc((void **)&e->h_win_flag, (size_t)e->log_calls * sizeof(uint32_t), hipHostMallocCoherent);
        if (he3 != hipSuccess) {
            st = azd::hip_fail(he3, "hipHostMalloc");
            azd_engine_destroy(e);
            return st;
        }
        e->pool.win_log = e->h_win_log;
        e->pool.win_flag = e->h_win_flag;
    }
    TRY(e->alloc(&e->d_stage_parents, B * (size_t)(dense ? 8 * a.n : ramsey ? a.E : a.n)));
    TRY(e->alloc(&e->d_stage_perm, B * (size_t)(dense ? 17 * a.KW : a.KW)));
    {
        hipError_t he = hipHostMalloc((void **)&e->h_status, sizeof(azd::StatusRec));
        if (he == hipSuccess) he = hipHostMalloc((void **)&e->h_argmin, sizeof(azd::ArgminRec));
        if (he != hipSuccess) {
            st = azd::hip_fail(he, "hipHostMalloc");
            azd_engine_destroy(e);
            return st;
        }
    }
    hipError_t he = hipMemsetAsync(a.counters, 0, B * azd::NUM_COUNTERS * 8, e->stream);
    if (he == hipSuccess) he = hipMemsetAsync(a.status, 0, sizeof(azd::StatusRec), e->stream);
    if (he == hipSuccess) he = hipMemsetAsync(a.argmin, 0, sizeof(azd::ArgminRec), e->stream);
    if (he == hipSuccess) he = hipMemsetAsync(a.flags, 0, B * 4, e->stream);
    if (he == hipSuccess) he = hipMemsetAsync(a.h_theta, 0, B * a.A * 4, e->stream);
    if (he == hipSuccess) he = hipMemsetAsync(a.state_vecs, 0, B * a.S * 4, e->stream); // vec![0.; ..] (optimizer/mod.rs:65)
    if (he == hipSuccess) he = hipMemsetAsync(a.weights, 0, B * a.A * 4, e->stream);
    if (he == hipSuccess) he = hipStreamSynchronize(e->stream);
    if (he != hipSuccess) {
        st = azd::hip_fail(he, "engine init memset");
        azd_engine_destroy(e);
        return st;
    }
#undef TRY
    *out = e;
    return AZD_OK;
}

int azd_engine_destroy(azd_engine *e) {
    if (!e) return AZD_OK;
    (void)hipSetDevice(e->cfg.device);
    if (e->stream) (void)hipStreamSynchronize(e->stream);
    for (void *p : e->allocs) (void)hipFree(p);
    if (e->h_status) (void)hipHostFree(e->h_status);
    if (e->h_argmin) (void)hipHostFree(e->h_argmin);
    if (e->h_pargs) (void)hipHostFree(e->h_pargs);
    if (e->h_win_log) (void)hipHostFree(e->h_win_log);
    if (e->h_win_flag) (void)hipHostFree(e->h_win_flag);
    if (e->d_pool) (void)hipFree(e->d_pool);
    for (void *p : {(void *)e->d_grp_tile, (void *)e->d_grp_lds, (void *)e->d_grp_desc, (void *)e->d_grp_cnt, (void *)e->d_grp_x, (void *)e->d_grp_flag})
        if (p) (void)hipFree(p);
    if (e->call_graph) (void)hipGraphExecDestroy(e->call_graph);
    for (int i = 0; i < azd_engine::MAX_SUBS; ++i) {
        if (e->sub_graph[i]) (void)hipGraphExecDestroy(e->sub_graph[i]);
        if (e->sub_join[i]) (void)hipEventDestroy(e->sub_join[i]);
        if (e->sub_stream[i]) (void)hipStreamDestroy(e->sub_stream[i]);
    }
    if (e->sub_fork) (void)hipEventDestroy(e->sub_fork);
    if (e->ext_done) (void)hipEventDestroy(e->ext_done);
    if (e->ext_fork) (void)hipEventDestroy(e->ext_fork);
    for (int x = 0; x < azd_engine::EXT_STREAMS; ++x) {
        if (e->ext_graph[x]) (void)hipGraphExecDestroy(e->ext_graph[x]);
        for (int i = 0; i < azd_engine::EXT_IN_FLIGHT; ++i)
            if (e->ext_ring[x][i]) (void)hipEventDestroy(e->ext_ring[x][i]);
        if (e->ext_stream[x]) (void)hipStreamDestroy(e->ext_stream[x]);
    }
    for (auto &it : e->ev_inflight) {
        (void)hipEventDestroy(it.second.first);
        (void)hipEventDestroy(it.second.second);
    }
    for (hipEvent_t ev : e->ev_pool) (void)hipEventDestroy(ev);
    if (e->stream) (void)hipStreamDestroy(e->stream);
    delete e;
    return AZD_OK;
}

// optimizer/mod.rs:61-70
int azd_engine_par_new_begin(azd_engine *e, const uint8_t *parents, const uint64_t *permitted) {
    if (!e || !parents || !permitted) return AZD_ERR_INVALID_ARGUMENT;
    AZD_ENTER(e);
    int st = upload_roots(e, parents, permitted);
    if (st) return st;
    const azd::Arenas &a = e->a;
    AZD_HIP(hipMemsetAsync(a.counters, 0, (size_t)a.B * azd::NUM_COUNTERS * 8, e->stream));
    e->counters_by_wave = false;
    AZD_HIP(hipMemsetAsync(a.status, 0, sizeof(azd::StatusRec), e->stream));
    e->seen_improved = 0;
    azd::launch_init_roots(a, e->d_stage_parents, e->d_stage_perm, e->stream);
    AZD_HIP(hipMemsetAsync(a.h_theta, 0, (size_t)a.B * a.A * 4, e->stream)); // vec![0.; ..] at :71
    AZD_HIP(hipGetLastError());
    return AZD_OK;
}
static int new_finish(azd_engine *e) {
    azd::launch_add_actions(e->a, 1, e->stream);
    azd::launch_argmin(e->a, 1, e->stream);
    e->initialised = true;
    return sync_status(e);
}
// optimizer/mod.rs:74-101
int azd_engine_par_new_end(azd_engine *e, const float *h_theta) {
    if (!e || !h_theta) return AZD_ERR_INVALID_ARGUMENT;
    AZD_ENTER(e);
    AZD_HIP(hipMemcpyAsync(e->a.h_theta, h_theta, (size_t)e->a.B * e->a.A * 4, hipMemcpyHostToDevice, e->stream));
    return new_finish(e);
}
int azd_engine_par_new(azd_engine *e, const uint8_t *parents, const uint64_t *permitted) {
    if (!e) return AZD_ERR_INVALID_ARGUMENT;
    if (!e->ev) return AZD_ERR_NO_EVALUATOR;
    int st = azd_engine_par_new_begin(e, parents, permitted);
    if (st) return st;
    st = run_evaluator(e); // :72
    if (st) return st;
    return new_finish(e);
}

// optimizer/mod.rs:159-174
int azd_engine_roll_out_begin(azd_engine *e, const uint32_t *tol, int n_tol, uint32_t dflt) {
    if (!e || !e->initialised) return AZD_ERR_INVALID_ARGUMENT;
    AZD_ENTER(e);
    azd::TolTable t;
    int st = fill_tol(t, tol, n_tol, dflt);
    if (st) return st;
    e->time_begin(0);
    azd::launch_rollout(e->a, t, e->stream);
    e->time_end();
    return sync_status(e);
}
// optimizer/mod.rs:177-190
int azd_engine_roll_out_end(azd_engine *e, const float *h_theta, int *improved) {
    if (!e || !h_theta || !e->initialised) return AZD_ERR_INVALID_ARGUMENT;
    AZD_ENTER(e);
    AZD_HIP(hipMemcpyAsync(e->a.h_theta, h_theta, (size_t)e->a.B * e->a.A * 4, hipMemcpyHostToDevice, e->stream));
    azd::launch_add_actions(e->a, 0, e->stream);
    azd::launch_argmin(e->a, 0, e->stream);
    int st = sync_status(e);
    if (improved) *improved = (int)(e->h_status->improved - e->seen_improved);
    e->seen_improved = e->h_status->improved;
    return st;
}
// Evaluator groups (pool_step.inc: pool_eval_group): which member keeps which column tile of which layer in its LDS.
// Greedy by size: the largest tiles first, each to the member that holds the fewest bytes so far and fewer than W tiles of that
// layer.  The smallest group (a multiple of 8 workgroups) and the fewest waves per slot that fit the LDS are taken: a batch's
// latency does not depend on g (a layer is one tile's MFMA chain per wave either way), the number of groups the CUs make does.
struct GroupPlan {
    int g = 0, W = 0;
    size_t lds_bytes = 0;      // the fullest member's
    uint32_t xstride = 0;      // floats per exchange buffer
    std::vector<int16_t> tile; // [L][g][W]
    std::vector<uint32_t> lds; // [L][g][W]
};
static bool plan_groups(const azd::FusedEval &fe, size_t lds_budget, int avail_wgs, int force_g, GroupPlan *out) {
    const int L = fe.n_layers;
    if (L < 1 || L > 7 || fe.bf16) return false;
    std::vector<int> tiles(L), steps(L);
    int max_tiles = 0;
    uint32_t xs = 0;
    for (int l = 0; l < L; ++l) {
        steps[l] = (fe.dims[l] + 15) / 16;
        tiles[l] = (fe.dims[l + 1] + 15) / 16;
        max_tiles = std::max(max_tiles, tiles[l]);
        if (l < L - 1) xs = std::max<uint32_t>(xs, (uint32_t)tiles[l] * 256u);
        if (l < L - 1 && fe.dims[l + 1] % 16 != 0) return false; // a hidden layer's output is the next layer's whole k-steps
    }
    // the fewest waves per slot first (W = 1: sixteen batch slots per group), then the smallest group that fits the LDS and leaves
    // room for two groups (config A: W = 1 needs g = 64 of the 192 CUs the searchers leave: 5.7 M expansions/s against 5.2 at g = 40, W = 2)
    for (int W : {1, 2, 4})
    for (int g = force_g > 0 ? force_g : 8; g <= avail_wgs && g <= 128; g += 8) {
        {
            if (force_g <= 0 && W < 4 && 2 * g > avail_wgs) break; // (a lone group only as the last resort)
            if ((max_tiles + g - 1) / g > W) {
                if (force_g > 0) break; // (a forced size is tried with every W, never enlarged)
                continue;
            }
            struct Item { int l, t; size_t bytes; };
            std::vector<Item> items;
            for (int l = 0; l < L; ++l)
                for (int t = 0; t < tiles[l]; ++t) items.push_back({l, t, (size_t)steps[l] * 1024});
            std::stable_sort(items.begin(), items.end(), [](const Item &x, const Item &y) { return x.bytes > y.bytes; });
            std::vector<size_t> load((size_t)g, 0);
            std::vector<int> cnt((size_t)L * g, 0);
            std::vector<int16_t> tile((size_t)L * g * W, (int16_t)-1);
            std::vector<uint32_t> lds((size_t)L * g * W, 0u);
            bool ok = true;
            for (const Item &it : items) {
                int best = -1;
                for (int m = 0; m < g; ++m)
                    if (cnt[(size_t)it.l * g + m] < W && (best < 0 || load[(size_t)m] < load[(size_t)best])) best = m;
                if (best < 0 || load[(size_t)best] + it.bytes > lds_budget) {
                    ok = false;
                    break;
                }
                const int w = cnt[(size_t)it.l * g + best]++;
                tile[((size_t)it.l * g + best) * W + w] = (int16_t)it.t;
                lds[((size_t)it.l * g + best) * W + w] = (uint32_t)load[(size_t)best];
                load[(size_t)best] += it.bytes;
            }
            if (!ok) {
                if (force_g > 0) break;
                continue;
            }
            out->g = g;
            out->W = W;
            out->lds_bytes = *std::max_element(load.begin(), load.end());
            out->xstride = xs ? xs : 256u;
            out->tile = tile;
            out->lds = lds;
            return true;
        }
        if (force_g > 0) break;
    }
    return false;
}

static int pool_clear(azd_engine *e, const azd::PoolArgs &pool) { // empty queues, nobody claimed, no call done
    AZD_HIP(hipMemsetAsync(pool.ctl, 0, sizeof(azd::PoolCtl), e->stream));
    AZD_HIP(hipMemsetAsync(pool.ready_slots, 0, e->pool_slot_words * sizeof(uint32_t), e->stream));
    AZD_HIP(hipMemsetAsync(pool.join, 0, (size_t)e->a.B * sizeof(uint32_t), e->stream));
    e->pool_clean = true;
    return AZD_OK;
}

// What follows a pool launch of k calls on the host: its status block (the two sides' busy shares, for the split feedback) and
// -- when a wait ran into its bound -- the take-over.
// A wait that ran into its bound ends a pool launch instead of hanging it (PoolCtl::abort; k_argmin_log1 has put the flag into
// the status block and left the log alone).  The trees are consistent -- a wave never leaves an agent inside a call -- so the
// asynchronous step, whose workgroups need no company, takes the launch over where every agent stands, and this engine stays
// with it (*took_over; the asynchronous step's LDS plan in *as_out / *ab_out).
static int pool_finish_launch(azd_engine *e, const azd::FusedEval &fe, const azd::PoolArgs &pool, int k, bool *took_over,
                              uint32_t *as_out, size_t *ab_out) {
    *took_over = false;
    int st = fetch_status(e);
    if (st) return st;
    if (e->h_status->pool_ticks > 0 && !e->h_status->pool_abort) {
        const double T = (double)e->h_status->pool_ticks;
        e->pool_util_eval = e->pool_eval_wgs > 0 ? (double)e->h_status->pool_eval_busy / (T * e->pool_eval_wgs) : 0.0;
        e->pool_util_search = e->pool_search_waves > 0 ? (double)e->h_status->pool_search_busy / (T * e->pool_search_waves) : 0.0;
    }
    if (!e->h_status->pool_abort) return AZD_OK;
    e->pool_clean = false;
    e->log_clean = false; // (it holds the aborted launch's candidates, which the take-over's replay needs: not cleared here)
    e->pool_failed = true;
    e->pool_step = false;
    uint32_t as = 0;
    size_t ab = 0;
    const char *why_t = "";
    if (!azd::async_plan(e->a, fe, &as, &ab, &why_t)) {
        e->time_collect();
        azd::g_last_error = std::string("pool step: a queue wait ran into its bound, and the asynchronous step cannot take over: ") + why_t;
        return AZD_ERR_UNREACHABLE;
    }
    azd::launch_pool_resume_scan(e->a, pool, k, e->d_resume, e->stream);
    st = pool_clear(e, pool);
    if (st) return st;
    azd::StepLaunch sl;
    sl.n_calls = k;
    sl.log_key = e->d_log_key;
    sl.resume = e->d_resume;
    sl.ctl = nullptr;
    sl.hashed = fe.kind == 4;
    sl.window = 0;
    e->time_begin(0);
    azd::launch_async(e->a, e->d_pargs, sl, fe.params, fe.wpk, as, ab, e->stream);
    e->time_end();
    e->log_clean = true; // k_argmin_log1 has replayed and cleared it
    e->step_form = AZD_STEP_ASYNC;
    e->step_reason = "pool step aborted (a queue wait ran into its bound: no evaluator or searcher workgroup made progress); "
                     "the asynchronous step took the launch over and serves this engine from here on";
    *took_over = true;
    *as_out = as;
    *ab_out = ab;
    return AZD_OK;
}

// The pool step of the dense-graph space (BASELINE configs[4]): searcher workgroups only (k_pool_search) on part of the chip; the
// model -- too large for an evaluator workgroup's LDS -- is served by batched GEMM launches over the rows the searchers have posted,
// replayed from a graph on a second stream for as long as the searchers run.  Agents advance independently, so a launch no longer
// lasts as long as its slowest agent per call (launch-per-phase form: a roll-out launch took 1.2 ms where the mean agent needed 0.06).
// *ran = false: the form cannot run here (why in e->step_reason) and the caller takes the launch-per-phase form.
// *left_out: calls still to run when an aborted launch had to be completed by the launch-per-phase kernels (the caller runs them).
static int dense_pool_run(azd_engine *e, const azd::TolTable &t, int n_calls, bool *ran, int *left_out) {
    *ran = false;
    *left_out = 0;
    const azd::Arenas &a = e->a;
    const char *why = "";
    uint32_t dyn_stride = 0;
    size_t dyn_bytes = 0;
    // wavefronts per searcher workgroup: 16, or as many as the LDS holds (roots of more than 640 slots: 12 -- a wave's block and its
    // selection scratch are 12 KB there)
    int waves = 16;
    if (const char *env = getenv("AZD_DENSE_POOL_WAVES")) {
        waves = atoi(env);
        if (waves < 4 || waves > 16) { // (round-4 verdict, 7c: a knob out of range is refused, not silently bent)
            azd::g_last_error = "AZD_DENSE_POOL_WAVES must be 4..16 (wavefronts per searcher workgroup of the dense-graph pool step)";
            return AZD_ERR_INVALID_ARGUMENT;
        }
    }
    while (waves > 4 && !azd::dense_pool_plan(a, waves, &dyn_stride, &dyn_bytes, &why)) waves -= 1;
    azd::FusedEval fe;
    const bool hashed = e->ev->fused_desc(&fe) && fe.kind == 4; // the test harness' fixed prediction stream, served like a model's rows
    if (!e->pool_step || !e->persist_enabled || e->pool_failed || e->ext_unsupported || (!a.state_vecs16 && !hashed) ||
        n_calls < 1 || !azd::dense_pool_plan(a, waves, &dyn_stride, &dyn_bytes, &why)) {
        e->step_reason = !e->pool_step || !e->persist_enabled ? "dense-graph space: the pool step is not configured for this engine"
                         : e->pool_failed                      ? "an earlier pool launch of this engine aborted: launch-per-phase form"
                         : ((!a.state_vecs16 && !hashed) || e->ext_unsupported)
                             ? "dense-graph space: the pool step needs an evaluator that serves gathered bf16 rows (ActionModel with bf16 storage)"
                             : why;
        return AZD_OK;
    }
    const int per_cu = azd::dense_pool_search_resident(a, waves, dyn_bytes);
    // searcher workgroups: no more waves than twice the agents, and no more than half the chip's wave slots -- the GEMM launches
    // need the rest
    int n_search = (2 * a.B + waves - 1) / waves;
    if (n_search > e->n_cus * 8 / waves) n_search = e->n_cus * 8 / waves;
    if (const char *env = getenv("AZD_DENSE_POOL_SEARCH_WGS")) n_search = atoi(env) > 0 ? atoi(env) : n_search;
    // The evaluator of this form is a stream of GEMM LAUNCHES beside the searchers' persistent kernel: they run only where a CU has
    // LDS and registers left, and a searcher workgroup (1024 threads' worth of LDS blocks) leaves none.  A setting of the two knobs
    // that lets the searchers cover more than three quarters of the CUs used to be accepted and then cost a 4-s wait bound, an
    // abort and the engine's demotion to the launch-per-phase form (gpurun_out/ew.txt, round 4): refused up front instead.
    // (workgroups, not wave slots: the dispatcher deals one workgroup to every CU before it doubles up, so 256 workgroups of 8 waves
    // sit on 256 CUs although two would fit one)
    if (per_cu >= 1 && (getenv("AZD_DENSE_POOL_SEARCH_WGS") || getenv("AZD_DENSE_POOL_WAVES")) && n_search > e->n_cus - e->n_cus / 4) {
        static thread_local char msg[256];
        snprintf(msg, sizeof msg, "AZD_DENSE_POOL_SEARCH_WGS / AZD_DENSE_POOL_WAVES: %d searcher workgroups of %d waves would sit on %d of %d CUs; "
                 "the evaluator's GEMM launches need at least a quarter of the chip free (at most %d workgroups)", n_search, waves,
                 n_search < e->n_cus ? n_search : e->n_cus, e->n_cus, e->n_cus - e->n_cus / 4);
        azd::g_last_error = msg;
        return AZD_ERR_INVALID_ARGUMENT;
    }
    if (per_cu < 1 || n_search > e->n_cus * per_cu * 7 / 8) n_search = per_cu < 1 ? 0 : e->n_cus * per_cu * 7 / 8;
    if (n_search < 1) {
        e->step_reason = "dense-graph space: the device holds no searcher workgroup of the pool step";
        return AZD_OK;
    }
    // (one stream by default: with two, each batch is half as large and takes as long -- the GEMMs' k loops are latency-bound at these
    // batch sizes and the streams share the same CUs -- so an agent's cycle, which sets the rate, gets no shorter: 18.8 M expansions/s
    // with one stream against 17.8 with two at config E)
    int n_ext = 1;
    if (const char *env = getenv("AZD_DENSE_POOL_STREAMS")) n_ext = atoi(env);
    n_ext = n_ext < 1 ? 1 : n_ext > azd_engine::EXT_STREAMS ? azd_engine::EXT_STREAMS : n_ext;
    if (!e->ext_done) AZD_HIP(hipEventCreateWithFlags(&e->ext_done, hipEventDisableTiming));
    if (!e->ext_fork) AZD_HIP(hipEventCreateWithFlags(&e->ext_fork, hipEventDisableTiming));
    for (int x = 0; x < n_ext; ++x) {
        if (!e->ext_stream[x]) AZD_HIP(hipStreamCreateWithFlags(&e->ext_stream[x], hipStreamNonBlocking));
        for (int i = 0; i < azd_engine::EXT_IN_FLIGHT; ++i)
            if (!e->ext_ring[x][i]) AZD_HIP(hipEventCreateWithFlags(&e->ext_ring[x][i], hipEventDisableTiming));
    }
    azd::PoolArgs pool = e->pool;
    pool.n_eval = 0;
    pool.ready_lanes = 0;
    pool.n_express = 0;
    pool.express_waves = 0;
    pool.express_shift = 0;
    pool.early_post = 1; // the request leaves with the row: the GEMMs run while the wave computes lambda_1 and the matching
    if (const char *env = getenv("AZD_POOL_EARLY_POST")) pool.early_post = atoi(env);
    pool.eval_stride = pool.eval_out_off = 0;
    pool.eval_rows = 0;
    pool.debug_abort_call = 0;
    if (const char *env = getenv("AZD_POOL_DEBUG_ABORT_CALL")) pool.debug_abort_call = (uint32_t)atoi(env); // test hook: k_ext_deliver
    // the evaluator's graphs, one per stream: collect, the layers over the collected rows, hand back.  Every stream may find the
    // whole population posted, so each has row lists and activation rows of its own.
    if (!hashed) {
        const int st_r = e->ev->ensure_rows(n_ext * a.B);
        if (st_r) return st_r;
    }
    int rounds = 4; // collect / layers / hand-back rounds per replay of the evaluator's graph
    if (const char *env = getenv("AZD_DENSE_POOL_ROUNDS")) rounds = atoi(env);
    rounds = rounds < 1 ? 1 : rounds > 16 ? 16 : rounds;
    const uint64_t layout = e->ev->layout_version + (hashed ? 1ull << 63 : 0ull) + ((uint64_t)rounds << 56);
    if (e->ext_graph_n != n_ext || e->ext_graph_layout != layout) {
        for (int x = 0; x < azd_engine::EXT_STREAMS; ++x)
            if (e->ext_graph[x]) {
                (void)hipGraphExecDestroy(e->ext_graph[x]);
                e->ext_graph[x] = nullptr;
            }
        e->ext_graph_n = 0;
        for (int x = 0; x < n_ext; ++x) {
            uint32_t *rows = e->d_ext_rows + (size_t)x * a.B, *home = e->d_ext_home + (size_t)x * a.B, *cnt = e->d_ext_n + x;
            hipGraph_t g = nullptr;
            AZD_HIP(hipStreamBeginCapture(e->ext_stream[x], hipStreamCaptureModeThreadLocal));
            int st_g = AZD_OK;
            // several rounds per graph: between two graphs on a stream the GPU idles ~18 us, between two kernels of one graph not at all
            for (int r = 0; r < rounds && st_g == AZD_OK; ++r) {
                azd::launch_ext_take(pool, rows, home, cnt, e->d_ext_t0 + x, e->ext_stream[x]);
                if (hashed) azd::launch_ext_hash_rows(e->d_pargs, rows, cnt, (uint32_t)a.B, a.h_theta, e->ext_stream[x]);
                else st_g = e->ev->write_predictions_gathered(rows, cnt, a.B, a.state_vecs16, a.S16, a.h_theta, e->ext_stream[x], x * a.B);
                azd::launch_ext_deliver(pool, a, rows, home, cnt, (uint32_t)a.B, e->d_ext_t0 + x, e->ext_stream[x]);
            }
            const hipError_t he = hipStreamEndCapture(e->ext_stream[x], &g);
            if (st_g) {
                if (g) (void)hipGraphDestroy(g);
                if (st_g != AZD_ERR_UNSUPPORTED) return st_g;
                e->ext_unsupported = true;
                e->step_reason = "dense-graph space: the pool step needs an evaluator that serves gathered bf16 rows (ActionModel with bf16 storage)";
                return AZD_OK;
            }
            if (he != hipSuccess) return azd::hip_fail(he, "hipStreamEndCapture");
            const hipError_t hi = hipGraphInstantiate(&e->ext_graph[x], g, nullptr, nullptr, 0);
            (void)hipGraphDestroy(g);
            if (hi != hipSuccess) return azd::hip_fail(hi, "hipGraphInstantiate");
        }
        e->ext_graph_n = n_ext;
        e->ext_graph_layout = layout;
    }
    if (!hashed) {
        memset(&fe, 0, sizeof(fe));
        fe.kind = 3; // what the searchers look at: requests are posted for an evaluator
    }
    e->step_form = AZD_STEP_POOL;
    e->step_reason.clear();
    e->pool_eval_wgs = 0;
    e->pool_search_wgs = n_search;
    e->pool_search_waves = n_search * waves;
    e->ext_iterations = 0;
    int left = n_calls;
    while (left > 0) {
        const int k = left < e->log_calls ? left : e->log_calls;
        {   // the argument block (re-sent only when it changed)
            azd::PersistArgs now;
            memset(&now, 0, sizeof(now));
            now.a = a;
            now.tol = t;
            now.ev = fe;
            now.ev.call_base = hashed ? e->ev->calls : 0; // hash stream: index of the launch's first call
            now.pool = pool;
            if (!e->pargs_valid || memcmp(&e->pargs_sent, &now, sizeof(now)) != 0) {
                AZD_HIP(hipStreamSynchronize(e->stream));
                memcpy(e->h_pargs, &now, sizeof(now));
                AZD_HIP(hipMemcpyAsync(e->d_pargs, e->h_pargs, sizeof(azd::PersistArgs), hipMemcpyHostToDevice, e->stream));
                memcpy(&e->pargs_sent, &now, sizeof(now));
                e->pargs_valid = true;
            }
        }
        int st = pool_clear(e, pool); // (always: a collect that ran past the end of the last launch may have touched the control block)
        if (st) return st;
        if (!e->log_clean) {
            AZD_HIP(hipMemsetAsync(e->d_log_key, 0xFF, (size_t)e->log_calls * sizeof(unsigned long long), e->stream));
            e->log_clean = true;
        }
        AZD_HIP(hipMemsetAsync(e->d_ext_n, 0, sizeof(uint32_t) * azd_engine::EXT_STREAMS, e->stream));
        AZD_HIP(hipEventRecord(e->ext_fork, e->stream));
        for (int x = 0; x < n_ext; ++x) AZD_HIP(hipStreamWaitEvent(e->ext_stream[x], e->ext_fork, 0)); // the first collect sees the cleared queues
        azd::StepLaunch sl;
        sl.n_calls = k;
        sl.log_key = e->d_log_key;
        sl.resume = nullptr;
        sl.ctl = pool.ctl;
        sl.hashed = hashed ? 1 : 0;
        sl.window = 0;
        e->time_begin(0);
        azd::dense_launch_pool_search(a, e->d_pargs, sl, n_search, waves, dyn_stride, dyn_bytes, e->stream);
#ifndef AZD_PHASE_PROFILE
        e->counters_by_wave = true;
#endif
        e->time_end();
        AZD_HIP(hipGetLastError());
        AZD_HIP(hipEventRecord(e->ext_done, e->stream));
        e->pool_clean = false;
        // the evaluator: replayed, stream after stream, until the searchers (and the argmin replay behind them) are through; ext_depth
        // replays queued per stream at a time -- a replay that finds nothing posted costs a few microseconds
        unsigned long long it = 0;
        int depth = e->ext_depth;
        if (const char *env = getenv("AZD_DENSE_POOL_DEPTH")) depth = atoi(env);
        depth = depth < 1 ? 1 : depth > azd_engine::EXT_IN_FLIGHT ? azd_engine::EXT_IN_FLIGHT : depth;
        for (;;) {
            const hipError_t q = hipEventQuery(e->ext_done);
            if (q == hipSuccess) break;
            if (q != hipErrorNotReady) return azd::hip_fail(q, "hipEventQuery");
            const int x = (int)(it % (unsigned long long)n_ext);
            const unsigned long long round = it / (unsigned long long)n_ext;
            hipEvent_t slot = e->ext_ring[x][round % (unsigned long long)depth];
            if (round >= (unsigned long long)depth) AZD_HIP(hipEventSynchronize(slot));
            AZD_HIP(hipGraphLaunch(e->ext_graph[x], e->ext_stream[x]));
            AZD_HIP(hipEventRecord(slot, e->ext_stream[x]));
            it += 1;
        }
        for (int x = 0; x < n_ext; ++x) AZD_HIP(hipStreamSynchronize(e->ext_stream[x]));
        e->ext_iterations += it;
        if (getenv("AZD_DENSE_POOL_DEBUG")) fprintf(stderr, "dense pool: %d calls, %d searcher workgroups, %llu evaluator replays\n", k, n_search, it);
        left -= k;
        e->ev->calls += (uint64_t)k;
        st = fetch_status(e);
        if (st) return st;
        if (e->h_status->pool_ticks > 0) {
            const double T = (double)e->h_status->pool_ticks;
            e->pool_util_eval = (double)e->h_status->pool_eval_busy / (T * n_ext); // the share of the launch an evaluator stream held a batch
            e->pool_util_search = (double)e->h_status->pool_search_busy / (T * e->pool_search_waves);
            // (A controller on these two shares, as for the in-kernel evaluator, was measured and not kept: over whole epochs the rate is flat
            // between 112 and 160 searcher workgroups -- config E 19.7-20.0 M expansions/s, 612-slot roots 10.2-10.3 M.)
        }
        if (e->h_status->pool_abort) {
            // A wait ran into its bound (nothing made progress for 4 s: the GEMM launches never got CUs, say).  The trees are
            // consistent -- a wave never leaves an agent inside a call -- but the agents stand at different calls.  No other
            // CU-resident form of this space exists to take the launch over, so the launch-per-phase kernels complete it: agent by
            // agent from where each one stands (k_pool_resume_scan), the ones that are through sitting out (FLAG_PARKED), every
            // candidate logged under the call it really belongs to, one replay of the log at the end -- the results of an
            // undisturbed launch.  This engine stays with the launch-per-phase form.
            e->pool_failed = true;
            e->log_clean = false;
            azd::launch_pool_resume_scan(a, pool, k, e->d_resume, e->stream);
            std::vector<uint32_t> res((size_t)a.B);
            AZD_HIP(hipMemcpyAsync(res.data(), e->d_resume, res.size() * 4, hipMemcpyDeviceToHost, e->stream));
            AZD_HIP(hipStreamSynchronize(e->stream));
            st = pool_clear(e, pool);
            if (st) return st;
            int rounds = 0;
            bool any_pending = false;
            for (uint32_t r : res) {
                const int rem = k - (int)(r & 0x7FFFFFFFu);
                rounds = rem > rounds ? rem : rounds;
                any_pending = any_pending || (r >> 31) != 0u;
            }
            auto evaluate_rows = [&]() -> int {
                if (hashed) { // the fixed stream's rows depend on the call: not reproducible outside the launch that posted them
                    azd::g_last_error = "dense pool step aborted with the test harness' prediction stream: no recovery";
                    return AZD_ERR_UNREACHABLE;
                }
                const uint64_t calls_before = e->ev->calls;
                const int s2 = e->ev->write_predictions_dev16(a.B, a.state_vecs, a.state_vecs16, a.S16, a.h_theta, e->stream);
                e->ev->calls = calls_before;
                return s2;
            };
            if (any_pending) { // the rows that were still due, and add_actions for the nodes waiting for them
                azd::launch_park(a, e->d_resume, k, -1, 1, e->stream);
                st = evaluate_rows();
                if (st) return st;
                azd::launch_add_actions(a, 0, e->stream);
            }
            for (int r = 0; r < rounds; ++r) {
                azd::launch_park(a, e->d_resume, k, r, 1, e->stream);
                azd::launch_rollout(a, t, e->stream);
                st = evaluate_rows();
                if (st) return st;
                azd::launch_add_actions(a, 0, e->stream);
                azd::launch_log_candidates_resume(a, e->d_log_key, e->d_resume, k, r, e->stream);
            }
            azd::launch_park(a, e->d_resume, k, 0, 0, e->stream); // everyone back
            azd::launch_argmin_log(a, k, e->d_log_key, e->stream);  // replays the k calls and leaves the log clean
            e->log_clean = true;
            AZD_HIP(hipGetLastError());
            st = fetch_status(e);
            if (st) return st;
            e->step_form = AZD_STEP_PER_CALL;
            e->step_reason = "dense pool step aborted (a queue wait ran into its bound: the searchers or the evaluator's launches made no "
                             "progress); the launch-per-phase kernels completed the launch and serve this engine from here on";
            *ran = true;
            *left_out = left;
            return AZD_OK;
        }
    }
    *ran = true;
    return AZD_OK;
}

// optimizer/mod.rs:159-190, n_calls times.  ahead: azd_engine_run_ahead -- the launch is left running and its calls are handed
// out by window_serve; *accepted = 0 when this engine's step form cannot do that (nothing is launched then).
static int roll_out_impl(azd_engine *e, const azd::TolTable &t, int n_calls, int *improved, bool ahead, int *accepted) {
    int st = AZD_OK;
    if (accepted) *accepted = 0;
    if (e->a.space == azd::SPACE_DENSE && !getenv("AZD_DENSE_NO_POOL")) {
        if (ahead) return AZD_OK; // (the evaluator's launches need this thread: nothing can run ahead of the host)
        bool ran = false;
        int left_after = 0;
        st = dense_pool_run(e, t, n_calls, &ran, &left_after);
        if (st) return st;
        if (ran && left_after == 0) {
            st = check_status(e);
            if (improved) *improved = (int)(e->h_status->improved - e->seen_improved);
            e->seen_improved = e->h_status->improved;
            return st;
        }
        if (ran) n_calls = left_after; // (an aborted launch was completed call by call: what is left runs the same way, below)
    }
    const std::string dense_reason = e->a.space == azd::SPACE_DENSE ? e->step_reason : std::string();
    azd::FusedEval fe;
    uint32_t dyn_stride = 0;
    size_t dyn_bytes = 0;
    const bool fusable = e->persist_enabled && e->ev->fused_desc(&fe);
    const char *why_a = "", *why_b = "";
    const char *why_p = "";
    azd::PoolArgs pool = e->pool;
    bool use_pool = fusable && e->pool_step && azd::pool_plan(e->a, fe, &pool, &dyn_stride, &dyn_bytes, &why_p);
    bool use_async = fusable && !use_pool && !e->barrier_step && azd::async_plan(e->a, fe, &dyn_stride, &dyn_bytes, &why_a);
    bool use_barrier = fusable && !use_pool && !use_async && azd::persist_plan(e->a, fe, &dyn_stride, &dyn_bytes, &why_b);
    // which form runs is part of the result a caller may want to check (azd_engine_step_form): the launch-per-phase
    // form is several times slower than the CU-resident ones
    e->step_form = use_pool ? AZD_STEP_POOL : use_async ? AZD_STEP_ASYNC : use_barrier ? AZD_STEP_BARRIER : AZD_STEP_PER_CALL;
    e->step_reason.clear();
    if (e->pool_step && fusable && !use_pool) e->step_reason = std::string(why_p) + "; ";
    if (!use_async && !use_pool) {
        if (!e->persist_enabled) e->step_reason = "AZD_ENGINE_NO_PERSISTENT_STEP";
        else if (!fusable) e->step_reason = "the evaluator cannot run inside the kernel (external model, more than 7 layers, or a layer width that is not a multiple of 4)";
        else if (e->barrier_step) e->step_reason = "AZD_ENGINE_BARRIER_STEP";
        else e->step_reason += why_a;
        if (fusable && !use_barrier && *why_b) e->step_reason += std::string("; ") + why_b;
    }
    if (e->pool_failed && use_async)
        e->step_reason = "an earlier pool launch of this engine aborted (a queue wait ran into its bound): asynchronous step";
    int pool_blocks = 0;
    bool status_fresh = false; // h_status already holds the status behind the last launch
    if (use_pool) {
        // Split of the CUs between the two roles, in proportion to the CU time a row costs an evaluator and a call costs
        // the searchers.  AZD_POOL_EVAL_WGS / AZD_POOL_SEARCH_WGS override (experiments).
        const int B = e->a.B;
        int n_eval = 0;
        if (fe.kind >= 3) {
            double flop = 0;
            if (fe.kind == 3)
                for (int l = 0; l < fe.n_layers; ++l) flop += 2.0 * fe.dims[l] * fe.dims[l + 1];
            else flop = 2.0 * 256.0 * (e->a.S + 512.0 + e->a.A); // hashed rows (test harness): the split a 3 x 256 model would get
            // measured (profiles/r02_pool_probe.txt): a 16-row fp32 batch of the 3 x 256 MLP takes 25 us of an evaluator
            // CU, bf16 storage 16 us; pure search 4.65 us of a CU per call on young trees.  Whole epochs (older, larger
            // trees) want a few more evaluators than that ratio says: best 88-100 of 256 at 4096 agents fp32, 80 at 8192,
            // 56 at 8192 bf16 (gpurun sweeps r2c/sw_*), which the constants below reproduce
            const double eval_us_per_row = flop / (256.0 * 2400.0) / (fe.bf16 ? 1.85 : 1.0) / 0.33;
            // (a Ramsey call costs the searchers less: no lambda_1, five selections per expansion against eleven; best 126 of 256
            // for config D, 113 cost 10 %, 134 5 % -- gpurun sweep r2m/D_*)
            // (round 4: a c21 call costs the searchers a quarter less -- one round trip per selection level, 3.6 rounds of lambda_1 -- and the
            // measured-feedback split settles at 100-105 of 256 where it used to settle at 89-96; a first guess of 88 cost a process that
            // makes few launches -- the bench's 20-call window -- 8 %: 27.0 against 29.4 M expansions/s at 88 / 104 evaluator workgroups)
            const double search_us_per_call = e->a.space == azd::SPACE_RAMSEY ? 3.6 : 3.65;
            n_eval = (int)(e->n_cus * eval_us_per_row / (eval_us_per_row + search_us_per_call) + 0.5);
            // populations well beyond the searching waves keep the evaluator queues deep enough for 32-row batches (two row
            // tiles per weight fragment, where they fit the LDS), which cost an evaluator 1.2 us per row instead of 1.6:
            // best 76-84 of 256 at 8192 agents fp32 (88-92 at 4096), 50-58 at 8192 bf16 (gpurun sweeps r2m/B8_*, C_*)
            if (pool.eval_rows > 16 && B >= 6144) n_eval = (int)(n_eval * 0.91 + 0.5);
            // small populations leave CUs free (the searchers need no more waves than twice the agents): evaluators may take
            // them down to ~4 rows per batch -- rows then rarely queue behind a running batch (config A, 512 agents and a
            // 2.5 MFLOP model: 33 / 64 / 128 evaluator workgroups -> 2.70 / 2.92 / 3.09 M expansions/s; gpurun r2t/A_*)
            const int cap = (B + 3) / 4 + 1;
            n_eval = n_eval > cap ? cap : n_eval;
            {   // at most half the chip, or whatever the searchers of a small population leave
                const int want_search = (B + 7) / 8;
                const int most = want_search < e->n_cus / 2 ? e->n_cus - want_search : e->n_cus / 2;
                n_eval = n_eval > most ? most : n_eval;
            }
            n_eval = n_eval < 1 ? 1 : n_eval;
            if (pool_feedback_on() && e->pool_fb.n_eval > 0) { // the split the last launches' busy shares ask for, under the same caps
                const int fb = e->pool_fb.n_eval;
                const int want_search = (B + 7) / 8, most = want_search < e->n_cus / 2 ? e->n_cus - want_search : e->n_cus / 2;
                n_eval = fb > cap ? cap : fb;
                n_eval = n_eval > most ? most : n_eval;
                n_eval = n_eval < 1 ? 1 : n_eval;
            }
            // a SHORT launch is fill and drain (its length is its slowest agents' few calls, each with an evaluator round trip in it): a seventh
            // more evaluator workgroups than the busy shares of whole epochs ask for -- bench.py --steps 20, 4096 agents, same box, four
            // alternating runs each: 104 workgroups 27.7 / 28.6 / 30.1 / 29.4 M expansions/s, 116: 29.9 / 29.2 / 29.3 / 29.8, 124: 29.5 / 29.8 /
            // 29.6 / 29.6 (round 5; the launch over its median agent 1.50 -> 1.44 -> 1.37)
            if (n_calls <= 64 && fe.kind == 3) {
                const int want_search = (B + 7) / 8, most = want_search < e->n_cus / 2 ? e->n_cus - want_search : e->n_cus / 2;
                n_eval += n_eval / 7;
                n_eval = n_eval > most ? most : n_eval;
                n_eval = n_eval > cap ? cap : n_eval;
            }
            if (const char *env = getenv("AZD_POOL_EVAL_WGS")) n_eval = atoi(env) > 0 ? atoi(env) : n_eval;
        }
        int n_search = e->n_cus - n_eval;
        const int want = (B + 7) / 8; // no more waves than twice the agents: a wave without an agent only polls
        n_search = n_search > want ? want : n_search;
        n_search = n_search < 1 ? 1 : n_search;
        if (const char *env = getenv("AZD_POOL_SEARCH_WGS")) n_search = atoi(env) > 0 ? atoi(env) : n_search;
        // Searcher and evaluator workgroups spin-wait on each other: every one of them must be RESIDENT, whatever the overrides
        // above ask for and whatever the device can hold (a CU mask, a partition, another kernel's LDS).  Clamp the grid to the
        // co-resident capacity the runtime reports, evaluators first (they are dispatched first: a grid of evaluators alone
        // would never let a searcher in); with no room for one of each the call takes the asynchronous step instead.
        int capacity = azd::pool_max_resident(e->a, dyn_bytes, e->n_cus);
        if (const char *env = getenv("AZD_POOL_MAX_RESIDENT")) capacity = atoi(env); // tests: a device that holds fewer workgroups
        if (fe.kind < 3) n_eval = 0; // TrivialModel / in-wave hash stream: nothing to serve
        if (n_eval + n_search > capacity) {
            if (n_eval > capacity / 2) n_eval = capacity / 2;
            if (fe.kind >= 3 && n_eval < 1) n_eval = 1;
            n_search = capacity - n_eval;
        }
        if (n_search < 1 || capacity < 1) {
            use_pool = false;
            char buf[160];
            snprintf(buf, sizeof(buf), "pool step: the device holds %d of its workgroups at once; it needs an evaluator and a searcher resident together; ", capacity);
            e->step_reason = buf;
        }
        pool.n_eval = n_eval;
        // With more agents than searching waves an agent's cycle is mostly waiting for a wave (92 of 179 us at 8192 agents);
        // in lane mode the agents behind the mean progress are taken first, so that the slow chains do not also queue.
        // (lane mode: measured +4 % at 8192 agents fp32, -4 % at config C, +-0 at config D -- within run-to-run noise: those
        // populations are bound by the searchers' capacity, not by the order they are served in.  Off unless asked for.)
        pool.ready_lanes = 0;
        if (const char *env = getenv("AZD_POOL_READY_LANES")) pool.ready_lanes = atoi(env);
        // Express mode (default with a model evaluator and a chip-sized grid): one searcher workgroup per XCD serves only the
        // agents that lag behind the progress of their XCD's unfinished agents (by more than 1/16 of it), and every other wave
        // takes one of those first when more of them wait than express waves stand by.  A launch lasts as long as its slowest
        // agent's chain of calls; in that chain ~10 us per call were the agent waiting for a wave.  4096 agents fp32: 32.7-33.3
        // -> 34.1-34.6 M expansions/s over whole epochs, 25.4-25.7 -> 26.0 in a 20-call launch (gpurun r3 sweeps: 8 / 16 / 24
        // express workgroups, thresholds 1/4 .. 1/64, 4 / 8 / 16 waves each: 8-16 workgroups and 1/16-1/32 are the flat optimum,
        // the waves per workgroup do not matter -- what helps is the place in the queue, not the emptier CU).
        pool.n_express = (fe.kind >= 3 && n_search >= 64) ? 8 : 0;
        pool.express_waves = 8;
        pool.express_shift = 4;
        if (const char *env = getenv("AZD_POOL_EXPRESS_WGS")) pool.n_express = atoi(env);
        if (const char *env = getenv("AZD_POOL_EXPRESS_WAVES")) pool.express_waves = (uint32_t)atoi(env);
        if (const char *env = getenv("AZD_POOL_EXPRESS_SHIFT")) pool.express_shift = (uint32_t)atoi(env);
        if (pool.n_express > n_search / 2) pool.n_express = n_search / 2;
        if (pool.n_express < 0) pool.n_express = 0;
        if (pool.ready_lanes != 1) pool.ready_lanes = (pool.n_express > 0 && fe.kind >= 3) ? 2 : 0;
        if (pool.ready_lanes != 2) pool.n_express = 0;
        if (pool.express_waves < 1 || pool.express_waves > 16) pool.express_waves = 4;
        // Early post: the row leaves, and the evaluator is asked, before the wave computes the new node's cost and writes the
        // tree back (the agent is queued again by whoever is later, PoolArgs::join).  That takes ~15 us off an agent's cycle
        // and costs the wave a second drain of its stores (~1.5 us): +10 % where agents rarely wait for a wave (512..2048
        // agents, gpurun r2k/e_*), -3 % where the searchers' capacity is the bound (4096 agents and beyond).
        // Larger populations: only by a wave that had to wait for its agent (mode 2), which is the state of the last fifth of a
        // launch, when the slowest chains are all that is left.
        pool.early_post = (double)B <= 1.25 * n_search * 16 ? 1 : 2; // 16 waves per searcher workgroup
        // (a SHORT launch is fill and drain: its length is its slowest agent's few calls, not the searchers' capacity -- always early
        // there: 28.4 -> 29.05 M expansions/s in the driver's 20-call window, same-box A/B of two runs each, round 5)
        if (n_calls <= 64) pool.early_post = 1;
        if (const char *env = getenv("AZD_POOL_EARLY_POST")) pool.early_post = atoi(env);
        pool.debug_abort_call = 0;
        if (const char *env = getenv("AZD_POOL_DEBUG_ABORT_CALL")) pool.debug_abort_call = (uint32_t)atoi(env);
        // ---- evaluator groups (pool_eval_group): where a classic batch is long -- a model whose weights one CU streams in tens of
        // microseconds -- and the population small enough that a few groups carry its rows.  AZD_POOL_EVAL_GROUP = 0: never;
        // = g: groups of g workgroups whatever the model (tests, experiments).  f32 weight storage only.
        pool.grp_g = pool.grp_w = pool.grp_groups = 0;
        pool.grp_tile = nullptr;
        pool.grp_lds = nullptr;
        pool.grp_desc = pool.grp_cnt = nullptr;
        pool.grp_x = nullptr;
        pool.grp_flag = nullptr;
        pool.grp_xstride = 0;
        if (use_pool && fe.kind == 3 && !fe.bf16 && e->a.space == azd::SPACE_C21 && (size_t)B * (size_t)e->a.S * 4 < (1ull << 31)) {
            double wbytes = 0;
            for (int l = 0; l < fe.n_layers; ++l) wbytes += 4.0 * fe.dims[l] * fe.dims[l + 1];
            int force_g = 0;
            bool want = wbytes > 2.5e6 && B <= 1536; // (measured with config A's model: groups 5.7 / 9.8 / 11.0 M expansions/s at 512 / 1024 / 2048 agents, the classic form 3.5 / 6.8 / 13.2)
            if (const char *env = getenv("AZD_POOL_EVAL_GROUP")) {
                force_g = atoi(env);
                want = force_g > 0;
            }
            GroupPlan gpl;
            const int avail = capacity - n_search; // what the searchers leave (>= n_eval): idle CUs of a small population included
            if (want && plan_groups(fe, (size_t)160 * 1024 - 2048, avail, force_g, &gpl)) {
                const int n_groups = std::max(1, avail / gpl.g);
                const int NS = 16 / gpl.W;
                const size_t slots = (size_t)n_groups * NS;
                if (!e->d_grp_tile) {
                    AZD_HIP(hipMalloc(&e->d_grp_tile, (size_t)8 * 128 * 4 * sizeof(int16_t)));
                    AZD_HIP(hipMalloc(&e->d_grp_lds, (size_t)8 * 128 * 4 * sizeof(uint32_t)));
                }
                if (slots > e->grp_slots_alloc || slots * 2 * gpl.xstride > e->grp_x_floats || slots * gpl.g * 16 > e->grp_flag_words) {
                    AZD_HIP(hipStreamSynchronize(e->stream));
                    (void)hipFree(e->d_grp_desc);
                    (void)hipFree(e->d_grp_cnt);
                    (void)hipFree(e->d_grp_x);
                    (void)hipFree(e->d_grp_flag);
                    e->d_grp_flag = nullptr;
                    e->d_grp_desc = e->d_grp_cnt = nullptr;
                    e->d_grp_x = nullptr;
                    AZD_HIP(hipMalloc(&e->d_grp_desc, slots * 64 * 4));
                    AZD_HIP(hipMalloc(&e->d_grp_cnt, slots * 8 * 32 * 4));
                    AZD_HIP(hipMalloc(&e->d_grp_x, slots * 2 * gpl.xstride * 4));
                    AZD_HIP(hipMalloc(&e->d_grp_flag, slots * gpl.g * 16 * 4));
                    e->grp_flag_words = slots * gpl.g * 16;
                    e->grp_slots_alloc = slots;
                    e->grp_x_floats = slots * 2 * gpl.xstride;
                }
                if (gpl.tile != e->grp_tile_host || gpl.lds != e->grp_lds_host) {
                    AZD_HIP(hipStreamSynchronize(e->stream));
                    AZD_HIP(hipMemcpy(e->d_grp_tile, gpl.tile.data(), gpl.tile.size() * sizeof(int16_t), hipMemcpyHostToDevice));
                    AZD_HIP(hipMemcpy(e->d_grp_lds, gpl.lds.data(), gpl.lds.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
                    e->grp_tile_host = gpl.tile;
                    e->grp_lds_host = gpl.lds;
                }
                pool.grp_g = gpl.g;
                pool.grp_w = gpl.W;
                pool.grp_groups = n_groups;
                pool.grp_tile = e->d_grp_tile;
                pool.grp_lds = e->d_grp_lds;
                pool.grp_desc = e->d_grp_desc;
                pool.grp_cnt = e->d_grp_cnt;
                pool.grp_x = e->d_grp_x;
                pool.grp_flag = e->d_grp_flag;
                pool.grp_xstride = gpl.xstride;
                n_eval = n_groups * gpl.g; // every evaluator workgroup is a group member
                pool.n_eval = n_eval;
                if (gpl.lds_bytes > dyn_bytes) dyn_bytes = gpl.lds_bytes;
            }
        }
        pool_blocks = n_eval + n_search;
        e->pool_grp_g = pool.grp_g;
        e->pool_grp_groups = pool.grp_groups;
        e->pool_grp_w = pool.grp_w;
        e->pool_eval_wgs = n_eval;
        e->pool_search_wgs = n_search;
        e->pool_search_waves = (n_search - pool.n_express) * 16 + pool.n_express * (int)pool.express_waves;
    }
    if (e->pool_step && fusable && !use_pool) { // the pool step was wanted and cannot run: the next form down
        use_async = !e->barrier_step && azd::async_plan(e->a, fe, &dyn_stride, &dyn_bytes, &why_a);
        use_barrier = !use_async && azd::persist_plan(e->a, fe, &dyn_stride, &dyn_bytes, &why_b);
        if (!use_async) e->step_reason += why_a;
        if (!use_async && !use_barrier && *why_b) e->step_reason += std::string("; ") + why_b;
    }
    e->step_form = use_pool ? AZD_STEP_POOL : use_async ? AZD_STEP_ASYNC : use_barrier ? AZD_STEP_BARRIER : AZD_STEP_PER_CALL;
    if (e->a.space == azd::SPACE_DENSE) // why the space's pool step did not run: the launch-per-phase form follows
        e->step_reason = dense_reason.empty() ? "AZD_DENSE_NO_POOL: the dense-graph space's pool step was switched off" : dense_reason;
    if (ahead) { // only the pool step publishes its calls while it runs, one launch's worth of them
        const bool ok = use_pool && fe.kind >= 3 && n_calls >= 1 && n_calls <= e->log_calls && !e->timing;
        if (!ok) return AZD_OK; // a hint: the calls run when they are asked for
        *accepted = 1;
    }
    if (use_pool || use_async || use_barrier) {
        // CU-resident forms: the whole call chain, n_calls times, in one launch per <= log_calls calls.  What a launch costs
        // the host: the argument block is re-sent only when it changed (per-launch values are kernel arguments), the log and
        // the pool's control block are left clean by k_argmin_log1, the abort flag comes back with the status block.
        auto send_args = [&](const azd::PoolArgs &pl) -> int {
            azd::PersistArgs &pa = e->pargs_sent;
            azd::PersistArgs now;
            memset(&now, 0, sizeof(now)); // (padding bytes compare equal)
            now.a = e->a;
            now.tol = t;
            now.ev = fe;
            now.ev.call_base = (fe.kind == 2 || fe.kind == 4) ? e->ev->calls : 0; // hash stream: index of the launch's first call
            now.pool = pl;
            if (e->pargs_valid && memcmp(&pa, &now, sizeof(now)) == 0) return AZD_OK;
            AZD_HIP(hipStreamSynchronize(e->stream)); // the pinned block may still be in flight from the copy before
            memcpy(e->h_pargs, &now, sizeof(now));
            AZD_HIP(hipMemcpyAsync(e->d_pargs, e->h_pargs, sizeof(azd::PersistArgs), hipMemcpyHostToDevice, e->stream));
            memcpy(&pa, &now, sizeof(now));
            e->pargs_valid = true;
            return AZD_OK;
        };
        const bool fb_on = use_pool && fe.kind == 3 && pool_feedback_on() && n_calls >= 100 && !getenv("AZD_POOL_EVAL_WGS");
        int left = n_calls;
        while (left > 0) {
            const int k = left < e->log_calls ? left : e->log_calls;
            status_fresh = false;
            st = send_args(pool);
            if (st) return st;
            if (use_pool && !e->pool_clean) {
                st = pool_clear(e, pool);
                if (st) return st;
            }
            if (use_pool && pool.grp_g > 0) { // batch numbers and arrival counts of the groups' slots start from zero in every launch
                const size_t slots = (size_t)pool.grp_groups * (16 / pool.grp_w);
                AZD_HIP(hipMemsetAsync(pool.grp_desc, 0, slots * 64 * 4, e->stream));
                AZD_HIP(hipMemsetAsync(pool.grp_cnt, 0, slots * 8 * 32 * 4, e->stream));
                AZD_HIP(hipMemsetAsync(pool.grp_flag, 0, slots * pool.grp_g * 16 * 4, e->stream));
            }
            if (!use_barrier && !e->log_clean) {
                AZD_HIP(hipMemsetAsync(e->d_log_key, 0xFF, (size_t)e->log_calls * sizeof(unsigned long long), e->stream));
                e->log_clean = true;
            }
            if (ahead) {
                // the window's hand-over words (no launch that writes them is in flight: a window is drained before the next opens),
                // and the argmin record the window starts from: its cost for the host's per-call compare, a copy for argmin_data
                AZD_HIP(hipMemsetAsync(pool.win_count, 0, (size_t)e->log_calls * sizeof(uint32_t), e->stream));
                AZD_HIP(hipMemcpyAsync(e->d_argmin_side, e->a.argmin, sizeof(azd::ArgminRec), hipMemcpyDeviceToDevice, e->stream));
                if (e->d_argmin_r_side)
                    AZD_HIP(hipMemcpyAsync(e->d_argmin_r_side, e->a.argmin_r, sizeof(azd::RamseyArgminRec), hipMemcpyDeviceToDevice, e->stream));
                AZD_HIP(hipMemcpyAsync(e->h_argmin, e->a.argmin, sizeof(azd::ArgminRec), hipMemcpyDeviceToHost, e->stream));
                st = fetch_status(e); // (synchronises)
                if (st) return st;
                memset(e->h_win_flag, 0, (size_t)e->log_calls * sizeof(uint32_t));
                __atomic_thread_fence(__ATOMIC_SEQ_CST);
            }
            azd::StepLaunch sl;
            sl.n_calls = k;
            sl.log_key = e->d_log_key;
            sl.resume = nullptr;
            sl.ctl = use_pool ? pool.ctl : nullptr;
            sl.hashed = fe.kind == 4;
            sl.window = ahead ? 1 : 0;
            sl.groups = (use_pool && pool.grp_g > 0) ? 1 : 0;
            e->time_begin(0);
            if (use_pool) {
                azd::launch_pool(e->a, e->d_pargs, sl, fe.params, fe.wpk, pool_blocks, dyn_stride, dyn_bytes, e->stream);
#ifndef AZD_PHASE_PROFILE
                e->counters_by_wave = true;
#endif
            } else if (use_async) azd::launch_async(e->a, e->d_pargs, sl, fe.params, fe.wpk, dyn_stride, dyn_bytes, e->stream);
            else {
                azd::launch_persist(e->a, e->d_pargs, sl, e->d_log_node, dyn_stride, dyn_bytes, e->stream);
                e->log_clean = false;
            }
            e->time_end();
            left -= k;
            if (ahead) { // the launch is on its way; window_serve / window_drain do the rest
                AZD_HIP(hipGetLastError());
                e->ev->calls += (uint64_t)k;
                azd_engine::Window &w = e->win;
                w = azd_engine::Window();
                w.open = true;
                w.n = k;
                w.tol = t;
                w.best_ord = host_ordf(e->h_argmin->eval);
                w.base_improved = e->h_status->improved;
                w.fe = fe;
                w.pool = pool;
                if (improved) *improved = 0;
                return AZD_OK;
            }
            if (use_pool) {
                bool took_over = false;
                st = pool_finish_launch(e, fe, pool, k, &took_over, &dyn_stride, &dyn_bytes);
                if (st) return st;
                status_fresh = left == 0 && !took_over; // the last launch's status is in, and nothing ran behind it
                if (took_over) {
                    use_pool = false;
                    use_async = true;
                }
            }
            e->ev->calls += (uint64_t)k;
        }
        AZD_HIP(hipGetLastError());
        if (fb_on && use_pool && status_fresh && e->h_status->pool_ticks > 0) { // the launch is over and its busy shares are in
            // where the best split sits: at equal shares when the agents hardly outnumber the searching waves (their cycle, not the
            // chip, sets the pace: configs A, B, the 384 x 384 model), at u_e - u_s = +0.06 when they queue for waves (8192 agents:
            // a slightly starved evaluator side fills 32-row batches, which cost it a quarter less per row)
            const double crowd = e->pool_search_waves > 0 ? (double)e->a.B / e->pool_search_waves - 1.5 : 0.0;
            const double d = e->pool_util_eval - e->pool_util_search - 0.06 * (crowd < 0 ? 0.0 : crowd > 1 ? 1.0 : crowd);
            // (large steps while the first guess is being corrected; afterwards at most 6 workgroups per launch, so that one disturbed
            // launch -- another tenant's burst on the box, a first dispatch under a profiler -- cannot carry the split far: a run whose
            // warm-up launch moved it from 89 to 102 evaluators stayed 7 % low for the two launches it had left to walk back)
            const int cur = e->pool_eval_wgs, lim = e->pool_fb.updates < 2 ? (cur / 4 > 4 ? cur / 4 : 4) : 6;
            e->pool_fb.updates += 1;
            int mv = (int)(100.0 * d + (d >= 0 ? 0.5 : -0.5));
            mv = mv > lim ? lim : mv < -lim ? -lim : mv;
            int next = cur + mv;
            next = next > e->n_cus / 2 ? e->n_cus / 2 : next < 1 ? 1 : next;
            e->pool_fb.n_eval = next;
        }
    } else {
        // One call = roll-out, model call, add_actions, argmin: four to eight launches.  With the MLP evaluator (whose
        // launches take no per-call arguments) the sequence is captured once into a hipGraph and replayed per call, so
        // a call costs one graph launch instead of a launch per phase (BASELINE configs[4]: "hipGraph-captured episode
        // step").  Not while per-launch timing is on (the events would be captured too), nor for evaluators whose
        // kernels take the call index.
        const bool graphable = e->graph_enabled && !e->timing && n_calls >= 2 && e->ev->replayable(e->a.B);
        // Sub-populations: a roll-out launch lasts as long as its slowest agent (config E: 1.2 ms against a mean agent's 0.06), so
        // the population is cut into S parts, each with a stream and a graph of its own (roll-out, model rows, add_actions,
        // candidates into the per-call log); the parts run n_calls calls independently -- trees never exchange data and the
        // model is constant -- and one part's kernels fill the CUs another part's stragglers leave idle.  The per-call argmin
        // is replayed from the log afterwards, as in the CU-resident forms.  Needs an evaluator whose rows may run concurrently.
        // (Config E, 8192 agents: 1 / 2 / 4 / 8 parts -> 11.7 / 12.4 / 8.0 / 6.0 M expansions/s: every part adds seven graph nodes per
        // call for the host to enqueue, and beyond two parts the host, not the GPU, sets the pace.)
        int n_subs = (e->a.B >= 2048 && e->ev->rows_concurrent()) ? 2 : 1;
        if (const char *env = getenv("AZD_PER_CALL_STREAMS")) n_subs = atoi(env);
        if (n_subs > azd_engine::MAX_SUBS) n_subs = azd_engine::MAX_SUBS;
        if (n_subs > 1 && (!e->ev->rows_concurrent() || e->a.B > 65536 || e->a.node_cap > 65536)) n_subs = 1;
        if (graphable && n_subs > 1) {
            const int per = (e->a.B + n_subs - 1) / n_subs;
            if (!e->sub_fork) AZD_HIP(hipEventCreateWithFlags(&e->sub_fork, hipEventDisableTiming));
            for (int i = 0; i < n_subs; ++i) {
                if (!e->sub_stream[i]) AZD_HIP(hipStreamCreateWithFlags(&e->sub_stream[i], hipStreamNonBlocking));
                if (!e->sub_join[i]) AZD_HIP(hipEventCreateWithFlags(&e->sub_join[i], hipEventDisableTiming));
            }
            if (e->sub_graph_n != n_subs || memcmp(&e->call_graph_tol, &t, sizeof(t)) != 0 || e->call_graph_layout != e->ev->layout_version) {
                for (int i = 0; i < azd_engine::MAX_SUBS; ++i)
                    if (e->sub_graph[i]) {
                        (void)hipGraphExecDestroy(e->sub_graph[i]);
                        e->sub_graph[i] = nullptr;
                    }
                e->sub_graph_n = 0;
                for (int i = 0; i < n_subs; ++i) {
                    azd::Arenas as = e->a;
                    as.t0 = i * per;
                    as.tn = e->a.B - as.t0 < per ? e->a.B - as.t0 : per;
                    if (as.tn <= 0) continue;
                    hipGraph_t g = nullptr;
                    AZD_HIP(hipStreamBeginCapture(e->sub_stream[i], hipStreamCaptureModeThreadLocal));
                    azd::launch_rollout(as, t, e->sub_stream[i]);
                    st = e->ev->write_predictions_rows(as.t0, as.tn, e->a.state_vecs, e->a.state_vecs16, e->a.S16, e->a.h_theta, e->sub_stream[i]);
                    azd::launch_add_actions(as, 0, e->sub_stream[i]);
                    azd::launch_log_candidates(as, e->d_log_key, e->d_call_ctr + i, e->sub_stream[i]);
                    hipError_t he = hipStreamEndCapture(e->sub_stream[i], &g);
                    if (st) {
                        if (g) (void)hipGraphDestroy(g);
                        return st;
                    }
                    if (he != hipSuccess) return azd::hip_fail(he, "hipStreamEndCapture");
                    he = hipGraphInstantiate(&e->sub_graph[i], g, nullptr, nullptr, 0);
                    (void)hipGraphDestroy(g);
                    if (he != hipSuccess) return azd::hip_fail(he, "hipGraphInstantiate");
                }
                e->sub_graph_n = n_subs;
                e->call_graph_tol = t;
                e->call_graph_layout = e->ev->layout_version;
                if (e->call_graph) { // (the single-stream graph was captured for another tol table or layout)
                    (void)hipGraphExecDestroy(e->call_graph);
                    e->call_graph = nullptr;
                }
            }
            int left = n_calls;
            while (left > 0) {
                const int k = left < e->log_calls ? left : e->log_calls;
                if (!e->log_clean) {
                    AZD_HIP(hipMemsetAsync(e->d_log_key, 0xFF, (size_t)e->log_calls * sizeof(unsigned long long), e->stream));
                    e->log_clean = true;
                }
                AZD_HIP(hipMemsetAsync(e->d_call_ctr, 0, sizeof(uint32_t) * azd_engine::MAX_SUBS, e->stream));
                AZD_HIP(hipEventRecord(e->sub_fork, e->stream));
                for (int i = 0; i < n_subs; ++i)
                    if (e->sub_graph[i]) AZD_HIP(hipStreamWaitEvent(e->sub_stream[i], e->sub_fork, 0));
                for (int c = 0; c < k; ++c)
                    for (int i = 0; i < n_subs; ++i)
                        if (e->sub_graph[i]) AZD_HIP(hipGraphLaunch(e->sub_graph[i], e->sub_stream[i]));
                for (int i = 0; i < n_subs; ++i)
                    if (e->sub_graph[i]) {
                        AZD_HIP(hipEventRecord(e->sub_join[i], e->sub_stream[i]));
                        AZD_HIP(hipStreamWaitEvent(e->stream, e->sub_join[i], 0));
                    }
                azd::launch_argmin_log(e->a, k, e->d_log_key, e->stream); // replays the k calls and leaves the log clean
                left -= k;
            }
            e->ev->calls += (uint64_t)n_calls;
            e->step_form = AZD_STEP_PER_CALL_GRAPH;
        } else if (graphable) {
            if (!e->call_graph || e->sub_graph_n != 0 || memcmp(&e->call_graph_tol, &t, sizeof(t)) != 0 || e->call_graph_layout != e->ev->layout_version) {
                e->sub_graph_n = 0; // (the sub-population graphs, if any, belong to another tol table or layout from here on)
                if (e->call_graph) (void)hipGraphExecDestroy(e->call_graph);
                e->call_graph = nullptr;
                hipGraph_t g = nullptr;
                AZD_HIP(hipStreamBeginCapture(e->stream, hipStreamCaptureModeThreadLocal));
                azd::launch_rollout(e->a, t, e->stream);
                const uint64_t calls_before = e->ev->calls;
                st = e->ev->write_predictions_dev16(e->a.B, e->a.state_vecs, e->a.state_vecs16, e->a.S16, e->a.h_theta, e->stream);
                e->ev->calls = calls_before;
                azd::launch_add_actions(e->a, 0, e->stream);
                azd::launch_argmin(e->a, 0, e->stream);
                hipError_t he = hipStreamEndCapture(e->stream, &g);
                if (st) {
                    if (g) (void)hipGraphDestroy(g);
                    return st;
                }
                if (he != hipSuccess) return azd::hip_fail(he, "hipStreamEndCapture");
                he = hipGraphInstantiate(&e->call_graph, g, nullptr, nullptr, 0);
                (void)hipGraphDestroy(g);
                if (he != hipSuccess) return azd::hip_fail(he, "hipGraphInstantiate");
                e->call_graph_tol = t;
                e->call_graph_layout = e->ev->layout_version;
            }
            for (int c = 0; c < n_calls; ++c) AZD_HIP(hipGraphLaunch(e->call_graph, e->stream));
            e->ev->calls += (uint64_t)n_calls;
            e->step_form = AZD_STEP_PER_CALL_GRAPH;
        } else
            for (int c = 0; c < n_calls; ++c) {
                e->time_begin(0);
                azd::launch_rollout(e->a, t, e->stream);
                e->time_end();
                st = run_evaluator(e); // :175-176
                if (st) return st;
                azd::launch_add_actions(e->a, 0, e->stream);
                azd::launch_argmin(e->a, 0, e->stream); // :190
            }
    }
    st = status_fresh ? check_status(e) : sync_status(e);
    if (improved) *improved = (int)(e->h_status->improved - e->seen_improved);
    e->seen_improved = e->h_status->improved;
    return st;
}

// ---------------------------------------------------------------- run-ahead window
// The launch behind the window is over (or is waited for here): status in, busy shares noted, an aborted launch taken over.
static int window_drain(azd_engine *e) {
    azd_engine::Window &w = e->win;
    if (w.drained) return AZD_OK;
    bool took_over = false;
    uint32_t as = 0;
    size_t ab = 0;
    int st = pool_finish_launch(e, w.fe, w.pool, w.n, &took_over, &as, &ab);
    if (st) return st;
    if (took_over) {
        st = fetch_status(e);
        if (st) return st;
    }
    w.drained = true;
    return check_status(e);
}
// The window ends: every call it ran is accounted for, whether handed out or not.
static int window_close(azd_engine *e) {
    azd_engine::Window &w = e->win;
    if (!w.open) return AZD_OK;
    const int st = window_drain(e);
    w.open = false;
    if (st) return st;
    // every improvement the host was told of, call by call, is one the device's replay of the same log counted
    const unsigned long long total = e->h_status->improved - w.base_improved;
    if (w.consumed == w.n && !w.lumped && w.reported != total) {
        char buf[160];
        snprintf(buf, sizeof(buf), "run-ahead window: %llu improvements handed out call by call, %llu in the device's replay", w.reported, total);
        azd::g_last_error = buf;
        e->seen_improved = e->h_status->improved;
        return AZD_ERR_UNREACHABLE;
    }
    e->seen_improved = e->h_status->improved;
    return AZD_OK;
}
// k calls of the window, in order: each is waited for (the kernel publishes a call once its last agent is through it) and
// compared with the best cost so far exactly as k_argmin_log1 will compare it (strict, optimizer/mod.rs:211).
static int window_serve(azd_engine *e, int k, int *improved) {
    azd_engine::Window &w = e->win;
    int imp = 0;
    for (int i = w.consumed; i < w.consumed + k; ++i) {
        uint32_t spins = 0;
        while (!w.drained && __atomic_load_n(&e->h_win_flag[i], __ATOMIC_ACQUIRE) == 0u) {
            if ((++spins & 255u) == 0u) {
                const hipError_t q = hipStreamQuery(e->stream);
                if (q == hipSuccess) { // the launch is over: whatever it has not published it will not (an abort)
                    const int st = window_drain(e);
                    if (st) {
                        w.open = false;
                        e->seen_improved = e->h_status->improved;
                        return st;
                    }
                } else if (q != hipErrorNotReady) return azd::hip_fail(q, "hipStreamQuery");
            }
#if defined(__x86_64__)
            __builtin_ia32_pause();
#endif
        }
        if (__atomic_load_n(&e->h_win_flag[i], __ATOMIC_ACQUIRE) != 0u) {
            const unsigned long long key = e->h_win_log[i];
            if (key != ~0ull && (uint32_t)(key >> 32) < w.best_ord) {
                w.best_ord = (uint32_t)(key >> 32);
                w.best_key = key;
                imp += 1;
            }
        } else if (!w.lumped) {
            // an aborted launch: the take-over completed the calls from here on and does not publish them one by one -- what they
            // improved is reported with this call, and the argmin record is the one the launch ended with
            const unsigned long long total = e->h_status->improved - w.base_improved;
            imp += (int)(total - w.reported - (unsigned long long)imp);
            w.lumped = true;
        }
    }
    w.consumed += k;
    w.reported += (unsigned long long)imp;
    if (improved) *improved = imp;
    if (w.consumed == w.n) return window_close(e);
    return AZD_OK;
}
// argmin_data inside a window: the record as of the calls handed out.  The winner's replay reads its tree, so the launch is
// waited for; the calls not yet handed out stay in the window.
static int window_argmin_side(azd_engine *e, bool *side) {
    azd_engine::Window &w = e->win;
    *side = false;
    if (!w.open) return AZD_OK;
    const int st = window_drain(e);
    if (st) {
        w.open = false;
        e->seen_improved = e->h_status->improved;
        return st;
    }
    if (w.lumped) return AZD_OK; // (the device's record is all there is)
    if (w.best_key != w.side_key) {
        azd::Arenas a2 = e->a;
        a2.argmin = e->d_argmin_side;
        a2.argmin_r = e->d_argmin_r_side;
        azd::launch_argmin_one(a2, (int)((w.best_key >> 16) & 0xFFFFull), (uint32_t)(w.best_key & 0xFFFFull), e->stream);
        AZD_HIP(hipGetLastError());
        w.side_key = w.best_key;
    }
    *side = true;
    return AZD_OK;
}

int azd_engine_par_roll_out_episodes(azd_engine *e, const uint32_t *tol, int n_tol, uint32_t dflt, int n_calls,
                                     int *improved) {
    if (!e || n_calls < 0 || !e->initialised) return AZD_ERR_INVALID_ARGUMENT;
    if (!e->ev) return AZD_ERR_NO_EVALUATOR;
    AZD_HIP(hipSetDevice(e->cfg.device));
    azd::TolTable t;
    int st = fill_tol(t, tol, n_tol, dflt);
    if (st) return st;
    if (e->win.open) {
        // calls that were run ahead: handed out without a launch.  Other calls (another tolerance table, more calls than the
        // window has left) close the window and run as usual.
        if (memcmp(&e->win.tol, &t, sizeof(t)) == 0 && n_calls <= e->win.n - e->win.consumed) {
            if (n_calls == 0) {
                if (improved) *improved = 0;
                return AZD_OK;
            }
            return window_serve(e, n_calls, improved);
        }
        st = window_close(e);
        if (st) return st;
    }
    return roll_out_impl(e, t, n_calls, improved, false, nullptr);
}

// Run-ahead window: the next n_calls calls of par_roll_out_episodes(tol, ...) are started now, in one launch, and the calls
// that ask for them -- one by one, or in any chunks -- are answered from what the kernel publishes as it goes.
int azd_engine_run_ahead(azd_engine *e, const uint32_t *tol, int n_tol, uint32_t dflt, int n_calls, int *accepted) {
    if (accepted) *accepted = 0;
    if (!e || n_calls < 0 || !e->initialised) return AZD_ERR_INVALID_ARGUMENT;
    if (!e->ev) return AZD_ERR_NO_EVALUATOR;
    AZD_ENTER(e);
    azd::TolTable t;
    int st = fill_tol(t, tol, n_tol, dflt);
    if (st) return st;
    if (n_calls == 0) return AZD_OK;
    int ok = 0;
    st = roll_out_impl(e, t, n_calls, nullptr, true, &ok);
    if (accepted) *accepted = ok;
    return st;
}

// optimizer/mod.rs:262-278
int azd_engine_observe_dev(azd_engine *e, uint32_t n_obs_tol, const float **d_state_vecs, const float **d_obs,
                           const float **d_w) {
    if (!e || !e->initialised) return AZD_ERR_INVALID_ARGUMENT;
    AZD_ENTER(e);
    azd::launch_observe(e->a, n_obs_tol, e->stream);
    AZD_HIP(hipStreamSynchronize(e->stream));
    AZD_HIP(hipGetLastError());
    if (d_state_vecs) *d_state_vecs = e->a.state_vecs;
    if (d_obs) *d_obs = e->a.obs;
    if (d_w) *d_w = e->a.weights;
    return AZD_OK;
}
int azd_engine_observe(azd_engine *e, uint32_t n_obs_tol, float *state_vecs, float *observations, float *weights) {
    int st = azd_engine_observe_dev(e, n_obs_tol, nullptr, nullptr, nullptr);
    if (st) return st;
    const azd::Arenas &a = e->a;
    if (state_vecs) AZD_HIP(hipMemcpy(state_vecs, a.state_vecs, (size_t)a.B * a.S * 4, hipMemcpyDeviceToHost));
    if (observations) AZD_HIP(hipMemcpy(observations, a.obs, (size_t)a.B * a.A * 4, hipMemcpyDeviceToHost));
    if (weights) AZD_HIP(hipMemcpy(weights, a.weights, (size_t)a.B * a.A * 4, hipMemcpyDeviceToHost));
    return AZD_OK;
}
int azd_engine_par_update_model(azd_engine *e, uint32_t n_obs_tol, float *loss) {
    if (!e || !e->initialised) return AZD_ERR_INVALID_ARGUMENT;
    if (!e->ev) return AZD_ERR_NO_EVALUATOR;
    AZD_ENTER(e);
    azd::launch_observe(e->a, n_obs_tol, e->stream);
    AZD_HIP(hipGetLastError());
    float l = 0.f;
    int st = e->ev->update_model_dev(e->a.B, e->a.state_vecs, e->a.obs, e->a.weights, &l, e->stream); // :279-280
    if (st) return st;
    AZD_HIP(hipStreamSynchronize(e->stream));
    if (loss) *loss = l;
    return AZD_OK;
}

// ---- the epoch exchange of a sharded population, for hosts that are not Python (azdopt_amd/parallel.py does the
// same through torch.distributed): optimizer/mod.rs:249-281 where the batch is the union of all ranks' agents.
// RCCL is bound at first use (dlopen), so the library carries no link-time dependency on it.
namespace {
struct RcclApi {
    void *lib = nullptr;
    int (*allGather)(const void *, void *, size_t, int, void *, hipStream_t) = nullptr;
    int (*commCount)(void *, int *) = nullptr;
    const char *(*errorString)(int) = nullptr;
    bool tried = false;
};
RcclApi *rccl() {
    static RcclApi api;
    if (!api.tried) {
        api.tried = true;
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            api.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (api.lib) break;
        }
        if (api.lib) {
            api.allGather = (decltype(api.allGather))dlsym(api.lib, "ncclAllGather");
            api.commCount = (decltype(api.commCount))dlsym(api.lib, "ncclCommCount");
            api.errorString = (decltype(api.errorString))dlsym(api.lib, "ncclGetErrorString");
        }
    }
    return &api;
}
constexpr int NCCL_FLOAT32 = 7; // ncclFloat32 (rccl.h)
} // namespace

int azd_engine_par_update_model_sharded(azd_engine *e, uint32_t n_obs_tol, void *nccl_comm, float *loss) {
    if (!e || !e->initialised || !nccl_comm) return AZD_ERR_INVALID_ARGUMENT;
    if (!e->ev) return AZD_ERR_NO_EVALUATOR;
    AZD_ENTER(e);
    RcclApi &api = *rccl();
    if (!api.allGather || !api.commCount) {
        azd::g_last_error = "librccl.so could not be loaded (ncclAllGather / ncclCommCount)";
        return AZD_ERR_UNSUPPORTED;
    }
    int world = 0;
    int rc = api.commCount(nccl_comm, &world);
    if (rc != 0 || world <= 0) {
        azd::g_last_error = std::string("ncclCommCount: ") + (api.errorString ? api.errorString(rc) : "failed");
        return AZD_ERR_INVALID_ARGUMENT;
    }
    const azd::Arenas &a = e->a;
    const size_t B = (size_t)a.B, rows = B * (size_t)world;
    const size_t ns = rows * a.S, na = rows * a.A;
    if (rows > e->pool_rows) {
        if (e->d_pool) (void)hipFree(e->d_pool);
        e->d_pool = nullptr;
        e->pool_rows = 0;
        AZD_HIP(hipMalloc(&e->d_pool, (ns + 2 * na) * sizeof(float)));
        e->pool_rows = rows;
    }
    if (rows > e->pool_rows || rows > (size_t)INT32_MAX) { // (the pooled buffers hold `pool_rows` rows: sized just above)
        azd::g_last_error = "pooled batch does not fit its buffers";
        return AZD_ERR_CAPACITY;
    }
    float *g_sv = e->d_pool, *g_obs = g_sv + ns, *g_w = g_obs + na;
    azd::launch_observe(a, n_obs_tol, e->stream); // :262-278 on this rank's trees
    AZD_HIP(hipGetLastError());
    // rank order = global agent order (shards are contiguous agent ranges); the collectives run on the engine's
    // stream, behind k_observe and ahead of the optimiser step
    const float *src[3] = {a.state_vecs, a.obs, a.weights};
    float *dst[3] = {g_sv, g_obs, g_w};
    const size_t cnt[3] = {B * (size_t)a.S, B * (size_t)a.A, B * (size_t)a.A};
    for (int i = 0; i < 3; ++i) {
        rc = api.allGather(src[i], dst[i], cnt[i], NCCL_FLOAT32, nccl_comm, e->stream);
        if (rc != 0) {
            (void)hipStreamSynchronize(e->stream); // k_observe and the collectives already queued: nothing of this call is left in flight
            azd::g_last_error = std::string("ncclAllGather: ") + (api.errorString ? api.errorString(rc) : "failed");
            return AZD_ERR_HIP;
        }
    }
    float l = 0.f;
    int st = e->ev->update_model_dev((int)rows, g_sv, g_obs, g_w, &l, e->stream); // :279-280, the same step on every rank
    if (st) return st;
    AZD_HIP(hipStreamSynchronize(e->stream));
    if (loss) *loss = l;
    return AZD_OK;
}

// optimizer/mod.rs:317-346
int azd_engine_reset_begin(azd_engine *e, const uint8_t *parents, const uint64_t *permitted) {
    if (!e || !parents || !permitted || !e->initialised) return AZD_ERR_INVALID_ARGUMENT;
    AZD_ENTER(e);
    int st = upload_roots(e, parents, permitted);
    if (st) return st;
    const azd::Arenas &a = e->a;
    AZD_HIP(hipMemsetAsync(&a.status->failed, 0, sizeof(unsigned long long), e->stream));
    azd::launch_init_roots(a, e->d_stage_parents, e->d_stage_perm, e->stream);
    AZD_HIP(hipMemsetAsync(a.h_theta, 0, (size_t)a.B * a.A * 4, e->stream)); // h_theta_host.fill(0.) at :347
    AZD_HIP(hipGetLastError());
    return AZD_OK;
}
static int reset_finish(azd_engine *e) {
    azd::launch_add_actions(e->a, 1, e->stream); // :350-358; num_inspected_nodes = 0 via cand_* in init_roots
    return sync_status(e);
}
int azd_engine_reset_end(azd_engine *e, const float *h_theta) {
    if (!e || !h_theta) return AZD_ERR_INVALID_ARGUMENT;
    AZD_ENTER(e);
    AZD_HIP(hipMemcpyAsync(e->a.h_theta, h_theta, (size_t)e->a.B * e->a.A * 4, hipMemcpyHostToDevice, e->stream));
    return reset_finish(e);
}
int azd_engine_par_reset_trees(azd_engine *e, const uint8_t *parents, const uint64_t *permitted) {
    if (!e) return AZD_ERR_INVALID_ARGUMENT;
    if (!e->ev) return AZD_ERR_NO_EVALUATOR;
    int st = azd_engine_reset_begin(e, parents, permitted);
    if (st) return st;
    st = run_evaluator(e); // :348
    if (st) return st;
    return reset_finish(e);
}

// par_reset_trees with the c21 driver's modify_root policy (04-c21-tree.rs:172-206) evaluated on the
// device: no host round trip at the epoch boundary.
static int c21_policy_args_ok(azd_engine *e, int kmin, int kmax) {
    if (!e || !e->initialised) return AZD_ERR_INVALID_ARGUMENT;
    if (e->a.space == azd::SPACE_DENSE) {
        if (e->a.path_kind == azd::PATH_SEQUENCE) {
            azd::g_last_error = "the device root policy of the dense-graph space handles ActionSet keys only";
            return AZD_ERR_UNSUPPORTED;
        }
        if (kmin < 1 || kmax < kmin || kmax > e->a.E || kmax > e->dense_slots) {
            azd::g_last_error = "dense root policy: need 1 <= kmin <= kmax <= min(E, 64 * key words) (azd_engine_config::max_slots)";
            return AZD_ERR_INVALID_ARGUMENT;
        }
        return AZD_OK;
    }
    if (e->a.path_kind == azd::PATH_SEQUENCE && e->a.node_cap > 4096) {
        azd::g_last_error = "the device root policy handles sequence-keyed trees of at most 4096 nodes";
        return AZD_ERR_UNSUPPORTED;
    }
    if (e->a.space == azd::SPACE_RAMSEY) {
        if (kmin < 1 || kmax < kmin || kmax > e->a.E || kmax * (e->a.C - 1) > azd::MAX_NODE_ACTIONS) return AZD_ERR_INVALID_ARGUMENT;
        return AZD_OK;
    }
    if (kmin < 1 || kmax < kmin || kmax > e->a.A || kmax > azd::MAX_NODE_ACTIONS) return AZD_ERR_INVALID_ARGUMENT;
    return AZD_OK;
}
int azd_engine_par_reset_trees_c21(azd_engine *e, uint64_t seed, uint64_t epoch, int kmin, int kmax) {
    int st = c21_policy_args_ok(e, kmin, kmax);
    if (st) return st;
    if (!e->ev) return AZD_ERR_NO_EVALUATOR;
    AZD_ENTER(e);
    const azd::Arenas &a = e->a;
    if (a.space == azd::SPACE_DENSE)
        azd::dense_launch_modify_roots(a, seed, epoch, e->cfg.first_agent, kmin, kmax, e->d_stage_parents, e->d_stage_perm, e->d_stage_slots, e->stream);
    else azd::launch_c21_modify_roots(a, seed, epoch, e->cfg.first_agent, kmin, kmax, e->d_stage_parents, e->d_stage_perm, e->stream);
    AZD_HIP(hipMemsetAsync(&a.status->failed, 0, sizeof(unsigned long long), e->stream));
    azd::launch_init_roots(a, e->d_stage_parents, e->d_stage_perm, e->stream);
    AZD_HIP(hipMemsetAsync(a.h_theta, 0, (size_t)a.B * a.A * 4, e->stream));
    AZD_HIP(hipGetLastError());
    st = run_evaluator(e);
    if (st) return st;
    return reset_finish(e);
}
int azd_c21_modify_roots_dev(azd_engine *e, uint64_t seed, uint64_t epoch, int kmin, int kmax, uint8_t *parents_out,
                             uint64_t *permitted_out) {
    int st = c21_policy_args_ok(e, kmin, kmax);
    if (st) return st;
    if (!parents_out || !permitted_out) return AZD_ERR_INVALID_ARGUMENT;
    AZD_ENTER(e);
    const azd::Arenas &a = e->a;
    if (a.space == azd::SPACE_DENSE) { // roots_out: neighbourhoods (8 n bytes per root); permitted_out: slot masks in kw_host words per root
        azd::dense_launch_modify_roots(a, seed, epoch, e->cfg.first_agent, kmin, kmax, e->d_stage_parents, e->d_stage_perm, e->d_stage_slots, e->stream);
        const int ow = (a.E + 63) / 64;
        std::vector<uint64_t> sl((size_t)a.B * ow);
        AZD_HIP(hipMemcpyAsync(parents_out, e->d_stage_parents, (size_t)a.B * a.n * 8, hipMemcpyDeviceToHost, e->stream));
        AZD_HIP(hipMemcpyAsync(sl.data(), e->d_stage_slots, sl.size() * 8, hipMemcpyDeviceToHost, e->stream));
        AZD_HIP(hipStreamSynchronize(e->stream));
        AZD_HIP(hipGetLastError());
        memset(permitted_out, 0, (size_t)a.B * e->kw_host * 8);
        for (int i = 0; i < a.B; ++i) memcpy(permitted_out + (size_t)i * e->kw_host, &sl[(size_t)i * ow], (size_t)ow * 8);
        return AZD_OK;
    }
    azd::launch_c21_modify_roots(a, seed, epoch, e->cfg.first_agent, kmin, kmax, e->d_stage_parents, e->d_stage_perm, e->stream);
    AZD_HIP(hipMemcpyAsync(parents_out, e->d_stage_parents, (size_t)a.B * (a.space == azd::SPACE_RAMSEY ? a.E : a.n), hipMemcpyDeviceToHost, e->stream));
    AZD_HIP(hipMemcpyAsync(permitted_out, e->d_stage_perm, (size_t)a.B * a.KW * 8, hipMemcpyDeviceToHost, e->stream));
    AZD_HIP(hipStreamSynchronize(e->stream));
    AZD_HIP(hipGetLastError());
    return AZD_OK;
}

// space-neutral names of the two entry points above (the policy is the same for both spaces)
int azd_engine_par_reset_trees_policy(azd_engine *e, uint64_t seed, uint64_t epoch, int kmin, int kmax) {
    return azd_engine_par_reset_trees_c21(e, seed, epoch, kmin, kmax);
}
int azd_engine_modify_roots_dev(azd_engine *e, uint64_t seed, uint64_t epoch, int kmin, int kmax, uint8_t *roots_out,
                                uint64_t *permitted_out) {
    return azd_c21_modify_roots_dev(e, seed, epoch, kmin, kmax, roots_out, permitted_out);
}

int azd_engine_ramsey_argmin_data(azd_engine *e, azd_ramsey_argmin *out) {
    if (!e || !out || !e->initialised) return AZD_ERR_INVALID_ARGUMENT;
    if (e->a.space != azd::SPACE_RAMSEY) return AZD_ERR_UNSUPPORTED;
    AZD_HIP(hipSetDevice(e->cfg.device));
    bool side = false; // inside a run-ahead window: the record as of the calls handed out so far
    {
        const int st_w = window_argmin_side(e, &side);
        if (st_w) return st_w;
    }
    static_assert(sizeof(azd_ramsey_argmin) == sizeof(azd::RamseyArgminRec), "ABI struct mismatch");
    AZD_HIP(hipStreamSynchronize(e->stream));
    AZD_HIP(hipMemcpy(out, side ? e->d_argmin_r_side : e->a.argmin_r, sizeof(azd_ramsey_argmin), hipMemcpyDeviceToHost));
    return AZD_OK;
}
int azd_engine_dense_argmin_data(azd_engine *e, azd_dense_argmin *out) {
    if (!e || !out || !e->initialised) return AZD_ERR_INVALID_ARGUMENT;
    if (e->a.space != azd::SPACE_DENSE) return AZD_ERR_UNSUPPORTED;
    AZD_ENTER(e);
    static_assert(sizeof(azd_dense_argmin) == sizeof(azd::DenseArgminRec), "ABI struct mismatch");
    AZD_HIP(hipStreamSynchronize(e->stream));
    AZD_HIP(hipMemcpy(out, e->a.argmin_d, sizeof(azd_dense_argmin), hipMemcpyDeviceToHost));
    return AZD_OK;
}
int azd_engine_ramsey_agent_counts(azd_engine *e, int agent, int32_t *counts, int32_t *totals) {
    if (!e || agent < 0 || agent >= e->a.B) return AZD_ERR_INVALID_ARGUMENT;
    if (e->a.space != azd::SPACE_RAMSEY) return AZD_ERR_UNSUPPORTED;
    AZD_ENTER(e);
    AZD_HIP(hipStreamSynchronize(e->stream));
    const azd::Arenas &a = e->a;
    if (counts) AZD_HIP(hipMemcpy(counts, a.cur_counts + (size_t)agent * a.C * a.E, (size_t)a.C * a.E * 4, hipMemcpyDeviceToHost));
    if (totals) AZD_HIP(hipMemcpy(totals, a.cur_tot + (size_t)agent * 4, 16, hipMemcpyDeviceToHost));
    return AZD_OK;
}

int azd_engine_argmin_data(azd_engine *e, azd_argmin *out) {
    if (!e || !out || !e->initialised) return AZD_ERR_INVALID_ARGUMENT;
    if (e->a.space != azd::SPACE_C21) return AZD_ERR_UNSUPPORTED;
    AZD_HIP(hipSetDevice(e->cfg.device));
    bool side = false; // inside a run-ahead window: the record as of the calls handed out so far
    {
        const int st_w = window_argmin_side(e, &side);
        if (st_w) return st_w;
    }
    static_assert(sizeof(azd_argmin) == sizeof(azd::ArgminRec), "ABI struct mismatch");
    AZD_HIP(hipMemcpyAsync(e->h_argmin, side ? e->d_argmin_side : e->a.argmin, sizeof(azd::ArgminRec), hipMemcpyDeviceToHost, e->stream));
    AZD_HIP(hipStreamSynchronize(e->stream));
    memcpy(out, e->h_argmin, sizeof(azd_argmin));
    return AZD_OK;
}

int azd_engine_read_state_vecs(azd_engine *e, float *out) {
    if (!e || !out) return AZD_ERR_INVALID_ARGUMENT;
    AZD_ENTER(e);
    AZD_HIP(hipStreamSynchronize(e->stream));
    AZD_HIP(hipMemcpy(out, e->a.state_vecs, (size_t)e->a.B * e->a.S * 4, hipMemcpyDeviceToHost));
    return AZD_OK;
}
// Test entry: rows of the evaluator as the CU-resident step forms compute them (mlp_tile_task; k_tile_forward), for `rows` state
// vectors given by the host.  Needs an MLP evaluator the pool step can serve (the plan lays the rows out in LDS).
int azd_engine_debug_tile_forward(azd_engine *e, const float *states, float *predictions, int rows) {
    if (!e || !states || !predictions || rows < 1) return AZD_ERR_INVALID_ARGUMENT;
    AZD_ENTER(e);
    azd::FusedEval fe;
    azd::PoolArgs pool = e->pool;
    uint32_t dyn_stride = 0;
    size_t dyn_bytes = 0;
    const char *why = "";
    if (e->a.space == azd::SPACE_DENSE || !e->ev->fused_desc(&fe) || fe.kind != 3 || !azd::pool_plan(e->a, fe, &pool, &dyn_stride, &dyn_bytes, &why)) {
        azd::g_last_error = "debug_tile_forward: needs the c21 or the Ramsey space with an MLP evaluator the pool step can serve";
        return AZD_ERR_INVALID_ARGUMENT;
    }
    float *d_in = nullptr, *d_out = nullptr;
    hipError_t he = hipMalloc(&d_in, (size_t)rows * e->a.S * 4);
    const char *what = "hipMalloc";
    if (he == hipSuccess) he = hipMalloc(&d_out, (size_t)rows * e->a.A * 4);
    if (he == hipSuccess) what = "hipMemcpyAsync", he = hipMemcpyAsync(d_in, states, (size_t)rows * e->a.S * 4, hipMemcpyHostToDevice, e->stream);
    if (he == hipSuccess) what = "k_tile_forward", he = azd::launch_tile_forward(fe, pool, rows, d_in, d_out, e->stream);
    if (he == hipSuccess) what = "hipMemcpyAsync", he = hipMemcpyAsync(predictions, d_out, (size_t)rows * e->a.A * 4, hipMemcpyDeviceToHost, e->stream);
    if (he == hipSuccess) what = "hipStreamSynchronize", he = hipStreamSynchronize(e->stream);
    else (void)hipStreamSynchronize(e->stream); // nothing of this call may still be running on the buffers freed below
    (void)hipFree(d_in);
    (void)hipFree(d_out);
    return he == hipSuccess ? AZD_OK : azd::hip_fail(he, what);
}
int azd_engine_read_predictions(azd_engine *e, float *out) {
    if (!e || !out) return AZD_ERR_INVALID_ARGUMENT;
    AZD_ENTER(e);
    AZD_HIP(hipStreamSynchronize(e->stream));
    AZD_HIP(hipMemcpy(out, e->a.h_theta, (size_t)e->a.B * e->a.A * 4, hipMemcpyDeviceToHost));
    return AZD_OK;
}

int azd_engine_tree_sizes(azd_engine *e, int agent, int *n_nodes, int *n_arcs, int *n_preds) {
    if (!e || agent < 0 || agent >= e->a.B) return AZD_ERR_INVALID_ARGUMENT;
    AZD_ENTER(e);
    AZD_HIP(hipStreamSynchronize(e->stream));
    uint32_t v[3];
    AZD_HIP(hipMemcpy(&v[0], e->a.n_nodes + agent, 4, hipMemcpyDeviceToHost));
    AZD_HIP(hipMemcpy(&v[1], e->a.n_arcs + agent, 4, hipMemcpyDeviceToHost));
    AZD_HIP(hipMemcpy(&v[2], e->a.n_preds + agent, 4, hipMemcpyDeviceToHost));
    if (n_nodes) *n_nodes = (int)v[0];
    if (n_arcs) *n_arcs = (int)v[1];
    if (n_preds) *n_preds = (int)v[2];
    return AZD_OK;
}

int azd_engine_export_tree(azd_engine *e, int agent, float *c, float *c_star, uint32_t *n_t, uint32_t *exhausted,
                           uint32_t *act_begin, uint32_t *act_end, uint64_t *keys, uint32_t *arc_src,
                           uint32_t *arc_dst, uint32_t *arc_pp, uint32_t *pred_a_id, float *pred_g,
                           int32_t *pred_arc) {
    int nn, na, np;
    int st = azd_engine_tree_sizes(e, agent, &nn, &na, &np);
    if (st) return st;
    const azd::Arenas &a = e->a;
    std::vector<azd::NodeRec> nodes((size_t)nn);
    std::vector<azd::ArcRec> arcs((size_t)na);
    std::vector<azd::PredRec> preds((size_t)np);
    AZD_HIP(hipMemcpy(nodes.data(), a.nodes + (size_t)agent * a.node_cap, nodes.size() * sizeof(azd::NodeRec), hipMemcpyDeviceToHost));
    if (na) AZD_HIP(hipMemcpy(arcs.data(), a.arcs + (size_t)agent * a.arc_cap, arcs.size() * sizeof(azd::ArcRec), hipMemcpyDeviceToHost));
    if (np) AZD_HIP(hipMemcpy(preds.data(), a.preds + (size_t)agent * a.pred_cap, preds.size() * sizeof(azd::PredRec), hipMemcpyDeviceToHost));
    std::vector<float> g_exp((size_t)np); // g of the expanded predictions (the record holds the child's summary in its place)
    if (np) AZD_HIP(hipMemcpy(g_exp.data(), a.pred_g + (size_t)agent * a.pred_cap, g_exp.size() * 4, hipMemcpyDeviceToHost));
    if (keys && a.space == azd::SPACE_DENSE) { // device keys are sets of RANKS: back to action-id sets (kw_host words per node)
        const int KW = a.KW, MAXS = e->dense_slots;
        std::vector<uint64_t> rk((size_t)nn * KW);
        std::vector<uint16_t> tab((size_t)MAXS);
        AZD_HIP(hipMemcpy(rk.data(), a.keys + (size_t)agent * a.node_cap * KW, rk.size() * 8, hipMemcpyDeviceToHost));
        AZD_HIP(hipMemcpy(tab.data(), a.root_aid + (size_t)agent * MAXS, tab.size() * 2, hipMemcpyDeviceToHost));
        memset(keys, 0, (size_t)nn * e->kw_host * 8);
        for (int i = 0; i < nn; ++i)
            for (int r = 0; r < MAXS; ++r)
                if ((rk[(size_t)i * KW + (r >> 6)] >> (r & 63)) & 1ull) keys[(size_t)i * e->kw_host + (tab[(size_t)r] >> 6)] |= 1ull << (tab[(size_t)r] & 63);
    } else if (keys) AZD_HIP(hipMemcpy(keys, a.keys + (size_t)agent * a.node_cap * a.KW, (size_t)nn * a.KW * 8, hipMemcpyDeviceToHost));
    for (int i = 0; i < nn; ++i) {
        if (c) c[i] = nodes[(size_t)i].c;
        if (c_star) c_star[i] = nodes[(size_t)i].c_star;
        if (n_t) n_t[i] = azd::node_nt(nodes[(size_t)i]);
        if (exhausted) exhausted[i] = azd::node_exhausted(nodes[(size_t)i]);
        if (act_begin) act_begin[i] = nodes[(size_t)i].act_begin;
        if (act_end) act_end[i] = nodes[(size_t)i].act_end;
    }
    for (int i = 0; i < na; ++i) {
        if (arc_src) arc_src[i] = arcs[(size_t)i].src;
        if (arc_dst) arc_dst[i] = arcs[(size_t)i].dst;
        if (arc_pp) arc_pp[i] = arcs[(size_t)i].pp;
    }
    for (int i = 0; i < np; ++i) {
        const azd::PredRec &pr = preds[(size_t)i];
        const bool ex = azd::pred_expanded(pr);
        if (pred_a_id) pred_a_id[i] = azd::pred_aid(pr);
        if (pred_g) {
            const uint32_t gb = pr.w1;
            float gv;
            memcpy(&gv, &gb, 4);
            pred_g[i] = ex ? g_exp[(size_t)i] : gv;
        }
        if (pred_arc) pred_arc[i] = ex ? (int32_t)azd::pred_arc(pr) : -1;
    }
    return AZD_OK;
}

int azd_engine_agent_state(azd_engine *e, int agent, uint8_t *parents, uint64_t *permitted, uint64_t *path,
                           uint32_t *state_pos, double *lambda_1, int *matching_size) {
    if (!e || agent < 0 || agent >= e->a.B) return AZD_ERR_INVALID_ARGUMENT;
    AZD_ENTER(e);
    AZD_HIP(hipStreamSynchronize(e->stream));
    const azd::Arenas &a = e->a;
    if (a.space == azd::SPACE_DENSE) { // `parents` receives the neighbourhoods (8 n bytes); masks are kw_host words
        const int KW = a.KW, MAXS = e->dense_slots;
        std::vector<uint16_t> tab((size_t)MAXS);
        uint64_t rem[16], pth[16];
        AZD_HIP(hipMemcpy(tab.data(), a.root_aid + (size_t)agent * MAXS, tab.size() * 2, hipMemcpyDeviceToHost));
        AZD_HIP(hipMemcpy(rem, a.cur_perm + (size_t)agent * KW, (size_t)KW * 8, hipMemcpyDeviceToHost));
        AZD_HIP(hipMemcpy(pth, a.cur_path + (size_t)agent * KW, (size_t)KW * 8, hipMemcpyDeviceToHost));
        if (parents) AZD_HIP(hipMemcpy(parents, a.cur_adj + (size_t)agent * 64, (size_t)a.n * 8, hipMemcpyDeviceToHost));
        if (permitted) memset(permitted, 0, (size_t)e->kw_host * 8);
        if (path) memset(path, 0, (size_t)e->kw_host * 8);
        for (int r = 0; r < MAXS; ++r) {
            if (permitted && ((rem[r >> 6] >> (r & 63)) & 1ull)) { // open SLOTS, as in the packed root
                const int slot = tab[(size_t)r] % a.E;
                permitted[slot >> 6] |= 1ull << (slot & 63);
            }
            if (path && ((pth[r >> 6] >> (r & 63)) & 1ull)) path[tab[(size_t)r] >> 6] |= 1ull << (tab[(size_t)r] & 63);
        }
        if (state_pos) AZD_HIP(hipMemcpy(state_pos, a.state_pos + agent, 4, hipMemcpyDeviceToHost));
        if (lambda_1) AZD_HIP(hipMemcpy(lambda_1, a.cur_lambda + agent, 8, hipMemcpyDeviceToHost));
        if (matching_size) AZD_HIP(hipMemcpy(matching_size, a.cur_mu + agent, 4, hipMemcpyDeviceToHost));
        return AZD_OK;
    }
    if (a.space == azd::SPACE_RAMSEY) { // `parents` receives the colour of every edge (E bytes)
        uint32_t nbr[128];
        AZD_HIP(hipMemcpy(nbr, a.cur_nbr + (size_t)agent * 128, sizeof(nbr), hipMemcpyDeviceToHost));
        if (parents) {
            int pos = 0;
            for (int v = 0; v < a.n; ++v)
                for (int u = 0; u < v; ++u, ++pos) {
                    parents[pos] = 0;
                    for (int c = 0; c < a.C; ++c)
                        if ((nbr[c * 32 + v] >> u) & 1u) parents[pos] = (uint8_t)c;
                }
        }
        if (permitted) AZD_HIP(hipMemcpy(permitted, a.cur_perm + (size_t)agent * a.KW, (size_t)a.KW * 8, hipMemcpyDeviceToHost));
        if (path) AZD_HIP(hipMemcpy(path, a.cur_path + (size_t)agent * a.KW, (size_t)a.KW * 8, hipMemcpyDeviceToHost));
        if (state_pos) AZD_HIP(hipMemcpy(state_pos, a.state_pos + agent, 4, hipMemcpyDeviceToHost));
        if (lambda_1) *lambda_1 = 0.0;
        if (matching_size) *matching_size = 0;
        return AZD_OK;
    }
    uint8_t par[azd::PARENTS_STRIDE];
    AZD_HIP(hipMemcpy(par, a.cur_parents + (size_t)agent * azd::PARENTS_STRIDE, azd::PARENTS_STRIDE, hipMemcpyDeviceToHost));
    if (parents) memcpy(parents, par, (size_t)a.n);
    if (permitted) AZD_HIP(hipMemcpy(permitted, a.cur_perm + (size_t)agent * a.KW, (size_t)a.KW * 8, hipMemcpyDeviceToHost));
    if (path) AZD_HIP(hipMemcpy(path, a.cur_path + (size_t)agent * a.KW, (size_t)a.KW * 8, hipMemcpyDeviceToHost));
    if (state_pos) AZD_HIP(hipMemcpy(state_pos, a.state_pos + agent, 4, hipMemcpyDeviceToHost));
    if (lambda_1) AZD_HIP(hipMemcpy(lambda_1, a.cur_lambda + agent, 8, hipMemcpyDeviceToHost));
    if (matching_size) AZD_HIP(hipMemcpy(matching_size, a.cur_mu + agent, 4, hipMemcpyDeviceToHost));
    return AZD_OK;
}

int azd_engine_counters(azd_engine *e, uint64_t *out) {
    if (!e || !out) return AZD_ERR_INVALID_ARGUMENT;
    AZD_ENTER(e);
    AZD_HIP(hipStreamSynchronize(e->stream));
    const azd::Arenas &a = e->a;
    std::vector<unsigned long long> h((size_t)a.B * azd::NUM_COUNTERS);
    std::vector<uint32_t> fl((size_t)a.B);
    AZD_HIP(hipMemcpy(h.data(), a.counters, h.size() * 8, hipMemcpyDeviceToHost));
    AZD_HIP(hipMemcpy(fl.data(), a.flags, fl.size() * 4, hipMemcpyDeviceToHost));
    for (int k = 0; k < AZD_CTR_COUNT; ++k) out[k] = 0;
    for (int i = 0; i < a.B; ++i) {
        for (int k = 0; k < azd::NUM_COUNTERS; ++k) {
            unsigned long long v = h[(size_t)i * azd::NUM_COUNTERS + k];
            if (k >= AZD_CTR_COUNT) continue;
            if (k == AZD_CTR_MAX_FRONTIER || k == AZD_CTR_MAX_DEPTH || k == AZD_CTR_TICKS_MAX_CALL) out[k] = std::max<uint64_t>(out[k], v);
            else out[k] += v;
        }
    }
    uint64_t failed = 0;
    for (uint32_t f : fl) failed += f != 0;
    out[AZD_CTR_FAILED_AGENTS] = failed;
    return AZD_OK;
}

int azd_engine_agent_counters(azd_engine *e, uint64_t *out) {
    if (!e || !out) return AZD_ERR_INVALID_ARGUMENT;
    AZD_ENTER(e);
    AZD_HIP(hipStreamSynchronize(e->stream));
    AZD_HIP(hipMemcpy(out, e->a.counters, (size_t)e->a.B * azd::NUM_COUNTERS * 8, hipMemcpyDeviceToHost));
    return AZD_OK;
}
int azd_engine_agent_counters_per_agent(azd_engine *e) { return e ? (e->counters_by_wave ? 0 : 1) : AZD_ERR_INVALID_ARGUMENT; }

int azd_engine_set_timing(azd_engine *e, int enabled) {
    if (!e) return AZD_ERR_INVALID_ARGUMENT;
    e->timing = enabled != 0;
    e->rollout_ms = e->evaluator_ms = 0;
    e->rollout_launches = 0;
    return AZD_OK;
}
int azd_engine_timing(azd_engine *e, double *tree_ms, double *evaluator_ms, uint64_t *tree_launches) {
    if (!e) return AZD_ERR_INVALID_ARGUMENT;
    if (tree_ms) *tree_ms = e->rollout_ms;
    if (evaluator_ms) *evaluator_ms = e->evaluator_ms;
    if (tree_launches) *tree_launches = e->rollout_launches;
    return AZD_OK;
}
void *azd_engine_stream(azd_engine *e) { return e ? (void *)e->stream : nullptr; }
int azd_engine_pool_utilisation(azd_engine *e, double *eval_busy, double *search_busy) {
    if (!e) return AZD_ERR_INVALID_ARGUMENT;
    if (eval_busy) *eval_busy = e->pool_util_eval;
    if (search_busy) *search_busy = e->pool_util_search;
    return AZD_OK;
}
int azd_engine_pool_groups(azd_engine *e, int *members, int *groups, int *waves_per_slot) {
    if (!e) return AZD_ERR_INVALID_ARGUMENT;
    if (members) *members = e->pool_grp_g;
    if (groups) *groups = e->pool_grp_groups;
    if (waves_per_slot) *waves_per_slot = e->pool_grp_w;
    return AZD_OK;
}
int azd_engine_pool_agent_finish(azd_engine *e, uint64_t *ticks_out) {
    if (!e || !ticks_out) return AZD_ERR_INVALID_ARGUMENT;
    AZD_ENTER(e);
    if (!e->pool.stamp) {
        azd::g_last_error = "pool_agent_finish: this engine has run no pool step";
        return AZD_ERR_INVALID_ARGUMENT;
    }
    AZD_HIP(hipStreamSynchronize(e->stream));
    AZD_HIP(hipMemcpy(ticks_out, e->pool.stamp, (size_t)e->a.B * 8, hipMemcpyDeviceToHost));
    return AZD_OK;
}
int azd_engine_pool_split(azd_engine *e, int *eval_wgs, int *search_wgs) {
    if (!e) return AZD_ERR_INVALID_ARGUMENT;
    if (eval_wgs) *eval_wgs = e->pool_eval_wgs;
    if (search_wgs) *search_wgs = e->pool_search_wgs;
    return AZD_OK;
}
int azd_debug_probe_xcc(int device, uint32_t *out, int n_blocks) {
    if (!out || n_blocks <= 0) return AZD_ERR_INVALID_ARGUMENT;
    int st = azd::device_ok(device);
    if (st) return st;
    AZD_HIP(hipSetDevice(device));
    uint32_t *d = nullptr;
    AZD_HIP(hipMalloc(&d, (size_t)n_blocks * 4));
    azd::launch_probe_xcc(d, n_blocks, nullptr);
    hipError_t he = hipDeviceSynchronize();
    if (he == hipSuccess) he = hipMemcpy(out, d, (size_t)n_blocks * 4, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (he != hipSuccess) return azd::hip_fail(he, "probe_xcc");
    return AZD_OK;
}
int azd_engine_step_form(azd_engine *e, int *form, const char **reason) {
    if (!e) return AZD_ERR_INVALID_ARGUMENT;
    if (form) *form = e->step_form;
    if (reason) *reason = e->step_reason.c_str();
    return AZD_OK;
}

int azd_debug_probe_math(int device, const float *in, float *out, int n) {
    if (!in || !out || n <= 0) return AZD_ERR_INVALID_ARGUMENT;
    int st = azd::device_ok(device);
    if (st) return st;
    AZD_HIP(hipSetDevice(device));
    float *d_in = nullptr, *d_out = nullptr;
    AZD_HIP(hipMalloc(&d_in, (size_t)n * 8));
    AZD_HIP(hipMalloc(&d_out, (size_t)n * 16));
    AZD_HIP(hipMemcpy(d_in, in, (size_t)n * 8, hipMemcpyHostToDevice));
    azd::launch_probe_math(d_in, d_out, n, nullptr);
    hipError_t he = hipDeviceSynchronize();
    if (he == hipSuccess) he = hipMemcpy(out, d_out, (size_t)n * 16, hipMemcpyDeviceToHost);
    (void)hipFree(d_in);
    (void)hipFree(d_out);
    if (he != hipSuccess) return azd::hip_fail(he, "probe_math");
    return AZD_OK;
}

int azd_debug_probe_cost(int device, const uint8_t *parents, int n, int count, int reps, int full, double *lambda_1,
                         int *matching_size, float *ms) {
    if (!parents || !lambda_1 || !matching_size || n < 4 || n > AZD_C21_MAX_N || count <= 0 || reps <= 0)
        return AZD_ERR_INVALID_ARGUMENT;
    int st = azd::device_ok(device);
    if (st) return st;
    AZD_HIP(hipSetDevice(device));
    for (int i = 0; i < count; ++i) {
        const uint8_t *p = parents + (size_t)i * n;
        if (p[0] != 0 || p[n - 1] != 0) return AZD_ERR_INVALID_ARGUMENT;
        for (int v = 1; v < n; ++v)
            if (p[v] >= v) return AZD_ERR_INVALID_ARGUMENT;
    }
    uint8_t *d_p = nullptr;
    double *d_l = nullptr;
    int *d_m = nullptr;
    hipEvent_t e0, e1;
    AZD_HIP(hipMalloc(&d_p, (size_t)count * n));
    AZD_HIP(hipMalloc(&d_l, (size_t)count * 8));
    AZD_HIP(hipMalloc(&d_m, (size_t)count * 4));
    AZD_HIP(hipEventCreate(&e0));
    AZD_HIP(hipEventCreate(&e1));
    AZD_HIP(hipMemcpy(d_p, parents, (size_t)count * n, hipMemcpyHostToDevice));
    azd::launch_probe_cost(d_p, n, count, 1, full, d_l, d_m, nullptr); // warm-up
    AZD_HIP(hipEventRecord(e0, nullptr));
    azd::launch_probe_cost(d_p, n, count, reps, full, d_l, d_m, nullptr);
    AZD_HIP(hipEventRecord(e1, nullptr));
    hipError_t he = hipDeviceSynchronize();
    float t = 0.f;
    if (he == hipSuccess) he = hipEventElapsedTime(&t, e0, e1);
    if (he == hipSuccess) he = hipMemcpy(lambda_1, d_l, (size_t)count * 8, hipMemcpyDeviceToHost);
    if (he == hipSuccess) he = hipMemcpy(matching_size, d_m, (size_t)count * 4, hipMemcpyDeviceToHost);
    (void)hipFree(d_p);
    (void)hipFree(d_l);
    (void)hipFree(d_m);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (he != hipSuccess) return azd::hip_fail(he, "probe_cost");
    if (ms) *ms = t;
    return AZD_OK;
}

// ------------------------------------------------------------------ c21 root policy (host)
// 04-c21-tree.rs:172-206 applied to node_data() (tree/mod.rs:302-307: BTreeMap order, i.e. keys
// compared lexicographically over their ascending elements; n[0] is the root).
static bool key_less(const uint64_t *x, const uint64_t *y, int KW) {
    // d = lowest differing action id; the set holding d is smaller iff the other has an element > d
    for (int w = 0; w < KW; ++w) {
        uint64_t diff = x[w] ^ y[w];
        if (!diff) continue;
        int b = __builtin_ctzll(diff);
        bool x_has = (x[w] >> b) & 1ull;
        const uint64_t *other = x_has ? y : x;
        bool other_has_more = (b < 63 ? (other[w] >> (b + 1)) != 0 : false);
        for (int w2 = w + 1; w2 < KW && !other_has_more; ++w2) other_has_more = other[w2] != 0;
        // x_has: x < y iff y continues; else y has d: x < y iff x does NOT continue (x is a proper prefix)
        return x_has ? other_has_more : !other_has_more;
    }
    return false;
}

int azd_c21_modify_roots(azd_engine *e, uint64_t seed, uint64_t epoch, int kmin, int kmax, uint8_t *parents_out,
                         uint64_t *permitted_out) {
    if (!e || !parents_out || !permitted_out || !e->initialised) return AZD_ERR_INVALID_ARGUMENT;
    if (e->a.space != azd::SPACE_C21 || e->a.path_kind != azd::PATH_SET) return AZD_ERR_UNSUPPORTED;
    AZD_ENTER(e);
    AZD_HIP(hipStreamSynchronize(e->stream));
    const azd::Arenas &a = e->a;
    const int B = a.B, KW = a.KW, n = a.n;
    std::vector<uint32_t> n_nodes((size_t)B);
    AZD_HIP(hipMemcpy(n_nodes.data(), a.n_nodes, (size_t)B * 4, hipMemcpyDeviceToHost));
    std::vector<uint8_t> rp((size_t)B * azd::PARENTS_STRIDE);
    std::vector<uint64_t> rm((size_t)B * KW);
    AZD_HIP(hipMemcpy(rp.data(), a.root_parents, rp.size(), hipMemcpyDeviceToHost));
    AZD_HIP(hipMemcpy(rm.data(), a.root_perm, rm.size() * 8, hipMemcpyDeviceToHost));
    uint32_t max_nodes = 0;
    for (uint32_t v : n_nodes) max_nodes = std::max(max_nodes, v);
    std::vector<azd::NodeRec> nodes((size_t)max_nodes);
    std::vector<uint64_t> keys((size_t)max_nodes * KW);
    const uint64_t domain = azd::DOMAIN_RESET ^ (epoch << 32);
    // (parent, child) of an action id
    auto act = [&](uint8_t *par, uint64_t *perm, int aid) {
        int c = 2;
        while (c * (c + 1) / 2 - 1 <= aid) ++c;
        int p = aid - (c * (c - 1) / 2 - 1);
        par[c] = (uint8_t)p;
        for (int u = 0; u < c; ++u) {
            int id = c * (c - 1) / 2 + u - 1;
            perm[id >> 6] &= ~(1ull << (id & 63));
        }
    };
    std::vector<uint32_t> keep;
    for (int i = 0; i < B; ++i) {
        const uint64_t agent = e->cfg.first_agent + (uint64_t)i;
        const uint32_t nn = n_nodes[(size_t)i];
        AZD_HIP(hipMemcpy(nodes.data(), a.nodes + (size_t)i * a.node_cap, (size_t)nn * sizeof(azd::NodeRec), hipMemcpyDeviceToHost));
        AZD_HIP(hipMemcpy(keys.data(), a.keys + (size_t)i * a.node_cap * KW, (size_t)nn * KW * 8, hipMemcpyDeviceToHost));
        uint8_t par[azd::PARENTS_STRIDE];
        uint64_t perm[azd::MAX_KW] = {0, 0, 0, 0};
        memcpy(par, &rp[(size_t)i * azd::PARENTS_STRIDE], azd::PARENTS_STRIDE);
        for (int w = 0; w < KW; ++w) perm[w] = rm[(size_t)i * KW + w];
        uint8_t *po = parents_out + (size_t)i * n;
        uint64_t *mo = permitted_out + (size_t)i * KW;
        const float c_root = nodes[0].c, c_root_star = nodes[0].c_star;
        const uint64_t r0 = azd::stream_key(seed, domain, agent, 0), r1 = azd::stream_key(seed, domain, agent, 1);
        int k_new;
        keep.clear();
        if (c_root == c_root_star) {
            int kcur = 0;
            for (int w = 0; w < KW; ++w) kcur += __builtin_popcountll(perm[w]);
            if (kcur == kmax) {
                int k = kmin + (int)azd::draw_below(r1, (uint32_t)(kmax - kmin + 1));
                azd::c21_fresh_root(seed, domain, agent, n, k, po, mo);
                continue;
            }
            for (uint32_t v = 0; v < nn; ++v)
                if (nodes[v].c == c_root) keep.push_back(v);
            k_new = kcur + (int)azd::draw_below(r1, (uint32_t)(kmax - kcur + 1));
        } else {
            const float thr = (c_root + 3.0f * c_root_star) / 4.0f;
            for (uint32_t v = 0; v < nn; ++v)
                if (nodes[v].c <= thr) keep.push_back(v);
            k_new = kmin + (int)azd::draw_below(r1, (uint32_t)(kmax - kmin + 1));
        }
        const uint32_t r = azd::draw_below(r0, (uint32_t)keep.size());
        std::nth_element(keep.begin(), keep.begin() + r, keep.end(), [&](uint32_t x, uint32_t y) {
            return key_less(&keys[(size_t)x * KW], &keys[(size_t)y * KW], KW);
        });
        const uint64_t *key = &keys[(size_t)keep[r] * KW];
        for (int w = 0; w < KW; ++w) {
            uint64_t bits = key[w];
            while (bits) {
                int b = __builtin_ctzll(bits);
                bits &= bits - 1;
                act(par, perm, w * 64 + b);
            }
        }
        memcpy(po, par, (size_t)n);
        azd::c21_shuffle_permitted(seed, domain, agent, n, k_new, mo);
    }
    return AZD_OK;
}

} // extern "C"
