// engine_types.h -- device data layout shared by the tree kernels and the host engine.
//
// HBM layout (DESIGN.md "Data layout"): every tree owns a fixed-capacity slice of
// five tree-major arenas.  Records are sized to the 16-B-per-lane vector-load
// sweet spot of gfx950: a node is two dwordx4, an arc and a prediction one each.
//
//   nodes  [B][node_cap]       NodeRec  32 B   StateWeight (tree/state_weight.rs:4-10) + in-list head
//   keys   [B][node_cap][KW]   u64             ActionSet of the node as a bit mask (path/set.rs:6-9)
//   arcs   [B][arc_cap]        ArcRec   16 B   petgraph edge: endpoints, ActionWeight.prediction_pos, next-in
//   preds  [B][pred_cap]       PredRec  16 B   ActionPrediction (tree/arc_weight.rs:11-16) + child node
//   ht     [B][ht_cap]         u32             open-addressing transposition table over `keys`
//                                              (stands in for BTreeMap<P, NodeIndex>, tree/mod.rs:29)
#pragma once
#include <stddef.h>
#include <stdint.h>

namespace azd {

constexpr uint32_t NONE = 0xFFFFFFFFu;
constexpr int PARENTS_STRIDE = 32; // bytes per packed parents row on the device
constexpr int MAX_N = 24;
constexpr int MAX_KW = 6;                   // c21 uses up to 4 key words, the Ramsey space up to 6
constexpr int PATH_SET = 0, PATH_SEQUENCE = 1;   // = AZD_PATH_*
constexpr int SPACE_C21 = 1, SPACE_RAMSEY = 2, SPACE_DENSE = 3; // = AZD_SPACE_* of include/azdopt_amd.h
constexpr int PRED_CHUNKS = 2;               // a node holds at most 64*PRED_CHUNKS legal actions
constexpr int MAX_NODE_ACTIONS = 64 * PRED_CHUNKS;
constexpr int FRONTIER_CAP = 256;            // LDS-staged cascade frontier per tree (a level's entries beyond it go to Arenas::fr_spill)
constexpr int PATH_STACK = 32;               // nodes of the current path kept per agent, root first (deeper levels: not kept)
constexpr int MAX_TOL = 32;
constexpr int NUM_COUNTERS = 32; // 0..15 public counters, 16..24 phase ticks (AZD_PHASE_PROFILE builds), 25..27 evaluator service

// (round 4) The two words a cascade rewrites -- c_star and (n_t, exhausted) -- open the record: one 8-byte store.  n_t counts
// the cascades that passed through the node, at most one per arc ever added to the tree, so it is bounded by arc_cap
// (<= 65535: engine limit); exhausted <= the node's action count (<= 2047).
struct __attribute__((aligned(16))) NodeRec {
    float c_star;       // best eval seen at or below
    uint32_t nt_ex;     // n_t (bits 0-19) | exhausted_children (bits 20-31)
    float c;            // evaluate(cost(state))
    uint32_t act_begin; // actions: Range<u32> into preds
    uint32_t act_end;
    uint32_t first_in;  // head of the list of LATER arcs into this node (transpositions), NONE if empty
    uint32_t in_src;    // source of the arc that created the node (NONE for the root) ...
    uint32_t in_pp;     // ... and the prediction (index into the tree's preds) that arc expands: where the PARENT keeps this node's summary
};
static_assert(sizeof(NodeRec) == 32, "NodeRec must be 32 B");

struct __attribute__((aligned(16))) ArcRec {
    uint32_t src, dst;
    uint32_t pp;      // ActionWeight.prediction_pos
    uint32_t next_in; // next arc into dst (older), NONE at the tail
};
static_assert(sizeof(ArcRec) == 16, "ArcRec must be 16 B");

// (round 4) A prediction carries what next_action needs to know about the CHILD it leads to, so that a selection level is one
// round trip (the node's predictions) instead of two (the predictions, then a gather of the children's records):
//   unexpanded   w0 = c_theta_star = c - g as f32 (max_curiosity's candidate value, next_action.rs:63-66; c is the node's own
//                cost, fixed when the predictions are written), w1 = g, w2 = NONE, w3 = a_id
//   expanded     w0 = the child's c_star, w1 = the child's n_t (bits 0-19) | its action count (20-30) | active (31),
//                w2 = child node (bits 0-15) | arc id (16-31), w3 = a_id (bits 0-11) | the child's act_begin (12-31)
// w0 / w1 of an expanded prediction are rewritten (one 8-byte store) by whoever changes the child: the cascade
// (empty_transitions.rs:50-127) and the child's add_actions.  g of an expanded prediction moves to Arenas::pred_g (the search
// never reads it again; the tree export does).  Limits (engine.hip checks them): node_cap <= 65536, arc_cap <= 65535,
// pred_cap <= 2^20, ACTION_DIM <= 4096, <= 2047 actions per node.
struct __attribute__((aligned(16))) PredRec {
    uint32_t w0, w1, w2, w3;
};
static_assert(sizeof(PredRec) == 16, "PredRec must be 16 B");
#if defined(__HIPCC__)
#define AZD_HD __host__ __device__ inline
#else
#define AZD_HD inline
#endif
AZD_HD uint32_t node_nt(const NodeRec &r) { return r.nt_ex & 0xFFFFFu; }
AZD_HD uint32_t node_exhausted(const NodeRec &r) { return r.nt_ex >> 20; }
AZD_HD uint32_t node_pack(uint32_t n_t, uint32_t exhausted) { return n_t | (exhausted << 20); }
AZD_HD bool pred_expanded(const PredRec &p) { return p.w2 != NONE; }
AZD_HD uint32_t pred_aid(const PredRec &p) { return p.w3 & 0xFFFu; }
AZD_HD uint32_t pred_child(const PredRec &p) { return p.w2 & 0xFFFFu; }
AZD_HD uint32_t pred_arc(const PredRec &p) { return p.w2 >> 16; }
AZD_HD uint32_t pred_child_nt(const PredRec &p) { return p.w1 & 0xFFFFFu; }
AZD_HD uint32_t pred_child_count(const PredRec &p) { return (p.w1 >> 20) & 0x7FFu; }
AZD_HD bool pred_child_active(const PredRec &p) { return (p.w1 >> 31) != 0u; }
AZD_HD uint32_t pred_child_begin(const PredRec &p) { return p.w3 >> 12; }
// the summary words of node r as its parents' predictions keep them
AZD_HD uint32_t summary_w1(const NodeRec &r) {
    const uint32_t cnt = r.act_end - r.act_begin;
    return node_nt(r) | (cnt << 20) | (node_exhausted(r) < cnt ? 0x80000000u : 0u);
}

// flags[agent] bits
enum : uint32_t {
    FLAG_NODE_CAP = 1u << 0,
    FLAG_ARC_CAP = 1u << 1,
    FLAG_PRED_CAP = 1u << 2,
    FLAG_FRONTIER_CAP = 1u << 3,
    FLAG_HT_FULL = 1u << 4,
    FLAG_UNREACHABLE = 1u << 5,
    FLAG_NODE_ACTIONS = 1u << 6,
    FLAG_LOOP_GUARD = 1u << 7,
    // not a failure: the agent sits out the launches of a recovery round (engine.hip: an aborted dense pool launch is completed by
    // the launch-per-phase kernels, agent by agent from where each one stands); set and cleared by k_park
    FLAG_PARKED = 1u << 30,
};

struct ArgminRec { // device copy of azd_argmin
    uint8_t parents[32];
    uint64_t permitted[4];
    double lambda_1;
    int32_t matching_size;
    int32_t matching[32];
    float eval;
    int32_t agent;
    uint32_t node;
};

struct RamseyArgminRec { // device copy of azd_ramsey_argmin
    uint8_t colors[256];
    uint64_t permitted[4];
    int32_t totals[4];
    float eval;
    int32_t agent;
    uint32_t node;
};

struct DenseArgminRec { // device copy of azd_dense_argmin
    uint64_t adj[64];
    uint64_t permitted[40]; // modifiable slots (colex positions) still open
    double lambda_1;
    int32_t matching_size;
    float eval;
    int32_t agent;
    uint32_t node;
};

struct StatusRec { // small device block copied back after every host-visible call
    unsigned long long improved;   // k_argmin calls that improved the argmin
    unsigned long long expansions; // sum over agents and calls (metric numerator)
    unsigned long long failed;     // agents with a non-zero flag
    unsigned long long pool_abort; // the pool launch before this read-back ended on PoolCtl::abort (k_argmin_log1 copies the flag here)
    unsigned long long pool_eval_busy, pool_search_busy, pool_ticks, pool_pad; // PoolCtl's busy sums and t_last - t_first of that launch
};

struct Arenas {
    NodeRec *nodes;
    uint64_t *keys;
    ArcRec *arcs;
    PredRec *preds;
    float *pred_g;          // [B][pred_cap] g_theta_sa of the EXPANDED predictions (PredRec: w1 holds it until then); written once, read by the export only
    uint32_t *ht;
    uint32_t node_cap, arc_cap, pred_cap, ht_cap; // ht_cap is a power of two >= 128
    // per agent
    uint8_t *root_parents;  // [B][PARENTS_STRIDE]
    uint64_t *root_perm;    // [B][KW]
    uint8_t *cur_parents;
    uint64_t *cur_perm;
    uint64_t *cur_path;
    double *cur_lambda;
    int32_t *cur_mu;
    uint32_t *state_pos;
    uint32_t *n_nodes, *n_arcs, *n_preds;
    uint32_t *flags;
    uint32_t fr_lds;        // entries of a frontier level kept in LDS: FRONTIER_CAP (test hook AZD_DEBUG_FRONTIER_LDS: 64, 128 or 192)
    uint32_t *fr_spill;     // [B][2][node_cap][2] cascade frontier entries beyond the FRONTIER_CAP the wave's LDS holds: (node, x) per level
    float *cand_c;          // first-min eval over nodes created since the last argmin inspection
    uint32_t *cand_node;
    unsigned long long *counters; // [B][NUM_COUNTERS]
    float *state_vecs;      // [B][S]
    uint16_t *state_vecs16; // [B][S16] the same rows as bf16 (RNE), zero beyond S, written beside the f32 rows by the spaces that have the
    int S16;                // hook (SP::write_vec16): the bf16 evaluator's GEMM takes them as they are; null: no copy is kept
    float *h_theta;         // [B][A]
    float *obs;             // [B][A]
    float *weights;         // [B][A]
    ArgminRec *argmin;
    StatusRec *status;
    int n, A, S, KW, B;
    int t0, tn;             // launch-per-phase kernels over a SUB-population: agents t0 .. t0 + tn - 1 (tn = 0: all B); see engine.hip
    float eval_slope;       // squish slope 1/(C_UPPER - C_LOWER), 04-c21-tree.rs:58-74
    double lam_lo, lam_hi;  // c21: initial bracket of the lambda_1 multisection (c21_host.cpp:c21_lambda_bracket)
    // ---- Ramsey space (space_ramsey.inc); unused (null / 0) for c21
    int space;              // SPACE_C21 / SPACE_RAMSEY
    int C, E;               // colours, edges N(N-1)/2;  A = E*C, S = E*(2C+1)
    int sizes[4];           // clique size per colour (2..5)
    float cweights[4];      // weight per colour in evaluate
    uint32_t *root_nbr, *cur_nbr;      // [B][4*32] neighbourhood bitsets per colour
    int32_t *root_counts, *cur_counts; // [B][C*E]
    int32_t *root_tot, *cur_tot;       // [B][4]
    RamseyArgminRec *argmin_r;
    // ---- dense-graph space (space_dense.inc); null for the other spaces.  E = N(N-1)/2 edge slots, A = 2E, S = 3E + 1,
    // KW = 2, 4, 10 or 16 (keys over the ranks of the root's modifiable slots: at most 64 KW of them)
    uint64_t *root_adj, *cur_adj; // [B][64] neighbourhood bitsets
    uint16_t *root_aid;           // [B][64 KW] rank -> action id, ascending (0xFFFF beyond the root's k slots)
    uint32_t dense_p24;           // edge probability of a fresh root of the device root policy, x 2^24 (azd_engine_config::dense_p)
    DenseArgminRec *argmin_d;
    uint8_t *node_mate; // [B][node_cap][64] a maximum matching of every tree node's graph (0xFF: unmatched): a new node repairs its parent's
    // ---- path encoding P (az-discrete-opt/src/path/): PATH_SET = ActionSet (= ActionMultiset on
    // ActionsNeverRepeat spaces), PATH_SEQUENCE = ActionSequence (= OrderedActionSet): no transpositions
    int path_kind;
    // ---- Layered<L, Space> (az-discrete-opt/src/space/layered.rs; nabla/space/mod.rs:41-111): the evaluator
    // sees the last `layers` states of the current path; S = layers * S_inner.  The ring of states is not
    // stored: a state of the path is the root plus a prefix of the path's actions, kept in order in cur_seq.
    int layers, S_inner;
    uint16_t *cur_seq; // [B][MAX_NODE_ACTIONS] actions of the current path in the order taken (layers > 1)
    // The nodes of the current path, root first: the cascade that follows a terminal or a transposition visits them all
    // (they are ancestors of the arc), and fetches their records in one round trip instead of one per level.
    uint32_t *cur_stack; // [B][PATH_STACK]
};

// what the persistent step needs to run the evaluator inside the kernel
struct FusedEval {
    int kind;                // 0 not fusable (external), 1 TrivialModel, 2 hash stream, 3 MLP,
                             // 4 hash stream served by the pool step's EVALUATOR workgroups (test harness: the rows of the fixed
                             // prediction stream take the MLP's way through the queues, so the oracle can check whole launches)
    uint64_t seed, first_agent, call_base; // hash stream; call_base = index of the launch's first call (0 for the other kinds, whose argument block then stays the same from launch to launch)
    const float *params;     // MLP: flat parameters (per layer W[out][in] then b[out])
    int n_layers;
    int dims[8];
    int final_act;
    int max_hidden;
    // the two ping-pong activation buffers of a row are sized independently: layer l < L-1 writes its output to
    // buffer l & 1, so buffer 0 holds the widest EVEN-layer output and buffer 1 the widest ODD-layer output
    // (04-c21-tree.rs:46-52: 304-512-1024-512-152 needs 512 + 1024 floats, not 2 x 1024)
    int hid[2];
    long long w_off[7], b_off[7];
    // bf16 weight storage (AZD_STORAGE_BF16): the same layout as `params` in 16-bit words; products are
    // exact in f32, accumulation is f32 (v_mfma_f32_16x16x16_bf16), activations are rounded to bf16 as
    // they are read; biases stay f32 in `params`
    int bf16;
    const uint16_t *w16;
    // the asynchronous step's weight stream: the same values in MFMA-fragment order, so that one k-step
    // of a 16-column tile is ONE contiguous piece (1 KB f32 / 512 B bf16) instead of 16 pieces of 64 B
    // (mlp_kernels.hip:k_pack_weights).  Layer l, tile j, k-step s, lane, i = 0..3:
    //   wpk[p_off[l] + ((j * ceil(K/16) + s) * 64 + lane) * 4 + i] = W_l[16 j + (lane & 15)][16 s + 4 (lane >> 4) + i]
    // (0 outside the matrix); f32 words, or bf16 halves under AZD_STORAGE_BF16
    const void *wpk;
    long long p_off[7];
};

struct TolTable {
    uint32_t tol[MAX_TOL];
    int n_tol;
    uint32_t tol_default;
};

// ---- pool step (pool_step.inc): agents are not bound to waves.  Searcher workgroups pull ready agents from
// per-XCD queues, evaluator workgroups pull batches of posted rows from per-XCD queues; all in device memory.
constexpr int POOL_XCDS = 8;
struct PoolQ { // head and tail on cache lines of their own
    uint32_t head, pad0[31];
    uint32_t tail, pad1[31];
};
struct PoolCtl {
    uint32_t claim_next, pad0[31];  // next agent no searcher has taken yet in this launch
    uint32_t done_agents, pad1[31]; // agents through all their calls
    uint32_t abort, pad2[31];       // a wait ran into its bound: every loop leaves
    // agents whose prediction row has arrived, by home XCD: ticket queues of agent ids.  Plain mode
    // (PoolArgs::ready_lanes == 0): everything goes through ready[1].  Lane mode (more agents than searching waves):
    // ready[0] takes the agents behind the mean progress; a wave looking for work takes a ticket of ready[0] while that
    // queue holds agents or has fewer than a handful of waves standing by, so a lagging agent never waits for a wave.
    PoolQ ready[2][POOL_XCDS];
    PoolQ evalq[2][POOL_XCDS];      // agents waiting for a prediction row, by home XCD; [0] = agents that lag behind, served first
    struct {
        uint32_t v, pad[31];
    } sum_calls[POOL_XCDS], claimed[POOL_XCDS], // calls completed by / agents living on each XCD: their ratio is the mean progress
      done_x[POOL_XCDS],                        // agents of each XCD that are through all their calls (they no longer count for it)
      max_lag[POOL_XCDS];                       // express mode: the largest lag behind the XCD's progress any of its agents has posted with
    // what the launch's two sides were busy with, in ticks of the 100 MHz wall clock (the host balances the split by them):
    unsigned long long eval_busy;   // evaluator workgroups: from a batch taken to the batch released, summed over workgroups
    unsigned long long search_busy; // searcher waves: with an agent in hand (add_actions + calls), summed over waves
    unsigned long long t_first;     // when the first workgroup started (first writer) ...
    unsigned long long t_last;      // ... and the last one ended
};
struct PendRec { // what a call that ended on a new node leaves for the add_actions that follows the evaluator
    uint32_t pos;
    float c;
    uint32_t n_preds;
    uint32_t call; // calls of this launch the agent has completed
};
struct PoolArgs {
    PoolCtl *ctl;
    uint32_t *ready_slots; // [2][POOL_XCDS][qcap]  agent + 1, 0 = empty
    uint32_t *eval_slots;  // [2][POOL_XCDS][qcap]  (agent + 1) | home XCD << 24
    int ready_lanes;       // 1: lane mode; 2: express mode -- the last n_express searcher workgroups run express_waves waves each (a wave
                           // searches faster with its SIMD to itself: 39 us per call against 50 with 16 waves per CU) and serve ready[0]
                           // only, which takes the agents that lag behind their XCD's progress: the launch lasts as long as the
                           // slowest agent's chain of calls
    int n_express;
    uint32_t express_waves;
    uint32_t express_shift; // an agent lags when its calls + 2 + (ref >> express_shift) < ref, ref = mean calls of the XCD's unfinished agents
    int early_post;        // the request for a prediction row leaves before the new node's cost is computed: 1 always, 2 when
                           // the wave had to wait for the agent (waves idle: latency-bound), 0 never
    uint32_t qcap;         // power of two >= 2 * B
    uint32_t *join;        // [B] +1 by the wave that posted the agent's row once its own stores are out, +1 by
                           // the evaluator once the prediction row is out; whoever brings it to an even count queues the agent
    PendRec *pend;         // [B]
    unsigned long long *stamp; // [B] diagnostic build (make PROFILE=1): when the agent was posted / its row stored
    int n_eval;            // blocks [0, n_eval) are evaluator workgroups
    uint32_t eval_stride;  // floats per row of an evaluator batch in LDS
    uint32_t eval_out_off; // offset (floats) of the head's output inside a row
    uint32_t eval_rows;    // rows an evaluator batch may hold: 16, or 32 (two MFMA row tiles per weight fragment)
    uint32_t *post_call;   // [B] FusedEval::kind 4 only: the call whose row the agent has posted (the evaluator hashes it)
    // Evaluator GROUPS (round 5; pool_step.inc: pool_eval_group): blocks [0, grp_groups * grp_g) serve batches g workgroups at a time,
    // member j keeping the weight fragments of its column tiles of every layer in LDS for the whole launch; the other evaluator
    // blocks [grp_groups * grp_g, n_eval) are workgroups of the classic form.  grp_g = 0: no groups.
    int grp_g, grp_w, grp_groups;    // members per group; waves per member and batch slot (= column tiles per member and layer, at most); groups
    const int16_t *grp_tile;         // [layer][member][w]: the column tile, -1 none
    const uint32_t *grp_lds;         // [layer][member][w]: byte offset of the tile's fragments in the member's LDS
    uint32_t *grp_desc;              // [group][slot][64]: word 0 batch number (0xFFFFFFFF: leave), 1 rows, 16.. agents, 32.. request bytes
    uint32_t *grp_cnt;               // [group][slot][8][32]: column tiles of layer l done, over all batches of the slot (a line each)
    uint32_t *grp_flag;              // [group][slot][member][16]: word l = the last batch whose layer l is complete, word 7 = the last batch published
                                     // (0xFFFFFFFF: leave); written by ONE wave (the layer's last arriver / the slot's leader), polled by that member only
    float *grp_x;                    // [group][slot][2][grp_xstride]: a batch's activations between the layers, fragment-major
    uint32_t grp_xstride;            // floats: 256 x the widest hidden layer's column tiles
    uint32_t debug_abort_call; // test hook (AZD_POOL_DEBUG_ABORT_CALL = k): evaluator workgroup 0 raises PoolCtl::abort after its k-th batch; 0: off
    // Run-ahead window (azd_engine_run_ahead; the kernel's window instantiation only): the wave that completes a call for the
    // LAST agent hands the call's argmin candidate to the host while the launch goes on
    uint32_t *win_count;           // [log_calls] agents through call i (device memory, zero when the launch starts)
    unsigned long long *win_log;   // [log_calls] pinned host memory: log_key[i] once every agent is through call i
    uint32_t *win_flag;            // [log_calls] pinned host memory: 1 once win_log[i] is in
};

// Per-launch values of the CU-resident step forms that are not part of the argument block in device memory (PersistArgs, which
// is re-sent only when something in it changed: with a model evaluator, never between the launches of an epoch).
struct StepLaunch {
    int n_calls;
    unsigned long long *log_key;   // [>= n_calls] per-call argmin candidates, all ones when the launch starts (k_argmin_log1 leaves them so)
    // k_async taking over from an aborted pool launch (engine.hip): [B] calls the agent has completed | 1u << 31 if its last
    // call ended on a new node whose prediction row / add_actions are still due.  Null: every agent starts at call 0.
    const uint32_t *resume;
    PoolCtl *ctl;                  // pool step: the control block k_argmin_log1 reports the abort flag from and clears; else null
    int hashed;                    // pool step: FusedEval kind 4, the harness' instantiation of the kernel (k_pool<SP, 1>)
    int groups = 0;                // pool step: PoolArgs::grp_g > 0 -- the instantiation with the evaluator groups (k_pool<SP, 4>; c21 space)
    int window;                    // pool step: publish every completed call to the host (PoolArgs::win_*)
};

struct PersistArgs { // argument block of the persistent step, read from device memory
    Arenas a;
    TolTable tol;
    FusedEval ev;
    PoolArgs pool;
};

// kernel launchers (tree_kernels.hip); all asynchronous on `stream`.  The *_plan functions lay out the LDS of a
// CU-resident step; when the model or the population does not fit they return false and say why in *why.
void launch_init_roots(const Arenas &a, const uint8_t *d_parents, const uint64_t *d_permitted, void *stream);
void launch_add_actions(const Arenas &a, int root_mode, void *stream);
void launch_rollout(const Arenas &a, const TolTable &tol, void *stream);
void launch_argmin(const Arenas &a, int init_mode, void *stream);
// launch-per-phase form over sub-populations on streams of their own (engine.hip): the candidates of agents a.t0 .. a.t0 + a.tn - 1
// since their last inspection go into log_key[*call_ctr] (atomic min), the counter is bumped; replayed by launch_argmin_log
void launch_log_candidates(const Arenas &a, unsigned long long *log_key, uint32_t *call_ctr, void *stream);
void launch_argmin_log(const Arenas &a, int n_calls, unsigned long long *log_key, void *stream);
// one candidate (agent, node) replayed into the argmin records `a` points at; StatusRec untouched (run-ahead window, engine.hip)
void launch_argmin_one(const Arenas &a, int agent, uint32_t node, void *stream);
void ramsey_launch_argmin_one(const Arenas &a, int agent, uint32_t node, void *stream);
void launch_observe(const Arenas &a, uint32_t n_obs_tol, void *stream);
bool async_plan(const Arenas &a, const FusedEval &ev, uint32_t *dyn_stride, size_t *dyn_bytes, const char **why = nullptr);
void launch_async(const Arenas &a, const PersistArgs *d_args, const StepLaunch &sl,
                  const float *params, const void *wpk, uint32_t dyn_stride, size_t dyn_bytes, void *stream);
bool persist_plan(const Arenas &a, const FusedEval &ev, uint32_t *dyn_stride, size_t *dyn_bytes, const char **why = nullptr);
// pool step (pool_kernels.hip / ramsey_pool_kernels.hip): the plan also lays out an evaluator batch (pool->eval_*)
bool pool_plan(const Arenas &a, const FusedEval &ev, PoolArgs *pool, uint32_t *dyn_stride, size_t *dyn_bytes, const char **why = nullptr);
void launch_pool(const Arenas &a, const PersistArgs *d_args, const StepLaunch &sl, const float *params,
                 const void *wpk, int n_blocks, uint32_t dyn_stride, size_t dyn_bytes, void *stream);
// workgroups of k_pool the device can hold at once with this much dynamic LDS (occupancy query x CUs): the pool step's
// searcher and evaluator workgroups spin-wait on each other, so all of them must be resident together
int pool_max_resident(const Arenas &a, size_t dyn_bytes, int n_cus);
int ramsey_pool_max_resident(const Arenas &a, size_t dyn_bytes, int n_cus);
// after an aborted pool launch: resume[t] for k_async (StepLaunch::resume) from the trees and PoolArgs::pend
void launch_pool_resume_scan(const Arenas &a, const PoolArgs &pool, int n_calls, uint32_t *resume, void *stream);
// test entry: the in-kernel evaluator's forward (pool_eval's staging + mlp_tile_task) for rows given by the host
hipError_t launch_tile_forward(const FusedEval &ev, const PoolArgs &pool, int n_rows, const float *states, float *out, void *stream);
bool ramsey_pool_plan(const Arenas &a, const FusedEval &ev, PoolArgs *pool, uint32_t *dyn_stride, size_t *dyn_bytes, const char **why = nullptr);
void ramsey_launch_pool(const Arenas &a, const PersistArgs *d_args, const StepLaunch &sl, const float *params,
                        const void *wpk, int n_blocks, uint32_t dyn_stride, size_t dyn_bytes, void *stream);
void launch_probe_xcc(uint32_t *d_out, int n_blocks, void *stream); // HW_REG_XCC_ID of every block of a launch (tests)

// the launch-per-phase kernels for SPACE_DENSE (dense_kernels.hip); the c21 entry points forward to them.  The space's
// state vector (3E + 1 floats) does not fit the CU-resident forms' LDS plans: it runs one launch per phase.
void dense_launch_init_roots(const Arenas &a, const uint8_t *d_adj, const uint64_t *d_packed, void *stream);
void dense_launch_add_actions(const Arenas &a, int root_mode, void *stream);
void dense_launch_rollout(const Arenas &a, const TolTable &tol, void *stream);
void dense_launch_argmin(const Arenas &a, int init_mode, void *stream);
void dense_launch_argmin_log(const Arenas &a, int n_calls, unsigned long long *log_key, void *stream);
void dense_launch_observe(const Arenas &a, uint32_t n_obs_tol, void *stream);
// pool step of the dense-graph space: searcher workgroups only (k_pool_search); the evaluator is a stream of batched GEMM launches
// over the rows the searchers have posted, collected by k_ext_take and handed back by k_ext_deliver (pool_step.inc)
bool dense_pool_plan(const Arenas &a, int waves, uint32_t *dyn_stride, size_t *dyn_bytes, const char **why);
void dense_launch_pool_search(const Arenas &a, const PersistArgs *d_args, const StepLaunch &sl, int n_blocks, int waves, uint32_t dyn_stride,
                              size_t dyn_bytes, void *stream);
int dense_pool_search_resident(const Arenas &a, int waves, size_t dyn_bytes);
void launch_ext_take(const PoolArgs &pool, uint32_t *rows, uint32_t *home, uint32_t *n, unsigned long long *t0, void *stream);
// recovery of an aborted dense pool launch (engine.hip): park (round >= 0: the agents whose calls are through by that round; -1: the
// agents that are not waiting for a row) / unpark (mode 0), and the candidates of round r under the call each agent is really in
void launch_park(const Arenas &a, const uint32_t *resume, int n_calls, int round, int park, void *stream);
void launch_log_candidates_resume(const Arenas &a, unsigned long long *log_key, const uint32_t *resume, int n_calls, int round, void *stream);
void launch_ext_hash_rows(const PersistArgs *d_args, const uint32_t *rows, const uint32_t *n, uint32_t cap, float *h_theta, void *stream);
void launch_ext_deliver(const PoolArgs &pool, const Arenas &a, const uint32_t *rows, const uint32_t *home, const uint32_t *n, uint32_t cap,
                        const unsigned long long *t0, void *stream);
void launch_persist(const Arenas &a, const PersistArgs *d_args, const StepLaunch &sl,
                    uint32_t *log_node, uint32_t dyn_stride, size_t dyn_bytes, void *stream);
void launch_c21_modify_roots(const Arenas &a, uint64_t seed, uint64_t epoch, uint64_t first_agent, int kmin, int kmax,
                             uint8_t *d_parents, uint64_t *d_perm, void *stream);
// dense-graph space: d_packed = what k_init_roots takes (17 KW words per root), d_slots = the drawn slot masks ((E + 63) / 64 words per root)
void dense_launch_modify_roots(const Arenas &a, uint64_t seed, uint64_t epoch, uint64_t first_agent, int kmin, int kmax,
                               uint8_t *d_adj, uint64_t *d_packed, uint64_t *d_slots, void *stream);
void launch_hash_predictions(float *d_out, int batch, int action_dim, uint64_t seed, uint64_t first_agent,
                             uint64_t call, void *stream);
void launch_probe_cost(const uint8_t *d_parents, int n, int count, int reps, int full, double *d_lam, int *d_mu,
                       void *stream);
void launch_probe_math(const float *d_in, float *d_out, int n, void *stream); // sqrtf / sub parity probe (tests)

// the same launchers for SPACE_RAMSEY (ramsey_kernels.hip); the functions above forward to them
void ramsey_launch_init_roots(const Arenas &a, const uint8_t *d_colors, const uint64_t *d_permitted, void *stream);
void ramsey_launch_add_actions(const Arenas &a, int root_mode, void *stream);
void ramsey_launch_rollout(const Arenas &a, const TolTable &tol, void *stream);
void ramsey_launch_argmin(const Arenas &a, int init_mode, void *stream);
void ramsey_launch_argmin_log(const Arenas &a, int n_calls, unsigned long long *log_key, void *stream);
void ramsey_launch_observe(const Arenas &a, uint32_t n_obs_tol, void *stream);
void ramsey_launch_modify_roots(const Arenas &a, uint64_t seed, uint64_t epoch, uint64_t first_agent, int kmin, int kmax,
                                uint8_t *d_colors, uint64_t *d_perm, void *stream);
bool ramsey_async_plan(const Arenas &a, const FusedEval &ev, uint32_t *dyn_stride, size_t *dyn_bytes, const char **why = nullptr);
void ramsey_launch_async(const Arenas &a, const PersistArgs *d_args, const StepLaunch &sl,
                         const float *params, const void *wpk, uint32_t dyn_stride, size_t dyn_bytes, void *stream);
bool ramsey_persist_plan(const Arenas &a, const FusedEval &ev, uint32_t *dyn_stride, size_t *dyn_bytes, const char **why = nullptr);
void ramsey_launch_persist(const Arenas &a, const PersistArgs *d_args, const StepLaunch &sl,
                           uint32_t *log_node, uint32_t dyn_stride, size_t dyn_bytes, void *stream);

} // namespace azd
