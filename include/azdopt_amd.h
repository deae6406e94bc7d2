/*
 * azdopt_amd.h -- C ABI of the MI355X-native self-play hot path of ariasanovsky/azdopt.
 *
 * The reference has no FFI layer; its seams are Rust traits.  Each entry point
 * below replaces one trait method / pub fn (cited as file:line relative to the
 * reference root) with the same argument meaning, ownership and layout:
 *
 *   evaluator seam   trait NablaModel              az-discrete-opt/src/nabla/model/mod.rs:4-8
 *                    ActionModel::{new,...}        az-discrete-opt/src/nabla/model/dfdx.rs:36-131
 *   engine seam      NablaOptimizer pub methods    az-discrete-opt/src/nabla/optimizer/mod.rs:30-36,39,121,249,284,361
 *   space seam       NablaStateActionSpace         az-discrete-opt/src/nabla/space/mod.rs:5-39
 *                    (device-resident spaces are built in and selected by id;
 *                     the trait stays the host-side description)
 *
 * Conventions: plain pointers and sizes, no torch types.  Host slices are lent
 * for the duration of the call (as the Rust `&[f32]` / `&mut [f32]` are) and are
 * never retained.  All matrices are row-major, agent-major.  Every function
 * returns an azd_status (0 = OK); nothing aborts the process (the reference
 * panics with panic='abort', graph-state/Cargo.toml:59,62).
 *
 * The library has NO CPU fallback: every compute entry point runs HIP kernels
 * on a gfx950 device and fails with AZD_ERR_NO_DEVICE when none is present.
 */
#ifndef AZDOPT_AMD_H
#define AZDOPT_AMD_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef enum azd_status {
    AZD_OK = 0,
    AZD_ERR_INVALID_ARGUMENT = 1,
    AZD_ERR_NO_DEVICE = 2,
    AZD_ERR_HIP = 3,
    AZD_ERR_CAPACITY = 4,      /* a tree arena (nodes / arcs / predictions / frontier) overflowed */
    AZD_ERR_UNREACHABLE = 5,   /* the reference's unreachable!() at tree/mod.rs:227 */
    AZD_ERR_NO_EVALUATOR = 6,  /* engine created without an evaluator: use the *_begin/_end pair */
    AZD_ERR_OUT_OF_MEMORY = 7,
    AZD_ERR_UNSUPPORTED = 8
} azd_status;

const char *azd_status_string(int status);
/* text of the last HIP error seen by the calling thread ("" if none) */
const char *azd_last_error(void);
int azd_version(void);
/* number of visible gfx950 devices (0 on a box without a GPU; never fails) */
int azd_device_count(void);

/* ------------------------------------------------------------------------- */
/* Space seam: c21 = ROTModifyParentsOnce<N, Conjecture2Dot1Cost>            */
/*   graph-state/src/rooted_tree/space.rs:14-125                             */
/* ------------------------------------------------------------------------- */
#define AZD_SPACE_C21 1
#define AZD_C21_MAX_N 24
int azd_c21_state_dim(int n);  /* (N-1)(N-2)-2     space.rs:46 */
int azd_c21_action_dim(int n); /* (N-1)(N-2)/2-1   space.rs:48 */
int azd_c21_key_words(int n);  /* u64 words of an ActionSet bit mask */

/* Host-side `init_states` closure of the driver (04-c21-tree.rs:108-112;
 * rooted_tree/mod.rs:14-20; modify_parent_once.rs:14-25) with a seeded,
 * counter-based generator instead of thread_rng (spec: DESIGN.md).
 * Packed root format used everywhere below:
 *   parents   [count][n]  u8   parents[v] < v, parents[0]=parents[1]=parents[n-1]=0
 *   permitted [count][KW] u64  bit a set <=> action id a is permitted */
int azd_c21_generate_roots(uint64_t seed, uint64_t epoch, uint64_t first_agent, int count, int n,
                           int kmin, int kmax, uint8_t *parents, uint64_t *permitted);

/* ------------------------------------------------------------------------- */
/* Space seam: Ramsey = RamseySpaceNoEdgeRecolor<B32, N, E, C>               */
/*   graph-state/src/ramsey_counts/space.rs:10-176 (drivers 01-r333.rs,      */
/*   02-r44.rs): E = N(N-1)/2 edges in colex order (simple_graph/edge.rs),   */
/*   action id = edge + new_colour * E.  Built for E <= 256, 2..4 colours,   */
/*   clique sizes 2..5, E*C <= 384 (r333: N=16, r44: N=17).                  */
/* ------------------------------------------------------------------------- */
#define AZD_SPACE_RAMSEY 2
#define AZD_RAMSEY_MAX_N 23
int azd_ramsey_state_dim(int n, int n_colors);  /* E(2C+1)  space.rs:40 */
int azd_ramsey_action_dim(int n, int n_colors); /* EC       space.rs:42 */
int azd_ramsey_key_words(int n, int n_colors);
/* Host-side `init_state` closure of the drivers (01-r333.rs:84-90: ColoredCompleteBitsetGraph::
 * generate with uniform colour weights + RamseyCountsNoRecolor::generate), seeded.
 * Packed root format of this space, passed through the same `parents` / `permitted`
 * arguments of the engine calls below:
 *   colors    [count][E]  u8   colour of the edge at each colex position
 *   permitted [count][KW] u64  bit e set <=> edge position e may still be recoloured */
int azd_ramsey_generate_roots(uint64_t seed, uint64_t epoch, uint64_t first_agent, int count, int n,
                              int n_colors, int kmin, int kmax, uint8_t *colors, uint64_t *permitted);

/* ------------------------------------------------------------------------- */
/* Space seam: dense graphs (BUILD-DEFINED: the reference has no live         */
/*   NablaStateActionSpace over general graphs, examples/05-ah.rs is a stub)  */
/*   built from ConnectedBitsetGraph (connected_bitset_graph/mod.rs:45-71,    */
/*   139-158, 226-338), AddOrDeleteEdge (bitset_graph/space/action.rs:4-27)   */
/*   and the STATE = E + ACTION + 1 hint (05-ah.rs:39-40); definition in      */
/*   oracle/dense_graph.inc.  BASELINE.json configs[4]: N = 50.               */
/* ------------------------------------------------------------------------- */
#define AZD_SPACE_DENSE 3
#define AZD_DENSE_MAX_N 64
#define AZD_DENSE_MAX_SLOTS 1024 /* modifiable edge slots of a root = legal actions a node can hold (azd_engine_config::max_slots) */
int azd_dense_state_dim(int n);  /* 3E + 1 */
int azd_dense_action_dim(int n); /* 2E: Add(e) = e, Delete(e) = E + e */
int azd_dense_key_words(int n);  /* u64 words of an action-id set */
/* Seeded `init_states`: a connected G(n, p) (redrawn until connected, as ConnectedBitsetGraph::generate does) and
 * k in [kmin, kmax] modifiable edge slots.  Packed root format of this space:
 *   adj   [count][n]  u64  neighbourhood bitsets (passed as the `parents` bytes of the engine calls: 8 n bytes per root)
 *   slots [count][KW] u64  bit e set <=> the edge slot at colex position e may be modified (once) */
int azd_dense_generate_roots(uint64_t seed, uint64_t epoch, uint64_t first_agent, int count, int n, int kmin, int kmax,
                             double p, uint64_t *adj, uint64_t *slots);

/* ------------------------------------------------------------------------- */
/* Evaluator seam (NablaModel)                                               */
/* ------------------------------------------------------------------------- */
typedef struct azd_evaluator azd_evaluator;

typedef struct azd_adam_config { /* dfdx AdamConfig as set at 04-c21-tree.rs:87-92 */
    float lr;
    float beta1;
    float beta2;
    float eps;
    float l2; /* WeightDecay::L2 */
} azd_adam_config;

#define AZD_ACT_NONE 0
#define AZD_ACT_RELU 1
#define AZD_ACT_SIGMOID 2

/* ActionModel::new (model/dfdx.rs:36-53): an MLP state_dim -> hidden[0] -> ... -> action_dim,
 * ReLU between layers, `final_act` on the head (04-c21-tree.rs:46-52), fp32, Adam. */
int azd_evaluator_create_mlp(azd_evaluator **out, int device, int max_batch, int state_dim,
                             int action_dim, const int *hidden, int n_hidden, int final_act,
                             const azd_adam_config *adam, uint64_t seed);
/* TrivialModel (model/mod.rs:10-23): leaves predictions untouched, loss 0. */
int azd_evaluator_create_trivial(azd_evaluator **out, int device, int state_dim, int action_dim);
/* Fixed prediction stream h(agent, call, a): the parity harness'
 * stand-in for a model, so MLP rounding cannot perturb tree topology. */
int azd_evaluator_create_hash_stream(azd_evaluator **out, int device, int state_dim, int action_dim,
                                     uint64_t seed, uint64_t first_agent);
int azd_evaluator_destroy(azd_evaluator *ev);

/* NablaModel::write_predictions (model/mod.rs:5; dfdx.rs:69-84).
 * states: batch*state_dim, predictions: batch*action_dim, host memory. */
int azd_evaluator_write_predictions(azd_evaluator *ev, int batch, const float *states,
                                    float *predictions);
/* NablaModel::update_model (model/mod.rs:6-7; dfdx.rs:86-131): w /= sum(w);
 * L = sum w (pred - obs)^2; backward; Adam.  Returns L in *loss. */
int azd_evaluator_update_model(azd_evaluator *ev, int batch, const float *states,
                               const float *observations, const float *action_weights, float *loss);
/* Same two calls on DEVICE pointers (what the engine uses internally and what a
 * multi-GPU host calls after its all-gather).  `stream` is a hipStream_t or NULL. */
int azd_evaluator_write_predictions_dev(azd_evaluator *ev, int batch, const float *d_states,
                                        float *d_predictions, void *stream);
int azd_evaluator_update_model_dev(azd_evaluator *ev, int batch, const float *d_states,
                                   const float *d_observations, const float *d_action_weights,
                                   float *loss, void *stream);
/* flat parameter vector: per layer W[out][in] then b[out] (dfdx Linear layout) */
/* Weight storage of the MLP evaluator for INFERENCE (write_predictions and the in-kernel evaluator):
 * AZD_STORAGE_BF16 = weights and layer inputs rounded to bf16 (RNE), products exact, f32 accumulate
 * (v_mfma_f32_16x16x16_bf16), biases and outputs f32; the optimiser keeps f32 master weights and the bf16
 * copy is refreshed after every step (BASELINE config C: "bf16 storage / fp32 accumulate").  Default f32. */
#define AZD_STORAGE_F32 0
#define AZD_STORAGE_BF16 1
int azd_evaluator_set_weight_storage(azd_evaluator *ev, int dtype);
int64_t azd_evaluator_num_params(azd_evaluator *ev);
int azd_evaluator_get_params(azd_evaluator *ev, float *out);
int azd_evaluator_set_params(azd_evaluator *ev, const float *in);
/* number of evaluator invocations so far (the `call` index of the hash stream) */
uint64_t azd_evaluator_calls(azd_evaluator *ev);

/* ------------------------------------------------------------------------- */
/* Engine seam (NablaOptimizer<Space, M, ActionSet>)                         */
/* ------------------------------------------------------------------------- */
typedef struct azd_engine azd_engine;

/* Upper limits of the per-tree arenas (what the packed prediction record holds; the reference's petgraph indices are u32 and
 * have none): azd_engine_create refuses a configuration beyond them with AZD_ERR_INVALID_ARGUMENT and an azd_last_error()
 * that names the argument. */
#define AZD_MAX_NODE_CAPACITY 65536
#define AZD_MAX_ARC_CAPACITY 65535
#define AZD_MAX_PREDICTION_CAPACITY (1 << 20)

typedef struct azd_engine_config {
    int space_id;      /* AZD_SPACE_C21 or AZD_SPACE_RAMSEY */
    int n;             /* vertices N (c21: 4..AZD_C21_MAX_N; Ramsey: 3..AZD_RAMSEY_MAX_N) */
    int batch;         /* BATCH: agents (trees) owned by this engine / GPU */
    int device;        /* HIP device ordinal */
    /* per-tree arena capacities; 0 = default sized for 800 calls per epoch (4096 / 8192 / 32768).  Upper limits, from what a
     * prediction record packs (node ids 16 bits, arc ids 16, a node's first prediction 20): 65536 nodes, 65535 arcs, 1048576
     * predictions per tree; ACTION_DIM <= 4096, <= 2047 actions per node.  Beyond them azd_engine_create returns
     * AZD_ERR_INVALID_ARGUMENT; an arena that fills during a run stops its agent and surfaces as AZD_ERR_CAPACITY. */
    int node_capacity;
    int arc_capacity;
    int prediction_capacity;
    uint64_t first_agent; /* global id of agent 0 (multi-GPU sharding) */
    uint32_t flags;       /* AZD_ENGINE_* */
    /* AZD_SPACE_RAMSEY only (RamseySpaceNoEdgeRecolor::new(sizes, weights), space.rs:17-24) */
    int n_colors;         /* C */
    int clique_sizes[4];  /* SIZES: forbidden clique size per colour */
    float color_weights[4];
    /* path encoding P of NablaOptimizer<Space, M, P> (az-discrete-opt/src/path/, licences in
     * space/axioms.rs:12-19); both built spaces are ActionsNeverRepeat + ActionOrderIndependent */
    int path_kind;        /* AZD_PATH_* */
    /* Layered<L, Space> history wrapper (az-discrete-opt/src/space/layered.rs; nabla/space/mod.rs:41-111):
     * 0 or 1 = plain space; L = 2..8: STATE_DIM becomes L * inner STATE_DIM and a state vector carries the
     * last L states of the agent's path, oldest first; chunks of states the ring does not hold yet keep
     * their earlier contents, as in the reference.  A root installed by par_new / par_reset_trees is a ring
     * of one (Layers::new). */
    int layers;
    /* AZD_SPACE_DENSE only: the most modifiable edge slots a root may bring (= legal actions a node can hold); 0 = 128.  The
     * engine keeps transposition keys over the ranks of a root's slots in 2, 4, 10 or 16 words: up to 128, 256, 640 or 1024
     * slots (E / 2 = 612 at N = 50).  dense_p: edge probability of the fresh roots the device root policy draws (the `p` of
     * azd_dense_generate_roots; 0 = 0.2). */
    int max_slots;
    float dense_p;
} azd_engine_config;
/* ActionSet (path/set.rs): key = set of actions taken; equal sets share a node (transpositions).
 * ActionMultiset (path/multiset.rs) coincides with it on ActionsNeverRepeat spaces: every count is 1,
 * so identity, Ord and len are the set's. */
#define AZD_PATH_SET 0
/* ActionSequence (path/sequence.rs) and OrderedActionSet (path/ord_set.rs, also a Vec that
 * push_unchecked appends to): key = actions in the order taken; a path only ever meets itself, so
 * the search graph is a tree, and BTreeMap order = lexicographic order of the sequences. */
#define AZD_PATH_SEQUENCE 1

/* run every phase of a call as its own kernel launch instead of the CU-resident persistent step
 * (the two forms produce identical trees; the persistent step is the fast one) */
#define AZD_ENGINE_NO_PERSISTENT_STEP 1u
/* The CU-resident step comes in three forms with identical results.  With none of the three flags below the
 * engine takes the pool step for populations of 256 agents and more and the asynchronous step below that, and
 * falls back by itself (pool -> asynchronous -> lock-step -> one launch per phase) when a form cannot take the
 * model; azd_engine_step_form says which form ran and why.
 * AZD_ENGINE_ASYNC_STEP: k_async -- one workgroup = 16 agents on 16 wavefronts, the agents of a workgroup drift
 * apart, an agent waiting for its prediction row serves MFMA tile tasks of the workgroup's evaluator (hidden layer
 * widths in multiples of 16, input width in multiples of 4).
 * AZD_ENGINE_BARRIER_STEP: k_persist -- the same residency in lock-step (a workgroup barrier around the evaluator
 * on every call), 12 % slower at 4096 agents (profiles/README.md). */
#define AZD_ENGINE_ASYNC_STEP 2u
#define AZD_ENGINE_BARRIER_STEP 4u
/* Third CU-resident form, identical results again (k_pool): agents are not bound to wavefronts -- searcher
 * workgroups pull ready agents from per-XCD queues and never wait for a prediction row, evaluator workgroups on
 * CUs of their own pull batches of up to 16 posted rows (full MFMA row tiles, an undisturbed weight stream).
 * Populations larger than the resident waves share them instead of running as serial rounds of workgroups. */
#define AZD_ENGINE_POOL_STEP 8u

/* ArgminData<State, Cost> (az-discrete-opt/src/log.rs:1-11) for the c21 space */
typedef struct azd_argmin {
    uint8_t parents[32];
    uint64_t permitted[4];
    double lambda_1;           /* Conjecture2Dot1Cost.lambda_1 */
    int32_t matching_size;     /* Conjecture2Dot1Cost.matching.len() */
    int32_t matching[32];      /* (parent, child) pairs of the matching */
    float eval;
    int32_t agent;             /* tree the state was found in */
    uint32_t node;             /* its node index in that tree */
} azd_argmin;

/* ArgminData<RamseyCountsNoRecolor, TotalCounts<C>> for the Ramsey space */
typedef struct azd_ramsey_argmin {
    uint8_t colors[256];       /* colour per colex edge position */
    uint64_t permitted[4];     /* permitted edge positions */
    int32_t totals[4];         /* TotalCounts: monochromatic cliques per colour */
    float eval;
    int32_t agent;
    uint32_t node;
} azd_ramsey_argmin;

/* ArgminData for the dense-graph space: the graph, its open slots, Conjecture2Dot1Cost */
typedef struct azd_dense_argmin {
    uint64_t adj[64];
    uint64_t permitted[40];    /* BITMAP over the E edge slots (colex positions) still modifiable: (E + 63) / 64 words
                                * (32 at AZD_DENSE_MAX_N), NOT an action-id set of azd_dense_key_words() words */
    double lambda_1;
    int32_t matching_size;
    float eval;
    int32_t agent;
    uint32_t node;
} azd_dense_argmin;

enum { /* indices into azd_engine_counters' output */
    AZD_CTR_EXPANSIONS = 0,     /* calls that ended on a new non-terminal node (metric numerator) */
    AZD_CTR_TERMINALS = 1,
    AZD_CTR_TRANSPOSITIONS = 2,
    AZD_CTR_VISITED_STEPS = 3,
    AZD_CTR_SELECT_CALLS = 4,
    AZD_CTR_SUM_DEG = 5,
    AZD_CTR_SUM_ACTIONS = 6,
    AZD_CTR_CASCADE_NODES = 7,
    AZD_CTR_NEW_PREDS = 8,
    AZD_CTR_ROOT_EXHAUSTED = 9,
    AZD_CTR_MAX_FRONTIER = 10,
    AZD_CTR_MAX_DEPTH = 11,
    AZD_CTR_CURIOSITY_PAIRS = 12,
    AZD_CTR_EVAL_LAYER_CLOCKS = 13, /* pool step: shader clocks the evaluator workgroups spent in the MLP's layers, summed (s_memtime); with
                                     * slot 31's 100 MHz ticks of the same spans it gives the clock the evaluator CUs actually held */
    AZD_CTR_TICKS_ADD_ACTIONS = 14, /* pool step, AZD_WAVE_PHASES builds: ticks from an agent taken to its arrived row's actions added */
    AZD_CTR_FAILED_AGENTS = 15,
    /* 16..22: 100 MHz ticks per phase of the roll-out kernel, summed over agents; filled only by
     * the diagnostic build (make PROFILE=1), 0 otherwise */
    AZD_CTR_TICKS_TOTAL = 16,
    AZD_CTR_TICKS_SELECT = 17,
    AZD_CTR_TICKS_LOOKUP = 18,
    AZD_CTR_TICKS_NEWNODE = 19,
    AZD_CTR_TICKS_CASCADE = 20,
    AZD_CTR_TICKS_MAX_CALL = 21, /* max over agents and calls of one agent's ticks in one call */
    AZD_CTR_TICKS_LAMBDA = 22,   /* inside NEWNODE: lambda_1 */
    AZD_CTR_TICKS_MATCHING = 23, /* inside NEWNODE: matching */
    AZD_CTR_TICKS_WAIT = 24,     /* waiting for / serving the in-kernel evaluator */
    /* evaluator service of the asynchronous step (always counted) */
    AZD_CTR_EVAL_BATCHES = 25,
    AZD_CTR_EVAL_ROWS = 26,
    AZD_CTR_EVAL_TILES = 27,
    AZD_CTR_TICKS_TILES = 28,   /* diagnostic build: ticks inside evaluator tile tasks */
    AZD_CTR_TICKS_BATCH = 29,   /* pool step, every build: 100 MHz ticks from batch taken to batch released, summed over the evaluator
                                 * workgroups (bench.py's evaluator_weight_stream / us_per_batch read it); the asynchronous step fills it
                                 * in the diagnostic build only */
    AZD_CTR_COUNT = 32
};

/* Allocates the device arenas.  `ev` may be NULL (external evaluator: drive the
 * engine with the *_begin/_end pairs and supply predictions yourself). */
int azd_engine_create(azd_engine **out, const azd_engine_config *cfg, azd_evaluator *ev);
int azd_engine_destroy(azd_engine *e);

/* NablaOptimizer::par_new (optimizer/mod.rs:39-118).  `init_states` stays on the
 * host: the caller passes the packed roots it produced. */
int azd_engine_par_new(azd_engine *e, const uint8_t *parents, const uint64_t *permitted);
/* NablaOptimizer::par_roll_out_episodes (optimizer/mod.rs:121-191), `n_calls`
 * times back to back without a host round trip.  n_as_tol crosses the ABI as
 * table + default (04-c21-tree.rs:136-138).  *improved = number of calls that
 * returned ArgminImprovement::Improved. */
int azd_engine_par_roll_out_episodes(azd_engine *e, const uint32_t *n_as_tol, int n_tol,
                                     uint32_t n_as_tol_default, int n_calls, int *improved);
/* Run-ahead window, for a host that asks for its episodes ONE CALL AT A TIME as the reference's drivers do
 * (04-c21-tree.rs:142-150: par_roll_out_episodes, then a look at the ArgminImprovement, `episodes` times per epoch): a
 * launch of the CU-resident step costs ~0.6 ms whatever it holds, so 800 launches of one call run at a fifth of the speed
 * of one launch of 800.  azd_engine_run_ahead starts the next `n_calls` calls NOW, in one launch, and returns at once;
 * the azd_engine_par_roll_out_episodes calls that follow (same n_as_tol; one call or any chunk at a time) launch nothing:
 * each waits until the kernel has published the calls it asks for -- a call is published when its last agent is through it
 * -- and returns their ArgminImprovements exactly as separate launches would have.  azd_engine_argmin_data (and the Ramsey
 * form) inside the window return the record as of the calls handed out so far; the winner's replay reads its tree, so they
 * wait for the launch first.  Any other engine call closes the window: it waits for the launch, and the calls that were not
 * asked for have run and stay run (counters and trees show them).  The evaluator must not be updated by calls of its own
 * while a window is open (azd_engine_par_update_model closes it like every engine call).
 * *accepted = 0 when this engine's step cannot publish calls while it runs (any form but the pool step; n_calls beyond one
 * launch, 1024; per-launch timing on): nothing is started then and the calls run when they are asked for, so a host may
 * call this unconditionally. */
int azd_engine_run_ahead(azd_engine *e, const uint32_t *n_as_tol, int n_tol, uint32_t n_as_tol_default, int n_calls,
                         int *accepted);
/* NablaOptimizer::par_update_model (optimizer/mod.rs:249-281) */
int azd_engine_par_update_model(azd_engine *e, uint32_t n_obs_tol, float *loss);
/* par_update_model for a population SHARDED over several GPUs (one engine per process and GPU, contiguous agent
 * ranges in rank order; the reference has no such form -- optimizer/mod.rs:249-281 on one device): the training
 * triple of this rank (optimizer/mod.rs:262-278) is all-gathered over RCCL on the engine's stream into
 * engine-owned buffers, then every rank takes the identical optimiser step on the pooled batch*world rows (the
 * loss normaliser sum(w) is global over the batch, model/dfdx.rs:106,110).  `nccl_comm` is the caller's ncclComm_t
 * (ncclCommInitRank on this engine's device); the evaluator must have been created with max_batch >= batch*world.
 * librccl.so is bound at first use; AZD_ERR_UNSUPPORTED if it cannot be loaded. */
int azd_engine_par_update_model_sharded(azd_engine *e, uint32_t n_obs_tol, void *nccl_comm, float *loss);
/* NablaOptimizer::par_reset_trees (optimizer/mod.rs:284-360) with the
 * `modify_root` closure applied by the caller (see azd_c21_modify_roots). */
int azd_engine_par_reset_trees(azd_engine *e, const uint8_t *parents, const uint64_t *permitted);
/* NablaOptimizer::argmin_data (optimizer/mod.rs:361) */
int azd_engine_argmin_data(azd_engine *e, azd_argmin *out);
int azd_engine_ramsey_argmin_data(azd_engine *e, azd_ramsey_argmin *out); /* AZD_SPACE_RAMSEY engines */
int azd_engine_dense_argmin_data(azd_engine *e, azd_dense_argmin *out);   /* AZD_SPACE_DENSE engines */
/* the agent's live per-edge clique counts [C][E] and totals [4] (RamseyCounts, mod.rs:12-17) */
int azd_engine_ramsey_agent_counts(azd_engine *e, int agent, int32_t *counts, int32_t *totals);

/* The `modify_root` policy of the c21 driver (04-c21-tree.rs:172-206), seeded;
 * reads each tree's node_data() (tree/mod.rs:302-307) and writes new packed roots. */
int azd_c21_modify_roots(azd_engine *e, uint64_t seed, uint64_t epoch, int kmin, int kmax,
                         uint8_t *parents_out, uint64_t *permitted_out);

/* The same policy evaluated on the device (one wavefront per tree, radix select of the chosen key
 * in BTreeMap order): _dev returns the roots it would install, par_reset_trees_c21 = policy +
 * par_reset_trees with no host round trip. */
int azd_c21_modify_roots_dev(azd_engine *e, uint64_t seed, uint64_t epoch, int kmin, int kmax,
                             uint8_t *parents_out, uint64_t *permitted_out);
int azd_engine_par_reset_trees_c21(azd_engine *e, uint64_t seed, uint64_t epoch, int kmin, int kmax);
/* The Ramsey drivers' modify_root (02-r44.rs:196-228) is the same policy over permitted EDGES
 * (k' of the E edges instead of k' of ACTION_DIM; a fresh root is a fresh random colouring), and
 * the device kernel serves both spaces.  Space-neutral names of the two calls above: */
int azd_engine_modify_roots_dev(azd_engine *e, uint64_t seed, uint64_t epoch, int kmin, int kmax,
                                uint8_t *roots_out, uint64_t *permitted_out);
int azd_engine_par_reset_trees_policy(azd_engine *e, uint64_t seed, uint64_t epoch, int kmin, int kmax);

/* Split-phase forms of the three calls above, cut at the model call
 * (optimizer/mod.rs:72, :175-176, :348), for an external NablaModel:
 * *_begin leaves the state vectors ready, *_end consumes h_theta (host, batch*action_dim). */
int azd_engine_par_new_begin(azd_engine *e, const uint8_t *parents, const uint64_t *permitted);
int azd_engine_par_new_end(azd_engine *e, const float *h_theta);
int azd_engine_roll_out_begin(azd_engine *e, const uint32_t *n_as_tol, int n_tol,
                              uint32_t n_as_tol_default);
int azd_engine_roll_out_end(azd_engine *e, const float *h_theta, int *improved);
int azd_engine_reset_begin(azd_engine *e, const uint8_t *parents, const uint64_t *permitted);
int azd_engine_reset_end(azd_engine *e, const float *h_theta);
/* par_update_model without the model call: fills the training triple
 * (optimizer/mod.rs:262-278).  Host copies (any may be NULL) ... */
int azd_engine_observe(azd_engine *e, uint32_t n_obs_tol, float *state_vecs, float *observations,
                       float *action_weights);
/* ... or the device buffers themselves, for an all-gather across GPUs. */
int azd_engine_observe_dev(azd_engine *e, uint32_t n_obs_tol, const float **d_state_vecs,
                           const float **d_observations, const float **d_action_weights);
int azd_engine_read_state_vecs(azd_engine *e, float *state_vecs); /* batch*state_dim */
int azd_engine_read_predictions(azd_engine *e, float *h_theta);   /* last h_theta, batch*action_dim */
/* Test entry: `rows` prediction rows exactly as the CU-resident step forms' in-kernel evaluator computes them (the same sums in
 * the same order; the model call of model/dfdx.rs:69-84), for state vectors the host hands over: what a checker feeds its
 * restatement with to follow a whole launch of the product kernel with the real model.  c21 / Ramsey space, MLP evaluator
 * (f32 or bf16 storage). */
int azd_engine_debug_tile_forward(azd_engine *e, const float *states, float *predictions, int rows);

/* Introspection (SearchTree::{nodes, positions, node_data}, tree/mod.rs:302-315;
 * get_trees, optimizer/mod.rs:34-36): raw arrays instead of graphviz. */
int azd_engine_tree_sizes(azd_engine *e, int agent, int *n_nodes, int *n_arcs, int *n_predictions);
int azd_engine_export_tree(azd_engine *e, int agent, float *c, float *c_star, uint32_t *n_t,
                           uint32_t *exhausted, uint32_t *act_begin, uint32_t *act_end,
                           uint64_t *keys, uint32_t *arc_src, uint32_t *arc_dst, uint32_t *arc_pp,
                           uint32_t *pred_a_id, float *pred_g, int32_t *pred_arc);
int azd_engine_agent_state(azd_engine *e, int agent, uint8_t *parents, uint64_t *permitted,
                           uint64_t *path, uint32_t *state_pos, double *lambda_1,
                           int *matching_size);
int azd_engine_counters(azd_engine *e, uint64_t *out /* [AZD_CTR_COUNT] */);
/* the same counters per agent, unreduced: out[batch][AZD_CTR_COUNT] (load-balance diagnostics).  The launch-per-phase and
 * asynchronous forms attribute every count to its agent.  The pool step of the product build adds a searcher wave's counts once,
 * when the wave leaves the launch, to ONE agent's block (the sums and maxima of azd_engine_counters are the same; the per-call
 * read-modify-write of the agent's block was a dependent round trip on the wave's time): per-agent attribution under the pool
 * step is the diagnostic build's (make PROFILE=1, tools/slow_agents.py). */
int azd_engine_agent_counters(azd_engine *e, uint64_t *out);
/* 1 while every block of azd_engine_agent_counters holds its own agent's counts; 0 once a pool launch of the product build has
 * run since the counters were last cleared (azd_engine_par_new): the blocks then hold what searcher waves counted and only their
 * sums / maxima (azd_engine_counters) mean anything.  Load-balance tools must check this before reading per-agent values. */
int azd_engine_agent_counters_per_agent(azd_engine *e);
/* ms of GPU time spent in the tree kernels / evaluator since creation (HIP events
 * on the engine's stream; enabled by azd_engine_set_timing) */
int azd_engine_set_timing(azd_engine *e, int enabled);
int azd_engine_timing(azd_engine *e, double *tree_ms, double *evaluator_ms, uint64_t *tree_launches);
void *azd_engine_stream(azd_engine *e); /* hipStream_t the engine launches on */
/* Which form of the step the last azd_engine_par_roll_out_episodes ran (identical results, very different
 * speed): the asynchronous CU-resident step, the lock-step CU-resident step, or one launch per phase and
 * call (external evaluators, models the in-kernel evaluator cannot take).  *reason (owned by the engine,
 * valid until the next call) says why a faster form was not taken, "" when the preferred one ran.
 * The pool step needs its searcher and evaluator workgroups resident together: the grid is clamped to what the occupancy query
 * reports, a device without room for one of each runs the asynchronous step, and a pool launch whose waits run into their
 * bound (4 s without progress) is TAKEN OVER by the asynchronous step where every agent stands -- same results, the call
 * succeeds, *reason says so and the engine stays with the asynchronous step (dense-graph space, whose evaluator runs beside the
 * kernel: completed by the launch-per-phase kernels in the same way, and the engine stays with those). */
#define AZD_STEP_NONE 0
#define AZD_STEP_ASYNC 1
#define AZD_STEP_BARRIER 2
#define AZD_STEP_PER_CALL 3
#define AZD_STEP_POOL 4
#define AZD_STEP_PER_CALL_GRAPH 5 /* AZD_STEP_PER_CALL with the launches of a call captured in a hipGraph and replayed */
int azd_engine_step_form(azd_engine *e, int *form, const char **reason);
/* how the last pool-step launch split the CUs: evaluator / searcher workgroups */
/* Evaluator groups of the last pool launch: `members` workgroups serve a batch together, each keeping the weight fragments of its
 * column tiles of every layer in LDS for the whole launch (no weight stream per batch; the batch's activations travel between the
 * layers through device memory), `groups` of them, a workgroup's waves in slots of `waves_per_slot`.  0 members: the classic form
 * (one workgroup per batch, weights streamed from L2).  The engine forms groups where a classic batch is long (fp32 weights beyond
 * 2.5 MB, at most 2048 agents: BASELINE configs[0], the reference's 304-512-1024-512-152); AZD_POOL_EVAL_GROUP = 0 / g overrides.
 * Same rows to the bit either way. */
int azd_engine_pool_groups(azd_engine *e, int *members, int *groups, int *waves_per_slot);
/* Pool step (product build): for every agent, when it was through with the calls of the LAST pool launch, in 100 MHz ticks from the
 * launch's first workgroup -- the launch lasts as long as its slowest agent's chain of calls (tree/mod.rs:113-232 is serial per
 * tree), so max over agents = the launch, and the distance of the median from it is what the epoch's barrier costs.
 * out[batch].  The diagnostic build (make PROFILE=1) uses the array for its own stamps: values are not finish times there. */
int azd_engine_pool_agent_finish(azd_engine *e, uint64_t *ticks_out);
int azd_engine_pool_split(azd_engine *e, int *eval_wgs, int *search_wgs);
/* ... and how busy its two sides were: the share of the launch an evaluator workgroup spent on batches, and the share a
 * searcher wave spent with an agent in hand (the engine moves the split towards equal shares from launch to launch) */
int azd_engine_pool_utilisation(azd_engine *e, double *eval_busy, double *search_busy);
/* HW_REG_XCC_ID read by every block of an n_blocks launch (the pool step keeps a tree on the XCD that first took it) */
int azd_debug_probe_xcc(int device, uint32_t *out, int n_blocks);
/* Test harness of the pool step: with `on`, a hash-stream evaluator's rows are served by the pool step's EVALUATOR workgroups
 * (through the queues, the early post and the join counter, like the MLP's) instead of by the searching wave itself, so that
 * whole launches of that machinery can be compared with the CPU oracle.  Same values either way.  Environment hooks of the
 * same harness: AZD_POOL_DEBUG_ABORT_CALL=k (agent 0 raises the launch's abort flag after its k-th call: the take-over by the
 * asynchronous step), AZD_POOL_MAX_RESIDENT=w (pretend the device holds w workgroups of the pool kernel at once). */
int azd_debug_hash_stream_via_evaluators(azd_evaluator *ev, int on);

/* Parity probe for the two f32 primitives the selection rule (tree/next_action.rs:70,81) depends
 * on bit-for-bit.  in: 2*n floats (pairs x, y); out: 4*n floats per pair:
 * [0] the kernels' sqrt(|x - y|), [1] sqrtf, [2] __fsqrt_rn (diagnostics), [3] x - (x - y). */
int azd_debug_probe_math(int device, const float *in, float *out, int n);
/* c21 cost kernel in isolation (ordered_edge.rs:72-124): lambda_1 (full = 1: full f64 bracket as
 * ArgminData reports it; 0: the f32-exact early stop used for node costs) and the matching size of
 * `count` trees, one wavefront each, repeated `reps` times; *ms = GPU time of the timed launch. */
int azd_debug_probe_cost(int device, const uint8_t *parents, int n, int count, int reps, int full,
                         double *lambda_1, int *matching_size, float *ms);

/* The evaluator's bf16 forward GEMM in isolation (timing, tests): Y[M, N] = act(A[M, K] . W[N, K]^T + bias) on device
 * pointers; A and W are bf16 rows of pitch Kp (a multiple of 64, zero beyond K), Y f32 or bf16 with pitch ldy;
 * *ms = mean GPU time of `reps` launches. */
int azd_debug_gemm_bf16(int device, int M, int N, int Kp, const void *d_a, const void *d_w, const float *d_bias, void *d_y, int ldy,
                        int out_bf16, int act, int reps, float *ms);

#ifdef __cplusplus
}
#endif
#endif /* AZDOPT_AMD_H */
