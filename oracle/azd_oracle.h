/*
 * azd_oracle.h -- C API of the CPU ORACLE.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library.  The product
 * (azdopt_amd/, include/azdopt_amd.h) never links, imports or calls it.
 *
 * It is a CPU restatement of the reference's data-parallel tree-search path:
 *   az-discrete-opt/src/nabla/tree/{mod,next_action,empty_transitions,
 *       graph_operations,state_weight,arc_weight}.rs
 *   az-discrete-opt/src/nabla/optimizer/mod.rs
 *   az-discrete-opt/src/nabla/model/{mod,dfdx}.rs
 *   graph-state/src/rooted_tree/{mod,modify_parent_once,ordered_edge,space}.rs
 *   graph-state/src/simple_graph/edge.rs
 *   graph-state/src/ramsey_counts/{mod,no_recolor,space}.rs,
 *   graph-state/src/simple_graph/bitset_graph/mod.rs (Ramsey space; drivers 01-r333.rs, 02-r44.rs)
 *   graph-state/examples/04-c21-tree.rs (driver semantics)
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   - space layer (edge index, parent modifications, lambda_1, matching):
 *     pinned by the reference's own unit-test vectors (tests/golden/).
 *   - tree / optimizer / model layers: the reference holds NO test, fixture or
 *     golden output for them and cannot be built here (Rust nightly + CUDA
 *     dfdx, no toolchain, no network)  ==> PARITY UNPINNED by the reference;
 *     pinned instead by an independent second restatement (oracle/py_oracle.py)
 *     that must agree bit-for-bit, plus hand-worked micro-cases.
 */
#ifndef AZD_ORACLE_H
#define AZD_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_engine orc_engine;
typedef struct orc_mlp orc_mlp;

/* ---- dimensions (rooted_tree/space.rs:46-48) ---- */
int orc_state_dim(int n);
int orc_action_dim(int n);
int orc_key_words(int n);

/* ---- space-level functions, exposed for the golden-vector tests ---- */
int orc_edge_colex_position(int u, int v);                 /* edge.rs:48-53  */
void orc_edge_from_colex_position(int pos, int *mx, int *mn); /* edge.rs:55-65 */
int orc_action_index(int parent, int child);               /* ordered_edge.rs:35-38 */
void orc_action_from_index(int index, int *parent, int *child); /* :40-42 */
/* ordered_edge.rs:52-70; writes (parent, child) pairs, returns count */
int orc_all_possible_parent_modifications(const uint8_t *parents, int n, int *out_pairs);
double orc_lambda1_jacobi(const uint8_t *parents, int n);  /* dense symmetric eigen-solve (what faer does) */
double orc_lambda1_sturm(const uint8_t *parents, int n);   /* cost contract, full f64 bracket (ArgminData) */
double orc_lambda1_node(const uint8_t *parents, int n);    /* cost contract, f32-exact early stop (node costs) */
double orc_lambda1_plain(const uint8_t *parents, int n, int node_mode); /* the same bracket by plain 33-section only (yardstick) */
int orc_lambda1_rounds(const uint8_t *parents, int n, int node_mode, int windowed); /* rounds the solve took */
int orc_maximum_matching(const uint8_t *parents, int n, int *out_pairs); /* ordered_edge.rs:94-124 */
float orc_c21_eval(int n, double lambda1, int matching_size);           /* 04-c21-tree.rs:58-74,98-102 */

/* ---- seeded generators (the reference is unseeded; spec in DESIGN.md) ---- */
uint64_t orc_key4(uint64_t a, uint64_t b, uint64_t c, uint64_t d);
void orc_gen_roots(uint64_t seed, uint64_t epoch, uint64_t first_agent, int count, int n,
                   int kmin, int kmax, uint8_t *parents, uint64_t *permitted);
void orc_hash_predictions(uint64_t seed, uint64_t first_agent, int count, int action_dim,
                          uint64_t call, float *out);

/* ---- engine = NablaOptimizer<ROTModifyParentsOnce<N>, M, ActionSet> ---- */
orc_engine *orc_create(int n, int batch, int threads);
/* ---- engine = NablaOptimizer<RamseySpaceNoEdgeRecolor<B32, N, E, C>, M, ActionSet>
 * (ramsey_counts/space.rs; drivers 01-r333.rs, 02-r44.rs).  Packed roots: `parents` carries the
 * colour of every edge in colex order (E bytes per agent), `permitted` the permitted edge
 * positions (E bits in KW = ceil(E*C/64) words per agent).  The c21-named calls below
 * (orc_c21_modify_roots, orc_argmin's lambda1/matching) have Ramsey counterparts. ---- */
orc_engine *orc_create_ramsey(int n, int n_colors, const int *sizes, const float *weights, int batch, int threads);
void orc_gen_ramsey_roots(uint64_t seed, uint64_t epoch, uint64_t first_agent, int count, int n, int n_colors,
                          int kmin, int kmax, uint8_t *colors, uint64_t *permitted);
void orc_argmin_totals(orc_engine *e, int32_t *totals /* [4] */);
void orc_agent_totals(orc_engine *e, int agent, int32_t *totals /* [4] */);
void orc_agent_counts(orc_engine *e, int agent, int32_t *counts /* [C*E] */);
int orc_engine_state_dim(orc_engine *e);
int orc_engine_action_dim(orc_engine *e);
int orc_engine_key_words(orc_engine *e);
int orc_engine_root_bytes(orc_engine *e);
void orc_ramsey_counts_new(int n, int n_colors, const int *sizes, const uint8_t *colors, int32_t *counts,
                           int32_t *totals);                       /* ramsey_counts/mod.rs:20-68 */
void orc_ramsey_act_sequence(int n, int n_colors, const int *sizes, uint8_t *colors, const int *actions,
                             int n_actions, int32_t *counts, int32_t *totals); /* mod.rs:78-164 */
void orc_destroy(orc_engine *e);
/* path encoding P of NablaOptimizer<Space, M, P> (az-discrete-opt/src/path/): 0 = ActionSet (= ActionMultiset on
 * ActionsNeverRepeat spaces), 1 = ActionSequence (= OrderedActionSet).  Call before orc_new_begin. */
void orc_set_path_kind(orc_engine *e, int kind);
/* Layered<L, Space> history wrapper (az-discrete-opt/src/space/layered.rs; nabla/space/mod.rs:41-111): the
 * evaluator sees the last L states of the path.  Call before orc_new_begin; STATE_DIM becomes L * inner. */
void orc_set_layers(orc_engine *e, int layers);
/* par_new (optimizer/mod.rs:39-118) split around the model call at :72 */
void orc_new_begin(orc_engine *e, const uint8_t *parents, const uint64_t *permitted);
void orc_new_end(orc_engine *e, const float *h_theta);
/* par_roll_out_episodes (:121-191) split around the model call at :175 */
void orc_rollout_begin(orc_engine *e, const uint32_t *tol, int ntol, uint32_t tol_default);
int orc_rollout_end(orc_engine *e, const float *h_theta); /* 1 = ArgminImprovement::Improved */
/* par_update_model (:249-281) without the model call at :279 */
void orc_observe(orc_engine *e, uint32_t n_obs_tol, float *obs, float *weights);
/* par_reset_trees (:284-360) split around the model call at :348; the
 * modify_root closure is applied by the caller, who passes the new roots */
void orc_reset_begin(orc_engine *e, const uint8_t *parents, const uint64_t *permitted);
void orc_reset_end(orc_engine *e, const float *h_theta);
/* the modify_root policy of 04-c21-tree.rs:172-206 with the seeded generator */
void orc_c21_modify_roots(orc_engine *e, uint64_t seed, uint64_t epoch, uint64_t first_agent,
                          int kmin, int kmax, uint8_t *parents_out, uint64_t *permitted_out);

const float *orc_state_vecs(orc_engine *e); /* [batch * STATE_DIM] */
/* counters: see ORC_CTR_* */
enum {
    ORC_CTR_EXPANSIONS = 0, /* rollout calls that ended on a new non-terminal node */
    ORC_CTR_TERMINALS = 1,
    ORC_CTR_TRANSPOSITIONS = 2,
    ORC_CTR_VISITED_STEPS = 3,
    ORC_CTR_SELECT_CALLS = 4,
    ORC_CTR_SUM_DEG = 5,      /* sum over select calls of out-degree */
    ORC_CTR_SUM_ACTIONS = 6,  /* sum over select calls of |node.actions| */
    ORC_CTR_CASCADE_NODES = 7,
    ORC_CTR_NEW_PREDS = 8,
    ORC_CTR_ROOT_EXHAUSTED = 9,
    ORC_CTR_MAX_FRONTIER = 10,
    ORC_CTR_MAX_DEPTH = 11,
    ORC_CTR_CURIOSITY_PAIRS = 12,
    ORC_CTR_COUNT = 16
};
void orc_counters(orc_engine *e, uint64_t *out /* [ORC_CTR_COUNT] */);
void orc_argmin(orc_engine *e, uint8_t *parents, uint64_t *permitted, double *lambda1,
                int *matching_size, float *eval);
void orc_tree_sizes(orc_engine *e, int agent, int *n_nodes, int *n_edges, int *n_preds);
void orc_export_tree(orc_engine *e, int agent, float *c, float *c_star, uint32_t *n_t,
                     uint32_t *exhausted, uint32_t *act_begin, uint32_t *act_end,
                     uint64_t *keys /* [n_nodes*KW] */, uint32_t *e_src, uint32_t *e_dst,
                     uint32_t *e_pp, uint32_t *p_aid, float *p_g, int32_t *p_edge);
void orc_agent_state(orc_engine *e, int agent, uint8_t *parents, uint64_t *permitted,
                     uint64_t *path, uint32_t *state_pos, double *lambda1, int *matching_size);

/* ---- evaluator = ActionModel (model/dfdx.rs) as a plain fp32 MLP + Adam ---- */
/* dims[n_layers+1]; hidden activations ReLU; final_act: 0 none, 1 ReLU, 2 Sigmoid */
/* ---- the build-defined dense-graph space (oracle/dense_graph.inc; BASELINE configs[4]) */
orc_engine *orc_create_dense(int n, int batch, int threads);
/* edge probability (x 2^24) of the fresh roots orc_c21_modify_roots draws for the dense-graph space; default 0.2 */
void orc_set_dense_p(orc_engine *e, uint32_t p24);
void orc_gen_dense_roots(uint64_t seed, uint64_t epoch, uint64_t first_agent, int count, int n, int kmin, int kmax,
                         uint32_t p24, uint64_t *adj, uint64_t *slots);
int orc_dense_matching_tutte(const uint64_t *adj, int n);
int orc_dense_matching_exact(const uint64_t *adj, int n); /* Edmonds: the matching number the space's cost uses */     /* rank(Tutte over GF(2^31 - 1)) / 2 */
int orc_dense_matching_reference(const uint64_t *adj, int n); /* connected_bitset_graph/mod.rs:235-317, literally */
int orc_dense_is_cut_edge(const uint64_t *adj, int v, int u); /* :47-71 */
double orc_dense_lambda1(const uint64_t *adj, int n);
orc_mlp *orc_mlp_create(int n_layers, const int *dims, int final_act, float lr, float beta1,
                        float beta2, float eps, float l2, uint64_t seed, int threads);
void orc_mlp_destroy(orc_mlp *m);
int64_t orc_mlp_num_params(orc_mlp *m);
void orc_mlp_get_params(orc_mlp *m, float *out); /* per layer: W[out][in] then b[out] */
void orc_mlp_set_params(orc_mlp *m, const float *in);
void orc_mlp_forward(orc_mlp *m, int batch, const float *states, float *preds);
/* the same rows bit for bit from a blocked, AVX2-vectorised loop nest (the CPU-baseline timing uses this one);
 * returns 1 if the vector path ran, 0 if it fell back to orc_mlp_forward */
int orc_mlp_forward_fast(orc_mlp *m, int batch, const float *states, float *preds);
float orc_mlp_update(orc_mlp *m, int batch, const float *states, const float *obs,
                     const float *weights);

#ifdef __cplusplus
}
#endif
#endif
