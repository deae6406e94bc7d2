/*
 * azd_oracle.cpp -- CPU ORACLE (test infrastructure; see azd_oracle.h).
 *
 * A literal restatement of the reference algorithm with the same container
 * semantics the reference gets from its third-party crates:
 *   - petgraph 0.6.4 `Graph::add_edge` pushes the new edge at the HEAD of both
 *     endpoints' adjacency lists  => every adjacency walk is newest-edge-first.
 *   - `Iterator::min_by` keeps the FIRST of equal minima, `max_by` the LAST of
 *     equal maxima.
 *   - `BTreeMap<ActionSet, NodeIndex>` orders keys lexicographically over their
 *     ascending elements (= std::map<std::set<uint32_t>>).
 * Every function cites the reference file:line it follows (paths relative to
 * /root/reference).  No reference source text is copied.
 */
#include "azd_oracle.h"

#include <algorithm>
#include <cassert>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <set>
#include <utility>
#include <vector>

#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

constexpr int MAXN = 32;
constexpr int MAXC = 4;   /* colours of the Ramsey space */
constexpr int SPACE_C21 = 0, SPACE_RAMSEY = 1, SPACE_DENSE = 2;
constexpr uint32_t NONE = 0xFFFFFFFFu;

/* ------------------------------------------------------------------ */
/* Seeded generator (build-defined; the reference uses thread_rng)     */
/* ------------------------------------------------------------------ */
inline uint64_t splitmix(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
inline uint64_t key4(uint64_t a, uint64_t b, uint64_t c, uint64_t d) {
    return splitmix(splitmix(splitmix(splitmix(a) ^ b) ^ c) ^ d);
}
/* uniform in [0, n) from the high 32 bits (multiply-shift) */
inline uint32_t below(uint64_t r, uint32_t n) { return (uint32_t)(((r >> 32) * (uint64_t)n) >> 32); }

constexpr uint64_t DOMAIN_ROOT = 0x726f6f74ull;   /* "root" */
constexpr uint64_t DOMAIN_PRED = 0x70726564ull;   /* "pred" */
constexpr uint64_t DOMAIN_RESET = 0x72657365ull;  /* "rese" */

/* ------------------------------------------------------------------ */
/* Edge / action indexing                                              */
/* ------------------------------------------------------------------ */
/* graph-state/src/simple_graph/edge.rs:48-53 */
inline int colex_position(int mx, int mn) {
    int last_pos = mx * (mx + 1) / 2;
    int diff = mx - mn;
    return last_pos - diff;
}
/* edge.rs:55-65 */
inline void from_colex_position(int pos, int *mx, int *mn) {
    int v = 1;
    for (;;) {
        int last_position = v * (v + 1) / 2;
        if (pos < last_position) {
            int diff = last_position - pos;
            *mx = v;
            *mn = v - diff;
            return;
        }
        v += 1;
    }
}
/* rooted_tree/ordered_edge.rs:35-38  index_ignoring_edge_0_1 */
inline int action_index(int parent, int child) {
    int mx = std::max(parent, child), mn = std::min(parent, child);
    return colex_position(mx, mn) - 1;
}
/* ordered_edge.rs:40-42 */
inline void action_from_index(int index, int *parent, int *child) {
    int mx, mn;
    from_colex_position(index + 1, &mx, &mn);
    *parent = mn;
    *child = mx;
}

inline int state_dim(int n) { return (n - 1) * (n - 2) - 2; }      /* space.rs:46 */
inline int action_dim(int n) { return (n - 1) * (n - 2) / 2 - 1; } /* space.rs:48 */
inline int key_words(int n) { return (action_dim(n) + 63) / 64; }

/* ------------------------------------------------------------------ */
/* c21 state: ROTWithActionPermissions<N> (modify_parent_once.rs:8-12) */
/* ------------------------------------------------------------------ */
/* The Ramsey state RamseyCountsNoRecolor<N, E, C, B> (ramsey_counts/no_recolor.rs:9-13 around
 * ramsey_counts/mod.rs:12-17) lives in the same struct: `nbr` = graphs[c].neighborhoods[v],
 * `counts` = counts[c][e] (row-major C x E), `total` = total_counts, `permitted` = permitted_edges. */
struct State {
    uint8_t parents[MAXN];
    std::set<uint32_t> permitted; /* BTreeSet<usize> */
    uint32_t nbr[MAXC][MAXN];
    uint64_t adj[64]; /* dense-graph space (dense_graph.inc): ConnectedBitsetGraph<N, B64>::neighborhoods; `permitted` = remaining slots */
    std::vector<int32_t> counts;
    int32_t total[MAXC];
    /* Layered<L, Space> (az-discrete-opt/src/space/layered.rs, nabla/space/mod.rs:41-111): the state is
     * Layers<State, L> (state/layers.rs), a ring of the last L states; everything above is back(), `older`
     * holds the states before it, oldest first (at most L - 1).  Empty for a plain space. */
    std::vector<State> older;
};

struct Cost { /* Conjecture2Dot1Cost, connected_bitset_graph/mod.rs:340-344; TotalCounts<C>, ramsey_counts/mod.rs:192-193 */
    double lambda1 = 0.0;
    std::vector<std::pair<int, int>> matching;
    int mu = 0; /* dense-graph space: the matching number (the matching itself is not constructed) */
    int32_t totals[MAXC] = {0, 0, 0, 0};
};

/* dense symmetric eigen-solve, cyclic Jacobi, f64: stands in for
 * faer 0.15 `selfadjoint_eigenvalues` at ordered_edge.rs:74-78 */
double lambda1_jacobi(const uint8_t *parents, int n) {
    std::vector<double> a((size_t)n * n, 0.0);
    /* adjacency_matrix, ordered_edge.rs:84-91 (all vertices 1..N-1) */
    for (int i = 1; i < n; ++i) {
        a[(size_t)i * n + parents[i]] = 1.0;
        a[(size_t)parents[i] * n + i] = 1.0;
    }
    for (int sweep = 0; sweep < 100; ++sweep) {
        double off = 0.0;
        for (int p = 0; p < n; ++p)
            for (int q = p + 1; q < n; ++q) off += a[(size_t)p * n + q] * a[(size_t)p * n + q];
        if (off < 1e-300) break;
        for (int p = 0; p < n; ++p)
            for (int q = p + 1; q < n; ++q) {
                double apq = a[(size_t)p * n + q];
                if (apq == 0.0) continue;
                double app = a[(size_t)p * n + p], aqq = a[(size_t)q * n + q];
                double theta = (aqq - app) / (2.0 * apq);
                double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < n; ++k) {
                    double akp = a[(size_t)k * n + p], akq = a[(size_t)k * n + q];
                    a[(size_t)k * n + p] = c * akp - s * akq;
                    a[(size_t)k * n + q] = s * akp + c * akq;
                }
                for (int k = 0; k < n; ++k) {
                    double apk = a[(size_t)p * n + k], aqk = a[(size_t)q * n + k];
                    a[(size_t)p * n + k] = c * apk - s * aqk;
                    a[(size_t)q * n + k] = s * apk + c * aqk;
                }
            }
    }
    double best = a[0];
    for (int i = 1; i < n; ++i) best = std::max(best, a[(size_t)i * n + i]);
    return best;
}

/* Cost contract for lambda_1 (DESIGN.md "lambda_1").  parents[v] < v, so the subtree T_v hangs
 * below v.  With phi_v = characteristic polynomial of T_v and psi_v = prod_{children c} phi_c
 * (that of T_v - v):   phi_v = x psi_v - sum_c psi_c prod_{c' != c} phi_c'.
 * xI - A is positive definite  <=>  every phi_v(x) > 0  <=>  x > lambda_1 (the pivots of the
 * leaf-first LDL^T are phi_v / psi_v).  Division-free: each vertex folds itself into its parent's
 * running pair (P, Q) = (prod phi_c, sum_c psi_c prod_{c' != c} phi_c'):
 *     phi = x P[v] - Q[v];  Q[p] = Q[p] phi + P[p] P[v];  P[p] = P[p] phi      (v = N-1 .. 1)
 * every operation a single IEEE f64 multiply / add / subtract (no fused ops).  Ten rounds of
 * 64-way multisection of [1, N] leave hi - lo ~ 1 ulp; lambda_1 := hi, the smallest tested x
 * that is positive definite. */
inline bool posdef_at(const uint8_t *parents, int n, double x, double *phi0_out = nullptr) {
    double P[MAXN], Q[MAXN];
    for (int v = 0; v < n; ++v) {
        P[v] = 1.0;
        Q[v] = 0.0;
    }
    bool ok = true;
    for (int v = n - 1; v >= 1; --v) {
        double xp = x * P[v];
        double phi = xp - Q[v];
        if (!(phi > 0.0)) ok = false;
        int p = parents[v];
        double qphi = Q[p] * phi;
        double ppsi = P[p] * P[v];
        Q[p] = qphi + ppsi;
        P[p] = P[p] * phi;
    }
    double xp0 = x * P[0];
    double phi0 = xp0 - Q[0];
    if (!(phi0 > 0.0)) ok = false;
    if (phi0_out) *phi0_out = phi0; /* the tree's characteristic polynomial at x (round 4: the secant window below) */
    return ok;
}
/* 32 trial points per round (33-section of [lo, hi]), at most 12 rounds (33^12 > 2^60).
 * node_mode: stop as soon as both ends of the bracket round to the same f32 -- the f32 value the
 * evaluation uses (04-c21-tree.rs:100 `*lambda_1 as f32`) is then already the one the full-precision
 * bracket would give, so tree topology cannot depend on the early stop.  Returns hi. */
/* Initial bracket (round 2; [1, N] before): among the trees on n vertices the path has the smallest lambda_1,
 * 2 cos(pi / (n + 1)), and the star the largest, sqrt(n - 1); both rounded to f32 and widened by 2^-20, so that every
 * restatement and the device start from the same doubles whatever their libm's last bit.  Saves most node costs the sixth
 * round (2.35 / 33^5 is below an f32 ulp, 17 / 33^5 is not). */
void lambda1_bracket(int n, double *lo, double *hi) {
    *lo = (double)(float)(2.0 * std::cos(3.14159265358979323846 / (double)(n + 1))) - 0x1p-20;
    *hi = (double)(float)std::sqrt((double)(n - 1)) + 0x1p-20;
}
/* Round 4: once both ends of the bracket carry a value of the tree's characteristic polynomial phi_0 with the signs of a simple
 * crossing (phi_0(lo) < 0 < phi_0(hi)), the 32 trial points of a round go into a window around the secant's estimate of the root
 * -- half-width 8 (span / 2)^2, at least span / 1024 -- clipped to the bracket, instead of across the whole bracket.  The ends are
 * only ever replaced by trial points, so the bracket stays a bracket; 5.25 -> 3.6 rounds per node cost.
 * Round 5: a window may MISS the root (all 32 points on one side of it: first == 0 or first == 32).  A near miss is harmless -- the
 * end it moves lands beside the root and the next secant is good: half of all random trees have one and still finish in 3.6 rounds
 * -- but with lambda_2 close to lambda_1 (the double brooms) |phi'' / phi'| is large, the estimate stays on one side and every
 * round misses again: the bracket then creeps instead of shrinking.  The THIRD miss of a solve therefore switches the window off
 * for the rest of it: the plain 33-section of [lo, hi] from there.  Every round that is not a miss shrinks the bracket 33-fold at
 * least and there are at most three misses, so 15 rounds give what 12 plain rounds give (33^12 > 2^60).
 * The device runs exactly these operations (space_c21.inc:lambda1_impl), one IEEE f64 operation each.
 * windowed = false: the plain 33-section in every round (what rounds 1-3 ran; kept as the yardstick of the tests). */
constexpr int LAMBDA1_MAX_MISSES = 3;
double lambda1_sturm(const uint8_t *parents, int n, bool node_mode, bool windowed = true, int *rounds_out = nullptr) {
    double lo, hi;
    lambda1_bracket(n, &lo, &hi);
    double flo = 0.0, fhi = 0.0;
    bool have_lo = false, have_hi = false;
    int misses = windowed ? 0 : LAMBDA1_MAX_MISSES;
    int round = 0;
    for (; round < 12 + LAMBDA1_MAX_MISSES; ++round) {
        if (node_mode && (float)lo == (float)hi) break;
        double wlo = lo, whi = hi;
        bool in_window = false;
        if (misses < LAMBDA1_MAX_MISSES && have_lo && have_hi && flo < 0.0 && fhi > 0.0) {
            double span = hi - lo;
            double den = flo - fhi;
            double tt = flo / den;
            double st = span * tt;
            double c = lo + st;
            double half = span * 0.5;
            double hh = half * half;
            double d = 8.0 * hh;
            double dmin = span * 0.0009765625;
            if (d < dmin) d = dmin;
            if (d < half) {
                double wa = c - d, wb = c + d;
                if (wa > lo) wlo = wa;
                if (wb < hi) whi = wb;
                in_window = true;
            }
        }
        double ws = whi - wlo;
        double w = ws / 33.0;
        int first = 32;
        double xs[32], ph[32];
        for (int j = 0; j < 32; ++j) {
            double step = w * (double)(j + 1);
            xs[j] = wlo + step;
        }
        for (int j = 0; j < 32; ++j)
            if (posdef_at(parents, n, xs[j], &ph[j])) {
                first = j;
                break;
            }
        if (first > 0) {
            lo = xs[first - 1];
            flo = ph[first - 1];
            have_lo = true;
        }
        if (first < 32) {
            hi = xs[first];
            fhi = ph[first];
            have_hi = true;
        }
        if (in_window && (first == 0 || first == 32)) ++misses;
    }
    if (rounds_out) *rounds_out = round;
    return hi;
}

/* ordered_edge.rs:94-124 maximum_matching ("unoptimized" leaf stripping) */
void maximum_matching(const uint8_t *parents, int n, std::vector<std::pair<int, int>> &m) {
    m.clear();
    bool available[MAXN];
    for (int i = 0; i < n; ++i) available[i] = true;
    for (;;) {
        bool next_leaf[MAXN];
        for (int i = 0; i < n; ++i) next_leaf[i] = available[i];
        for (int i = 1; i < n; ++i)
            if (available[i]) next_leaf[parents[i]] = false;
        for (int i = 1; i < n; ++i) {
            if (next_leaf[i]) {
                available[i] = false;
                int parent = parents[i];
                if (available[parent]) {
                    available[parent] = false;
                    m.push_back({parent, i});
                }
            }
        }
        int num_available = 0;
        for (int i = 0; i < n; ++i) num_available += available[i] ? 1 : 0;
        if (num_available < 2) break;
    }
}

/* 04-c21-tree.rs:58-74 (C_LOWER_BOUND, C_UPPER_BOUND, squish) and :98-102 */
float c21_eval(int n, double lambda1, int matching_size) {
    int isq = 0;
    while ((isq + 1) * (isq + 1) <= n - 1) ++isq;
    int sq = (isq * isq == n - 1) ? isq : isq + 1;
    int mu_max = (n + 1) / 2;
    int c_upper = sq + mu_max, c_lower = 2;
    const float slope = 1.0f / (float)(c_upper - c_lower);
    float c = (float)matching_size + (float)lambda1;
    float x = c - (float)c_lower;
    return slope * x;
}

#include "dense_graph.inc"

/* ------------------------------------------------------------------ */
/* Ramsey space primitives                                             */
/* ------------------------------------------------------------------ */
inline int popc(uint32_t x) { return __builtin_popcount(x); }
/* simple_graph/bitset_graph/mod.rs:164-185 count_cliques_inside */
int count_cliques_inside(const uint32_t *nbr, uint32_t common, int size) {
    if (size == 0) return 1;
    if (size == 1) return popc(common);
    int sum = 0;
    for (uint32_t rest = common; rest; rest &= rest - 1) {
        int u = __builtin_ctz(rest);
        uint32_t n_u = common & nbr[u] & ((1u << u) - 1u); /* range_to(u) */
        if (1 + popc(n_u) >= size) sum += count_cliques_inside(nbr, n_u, size - 1);
    }
    return sum;
}
/* ColoredCompleteBitsetGraph::color, bitset_graph/mod.rs:45-56 */
inline int edge_color(const State &s, int C, int u, int v) {
    for (int c = 0; c < C; ++c)
        if ((s.nbr[c][u] >> v) & 1u) return c;
    return -1;
}
/* RamseyCounts::new, ramsey_counts/mod.rs:20-68 */
void ramsey_counts_new(State &s, int n, int C, int E, const int *sizes) {
    s.counts.assign((size_t)C * E, 0);
    for (int c = 0; c < C; ++c) {
        int total = 0, pos = 0;
        for (int v = 0; v < n; ++v)
            for (int u = 0; u < v; ++u, ++pos) {
                uint32_t common = s.nbr[c][v] & s.nbr[c][u];
                int cnt = count_cliques_inside(s.nbr[c], common, sizes[c] - 2);
                if ((s.nbr[c][v] >> u) & 1u) total += cnt;
                s.counts[(size_t)c * E + pos] = cnt;
            }
        s.total[c] = total / (sizes[c] * (sizes[c] - 1) / 2);
    }
}
/* reassign_color_count_adjustment, ramsey_counts/mod.rs:101-164 */
void ramsey_adjust(State &s, int E, bool subtract, int u, int v, int color, int size) {
    if (size <= 2) return;
    int32_t *counts = &s.counts[(size_t)color * E];
    const uint32_t *nbr = s.nbr[color];
    uint32_t n_u = nbr[u], n_v = nbr[v], n_uv = n_u & n_v;
    for (uint32_t r = n_u; r; r &= r - 1) { /* edge {v, w}, w in n_u */
        int w = __builtin_ctz(r);
        int change = count_cliques_inside(nbr, n_uv & nbr[w], size - 3);
        if (change != 0) counts[colex_position(std::max(v, w), std::min(v, w))] += subtract ? -change : change;
    }
    for (uint32_t r = n_v; r; r &= r - 1) { /* edge {u, w}, w in n_v */
        int w = __builtin_ctz(r);
        int change = count_cliques_inside(nbr, n_uv & nbr[w], size - 3);
        if (change != 0) counts[colex_position(std::max(u, w), std::min(u, w))] += subtract ? -change : change;
    }
    if (size == 3) return;
    for (uint32_t r = n_uv; r; r &= r - 1) { /* edge {w, x}, w < x both in n_uv (tuple_combinations) */
        int w = __builtin_ctz(r);
        for (uint32_t q = r & (r - 1); q; q &= q - 1) {
            int x = __builtin_ctz(q);
            int change = count_cliques_inside(nbr, n_uv & nbr[w] & nbr[x], size - 4);
            if (change != 0) counts[colex_position(x, w)] += subtract ? -change : change;
        }
    }
}
/* reassign_color, ramsey_counts/mod.rs:78-99 */
void ramsey_reassign_color(State &s, int C, int E, const int *sizes, int edge_pos, int new_color) {
    int v, u;
    from_colex_position(edge_pos, &v, &u);
    int old_color = edge_color(s, C, u, v);
    s.nbr[old_color][v] ^= 1u << u;
    s.nbr[old_color][u] ^= 1u << v;
    ramsey_adjust(s, E, true, u, v, old_color, sizes[old_color]);
    ramsey_adjust(s, E, false, u, v, new_color, sizes[new_color]);
    s.nbr[new_color][v] ^= 1u << u;
    s.nbr[new_color][u] ^= 1u << v;
    s.total[old_color] -= s.counts[(size_t)old_color * E + edge_pos];
    s.total[new_color] += s.counts[(size_t)new_color * E + edge_pos];
}

/* One runtime-tagged space: kind 0 = ROTModifyParentsOnce<N, Conjecture2Dot1Cost>
 * (rooted_tree/space.rs:14-125), kind 1 = RamseySpaceNoEdgeRecolor<B32, N, E, C>
 * (ramsey_counts/space.rs:10-176). */
struct Space {
    int n, A, S, KW;
    int kind = SPACE_C21;
    int layers = 1, S_inner = 0; /* Layered<L, _>: STATE_DIM = L * inner STATE_DIM (nabla/space/mod.rs:53) */
    int C = 0, E = 0, root_bytes = 0;
    uint32_t p24 = 3355443; /* dense-graph space: edge probability of a fresh root, x 2^24 (0.2) */
    int sizes[MAXC] = {0, 0, 0, 0};
    float weights[MAXC] = {0, 0, 0, 0};
    /* space.rs:56-73 act; ramsey_counts/space.rs:71-86 */
    void act(State &s, int index) const {
        if (layers > 1) { /* Layered::act = push_op(clone of back, acted on) (nabla/space/mod.rs:65-71) */
            State prev = s;
            prev.older.clear();
            s.older.push_back(std::move(prev));
            if ((int)s.older.size() > layers - 1) s.older.erase(s.older.begin()); /* the ring drops its oldest */
        }
        if (kind == SPACE_DENSE) { /* AddOrDeleteEdge::from_action_index (action.rs:20-27); add_or_remove_edge_unchecked */
            int slot = index % E, mx, mn;
            from_colex_position(slot, &mx, &mn);
            s.adj[mx] ^= 1ull << mn;
            s.adj[mn] ^= 1ull << mx;
            s.permitted.erase((uint32_t)slot); /* every slot at most once */
            return;
        }
        if (kind == SPACE_RAMSEY) {
            int edge_pos = index % E, new_color = index / E; /* `action`, :48-54 */
            ramsey_reassign_color(s, C, E, sizes, edge_pos, new_color);
            s.permitted.erase((uint32_t)edge_pos);
            return;
        }
        int parent, child;
        action_from_index(index, &parent, &child);
        s.parents[child] = (uint8_t)parent; /* set_parent, ordered_edge.rs:46-50 */
        for (int u = 0; u < child; ++u) s.permitted.erase((uint32_t)action_index(u, child));
    }
    /* rooted_tree/mod.rs:60-72 edge_indices_ignoring_0_1_and_last_vertex */
    void current_edge_positions(const State &s, std::vector<uint32_t> &out) const {
        out.clear();
        for (int child = 2; child < n - 1; ++child) out.push_back((uint32_t)action_index(s.parents[child], child));
    }
    /* space.rs:75-89 action_data: permitted ids (ascending) that are not current edges.
     * Ramsey (ramsey_counts/space.rs:88-120): edges ascending, then new colours ascending;
     * a_id = e_pos + new_color * E; `r` receives the reward hint folded to the f32
     * r_sa = old_count * w[old] - new_count * w[new] of g_theta_star_sa (:163-165). */
    void action_data(const State &s, std::vector<uint32_t> &out, std::vector<float> *r = nullptr) const {
        if (kind == SPACE_DENSE) { /* action_kinds (mod.rs:139-158) over the remaining slots, ascending action id */
            out.clear();
            if (r) r->clear();
            for (uint32_t slot : s.permitted) { /* Add(e) = e */
                int mx, mn;
                from_colex_position((int)slot, &mx, &mn);
                if (!dense_has_edge(s.adj, mx, mn)) out.push_back(slot);
            }
            for (uint32_t slot : s.permitted) { /* Delete(e) = E + e, unless e is a cut edge */
                int mx, mn;
                from_colex_position((int)slot, &mx, &mn);
                if (dense_has_edge(s.adj, mx, mn) && !dense_is_cut_edge(s.adj, mx, mn)) out.push_back((uint32_t)E + slot);
            }
            if (r) r->assign(out.size(), 0.f);
            return;
        }
        if (kind == SPACE_RAMSEY) {
            out.clear();
            if (r) r->clear();
            int pos = 0;
            for (int v = 0; v < n; ++v)
                for (int u = 0; u < v; ++u, ++pos) {
                    if (!s.permitted.count((uint32_t)pos)) continue;
                    int old_color = edge_color(s, C, v, u);
                    for (int nc = 0; nc < C; ++nc) {
                        if (nc == old_color) continue;
                        out.push_back((uint32_t)(pos + nc * E));
                        if (r) {
                            int old_count = s.counts[(size_t)old_color * E + pos], new_count = s.counts[(size_t)nc * E + pos];
                            r->push_back((float)old_count * weights[old_color] - (float)new_count * weights[nc]);
                        }
                    }
                }
            return;
        }
        if (r) r->clear();
        std::vector<uint32_t> cur;
        current_edge_positions(s, cur);
        out.clear();
        for (uint32_t a : s.permitted)
            if (std::find(cur.begin(), cur.end(), a) == cur.end()) out.push_back(a);
    }
    bool is_terminal(const State &s) const { /* nabla/space/mod.rs:23-25 */
        std::vector<uint32_t> d;
        action_data(s, d);
        return d.empty();
    }
    /* space.rs:91-101 */
    /* Layered::write_vec (nabla/space/mod.rs:80-96): buffer().iter() zipped with chunks_exact_mut: the states
     * the ring holds, oldest first, each into its own chunk; chunks beyond the ring's length are NOT written
     * (they keep whatever an earlier call left there) */
    void write_vec(const State &s, float *v) const {
        if (layers > 1) {
            size_t j = 0;
            for (; j < s.older.size(); ++j) write_vec_inner(s.older[j], v + j * (size_t)S_inner);
            write_vec_inner(s, v + j * (size_t)S_inner);
            return;
        }
        write_vec_inner(s, v);
    }
    void write_vec_inner(const State &s, float *v) const {
        const int S = S_inner;
        for (int i = 0; i < S; ++i) v[i] = 0.f;
        if (kind == SPACE_DENSE) { /* dense_graph.inc header: edge bools, modifiable absent, modifiable present, remaining / E */
            int pos = 0;
            for (int x = 0; x < n; ++x)
                for (int u = 0; u < x; ++u, ++pos) v[pos] = dense_has_edge(s.adj, x, u) ? 1.0f : 0.f;
            for (uint32_t slot : s.permitted) {
                int mx, mn;
                from_colex_position((int)slot, &mx, &mn);
                v[(dense_has_edge(s.adj, mx, mn) ? 2 * E : E) + (int)slot] = 1.f;
            }
            v[3 * E] = (float)s.permitted.size() / (float)E;
            return;
        }
        if (kind == SPACE_RAMSEY) { /* ramsey_counts/space.rs:122-153 */
            for (int i = 0; i < C * E; ++i) v[i] = (float)s.counts[i];
            for (int c = 0; c < C; ++c) {
                int pos = 0;
                for (int x = 0; x < n; ++x)
                    for (int u = 0; u < x; ++u, ++pos) v[C * E + c * E + pos] = ((s.nbr[c][x] >> u) & 1u) ? 1.0f : 0.f;
            }
            for (uint32_t e : s.permitted) v[2 * C * E + e] = 1.f;
            return;
        }
        std::vector<uint32_t> cur;
        current_edge_positions(s, cur);
        for (uint32_t e : cur) v[e] = 1.f;
        for (uint32_t a : s.permitted) v[A + a] = 1.f;
    }
    /* space.rs:103-105 + ordered_edge.rs:72-82 */
    /* full = false: node costs during the search (f32-exact early stop);
     * full = true: the ArgminData cost reported to the user (full f64 bracket) */
    Cost cost(const State &s, bool full = false) const {
        Cost c;
        if (kind == SPACE_DENSE) { /* conjecture_2_1_cost, mod.rs:319-338 */
            c.lambda1 = dense_lambda1(s.adj, n);
            c.mu = dense_matching_exact(s.adj, n);
            return c;
        }
        if (kind == SPACE_RAMSEY) { /* ramsey_counts/space.rs:155-157 */
            for (int i = 0; i < C; ++i) c.totals[i] = s.total[i];
            return c;
        }
        c.lambda1 = lambda1_sturm(s.parents, n, !full);
        maximum_matching(s.parents, n, c.matching);
        return c;
    }
    float evaluate(const Cost &c) const {
        if (kind == SPACE_RAMSEY) { /* ramsey_counts/space.rs:159-165: f32 sum in colour order from 0 */
            float sum = 0.0f;
            for (int i = 0; i < C; ++i) sum = sum + (float)c.totals[i] * weights[i];
            return sum;
        }
        if (kind == SPACE_DENSE) return c21_eval(n, c.lambda1, c.mu); /* the same squish with N's bounds */
        return c21_eval(n, c.lambda1, (int)c.matching.size());
    }
    /* 04-c21-tree.rs:103; ramsey_counts/space.rs:167-172 */
    float g_theta_star_sa(float c_s, float r_sa, float h_theta_sa) const {
        if (kind == SPACE_RAMSEY) return c_s * h_theta_sa + r_sa * (1.0f - h_theta_sa);
        return c_s - h_theta_sa;
    }
    /* 04-c21-tree.rs:104; ramsey_counts/space.rs:174-177 */
    float h_sa(float /*c_s*/, float c_as, float c_as_star) const {
        if (kind == SPACE_RAMSEY) return 1.0f - c_as_star / c_as;
        return c_as_star;
    }
};

/* ------------------------------------------------------------------ */
/* SearchTree<ActionSet>                                               */
/* ------------------------------------------------------------------ */
struct StateWeight { /* tree/state_weight.rs:4-21 */
    float c, c_t_star;
    uint32_t n_t = 0, exhausted_children = 0;
    uint32_t act_start = 0, act_end = 0;
    bool is_active() const { return act_start + exhausted_children < act_end; } /* :31-33 */
};
struct GNode { /* petgraph Node: weight + heads of the out / in adjacency lists */
    StateWeight w;
    uint32_t next[2] = {NONE, NONE};
};
struct GEdge { /* petgraph Edge: weight (prediction_pos) + next-out / next-in + endpoints */
    uint32_t prediction_pos;
    uint32_t next[2];
    uint32_t node[2];
};
struct Prediction { /* tree/arc_weight.rs:11-16 */
    uint32_t a_id;
    float g_theta_sa;
    uint32_t edge_id = NONE;
};

/* ActionPath encodings (az-discrete-opt/src/path/{set,multiset,sequence,ord_set}.rs).  All derive Ord on their container, and the
 * containers compare lexicographically over what they iterate, so one ordered vector serves:
 *   ActionSet (set.rs:6-37, BTreeSet<usize>)            sorted unique elements
 *   ActionMultiset (multiset.rs:5-42, BTreeMap<a,count>) for ActionsNeverRepeat spaces every count is 1:
 *                                                        same identity, same order, same len as the set
 *   ActionSequence (sequence.rs:3-36, Vec<usize>) and OrderedActionSet (ord_set.rs:3-34, also a Vec
 *   that push_unchecked appends to)                      actions in the order taken */
constexpr int PATH_SET = 0, PATH_SEQUENCE = 1;
struct Path {
    std::vector<uint32_t> v;
    bool operator<(const Path &o) const { return v < o.v; }
    size_t size() const { return v.size(); }
    bool empty() const { return v.empty(); }
    void clear() { v.clear(); }
    void push(uint32_t a, int kind) { /* push_unchecked */
        if (kind == PATH_SEQUENCE) v.push_back(a);
        else v.insert(std::lower_bound(v.begin(), v.end(), a), a);
    }
    std::vector<uint32_t>::const_iterator begin() const { return v.begin(); } /* actions_taken */
    std::vector<uint32_t>::const_iterator end() const { return v.end(); }
};

struct Counters {
    uint64_t v[ORC_CTR_COUNT] = {0};
};

struct Tree { /* tree/mod.rs:28-32 */
    std::map<Path, uint32_t> positions;
    std::vector<GNode> nodes;
    std::vector<GEdge> edges;
    std::vector<Prediction> predictions;

    void clear() { /* :45-49 */
        positions.clear();
        nodes.clear();
        edges.clear();
        predictions.clear();
    }
    /* graph_operations.rs:8-16 */
    uint32_t add_node(const Path &p, float c) {
        GNode nd;
        nd.w.c = c;
        nd.w.c_t_star = c;
        uint32_t index = (uint32_t)nodes.size();
        nodes.push_back(nd);
        positions.emplace(p, index);
        return index;
    }
    /* graph_operations.rs:18-30; petgraph add_edge head insertion */
    uint32_t add_arc(uint32_t parent, uint32_t child, uint32_t prediction_pos) {
        GEdge e;
        e.prediction_pos = prediction_pos;
        e.node[0] = parent;
        e.node[1] = child;
        e.next[0] = nodes[parent].next[0];
        e.next[1] = nodes[child].next[1];
        uint32_t idx = (uint32_t)edges.size();
        nodes[parent].next[0] = idx;
        nodes[child].next[1] = idx;
        edges.push_back(e);
        predictions[prediction_pos].edge_id = idx;
        return idx;
    }
    /* graph_operations.rs:32-56 */
    void add_actions(uint32_t id, const Space &space, const State &state, const float *h_theta, Counters &ctr) {
        float c = nodes[id].w.c;
        uint32_t start = (uint32_t)predictions.size();
        std::vector<uint32_t> acts;
        std::vector<float> r;
        space.action_data(state, acts, &r);
        for (size_t k = 0; k < acts.size(); ++k) {
            uint32_t a_id = acts[k];
            Prediction p;
            p.a_id = a_id;
            p.g_theta_sa = space.g_theta_star_sa(c, r.empty() ? 0.f : r[k], h_theta[a_id]);
            predictions.push_back(p);
        }
        uint32_t end = (uint32_t)predictions.size();
        nodes[id].w.act_start = start;
        nodes[id].w.act_end = end;
        ctr.v[ORC_CTR_NEW_PREDS] += end - start;
    }

    /* next_action.rs:28-53: first-min of (n_t, c_t_star) over ACTIVE children, newest edge first */
    bool revisit_choice(uint32_t pos, uint32_t *edge, uint32_t *n_t) const {
        bool have = false;
        uint32_t be = 0, bn = 0;
        float bc = 0.f;
        for (uint32_t e = nodes[pos].next[0]; e != NONE; e = edges[e].next[0]) {
            const StateWeight &cw = nodes[edges[e].node[1]].w;
            if (!cw.is_active()) continue;
            uint32_t n = cw.n_t;
            float cs = cw.c_t_star;
            if (!have) {
                have = true; be = e; bn = n; bc = cs;
            } else {
                /* replace only when strictly less: (n, cs) < (bn, bc) */
                bool less = (n < bn) || (n == bn && cs < bc);
                if (less) { be = e; bn = n; bc = cs; }
            }
        }
        if (have) { *edge = be; *n_t = bn; }
        return have;
    }
    /* next_action.rs:55-88 */
    bool max_curiosity(uint32_t pos, uint32_t *out_pp, Counters &ctr) const {
        float c_s = nodes[pos].w.c;
        std::vector<float> c_t_star_values;
        for (uint32_t e = nodes[pos].next[0]; e != NONE; e = edges[e].next[0])
            c_t_star_values.push_back(nodes[edges[e].node[1]].w.c_t_star);
        uint32_t start = nodes[pos].w.act_start, end = nodes[pos].w.act_end;
        bool have = false;
        uint32_t best = 0;
        float bestv = 0.f;
        if (c_t_star_values.empty()) {
            for (uint32_t i = start; i < end; ++i) {
                const Prediction &p = predictions[i];
                if (p.edge_id != NONE) continue;
                float v = c_s - p.g_theta_sa;
                if (!have || v < bestv) { have = true; best = i; bestv = v; } /* min_by: first min */
            }
        } else {
            for (uint32_t i = start; i < end; ++i) {
                const Prediction &p = predictions[i];
                if (p.edge_id != NONE) continue;
                float c_theta_star = c_s - p.g_theta_sa;
                float curiosity = 0.f; /* f32 sum, sequential, newest child first */
                for (float c_t_star : c_t_star_values) curiosity = curiosity + std::sqrt(std::fabs(c_t_star - c_theta_star));
                ctr.v[ORC_CTR_CURIOSITY_PAIRS] += c_t_star_values.size();
                if (!have || !(curiosity < bestv)) { have = true; best = i; bestv = curiosity; } /* max_by: last max */
            }
        }
        if (have) *out_pp = best;
        return have;
    }
    enum Choice { CH_NONE, CH_VISITED, CH_UNVISITED };
    /* next_action.rs:11-26 */
    Choice next_action(uint32_t pos, uint32_t tol, uint32_t *out, Counters &ctr) const {
        const StateWeight &w = nodes[pos].w;
        if (!w.is_active()) return CH_NONE;
        ctr.v[ORC_CTR_SELECT_CALLS] += 1;
        ctr.v[ORC_CTR_SUM_ACTIONS] += w.act_end - w.act_start;
        {
            uint64_t deg = 0;
            for (uint32_t e = nodes[pos].next[0]; e != NONE; e = edges[e].next[0]) ++deg;
            ctr.v[ORC_CTR_SUM_DEG] += deg;
        }
        uint32_t re = 0, rn = 0;
        bool have_r = revisit_choice(pos, &re, &rn);
        if (have_r && rn < tol) { *out = re; return CH_VISITED; }
        uint32_t pp = 0;
        if (max_curiosity(pos, &pp, ctr)) { *out = pp; return CH_UNVISITED; }
        if (have_r) { *out = re; return CH_VISITED; }
        return CH_NONE;
    }

    /* empty_transitions.rs:50-87 (old=false) and :89-127 (old=true) */
    void cascade(uint32_t edge_id, bool old, Counters &ctr) {
        struct Info { float c_t_star; uint32_t newly_exhausted_children; };
        const GEdge &a_t = edges[edge_id];
        const StateWeight &s_t = nodes[a_t.node[1]].w;
        uint32_t n_t_s_t = s_t.n_t;
        Info info;
        info.c_t_star = s_t.c_t_star;
        info.newly_exhausted_children = old ? (s_t.is_active() ? 0u : 1u) : 1u;
        std::map<uint32_t, Info> current_nodes, next_nodes; /* BTreeMap<NodeIndex, Info> */
        current_nodes.emplace(a_t.node[0], info);
        for (;;) {
            if (current_nodes.empty()) { /* pop_front, :21-26 */
                std::swap(current_nodes, next_nodes);
                if (current_nodes.empty()) break;
                ctr.v[ORC_CTR_MAX_FRONTIER] = std::max<uint64_t>(ctr.v[ORC_CTR_MAX_FRONTIER], current_nodes.size());
            }
            auto it = current_nodes.begin();
            uint32_t child_index = it->first;
            Info ancestor_info = it->second;
            current_nodes.erase(it);
            ctr.v[ORC_CTR_CASCADE_NODES] += 1;
            StateWeight &child = nodes[child_index].w;
            child.exhausted_children += ancestor_info.newly_exhausted_children;
            if (child.c_t_star > ancestor_info.c_t_star) child.c_t_star = ancestor_info.c_t_star;
            else child.n_t += 1;
            if (old) child.n_t = std::max(child.n_t, n_t_s_t); /* :110 */
            Info new_child_info;
            new_child_info.c_t_star = ancestor_info.c_t_star;
            new_child_info.newly_exhausted_children = child.is_active() ? 0u : 1u;
            for (uint32_t e = nodes[child_index].next[1]; e != NONE; e = edges[e].next[1]) {
                uint32_t parent_id = edges[e].node[0];
                auto f = next_nodes.find(parent_id);
                if (f == next_nodes.end()) next_nodes.emplace(parent_id, new_child_info);
                else {
                    f->second.c_t_star = std::fmin(f->second.c_t_star, new_child_info.c_t_star);
                    f->second.newly_exhausted_children += new_child_info.newly_exhausted_children;
                }
            }
        }
    }

    /* tree/mod.rs:113-232.  Returns false on the reference's unreachable!() */
    bool roll_out_episodes(const Space &space, const State &root, State &state, Cost &cost, Path &path,
                           uint32_t &state_pos, const uint32_t *tol, int ntol, uint32_t tol_default, Counters &ctr,
                           int path_kind = PATH_SET) {
        for (;;) {
            size_t len = path.size();
            uint32_t t = len < (size_t)ntol ? tol[len] : tol_default;
            uint32_t sel = 0;
            Choice ch = next_action(state_pos, t, &sel, ctr);
            if (ch == CH_VISITED) { /* :139-151 */
                uint32_t prediction_pos = edges[sel].prediction_pos;
                uint32_t action_id = predictions[prediction_pos].a_id;
                path.push(action_id, path_kind);
                space.act(state, (int)action_id);
                state_pos = edges[sel].node[1];
                ctr.v[ORC_CTR_VISITED_STEPS] += 1;
            } else if (ch == CH_UNVISITED) { /* :160-218 */
                uint32_t prediction_pos = sel;
                uint32_t action_id = predictions[prediction_pos].a_id;
                path.push(action_id, path_kind);
                ctr.v[ORC_CTR_MAX_DEPTH] = std::max<uint64_t>(ctr.v[ORC_CTR_MAX_DEPTH], path.size());
                auto f = positions.find(path);
                if (f != positions.end()) { /* transposition, :172-179 */
                    uint32_t arc = add_arc(state_pos, f->second, prediction_pos);
                    cascade(arc, true, ctr);
                    state = root;
                    path.clear();
                    state_pos = 0;
                    ctr.v[ORC_CTR_TRANSPOSITIONS] += 1;
                } else { /* :180-216 */
                    space.act(state, (int)action_id);
                    cost = space.cost(state);
                    float c_as = space.evaluate(cost);
                    uint32_t next_pos = add_node(path, c_as);
                    uint32_t arc = add_arc(state_pos, next_pos, prediction_pos);
                    if (space.is_terminal(state)) {
                        cascade(arc, false, ctr);
                        state = root;
                        path.clear();
                        state_pos = 0;
                        ctr.v[ORC_CTR_TERMINALS] += 1;
                    } else {
                        state_pos = next_pos;
                        ctr.v[ORC_CTR_EXPANSIONS] += 1;
                        return true;
                    }
                }
            } else { /* :220-229 */
                if (path.empty()) {
                    ctr.v[ORC_CTR_ROOT_EXHAUSTED] += 1;
                    return true;
                }
                return false; /* unreachable!() in the reference */
            }
        }
    }

    /* tree/mod.rs:242-264 */
    void write_observations(const Space &space, float *observations, float *weights, uint32_t n_t_as_tol) const {
        float c_s = nodes[0].w.c;
        for (uint32_t e = nodes[0].next[0]; e != NONE; e = edges[e].next[0]) {
            const StateWeight &cw = nodes[edges[e].node[1]].w;
            if (!cw.is_active() || cw.n_t >= n_t_as_tol) {
                float h = space.h_sa(c_s, cw.c, cw.c_t_star);
                uint32_t a = predictions[edges[e].prediction_pos].a_id;
                observations[a] = h;
                weights[a] = 1.0f;
            }
        }
    }
};

} // namespace

/* ------------------------------------------------------------------ */
/* NablaOptimizer (optimizer/mod.rs:7-22)                              */
/* ------------------------------------------------------------------ */
struct orc_engine {
    Space space;
    int B, threads;
    std::vector<State> roots, states;
    std::vector<Cost> costs;
    std::vector<Path> paths;
    int path_kind = PATH_SET;
    std::vector<uint32_t> last_positions;
    std::vector<float> state_vecs;
    std::vector<Tree> trees;
    std::vector<size_t> num_inspected_nodes;
    std::vector<Counters> ctrs;
    std::vector<uint8_t> failed;
    /* ArgminData, log.rs:1-11 */
    State argmin_state;
    Cost argmin_cost;
    float argmin_eval = 0.f;
};

namespace {

/* Packed roots.  c21: parents[n] + permitted action ids (A bits).  Ramsey: colour of every edge in
 * colex order, colors[E], + permitted edge positions (E bits); both masks are KW words per agent. */
void unpack_state(const Space &sp, const uint8_t *parents, const uint64_t *permitted, State &s) {
    s.older.clear(); /* Layers::new(s): a ring of one */
    std::memset(s.parents, 0, sizeof(s.parents));
    if (sp.kind == SPACE_DENSE) { /* packed root: neighbourhoods as n u64 (8 n bytes) + modifiable slots (E bits) */
        std::memset(s.adj, 0, sizeof(s.adj));
        std::memcpy(s.adj, parents, (size_t)sp.n * 8);
        s.permitted.clear();
        for (int e = 0; e < sp.E; ++e)
            if ((permitted[e >> 6] >> (e & 63)) & 1ull) s.permitted.insert((uint32_t)e);
        return;
    }
    if (sp.kind == SPACE_RAMSEY) {
        std::memset(s.nbr, 0, sizeof(s.nbr));
        int pos = 0;
        for (int v = 0; v < sp.n; ++v)
            for (int u = 0; u < v; ++u, ++pos) {
                int c = parents[pos];
                s.nbr[c][v] |= 1u << u;
                s.nbr[c][u] |= 1u << v;
            }
        ramsey_counts_new(s, sp.n, sp.C, sp.E, sp.sizes);
        s.permitted.clear();
        for (int e = 0; e < sp.E; ++e)
            if ((permitted[e >> 6] >> (e & 63)) & 1ull) s.permitted.insert((uint32_t)e);
        return;
    }
    for (int i = 0; i < sp.n; ++i) s.parents[i] = parents[i];
    s.permitted.clear();
    for (int a = 0; a < sp.A; ++a)
        if ((permitted[a >> 6] >> (a & 63)) & 1ull) s.permitted.insert((uint32_t)a);
}
void pack_state(const Space &sp, const State &s, uint8_t *parents, uint64_t *permitted) {
    if (sp.kind == SPACE_DENSE) std::memcpy(parents, s.adj, (size_t)sp.n * 8);
    else if (sp.kind == SPACE_RAMSEY) {
        int pos = 0;
        for (int v = 0; v < sp.n; ++v)
            for (int u = 0; u < v; ++u, ++pos) parents[pos] = (uint8_t)edge_color(s, sp.C, v, u);
    } else
        for (int i = 0; i < sp.n; ++i) parents[i] = s.parents[i];
    for (int w = 0; w < sp.KW; ++w) permitted[w] = 0;
    for (uint32_t a : s.permitted) permitted[a >> 6] |= 1ull << (a & 63);
}

/* optimizer/mod.rs:194-246 */
int update_argmin(orc_engine *e) {
    const float min_eval = e->argmin_eval;
    int best_tree = -1;
    size_t best_node = 0;
    float best_c = 0.f;
    for (int t = 0; t < e->B; ++t) {
        Tree &tr = e->trees[t];
        size_t num = e->num_inspected_nodes[t];
        if (num < tr.nodes.size()) {
            bool have = false;
            size_t bi = 0;
            float bc = 0.f;
            for (size_t i = num; i < tr.nodes.size(); ++i) {
                float c = tr.nodes[i].w.c;
                if (!(c < min_eval)) continue;
                if (!have || c < bc) { have = true; bi = i; bc = c; }
            }
            e->num_inspected_nodes[t] = tr.nodes.size();
            /* rayon min_by across trees: tie order unspecified in the reference;
             * this build fixes it to the lowest tree index (first min) */
            if (have && (best_tree < 0 || bc < best_c)) { best_tree = t; best_node = bi; best_c = bc; }
        }
    }
    if (best_tree < 0) return 0;
    Tree &tr = e->trees[best_tree];
    e->argmin_state = e->roots[best_tree];
    for (auto &kv : tr.positions)
        if (kv.second == best_node) {
            for (uint32_t a : kv.first) e->space.act(e->argmin_state, (int)a);
            break;
        }
    e->argmin_cost = e->space.cost(e->argmin_state, true);
    e->argmin_eval = e->space.evaluate(e->argmin_cost);
    return 1;
}

} // namespace

extern "C" {

int orc_state_dim(int n) { return state_dim(n); }
int orc_action_dim(int n) { return action_dim(n); }
int orc_key_words(int n) { return key_words(n); }
int orc_edge_colex_position(int u, int v) { return colex_position(std::max(u, v), std::min(u, v)); }
void orc_edge_from_colex_position(int pos, int *mx, int *mn) { from_colex_position(pos, mx, mn); }
int orc_action_index(int parent, int child) { return action_index(parent, child); }
void orc_action_from_index(int index, int *parent, int *child) { action_from_index(index, parent, child); }

/* ordered_edge.rs:52-70 */
int orc_all_possible_parent_modifications(const uint8_t *parents, int n, int *out_pairs) {
    int cnt = 0;
    for (int child = 0; child < n - 1; ++child) {
        if (child == 0) continue; /* parent(0) = None, rooted_tree/mod.rs:45-51 */
        int parent = parents[child];
        for (int np = 0; np < child; ++np) {
            if (np == parent) continue;
            out_pairs[2 * cnt] = np;
            out_pairs[2 * cnt + 1] = child;
            ++cnt;
        }
    }
    return cnt;
}
double orc_lambda1_jacobi(const uint8_t *parents, int n) { return lambda1_jacobi(parents, n); }
double orc_lambda1_sturm(const uint8_t *parents, int n) { return lambda1_sturm(parents, n, false); }
double orc_lambda1_node(const uint8_t *parents, int n) { return lambda1_sturm(parents, n, true); }
double orc_lambda1_plain(const uint8_t *parents, int n, int node_mode) { return lambda1_sturm(parents, n, node_mode != 0, false); }
int orc_lambda1_rounds(const uint8_t *parents, int n, int node_mode, int windowed) {
    int r = 0;
    lambda1_sturm(parents, n, node_mode != 0, windowed != 0, &r);
    return r;
}
int orc_maximum_matching(const uint8_t *parents, int n, int *out_pairs) {
    std::vector<std::pair<int, int>> m;
    maximum_matching(parents, n, m);
    if (out_pairs)
        for (size_t i = 0; i < m.size(); ++i) {
            out_pairs[2 * i] = m[i].first;
            out_pairs[2 * i + 1] = m[i].second;
        }
    return (int)m.size();
}
float orc_c21_eval(int n, double lambda1, int matching_size) { return c21_eval(n, lambda1, matching_size); }

uint64_t orc_key4(uint64_t a, uint64_t b, uint64_t c, uint64_t d) { return key4(a, b, c, d); }

/* Root generator spec (stands in for rooted_tree/mod.rs:14-20 +
 * modify_parent_once.rs:14-25 + 04-c21-tree.rs:108-112 with thread_rng):
 *   stream(i) = key4(seed, DOMAIN_ROOT ^ (epoch << 32), agent, i)
 *   k = kmin + below(stream(0), kmax - kmin + 1)
 *   parents[v] = below(stream(v), v) for v = 2..N-2, else 0
 *   permitted = first k of a Fisher-Yates shuffle of 0..A-1,
 *               step j swaps j with j + below(stream(64 + j), A - j) */
static void gen_one_root(uint64_t seed, uint64_t domain, uint64_t agent, int n, int k, uint8_t *parents,
                         uint64_t *permitted, uint64_t draw_base) {
    int A = action_dim(n), KW = key_words(n);
    for (int v = 0; v < n; ++v) parents[v] = 0;
    for (int v = 2; v <= n - 2; ++v) parents[v] = (uint8_t)below(key4(seed, domain, agent, draw_base + (uint64_t)v), (uint32_t)v);
    std::vector<uint32_t> perm(A);
    for (int i = 0; i < A; ++i) perm[i] = (uint32_t)i;
    for (int w = 0; w < KW; ++w) permitted[w] = 0;
    for (int j = 0; j < k; ++j) {
        uint32_t r = (uint32_t)j + below(key4(seed, domain, agent, draw_base + 64 + (uint64_t)j), (uint32_t)(A - j));
        std::swap(perm[j], perm[r]);
        permitted[perm[j] >> 6] |= 1ull << (perm[j] & 63);
    }
}
static void gen_permitted_of(uint64_t seed, uint64_t domain, uint64_t agent, int A, int KW, int k, uint64_t *permitted,
                             uint64_t draw_base) {
    std::vector<uint32_t> perm(A);
    for (int i = 0; i < A; ++i) perm[i] = (uint32_t)i;
    for (int w = 0; w < KW; ++w) permitted[w] = 0;
    for (int j = 0; j < k; ++j) {
        uint32_t r = (uint32_t)j + below(key4(seed, domain, agent, draw_base + 64 + (uint64_t)j), (uint32_t)(A - j));
        std::swap(perm[j], perm[r]);
        permitted[perm[j] >> 6] |= 1ull << (perm[j] & 63);
    }
}
/* Ramsey root generator spec (stands in for ColoredCompleteBitsetGraph::generate with uniform
 * colour weights, bitset_graph/mod.rs:21-38, + RamseyCountsNoRecolor::generate, no_recolor.rs:38-54):
 *   colour[e] = below(stream(1024 + e), C) for every edge position e;
 *   permitted edges = first k of the Fisher-Yates shuffle of 0..E-1 (draws 64 + j). */
static void gen_ramsey_root(uint64_t seed, uint64_t domain, uint64_t agent, int E, int C, int KW, int k, uint8_t *colors,
                            uint64_t *permitted) {
    for (int e = 0; e < E; ++e) colors[e] = (uint8_t)below(key4(seed, domain, agent, 1024 + (uint64_t)e), (uint32_t)C);
    gen_permitted_of(seed, domain, agent, E, KW, k, permitted, 0);
}
void orc_gen_ramsey_roots(uint64_t seed, uint64_t epoch, uint64_t first_agent, int count, int n, int n_colors, int kmin,
                          int kmax, uint8_t *colors, uint64_t *permitted) {
    int E = n * (n - 1) / 2, KW = (E * n_colors + 63) / 64;
    uint64_t domain = DOMAIN_ROOT ^ (epoch << 32);
    for (int i = 0; i < count; ++i) {
        uint64_t agent = first_agent + (uint64_t)i;
        int k = kmin + (int)below(key4(seed, domain, agent, 0), (uint32_t)(kmax - kmin + 1));
        gen_ramsey_root(seed, domain, agent, E, n_colors, KW, k, colors + (size_t)i * E, permitted + (size_t)i * KW);
    }
}
/* Root generator spec (stands in for ConnectedBitsetGraph::generate(p), mod.rs:84-97, which redraws G(n, p) until it is
 * connected, + a permitted-slot set in the image of modify_parent_once.rs:14-25), seeded:
 *   stream(i) = key4(seed, DOMAIN_ROOT ^ (epoch << 32), agent, i)
 *   k = kmin + below(stream(0), kmax - kmin + 1)
 *   attempt t = 0, 1, ...: edge at slot e present iff (stream(4096 + t E + e) >> 40) < p24 (p24 = p * 2^24); first connected
 *   permitted slots = first k of the Fisher-Yates shuffle of 0..E-1 (draws 64 + j, as for the other spaces) */
static void gen_dense_root(uint64_t seed, uint64_t domain, uint64_t agent, int n, int k, uint32_t p24, uint64_t *adj, uint64_t *slots) {
    const int E = n * (n - 1) / 2, PW = (E + 63) / 64;
    for (uint64_t t = 0;; ++t) {
        for (int v = 0; v < n; ++v) adj[v] = 0;
        int e = 0;
        for (int v = 1; v < n; ++v)
            for (int u = 0; u < v; ++u, ++e)
                if ((uint32_t)(key4(seed, domain, agent, 4096ull + t * (uint64_t)E + (uint64_t)e) >> 40) < p24) {
                    adj[v] |= 1ull << u;
                    adj[u] |= 1ull << v;
                }
        if (dense_is_connected(adj, n)) break;
    }
    gen_permitted_of(seed, domain, agent, E, PW, k, slots, 0);
}
void orc_gen_dense_roots(uint64_t seed, uint64_t epoch, uint64_t first_agent, int count, int n, int kmin, int kmax, uint32_t p24,
                         uint64_t *adj, uint64_t *slots) {
    const int E = n * (n - 1) / 2, KW = (2 * E + 63) / 64;
    uint64_t domain = DOMAIN_ROOT ^ (epoch << 32);
    for (int i = 0; i < count; ++i) {
        uint64_t agent = first_agent + (uint64_t)i;
        int k = kmin + (int)below(key4(seed, domain, agent, 0), (uint32_t)(kmax - kmin + 1));
        uint64_t *so = slots + (size_t)i * KW;
        for (int w = 0; w < KW; ++w) so[w] = 0;
        gen_dense_root(seed, domain, agent, n, k, p24, adj + (size_t)i * n, so);
    }
}
/* the dense-graph primitives, exposed for the golden-vector tests */
int orc_dense_matching_tutte(const uint64_t *adj, int n) { return dense_matching_tutte(adj, n); }
int orc_dense_matching_exact(const uint64_t *adj, int n) { return dense_matching_exact(adj, n); }
int orc_dense_matching_reference(const uint64_t *adj, int n) { return dense_matching_reference(adj, n); }
int orc_dense_is_cut_edge(const uint64_t *adj, int v, int u) { return dense_is_cut_edge(adj, v, u) ? 1 : 0; }
double orc_dense_lambda1(const uint64_t *adj, int n) { return dense_lambda1(adj, n); }
void orc_gen_roots(uint64_t seed, uint64_t epoch, uint64_t first_agent, int count, int n, int kmin, int kmax,
                   uint8_t *parents, uint64_t *permitted) {
    int KW = key_words(n);
    uint64_t domain = DOMAIN_ROOT ^ (epoch << 32);
    for (int i = 0; i < count; ++i) {
        uint64_t agent = first_agent + (uint64_t)i;
        int k = kmin + (int)below(key4(seed, domain, agent, 0), (uint32_t)(kmax - kmin + 1));
        gen_one_root(seed, domain, agent, n, k, parents + (size_t)i * n, permitted + (size_t)i * KW, 0);
    }
}
/* h(agent, call, a) = top 24 bits of key4(seed ^ DOMAIN_PRED, agent, call, a) * 2^-24 in [0,1) */
void orc_hash_predictions(uint64_t seed, uint64_t first_agent, int count, int action_dim_, uint64_t call,
                          float *out) {
    for (int i = 0; i < count; ++i)
        for (int a = 0; a < action_dim_; ++a) {
            uint64_t r = key4(seed ^ DOMAIN_PRED, first_agent + (uint64_t)i, call, (uint64_t)a);
            out[(size_t)i * action_dim_ + a] = (float)(r >> 40) * (1.0f / 16777216.0f);
        }
}

static void engine_init(orc_engine *e, int batch, int threads);
orc_engine *orc_create(int n, int batch, int threads) {
    if (n < 4 || n > MAXN) return nullptr;
    orc_engine *e = new orc_engine();
    e->space.n = n;
    e->space.A = action_dim(n);
    e->space.S = state_dim(n);
    e->space.KW = key_words(n);
    e->space.root_bytes = n;
    engine_init(e, batch, threads);
    return e;
}
/* NablaOptimizer over the build-defined dense-graph space (dense_graph.inc) */
orc_engine *orc_create_dense(int n, int batch, int threads) {
    if (n < 4 || n > 64) return nullptr;
    orc_engine *e = new orc_engine();
    Space &sp = e->space;
    sp.kind = SPACE_DENSE;
    sp.n = n;
    sp.E = n * (n - 1) / 2;
    sp.A = 2 * sp.E;     /* AddOrDeleteEdge, action.rs:10-27 */
    sp.S = 3 * sp.E + 1; /* E + ACTION + 1, 05-ah.rs:39-40 */
    sp.KW = (sp.A + 63) / 64;
    sp.root_bytes = 8 * n;
    engine_init(e, batch, threads);
    return e;
}
void orc_set_dense_p(orc_engine *e, uint32_t p24) { e->space.p24 = p24; }
/* NablaOptimizer<RamseySpaceNoEdgeRecolor<B32, N, E, C>, M, ActionSet> (01-r333.rs:35-46, 02-r44.rs:35-47) */
orc_engine *orc_create_ramsey(int n, int n_colors, const int *sizes, const float *weights, int batch, int threads) {
    if (n < 2 || n > MAXN || n_colors < 2 || n_colors > MAXC) return nullptr;
    for (int c = 0; c < n_colors; ++c)
        if (sizes[c] < 2) return nullptr;
    orc_engine *e = new orc_engine();
    Space &sp = e->space;
    sp.kind = SPACE_RAMSEY;
    sp.n = n;
    sp.C = n_colors;
    sp.E = n * (n - 1) / 2;
    sp.A = sp.E * sp.C;                /* ramsey_counts/space.rs:42 */
    sp.S = sp.E * (2 * sp.C + 1);      /* :40 */
    sp.KW = (sp.A + 63) / 64;
    sp.root_bytes = sp.E;
    for (int c = 0; c < n_colors; ++c) {
        sp.sizes[c] = sizes[c];
        sp.weights[c] = weights[c];
    }
    engine_init(e, batch, threads);
    return e;
}
static void engine_init(orc_engine *e, int batch, int threads) {
    e->space.S_inner = e->space.S;
    e->B = batch;
    e->threads = threads < 1 ? 1 : threads;
    e->roots.resize(batch);
    e->states.resize(batch);
    e->costs.resize(batch);
    e->paths.resize(batch);
    e->last_positions.assign(batch, 0);
    e->state_vecs.assign((size_t)batch * e->space.S, 0.f);
    e->trees.resize(batch);
    e->num_inspected_nodes.assign(batch, 0);
    e->ctrs.resize(batch);
    e->failed.assign(batch, 0);
}
void orc_destroy(orc_engine *e) { delete e; }
/* Layered<L, Space>: call before orc_new_begin.  state_vecs keep their contents across calls, as the
 * optimizer's buffer does (chunks beyond the ring's length are never rewritten). */
void orc_set_layers(orc_engine *e, int layers) {
    e->space.layers = layers < 1 ? 1 : layers;
    e->space.S = e->space.S_inner * e->space.layers;
    e->state_vecs.assign((size_t)e->B * e->space.S, 0.f);
}
/* 0 = ActionSet / ActionMultiset, 1 = ActionSequence / OrderedActionSet; call before orc_new_begin */
void orc_set_path_kind(orc_engine *e, int kind) { e->path_kind = kind; }
const float *orc_state_vecs(orc_engine *e) { return e->state_vecs.data(); }

/* optimizer/mod.rs:61-70 */
void orc_new_begin(orc_engine *e, const uint8_t *parents, const uint64_t *permitted) {
    const Space &sp = e->space;
#pragma omp parallel for num_threads(e->threads) schedule(dynamic, 16)
    for (int i = 0; i < e->B; ++i) {
        unpack_state(sp, parents + (size_t)i * sp.root_bytes, permitted + (size_t)i * sp.KW, e->roots[i]);
        e->states[i] = e->roots[i];
        e->costs[i] = sp.cost(e->roots[i]);
        e->paths[i].clear();
        sp.write_vec(e->states[i], &e->state_vecs[(size_t)i * sp.S]);
        e->last_positions[i] = 0;
        e->num_inspected_nodes[i] = 0;
    }
}
/* optimizer/mod.rs:74-101 */
void orc_new_end(orc_engine *e, const float *h) {
    const Space &sp = e->space;
#pragma omp parallel for num_threads(e->threads) schedule(dynamic, 16)
    for (int i = 0; i < e->B; ++i) {
        Tree &t = e->trees[i];
        t.clear();
        float c = sp.evaluate(e->costs[i]);
        uint32_t root_id = t.add_node(Path(), c);
        t.add_actions(root_id, sp, e->states[i], h + (size_t)i * sp.A, e->ctrs[i]);
    }
    int best = 0;
    float be = sp.evaluate(e->costs[0]);
    for (int i = 1; i < e->B; ++i) {
        float ev = sp.evaluate(e->costs[i]);
        if (ev < be) { be = ev; best = i; }
    }
    e->argmin_state = e->states[best];
    e->argmin_cost = sp.cost(e->states[best], true);
    e->argmin_eval = be;
}

/* optimizer/mod.rs:159-174 */
void orc_rollout_begin(orc_engine *e, const uint32_t *tol, int ntol, uint32_t tol_default) {
    const Space &sp = e->space;
#pragma omp parallel for num_threads(e->threads) schedule(dynamic, 4)
    for (int i = 0; i < e->B; ++i) {
        bool ok = e->trees[i].roll_out_episodes(sp, e->roots[i], e->states[i], e->costs[i], e->paths[i],
                                                e->last_positions[i], tol, ntol, tol_default, e->ctrs[i], e->path_kind);
        if (!ok) e->failed[i] = 1;
        if (!e->paths[i].empty()) sp.write_vec(e->states[i], &e->state_vecs[(size_t)i * sp.S]);
    }
}
/* optimizer/mod.rs:177-190 */
int orc_rollout_end(orc_engine *e, const float *h) {
    const Space &sp = e->space;
#pragma omp parallel for num_threads(e->threads) schedule(dynamic, 16)
    for (int i = 0; i < e->B; ++i) {
        if (!e->paths[i].empty())
            e->trees[i].add_actions(e->last_positions[i], sp, e->states[i], h + (size_t)i * sp.A, e->ctrs[i]);
    }
    return update_argmin(e);
}

/* optimizer/mod.rs:262-278 */
void orc_observe(orc_engine *e, uint32_t n_obs_tol, float *obs, float *weights) {
    const Space &sp = e->space;
#pragma omp parallel for num_threads(e->threads) schedule(dynamic, 16)
    for (int i = 0; i < e->B; ++i) {
        sp.write_vec(e->roots[i], &e->state_vecs[(size_t)i * sp.S]);
        float *o = obs + (size_t)i * sp.A, *w = weights + (size_t)i * sp.A;
        for (int a = 0; a < sp.A; ++a) { o[a] = 0.f; w[a] = 0.f; }
        e->trees[i].write_observations(sp, o, w, n_obs_tol);
    }
}

/* optimizer/mod.rs:317-346 (modify_root already applied by the caller) */
void orc_reset_begin(orc_engine *e, const uint8_t *parents, const uint64_t *permitted) {
    const Space &sp = e->space;
#pragma omp parallel for num_threads(e->threads) schedule(dynamic, 16)
    for (int i = 0; i < e->B; ++i) {
        e->last_positions[i] = 0;
        unpack_state(sp, parents + (size_t)i * sp.root_bytes, permitted + (size_t)i * sp.KW, e->roots[i]);
        e->states[i] = e->roots[i];
        e->costs[i] = sp.cost(e->roots[i]);
        sp.write_vec(e->states[i], &e->state_vecs[(size_t)i * sp.S]);
        e->paths[i].clear();
    }
}
/* optimizer/mod.rs:350-359 */
void orc_reset_end(orc_engine *e, const float *h) {
    const Space &sp = e->space;
#pragma omp parallel for num_threads(e->threads) schedule(dynamic, 16)
    for (int i = 0; i < e->B; ++i) {
        Tree &t = e->trees[i];
        t.clear();
        float c = sp.evaluate(e->costs[i]);
        uint32_t root_id = t.add_node(Path(), c);
        t.add_actions(root_id, sp, e->roots[i], h + (size_t)i * sp.A, e->ctrs[i]);
        e->num_inspected_nodes[i] = 0;
    }
}

/* 04-c21-tree.rs:172-206 and 02-r44.rs:196-228 (the same policy over either space) with the seeded generator:
 *   stream(i) = key4(seed, DOMAIN_RESET ^ (epoch << 32), agent, i)
 *   draw 0: node choice; draw 1: new permitted count; draws 2.. / 64..: as the
 *   root generator (fresh tree) or the permitted shuffle. */
void orc_c21_modify_roots(orc_engine *e, uint64_t seed, uint64_t epoch, uint64_t first_agent, int kmin, int kmax,
                          uint8_t *parents_out, uint64_t *permitted_out) {
    const Space &sp = e->space;
    uint64_t domain = DOMAIN_RESET ^ (epoch << 32);
    for (int i = 0; i < e->B; ++i) {
        uint64_t agent = first_agent + (uint64_t)i;
        State state = e->roots[i];
        const Tree &t = e->trees[i];
        /* node_data(): BTreeMap order; n[0] is the root (empty set sorts first) */
        std::vector<std::pair<const Path *, const StateWeight *>> n;
        for (auto &kv : t.positions) n.push_back({&kv.first, &t.nodes[kv.second].w});
        float c_root = n[0].second->c, c_root_star = n[0].second->c_t_star;
        uint8_t *po = parents_out + (size_t)i * sp.root_bytes;
        std::vector<uint64_t> scratch(sp.KW);
        uint64_t *mo = permitted_out + (size_t)i * sp.KW;
        uint64_t r0 = key4(seed, domain, agent, 0), r1 = key4(seed, domain, agent, 1);
        if (c_root == c_root_star) {
            int num_permitted = (int)state.permitted.size();
            if (num_permitted == kmax) {
                int k = kmin + (int)below(r1, (uint32_t)(kmax - kmin + 1));
                if (sp.kind == SPACE_RAMSEY) gen_ramsey_root(seed, domain, agent, sp.E, sp.C, sp.KW, k, po, mo);
                else if (sp.kind == SPACE_DENSE) { /* a fresh connected G(n, p) + k of its E slots */
                    for (int w = 0; w < sp.KW; ++w) mo[w] = 0;
                    gen_dense_root(seed, domain, agent, sp.n, k, sp.p24, reinterpret_cast<uint64_t *>(po), mo);
                } else gen_one_root(seed, domain, agent, sp.n, k, po, mo, 0);
                continue;
            }
            std::vector<const Path *> keep;
            for (auto &pw : n)
                if (pw.second->c == c_root) keep.push_back(pw.first);
            const Path *p = keep[below(r0, (uint32_t)keep.size())];
            for (uint32_t a : *p) sp.act(state, (int)a);
            int k = num_permitted + (int)below(r1, (uint32_t)(kmax - num_permitted + 1));
            pack_state(sp, state, po, scratch.data());
            gen_permitted_of(seed, domain, agent, sp.kind == SPACE_C21 ? sp.A : sp.E, sp.KW, k, mo, 0); /* dense: k of the E edge slots */
        } else {
            float c_threshold = (c_root + 3.0f * c_root_star) / 4.0f;
            std::vector<const Path *> keep;
            for (auto &pw : n)
                if (pw.second->c <= c_threshold) keep.push_back(pw.first);
            const Path *p = keep[below(r0, (uint32_t)keep.size())];
            for (uint32_t a : *p) sp.act(state, (int)a);
            int k = kmin + (int)below(r1, (uint32_t)(kmax - kmin + 1));
            pack_state(sp, state, po, scratch.data());
            gen_permitted_of(seed, domain, agent, sp.kind == SPACE_C21 ? sp.A : sp.E, sp.KW, k, mo, 0);
        }
    }
}

void orc_counters(orc_engine *e, uint64_t *out) {
    for (int k = 0; k < ORC_CTR_COUNT; ++k) out[k] = 0;
    for (int i = 0; i < e->B; ++i)
        for (int k = 0; k < ORC_CTR_COUNT; ++k) {
            if (k == ORC_CTR_MAX_FRONTIER || k == ORC_CTR_MAX_DEPTH) out[k] = std::max(out[k], e->ctrs[i].v[k]);
            else out[k] += e->ctrs[i].v[k];
        }
    uint64_t failed = 0;
    for (int i = 0; i < e->B; ++i) failed += e->failed[i];
    out[ORC_CTR_COUNT - 1] = failed;
}

void orc_argmin(orc_engine *e, uint8_t *parents, uint64_t *permitted, double *lambda1, int *matching_size,
                float *eval) {
    pack_state(e->space, e->argmin_state, parents, permitted);
    *lambda1 = e->argmin_cost.lambda1;
    *matching_size = e->space.kind == SPACE_DENSE ? e->argmin_cost.mu : (int)e->argmin_cost.matching.size();
    *eval = e->argmin_eval;
}

void orc_argmin_totals(orc_engine *e, int32_t *totals) {
    for (int c = 0; c < MAXC; ++c) totals[c] = e->argmin_cost.totals[c];
}
void orc_agent_totals(orc_engine *e, int agent, int32_t *totals) {
    for (int c = 0; c < MAXC; ++c) totals[c] = e->costs[agent].totals[c];
}
/* the agent's live per-edge counts [C][E] (test access to RamseyCounts::counts) */
void orc_agent_counts(orc_engine *e, int agent, int32_t *counts) {
    const State &s = e->states[agent];
    for (size_t i = 0; i < s.counts.size(); ++i) counts[i] = s.counts[i];
}
int orc_engine_state_dim(orc_engine *e) { return e->space.S; }
int orc_engine_action_dim(orc_engine *e) { return e->space.A; }
int orc_engine_key_words(orc_engine *e) { return e->space.KW; }
int orc_engine_root_bytes(orc_engine *e) { return e->space.root_bytes; }

/* ---- Ramsey space functions, exposed for the golden-vector tests ---- */
/* RamseyCounts::new (ramsey_counts/mod.rs:20-68) on colours given per colex edge position */
void orc_ramsey_counts_new(int n, int n_colors, const int *sizes, const uint8_t *colors, int32_t *counts, int32_t *totals) {
    Space sp;
    sp.kind = SPACE_RAMSEY; sp.n = n; sp.C = n_colors; sp.E = n * (n - 1) / 2; sp.A = sp.E * sp.C; sp.KW = (sp.A + 63) / 64;
    for (int c = 0; c < n_colors; ++c) sp.sizes[c] = sizes[c];
    State s;
    std::vector<uint64_t> none(sp.KW, 0);
    unpack_state(sp, colors, none.data(), s);
    for (size_t i = 0; i < s.counts.size(); ++i) counts[i] = s.counts[i];
    for (int c = 0; c < n_colors; ++c) totals[c] = s.total[c];
}
/* reassign_color (ramsey_counts/mod.rs:78-99) applied in sequence to `n_actions` action ids; returns
 * the incrementally maintained counts/totals and the final colours */
void orc_ramsey_act_sequence(int n, int n_colors, const int *sizes, uint8_t *colors, const int *actions, int n_actions,
                             int32_t *counts, int32_t *totals) {
    Space sp;
    sp.kind = SPACE_RAMSEY; sp.n = n; sp.C = n_colors; sp.E = n * (n - 1) / 2; sp.A = sp.E * sp.C; sp.KW = (sp.A + 63) / 64;
    for (int c = 0; c < n_colors; ++c) sp.sizes[c] = sizes[c];
    State s;
    std::vector<uint64_t> all(sp.KW, ~0ull);
    unpack_state(sp, colors, all.data(), s);
    for (int i = 0; i < n_actions; ++i) sp.act(s, actions[i]);
    for (size_t i = 0; i < s.counts.size(); ++i) counts[i] = s.counts[i];
    for (int c = 0; c < n_colors; ++c) totals[c] = s.total[c];
    std::vector<uint64_t> scratch(sp.KW);
    pack_state(sp, s, colors, scratch.data());
}

void orc_tree_sizes(orc_engine *e, int agent, int *n_nodes, int *n_edges, int *n_preds) {
    const Tree &t = e->trees[agent];
    *n_nodes = (int)t.nodes.size();
    *n_edges = (int)t.edges.size();
    *n_preds = (int)t.predictions.size();
}

void orc_export_tree(orc_engine *e, int agent, float *c, float *c_star, uint32_t *n_t, uint32_t *exhausted,
                     uint32_t *act_begin, uint32_t *act_end, uint64_t *keys, uint32_t *e_src, uint32_t *e_dst,
                     uint32_t *e_pp, uint32_t *p_aid, float *p_g, int32_t *p_edge) {
    const Tree &t = e->trees[agent];
    int KW = e->space.KW;
    for (size_t i = 0; i < t.nodes.size(); ++i) {
        const StateWeight &w = t.nodes[i].w;
        c[i] = w.c; c_star[i] = w.c_t_star; n_t[i] = w.n_t; exhausted[i] = w.exhausted_children;
        act_begin[i] = w.act_start; act_end[i] = w.act_end;
        for (int k = 0; k < KW; ++k) keys[i * KW + k] = 0;
    }
    for (auto &kv : t.positions)
        for (uint32_t a : kv.first) keys[(size_t)kv.second * KW + (a >> 6)] |= 1ull << (a & 63);
    for (size_t i = 0; i < t.edges.size(); ++i) {
        e_src[i] = t.edges[i].node[0]; e_dst[i] = t.edges[i].node[1]; e_pp[i] = t.edges[i].prediction_pos;
    }
    for (size_t i = 0; i < t.predictions.size(); ++i) {
        p_aid[i] = t.predictions[i].a_id; p_g[i] = t.predictions[i].g_theta_sa;
        p_edge[i] = t.predictions[i].edge_id == NONE ? -1 : (int32_t)t.predictions[i].edge_id;
    }
}

void orc_agent_state(orc_engine *e, int agent, uint8_t *parents, uint64_t *permitted, uint64_t *path,
                     uint32_t *state_pos, double *lambda1, int *matching_size) {
    pack_state(e->space, e->states[agent], parents, permitted);
    for (int w = 0; w < e->space.KW; ++w) path[w] = 0;
    for (uint32_t a : e->paths[agent]) path[a >> 6] |= 1ull << (a & 63);
    *state_pos = e->last_positions[agent];
    *lambda1 = e->costs[agent].lambda1;
    *matching_size = e->space.kind == SPACE_DENSE ? e->costs[agent].mu : (int)e->costs[agent].matching.size();
}

} // extern "C"

/* ------------------------------------------------------------------ */
/* Evaluator: ActionModel (model/dfdx.rs) as plain fp32 MLP + Adam.    */
/* dfdx 0.13 is absent from /root/reference: restated from its         */
/* published behaviour (Linear init U(-1/sqrt(in), 1/sqrt(in)) for W   */
/* and b; y = x W^T + b; Adam with WeightDecay::L2 = g += wd*p before  */
/* the moments, bias-corrected).  PARITY UNPINNED; tolerance-checked   */
/* against torch CPU in tests/.                                        */
/* ------------------------------------------------------------------ */
struct orc_mlp {
    int L;
    std::vector<int> dims;
    int final_act;
    float lr, b1, b2, eps, l2;
    int threads;
    int t = 0;
    std::vector<std::vector<float>> W, b, mW, vW, mb, vb;
};

extern "C" {

orc_mlp *orc_mlp_create(int n_layers, const int *dims, int final_act, float lr, float beta1, float beta2,
                        float eps, float l2, uint64_t seed, int threads) {
    orc_mlp *m = new orc_mlp();
    m->L = n_layers;
    m->dims.assign(dims, dims + n_layers + 1);
    m->final_act = final_act;
    m->lr = lr; m->b1 = beta1; m->b2 = beta2; m->eps = eps; m->l2 = l2;
    m->threads = threads < 1 ? 1 : threads;
    m->W.resize(n_layers); m->b.resize(n_layers);
    m->mW.resize(n_layers); m->vW.resize(n_layers); m->mb.resize(n_layers); m->vb.resize(n_layers);
    for (int l = 0; l < n_layers; ++l) {
        int in = dims[l], out = dims[l + 1];
        float bound = 1.0f / std::sqrt((float)in);
        m->W[l].resize((size_t)in * out);
        m->b[l].resize(out);
        for (size_t i = 0; i < m->W[l].size(); ++i) {
            uint64_t r = key4(seed, 0x6d6c7057ull, (uint64_t)l, (uint64_t)i);
            float u = (float)(r >> 40) * (1.0f / 16777216.0f);
            m->W[l][i] = (2.0f * u - 1.0f) * bound;
        }
        for (int i = 0; i < out; ++i) {
            uint64_t r = key4(seed, 0x6d6c7062ull, (uint64_t)l, (uint64_t)i);
            float u = (float)(r >> 40) * (1.0f / 16777216.0f);
            m->b[l][i] = (2.0f * u - 1.0f) * bound;
        }
        m->mW[l].assign(m->W[l].size(), 0.f); m->vW[l].assign(m->W[l].size(), 0.f);
        m->mb[l].assign(out, 0.f); m->vb[l].assign(out, 0.f);
    }
    return m;
}
void orc_mlp_destroy(orc_mlp *m) { delete m; }
int64_t orc_mlp_num_params(orc_mlp *m) {
    int64_t n = 0;
    for (int l = 0; l < m->L; ++l) n += (int64_t)m->W[l].size() + (int64_t)m->b[l].size();
    return n;
}
void orc_mlp_get_params(orc_mlp *m, float *out) {
    for (int l = 0; l < m->L; ++l) {
        std::memcpy(out, m->W[l].data(), m->W[l].size() * 4); out += m->W[l].size();
        std::memcpy(out, m->b[l].data(), m->b[l].size() * 4); out += m->b[l].size();
    }
}
void orc_mlp_set_params(orc_mlp *m, const float *in) {
    for (int l = 0; l < m->L; ++l) {
        std::memcpy(m->W[l].data(), in, m->W[l].size() * 4); in += m->W[l].size();
        std::memcpy(m->b[l].data(), in, m->b[l].size() * 4); in += m->b[l].size();
    }
}

static void layer_forward(const orc_mlp *m, int l, int batch, const float *x, float *y) {
    int in = m->dims[l], out = m->dims[l + 1];
    int act = (l == m->L - 1) ? m->final_act : 1;
    const float *W = m->W[l].data(), *bb = m->b[l].data();
#pragma omp parallel for num_threads(m->threads) schedule(static)
    for (int r = 0; r < batch; ++r) {
        const float *xr = x + (size_t)r * in;
        float *yr = y + (size_t)r * out;
        for (int o = 0; o < out; ++o) {
            const float *w = W + (size_t)o * in;
            float s = 0.f;
            for (int i = 0; i < in; ++i) s += xr[i] * w[i];
            s += bb[o];
            if (act == 1) s = s > 0.f ? s : 0.f;
            else if (act == 2) s = 1.0f / (1.0f + std::exp(-s));
            yr[o] = s;
        }
    }
}

/* The same layer, blocked and vectorised for the CPU-baseline timing (bench.py cpu_baseline): 16 rows at a
 * time, transposed so that one 8-float vector holds one input feature of 8 rows, 4 output neurons per pass.
 * Every output element is still the sum over i = 0..in-1 in order of separately rounded products plus the
 * bias (no FMA, no reassociation), so the result equals layer_forward's bit for bit (tests/test_oracle_mlp_fast.py). */
typedef float v8f __attribute__((vector_size(32)));
typedef float v8fu __attribute__((vector_size(32), aligned(4)));
__attribute__((target("avx2"))) static void layer_forward_blocked(const orc_mlp *m, int l, int batch, const float *x, float *y) {
    const int in = m->dims[l], out = m->dims[l + 1];
    const int act = (l == m->L - 1) ? m->final_act : 1;
    const float *W = m->W[l].data(), *bb = m->b[l].data();
    const int n_blocks = (batch + 15) / 16;
#pragma omp parallel num_threads(m->threads)
    {
        std::vector<float> xT((size_t)in * 16);
#pragma omp for schedule(static)
        for (int blk = 0; blk < n_blocks; ++blk) {
            const int rb = blk * 16, nr = batch - rb < 16 ? batch - rb : 16;
            for (int r = 0; r < 16; ++r)
                for (int i = 0; i < in; ++i) xT[(size_t)i * 16 + r] = r < nr ? x[(size_t)(rb + r) * in + i] : 0.f;
            for (int o = 0; o < out; o += 4) {
                const int no = out - o < 4 ? out - o : 4;
                const float *w0 = W + (size_t)o * in, *w1 = W + (size_t)(o + (no > 1 ? 1 : 0)) * in;
                const float *w2 = W + (size_t)(o + (no > 2 ? 2 : 0)) * in, *w3 = W + (size_t)(o + (no > 3 ? 3 : 0)) * in;
                v8f a00 = {0}, a01 = {0}, a10 = {0}, a11 = {0}, a20 = {0}, a21 = {0}, a30 = {0}, a31 = {0};
                for (int i = 0; i < in; ++i) {
                    const v8f x0 = *reinterpret_cast<const v8fu *>(&xT[(size_t)i * 16]);
                    const v8f x1 = *reinterpret_cast<const v8fu *>(&xT[(size_t)i * 16 + 8]);
                    const float s0 = w0[i], s1 = w1[i], s2 = w2[i], s3 = w3[i];
                    a00 += x0 * s0; a01 += x1 * s0;
                    a10 += x0 * s1; a11 += x1 * s1;
                    a20 += x0 * s2; a21 += x1 * s2;
                    a30 += x0 * s3; a31 += x1 * s3;
                }
                const v8f acc[4][2] = {{a00, a01}, {a10, a11}, {a20, a21}, {a30, a31}};
                for (int q = 0; q < no; ++q)
                    for (int r = 0; r < nr; ++r) {
                        float s = acc[q][r >> 3][r & 7] + bb[o + q];
                        if (act == 1) s = s > 0.f ? s : 0.f;
                        else if (act == 2) s = 1.0f / (1.0f + std::exp(-s));
                        y[(size_t)(rb + r) * out + o + q] = s;
                    }
            }
        }
    }
}

/* model/dfdx.rs:69-84, blocked (see layer_forward_blocked); falls back to orc_mlp_forward without AVX2 */
int orc_mlp_forward_fast(orc_mlp *m, int batch, const float *states, float *preds) {
    if (!__builtin_cpu_supports("avx2")) {
        orc_mlp_forward(m, batch, states, preds);
        return 0;
    }
    std::vector<float> a, bbuf;
    const float *x = states;
    for (int l = 0; l < m->L; ++l) {
        float *y;
        if (l == m->L - 1) y = preds;
        else {
            std::vector<float> &dst = (l & 1) ? bbuf : a;
            dst.resize((size_t)batch * m->dims[l + 1]);
            y = dst.data();
        }
        layer_forward_blocked(m, l, batch, x, y);
        x = y;
    }
    return 1;
}

/* model/dfdx.rs:69-84 */
void orc_mlp_forward(orc_mlp *m, int batch, const float *states, float *preds) {
    std::vector<float> a, bbuf;
    const float *x = states;
    for (int l = 0; l < m->L; ++l) {
        float *y;
        if (l == m->L - 1) y = preds;
        else {
            std::vector<float> &dst = (l & 1) ? bbuf : a;
            dst.resize((size_t)batch * m->dims[l + 1]);
            y = dst.data();
        }
        layer_forward(m, l, batch, x, y);
        x = y;
    }
}

/* model/dfdx.rs:86-131 */
float orc_mlp_update(orc_mlp *m, int batch, const float *states, const float *obs, const float *weights) {
    int L = m->L;
    std::vector<std::vector<float>> acts(L + 1);
    acts[0].assign(states, states + (size_t)batch * m->dims[0]);
    for (int l = 0; l < L; ++l) {
        acts[l + 1].resize((size_t)batch * m->dims[l + 1]);
        layer_forward(m, l, batch, acts[l].data(), acts[l + 1].data());
    }
    int A = m->dims[L];
    float weight_sum = 0.f; /* :106 sequential f32 sum */
    for (size_t i = 0; i < (size_t)batch * A; ++i) weight_sum += weights[i];
    double loss = 0.0;
    std::vector<float> delta((size_t)batch * A);
    for (size_t i = 0; i < (size_t)batch * A; ++i) {
        float w = weights[i] / weight_sum; /* :110 */
        float p = acts[L][i];
        float d = p - obs[i];
        loss += (double)(w * d * d);
        float dp = 2.0f * w * d;
        if (m->final_act == 2) dp *= p * (1.0f - p);
        else if (m->final_act == 1) dp = p > 0.f ? dp : 0.f;
        delta[i] = dp;
    }
    m->t += 1;
    float bc1 = 1.0f - std::pow(m->b1, (float)m->t), bc2 = 1.0f - std::pow(m->b2, (float)m->t);
    for (int l = L - 1; l >= 0; --l) {
        int in = m->dims[l], out = m->dims[l + 1];
        const float *x = acts[l].data();
        std::vector<float> gW((size_t)in * out, 0.f), gb(out, 0.f), dx;
        if (l > 0) dx.assign((size_t)batch * in, 0.f);
#pragma omp parallel for num_threads(m->threads) schedule(static)
        for (int o = 0; o < out; ++o) {
            float *g = &gW[(size_t)o * in];
            float sb = 0.f;
            for (int r = 0; r < batch; ++r) {
                float dz = delta[(size_t)r * out + o];
                if (dz == 0.f) continue;
                sb += dz;
                const float *xr = x + (size_t)r * in;
                for (int i = 0; i < in; ++i) g[i] += dz * xr[i];
            }
            gb[o] = sb;
        }
        if (l > 0) {
            const float *W = m->W[l].data();
#pragma omp parallel for num_threads(m->threads) schedule(static)
            for (int r = 0; r < batch; ++r) {
                float *dxr = &dx[(size_t)r * in];
                for (int o = 0; o < out; ++o) {
                    float dz = delta[(size_t)r * out + o];
                    if (dz == 0.f) continue;
                    const float *w = W + (size_t)o * in;
                    for (int i = 0; i < in; ++i) dxr[i] += dz * w[i];
                }
                const float *xr = x + (size_t)r * in; /* ReLU'(z) = [x > 0] on the hidden output */
                for (int i = 0; i < in; ++i) dxr[i] = xr[i] > 0.f ? dxr[i] : 0.f;
            }
        }
        auto adam = [&](std::vector<float> &p, std::vector<float> &g, std::vector<float> &mm, std::vector<float> &vv) {
            for (size_t i = 0; i < p.size(); ++i) {
                float gi = g[i] + m->l2 * p[i];
                mm[i] = m->b1 * mm[i] + (1.0f - m->b1) * gi;
                vv[i] = m->b2 * vv[i] + (1.0f - m->b2) * gi * gi;
                float mh = mm[i] / bc1, vh = vv[i] / bc2;
                p[i] -= m->lr * mh / (std::sqrt(vh) + m->eps);
            }
        };
        adam(m->W[l], gW, m->mW[l], m->vW[l]);
        adam(m->b[l], gb, m->mb[l], m->vb[l]);
        if (l > 0) delta.swap(dx);
    }
    return (float)loss;
}

} // extern "C"
