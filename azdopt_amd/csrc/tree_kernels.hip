// tree_kernels.hip -- the data-parallel tree-search step on gfx950.
//
// One 64-lane wavefront owns one agent (= one SearchTree + its state).  Control
// flow is wave-uniform; lanes parallelise the per-node work:
//   * lanes <-> the node's action predictions (one dwordx4 PredRec per lane),
//   * lanes <-> children for the gather of their 32-B NodeRec,
//   * wave reductions (shuffle butterflies over 64 lanes) for the child argmin /
//     curiosity argmax with the reference's first-min / last-max tie rules,
//   * lanes <-> 64 probe slots of the transposition table,
//   * lanes <-> 64 trial points of the lambda_1 multisection,
//   * ballot + prefix popcount to append the legal actions of a new node.
// Built with -ffp-contract=off: every f32/f64 operation below is a single IEEE
// operation so results are bit-identical to the CPU oracle.
//
// Reference semantics (file:line relative to the reference root) are cited at
// each device function.
#include <hip/hip_runtime.h>

#include "engine_types.h"

namespace azd {

#define LANE ((int)(threadIdx.x & 63))
// Intra-wave LDS hand-off: a wave's LDS operations complete in issue order, so lanes only need
// the compiler kept from reordering across this point plus the lgkmcnt drain the fences emit.
// (No s_barrier: one wave per agent, and k_argmin's tail runs on a single wave of a larger block.)
#define WAVE_SYNC()                                              \
    do {                                                         \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   \
        __builtin_amdgcn_wave_barrier();                         \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");   \
    } while (0)

// Diagnostic build (make PROFILE=1): 100 MHz wall-clock stamps per phase, accumulated into counter
// slots 16..22.  In the product build PH_NOW() folds to 0 and the adds vanish.
#ifdef AZD_PHASE_PROFILE
#define PH_NOW() ((unsigned long long)wall_clock64())
#else
#define PH_NOW() (0ull)
#endif

// per-call counters live in the wave's LDS block `s` (lane 0 updates; values are wave-uniform)
#define CTR_ADD(K, V)                                            \
    do {                                                         \
        if (LANE == 0) s.ctr[K] += (unsigned long long)(V);      \
    } while (0)
#define CTR_MAX(K, V)                                                                        \
    do {                                                                                     \
        if (LANE == 0 && (unsigned long long)(V) > s.ctr[K]) s.ctr[K] = (unsigned long long)(V); \
    } while (0)

// ---------------------------------------------------------------- small helpers
__device__ __forceinline__ uint32_t ordf(float f) {
    // order-preserving map f32 -> u32 (after folding -0.0 into +0.0, as partial_cmp treats them equal)
    f = f + 0.0f;
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
// Correctly rounded f32 square root (the oracle's sqrtss / np.sqrt): an f64 square root of an f32
// value rounded back to f32 is exact-rounded (53 >= 2*24 + 2 bits), independent of the f32 sqrt
// expansion hipcc picks.
__device__ __forceinline__ float azd_sqrt(float x) { return (float)sqrt((double)x); }
// Wave-wide reductions on the DPP cross-lane path (row_shr 1/2/4/8, row_bcast 15/31, then one
// v_readlane): ~13 VALU ops per 32-bit reduction.  The __shfl_xor butterfly costs a dependent
// ds_bpermute (an LDS-crossbar round trip) per step, 12 of them for a 64-bit key -- measured as the
// largest single cost of a selection step.
#define AZD_DPP_STEP(OP, ID, V, CTRL, ROWM, BANKM)                                                   \
    do {                                                                                             \
        uint32_t _t = (uint32_t)__builtin_amdgcn_update_dpp((int)(ID), (int)(V), CTRL, ROWM, BANKM, false); \
        V = OP(V, _t);                                                                               \
    } while (0)
__device__ __forceinline__ uint32_t u32_min(uint32_t a, uint32_t b) { return a < b ? a : b; }
__device__ __forceinline__ uint32_t u32_max(uint32_t a, uint32_t b) { return a > b ? a : b; }
__device__ __forceinline__ uint32_t u32_or(uint32_t a, uint32_t b) { return a | b; }
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
    AZD_DPP_STEP(u32_min, 0xFFFFFFFFu, v, 0x111, 0xf, 0xf);
    AZD_DPP_STEP(u32_min, 0xFFFFFFFFu, v, 0x112, 0xf, 0xf);
    AZD_DPP_STEP(u32_min, 0xFFFFFFFFu, v, 0x114, 0xf, 0xe);
    AZD_DPP_STEP(u32_min, 0xFFFFFFFFu, v, 0x118, 0xf, 0xc);
    AZD_DPP_STEP(u32_min, 0xFFFFFFFFu, v, 0x142, 0xa, 0xf);
    AZD_DPP_STEP(u32_min, 0xFFFFFFFFu, v, 0x143, 0xc, 0xf);
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
    AZD_DPP_STEP(u32_max, 0u, v, 0x111, 0xf, 0xf);
    AZD_DPP_STEP(u32_max, 0u, v, 0x112, 0xf, 0xf);
    AZD_DPP_STEP(u32_max, 0u, v, 0x114, 0xf, 0xe);
    AZD_DPP_STEP(u32_max, 0u, v, 0x118, 0xf, 0xc);
    AZD_DPP_STEP(u32_max, 0u, v, 0x142, 0xa, 0xf);
    AZD_DPP_STEP(u32_max, 0u, v, 0x143, 0xc, 0xf);
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ uint32_t wave_or_u32(uint32_t v) {
    AZD_DPP_STEP(u32_or, 0u, v, 0x111, 0xf, 0xf);
    AZD_DPP_STEP(u32_or, 0u, v, 0x112, 0xf, 0xf);
    AZD_DPP_STEP(u32_or, 0u, v, 0x114, 0xf, 0xe);
    AZD_DPP_STEP(u32_or, 0u, v, 0x118, 0xf, 0xc);
    AZD_DPP_STEP(u32_or, 0u, v, 0x142, 0xa, 0xf);
    AZD_DPP_STEP(u32_or, 0u, v, 0x143, 0xc, 0xf);
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
// 64-bit lexicographic min / max = two 32-bit reductions (high word, then low word among the ties)
__device__ __forceinline__ uint64_t wave_min_u64(uint64_t v) {
    const uint32_t hi = (uint32_t)(v >> 32), lo = (uint32_t)v;
    const uint32_t mh = wave_min_u32(hi);
    const uint32_t ml = wave_min_u32(hi == mh ? lo : 0xFFFFFFFFu);
    return ((uint64_t)mh << 32) | ml;
}
__device__ __forceinline__ uint64_t wave_max_u64(uint64_t v) {
    const uint32_t hi = (uint32_t)(v >> 32), lo = (uint32_t)v;
    const uint32_t mh = wave_max_u32(hi);
    const uint32_t ml = wave_max_u32(hi == mh ? lo : 0u);
    return ((uint64_t)mh << 32) | ml;
}
__device__ __forceinline__ int first_lane(uint64_t mask) { return __ffsll((unsigned long long)mask) - 1; }
__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ uint64_t uni64(uint64_t v) {
    uint32_t lo = uni((uint32_t)v), hi = uni((uint32_t)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}

template <int KW>
__device__ __forceinline__ void clear_bit_range(uint64_t (&m)[KW], int lo, int len) {
#pragma unroll
    for (int w = 0; w < KW; ++w) {
        int a = lo - 64 * w, b = lo + len - 64 * w; // [a, b) within this word
        a = a < 0 ? 0 : a;
        b = b > 64 ? 64 : b;
        if (b > a) {
            uint64_t bits = (b - a == 64) ? ~0ull : (((1ull << (b - a)) - 1ull) << a);
            m[w] &= ~bits;
        }
    }
}
template <int KW>
__device__ __forceinline__ bool mask_empty(const uint64_t (&m)[KW]) {
    uint64_t o = 0;
#pragma unroll
    for (int w = 0; w < KW; ++w) o |= m[w];
    return o == 0;
}
template <int KW>
__device__ __forceinline__ int mask_count(const uint64_t (&m)[KW]) {
    int c = 0;
#pragma unroll
    for (int w = 0; w < KW; ++w) c += __popcll(m[w]);
    return c;
}
template <int KW>
__device__ __forceinline__ uint32_t key_hash(const uint64_t (&k)[KW]) {
    uint64_t h = 0x9E3779B97F4A7C15ull;
#pragma unroll
    for (int w = 0; w < KW; ++w) {
        h = (h ^ k[w]) * 0xFF51AFD7ED558CCDull;
        h ^= h >> 32;
    }
    return (uint32_t)h;
}

// ---------------------------------------------------------------- per-wave LDS
// One block = one wave.  The selection scratch (kids / tmp_arc), the cascade frontier and the
// lambda_1 columns are never live together, so they share the dynamic LDS region (8 KiB per agent
// at N = 19).
struct WaveLds {
    uint8_t par[PARENTS_STRIDE];     // parents of the agent's current state
    uint8_t act_parent[256];         // action id -> (parent, child), ordered_edge.rs:40-42
    uint8_t act_child[256];
    unsigned long long ctr[NUM_COUNTERS]; // per-call counters (wave-uniform; kept out of the register file)
};
// views of the dynamic region
extern __shared__ double azd_dyn_lds[];
__device__ __forceinline__ char *lds_base(uint32_t dyn) { return (char *)azd_dyn_lds + dyn; } // dyn: byte offset of this wave's region
__device__ __forceinline__ float *lds_kids(uint32_t dyn) { return (float *)lds_base(dyn); } // [MAX_NODE_ACTIONS] c_star, newest arc first
__device__ __forceinline__ uint32_t *lds_tmp_arc(uint32_t dyn) { return (uint32_t *)lds_base(dyn) + MAX_NODE_ACTIONS; }
__device__ __forceinline__ uint32_t *lds_fr_id(uint32_t dyn) { return (uint32_t *)lds_base(dyn); }                 // [2][FRONTIER_CAP]
__device__ __forceinline__ uint32_t *lds_fr_x(uint32_t dyn) { return (uint32_t *)lds_base(dyn) + 2 * FRONTIER_CAP; } // [2][FRONTIER_CAP]
__device__ __forceinline__ double *lds_pq(uint32_t dyn) { return (double *)lds_base(dyn); } // [2][n-2][32] lambda_1 (P, Q) columns
static size_t dyn_lds_bytes(int n) {
    size_t fr = (size_t)4 * FRONTIER_CAP * sizeof(uint32_t);
    size_t sel = (size_t)2 * MAX_NODE_ACTIONS * sizeof(uint32_t);
    size_t pq = (size_t)2 * (n > 2 ? n - 2 : 1) * 32 * sizeof(double);
    size_t m = fr > sel ? fr : sel;
    return m > pq ? m : pq;
}

// action id -> (parent, child): index(parent, child) = child(child-1)/2 + parent - 1
// (edge.rs:48-65 colex position minus the skipped edge 0-1, ordered_edge.rs:35-42)
__device__ __forceinline__ void build_action_table(WaveLds &s, int A) {
    for (int i = LANE; i < A; i += 64) {
        int c = 2;
        while (c * (c + 1) / 2 - 1 <= i) ++c;
        s.act_child[i] = (uint8_t)c;
        s.act_parent[i] = (uint8_t)(i - (c * (c - 1) / 2 - 1));
    }
}

// bit a set <=> action a is a current parent edge (rooted_tree/mod.rs:60-72)
template <int KW>
__device__ __forceinline__ void current_edges(const WaveLds &s, int A, uint64_t (&out)[KW]) {
#pragma unroll
    for (int w = 0; w < KW; ++w) {
        int i = w * 64 + LANE;
        bool e = (i < A) && (s.par[s.act_child[i]] == s.act_parent[i]);
        out[w] = __ballot(e);
    }
}

// space.rs:56-73 `act`: parents[child] = parent; drop every permitted (u, child), u < child --
// those ids are the contiguous range [child(child-1)/2 - 1, +child)
template <int KW>
__device__ __forceinline__ void do_act(WaveLds &s, uint64_t (&perm)[KW], uint32_t a) {
    int p = s.act_parent[a], c = s.act_child[a];
    if (LANE == 0) s.par[c] = (uint8_t)p;
    clear_bit_range<KW>(perm, c * (c - 1) / 2 - 1, c);
    WAVE_SYNC();
}

__device__ __forceinline__ uint64_t wave_or_u64(uint64_t v) {
    return ((uint64_t)wave_or_u32((uint32_t)(v >> 32)) << 32) | wave_or_u32((uint32_t)v);
}

// lambda_1 cost contract (DESIGN.md "lambda_1"; stands in for faer at ordered_edge.rs:72-82):
// phi_v = characteristic polynomial of the subtree below v is positive for every v  <=>  x > lambda_1.
// Division-free fold of each vertex into its parent's running pair (P, Q), initially (1, 0):
//     phi = x P[v] - Q[v];   Q[p] = Q[p] phi + P[p] P[v];   P[p] = P[p] phi        (v = N-1 .. 1)
// at 32 trial points per round (33-section of the bracket; lanes l and l+32 duplicate each other),
// at most 12 rounds.  FULL = false (node costs): stop once both bracket ends round to the same f32,
// which is all `lambda_1 as f32` (04-c21-tree.rs:100) can see.
// Shape of the loop: the kernel waits for its slowest agent, so single-wave latency is what counts,
// and on this machine that means NO BRANCHES in the vertex loop (a taken scalar branch costs more
// than the six f64 operations of a fold).  Every vertex runs the same straight-line body; the
// wave-uniform tree only enters through LDS addresses: per-lane (P, Q) live in LDS columns
// [slot][lane], slot(v) = v for v = 0..n-3 (vertices n-1 and n-2 are leaves by construction and are
// peeled), and parents are packed 5 bits each into two scalars, so the loop issues no LDS read for
// the tree itself.  Four ds_read_b64 go out together per vertex, then six DP ops, then two writes.
// parents[] as two wave-uniform scalars, 5 bits per vertex (12 vertices per word)
struct PackedTree {
    uint64_t par0, par1;
};
__device__ __forceinline__ PackedTree pack_tree(const WaveLds &s, int n) {
    uint64_t w0 = 0, w1 = 0;
    if (LANE >= 1 && LANE < n) {
        uint64_t pv = s.par[LANE];
        if (LANE < 12) w0 = pv << (5 * LANE);
        else w1 = pv << (5 * (LANE - 12));
    }
    PackedTree t;
    t.par0 = uni64(wave_or_u64(w0));
    t.par1 = uni64(wave_or_u64(w1));
    return t;
}

// |maximum matching| of the tree.  The reference strips leaves round by round
// (ordered_edge.rs:94-124; restated literally in matching_wave below, which k_argmin uses to report
// the edges); that procedure only ever matches a leaf of the remaining forest with its parent,
// which is always safe, so its size is the matching number -- here computed by the one-pass
// children-before-parents greedy on the scalar unit (tests compare both against the oracle).
__device__ __forceinline__ int matching_size_wave(const PackedTree &t, int n) {
    uint32_t matched = 0;
    int m = 0;
    for (int v = n - 1; v >= 1; --v) {
        const uint32_t p = (uint32_t)((v < 12 ? (t.par0 >> (5 * v)) : (t.par1 >> (5 * (v - 12)))) & 31ull);
        const uint32_t free_both = ~(matched >> v) & ~(matched >> p) & 1u;
        matched |= (free_both << v) | (free_both << p);
        m += (int)free_both;
    }
    return m;
}

// NS > 0: the vertex count as a compile-time constant (the vertex loop unrolls: no loop branch,
// constant LDS offsets and field shifts); NS = 0: runtime n.
template <bool FULL, int NS>
__device__ double lambda1_impl(const PackedTree &t, int n_rt, uint32_t dyn) {
    const int n = NS ? NS : n_rt;
    uint64_t par0 = t.par0, par1 = t.par1;
    const int nslot = n > 2 ? n - 2 : 1;
    const int l = LANE & 31;
    double *Pm = lds_pq(dyn) + l;
    double *Qm = Pm + nslot * 32;
    const int p_last = (int)(((n - 1) < 12 ? (par0 >> (5 * (n - 1))) : (par1 >> (5 * (n - 13)))) & 31ull);
    const int p_prev = (int)(((n - 2) < 12 ? (par0 >> (5 * (n - 2))) : (par1 >> (5 * (n - 14)))) & 31ull);
    double lo = 1.0, hi = (double)n;
    for (int round = 0; round < 12; ++round) {
        if (!FULL && (float)lo == (float)hi) break;
        const double w = (hi - lo) / 33.0;
        const double step = w * (double)(l + 1);
        const double x = lo + step;
        for (int v = 0; v < nslot; ++v) {
            Pm[v * 32] = 1.0;
            Qm[v * 32] = 0.0;
        }
        bool ok = x > 0.0;
        // peeled leaves n-1 and n-2 (n >= 4): phi = x*1 - 0 = x, psi = 1
        {
            const double pp = Pm[p_last * 32], qp = Qm[p_last * 32];
            const double qphi = qp * x, ppsi = pp * 1.0;
            Qm[p_last * 32] = qphi + ppsi;
            Pm[p_last * 32] = pp * x;
        }
        {
            const double pp = Pm[p_prev * 32], qp = Qm[p_prev * 32];
            const double qphi = qp * x, ppsi = pp * 1.0;
            Qm[p_prev * 32] = qphi + ppsi;
            Pm[p_prev * 32] = pp * x;
        }
        for (int v = n - 3; v >= 1; --v) {
            const int p = (int)((v < 12 ? (par0 >> (5 * v)) : (par1 >> (5 * (v - 12)))) & 31ull);
            const double pv = Pm[v * 32], qv = Qm[v * 32];
            const double pp = Pm[p * 32], qp = Qm[p * 32];
            const double xp = x * pv;
            const double phi = xp - qv;
            ok = ok && (phi > 0.0);
            const double qphi = qp * phi;
            const double ppsi = pp * pv;
            Qm[p * 32] = qphi + ppsi;
            Pm[p * 32] = pp * phi;
        }
        const double xp0 = x * Pm[0];
        const double phi0 = xp0 - Qm[0];
        ok = ok && (phi0 > 0.0);
        const uint32_t m = (uint32_t)__ballot(ok);
        const int first = m ? (__ffs((int)m) - 1) : 32;
        const double x_prev = __shfl(x, first > 0 ? first - 1 : 0, 64);
        const double x_first = __shfl(x, first < 32 ? first : 31, 64);
        const double nlo = first > 0 ? x_prev : lo;
        const double nhi = first < 32 ? x_first : hi;
        lo = nlo;
        hi = nhi;
    }
    return hi;
}

template <bool FULL>
__device__ __forceinline__ double lambda1_wave(const PackedTree &t, int n, uint32_t dyn) {
    // lambda1_impl<FULL, 19> (static unroll) is 1.8x faster in isolation (tools/probe_cost.py) but its
    // register pressure spills the search loop around it: measured slower end to end, so not used.
    return lambda1_impl<FULL, 0>(t, n, dyn);
}

// ordered_edge.rs:94-124 maximum_matching (leaf stripping), on bit masks.  Returns |matching|;
// optionally writes the (parent, child) pairs.
__device__ int matching_wave(const WaveLds &s, int n, int32_t *pairs_out) {
    uint32_t avail = (n >= 32) ? 0xFFFFFFFFu : ((1u << n) - 1u);
    int m = 0;
    for (int guard = 0; guard < 64; ++guard) {
        uint32_t leaf = avail;
        for (int i = 1; i < n; ++i)
            if ((avail >> i) & 1u) leaf &= ~(1u << s.par[i]);
        for (int i = 1; i < n; ++i) {
            if ((leaf >> i) & 1u) {
                avail &= ~(1u << i);
                int p = s.par[i];
                if ((avail >> p) & 1u) {
                    avail &= ~(1u << p);
                    if (pairs_out && LANE == 0) {
                        pairs_out[2 * m] = p;
                        pairs_out[2 * m + 1] = i;
                    }
                    ++m;
                }
            }
        }
        if (__popc(avail) < 2) break;
    }
    return m;
}

// 04-c21-tree.rs:98-102: squish(matching.len() as f32 + lambda_1 as f32)
__device__ __forceinline__ float c21_eval(float slope, double lambda1, int mu) {
    float c = (float)mu + (float)lambda1;
    float x = c - 2.0f;
    return slope * x;
}

// space.rs:91-101 write_vec: [0, A) one-hot of current parent edges, [A, 2A) permitted mask
template <int KW>
__device__ __forceinline__ void write_state_vec(const WaveLds &s, const uint64_t (&perm)[KW], int A, float *row) {
    uint64_t cur[KW];
    current_edges<KW>(s, A, cur);
#pragma unroll
    for (int w = 0; w < KW; ++w) {
        int i = w * 64 + LANE;
        if (i < A) {
            row[i] = (float)((cur[w] >> LANE) & 1ull);
            row[A + i] = (float)((perm[w] >> LANE) & 1ull);
        }
    }
}

// ---------------------------------------------------------------- transposition table
// Wave-parallel linear probing: 64 consecutive slots per probe, keys compared by the lanes
// that hit an occupied slot.  Stands in for BTreeMap::get / insert (tree/mod.rs:170,188).
template <int KW>
__device__ uint32_t ht_lookup(const uint32_t *ht, uint32_t mask, const uint64_t *keys, const uint64_t (&k)[KW],
                              uint32_t *ins_slot) {
    uint32_t h = key_hash<KW>(k) & mask;
    for (uint32_t probe = 0; probe <= mask; probe += 64) {
        uint32_t slot = (h + probe + (uint32_t)LANE) & mask;
        uint32_t nd = ht[slot];
        bool empty = nd == NONE;
        bool match = !empty;
        if (!empty) {
#pragma unroll
            for (int w = 0; w < KW; ++w) match = match && (keys[(size_t)nd * KW + w] == k[w]);
        }
        uint64_t mm = __ballot(match), me = __ballot(empty);
        int fm = mm ? first_lane(mm) : 64, fe = me ? first_lane(me) : 64;
        if (fm < fe) return (uint32_t)__shfl((int)nd, fm, 64);
        if (fe < 64) {
            *ins_slot = (h + probe + (uint32_t)fe) & mask;
            return NONE;
        }
    }
    *ins_slot = NONE;
    return NONE;
}

// ---------------------------------------------------------------- per-agent context
template <int KW>
struct Agent {
    NodeRec *nodes;
    uint64_t *keys;
    ArcRec *arcs;
    PredRec *preds;
    uint32_t *ht;
    uint32_t n_nodes, n_arcs, n_preds;
    uint32_t flags;
    float cand_c;
    uint32_t cand_node;
};

__device__ __forceinline__ bool node_active(const NodeRec &r) { return r.act_begin + r.exhausted < r.act_end; }

// empty_transitions.rs:50-87 (old = false) / :89-127 (old = true).  Level-synchronous sweep over
// the ancestors of arc (src -> dst); with ActionSet keys the DAG is layered (depth = |set|), so a
// frontier holds one depth only and the order inside a level cannot matter; the in-list order
// cannot matter either because the merge is (min, +).  The propagated c_t_star never changes
// along the sweep (every emitted Info carries the value it received), so it is one scalar.
template <int KW>
__device__ void cascade(const Arenas &a, Agent<KW> &ag, WaveLds &s, uint32_t dyn, uint32_t src, uint32_t dst, bool old) {
    NodeRec t = ag.nodes[dst];
    const uint32_t n_t_target = t.n_t;
    const float c = t.c_star;
    // apply one Info to node u; returns the x it sends up (0 / 1) and its record
    auto visit = [&](uint32_t u, uint32_t x, NodeRec &r) -> uint32_t {
        r = ag.nodes[u];
        r.exhausted += x;
        if (r.c_star > c) r.c_star = c;
        else r.n_t += 1;
        if (old) r.n_t = r.n_t > n_t_target ? r.n_t : n_t_target;
        if (LANE == 0) {
            ag.nodes[u].c_star = r.c_star;
            ag.nodes[u].n_t = r.n_t;
            ag.nodes[u].exhausted = r.exhausted;
        }
        CTR_ADD(7, 1);
        return node_active(r) ? 0u : 1u;
    };
    // ---- fast path: while the frontier is one node whose only in-arc is the one that created it
    // (NodeRec.in_src), the sweep is a plain walk up the creating chain: one record load per level,
    // no LDS frontier.
    uint32_t u1 = src, x1 = old ? (node_active(t) ? 0u : 1u) : 1u;
    for (;;) {
        CTR_MAX(10, 1);
        NodeRec r;
        const uint32_t up_x = visit(u1, x1, r);
        if (r.first_in == NONE) {
            if (r.in_src == NONE) return; // root done
            u1 = r.in_src;
            x1 = up_x;
            continue;
        }
        // several parents: seed the LDS frontier with them and fall through to the general sweep
        uint32_t n0 = 0;
        if (LANE == 0 && r.in_src != NONE) {
            lds_fr_id(dyn)[0] = r.in_src;
            lds_fr_x(dyn)[0] = up_x;
        }
        if (r.in_src != NONE) n0 = 1;
        WAVE_SYNC();
        for (uint32_t e = r.first_in; e != NONE;) {
            ArcRec ar = ag.arcs[e];
            // parents of one node are distinct (one arc per (parent, action)), so no merge here
            if (n0 >= FRONTIER_CAP) {
                ag.flags |= FLAG_FRONTIER_CAP;
                return;
            }
            if (LANE == 0) {
                lds_fr_id(dyn)[n0] = ar.src;
                lds_fr_x(dyn)[n0] = up_x;
            }
            n0 += 1;
            e = ar.next_in;
        }
        WAVE_SYNC();
        // ---- general level-synchronous sweep with an LDS-staged frontier
        int cur = 0;
        uint32_t n_cur = n0;
        while (n_cur != 0) {
            uint32_t n_nxt = 0;
            const int nxt = cur ^ 1;
            CTR_MAX(10, n_cur);
            for (uint32_t i = 0; i < n_cur; ++i) {
                const uint32_t u = lds_fr_id(dyn)[cur * FRONTIER_CAP + i];
                const uint32_t x = lds_fr_x(dyn)[cur * FRONTIER_CAP + i];
                NodeRec ru;
                const uint32_t ux = visit(u, x, ru);
                uint32_t e = ru.first_in;
                uint32_t p = ru.in_src;
                bool from_list = false;
                if (p == NONE) {
                    if (e == NONE) continue; // the root
                    ArcRec ar = ag.arcs[e];
                    p = ar.src;
                    e = ar.next_in;
                    from_list = true;
                }
                (void)from_list;
                for (;;) {
                    int found = -1;
                    for (uint32_t base = 0; base < n_nxt; base += 64) {
                        uint32_t j = base + (uint32_t)LANE;
                        bool hit = (j < n_nxt) && (lds_fr_id(dyn)[nxt * FRONTIER_CAP + j] == p);
                        uint64_t m = __ballot(hit);
                        if (m) {
                            found = (int)base + first_lane(m);
                            break;
                        }
                    }
                    if (found >= 0) {
                        if (LANE == 0) lds_fr_x(dyn)[nxt * FRONTIER_CAP + found] += ux;
                    } else {
                        if (n_nxt >= FRONTIER_CAP) {
                            ag.flags |= FLAG_FRONTIER_CAP;
                            return;
                        }
                        if (LANE == 0) {
                            lds_fr_id(dyn)[nxt * FRONTIER_CAP + n_nxt] = p;
                            lds_fr_x(dyn)[nxt * FRONTIER_CAP + n_nxt] = ux;
                        }
                        n_nxt += 1;
                    }
                    WAVE_SYNC();
                    if (e == NONE) break;
                    ArcRec ar = ag.arcs[e];
                    p = ar.src;
                    e = ar.next_in;
                }
            }
            cur = nxt;
            n_cur = n_nxt;
        }
        return;
    }
}

// graph_operations.rs:18-30 add_arc (+ petgraph head insertion into dst's in-list)
// `creating`: the arc that creates dst -- its source is stored in the node (NodeRec.in_src); only
// later arcs into an existing node (transpositions) are chained through first_in / next_in.
template <int KW>
__device__ __forceinline__ uint32_t add_arc(Agent<KW> &ag, uint32_t src, uint32_t dst, uint32_t pp, bool creating) {
    uint32_t e = ag.n_arcs;
    if (LANE == 0) {
        ArcRec ar;
        ar.src = src;
        ar.dst = dst;
        ar.pp = pp;
        ar.next_in = creating ? NONE : ag.nodes[dst].first_in;
        ag.arcs[e] = ar;
        if (creating) ag.nodes[dst].in_src = src;
        else ag.nodes[dst].first_in = e;
        ag.preds[pp].arc = e;
        ag.preds[pp].child = dst;
    }
    ag.n_arcs = e + 1;
    return e;
}

// ---------------------------------------------------------------- kernels
#ifndef AZD_TU_ASYNC
template <int KW, bool BIG>
__global__ __launch_bounds__(64) void k_init_roots(Arenas a, const uint8_t *__restrict__ parents,
                                                   const uint64_t *__restrict__ permitted) {
    __shared__ WaveLds s;
    const uint32_t dyn = 0;
    const int t = blockIdx.x;
    const int n = a.n, A = a.A;
    build_action_table(s, A);
    if (LANE < PARENTS_STRIDE) {
        uint8_t p = LANE < n ? parents[(size_t)t * n + LANE] : 0;
        s.par[LANE] = p;
        a.root_parents[(size_t)t * PARENTS_STRIDE + LANE] = p;
        a.cur_parents[(size_t)t * PARENTS_STRIDE + LANE] = p;
    }
    uint64_t perm[KW];
#pragma unroll
    for (int w = 0; w < KW; ++w) {
        perm[w] = permitted[(size_t)t * KW + w];
        if (LANE == 0) {
            a.root_perm[(size_t)t * KW + w] = perm[w];
            a.cur_perm[(size_t)t * KW + w] = perm[w];
            a.cur_path[(size_t)t * KW + w] = 0;
        }
    }
    WAVE_SYNC();
    // optimizer/mod.rs:63 costs = space.cost(root)
    const PackedTree pt = pack_tree(s, n);
    double lam = lambda1_wave<false>(pt, n, dyn);
    int mu = matching_size_wave(pt, n);
    float c = c21_eval(a.eval_slope, lam, mu);
    // SearchTree::clear + add_node(P::new(), StateWeight::new(c)) (optimizer/mod.rs:81-84, :353-356)
    uint32_t *ht = a.ht + (size_t)t * a.ht_cap;
    for (uint32_t i = LANE; i < a.ht_cap; i += 64) ht[i] = NONE;
    WAVE_SYNC();
    uint64_t zero[KW];
#pragma unroll
    for (int w = 0; w < KW; ++w) zero[w] = 0;
    if (LANE == 0) {
        NodeRec r;
        r.c = c; r.c_star = c; r.n_t = 0; r.exhausted = 0; r.act_begin = 0; r.act_end = 0; r.first_in = NONE; r.in_src = NONE;
        a.nodes[(size_t)t * a.node_cap] = r;
#pragma unroll
        for (int w = 0; w < KW; ++w) a.keys[((size_t)t * a.node_cap) * KW + w] = 0;
        ht[key_hash<KW>(zero) & (a.ht_cap - 1)] = 0;
        a.cur_lambda[t] = lam;
        a.cur_mu[t] = mu;
        a.state_pos[t] = 0;
        a.n_nodes[t] = 1;
        a.n_arcs[t] = 0;
        a.n_preds[t] = 0;
        a.flags[t] = 0;
        a.cand_c[t] = c; // num_inspected_nodes = 0: the root is inspected by the next argmin pass
        a.cand_node[t] = 0;
    }
    write_state_vec<KW>(s, perm, A, a.state_vecs + (size_t)t * a.S);
}

#endif // !AZD_TU_ASYNC
// graph_operations.rs:32-56 add_actions for the node the agent stands on.
// root_mode = 1: par_new / par_reset_trees (every agent, node 0); 0: after a roll-out (agents
// whose path is non-empty, optimizer/mod.rs:186).
template <int KW>
__device__ void add_actions_agent(const Arenas &a, WaveLds &s, const int t, const int root_mode) {
    if (a.flags[t] != 0) return;
    uint64_t perm[KW], path[KW];
#pragma unroll
    for (int w = 0; w < KW; ++w) {
        perm[w] = a.cur_perm[(size_t)t * KW + w];
        path[w] = a.cur_path[(size_t)t * KW + w];
    }
    if (!root_mode && mask_empty<KW>(path)) return;
    const int A = a.A;
    if (LANE < PARENTS_STRIDE) s.par[LANE] = a.cur_parents[(size_t)t * PARENTS_STRIDE + LANE];
    WAVE_SYNC();
    uint64_t cur[KW], legal[KW];
    current_edges<KW>(s, A, cur);
#pragma unroll
    for (int w = 0; w < KW; ++w) legal[w] = perm[w] & ~cur[w]; // space.rs:75-89 action_data
    const uint32_t cnt = (uint32_t)mask_count<KW>(legal);
    const uint32_t begin = a.n_preds[t];
    uint32_t fl = 0;
    if (cnt > MAX_NODE_ACTIONS) fl |= FLAG_NODE_ACTIONS;
    if (begin + cnt > a.pred_cap) fl |= FLAG_PRED_CAP;
    if (fl) {
        if (LANE == 0) a.flags[t] = fl;
        return;
    }
    const uint32_t node = a.state_pos[t];
    NodeRec *nodes = a.nodes + (size_t)t * a.node_cap;
    PredRec *preds = a.preds + (size_t)t * a.pred_cap;
    const float c = nodes[node].c;
    const float *h = a.h_theta + (size_t)t * A;
    uint32_t before = 0;
#pragma unroll
    for (int w = 0; w < KW; ++w) {
        int i = w * 64 + LANE;
        if ((legal[w] >> LANE) & 1ull) {
            uint32_t rank = before + (uint32_t)__popcll(legal[w] & ((1ull << LANE) - 1ull));
            PredRec p;
            p.a_id = (uint32_t)i;
            p.g = c - h[i]; // g_theta_star_sa = c_s - h_theta_sa (04-c21-tree.rs:103)
            p.arc = NONE;
            p.child = NONE;
            preds[begin + rank] = p;
        }
        before += (uint32_t)__popcll(legal[w]);
    }
    if (LANE == 0) {
        nodes[node].act_begin = begin;
        nodes[node].act_end = begin + cnt;
        a.n_preds[t] = begin + cnt;
        a.counters[(size_t)t * NUM_COUNTERS + 8] += cnt;
    }
}

#ifndef AZD_TU_ASYNC
template <int KW>
__global__ __launch_bounds__(64) void k_add_actions(Arenas a, int root_mode) {
    __shared__ WaveLds s;
    build_action_table(s, a.A);
    add_actions_agent<KW>(a, s, (int)blockIdx.x, root_mode);
}

#endif // !AZD_TU_ASYNC
// tree/mod.rs:113-232 roll_out_episodes for every agent (optimizer/mod.rs:159-174)
// One agent's call, run by one wavefront.  `s` is the wave's LDS block (action tables already
// built), `dyn` the byte offset of its scratch region.  Returns true iff the call ended on a new
// non-terminal node (the agent needs a prediction row).
template <int KW>
__device__ bool rollout_agent(const Arenas &a, const TolTable &tol, WaveLds &s, const uint32_t dyn, const int t) {
    if (a.flags[t] != 0) return false;
    const int n = a.n, A = a.A;
    Agent<KW> ag;
    ag.nodes = a.nodes + (size_t)t * a.node_cap;
    ag.keys = a.keys + (size_t)t * a.node_cap * KW;
    ag.arcs = a.arcs + (size_t)t * a.arc_cap;
    ag.preds = a.preds + (size_t)t * a.pred_cap;
    ag.ht = a.ht + (size_t)t * a.ht_cap;
    ag.n_nodes = a.n_nodes[t];
    ag.n_arcs = a.n_arcs[t];
    ag.n_preds = a.n_preds[t];
    ag.flags = 0;
    ag.cand_c = a.cand_c[t];
    ag.cand_node = a.cand_node[t];
    if (LANE < 24) s.ctr[LANE] = 0; // slots 24.. belong to the asynchronous step's evaluator service

    if (LANE < PARENTS_STRIDE) s.par[LANE] = a.cur_parents[(size_t)t * PARENTS_STRIDE + LANE];
    uint64_t perm[KW], path[KW];
#pragma unroll
    for (int w = 0; w < KW; ++w) {
        perm[w] = a.cur_perm[(size_t)t * KW + w];
        path[w] = a.cur_path[(size_t)t * KW + w];
    }
    uint32_t pos = a.state_pos[t];
    double cur_lambda = a.cur_lambda[t];
    int cur_mu = a.cur_mu[t];
    WAVE_SYNC();

    bool expanded_new = false;
    // The record of the node stepped into comes out of the parent's child gather (nothing modifies the
    // tree between that gather and the step), which removes one of the three dependent loads per level.
    NodeRec rec_next;
    bool have_rec = false;
    const unsigned long long ph_begin = PH_NOW();
    for (uint32_t guard = 0;; ++guard) {
        if (guard > (1u << 22)) {
            ag.flags |= FLAG_LOOP_GUARD;
            break;
        }
        // ---- next_action (next_action.rs:11-26)
        const unsigned long long ph_sel0 = PH_NOW();
        NodeRec rec;
        if (have_rec) rec = rec_next;
        else rec = ag.nodes[pos];
        have_rec = false;
        const int depth = mask_count<KW>(path);
        const uint32_t tl = depth < tol.n_tol ? tol.tol[depth] : tol.tol_default;
        int kind = 0; // 0 None, 1 Visited, 2 Unvisited
        uint32_t sel_pp = 0, sel_child = 0, sel_aid = 0;
        if (node_active(rec)) {
            const uint32_t nact = rec.act_end - rec.act_begin;
            CTR_ADD(4, 1);
            CTR_ADD(6, nact);
            PredRec p[PRED_CHUNKS];
            bool valid[PRED_CHUNKS], expd[PRED_CHUNKS];
            float k_cstar[PRED_CHUNKS];
            NodeRec crs[PRED_CHUNKS];
            uint64_t rkey[PRED_CHUNKS];
            uint32_t n_exp = 0;
            uint64_t exp_mask[PRED_CHUNKS];
#pragma unroll
            for (int ch = 0; ch < PRED_CHUNKS; ++ch) {
                uint32_t idx = (uint32_t)(ch * 64 + LANE);
                valid[ch] = idx < nact;
                p[ch].a_id = 0; p[ch].g = 0.f; p[ch].arc = NONE; p[ch].child = NONE;
                if (valid[ch]) p[ch] = ag.preds[rec.act_begin + idx];
                expd[ch] = valid[ch] && p[ch].arc != NONE;
                k_cstar[ch] = 0.f;
                rkey[ch] = ~0ull;
                if (expd[ch]) {
                    NodeRec cr = ag.nodes[p[ch].child];
                    crs[ch] = cr;
                    k_cstar[ch] = cr.c_star;
                    if (node_active(cr)) rkey[ch] = ((uint64_t)cr.n_t << 32) | (uint64_t)ordf(cr.c_star);
                }
                exp_mask[ch] = __ballot(expd[ch]);
                n_exp += (uint32_t)__popcll(exp_mask[ch]);
            }
            CTR_ADD(5, n_exp);
            // ---- revisit_choice (next_action.rs:28-53): first-min of (n_t, c_t_star) over ACTIVE
            // children in newest-arc-first order  ==  min key, ties -> largest arc id
            uint64_t kmin = rkey[0];
#pragma unroll
            for (int ch = 1; ch < PRED_CHUNKS; ++ch) kmin = rkey[ch] < kmin ? rkey[ch] : kmin;
            kmin = wave_min_u64(kmin);
            bool have_r = kmin != ~0ull;
            uint32_t r_arc = 0, r_nt = (uint32_t)(kmin >> 32);
            if (have_r) {
                uint32_t best = 0;
#pragma unroll
                for (int ch = 0; ch < PRED_CHUNKS; ++ch)
                    if (rkey[ch] == kmin) best = p[ch].arc + 1 > best ? p[ch].arc + 1 : best;
                r_arc = wave_max_u32(best) - 1;
            }
            if (have_r && r_nt < tl) kind = 1;
            else {
                // ---- max_curiosity (next_action.rs:55-88)
                // children's c_t_star (ALL children), newest arc first: rank = #arcs with larger id
                if (n_exp > 0) {
                    uint32_t off = 0;
#pragma unroll
                    for (int ch = 0; ch < PRED_CHUNKS; ++ch) {
                        if (expd[ch]) lds_tmp_arc(dyn)[off + (uint32_t)__popcll(exp_mask[ch] & ((1ull << LANE) - 1ull))] = p[ch].arc;
                        off += (uint32_t)__popcll(exp_mask[ch]);
                    }
                    WAVE_SYNC();
#pragma unroll
                    for (int ch = 0; ch < PRED_CHUNKS; ++ch) {
                        if (expd[ch]) {
                            uint32_t rank = 0;
                            for (uint32_t j = 0; j < n_exp; ++j) rank += lds_tmp_arc(dyn)[j] > p[ch].arc ? 1u : 0u;
                            lds_kids(dyn)[rank] = k_cstar[ch];
                        }
                    }
                    WAVE_SYNC();
                }
                uint64_t best = 0;
                bool has_cand = false;
#pragma unroll
                for (int ch = 0; ch < PRED_CHUNKS; ++ch) {
                    if (valid[ch] && !expd[ch]) {
                        uint32_t idx = (uint32_t)(ch * 64 + LANE);
                        float v = rec.c - p[ch].g; // c_theta_star
                        uint64_t key;
                        if (n_exp == 0) {
                            // min_by: first minimum  ->  maximise (~ord(v), ~idx)
                            key = ((uint64_t)(~ordf(v)) << 32) | (uint64_t)(0xFFFFFFFFu - idx);
                        } else {
                            float sum = 0.0f; // f32 sum, sequential, newest child first
                            for (uint32_t j = 0; j < n_exp; ++j) sum = sum + azd_sqrt(fabsf(lds_kids(dyn)[j] - v));
                            // max_by: last maximum  ->  maximise (ord(sum), idx)
                            key = ((uint64_t)ordf(sum) << 32) | (uint64_t)idx;
                        }
                        best = (!has_cand || key > best) ? key : best;
                        has_cand = true;
                    }
                }
                CTR_ADD(12, (unsigned long long)n_exp * (unsigned long long)(nact - n_exp));
                const bool any_cand = __ballot(has_cand) != 0;
                best = wave_max_u64(has_cand ? best : 0ull);
                if (any_cand) {
                    uint32_t idx = (uint32_t)(best & 0xFFFFFFFFull);
                    if (n_exp == 0) idx = 0xFFFFFFFFu - idx;
                    kind = 2;
                    sel_pp = rec.act_begin + idx;
                    sel_aid = (uint32_t)__shfl((int)(idx >= 64 ? p[1].a_id : p[0].a_id), (int)(idx & 63u), 64);
                } else if (have_r) kind = 1;
            }
            if (kind == 1) {
                // locate the chosen arc's prediction
                uint32_t a_id = 0, child = 0;
                NodeRec mcr;
                mcr.c = 0.f; mcr.c_star = 0.f; mcr.n_t = 0; mcr.exhausted = 0; mcr.act_begin = 0; mcr.act_end = 0; mcr.first_in = NONE; mcr.in_src = NONE;
                bool mine = false;
#pragma unroll
                for (int ch = 0; ch < PRED_CHUNKS; ++ch)
                    if (expd[ch] && p[ch].arc == r_arc) { mine = true; a_id = p[ch].a_id; child = p[ch].child; mcr = crs[ch]; }
                uint64_t mm = __ballot(mine);
                int src_lane = first_lane(mm);
                sel_aid = (uint32_t)__shfl((int)a_id, src_lane, 64);
                sel_child = (uint32_t)__shfl((int)child, src_lane, 64);
                rec_next.c = __shfl(mcr.c, src_lane, 64);
                rec_next.c_star = __shfl(mcr.c_star, src_lane, 64);
                rec_next.n_t = (uint32_t)__shfl((int)mcr.n_t, src_lane, 64);
                rec_next.exhausted = (uint32_t)__shfl((int)mcr.exhausted, src_lane, 64);
                rec_next.act_begin = (uint32_t)__shfl((int)mcr.act_begin, src_lane, 64);
                rec_next.act_end = (uint32_t)__shfl((int)mcr.act_end, src_lane, 64);
                rec_next.first_in = (uint32_t)__shfl((int)mcr.first_in, src_lane, 64);
                rec_next.in_src = (uint32_t)__shfl((int)mcr.in_src, src_lane, 64);
                have_rec = true;
            }
        }
        kind = (int)uni((uint32_t)kind);
        sel_aid = uni(sel_aid);
        CTR_ADD(17, PH_NOW() - ph_sel0);

        if (kind == 1) { // Visited (tree/mod.rs:139-151)
            sel_child = uni(sel_child);
            path[sel_aid >> 6] |= 1ull << (sel_aid & 63u);
            do_act<KW>(s, perm, sel_aid);
            pos = sel_child;
            CTR_ADD(3, 1);
            continue;
        }
        if (kind == 0) { // None (tree/mod.rs:220-229)
            if (mask_empty<KW>(path)) CTR_ADD(9, 1);
            else ag.flags |= FLAG_UNREACHABLE;
            break;
        }
        // ---- Unvisited(prediction_pos) (tree/mod.rs:160-218)
        sel_pp = uni(sel_pp);
        path[sel_aid >> 6] |= 1ull << (sel_aid & 63u);
        {
            uint64_t d = (uint64_t)mask_count<KW>(path);
            CTR_MAX(11, d);
        }
        if (ag.n_arcs >= a.arc_cap) {
            ag.flags |= FLAG_ARC_CAP;
            break;
        }
        uint32_t ins_slot = NONE;
        const unsigned long long ph_lk0 = PH_NOW();
        uint32_t hit = ht_lookup<KW>(ag.ht, a.ht_cap - 1, ag.keys, path, &ins_slot);
        hit = uni(hit);
        CTR_ADD(18, PH_NOW() - ph_lk0);
        bool reset_to_root = false;
        if (hit != NONE) { // transposition (tree/mod.rs:172-179)
            add_arc<KW>(ag, pos, hit, sel_pp, false);
            WAVE_SYNC();
            const unsigned long long ph_c0 = PH_NOW();
            cascade<KW>(a, ag, s, dyn, pos, hit, true);
            CTR_ADD(20, PH_NOW() - ph_c0);
            CTR_ADD(2, 1);
            reset_to_root = true;
        } else { // new node (tree/mod.rs:180-216)
            ins_slot = uni(ins_slot);
            if (ag.n_nodes >= a.node_cap) {
                ag.flags |= FLAG_NODE_CAP;
                break;
            }
            if (ins_slot == NONE) {
                ag.flags |= FLAG_HT_FULL;
                break;
            }
            const unsigned long long ph_n0 = PH_NOW();
            do_act<KW>(s, perm, sel_aid);
            const unsigned long long ph_l0 = PH_NOW();
            const PackedTree pt = pack_tree(s, n);
            cur_lambda = lambda1_wave<false>(pt, n, dyn);
            const unsigned long long ph_l1 = PH_NOW();
            cur_mu = matching_size_wave(pt, n);
            CTR_ADD(22, ph_l1 - ph_l0);
            CTR_ADD(23, PH_NOW() - ph_l1);
            const float c_as = c21_eval(a.eval_slope, cur_lambda, cur_mu);
            CTR_ADD(19, PH_NOW() - ph_n0);
            const uint32_t v = ag.n_nodes;
            if (LANE == 0) {
                NodeRec r;
                r.c = c_as; r.c_star = c_as; r.n_t = 0; r.exhausted = 0; r.act_begin = 0; r.act_end = 0; r.first_in = NONE; r.in_src = NONE;
                ag.nodes[v] = r;
#pragma unroll
                for (int w = 0; w < KW; ++w) ag.keys[(size_t)v * KW + w] = path[w];
                ag.ht[ins_slot] = v;
            }
            ag.n_nodes = v + 1;
            if (c_as < ag.cand_c || ag.cand_node == NONE) { // first-min over nodes since last inspection
                ag.cand_c = c_as;
                ag.cand_node = v;
            }
            WAVE_SYNC();
            add_arc<KW>(ag, pos, v, sel_pp, true);
            WAVE_SYNC();
            uint64_t cur[KW], legal[KW];
            current_edges<KW>(s, A, cur);
#pragma unroll
            for (int w = 0; w < KW; ++w) legal[w] = perm[w] & ~cur[w];
            if (mask_empty<KW>(legal)) { // terminal: is_terminal, nabla/space/mod.rs:23-25
                const unsigned long long ph_c0 = PH_NOW();
                cascade<KW>(a, ag, s, dyn, pos, v, false);
                CTR_ADD(20, PH_NOW() - ph_c0);
                CTR_ADD(1, 1);
                reset_to_root = true;
            } else {
                pos = v;
                CTR_ADD(0, 1);
                expanded_new = true;
                break;
            }
        }
        if (ag.flags) break;
        if (reset_to_root) { // state.clone_from(root); path.clear(); state_pos = root
            WAVE_SYNC();
            if (LANE < PARENTS_STRIDE) s.par[LANE] = a.root_parents[(size_t)t * PARENTS_STRIDE + LANE];
#pragma unroll
            for (int w = 0; w < KW; ++w) {
                perm[w] = a.root_perm[(size_t)t * KW + w];
                path[w] = 0;
            }
            pos = 0;
            WAVE_SYNC();
        }
    }

    // ---- write back; optimizer/mod.rs:171-173 write_vec iff the path is non-empty
    CTR_ADD(16, PH_NOW() - ph_begin);
    CTR_ADD(21, PH_NOW() - ph_begin);
    WAVE_SYNC();
    if (expanded_new) write_state_vec<KW>(s, perm, A, a.state_vecs + (size_t)t * a.S);
    if (LANE < PARENTS_STRIDE) a.cur_parents[(size_t)t * PARENTS_STRIDE + LANE] = s.par[LANE];
    if (LANE == 0) {
#pragma unroll
        for (int w = 0; w < KW; ++w) {
            a.cur_perm[(size_t)t * KW + w] = perm[w];
            a.cur_path[(size_t)t * KW + w] = path[w];
        }
        a.state_pos[t] = pos;
        a.cur_lambda[t] = cur_lambda;
        a.cur_mu[t] = cur_mu;
        a.n_nodes[t] = ag.n_nodes;
        a.n_arcs[t] = ag.n_arcs;
        a.cand_c[t] = ag.cand_c;
        a.cand_node[t] = ag.cand_node;
        if (ag.flags) {
            a.flags[t] = ag.flags;
            atomicAdd(&a.status->failed, 1ull);
        }
        if (expanded_new) atomicAdd(&a.status->expansions, 1ull);
    }
    // per-call counters -> the agent's global block, one counter per lane
    if (LANE < 24) {
        const int k = LANE;
        const unsigned long long v = s.ctr[k];
        unsigned long long *ctr = a.counters + (size_t)t * NUM_COUNTERS;
        if (k == 10 || k == 11 || k == 21) {
            if (v > ctr[k]) ctr[k] = v;
        } else if (v) ctr[k] += v;
    }
    return expanded_new;
}

#ifndef AZD_TU_ASYNC
template <int KW, bool BIG>
__global__ __launch_bounds__(64) void k_rollout(Arenas a, TolTable tol) {
    __shared__ WaveLds s;
    build_action_table(s, a.A);
    rollout_agent<KW>(a, tol, s, 0u, (int)blockIdx.x);
}


#endif // !AZD_TU_ASYNC
static_assert(PRED_CHUNKS == 2, "selection code addresses prediction chunks 0 and 1 explicitly");

// optimizer/mod.rs:226-241: ArgminData.state = roots[tree] with the winner's ActionSet replayed
// (acts commute, ascending order), cost recomputed at full precision.  One wave.
template <int KW>
__device__ void argmin_replay(const Arenas &a, WaveLds &s, const uint32_t dyn, const int wt, const uint32_t win_node,
                              const int init_mode) {
    const int n = a.n, A = a.A;
    build_action_table(s, A);
    if (LANE < PARENTS_STRIDE) s.par[LANE] = a.root_parents[(size_t)wt * PARENTS_STRIDE + LANE];
    uint64_t perm[KW], key[KW];
#pragma unroll
    for (int w = 0; w < KW; ++w) {
        perm[w] = a.root_perm[(size_t)wt * KW + w];
        key[w] = a.keys[((size_t)wt * a.node_cap + win_node) * KW + w];
    }
    WAVE_SYNC();
#pragma unroll
    for (int w = 0; w < KW; ++w) {
        uint64_t bits = key[w];
        while (bits) {
            int b = __ffsll((unsigned long long)bits) - 1;
            bits &= bits - 1;
            do_act<KW>(s, perm, (uint32_t)(w * 64 + b));
        }
    }
    ArgminRec *out = a.argmin;
    double lam = lambda1_wave<true>(pack_tree(s, n), n, dyn);
    int mu = matching_wave(s, n, out->matching);
    float ev = c21_eval(a.eval_slope, lam, mu);
    if (LANE < 32) out->parents[LANE] = LANE < n ? s.par[LANE] : 0;
    if (LANE == 0) {
#pragma unroll
        for (int w = 0; w < 4; ++w) out->permitted[w] = 0;
#pragma unroll
        for (int w = 0; w < KW; ++w) out->permitted[w] = perm[w];
        out->lambda_1 = lam;
        out->matching_size = mu;
        out->eval = ev;
        out->agent = wt;
        out->node = win_node;
        if (!init_mode) atomicAdd(&a.status->improved, 1ull);
    }
}

#ifndef AZD_TU_ASYNC
// optimizer/mod.rs:194-246 par_update_argmmim_data (init_mode = 0) and the argmin of par_new
// (:92-101, init_mode = 1).  One block: a strided scan over agents for the lexicographic min of
// (c, agent) among candidates with c < best (strict; cross-tree ties -> lowest agent, which the
// reference leaves to rayon), then wave 0 replays the winner's ActionSet from its root and
// recomputes the cost (:226-241).
template <int KW, bool BIG>
__global__ __launch_bounds__(1024) void k_argmin(Arenas a, int init_mode) {
    __shared__ unsigned long long s_best[17];
    __shared__ WaveLds s;
    const uint32_t dyn = 0;
    const int tid = threadIdx.x;
    const float best_eval = init_mode ? __int_as_float(0x7f800000) : a.argmin->eval;
    unsigned long long mine = ~0ull;
    for (int t = tid; t < a.B; t += blockDim.x) {
        uint32_t node = a.cand_node[t];
        float c = a.cand_c[t];
        if (node != NONE && a.flags[t] == 0 && c < best_eval) {
            unsigned long long key = ((unsigned long long)ordf(c) << 32) | (unsigned long long)(uint32_t)t;
            mine = key < mine ? key : mine;
        }
    }
    mine = wave_min_u64(mine);
    if ((tid & 63) == 0) s_best[tid >> 6] = mine;
    __syncthreads();
    if (tid < 64) {
        unsigned long long v = tid < (int)(blockDim.x >> 6) ? s_best[tid] : ~0ull;
        v = wave_min_u64(v);
        if (tid == 0) s_best[16] = v;
    }
    __syncthreads();
    const unsigned long long win = s_best[16];
    uint32_t win_node = NONE;
    int wt = -1;
    if (win != ~0ull) {
        wt = (int)(win & 0xFFFFFFFFull);
        win_node = a.cand_node[wt];
    }
    __syncthreads();
    for (int t = tid; t < a.B; t += blockDim.x) a.cand_node[t] = NONE; // num_inspected_nodes = nodes.len()
    if (tid >= 64 || wt < 0) return;
    // ---- single wave from here on (WAVE_SYNC only)
    argmin_replay<KW>(a, s, dyn, wt, win_node, init_mode);
}

// optimizer/mod.rs:262-278: state_vecs <- root vectors; obs/weights zeroed then filled by
// SearchTree::write_observations (tree/mod.rs:242-264); h_sa = c_child* (04-c21-tree.rs:104)
template <int KW>
__global__ __launch_bounds__(64) void k_observe(Arenas a, uint32_t n_obs_tol) {
    __shared__ WaveLds s;
    const int t = blockIdx.x;
    const int A = a.A;
    build_action_table(s, A);
    if (LANE < PARENTS_STRIDE) s.par[LANE] = a.root_parents[(size_t)t * PARENTS_STRIDE + LANE];
    uint64_t perm[KW];
#pragma unroll
    for (int w = 0; w < KW; ++w) perm[w] = a.root_perm[(size_t)t * KW + w];
    WAVE_SYNC();
    write_state_vec<KW>(s, perm, A, a.state_vecs + (size_t)t * a.S);
    float *obs = a.obs + (size_t)t * A, *wts = a.weights + (size_t)t * A;
    for (int i = LANE; i < A; i += 64) {
        obs[i] = 0.f;
        wts[i] = 0.f;
    }
    WAVE_SYNC();
    const NodeRec *nodes = a.nodes + (size_t)t * a.node_cap;
    const PredRec *preds = a.preds + (size_t)t * a.pred_cap;
    const NodeRec root = nodes[0];
    const uint32_t nact = root.act_end - root.act_begin;
    for (uint32_t idx = LANE; idx < nact; idx += 64) {
        PredRec p = preds[root.act_begin + idx];
        if (p.arc != NONE) {
            NodeRec cr = nodes[p.child];
            if (!node_active(cr) || cr.n_t >= n_obs_tol) {
                obs[p.a_id] = cr.c_star;
                wts[p.a_id] = 1.0f;
            }
        }
    }
}

#endif // !AZD_TU_ASYNC
// fixed prediction stream h(agent, call, a) = top 24 bits of key4(seed ^ "pred", agent, call, a) * 2^-24
__device__ __forceinline__ uint64_t splitmix(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
#ifndef AZD_TU_ASYNC
__global__ void k_hash_predictions(float *out, int batch, int action_dim, uint64_t seed, uint64_t first_agent,
                                   uint64_t call) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t total = (size_t)batch * action_dim;
    if (i >= total) return;
    uint64_t agent = first_agent + i / action_dim, act = i % action_dim;
    uint64_t r = splitmix(splitmix(splitmix(splitmix(seed ^ 0x70726564ull) ^ agent) ^ call) ^ act);
    out[i] = (float)(r >> 40) * (1.0f / 16777216.0f);
}

#endif // !AZD_TU_ASYNC
// BIG is a spare specialisation flag (n > 19); the kernels no longer depend on it
#define DISPATCH_KW(A, FN, ...)                                   \
    switch ((A).KW) {                                             \
    case 1: FN<1, false>(__VA_ARGS__); break;                     \
    case 2: FN<2, false>(__VA_ARGS__); break;                     \
    case 3:                                                       \
        if ((A).n > 19) FN<3, true>(__VA_ARGS__);                 \
        else FN<3, false>(__VA_ARGS__);                           \
        break;                                                    \
    default: FN<4, true>(__VA_ARGS__); break;                     \
    }

#include "persistent_step.inc"
#ifdef AZD_TU_ASYNC
#include "async_step.inc"
template <int KW, bool BIG>
static void l_async(const Arenas &a, const PersistArgs *d_args, int n_calls, unsigned long long *log_key, float *act_scratch,
                    const float *params, uint32_t dyn_stride, size_t dyn_bytes, hipStream_t st) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void *)k_async<KW>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 16 * 1024);
        attr_set = true;
    }
    const int n_wg = (a.B + PERSIST_WAVES - 1) / PERSIST_WAVES;
    (void)hipMemsetAsync(log_key, 0xFF, (size_t)n_calls * sizeof(unsigned long long), st);
    k_async<KW><<<dim3(n_wg), dim3(PERSIST_WAVES * 64), dyn_bytes, st>>>(d_args, n_calls, log_key, act_scratch, dyn_stride, params, a.state_vecs, a.h_theta);
    k_argmin_log1<KW, BIG><<<dim3(1), dim3(64), dyn_lds_bytes(a.n), st>>>(a, n_calls, log_key);
}
void launch_async(const Arenas &a, const PersistArgs *d_args, int n_calls, unsigned long long *log_key, float *act_scratch,
                  const float *params, uint32_t dyn_stride, size_t dyn_bytes, void *stream) {
    DISPATCH_KW(a, l_async, a, d_args, n_calls, log_key, act_scratch, params, dyn_stride, dyn_bytes, (hipStream_t)stream);
}
// LDS plan of the asynchronous step (no evaluator buffers in LDS)
bool async_plan(const Arenas &a, const FusedEval &ev, uint32_t *dyn_stride, size_t *dyn_bytes) {
    (void)ev;
    if (a.B > 65536 || a.node_cap > 65536) return false; // (agent, node) are packed 16 + 16 bits in the argmin log
    size_t stride = (dyn_lds_bytes(a.n) + 15) & ~(size_t)15;
    size_t total = stride * PERSIST_WAVES;
    const size_t static_lds = PERSIST_WAVES * (sizeof(WaveLds) + 16) + sizeof(AsyncCtl) + 256;
    if (total + static_lds > 160 * 1024) return false;
    *dyn_stride = (uint32_t)stride;
    *dyn_bytes = total;
    return true;
}
#else
#include "root_policy.inc"

// parity probe for the f32 primitives the selection rule depends on; four outputs per input pair:
//   [0] the kernel's own sqrt(|x - y|) (azd_sqrt)   [1] sqrtf   [2] __fsqrt_rn   [3] x - (x - y)
__global__ void k_probe_math(const float *in, float *out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float x = in[2 * i], y = in[2 * i + 1];
    float d = fabsf(x - y);
    out[4 * i] = azd_sqrt(d);
    out[4 * i + 1] = sqrtf(d);
    out[4 * i + 2] = __fsqrt_rn(d);
    float g = x - y;
    out[4 * i + 3] = x - g;
}

// lambda_1 / matching probe: one wave per tree, `reps` repetitions (timing), result of the last one
__global__ __launch_bounds__(64) void k_probe_cost(const uint8_t *__restrict__ parents, int n, int count, int reps,
                                                   int full, double *__restrict__ lam_out, int *__restrict__ mu_out) {
    __shared__ WaveLds s;
    const uint32_t dyn = 0;
    const int t = blockIdx.x;
    if (t >= count) return;
    if (LANE < PARENTS_STRIDE) s.par[LANE] = LANE < n ? parents[(size_t)t * n + LANE] : 0;
    WAVE_SYNC();
    double lam = 0.0;
    int mu = 0;
    for (int r = 0; r < reps; ++r) {
        const PackedTree pt = pack_tree(s, n);
        lam = full ? lambda1_wave<true>(pt, n, dyn) : lambda1_wave<false>(pt, n, dyn);
        mu = full ? matching_wave(s, n, nullptr) : matching_size_wave(pt, n);
        WAVE_SYNC();
    }
    if (LANE == 0) {
        lam_out[t] = lam;
        mu_out[t] = mu;
    }
}

// ---------------------------------------------------------------- launchers
template <int KW, bool BIG>
static void l_init_roots(const Arenas &a, const uint8_t *p, const uint64_t *m, hipStream_t st) {
    k_init_roots<KW, BIG><<<dim3(a.B), dim3(64), dyn_lds_bytes(a.n), st>>>(a, p, m);
}
template <int KW, bool BIG>
static void l_add_actions(const Arenas &a, int root_mode, hipStream_t st) {
    k_add_actions<KW><<<dim3(a.B), dim3(64), dyn_lds_bytes(a.n), st>>>(a, root_mode);
}
template <int KW, bool BIG>
static void l_rollout(const Arenas &a, const TolTable &tol, hipStream_t st) {
    k_rollout<KW, BIG><<<dim3(a.B), dim3(64), dyn_lds_bytes(a.n), st>>>(a, tol);
}
template <int KW, bool BIG>
static void l_argmin(const Arenas &a, int init_mode, hipStream_t st) {
    k_argmin<KW, BIG><<<dim3(1), dim3(1024), dyn_lds_bytes(a.n), st>>>(a, init_mode);
}
template <int KW, bool BIG>
static void l_observe(const Arenas &a, uint32_t tol, hipStream_t st) {
    k_observe<KW><<<dim3(a.B), dim3(64), dyn_lds_bytes(a.n), st>>>(a, tol);
}

void launch_init_roots(const Arenas &a, const uint8_t *d_parents, const uint64_t *d_permitted, void *stream) {
    DISPATCH_KW(a, l_init_roots, a, d_parents, d_permitted, (hipStream_t)stream);
}
void launch_add_actions(const Arenas &a, int root_mode, void *stream) {
    DISPATCH_KW(a, l_add_actions, a, root_mode, (hipStream_t)stream);
}
void launch_rollout(const Arenas &a, const TolTable &tol, void *stream) {
    DISPATCH_KW(a, l_rollout, a, tol, (hipStream_t)stream);
}
void launch_argmin(const Arenas &a, int init_mode, void *stream) {
    DISPATCH_KW(a, l_argmin, a, init_mode, (hipStream_t)stream);
}
void launch_observe(const Arenas &a, uint32_t n_obs_tol, void *stream) {
    DISPATCH_KW(a, l_observe, a, n_obs_tol, (hipStream_t)stream);
}
template <int KW, bool BIG>
static void l_persist(const Arenas &a, const PersistArgs *d_args, int n_calls, unsigned long long *log_key,
                      uint32_t *log_node, uint32_t dyn_stride, size_t dyn_bytes, hipStream_t st) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void *)k_persist<KW>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 16 * 1024);
        attr_set = true;
    }
    const int n_wg = (a.B + PERSIST_WAVES - 1) / PERSIST_WAVES;
    k_persist<KW><<<dim3(n_wg), dim3(PERSIST_WAVES * 64), dyn_bytes, st>>>(d_args, n_calls, log_key, log_node, dyn_stride);
    k_argmin_log<KW, BIG><<<dim3(1), dim3(64), dyn_lds_bytes(a.n), st>>>(a, n_calls, n_wg, log_key, log_node);
}
// LDS plan of the persistent step; returns false when the workgroup does not fit a CU
bool persist_plan(const Arenas &a, const FusedEval &ev, uint32_t *dyn_stride, size_t *dyn_bytes) {
    size_t stride = (dyn_lds_bytes(a.n) + 15) & ~(size_t)15;
    size_t total = stride * PERSIST_WAVES;
    if (ev.kind == 3) {
        size_t mlp = (size_t)PERSIST_WAVES * ((size_t)(ev.dims[0] + 4) + 2 * (size_t)(ev.max_hidden + 4)) * sizeof(float);
        if (mlp > total) total = mlp;
    }
    const size_t static_lds = PERSIST_WAVES * (sizeof(WaveLds) + 16) + 256;
    if (total + static_lds > 160 * 1024) return false;
    *dyn_stride = (uint32_t)stride;
    *dyn_bytes = total;
    return true;
}
void launch_persist(const Arenas &a, const PersistArgs *d_args, int n_calls, unsigned long long *log_key,
                    uint32_t *log_node, uint32_t dyn_stride, size_t dyn_bytes, void *stream) {
    DISPATCH_KW(a, l_persist, a, d_args, n_calls, log_key, log_node, dyn_stride, dyn_bytes, (hipStream_t)stream);
}
template <int KW, bool BIG>
static void l_modify_roots(const Arenas &a, uint64_t seed, uint64_t epoch, uint64_t first_agent, int kmin, int kmax,
                           uint8_t *d_parents, uint64_t *d_perm, hipStream_t st) {
    k_c21_modify_roots<KW><<<dim3(a.B), dim3(64), dyn_lds_bytes(a.n), st>>>(a, seed, epoch, first_agent, kmin, kmax, d_parents, d_perm);
}
void launch_c21_modify_roots(const Arenas &a, uint64_t seed, uint64_t epoch, uint64_t first_agent, int kmin, int kmax,
                             uint8_t *d_parents, uint64_t *d_perm, void *stream) {
    DISPATCH_KW(a, l_modify_roots, a, seed, epoch, first_agent, kmin, kmax, d_parents, d_perm, (hipStream_t)stream);
}
void launch_hash_predictions(float *d_out, int batch, int action_dim, uint64_t seed, uint64_t first_agent,
                             uint64_t call, void *stream) {
    size_t total = (size_t)batch * action_dim;
    int threads = 256;
    int blocks = (int)((total + threads - 1) / threads);
    hipLaunchKernelGGL(k_hash_predictions, dim3(blocks), dim3(threads), 0, (hipStream_t)stream, d_out, batch, action_dim,
                       seed, first_agent, call);
}
void launch_probe_cost(const uint8_t *d_parents, int n, int count, int reps, int full, double *d_lam, int *d_mu,
                       void *stream) {
    k_probe_cost<<<dim3(count), dim3(64), dyn_lds_bytes(n), (hipStream_t)stream>>>(d_parents, n, count, reps, full, d_lam, d_mu);
}
void launch_probe_math(const float *d_in, float *d_out, int n, void *stream) {
    hipLaunchKernelGGL(k_probe_math, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, d_in, d_out, n);
}

#endif // AZD_TU_ASYNC

} // namespace azd
