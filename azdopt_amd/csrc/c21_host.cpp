// c21_host.cpp -- host side of the c21 space seam: dimensions and the seeded root generator
// that stands in for the driver's `init_states` closure (graph-state/examples/04-c21-tree.rs:108-112,
// graph-state/src/rooted_tree/mod.rs:14-20, modify_parent_once.rs:14-25; the reference draws from
// thread_rng, this build from a counter-based generator so that runs are reproducible and
// shard-invariant -- the stream of an agent depends only on (seed, epoch, global agent id)).
#include "c21_host.h"

#include <cmath>
#include <vector>

namespace azd {

uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
uint64_t stream_key(uint64_t seed, uint64_t domain, uint64_t agent, uint64_t draw) {
    return splitmix64(splitmix64(splitmix64(splitmix64(seed) ^ domain) ^ agent) ^ draw);
}
uint32_t draw_below(uint64_t r, uint32_t n) { return (uint32_t)(((r >> 32) * (uint64_t)n) >> 32); }

int c21_state_dim(int n) { return (n - 1) * (n - 2) - 2; }
int c21_action_dim(int n) { return (n - 1) * (n - 2) / 2 - 1; }
int c21_key_words(int n) { return (c21_action_dim(n) + 63) / 64; }

float c21_eval_slope(int n) { // 04-c21-tree.rs:58-74
    int r = 0;
    while ((r + 1) * (r + 1) <= n - 1) ++r;
    int sq = (r * r == n - 1) ? r : r + 1;
    int upper = sq + (n + 1) / 2;
    return 1.0f / (float)(upper - 2);
}

// initial bracket of the lambda_1 multisection: path (smallest lambda_1 among trees on n vertices, 2 cos(pi / (n + 1)))
// and star (largest, sqrt(n - 1)), rounded to f32 and widened by 2^-20 -- the same doubles as the oracle's
void c21_lambda_bracket(int n, double *lo, double *hi) {
    *lo = (double)(float)(2.0 * std::cos(3.14159265358979323846 / (double)(n + 1))) - 0x1p-20;
    *hi = (double)(float)std::sqrt((double)(n - 1)) + 0x1p-20;
}

void c21_shuffle_permitted(uint64_t seed, uint64_t domain, uint64_t agent, int n, int k, uint64_t *permitted) {
    const int A = c21_action_dim(n), KW = c21_key_words(n);
    std::vector<uint32_t> perm((size_t)A);
    for (int i = 0; i < A; ++i) perm[(size_t)i] = (uint32_t)i;
    for (int w = 0; w < KW; ++w) permitted[w] = 0;
    for (int j = 0; j < k; ++j) {
        uint32_t r = (uint32_t)j + draw_below(stream_key(seed, domain, agent, 64 + (uint64_t)j), (uint32_t)(A - j));
        uint32_t tmp = perm[(size_t)j];
        perm[(size_t)j] = perm[r];
        perm[r] = tmp;
        permitted[perm[(size_t)j] >> 6] |= 1ull << (perm[(size_t)j] & 63);
    }
}

void c21_fresh_root(uint64_t seed, uint64_t domain, uint64_t agent, int n, int k, uint8_t *parents,
                    uint64_t *permitted) {
    for (int v = 0; v < n; ++v) parents[v] = 0;
    for (int v = 2; v <= n - 2; ++v) parents[v] = (uint8_t)draw_below(stream_key(seed, domain, agent, (uint64_t)v), (uint32_t)v);
    c21_shuffle_permitted(seed, domain, agent, n, k, permitted);
}

void c21_generate_roots(uint64_t seed, uint64_t epoch, uint64_t first_agent, int count, int n, int kmin, int kmax,
                        uint8_t *parents, uint64_t *permitted) {
    const int KW = c21_key_words(n);
    const uint64_t domain = DOMAIN_ROOT ^ (epoch << 32);
    for (int i = 0; i < count; ++i) {
        uint64_t agent = first_agent + (uint64_t)i;
        int k = kmin + (int)draw_below(stream_key(seed, domain, agent, 0), (uint32_t)(kmax - kmin + 1));
        c21_fresh_root(seed, domain, agent, n, k, parents + (size_t)i * n, permitted + (size_t)i * KW);
    }
}

// ---- Ramsey space: seeded stand-in for the drivers' init_state closure (01-r333.rs:84-90,
// 02-r44.rs:84-90: ColoredCompleteBitsetGraph::generate with uniform colour weights +
// RamseyCountsNoRecolor::generate).  colour[e] = below(draw 1024 + e, C); permitted edges = first k of
// the Fisher-Yates shuffle of 0..E-1 (draws 64 + j).
int ramsey_edges(int n) { return n * (n - 1) / 2; }
int ramsey_state_dim(int n, int c) { return ramsey_edges(n) * (2 * c + 1); }
int ramsey_action_dim(int n, int c) { return ramsey_edges(n) * c; }
int ramsey_key_words(int n, int c) { return (ramsey_action_dim(n, c) + 63) / 64; }

void shuffle_mask(uint64_t seed, uint64_t domain, uint64_t agent, int universe, int words, int k, uint64_t *mask) {
    std::vector<uint32_t> perm((size_t)universe);
    for (int i = 0; i < universe; ++i) perm[(size_t)i] = (uint32_t)i;
    for (int w = 0; w < words; ++w) mask[w] = 0;
    for (int j = 0; j < k; ++j) {
        uint32_t r = (uint32_t)j + draw_below(stream_key(seed, domain, agent, 64 + (uint64_t)j), (uint32_t)(universe - j));
        uint32_t tmp = perm[(size_t)j];
        perm[(size_t)j] = perm[r];
        perm[r] = tmp;
        mask[perm[(size_t)j] >> 6] |= 1ull << (perm[(size_t)j] & 63);
    }
}
void ramsey_fresh_root(uint64_t seed, uint64_t domain, uint64_t agent, int n, int c, int k, uint8_t *colors,
                       uint64_t *permitted) {
    const int E = ramsey_edges(n);
    for (int e = 0; e < E; ++e) colors[e] = (uint8_t)draw_below(stream_key(seed, domain, agent, 1024 + (uint64_t)e), (uint32_t)c);
    shuffle_mask(seed, domain, agent, E, ramsey_key_words(n, c), k, permitted);
}
void ramsey_generate_roots(uint64_t seed, uint64_t epoch, uint64_t first_agent, int count, int n, int c, int kmin,
                           int kmax, uint8_t *colors, uint64_t *permitted) {
    const int E = ramsey_edges(n), KW = ramsey_key_words(n, c);
    const uint64_t domain = DOMAIN_ROOT ^ (epoch << 32);
    for (int i = 0; i < count; ++i) {
        uint64_t agent = first_agent + (uint64_t)i;
        int k = kmin + (int)draw_below(stream_key(seed, domain, agent, 0), (uint32_t)(kmax - kmin + 1));
        ramsey_fresh_root(seed, domain, agent, n, c, k, colors + (size_t)i * E, permitted + (size_t)i * KW);
    }
}

// ---- dense-graph space: seeded stand-in for ConnectedBitsetGraph::generate(p) (connected_bitset_graph/mod.rs:84-97:
// redraw G(n, p) until it is connected) + a modifiable-slot set in the image of modify_parent_once.rs:14-25.
//   attempt t = 0, 1, ...: the edge at slot e is present iff (draw(4096 + t E + e) >> 40) < p24 (p24 = p 2^24);
//   modifiable slots = first k of the Fisher-Yates shuffle of 0..E-1 (draws 64 + j)
int dense_edges(int n) { return n * (n - 1) / 2; }
int dense_state_dim(int n) { return 3 * dense_edges(n) + 1; }
int dense_action_dim(int n) { return 2 * dense_edges(n); }
int dense_key_words(int n) { return (dense_action_dim(n) + 63) / 64; }
bool dense_connected(const uint64_t *adj, int n) {
    uint64_t seen = 1ull, frontier = 1ull;
    const uint64_t all = n >= 64 ? ~0ull : ((1ull << n) - 1ull);
    while (frontier) {
        uint64_t next = 0;
        while (frontier) {
            const int w = __builtin_ctzll(frontier);
            frontier &= frontier - 1;
            next |= adj[w];
        }
        frontier = next & ~seen;
        seen |= next;
    }
    return (seen & all) == all;
}
void dense_generate_roots(uint64_t seed, uint64_t epoch, uint64_t first_agent, int count, int n, int kmin, int kmax,
                          uint32_t p24, uint64_t *adj_out, uint64_t *slots) {
    const int E = dense_edges(n), KW = dense_key_words(n);
    const uint64_t domain = DOMAIN_ROOT ^ (epoch << 32);
    for (int i = 0; i < count; ++i) {
        const uint64_t agent = first_agent + (uint64_t)i;
        const int k = kmin + (int)draw_below(stream_key(seed, domain, agent, 0), (uint32_t)(kmax - kmin + 1));
        uint64_t *adj = adj_out + (size_t)i * n;
        for (uint64_t t = 0;; ++t) {
            for (int v = 0; v < n; ++v) adj[v] = 0;
            int e = 0;
            for (int v = 1; v < n; ++v)
                for (int u = 0; u < v; ++u, ++e)
                    if ((uint32_t)(stream_key(seed, domain, agent, 4096ull + t * (uint64_t)E + (uint64_t)e) >> 40) < p24) {
                        adj[v] |= 1ull << u;
                        adj[u] |= 1ull << v;
                    }
            if (dense_connected(adj, n)) break;
        }
        uint64_t *so = slots + (size_t)i * KW;
        for (int w = 0; w < KW; ++w) so[w] = 0;
        shuffle_mask(seed, domain, agent, E, (E + 63) / 64, k, so);
    }
}

} // namespace azd
