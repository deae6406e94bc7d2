// c21_host.h -- host-side helpers of the c21 space seam (see c21_host.cpp)
#pragma once
#include <stdint.h>

namespace azd {

constexpr uint64_t DOMAIN_ROOT = 0x726f6f74ull;  // "root"
constexpr uint64_t DOMAIN_PRED = 0x70726564ull;  // "pred"
constexpr uint64_t DOMAIN_RESET = 0x72657365ull; // "rese"

uint64_t splitmix64(uint64_t x);
uint64_t stream_key(uint64_t seed, uint64_t domain, uint64_t agent, uint64_t draw);
uint32_t draw_below(uint64_t r, uint32_t n);

int c21_state_dim(int n);
int c21_action_dim(int n);
int c21_key_words(int n);
float c21_eval_slope(int n);
void c21_lambda_bracket(int n, double *lo, double *hi);

void c21_shuffle_permitted(uint64_t seed, uint64_t domain, uint64_t agent, int n, int k, uint64_t *permitted);
void c21_fresh_root(uint64_t seed, uint64_t domain, uint64_t agent, int n, int k, uint8_t *parents,
                    uint64_t *permitted);
void c21_generate_roots(uint64_t seed, uint64_t epoch, uint64_t first_agent, int count, int n, int kmin, int kmax,
                        uint8_t *parents, uint64_t *permitted);

// Ramsey space (ramsey_counts/space.rs:40-42): E = N(N-1)/2, ACTION = E*C, STATE = E*(2C+1)
int ramsey_edges(int n);
int ramsey_state_dim(int n, int c);
int ramsey_action_dim(int n, int c);
int ramsey_key_words(int n, int c);
void shuffle_mask(uint64_t seed, uint64_t domain, uint64_t agent, int universe, int words, int k, uint64_t *mask);
void ramsey_fresh_root(uint64_t seed, uint64_t domain, uint64_t agent, int n, int c, int k, uint8_t *colors,
                       uint64_t *permitted);
void ramsey_generate_roots(uint64_t seed, uint64_t epoch, uint64_t first_agent, int count, int n, int c, int kmin,
                           int kmax, uint8_t *colors, uint64_t *permitted);

// dense-graph space (space_dense.inc; oracle/dense_graph.inc): E = N(N-1)/2 slots, ACTION = 2E (AddOrDeleteEdge,
// bitset_graph/space/action.rs:10-27), STATE = 3E + 1 (= E + ACTION + 1, 05-ah.rs:39-40)
int dense_edges(int n);
int dense_state_dim(int n);
int dense_action_dim(int n);
int dense_key_words(int n); // u64 words of an action-id set (host-visible keys, root slot masks)
bool dense_connected(const uint64_t *adj, int n);
void dense_generate_roots(uint64_t seed, uint64_t epoch, uint64_t first_agent, int count, int n, int kmin, int kmax,
                          uint32_t p24, uint64_t *adj, uint64_t *slots);

} // namespace azd
