// Test program: a NablaModel on the HOST side of the boundary (azdopt::HostModel) drives the engine call by call through
// include/azdopt_amd.hpp; tests/test_gpu_examples.py runs the same loop through the Python host's *_begin / *_end calls.
#include <cstdio>
#include <cstdlib>

#include "azdopt_amd.hpp"

namespace {
// predictions and "loss" by integer arithmetic (one f32 division): the same bits from any host
struct CountingModel : azdopt::HostModel {
    int S, A;
    CountingModel(int s, int a) : S(s), A(a) {}
    void write_predictions(int batch, const float *x, float *predictions) override {
        for (int b = 0; b < batch; ++b) {
            int nz = 0;
            for (int i = 0; i < S; ++i) nz += x[(size_t)b * S + i] != 0.f;
            for (int a = 0; a < A; ++a) predictions[(size_t)b * A + a] = (float)((a * 7 + nz * 13) % 97) / 97.0f;
        }
    }
    float update_model(int batch, const float *, const float *, const float *action_weights) override {
        int nz = 0;
        for (size_t i = 0; i < (size_t)batch * A; ++i) nz += action_weights[i] != 0.f;
        return (float)nz;
    }
};
} // namespace

int main(int argc, char **argv) {
    const int batch = argc > 1 ? std::atoi(argv[1]) : 24, calls = argc > 2 ? std::atoi(argv[2]) : 30;
    const uint64_t seed = 2;
    try {
        const azdopt::ROTModifyParentsOnce space(13);
        CountingModel model(space.STATE_DIM(), space.ACTION_DIM());
        const int kmin = 3, kmax = space.ACTION_DIM() / 2;
        auto opt = azdopt::NablaOptimizer<azdopt::ROTModifyParentsOnce>::par_new(space, space.generate_roots(seed, batch, kmin, kmax), model, batch);
        const azdopt::Tolerance tol = {{8, 4, 2}, 1};
        for (int epoch = 0; epoch < 2; ++epoch) {
            const int improved = opt.par_roll_out_episodes(tol, calls);
            const auto a = opt.argmin_data();
            std::printf("improved %d eval %.9g lambda_1 %.17g matching %zu\n", improved, a.eval, a.lambda_1, a.matching.size());
            std::printf("loss %.9g\n", opt.par_update_model(3));
            opt.par_reset_trees_policy(seed, (uint64_t)epoch, kmin, kmax);
        }
        const auto c = opt.counters();
        std::printf("expansions %llu transpositions %llu terminals %llu\n", (unsigned long long)c[AZD_CTR_EXPANSIONS],
                    (unsigned long long)c[AZD_CTR_TRANSPOSITIONS], (unsigned long long)c[AZD_CTR_TERMINALS]);
        // the dense-graph space (BASELINE configs[4]) with a model on the device
        const azdopt::DenseGraphSpace dense(12, 0.3);
        azdopt::ActionModel mlp(batch, dense.STATE_DIM(), dense.ACTION_DIM(), {64}, azdopt::AdamConfig(), seed);
        auto dopt = azdopt::NablaOptimizer<azdopt::DenseGraphSpace>::par_new(dense, dense.generate_roots(seed, batch, 4, 20), mlp, batch);
        const int dimp = dopt.par_roll_out_episodes({{20, 10, 5}, 3}, calls);
        const auto da = dopt.argmin_data();
        std::printf("dense improved %d eval %.9g lambda_1 %.17g matching %d loss %.9g\n", dimp, da.eval, da.lambda_1, da.matching_number,
                    dopt.par_update_model(3));
        // N = 64 (AZD_DENSE_MAX_N): the argmin's open-slot bitmap is (E + 63) / 64 = 32 words, the host key width 63 -- the record
        // must be copied by the former; then the device root policy (modify_root) with up to 200 slots per root
        const azdopt::DenseGraphSpace big(64, 0.15, 200);
        azdopt::TrivialModel triv(big.STATE_DIM(), big.ACTION_DIM());
        auto bopt = azdopt::NablaOptimizer<azdopt::DenseGraphSpace>::par_new(big, big.generate_roots(seed, 4, 150, 200), triv, 4);
        bopt.par_roll_out_episodes({{4, 2}, 1}, 6);
        const auto ba = bopt.argmin_data();
        size_t open = 0;
        for (uint64_t w : ba.permitted) open += (size_t)__builtin_popcountll(w);
        std::printf("dense64 slot words %zu open slots %zu eval %.9g\n", ba.permitted.size(), open, ba.eval);
        bopt.par_reset_trees_policy(seed, 1, 150, 200);
        bopt.par_roll_out_episodes({{4, 2}, 1}, 3);
        std::printf("dense64 after policy eval %.9g\n", bopt.argmin_data().eval);
    } catch (const azdopt::Error &e) {
        std::fprintf(stderr, "azdopt error %d: %s\n", e.status(), e.what());
        return 1;
    }
    return 0;
}
