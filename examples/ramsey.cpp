// The reference's Ramsey drivers (graph-state/examples/01-r333.rs, 02-r44.rs) from a compiled host, over
// include/azdopt_amd.hpp; the same loop and console lines as examples/ramsey.py.
//
//   g++ -O2 -std=c++17 -Iinclude examples/ramsey.cpp -o examples/ramsey -Lazdopt_amd -lazdopt_amd -Wl,-rpath,'$ORIGIN/../azdopt_amd'
//   examples/ramsey r333|r44 [epochs 250] [episodes 0 = the driver's] [batch 0 = the driver's] [stride 1] [seed 0] [hidden ...]
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "azdopt_amd.hpp"

namespace {
struct Driver { // N, SIZES, BATCH, episodes, n_as_tol, num_permitted_edges_range.start
    const char *name;
    int n;
    std::vector<int> sizes;
    int batch, episodes, kmin;
    azdopt::Tolerance tol;
};
const Driver DRIVERS[] = {
    {"r333", 16, {3, 3, 3}, 256, 6400, 10, {{200, 200, 200, 100, 100, 100, 50, 50, 50, 25, 25, 25}, 10}}, // 01-r333.rs:35-38,61,83,126-130
    {"r44", 17, {4, 4}, 512, 3200, 12, {{200, 200, 100, 100, 50, 50, 25, 25}, 10}},                       // 02-r44.rs:35-38,61,83,126-130
};

bool process_argmin(const azdopt::RamseyArgmin &a) {
    std::printf("%.9g\tTotalCounts([", a.eval);
    for (size_t c = 0; c < a.clique_counts.size(); ++c) std::printf("%s%d", c ? ", " : "", a.clique_counts[c]);
    std::printf("])\n");
    if (a.eval == 0.f) {
        std::printf("state is optimal\n");
        return true;
    }
    return false;
}
} // namespace

int main(int argc, char **argv) {
    const Driver *d = nullptr;
    for (const Driver &x : DRIVERS)
        if (argc > 1 && std::strcmp(argv[1], x.name) == 0) d = &x;
    if (!d) {
        std::fprintf(stderr, "usage: %s r333|r44 [epochs] [episodes] [batch] [stride] [seed] [hidden ...]\n", argv[0]);
        return 2;
    }
    const int epochs = argc > 2 ? std::atoi(argv[2]) : 250;
    const int episodes = argc > 3 && std::atoi(argv[3]) > 0 ? std::atoi(argv[3]) : d->episodes;
    const int batch = argc > 4 && std::atoi(argv[4]) > 0 ? std::atoi(argv[4]) : d->batch;
    const int stride = argc > 5 ? std::atoi(argv[5]) : 1;
    const uint64_t seed = argc > 6 ? std::strtoull(argv[6], nullptr, 10) : 0;
    std::vector<int> hidden;
    for (int i = 7; i < argc; ++i) hidden.push_back(std::atoi(argv[i]));
    if (hidden.empty()) hidden = {512, 1024, 512};
    try {
        const azdopt::RamseySpaceNoEdgeRecolor space(d->n, d->sizes);
        azdopt::ActionModel model(batch, space.STATE_DIM(), space.ACTION_DIM(), hidden, azdopt::AdamConfig(), seed);
        const int C = space.C();
        // ..=(E / 2), capped by what a node holds (128 predictions: (C - 1) per permitted edge)
        const int kmin = d->kmin, kmax = std::min(space.E() / 2, 128 / (C - 1));
        auto opt = azdopt::NablaOptimizer<azdopt::RamseySpaceNoEdgeRecolor>::par_new(
            space, space.generate_roots(seed, batch, kmin, kmax), model, batch, 0, 0, 0, episodes * 2 + 64, episodes * 3 + 64,
            (episodes + 1) * kmax * (C - 1) + 128);
        if (process_argmin(opt.argmin_data())) return 0;
        for (int epoch = 1; epoch <= epochs; ++epoch) {
            std::printf("==== EPOCH: %d ====\n", epoch);
            // the drivers ask for their episodes one call at a time (02-r44.rs:135-143): the epoch's calls are started in one launch
            // and the loop below is answered as they complete (a no-op where the engine cannot do that)
            if (stride < episodes) opt.run_ahead(d->tol, episodes);
            for (int done = 0; done < episodes;) {
                const int k = stride < episodes - done ? stride : episodes - done;
                if (opt.par_roll_out_episodes(d->tol, k) && process_argmin(opt.argmin_data())) return 0;
                done += k;
            }
            std::printf("==== EPISODE: %d ====\n", episodes);
            std::printf("loss: %.9g\n", opt.par_update_model(200));
            opt.par_reset_trees_policy(seed, (uint64_t)epoch, kmin, kmax);
        }
    } catch (const azdopt::Error &e) {
        std::fprintf(stderr, "azdopt error %d: %s\n", e.status(), e.what());
        return 1;
    }
    return 0;
}
