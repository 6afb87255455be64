// ThreadSanitizer probe of the header's parallel host builders (merge sort, task-parallel node build): see tools/sanitize_cpu.sh.
#define RAKAU_AMD_DROP_IN
#include "../../include/rakau_amd/tree.hpp"
#include <random>
#include <cstdio>
using namespace rakau; using namespace rakau::kwargs;
int main() {
    std::mt19937 rng(1);
    const std::size_t n = 300000;
    std::vector<float> x(n), y(n), z(n), m(n);
    std::normal_distribution<float> g(0.f, 1.f); std::uniform_real_distribution<float> u(0.1f, 1.f);
    for (std::size_t i = 0; i < n; ++i) { x[i] = g(rng); y[i] = g(rng); z[i] = g(rng); m[i] = u(rng); }
    octree<float> t{x_coords = x, y_coords = y, z_coords = z, masses = m};
    quadtree<float, mac::bh_geom> q{x_coords = x, y_coords = y, masses = m};
    t.update_particles_u([n](const auto &its) { for (std::size_t i = 0; i < n; ++i) its[0][i] *= 0.99f; });
    std::printf("nodes %zu %zu\n", t.nodes().size(), q.nodes().size());
    return 0;
}
