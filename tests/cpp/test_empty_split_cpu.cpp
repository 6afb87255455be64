// -DRAKAU_AMD_EMPTY_SPLIT_IS_CPU: a call without a `split` argument means what it means in the reference -- no accelerator
// share, the CPU engine (tree.hpp:3114-3117 of the reference) -- instead of this library's default "everything on device 0".
// Runs without a GPU: the results must equal those of the explicit CPU-only split = {1} bit for bit.
#define RAKAU_AMD_DROP_IN
#include <rakau_amd/tree.hpp>

#include <cstdio>
#include <random>
#include <vector>

using namespace rakau;
using namespace rakau::kwargs;

int main()
{
    std::mt19937 rng(5);
    std::uniform_real_distribution<float> d(-1.f, 1.f);
    const std::size_t s = 3000;
    std::vector<float> m(s), x(s), y(s), z(s);
    for (std::size_t i = 0; i < s; ++i) {
        m[i] = 0.5f + 0.5f * (d(rng) + 1.f), x[i] = d(rng), y[i] = d(rng), z[i] = d(rng);
    }
    octree<float> t{x_coords = x.data(), y_coords = y.data(), z_coords = z.data(), masses = m.data(), nparts = s};
    std::vector<float> ax(s), ay(s), az(s), bx(s), by(s), bz(s);
    t.accs_u({ax.data(), ay.data(), az.data()}, 0.75f);                                          // no split: the CPU
    t.accs_u({bx.data(), by.data(), bz.data()}, 0.75f, split = std::vector<double>{1.});         // CPU only, explicitly
    const bool same = ax == bx && ay == by && az == bz;
    std::printf("empty split is the CPU engine: %s\n", same ? "yes" : "NO");
    return same ? 0 : 1;
}
