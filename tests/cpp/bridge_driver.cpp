// Run-time check of integration/rakau_amd_bridge.cpp: rakau::rocm_state (the reference's accelerator seam, declared by the
// reference's own rocm_fwd.hpp) constructed and called the way tree::rocm_init_state() / tree::acc_pot_impl() do
// (tree.hpp:1495-1508, 3047-3113 of the reference), on a tree built by the rakau_amd header -- whose node records have
// the reference's layout. Compiled by tests/test_integration_bridge.py when a checkout of the reference is present
// (only its headers are read, at compile time); the binary needs a GPU to run.
#include <algorithm>
#include <array>
#include <cstdint>
#include <cstdio>
#include <random>
#include <vector>

#include <rakau/detail/rocm_fwd.hpp>

#include "../../include/rakau_amd/tree.hpp"
#include "../../integration/rakau_amd_bridge.hpp"

template <typename F, rakau::mac RM, rakau_amd::mac AM>
static int run()
{
    using ref_node = rakau::tree_node_t<3, F, std::uint64_t, RM>;
    using our_tree = rakau_amd::tree<3, F, std::uint64_t, AM>;
    static_assert(sizeof(ref_node) == sizeof(rakau_amd::tree_node_t<3, F, std::uint64_t, AM>), "node record layouts differ");
    const std::size_t n = 20000;
    std::mt19937 rng(7);
    std::uniform_real_distribution<F> u(F(-1), F(1)), um(F(0.1), F(1));
    std::vector<F> x(n), y(n), z(n), m(n);
    for (std::size_t i = 0; i < n; ++i) {
        x[i] = u(rng), y[i] = u(rng), z[i] = u(rng), m[i] = um(rng);
    }
    namespace kw = rakau_amd::kwargs;
    our_tree t{kw::x_coords = x, kw::y_coords = y, kw::z_coords = z, kw::masses = m, kw::ncrit = 100};
    const auto parts = t.p_its_u();
    // What the (patched) reference tree does right before m_rocm.emplace(...), tree.hpp:1504.
    rakau::rakau_amd_set_ncrit(t.ncrit());
    rakau::rocm_state<3, F, std::uint64_t, RM> st(parts, reinterpret_cast<const std::uint64_t *>(t.c_it_u()), static_cast<int>(n),
                                                  reinterpret_cast<const ref_node *>(t.nodes().data()),
                                                  static_cast<int>(t.nodes().size()));
    // A split index on a critical-node boundary (tree.hpp:3053-3063).
    const auto &cn = t.crit_nodes();
    const auto split = static_cast<int>(cn[cn.size() / 2].begin);
    const F theta = F(0.6), mac_value = AM == rakau_amd::mac::bh ? F(1) / (theta * theta) : F(1) / theta, G = F(1.5), eps2 = F(1e-4);
    std::array<std::vector<F>, 4> full;
    t.accs_pots_u(full, theta, kw::G = G, kw::eps = std::sqrt(eps2));
    // offset_output = true: full-size arrays, results written from `split` on; false: compact arrays.
    std::array<std::vector<F>, 4> a, b;
    for (auto &v : a) v.assign(n, F(-1));
    for (auto &v : b) v.assign(n - split, F(-1));
    st.template acc_pot<2>(split, static_cast<int>(n), std::array<F *, 4>{a[0].data(), a[1].data(), a[2].data(), a[3].data()}, mac_value, G,
                           eps2, true);
    st.template acc_pot<2>(split, static_cast<int>(n), std::array<F *, 4>{b[0].data(), b[1].data(), b[2].data(), b[3].data()}, mac_value, G,
                           eps2, false);
    std::array<std::vector<F>, 3> acc;
    for (auto &v : acc) v.assign(n, F(0));
    st.template acc_pot<0>(0, static_cast<int>(n), std::array<F *, 3>{acc[0].data(), acc[1].data(), acc[2].data()}, mac_value, G, eps2, true);
    std::vector<F> pot(n, F(0));
    st.template acc_pot<1>(0, static_cast<int>(n), std::array<F *, 1>{pot.data()}, mac_value, G, eps2, true);
    int bad = 0;
    for (std::size_t i = 0; i < n; ++i) {
        for (int k = 0; k < 4; ++k) {
            if (i < static_cast<std::size_t>(split)) {
                bad += a[k][i] != F(-1);
            } else {
                // eps = sqrt(eps2) squared again may differ from eps2 by an ulp: compare with a tolerance there, exactly
                // between the two calls of the seam.
                bad += a[k][i] != b[k][i - split];
                bad += !(std::abs(a[k][i] - full[k][i]) <= F(1e-4) * std::abs(full[k][i]));
            }
        }
        for (int k = 0; k < 3; ++k) bad += i >= static_cast<std::size_t>(split) && acc[k][i] != a[k][i];
        bad += i >= static_cast<std::size_t>(split) && pot[i] != a[3][i];
    }
    bool threw = false;
    try {
        st.template acc_pot<0>(split + 1, static_cast<int>(n), std::array<F *, 3>{acc[0].data(), acc[1].data(), acc[2].data()}, mac_value, G, eps2,
                               true);
    } catch (const std::invalid_argument &) {
        threw = true; // not a critical-node boundary
    }
    bad += !threw;
    return bad;
}

int main()
{
    if (!rakau::rocm_has_accelerator()) {
        std::puts("bridge: no accelerator");
        return 2;
    }
    int bad = rakau::rocm_min_size() != 64u;
    bad += run<float, rakau::mac::bh, rakau_amd::mac::bh>();
    bad += run<double, rakau::mac::bh_geom, rakau_amd::mac::bh_geom>();
    std::printf("bridge checks: %d failure(s)\n", bad);
    return bad != 0;
}
