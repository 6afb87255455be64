// Run-time check of integration/rakau_amd_cuda_bridge.cpp: rakau::cuda_acc_pot_impl (the reference's multi-GPU seam,
// declared by the reference's own cuda_fwd.hpp) called the way tree::acc_pot_impl() calls it (tree.hpp:3150-3222 of the
// reference: split_indices whose first entry lies on a critical-node boundary, full-size or compact outputs), on a tree
// built by the rakau_amd header -- whose node records have the reference's layout. Run with RK_ALIAS_DEVICES=4 on a
// 1-GPU box: four logical devices, every device share bit-identical to what one device computes for the same particles.
// Compiled by tests/test_integration_bridge.py when a checkout of the reference is present (only its headers are read,
// at compile time); the binary needs a GPU to run.
#include <algorithm>
#include <array>
#include <cstdint>
#include <cstdio>
#include <random>
#include <vector>

#include <rakau/detail/cuda_fwd.hpp>

#include "../../include/rakau_amd/tree.hpp"
#include "../../integration/rakau_amd_cuda_bridge.hpp"

template <typename F, typename UInt, rakau::mac RM, rakau_amd::mac AM>
static int run(bool announce)
{
    using ref_node = rakau::tree_node_t<3, F, UInt, RM>;
    using our_tree = rakau_amd::tree<3, F, UInt, AM>;
    static_assert(sizeof(ref_node) == sizeof(rakau_amd::tree_node_t<3, F, UInt, AM>), "node record layouts differ");
    using size_type = rakau::tree_size_t<F>;
    const std::size_t n = 30000;
    std::mt19937 rng(11);
    std::uniform_real_distribution<F> u(F(-1), F(1)), um(F(0.1), F(1));
    std::vector<F> x(n), y(n), z(n), m(n);
    for (std::size_t i = 0; i < n; ++i) {
        x[i] = u(rng), y[i] = u(rng), z[i] = u(rng), m[i] = um(rng);
    }
    namespace kw = rakau_amd::kwargs;
    const std::size_t ncrit = announce ? 100 : rakau_amd::default_ncrit;
    our_tree t{kw::x_coords = x, kw::y_coords = y, kw::z_coords = z, kw::masses = m, kw::ncrit = ncrit};
    const auto parts = t.p_its_u();
    const auto *nodes = reinterpret_cast<const ref_node *>(t.nodes().data());
    const size_type n_nodes = t.nodes().size();
    const auto *codes = reinterpret_cast<const UInt *>(t.c_it_u());
    if (announce) {
        // What the patched reference tree does in rocm_init_state(), INTEGRATION.md section B.
        rakau::rakau_amd_tree_ready(nodes, t.ncrit());
    }
    const auto &cn = t.crit_nodes();
    // tree.hpp:3150-3185: the first index on a critical-node boundary, the others wherever the fractions fall.
    const size_type s0 = cn[cn.size() / 5].begin;
    const std::vector<size_type> split4{s0, s0 + (n - s0) / 4 + 3, s0 + (n - s0) / 2 + 1, s0 + 3 * (n - s0) / 4 + 7, n};
    const std::vector<size_type> split1{s0, n};
    const F theta = F(0.6), mac_value = AM == rakau_amd::mac::bh ? F(1) / (theta * theta) : F(1) / theta, G = F(1.5), eps2 = F(1e-4);
    int bad = 0;
    // Q = 2, offset outputs: one device against four.
    std::array<std::vector<F>, 4> one, four, compact;
    for (auto &v : one) v.assign(n, F(-1));
    for (auto &v : four) v.assign(n, F(-1));
    for (auto &v : compact) v.assign(n - s0, F(-1));
    auto ptrs4 = [](std::array<std::vector<F>, 4> &a) { return std::array<F *, 4>{a[0].data(), a[1].data(), a[2].data(), a[3].data()}; };
    rakau::cuda_acc_pot_impl<2u, 3u, F, UInt, RM>(ptrs4(one), split1, nodes, n_nodes, parts, codes, n, mac_value, G, eps2, true);
    rakau::cuda_acc_pot_impl<2u, 3u, F, UInt, RM>(ptrs4(four), split4, nodes, n_nodes, parts, codes, n, mac_value, G, eps2, true);
    // Compact outputs (the non-pointer iterator branch of tree.hpp:3211-3222): element 0 is particle split_indices[0].
    rakau::cuda_acc_pot_impl<2u, 3u, F, UInt, RM>(ptrs4(compact), split4, nodes, n_nodes, parts, codes, n, mac_value, G, eps2, false);
    // The header's own traversal of the same tree (same engine underneath; eps = sqrt(eps2) squared again may differ
    // from eps2 by an ulp, hence a tolerance here and exact comparisons between the calls of the seam).
    std::array<std::vector<F>, 4> full;
    t.accs_pots_u(full, theta, kw::G = G, kw::eps = std::sqrt(eps2));
    for (std::size_t i = 0; i < n; ++i) {
        for (int k = 0; k < 4; ++k) {
            if (i < s0) {
                bad += one[k][i] != F(-1);
                bad += four[k][i] != F(-1);
            } else {
                bad += one[k][i] != four[k][i];
                bad += compact[k][i - s0] != four[k][i];
                bad += !(std::abs(four[k][i] - full[k][i]) <= F(1e-4) * std::abs(full[k][i]));
            }
        }
    }
    // Q = 0 and Q = 1 agree with Q = 2 bit for bit (the engine uses the same expressions).
    std::array<std::vector<F>, 3> acc;
    for (auto &v : acc) v.assign(n, F(0));
    rakau::cuda_acc_pot_impl<0u, 3u, F, UInt, RM>(std::array<F *, 3>{acc[0].data(), acc[1].data(), acc[2].data()}, split4, nodes, n_nodes,
                                                  parts, codes, n, mac_value, G, eps2, true);
    std::vector<F> pot(n, F(0));
    rakau::cuda_acc_pot_impl<1u, 3u, F, UInt, RM>(std::array<F *, 1>{pot.data()}, split4, nodes, n_nodes, parts, codes, n, mac_value, G,
                                                  eps2, true);
    for (std::size_t i = s0; i < n; ++i) {
        for (int k = 0; k < 3; ++k) bad += acc[k][i] != four[k][i];
        bad += pot[i] != four[3][i];
    }
    // A split whose devices all end up with nothing after the first index, and one that gives everything to device 2.
    const std::vector<size_type> split_last{n, n, n};
    rakau::cuda_acc_pot_impl<1u, 3u, F, UInt, RM>(std::array<F *, 1>{pot.data()}, split_last, nodes, n_nodes, parts, codes, n, mac_value, G,
                                                  eps2, true);
    std::vector<F> pot2(n, F(0));
    const std::vector<size_type> split_dev2{s0, s0, s0, n};
    rakau::cuda_acc_pot_impl<1u, 3u, F, UInt, RM>(std::array<F *, 1>{pot2.data()}, split_dev2, nodes, n_nodes, parts, codes, n, mac_value,
                                                  G, eps2, true);
    for (std::size_t i = s0; i < n; ++i) bad += pot2[i] != pot[i];
    // First index off a critical-node boundary: the engine refuses (and names ncrit) instead of computing other groups.
    bool threw = false;
    try {
        const std::vector<size_type> off{s0 + 1, n};
        rakau::cuda_acc_pot_impl<1u, 3u, F, UInt, RM>(std::array<F *, 1>{pot2.data()}, off, nodes, n_nodes, parts, codes, n, mac_value, G,
                                                      eps2, true);
    } catch (const std::invalid_argument &) {
        threw = true;
    }
    bad += !threw;
    // More accelerators than devices (tree.hpp:3135-3141).
    threw = false;
    try {
        const std::vector<size_type> many(rakau::cuda_device_count() + 2u, n);
        rakau::cuda_acc_pot_impl<1u, 3u, F, UInt, RM>(std::array<F *, 1>{pot2.data()}, many, nodes, n_nodes, parts, codes, n, mac_value, G,
                                                      eps2, true);
    } catch (const std::invalid_argument &) {
        threw = true;
    }
    bad += !threw;
    if (announce) {
        // The particles change in place (update_particles_u keeps the arrays' addresses): rocm_reset_state() ->
        // rakau_amd_invalidate, rocm_init_state() -> rakau_amd_tree_ready. The results must follow the new masses.
        rakau::rakau_amd_invalidate(nodes);
        t.update_masses_u([n](auto m_it) {
            for (std::size_t i = 0; i < n; ++i) {
                m_it[i] *= F(2);
            }
        });
        const auto parts2 = t.p_its_u();
        const auto *nodes2 = reinterpret_cast<const ref_node *>(t.nodes().data());
        rakau::rakau_amd_tree_ready(nodes2, t.ncrit());
        std::vector<F> pot3(n, F(0));
        rakau::cuda_acc_pot_impl<1u, 3u, F, UInt, RM>(std::array<F *, 1>{pot3.data()}, split4, nodes2, t.nodes().size(), parts2, codes, n,
                                                      mac_value, G, eps2, true);
        // Every pair's product of masses is four times what it was.
        for (std::size_t i = s0; i < n; ++i) bad += !(std::abs(pot3[i] - F(4) * pot[i]) <= F(1e-4) * std::abs(pot3[i]));
        rakau::rakau_amd_invalidate(nodes2);
    }
    return bad;
}

// `cuda_bridge_driver timing <n>`: what a call through the stateless seam costs with and without the life-time hooks (one
// device; uniform particles, theta 0.75, accelerations into pageable arrays): INTEGRATION.md section B.2.
#include <chrono>
#include <cstdlib>
#include <cstring>
static int timing(std::size_t n)
{
    using F = float;
    using ref_node = rakau::tree_node_t<3, F, std::uint64_t, rakau::mac::bh>;
    using size_type = rakau::tree_size_t<F>;
    std::mt19937 rng(11);
    std::uniform_real_distribution<F> u(F(-1), F(1)), um(F(0.1), F(1));
    std::vector<F> x(n), y(n), z(n), m(n);
    for (std::size_t i = 0; i < n; ++i) {
        x[i] = u(rng), y[i] = u(rng), z[i] = u(rng), m[i] = um(rng);
    }
    namespace kw = rakau_amd::kwargs;
    rakau_amd::tree<3, F, std::uint64_t, rakau_amd::mac::bh> t{kw::x_coords = x, kw::y_coords = y, kw::z_coords = z, kw::masses = m};
    const auto parts = t.p_its_u();
    const auto *nodes = reinterpret_cast<const ref_node *>(t.nodes().data());
    const size_type n_nodes = t.nodes().size();
    const auto *codes = reinterpret_cast<const std::uint64_t *>(t.c_it_u());
    std::array<std::vector<F>, 3> acc;
    for (auto &v : acc) v.assign(n, F(0));
    const std::array<F *, 3> out{acc[0].data(), acc[1].data(), acc[2].data()};
    const std::vector<size_type> split{0, n};
    const F mac_value = F(1) / (F(0.75) * F(0.75));
    auto call_ms = [&]() {
        const auto t0 = std::chrono::steady_clock::now();
        rakau::cuda_acc_pot_impl<0u, 3u, F, std::uint64_t, rakau::mac::bh>(out, split, nodes, n_nodes, parts, codes, n, mac_value, F(1), F(0), true);
        return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    };
    std::printf("n = %zu, %zu nodes, one device\n", n, static_cast<std::size_t>(n_nodes));
    std::printf("unannounced tree (state built from the arguments and destroyed, every call), ms:");
    for (int i = 0; i < 5; ++i) std::printf(" %.2f", call_ms());
    rakau::rakau_amd_tree_ready(nodes, t.ncrit());
    std::printf("\nannounced tree (rakau_amd_tree_ready: state resident between calls), ms:");
    for (int i = 0; i < 8; ++i) std::printf(" %.2f", call_ms());
    rakau::rakau_amd_invalidate(nodes);
    std::printf("\n");
    return 0;
}

int main(int argc, char **argv)
{
    if (argc > 2 && !std::strcmp(argv[1], "timing")) {
        return timing(static_cast<std::size_t>(std::atoll(argv[2])));
    }
    if (rakau::cuda_device_count() < 4u) {
        std::printf("cuda bridge: %u device(s); run with RK_ALIAS_DEVICES=4\n", rakau::cuda_device_count());
        return 2;
    }
    int bad = rakau::cuda_min_size() != 64u;
    bad += run<float, std::uint64_t, rakau::mac::bh, rakau_amd::mac::bh>(true);
    bad += run<float, std::uint64_t, rakau::mac::bh, rakau_amd::mac::bh>(false);
    bad += run<double, std::uint64_t, rakau::mac::bh_geom, rakau_amd::mac::bh_geom>(true);
    bad += run<float, std::uint32_t, rakau::mac::bh, rakau_amd::mac::bh>(false);
    std::printf("cuda bridge checks: %d failure(s)\n", bad);
    return bad != 0;
}
