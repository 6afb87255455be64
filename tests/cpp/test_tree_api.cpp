// Compile-and-run check of the C++17 front door (include/rakau_amd/tree.hpp): the call shapes of the
// reference's tests (test/basic.cpp, test/readme_example.cpp, test/accuracy_acc.cpp) must compile unchanged
// modulo the namespace, and behave the same. Usage: test_tree_api [gpu]
//   without "gpu": constructor / accessor / error-path checks only (no device needed);
//   with "gpu":    also runs the acc/pot overloads on device 0 and checks them against exact_*.
#define RAKAU_AMD_DROP_IN
#include "../../include/rakau_amd/tree.hpp"

#include <cstdio>
#include <cstring>
#include <iostream>
#include <random>
#include <vector>

using namespace rakau;
using namespace rakau::kwargs;

static int failures = 0;
#define CHECK(cond)                                                                                                    \
    do {                                                                                                               \
        if (!(cond)) {                                                                                                 \
            std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond);                                              \
            ++failures;                                                                                                \
        }                                                                                                              \
    } while (0)
#define CHECK_THROWS(expr, ExcType, needle)                                                                            \
    do {                                                                                                               \
        bool ok_ = false;                                                                                              \
        try {                                                                                                          \
            expr;                                                                                                      \
        } catch (const ExcType &e_) {                                                                                  \
            ok_ = std::strstr(e_.what(), needle) != nullptr;                                                           \
            if (!ok_) std::printf("  message was: %s\n", e_.what());                                                   \
        } catch (...) {                                                                                                \
        }                                                                                                              \
        if (!ok_) {                                                                                                    \
            std::printf("FAILED %s:%d: %s did not throw %s(\"%s\")\n", __FILE__, __LINE__, #expr, #ExcType, needle);   \
            ++failures;                                                                                                \
        }                                                                                                              \
    } while (0)

template <typename F>
static std::vector<F> uniform_particles(std::size_t n, F size, std::mt19937 &rng)
{
    std::vector<F> v(n * 4);
    std::uniform_real_distribution<F> md(F(0), F(1)), rd(-size / F(2), size / F(2));
    for (std::size_t i = 0; i < n; ++i) v[i] = md(rng);
    for (std::size_t i = n; i < 4 * n; ++i) v[i] = rd(rng);
    return v;
}

template <typename F, mac M>
static void host_checks()
{
    std::mt19937 rng(7);
    const std::size_t s = 3000;
    auto parts = uniform_particles<F>(s, F(1), rng);
    // Iterators + nparts (test/accuracy_acc.cpp:63-71).
    octree<F, M> t{x_coords = parts.begin() + s, y_coords = parts.begin() + 2 * s, z_coords = parts.begin() + 3 * s,
                   masses = parts.begin(),       nparts = s,                      box_size = F(1),
                   max_leaf_n = 8,               ncrit = 64};
    CHECK(t.nparts() == s);
    CHECK(t.box_size() == F(1) && !t.box_size_deduced());
    CHECK(t.max_leaf_n() == 8u && t.ncrit() == 64u);
    CHECK(t.nodes().size() > 1 && t.nodes()[0].code == 1u && t.nodes()[0].n_children == t.nodes().size() - 1);
    // perm / inv_perm are inverse permutations; p_its_o gives back the original order.
    for (std::size_t i = 0; i < s; i += 97) {
        CHECK(t.inv_perm()[t.perm()[i]] == i);
        CHECK(t.p_its_o()[0][static_cast<std::ptrdiff_t>(i)] == parts[s + i]);
        CHECK(t.p_its_u()[3][t.inv_perm()[i]] == parts[i]);
    }
    // Ranges (vectors) without nparts; deduced box.
    std::vector<F> xs(parts.begin() + s, parts.begin() + 2 * s), ys(parts.begin() + 2 * s, parts.begin() + 3 * s),
        zs(parts.begin() + 3 * s, parts.end()), ms(parts.begin(), parts.begin() + s);
    octree<F, M> t2{x_coords = xs, y_coords = ys, z_coords = zs, masses = ms};
    CHECK(t2.nparts() == s && t2.box_size_deduced() && t2.box_size() > F(0.9));
    // Copy / move.
    octree<F, M> t3(t2), t4(std::move(t2));
    CHECK(t3.nodes().size() == t4.nodes().size() && t2.nparts() == 0);
    // Error paths (messages of tree.hpp:1350-1362, 1644-1658, 3364-3370).
    CHECK_THROWS((octree<F, M>{x_coords = xs, y_coords = ys, z_coords = zs, masses = ms, max_leaf_n = 0}),
                 std::invalid_argument, "maximum number of particles per leaf must be nonzero");
    CHECK_THROWS((octree<F, M>{x_coords = xs, y_coords = ys, z_coords = zs, masses = ms, ncrit = 0}),
                 std::invalid_argument, "critical number of particles");
    CHECK_THROWS((octree<F, M>{x_coords = xs, y_coords = ys, z_coords = zs, masses = ms, box_size = -1}),
                 std::invalid_argument, "box size must be a finite non-negative value");
    std::vector<F> shorter(xs.begin(), xs.end() - 1);
    CHECK_THROWS((octree<F, M>{x_coords = shorter, y_coords = ys, z_coords = zs, masses = ms}), std::invalid_argument,
                 "inconsistent sizes");
    std::array<std::vector<F>, 3> accs;
    CHECK_THROWS(t.accs_u(accs, F(0)), std::domain_error, "MAC value must be finite and positive");
    CHECK_THROWS(t.accs_u(accs, F(0.5), eps = -1), std::domain_error, "softening length must be finite");
    CHECK_THROWS(t.accs_u({accs[0].data(), accs[1].data()}, F(0.5)), std::invalid_argument,
                 "iterators is required instead");
    CHECK_THROWS(t.accs_u(accs, F(0.5), split = std::vector<double>{0., 0.}), std::invalid_argument,
                 "cannot all be zero");
    // exact_* are host computations.
    const auto ea = t.exact_acc_u(5), eo = t.exact_acc_o(t.perm()[5]);
    CHECK(ea[0] == eo[0] && ea[1] == eo[1] && ea[2] == eo[2]);
}

template <typename F, mac M>
static void gpu_checks()
{
    std::mt19937 rng(11);
    const std::size_t s = 2000;
    auto parts = uniform_particles<F>(s, F(1), rng);
    octree<F, M> t{x_coords = parts.data() + s, y_coords = parts.data() + 2 * s, z_coords = parts.data() + 3 * s,
                   masses = parts.data(),       nparts = s,                      box_size = F(2)};
    const F theta = F(0.001);
    const double tol = std::is_same_v<F, double> ? 5e-10 : 5e-2;
    std::array<std::vector<F>, 3> accs;
    t.accs_o(accs, theta);
    CHECK(accs[0].size() == s);
    std::vector<F> pots;
    t.pots_o(pots, theta, G = F(2));
    std::array<std::vector<F>, 4> ap;
    t.accs_pots_u(ap, theta, eps = F(0.01));
    for (std::size_t i = 0; i < s; i += 101) {
        const auto e = t.exact_acc_o(i);
        for (int k = 0; k < 3; ++k) CHECK(std::abs((e[k] - accs[k][i]) / e[k]) < tol);
        CHECK(std::abs((t.exact_pot_o(i, G = F(2)) - pots[i]) / pots[i]) < tol);
        const auto e2 = t.exact_acc_pot_u(i, eps = F(0.01));
        for (int k = 0; k < 4; ++k) CHECK(std::abs((e2[k] - ap[k][i]) / e2[k]) < tol);
    }
    // Iterator-array and initializer-list outputs; G = 0 gives exact zeros (test/g_constant_acc.cpp:66-70).
    std::vector<F> ax(s), ay(s), az(s);
    t.accs_u(std::array{ax.begin(), ay.begin(), az.begin()}, F(0.75));
    std::vector<F> bx(s), by(s), bz(s);
    t.accs_u({bx.data(), by.data(), bz.data()}, F(0.75), split = std::vector<double>{0., 1.});
    CHECK(ax == bx && ay == by && az == bz);
    // split = {cpu, dev0}: the CPU engine of the header computes the first half while the GPU computes the rest
    // (tree.hpp:3047-3113 of the reference); same interaction lists, so the halves agree with the GPU-only result to
    // rounding, and the device half exactly.
    t.accs_u({bx.data(), by.data(), bz.data()}, F(0.75), split = std::vector<double>{0.5, 0.5});
    {
        double worst = 0;
        std::size_t n_same = 0;
        for (std::size_t i = 0; i < s; ++i) {
            const double d[3] = {double(ax[i]) - bx[i], double(ay[i]) - by[i], double(az[i]) - bz[i]};
            const double nrm = std::sqrt(double(ax[i]) * ax[i] + double(ay[i]) * ay[i] + double(az[i]) * az[i]);
            worst = std::max(worst, std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]) / nrm);
            n_same += ax[i] == bx[i] && ay[i] == by[i] && az[i] == bz[i];
        }
        CHECK(worst < (std::is_same_v<F, double> ? 1e-12 : 2e-5));
        CHECK(n_same >= s / 2 - 200 && ax[s - 1] == bx[s - 1]);
    }
    // CPU only (split = {1}): the reference's default engine.
    t.accs_u({bx.data(), by.data(), bz.data()}, F(0.75), split = std::vector<double>{1.});
    for (std::size_t i = 0; i < s; i += 97) CHECK(std::abs(double(ax[i]) - bx[i]) <= 2e-5 * (std::abs(double(ax[i])) + 1e-3));
    t.accs_u({bx.data(), by.data(), bz.data()}, F(0.75), G = 0);
    for (std::size_t i = 0; i < s; ++i) CHECK(bx[i] == F(0) && by[i] == F(0) && bz[i] == F(0));
    // update_particles_o: shift everything; accelerations are translation invariant up to rounding.
    t.update_particles_o([s](const auto &its) {
        for (std::size_t i = 0; i < s; ++i) its[0][static_cast<std::ptrdiff_t>(i)] += F(0.125);
    });
    std::array<std::vector<F>, 3> acc2;
    t.accs_o(acc2, theta);
    for (std::size_t i = 0; i < s; i += 53) {
        CHECK(std::abs((acc2[0][i] - accs[0][i]) / accs[0][i]) < (std::is_same_v<F, double> ? 1e-9 : 1e-1));
    }
}

// test/morton.cpp:35-62: encode random coordinate tuples and decode them back, for both code widths and dimensions.
template <std::size_t ND, typename UInt>
static void morton_checks()
{
    constexpr auto cbits = cbits_v<UInt, ND>;
    morton_encoder<ND, UInt> me;
    morton_decoder<ND, UInt> md;
    std::mt19937_64 rng(ND * 100 + sizeof(UInt));
    std::uniform_int_distribution<UInt> udist(0, (UInt(1) << cbits) - 1u);
    UInt b1[ND], b2[ND];
    for (int i = 0; i < 10000; ++i) {
        for (auto &v : b1) v = udist(rng);
        const UInt code = me(&b1[0]);
        md(&b2[0], code);
        CHECK(std::equal(b1, b1 + ND, b2));
    }
    // x -> bit 0, y -> bit 1 (, z -> bit 2): the child order the node-centre tests rely on.
    UInt unit[ND] = {};
    unit[0] = 1;
    CHECK(me(&unit[0]) == UInt(1));
    unit[0] = 0, unit[1] = 1;
    CHECK(me(&unit[0]) == UInt(2));
}

// test/update_masses.cpp:36-150: mass updates change node masses (and nothing else) exactly; all-zero masses turn the
// centres of mass into the geometric node centres; _u and _o flavours.
template <typename F, mac M>
static void update_masses_checks()
{
    std::mt19937 rng(0);
    const std::size_t s = 10000;
    auto parts = uniform_particles<F>(s, F(1), rng);
    octree<F, M> t{x_coords = parts.begin() + s, y_coords = parts.begin() + 2 * s, z_coords = parts.begin() + 3 * s,
                   masses = parts.begin(),       nparts = s,                      box_size = F(10)};
    const auto t2(t);
    t.update_masses_u([](auto) {});
    CHECK(t.nodes() == t2.nodes());
    t.update_masses_o([](auto) {});
    CHECK(t.nodes() == t2.nodes());
    auto doubled = [&](const auto &tr) {
        bool ok = tr.nodes().size() == t2.nodes().size();
        for (std::size_t i = 0; ok && i < tr.nodes().size(); ++i) {
            ok = tr.nodes()[i].props[3] == t2.nodes()[i].props[3] * 2
                 && std::equal(tr.nodes()[i].props, tr.nodes()[i].props + 3, t2.nodes()[i].props);
        }
        return ok;
    };
    t.update_masses_u([s](auto it) {
        for (std::size_t i = 0; i < s; ++i) *(it + static_cast<std::ptrdiff_t>(i)) *= 2;
    });
    CHECK(!(t.nodes() == t2.nodes()));
    CHECK(doubled(t));
    t = t2;
    t.update_masses_o([s](auto it) {
        for (std::size_t i = 0; i < s; ++i) *(it + static_cast<std::ptrdiff_t>(i)) *= 2;
    });
    CHECK(doubled(t));
    auto zeroed = [&](const auto &tr) {
        bool ok = true;
        for (std::size_t i = 0; ok && i < tr.nodes().size(); ++i) {
            F c[3];
            get_node_centre(c, tr.nodes()[i].code, F(10));
            ok = tr.nodes()[i].props[3] == F(0) && std::equal(c, c + 3, tr.nodes()[i].props);
        }
        return ok;
    };
    t = t2;
    t.update_masses_u([s](auto it) {
        for (std::size_t i = 0; i < s; ++i) *(it + static_cast<std::ptrdiff_t>(i)) = 0;
    });
    CHECK(zeroed(t));
    t = t2;
    t.update_masses_o([s](auto it) {
        for (std::size_t i = 0; i < s; ++i) *(it + static_cast<std::ptrdiff_t>(i)) = 0;
    });
    CHECK(zeroed(t));
    // A few individual particles (update_masses.cpp:118-150).
    const std::vector<std::size_t> indices{1, 100, 123, 1045, 9800};
    t = t2;
    t.update_masses_u([&indices](auto it) {
        for (auto idx : indices) *(it + static_cast<std::ptrdiff_t>(idx)) += 1;
    });
    for (auto idx : indices) CHECK(t.p_its_u()[3][idx] == t2.p_its_u()[3][idx] + 1);
    CHECK(!(t.nodes()[0] == t2.nodes()[0]));
    t = t2;
    t.update_masses_o([&indices](auto it) {
        for (auto idx : indices) *(it + static_cast<std::ptrdiff_t>(idx)) += 1;
    });
    for (auto idx : indices) {
        CHECK(t.p_its_o()[3][static_cast<std::ptrdiff_t>(idx)] == t2.p_its_o()[3][static_cast<std::ptrdiff_t>(idx)] + 1);
    }
    // Positions, codes and permutations are untouched by a mass update.
    CHECK(std::equal(t.c_it_u(), t.c_it_u() + s, t2.c_it_u()) && t.perm() == t2.perm());
}

// test/update.cpp:34-213: position updates through update_particles_u / _o -- ordered iterators, perm / last_perm /
// inv_perm bookkeeping across no-op updates, coordinate swaps, shifts and rescalings.
template <typename F, mac M>
static void update_positions_checks()
{
    std::mt19937 rng(5);
    const std::size_t s = 10000;
    auto parts = uniform_particles<F>(s, F(1), rng);
    octree<F, M> t{x_coords = parts.begin() + s, y_coords = parts.begin() + 2 * s, z_coords = parts.begin() + 3 * s,
                   masses = parts.begin(),       nparts = s,                      box_size = F(10)};
    const auto t2(t);
    using size_type = typename decltype(t)::size_type;
    using diff_t = std::ptrdiff_t;
    CHECK(t.perm() == t.last_perm());
    std::vector<size_type> track(1000);
    std::uniform_int_distribution<size_type> idist(0, s - 1);
    for (auto &v : track) v = idist(rng);
    auto tracked_ok = [&](const auto &tr, F shift, F scale, int rot) {
        // rot = 0: (x, y, z); 1: (y, z, x) in the x, y, z slots. Values = (orig + shift) * scale.
        auto pro = tr.p_its_o();
        bool ok = true;
        for (auto idx : track) {
            const F ox = parts[s + idx], oy = parts[2 * s + idx], oz = parts[3 * s + idx];
            const F e0 = rot ? oy : ox, e1 = rot ? oz : oy, e2 = rot ? ox : oz;
            ok = ok && pro[0][static_cast<diff_t>(idx)] == (e0 + shift) * scale
                 && pro[1][static_cast<diff_t>(idx)] == (e1 + shift) * scale
                 && pro[2][static_cast<diff_t>(idx)] == (e2 + shift) * scale && pro[3][static_cast<diff_t>(idx)] == parts[idx];
        }
        return ok;
    };
    CHECK(tracked_ok(t, F(0), F(1), 0));
    auto orig_perm = t.perm(), orig_inv = t.inv_perm();
    std::vector<size_type> iota(s);
    for (std::size_t i = 0; i < s; ++i) iota[i] = i;
    t.update_particles_u([](const auto &) {});
    CHECK(orig_perm == t.perm() && iota == t.last_perm() && orig_inv == t.inv_perm() && tracked_ok(t, F(0), F(1), 0));
    t.update_particles_o([](const auto &) {});
    CHECK(orig_perm == t.perm() && iota == t.last_perm() && orig_inv == t.inv_perm() && tracked_ok(t, F(0), F(1), 0));
    for (int k = 0; k < 4; ++k) CHECK(std::equal(t.p_its_u()[k], t.p_its_u()[k] + s, t2.p_its_u()[k]));
    // x, y, z -> y, z, x through the ordered iterators; the old x (Morton order) must reappear as z through last_perm.
    std::vector<F> x_old(t.p_its_u()[0], t.p_its_u()[0] + s), x_orig(x_old), x_new(s);
    t.update_particles_o([s](const auto &its) {
        for (std::size_t i = 0; i < s; ++i) {
            const auto d = static_cast<diff_t>(i);
            std::swap(*(its[0] + d), *(its[1] + d));
            std::swap(*(its[1] + d), *(its[2] + d));
        }
    });
    CHECK(tracked_ok(t, F(0), F(1), 1));
    auto follow = [&](F shift, F scale) {
        const auto lp = t.last_perm();
        for (std::size_t i = 0; i < s; ++i) x_new[i] = (x_old[lp[i]] + shift) * scale;
        x_old = x_new;
    };
    follow(F(0), F(1));
    CHECK(std::equal(x_new.begin(), x_new.end(), t.p_its_u()[2]));
    t.update_particles_o([s](const auto &its) {
        for (std::size_t i = 0; i < s; ++i) {
            const auto d = static_cast<diff_t>(i);
            std::swap(*(its[2] + d), *(its[1] + d));
            std::swap(*(its[0] + d), *(its[1] + d));
        }
    });
    CHECK(tracked_ok(t, F(0), F(1), 0));
    follow(F(0), F(1));
    CHECK(std::equal(x_new.begin(), x_new.end(), t.p_its_u()[0]) && x_new == x_orig);
    t.update_particles_u([s](const auto &its) {
        for (std::size_t i = 0; i < s; ++i)
            for (std::size_t j = 0; j < 3; ++j) *(its[j] + static_cast<diff_t>(i)) += F(1);
    });
    CHECK(tracked_ok(t, F(1), F(1), 0));
    follow(F(1), F(1));
    CHECK(std::equal(x_new.begin(), x_new.end(), t.p_its_u()[0]));
    t.update_particles_u([s](const auto &its) {
        for (std::size_t i = 0; i < s; ++i)
            for (std::size_t j = 0; j < 3; ++j) *(its[j] + static_cast<diff_t>(i)) /= F(2);
    });
    CHECK(tracked_ok(t, F(1), F(0.5), 0));
    follow(F(0), F(0.5));
    CHECK(std::equal(x_new.begin(), x_new.end(), t.p_its_u()[0]));
    // Moving a particle out of the box is an error, as at construction (tree.hpp:381-429).
    CHECK_THROWS(t.update_particles_u([](const auto &its) { *(its[0]) = F(100); }), std::invalid_argument,
                 "outside the allowed bounds");
}

// Quadtrees: the call shapes of test/node_centre.cpp:23-53 and the 2-D flavour of the accuracy checks.
template <typename F, mac M>
static void quadtree_host_checks()
{
    using tree_t = quadtree<F, M>;
    tree_t t;
    CHECK(t.nodes().size() == 0u);
    t = tree_t{x_coords = std::vector<F>{-1, -1, 1, 1}, y_coords = std::vector<F>{-1, 1, -1, 1},
               masses = std::vector<F>{1, 1, 1, 1}, box_size = 10, max_leaf_n = 1};
    CHECK(t.nodes().size() == 5u);
    CHECK(t.nodes()[0].code == 1u && t.nodes()[1].code == 4u && t.nodes()[4].code == 7u);
    CHECK(t.nodes()[1].props[0] == F(-1) && t.nodes()[1].props[1] == F(-1) && t.nodes()[1].props[2] == F(1));
    CHECK(t.nodes()[2].props[0] == F(1) && t.nodes()[2].props[1] == F(-1));
    CHECK(t.nodes()[0].props[2] == F(4));
    std::mt19937 rng(3);
    const std::size_t s = 1500;
    std::vector<F> m(s), x(s), y(s);
    std::uniform_real_distribution<F> md(F(0), F(1)), rd(F(-0.5), F(0.5));
    for (std::size_t i = 0; i < s; ++i) m[i] = md(rng), x[i] = rd(rng), y[i] = rd(rng);
    tree_t u{x_coords = x, y_coords = y, masses = m};
    CHECK(u.nparts() == s && u.perm().size() == s);
    for (std::size_t i = 1; i < s; ++i) CHECK(u.c_it_u()[i - 1] <= u.c_it_u()[i]);
    for (std::size_t i = 0; i < s; ++i) CHECK(u.p_its_o()[0][static_cast<std::ptrdiff_t>(i)] == x[i]);
    const auto e = u.exact_acc_pot_o(7);
    CHECK(e.size() == 3u && std::isfinite(e[0]) && std::isfinite(e[2]));
    CHECK_THROWS((tree_t{x_coords = x, y_coords = std::vector<F>(3), masses = m}), std::invalid_argument,
                 "inconsistent sizes");
}

template <typename F, mac M>
static void quadtree_gpu_checks(bool on_device)
{
    std::mt19937 rng(5);
    const std::size_t s = 2500;
    std::vector<F> m(s), x(s), y(s);
    std::uniform_real_distribution<F> md(F(0), F(1)), rd(F(-0.5), F(0.5));
    for (std::size_t i = 0; i < s; ++i) m[i] = md(rng), x[i] = rd(rng), y[i] = rd(rng);
    quadtree<F, M> t{x_coords = x.data(), y_coords = y.data(), masses = m.data(), nparts = s, box_size = F(2),
                     device_build = on_device};
    const F theta = F(0.001);
    const double tol = std::is_same_v<F, double> ? 5e-10 : 5e-2;
    std::array<std::vector<F>, 2> accs;
    t.accs_o(accs, theta);
    std::vector<F> pots;
    t.pots_o(pots, theta, G = F(2));
    std::array<std::vector<F>, 3> ap;
    t.accs_pots_u(ap, theta, eps = F(0.01));
    for (std::size_t i = 0; i < s; i += 101) {
        const auto e = t.exact_acc_o(i);
        for (int k = 0; k < 2; ++k) CHECK(std::abs((e[k] - accs[k][i]) / e[k]) < tol);
        CHECK(std::abs((t.exact_pot_o(i, G = F(2)) - pots[i]) / pots[i]) < tol);
        const auto e2 = t.exact_acc_pot_u(i, eps = F(0.01));
        for (int k = 0; k < 3; ++k) CHECK(std::abs((e2[k] - ap[k][i]) / e2[k]) < tol);
    }
    std::vector<F> ax(s), ay(s);
    t.accs_u({ax.data(), ay.data()}, F(0.75), G = 0);
    for (std::size_t i = 0; i < s; ++i) CHECK(ax[i] == F(0) && ay[i] == F(0));
    t.update_particles_u([s](const auto &its) {
        for (std::size_t i = 0; i < s; ++i) its[1][i] *= F(0.5);
    });
    std::array<std::vector<F>, 2> acc2;
    t.accs_o(acc2, theta);
    for (std::size_t i = 0; i < s; i += 97) {
        const auto e = t.exact_acc_o(i);
        for (int k = 0; k < 2; ++k) CHECK(std::abs((e[k] - acc2[k][i]) / e[k]) < tol);
    }
}

// 32-bit Morton codes (tree<NDim, F, std::uint32_t, MAC>, one of the reference's instantiations): 10 / 15 bits per
// coordinate, so the tree is at most 10 / 15 levels deep.
template <std::size_t ND, typename F>
static void narrow_code_checks(bool gpu)
{
    using tree_t = tree<ND, F, std::uint32_t, mac::bh>;
    std::mt19937 rng(9);
    const std::size_t s = 4000;
    std::vector<F> m(s), c[3];
    std::uniform_real_distribution<F> md(F(0), F(1)), rd(F(-0.5), F(0.5));
    for (std::size_t i = 0; i < s; ++i) m[i] = md(rng);
    for (auto &v : c) {
        v.resize(s);
        for (auto &x : v) x = rd(rng);
    }
    auto make = [&](bool dev) {
        if constexpr (ND == 3) {
            return tree_t{x_coords = c[0], y_coords = c[1], z_coords = c[2], masses = m, box_size = F(1), max_leaf_n = 2,
                          device_build = dev};
        } else {
            return tree_t{x_coords = c[0], y_coords = c[1], masses = m, box_size = F(1), max_leaf_n = 2,
                          device_build = dev};
        }
    };
    tree_t t = make(false);
    const unsigned cb = ND == 3 ? 10u : 15u;
    unsigned max_level = 0;
    for (const auto &n : t.nodes()) max_level = std::max<unsigned>(max_level, n.level);
    CHECK(max_level <= cb && max_level >= 5u);
    for (std::size_t i = 0; i < s; ++i) CHECK(t.c_it_u()[i] < (std::uint32_t(1) << (cb * ND)));
    for (std::size_t i = 1; i < s; ++i) CHECK(t.c_it_u()[i - 1] <= t.c_it_u()[i]);
    if (gpu) {
        CHECK_THROWS(make(true), std::invalid_argument, "64-bit Morton codes");
        std::array<std::vector<F>, ND> accs;
        t.accs_o(accs, F(0.001));
        const double tol = std::is_same_v<F, double> ? 5e-10 : 5e-2;
        for (std::size_t i = 0; i < s; i += 211) {
            const auto e = t.exact_acc_o(i);
            for (std::size_t k = 0; k < ND; ++k) CHECK(std::abs((e[k] - accs[k][i]) / e[k]) < tol);
        }
    }
}

// Output vectors with rakau_amd::pinned_allocator (the `std::vector<F, Allocator>` overloads, tree.hpp:3406-3497 of the
// reference): the kernels write into them directly; the results are the bits of the pageable path.
template <typename F>
static void pinned_output_checks()
{
    std::mt19937 rng(11);
    const std::size_t s = 300000;
    std::vector<F> m(s), x(s), y(s), z(s);
    std::uniform_real_distribution<F> md(F(0.1), F(1)), rd(F(-0.5), F(0.5));
    for (std::size_t i = 0; i < s; ++i) m[i] = md(rng), x[i] = rd(rng), y[i] = rd(rng), z[i] = rd(rng);
    octree<F, mac::bh> t{x_coords = x, y_coords = y, z_coords = z, masses = m, box_size = F(1)};
    using pvec = std::vector<F, pinned_allocator<F>>;
    std::array<std::vector<F>, 3> a0;
    std::array<pvec, 3> a1;
    t.accs_u(a0, F(0.75));
    t.accs_u(a1, F(0.75));
    bool same = a1[0].size() == s;
    for (std::size_t k = 0; same && k < 3; ++k) same = std::memcmp(a0[k].data(), a1[k].data(), s * sizeof(F)) == 0;
    CHECK(same);
    std::array<std::vector<F>, 4> b0;
    std::array<pvec, 4> b1;
    t.accs_pots_o(b0, F(0.75), eps = F(0.01), G = F(3));
    t.accs_pots_o(b1, F(0.75), eps = F(0.01), G = F(3));
    same = true;
    for (std::size_t k = 0; same && k < 4; ++k) same = std::memcmp(b0[k].data(), b1[k].data(), s * sizeof(F)) == 0;
    CHECK(same);
    // A CPU share next to the device: both engines write into the same pinned vectors.
    pvec p0, p1;
    std::vector<F> q0;
    t.pots_u(q0, F(0.75), split = std::vector<double>{0.25, 0.75});
    t.pots_u(p0, F(0.75), split = std::vector<double>{0.25, 0.75});
    t.pots_u(p1, F(0.75));
    CHECK(std::memcmp(q0.data(), p0.data(), s * sizeof(F)) == 0);
    double worst = 0;
    for (std::size_t i = 0; i < s; ++i) worst = std::max(worst, std::abs(double(p0[i] - p1[i]) / double(p1[i])));
    CHECK(worst < (std::is_same_v<F, double> ? 1e-12 : 1e-4));
}

int main(int argc, char **argv)
{
    const bool gpu = argc > 1 && std::strcmp(argv[1], "gpu") == 0;
    host_checks<float, mac::bh>();
    host_checks<double, mac::bh_geom>();
    morton_checks<2, std::uint32_t>();
    morton_checks<3, std::uint32_t>();
    morton_checks<2, std::uint64_t>();
    morton_checks<3, std::uint64_t>();
    update_positions_checks<double, mac::bh>();
    update_positions_checks<float, mac::bh_geom>();
    update_masses_checks<double, mac::bh>();
    update_masses_checks<float, mac::bh_geom>();
    quadtree_host_checks<double, mac::bh>();
    quadtree_host_checks<float, mac::bh_geom>();
    narrow_code_checks<3, double>(gpu);
    narrow_code_checks<2, float>(gpu);
    if (gpu) {
        quadtree_gpu_checks<double, mac::bh>(false);
        quadtree_gpu_checks<double, mac::bh_geom>(true);
        quadtree_gpu_checks<float, mac::bh>(true);
        gpu_checks<float, mac::bh>();
        gpu_checks<double, mac::bh>();
        gpu_checks<double, mac::bh_geom>();
        pinned_output_checks<float>();
        pinned_output_checks<double>();
    }
    std::printf("%s: %d failure(s)\n", gpu ? "gpu+host checks" : "host checks", failures);
    return failures ? 1 : 0;
}
