// C ABI over rakau_amd::octree<F, MAC> / quadtree<F, MAC> (declared in include/rakau_amd_tree.h).
#include "../../include/rakau_amd/tree.hpp"
#include "../../include/rakau_amd_tree.h"

#include <dlfcn.h>

#include <variant>

namespace
{

using namespace rakau_amd;

template <std::size_t ND, typename F, mac M>
using narrow_tree = tree<ND, F, std::uint32_t, M>; // 32-bit Morton codes

using any_tree
    = std::variant<octree<float, mac::bh>, octree<float, mac::bh_geom>, octree<double, mac::bh>,
                   octree<double, mac::bh_geom>, quadtree<float, mac::bh>, quadtree<float, mac::bh_geom>,
                   quadtree<double, mac::bh>, quadtree<double, mac::bh_geom>, narrow_tree<3, float, mac::bh>,
                   narrow_tree<3, float, mac::bh_geom>, narrow_tree<3, double, mac::bh>,
                   narrow_tree<3, double, mac::bh_geom>, narrow_tree<2, float, mac::bh>,
                   narrow_tree<2, float, mac::bh_geom>, narrow_tree<2, double, mac::bh>,
                   narrow_tree<2, double, mac::bh_geom>>;

thread_local std::string g_tree_err;

} // namespace

struct rk_tree {
    any_tree t;
};

// Defined in rk_state.hip: lets this translation unit report through rk_last_error().
extern "C" void rk_set_last_error_(const char *msg);

namespace
{

template <typename Fn>
int guard(Fn &&f) noexcept
{
    try {
        f();
        return RK_OK;
    } catch (const std::domain_error &e) {
        rk_set_last_error_(e.what());
        return RK_EDOMAIN;
    } catch (const std::invalid_argument &e) {
        rk_set_last_error_(e.what());
        return RK_EINVAL;
    } catch (const std::overflow_error &e) {
        rk_set_last_error_(e.what());
        return RK_EOVERFLOW;
    } catch (const std::bad_alloc &) {
        rk_set_last_error_("out of memory");
        return RK_ENOMEM;
    } catch (const std::exception &e) {
        rk_set_last_error_(e.what());
        return RK_ERUNTIME;
    }
}

template <typename Tree>
struct fp_of;
template <std::size_t ND, typename F, typename UInt, mac M>
struct fp_of<tree<ND, F, UInt, M>> {
    using type = F;
    using code_type = UInt;
    static constexpr std::size_t ndim = ND;
};

// src: the ND coordinate arrays followed by the masses.
template <std::size_t ND, typename F, mac M, typename UInt>
tree<ND, F, UInt, M> make_tree(const void *const *src, std::int64_t n, double box, std::uint64_t max_leaf_n,
                               std::uint64_t ncrit, bool dev)
{
    using tree_t = tree<ND, F, UInt, M>;
    const auto *xs = static_cast<const F *>(src[0]), *ys = static_cast<const F *>(src[1]),
               *ms = static_cast<const F *>(src[ND]);
    if constexpr (ND == 3) {
        const auto *zs = static_cast<const F *>(src[2]);
        if (box == 0.) {
            return tree_t{kwargs::x_coords = xs,         kwargs::y_coords = ys,
                          kwargs::z_coords = zs,         kwargs::masses = ms,
                          kwargs::nparts = n,            kwargs::max_leaf_n = max_leaf_n,
                          kwargs::ncrit = ncrit,         kwargs::device_build = dev};
        }
        return tree_t{kwargs::x_coords = xs,         kwargs::y_coords = ys, kwargs::z_coords = zs,
                      kwargs::masses = ms,           kwargs::nparts = n,    kwargs::box_size = box,
                      kwargs::max_leaf_n = max_leaf_n, kwargs::ncrit = ncrit, kwargs::device_build = dev};
    } else {
        if (box == 0.) {
            return tree_t{kwargs::x_coords = xs, kwargs::y_coords = ys,           kwargs::masses = ms,
                          kwargs::nparts = n,    kwargs::max_leaf_n = max_leaf_n, kwargs::ncrit = ncrit,
                          kwargs::device_build = dev};
        }
        return tree_t{kwargs::x_coords = xs,   kwargs::y_coords = ys,           kwargs::masses = ms,
                      kwargs::nparts = n,      kwargs::box_size = box,          kwargs::max_leaf_n = max_leaf_n,
                      kwargs::ncrit = ncrit,   kwargs::device_build = dev};
    }
}

} // namespace

extern "C" {

int rk_tree_create_nd(rk_tree **out, int ndim, int fp, int mac_kind, const void *const *src, int64_t nparts,
                      double box_size, uint64_t max_leaf_n, uint64_t ncrit, int flags)
{
    return guard([&] {
        if (!out) {
            throw std::invalid_argument("null output pointer");
        }
        *out = nullptr;
        if (ndim != 2 && ndim != 3) {
            throw std::invalid_argument("ndim must be 2 (quadtree) or 3 (octree)");
        }
        if (nparts < 0 || !src) {
            throw std::invalid_argument("invalid particle arrays");
        }
        for (int j = 0; j < ndim + 1 && nparts > 0; ++j) {
            if (!src[j]) {
                throw std::invalid_argument("invalid particle arrays");
            }
        }
        const bool dev = (flags & 1) != 0, narrow = (flags & 2) != 0;
        std::unique_ptr<rk_tree> t;
        auto make = [&](auto nd, auto code_tag) {
            constexpr std::size_t ND = decltype(nd)::value;
            using UInt = typename decltype(code_tag)::type;
            switch (fp * 2 + mac_kind) {
                case 0:
                    t.reset(new rk_tree{make_tree<ND, float, mac::bh, UInt>(src, nparts, box_size, max_leaf_n, ncrit, dev)});
                    break;
                case 1:
                    t.reset(new rk_tree{
                        make_tree<ND, float, mac::bh_geom, UInt>(src, nparts, box_size, max_leaf_n, ncrit, dev)});
                    break;
                case 2:
                    t.reset(new rk_tree{make_tree<ND, double, mac::bh, UInt>(src, nparts, box_size, max_leaf_n, ncrit, dev)});
                    break;
                case 3:
                    t.reset(new rk_tree{
                        make_tree<ND, double, mac::bh_geom, UInt>(src, nparts, box_size, max_leaf_n, ncrit, dev)});
                    break;
                default:
                    throw std::invalid_argument("invalid fp / mac selector");
            }
        };
        struct wide {
            using type = std::size_t;
        };
        struct narrow_t {
            using type = std::uint32_t;
        };
        if (ndim == 3) {
            narrow ? make(std::integral_constant<std::size_t, 3>{}, narrow_t{})
                   : make(std::integral_constant<std::size_t, 3>{}, wide{});
        } else {
            narrow ? make(std::integral_constant<std::size_t, 2>{}, narrow_t{})
                   : make(std::integral_constant<std::size_t, 2>{}, wide{});
        }
        *out = t.release();
    });
}

int rk_tree_create(rk_tree **out, int fp, int mac_kind, const void *x, const void *y, const void *z, const void *m,
                   int64_t nparts, double box_size, uint64_t max_leaf_n, uint64_t ncrit, int flags)
{
    const void *src[4] = {x, y, z, m};
    return rk_tree_create_nd(out, 3, fp, mac_kind, src, nparts, box_size, max_leaf_n, ncrit, flags);
}

void rk_tree_destroy(rk_tree *t)
{
    delete t;
}

int rk_tree_info(const rk_tree *t, int64_t info[8], double *box_size)
{
    return guard([&] {
        if (!t || !info || !box_size) {
            throw std::invalid_argument("null argument");
        }
        std::visit(
            [&](const auto &tr) {
                info[0] = static_cast<int64_t>(tr.nparts());
                info[1] = static_cast<int64_t>(tr.nodes().size());
                info[2] = static_cast<int64_t>(tr.crit_nodes().size());
                info[3] = static_cast<int64_t>(tr.max_leaf_n());
                info[4] = static_cast<int64_t>(tr.ncrit());
                info[5] = tr.box_size_deduced();
                info[6] = static_cast<int64_t>(sizeof(tr.nodes()[0]));
                info[7] = static_cast<int64_t>(fp_of<std::decay_t<decltype(tr)>>::ndim);
                *box_size = static_cast<double>(tr.box_size());
            },
            t->t);
    });
}

int rk_tree_get(const rk_tree *t, int what, void *dst)
{
    return guard([&] {
        if (!t || !dst) {
            throw std::invalid_argument("null argument");
        }
        std::visit(
            [&](const auto &tr) {
                using F = typename fp_of<std::decay_t<decltype(tr)>>::type;
                constexpr std::size_t ND = fp_of<std::decay_t<decltype(tr)>>::ndim;
                const auto n = tr.nparts();
                if (what >= 0 && what <= 3) {
                    // 0, 1, 2 = x, y, z; 3 = masses.
                    if (what == 2 && ND == 2) {
                        throw std::invalid_argument("a quadtree has no z coordinates");
                    }
                    std::memcpy(dst, tr.p_its_u()[what == 3 ? ND : static_cast<std::size_t>(what)], n * sizeof(F));
                } else if (what == 4) {
                    // Codes in the tree's own width (uint64 or uint32).
                    std::memcpy(dst, tr.c_it_u(), n * sizeof(typename fp_of<std::decay_t<decltype(tr)>>::code_type));
                } else if (what == 5) {
                    std::memcpy(dst, tr.perm().data(), n * sizeof(std::uint64_t));
                } else if (what == 6) {
                    std::memcpy(dst, tr.last_perm().data(), n * sizeof(std::uint64_t));
                } else if (what == 7) {
                    std::memcpy(dst, tr.inv_perm().data(), n * sizeof(std::uint64_t));
                } else if (what == 8) {
                    auto *o = static_cast<std::uint64_t *>(dst);
                    for (const auto &c : tr.crit_nodes()) {
                        *o++ = c.code;
                        *o++ = c.begin;
                        *o++ = c.end;
                    }
                } else {
                    throw std::invalid_argument("invalid array selector");
                }
            },
            t->t);
    });
}

int rk_tree_nodes(const rk_tree *t, const void **ptr, int64_t *count, int64_t *stride)
{
    return guard([&] {
        if (!t || !ptr || !count || !stride) {
            throw std::invalid_argument("null argument");
        }
        std::visit(
            [&](const auto &tr) {
                *ptr = tr.nodes().data();
                *count = static_cast<int64_t>(tr.nodes().size());
                *stride = static_cast<int64_t>(sizeof(tr.nodes()[0]));
            },
            t->t);
    });
}

int rk_tree_state(const rk_tree *t, rk_state **state)
{
    return guard([&] {
        if (!t || !state) {
            throw std::invalid_argument("null argument");
        }
        std::visit([&](const auto &tr) { *state = tr.device_state(0); }, t->t);
    });
}

int rk_tree_acc_pot(const rk_tree *t, int q, int ordered, void *const *out, double theta, double G, double eps,
                    const double *split, int n_split)
{
    return guard([&] {
        if (!t || !out) {
            throw std::invalid_argument("null argument");
        }
        const std::vector<double> sp(split, split + (split ? n_split : 0));
        std::visit(
            [&](const auto &tr) {
                using F = typename fp_of<std::decay_t<decltype(tr)>>::type;
                constexpr std::size_t ND = fp_of<std::decay_t<decltype(tr)>>::ndim;
                auto o = [&](int k) { return static_cast<F *>(out[k]); };
                std::array<F *, ND> oa;
                std::array<F *, ND + 1u> oap;
                for (std::size_t k = 0; k < ND + 1u; ++k) {
                    oap[k] = o(static_cast<int>(k));
                    if (k < ND) {
                        oa[k] = oap[k];
                    }
                }
                const F th = static_cast<F>(theta);
                switch (q * 2 + (ordered ? 1 : 0)) {
                    case 0:
                        tr.accs_u(oa, th, kwargs::G = G, kwargs::eps = eps, kwargs::split = sp);
                        break;
                    case 1:
                        tr.accs_o(oa, th, kwargs::G = G, kwargs::eps = eps, kwargs::split = sp);
                        break;
                    case 2:
                        tr.pots_u(o(0), th, kwargs::G = G, kwargs::eps = eps, kwargs::split = sp);
                        break;
                    case 3:
                        tr.pots_o(o(0), th, kwargs::G = G, kwargs::eps = eps, kwargs::split = sp);
                        break;
                    case 4:
                        tr.accs_pots_u(oap, th, kwargs::G = G, kwargs::eps = eps, kwargs::split = sp);
                        break;
                    case 5:
                        tr.accs_pots_o(oap, th, kwargs::G = G, kwargs::eps = eps, kwargs::split = sp);
                        break;
                    default:
                        throw std::invalid_argument("q must be 0, 1 or 2");
                }
            },
            t->t);
    });
}

// The CPU engine compiled for AVX-512 lives in its own shared object next to this library (rk_cpu_engine.cpp); it is
// loaded on first use, and only on CPUs that have AVX-512F + DQ + VL. RAKAU_AMD_CPU_ISA=avx2 keeps the caller's flavour.
int rk_cpu_engine_run(const rk_cpu_job *job)
{
    using entry_t = int (*)(const rk_cpu_job *);
    static const entry_t entry = []() -> entry_t {
        const char *e = std::getenv("RAKAU_AMD_CPU_ISA");
        if (e && std::string(e) != "avx512") {
            return nullptr;
        }
        __builtin_cpu_init();
        if (!__builtin_cpu_supports("avx512f") || !__builtin_cpu_supports("avx512dq") || !__builtin_cpu_supports("avx512vl")) {
            return nullptr;
        }
        Dl_info info;
        if (!dladdr(reinterpret_cast<const void *>(&rk_cpu_engine_run), &info) || !info.dli_fname) {
            return nullptr;
        }
        std::string path(info.dli_fname);
        const auto slash = path.find_last_of('/');
        path = (slash == std::string::npos ? std::string() : path.substr(0, slash + 1)) + "librakau_amd_cpu512.so";
        void *h = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
        if (!h) {
            return nullptr;
        }
        return reinterpret_cast<entry_t>(dlsym(h, "rk_cpu_engine_entry"));
    }();
    if (!entry) {
        return -1;
    }
    if (!job) {
        return RK_OK; // probe: the AVX-512 flavour is available
    }
    const int rc = entry(job);
    if (rc != RK_OK) {
        rk_set_last_error_("the AVX-512 flavour of the CPU engine failed");
    }
    return rc;
}

int rk_tree_cpu_acc_pot(const rk_tree *t, int q, void *const *out, double theta, double G, double eps, int flavour,
                        unsigned nthreads)
{
    return guard([&] {
        if (!t || !out) {
            throw std::invalid_argument("null argument");
        }
        if (flavour < 0 || flavour > 2) {
            throw std::invalid_argument("flavour must be 0 (automatic), 1 (scalar) or 2 (SIMD, exact arithmetic)");
        }
        std::visit(
            [&](const auto &tr) {
                using F = typename fp_of<std::decay_t<decltype(tr)>>::type;
                constexpr std::size_t ND = fp_of<std::decay_t<decltype(tr)>>::ndim;
                const auto fl = static_cast<cpu_flavour>(flavour);
                const F th = static_cast<F>(theta), g = static_cast<F>(G), e = static_cast<F>(eps);
                auto ptrs = [&](auto tag) {
                    std::array<F *, decltype(tag)::value> a;
                    for (std::size_t k = 0; k < a.size(); ++k) {
                        a[k] = static_cast<F *>(out[k]);
                    }
                    return a;
                };
                switch (q) {
                    case 0: tr.template cpu_acc_pot_u<0>(ptrs(std::integral_constant<std::size_t, ND>{}), th, g, e, fl, nthreads); break;
                    case 1: tr.template cpu_acc_pot_u<1>(ptrs(std::integral_constant<std::size_t, 1>{}), th, g, e, fl, nthreads); break;
                    case 2: tr.template cpu_acc_pot_u<2>(ptrs(std::integral_constant<std::size_t, ND + 1u>{}), th, g, e, fl, nthreads); break;
                    default: throw std::invalid_argument("q must be 0, 1 or 2");
                }
            },
            t->t);
    });
}

int rk_tree_exact(const rk_tree *t, int q, int ordered, int64_t idx, double G, double eps, void *out)
{
    return guard([&] {
        if (!t || !out) {
            throw std::invalid_argument("null argument");
        }
        std::visit(
            [&](const auto &tr) {
                using F = typename fp_of<std::decay_t<decltype(tr)>>::type;
                auto *o = static_cast<F *>(out);
                if (idx < 0 || static_cast<std::size_t>(idx) >= tr.nparts()) {
                    throw std::invalid_argument("particle index out of range");
                }
                const auto i = static_cast<std::size_t>(idx);
                if (q == 0) {
                    const auto r = ordered ? tr.exact_acc_o(i, kwargs::G = G, kwargs::eps = eps)
                                           : tr.exact_acc_u(i, kwargs::G = G, kwargs::eps = eps);
                    std::copy(r.begin(), r.end(), o);
                } else if (q == 1) {
                    o[0] = ordered ? tr.exact_pot_o(i, kwargs::G = G, kwargs::eps = eps)
                                   : tr.exact_pot_u(i, kwargs::G = G, kwargs::eps = eps);
                } else if (q == 2) {
                    const auto r = ordered ? tr.exact_acc_pot_o(i, kwargs::G = G, kwargs::eps = eps)
                                           : tr.exact_acc_pot_u(i, kwargs::G = G, kwargs::eps = eps);
                    std::copy(r.begin(), r.end(), o);
                } else {
                    throw std::invalid_argument("q must be 0, 1 or 2");
                }
            },
            t->t);
    });
}

int rk_tree_update_particles(rk_tree *t, const void *x, const void *y, const void *z, const void *m)
{
    return guard([&] {
        if (!t) {
            throw std::invalid_argument("null argument");
        }
        std::visit(
            [&](auto &tr) {
                using F = typename fp_of<std::decay_t<decltype(tr)>>::type;
                constexpr std::size_t ND = fp_of<std::decay_t<decltype(tr)>>::ndim;
                const auto n = tr.nparts();
                const void *all[4] = {x, y, z, m};
                const void *src[ND + 1u];
                for (std::size_t j = 0; j < ND; ++j) {
                    src[j] = all[j];
                }
                src[ND] = m; // a quadtree ignores z
                tr.update_particles_u([&](const auto &its) {
                    for (std::size_t j = 0; j < ND + 1u; ++j) {
                        if (src[j]) {
                            std::copy(static_cast<const F *>(src[j]), static_cast<const F *>(src[j]) + n, its[j]);
                        }
                    }
                });
            },
            t->t);
    });
}

} // extern "C"
