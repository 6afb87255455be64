// ORACLE -- TEST INFRASTRUCTURE ONLY. NOT PART OF THE PRODUCT PATH.
//
// CPU restatement of bluescarni/rakau's Barnes-Hut algorithm for the hot path
// accs_u()/pots_u()/accs_pots_u() (and the _o variants), written from the reference's
// *behaviour*; every function cites the reference file:line it follows
// (paths relative to /root/reference).  It restates the scalar flavour of the reference
// (the RAKAU_DISABLE_SIMD code path, include/rakau/tree.hpp:2258-2320, 2432-2470,
// 2564-2589, 2740-2777) with fma_wrap == std::fma (the FP_FAST_FMA flavour,
// tree.hpp:180-205).
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
// library.  The product (rakau_amd/) never links, imports or calls it.
//
// Pinning status: PARITY UNPINNED. The reference cannot be built in this image (TBB, xsimd and Boost are absent and no
// stand-ins may be written), and it ships no golden vectors, so no output of a run of the reference backs this file.
// What constrains it instead (tests/test_oracle_*.py, tests/test_cpu_engine.py):
//   (1) the reference's own known-answer tests restated on this oracle
//       (test/accuracy_*.cpp, g_constant_*.cpp, zero_masses.cpp, softening_*.cpp,
//        ordering_*.cpp: agreement with the direct sum to the tolerances those tests state,
//        bit-exact G scaling, exact zeros);
//   (2) bit-for-bit agreement with an independently structured second implementation (the scalar flavour of
//       include/rakau_amd/cpu_engine.hpp) on trees built by an independent builder;
//   (3) consistency with the figures SURVEY.md section 8(c) recorded (node count, deduced box size, accs_u at Morton
//       index 0 for the default-seeded benchmark Plummer sphere) -- obtained there from a stand-in build, hence a
//       consistency check, not a pin.
//
// Build: see oracle/Makefile  (g++ -O2 -ffp-contract=off -mfma; no -march=native: the .so
// travels to the GPU box).

#include <algorithm>
#include <array>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <memory>
#include <numeric>
#include <random>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

namespace
{

using u64 = std::uint64_t;
// Width of the interleaved partial sums in compute_node_properties (orc_set_simd_width); 1 = scalar association.
unsigned g_simd_width = 1;

thread_local std::string g_last_error;

// Number of bits per coordinate in the 64-bit Morton code of an ND-dimensional tree.
// Reference: include/rakau/detail/tree_fwd.hpp:141-150 (64/3 - !(64%3) = 21 for octrees, 64/2 - 1 = 31 for quadtrees).
// With 32-bit codes (UInt = std::uint32_t, one of the reference's instantiations): 10 and 15.
constexpr unsigned cbits_for(unsigned code_bits, unsigned nd)
{
    return code_bits / nd - !(code_bits % nd);
}
static_assert(cbits_for(64, 3) == 21 && cbits_for(64, 2) == 31 && cbits_for(32, 3) == 10 && cbits_for(32, 2) == 15);

// 3D Morton encoding, x -> bit 0, y -> bit 1, z -> bit 2.
// Reference: include/rakau/detail/libmorton/morton3D.h:38-50 (as used at tree.hpp:222-242).
inline u64 spread3(u64 v)
{
    v &= 0x1fffffULL;
    v = (v | (v << 32)) & 0x1f00000000ffffULL;
    v = (v | (v << 16)) & 0x1f0000ff0000ffULL;
    v = (v | (v << 8)) & 0x100f00f00f00f00fULL;
    v = (v | (v << 4)) & 0x10c30c30c30c30c3ULL;
    v = (v | (v << 2)) & 0x1249249249249249ULL;
    return v;
}
inline u64 compact3(u64 v)
{
    v &= 0x1249249249249249ULL;
    v = (v ^ (v >> 2)) & 0x10c30c30c30c30c3ULL;
    v = (v ^ (v >> 4)) & 0x100f00f00f00f00fULL;
    v = (v ^ (v >> 8)) & 0x1f0000ff0000ffULL;
    v = (v ^ (v >> 16)) & 0x1f00000000ffffULL;
    v = (v ^ (v >> 32)) & 0x1fffffULL;
    return v;
}
// 2D Morton encoding of two 31-bit (in general, up to 32-bit) coordinates, x -> bit 0, y -> bit 1.
// Reference: include/rakau/detail/libmorton/morton2D.h (magic-bits flavour), as used at tree.hpp:207-220.
inline u64 spread2(u64 v)
{
    v &= 0xffffffffULL;
    v = (v | (v << 16)) & 0x0000ffff0000ffffULL;
    v = (v | (v << 8)) & 0x00ff00ff00ff00ffULL;
    v = (v | (v << 4)) & 0x0f0f0f0f0f0f0f0fULL;
    v = (v | (v << 2)) & 0x3333333333333333ULL;
    v = (v | (v << 1)) & 0x5555555555555555ULL;
    return v;
}
inline u64 compact2(u64 v)
{
    v &= 0x5555555555555555ULL;
    v = (v ^ (v >> 1)) & 0x3333333333333333ULL;
    v = (v ^ (v >> 2)) & 0x0f0f0f0f0f0f0f0fULL;
    v = (v ^ (v >> 4)) & 0x00ff00ff00ff00ffULL;
    v = (v ^ (v >> 8)) & 0x0000ffff0000ffffULL;
    v = (v ^ (v >> 16)) & 0xffffffffULL;
    return v;
}
template <unsigned ND>
inline u64 morton_encode(const u64 *d)
{
    if constexpr (ND == 3) {
        return spread3(d[0]) | (spread3(d[1]) << 1) | (spread3(d[2]) << 2);
    } else {
        return spread2(d[0]) | (spread2(d[1]) << 1);
    }
}
// Coordinate j of a Morton code.
template <unsigned ND>
inline u64 morton_coord(u64 code, unsigned j)
{
    if constexpr (ND == 3) {
        return compact3(code >> j);
    } else {
        return compact2(code >> j);
    }
}

// Level of a nodal code. Reference: tree_fwd.hpp:212-228.
template <unsigned ND>
inline unsigned tree_level(u64 n)
{
    return (63u - static_cast<unsigned>(__builtin_clzll(n))) / ND;
}

template <typename F, unsigned ND>
struct node_t {
    // Reference: tree_fwd.hpp:77-116 (begin,end,n_children,code,level,props[NDim+1], dim2 | dim,delta).
    u64 begin, end, n_children, code, level;
    F props[ND + 1];
    F dim; // dim2 for mac==bh, dim for mac==bh_geom
    F delta; // bh_geom only
};

struct cnode_t {
    // Reference: tree_fwd.hpp:119-125.
    u64 code, begin, end;
};

template <typename F, unsigned ND>
struct tree_t {
    static constexpr unsigned NDim = ND;
    unsigned cbits = cbits_for(64, ND); // bits per coordinate; set before construct() for 32-bit codes
    using fp_type = F;
    int mac = 0; // 0 = bh, 1 = bh_geom
    F box_size = 0;
    bool box_deduced = false;
    u64 max_leaf_n = 16, ncrit = 128;
    std::vector<F> parts[ND + 1]; // x, y, (z,) m in Morton order
    std::vector<u64> codes, perm, last_perm, inv_perm;
    std::vector<node_t<F, ND>> nodes;
    std::vector<cnode_t> crit;

    u64 nparts() const
    {
        return parts[0].size();
    }

    // Reference: tree.hpp:381-429 (disc_single_coord, Clamp == false).
    u64 disc_single_coord(F x, F inv_box_size) const
    {
        const u64 factor = u64(1) << cbits;
        F tmp = std::fma(x, inv_box_size, F(1) / F(2));
        tmp *= F(factor);
        if (!std::isfinite(tmp)) {
            throw std::invalid_argument("While trying to discretise the input coordinate " + std::to_string(x)
                                        + " in a box of size " + std::to_string(F(1) / inv_box_size)
                                        + ", the non-finite value " + std::to_string(tmp) + " was generated");
        }
        if (tmp < F(0) || tmp >= F(factor)) {
            throw std::invalid_argument("The discretisation of the input coordinate " + std::to_string(x)
                                        + " in a box of size " + std::to_string(F(1) / inv_box_size)
                                        + " produced the floating-point value " + std::to_string(tmp)
                                        + ", which is outside the allowed bounds");
        }
        auto retval = static_cast<u64>(tmp);
        if (retval >= factor) {
            throw std::invalid_argument("The discretisation of the input coordinate produced an integral value "
                                        "which is outside the allowed bounds");
        }
        return retval;
    }

    // Reference: tree.hpp:444-482 (get_node_dim, get_node_centre).
    static F get_node_dim(u64 level, F box)
    {
        return box / static_cast<F>(u64(1) << level);
    }
    void get_node_centre(F (&out)[ND], u64 code) const
    {
        const auto level = tree_level<ND>(code);
        const u64 c_code = (code - (u64(1) << (level * NDim))) << ((cbits - level) * NDim);
        const F node_dim_2 = get_node_dim(level, box_size) * (F(1) / F(2));
        const F cell_size = box_size * (F(1) / static_cast<F>(u64(1) << cbits));
        for (unsigned j = 0; j < ND; ++j) {
            out[j] = std::fma(static_cast<F>(morton_coord<ND>(c_code, j)), cell_size,
                              node_dim_2 - box_size * (F(1) / F(2)));
        }
    }

    // Reference: tree.hpp:1116-1237 (compute_node_properties, scalar branch 1162-1168). With orc_set_simd_width(W > 1) the
    // octree sums follow the association of the reference's SIMD branch (tree.hpp:1134-1161): W interleaved partial sums
    // over the first size - size % W particles, added horizontally (pairwise, as xsimd::hadd does on AVX), then the scalar
    // tail -- the node properties a default (SIMD-enabled) build of the reference produces for batch_size = W.
    void compute_node_properties(node_t<F, ND> &node) const
    {
        const auto begin = node.begin, end = node.end;
        F tot_mass(0), com[ND] = {};
        u64 i = begin;
        const unsigned W = g_simd_width;
        if (W > 1 && ND == 3) {
            const u64 size = end - begin, vec_end = begin + (size - size % W);
            F acc[ND + 1][16] = {};
            for (; i < vec_end; i += W) {
                for (unsigned l = 0; l < W; ++l) {
                    const F mass = parts[ND][i + l];
                    acc[ND][l] += mass;
                    for (unsigned j = 0; j < ND; ++j) {
                        acc[j][l] = std::fma(mass, parts[j][i + l], acc[j][l]);
                    }
                }
            }
            for (unsigned j = 0; j <= ND; ++j) {
                for (unsigned w = 1; w < W; w <<= 1) {
                    for (unsigned l = 0; l + w < W; l += 2 * w) {
                        acc[j][l] += acc[j][l + w];
                    }
                }
            }
            tot_mass = acc[ND][0];
            for (unsigned j = 0; j < ND; ++j) {
                com[j] = acc[j][0];
            }
        }
        for (; i < end; ++i) {
            const F mass = parts[ND][i];
            tot_mass += mass;
            for (unsigned j = 0; j < ND; ++j) {
                com[j] = std::fma(mass, parts[j][i], com[j]);
            }
        }
        F geo[ND] = {};
        if (mac == 1) {
            get_node_centre(geo, node.code);
        }
        if (tot_mass == F(0)) {
            if (mac == 0) {
                get_node_centre(com, node.code);
            } else {
                std::copy(geo, geo + ND, com);
            }
        } else {
            const F inv = F(1) / tot_mass;
            for (unsigned j = 0; j < ND; ++j) {
                com[j] *= inv;
            }
        }
        for (unsigned j = 0; j < ND; ++j) {
            if (!std::isfinite(com[j])) {
                throw std::invalid_argument(
                    "The computation of the centre of mass of a node produced a non-finite value");
            }
            node.props[j] = com[j];
        }
        if (!std::isfinite(tot_mass)) {
            throw std::invalid_argument("The computation of the total mass in a node produced the non-finite value "
                                        + std::to_string(tot_mass));
        }
        node.props[ND] = tot_mass;
        const F node_dim = get_node_dim(node.level, box_size);
        if (mac == 0) {
            node.dim = node_dim * node_dim;
            node.delta = F(0);
            if (!std::isfinite(node.dim)) {
                throw std::invalid_argument(
                    "The computation of the square of the dimension of a node produced a non-finite value");
            }
        } else {
            node.dim = node_dim;
            F delta2 = (com[0] - geo[0]) * (com[0] - geo[0]);
            for (unsigned j = 1; j < ND; ++j) {
                delta2 = std::fma(com[j] - geo[j], com[j] - geo[j], delta2);
            }
            node.delta = std::sqrt(delta2);
            if (!std::isfinite(node.dim) || !std::isfinite(node.delta)) {
                throw std::invalid_argument("The computation of the dimension of a node produced a non-finite value");
            }
        }
    }

    // Reference: tree.hpp:723-833 (build_tree_ser_impl). Depth-first construction: children of the node
    // `parent_code` at `parent_level` whose particles have codes in [begin, end). Returns the number
    // of descendants appended.
    u64 build_children(u64 parent_level, u64 parent_code, u64 begin, u64 end, bool crit_ancestor)
    {
        if (parent_level >= cbits) {
            return 0;
        }
        u64 retval = 0;
        const u64 node_prefix = parent_code - (u64(1) << (parent_level * NDim));
        const unsigned shift = (cbits - static_cast<unsigned>(parent_level) - 1u) * NDim;
        const u64 *cb = codes.data();
        u64 cur = begin;
        for (u64 i = 0; i < (u64(1) << NDim); ++i) {
            // equal_range of (node_prefix << NDim) + i on the shifted codes (tree.hpp:763-764).
            const u64 key = (node_prefix << NDim) + i;
            const u64 *lo = std::lower_bound(cb + cur, cb + end, key,
                                             [shift](u64 c, u64 k) { return (c >> shift) < k; });
            const u64 *hi = std::upper_bound(lo, cb + end, key, [shift](u64 k, u64 c) { return k < (c >> shift); });
            const u64 it_start = static_cast<u64>(lo - cb), it_end = static_cast<u64>(hi - cb);
            const u64 npart = it_end - it_start;
            cur = it_end;
            if (!npart) {
                continue;
            }
            node_t<F, ND> nn{};
            nn.begin = it_start;
            nn.end = it_end;
            nn.n_children = 0;
            nn.code = (parent_code << NDim) + i;
            nn.level = parent_level + 1u;
            compute_node_properties(nn);
            nodes.push_back(nn);
            const auto idx = nodes.size() - 1u;
            // tree.hpp:801-807.
            const bool critical_node
                = !crit_ancestor && (npart <= ncrit || npart <= max_leaf_n || parent_level + 1u == cbits);
            if (critical_node) {
                crit.push_back({nn.code, nn.begin, nn.end});
            }
            if (npart > max_leaf_n) {
                const u64 cc = build_children(parent_level + 1u, nn.code, it_start, it_end,
                                              critical_node || crit_ancestor);
                nodes[idx].n_children = cc;
            }
            retval += nodes[idx].n_children + 1u;
        }
        return retval;
    }

    // Reference: tree.hpp:932-1111 (build_tree). The parallel variant concatenates subtrees in nodal-code
    // (= depth-first) order, so a serial depth-first build yields the identical array.
    void build_tree()
    {
        nodes.clear();
        crit.clear();
        const u64 np = codes.size();
        if (!np) {
            return;
        }
        node_t<F, ND> root{};
        root.begin = 0;
        root.end = np;
        root.code = 1;
        root.level = 0;
        nodes.push_back(root);
        compute_node_properties(nodes[0]);
        const bool root_is_crit = np <= ncrit || np <= max_leaf_n;
        if (root_is_crit) {
            crit.push_back({u64(1), u64(0), np});
        }
        if (np > max_leaf_n) {
            const u64 cc = build_children(0, 1, 0, np, root_is_crit);
            nodes[0].n_children = cc;
        }
    }

    // Reference: tree.hpp:1279-1319 (determine_box_size).
    static F determine_box_size(const F *const *c, u64 n)
    {
        F mx(0);
        for (unsigned j = 0; j < ND; ++j) {
            for (u64 i = 0; i < n; ++i) {
                const F tmp = std::abs(c[j][i]);
                if (!std::isfinite(tmp)) {
                    throw std::invalid_argument("While trying to automatically determine the domain size, a "
                                                "non-finite coordinate with absolute value "
                                                + std::to_string(tmp) + " was encountered");
                }
                mx = std::max(mx, tmp);
            }
        }
        F retval = mx * F(2);
        retval = std::fma(retval, F(1) / F(20), retval);
        if (!std::isfinite(retval)) {
            throw std::invalid_argument("The automatic deduction of the domain size produced the non-finite value "
                                        + std::to_string(retval));
        }
        return retval;
    }

    // Reference: tree.hpp:1330-1487 (construct_impl).
    // src: the ND coordinate arrays followed by the masses.
    void construct(const F *const *src, u64 n, F box, u64 mln, u64 nc)
    {
        box_size = box;
        box_deduced = (box == F(0));
        max_leaf_n = mln;
        ncrit = nc;
        if (!std::isfinite(box_size) || box_size < F(0)) {
            throw std::invalid_argument("The box size must be a finite non-negative value, but it is "
                                        + std::to_string(box) + " instead");
        }
        if (!mln) {
            throw std::invalid_argument("The maximum number of particles per leaf must be nonzero");
        }
        if (!nc) {
            throw std::invalid_argument("The critical number of particles for the vectorised computation of the "
                                        "potentials/accelerations must be nonzero");
        }
        for (unsigned j = 0; j < ND + 1; ++j) {
            parts[j].assign(src[j], src[j] + n);
        }
        codes.resize(n);
        perm.resize(n);
        last_perm.resize(n);
        inv_perm.resize(n);
        std::iota(perm.begin(), perm.end(), u64(0));
        if (box_deduced) {
            const F *c[ND];
            for (unsigned j = 0; j < ND; ++j) {
                c[j] = parts[j].data();
            }
            box_size = determine_box_size(c, n);
        }
        sort_and_build();
    }

    // Morton encode + indirect sort + permute + build (tree.hpp:1435-1486; also the tail of
    // sync, tree.hpp:3678-3743). NOTE: the reference uses tbb::parallel_sort, which is not stable;
    // the order of particles with identical codes is unspecified there. A stable sort is one valid
    // instance of it.
    void sort_and_build()
    {
        const u64 n = nparts();
        const F inv_box_size = F(1) / box_size;
        for (u64 i = 0; i < n; ++i) {
            u64 d[ND];
            for (unsigned j = 0; j < ND; ++j) {
                d[j] = disc_single_coord(parts[j][i], inv_box_size);
            }
            codes[i] = morton_encode<ND>(d);
        }
        std::vector<u64> idx(n);
        std::iota(idx.begin(), idx.end(), u64(0));
        std::stable_sort(idx.begin(), idx.end(), [this](u64 a, u64 b) { return codes[a] < codes[b]; });
        // apply_isort (tree.hpp:493-507).
        auto apply = [&idx, n](auto &v) {
            auto nv = v;
            for (u64 i = 0; i < n; ++i) {
                nv[i] = v[idx[i]];
            }
            v = std::move(nv);
        };
        apply(codes);
        for (unsigned j = 0; j < ND + 1; ++j) {
            apply(parts[j]);
        }
        // On construction perm == iota, so perm becomes idx; in general perm is permuted by idx
        // (tree.hpp:3712-3727).
        apply(perm);
        last_perm = idx;
        for (u64 i = 0; i < n; ++i) {
            inv_perm[perm[i]] = i;
        }
        build_tree();
    }

    // ---------------------------------------------------------------------------------------------
    // Traversal (the hot path).
    // ---------------------------------------------------------------------------------------------

    struct scratch_t {
        std::vector<F> tgt[ND + 1], res[ND + 1], tmp[ND + 2];
    };

    // Per-group statistics used for the roofline's algorithmic work count.
    struct stats_t {
        // Per-group counts: MAC evaluations, accepted nodes, opened leaves, source particles of opened
        // leaves, unordered pairs inside the group. w_*: the same weighted by the group size, i.e.
        // particle-level interaction counts (targets x sources); w_self counts ordered pairs.
        u64 visits = 0, com = 0, leaves = 0, pp = 0, self_pairs = 0, w_visits = 0, w_com = 0, w_pp = 0, w_self = 0;
    };

    // Reference: tree.hpp:2073-2321 (tree_self_interactions), scalar branch 2258-2320.
    template <unsigned Q>
    static void self_interactions(F eps2, u64 tgt_size, const F *const *p, F *const *res)
    {
        constexpr unsigned nres = Q == 0 ? ND : (Q == 1 ? 1 : ND + 1);
        constexpr unsigned pot_idx = Q == 1 ? 0 : ND;
        const F *m_ptr = p[ND];
        F diffs[ND], pos1[ND];
        for (u64 i1 = 0; i1 < tgt_size; ++i1) {
            for (unsigned j = 0; j < ND; ++j) {
                pos1[j] = p[j][i1];
            }
            const F m1 = m_ptr[i1];
            F a1[nres];
            for (unsigned j = 0; j < nres; ++j) {
                a1[j] = F(0);
            }
            for (u64 i2 = i1 + 1u; i2 < tgt_size; ++i2) {
                F dist2(eps2);
                for (unsigned j = 0; j < ND; ++j) {
                    diffs[j] = p[j][i2] - pos1[j];
                    dist2 = std::fma(diffs[j], diffs[j], dist2);
                }
                const F dist = std::sqrt(dist2), m2 = m_ptr[i2];
                if constexpr (Q == 0 || Q == 2) {
                    const F dist3 = dist2 * dist, m2_dist3 = m2 / dist3, m1_dist3 = m1 / dist3;
                    for (unsigned j = 0; j < ND; ++j) {
                        a1[j] = std::fma(m2_dist3, diffs[j], a1[j]);
                        res[j][i2] = std::fma(m1_dist3, -diffs[j], res[j][i2]);
                    }
                }
                if constexpr (Q == 1 || Q == 2) {
                    const F mut_pot = m1 / dist * m2;
                    a1[pot_idx] -= mut_pot;
                    res[pot_idx][i2] -= mut_pot;
                }
            }
            if constexpr (Q == 0 || Q == 2) {
                for (unsigned j = 0; j < ND; ++j) {
                    res[j][i1] += a1[j];
                }
            }
            if constexpr (Q == 1 || Q == 2) {
                res[pot_idx][i1] += a1[pot_idx];
            }
        }
    }

    // Reference: tree.hpp:2327-2471 (tree_acc_pot_leaf), scalar branch 2432-2470.
    template <unsigned Q>
    void leaf_interactions(F eps2, const node_t<F, ND> &src, u64 tgt_size, const F *const *p, F *const *res) const
    {
        constexpr unsigned pot_idx = Q == 1 ? 0 : ND;
        F pos1[ND], diffs[ND];
        for (u64 i1 = 0; i1 < tgt_size; ++i1) {
            for (unsigned j = 0; j < ND; ++j) {
                pos1[j] = p[j][i1];
            }
            F m1 = F(0);
            if constexpr (Q == 1 || Q == 2) {
                m1 = p[ND][i1];
            }
            for (u64 i2 = src.begin; i2 < src.end; ++i2) {
                F dist2(eps2);
                for (unsigned j = 0; j < ND; ++j) {
                    diffs[j] = parts[j][i2] - pos1[j];
                    dist2 = std::fma(diffs[j], diffs[j], dist2);
                }
                const F dist = std::sqrt(dist2), m2 = parts[ND][i2];
                if constexpr (Q == 0 || Q == 2) {
                    const F dist3 = dist * dist2, m_dist3 = m2 / dist3;
                    for (unsigned j = 0; j < ND; ++j) {
                        res[j][i1] = std::fma(diffs[j], m_dist3, res[j][i1]);
                    }
                }
                if constexpr (Q == 1 || Q == 2) {
                    res[pot_idx][i1] = std::fma(-m1, m2 / dist, res[pot_idx][i1]);
                }
            }
        }
    }

    // Reference: tree.hpp:2597-2793 (tree_acc_pot_mac_check, scalar branch 2740-2777) fused with
    // tree.hpp:2477-2590 (tree_acc_pot_src_com, scalar branch 2564-2589).
    // Returns the index of the next node in the traversal.
    template <unsigned Q>
    u64 mac_check(u64 src_idx, F mac_value, F eps2, u64 tgt_size, const F *const *p, F *const *res, F *const *tmp,
                  stats_t *st) const
    {
        constexpr unsigned pot_idx = Q == 1 ? 0 : ND;
        constexpr unsigned dist_idx = Q == 1 ? 0 : ND + 1; // tmp: ND differences, dist^3, dist
        const auto &src = nodes[src_idx];
        const u64 n_children_src = src.n_children;
        // tree.hpp:2632-2642.
        F mac_lh;
        if (mac == 0) {
            mac_lh = src.dim * mac_value;
        } else {
            const F t = std::fma(src.dim, mac_value, src.delta);
            mac_lh = t * t;
        }
        bool mac_flag = true;
        for (u64 i = 0; i < tgt_size; ++i) {
            F dist2(0);
            for (unsigned j = 0; j < ND; ++j) {
                const F diff = src.props[j] - p[j][i];
                if constexpr (Q == 0 || Q == 2) {
                    tmp[j][i] = diff;
                }
                dist2 = std::fma(diff, diff, dist2);
            }
            if (mac_lh >= dist2) {
                mac_flag = false;
                break;
            }
            dist2 += eps2;
            const F dist = std::sqrt(dist2);
            if constexpr (Q == 0 || Q == 2) {
                tmp[ND][i] = dist * dist2;
            }
            if constexpr (Q == 1 || Q == 2) {
                tmp[dist_idx][i] = dist;
            }
        }
        if (st) {
            ++st->visits;
            st->w_visits += tgt_size;
        }
        if (mac_flag) {
            // tree.hpp:2564-2589.
            const F m_src = src.props[ND];
            for (u64 i = 0; i < tgt_size; ++i) {
                if constexpr (Q == 0 || Q == 2) {
                    const F m_src_dist3 = m_src / tmp[ND][i];
                    for (unsigned j = 0; j < ND; ++j) {
                        res[j][i] = std::fma(tmp[j][i], m_src_dist3, res[j][i]);
                    }
                }
                if constexpr (Q == 1 || Q == 2) {
                    res[pot_idx][i] = std::fma(-p[ND][i], m_src / tmp[dist_idx][i], res[pot_idx][i]);
                }
            }
            if (st) {
                ++st->com;
                st->w_com += tgt_size;
            }
            return src_idx + n_children_src + 1u;
        }
        if (!n_children_src) {
            leaf_interactions<Q>(eps2, src, tgt_size, p, res);
            if (st) {
                ++st->leaves;
                st->pp += src.end - src.begin;
                st->w_pp += tgt_size * (src.end - src.begin);
            }
        }
        return src_idx + 1u;
    }

    // Reference: tree.hpp:2798-2849 (tree_acc_pot).
    template <unsigned Q>
    void tree_acc_pot(F mac_value, F eps2, u64 tgt_size, u64 tgt_code, const F *const *p, F *const *res,
                      F *const *tmp, stats_t *st) const
    {
        const u64 tgt_level = tree_level<ND>(tgt_code);
        const u64 tree_size = nodes.size();
        for (u64 src_idx = 0; src_idx < tree_size;) {
            const auto &src = nodes[src_idx];
            const u64 src_code = src.code, n_children_src = src.n_children, src_level = src.level;
            // NOTE: the reference evaluates tgt_code >> ((tgt_level - src_level) * NDim) with unsigned
            // wrap-around when src_level > tgt_level (tree.hpp:2828); that never compares equal on x86.
            // Here the level test is explicit.
            const bool anc_or_self
                = src_level <= tgt_level && (tgt_code >> ((tgt_level - src_level) * NDim)) == src_code;
            if (anc_or_self) {
                src_idx += 1u + (src_code == tgt_code ? n_children_src : 0u);
            } else {
                src_idx = mac_check<Q>(src_idx, mac_value, eps2, tgt_size, p, res, tmp, st);
            }
        }
        self_interactions<Q>(eps2, tgt_size, p, res);
        if (st) {
            st->self_pairs += tgt_size * (tgt_size - 1u) / 2u;
            st->w_self += tgt_size * (tgt_size - 1u);
        }
    }

    // Reference: tree.hpp:2871-3022 (cpu_run) for critical nodes [c_begin, c_end), with the output
    // either in Morton order (ordered == false) or scattered through m_perm (tree.hpp:3320-3330).
    template <unsigned Q>
    void cpu_run(u64 c_begin, u64 c_end, F *const *out, bool ordered, F mac_value, F G, F eps2, unsigned nthreads,
                 stats_t *stats_out) const
    {
        constexpr unsigned nres = Q == 0 ? ND : (Q == 1 ? 1 : ND + 1);
        std::atomic<u64> next(c_begin);
        std::vector<stats_t> tstats(nthreads);
        auto worker = [&](unsigned tid) {
            scratch_t s;
            const u64 chunk = 16;
            for (;;) {
                const u64 b = next.fetch_add(chunk);
                if (b >= c_end) {
                    break;
                }
                const u64 e = std::min(c_end, b + chunk);
                for (u64 ci = b; ci < e; ++ci) {
                    const u64 tgt_code = crit[ci].code, tgt_begin = crit[ci].begin,
                              tgt_size = crit[ci].end - tgt_begin;
                    const F *p[ND + 1];
                    F *res[ND + 1] = {};
                    F *tmp[ND + 2];
                    for (unsigned j = 0; j < ND + 1; ++j) {
                        s.tgt[j].assign(parts[j].data() + tgt_begin, parts[j].data() + tgt_begin + tgt_size);
                        p[j] = s.tgt[j].data();
                    }
                    for (unsigned j = 0; j < nres; ++j) {
                        s.res[j].assign(tgt_size, F(0));
                        res[j] = s.res[j].data();
                    }
                    for (unsigned j = 0; j < ND + 2; ++j) {
                        s.tmp[j].resize(tgt_size);
                        tmp[j] = s.tmp[j].data();
                    }
                    tree_acc_pot<Q>(mac_value, eps2, tgt_size, tgt_code, p, res, tmp,
                                    stats_out ? &tstats[tid] : nullptr);
                    // tree.hpp:2986-3002.
                    if (G != F(1)) {
                        for (unsigned j = 0; j < nres; ++j) {
                            for (u64 k = 0; k < tgt_size; ++k) {
                                res[j][k] *= G;
                            }
                        }
                    }
                    // tree.hpp:3004-3007.
                    for (unsigned j = 0; j < nres; ++j) {
                        if (ordered) {
                            for (u64 k = 0; k < tgt_size; ++k) {
                                out[j][perm[tgt_begin + k]] = res[j][k];
                            }
                        } else {
                            std::copy(res[j], res[j] + tgt_size, out[j] + tgt_begin);
                        }
                    }
                }
            }
        };
        if (nthreads <= 1) {
            worker(0);
        } else {
            std::vector<std::thread> th;
            for (unsigned t = 0; t < nthreads; ++t) {
                th.emplace_back(worker, t);
            }
            for (auto &t : th) {
                t.join();
            }
        }
        if (stats_out) {
            for (auto &t : tstats) {
                stats_out->visits += t.visits;
                stats_out->com += t.com;
                stats_out->leaves += t.leaves;
                stats_out->pp += t.pp;
                stats_out->self_pairs += t.self_pairs;
                stats_out->w_visits += t.w_visits;
                stats_out->w_com += t.w_com;
                stats_out->w_pp += t.w_pp;
                stats_out->w_self += t.w_self;
            }
        }
    }

    // Reference: tree.hpp:3293-3334 (acc_pot_dispatch): validation and theta -> mac_value transform.
    template <unsigned Q>
    void acc_pot(F *const *out, bool ordered, F theta, F G, F eps, unsigned nthreads, u64 c_begin, u64 c_end,
                 stats_t *st) const
    {
        if (!std::isfinite(theta) || theta <= F(0)) {
            throw std::domain_error("The MAC value must be finite and positive, but it is " + std::to_string(theta)
                                    + " instead");
        }
        const F mac_value = mac == 0 ? F(1) / (theta * theta) : F(1) / theta;
        if (!std::isfinite(mac_value) || mac_value <= F(0)) {
            throw std::domain_error("The transformed MAC value must be finite and positive, but it is "
                                    + std::to_string(mac_value) + " instead");
        }
        // tree.hpp:3268-3281.
        if (!std::isfinite(eps) || eps < F(0)) {
            throw std::domain_error("The softening length must be finite and non-negative, but it is "
                                    + std::to_string(eps) + " instead");
        }
        const F eps2 = eps * eps;
        if (!std::isfinite(eps2) || eps2 < F(0)) {
            throw std::domain_error("The square of the softening length must be finite and non-negative, but it is "
                                    + std::to_string(eps2) + " instead");
        }
        // tree.hpp:3283-3289.
        if (!std::isfinite(G)) {
            throw std::domain_error("The value of the gravitational constant G must be finite, but it is "
                                    + std::to_string(G) + " instead");
        }
        c_end = std::min<u64>(c_end, crit.size());
        cpu_run<Q>(c_begin, c_end, out, ordered, mac_value, G, eps2, nthreads, st);
    }

    // Reference: tree.hpp:3531-3569 (exact_acc_pot_impl).
    template <unsigned Q>
    void exact(F *retval, bool ordered, u64 orig_idx, F G, F eps) const
    {
        constexpr unsigned nres = Q == 0 ? ND : (Q == 1 ? 1 : ND + 1);
        constexpr unsigned pot_idx = Q == 1 ? 0 : ND;
        if (!std::isfinite(eps) || eps < F(0)) {
            throw std::domain_error("The softening length must be finite and non-negative");
        }
        const F eps2 = eps * eps;
        if (!std::isfinite(G)) {
            throw std::domain_error("The value of the gravitational constant G must be finite");
        }
        const u64 size = nparts();
        for (unsigned j = 0; j < nres; ++j) {
            retval[j] = F(0);
        }
        F diffs[ND];
        const u64 idx = ordered ? inv_perm[orig_idx] : orig_idx;
        for (u64 i = 0; i < size; ++i) {
            if (i == idx) {
                continue;
            }
            F dist2(eps2);
            for (unsigned j = 0; j < ND; ++j) {
                diffs[j] = parts[j][i] - parts[j][idx];
                dist2 = std::fma(diffs[j], diffs[j], dist2);
            }
            const F inv_dist = F(1) / std::sqrt(dist2), Gmi_dist = G * parts[ND][i] * inv_dist;
            if constexpr (Q == 0 || Q == 2) {
                const F Gmi_dist3 = inv_dist * inv_dist * Gmi_dist;
                for (unsigned j = 0; j < ND; ++j) {
                    retval[j] = std::fma(diffs[j], Gmi_dist3, retval[j]);
                }
            }
            if constexpr (Q == 1 || Q == 2) {
                retval[pot_idx] = std::fma(-Gmi_dist, parts[ND][idx], retval[pot_idx]);
            }
        }
    }
};

// Reference: benchmark/common.hpp:39-126 (get_plummer_sphere, serial branch 95-124) with the
// default-seeded thread-local std::mt19937 of common.hpp:36. Output layout: m, x, y, z blocks of n.
template <typename F>
void plummer(F *retval, u64 n, F a, F size, std::uint32_t seed)
{
    std::mt19937 rng(seed);
    if (!std::isfinite(a) || a <= F(0)) {
        throw std::invalid_argument("The Plummer 'a' parameter must be finite and positive");
    }
    if (!std::isfinite(size) || size < F(0)) {
        throw std::invalid_argument("The Plummer 'size' parameter must be finite and non-negative");
    }
    const F size_limit = (size > F(0)) ? (size / F(2) - size / F(100)) : std::numeric_limits<F>::infinity();
    auto check_bounds = [size_limit](F x, F y, F z) {
        return x >= -size_limit && x < size_limit && y >= -size_limit && y < size_limit && z >= -size_limit
               && z < size_limit;
    };
    std::uniform_real_distribution<F> udist(F(0), F(1));
    std::uniform_real_distribution<F> mdist(F(0.1), F(1.9));
    std::generate(retval, retval + n, [&]() { return mdist(rng); });
    // boost::math::constants::pi<F>() == pi rounded to F.
    const F pi = static_cast<F>(3.141592653589793238462643383279502884L);
    for (u64 i = 0; i < n;) {
        F r;
        do {
            r = a / std::sqrt(std::pow(udist(rng), F(-2) / F(3)) - F(1));
        } while (!std::isfinite(r));
        const F u = udist(rng), v = udist(rng);
        const F lon = std::clamp(F(2) * pi * u, F(0), F(2) * pi);
        const F colat = std::acos(std::clamp(F(2) * v - F(1), F(-1), F(1)));
        const F x = r * std::cos(lon) * std::sin(colat), y = r * std::sin(lon) * std::sin(colat),
                z = r * std::cos(colat);
        if (check_bounds(x, y, z)) {
            retval[n + i] = x;
            retval[2 * n + i] = y;
            retval[3 * n + i] = z;
            ++i;
        }
    }
}

// Reference: test/test_utils.hpp:41-59 (get_uniform_particles<NDim>): masses U[0,1) then the NDim*n coordinates
// U[-size/2, size/2), all drawn from one engine. The engine is passed by the caller so that a sequence of
// calls continues one stream, as the reference tests do with their file-static rng.
template <typename F>
void uniform_particles(F *retval, u64 n, F size, std::mt19937 &rng, unsigned ndim)
{
    std::uniform_real_distribution<F> mdist(F(0), F(1));
    std::generate(retval, retval + n, [&]() { return mdist(rng); });
    std::uniform_real_distribution<F> rdist(-size / F(2), size / F(2));
    std::generate(retval + n, retval + (ndim + 1) * n, [&]() { return rdist(rng); });
}

struct handle_t {
    int fp;   // 0 = float, 1 = double
    int ndim; // 2 or 3
    std::unique_ptr<tree_t<float, 3>> tf3;
    std::unique_ptr<tree_t<double, 3>> td3;
    std::unique_ptr<tree_t<float, 2>> tf2;
    std::unique_ptr<tree_t<double, 2>> td2;
};

// Calls f(tree) on whichever tree the handle holds.
template <typename H, typename Fn>
void visit(H *h, Fn &&f)
{
    if (h->ndim == 3) {
        if (h->fp == 0) {
            f(*h->tf3);
        } else {
            f(*h->td3);
        }
    } else {
        if (h->fp == 0) {
            f(*h->tf2);
        } else {
            f(*h->td2);
        }
    }
}

template <typename Fn>
int guard(Fn &&f)
{
    try {
        f();
        return 0;
    } catch (const std::domain_error &e) {
        g_last_error = e.what();
        return 2;
    } catch (const std::invalid_argument &e) {
        g_last_error = e.what();
        return 1;
    } catch (const std::overflow_error &e) {
        g_last_error = e.what();
        return 3;
    } catch (const std::exception &e) {
        g_last_error = e.what();
        return 4;
    }
}

template <typename T>
void acc_pot_q(const T &t, int q, void *const *out, int ordered, double theta, double G, double eps, unsigned nthreads,
               u64 c_begin, u64 c_end, u64 *stats)
{
    using F = typename T::fp_type;
    F *o[4] = {static_cast<F *>(out[0]), static_cast<F *>(out[1]), static_cast<F *>(out[2]),
               static_cast<F *>(out[3])};
    typename T::stats_t st;
    auto *sp = stats ? &st : nullptr;
    switch (q) {
        case 0:
            t.template acc_pot<0>(o, ordered, F(theta), F(G), F(eps), nthreads, c_begin, c_end, sp);
            break;
        case 1:
            t.template acc_pot<1>(o, ordered, F(theta), F(G), F(eps), nthreads, c_begin, c_end, sp);
            break;
        case 2:
            t.template acc_pot<2>(o, ordered, F(theta), F(G), F(eps), nthreads, c_begin, c_end, sp);
            break;
        default:
            throw std::invalid_argument("q must be 0, 1 or 2");
    }
    if (stats) {
        stats[0] = st.visits;
        stats[1] = st.com;
        stats[2] = st.leaves;
        stats[3] = st.pp;
        stats[4] = st.self_pairs;
        stats[5] = st.w_visits;
        stats[6] = st.w_com;
        stats[7] = st.w_pp;
        stats[8] = st.w_self;
    }
}

} // namespace

extern "C" {

const char *orc_last_error()
{
    return g_last_error.c_str();
}

// fp: 0 = float, 1 = double.  out: 4*n values laid out m | x | y | z (benchmark/benchmark_acc.cpp:44-52).
int orc_plummer(int fp, void *out, u64 n, double a, double size, unsigned seed)
{
    return guard([&] {
        if (fp == 0) {
            plummer<float>(static_cast<float *>(out), n, float(a), float(size), seed);
        } else {
            plummer<double>(static_cast<double *>(out), n, a, size, seed);
        }
    });
}

// A persistent engine so that consecutive calls continue one stream (test/accuracy_acc.cpp:39).
void *orc_rng_create(unsigned seed)
{
    return new std::mt19937(seed);
}
void orc_rng_destroy(void *r)
{
    delete static_cast<std::mt19937 *>(r);
}
// out: (ndim + 1) * n values laid out m | x | y (| z).
int orc_uniform_nd(int ndim, int fp, void *out, u64 n, double size, void *rng)
{
    return guard([&] {
        if (ndim != 2 && ndim != 3) {
            throw std::invalid_argument("ndim must be 2 or 3");
        }
        auto &r = *static_cast<std::mt19937 *>(rng);
        if (fp == 0) {
            uniform_particles<float>(static_cast<float *>(out), n, float(size), r, unsigned(ndim));
        } else {
            uniform_particles<double>(static_cast<double *>(out), n, size, r, unsigned(ndim));
        }
    });
}
int orc_uniform(int fp, void *out, u64 n, double size, void *rng)
{
    return orc_uniform_nd(3, fp, out, n, size, rng);
}

// Association of the octree node sums of trees built from now on: 1 = the reference's scalar build (default), 4 / 8 / 16 =
// its SIMD build with that batch size (see compute_node_properties). Process-wide; for the envelope test of the device builder.
void orc_set_simd_width(int w)
{
    g_simd_width = (w == 4 || w == 8 || w == 16) ? static_cast<unsigned>(w) : 1u;
}

// mac: 0 = bh, 1 = bh_geom. box_size == 0 -> deduced. src: the ndim coordinate arrays followed by the masses.
// Returns nullptr on error (see orc_last_error()).
void *orc_tree_create_ex(int ndim, int code_bits, int fp, int mac, const void *const *src, u64 n, double box_size,
                         u64 max_leaf_n, u64 ncrit, int *status);

void *orc_tree_create_nd(int ndim, int fp, int mac, const void *const *src, u64 n, double box_size, u64 max_leaf_n,
                         u64 ncrit, int *status)
{
    return orc_tree_create_ex(ndim, 64, fp, mac, src, n, box_size, max_leaf_n, ncrit, status);
}

// code_bits: 64 or 32 = width of the Morton codes (the UInt template parameter of the reference's tree).
void *orc_tree_create_ex(int ndim, int code_bits, int fp, int mac, const void *const *src, u64 n, double box_size,
                         u64 max_leaf_n, u64 ncrit, int *status)
{
    auto h = std::make_unique<handle_t>();
    h->fp = fp;
    h->ndim = ndim;
    const int rc = guard([&] {
        if (ndim != 2 && ndim != 3) {
            throw std::invalid_argument("ndim must be 2 or 3");
        }
        if (code_bits != 32 && code_bits != 64) {
            throw std::invalid_argument("code_bits must be 32 or 64");
        }
        auto make = [&](auto &ptr) {
            using T = typename std::remove_reference_t<decltype(ptr)>::element_type;
            using F = typename T::fp_type;
            ptr = std::make_unique<T>();
            ptr->mac = mac;
            ptr->cbits = cbits_for(unsigned(code_bits), unsigned(ndim));
            const F *s[4] = {};
            for (int j = 0; j < ndim + 1; ++j) {
                s[j] = static_cast<const F *>(src[j]);
            }
            ptr->construct(s, n, F(box_size), max_leaf_n, ncrit);
        };
        if (ndim == 3) {
            if (fp == 0) {
                make(h->tf3);
            } else {
                make(h->td3);
            }
        } else {
            if (fp == 0) {
                make(h->tf2);
            } else {
                make(h->td2);
            }
        }
    });
    if (status) {
        *status = rc;
    }
    return rc ? nullptr : h.release();
}

void *orc_tree_create(int fp, int mac, const void *x, const void *y, const void *z, const void *m, u64 n,
                      double box_size, u64 max_leaf_n, u64 ncrit, int *status)
{
    const void *src[4] = {x, y, z, m};
    return orc_tree_create_nd(3, fp, mac, src, n, box_size, max_leaf_n, ncrit, status);
}

void orc_tree_destroy(void *hp)
{
    delete static_cast<handle_t *>(hp);
}

// info[0..3] = nparts, n_nodes, n_crit, ndim; box = box size.
void orc_tree_info(void *hp, u64 *info, double *box)
{
    auto *h = static_cast<handle_t *>(hp);
    visit(h, [&](auto &t) {
        info[0] = t.nparts();
        info[1] = t.nodes.size();
        info[2] = t.crit.size();
        info[3] = t.NDim;
        *box = t.box_size;
    });
}

// Copy out the Morton-ordered particle SoA (p_its_u, tree.hpp:3638-3641), codes and permutations.
// Any pointer may be null; z is ignored for 2D trees.
void orc_tree_get_parts(void *hp, void *x, void *y, void *z, void *m, u64 *codes, u64 *perm, u64 *last_perm,
                        u64 *inv_perm)
{
    auto *h = static_cast<handle_t *>(hp);
    visit(h, [&](auto &t) {
        using T = std::remove_reference_t<decltype(t)>;
        using F = typename T::fp_type;
        void *coords[3] = {x, y, z};
        for (unsigned j = 0; j < T::NDim; ++j) {
            if (coords[j]) {
                std::memcpy(coords[j], t.parts[j].data(), t.nparts() * sizeof(F));
            }
        }
        if (m) {
            std::memcpy(m, t.parts[T::NDim].data(), t.nparts() * sizeof(F));
        }
        if (codes) {
            std::memcpy(codes, t.codes.data(), t.nparts() * 8);
        }
        if (perm) {
            std::memcpy(perm, t.perm.data(), t.nparts() * 8);
        }
        if (last_perm) {
            std::memcpy(last_perm, t.last_perm.data(), t.nparts() * 8);
        }
        if (inv_perm) {
            std::memcpy(inv_perm, t.inv_perm.data(), t.nparts() * 8);
        }
    });
}

// Copy out the node array (nodes(), tree.hpp:3670-3673) as SoA:
// topo[5*i + {0..4}] = begin, end, n_children, code, level; props[(ndim+1)*i + {0..ndim}] = COM, mass;
// dims[2*i + {0,1}] = {dim2, 0} (bh) or {dim, delta} (bh_geom).
void orc_tree_get_nodes(void *hp, u64 *topo, void *props, void *dims)
{
    auto *h = static_cast<handle_t *>(hp);
    visit(h, [&](auto &t) {
        using T = std::remove_reference_t<decltype(t)>;
        using F = typename T::fp_type;
        constexpr unsigned np = T::NDim + 1;
        auto *pr = static_cast<F *>(props);
        auto *dm = static_cast<F *>(dims);
        for (std::size_t i = 0; i < t.nodes.size(); ++i) {
            const auto &n = t.nodes[i];
            if (topo) {
                topo[5 * i] = n.begin;
                topo[5 * i + 1] = n.end;
                topo[5 * i + 2] = n.n_children;
                topo[5 * i + 3] = n.code;
                topo[5 * i + 4] = n.level;
            }
            if (pr) {
                for (unsigned j = 0; j < np; ++j) {
                    pr[np * i + j] = n.props[j];
                }
            }
            if (dm) {
                dm[2 * i] = n.dim;
                dm[2 * i + 1] = n.delta;
            }
        }
    });
}

// crit[3*i + {0,1,2}] = code, begin, end.
void orc_tree_get_crit(void *hp, u64 *crit)
{
    auto *h = static_cast<handle_t *>(hp);
    visit(h, [&](auto &t) {
        const auto &c = t.crit;
        for (std::size_t i = 0; i < c.size(); ++i) {
            crit[3 * i] = c[i].code;
            crit[3 * i + 1] = c[i].begin;
            crit[3 * i + 2] = c[i].end;
        }
    });
}

// q: 0 accs (ndim outputs), 1 pots (1), 2 accs+pots (ndim + 1). ordered: 0 -> *_u, 1 -> *_o.
// Only critical nodes [c_begin, c_end) are processed (c_end is clamped); outputs of the other
// particles are left untouched. stats (may be null, else 9 entries): visits, com, leaves, pp, self_pairs,
// w_visits, w_com, w_pp, w_self (see stats_t). out always has room for 4 pointers.
int orc_acc_pot(void *hp, int q, void *const *out, int ordered, double theta, double G, double eps,
                unsigned nthreads, u64 c_begin, u64 c_end, u64 *stats)
{
    auto *h = static_cast<handle_t *>(hp);
    return guard([&] {
        visit(h, [&](auto &t) { acc_pot_q(t, q, out, ordered, theta, G, eps, nthreads, c_begin, c_end, stats); });
    });
}

// exact_{acc,pot,acc_pot}_{u,o} (tree.hpp:3572-3616). out has ndim/1/ndim+1 entries.
int orc_exact(void *hp, int q, void *out, int ordered, u64 idx, double G, double eps)
{
    auto *h = static_cast<handle_t *>(hp);
    return guard([&] {
        visit(h, [&](auto &t) {
            using F = typename std::remove_reference_t<decltype(t)>::fp_type;
            auto *o = static_cast<F *>(out);
            switch (q) {
                case 0:
                    t.template exact<0>(o, ordered, idx, F(G), F(eps));
                    break;
                case 1:
                    t.template exact<1>(o, ordered, idx, F(G), F(eps));
                    break;
                case 2:
                    t.template exact<2>(o, ordered, idx, F(G), F(eps));
                    break;
                default:
                    throw std::invalid_argument("q must be 0, 1 or 2");
            }
        });
    });
}

} // extern "C"
