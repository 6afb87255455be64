// rakau_amd::tree -- C++17 front door of the MI355X-native Barnes-Hut traversal engine.
//
// Drop-in for the acc/pot surface of rakau::tree<> (include/rakau/tree.hpp of the reference; line
// numbers below refer to that file): same class template, same kwargs constructor, same
// accs_u/pots_u/accs_pots_u (+ _o) overloads, same exact_* helpers, accessors and update_* calls,
// same exception types and messages. What differs is what runs underneath:
//
//   * the tree is built on the host by this header's own builder (Morton encode -> parallel merge
//     sort -> task-parallel depth-first node construction), producing the reference's data contract
//     (tree_node_t / tree_cnode_t, include/rakau/detail/tree_fwd.hpp:77-125);
//   * every acc/pot call is served by hand-written HIP kernels through the C ABI of
//     include/rakau_amd.h; a call without a gfx950 device throws, it never falls back to the CPU;
//   * `split` keeps the reference's spelling and meaning {host, dev0, dev1, ...} (tree.hpp:3047-3240):
//     the critical nodes below the first cut are computed by this header's own CPU engine
//     (cpu_engine.hpp) on the calling thread WHILE one host thread per device runs the device
//     shares. Deliberate differences: an empty split (the default) means "everything on device 0"
//     (the reference: CPU; -DRAKAU_AMD_EMPTY_SPLIT_IS_CPU restores that), and every cut is snapped to
//     a critical-node boundary (the reference snaps the first one).
//
// Define RAKAU_AMD_DROP_IN before including to get `namespace rakau = rakau_amd;`.
#ifndef RAKAU_AMD_TREE_HPP
#define RAKAU_AMD_TREE_HPP

#include <algorithm>
#include <array>
#include <cassert>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <future>
#include <initializer_list>
#include <iterator>
#include <limits>
#include <memory>
#include <mutex>
#include <new>
#include <numeric>
#include <ostream>
#include <stdexcept>
#include <string>
#include <thread>
#include <tuple>
#include <type_traits>
#include <utility>
#include <vector>

#include "../rakau_amd.h"
#include "cpu_engine.hpp"
#include "kwargs.hpp"

namespace rakau_amd
{

// Multipole acceptance criteria (tree_fwd.hpp:46).
enum class mac { bh, bh_geom };

// Allocator for output vectors in pinned host memory (rk_host_alloc): with
//     std::array<std::vector<float, rakau_amd::pinned_allocator<float>>, 3> accs;  t.accs_u(accs, theta);
// -- the `std::vector<F, Allocator>` overloads of the reference's API, tree.hpp:3406-3497 -- the kernels write the
// results into the vectors themselves instead of a staging buffer that host threads then copy (rakau_amd.h).
template <typename T>
struct pinned_allocator {
    using value_type = T;
    pinned_allocator() = default;
    template <typename U>
    pinned_allocator(const pinned_allocator<U> &) noexcept
    {
    }
    T *allocate(std::size_t n)
    {
        void *p = nullptr;
        if (n > static_cast<std::size_t>(std::numeric_limits<std::int64_t>::max()) / sizeof(T)
            || rk_host_alloc(&p, static_cast<std::int64_t>(n * sizeof(T))) != RK_OK || (n && !p)) {
            throw std::bad_alloc();
        }
        return static_cast<T *>(p);
    }
    void deallocate(T *p, std::size_t) noexcept
    {
        (void)rk_host_free(p);
    }
    template <typename U>
    bool operator==(const pinned_allocator<U> &) const noexcept
    {
        return true;
    }
    template <typename U>
    bool operator!=(const pinned_allocator<U> &) const noexcept
    {
        return false;
    }
};

inline namespace detail
{

template <typename F>
using tree_size_t = std::size_t;

// Node records: the data contract of the boundary (tree_fwd.hpp:77-116). n_children counts all
// descendants, so that the next sibling of node i is i + n_children + 1.
template <std::size_t NDim, typename F, typename UInt>
struct base_tree_node_t {
    tree_size_t<F> begin, end, n_children;
    UInt code, level;
    friend bool operator==(const base_tree_node_t &a, const base_tree_node_t &b)
    {
        return a.begin == b.begin && a.end == b.end && a.n_children == b.n_children && a.code == b.code
               && a.level == b.level;
    }
};

template <std::size_t NDim, typename F, typename UInt, mac MAC>
struct tree_node_t;

template <std::size_t NDim, typename F, typename UInt>
struct tree_node_t<NDim, F, UInt, mac::bh> : base_tree_node_t<NDim, F, UInt> {
    F props[NDim + 1u], dim2;
    friend bool operator==(const tree_node_t &a, const tree_node_t &b)
    {
        using base = base_tree_node_t<NDim, F, UInt>;
        return std::equal(std::begin(a.props), std::end(a.props), std::begin(b.props)) && a.dim2 == b.dim2
               && static_cast<const base &>(a) == static_cast<const base &>(b);
    }
};

template <std::size_t NDim, typename F, typename UInt>
struct tree_node_t<NDim, F, UInt, mac::bh_geom> : base_tree_node_t<NDim, F, UInt> {
    F props[NDim + 1u], dim, delta;
    friend bool operator==(const tree_node_t &a, const tree_node_t &b)
    {
        using base = base_tree_node_t<NDim, F, UInt>;
        return std::equal(std::begin(a.props), std::end(a.props), std::begin(b.props)) && a.dim == b.dim
               && a.delta == b.delta && static_cast<const base &>(a) == static_cast<const base &>(b);
    }
};

// Critical node (tree_fwd.hpp:119-125).
template <typename F, typename UInt>
struct tree_cnode_t {
    UInt code;
    tree_size_t<F> begin, end;
};

template <unsigned Q, std::size_t NDim>
inline constexpr std::size_t tree_nvecs_res = (Q == 0u ? NDim : (Q == 1u ? 1u : NDim + 1u));

// Bits per coordinate in the Morton code (tree_fwd.hpp:141-150).
template <typename UInt, std::size_t NDim>
inline constexpr unsigned cbits_v
    = static_cast<unsigned>(std::numeric_limits<UInt>::digits / NDim - !(std::numeric_limits<UInt>::digits % NDim));

// Level of a nodal code (tree_fwd.hpp:212-228).
template <std::size_t NDim>
inline unsigned tree_level(std::uint64_t n)
{
    return (63u - static_cast<unsigned>(__builtin_clzll(n))) / static_cast<unsigned>(NDim);
}

// Default tree parameters (tree.hpp:584-595; 128 is the non-AVX-512 value).
inline constexpr unsigned default_max_leaf_n = 16;
inline constexpr unsigned default_ncrit = 128;

template <typename T>
using uncvref_t = std::remove_cv_t<std::remove_reference_t<T>>;

template <typename T, typename = void>
struct is_range : std::false_type {
};
template <typename T>
struct is_range<T, std::void_t<decltype(std::begin(std::declval<T>())), decltype(std::end(std::declval<T>()))>>
    : std::true_type {
};

// Range-checked integral conversion (stands where the reference uses boost::numeric_cast).
template <typename To, typename From>
inline To checked_cast(From x)
{
    if constexpr (std::is_floating_point_v<To>) {
        return static_cast<To>(x);
    } else if constexpr (std::is_floating_point_v<From>) {
        if (!(x >= static_cast<From>(std::numeric_limits<To>::min()))
            || !(x < static_cast<From>(std::numeric_limits<To>::max()))) {
            throw std::overflow_error("floating-point value out of the range of the target integral type");
        }
        return static_cast<To>(x);
    } else {
        if constexpr (std::is_signed_v<From>) {
            if (x < 0 && !std::is_signed_v<To>) {
                throw std::overflow_error("negative value converted to an unsigned type");
            }
        }
        using C = std::common_type_t<std::make_unsigned_t<To>, std::make_unsigned_t<From>>;
        if (x > 0 && static_cast<C>(x) > static_cast<C>(std::numeric_limits<To>::max())) {
            throw std::overflow_error("integral value out of the range of the target type");
        }
        return static_cast<To>(x);
    }
}

// Random-access iterator visiting base[index[i]] (stands where the reference uses
// boost::permutation_iterator: p_its_o(), c_it_o(), ordered outputs).
template <typename BaseIt, typename IdxIt>
class perm_iterator
{
    BaseIt m_base;
    IdxIt m_idx;

public:
    using iterator_category = std::random_access_iterator_tag;
    using value_type = typename std::iterator_traits<BaseIt>::value_type;
    using reference = typename std::iterator_traits<BaseIt>::reference;
    using pointer = typename std::iterator_traits<BaseIt>::pointer;
    using difference_type = typename std::iterator_traits<IdxIt>::difference_type;

    perm_iterator() = default;
    perm_iterator(BaseIt b, IdxIt i) : m_base(b), m_idx(i) {}
    reference operator*() const
    {
        return *(m_base + static_cast<typename std::iterator_traits<BaseIt>::difference_type>(*m_idx));
    }
    reference operator[](difference_type n) const
    {
        return *(*this + n);
    }
    perm_iterator &operator++()
    {
        ++m_idx;
        return *this;
    }
    perm_iterator operator++(int)
    {
        auto t = *this;
        ++m_idx;
        return t;
    }
    perm_iterator &operator--()
    {
        --m_idx;
        return *this;
    }
    perm_iterator operator--(int)
    {
        auto t = *this;
        --m_idx;
        return t;
    }
    perm_iterator &operator+=(difference_type n)
    {
        m_idx += n;
        return *this;
    }
    perm_iterator &operator-=(difference_type n)
    {
        m_idx -= n;
        return *this;
    }
    friend perm_iterator operator+(perm_iterator a, difference_type n)
    {
        a += n;
        return a;
    }
    friend perm_iterator operator+(difference_type n, perm_iterator a)
    {
        a += n;
        return a;
    }
    friend perm_iterator operator-(perm_iterator a, difference_type n)
    {
        a -= n;
        return a;
    }
    friend difference_type operator-(const perm_iterator &a, const perm_iterator &b)
    {
        return a.m_idx - b.m_idx;
    }
    friend bool operator==(const perm_iterator &a, const perm_iterator &b)
    {
        return a.m_idx == b.m_idx;
    }
    friend bool operator!=(const perm_iterator &a, const perm_iterator &b)
    {
        return a.m_idx != b.m_idx;
    }
    friend bool operator<(const perm_iterator &a, const perm_iterator &b)
    {
        return a.m_idx < b.m_idx;
    }
    friend bool operator>(const perm_iterator &a, const perm_iterator &b)
    {
        return a.m_idx > b.m_idx;
    }
    friend bool operator<=(const perm_iterator &a, const perm_iterator &b)
    {
        return a.m_idx <= b.m_idx;
    }
    friend bool operator>=(const perm_iterator &a, const perm_iterator &b)
    {
        return a.m_idx >= b.m_idx;
    }
};

// Number of host threads used by the builder (hardware threads narrowed by affinity and the cgroup CPU quota).
inline unsigned host_threads()
{
    return std::clamp(usable_hw_threads(), 1u, 64u);
}

// Run f(begin, end) over [0, n) on several threads (contiguous blocks).
template <typename Fn>
inline void parallel_blocks(std::size_t n, std::size_t min_block, Fn &&f)
{
    const std::size_t nt = std::min<std::size_t>(host_threads(), min_block ? (n + min_block - 1) / min_block : 1);
    if (nt <= 1 || n == 0) {
        f(std::size_t(0), n);
        return;
    }
    std::vector<std::thread> th;
    std::exception_ptr ep;
    std::mutex mx;
    const std::size_t blk = (n + nt - 1) / nt;
    for (std::size_t t = 0; t < nt; ++t) {
        const std::size_t b = t * blk, e = std::min(n, b + blk);
        if (b >= e) {
            break;
        }
        th.emplace_back([&, b, e] {
            try {
                f(b, e);
            } catch (...) {
                std::lock_guard<std::mutex> lk(mx);
                if (!ep) {
                    ep = std::current_exception();
                }
            }
        });
    }
    for (auto &t : th) {
        t.join();
    }
    if (ep) {
        std::rethrow_exception(ep);
    }
}

// Per-thread staging for results that cannot be written where the caller wants them (original-order outputs, generic output
// iterators): k arrays of n values, kept between calls and only ever grown. Pinned (rk_host_alloc: the kernels write into them
// directly) when that succeeds, plain memory otherwise (no device: the CPU engine writes them).
// Staging a thread keeps between calls, in all (larger sets are released when the call that needed them is over).
#ifndef RAKAU_AMD_STAGE_KEEP_MB
#define RAKAU_AMD_STAGE_KEEP_MB 16
#endif
constexpr std::size_t stage_keep_bytes = std::size_t(RAKAU_AMD_STAGE_KEEP_MB) << 20;
template <typename F>
struct stage_buffers {
    F *p[4] = {nullptr, nullptr, nullptr, nullptr};
    std::size_t cap = 0;
    bool pinned = false;
    ~stage_buffers()
    {
        release();
    }
    void release() noexcept
    {
        for (auto &q : p) {
            if (q) {
                if (pinned) {
                    (void)rk_host_free(q);
                } else {
                    ::operator delete(q);
                }
                q = nullptr;
            }
        }
        cap = 0;
    }
    static void get(std::size_t n, std::size_t k, F *out[4])
    {
        stage_buffers &sb = instance();
        if (sb.cap < n) {
            sb.release();
            sb.cap = n;
            sb.pinned = rk_has_accelerator() != 0;
        }
        const std::size_t bytes = std::max<std::size_t>(sb.cap, 1) * sizeof(F);
        for (std::size_t j = 0; j < k; ++j) {
            if (sb.p[j]) {
                continue;
            }
            void *q = nullptr;
            if (sb.pinned && (rk_host_alloc(&q, static_cast<std::int64_t>(bytes)) != RK_OK || !q)) {
                // Pinned memory refused: everything in plain memory from here on (mixing the two would only confuse release()).
                const std::size_t keep = sb.cap;
                sb.release();
                sb.cap = keep;
                sb.pinned = false;
                j = static_cast<std::size_t>(-1);
                continue;
            }
            if (!sb.pinned) {
                q = ::operator new(bytes);
            }
            sb.p[j] = static_cast<F *>(q);
        }
        for (std::size_t j = 0; j < 4; ++j) {
            out[j] = sb.p[j];
        }
    }
    // Called when a staged call is over: buffers beyond `keep_bytes` in all are given back at once, so that a pool of threads
    // calling the staged overloads on large trees does not keep 4 x nparts x sizeof(F) of PINNED memory per thread for the life of
    // the process; small ones stay for the next call. trim(0) releases everything the calling thread holds.
    static void trim(std::size_t keep_bytes)
    {
        stage_buffers &sb = instance();
        std::size_t held = 0;
        for (const auto *q : sb.p) {
            held += q ? std::max<std::size_t>(sb.cap, 1) * sizeof(F) : 0;
        }
        if (held > keep_bytes) {
            sb.release();
        }
    }

private:
    static stage_buffers &instance()
    {
        thread_local stage_buffers sb;
        return sb;
    }
};

// Bit interleaving for 3-D Morton codes: x -> bit 0, y -> bit 1, z -> bit 2 (the order produced by the
// reference's encoder, tree.hpp:222-242 / libmorton/morton3D.h:38-50).
inline std::uint64_t morton_spread3(std::uint64_t v)
{
    v &= 0x1fffffULL;
    v = (v | (v << 32)) & 0x1f00000000ffffULL;
    v = (v | (v << 16)) & 0x1f0000ff0000ffULL;
    v = (v | (v << 8)) & 0x100f00f00f00f00fULL;
    v = (v | (v << 4)) & 0x10c30c30c30c30c3ULL;
    v = (v | (v << 2)) & 0x1249249249249249ULL;
    return v;
}
inline std::uint64_t morton_compact3(std::uint64_t v)
{
    v &= 0x1249249249249249ULL;
    v = (v ^ (v >> 2)) & 0x10c30c30c30c30c3ULL;
    v = (v ^ (v >> 4)) & 0x100f00f00f00f00fULL;
    v = (v ^ (v >> 8)) & 0x1f0000ff0000ffULL;
    v = (v ^ (v >> 16)) & 0x1f00000000ffffULL;
    v = (v ^ (v >> 32)) & 0x1fffffULL;
    return v;
}

// 2-D flavour: x -> even bits, y -> odd bits (libmorton/morton2D.h as used at tree.hpp:207-220 of the reference).
inline std::uint64_t morton_spread2(std::uint64_t v)
{
    v &= 0xffffffffULL;
    v = (v | (v << 16)) & 0x0000ffff0000ffffULL;
    v = (v | (v << 8)) & 0x00ff00ff00ff00ffULL;
    v = (v | (v << 4)) & 0x0f0f0f0f0f0f0f0fULL;
    v = (v | (v << 2)) & 0x3333333333333333ULL;
    v = (v | (v << 1)) & 0x5555555555555555ULL;
    return v;
}
inline std::uint64_t morton_compact2(std::uint64_t v)
{
    v &= 0x5555555555555555ULL;
    v = (v ^ (v >> 1)) & 0x3333333333333333ULL;
    v = (v ^ (v >> 2)) & 0x0f0f0f0f0f0f0f0fULL;
    v = (v ^ (v >> 4)) & 0x00ff00ff00ff00ffULL;
    v = (v ^ (v >> 8)) & 0x0000ffff0000ffffULL;
    v = (v ^ (v >> 16)) & 0xffffffffULL;
    return v;
}
template <std::size_t NDim>
inline std::uint64_t morton_encode(const std::uint64_t *d)
{
    if constexpr (NDim == 3u) {
        return morton_spread3(d[0]) | (morton_spread3(d[1]) << 1) | (morton_spread3(d[2]) << 2);
    } else {
        return morton_spread2(d[0]) | (morton_spread2(d[1]) << 1);
    }
}
template <std::size_t NDim>
inline std::uint64_t morton_coord(std::uint64_t code, std::size_t j)
{
    if constexpr (NDim == 3u) {
        return morton_compact3(code >> j);
    } else {
        return morton_compact2(code >> j);
    }
}

// Owner of one rk_state (device-resident copy of the tree on one GPU).
struct device_state {
    rk_state *h = nullptr;
    bool perm_set = false; // the state has been given the tree's permutation (original-order outputs written by the kernels)
    device_state() = default;
    device_state(const device_state &) = delete;
    device_state &operator=(const device_state &) = delete;
    device_state(device_state &&o) noexcept : h(o.h), perm_set(o.perm_set)
    {
        o.h = nullptr;
    }
    device_state &operator=(device_state &&o) noexcept
    {
        if (this != &o) {
            reset();
            h = o.h;
            perm_set = o.perm_set;
            o.h = nullptr;
        }
        return *this;
    }
    ~device_state()
    {
        reset();
    }
    void reset()
    {
        if (h) {
            rk_state_destroy(h);
            h = nullptr;
        }
        perm_set = false;
    }
};

// Map a C ABI status onto the exception types the reference throws ("Errors").
inline void throw_status(int rc)
{
    if (rc == RK_OK) {
        return;
    }
    const std::string msg = rk_last_error();
    switch (rc) {
        case RK_EINVAL:
            throw std::invalid_argument(msg);
        case RK_EDOMAIN:
            throw std::domain_error(msg);
        case RK_EOVERFLOW:
            throw std::overflow_error(msg);
        case RK_ENOMEM:
            throw std::bad_alloc();
        default:
            throw std::runtime_error(msg);
    }
}

} // namespace detail

// Morton encoding / decoding of NDim discretised coordinates (the functors of tree.hpp:218-374 of the reference):
// the encoder reads NDim values of cbits_v<UInt, NDim> bits each from an iterator, the decoder writes them back.
template <std::size_t NDim, typename UInt>
struct morton_encoder {
    static_assert(NDim == 2u || NDim == 3u);
    template <typename It>
    UInt operator()(It it) const
    {
        std::uint64_t d[NDim];
        for (std::size_t j = 0; j < NDim; ++j) {
            d[j] = static_cast<std::uint64_t>(*(it + static_cast<std::ptrdiff_t>(j)));
        }
        return static_cast<UInt>(detail::morton_encode<NDim>(d));
    }
};
template <std::size_t NDim, typename UInt>
struct morton_decoder {
    static_assert(NDim == 2u || NDim == 3u);
    template <typename It>
    void operator()(It it, UInt code) const
    {
        for (std::size_t j = 0; j < NDim; ++j) {
            *(it + static_cast<std::ptrdiff_t>(j)) = static_cast<UInt>(detail::morton_coord<NDim>(code, j));
        }
    }
};

// Size of a node at `node_level`, and geometric centre of the node with nodal code `node_code`, in a domain of size
// `box_size` (tree.hpp:444-482 of the reference; same arithmetic as tree::node_centre()).
template <typename UInt, typename F>
inline F get_node_dim(UInt node_level, F box_size)
{
    return box_size / static_cast<F>(UInt(1) << node_level);
}
template <typename F, std::size_t NDim, typename UInt>
inline void get_node_centre(F (&out)[NDim], UInt node_code, F box_size)
{
    constexpr unsigned cbits = cbits_v<UInt, NDim>;
    const unsigned level = tree_level<NDim>(node_code);
    const UInt first_cell = static_cast<UInt>((node_code - (UInt(1) << (level * NDim))) << ((cbits - level) * NDim));
    const F half_dim = get_node_dim(static_cast<UInt>(level), box_size) * (F(1) / F(2));
    const F cell = box_size * (F(1) / static_cast<F>(UInt(1) << cbits));
    for (std::size_t j = 0; j < NDim; ++j) {
        out[j] = std::fma(static_cast<F>(detail::morton_coord<NDim>(first_cell, j)), cell,
                          half_dim - box_size * (F(1) / F(2)));
    }
}

template <typename F>
using f_vector = std::vector<F>;

template <std::size_t NDim, typename F, typename UInt, mac MAC>
class tree
{
    static_assert(NDim == 2u || NDim == 3u, "rakau_amd::tree provides quadtrees (NDim = 2) and octrees (NDim = 3).");
    static constexpr unsigned n_children_max = 1u << NDim;
    static_assert(std::is_same_v<F, float> || std::is_same_v<F, double>,
                  "The type F must be float or double (the precisions the device kernels are built for).");
    static_assert(std::is_integral_v<UInt> && std::is_unsigned_v<UInt>
                      && (std::numeric_limits<UInt>::digits == 64 || std::numeric_limits<UInt>::digits == 32),
                  "rakau_amd::tree provides 64-bit and 32-bit Morton codes (the reference's UInt instantiations).");
    static constexpr bool wide_codes = std::numeric_limits<UInt>::digits == 64;
    static constexpr unsigned cbits = cbits_v<UInt, NDim>;

public:
    using size_type = tree_size_t<F>;

private:
    using node_type = tree_node_t<NDim, F, UInt, MAC>;
    using tree_type = std::vector<node_type>;
    using cnode_type = tree_cnode_t<F, UInt>;
    using cnode_list_type = std::vector<cnode_type>;

    template <unsigned Q>
    static constexpr std::size_t nvecs_res = tree_nvecs_res<Q, NDim>;

    // ------------------------------------------------------------------------------------------
    // Construction.
    // ------------------------------------------------------------------------------------------

    // Discretise one coordinate (behaviour of tree.hpp:381-429, Clamp == false).
    static UInt disc_single_coord(F x, F inv_box_size)
    {
        constexpr UInt factor = UInt(1) << cbits;
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
        const auto retval = static_cast<UInt>(tmp);
        if (retval >= factor) {
            throw std::invalid_argument("The discretisation of the input coordinate " + std::to_string(x)
                                        + " in a box of size " + std::to_string(F(1) / inv_box_size)
                                        + " produced the integral value " + std::to_string(retval)
                                        + ", which is outside the allowed bounds");
        }
        return retval;
    }

    static F node_dim_of(UInt level, F box)
    {
        return box / static_cast<F>(UInt(1) << level);
    }

    // Geometric centre of a node (behaviour of tree.hpp:452-482).
    void node_centre(F (&out)[NDim], UInt code) const
    {
        const unsigned level = tree_level<NDim>(code);
        const UInt first_cell = static_cast<UInt>((code - (UInt(1) << (level * NDim))) << ((cbits - level) * NDim));
        const F half_dim = node_dim_of(level, m_box_size) * (F(1) / F(2));
        const F cell = m_box_size * (F(1) / static_cast<F>(UInt(1) << cbits));
        for (std::size_t j = 0; j < NDim; ++j) {
            out[j] = std::fma(static_cast<F>(detail::morton_coord<NDim>(first_cell, j)), cell,
                              half_dim - m_box_size * (F(1) / F(2)));
        }
    }

    // Mass, centre of mass and MAC size of a node (behaviour of tree.hpp:1116-1237, scalar summation).
    void fill_node_properties(node_type &node) const
    {
        F tot_mass(0), com[NDim] = {};
        const F *cs[NDim];
        for (std::size_t j = 0; j < NDim; ++j) {
            cs[j] = m_parts[j].data();
        }
        const F *ms = m_parts[NDim].data();
        for (size_type i = node.begin; i < node.end; ++i) {
            const F mass = ms[i];
            tot_mass += mass;
            for (std::size_t j = 0; j < NDim; ++j) {
                com[j] = std::fma(mass, cs[j][i], com[j]);
            }
        }
        [[maybe_unused]] F centre[NDim] = {};
        if constexpr (MAC == mac::bh_geom) {
            node_centre(centre, node.code);
        }
        if (tot_mass == F(0)) {
            if constexpr (MAC == mac::bh) {
                node_centre(com, node.code);
            } else {
                std::copy(std::begin(centre), std::end(centre), std::begin(com));
            }
        } else {
            const F inv = F(1) / tot_mass;
            for (auto &c : com) {
                c *= inv;
            }
        }
        for (std::size_t j = 0; j < NDim; ++j) {
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
        node.props[NDim] = tot_mass;
        const F dim = node_dim_of(node.level, m_box_size);
        if constexpr (MAC == mac::bh) {
            node.dim2 = dim * dim;
            if (!std::isfinite(node.dim2)) {
                throw std::invalid_argument(
                    "The computation of the square of the dimension of a node produced the non-finite value "
                    + std::to_string(node.dim2));
            }
        } else {
            node.dim = dim;
            if (!std::isfinite(node.dim)) {
                throw std::invalid_argument("The computation of the dimension of a node produced the non-finite value "
                                            + std::to_string(node.dim));
            }
            F delta2 = (com[0] - centre[0]) * (com[0] - centre[0]);
            for (std::size_t j = 1; j < NDim; ++j) {
                delta2 = std::fma(com[j] - centre[j], com[j] - centre[j], delta2);
            }
            node.delta = std::sqrt(delta2);
            if (!std::isfinite(node.delta)) {
                throw std::invalid_argument("The computation of the distance between the centre of mass "
                                            "and the geometric centre of a node produced the non-finite value "
                                            + std::to_string(node.delta));
            }
        }
    }

    // A finished piece of the tree: nodes in depth-first order plus the critical nodes found in it.
    struct subtree {
        tree_type nodes;
        cnode_list_type crit;
    };

    // Particles per task below which a subtree is built by one thread. The reference switches from
    // parallel to serial construction at 40000 particles (tree.hpp:903); here the threshold also grows
    // with N so that the number of tasks stays near 8 per host thread.
    size_type task_threshold() const
    {
        return std::max<size_type>(40000, m_codes.size() / (8u * host_threads()));
    }

    // Split [begin, end) of the sorted codes into the (up to 2^NDim) children of a node at `level`.
    // Returns the child boundaries: child i owns [b[i], b[i + 1]).
    std::array<size_type, n_children_max + 1u> child_bounds(unsigned level, size_type begin, size_type end) const
    {
        const unsigned shift = (cbits - level - 1u) * NDim;
        std::array<size_type, n_children_max + 1u> b;
        b[0] = begin;
        const UInt *codes = m_codes.data();
        for (unsigned i = 1; i < n_children_max; ++i) {
            // First particle whose NDim-bit digit at this level is >= i.
            b[i] = static_cast<size_type>(
                std::lower_bound(codes + b[i - 1], codes + end, i,
                                 [shift](UInt c, unsigned digit) { return ((c >> shift) & (n_children_max - 1u)) < digit; })
                - codes);
        }
        b[n_children_max] = end;
        return b;
    }

    // Depth-first construction of everything below the node (parent_code, parent_level) that owns
    // particles [begin, end). Appends to out; returns the number of nodes appended. Semantics of
    // tree.hpp:723-833.
    size_type build_below(subtree &out, unsigned parent_level, UInt parent_code, size_type begin, size_type end,
                          bool crit_ancestor) const
    {
        if (parent_level >= cbits) {
            return 0;
        }
        const auto b = child_bounds(parent_level, begin, end);
        size_type appended = 0;
        for (unsigned i = 0; i < n_children_max; ++i) {
            const size_type npart = b[i + 1] - b[i];
            if (!npart) {
                continue;
            }
            node_type n{};
            n.begin = b[i];
            n.end = b[i + 1];
            n.n_children = 0;
            n.code = static_cast<UInt>((parent_code << NDim) + i);
            n.level = parent_level + 1u;
            fill_node_properties(n);
            const size_type idx = out.nodes.size();
            out.nodes.push_back(n);
            // Critical node rule (tree.hpp:801-803).
            const bool critical
                = !crit_ancestor && (npart <= m_ncrit || npart <= m_max_leaf_n || parent_level + 1u == cbits);
            if (critical) {
                out.crit.push_back(cnode_type{n.code, n.begin, n.end});
            }
            if (npart > m_max_leaf_n) {
                const size_type below
                    = build_below(out, parent_level + 1u, n.code, n.begin, n.end, critical || crit_ancestor);
                out.nodes[idx].n_children = below;
            }
            appended += out.nodes[idx].n_children + 1u;
        }
        return appended;
    }

    // Task-parallel variant for nodes holding many particles: children with >= task_threshold() particles
    // become asynchronous tasks, the pieces are concatenated in child order (= nodal-code order, which is
    // what the reference obtains by sorting its partial trees, tree.hpp:984-1018).
    subtree build_below_par(unsigned parent_level, UInt parent_code, size_type begin, size_type end,
                            bool crit_ancestor) const
    {
        subtree out;
        if (parent_level >= cbits) {
            return out;
        }
        const auto b = child_bounds(parent_level, begin, end);
        struct piece {
            node_type head;
            bool critical = false;
            std::future<subtree> fut;
            subtree ready;
            bool has_future = false;
        };
        std::vector<piece> pieces;
        pieces.reserve(n_children_max);
        for (unsigned i = 0; i < n_children_max; ++i) {
            const size_type npart = b[i + 1] - b[i];
            if (!npart) {
                continue;
            }
            piece p;
            p.head = node_type{};
            p.head.begin = b[i];
            p.head.end = b[i + 1];
            p.head.code = static_cast<UInt>((parent_code << NDim) + i);
            p.head.level = parent_level + 1u;
            p.critical = !crit_ancestor && (npart <= m_ncrit || npart <= m_max_leaf_n || parent_level + 1u == cbits);
            const bool anc = p.critical || crit_ancestor;
            const auto code = p.head.code;
            const auto pb = p.head.begin, pe = p.head.end;
            const unsigned lvl = parent_level + 1u;
            if (npart > m_max_leaf_n) {
                if (npart >= task_threshold()) {
                    p.has_future = true;
                    p.fut = std::async(std::launch::async,
                                       [this, lvl, code, pb, pe, anc] { return build_below_par(lvl, code, pb, pe, anc); });
                } else {
                    build_below(p.ready, lvl, code, pb, pe, anc);
                }
            }
            pieces.push_back(std::move(p));
        }
        // Node properties of the heads are computed while the tasks run.
        for (auto &p : pieces) {
            fill_node_properties(p.head);
        }
        for (auto &p : pieces) {
            subtree sub = p.has_future ? p.fut.get() : std::move(p.ready);
            p.head.n_children = sub.nodes.size();
            out.nodes.push_back(p.head);
            if (p.critical) {
                out.crit.push_back(cnode_type{p.head.code, p.head.begin, p.head.end});
            }
            out.nodes.insert(out.nodes.end(), sub.nodes.begin(), sub.nodes.end());
            out.crit.insert(out.crit.end(), sub.crit.begin(), sub.crit.end());
        }
        return out;
    }

    // Build m_tree / m_crit_nodes from the sorted codes (semantics of tree.hpp:932-1111).
    void build_tree()
    {
        m_tree.clear();
        m_crit_nodes.clear();
        const size_type np = m_codes.size();
        if (!np) {
            return;
        }
        node_type root{};
        root.begin = 0;
        root.end = np;
        root.code = 1;
        root.level = 0;
        const bool root_is_crit = np <= m_ncrit || np <= m_max_leaf_n;
        subtree below;
        std::future<void> root_props = std::async(std::launch::async, [this, &root] { fill_node_properties(root); });
        try {
            if (np > m_max_leaf_n) {
                below = build_below_par(0, 1, 0, np, root_is_crit);
            }
        } catch (...) {
            root_props.wait();
            throw;
        }
        root_props.get();
        root.n_children = below.nodes.size();
        m_tree.reserve(below.nodes.size() + 1u);
        m_tree.push_back(root);
        m_tree.insert(m_tree.end(), below.nodes.begin(), below.nodes.end());
        if (root_is_crit) {
            m_crit_nodes.push_back(cnode_type{UInt(1), size_type(0), np});
        }
        m_crit_nodes.insert(m_crit_nodes.end(), below.crit.begin(), below.crit.end());
        assert(!m_crit_nodes.empty() && m_crit_nodes[0].begin == 0u && m_crit_nodes.back().end == np);
    }

    // Box size = 2 * max |coordinate| plus 5% (behaviour of tree.hpp:1279-1319).
    template <typename It>
    static F determine_box_size(const std::array<It, NDim + 1u> &cm_it, size_type n)
    {
        std::mutex mx;
        F global_max(0);
        parallel_blocks(n, 1u << 18, [&](std::size_t b, std::size_t e) {
            F local(0);
            for (std::size_t j = 0; j < NDim; ++j) {
                for (std::size_t i = b; i < e; ++i) {
                    const F tmp = std::abs(*(cm_it[j] + static_cast<typename std::iterator_traits<It>::difference_type>(i)));
                    if (!std::isfinite(tmp)) {
                        throw std::invalid_argument("While trying to automatically determine the domain size, a "
                                                    "non-finite coordinate with absolute value "
                                                    + std::to_string(tmp) + " was encountered");
                    }
                    local = std::max(local, tmp);
                }
            }
            std::lock_guard<std::mutex> lk(mx);
            global_max = std::max(global_max, local);
        });
        F retval = global_max * F(2);
        retval = std::fma(retval, F(1) / F(20), retval);
        if (!std::isfinite(retval)) {
            throw std::invalid_argument("The automatic deduction of the domain size produced the non-finite value "
                                        + std::to_string(retval));
        }
        return retval;
    }

    // Compute the Morton codes of the current particles and the permutation that sorts them.
    // Ties are broken by index, i.e. the sort is stable (the reference's tbb::parallel_sort leaves the
    // order of equal codes unspecified, tree.hpp:1267-1274).
    std::vector<size_type> encode_and_sort()
    {
        const size_type np = m_parts[0].size();
        const F inv_box_size = F(1) / m_box_size;
        struct key {
            UInt code;
            size_type idx;
        };
        std::vector<key> keys(np), tmp(np);
        parallel_blocks(np, 1u << 16, [&](std::size_t b, std::size_t e) {
            for (std::size_t i = b; i < e; ++i) {
                std::uint64_t d[NDim];
                for (std::size_t j = 0; j < NDim; ++j) {
                    d[j] = disc_single_coord(m_parts[j][i], inv_box_size);
                }
                keys[i].code = static_cast<UInt>(detail::morton_encode<NDim>(d));
                keys[i].idx = i;
            }
        });
        auto less = [](const key &a, const key &b) { return a.code < b.code || (a.code == b.code && a.idx < b.idx); };
        // Parallel merge sort: sort equal chunks, then merge pairs of runs until one run is left.
        std::size_t nruns = 1;
        while (nruns < host_threads() && np / (nruns * 2) >= (1u << 15)) {
            nruns *= 2;
        }
        std::vector<std::size_t> bounds(nruns + 1);
        for (std::size_t r = 0; r <= nruns; ++r) {
            bounds[r] = np * r / nruns;
        }
        {
            std::vector<std::thread> th;
            for (std::size_t r = 0; r < nruns; ++r) {
                th.emplace_back([&, r] { std::sort(keys.begin() + bounds[r], keys.begin() + bounds[r + 1], less); });
            }
            for (auto &t : th) {
                t.join();
            }
        }
        key *src = keys.data(), *dst = tmp.data();
        for (std::size_t width = 1; width < nruns; width *= 2) {
            std::vector<std::thread> th;
            for (std::size_t r = 0; r < nruns; r += 2 * width) {
                const std::size_t lo = bounds[r], mid = bounds[std::min(r + width, nruns)],
                                  hi = bounds[std::min(r + 2 * width, nruns)];
                th.emplace_back([=] { std::merge(src + lo, src + mid, src + mid, src + hi, dst + lo, less); });
            }
            for (auto &t : th) {
                t.join();
            }
            std::swap(src, dst);
        }
        std::vector<size_type> order(np);
        m_codes.resize(np);
        parallel_blocks(np, 1u << 18, [&](std::size_t b, std::size_t e) {
            for (std::size_t i = b; i < e; ++i) {
                order[i] = src[i].idx;
                m_codes[i] = src[i].code;
            }
        });
        return order;
    }

    template <typename Vec>
    static void apply_order(Vec &v, const std::vector<size_type> &order)
    {
        Vec nv(v.size());
        parallel_blocks(v.size(), 1u << 18, [&](std::size_t b, std::size_t e) {
            for (std::size_t i = b; i < e; ++i) {
                nv[i] = v[order[i]];
            }
        });
        v = std::move(nv);
    }

    // Device-side construction (rk_state_build): the GPU encodes, sorts and builds; the host receives the arrays its
    // accessors expose (particles in Morton order, codes, permutation, nodes, critical nodes) and keeps the resulting
    // traversal state as the replica of device 0.
    void sort_and_build_on_device()
    {
        if constexpr (wide_codes) {
            sort_and_build_on_device_impl();
        }
    }
    void sort_and_build_on_device_impl()
    {
        const size_type np = m_parts[0].size();
        if (!m_box_size_deduced && m_box_size == F(0) && np) {
            // The C ABI spells "deduce the box" as box_size == 0. An EXPLICIT zero box cannot hold a particle: report
            // it as the host path (and the reference, tree.hpp:381-429) do instead of silently deducing one.
            for (std::size_t j = 0; j < NDim; ++j) {
                disc_single_coord(m_parts[j][0], F(1) / m_box_size);
            }
        }
        const void *parts[4] = {};
        for (std::size_t j = 0; j < NDim + 1u; ++j) {
            parts[j] = m_parts[j].data();
        }
        detail::device_state ds;
        throw_status(rk_state_build_nd(&ds.h, static_cast<int>(NDim), std::is_same_v<F, float> ? RK_F32 : RK_F64,
                                       MAC == mac::bh ? RK_MAC_BH : RK_MAC_BH_GEOM, 0, parts, 0,
                                       static_cast<std::int64_t>(np),
                                       m_box_size_deduced ? 0. : static_cast<double>(m_box_size), m_max_leaf_n, m_ncrit));
        std::int64_t info[8], tinfo[4];
        double box = 0.;
        throw_status(rk_state_info(ds.h, info));
        throw_status(rk_state_tree_info(ds.h, &box, tinfo));
        m_box_size = static_cast<F>(box);
        {
            // One transfer of the {x, y, z, m} records (z = 0 for quadtrees), de-interleaved by the host threads.
            std::vector<F> aos(static_cast<std::size_t>(np) * 4u);
            if (np) {
                throw_status(rk_state_download(ds.h, 8, aos.data()));
            }
            parallel_blocks(np, 1u << 16, [&](std::size_t b, std::size_t e) {
                for (std::size_t i = b; i < e; ++i) {
                    for (std::size_t j = 0; j < NDim; ++j) {
                        m_parts[j][i] = aos[4u * i + j];
                    }
                    m_parts[NDim][i] = aos[4u * i + 3u];
                }
            });
        }
        m_codes.resize(np);
        std::vector<std::uint64_t> order(np);
        m_tree.resize(static_cast<std::size_t>(info[1]));
        std::vector<std::uint64_t> crit(static_cast<std::size_t>(info[2]) * 3u);
        if (np) {
            throw_status(rk_state_download(ds.h, 4, m_codes.data()));
            throw_status(rk_state_download(ds.h, 5, order.data()));
            throw_status(rk_state_download(ds.h, 6, m_tree.data()));
            throw_status(rk_state_download(ds.h, 7, crit.data()));
        }
        m_crit_nodes.resize(crit.size() / 3u);
        for (std::size_t i = 0; i < m_crit_nodes.size(); ++i) {
            m_crit_nodes[i] = cnode_type{static_cast<UInt>(crit[3 * i]), static_cast<size_type>(crit[3 * i + 1]),
                                         static_cast<size_type>(crit[3 * i + 2])};
        }
        // order[i] = index (in the order before this build) of the particle now at Morton position i.
        std::vector<size_type> new_perm(np);
        m_last_perm.resize(np);
        m_inv_perm.resize(np);
        parallel_blocks(np, 1u << 16, [&](std::size_t b, std::size_t e) {
            for (std::size_t i = b; i < e; ++i) {
                m_last_perm[i] = static_cast<size_type>(order[i]);
                new_perm[i] = m_perm[m_last_perm[i]];
            }
        });
        m_perm = std::move(new_perm);
        parallel_blocks(np, 1u << 16, [&](std::size_t b, std::size_t e) {
            for (std::size_t i = b; i < e; ++i) {
                m_inv_perm[m_perm[i]] = i;
            }
        });
        std::lock_guard<std::mutex> lk(m_dev_mutex);
        m_dev.clear();
        m_dev.emplace_back(std::move(ds));
    }

    // Re-establish codes, Morton order, permutations and the tree for the current particle data
    // (construction: tree.hpp:1435-1486; after an update: tree.hpp:3678-3743).
    void sort_and_build()
    {
        if (m_device_build) {
            if constexpr (!wide_codes) {
                throw std::invalid_argument("kwargs::device_build needs 64-bit Morton codes: the device builder produces "
                                            "the 21 / 31 bits per coordinate of tree<NDim, F, std::uint64_t>");
            }
            if (rk_has_accelerator()) {
                sort_and_build_on_device();
                return;
            }
        }
        const size_type np = m_parts[0].size();
        if (m_box_size_deduced) {
            m_box_size = determine_box_size(p_its_u(), np);
        }
        auto order = encode_and_sort();
        for (auto &p : m_parts) {
            apply_order(p, order);
        }
        apply_order(m_perm, order);
        m_inv_perm.resize(np);
        parallel_blocks(np, 1u << 18, [&](std::size_t b, std::size_t e) {
            for (std::size_t i = b; i < e; ++i) {
                m_inv_perm[m_perm[i]] = i;
            }
        });
        m_last_perm = std::move(order);
        build_tree();
    }

    template <typename It>
    void construct_impl(F box_size, bool deduced, const std::array<It, NDim + 1u> &its, size_type n,
                        size_type max_leaf_n, size_type ncrit)
    {
        m_box_size = box_size;
        m_box_size_deduced = deduced;
        m_max_leaf_n = max_leaf_n;
        m_ncrit = ncrit;
        // Parameter checks and messages of tree.hpp:1350-1362.
        if (!std::isfinite(m_box_size) || m_box_size < F(0)) {
            throw std::invalid_argument("The box size must be a finite non-negative value, but it is "
                                        + std::to_string(box_size) + " instead");
        }
        if (!max_leaf_n) {
            throw std::invalid_argument("The maximum number of particles per leaf must be nonzero");
        }
        if (!ncrit) {
            throw std::invalid_argument("The critical number of particles for the vectorised computation of the "
                                        "potentials/accelerations must be nonzero");
        }
        for (std::size_t j = 0; j < NDim + 1u; ++j) {
            m_parts[j].resize(n);
            auto it = its[j];
            F *dst = m_parts[j].data();
            parallel_blocks(n, 1u << 18, [&](std::size_t b, std::size_t e) {
                using diff_t = typename std::iterator_traits<It>::difference_type;
                std::copy(it + static_cast<diff_t>(b), it + static_cast<diff_t>(e), dst + b);
            });
        }
        m_perm.resize(n);
        std::iota(m_perm.begin(), m_perm.end(), size_type(0));
        sort_and_build();
    }

    void adopt_vectors(F box_size, bool deduced, std::array<f_vector<F>, NDim + 1u> &&data, size_type max_leaf_n,
                       size_type ncrit)
    {
        std::array<const F *, NDim + 1u> its;
        for (std::size_t j = 0; j < NDim + 1u; ++j) {
            its[j] = data[j].data();
        }
        construct_impl(box_size, deduced, its, data[0].size(), max_leaf_n, ncrit);
    }

public:
    // Default constructor: empty tree (tree.hpp:1522-1525).
    tree() : m_box_size(0), m_box_size_deduced(false), m_max_leaf_n(default_max_leaf_n), m_ncrit(default_ncrit) {}

private:
    template <typename... KwArgs>
    struct generic_ctor_enabler_impl : std::false_type {
    };
    template <typename T>
    struct generic_ctor_enabler_impl<T> : std::negation<std::is_same<tree, uncvref_t<T>>> {
    };
    template <typename T, typename U, typename... Args>
    struct generic_ctor_enabler_impl<T, U, Args...> : std::true_type {
    };
    template <typename... KwArgs>
    using generic_ctor_enabler = std::enable_if_t<generic_ctor_enabler_impl<KwArgs &&...>::value, int>;

    // Compile-time checks over kwargs::coords<0> .. coords<NDim - 1>.
    template <typename P, std::size_t... I>
    static constexpr bool kw_has_coords(std::index_sequence<I...>)
    {
        return (P::has(kwargs::coords<I>) && ...);
    }
    template <typename P, std::size_t... I>
    static constexpr bool kw_dup_coords(std::index_sequence<I...>)
    {
        return (P::duplicated(kwargs::coords<I>) || ...);
    }
    template <typename D, typename P, std::size_t... I>
    static constexpr bool kw_same_coords(std::index_sequence<I...>)
    {
        return (std::is_same_v<D, uncvref_t<decltype(std::declval<const P &>()(kwargs::coords<I>))>> && ...);
    }
    using dim_seq = std::make_index_sequence<NDim>;

    template <typename P, typename M, std::size_t... I>
    void construct_from_ranges(const P &p, const M &m, F box_size, bool deduced, size_type max_leaf_n, size_type ncrit,
                               std::index_sequence<I...>)
    {
        const auto size_of = [](const auto &r) { return checked_cast<size_type>(std::distance(std::begin(r), std::end(r))); };
        const size_type sizes[] = {size_of(p(kwargs::coords<I>))...};
        const size_type n = sizes[0];
        for (const size_type sz : sizes) {
            if (sz != n) {
                throw std::invalid_argument("The input ranges for the particle coordinates have inconsistent sizes");
            }
        }
        const auto nm = size_of(m);
        if (nm != n) {
            throw std::invalid_argument("The size of the input range for the particle masses (" + std::to_string(nm)
                                        + ") is different from the size of "
                                          "the input ranges for the particle coordinates ("
                                        + std::to_string(n) + ")");
        }
        const auto cbegin_of = [](const auto &r) { return std::begin(r); }; // const iterators whatever was bound
        construct_impl(box_size, deduced, std::array{cbegin_of(p(kwargs::coords<I>))..., cbegin_of(m)}, n, max_leaf_n,
                       ncrit);
    }
    template <typename D, typename P, std::size_t... I>
    void construct_from_iterators(const P &p, F box_size, bool deduced, size_type n, size_type max_leaf_n,
                                  size_type ncrit, std::index_sequence<I...>)
    {
        construct_impl(box_size, deduced, std::array<D, NDim + 1u>{p(kwargs::coords<I>)..., p(kwargs::masses)}, n,
                       max_leaf_n, ncrit);
    }

public:
    // Generic constructor with keyword arguments (tree.hpp:1573-1733): coordinates and masses as ranges
    // or as iterators + kwargs::nparts; optional kwargs::box_size, max_leaf_n, ncrit.
    template <typename... KwArgs, generic_ctor_enabler<KwArgs &&...> = 0>
    explicit tree(KwArgs &&... args)
    {
        kw_detail::parser<KwArgs...> p{std::forward<KwArgs>(args)...};
        using P = decltype(p);
        static_assert(!P::has_unnamed_arguments(),
                      "All the arguments for the generic constructor must be keyword arguments.");
        static_assert(kw_has_coords<P>(dim_seq{}) && P::has(kwargs::masses),
                      "The generic tree constructor needs particle coordinates for every dimension, and particle "
                      "masses.");
        static_assert(!kw_dup_coords<P>(dim_seq{}) && !P::duplicated(kwargs::masses)
                          && !P::duplicated(kwargs::box_size) && !P::duplicated(kwargs::max_leaf_n)
                          && !P::duplicated(kwargs::ncrit) && !P::duplicated(kwargs::nparts),
                      "The generic constructor cannot have duplicate keyword arguments.");

        F box_size(0);
        bool deduced = true;
        if constexpr (P::has(kwargs::box_size)) {
            box_size = checked_cast<F>(p(kwargs::box_size));
            deduced = false;
        }
        if constexpr (P::has(kwargs::device_build)) {
            m_device_build = static_cast<bool>(p(kwargs::device_build));
        }
        size_type max_leaf_n = default_max_leaf_n, ncrit = default_ncrit;
        if constexpr (P::has(kwargs::max_leaf_n)) {
            max_leaf_n = checked_cast<size_type>(p(kwargs::max_leaf_n));
        }
        if constexpr (P::has(kwargs::ncrit)) {
            ncrit = checked_cast<size_type>(p(kwargs::ncrit));
        }

        using data_t = uncvref_t<decltype(p(kwargs::coords<0>))>;
        static_assert(kw_same_coords<data_t, P>(dim_seq{}),
                      "All particle data in the generic tree constructor must be passed in as the same type.");
        static_assert(std::is_same_v<data_t, uncvref_t<decltype(p(kwargs::masses))>>,
                      "The type of the particle masses data is not consistent with the type of the particle "
                      "coordinates data.");

        if constexpr (is_range<const data_t &>::value) {
            static_assert(!P::has(kwargs::nparts), "If the particle coordinates are provided as ranges, the "
                                                   "'nparts' keyword argument must not be provided.");
            construct_from_ranges(p, p(kwargs::masses), box_size, deduced, max_leaf_n, ncrit, dim_seq{});
        } else {
            static_assert(P::has(kwargs::nparts), "If the particle coordinates are provided as iterators, the "
                                                  "'nparts' keyword argument must also be provided.");
            const auto n = checked_cast<size_type>(p(kwargs::nparts));
            construct_from_iterators<data_t>(p, box_size, deduced, n, max_leaf_n, ncrit, dim_seq{});
        }
    }

    // Copy / move: host data is copied or moved, device replicas are re-created on demand
    // (the reference re-creates its accelerator views after every copy/move, tree.hpp:1742-1824).
    tree(const tree &o)
        : m_box_size(o.m_box_size), m_box_size_deduced(o.m_box_size_deduced), m_max_leaf_n(o.m_max_leaf_n),
          m_ncrit(o.m_ncrit), m_device_build(o.m_device_build), m_parts(o.m_parts), m_codes(o.m_codes), m_perm(o.m_perm),
          m_last_perm(o.m_last_perm),
          m_inv_perm(o.m_inv_perm), m_tree(o.m_tree), m_crit_nodes(o.m_crit_nodes)
    {
    }
    tree(tree &&o) noexcept
        : m_box_size(o.m_box_size), m_box_size_deduced(o.m_box_size_deduced), m_max_leaf_n(o.m_max_leaf_n),
          m_ncrit(o.m_ncrit), m_device_build(o.m_device_build), m_parts(std::move(o.m_parts)), m_codes(std::move(o.m_codes)),
          m_perm(std::move(o.m_perm)), m_last_perm(std::move(o.m_last_perm)), m_inv_perm(std::move(o.m_inv_perm)),
          m_tree(std::move(o.m_tree)), m_crit_nodes(std::move(o.m_crit_nodes)), m_dev(std::move(o.m_dev))
    {
        o.clear();
    }
    tree &operator=(const tree &o)
    {
        if (this != &o) {
            tree tmp(o);
            *this = std::move(tmp);
        }
        return *this;
    }
    tree &operator=(tree &&o) noexcept
    {
        if (this != &o) {
            reset_device_state();
            m_box_size = o.m_box_size;
            m_box_size_deduced = o.m_box_size_deduced;
            m_max_leaf_n = o.m_max_leaf_n;
            m_ncrit = o.m_ncrit;
            m_device_build = o.m_device_build;
            m_parts = std::move(o.m_parts);
            m_codes = std::move(o.m_codes);
            m_perm = std::move(o.m_perm);
            m_last_perm = std::move(o.m_last_perm);
            m_inv_perm = std::move(o.m_inv_perm);
            m_tree = std::move(o.m_tree);
            m_crit_nodes = std::move(o.m_crit_nodes);
            m_dev = std::move(o.m_dev);
            o.clear();
        }
        return *this;
    }
    ~tree() = default;

    // Reset to a default-constructed state.
    void clear() noexcept
    {
        reset_device_state();
        m_box_size = F(0);
        m_box_size_deduced = false;
        m_max_leaf_n = default_max_leaf_n;
        m_ncrit = default_ncrit;
        for (auto &p : m_parts) {
            p.clear();
        }
        m_codes.clear();
        m_perm.clear();
        m_last_perm.clear();
        m_inv_perm.clear();
        m_tree.clear();
        m_crit_nodes.clear();
    }

    friend std::ostream &operator<<(std::ostream &os, const tree &t)
    {
        os << "Box size                 : " << t.m_box_size << (t.m_box_size_deduced ? " (deduced)" : "") << '\n';
        os << "Total number of particles: " << t.nparts() << '\n';
        os << "Total number of nodes    : " << t.m_tree.size() << '\n';
        os << "Critical nodes           : " << t.m_crit_nodes.size() << '\n';
        return os;
    }

private:
    // ------------------------------------------------------------------------------------------
    // Device state (the seam the reference fills with rocm_state, tree.hpp:1495-1519, 3882-3884).
    // ------------------------------------------------------------------------------------------
    void reset_device_state() noexcept
    {
        m_dev.clear();
    }

    rk_state *device_state_for(int device) const
    {
        std::lock_guard<std::mutex> lk(m_dev_mutex);
        if (m_dev.size() <= static_cast<std::size_t>(device)) {
            m_dev.resize(static_cast<std::size_t>(device) + 1u);
        }
        auto &d = m_dev[static_cast<std::size_t>(device)];
        if (!d.h && device != 0 && m_dev[0].h) {
            // Further devices get a replica of device 0's state: buffers travel device to device (xGMI peer copies)
            // instead of being converted and uploaded from host memory once more (the reference re-uploads everything
            // to every device on every call, src/rakau_cuda.cu:410-527).
            throw_status(rk_state_clone(&d.h, m_dev[0].h, device));
        }
        if (!d.h) {
            const void *parts[4] = {};
            for (std::size_t j = 0; j < NDim + 1u; ++j) {
                parts[j] = m_parts[j].data();
            }
            // The seam takes node records with 64-bit code / level fields (tree_node_t<NDim, F, uint64_t, MAC>); trees
            // with 32-bit codes hand over a widened copy (the device side only uses the topology and the properties).
            using wide_node = tree_node_t<NDim, F, std::uint64_t, MAC>;
            std::vector<wide_node> widened;
            const void *nodes = m_tree.data();
            if constexpr (!wide_codes) {
                widened.resize(m_tree.size());
                parallel_blocks(m_tree.size(), 1u << 15, [&](std::size_t b, std::size_t e) {
                    for (std::size_t i = b; i < e; ++i) {
                        const auto &n = m_tree[i];
                        auto &w = widened[i];
                        w.begin = n.begin, w.end = n.end, w.n_children = n.n_children, w.code = n.code, w.level = n.level;
                        std::copy(std::begin(n.props), std::end(n.props), std::begin(w.props));
                        if constexpr (MAC == mac::bh) {
                            w.dim2 = n.dim2;
                        } else {
                            w.dim = n.dim, w.delta = n.delta;
                        }
                    }
                });
                nodes = widened.data();
            }
            throw_status(rk_state_create_nd(&d.h, static_cast<int>(NDim), std::is_same_v<F, float> ? RK_F32 : RK_F64,
                                            MAC == mac::bh ? RK_MAC_BH : RK_MAC_BH_GEOM, device, parts, nullptr,
                                            static_cast<std::int64_t>(nparts()), nodes,
                                            static_cast<std::int64_t>(m_tree.size()),
                                            static_cast<std::int64_t>(sizeof(wide_node)), m_ncrit));
        }
        return d.h;
    }

    // ------------------------------------------------------------------------------------------
    // acc/pot dispatch (behaviour of tree.hpp:2853-3357).
    // ------------------------------------------------------------------------------------------

    // Validation of `split` common to all code paths (tree.hpp:2857-2868).
    static void check_split(const std::vector<double> &split)
    {
        if (std::any_of(split.begin(), split.end(), [](double x) { return !std::isfinite(x); })) {
            throw std::invalid_argument("The 'split' parameter cannot contain non-finite values");
        }
        if (std::any_of(split.begin(), split.end(), [](double x) { return x < 0.; })) {
            throw std::invalid_argument("The 'split' parameter must contain only non-negative values");
        }
        if (!split.empty() && std::all_of(split.begin(), split.end(), [](double x) { return x == 0.; })) {
            throw std::invalid_argument("The values in the 'split' parameter cannot all be zero");
        }
    }

    // Particle indices at which the work is cut: split = {cpu, dev0, dev1, ...} (tree.hpp:3150-3187). The result has
    // split.size() + 1 entries: [cuts[0], cuts[1]) is the share of the CPU engine, [cuts[d + 1], cuts[d + 2]) the share
    // of device d. Every cut sits on a critical-node boundary (the snapping rule of tree.hpp:3053-3063 applied to every
    // boundary: this engine's unit of work is the critical node on the devices too).
    // An EMPTY split means "everything on device 0" here (the reference's default is the CPU; this library exists to
    // offload), a split of size one is the reference's "CPU only". Compile with -DRAKAU_AMD_EMPTY_SPLIT_IS_CPU for the
    // reference's meaning of the empty split (tree.hpp:3114-3117 of the reference: no accelerator share, CPU engine).
    std::vector<size_type> split_cuts(const std::vector<double> &split) const
    {
        check_split(split);
        const size_type np = nparts();
        if (split.empty()) {
#if defined(RAKAU_AMD_EMPTY_SPLIT_IS_CPU)
            return {size_type(0), np};
#else
            return {size_type(0), size_type(0), np};
#endif
        }
        if (split.size() == 1u) {
            return {size_type(0), np};
        }
        const auto n_dev = static_cast<std::size_t>(rk_device_count());
        if (split.size() - 1u > n_dev) {
            // Message of tree.hpp:3136-3141.
            throw std::invalid_argument(
                "Cannot split the computation of accelerations/potentials: the split vector refers to "
                + std::to_string(split.size() - 1u) + " accelerators, but only " + std::to_string(n_dev)
                + " were detected");
        }
        const double total = std::accumulate(split.begin(), split.end(), 0.);
        std::vector<size_type> cuts(split.size() + 1u);
        cuts[0] = 0;
        double acc = 0.;
        for (std::size_t d = 0; d + 1u < split.size(); ++d) {
            acc += split[d];
            auto idx = checked_cast<size_type>(acc / total * static_cast<double>(np));
            const auto it = std::lower_bound(m_crit_nodes.begin(), m_crit_nodes.end(), idx,
                                             [](const cnode_type &cn, size_type v) { return cn.begin < v; });
            idx = (it == m_crit_nodes.end()) ? np : it->begin;
            cuts[d + 1u] = std::max(idx, cuts[d]);
        }
        cuts.back() = np;
        // A device share below the minimum size sends the whole call to the CPU engine (tree.hpp:3191-3199,
        // 3114-3117: "not enough particles, run on the cpu").
        for (std::size_t d = 1; d + 1u < cuts.size(); ++d) {
            if (cuts[d + 1u] - cuts[d] < rk_min_size()) {
                return {size_type(0), np};
            }
        }
        return cuts;
    }

    // The CPU engine on the critical nodes whose particles are [0, p_end) (p_end is a critical-node boundary).
    template <unsigned Q>
    void cpu_run(const std::array<F *, nvecs_res<Q>> &res, size_type p_end, F mac_value, F G, F eps2,
                 detail::cpu_flavour flavour = detail::cpu_flavour::automatic, unsigned nthreads = 0) const
    {
        const auto it = std::lower_bound(m_crit_nodes.begin(), m_crit_nodes.end(), p_end,
                                         [](const cnode_type &cn, size_type v) { return cn.begin < v; });
        const auto c_end = static_cast<std::size_t>(it - m_crit_nodes.begin());
        std::array<const F *, NDim + 1u> parts;
        for (std::size_t j = 0; j < NDim + 1u; ++j) {
            parts[j] = m_parts[j].data();
        }
        F *out[NDim + 1u] = {};
        for (std::size_t j = 0; j < nvecs_res<Q>; ++j) {
            out[j] = res[j];
        }
        nthreads = nthreads ? nthreads : usable_hw_threads();
        if (flavour != detail::cpu_flavour::scalar && detail::cpu::native_width<F> * sizeof(F) < 64u) {
            // This translation unit was not compiled for AVX-512: let the library run the engine's AVX-512 build if
            // the CPU has it (same algorithm, wider batches).
            rk_cpu_job job{};
            job.q = static_cast<int>(Q), job.ndim = static_cast<int>(NDim), job.fp = std::is_same_v<F, float> ? RK_F32 : RK_F64;
            job.code_bits = static_cast<int>(sizeof(UInt) * 8u), job.mac = MAC == mac::bh ? RK_MAC_BH : RK_MAC_BH_GEOM;
            job.flavour = static_cast<int>(flavour), job.nthreads = nthreads;
            job.tree = m_tree.data(), job.tree_size = m_tree.size();
            job.crit = m_crit_nodes.data(), job.c_begin = 0, job.c_end = c_end;
            for (std::size_t j = 0; j < NDim + 1u; ++j) {
                job.parts[j] = parts[j];
                job.out[j] = out[j];
            }
            job.mac_value = static_cast<double>(mac_value), job.G = static_cast<double>(G), job.eps2 = static_cast<double>(eps2);
            const int rc = rk_cpu_engine_run(&job);
            if (rc == RK_OK) {
                return;
            }
            if (rc > 0) {
                throw_status(rc);
            }
        }
        detail::cpu::run<Q, NDim, MAC == mac::bh>(m_tree.data(), m_tree.size(), m_crit_nodes.data(), std::size_t(0), c_end,
                                                  parts, out, mac_value, G, eps2, flavour, nthreads);
    }

    // Device 0's state, then replicas of it on every further device that has a share and none yet: one rk_state_clone_all
    // call, whose peer copies fan out as a doubling tree over xGMI (the reference converts and uploads tree and particles
    // to every device from the host on every call: src/rakau_cuda.cu:410-527).
    void replicate_to_devices(const std::vector<size_type> &cuts, std::size_t n_dev) const
    {
        rk_state *first = device_state_for(0);
        std::lock_guard<std::mutex> lk(m_dev_mutex);
        if (m_dev.size() < n_dev) {
            m_dev.resize(n_dev);
        }
        std::vector<int> need;
        for (std::size_t d = 1; d < n_dev; ++d) {
            if (cuts[d + 2u] > cuts[d + 1u] && !m_dev[d].h) {
                need.push_back(static_cast<int>(d));
            }
        }
        if (need.empty()) {
            return;
        }
        std::vector<rk_state *> made(need.size(), nullptr);
        throw_status(rk_state_clone_all(made.data(), first, need.data(), static_cast<int>(need.size())));
        for (std::size_t i = 0; i < need.size(); ++i) {
            m_dev[static_cast<std::size_t>(need[i])].h = made[i];
        }
    }

    // Run the engines for [0, nparts) and leave the results in res[j][0..nparts) (Morton order): the devices on one
    // host thread each (rocm_state::acc_pot / cuda_acc_pot_impl of the reference), the CPU share on the calling
    // thread meanwhile; every future is joined so that exceptions propagate (tree.hpp:3071-3113).
    template <unsigned Q>
    void device_run(const std::array<F *, nvecs_res<Q>> &res, F mac_value, F G, F eps2,
                    const std::vector<double> &split) const
    {
        const auto cuts = split_cuts(split);
        if (!nparts()) {
            return;
        }
        const std::size_t n_dev = cuts.size() - 2u; // devices with a share
        bool any_dev = false;
        for (std::size_t d = 0; d < n_dev; ++d) {
            any_dev = any_dev || cuts[d + 2u] > cuts[d + 1u];
        }
        if (any_dev && !rk_has_accelerator()) {
            throw std::runtime_error("rakau_amd: no gfx950 accelerator is available for the device share of the "
                                     "computation (split = {1} selects the CPU engine)");
        }
        void *out[4] = {};
        for (std::size_t j = 0; j < nvecs_res<Q>; ++j) {
            out[j] = res[j];
        }
        auto run_one = [&](int device, size_type b, size_type e) {
            if (b == e) {
                return;
            }
            throw_status(rk_acc_pot(device_state_for(device), static_cast<int>(Q), static_cast<std::int64_t>(b),
                                    static_cast<std::int64_t>(e), out, static_cast<double>(mac_value),
                                    static_cast<double>(G), static_cast<double>(eps2), 1));
        };
        const bool cpu_share = cuts[1] > cuts[0];
        if (!cpu_share && n_dev == 1u) {
            run_one(0, cuts[1], cuts[2]); // the common case: no thread
            return;
        }
        if (n_dev > 1u) {
            // Every replica exists before the first device thread starts: the threads then only look handles up, and no
            // state is created (or has its host mirrors filled) while another thread traverses with it.
            replicate_to_devices(cuts, n_dev);
        }
        std::vector<std::future<void>> futs;
        // Device 0 gets a thread of its own only if this thread is busy with the CPU share.
        for (std::size_t d = cpu_share ? 0u : 1u; d < n_dev; ++d) {
            futs.emplace_back(std::async(std::launch::async, run_one, static_cast<int>(d), cuts[d + 1u], cuts[d + 2u]));
        }
        std::exception_ptr ep;
        try {
            if (cpu_share) {
                cpu_run<Q>(res, cuts[1], mac_value, G, eps2);
            } else {
                run_one(0, cuts[1], cuts[2]);
            }
        } catch (...) {
            ep = std::current_exception();
        }
        for (auto &f : futs) {
            try {
                f.get();
            } catch (...) {
                if (!ep) {
                    ep = std::current_exception();
                }
            }
        }
        if (ep) {
            std::rethrow_exception(ep);
        }
    }

    static F compute_eps2(F eps)
    {
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
        return eps2;
    }
    static void check_G_const(F G)
    {
        // tree.hpp:3283-3289.
        if (!std::isfinite(G)) {
            throw std::domain_error("The value of the gravitational constant G must be finite, but it is "
                                    + std::to_string(G) + " instead");
        }
    }

    // Original-order outputs without a host-side scatter: when device 0 computes everything (no CPU share, one device), the
    // kernels scatter through perm themselves (rk_acc_pot with RK_OUT_ORDERED: into a buffer in HBM, from where the ordered
    // arrays travel in one piece each). Returns false if the call is not of that kind (the caller then stages and scatters).
    template <unsigned Q, typename It>
    bool ordered_on_device(const std::array<It, nvecs_res<Q>> &out, F mac_value, F G, F eps2, const std::vector<double> &split) const
    {
        if constexpr (std::is_same_v<It, F *>) {
            const auto cuts = split_cuts(split);
            if (!nparts() || cuts.size() != 3u || cuts[1] > cuts[0] || !rk_has_accelerator()) {
                return false;
            }
            rk_state *st = device_state_for(0);
            {
                std::lock_guard<std::mutex> lk(m_dev_mutex);
                if (!m_dev[0].perm_set) {
                    static_assert(sizeof(size_type) == sizeof(std::uint64_t));
                    throw_status(rk_state_set_perm(st, reinterpret_cast<const std::uint64_t *>(m_perm.data())));
                    m_dev[0].perm_set = true;
                }
            }
            void *o[4] = {};
            for (std::size_t j = 0; j < nvecs_res<Q>; ++j) {
                o[j] = out[j];
            }
            throw_status(rk_acc_pot(st, static_cast<int>(Q), 0, static_cast<std::int64_t>(nparts()), o, static_cast<double>(mac_value),
                                    static_cast<double>(G), static_cast<double>(eps2), RK_OUT_OFFSET | RK_OUT_ORDERED));
            return true;
        } else {
            (void)out, (void)mac_value, (void)G, (void)eps2, (void)split;
            return false;
        }
    }

    template <bool Ordered, unsigned Q, typename It>
    void acc_pot_dispatch(const std::array<It, nvecs_res<Q>> &out, F orig_mac_value, F G, F eps,
                          const std::vector<double> &split) const
    {
        // Checks and transforms of tree.hpp:3299-3319.
        if (!std::isfinite(orig_mac_value) || orig_mac_value <= F(0)) {
            throw std::domain_error("The MAC value must be finite and positive, but it is "
                                    + std::to_string(orig_mac_value) + " instead");
        }
        const F mac_value
            = MAC == mac::bh ? F(1) / (orig_mac_value * orig_mac_value) : F(1) / orig_mac_value;
        if (!std::isfinite(mac_value) || mac_value <= F(0)) {
            throw std::domain_error("The transformed MAC value must be finite and positive, but it is "
                                    + std::to_string(mac_value) + " instead");
        }
        const F eps2 = compute_eps2(eps);
        check_G_const(G);
        const size_type np = nparts();
        if constexpr (!Ordered && std::is_same_v<It, F *>) {
            device_run<Q>(out, mac_value, G, eps2, split);
        } else if (Ordered && std::is_same_v<It, F *> && ordered_on_device<Q>(out, mac_value, G, eps2, split)) {
            // (accs_o / pots_o / accs_pots_o into plain arrays, everything on device 0: the kernels scattered through perm)
        } else {
            // Generic output iterators and/or original-order output: stage in Morton-order buffers,
            // then copy/scatter (the reference stages accelerator results the same way, tree.hpp:3081-3106).
            // The buffers are kept per thread between calls, in pinned memory where a device is there to write them (the
            // kernels then store the results into them directly: no staging inside the library, no 48 MB of fresh pages per
            // call at 4M), and the scatter through perm runs on the host threads: accs_o() at 4M 30 ms -> 6 ms.
            F *ptrs_raw[4] = {};
            detail::stage_buffers<F>::get(np, nvecs_res<Q>, ptrs_raw);
            std::array<F *, nvecs_res<Q>> ptrs;
            for (std::size_t j = 0; j < nvecs_res<Q>; ++j) {
                ptrs[j] = ptrs_raw[j];
            }
            device_run<Q>(ptrs, mac_value, G, eps2, split);
            using diff_t = typename std::iterator_traits<It>::difference_type;
            if constexpr (Ordered) {
                (void)checked_cast<diff_t>(np); // every index below fits
            }
            // One pass over the particles for all result arrays (perm is read once per particle), on the host threads.
            detail::parallel_blocks(np, std::size_t(1) << 17, [&](std::size_t b, std::size_t e) {
                if constexpr (Ordered) {
                    // out[perm[i]] = res[i] (tree.hpp:3320-3330), as a gather: out[k] = res[inv_perm[k]] -- the writes stream
                    // through the caller's arrays, the random accesses are reads.
                    for (std::size_t k = b; k < e; ++k) {
                        const auto i = m_inv_perm[k];
                        for (std::size_t j = 0; j < nvecs_res<Q>; ++j) {
                            *(out[j] + static_cast<diff_t>(k)) = ptrs[j][i];
                        }
                    }
                } else {
                    for (std::size_t j = 0; j < nvecs_res<Q>; ++j) {
                        for (std::size_t i = b; i < e; ++i) {
                            *(out[j] + static_cast<diff_t>(i)) = ptrs[j][i];
                        }
                    }
                }
            });
            detail::stage_buffers<F>::trim(detail::stage_keep_bytes);
        }
    }
    template <bool Ordered, unsigned Q, typename Allocator>
    void acc_pot_dispatch(std::array<std::vector<F, Allocator>, nvecs_res<Q>> &out, F mac_value, F G, F eps,
                          const std::vector<double> &split) const
    {
        std::array<F *, nvecs_res<Q>> ptrs;
        for (std::size_t j = 0; j < nvecs_res<Q>; ++j) {
            out[j].resize(m_parts[0].size());
            ptrs[j] = out[j].data();
        }
        acc_pot_dispatch<Ordered, Q>(ptrs, mac_value, G, eps, split);
    }
    template <bool Ordered, unsigned Q, typename Allocator>
    void acc_pot_dispatch(std::vector<F, Allocator> &out, F mac_value, F G, F eps,
                          const std::vector<double> &split) const
    {
        static_assert(Q == 1u);
        out.resize(m_parts[0].size());
        acc_pot_dispatch<Ordered, Q>(std::array<F *, 1>{out.data()}, mac_value, G, eps, split);
    }
    template <unsigned Q, typename It>
    static auto ilist_to_array(std::initializer_list<It> ilist)
    {
        // Message of tree.hpp:3364-3370.
        if (ilist.size() != nvecs_res<Q>) {
            throw std::invalid_argument(
                "An initializer list containing " + std::to_string(ilist.size())
                + " iterators was used as the output for the computation of the accelerations/potentials in a "
                + std::to_string(NDim) + "-dimensional tree, but a list with " + std::to_string(nvecs_res<Q>)
                + " iterators is required instead");
        }
        std::array<It, nvecs_res<Q>> retval;
        std::copy(ilist.begin(), ilist.end(), retval.begin());
        return retval;
    }
    // Defaults G = 1, eps = 0, split = {} (tree.hpp:3376-3403).
    template <typename... Args>
    static auto parse_accpot_kwargs(Args &&... args)
    {
        kw_detail::parser<Args...> p{std::forward<Args>(args)...};
        using P = decltype(p);
        static_assert(!P::has_unnamed_arguments(),
                      "Only keyword arguments can be passed in the parameter pack of the "
                      "functions for the computation of accelerations and potentials");
        static_assert(!P::duplicated(kwargs::G) && !P::duplicated(kwargs::eps) && !P::duplicated(kwargs::split),
                      "The functions for the computation of accelerations and/or potentials cannot "
                      "have duplicate keyword arguments.");
        F G(1), eps(0);
        std::vector<double> split;
        if constexpr (P::has(kwargs::G)) {
            G = checked_cast<F>(p(kwargs::G));
        }
        if constexpr (P::has(kwargs::eps)) {
            eps = checked_cast<F>(p(kwargs::eps));
        }
        if constexpr (P::has(kwargs::split)) {
            const auto &s = p(kwargs::split);
            split.assign(std::begin(s), std::end(s));
        }
        return std::tuple{G, eps, std::move(split)};
    }

public:
#define RAKAU_AMD_ACCPOT_API(NAME, ORDERED)                                                                            \
    template <typename Allocator, typename... KwArgs>                                                                  \
    void accs_##NAME(std::array<std::vector<F, Allocator>, NDim> &out, F mac_value, KwArgs &&... args) const           \
    {                                                                                                                  \
        const auto [G, eps, split] = parse_accpot_kwargs(std::forward<KwArgs>(args)...);                               \
        acc_pot_dispatch<ORDERED, 0>(out, mac_value, G, eps, split);                                                   \
    }                                                                                                                  \
    template <typename It, typename... KwArgs>                                                                         \
    void accs_##NAME(const std::array<It, NDim> &out, F mac_value, KwArgs &&... args) const                            \
    {                                                                                                                  \
        const auto [G, eps, split] = parse_accpot_kwargs(std::forward<KwArgs>(args)...);                               \
        acc_pot_dispatch<ORDERED, 0>(out, mac_value, G, eps, split);                                                   \
    }                                                                                                                  \
    template <typename It, typename... KwArgs>                                                                         \
    void accs_##NAME(std::initializer_list<It> out, F mac_value, KwArgs &&... args) const                              \
    {                                                                                                                  \
        accs_##NAME(ilist_to_array<0>(out), mac_value, std::forward<KwArgs>(args)...);                                 \
    }                                                                                                                  \
    template <typename Allocator, typename... KwArgs>                                                                  \
    void pots_##NAME(std::vector<F, Allocator> &out, F mac_value, KwArgs &&... args) const                             \
    {                                                                                                                  \
        const auto [G, eps, split] = parse_accpot_kwargs(std::forward<KwArgs>(args)...);                               \
        acc_pot_dispatch<ORDERED, 1>(out, mac_value, G, eps, split);                                                   \
    }                                                                                                                  \
    template <typename It, typename... KwArgs, std::enable_if_t<!is_range<It>::value, int> = 0>                        \
    void pots_##NAME(It out, F mac_value, KwArgs &&... args) const                                                     \
    {                                                                                                                  \
        const auto [G, eps, split] = parse_accpot_kwargs(std::forward<KwArgs>(args)...);                               \
        acc_pot_dispatch<ORDERED, 1>(std::array<It, 1>{out}, mac_value, G, eps, split);                                \
    }                                                                                                                  \
    template <typename Allocator, typename... KwArgs>                                                                  \
    void accs_pots_##NAME(std::array<std::vector<F, Allocator>, NDim + 1u> &out, F mac_value, KwArgs &&... args)       \
        const                                                                                                          \
    {                                                                                                                  \
        const auto [G, eps, split] = parse_accpot_kwargs(std::forward<KwArgs>(args)...);                               \
        acc_pot_dispatch<ORDERED, 2>(out, mac_value, G, eps, split);                                                   \
    }                                                                                                                  \
    template <typename It, typename... KwArgs>                                                                         \
    void accs_pots_##NAME(const std::array<It, NDim + 1u> &out, F mac_value, KwArgs &&... args) const                  \
    {                                                                                                                  \
        const auto [G, eps, split] = parse_accpot_kwargs(std::forward<KwArgs>(args)...);                               \
        acc_pot_dispatch<ORDERED, 2>(out, mac_value, G, eps, split);                                                   \
    }                                                                                                                  \
    template <typename It, typename... KwArgs>                                                                         \
    void accs_pots_##NAME(std::initializer_list<It> out, F mac_value, KwArgs &&... args) const                         \
    {                                                                                                                  \
        accs_pots_##NAME(ilist_to_array<2>(out), mac_value, std::forward<KwArgs>(args)...);                            \
    }

    // Engine hook for tests and benchmarks: the CPU engine alone on the whole tree (what split = {1} runs) with an
    // explicit arithmetic flavour and thread count (0 = all usable host threads). Q = 0 / 1 / 2 as in
    // accs_u / pots_u / accs_pots_u; results in Morton order; same checks and transforms as every acc/pot call.
    template <unsigned Q>
    void cpu_acc_pot_u(const std::array<F *, nvecs_res<Q>> &out, F orig_mac_value, F G, F eps, cpu_flavour flavour,
                       unsigned nthreads) const
    {
        if (!std::isfinite(orig_mac_value) || orig_mac_value <= F(0)) {
            throw std::domain_error("The MAC value must be finite and positive, but it is "
                                    + std::to_string(orig_mac_value) + " instead");
        }
        const F mac_value = MAC == mac::bh ? F(1) / (orig_mac_value * orig_mac_value) : F(1) / orig_mac_value;
        if (!std::isfinite(mac_value) || mac_value <= F(0)) {
            throw std::domain_error("The transformed MAC value must be finite and positive, but it is "
                                    + std::to_string(mac_value) + " instead");
        }
        const F eps2 = compute_eps2(eps);
        check_G_const(G);
        cpu_run<Q>(out, nparts(), mac_value, G, eps2, flavour, nthreads);
    }

    // accs_u / pots_u / accs_pots_u: results in Morton order (tree.hpp:3406-3451).
    RAKAU_AMD_ACCPOT_API(u, false)
    // accs_o / pots_o / accs_pots_o: results in the original particle order (tree.hpp:3452-3497).
    RAKAU_AMD_ACCPOT_API(o, true)
#undef RAKAU_AMD_ACCPOT_API

private:
    // Direct summation for one particle (behaviour of tree.hpp:3531-3569).
    template <bool Ordered, unsigned Q>
    auto exact_acc_pot_impl(size_type orig_idx, F G, F eps) const
    {
        const F eps2 = compute_eps2(eps);
        check_G_const(G);
        const size_type size = m_parts[0].size();
        std::array<F, nvecs_res<Q>> retval{};
        F diffs[NDim];
        const size_type idx = Ordered ? m_inv_perm[orig_idx] : orig_idx;
        for (size_type i = 0; i < size; ++i) {
            if (i == idx) {
                continue;
            }
            F dist2(eps2);
            for (std::size_t j = 0; j < NDim; ++j) {
                diffs[j] = m_parts[j][i] - m_parts[j][idx];
                dist2 = std::fma(diffs[j], diffs[j], dist2);
            }
            const F inv_dist = F(1) / std::sqrt(dist2), Gmi_dist = G * m_parts[NDim][i] * inv_dist;
            if constexpr (Q == 0u || Q == 2u) {
                const F Gmi_dist3 = inv_dist * inv_dist * Gmi_dist;
                for (std::size_t j = 0; j < NDim; ++j) {
                    retval[j] = std::fma(diffs[j], Gmi_dist3, retval[j]);
                }
            }
            if constexpr (Q == 1u || Q == 2u) {
                constexpr std::size_t pot_idx = Q == 1u ? 0u : NDim;
                retval[pot_idx] = std::fma(-Gmi_dist, m_parts[NDim][idx], retval[pot_idx]);
            }
        }
        return retval;
    }

public:
    template <typename... KwArgs>
    std::array<F, NDim> exact_acc_u(size_type idx, KwArgs &&... args) const
    {
        const auto [G, eps, split] = parse_accpot_kwargs(std::forward<KwArgs>(args)...);
        (void)split;
        return exact_acc_pot_impl<false, 0>(idx, G, eps);
    }
    template <typename... KwArgs>
    F exact_pot_u(size_type idx, KwArgs &&... args) const
    {
        const auto [G, eps, split] = parse_accpot_kwargs(std::forward<KwArgs>(args)...);
        (void)split;
        return exact_acc_pot_impl<false, 1>(idx, G, eps)[0];
    }
    template <typename... KwArgs>
    std::array<F, NDim + 1u> exact_acc_pot_u(size_type idx, KwArgs &&... args) const
    {
        const auto [G, eps, split] = parse_accpot_kwargs(std::forward<KwArgs>(args)...);
        (void)split;
        return exact_acc_pot_impl<false, 2>(idx, G, eps);
    }
    template <typename... KwArgs>
    std::array<F, NDim> exact_acc_o(size_type idx, KwArgs &&... args) const
    {
        const auto [G, eps, split] = parse_accpot_kwargs(std::forward<KwArgs>(args)...);
        (void)split;
        return exact_acc_pot_impl<true, 0>(idx, G, eps);
    }
    template <typename... KwArgs>
    F exact_pot_o(size_type idx, KwArgs &&... args) const
    {
        const auto [G, eps, split] = parse_accpot_kwargs(std::forward<KwArgs>(args)...);
        (void)split;
        return exact_acc_pot_impl<true, 1>(idx, G, eps)[0];
    }
    template <typename... KwArgs>
    std::array<F, NDim + 1u> exact_acc_pot_o(size_type idx, KwArgs &&... args) const
    {
        const auto [G, eps, split] = parse_accpot_kwargs(std::forward<KwArgs>(args)...);
        (void)split;
        return exact_acc_pot_impl<true, 2>(idx, G, eps);
    }

    // ------------------------------------------------------------------------------------------
    // Accessors (tree.hpp:3638-3673, 3818-3837).
    // ------------------------------------------------------------------------------------------
    std::array<const F *, NDim + 1u> p_its_u() const
    {
        std::array<const F *, NDim + 1u> r;
        for (std::size_t j = 0; j < NDim + 1u; ++j) {
            r[j] = m_parts[j].data();
        }
        return r;
    }
    auto p_its_o() const
    {
        using it_t = perm_iterator<const F *, typename std::vector<size_type>::const_iterator>;
        std::array<it_t, NDim + 1u> r;
        for (std::size_t j = 0; j < NDim + 1u; ++j) {
            r[j] = it_t(m_parts[j].data(), m_inv_perm.begin());
        }
        return r;
    }
    const UInt *c_it_u() const
    {
        return m_codes.data();
    }
    auto c_it_o() const
    {
        return perm_iterator<const UInt *, typename std::vector<size_type>::const_iterator>(m_codes.data(),
                                                                                           m_inv_perm.begin());
    }
    const auto &perm() const
    {
        return m_perm;
    }
    const auto &last_perm() const
    {
        return m_last_perm;
    }
    const auto &inv_perm() const
    {
        return m_inv_perm;
    }
    const auto &nodes() const
    {
        return m_tree;
    }
    // The target groups of the traversal (m_crit_nodes of the reference; not public there).
    const auto &crit_nodes() const
    {
        return m_crit_nodes;
    }
    F box_size() const
    {
        return m_box_size;
    }
    bool box_size_deduced() const
    {
        return m_box_size_deduced;
    }
    bool device_build() const
    {
        return m_device_build;
    }
    size_type max_leaf_n() const
    {
        return m_max_leaf_n;
    }
    size_type ncrit() const
    {
        return m_ncrit;
    }
    size_type nparts() const
    {
        return m_parts[0].size();
    }
    // Device-resident state on `device` (created on first use); for callers that keep outputs in HBM.
    rk_state *device_state(int device = 0) const
    {
        return device_state_for(device);
    }

private:
    std::array<F *, NDim + 1u> mutable_its_u()
    {
        std::array<F *, NDim + 1u> r;
        for (std::size_t j = 0; j < NDim + 1u; ++j) {
            r[j] = m_parts[j].data();
        }
        return r;
    }
    auto mutable_its_o()
    {
        using it_t = perm_iterator<F *, typename std::vector<size_type>::const_iterator>;
        std::array<it_t, NDim + 1u> r;
        for (std::size_t j = 0; j < NDim + 1u; ++j) {
            r[j] = it_t(m_parts[j].data(), m_inv_perm.begin());
        }
        return r;
    }
    // Particle update (tree.hpp:3744-3776): the functor moves particles, then everything is re-derived.
    // The device replicas are dropped BEFORE host data is touched and re-created on the next acc/pot call
    // (the reset-before-mutate ordering of tree.hpp:3681).
    template <bool Ordered, typename Func>
    void update_particles_dispatch(Func &&f)
    {
        reset_device_state();
        try {
            if constexpr (Ordered) {
                std::forward<Func>(f)(mutable_its_o());
            } else {
                std::forward<Func>(f)(mutable_its_u());
            }
            sort_and_build();
        } catch (...) {
            clear();
            throw;
        }
    }
    // Mass update (tree.hpp:3779-3804): positions are untouched, only node properties are recomputed.
    template <bool Ordered, typename Func>
    void update_masses_dispatch(Func &&f)
    {
        reset_device_state();
        try {
            if constexpr (Ordered) {
                std::forward<Func>(f)(mutable_its_o()[NDim]);
            } else {
                std::forward<Func>(f)(mutable_its_u()[NDim]);
            }
            parallel_blocks(m_tree.size(), 1u << 10, [this](std::size_t b, std::size_t e) {
                for (std::size_t i = b; i < e; ++i) {
                    fill_node_properties(m_tree[i]);
                }
            });
        } catch (...) {
            clear();
            throw;
        }
    }

public:
    template <typename Func>
    void update_particles_u(Func &&f)
    {
        update_particles_dispatch<false>(std::forward<Func>(f));
    }
    template <typename Func>
    void update_particles_o(Func &&f)
    {
        update_particles_dispatch<true>(std::forward<Func>(f));
    }
    template <typename Func>
    void update_masses_u(Func &&f)
    {
        update_masses_dispatch<false>(std::forward<Func>(f));
    }
    template <typename Func>
    void update_masses_o(Func &&f)
    {
        update_masses_dispatch<true>(std::forward<Func>(f));
    }

private:
    F m_box_size;
    bool m_box_size_deduced;
    size_type m_max_leaf_n;
    size_type m_ncrit;
    // Extension: build (and rebuild after updates) on the GPU instead of the host.
    bool m_device_build = false;
    // Particles in Morton order: x, y, z, masses.
    std::array<f_vector<F>, NDim + 1u> m_parts;
    std::vector<UInt> m_codes;
    // m_perm: original order -> Morton order; m_last_perm: order before the last update -> Morton order;
    // m_inv_perm: Morton order -> original order (tree.hpp:3853-3877).
    std::vector<size_type> m_perm, m_last_perm, m_inv_perm;
    tree_type m_tree;
    cnode_list_type m_crit_nodes;
    // Device replicas, one per GPU, created on demand. Declared last so that they are destroyed first
    // (tree.hpp:3882-3884).
    mutable std::mutex m_dev_mutex;
    mutable std::vector<detail::device_state> m_dev;
};

template <typename F, mac MAC = mac::bh>
using octree = tree<3, F, std::size_t, MAC>;
template <typename F, mac MAC = mac::bh>
using quadtree = tree<2, F, std::size_t, MAC>;

} // namespace rakau_amd

#if defined(RAKAU_AMD_DROP_IN)
namespace rakau = rakau_amd;
#endif

#endif
