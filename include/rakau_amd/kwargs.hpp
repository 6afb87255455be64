// Minimal compile-time named arguments for the rakau_amd C++17 front door.
//
// Provides the spelling the reference's API uses (`kwargs::x_coords = ptr`, `kwargs::G = 2.`, ...;
// include/rakau/tree.hpp:599-626 of the reference) with a far smaller mechanism than the reference's
// igor library: a named argument is a (tag, reference) pair, and a parser looks tags up in a pack.
#ifndef RAKAU_AMD_KWARGS_HPP
#define RAKAU_AMD_KWARGS_HPP

#include <cstddef>
#include <tuple>
#include <type_traits>
#include <utility>

namespace rakau_amd
{
namespace kw_detail
{

template <typename Tag, typename T>
struct bound_arg {
    using tag_type = Tag;
    using value_type = T; // T is an lvalue or rvalue reference type
    T value;
};

template <typename T>
struct is_bound_arg : std::false_type {
};
template <typename Tag, typename T>
struct is_bound_arg<bound_arg<Tag, T>> : std::true_type {
};

template <typename Tag>
struct name {
    using tag_type = Tag;
    template <typename T>
    constexpr bound_arg<Tag, T &&> operator=(T &&x) const
    {
        return bound_arg<Tag, T &&>{std::forward<T>(x)};
    }
    // Brace initialisation, e.g. split = {1., 1.}.
    template <typename T>
    constexpr bound_arg<Tag, std::initializer_list<T> &&> operator=(std::initializer_list<T> &&l) const
    {
        return bound_arg<Tag, std::initializer_list<T> &&>{std::move(l)};
    }
};

template <typename... Args>
class parser
{
    std::tuple<Args &&...> m_args;

    template <typename Tag, std::size_t I = 0>
    static constexpr std::size_t index_of()
    {
        if constexpr (I == sizeof...(Args)) {
            return I;
        } else {
            using arg_t = std::remove_cv_t<std::remove_reference_t<std::tuple_element_t<I, std::tuple<Args...>>>>;
            if constexpr (is_bound_arg<arg_t>::value) {
                if constexpr (std::is_same_v<typename arg_t::tag_type, Tag>) {
                    return I;
                } else {
                    return index_of<Tag, I + 1>();
                }
            } else {
                return index_of<Tag, I + 1>();
            }
        }
    }
    template <typename Tag, std::size_t I = 0>
    static constexpr std::size_t count_of()
    {
        if constexpr (I == sizeof...(Args)) {
            return 0;
        } else {
            using arg_t = std::remove_cv_t<std::remove_reference_t<std::tuple_element_t<I, std::tuple<Args...>>>>;
            if constexpr (is_bound_arg<arg_t>::value) {
                return (std::is_same_v<typename arg_t::tag_type, Tag> ? 1u : 0u) + count_of<Tag, I + 1>();
            } else {
                return count_of<Tag, I + 1>();
            }
        }
    }

public:
    explicit constexpr parser(Args &&... args) : m_args(std::forward<Args>(args)...) {}

    template <typename Tag>
    static constexpr bool has(const name<Tag> &)
    {
        return index_of<Tag>() < sizeof...(Args);
    }
    template <typename... Tags>
    static constexpr bool has_all(const name<Tags> &...)
    {
        return ((index_of<Tags>() < sizeof...(Args)) && ...);
    }
    static constexpr bool has_unnamed_arguments()
    {
        return !(is_bound_arg<std::remove_cv_t<std::remove_reference_t<Args>>>::value && ...);
    }
    template <typename Tag>
    static constexpr bool duplicated(const name<Tag> &)
    {
        return count_of<Tag>() > 1u;
    }
    // Fetch the value bound to a name (preserving its value category).
    template <typename Tag>
    constexpr decltype(auto) operator()(const name<Tag> &) const
    {
        static_assert(index_of<Tag>() < sizeof...(Args), "named argument not present");
        using arg_t = std::remove_reference_t<std::tuple_element_t<index_of<Tag>(), std::tuple<Args...>>>;
        return static_cast<typename arg_t::value_type>(std::get<index_of<Tag>()>(m_args).value);
    }
};

template <typename... Args>
parser(Args &&...)->parser<Args...>;

} // namespace kw_detail

namespace kwargs
{

struct box_size_tag {
};
struct max_leaf_n_tag {
};
struct ncrit_tag {
};
template <std::size_t>
struct coords_tag {
};
struct masses_tag {
};
struct nparts_tag {
};
struct G_tag {
};
struct eps_tag {
};
struct split_tag {
};
struct device_build_tag {
};

// Tree construction.
inline constexpr kw_detail::name<box_size_tag> box_size{};
inline constexpr kw_detail::name<max_leaf_n_tag> max_leaf_n{};
inline constexpr kw_detail::name<ncrit_tag> ncrit{};
template <std::size_t N>
inline constexpr kw_detail::name<coords_tag<N>> coords{};
inline constexpr auto x_coords = coords<0>;
inline constexpr auto y_coords = coords<1>;
inline constexpr auto z_coords = coords<2>;
inline constexpr kw_detail::name<masses_tag> masses{};
inline constexpr kw_detail::name<nparts_tag> nparts{};
// Acceleration / potential computation.
inline constexpr kw_detail::name<G_tag> G{};
inline constexpr kw_detail::name<eps_tag> eps{};
inline constexpr kw_detail::name<split_tag> split{};
// Extension (not in rakau): device_build = true runs the tree construction itself on the GPU
// (rk_state_build) and only downloads the arrays the host-side accessors need.
inline constexpr kw_detail::name<device_build_tag> device_build{};

} // namespace kwargs
} // namespace rakau_amd

#endif
