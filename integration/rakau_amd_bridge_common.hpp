// Shared by the two bridge translation units (rakau_amd_bridge.cpp: the ROCm seam, rakau_amd_cuda_bridge.cpp: the CUDA
// seam): status code -> exception, the build's default ncrit, node records widened to 64-bit code fields.
#ifndef RAKAU_AMD_BRIDGE_COMMON_HPP
#define RAKAU_AMD_BRIDGE_COMMON_HPP

#include <cstddef>
#include <cstdint>
#include <new>
#include <stdexcept>
#include <type_traits>
#include <vector>

#include <rakau/detail/tree_fwd.hpp>

#include <rakau_amd.h>

namespace rakau_amd_bridge
{

// rakau's default_ncrit (tree.hpp:584-595 of the reference): 256 when the library is built for AVX-512, 128 otherwise.
// The reference tests xsimd's macros; a translation unit that has not seen xsimd (the bridges normally have not) gets the
// same answer from the compiler's own macro, because xsimd derives XSIMD_X86_INSTR_SET >= XSIMD_X86_AVX512_VERSION from
// __AVX512F__ and the bridges are compiled with the flags of the rest of the reference's library.
constexpr std::size_t default_ncrit_of_this_build =
#if defined(XSIMD_X86_INSTR_SET) && defined(XSIMD_X86_AVX512_VERSION)
#if XSIMD_X86_INSTR_SET >= XSIMD_X86_AVX512_VERSION
    256
#else
    128
#endif
#elif defined(__AVX512F__)
    256
#else
    128
#endif
    ;

// Status code -> the exception type the reference throws for that class of error ("Errors").
inline void rk_throw(int rc)
{
    switch (rc) {
        case RK_OK:
            return;
        case RK_EINVAL:
            throw std::invalid_argument(rk_last_error());
        case RK_EDOMAIN:
            throw std::domain_error(rk_last_error());
        case RK_EOVERFLOW:
            throw std::overflow_error(rk_last_error());
        case RK_ENOMEM:
            throw std::bad_alloc();
        default:
            throw std::runtime_error(rk_last_error());
    }
}

// The C ABI takes the node records with 64-bit code / level fields (tree_node_t<NDim, F, std::uint64_t, MAC>); trees with
// 32-bit codes hand over a widened copy (values unchanged). Of the NODE codes the engine reads the low NDim bits -- the
// octant inside the parent, which orders the sibling records (include/rakau_amd.h, rk_state_create) -- and nothing else;
// the sorted PARTICLE codes (m_codes) are not needed with critical-node grouping.
template <std::size_t NDim, typename F, typename UInt, rakau::mac MAC>
struct wide_nodes {
    using wide_node = rakau::tree_node_t<NDim, F, std::uint64_t, MAC>;
    std::vector<wide_node> widened;
    const void *data = nullptr;
    static constexpr std::int64_t stride = static_cast<std::int64_t>(sizeof(wide_node));
    wide_nodes(const rakau::tree_node_t<NDim, F, UInt, MAC> *tree, std::size_t tree_size) : data(tree)
    {
        if constexpr (!std::is_same_v<UInt, std::uint64_t>) {
            widened.resize(tree_size);
            for (std::size_t i = 0; i < tree_size; ++i) {
                const auto &n = tree[i];
                auto &w = widened[i];
                w.begin = n.begin, w.end = n.end, w.n_children = n.n_children, w.code = n.code, w.level = n.level;
                for (std::size_t j = 0; j < NDim + 1u; ++j) {
                    w.props[j] = n.props[j];
                }
                if constexpr (MAC == rakau::mac::bh) {
                    w.dim2 = n.dim2;
                } else {
                    w.dim = n.dim, w.delta = n.delta;
                }
            }
            data = widened.data();
        }
    }
};

} // namespace rakau_amd_bridge

#endif
