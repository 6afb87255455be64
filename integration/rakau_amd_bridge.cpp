// rakau_amd <-> rakau bridge: the definitions behind the reference's accelerator seam
// (include/rakau/detail/rocm_fwd.hpp:22-46), implemented on the rakau_amd C ABI (include/rakau_amd.h).
// This file REPLACES src/rakau_rocm.cpp of the reference in a RAKAU_WITH_ROCM build: same symbols, same explicit
// instantiations (src/rakau_rocm.cpp:333-370), no HCC. Build (the reference's headers are not part of this repository):
//
//   g++ -std=c++17 -O2 -fPIC -shared -I<rakau>/include -I<rakau_amd>/include integration/rakau_amd_bridge.cpp \
//       -L<rakau_amd>/rakau_amd/lib -lrakau_amd -o librakau_rocm.so
//
// tests/test_integration_bridge.py compiles it exactly like this whenever a checkout of the reference is present.
#include <array>
#include <cstddef>
#include <cstdint>
#include <new>
#include <stdexcept>
#include <type_traits>
#include <vector>

#include <rakau/detail/rocm_fwd.hpp>

#include "rakau_amd_bridge.hpp"
#include "rakau_amd_bridge_common.hpp"
#include <rakau_amd.h>

namespace rakau
{
inline namespace detail
{

namespace
{

using rakau_amd_bridge::default_ncrit_of_this_build;
using rakau_amd_bridge::rk_throw;
// Value announced by rakau_amd_set_ncrit() for the NEXT rocm_state constructed on this thread (0: none announced).
thread_local std::size_t t_next_ncrit = 0;

} // namespace

void rakau_amd_set_ncrit(std::size_t ncrit)
{
    t_next_ncrit = ncrit;
}

// rocm_fwd.hpp:22.
unsigned rocm_min_size()
{
    return rk_min_size();
}

// rocm_fwd.hpp:24.
bool rocm_has_accelerator()
{
    return rk_has_accelerator() != 0;
}

// rocm_fwd.hpp:29-30 (called from tree::rocm_init_state(), tree.hpp:1495-1508).
template <std::size_t NDim, typename F, typename UInt, mac MAC>
rocm_state<NDim, F, UInt, MAC>::rocm_state(const std::array<const F *, NDim + 1u> &parts, const UInt *codes, int nparts,
                                           const tree_node_t<NDim, F, UInt, MAC> *tree, int tree_size)
    : m_state(nullptr)
{
    static_assert(NDim == 2u || NDim == 3u, "the accelerator seam serves quadtrees and octrees");
    static_assert(std::is_same_v<F, float> || std::is_same_v<F, double>);
    const void *p[4] = {};
    for (std::size_t j = 0; j < NDim + 1u; ++j) {
        p[j] = parts[j];
    }
    // Trees with 32-bit codes hand over widened node records (rakau_amd_bridge_common.hpp).
    const rakau_amd_bridge::wide_nodes<NDim, F, UInt, MAC> wn(tree, static_cast<std::size_t>(tree_size));
    (void)codes;
    // The announcement is consumed by the constructor it was made for: a later tree on this thread that announces
    // nothing gets the build's default again, not the previous tree's value.
    const std::size_t ncrit = t_next_ncrit ? t_next_ncrit : default_ncrit_of_this_build;
    t_next_ncrit = 0;
    rk_state *s = nullptr;
    rk_throw(rk_state_create_nd(&s, static_cast<int>(NDim), std::is_same_v<F, float> ? RK_F32 : RK_F64,
                                MAC == mac::bh ? RK_MAC_BH : RK_MAC_BH_GEOM, /* device */ 0, p, nullptr, nparts, wn.data,
                                tree_size, wn.stride, ncrit));
    m_state = s;
}

// rocm_fwd.hpp:38 (tree::rocm_reset_state(), tree.hpp:1511-1519).
template <std::size_t NDim, typename F, typename UInt, mac MAC>
rocm_state<NDim, F, UInt, MAC>::~rocm_state()
{
    rk_state_destroy(static_cast<rk_state *>(m_state));
}

// rocm_fwd.hpp:41-42 (called from tree::acc_pot_impl(), tree.hpp:3078-3094). mac_value, G, eps2 and offset_output keep
// the meaning they have in src/rakau_rocm.cpp:100-116; [p_begin, p_end) starts on a critical-node boundary
// (tree.hpp:3053-3063) and ends at nparts.
template <std::size_t NDim, typename F, typename UInt, mac MAC>
template <unsigned Q>
void rocm_state<NDim, F, UInt, MAC>::acc_pot(int p_begin, int p_end, const std::array<F *, tree_nvecs_res<Q, NDim>> &out,
                                             F mac_value, F G, F eps2, bool offset_output) const
{
    void *o[4] = {};
    for (std::size_t j = 0; j < out.size(); ++j) {
        o[j] = out[j];
    }
    rk_throw(rk_acc_pot(static_cast<rk_state *>(m_state), static_cast<int>(Q), p_begin, p_end, o,
                        static_cast<double>(mac_value), static_cast<double>(G), static_cast<double>(eps2),
                        offset_output ? RK_OUT_OFFSET : RK_OUT_COMPACT));
}

// The instantiation list of src/rakau_rocm.cpp:333-370: NDim in {2, 3} x F in {float, double} x UInt in {32, 64 bit} x
// MAC in {bh, bh_geom}, acc_pot for Q in {0, 1, 2}.
#define RAKAU_AMD_BRIDGE_INST(ND, F, U, M)                                                                             \
    template class rocm_state<ND, F, U, M>;                                                                            \
    template void rocm_state<ND, F, U, M>::acc_pot<0>(int, int, const std::array<F *, tree_nvecs_res<0, ND>> &, F, F, F, \
                                                      bool) const;                                                     \
    template void rocm_state<ND, F, U, M>::acc_pot<1>(int, int, const std::array<F *, tree_nvecs_res<1, ND>> &, F, F, F, \
                                                      bool) const;                                                     \
    template void rocm_state<ND, F, U, M>::acc_pot<2>(int, int, const std::array<F *, tree_nvecs_res<2, ND>> &, F, F, F, \
                                                      bool) const;
#define RAKAU_AMD_BRIDGE_INST_MACS(ND, F, U) RAKAU_AMD_BRIDGE_INST(ND, F, U, mac::bh) RAKAU_AMD_BRIDGE_INST(ND, F, U, mac::bh_geom)
#define RAKAU_AMD_BRIDGE_INST_UINTS(ND, F)                                                                             \
    RAKAU_AMD_BRIDGE_INST_MACS(ND, F, std::uint32_t) RAKAU_AMD_BRIDGE_INST_MACS(ND, F, std::uint64_t)
RAKAU_AMD_BRIDGE_INST_UINTS(2, float)
RAKAU_AMD_BRIDGE_INST_UINTS(2, double)
RAKAU_AMD_BRIDGE_INST_UINTS(3, float)
RAKAU_AMD_BRIDGE_INST_UINTS(3, double)

} // namespace detail
} // namespace rakau
