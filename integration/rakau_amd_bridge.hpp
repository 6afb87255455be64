// rakau_amd <-> rakau bridge: the one declaration the reference's tree.hpp needs besides its own rocm_fwd.hpp.
//
// rocm_state's constructor (include/rakau/detail/rocm_fwd.hpp:29-30 of the reference) receives the particles, the codes
// and the node array, but not tree::m_ncrit. The rakau_amd engine works on the tree's critical nodes (the CPU engine's
// unit of work, tree.hpp:794-807), which it re-derives from the node array and ncrit. The tree passes the value with one
// call right before it builds the state (see the diff in INTEGRATION.md section B); without the call the library
// default of rakau (tree.hpp:584-595) is used.
#ifndef RAKAU_AMD_BRIDGE_HPP
#define RAKAU_AMD_BRIDGE_HPP

#include <cstddef>

namespace rakau
{
inline namespace detail
{

// ncrit of the tree whose rocm_state is constructed next ON THIS THREAD (consumed by that constructor).
void rakau_amd_set_ncrit(std::size_t ncrit) __attribute__((visibility("default")));

} // namespace detail
} // namespace rakau

#endif
