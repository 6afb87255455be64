// rakau_amd <-> rakau bridge, CUDA seam: the two declarations a RAKAU_WITH_CUDA build of the reference's tree.hpp needs
// besides its own cuda_fwd.hpp.
//
// cuda_acc_pot_impl (include/rakau/detail/cuda_fwd.hpp:25-29 of the reference) is STATELESS: tree, particles and codes
// arrive on every call and the reference uploads them to every device on every call (src/rakau_cuda.cu:446-481). The
// rakau_amd engine keeps a device-resident state per tree instead. Two facts the seam does not carry make that safe:
// tree::m_ncrit (the engine works on the tree's critical nodes) and the life time of the tree's arrays. A patched
// tree.hpp announces both through the hooks it already has (rocm_init_state / rocm_reset_state, tree.hpp:1495-1519,
// called at every construction / mutation / destruction site whatever the accelerator): see INTEGRATION.md section B.
// A tree that never announces itself is served correctly too -- with the build's default ncrit, and with a state that is
// built for the call and destroyed after it.
#ifndef RAKAU_AMD_CUDA_BRIDGE_HPP
#define RAKAU_AMD_CUDA_BRIDGE_HPP

#include <cstddef>

namespace rakau
{
inline namespace detail
{

// The node array at `tree` (m_tree.data()) is complete and will not change until rakau_amd_invalidate(tree): device
// replicas made for it by cuda_acc_pot_impl() are kept between calls. ncrit = tree::m_ncrit.
void rakau_amd_tree_ready(const void *tree, std::size_t ncrit) __attribute__((visibility("default")));
// The arrays behind `tree` are about to change or die: drop the replicas (no-op for an unknown pointer; nullptr drops
// every tree's).
void rakau_amd_invalidate(const void *tree) __attribute__((visibility("default")));

} // namespace detail
} // namespace rakau

#endif
