// Interaction census of the rakau_amd engine (gfx950 / CDNA4, wave64); the scalar depth-first cross-check kernels that
// used to live here are in rk_kernels_xcheck.hip (librakau_amd_xcheck.so).
//
// The unit of work is a *target group* = one critical node of the tree, exactly the unit the
// reference's CPU engine hands to a TBB task (include/rakau/tree.hpp:2923-3008 of the reference):
// the multipole acceptance test is passed only if EVERY particle of the group passes it
// (tree.hpp:2662-2672), so a group shares one interaction list, and that list is identical to the
// CPU engine's.
#include "rk_common.hpp"
#include "rk_device.hpp"

namespace rk
{

// ------------------------------------------------------------------------------------------------
// Interaction census: the same walk as k_dfs_wave without the arithmetic. Produces the particle-level
// interaction counts of the traversal (targets x MAC evaluations, targets x accepted nodes, targets x
// particles of opened leaves, ordered pairs inside groups) -- the algorithmic work the roofline figures
// in bench.py are computed from. Integer atomics: deterministic.
// ------------------------------------------------------------------------------------------------
template <typename F, int MAC>
__global__ void __launch_bounds__(256) k_census(const kparams<F> P, uint32_t g_begin, uint32_t g_end,
                                                unsigned long long *__restrict__ counts,
                                                unsigned long long *__restrict__ per_group)
{
    using v4 = typename vt<F>::v4;
    using v2 = typename vt<F>::v2;
    const uint32_t wave = __builtin_amdgcn_readfirstlane((blockIdx.x * 256 + threadIdx.x) >> 6);
    if (g_begin + wave >= g_end) {
        return;
    }
    const int lane = threadIdx.x & 63;
    const uint4 c = P.crit[g_begin + wave];
    const uint32_t tb = c.x, te = c.y, cnode = c.z;
    const unsigned long long T = te - tb;
    const uint32_t n_nodes = P.n_nodes;
    const F mac_value = P.mac_value;
    unsigned long long n_mac = 0, n_com = 0, n_pp = 0;
    uint32_t idx = 0;
    while (idx < n_nodes) {
        const uint4 topo = P.node_topo[idx];
        const uint32_t nch = topo.x;
        if (idx <= cnode && cnode <= idx + nch) {
            idx += (idx == cnode) ? nch + 1u : 1u;
            continue;
        }
        const v4 com = P.node_com[idx];
        const v2 mp = P.node_mac[idx];
        const F mac_lh = mac_lhs<F>(MAC, mp, mac_value);
        bool fail = false;
        for (uint32_t i = tb + lane; i < te; i += 64u) {
            const v4 t = P.part4[i];
            const F dx = com.x - t.x, dy = com.y - t.y, dz = com.z - t.z;
            const F d2 = rk_fma(dz, dz, rk_fma(dy, dy, dx * dx));
            fail |= (mac_lh >= d2);
        }
        n_mac += T;
        if (__builtin_amdgcn_ballot_w64(fail) != 0) {
            if (nch == 0) {
                n_pp += T * (topo.z - topo.y);
            }
            idx += 1u;
        } else {
            n_com += T;
            idx += nch + 1u;
        }
    }
    if (lane == 0) {
        atomicAdd(&counts[0], n_mac);
        atomicAdd(&counts[1], n_com);
        atomicAdd(&counts[2], n_pp);
        atomicAdd(&counts[3], T * (T - 1));
        if (per_group) {
            per_group[wave] = n_com + n_pp + T * (T - 1);
        }
    }
}

template <typename F>
void launch_census(const rk_state &s, const kparams<F> &p, int64_t g_begin, int64_t g_end,
                   unsigned long long *d_counts, unsigned long long *d_per_group, hipStream_t stream)
{
    const int64_t n = g_end - g_begin;
    if (n <= 0) {
        return;
    }
    const auto grid = static_cast<unsigned>((n + 3) / 4);
    if (s.mac == RK_MAC_BH) {
        hipLaunchKernelGGL((k_census<F, 0>), dim3(grid), dim3(256), 0, stream, p, static_cast<uint32_t>(g_begin),
                           static_cast<uint32_t>(g_end), d_counts, d_per_group);
    } else {
        hipLaunchKernelGGL((k_census<F, 1>), dim3(grid), dim3(256), 0, stream, p, static_cast<uint32_t>(g_begin),
                           static_cast<uint32_t>(g_end), d_counts, d_per_group);
    }
    RK_HIP(hipGetLastError());
}
template void launch_census<float>(const rk_state &, const kparams<float> &, int64_t, int64_t, unsigned long long *,
                                   unsigned long long *, hipStream_t);
template void launch_census<double>(const rk_state &, const kparams<double> &, int64_t, int64_t, unsigned long long *,
                                    unsigned long long *, hipStream_t);

// Makes the runtime load this translation unit's code object now (rk_init) instead of at the first launch.
void touch_kernels()
{
    hipFuncAttributes attr{};
    RK_HIP(hipFuncGetAttributes(&attr, reinterpret_cast<const void *>(&k_census<float, 0>)));
}

} // namespace rk
