// Cross-check kernels of the rakau_amd engine, built into librakau_amd_xcheck.so (loaded on demand by
// rk_set_kernel_variant(state, 1 | 4) / RK_BIG_DFS=1; never by the default path): variant 1, the scalar depth-first walk
// whose summation order is the CPU engine's, with its block-per-node form for oversized critical nodes. The split
// traversal (variant 4, rk_kernels_split.hip) is the other half of that library. Both were measured and lost to the list
// kernels (DESIGN.md sections 3.3, 3.5); they stay as independently written control flows that the tests compare with.
//
// The unit of work is a *target group* = one critical node of the tree, exactly the unit the
// reference's CPU engine hands to a TBB task (include/rakau/tree.hpp:2923-3008 of the reference):
// the multipole acceptance test is passed only if EVERY particle of the group passes it
// (tree.hpp:2662-2672), so a group shares one interaction list, and that list is identical to the
// CPU engine's.
#include "rk_common.hpp"
#include "rk_device.hpp"
#include "rk_xcheck.hpp"

namespace rk
{

// ------------------------------------------------------------------------------------------------
// Variant 1: one wavefront per target group, scalar depth-first traversal.
//
// Lane l holds targets l, l+64, ... (R per lane). The walk over the node array is wave-uniform:
// node records are fetched with scalar loads, the per-lane MAC outcome is combined with a ballot
// and the skip-pointer step is taken in SALU. The distance computed for the MAC is reused for the
// interaction, like tree_acc_pot_mac_check() + tree_acc_pot_src_com() do on the CPU
// (tree.hpp:2597-2793, 2477-2590), and the summation order is the CPU engine's.
// ------------------------------------------------------------------------------------------------
template <typename F, int Q, int MAC, int R>
__global__ void __launch_bounds__(256) k_dfs_wave(const kparams<F> P, const uint32_t *__restrict__ list, int n_list)
{
    using v4 = typename vt<F>::v4;
    using v2 = typename vt<F>::v2;
    constexpr int NR = nres_of(Q);
    const int wave = __builtin_amdgcn_readfirstlane((blockIdx.x * 256 + threadIdx.x) >> 6);
    if (wave >= n_list) {
        return;
    }
    const int lane = threadIdx.x & 63;
    const uint32_t g = __builtin_amdgcn_readfirstlane(list[wave]);
    const uint4 c = P.crit[g];
    const uint32_t tb = c.x, te = c.y, cnode = c.z;

    v4 tp[R];
    bool valid[R];
    F acc[R][NR];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const uint32_t i = tb + lane + 64u * r;
        valid[r] = i < te;
        tp[r] = P.part4[valid[r] ? i : tb];
#pragma unroll
        for (int k = 0; k < NR; ++k) {
            acc[r][k] = F(0);
        }
    }

    const uint32_t n_nodes = P.n_nodes;
    const F mac_value = P.mac_value, eps2 = P.eps2;
    uint32_t idx = 0;
    while (idx < n_nodes) {
        const uint4 topo = P.node_topo[idx];
        const uint32_t nch = topo.x;
        // Ancestor-or-self test (tree.hpp:2828-2838) on the depth-first index interval of the subtree.
        if (idx <= cnode && cnode <= idx + nch) {
            idx += (idx == cnode) ? nch + 1u : 1u;
            continue;
        }
        const v4 com = P.node_com[idx];
        const v2 mp = P.node_mac[idx];
        const F mac_lh = mac_lhs<F>(MAC, mp, mac_value);
        F dx[R], dy[R], dz[R], d2[R];
        bool fail = false;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            dx[r] = com.x - tp[r].x;
            dy[r] = com.y - tp[r].y;
            dz[r] = com.z - tp[r].z;
            d2[r] = rk_fma(dz[r], dz[r], rk_fma(dy[r], dy[r], dx[r] * dx[r]));
            fail |= valid[r] && (mac_lh >= d2[r]);
        }
        if (__builtin_amdgcn_ballot_w64(fail) != 0) {
            if (nch == 0) {
                // Opened leaf: all its particles act on all targets (tree.hpp:2432-2470).
                for (uint32_t j = topo.y; j < topo.z; ++j) {
                    const v4 s = P.part4[j];
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const F ex = s.x - tp[r].x, ey = s.y - tp[r].y, ez = s.z - tp[r].z;
                        const F e2 = rk_fma(ez, ez, rk_fma(ey, ey, rk_fma(ex, ex, eps2)));
                        interact<F, Q>(acc[r], ex, ey, ez, e2, s.w, tp[r].w);
                    }
                }
            }
            idx += 1u;
        } else {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                interact<F, Q>(acc[r], dx[r], dy[r], dz[r], d2[r] + eps2, com.w, tp[r].w);
            }
            idx += nch + 1u;
        }
    }

    // Interactions inside the group (tree.hpp:2073-2321), every ordered pair i != j.
    for (uint32_t j = tb; j < te; ++j) {
        const v4 s = P.part4[j];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const bool self = (j == tb + lane + 64u * r);
            const F ex = s.x - tp[r].x, ey = s.y - tp[r].y, ez = s.z - tp[r].z;
            F e2 = rk_fma(ez, ez, rk_fma(ey, ey, rk_fma(ex, ex, eps2)));
            e2 = self ? F(1) : e2;
            interact<F, Q>(acc[r], ex, ey, ez, e2, self ? F(0) : s.w, tp[r].w);
        }
    }

    const F G = P.G;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        if (valid[r]) {
            const uint32_t o = out_index(P, tb + lane + 64u * r);
#pragma unroll
            for (int k = 0; k < NR; ++k) {
                P.out[k][o] = acc[r][k] * G;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Fallback for groups larger than 512 particles (only possible with ncrit > 512, max_leaf_n > 512
// or > 512 particles sharing a deepest-level cell): one 256-thread workgroup per group, targets
// strided over the threads, accumulators kept in the output arrays. Correct for any size, not tuned.
// ------------------------------------------------------------------------------------------------
template <typename F, int Q, int MAC>
__global__ void __launch_bounds__(256) k_dfs_block(const kparams<F> P, const uint32_t *__restrict__ list, int n_list)
{
    using v4 = typename vt<F>::v4;
    using v2 = typename vt<F>::v2;
    constexpr int NR = nres_of(Q);
    const uint32_t g = list[blockIdx.x];
    const uint4 c = P.crit[g];
    const uint32_t tb = c.x, te = c.y, cnode = c.z;
    const uint32_t tid = threadIdx.x;
    for (uint32_t i = tb + tid; i < te; i += 256u) {
        for (int k = 0; k < NR; ++k) {
            P.out[k][out_index(P, i)] = F(0);
        }
    }
    const uint32_t n_nodes = P.n_nodes;
    const F mac_value = P.mac_value, eps2 = P.eps2;
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
        int fail = 0;
        for (uint32_t i = tb + tid; i < te; i += 256u) {
            const v4 t = P.part4[i];
            const F dx = com.x - t.x, dy = com.y - t.y, dz = com.z - t.z;
            const F d2 = rk_fma(dz, dz, rk_fma(dy, dy, dx * dx));
            fail |= (mac_lh >= d2);
        }
        const int any_fail = __syncthreads_or(fail);
        if (any_fail) {
            if (nch == 0) {
                for (uint32_t i = tb + tid; i < te; i += 256u) {
                    const v4 t = P.part4[i];
                    F acc[NR];
                    for (int k = 0; k < NR; ++k) {
                        acc[k] = P.out[k][out_index(P, i)];
                    }
                    for (uint32_t j = topo.y; j < topo.z; ++j) {
                        const v4 s = P.part4[j];
                        const F ex = s.x - t.x, ey = s.y - t.y, ez = s.z - t.z;
                        const F e2 = rk_fma(ez, ez, rk_fma(ey, ey, rk_fma(ex, ex, eps2)));
                        interact<F, Q>(acc, ex, ey, ez, e2, s.w, t.w);
                    }
                    for (int k = 0; k < NR; ++k) {
                        P.out[k][out_index(P, i)] = acc[k];
                    }
                }
            }
            idx += 1u;
        } else {
            for (uint32_t i = tb + tid; i < te; i += 256u) {
                const v4 t = P.part4[i];
                F acc[NR];
                for (int k = 0; k < NR; ++k) {
                    acc[k] = P.out[k][out_index(P, i)];
                }
                const F dx = com.x - t.x, dy = com.y - t.y, dz = com.z - t.z;
                const F d2 = rk_fma(dz, dz, rk_fma(dy, dy, dx * dx));
                interact<F, Q>(acc, dx, dy, dz, d2 + eps2, com.w, t.w);
                for (int k = 0; k < NR; ++k) {
                    P.out[k][out_index(P, i)] = acc[k];
                }
            }
            idx += nch + 1u;
        }
    }
    const F G = P.G;
    for (uint32_t i = tb + tid; i < te; i += 256u) {
        const v4 t = P.part4[i];
        F acc[NR];
        for (int k = 0; k < NR; ++k) {
            acc[k] = P.out[k][out_index(P, i)];
        }
        for (uint32_t j = tb; j < te; ++j) {
            const v4 s = P.part4[j];
            const bool self = (j == i);
            const F ex = s.x - t.x, ey = s.y - t.y, ez = s.z - t.z;
            F e2 = rk_fma(ez, ez, rk_fma(ey, ey, rk_fma(ex, ex, eps2)));
            e2 = self ? F(1) : e2;
            interact<F, Q>(acc, ex, ey, ez, e2, self ? F(0) : s.w, t.w);
        }
        for (int k = 0; k < NR; ++k) {
            P.out[k][out_index(P, i)] = acc[k] * G;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Launch logic.
// ------------------------------------------------------------------------------------------------
template <typename F, int Q, int MAC>
static void launch_qm(const rk_state &s, const kparams<F> &p, const int64_t cb[n_classes], const int64_t ce[n_classes],
                      hipStream_t stream)
{
    const auto *lists = static_cast<const uint32_t *>(s.buf[RK_BUF_CLASS]);
    auto wave_launch = [&](auto Rtag, int c) {
        constexpr int R = decltype(Rtag)::value;
        const int64_t n = ce[c] - cb[c];
        if (n <= 0) {
            return;
        }
        const auto grid = static_cast<unsigned>((n + 3) / 4);
        hipLaunchKernelGGL((k_dfs_wave<F, Q, MAC, R>), dim3(grid), dim3(256), 0, stream, p, lists + s.class_off[c] + cb[c],
                           static_cast<int>(n));
    };
    wave_launch(std::integral_constant<int, 1>{}, 0);
    wave_launch(std::integral_constant<int, 2>{}, 1);
    wave_launch(std::integral_constant<int, 4>{}, 2);
    wave_launch(std::integral_constant<int, 8>{}, 3);
    {
        const int64_t n = ce[big_class] - cb[big_class];
        if (n > 0) {
            hipLaunchKernelGGL((k_dfs_block<F, Q, MAC>), dim3(static_cast<unsigned>(n)), dim3(256), 0, stream, p,
                               lists + s.class_off[big_class] + cb[big_class], static_cast<int>(n));
        }
    }
}

template <typename F>
void launch_traversal(const rk_state &s, int q, const kparams<F> &p, const int64_t cb[n_classes],
                      const int64_t ce[n_classes], hipStream_t stream)
{
    const int key = q * 2 + s.mac;
    switch (key) {
        case 0: launch_qm<F, 0, 0>(s, p, cb, ce, stream); break;
        case 1: launch_qm<F, 0, 1>(s, p, cb, ce, stream); break;
        case 2: launch_qm<F, 1, 0>(s, p, cb, ce, stream); break;
        case 3: launch_qm<F, 1, 1>(s, p, cb, ce, stream); break;
        case 4: launch_qm<F, 2, 0>(s, p, cb, ce, stream); break;
        case 5: launch_qm<F, 2, 1>(s, p, cb, ce, stream); break;
        default: throw error(RK_EINVAL, "invalid q / mac combination");
    }
    RK_HIP(hipGetLastError());
}

template <typename F>
void launch_block(const rk_state &s, int q, const kparams<F> &p, const uint32_t *list, int64_t n, hipStream_t stream)
{
    if (n <= 0) {
        return;
    }
    const dim3 grid(static_cast<unsigned>(n)), block(256);
    const int cnt = static_cast<int>(n);
    switch (q * 2 + s.mac) {
        case 0: hipLaunchKernelGGL((k_dfs_block<F, 0, 0>), grid, block, 0, stream, p, list, cnt); break;
        case 1: hipLaunchKernelGGL((k_dfs_block<F, 0, 1>), grid, block, 0, stream, p, list, cnt); break;
        case 2: hipLaunchKernelGGL((k_dfs_block<F, 1, 0>), grid, block, 0, stream, p, list, cnt); break;
        case 3: hipLaunchKernelGGL((k_dfs_block<F, 1, 1>), grid, block, 0, stream, p, list, cnt); break;
        case 4: hipLaunchKernelGGL((k_dfs_block<F, 2, 0>), grid, block, 0, stream, p, list, cnt); break;
        case 5: hipLaunchKernelGGL((k_dfs_block<F, 2, 1>), grid, block, 0, stream, p, list, cnt); break;
        default: throw error(RK_EINVAL, "invalid q / mac combination");
    }
    RK_HIP(hipGetLastError());
}
template void launch_block<float>(const rk_state &, int, const kparams<float> &, const uint32_t *, int64_t, hipStream_t);
template void launch_block<double>(const rk_state &, int, const kparams<double> &, const uint32_t *, int64_t, hipStream_t);

template void launch_traversal<float>(const rk_state &, int, const kparams<float> &, const int64_t[n_classes],
                                      const int64_t[n_classes], hipStream_t);
template void launch_traversal<double>(const rk_state &, int, const kparams<double> &, const int64_t[n_classes],
                                       const int64_t[n_classes], hipStream_t);

// ------------------------------------------------------------------------------------------------
// The table librakau_amd.so binds (rk_xcheck.hpp): every entry catches, returns a status code, keeps the message.
// ------------------------------------------------------------------------------------------------
namespace
{
thread_local std::string t_xerr;
template <typename Fn>
int xguard(Fn &&f) noexcept
{
    try {
        f();
        return RK_OK;
    } catch (const error &e) {
        t_xerr = e.what();
        return e.code;
    } catch (const std::exception &e) {
        t_xerr = e.what();
        return RK_ERUNTIME;
    }
}
const char *x_last_error()
{
    return t_xerr.c_str();
}
template <typename F>
int x_traversal(const rk_state *s, int q, const kparams<F> *p, const int64_t *cb, const int64_t *ce, hipStream_t st)
{
    return xguard([&] { launch_traversal<F>(*s, q, *p, cb, ce, st); });
}
template <typename F>
int x_block(const rk_state *s, int q, const kparams<F> *p, const uint32_t *list, int64_t n, hipStream_t st)
{
    return xguard([&] { launch_block<F>(*s, q, *p, list, n, st); });
}
template <typename F>
int x_lists(const rk_state *s, const kparams<F> *p, int64_t g0, int64_t g1, hipStream_t st)
{
    return xguard([&] { launch_lists<F>(*s, *p, g0, g1, st); });
}
template <typename F>
int x_dense(const rk_state *s, int q, const kparams<F> *p, const int64_t *cb, const int64_t *ce, const hipStream_t *streams,
            unsigned class_mask, int what)
{
    return xguard([&] { launch_dense<F>(*s, q, *p, cb, ce, streams, class_mask, what); });
}
int x_touch()
{
    return xguard([] {
        hipFuncAttributes attr{};
        RK_HIP(hipFuncGetAttributes(&attr, reinterpret_cast<const void *>(&k_dfs_wave<float, 0, 0, 1>)));
        touch_split();
    });
}
const xcheck_vtable g_vtable = {xcheck_abi_tag(), x_last_error,          x_traversal<float>, x_traversal<double>,
                                x_block<float>,   x_block<double>,       x_lists<float>,     x_lists<double>,
                                x_dense<float>,   x_dense<double>,       x_touch};
} // namespace

} // namespace rk

extern "C" RK_EXPORT const rk::xcheck_vtable *rk_xcheck_entry(void)
{
    return &rk::g_vtable;
}
