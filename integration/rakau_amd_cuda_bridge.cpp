// rakau_amd <-> rakau bridge, CUDA seam: the definitions behind include/rakau/detail/cuda_fwd.hpp:23-30 of the reference
// -- cuda_min_size(), cuda_device_count(), cuda_acc_pot_impl<Q, NDim, F, UInt, MAC>() -- on the rakau_amd C ABI
// (include/rakau_amd.h). This file REPLACES src/rakau_cuda.cu in a RAKAU_WITH_CUDA build: same symbols, same explicit
// instantiations (src/rakau_cuda.cu:536-568), no nvcc. It is how the reference's own multi-GPU dispatch
// (tree.hpp:3131-3257) reaches this engine. Build (the reference's headers are not part of this repository):
//
//   g++ -std=c++17 -O2 -fPIC -shared -pthread -I<rakau>/include -I<rakau_amd>/include \
//       integration/rakau_amd_cuda_bridge.cpp -L<rakau_amd>/rakau_amd/lib -lrakau_amd -o librakau_cuda.so
//
// tests/test_integration_bridge.py compiles it exactly like this whenever a checkout of the reference is present.
//
// What a call does (the reference, src/rakau_cuda.cu:348-528: pin the outputs, upload tree + particles + codes into
// managed memory, advise every device, one stream and one kernel per device, copy the results back):
//   1. the state of the tree: looked up by the node array's address if the tree announced itself
//      (rakau_amd_tree_ready), otherwise built for this call from the arguments;
//   2. replicas on the further devices that have a share: ONE rk_state_clone_all() (asynchronous xGMI peer copies
//      fanning out as a doubling tree), kept with the state;
//   3. every interior cut of split_indices snapped forward to a critical-node boundary (the reference snaps only the
//      first, tree.hpp:3168-3185; this engine's unit of work is the critical node);
//   4. one host thread per device with a share, rk_acc_pot() on its replica for its Morton range; all joined, the first
//      exception re-thrown.
#include <algorithm>
#include <array>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstddef>
#include <cstdint>
#include <exception>
#include <future>
#include <map>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <vector>

#include <rakau/detail/cuda_fwd.hpp>

#include "rakau_amd_bridge_common.hpp"
#include "rakau_amd_cuda_bridge.hpp"
#include <rakau_amd.h>

namespace rakau
{
inline namespace detail
{

namespace
{

using rakau_amd_bridge::default_ncrit_of_this_build;
using rakau_amd_bridge::rk_throw;

// Everything resident for one tree: the state on each logical device (null where none was needed yet) and the first
// particle of every critical node (where cuts may fall).
struct resident {
    // What the state was built from; a mismatch (a caller that announced a tree and changed it without invalidating)
    // rebuilds instead of answering for the wrong tree.
    const void *tree = nullptr;
    std::uint64_t tree_size = 0, nparts = 0, ncrit = 0;
    const void *parts[4] = {};
    int ndim = 0, fp = 0, mac = 0, code_bits = 0;
    std::vector<rk_state *> dev;
    std::vector<std::uint64_t> crit_begin;
    std::mutex mtx; // one call at a time per tree (the seam's contract); guards dev
    ~resident()
    {
        for (auto *s : dev) {
            rk_state_destroy(s);
        }
    }
    resident() = default;
    resident(const resident &) = delete;
    resident &operator=(const resident &) = delete;
};

struct announced {
    std::size_t ncrit = 0;
    std::shared_ptr<resident> res; // empty until the first call
};

std::mutex g_mtx;
// Keyed by m_tree.data(). Never destroyed: a tree that is still announced when the process exits (a static tree) must not
// have its device states torn down from a static destructor, after the HIP runtime may have gone.
std::map<const void *, announced> &registry()
{
    static auto *r = new std::map<const void *, announced>;
    return *r;
}

template <std::size_t NDim, typename F, typename UInt, mac MAC>
void build_first_state(resident &r, int device, const tree_node_t<NDim, F, UInt, MAC> *tree, std::uint64_t tree_size,
                       const std::array<const F *, NDim + 1u> &parts, std::uint64_t nparts)
{
    const void *p[4] = {};
    for (std::size_t j = 0; j < NDim + 1u; ++j) {
        p[j] = parts[j];
    }
    const rakau_amd_bridge::wide_nodes<NDim, F, UInt, MAC> wn(tree, static_cast<std::size_t>(tree_size));
    rk_state *s = nullptr;
    rk_throw(rk_state_create_nd(&s, r.ndim, r.fp, r.mac, device, p, nullptr, static_cast<std::int64_t>(nparts), wn.data,
                                static_cast<std::int64_t>(tree_size), wn.stride, r.ncrit));
    r.dev[static_cast<std::size_t>(device)] = s;
}

// The first particle of every critical node: needed only when there are interior cuts to snap (more than one device).
void fetch_crit_begins(resident &r, rk_state *s)
{
    if (!r.crit_begin.empty()) {
        return;
    }
    std::int64_t info[8];
    rk_throw(rk_state_info(s, info));
    std::vector<std::int64_t> be(static_cast<std::size_t>(info[2]) * 2u);
    rk_throw(rk_state_crit_ranges(s, be.data()));
    r.crit_begin.resize(static_cast<std::size_t>(info[2]));
    for (std::size_t i = 0; i < r.crit_begin.size(); ++i) {
        r.crit_begin[i] = static_cast<std::uint64_t>(be[2u * i]);
    }
}

} // namespace

void rakau_amd_tree_ready(const void *tree, std::size_t ncrit)
{
    if (!tree) {
        return; // an empty tree never reaches the seam (every share is below cuda_min_size())
    }
    std::lock_guard<std::mutex> lk(g_mtx);
    auto &a = registry()[tree];
    a.ncrit = ncrit;
    a.res.reset(); // whatever lived at this address before is gone
}

void rakau_amd_invalidate(const void *tree)
{
    std::lock_guard<std::mutex> lk(g_mtx);
    if (!tree) {
        registry().clear();
    } else {
        registry().erase(tree);
    }
}

// cuda_fwd.hpp:23.
unsigned cuda_min_size()
{
    return rk_min_size();
}

// cuda_fwd.hpp:24.
unsigned cuda_device_count()
{
    const int n = rk_device_count();
    return n > 0 ? static_cast<unsigned>(n) : 0u;
}

// cuda_fwd.hpp:26-29 (called from tree::acc_pot_impl(), tree.hpp:3207-3222). Arguments as in src/rakau_cuda.cu:348-353:
// device i computes the particles [split_indices[i], split_indices[i + 1]); split_indices[0] lies on a critical-node
// boundary (tree.hpp:3168-3185); out[j] addresses element 0 of the full array (offset_output) or element
// split_indices[0] (compact), rakau_cuda.cu:524.
template <unsigned Q, std::size_t NDim, typename F, typename UInt, mac MAC>
void cuda_acc_pot_impl(const std::array<F *, tree_nvecs_res<Q, NDim>> &out,
                       const std::vector<tree_size_t<F>> &split_indices, const tree_node_t<NDim, F, UInt, MAC> *tree,
                       tree_size_t<F> tree_size, const std::array<const F *, NDim + 1u> &parts, const UInt *codes,
                       tree_size_t<F> nparts, F mac_value, F G, F eps2, bool offset_output)
{
    static_assert(NDim == 2u || NDim == 3u, "the accelerator seam serves quadtrees and octrees");
    static_assert(std::is_same_v<F, float> || std::is_same_v<F, double>);
    (void)codes;
    if (split_indices.size() < 2u) {
        return;
    }
    const std::size_t ngpus = split_indices.size() - 1u;
    if (ngpus > cuda_device_count()) {
        // tree.hpp:3135-3141 checks this before the call; a direct caller gets the same exception.
        throw std::invalid_argument("Cannot split the computation of accelerations/potentials: the split vector refers to "
                                    + std::to_string(ngpus) + " accelerators, but only "
                                    + std::to_string(cuda_device_count()) + " were detected");
    }

    // RK_BRIDGE_TIMING=1: phase times of the call on stderr (diagnostic).
    static const bool timing = std::getenv("RK_BRIDGE_TIMING") != nullptr;
    auto t_prev = std::chrono::steady_clock::now();
    const auto lap = [&](const char *what) {
        if (timing) {
            const auto t = std::chrono::steady_clock::now();
            std::fprintf(stderr, "RK_BRIDGE_TIMING %-28s %9.1f us\n", what, std::chrono::duration<double, std::micro>(t - t_prev).count());
            t_prev = t;
        }
    };
    // 1. The tree's resident data.
    std::shared_ptr<resident> res;
    bool keep = false;
    std::size_t ncrit = default_ncrit_of_this_build;
    {
        std::lock_guard<std::mutex> lk(g_mtx);
        const auto it = registry().find(tree);
        if (it != registry().end()) {
            keep = true;
            ncrit = it->second.ncrit;
            res = it->second.res;
        }
    }
    const int fp = std::is_same_v<F, float> ? RK_F32 : RK_F64, mc = MAC == mac::bh ? RK_MAC_BH : RK_MAC_BH_GEOM;
    auto matches = [&](const resident &r) {
        bool ok = r.tree == tree && r.tree_size == tree_size && r.nparts == nparts && r.ncrit == ncrit
                  && r.ndim == static_cast<int>(NDim) && r.fp == fp && r.mac == mc
                  && r.code_bits == static_cast<int>(sizeof(UInt) * 8u);
        for (std::size_t j = 0; j < NDim + 1u; ++j) {
            ok = ok && r.parts[j] == parts[j];
        }
        return ok;
    };
    if (!res || !matches(*res)) {
        res = std::make_shared<resident>();
        res->tree = tree, res->tree_size = tree_size, res->nparts = nparts, res->ncrit = ncrit;
        res->ndim = static_cast<int>(NDim), res->fp = fp, res->mac = mc, res->code_bits = static_cast<int>(sizeof(UInt) * 8u);
        for (std::size_t j = 0; j < NDim + 1u; ++j) {
            res->parts[j] = parts[j];
        }
        if (keep) {
            std::lock_guard<std::mutex> lk(g_mtx);
            const auto it = registry().find(tree);
            if (it != registry().end()) {
                it->second.res = res;
            }
        }
    }
    // From here on `res` keeps the states alive even if another thread invalidates the tree meanwhile (which the
    // seam's contract forbids anyway); an unannounced tree's states die with `res` at the end of this call.
    std::lock_guard<std::mutex> call_lock(res->mtx);

    // 3. Cuts: split_indices with every interior index moved forward to the next critical-node boundary.
    std::vector<std::uint64_t> cuts(split_indices.begin(), split_indices.end());
    if (cuts.back() != nparts || !std::is_sorted(cuts.begin(), cuts.end())) {
        throw std::invalid_argument("rakau_amd: split_indices must be sorted and end at the number of particles");
    }
    int first_dev = -1;
    for (std::size_t i = 0; i < ngpus; ++i) {
        if (cuts[i + 1u] > cuts[i]) {
            first_dev = static_cast<int>(i);
            break;
        }
    }
    if (first_dev < 0) {
        return; // nothing for the devices
    }
    if (res->dev.size() < ngpus) {
        res->dev.resize(ngpus, nullptr);
    }
    // The first state is made on the first device that has a share before snapping (should snapping empty that share,
    // the state still serves as the source of the replicas: where it lives is not worth a second upload).
    int src_dev = -1;
    for (std::size_t d = 0; d < res->dev.size(); ++d) {
        if (res->dev[d]) {
            src_dev = static_cast<int>(d);
            break;
        }
    }
    if (src_dev < 0) {
        src_dev = first_dev;
        build_first_state<NDim, F, UInt, MAC>(*res, src_dev, tree, tree_size, parts, nparts);
        lap("first state");
    }
    if (ngpus > 1u) {
        fetch_crit_begins(*res, res->dev[static_cast<std::size_t>(src_dev)]);
    }
    const auto &cb = res->crit_begin;
    for (std::size_t i = 0; i < ngpus; ++i) {
        // cuts[0] is on a boundary already (the engine refuses the call otherwise, naming ncrit); snapping it as well
        // would silently change which particles the CPU share is expected to have covered.
        if (i == 0u) {
            continue;
        }
        const auto it = std::lower_bound(cb.begin(), cb.end(), cuts[i]);
        cuts[i] = it == cb.end() ? static_cast<std::uint64_t>(nparts) : *it;
    }
    for (std::size_t i = 1; i < cuts.size(); ++i) {
        cuts[i] = std::max(cuts[i], cuts[i - 1u]);
    }

    // 2. Replicas for the devices that have a share and no state yet, all at once.
    std::vector<int> need;
    for (std::size_t d = 0; d < ngpus; ++d) {
        if (cuts[d + 1u] > cuts[d] && !res->dev[d]) {
            need.push_back(static_cast<int>(d));
        }
    }
    if (!need.empty()) {
        std::vector<rk_state *> made(need.size(), nullptr);
        rk_throw(rk_state_clone_all(made.data(), res->dev[static_cast<std::size_t>(src_dev)], need.data(),
                                    static_cast<int>(need.size())));
        for (std::size_t i = 0; i < need.size(); ++i) {
            res->dev[static_cast<std::size_t>(need[i])] = made[i];
        }
    }

    // 4. One host thread per device with a share (the calling thread takes the first of them).
    const std::uint64_t base = offset_output ? 0u : static_cast<std::uint64_t>(split_indices[0]);
    auto run_one = [&](std::size_t d) {
        void *o[4] = {};
        // Compact addressing relative to the device's own first particle: out[j] + (begin - base), no pointer ever
        // leaves the caller's array.
        for (std::size_t j = 0; j < out.size(); ++j) {
            o[j] = out[j] + (cuts[d] - base);
        }
        rk_throw(rk_acc_pot(res->dev[d], static_cast<int>(Q), static_cast<std::int64_t>(cuts[d]),
                            static_cast<std::int64_t>(cuts[d + 1u]), o, static_cast<double>(mac_value),
                            static_cast<double>(G), static_cast<double>(eps2), RK_OUT_COMPACT));
    };
    std::vector<std::size_t> busy;
    for (std::size_t d = 0; d < ngpus; ++d) {
        if (cuts[d + 1u] > cuts[d]) {
            busy.push_back(d);
        }
    }
    std::vector<std::future<void>> futs;
    for (std::size_t k = 1; k < busy.size(); ++k) {
        futs.emplace_back(std::async(std::launch::async, run_one, busy[k]));
    }
    std::exception_ptr ep;
    try {
        run_one(busy[0]);
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
    lap("device shares");
    if (ep) {
        std::rethrow_exception(ep);
    }
}

// The instantiation list of src/rakau_cuda.cu:536-568: NDim in {2, 3} x F in {float, double} x UInt in {32, 64 bit} x
// Q in {0, 1, 2} x MAC in {bh, bh_geom}.
#define RAKAU_AMD_CUDA_INST(Q, ND, F, U, M)                                                                            \
    template void cuda_acc_pot_impl<Q, ND, F, U, M>(                                                                   \
        const std::array<F *, tree_nvecs_res<Q, ND>> &, const std::vector<tree_size_t<F>> &,                           \
        const tree_node_t<ND, F, U, M> *, tree_size_t<F>, const std::array<const F *, ND + 1u> &, const U *,           \
        tree_size_t<F>, F, F, F, bool);
#define RAKAU_AMD_CUDA_INST_Q(ND, F, U, M)                                                                             \
    RAKAU_AMD_CUDA_INST(0, ND, F, U, M) RAKAU_AMD_CUDA_INST(1, ND, F, U, M) RAKAU_AMD_CUDA_INST(2, ND, F, U, M)
#define RAKAU_AMD_CUDA_INST_MACS(ND, F, U) RAKAU_AMD_CUDA_INST_Q(ND, F, U, mac::bh) RAKAU_AMD_CUDA_INST_Q(ND, F, U, mac::bh_geom)
#define RAKAU_AMD_CUDA_INST_UINTS(ND, F)                                                                               \
    RAKAU_AMD_CUDA_INST_MACS(ND, F, std::uint32_t) RAKAU_AMD_CUDA_INST_MACS(ND, F, std::uint64_t)
RAKAU_AMD_CUDA_INST_UINTS(2, float)
RAKAU_AMD_CUDA_INST_UINTS(2, double)
RAKAU_AMD_CUDA_INST_UINTS(3, float)
RAKAU_AMD_CUDA_INST_UINTS(3, double)

} // namespace detail
} // namespace rakau
