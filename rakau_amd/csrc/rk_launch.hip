// The launch side of a traversal call: graph cache, launch plans, the scratch of the cross-check split variant, and run_impl(),
// which turns (state, range, parameters) into the launch sequence of DESIGN.md section 3.5.
#include "rk_state_internal.hpp"

namespace rkst
{

// Executable graphs of launch sequences with parallel branches (class kernels forked onto side streams) are never destroyed:
// on this runtime hipGraphExecDestroy of one makes a LATER hipGraphLaunch of another such graph die of a segmentation fault
// inside libamdhip64 (tools/stress_graph_capture.py: within 500 key changes in every run, also with the device idle at the
// destroy; never when they are kept; never with linear graphs -- profiles/r03/graph_destroy_crash.txt). They are parked until
// the process ends instead, and only RK_GRAPH_FORKED_MAX (64) of them are ever made per process: after that, forked
// sequences are launched directly (1-4 % slower between 2M and 6M particles). RK_GRAPH_FORKED_MAX=0: never capture them.
int phys(int device);
std::atomic<int> g_forked_execs{0};
int forked_cap()
{
    static const int cap = [] {
        const char *m = std::getenv("RK_GRAPH_FORKED_MAX");
        return m ? std::max(std::atoi(m), 0) : 64;
    }();
    return cap;
}
// Forked executables nobody uses any more (their state went away, its tree was rebuilt, the cache evicted them), per
// physical device. They are not destroyed -- see above -- but RE-TARGETED: a new forked capture first tries
// hipGraphExecUpdate() on one of them (same topology -- pre-pass, fork, the class kernels, join -- with other kernel
// arguments), so a long-lived process that keeps meeting new signatures keeps replaying graphs without the number of
// executables growing.
std::mutex g_parked_mtx;
std::map<int, std::vector<hipGraphExec_t>> g_parked;
// One stream capture (and instantiation / re-targeting of what it captured) at a time in the process, whatever the precision of
// the state: the blocking host-output call captures too, and the device threads of a multi-device split make such calls side by
// side (captures are rare -- once per signature --; concurrent captures on logical devices that alias one GPU failed intermittently
// in round 4). Namespace scope: a static inside the template run_impl<F> was one mutex per precision.
std::mutex g_capture_mtx;
constexpr bool graph_update_enabled()
{
    return true;
}
bool forked_capture_allowed(int phys_dev)
{
    if (forked_cap() == 0) {
        return false;
    }
    if (g_forked_execs.load(std::memory_order_relaxed) < forked_cap()) {
        return true;
    }
    if (!graph_update_enabled()) {
        return false;
    }
    std::lock_guard<std::mutex> lk(g_parked_mtx);
    const auto it = g_parked.find(phys_dev);
    return it != g_parked.end() && !it->second.empty();
}
void retire_graph_exec(int phys_dev, hipGraphExec_t exec, bool forked)
{
    if (!exec) {
        return;
    }
    if (!forked) {
        (void)hipGraphExecDestroy(exec);
        return;
    }
    std::lock_guard<std::mutex> lk(g_parked_mtx);
    g_parked[phys_dev].push_back(exec);
}
// Forget every cached graph of the state (its buffers are about to change or go away). The caller has synchronised the
// device if a replay may still be in flight.
void drop_graph_exec(rk_state &s)
{
    for (auto &e : s.gcache) {
        retire_graph_exec(phys(s.device), e.exec, e.forked);
    }
    s.gcache.clear();
    s.gcache_plan.clear();
    s.plans.clear();
    s.seen_keys.clear();
}

// -1: not set (the environment variable RK_BUILD_EXACT decides, default off).
// Map [p_begin, p_end) onto per-class slices of the group lists.
void range_to_classes(rk_state &s, int64_t p_begin, int64_t p_end, int64_t cb[rk::n_classes],
                      int64_t ce[rk::n_classes], int64_t &g0_out, int64_t &g1_out, bool variant2 = false)
{
    if (p_begin < 0 || p_end < p_begin || p_end > s.nparts) {
        throw rk::error(RK_EINVAL, "invalid particle range [" + std::to_string(p_begin) + ", " + std::to_string(p_end)
                                       + ") for a tree with " + std::to_string(s.nparts) + " particles");
    }
    if (!s.mirrors_valid && variant2 && p_begin == 0 && p_end == s.nparts) {
        // Whole tree on a device-built state: the per-class counts are all that is needed.
        for (int c = 0; c < rk::n_classes; ++c) {
            cb[c] = 0;
            ce[c] = s.class2_count[c];
        }
        g0_out = 0;
        g1_out = s.n_crit;
        return;
    }
    ensure_mirrors(s);
    // First group starting at or after p_begin / p_end.
    const auto g0 = std::lower_bound(s.crit_begin.begin(), s.crit_begin.end(), p_begin) - s.crit_begin.begin();
    const auto g1 = std::lower_bound(s.crit_begin.begin(), s.crit_begin.end(), p_end) - s.crit_begin.begin();
    const bool b_ok = p_begin == s.nparts || (g0 < s.n_crit && s.crit_begin[g0] == p_begin);
    const bool e_ok = p_end == s.nparts || (g1 < s.n_crit && s.crit_begin[g1] == p_end);
    if (!b_ok || !e_ok) {
        // The usual cause behind a drop-in seam: the caller's tree was built with another ncrit than this state was told
        // (the reference's default is 256 when it is compiled for AVX-512, 128 otherwise: tree.hpp:589-595).
        throw rk::error(RK_EINVAL, "the particle range [" + std::to_string(p_begin) + ", " + std::to_string(p_end)
                                       + ") does not start and end at critical node boundaries of a tree with ncrit = "
                                       + std::to_string(s.ncrit)
                                       + " (was the state created with the ncrit the tree was built with?)");
    }
    for (int c = 0; c < rk::n_classes; ++c) {
        const auto &l = variant2 ? s.class2_list[c] : s.class_list[c];
        cb[c] = std::lower_bound(l.begin(), l.end(), static_cast<uint32_t>(g0)) - l.begin();
        ce[c] = std::lower_bound(l.begin(), l.end(), static_cast<uint32_t>(g1)) - l.begin();
    }
    g0_out = g0;
    g1_out = g1;
}

template <typename F>
rk::kparams<F> base_params(const rk_state &s, double mac_value, double G, double eps2)
{
    rk::kparams<F> p{};
    p.part4 = static_cast<const typename rk::vt<F>::v4 *>(s.buf[RK_BUF_PART4]);
    p.node_com = static_cast<const typename rk::vt<F>::v4 *>(s.buf[RK_BUF_NODE_COM]);
    p.node_mac = static_cast<const typename rk::vt<F>::v2 *>(s.buf[RK_BUF_NODE_MAC]);
    p.node_topo = static_cast<const uint4 *>(s.buf[RK_BUF_NODE_TOPO]);
    p.crit = static_cast<const uint4 *>(s.buf[RK_BUF_CRIT]);
    p.child_tab = static_cast<const uint32_t *>(s.buf[RK_BUF_CHILD]);
    p.node_rec = static_cast<const rk::node_rec<F> *>(s.buf[RK_BUF_NODE_REC]);
    p.crit_box = static_cast<const typename rk::vt<F>::v4 *>(s.buf[RK_BUF_CRIT_BOX]);
    p.n_nodes = static_cast<uint32_t>(s.tree_size);
    p.mac_value = static_cast<F>(mac_value);
    p.eps2 = static_cast<F>(eps2);
    p.G = static_cast<F>(G);
    p.mac = s.mac;
    return p;
}

template <typename F>
void census_impl(rk_state &s, int64_t p_begin, int64_t p_end, double mac_value, uint64_t counts[4],
                 uint64_t *per_group = nullptr)
{
    int64_t cb[rk::n_classes], ce[rk::n_classes], g0 = 0, g1 = 0;
    range_to_classes(s, p_begin, p_end, cb, ce, g0, g1, true); // validates the range
    auto p = base_params<F>(s, mac_value, 1., 0.);
    const size_t ng = per_group ? static_cast<size_t>(g1 - g0) : 0;
    struct pool_block {
        void *p = nullptr;
        ~pool_block()
        {
            rk::pool_free(p);
        }
    } blk;
    blk.p = rk::pool_alloc((4 + ng) * sizeof(unsigned long long));
    auto *d_counts = static_cast<unsigned long long *>(blk.p);
    RK_HIP(hipMemset(d_counts, 0, (4 + ng) * sizeof(unsigned long long)));
    rk::launch_census<F>(s, p, g0, g1, d_counts, ng ? d_counts + 4 : nullptr, nullptr);
    RK_HIP(hipMemcpy(counts, d_counts, 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    if (ng) {
        RK_HIP(hipMemcpy(per_group, d_counts + 4, ng * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    }
}

// Launch plan for the critical nodes [g_lo, g_hi): the dispatch order of a repeated call, per class. The weight ("work")
// of a node is its number of particles.
//  * lpt (calls of at most RK_PLAN_MAX_GROUPS nodes): sorted by decreasing work (longest processing time first), so
//    that a launch of only a few rounds of waves ends with its lightest nodes. (Sorting whole supergroups by their mean
//    work instead -- spatially compact runs that share the pre-pass lists -- measured 3-7 % slower from 100k particles
//    to the 0.5M-particle shards of the 4M tree: tools/archive/jobs_r02/r02_job27.sh.)
//  * otherwise: Morton order (neighbouring nodes share tree nodes and leaves in the L2), but the lightest quarter of the
//    nodes goes last: the device then drains over the duration of short waves instead of average ones (4M: 2.32-2.33
//    -> 2.27-2.28 ms; a full LPT order costs 60 % there: tools/archive/jobs_r02/r02_job41.sh), and every XCD works through one spatial
//    region of the range in ALL class kernels (same time, 8.5 % fewer bytes fetched past the L2: tools/archive/jobs_r02/r02_job46.sh).
// Launch-plan list buffers that nothing refers to any more. A launch still in flight (on a stream this library knows nothing about
// by then) may be reading one, so they are not handed back to the block cache at once -- rounds 2-4 drained the whole device for
// every one of them -- but parked here until the device is known to be idle anyway (release_tree(): a rebuild, a destroyed state)
// or 64 of them (a few hundred KB) have piled up, which costs one drain for all.
std::mutex g_retired_mtx;
std::vector<std::pair<int, void *>> g_retired_plan_buffers; // (physical device, buffer)
// Takes the CURRENT device's retired buffers off the list. The caller synchronises the device AFTERWARDS and only then hands
// them back to the block cache: whatever is retired during that synchronisation (by another state or thread on the same GPU,
// possibly while a kernel launched after the drain began still reads it) is not in the snapshot and waits for the next drain.
std::vector<void *> take_retired_plan_buffers()
{
    std::vector<void *> mine;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) {
        return mine;
    }
    std::lock_guard<std::mutex> lk(g_retired_mtx);
    auto keep = g_retired_plan_buffers.begin();
    for (auto &e : g_retired_plan_buffers) {
        if (e.first == dev) {
            mine.push_back(e.second);
        } else {
            *keep++ = e;
        }
    }
    g_retired_plan_buffers.erase(keep, g_retired_plan_buffers.end());
    return mine;
}
void retire_plan_buffer(int dev, void *b) noexcept
{
    size_t n = 0;
    bool listed = false;
    try {
        std::lock_guard<std::mutex> lk(g_retired_mtx);
        g_retired_plan_buffers.emplace_back(dev, b);
        listed = true;
        for (const auto &e : g_retired_plan_buffers) {
            n += e.first == dev ? 1u : 0u;
        }
    } catch (...) {
        n = 64; // (out of memory for the list itself: drain, then free the buffer directly if it is not on the list)
    }
    if (n >= 64) {
        int prev = 0;
        (void)hipGetDevice(&prev);
        (void)hipSetDevice(dev);
        std::vector<void *> mine;
        try {
            mine = take_retired_plan_buffers(); // snapshot first, drain second (see above)
        } catch (...) {
        }
        (void)hipDeviceSynchronize();
        for (void *r : mine) {
            rk::pool_free(r);
        }
        if (!listed) {
            rk::pool_free(b);
        }
        (void)hipSetDevice(prev);
    }
}

template <typename F>
void build_plan(rk_state &s, int64_t p_begin, int64_t p_end, int64_t g_lo, int64_t g_hi, double mac_value, int mode)
{
    // mode 1: heavy-first (sorted by decreasing work); 0: light-tail arrangement per class.
    const bool lpt = mode == 1;
    ensure_mirrors(s);
    // Weight of a node = its number of particles: as good a predictor of a wave's duration as the interaction census
    // (4M: 2.24-2.26 ms either way; tools/archive/jobs_r02/r02_job54.sh) and free, where the census is a traversal of its own (13 ms at 4M).
    if (s.work_cache.size() != static_cast<size_t>(s.n_crit)) {
        s.work_cache.resize(static_cast<size_t>(s.n_crit));
        for (int64_t g = 0; g < s.n_crit; ++g) {
            s.work_cache[static_cast<size_t>(g)]
                = static_cast<uint64_t>(s.crit_end[static_cast<size_t>(g)] - s.crit_begin[static_cast<size_t>(g)]);
        }
    }
    std::vector<uint32_t> lists;
    lists.reserve(static_cast<size_t>(g_hi - g_lo));
    std::vector<uint32_t> region_bound; // light-tail plans: first node of each of the 8 per-XCD regions (+ g_hi)
    // Light-tail arrangement of lists[first ..): Morton order, the lightest quarter of the nodes moved to the end (in Morton
    // order among themselves), one spatial region per XCD.
    auto arrange_light_tail = [&](const std::ptrdiff_t first) {

            // Morton order, the lightest quarter of the nodes moved to the end (in Morton order among themselves).
            constexpr double tail_frac = 0.25; // (2, 4 or 8 work quantiles instead: no better, tools/archive/jobs_r02/r02_job45.sh)
            std::vector<uint64_t> w;
            w.reserve(lists.size() - static_cast<size_t>(first));
            for (auto it = lists.begin() + first; it != lists.end(); ++it) {
                w.push_back(s.work_cache[*it]);
            }
            const size_t k = std::min(w.size() - 1u, static_cast<size_t>(static_cast<double>(w.size()) * tail_frac));
            std::nth_element(w.begin(), w.begin() + static_cast<std::ptrdiff_t>(k), w.end());
            const uint64_t thr = w[k];
            {
                // One spatial region of the range per XCD, the SAME regions for every class kernel: the members of a
                // supergroup (and neighbouring nodes generally) then run on one XCD whatever their class, and the
                // pre-pass lists, tree nodes and leaves they share are fetched into one L2 instead of several.
                // Regions are cut at equal weight. Entry i of the list is served by block i, i.e. by XCD i % 8:
                // the per-XCD queues (bulk in Morton order, then the light nodes) are interleaved and padded with
                // padding entries, which the kernels skip.
                if (region_bound.empty()) {
                    region_bound.assign(9, static_cast<uint32_t>(g_hi));
                    region_bound[0] = static_cast<uint32_t>(g_lo);
                    double total = 0., run = 0.;
                    for (int64_t g = g_lo; g < g_hi; ++g) {
                        total += static_cast<double>(s.work_cache[static_cast<size_t>(g)]);
                    }
                    int x = 1;
                    for (int64_t g = g_lo; g < g_hi && x < 8; ++g) {
                        run += static_cast<double>(s.work_cache[static_cast<size_t>(g)]);
                        while (x < 8 && run >= total * x / 8.) {
                            region_bound[static_cast<size_t>(x++)] = static_cast<uint32_t>(g + 1);
                        }
                    }
                }
                std::vector<uint32_t> q[8];
                for (int pass = 0; pass < 2; ++pass) { // bulk, then light
                    int x = 0;
                    for (auto it = lists.begin() + first; it != lists.end(); ++it) {
                        while (x < 7 && *it >= region_bound[static_cast<size_t>(x) + 1u]) {
                            ++x;
                        }
                        if ((s.work_cache[*it] >= thr) == (pass == 0)) {
                            q[x].push_back(*it);
                        }
                    }
                }
                size_t longest = 0;
                for (const auto &v : q) {
                    longest = std::max(longest, v.size());
                }
                lists.resize(static_cast<size_t>(first));
                for (size_t pos = 0; pos < longest; ++pos) {
                    for (const auto &v : q) {
                        lists.push_back(pos < v.size() ? v[pos] : rk::RK_PLAN_PAD_VALUE);
                    }
                }
            }
            };
    for (int c = 0; c < rk::n_classes; ++c) {
        s.plan.off[c] = static_cast<int64_t>(lists.size());
        if (c == rk::big_class) {
            continue; // served by the block-per-group kernel from the state's own list
        }
        const auto &l = s.class2_list[c];
        const auto b = std::lower_bound(l.begin(), l.end(), static_cast<uint32_t>(g_lo));
        const auto e = std::lower_bound(l.begin(), l.end(), static_cast<uint32_t>(g_hi));
        const auto first = static_cast<std::ptrdiff_t>(lists.size());
        lists.insert(lists.end(), b, e);
        if (lpt) {
            std::stable_sort(lists.begin() + first, lists.end(),
                             [&](uint32_t a, uint32_t b2) { return s.work_cache[a] > s.work_cache[b2]; });
        } else if (lists.size() - static_cast<size_t>(first) > 1u) {
            arrange_light_tail(first);
        }
    }
    s.plan.off[rk::n_classes] = static_cast<int64_t>(lists.size());
    s.plan.off_all = s.plan.n_all = s.plan.off_oth = s.plan.n_oth = s.plan.off_reg = s.plan.n_reg = 0;
    if (lpt) {
        // Merged heavy-first lists over the wave-kernel classes (stable: equal weights keep class, then Morton order): all
        // of them, all but R = 2.
        for (int pass = 0; pass < 2; ++pass) {
            const auto first = static_cast<std::ptrdiff_t>(lists.size());
            for (int c = 0; c < RK_MAX_R; ++c) {
                if (pass == 1 && c == 1) {
                    continue;
                }
                // (copied first: inserting a range of a vector into itself is undefined once it reallocates)
                const std::vector<uint32_t> part(lists.begin() + s.plan.off[c], lists.begin() + s.plan.off[c + 1]);
                lists.insert(lists.end(), part.begin(), part.end());
            }
            std::stable_sort(lists.begin() + first, lists.end(),
                             [&](uint32_t a, uint32_t b2) { return s.work_cache[a] > s.work_cache[b2]; });
            (pass == 0 ? s.plan.off_all : s.plan.off_oth) = first;
            (pass == 0 ? s.plan.n_all : s.plan.n_oth) = static_cast<int64_t>(lists.size()) - first;
        }
        {
            // The merged heavy-first list again as EIGHT QUEUES, one per XCD: the nodes of one eighth of the range's particles,
            // heavy-first inside the queue; entry i of the list is served by block i, i.e. by XCD i % 8 (shorter queues padded with
            // entries the kernels skip). This is what the one-launch kernels run over (round 6): dealt in chunks of 16 consecutive
            // entries of the ONE sorted list -- nodes of like size from anywhere in the range -- every XCD's L2 saw the leaves and
            // lower tree of the whole range; now it sees an eighth. Kernel ms, chunks -> queues (tools/jobs_r06/r06_job19.sh, _job20):
            // k_pc_any 100k 0.098 -> 0.095, 150k 0.133 -> 0.127, 200k 0.169 -> 0.164; k_list_any 250k 0.216 -> 0.209, 350k 0.228 ->
            // 0.220, 500k 0.315 -> 0.302, 1M 0.592 -> 0.580, 1.5M 0.853 -> 0.839; fp64 500k 0.554 -> 0.543, 1M 1.134 -> 1.114; the
            // 0.5M-particle shards of the 4M tree equal (0.310-0.324 / 0.313-0.330). Same nodes, same bits.
            const std::vector<uint32_t> all(lists.begin() + s.plan.off_all, lists.begin() + s.plan.off_all + s.plan.n_all);
            std::vector<uint32_t> q[8];
            const double span = static_cast<double>(p_end - p_begin);
            for (uint32_t g : all) {
                const double rel = static_cast<double>(s.crit_begin[g] - p_begin) / span;
                q[std::min(7, std::max(0, static_cast<int>(rel * 8.)))].push_back(g);
            }
            size_t longest = 0;
            for (const auto &v : q) {
                longest = std::max(longest, v.size());
            }
            const auto first = static_cast<std::ptrdiff_t>(lists.size());
            for (size_t pos = 0; pos < longest; ++pos) {
                for (const auto &v : q) {
                    lists.push_back(pos < v.size() ? v[pos] : rk::RK_PLAN_PAD_VALUE);
                }
            }
            s.plan.off_reg = first;
            s.plan.n_reg = static_cast<int64_t>(lists.size()) - first;
        }
    }
    // The list buffer: always a fresh one (the block cache makes that cheap), so that nothing in flight -- an earlier call on any
    // stream, a cached graph captured on the previous plan -- can be reading what the blocking copy below writes, and no wait is
    // needed here. The buffer it replaces is retired, not freed: see retire_plan_buffer().
    {
        void *buf = rk::pool_alloc(std::max<size_t>(lists.size(), 1) * sizeof(uint32_t));
        int dev = 0;
        RK_HIP(hipGetDevice(&dev));
        s.plan.hold = std::shared_ptr<void>(buf, [dev](void *b) { retire_plan_buffer(dev, b); });
        s.plan.d_lists = buf;
        s.plan.alloc = static_cast<int64_t>(lists.size());
    }
    if (!lists.empty()) {
        RK_HIP(hipMemcpy(s.plan.d_lists, lists.data(), lists.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    }
    s.plan.p_begin = p_begin, s.plan.p_end = p_end, s.plan.mac_value = mac_value;
}

// Calls over at most this many critical nodes take the one-launch producer / consumer kernel, larger ones k_list_any.
// Round 4 (three-wave workgroups, rk_kernels_pc.hip RK_PC_NCONS): k_pc_any / k_list_any kernel ms at 4.2k nodes 0.145 / 0.198,
// 5.6k 0.180 / 0.205, 6.5k 0.218 / 0.218, 8.4k 0.256 / 0.231 (round 3, five-wave workgroups: equal at 4.2k, limit 4000);
// fp64: 2.9k 0.194 / 0.245, 4.2k 0.257 / 0.281, 5.6k 0.328 / 0.305, 6.5k 0.405 / 0.343 (tools/pc_ring_probe.py).
int64_t pc_any_below_nodes(bool fp64)
{
    return fp64 ? int64_t(5000) : int64_t(6000);
}

bool super_cache_enabled()
{
    static const bool on = [] {
        const char *e = std::getenv("RK_SUPER_CACHE"); // 0 disables the reuse of the pre-pass lists across calls
        return !(e && std::atoi(e) == 0);
    }();
    return on;
}

// ---- split traversal (variant 4, rk_kernels_split.hip): scratch of a call ----
// (The automatic variant keeps to the fused kernels: measured on MI355X the split traversal is slower at every size;
// rk_set_kernel_variant(state, 4) selects it per state.)

// Longest list k_lists writes; longer ones (tiny opening angles) go to the fused kernel. A property of the call's
// parameters only, so that a node is served by the same kernel in every launch.
uint32_t split_max_len()
{
    return 32768u;
}

// Sizes the list pool for the critical nodes [g_lo, g_hi) of this call, (re)allocates it if it has to grow, digests the
// report of an earlier call and fills the kernel parameters. Returns whether the fallback launch is needed.
template <typename F>
bool prepare_split(rk_state &s, rk::kparams<F> &p, int64_t p_begin, int64_t p_end, int64_t g_lo, int64_t g_hi, double mac_value,
                   hipStream_t stream)
{
    const rk_state::sl_key key{p_begin, p_end, mac_value};
    if (!s.sl_rep_ev) {
        RK_HIP(hipEventCreateWithFlags(&s.sl_rep_ev, hipEventDisableTiming));
        RK_HIP(hipHostMalloc(reinterpret_cast<void **>(&s.sl_host), 8 * sizeof(uint32_t), hipHostMallocDefault));
        std::fill(s.sl_host, s.sl_host + 8, 0u);
    }
    // (A call on another stream than the previous one has already been ordered behind it: order_after_previous_call().)
    s.sl_used = true;
    if (s.sl_rep_pending && hipEventQuery(s.sl_rep_ev) == hipSuccess) {
        s.sl_rep_pending = false;
        const uint32_t used = s.sl_host[0], fallback = s.sl_host[1], exhausted = s.sl_host[3];
        s.sl_clean_key = s.sl_rep_key;
        s.sl_clean_mode = s.sl_rep_mode, s.sl_clean_npart = s.sl_rep_npart, s.sl_clean_nseg = s.sl_rep_nseg;
        s.sl_clean_valid = fallback == 0u;
        // Keep a quarter of the pool in reserve; double what a call that ran out of segments had.
        int64_t want = static_cast<int64_t>(used) + static_cast<int64_t>(used) / 4 + 1024;
        if (exhausted) {
            want = std::max<int64_t>(want, 2 * static_cast<int64_t>(used) + 4096);
        }
        s.sl_extra_hint = std::max(s.sl_extra_hint, want);
        const int64_t slots = s.sl_host[4];
        if (slots) {
            s.sl_part_hint = std::max(s.sl_part_hint, (exhausted ? 2 : 1) * (slots + slots / 4) + 256);
        }
    }
    const int64_t n_slot = g_hi - g_lo;
    // First guess of the pool: list lengths grow like theta^-3 (about 900 entries at 0.75 on a Plummer sphere).
    const double theta = s.mac == RK_MAC_BH ? 1. / std::sqrt(mac_value) : 1. / mac_value;
    const double est_len = std::min(900. * std::pow(0.75 / std::max(theta, 1e-3), 3.), static_cast<double>(split_max_len()));
    // (each of the two lists of a node holds about half of that; the first segment of each is fixed.)
    const int64_t extra_per_node = 2 * std::max<int64_t>(static_cast<int64_t>(std::ceil(0.75 * est_len / rk::SL_SEG)) - 1, 1);
    constexpr int64_t pool_max_seg = int64_t(24576) * (1ll << 20) / (rk::SL_SEG * 4); // the list pool is at most 24 GiB
    int64_t extra = std::max<int64_t>(n_slot * extra_per_node + 4096, s.sl_extra_hint);
    extra = std::min(extra, std::max<int64_t>(pool_max_seg - 2 * n_slot, 4096));
    const int64_t nseg = 2 * n_slot + extra;
    if (nseg >= (int64_t(1) << 31)) {
        throw rk::error(RK_EINVAL, "too many critical nodes for the split traversal");
    }
    // Calls over few critical nodes spread every node over several wavefronts (one per part of four tiles).
    // (read on every call: tests switch it between calls to compare the two forms bit for bit.)
    const int64_t parts_below = [] {
        const char *e = std::getenv("RK_SL_PARTS_BELOW");
        return e ? std::atoll(e) : int64_t(40000);
    }();
    const bool parts_mode = n_slot <= parts_below;
    int64_t npart = 0;
    if (parts_mode) {
        const double est_parts = std::ceil((800. + est_len + 256.) / 512.);
        npart = std::max<int64_t>(static_cast<int64_t>(1.5 * est_parts * static_cast<double>(n_slot)) + 1024, s.sl_part_hint);
        npart = std::min<int64_t>(npart, (int64_t(1) << 31) / 1024);
    }
    if (s.sl_nseg < nseg || s.sl_ncnt < s.n_crit || !s.sl_ctl || s.sl_npart < npart) {
        RK_HIP(hipDeviceSynchronize());
        drop_graph_exec(s); // it holds the old addresses
        if (s.sl_nseg < nseg) {
            for (void **b : {&s.sl_idx, &s.sl_next}) {
                rk::pool_free(*b);
                *b = nullptr;
            }
            s.sl_nseg = 0;
            const int64_t alloc = nseg + nseg / 8;
            s.sl_idx = rk::pool_alloc(static_cast<size_t>(alloc) * rk::SL_SEG * sizeof(uint32_t));
            s.sl_next = rk::pool_alloc(static_cast<size_t>(alloc) * sizeof(uint32_t));
            s.sl_nseg = alloc;
        }
        if (s.sl_ncnt < s.n_crit) {
            for (void **b : {&s.sl_cnt, &s.sl_fb, &s.sl_pbase}) {
                rk::pool_free(*b);
                *b = nullptr;
            }
            s.sl_ncnt = 0;
            s.sl_cnt = rk::pool_alloc(static_cast<size_t>(s.n_crit) * 2 * sizeof(uint32_t));
            s.sl_fb = rk::pool_alloc(static_cast<size_t>(s.n_crit) * sizeof(uint32_t));
            s.sl_pbase = rk::pool_alloc(static_cast<size_t>(s.n_crit) * sizeof(uint32_t));
            s.sl_ncnt = s.n_crit;
        }
        if (s.sl_npart < npart) {
            rk::pool_free(s.sl_part);
            s.sl_part = nullptr;
            s.sl_npart = 0;
            s.sl_part = rk::pool_alloc(static_cast<size_t>(npart) * 1024 * sizeof(F));
            s.sl_npart = npart;
        }
        if (!s.sl_ctl) {
            s.sl_ctl = rk::pool_alloc(8 * sizeof(uint32_t));
        }
    }
    p.sl_idx = static_cast<uint32_t *>(s.sl_idx);
    p.sl_next = static_cast<uint32_t *>(s.sl_next);
    p.sl_cnt = static_cast<uint32_t *>(s.sl_cnt);
    p.sl_ctl = static_cast<uint32_t *>(s.sl_ctl);
    p.sl_fb = static_cast<uint32_t *>(s.sl_fb);
    p.sl_g0 = static_cast<uint32_t>(g_lo);
    p.sl_nslot = static_cast<uint32_t>(2 * n_slot);
    p.sl_nseg = static_cast<uint32_t>(s.sl_nseg);
    p.sl_max_len = split_max_len();
    p.sl_parts_mode = parts_mode ? 1 : 0;
    p.sl_npart = static_cast<uint32_t>(parts_mode ? s.sl_npart : 0);
    p.sl_pbase = static_cast<uint32_t *>(s.sl_pbase);
    p.sl_part = s.sl_part;
    // The fallback launch is skipped only if a call of this very kind -- range, MAC value, one wave per node or per part, pools
    // at least as large -- has reported an empty fallback list.
    return !(s.sl_clean_valid && s.sl_clean_key == key && s.sl_clean_mode == p.sl_parts_mode
             && s.sl_clean_npart <= static_cast<int64_t>(p.sl_npart) && s.sl_clean_nseg <= static_cast<int64_t>(p.sl_nseg));
}

// ---- class kernels of a call that is the first of its kind, through a re-targeted executable graph ----
// Four class kernels launched directly are four dispatches on four hardware queues joined by events: the queues are served round
// robin, the R <= 2 kernels (8 waves per SIMD, short waves) take most slots first and the R = 3 kernel ends alone; replayed from a
// graph they are served in order from one queue, which is worth 2.5 % at 4M particles and 4.5 % at 2M (tools/jobs_r06/r06_job3.sh).
// Repeated calls get that from the graph cache; a call that is the first of its kind -- every traversal of a time-stepping loop --
// cannot afford a capture + instantiation per call (round 5: 1 %). But its class kernels ARE always the same four kernels: an
// explicit graph of four independent kernel nodes is instantiated once per (device, kernels, classes present) and RE-TARGETED per
// call -- hipGraphExecKernelNodeSetParams with the call's grid sizes and arguments costs nothing measurable
// (tools/ubench/graph_setparams.hip: 2398 us per call re-targeted, 2400 replayed unchanged, 2640-2650 forked directly) -- and
// launched into the call's stream behind the pre-pass, which was launched directly and hides the host time of all this.
// Executables of graphs with parallel branches are never destroyed (see forked_cap()); a state that goes away parks its own for
// the next state with the same kernels. They count towards RK_GRAPH_FORKED_MAX.
std::mutex g_class_graph_mtx;
std::vector<rk_state::class_graph> g_class_graph_pool;

void park_class_graphs(rk_state &s)
{
    if (s.class_graphs.empty()) {
        return;
    }
    // (the caller has synchronised the device: nothing of this state is in flight)
    std::lock_guard<std::mutex> lk(g_class_graph_mtx);
    for (auto &g : s.class_graphs) {
        g_class_graph_pool.push_back(g);
    }
    s.class_graphs.clear();
}

// Launches the class kernels of the call (all classes with nodes; order R = 3, 4, 1, 2 as in launch_list()) through the state's
// re-targeted graph. Returns false -- nothing launched -- when no executable is to be had (cap reached, runtime refuses).
template <typename F>
bool launch_classes_retargeted(rk_state &s, int q, const rk::kparams<F> &p, const int64_t cb[rk::n_classes], const int64_t ce[rk::n_classes],
                               hipStream_t stream)
{
    static const int order[4] = {2, 3, 0, 1};
    static_assert(RK_MAX_R == 4, "four class kernels");
    const int pdev = phys(s.device);
    rk_state::class_graph want;
    want.pdev = pdev;
    int n_nodes = 0;
    rk::kparams<F> args_p[4];
    const uint32_t *args_list[4];
    int args_n[4];
    const uint32_t *args_ndev[4];
    void *kargs[4][4];
    hipKernelNodeParams kp[4];
    for (int k = 0; k < 4; ++k) {
        const int c = order[k];
        const int64_t n = ce[c] - cb[c];
        if (n <= 0) {
            continue;
        }
        want.mask |= 1u << c;
        want.func[n_nodes] = rk::list_kernel_symbol<F>(s, q, c);
        if (!want.func[n_nodes]) {
            return false;
        }
        args_p[n_nodes] = p;
        args_list[n_nodes] = s.cur_lists + s.cur_off[c] + cb[c];
        args_n[n_nodes] = static_cast<int>(n);
        args_ndev[n_nodes] = nullptr;
        kargs[n_nodes][0] = &args_p[n_nodes], kargs[n_nodes][1] = &args_list[n_nodes], kargs[n_nodes][2] = &args_n[n_nodes],
        kargs[n_nodes][3] = &args_ndev[n_nodes];
        kp[n_nodes] = hipKernelNodeParams{};
        kp[n_nodes].func = const_cast<void *>(want.func[n_nodes]);
        kp[n_nodes].gridDim = dim3(static_cast<unsigned>(n));
        kp[n_nodes].blockDim = dim3(64);
        kp[n_nodes].sharedMemBytes = 0;
        kp[n_nodes].kernelParams = kargs[n_nodes];
        kp[n_nodes].extra = nullptr;
        ++n_nodes;
    }
    if (n_nodes < 2) {
        return false; // (a single class: a plain launch is the same thing)
    }
    const auto same = [&](const rk_state::class_graph &g) {
        return g.pdev == pdev && g.mask == want.mask && std::equal(g.func, g.func + 4, want.func);
    };
    rk_state::class_graph *use = nullptr;
    for (auto &g : s.class_graphs) {
        if (same(g)) {
            use = &g;
        }
    }
    if (!use) {
        {
            std::lock_guard<std::mutex> lk(g_class_graph_mtx);
            for (size_t i = 0; i < g_class_graph_pool.size() && !use; ++i) {
                if (same(g_class_graph_pool[i])) {
                    s.class_graphs.push_back(g_class_graph_pool[i]);
                    g_class_graph_pool.erase(g_class_graph_pool.begin() + static_cast<std::ptrdiff_t>(i));
                    use = &s.class_graphs.back();
                }
            }
        }
        if (!use) {
            if (forked_cap() == 0 || g_forked_execs.load(std::memory_order_relaxed) >= forked_cap()) {
                return false;
            }
            std::lock_guard<std::mutex> capture_lock(g_capture_mtx); // (graph construction stays out of other threads' captures)
            if (hipGraphCreate(&want.graph, 0) != hipSuccess) {
                (void)hipGetLastError();
                return false;
            }
            bool ok = true;
            for (int k = 0; k < n_nodes && ok; ++k) {
                ok = hipGraphAddKernelNode(&want.node[k], want.graph, nullptr, 0, &kp[k]) == hipSuccess;
            }
            ok = ok && hipGraphInstantiate(&want.exec, want.graph, nullptr, nullptr, 0) == hipSuccess;
            if (!ok) {
                (void)hipGetLastError();
                (void)hipGraphDestroy(want.graph); // (never instantiated or never launched: safe to destroy)
                return false;
            }
            g_forked_execs.fetch_add(1, std::memory_order_relaxed);
            s.class_graphs.push_back(want);
            use = &s.class_graphs.back();
            RK_HIP(hipGraphLaunch(use->exec, stream));
            return true;
        }
    }
    for (int k = 0; k < n_nodes; ++k) {
        if (hipGraphExecKernelNodeSetParams(use->exec, use->node[k], &kp[k]) != hipSuccess) {
            // (Nothing was launched yet: the caller falls back to direct launches. The executable is given up -- parked, never
            // destroyed -- since some of its nodes may carry this call's arguments and others the previous call's.)
            (void)hipGetLastError();
            std::lock_guard<std::mutex> lk(g_class_graph_mtx);
            use->mask = 0xffu; // matches nothing
            return false;
        }
    }
    RK_HIP(hipGraphLaunch(use->exec, stream));
    return true;
}

// Streams, events and the supergroup scratch a traversal call needs. Created with the state (so that the first call does not
// pay for them: 166 MB of scratch at 4M) and checked again by every call (a rebuilt tree may have more critical nodes).
template <typename F>
void ensure_call_resources(rk_state &s)
{
    if (!s.ev0) {
        RK_HIP(hipEventCreate(&s.ev0));
        RK_HIP(hipEventCreate(&s.ev1));
    }
    if (s.super_k < 0) {
        s.super_k = 16; // (8 ... 32 measure the same within 1 % at every size, round 5)
    }
    if (s.super_k > 0 && s.n_crit > 0) {
        const int64_t n_super = (s.n_crit + s.super_k - 1) / s.super_k;
        if (s.sup_alloc < n_super) {
            RK_HIP(hipDeviceSynchronize());
            for (void **b : {&s.sup_common, &s.sup_resid, &s.sup_cnt}) {
                rk::pool_free(*b);
                *b = nullptr;
            }
            s.sup_alloc = 0;
            s.sup_b = s.sup_e = 0;
            s.sup_common = rk::pool_alloc(static_cast<size_t>(n_super) * rk::SUP_CAPC * sizeof(typename rk::vt<F>::v4));
            s.sup_resid = rk::pool_alloc(static_cast<size_t>(n_super) * rk::SUP_CAPR * sizeof(uint32_t));
            s.sup_cnt = rk::pool_alloc(static_cast<size_t>(n_super) * sizeof(uint2));
            s.sup_alloc = n_super;
        }
    }
    // Side streams / events of the fork-join (created once, outside any capture).
    if (!s.aux_stream[0]) {
        for (int i = 0; i < rk::n_list_R - 1; ++i) {
            RK_HIP(hipStreamCreateWithFlags(&s.aux_stream[i], hipStreamNonBlocking));
            RK_HIP(hipEventCreateWithFlags(&s.ev_join[i], hipEventDisableTiming));
        }
        RK_HIP(hipEventCreateWithFlags(&s.ev_fork, hipEventDisableTiming));
        RK_HIP(hipStreamCreateWithFlags(&s.cap_stream, hipStreamNonBlocking));
    }
    if (!s.sup_ev) {
        RK_HIP(hipEventCreateWithFlags(&s.sup_ev, hipEventDisableTiming));
        RK_HIP(hipEventCreateWithFlags(&s.ev_done, hipEventDisableTiming));
    }
}

// A call on another stream than the state's previous one: everything that call enqueued must be finished before this one touches
// the state's scratch. Waits on the device for the event recorded behind the previous call (rk_common.hpp, last_done); only a
// state whose previous call recorded none -- timing events off and no stream change seen before -- drains the device, once.
void order_after_previous_call(rk_state &s, hipStream_t stream)
{
    if (s.has_last_stream && s.last_stream != stream) {
        if (s.last_done) {
            RK_HIP(hipStreamWaitEvent(stream, s.last_done, 0));
        } else {
            RK_HIP(hipDeviceSynchronize());
        }
        s.multi_stream = true;
    }
    s.last_stream = stream;
    s.has_last_stream = true;
}

void ensure_call_resources_any(rk_state &s)
{
    device_guard dg(s.device);
    if (s.fp == RK_F32) {
        ensure_call_resources<float>(s);
    } else {
        ensure_call_resources<double>(s);
    }
}

template <typename F>
void run_impl(rk_state &s, int q, int64_t p_begin, int64_t p_end, void *const *d_out, double mac_value, double G,
              double eps2, int offset_output, hipStream_t stream, bool allow_graph)
{
    // Variant 2 (LDS interaction lists) is the default; variant 1 is kept for cross-checks.
    const bool v2 = s.variant != 1;
    int64_t cb[rk::n_classes], ce[rk::n_classes], g_lo = 0, g_hi = 0;
    range_to_classes(s, p_begin, p_end, cb, ce, g_lo, g_hi, v2);
    auto p = base_params<F>(s, mac_value, G, eps2);
    // Kernel-side output slots are always {ax, ay, az, pot}; a quadtree's z slot is scratch.
    void *k_out[4] = {};
    if (s.ndim == 3) {
        std::copy(d_out, d_out + rk::nres_of(q), k_out);
    } else {
        if (q != 1 && !s.z_scratch) {
            s.z_scratch = rk::pool_alloc(static_cast<size_t>(std::max<int64_t>(s.nparts, 1)) * sizeof(F));
        }
        if (q == 1) {
            k_out[0] = d_out[0];
        } else {
            k_out[0] = d_out[0], k_out[1] = d_out[1], k_out[2] = s.z_scratch;
            if (q == 2) {
                k_out[3] = d_out[2];
            }
        }
    }
    d_out = k_out;
    for (int k = 0; k < rk::nres_of(q); ++k) {
        p.out[k] = static_cast<F *>(d_out[k]);
    }
    p.out_sub = (offset_output & 1) ? 0u : static_cast<uint32_t>(p_begin);
    p.perm = nullptr;
    if (offset_output & 2) {
        if (!s.bld_perm) {
            throw rk::error(RK_EINVAL, "original-order output needs the permutation: build the state on the device or "
                                       "call rk_state_set_perm() first");
        }
        p.perm = static_cast<const uint32_t *>(s.bld_perm);
    }
    p.dbg = nullptr;
    {
        p.xcd_mode = 1; // a contiguous slice of the list per XCD (launch plans choose their own mapping below)
        p.any_rev = 0;
        p.first_tab = nullptr;
    }
#ifdef RK_STAMPS
    {
        static unsigned long long *d_dbg = nullptr;
        if (!d_dbg) {
            RK_HIP(hipMalloc(&d_dbg, 8 * sizeof(unsigned long long)));
        }
        unsigned long long h[8];
        RK_HIP(hipMemcpy(h, d_dbg, sizeof(h), hipMemcpyDeviceToHost));
        fprintf(stderr, "RK_STAMPS prev: load %llu mac %llu classify %llu leaf %llu dense %llu self %llu rounds %llu other %llu\n", h[0],
                h[1], h[2], h[3], h[4], h[5], h[6], h[7]);
        RK_HIP(hipMemset(d_dbg, 0, sizeof(h)));
        p.dbg = d_dbg;
    }
#endif
#ifdef RK_COUNTS
    {
        // Diagnostic build: event counts of the PREVIOUS call (rk_list_common.hpp, RK_COUNT), 32 per lane-mapping class.
        static unsigned long long *d_cnt = nullptr;
        if (!d_cnt) {
            RK_HIP(hipMalloc(&d_cnt, 128 * sizeof(unsigned long long)));
            RK_HIP(hipMemset(d_cnt, 0, 128 * sizeof(unsigned long long)));
        }
        unsigned long long h[128];
        RK_HIP(hipDeviceSynchronize());
        RK_HIP(hipMemcpy(h, d_cnt, sizeof(h), hipMemcpyDeviceToHost));
        for (int r = 0; r < 4; ++r) {
            fprintf(stderr, "RK_COUNTS prev R=%d:", r + 1);
            for (int i = 0; i < 32; ++i) {
                fprintf(stderr, " %llu", h[r * 32 + i]);
            }
            fprintf(stderr, "\n");
        }
        RK_HIP(hipMemset(d_cnt, 0, sizeof(h)));
        p.dbg = d_cnt;
    }
#endif
#ifdef RK_TRACE
    {
        // Diagnostic build: per-wave {start, end, placement, size} records of the PREVIOUS call go to $RK_TRACE_FILE.
        static unsigned long long *d_tr = nullptr;
        static size_t tr_n = 0;
        const size_t n = static_cast<size_t>(s.n_crit) * 4;
        if (d_tr && tr_n == n) {
            RK_HIP(hipDeviceSynchronize());
            std::vector<unsigned long long> h(n);
            RK_HIP(hipMemcpy(h.data(), d_tr, n * 8, hipMemcpyDeviceToHost));
            if (const char *f = std::getenv("RK_TRACE_FILE")) {
                if (FILE *fp = std::fopen(f, "wb")) {
                    std::fwrite(h.data(), 8, n, fp);
                    std::fclose(fp);
                }
            }
        }
        if (tr_n != n) {
            if (d_tr) {
                (void)hipFree(d_tr);
            }
            RK_HIP(hipMalloc(&d_tr, n * 8));
            tr_n = n;
        }
        RK_HIP(hipMemset(d_tr, 0, n * 8));
        p.dbg = d_tr;
    }
#endif
    ensure_call_resources<F>(s);
    order_after_previous_call(s, stream);
    // allow_graph is false on the host-output path, which waits on ev1 for completion.
    const bool need_done_event = !allow_graph || s.want_done_event;
    if (s.timing && !s.keep_ev0) {
        RK_HIP(hipEventRecord(s.ev0, stream));
    }
    bool ran_super = false;
    p.super_k = 0;
    p.n_crit = static_cast<uint32_t>(s.n_crit);
    p.sup_common = nullptr;
    p.sup_resid = nullptr;
    p.sup_cnt = nullptr;
    if (v2) {
        // Supergroup pre-pass: K consecutive groups share the upper part of list building (16 of them).
        if (s.super_k > 0 && s.n_crit > 0) {
            p.super_k = static_cast<uint32_t>(s.super_k);
            p.sup_common = static_cast<typename rk::vt<F>::v4 *>(s.sup_common);
            p.sup_resid = static_cast<uint32_t *>(s.sup_resid);
            p.sup_cnt = static_cast<uint2 *>(s.sup_cnt);
        }
        static const bool serial = [] {
            // RK_SERIAL_CLASSES=1 keeps the class kernels on one stream (one after the other), which gives
            // per-kernel durations in a profile that add up to the step time.
            const char *e = std::getenv("RK_SERIAL_CLASSES");
            return e && std::atoi(e) != 0;
        }();
        static const bool use_graph = [] {
            const char *e = std::getenv("RK_GRAPH"); // 0 disables the hipGraph replay of a repeated call
            return !(e && std::atoi(e) == 0);
        }();
        // Default group lists: the state's own (ascending critical nodes per class).
        s.cur_lists = static_cast<const uint32_t *>(s.buf[RK_BUF_CLASS]);
        std::copy(s.class2_off, s.class2_off + rk::n_classes + 1, s.cur_off);
        const int64_t big_b = cb[rk::big_class], big_e = ce[rk::big_class];
        {
            // RK_PLAN: 0 = never reorder, 1 = reorder repeated calls (default), 2 = reorder every call.
            static const int plan_mode = [] {
                const char *e = std::getenv("RK_PLAN");
                return e ? std::atoi(e) : 1;
            }();
            static const int64_t plan_max_groups = [] {
                const char *e = std::getenv("RK_PLAN_MAX_GROUPS");
                // (60000 measured too: 2M particles = 54k nodes 1.19 instead of 1.22 ms, but the two 54k-node shards of the 4M
                // tree 1.34-1.37 instead of 1.25-1.28: the heavy-first order gives up the L2 locality of neighbouring nodes.)
                // Round 5: 45000 (rounds 2-4: 30000). With the list kernels' new occupancies the one launch is ahead further up:
                // whole trees of 31.9k / 38.2k / 44.3k nodes 0.823 / 0.921 / 1.029 -> 0.697 / 0.842 / 0.971 ms; at 54k nodes (2M)
                // 1.130 -> 1.107 but the 54k-node shards of the 4M tree 1.28 -> 1.32 (tools/archive/jobs_r05/r05_job32.sh).
                return e ? std::atoll(e) : int64_t(45000);
            }();
            bool cached = s.plan.d_lists && s.plan.p_begin == p_begin && s.plan.p_end == p_end
                          && s.plan.mac_value == mac_value;
            if (!cached) {
                // A plan that one of the cached graphs was captured with serves this range too (a caller alternating among a
                // few ranges gets its plans back together with its graphs).
                for (const auto *v : {&s.plans, &s.gcache_plan}) {
                    for (const auto &pl : *v) {
                        if (!cached && pl.d_lists && pl.p_begin == p_begin && pl.p_end == p_end && pl.mac_value == mac_value) {
                            s.plan = pl;
                            cached = true;
                        }
                    }
                }
            }
            // (tracked for every call, also on the host-output path and with RK_GRAPH=0, where no graph key is kept.)
            // A (range, MAC value) seen among the last calls gets a plan: also a caller that alternates among a few ranges.
            const rk_state::sl_key this_call{p_begin, p_end, mac_value};
            bool repeats = false;
            for (const auto &k : s.plan_keys) {
                repeats = repeats || k == this_call;
            }
            if (!repeats) {
                if (s.plan_keys.size() >= 8) {
                    s.plan_keys.erase(s.plan_keys.begin());
                }
                s.plan_keys.push_back(this_call);
            }
            // Beyond this many nodes the launch is so many rounds of waves deep that its tail no longer matters, and the
            // contiguous slice of the Morton order per XCD (xcd_mode 1) wins: 16M fp64 +0.6 %, 64M +1.5 % with a plan.
            constexpr int64_t plan_tail_max_groups = rk::FIRST_TAIL_MAX;
            const bool want = g_hi > g_lo
                              && (plan_mode == 2
                                  || (plan_mode == 1 && g_hi - g_lo <= plan_tail_max_groups && (cached || repeats)));
            if (want) {
                if (!cached) {
                    build_plan<F>(s, p_begin, p_end, g_lo, g_hi, mac_value,
                                  g_hi - g_lo <= plan_max_groups ? 1 : 0);
                    // Remember it (four plans; the oldest goes -- its buffer once nothing else holds it).
                    if (s.plans.size() >= 4) {
                        s.plans.erase(s.plans.begin());
                    }
                    s.plans.push_back(s.plan);
                }
                s.cur_lists = static_cast<const uint32_t *>(s.plan.d_lists);
                std::copy(s.plan.off, s.plan.off + rk::n_classes + 1, s.cur_off);
                for (int c = 0; c < rk::n_classes; ++c) {
                    cb[c] = 0;
                    ce[c] = s.plan.off[c + 1] - s.plan.off[c];
                }
                // Heavy-first order: deal chunks of consecutive list entries round-robin to the XCDs.
                p.xcd_mode = 0;
                // Light-tail order: the plan list interleaves the per-XCD queues itself (block i serves entry i).
                if (g_hi - g_lo > plan_max_groups) {
                    p.xcd_mode = 2;
                }
            }
        }
        // The launch sequence of one call: pre-pass, then the per-class kernels forked onto side streams (so that
        // the tail of one overlaps the others), joined back, then the big-group fallback. Stream-ordered work only,
        // so it can be recorded into a hipGraph.
        // Supergroup pre-pass: skipped when the scratch already holds these supergroups for this MAC value.
        const int64_t sb = (p.super_k && g_hi > g_lo) ? g_lo / s.super_k : 0,
                      se = (p.super_k && g_hi > g_lo) ? (g_hi - 1) / s.super_k + 1 : 0;
        const bool sup_cache = super_cache_enabled();
        // (The cached pre-pass output may have been written, or be in use, on another stream: order_after_previous_call() has
        // put this call behind the previous one in that case, so it may reuse or extend the lists.)
        // Variant 4: list building and dense evaluation as two kernels.
        const bool split = g_hi > g_lo && s.variant == 4;
        const bool need_super = se > sb && !(sup_cache && s.sup_mac == mac_value && s.sup_b <= sb && se <= s.sup_e);
        ran_super = need_super;
        bool split_fb = false;
        if (split) {
            split_fb = prepare_split<F>(s, p, p_begin, p_end, g_lo, g_hi, mac_value, stream);
        }
        // Class launches (what remains for calls that have no one-launch form: sub-ranges without a plan, trees beyond the
        // limits). Variant 0 (automatic): a call over few critical nodes cannot fill the device with one wave per node, and
        // ends with its longest serial chains running alone. Such calls hand lane-mapping classes to the producer /
        // consumer kernel (1 + R waves per node): all of them below 5 000 critical nodes, the class with the longest
        // chains (R = 2: 64 < T <= 128 targets on one wave) below 20 000. Both kernels give the same bits, so this is
        // a pure scheduling decision (measured: tools/archive/jobs_r02/r02_job11.sh, r02_job12.sh).
        constexpr int64_t pc_all_below = 5000, pc_r2_below = 20000;
        unsigned pc_mask = 0u;
        if (s.variant == 3) {
            pc_mask = 0xfu;
        } else if (s.variant == 0) {
            const int64_t ng = g_hi - g_lo;
            pc_mask = ng <= pc_all_below ? 0xfu : (ng <= pc_r2_below ? 0x2u : 0u);
        }
        // 0: per-class launches; 1: k_pc_any; 2: k_pc for R = 2 + k_list_any for the rest; 3: k_list_any (heavy-first plans
        // only, i.e. repeated calls over at most RK_PLAN_MAX_GROUPS critical nodes; RK_ANY=0 keeps the class launches).
        static const int any_env = [] {
            const char *e = std::getenv("RK_ANY");
            return e ? std::atoi(e) : -1;
        }();
        int any_mode = 0;
        if (!split && s.variant == 0 && any_env != 0 && s.cur_lists == static_cast<const uint32_t *>(s.plan.d_lists)
            && s.plan.n_all > 0 && s.plan.n_all == g_hi - g_lo - (big_e - big_b)) {
            // Measured (tools/archive/any_probe.py, profiles/r03/one_launch_kernels.txt): k_pc_any at 2.9k nodes 0.134 ms (class
            // launches 0.140, k_list_any 0.173); k_list_any at 9.4k nodes 0.23 (0.29-0.30; with R = 2 on k_pc 0.27), on the
            // 13.4k-node shards of the 4M tree 0.373-0.379 (0.406-0.412; 0.40), at 26k nodes 0.62 (0.66).
            // 4.2k nodes: class launches on the producer / consumer kernel 0.173, k_pc_any 0.196 (its five-wave workgroups
            // are admitted four per CU), k_list_any 0.193; 5.6k nodes: k_list_any 0.205 (class launches 0.25), 6.5k: 0.215
            // (0.27); 54k nodes (2M particles): 1.19 (1.22).
            const int64_t pc_any_below = pc_any_below_nodes(sizeof(F) == 8);
            // (Since forked launch sequences are no longer replayed from a graph, the class launches of 3.2k-5k nodes lost
            // their place -- queued calls, ms: 3.6k nodes 0.167, k_pc_any 0.155, k_list_any 0.177; 3.9k: 0.275 / 0.178 / 0.181;
            // 4.5k: 0.202 / 0.202 / 0.186; 5.1k: 0.335 / 0.230 / 0.192 -- tools/archive/any_probe3.py.)
            any_mode = any_env > 0 ? any_env : (g_hi - g_lo <= pc_any_below ? 1 : 3);
        }
        // A call WITHOUT a plan over all critical nodes of a tree that came with the light-tail arrangement of a first call (made on the
        // device with the tree, rk_build.hip: trees of FIRST_ORDER_MAX .. FIRST_TAIL_MAX critical nodes): the class kernels take their
        // nodes from its per-region queues -- every traversal of a time-stepping loop on 2M-8M particles is such a call.
        // Measured against what such calls ran before (examples/leapfrog, traversal ms, tools/jobs_r06/r06_job5.sh): 1.9M particles
        // (51k nodes; one launch over the class lists read backwards) 0.986 -> 0.973, 2.2M (60k) 1.13 -> 1.08, 3M (class kernels,
        // one Morton slice per XCD) 1.48 -> 1.42, 4M 1.92 -> 1.85, 6M 2.88 -> 2.80; the rebuild pays 10-15 us for it.
        if (s.cur_lists == static_cast<const uint32_t *>(s.buf[RK_BUF_CLASS]) && s.first_tail_valid && s.first_order && s.first_tab
            && s.variant == 0 && any_mode == 0 && !split && pc_mask == 0u && g_lo == 0 && g_hi == s.n_crit) {
            s.cur_lists = static_cast<const uint32_t *>(s.first_order);
            for (int c = 0; c < RK_MAX_R; ++c) {
                s.cur_off[c] = 0;
                cb[c] = 0;
                ce[c] = static_cast<int64_t>(s.first_grid[c]);
            }
            p.xcd_mode = 3;
            p.first_tab = static_cast<const uint32_t *>(s.first_tab);
        }
        // A small call WITHOUT a plan (the first call on a tree: every step of a time-stepping loop) that covers all critical
        // nodes: one launch too, over the state's own class lists read backwards -- R = 4 first, the lightest class last, which
        // is most of what the heavy-first plan buys -- instead of four class kernels forked onto side streams.
        const uint32_t *first_list = nullptr;
        int64_t first_n = 0;
        {
            constexpr bool any_first = true; // (RK_ANY=0 keeps the class launches for these calls too)
            static const int64_t any_first_max = [] {
                // First calls, ms (tools/archive/first_call_probe.py, Plummer; class launches -> this): 100k 0.191 -> 0.178, 350k
                // 0.488 -> 0.360, 1M 0.83 -> 0.76, 1.8M (47.6k nodes) 1.51 -> 1.25; leapfrog harness 100k 0.187 -> 0.166,
                // 2M (~50k nodes) 1.082 -> 1.069; beyond, the class kernels with a Morton slice per XCD win: 4M 1.97 vs 2.05.
                return int64_t(60000);
            }();
            const int64_t n_wave = s.class2_off[RK_MAX_R] - s.class2_off[0];
            if (any_first && any_mode == 0 && !split && s.variant == 0 && any_env != 0
                && s.cur_lists == static_cast<const uint32_t *>(s.buf[RK_BUF_CLASS]) && g_lo == 0 && g_hi == s.n_crit
                && g_hi <= any_first_max && n_wave > 0 && n_wave == g_hi - (big_e - big_b)) {
                first_list = s.cur_lists + s.class2_off[0];
                first_n = n_wave;
                any_mode = (any_env == 1 || any_env == 3) ? any_env : (g_hi <= pc_any_below_nodes(sizeof(F) == 8) ? 1 : 3);
                p.any_rev = 1;
                p.xcd_mode = 0; // chunks of consecutive entries dealt round-robin to the XCDs, as for a heavy-first plan
                if (s.first_order_valid && s.first_order && s.first_tab) {
                    // A small tree built (or converted) on the device comes with the order of a heavy-first plan -- eight per-XCD
                    // queues, nodes by decreasing size inside (rk_build.hip first_key()): 100k particles 0.144 -> 0.10 ms on k_pc_any.
                    first_list = static_cast<const uint32_t *>(s.first_order);
                    first_n = static_cast<int64_t>(s.first_grid[0]);
                    p.any_rev = 0;
                    p.xcd_mode = 4; // (the queue table: rk_list_common.hpp any_list_entry())
                    p.first_tab = static_cast<const uint32_t *>(s.first_tab);
                }
            }
        }
        // The class kernels run on side streams, forked from and joined back to the call's stream (a single launch
        // needs neither).
        const bool forked = !serial && any_mode != 1 && any_mode != 3;
        auto enqueue = [&](hipStream_t st, bool capturing) {
            if (need_super) {
                rk::launch_super<F>(s, p, sb, se, st);
            }
            if (split) {
                RK_HIP(hipMemsetAsync(s.sl_ctl, 0, 8 * sizeof(uint32_t), st));
                rk::launch_lists<F>(s, p, g_lo, g_hi, st);
            }
            hipStream_t streams[rk::n_list_R];
            for (int i = 0; i < rk::n_list_R; ++i) {
                streams[i] = (serial || i == 0) ? st : s.aux_stream[i - 1];
            }
            // The class kernels of a call that is not being captured: through the state's re-targeted executable graph when there
            // is one to be had (launch_classes_retargeted()), forked onto the side streams otherwise.
            bool classes_done = false;
            if (!capturing && forked && !split && any_mode == 0 && pc_mask == 0u && use_graph && allow_graph) {
                classes_done = launch_classes_retargeted<F>(s, q, p, cb, ce, st);
            }
            const bool fork_now = forked && !classes_done;
            if (fork_now) {
                RK_HIP(hipEventRecord(s.ev_fork, st));
                for (int i = 0; i < rk::n_list_R - 1; ++i) {
                    RK_HIP(hipStreamWaitEvent(s.aux_stream[i], s.ev_fork, 0));
                }
            }
            if (split) {
                if (p.sl_parts_mode) {
                    // One wavefront per part, then the per-node sums (same stream per class: ordered).
                    rk::launch_dense<F>(s, q, p, cb, ce, streams, 0xfu, 1);
                    rk::launch_dense<F>(s, q, p, cb, ce, streams, 0xfu, 2);
                } else {
                    rk::launch_dense<F>(s, q, p, cb, ce, streams, 0xfu, 0);
                }
            } else if (any_mode != 0) {
                // A small repeated call: one launch over the heavy-first list of ALL classes (or two: the R = 2 class on its
                // producer / consumer kernel, everything else on k_list_any) instead of four that start 25-45 us apart.
                const auto *pl = static_cast<const uint32_t *>(s.plan.d_lists);
                if (first_list) {
                    if (any_mode == 1) {
                        rk::launch_pc_any<F>(s, q, p, first_list, first_n, streams[0]);
                    } else {
                        rk::launch_list_any<F>(s, q, p, first_list, first_n, streams[0]);
                    }
                } else if (any_mode == 1) {
                    auto pr = p;
                    pr.xcd_mode = 2; // (the per-XCD queues of the plan: block i serves entry i)
                    rk::launch_pc_any<F>(s, q, pr, pl + s.plan.off_reg, s.plan.n_reg, streams[0]);
                } else if (any_mode == 2) {
                    rk::launch_pc<F>(s, q, p, cb, ce, streams, 0x2u);
                    rk::launch_list_any<F>(s, q, p, pl + s.plan.off_oth, s.plan.n_oth, streams[0]);
                } else {
                    auto pr = p;
                    pr.xcd_mode = 2;
                    rk::launch_list_any<F>(s, q, pr, pl + s.plan.off_reg, s.plan.n_reg, streams[0]);
                }
            } else if (!classes_done) {
                if (pc_mask) {
                    rk::launch_pc<F>(s, q, p, cb, ce, streams, pc_mask);
                }
                if (pc_mask != 0xfu) {
                    rk::launch_list<F>(s, q, p, cb, ce, streams, ~pc_mask);
                }
            }
            if (fork_now) {
                for (int i = 0; i < rk::n_list_R - 1; ++i) {
                    RK_HIP(hipEventRecord(s.ev_join[i], s.aux_stream[i]));
                    RK_HIP(hipStreamWaitEvent(st, s.ev_join[i], 0));
                }
            }
            // Critical nodes too large for one wavefront: a workgroup each, cut into chunks of targets (k_list<BIG>).
            // (Variant 1 walks them with its scalar block-per-node kernel instead: the cross-check, rk_kernels_xcheck.hip.)
            const auto *big_list = static_cast<const uint32_t *>(s.buf[RK_BUF_CLASS]) + s.class2_off[rk::big_class] + big_b;
            rk::launch_list_big<F>(s, q, p, big_list, big_e - big_b, st);
            if (split && split_fb) {
                // Nodes whose list k_lists did not complete (longer than the cap, or the pool ran out): the chunked form of
                // the fused kernel, over a list whose length is only known on the device. Skipped once a report of this
                // very call (range, MAC value) has shown the list to be empty.
                rk::launch_list_big<F>(s, q, p, static_cast<const uint32_t *>(s.sl_fb), g_hi - g_lo, st,
                                       static_cast<const uint32_t *>(s.sl_ctl) + 1);
                if (!capturing && !s.sl_rep_pending) {
                    RK_HIP(hipMemcpyAsync(s.sl_host, s.sl_ctl, 8 * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
                    RK_HIP(hipEventRecord(s.sl_rep_ev, st));
                    s.sl_rep_key = rk_state::sl_key{p_begin, p_end, mac_value};
                    s.sl_rep_mode = p.sl_parts_mode, s.sl_rep_npart = p.sl_npart, s.sl_rep_nseg = p.sl_nseg;
                    s.sl_rep_pending = true;
                }
            }
        };
        // The one-launch sequences (pre-pass + k_pc_any / k_list_any on one stream) are launched directly: replayed from a graph
        // they are 3-8 us slower per call (device-resident ms per step, graph / direct: 100k 0.1141-0.1153 / 0.1109-0.1115, 350k
        // 0.2598-0.2608 / 0.2522-0.2532, 1M 0.668 / 0.653-0.664; tools/archive/jobs_r04/r04_job57.sh) -- round 2 measured the opposite for
        // the four forked class kernels these sizes ran then. RK_GRAPH_LINEAR=1 captures them too.
        static const bool graph_linear = [] {
            const char *e = std::getenv("RK_GRAPH_LINEAR");
            return e && std::atoi(e) != 0;
        }();
        const bool one_launch_seq = (any_mode == 1 || any_mode == 3) && !split;
        if (use_graph && allow_graph && (!one_launch_seq || graph_linear)) {
            // A call that repeats the previous one (same range, outputs, parameters) replays a captured graph:
            // one hipGraphLaunch instead of a handful of runtime calls and stream hand-overs (the forked class kernels).
            rk_state::graph_key key{};
            key.q = q, key.p_begin = p_begin, key.p_end = p_end, key.mac_value = mac_value, key.G = G, key.eps2 = eps2;
            key.offset_output = offset_output, key.super_k = s.super_k, key.variant = s.variant;
            key.with_super = need_super ? 1 : 0;
            key.pad = split ? (split_fb ? 2 : 1) : 0;
            for (int k = 0; k < rk::nres_of(q); ++k) {
                key.out[k] = d_out[k];
            }
            key.perm = p.perm;
            static const size_t cache_cap = [] {
                const char *e = std::getenv("RK_GRAPH_CACHE");
                return static_cast<size_t>(e ? std::max(std::atoi(e), 1) : 8);
            }();
            const int pdev = phys(s.device);
            size_t hit = s.gcache.size();
            for (size_t i = 0; i < s.gcache.size(); ++i) {
                if (std::memcmp(&key, &s.gcache[i].key, sizeof(key)) == 0) {
                    hit = i;
                    break;
                }
            }
            bool seen = false;
            for (const auto &k : s.seen_keys) {
                seen = seen || std::memcmp(&key, &k, sizeof(key)) == 0;
            }
            if (!seen) {
                if (s.seen_keys.size() >= 2 * cache_cap) {
                    s.seen_keys.erase(s.seen_keys.begin());
                }
                s.seen_keys.push_back(key);
            }
            // What the captured sequence reads besides the state's own buffers: the launch plan (if this call uses one).
            const bool uses_plan = s.cur_lists == static_cast<const uint32_t *>(s.plan.d_lists) && s.plan.d_lists;
            if (hit < s.gcache.size()) {
                // Seen and captured before: replay, and move the entry to the most-recently-used end.
                if (hit + 1 != s.gcache.size()) {
                    std::rotate(s.gcache.begin() + static_cast<std::ptrdiff_t>(hit), s.gcache.begin() + static_cast<std::ptrdiff_t>(hit) + 1,
                                s.gcache.end());
                    std::rotate(s.gcache_plan.begin() + static_cast<std::ptrdiff_t>(hit),
                                s.gcache_plan.begin() + static_cast<std::ptrdiff_t>(hit) + 1, s.gcache_plan.end());
                }
                RK_HIP(hipGraphLaunch(s.gcache.back().exec, stream));
                ++s.graph_stats[0];
            } else if (!seen || (forked && !forked_capture_allowed(pdev) && !(graph_update_enabled() && forked_cap() > 0 && [&] {
                           for (const auto &e : s.gcache) {
                               if (e.forked) {
                                   return true; // one of this state's own forked executables can be re-targeted
                               }
                           }
                           return false;
                       }()))) {
                // First call of its kind (e.g. once per rebuilt tree in a time-stepping loop): launch directly, a capture +
                // instantiation would cost more than it saves. So are forked sequences when the process has made its share of
                // executable graphs with parallel branches and none is free to be re-targeted.
                enqueue(stream, false);
                ++s.graph_stats[2];
            } else {
                hipGraph_t graph = nullptr;
                // One capture at a time in the process (g_capture_mtx); released before the launch of what was captured.
                std::unique_lock<std::mutex> capture_lock(g_capture_mtx);
                RK_HIP(hipStreamBeginCapture(s.cap_stream, hipStreamCaptureModeThreadLocal));
                try {
                    enqueue(s.cap_stream, true);
                } catch (...) {
                    (void)hipStreamEndCapture(s.cap_stream, &graph);
                    if (graph) {
                        (void)hipGraphDestroy(graph);
                    }
                    throw;
                }
                RK_HIP(hipStreamEndCapture(s.cap_stream, &graph));
                hipGraphExec_t exec = nullptr;
                bool updated = false;
                if (forked && graph_update_enabled() && g_forked_execs.load(std::memory_order_relaxed) >= forked_cap()) {
                    // No new forked executable may be made: if none is parked either, give up this state's least recently
                    // used one (after a device synchronisation: it may be in flight) so that it can be re-targeted below.
                    bool parked;
                    {
                        std::lock_guard<std::mutex> lk(g_parked_mtx);
                        parked = !g_parked[pdev].empty();
                    }
                    for (size_t i = 0; !parked && i < s.gcache.size(); ++i) {
                        if (s.gcache[i].forked) {
                            RK_HIP(hipDeviceSynchronize());
                            retire_graph_exec(pdev, s.gcache[i].exec, true);
                            s.gcache.erase(s.gcache.begin() + static_cast<std::ptrdiff_t>(i));
                            s.gcache_plan.erase(s.gcache_plan.begin() + static_cast<std::ptrdiff_t>(i));
                            parked = true;
                        }
                    }
                }
                if (forked && graph_update_enabled()) {
                    // Re-target a parked executable of this device whose topology matches (it was retired after a device
                    // synchronisation, so it is not in flight).
                    std::vector<hipGraphExec_t> cand;
                    {
                        std::lock_guard<std::mutex> lk(g_parked_mtx);
                        cand.swap(g_parked[pdev]);
                    }
                    for (size_t i = cand.size(); i-- > 0 && !exec;) {
                        hipGraphNode_t err_node = nullptr;
                        hipGraphExecUpdateResult res{};
                        if (hipGraphExecUpdate(cand[i], graph, &err_node, &res) == hipSuccess) {
                            exec = cand[i];
                            cand.erase(cand.begin() + static_cast<std::ptrdiff_t>(i));
                            updated = true;
                        } else {
                            (void)hipGetLastError();
                        }
                    }
                    std::lock_guard<std::mutex> lk(g_parked_mtx);
                    auto &v = g_parked[pdev];
                    v.insert(v.end(), cand.begin(), cand.end());
                }
                if (!exec && forked && g_forked_execs.load(std::memory_order_relaxed) >= forked_cap()) {
                    // No parked executable took the new topology and no new one may be made: direct launch.
                    (void)hipGraphDestroy(graph);
                    enqueue(stream, false);
                    ++s.graph_stats[2];
                } else {
                    if (!exec) {
                        const hipError_t ie = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
                        (void)hipGraphDestroy(graph);
                        RK_HIP(ie);
                        if (forked) {
                            g_forked_execs.fetch_add(1, std::memory_order_relaxed);
                        }
                    } else {
                        (void)hipGraphDestroy(graph);
                    }
                    if (s.gcache.size() >= cache_cap) {
                        // Evict the least recently used entry. It may still be in flight on some stream: wait, then destroy
                        // (linear) or park it for re-targeting (forked).
                        RK_HIP(hipDeviceSynchronize());
                        retire_graph_exec(pdev, s.gcache.front().exec, s.gcache.front().forked);
                        s.gcache.erase(s.gcache.begin());
                        s.gcache_plan.erase(s.gcache_plan.begin());
                    }
                    s.gcache.push_back(rk_state::graph_entry{key, exec, forked});
                    s.gcache_plan.push_back(uses_plan ? s.plan : rk_state::launch_plan{});
                    capture_lock.unlock();
                    RK_HIP(hipGraphLaunch(exec, stream));
                    ++s.graph_stats[1];
                    s.graph_stats[3] += updated ? 1u : 0u;
                }
            }
        } else {
            enqueue(stream, false);
        }
    } else {
        rk::launch_traversal<F>(s, q, p, cb, ce, stream);
    }
    if (v2 && p.super_k && g_hi > g_lo) {
        const int64_t sb2 = g_lo / s.super_k, se2 = (g_hi - 1) / s.super_k + 1;
        if (s.sup_mac == mac_value && s.sup_e > s.sup_b && sb2 <= s.sup_e && s.sup_b <= se2) {
            s.sup_b = std::min(s.sup_b, sb2), s.sup_e = std::max(s.sup_e, se2); // overlapping or adjacent: the union
        } else if (!(s.sup_mac == mac_value && s.sup_b <= sb2 && se2 <= s.sup_e)) {
            s.sup_mac = mac_value, s.sup_b = sb2, s.sup_e = se2;
        }
        (void)ran_super; // (a call on another stream synchronises with sup_stream before it reuses or extends the lists)
    }
    // Every event record is a barrier packet between this call and the next one on the stream (~10 us each on the GPU):
    // timing events only if wanted (rk_state_set_timing), the completion event only where something waits on it.
    if (s.timing || need_done_event) {
        RK_HIP(hipEventRecord(s.ev1, stream));
        s.last_done = s.ev1;
    } else if (s.multi_stream) {
        RK_HIP(hipEventRecord(s.ev_done, stream)); // (3 us per call, only for callers that do change streams)
        s.last_done = s.ev_done;
    } else {
        s.last_done = nullptr;
    }
    s.timed = s.timing;
}

void check_call(const rk_state *s, int q, void *const *out, double mac_value, double G, double eps2)
{
    if (!s) {
        throw rk::error(RK_EINVAL, "null state");
    }
    if (q < 0 || q > 2) {
        throw rk::error(RK_EINVAL, "q must be 0 (accelerations), 1 (potentials) or 2 (both)");
    }
    if (!out) {
        throw rk::error(RK_EINVAL, "null output array");
    }
    for (int k = 0; k < user_nres(*s, q); ++k) {
        if (!out[k]) {
            throw rk::error(RK_EINVAL, "null output pointer");
        }
    }
    // Same domain checks as tree.hpp:3299-3319 of the reference, on the transformed values.
    if (!std::isfinite(mac_value) || mac_value <= 0.) {
        throw rk::error(RK_EDOMAIN, "The transformed MAC value must be finite and positive, but it is "
                                        + std::to_string(mac_value) + " instead");
    }
    if (!std::isfinite(eps2) || eps2 < 0.) {
        throw rk::error(RK_EDOMAIN, "The square of the softening length must be finite and non-negative, but it is "
                                        + std::to_string(eps2) + " instead");
    }
    if (!std::isfinite(G)) {
        throw rk::error(RK_EDOMAIN, "The value of the gravitational constant G must be finite, but it is "
                                        + std::to_string(G) + " instead");
    }
}


template void run_impl<float>(rk_state &, int, int64_t, int64_t, void *const *, double, double, double, int, hipStream_t, bool);
template void run_impl<double>(rk_state &, int, int64_t, int64_t, void *const *, double, double, double, int, hipStream_t, bool);

} // namespace rkst

extern "C" {

int rk_count_interactions(rk_state *s, int64_t p_begin, int64_t p_end, double mac_value, uint64_t counts[4])
{
    return guard([&] {
        if (!s || !counts) {
            throw rk::error(RK_EINVAL, "null argument");
        }
        if (!std::isfinite(mac_value) || mac_value <= 0.) {
            throw rk::error(RK_EDOMAIN, "The transformed MAC value must be finite and positive, but it is "
                                            + std::to_string(mac_value) + " instead");
        }
        std::fill(counts, counts + 4, uint64_t(0));
        if (!s->nparts) {
            return;
        }
        device_guard dg(s->device);
        if (s->fp == RK_F32) {
            census_impl<float>(*s, p_begin, p_end, mac_value, counts);
        } else {
            census_impl<double>(*s, p_begin, p_end, mac_value, counts);
        }
    });
}

int rk_group_work(rk_state *s, double mac_value, uint64_t *work)
{
    return guard([&] {
        if (!s || (!work && s->n_crit)) {
            throw rk::error(RK_EINVAL, "null argument");
        }
        if (!std::isfinite(mac_value) || mac_value <= 0.) {
            throw rk::error(RK_EDOMAIN, "The transformed MAC value must be finite and positive, but it is "
                                            + std::to_string(mac_value) + " instead");
        }
        if (!s->nparts) {
            return;
        }
        device_guard dg(s->device);
        uint64_t counts[4];
        if (s->fp == RK_F32) {
            census_impl<float>(*s, 0, s->nparts, mac_value, counts, work);
        } else {
            census_impl<double>(*s, 0, s->nparts, mac_value, counts, work);
        }
    });
}

} // extern "C"
