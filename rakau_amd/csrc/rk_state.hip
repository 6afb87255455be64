// Host side of the rakau_amd C ABI: states -- their life cycle, creation from the reference's arrays or by the device builder --
// and the entry points that query them. The launch sequence of a traversal is in rk_launch.hip, host-array outputs in
// rk_host_out.hip, replication in rk_replica.hip.
#include "rk_state_internal.hpp"

namespace rkst
{


thread_local std::string g_err;

// RK_BACKTRACE=1: print the native call stack (module + offset; resolve with addr2line against the same build) when the
// process dies of SIGSEGV / SIGABRT / SIGBUS, then die the same way. A debugging aid for crashes that only show up in long runs.
void crash_handler(int sig)
{
    void *frames[64];
    const int n = backtrace(frames, 64);
    const char msg[] = "rakau_amd: fatal signal, native stack:\n";
    (void)!write(2, msg, sizeof(msg) - 1);
    backtrace_symbols_fd(frames, n, 2);
    signal(sig, SIG_DFL);
    raise(sig);
}
const bool g_crash_handler_installed = [] {
    const char *e = std::getenv("RK_BACKTRACE");
    if (e && std::atoi(e) != 0) {
        for (int sig : {SIGSEGV, SIGABRT, SIGBUS}) {
            signal(sig, crash_handler);
        }
        return true;
    }
    return false;
}();


std::atomic<int> g_build_exact{-1};

// Device ordinals of the C ABI are LOGICAL. Normally logical == physical. RK_ALIAS_DEVICES=<n> (a test knob) makes the
// library report n devices and maps logical device d onto physical device d % (physical count): the multi-device host
// logic of the callers (one state and one host thread per device, replication, range cuts) then runs on a box with a
// single GPU. Speed is meaningless in that mode; results are not affected.
int physical_device_count()
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n < 0) {
        return 0;
    }
    return n;
}
int alias_devices()
{
    static const int n = [] {
        const char *e = std::getenv("RK_ALIAS_DEVICES");
        const int v = e ? std::atoi(e) : 0;
        return v > 0 ? (v > 64 ? 64 : v) : 0;
    }();
    return n;
}
int logical_device_count()
{
    const int real = physical_device_count();
    return (real > 0 && alias_devices() > 0) ? alias_devices() : real;
}
int phys(int device)
{
    const int real = physical_device_count();
    return (real > 0 && alias_devices() > 0) ? device % real : device;
}

// Number of output arrays the CALLER passes: ndim accelerations, one potential, or both (tree_nvecs_res).
int user_nres(const rk_state &s, int q)
{
    return q == 0 ? s.ndim : (q == 1 ? 1 : s.ndim + 1);
}

// Pinned staging buffers of the host-output path outlive their state: hipHostMalloc / hipHostFree of 48 MB (4M particles) cost
// 10-20 ms each, which a caller that rebuilds its tree -- and with it the state -- every time step would pay per step (the
// reference re-creates its rocm_state after every update_particles()). A few buffers are parked (per physical device; at most
// four, the smallest that fits is handed out); rk_pool_trim() frees them.
std::mutex g_stage_mtx;
struct parked_stage {
    int dev;
    void *p;
    size_t bytes;
};
std::vector<parked_stage> g_stages;
void *stage_take(int dev, size_t need, size_t &got)
{
    std::lock_guard<std::mutex> lk(g_stage_mtx);
    size_t best = g_stages.size();
    for (size_t i = 0; i < g_stages.size(); ++i) {
        if (g_stages[i].dev == dev && g_stages[i].bytes >= need && g_stages[i].bytes <= 2 * need + (size_t(1) << 20)
            && (best == g_stages.size() || g_stages[i].bytes < g_stages[best].bytes)) {
            best = i;
        }
    }
    if (best == g_stages.size()) {
        return nullptr;
    }
    void *p = g_stages[best].p;
    got = g_stages[best].bytes;
    g_stages.erase(g_stages.begin() + static_cast<std::ptrdiff_t>(best));
    return p;
}
void stage_give(int dev, void *p, size_t bytes)
{
    if (!p) {
        return;
    }
    constexpr size_t keep = 4;
    void *drop = p;
    {
        std::lock_guard<std::mutex> lk(g_stage_mtx);
        if (keep) {
            g_stages.push_back(parked_stage{dev, p, bytes});
            drop = nullptr;
            if (g_stages.size() > keep) {
                drop = g_stages.front().p; // the oldest one goes
                g_stages.erase(g_stages.begin());
            }
        }
    }
    if (drop) {
        (void)hipHostFree(drop);
    }
}
void stage_trim()
{
    std::vector<parked_stage> v;
    {
        std::lock_guard<std::mutex> lk(g_stage_mtx);
        v.swap(g_stages);
    }
    for (auto &e : v) {
        (void)hipHostFree(e.p);
    }
}

// Give the tree-dependent device buffers back to the pool (after a device sync: traversal kernels on other
// streams may still be reading them) and forget everything derived from them. Streams, events and the output /
// supergroup scratch survive, so that a state can be rebuilt in place every time step.
std::vector<void *> take_retired_plan_buffers(); // (launch-plan buffers parked until the device is idle: see build_plan())
void release_tree(rk_state *s)
{
    // The snapshot is taken BEFORE the drain: a buffer another thread retires while this one is blocked in the synchronisation
    // may still be read by a kernel the synchronisation does not cover; it waits for the next drain.
    const std::vector<void *> retired = take_retired_plan_buffers();
    (void)hipDeviceSynchronize();
    for (void *b : retired) {
        rk::pool_free(b);
    }
    for (int i = 0; i < RK_NBUF; ++i) {
        rk::pool_free(s->buf[i]);
        s->buf[i] = nullptr;
        s->buf_bytes[i] = 0;
    }
    for (void **b : {&s->bld_codes, &s->bld_perm, &s->bld_node_code}) {
        rk::pool_free(*b);
        *b = nullptr;
    }
    drop_graph_exec(*s);                 // (releases the plans the cached graphs hold)
    s->plan = rk_state::launch_plan{}; // the device was synchronised above: the buffer goes back to the pool
    s->work_cache.clear();
    s->sup_b = s->sup_e = 0;
    s->plan_keys.clear();
    s->sl_rep_pending = false; // the device was synchronised above
    s->sl_clean_valid = false;
    s->first_order_valid = false;
    s->first_tail_valid = false;
}

void free_state(rk_state *s)
{
    if (!s) {
        return;
    }
    int prev = 0;
    (void)hipGetDevice(&prev);
    (void)hipSetDevice(phys(s->device));
    release_tree(s);
    park_class_graphs(*s);
    for (void *b : {s->d_out, s->sup_common, s->sup_resid, s->sup_cnt, s->z_scratch, s->sl_idx, s->sl_next,
                    s->sl_cnt, s->sl_ctl, s->sl_fb, s->sl_pbase, s->sl_part, s->first_order, s->first_tab}) {
        rk::pool_free(b);
    }
    if (s->sl_host) {
        (void)hipHostFree(s->sl_host);
    }
    if (s->sl_rep_ev) {
        (void)hipEventDestroy(s->sl_rep_ev);
    }
    stage_give(phys(s->device), s->h_stage, s->h_stage_bytes); // (release_tree above synchronised the device)
    if (s->ev0) {
        (void)hipEventDestroy(s->ev0);
    }
    if (s->ev1) {
        (void)hipEventDestroy(s->ev1);
    }
    if (s->ev_fork) {
        (void)hipEventDestroy(s->ev_fork);
    }
    if (s->sup_ev) {
        (void)hipEventDestroy(s->sup_ev);
    }
    if (s->ev_mid) {
        (void)hipEventDestroy(s->ev_mid);
    }
    if (s->ev_done) {
        (void)hipEventDestroy(s->ev_done);
    }
    for (auto &e : s->ev_arr) {
        if (e) {
            (void)hipEventDestroy(e);
        }
    }
    if (s->cap_stream) {
        (void)hipStreamDestroy(s->cap_stream);
    }
    for (int i = 0; i < rk::n_list_R; ++i) {
        if (s->ev_join[i]) {
            (void)hipEventDestroy(s->ev_join[i]);
        }
        if (s->aux_stream[i]) {
            (void)hipStreamDestroy(s->aux_stream[i]);
        }
    }
    (void)hipSetDevice(prev);
    delete s;
}


void alloc_upload(rk_state &s, int which, const void *host, size_t bytes)
{
    s.buf_bytes[which] = static_cast<int64_t>(bytes);
    if (!bytes) {
        return;
    }
    s.buf[which] = rk::pool_alloc(bytes);
    if (host) {
        RK_HIP(hipMemcpy(s.buf[which], host, bytes, hipMemcpyHostToDevice));
    }
}

// Host array without value-initialisation: the first touch of the pages happens in the (parallel) loop that fills it.
template <typename T>
struct raw_array {
    explicit raw_array(size_t n) : m_p(new T[n]), m_n(n) {}
    T *data()
    {
        return m_p.get();
    }
    const T *data() const
    {
        return m_p.get();
    }
    size_t size() const
    {
        return m_n;
    }
    T &operator[](size_t i)
    {
        return m_p[i];
    }
    const T &operator[](size_t i) const
    {
        return m_p[i];
    }

private:
    std::unique_ptr<T[]> m_p;
    size_t m_n;
};

// Run f(begin, end) over [0, n) on a few host threads (the conversions of rk_state_create are memory-bound loops).
template <typename Fn>
void host_parallel_for(size_t n, Fn &&f)
{
    const size_t grain = size_t(1) << 16;
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    const auto n_thr = static_cast<unsigned>(std::min<size_t>(std::min(hw, 8u), (n + grain - 1) / grain));
    if (n_thr <= 1) {
        f(size_t(0), n);
        return;
    }
    std::vector<std::thread> thr;
    std::exception_ptr ep;
    std::mutex m;
    for (unsigned t = 0; t < n_thr; ++t) {
        thr.emplace_back([&, t] {
            try {
                f(n * t / n_thr, n * (t + 1) / n_thr);
            } catch (...) {
                std::lock_guard<std::mutex> lk(m);
                ep = std::current_exception();
            }
        });
    }
    for (auto &t : thr) {
        t.join();
    }
    if (ep) {
        std::rethrow_exception(ep);
    }
}

// Build the host mirrors (group ranges, class lists) from the crit array.
void build_host_mirrors(rk_state &s, const std::vector<uint4> &crit)
{
    s.n_crit = static_cast<int64_t>(crit.size());
    s.crit_begin.resize(crit.size());
    s.crit_end.resize(crit.size());
    s.max_group = 0;
    for (int c = 0; c < rk::n_classes; ++c) {
        s.class_list[c].clear();
        s.class2_list[c].clear();
    }
    for (size_t i = 0; i < crit.size(); ++i) {
        s.crit_begin[i] = crit[i].x;
        s.crit_end[i] = crit[i].y;
        const int64_t size = static_cast<int64_t>(crit[i].y) - crit[i].x;
        s.max_group = std::max(s.max_group, size);
        s.class_list[rk::class_of(size)].push_back(static_cast<uint32_t>(i));
        s.class2_list[rk::class2_of(size)].push_back(static_cast<uint32_t>(i));
    }
    // Device layout of RK_BUF_CLASS: the variant 1 lists, then the variant 2 lists.
    s.class_off[0] = 0;
    for (int c = 0; c < rk::n_classes; ++c) {
        s.class_off[c + 1] = s.class_off[c] + static_cast<int64_t>(s.class_list[c].size());
    }
    s.class2_off[0] = s.class_off[rk::n_classes];
    for (int c = 0; c < rk::n_classes; ++c) {
        s.class2_count[c] = static_cast<int64_t>(s.class2_list[c].size());
        s.class2_off[c + 1] = s.class2_off[c] + s.class2_count[c];
    }
    s.mirrors_valid = true;
}

std::vector<uint32_t> concat_class_lists(const rk_state &s)
{
    std::vector<uint32_t> lists;
    for (const auto &l : s.class_list) {
        lists.insert(lists.end(), l.begin(), l.end());
    }
    for (const auto &l : s.class2_list) {
        lists.insert(lists.end(), l.begin(), l.end());
    }
    return lists;
}

// Fill the host mirrors of a device-built state (and the cross-check kernel's half of the class lists) on first use.
void ensure_mirrors(rk_state &s)
{
    if (s.mirrors_valid) {
        return;
    }
    std::vector<uint4> crit(static_cast<size_t>(s.n_crit));
    if (!crit.empty()) {
        RK_HIP(hipMemcpy(crit.data(), s.buf[RK_BUF_CRIT], crit.size() * sizeof(uint4), hipMemcpyDeviceToHost));
    }
    int64_t dev_off[rk::n_classes + 1];
    std::copy(s.class2_off, s.class2_off + rk::n_classes + 1, dev_off);
    build_host_mirrors(s, crit);
    if (!std::equal(dev_off, dev_off + rk::n_classes + 1, s.class2_off)) {
        throw rk::error(RK_ERUNTIME, "internal error: device and host binning of the critical nodes disagree");
    }
    const std::vector<uint32_t> lists = concat_class_lists(s);
    if (!lists.empty()) {
        RK_HIP(hipMemcpy(s.buf[RK_BUF_CLASS], lists.data(), lists.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    }
    s.mirrors_valid = true;
}

template <typename F>
void create_impl(rk_state &s, const void *const parts[4], int64_t nparts, const void *tree, int64_t tree_size,
                 int64_t node_stride)
{
    using v4 = typename rk::vt<F>::v4;
    using v2 = typename rk::vt<F>::v2;
    
#ifdef RK_BUILD_TIMING
    constexpr bool timing = true; // diagnostic build (-DRK_BUILD_TIMING): phase times on stderr
#else
    constexpr bool timing = false;
#endif
    const auto t_start = std::chrono::steady_clock::now();
    auto t_prev = t_start;
    const auto lap = [&](const char *what) {
        if (timing) {
            const auto t = std::chrono::steady_clock::now();
            std::fprintf(stderr, "RK_BUILD_TIMING create: %-22s %8.1f us\n", what,
                         std::chrono::duration<double, std::micro>(t - t_prev).count());
            t_prev = t;
        }
    };
    // Offsets inside rakau::tree_node_t<NDim, F, uint64_t, MAC> (tree_fwd.hpp:77-116 of the reference).
    const auto nd = static_cast<size_t>(s.ndim);
    constexpr size_t off_props = 5 * sizeof(uint64_t);
    const size_t off_dim = off_props + (nd + 1) * sizeof(F);
    const size_t min_stride = off_dim + (s.mac == RK_MAC_BH ? 1 : 2) * sizeof(F);
    if (node_stride < static_cast<int64_t>(min_stride)) {
        throw rk::error(RK_EINVAL, "node_stride (" + std::to_string(node_stride)
                                       + ") is smaller than the node record of the selected F/MAC ("
                                       + std::to_string(min_stride) + ")");
    }
    // parts = the ndim coordinate arrays, then the masses. Quadtrees live in the z = 0 plane of the 3-D kernels:
    // dz = 0 adds exactly nothing to any distance or acceleration.
    const auto *x = static_cast<const F *>(parts[0]), *y = static_cast<const F *>(parts[1]),
               *z = nd == 3 ? static_cast<const F *>(parts[2]) : nullptr, *m = static_cast<const F *>(parts[nd]);
    const auto n = static_cast<size_t>(nparts), nn = static_cast<size_t>(tree_size);

    raw_array<v4> part4(n);
    host_parallel_for(n, [&](size_t b, size_t e) {
        for (size_t i = b; i < e; ++i) {
            part4[i].x = x[i];
            part4[i].y = y[i];
            part4[i].z = z ? z[i] : F(0);
            part4[i].w = m[i];
        }
    });

    lap("particles -> AoS");
    raw_array<v4> com(nn);
    raw_array<v2> macp(nn);
    raw_array<uint4> topo(nn);
    const auto *base = static_cast<const unsigned char *>(tree);
    host_parallel_for(nn, [&](size_t b, size_t e) {
        for (size_t i = b; i < e; ++i) {
            const unsigned char *rec = base + i * static_cast<size_t>(node_stride);
            uint64_t hdr[5];
            std::memcpy(hdr, rec, sizeof(hdr));
            F props[4] = {F(0), F(0), F(0), F(0)}, dim[2] = {F(0), F(0)};
            std::memcpy(props, rec + off_props, (nd + 1) * sizeof(F));
            if (nd == 2) {
                props[3] = props[2]; // {x, y, mass} -> {x, y, 0, mass}
                props[2] = F(0);
            }
            std::memcpy(dim, rec + off_dim, (s.mac == RK_MAC_BH ? 1 : 2) * sizeof(F));
            const uint64_t begin = hdr[0], end = hdr[1], nch = hdr[2];
            if (begin >= end || end > static_cast<uint64_t>(nparts) || nch > nn - 1 - i) {
                throw rk::error(RK_EINVAL, "inconsistent tree node at index " + std::to_string(i));
            }
            com[i].x = props[0];
            com[i].y = props[1];
            com[i].z = props[2];
            com[i].w = props[3];
            macp[i].x = dim[0];
            macp[i].y = dim[1];
            topo[i].x = static_cast<uint32_t>(nch);
            topo[i].y = static_cast<uint32_t>(begin);
            topo[i].z = static_cast<uint32_t>(end);
        }
    });
    // Slot of every internal node in the child table (serial: a running count).
    size_t n_internal = 0;
    for (size_t i = 0; i < nn; ++i) {
        topo[i].w = topo[i].x ? static_cast<uint32_t>(n_internal++) : 0xffffffffu;
    }
    lap("node records -> SoA");
    // Child table: the indices of the (up to 8) children of every internal node. In the depth-first
    // layout the first child of node i is i + 1 and the next sibling of c is c + n_children(c) + 1
    // (tree.hpp:2783 of the reference).
    std::vector<uint32_t> child(n_internal * 8, 0u);
    for (size_t i = 0; i < nn; ++i) {
        if (!topo[i].x) {
            continue;
        }
        const size_t slot = topo[i].w, last = i + topo[i].x;
        size_t c = i + 1, k = 0;
        while (c <= last) {
            if (k >= 8) {
                throw rk::error(RK_EINVAL, "tree node " + std::to_string(i) + " has more than 8 children");
            }
            child[slot * 8 + k++] = static_cast<uint32_t>(c);
            c += static_cast<size_t>(topo[c].x) + 1;
        }
        if (c != last + 1) {
            throw rk::error(RK_EINVAL, "inconsistent children counts below tree node " + std::to_string(i));
        }
    }

    lap("child table");
    // Records for the list kernel in sibling order: record 0 is the root; walking the depth-first array,
    // every internal node gets the next free run of records for its children.
    raw_array<rk::node_rec<F>> recs(nn);
    {
        // Serial part: depth-first index -> record index (a running count over the internal nodes).
        std::vector<uint32_t> rec_of(nn, 0u);
        uint32_t next = nn ? 1u : 0u;
        for (size_t i = 0; i < nn; ++i) {
            if (topo[i].x) {
                const uint32_t *ch = &child[static_cast<size_t>(topo[i].w) * 8];
                uint32_t cnt = 0;
                while (cnt < 8 && ch[cnt]) {
                    rec_of[ch[cnt]] = next + cnt;
                    ++cnt;
                }
                next += cnt;
            }
        }
        // Parallel part: fill the records.
        host_parallel_for(nn, [&](size_t b, size_t e) {
            for (size_t i = b; i < e; ++i) {
                auto &r = recs[rec_of[i]];
                r.com = com[i];
                r.mac = macp[i];
                r.dfs = static_cast<uint32_t>(i);
                r.nch = topo[i].x;
                r.pad[0] = r.pad[1] = 0;
                if (topo[i].x) {
                    const uint32_t *ch = &child[static_cast<size_t>(topo[i].w) * 8];
                    uint32_t cnt = 0;
                    while (cnt < 8 && ch[cnt]) {
                        ++cnt;
                    }
                    r.a = rec_of[ch[0]];
                    r.b = cnt;
                } else {
                    r.a = topo[i].y;
                    r.b = topo[i].z;
                }
            }
        });
        if (nn && next != nn) {
            throw rk::error(RK_EINVAL, "inconsistent tree: not every node is reachable from the root");
        }
    }

    lap("sibling-order records");
    // Critical nodes: the first node on each root->leaf path with at most ncrit particles or without
    // children (equivalent to the rule at tree.hpp:801-803 of the reference: a node has no children
    // iff it holds at most max_leaf_n particles or sits at the deepest level).
    std::vector<uint4> crit;
    for (size_t i = 0; i < nn;) {
        const uint64_t np = static_cast<uint64_t>(topo[i].z) - topo[i].y;
        if (np <= s.ncrit || topo[i].x == 0) {
            uint4 c;
            c.x = topo[i].y;
            c.y = topo[i].z;
            c.z = static_cast<uint32_t>(i);
            c.w = static_cast<uint32_t>(np);
            crit.push_back(c);
            i += static_cast<size_t>(topo[i].x) + 1;
        } else {
            ++i;
        }
    }
    // The groups must tile [0, nparts).
    uint64_t expect = 0;
    for (const auto &c : crit) {
        if (c.x != expect) {
            throw rk::error(RK_EINVAL, "the critical nodes derived from the tree do not tile the particle range");
        }
        expect = c.y;
    }
    if (expect != static_cast<uint64_t>(nparts)) {
        throw rk::error(RK_EINVAL, "the critical nodes derived from the tree do not cover all particles");
    }

    lap("critical nodes");
    // Tight bounding boxes of the target groups (used by the list kernel to take most MAC decisions without
    // visiting every target).
    std::vector<v4> boxes(crit.size() * 2);
    host_parallel_for(crit.size(), [&](size_t gb, size_t ge) {
        for (size_t gi = gb; gi < ge; ++gi) {
            const v4 &p0 = part4[crit[gi].x];
            F lo[3] = {p0.x, p0.y, p0.z}, hi[3] = {lo[0], lo[1], lo[2]};
            for (size_t i = crit[gi].x; i < crit[gi].y; ++i) {
                const F pv[3] = {part4[i].x, part4[i].y, part4[i].z};
                for (int k = 0; k < 3; ++k) {
                    lo[k] = std::min(lo[k], pv[k]);
                    hi[k] = std::max(hi[k], pv[k]);
                }
            }
            boxes[2 * gi].x = lo[0], boxes[2 * gi].y = lo[1], boxes[2 * gi].z = lo[2], boxes[2 * gi].w = F(0);
            boxes[2 * gi + 1].x = hi[0], boxes[2 * gi + 1].y = hi[1], boxes[2 * gi + 1].z = hi[2],
                                   boxes[2 * gi + 1].w = F(0);
        }
    });
    lap("group boxes");
    build_host_mirrors(s, crit);
    s.n_internal = static_cast<int64_t>(n_internal);
    const std::vector<uint32_t> lists = concat_class_lists(s);
    lap("mirrors + class lists");

    alloc_upload(s, RK_BUF_PART4, part4.data(), part4.size() * sizeof(v4));
    alloc_upload(s, RK_BUF_NODE_COM, com.data(), com.size() * sizeof(v4));
    alloc_upload(s, RK_BUF_NODE_MAC, macp.data(), macp.size() * sizeof(v2));
    alloc_upload(s, RK_BUF_NODE_TOPO, topo.data(), topo.size() * sizeof(uint4));
    alloc_upload(s, RK_BUF_CRIT, crit.data(), crit.size() * sizeof(uint4));
    alloc_upload(s, RK_BUF_CHILD, child.data(), child.size() * sizeof(uint32_t));
    alloc_upload(s, RK_BUF_CLASS, lists.data(), lists.size() * sizeof(uint32_t));
    alloc_upload(s, RK_BUF_NODE_REC, recs.data(), recs.size() * sizeof(rk::node_rec<F>));
    alloc_upload(s, RK_BUF_CRIT_BOX, boxes.data(), boxes.size() * sizeof(v4));
    lap("uploads");
}

void check_common(int fp, int mac)
{
    if (fp != RK_F32 && fp != RK_F64) {
        throw rk::error(RK_EINVAL, "fp must be RK_F32 or RK_F64");
    }
    if (mac != RK_MAC_BH && mac != RK_MAC_BH_GEOM) {
        throw rk::error(RK_EINVAL, "mac must be RK_MAC_BH or RK_MAC_BH_GEOM");
    }
}

void check_ndim(int ndim)
{
    if (ndim != 2 && ndim != 3) {
        throw rk::error(RK_EINVAL, "ndim must be 2 (quadtree) or 3 (octree)");
    }
}

void check_device(int device)
{
    const int n = logical_device_count();
    if (n <= 0) {
        throw rk::error(RK_ERUNTIME, "no HIP device is available: the rakau_amd engine needs a gfx950 GPU");
    }
    if (device < 0 || device >= n) {
        throw rk::error(RK_EINVAL, "invalid device ordinal " + std::to_string(device) + " (" + std::to_string(n)
                                       + " devices visible)");
    }
}

} // namespace rkst

namespace rk
{
bool exact_node_sums()
{
    const int v = g_build_exact.load();
    if (v >= 0) {
        return v != 0;
    }
    static const bool env = [] {
        const char *e = std::getenv("RK_BUILD_EXACT");
        return e && std::atoi(e) != 0;
    }();
    return env;
}
} // namespace rk

extern "C" {

const char *rk_last_error(void)
{
    return g_err.c_str();
}

// Internal (not declared in the public header): lets rk_tree_capi.cpp report through rk_last_error().
RK_EXPORT void rk_set_last_error_(const char *msg)
{
    g_err = msg ? msg : "";
}

unsigned rk_min_size(void)
{
    // One wavefront of targets, like rocm_min_size() (src/rakau_rocm.cpp of the reference).
    return 64u;
}

int rk_device_count(void)
{
    return logical_device_count();
}

// Everything a first call would otherwise pay for, paid now: HIP runtime and device context, the code objects of every
// kernel family, one block of the device-memory cache. Optional -- a first call does the same lazily.
int rk_init(int device)
{
    return guard([&] {
        check_device(device);
        device_guard dg(device);
        RK_HIP(hipFree(nullptr));
        rk::touch_kernels();
        rk::touch_list();
        rk::touch_pc();
        rk::touch_split();
        rk::touch_build();
        rk::pool_free(rk::pool_alloc(size_t(1) << 20));
        RK_HIP(hipDeviceSynchronize());
    });
}

int rk_has_accelerator(void)
{
    const int n = physical_device_count();
    for (int i = 0; i < n; ++i) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, i) == hipSuccess && std::strncmp(prop.gcnArchName, "gfx950", 6) == 0) {
            return 1;
        }
    }
    return 0;
}

int rk_state_create(rk_state **out, int fp, int mac, int device, const void *const parts[4], const uint64_t *codes,
                    int64_t nparts, const void *tree, int64_t tree_size, int64_t node_stride, uint64_t ncrit)
{
    return rk_state_create_nd(out, 3, fp, mac, device, parts, codes, nparts, tree, tree_size, node_stride, ncrit);
}

int rk_state_create_nd(rk_state **out, int ndim, int fp, int mac, int device, const void *const *parts,
                       const uint64_t *codes, int64_t nparts, const void *tree, int64_t tree_size, int64_t node_stride,
                       uint64_t ncrit)
{
    (void)codes;
    return guard([&] {
        if (!out) {
            throw rk::error(RK_EINVAL, "null output pointer");
        }
        *out = nullptr;
        check_common(fp, mac);
        check_ndim(ndim);
        if (nparts < 0 || tree_size < 0) {
            throw rk::error(RK_EINVAL, "negative size");
        }
        if (nparts > 0 && (!parts || !parts[0] || !parts[1] || !parts[2] || (ndim == 3 && !parts[3]) || !tree
                           || tree_size == 0)) {
            throw rk::error(RK_EINVAL, "null particle or tree array");
        }
        if (static_cast<uint64_t>(nparts) >= 0xffffffffull || static_cast<uint64_t>(tree_size) >= 0xffffffffull) {
            throw rk::error(RK_EOVERFLOW, "The number of particles or tree nodes (" + std::to_string(nparts) + ", "
                                              + std::to_string(tree_size)
                                              + ") is too large for the 32-bit device indices");
        }
        if (!ncrit) {
            throw rk::error(RK_EINVAL, "ncrit must be nonzero");
        }
        if (static_cast<uint64_t>(tree_size) >= rk::max_list_nodes) {
            throw rk::error(RK_EOVERFLOW, "The number of tree nodes (" + std::to_string(tree_size)
                                              + ") exceeds the 2^29 limit of the traversal kernel's node references");
        }
        check_device(device);
        device_guard dg(device);
        state_ptr s(new rk_state);
        s->ndim = ndim;
        s->fp = fp;
        s->mac = mac;
        s->device = device;
        s->nparts = nparts;
        s->tree_size = tree_size;
        s->ncrit = ncrit;
        if (nparts > 0) {
            // The device buffers are derived from the caller's arrays on the device (rk_build.hip: convert_device; 4M fp32:
            // an order of magnitude faster than the host loops of create_impl, which RK_CREATE_ON_HOST=1 still selects -- the
            // two give the same buffers).
            static const bool on_host = [] {
                const char *e = std::getenv("RK_CREATE_ON_HOST");
                return e && std::atoi(e) != 0;
            }();
            const size_t min_stride = 5 * sizeof(uint64_t)
                                      + static_cast<size_t>(ndim + 1 + (mac == RK_MAC_BH ? 1 : 2)) * (fp == RK_F32 ? 4u : 8u);
            if (node_stride < static_cast<int64_t>(min_stride)) {
                throw rk::error(RK_EINVAL, "node_stride (" + std::to_string(node_stride)
                                               + ") is smaller than the node record of the selected F/MAC ("
                                               + std::to_string(min_stride) + ")");
            }
            if (on_host) {
                if (fp == RK_F32) {
                    create_impl<float>(*s, parts, nparts, tree, tree_size, node_stride);
                } else {
                    create_impl<double>(*s, parts, nparts, tree, tree_size, node_stride);
                }
            } else if (fp == RK_F32) {
                if (ndim == 3) {
                    rk::convert_device<float, 3>(*s, parts, nparts, tree, tree_size, node_stride);
                } else {
                    rk::convert_device<float, 2>(*s, parts, nparts, tree, tree_size, node_stride);
                }
            } else if (ndim == 3) {
                rk::convert_device<double, 3>(*s, parts, nparts, tree, tree_size, node_stride);
            } else {
                rk::convert_device<double, 2>(*s, parts, nparts, tree, tree_size, node_stride);
            }
            ensure_call_resources_any(*s);
        }
        *out = s.release();
    });
}

void rk_state_destroy(rk_state *s)
{
    free_state(s);
}

int rk_state_info(const rk_state *s, int64_t info[8])
{
    return guard([&] {
        if (!s || !info) {
            throw rk::error(RK_EINVAL, "null argument");
        }
        info[0] = s->nparts;
        info[1] = s->tree_size;
        info[2] = s->n_crit;
        info[3] = s->max_group;
        info[4] = s->fp;
        info[5] = s->mac;
        info[6] = s->device;
        info[7] = static_cast<int64_t>(s->ncrit);
    });
}

int rk_state_crit_ranges(const rk_state *s, int64_t *begin_end)
{
    return guard([&] {
        if (!s || !begin_end) {
            throw rk::error(RK_EINVAL, "null argument");
        }
        device_guard dg(s->device);
        ensure_mirrors(*const_cast<rk_state *>(s));
        for (int64_t i = 0; i < s->n_crit; ++i) {
            begin_end[2 * i] = s->crit_begin[static_cast<size_t>(i)];
            begin_end[2 * i + 1] = s->crit_end[static_cast<size_t>(i)];
        }
    });
}

int rk_state_set_timing(rk_state *s, int on)
{
    return guard([&] {
        if (!s) {
            throw rk::error(RK_EINVAL, "null state");
        }
        s->timing = on != 0;
    });
}

int rk_last_kernel_ms(rk_state *s, float *ms)
{
    return guard([&] {
        if (!s || !ms) {
            throw rk::error(RK_EINVAL, "null argument");
        }
        if (!s->timed) {
            throw rk::error(RK_EINVAL, "no timed traversal has been run on this state (rk_state_set_timing)");
        }
        device_guard dg(s->device);
        RK_HIP(hipEventSynchronize(s->ev1));
        RK_HIP(hipEventElapsedTime(ms, s->ev0, s->ev1));
    });
}

// Run the device build into `s` (fp, mac, device, ncrit, max_leaf_n already set; no tree buffers held).
static void fill_from_build(rk_state &s, const void *const parts[4], bool on_device, int64_t nparts, double box_size)
{
    s.nparts = nparts;
    s.tree_size = 0;
    s.n_crit = 0;
    s.box_size = box_size;
    s.box_deduced = box_size == 0.;
    std::vector<uint4> crit;
    
#ifdef RK_BUILD_TIMING
    constexpr bool timing = true; // diagnostic build (-DRK_BUILD_TIMING): phase times on stderr
#else
    constexpr bool timing = false;
#endif
    const auto now = [] { return std::chrono::steady_clock::now(); };
    const auto t0 = now();
    auto t1 = t0, t2 = t0;
    if (nparts > 0) {
        std::string msg;
        if (s.fp == RK_F32) {
            if (s.ndim == 3) {
                rk::build_device<float, 3>(s, parts, on_device, nparts, box_size, s.max_leaf_n, msg);
            } else {
                rk::build_device<float, 2>(s, parts, on_device, nparts, box_size, s.max_leaf_n, msg);
            }
        } else {
            if (s.ndim == 3) {
                rk::build_device<double, 3>(s, parts, on_device, nparts, box_size, s.max_leaf_n, msg);
            } else {
                rk::build_device<double, 2>(s, parts, on_device, nparts, box_size, s.max_leaf_n, msg);
            }
        }
    }
    t1 = t2 = now();
    if (nparts == 0) {
        build_host_mirrors(s, crit); // empty tree: empty, valid mirrors
    }
    const auto t3 = now();
    if (timing) {
        const auto us = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
        std::fprintf(stderr, "RK_BUILD_TIMING n=%lld: device build %.0f us, crit download %.0f us, host mirrors %.0f us, "
                             "class lists + upload %.0f us\n",
                     static_cast<long long>(nparts), us(t0, t1), us(t1, t2), us(t2, t3), us(t3, now()));
    }
}

static void check_build_args(const void *const *parts, int ndim, int64_t nparts, double box_size)
{
    if (nparts < 0 || (nparts > 0 && (!parts || !parts[0] || !parts[1] || !parts[2] || (ndim == 3 && !parts[3])))) {
        throw rk::error(RK_EINVAL, "null particle array");
    }
    if (static_cast<uint64_t>(nparts) >= 0x7fffffffull) {
        throw rk::error(RK_EOVERFLOW, "The number of particles (" + std::to_string(nparts)
                                          + ") is too large for the 32-bit device indices");
    }
    // Parameter checks and messages of tree.hpp:1350-1362 of the reference.
    if (!std::isfinite(box_size) || box_size < 0.) {
        throw rk::error(RK_EINVAL, "The box size must be a finite non-negative value, but it is "
                                       + std::to_string(box_size) + " instead");
    }
}

static int state_build_impl(rk_state **out, int ndim, int fp, int mac, int device, const void *const *parts,
                            bool on_device, int64_t nparts, double box_size, uint64_t max_leaf_n, uint64_t ncrit)
{
    return guard([&] {
        if (!out) {
            throw rk::error(RK_EINVAL, "null output pointer");
        }
        *out = nullptr;
        check_common(fp, mac);
        check_ndim(ndim);
        check_build_args(parts, ndim, nparts, box_size);
        if (!max_leaf_n) {
            throw rk::error(RK_EINVAL, "The maximum number of particles per leaf must be nonzero");
        }
        if (!ncrit) {
            throw rk::error(RK_EINVAL, "The critical number of particles for the vectorised computation of the "
                                       "potentials/accelerations must be nonzero");
        }
        check_device(device);
        device_guard dg(device);
        state_ptr s(new rk_state);
        s->ndim = ndim;
        s->fp = fp;
        s->mac = mac;
        s->device = device;
        s->ncrit = ncrit;
        s->max_leaf_n = max_leaf_n;
        fill_from_build(*s, parts, on_device, nparts, box_size);
        ensure_call_resources_any(*s);
        *out = s.release();
    });
}

int rk_state_build(rk_state **out, int fp, int mac, int device, const void *const parts[4], int64_t nparts,
                   double box_size, uint64_t max_leaf_n, uint64_t ncrit)
{
    return state_build_impl(out, 3, fp, mac, device, parts, false, nparts, box_size, max_leaf_n, ncrit);
}

int rk_state_build_nd(rk_state **out, int ndim, int fp, int mac, int device, const void *const *parts, int on_device,
                      int64_t nparts, double box_size, uint64_t max_leaf_n, uint64_t ncrit)
{
    return state_build_impl(out, ndim, fp, mac, device, parts, on_device != 0, nparts, box_size, max_leaf_n, ncrit);
}

int rk_state_ndim(const rk_state *s)
{
    return s ? s->ndim : 0;
}

int rk_state_build_device(rk_state **out, int fp, int mac, int device, const void *const d_parts[4], int64_t nparts,
                          double box_size, uint64_t max_leaf_n, uint64_t ncrit)
{
    return state_build_impl(out, 3, fp, mac, device, d_parts, true, nparts, box_size, max_leaf_n, ncrit);
}

int rk_state_rebuild_device(rk_state *s, const void *const d_parts[4], int64_t nparts, double box_size)
{
    return guard([&] {
        if (!s) {
            throw rk::error(RK_EINVAL, "null state");
        }
        check_build_args(d_parts, s->ndim, nparts, box_size);
        device_guard dg(s->device);
        release_tree(s);
        try {
            fill_from_build(*s, d_parts, true, nparts, box_size);
        } catch (...) {
            // Leave an empty but valid state behind.
            release_tree(s);
            s->nparts = 0, s->tree_size = 0;
            build_host_mirrors(*s, {});
            throw;
        }
    });
}

void host_blocks_trim();
void rk_pool_trim(void)
{
    rk::pool_trim();
    stage_trim();
    host_blocks_trim();
}

void rk_set_build_exact(int on)
{
    g_build_exact.store(on ? 1 : 0);
}

int rk_state_set_perm(rk_state *s, const uint64_t *perm)
{
    return guard([&] {
        if (!s || (!perm && s->nparts)) {
            throw rk::error(RK_EINVAL, "null argument");
        }
        device_guard dg(s->device);
        // A traversal still in flight (any stream) may be reading the old permutation, and a captured launch sequence
        // must not outlive the buffer it was recorded with.
        RK_HIP(hipDeviceSynchronize());
        drop_graph_exec(*s);
        std::vector<uint32_t> p32(static_cast<size_t>(s->nparts));
        for (size_t i = 0; i < p32.size(); ++i) {
            if (perm[i] >= static_cast<uint64_t>(s->nparts)) {
                throw rk::error(RK_EINVAL, "invalid permutation entry");
            }
            p32[i] = static_cast<uint32_t>(perm[i]);
        }
        if (!s->bld_perm && !p32.empty()) {
            s->bld_perm = rk::pool_alloc(p32.size() * sizeof(uint32_t));
        }
        if (!p32.empty()) {
            RK_HIP(hipMemcpy(s->bld_perm, p32.data(), p32.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        }
    });
}

int rk_state_device_ptr(const rk_state *s, int what, void **ptr, int64_t *bytes)
{
    return guard([&] {
        if (!s || !ptr || !bytes) {
            throw rk::error(RK_EINVAL, "null argument");
        }
        const size_t n = static_cast<size_t>(s->nparts);
        switch (what) {
            case 0: *ptr = s->buf[RK_BUF_PART4], *bytes = s->buf_bytes[RK_BUF_PART4]; break;
            case 1: *ptr = s->bld_perm, *bytes = s->bld_perm ? static_cast<int64_t>(n * sizeof(uint32_t)) : 0; break;
            case 2: *ptr = s->bld_codes, *bytes = s->bld_codes ? static_cast<int64_t>(n * sizeof(uint64_t)) : 0; break;
            case 3: { // launch order of the first call (diagnostic): critical-node indices, uint32 -- heavy-first on a small tree,
                      // the queues of the light-tail arrangement on a large one
                const bool have = (s->first_order_valid || s->first_tail_valid) && s->first_order;
                *ptr = have ? s->first_order : nullptr;
                *bytes = have ? (s->class2_off[RK_MAX_R] - s->class2_off[0]) * static_cast<int64_t>(sizeof(uint32_t)) : 0;
                break;
            }
            case 4: { // queue table of the first-call order (FIRST_TAB_WORDS uint32; null unless the tree came with one)
                const bool have = (s->first_tail_valid || s->first_order_valid) && s->first_tab;
                *ptr = have ? s->first_tab : nullptr;
                *bytes = have ? static_cast<int64_t>(rk::FIRST_TAB_WORDS * sizeof(uint32_t)) : 0;
                break;
            }
            default: throw rk::error(RK_EINVAL, "invalid selector for rk_state_device_ptr");
        }
    });
}

int rk_state_tree_info(const rk_state *s, double *box_size, int64_t info[4])
{
    return guard([&] {
        if (!s || !box_size || !info) {
            throw rk::error(RK_EINVAL, "null argument");
        }
        *box_size = s->box_size;
        info[0] = s->box_deduced;
        info[1] = static_cast<int64_t>(s->max_leaf_n);
        info[2] = s->bld_codes != nullptr;
        info[3] = s->n_internal;
    });
}

int rk_state_download(const rk_state *s, int what, void *dst)
{
    return guard([&] {
        if (!s || !dst) {
            throw rk::error(RK_EINVAL, "null argument");
        }
        if (!s->nparts) {
            return;
        }
        device_guard dg(s->device);
        const size_t fsz = s->fp == RK_F32 ? 4 : 8, n = static_cast<size_t>(s->nparts),
                     nn = static_cast<size_t>(s->tree_size);
        auto fetch = [&](const void *dev, size_t bytes) {
            std::vector<unsigned char> h(bytes);
            RK_HIP(hipMemcpy(h.data(), dev, bytes, hipMemcpyDeviceToHost));
            return h;
        };
        if (what >= 0 && what <= 3) {
            if (what == 2 && s->ndim == 2) {
                throw rk::error(RK_EINVAL, "a quadtree has no z coordinates");
            }
            const auto h = fetch(s->buf[RK_BUF_PART4], n * 4 * fsz);
            for (size_t i = 0; i < n; ++i) {
                std::memcpy(static_cast<unsigned char *>(dst) + i * fsz, h.data() + (i * 4 + static_cast<size_t>(what)) * fsz,
                            fsz);
            }
            return;
        }
        if (what == 8) {
            // Particles as stored: {x, y, z, m} records in Morton order.
            RK_HIP(hipMemcpy(dst, s->buf[RK_BUF_PART4], n * 4 * fsz, hipMemcpyDeviceToHost));
            return;
        }
        if (((what == 4 || what == 6 || what == 7) && !s->bld_codes) || (what == 5 && !s->bld_perm)) {
            throw rk::error(RK_EINVAL, "this state was created from a host tree: codes, permutation and nodal codes "
                                       "live in the caller's tree");
        }
        if (what == 4) {
            RK_HIP(hipMemcpy(dst, s->bld_codes, n * sizeof(uint64_t), hipMemcpyDeviceToHost));
        } else if (what == 5) {
            const auto h = fetch(s->bld_perm, n * sizeof(uint32_t));
            auto *o = static_cast<uint64_t *>(dst);
            for (size_t i = 0; i < n; ++i) {
                uint32_t v;
                std::memcpy(&v, h.data() + i * 4, 4);
                o[i] = v;
            }
        } else if (what == 6) {
            // Node array in the reference's record layout (tree_fwd.hpp:77-116): begin, end, n_children, code, level,
            // props[4], dim2 | dim, delta.
            const auto topo = fetch(s->buf[RK_BUF_NODE_TOPO], nn * sizeof(uint4));
            const auto com = fetch(s->buf[RK_BUF_NODE_COM], nn * 4 * fsz);
            const auto macp = fetch(s->buf[RK_BUF_NODE_MAC], nn * 2 * fsz);
            const auto code = fetch(s->bld_node_code, nn * sizeof(uint64_t));
            const auto nd = static_cast<size_t>(s->ndim);
            const size_t off_props = 40, off_dim = off_props + (nd + 1) * fsz;
            const size_t stride = ((off_dim + (s->mac == RK_MAC_BH ? 1 : 2) * fsz + 7) / 8) * 8;
            auto *o = static_cast<unsigned char *>(dst);
            std::memset(o, 0, nn * stride);
            for (size_t i = 0; i < nn; ++i) {
                uint4 t;
                std::memcpy(&t, topo.data() + i * sizeof(uint4), sizeof(uint4));
                uint64_t c;
                std::memcpy(&c, code.data() + i * 8, 8);
                const uint64_t hdr[5]
                    = {t.y, t.z, t.x, c, (63u - static_cast<unsigned>(__builtin_clzll(c))) / static_cast<unsigned>(nd)};
                std::memcpy(o + i * stride, hdr, sizeof(hdr));
                // Device record {x, y, z, mass}; a quadtree's props are {x, y, mass}.
                std::memcpy(o + i * stride + off_props, com.data() + i * 4 * fsz, nd * fsz);
                std::memcpy(o + i * stride + off_props + nd * fsz, com.data() + (i * 4 + 3) * fsz, fsz);
                std::memcpy(o + i * stride + off_dim, macp.data() + i * 2 * fsz, (s->mac == RK_MAC_BH ? 1 : 2) * fsz);
            }
        } else if (what == 7) {
            // Critical nodes as {code, begin, end} triples.
            const auto code = fetch(s->bld_node_code, nn * sizeof(uint64_t));
            std::vector<uint4> crit(static_cast<size_t>(s->n_crit));
            RK_HIP(hipMemcpy(crit.data(), s->buf[RK_BUF_CRIT], crit.size() * sizeof(uint4), hipMemcpyDeviceToHost));
            auto *o = static_cast<uint64_t *>(dst);
            for (size_t g = 0; g < crit.size(); ++g) {
                uint64_t c;
                std::memcpy(&c, code.data() + static_cast<size_t>(crit[g].z) * 8, 8);
                o[3 * g] = c;
                o[3 * g + 1] = crit[g].x;
                o[3 * g + 2] = crit[g].y;
            }
        } else {
            throw rk::error(RK_EINVAL, "invalid selector for rk_state_download");
        }
    });
}

int rk_device_memcpy(void *dst, const void *src, int64_t bytes, int device)
{
    return guard([&] {
        if (bytes < 0 || (bytes > 0 && (!dst || !src))) {
            throw rk::error(RK_EINVAL, "invalid arguments to rk_device_memcpy");
        }
        check_device(device);
        device_guard dg(device);
        if (bytes) {
            RK_HIP(hipMemcpy(dst, src, static_cast<size_t>(bytes), hipMemcpyDeviceToDevice));
        }
    });
}

int rk_state_graph_stats(const rk_state *s, int64_t stats[6])
{
    return guard([&] {
        if (!s || !stats) {
            throw rk::error(RK_EINVAL, "null argument");
        }
        for (int i = 0; i < 4; ++i) {
            stats[i] = static_cast<int64_t>(s->graph_stats[i]);
        }
        stats[4] = static_cast<int64_t>(s->gcache.size());
        stats[5] = g_forked_execs.load(std::memory_order_relaxed);
    });
}

int rk_set_kernel_variant(rk_state *s, int variant)
{
    return guard([&] {
        if (!s || variant < 0 || variant > 4) {
            throw rk::error(RK_EINVAL, "invalid kernel variant");
        }
        if (variant == 1 || variant == 4) {
            (void)rk::xcheck(); // the cross-check kernels live in librakau_amd_xcheck.so: load it now, or say why not
        }
        s->variant = variant;
    });
}

} // extern "C"
