// Host side of the rakau_amd C ABI: state creation / replication and the acc_pot entry points.
#include "rk_common.hpp"
#include "rk_xcheck.hpp"

#include <dlfcn.h>
#include <functional>
#include <deque>
#include <condition_variable>
#include <execinfo.h>
#include <signal.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <memory>
#include <map>
#include <mutex>
#include <unordered_map>
#include <thread>
#include <atomic>

namespace
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
std::atomic<int> g_build_exact{-1};

// Layout tag of rk_state_export / rk_state_import ("rk04"): bump it whenever the buffer list or the meta block changes.
constexpr int64_t state_layout_tag = 0x726b3034;

template <typename Fn>
int guard(Fn &&f) noexcept
{
    try {
        f();
        return RK_OK;
    } catch (const rk::error &e) {
        g_err = e.what();
        return e.code;
    } catch (const std::bad_alloc &) {
        g_err = "out of host memory";
        return RK_ENOMEM;
    } catch (const std::exception &e) {
        g_err = e.what();
        return RK_ERUNTIME;
    }
}

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

struct device_guard {
    int prev = 0;
    explicit device_guard(int dev)
    {
        RK_HIP(hipGetDevice(&prev));
        cur = phys(dev);
        if (prev != cur) {
            RK_HIP(hipSetDevice(cur));
        }
    }
    ~device_guard()
    {
        if (prev != cur) {
            (void)hipSetDevice(prev);
        }
    }
    int cur = 0;
};

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

struct state_deleter {
    void operator()(rk_state *s) const
    {
        free_state(s);
    }
};
using state_ptr = std::unique_ptr<rk_state, state_deleter>;

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
    s.plan.off_all = s.plan.n_all = s.plan.off_oth = s.plan.n_oth = s.plan.off_123 = s.plan.n_123 = 0;
    if (lpt) {
        // Merged heavy-first lists over the wave-kernel classes (stable: equal weights keep class, then Morton order): all
        // of them, all but R = 2, all but R = 4.
        for (int pass = 0; pass < 3; ++pass) {
            const auto first = static_cast<std::ptrdiff_t>(lists.size());
            for (int c = 0; c < RK_MAX_R; ++c) {
                if ((pass == 1 && c == 1) || (pass == 2 && c == 3)) {
                    continue;
                }
                // (copied first: inserting a range of a vector into itself is undefined once it reallocates)
                const std::vector<uint32_t> part(lists.begin() + s.plan.off[c], lists.begin() + s.plan.off[c + 1]);
                lists.insert(lists.end(), part.begin(), part.end());
            }
            std::stable_sort(lists.begin() + first, lists.end(),
                             [&](uint32_t a, uint32_t b2) { return s.work_cache[a] > s.work_cache[b2]; });
            (pass == 0 ? s.plan.off_all : (pass == 1 ? s.plan.off_oth : s.plan.off_123)) = first;
            (pass == 0 ? s.plan.n_all : (pass == 1 ? s.plan.n_oth : s.plan.n_123)) = static_cast<int64_t>(lists.size()) - first;
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
              double eps2, int offset_output, hipStream_t stream, bool allow_graph = true)
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
        // one Morton slice per XCD) 1.48 -> 1.42, 4M 1.92 -> 1.85, 6M 2.88 -> 2.80; the rebuild pays 20-30 us for it.
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
                if (s.first_order_valid && s.first_order) {
                    // A small tree built (or converted) on the device comes with the order of a heavy-first plan, nodes by
                    // decreasing size (rk_build.hip k_first_order): 100k particles 0.144 -> 0.10 ms on k_pc_any.
                    first_list = static_cast<const uint32_t *>(s.first_order);
                    p.any_rev = 0;
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
            if (forked) {
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
                    rk::launch_pc_any<F>(s, q, p, pl + s.plan.off_all, s.plan.n_all, streams[0]);
                } else if (any_mode == 2) {
                    rk::launch_pc<F>(s, q, p, cb, ce, streams, 0x2u);
                    rk::launch_list_any<F>(s, q, p, pl + s.plan.off_oth, s.plan.n_oth, streams[0]);
                } else if (any_mode == 4) {
                    // The R = 4 class on its own kernel (first: its nodes are the longest), the rest on a k_list_any compiled for
                    // the registers of R = 3.
                    rk::launch_list<F>(s, q, p, cb, ce, streams, 0x8u);
                    rk::launch_list_any<F>(s, q, p, pl + s.plan.off_123, s.plan.n_123, streams[0], 3);
                } else {
                    rk::launch_list_any<F>(s, q, p, pl + s.plan.off_all, s.plan.n_all, streams[0]);
                }
            } else {
                if (pc_mask) {
                    rk::launch_pc<F>(s, q, p, cb, ce, streams, pc_mask);
                }
                if (pc_mask != 0xfu) {
                    rk::launch_list<F>(s, q, p, cb, ce, streams, ~pc_mask);
                }
            }
            if (forked) {
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

} // namespace

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

// ---- one process per GPU: the replicate step over RCCL (the collectives library is bound at run time) ----
namespace
{
struct rccl_api {
    using result_t = int;
    struct unique_id {
        char internal[128];
    };
    result_t (*get_unique_id)(unique_id *) = nullptr;
    result_t (*comm_init_rank)(void **, int, unique_id, int) = nullptr;
    result_t (*comm_destroy)(void *) = nullptr;
    result_t (*broadcast)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    result_t (*group_start)() = nullptr;
    result_t (*group_end)() = nullptr;
    const char *(*error_string)(result_t) = nullptr;
    bool ok = false;
    std::string why;
};

const rccl_api &rccl()
{
    static const rccl_api api = [] {
        rccl_api a;
        // The copy already mapped into the process (PyTorch-ROCm brings its own under the same soname), else the system's.
        void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
        if (!h) {
            h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
        }
        if (!h) {
            h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
        }
        if (!h) {
            a.why = std::string("cannot load librccl.so.1: ") + dlerror();
            return a;
        }
        auto sym = [&](const char *name) {
            void *p = dlsym(h, name);
            if (!p && a.why.empty()) {
                a.why = std::string("librccl.so.1 lacks ") + name;
            }
            return p;
        };
        a.get_unique_id = reinterpret_cast<decltype(a.get_unique_id)>(sym("ncclGetUniqueId"));
        a.comm_init_rank = reinterpret_cast<decltype(a.comm_init_rank)>(sym("ncclCommInitRank"));
        a.comm_destroy = reinterpret_cast<decltype(a.comm_destroy)>(sym("ncclCommDestroy"));
        a.broadcast = reinterpret_cast<decltype(a.broadcast)>(sym("ncclBroadcast"));
        a.group_start = reinterpret_cast<decltype(a.group_start)>(sym("ncclGroupStart"));
        a.group_end = reinterpret_cast<decltype(a.group_end)>(sym("ncclGroupEnd"));
        a.error_string = reinterpret_cast<decltype(a.error_string)>(sym("ncclGetErrorString"));
        a.ok = a.why.empty();
        return a;
    }();
    if (!api.ok) {
        throw rk::error(RK_ERUNTIME, "RCCL is not available: " + api.why);
    }
    return api;
}

void rccl_check(int r, const char *what)
{
    if (r != 0) {
        throw rk::error(RK_ERUNTIME, std::string("RCCL call failed: ") + what + ": " + rccl().error_string(r));
    }
}
} // namespace

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

namespace
{

// Device-side address of [p, p + bytes) if the whole range is host memory the device can write to (hipHostMalloc /
// hipHostRegister: rk_host_alloc(), a pinned torch tensor, a user's registered vector), nullptr for pageable memory.
void *device_view_of_host_range(void *p, size_t bytes)
{
    if (!p || !bytes) {
        return nullptr;
    }
    hipPointerAttribute_t a0{}, a1{};
    if (hipPointerGetAttributes(&a0, p) != hipSuccess
        || hipPointerGetAttributes(&a1, static_cast<unsigned char *>(p) + bytes - 1u) != hipSuccess) {
        (void)hipGetLastError(); // pageable memory is reported as an error: clear it
        return nullptr;
    }
    if (a0.type != hipMemoryTypeHost || a1.type != hipMemoryTypeHost || !a0.devicePointer || !a1.devicePointer) {
        return nullptr;
    }
    // One registration: the device view is contiguous over the range.
    if (static_cast<unsigned char *>(a1.devicePointer) - static_cast<unsigned char *>(a0.devicePointer)
        != static_cast<ptrdiff_t>(bytes - 1u)) {
        return nullptr;
    }
    return a0.devicePointer;
}

// memcpy with non-temporal stores for the 16-byte aligned body of the destination (movntdq on the host): the delivery of a
// staged result overwrites whole cache lines that nobody reads soon.
inline void stream_copy(unsigned char *d, const unsigned char *src, size_t n)
{
    typedef long long v2di __attribute__((vector_size(16)));
    typedef long long v2di_u __attribute__((vector_size(16), aligned(1)));
    if (n < 256) {
        std::memcpy(d, src, n);
        return;
    }
    const size_t head = (16 - (reinterpret_cast<uintptr_t>(d) & 15)) & 15;
    std::memcpy(d, src, head);
    d += head;
    src += head;
    n -= head;
    const size_t body = n & ~size_t(63);
    for (size_t i = 0; i < body; i += 64) {
        const v2di a = *reinterpret_cast<const v2di_u *>(src + i), b = *reinterpret_cast<const v2di_u *>(src + i + 16),
                   c = *reinterpret_cast<const v2di_u *>(src + i + 32), e = *reinterpret_cast<const v2di_u *>(src + i + 48);
        __builtin_nontemporal_store(a, reinterpret_cast<v2di *>(d + i));
        __builtin_nontemporal_store(b, reinterpret_cast<v2di *>(d + i + 16));
        __builtin_nontemporal_store(c, reinterpret_cast<v2di *>(d + i + 32));
        __builtin_nontemporal_store(e, reinterpret_cast<v2di *>(d + i + 48));
    }
    std::memcpy(d + body, src + body, n - body);
}

// The host threads that deliver staged results: created once (a thread costs ~20 us to start and to join, seven of them
// per call were 5% of a 4M-particle call), parked on a condition variable between calls. Several callers (the device
// threads of a multi-device split) may post jobs at the same time; every caller also works on its own job.
class delivery_pool
{
    struct job {
        const std::function<void(int)> *fn;
        int n_items, max_helpers;
        std::atomic<int> next{0}, done{0}, helpers{0};
    };
    std::mutex m_;
    std::condition_variable cv_;
    std::vector<std::thread> thr_;
    std::deque<std::shared_ptr<job>> jobs_;
    bool stop_ = false;

    static void work(job &j)
    {
        for (;;) {
            const int item = j.next.fetch_add(1, std::memory_order_relaxed);
            if (item >= j.n_items) {
                return;
            }
            (*j.fn)(item);
            j.done.fetch_add(1, std::memory_order_release);
        }
    }
    void loop()
    {
        for (;;) {
            std::shared_ptr<job> j;
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&] {
                    while (!jobs_.empty() && jobs_.front()->next.load(std::memory_order_relaxed) >= jobs_.front()->n_items) {
                        jobs_.pop_front();
                    }
                    return stop_ || !jobs_.empty();
                });
                if (stop_) {
                    return;
                }
                j = jobs_.front();
                if (j->helpers.fetch_add(1) >= j->max_helpers) { // enough hands on this one: look at the next, or sleep
                    j.reset();
                    for (auto &o : jobs_) {
                        if (o->next.load(std::memory_order_relaxed) < o->n_items && o->helpers.fetch_add(1) < o->max_helpers) {
                            j = o;
                            break;
                        }
                    }
                    if (!j) {
                        cv_.wait_for(lk, std::chrono::microseconds(200));
                        continue;
                    }
                }
            }
            work(*j);
        }
    }

public:
    static delivery_pool &get()
    {
        static delivery_pool p;
        return p;
    }
    ~delivery_pool()
    {
        {
            std::lock_guard<std::mutex> lk(m_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto &t : thr_) {
            t.join();
        }
    }
    // fn(item) for item in [0, n_items), on the caller's thread and up to n_thr - 1 pool threads; returns when all are done.
    void run(int n_items, int n_thr, const std::function<void(int)> &fn)
    {
        auto j = std::make_shared<job>();
        j->fn = &fn;
        j->n_items = n_items;
        j->max_helpers = n_thr - 1;
        if (n_thr > 1) {
            {
                std::lock_guard<std::mutex> lk(m_);
                while (static_cast<int>(thr_.size()) < n_thr - 1) {
                    thr_.emplace_back([this] { loop(); });
                }
                jobs_.push_back(j);
            }
            cv_.notify_all();
        }
        work(*j);
        while (j->done.load(std::memory_order_acquire) < n_items) {
            std::this_thread::yield();
        }
        // Workers that still hold the job only look at its counters (the shared_ptr keeps them alive); fn is not called again.
    }
};

} // namespace

// rk_host_alloc() / rk_host_free(): pinned, device-visible host memory. Blocks of 1 MiB and more that are freed are parked (at most
// eight of them, 512 MiB in all) and handed out again to requests they fit within a factor of two: pinning costs milliseconds per
// 16 MiB, and the C++ header's staged overloads give their buffers back after every call on a large tree (tree.hpp,
// stage_buffers::trim) instead of keeping them per thread for the life of the process. rk_pool_trim() releases what is parked.
namespace
{
std::mutex g_host_mtx;
std::unordered_map<void *, size_t> g_host_live; // blocks handed out by rk_host_alloc -> their size
struct parked_host {
    void *p;
    size_t bytes;
};
std::vector<parked_host> g_host_parked;
size_t g_host_parked_bytes = 0;
} // namespace

void host_blocks_trim()
{
    std::vector<parked_host> v;
    {
        std::lock_guard<std::mutex> lk(g_host_mtx);
        v.swap(g_host_parked);
        g_host_parked_bytes = 0;
    }
    for (auto &e : v) {
        (void)hipHostFree(e.p);
    }
}

int rk_host_alloc(void **ptr, int64_t bytes)
{
    return guard([&] {
        if (!ptr || bytes < 0) {
            throw rk::error(RK_EINVAL, "rk_host_alloc: null pointer or negative size");
        }
        *ptr = nullptr;
        if (!bytes) {
            return;
        }
        const auto need = static_cast<size_t>(bytes);
        {
            std::lock_guard<std::mutex> lk(g_host_mtx);
            size_t best = g_host_parked.size();
            for (size_t i = 0; i < g_host_parked.size(); ++i) {
                if (g_host_parked[i].bytes >= need && g_host_parked[i].bytes <= 2 * need
                    && (best == g_host_parked.size() || g_host_parked[i].bytes < g_host_parked[best].bytes)) {
                    best = i;
                }
            }
            if (best != g_host_parked.size()) {
                *ptr = g_host_parked[best].p;
                g_host_live.emplace(*ptr, g_host_parked[best].bytes);
                g_host_parked_bytes -= g_host_parked[best].bytes;
                g_host_parked.erase(g_host_parked.begin() + static_cast<std::ptrdiff_t>(best));
                return;
            }
        }
        RK_HIP(hipHostMalloc(ptr, need, hipHostMallocPortable));
        std::lock_guard<std::mutex> lk(g_host_mtx);
        g_host_live.emplace(*ptr, need);
    });
}

int rk_host_free(void *ptr)
{
    return guard([&] {
        if (!ptr) {
            return;
        }
        void *drop = ptr;
        {
            std::lock_guard<std::mutex> lk(g_host_mtx);
            const auto it = g_host_live.find(ptr);
            if (it != g_host_live.end()) {
                const size_t bytes = it->second;
                g_host_live.erase(it);
                if (bytes >= (size_t(1) << 20) && g_host_parked.size() < 8 && g_host_parked_bytes + bytes <= (size_t(512) << 20)) {
                    g_host_parked.push_back(parked_host{ptr, bytes});
                    g_host_parked_bytes += bytes;
                    drop = nullptr;
                }
            }
        }
        if (drop) {
            RK_HIP(hipHostFree(drop));
        }
    });
}

int rk_acc_pot_device(rk_state *s, int q, int64_t p_begin, int64_t p_end, void *const *d_out, double mac_value,
                      double G, double eps2, int offset_output, void *hip_stream)
{
    return guard([&] {
        check_call(s, q, d_out, mac_value, G, eps2);
        device_guard dg(s->device);
        auto stream = static_cast<hipStream_t>(hip_stream);
        if (s->fp == RK_F32) {
            run_impl<float>(*s, q, p_begin, p_end, d_out, mac_value, G, eps2, offset_output, stream);
        } else {
            run_impl<double>(*s, q, p_begin, p_end, d_out, mac_value, G, eps2, offset_output, stream);
        }
    });
}

int rk_acc_pot(rk_state *s, int q, int64_t p_begin, int64_t p_end, void *const *out, double mac_value, double G,
               double eps2, int offset_output)
{
    return guard([&] {
        check_call(s, q, out, mac_value, G, eps2);
        if (p_begin < 0 || p_end < p_begin || p_end > s->nparts) {
            throw rk::error(RK_EINVAL, "invalid particle range");
        }
        if (offset_output & ~(RK_OUT_OFFSET | RK_OUT_ORDERED)) {
            throw rk::error(RK_EINVAL, "rk_acc_pot(): invalid output flags");
        }
        if ((offset_output & RK_OUT_ORDERED) && (p_begin != 0 || p_end != s->nparts)) {
            throw rk::error(RK_EINVAL, "rk_acc_pot() with RK_OUT_ORDERED (original-order host outputs) takes the whole range "
                                       "[0, nparts): the results are scattered all over the output arrays");
        }
        const size_t fsz = s->fp == RK_F32 ? sizeof(float) : sizeof(double);
        const auto count = static_cast<size_t>(p_end - p_begin);
        const int nres = user_nres(*s, q);
        if (!count) {
            return;
        }
        device_guard dg(s->device);
        const size_t need = count * fsz * static_cast<size_t>(nres);
        if (offset_output & RK_OUT_ORDERED) {
            // accs_o / pots_o for host arrays (tree.hpp:3320-3330): the kernels scatter the results through perm into a buffer
            // in HBM (where random 4-byte stores cost nothing: the RK_OUT_ORDERED epilogue of rk_acc_pot_device), the ordered
            // arrays then travel in one piece each -- straight into pinned arrays, through the staging buffer and the host
            // threads otherwise, array k being delivered while array k + 1 is still on its way. A host-side scatter of
            // 3 x 4M values through a random permutation costs 6 ms on eight threads (30 ms on one); this, 1.3 ms.
            if (s->d_out_bytes < need) {
                if (s->d_out) {
                    RK_HIP(hipDeviceSynchronize());
                    rk::pool_free(s->d_out);
                    s->d_out = nullptr;
                    s->d_out_bytes = 0;
                }
                s->d_out = rk::pool_alloc(need);
                s->d_out_bytes = need;
            }
            const size_t arr = count * fsz;
            void *d_ptrs[4] = {};
            unsigned char *dst[4] = {};
            for (int k = 0; k < nres; ++k) {
                d_ptrs[k] = static_cast<unsigned char *>(s->d_out) + static_cast<size_t>(k) * arr;
                dst[k] = static_cast<unsigned char *>(out[k]);
            }
            if (s->fp == RK_F32) {
                run_impl<float>(*s, q, p_begin, p_end, d_ptrs, mac_value, G, eps2, RK_OUT_OFFSET | RK_OUT_ORDERED, nullptr, false);
            } else {
                run_impl<double>(*s, q, p_begin, p_end, d_ptrs, mac_value, G, eps2, RK_OUT_OFFSET | RK_OUT_ORDERED, nullptr, false);
            }
            bool pinned = true;
            for (int k = 0; pinned && k < nres; ++k) {
                pinned = device_view_of_host_range(dst[k], arr) != nullptr;
            }
            if (pinned || need < (size_t(1) << 20)) {
                for (int k = 0; k < nres; ++k) {
                    RK_HIP(hipMemcpyAsync(dst[k], d_ptrs[k], arr, hipMemcpyDeviceToHost, nullptr));
                }
                RK_HIP(hipStreamSynchronize(nullptr));
                return;
            }
            if (s->h_stage_bytes < need) {
                if (s->h_stage) {
                    RK_HIP(hipDeviceSynchronize());
                    stage_give(phys(s->device), s->h_stage, s->h_stage_bytes);
                    s->h_stage = nullptr;
                    s->h_stage_bytes = 0;
                }
                size_t got = 0;
                s->h_stage = stage_take(phys(s->device), need, got);
                if (s->h_stage) {
                    s->h_stage_bytes = got;
                } else {
                    RK_HIP(hipHostMalloc(&s->h_stage, need, hipHostMallocDefault));
                    s->h_stage_bytes = need;
                }
            }
            auto *stage = static_cast<unsigned char *>(s->h_stage);
            for (int k = 0; k < nres; ++k) {
                if (!s->ev_arr[k]) {
                    RK_HIP(hipEventCreateWithFlags(&s->ev_arr[k], hipEventDisableTiming));
                }
                RK_HIP(hipMemcpyAsync(stage + static_cast<size_t>(k) * arr, d_ptrs[k], arr, hipMemcpyDeviceToHost, nullptr));
                RK_HIP(hipEventRecord(s->ev_arr[k], nullptr));
            }
            static const int max_thr_o = [] {
                const char *e = std::getenv("RK_HOST_THREADS");
                const int v = e ? std::atoi(e) : 8;
                return v < 1 ? 1 : v;
            }();
            const size_t piece = size_t(2) << 20;
            for (int k = 0; k < nres; ++k) {
                RK_HIP(hipEventSynchronize(s->ev_arr[k]));
                const int n_items = static_cast<int>((arr + piece - 1) / piece);
                const int n_thr = std::max(1, std::min<int>({max_thr_o, n_items, static_cast<int>(std::thread::hardware_concurrency())}));
                delivery_pool::get().run(n_items, n_thr, [&](int item) {
                    const size_t off = static_cast<size_t>(item) * piece;
                    stream_copy(dst[k] + off, stage + static_cast<size_t>(k) * arr + off, std::min(piece, arr - off));
                });
            }
            return;
        }
        // No hipGraph capture on this path: the callers of the host entry point drive several devices from several host
        // threads (kwargs::split), and a capture in one thread makes legacy-stream operations of the others fail
        // (hipErrorStreamCaptureImplicit). The 35 us a replay saves vanish next to the transfer of the results.
        unsigned char *dst[4] = {};
        for (int k = 0; k < nres; ++k) {
            dst[k] = static_cast<unsigned char *>(out[k]) + (offset_output ? static_cast<size_t>(p_begin) * fsz : 0);
        }
        // Output arrays in pinned host memory (rk_host_alloc(), rakau_amd::pinned_allocator, hipHostRegister): the kernels
        // write the results where the caller wants them, nothing is staged or copied.
        {
            void *v_ptrs[4] = {};
            bool all = true;
            for (int k = 0; all && k < nres; ++k) {
                v_ptrs[k] = device_view_of_host_range(dst[k], count * fsz);
                all = v_ptrs[k] != nullptr;
            }
            if (all) {
                // Repeated calls are replayed from a hipGraph like device-output calls (round 5). The four class kernels of a large
                // call then sit in ONE queue and share the device as the replayed device-output step does; launched directly on four
                // streams, the R <= 2 kernels (8 waves per SIMD, short waves) take most of the slots first and end at 1.3 ms of a
                // 2.2 ms step, which leaves the R = 3 / 4 kernels to run among themselves at 6 waves per SIMD and the R = 3 kernel
                // alone for the last 0.2 ms (tools/seam_timeline.py): 4M 2.303 -> 2.234 ms per call, 1737 -> 1790 Mparticles/s
                // (tools/archive/jobs_r05/r05_job24.sh; round 4 measured the replay 0.01 ms SLOWER: the kernels were 4 % slower then and
                // better balanced at 7/7/6/5 waves per SIMD).
                constexpr bool host_graph = true;
                s->want_done_event = true;
                try {
                    if (s->fp == RK_F32) {
                        run_impl<float>(*s, q, p_begin, p_end, v_ptrs, mac_value, G, eps2, 0, nullptr, host_graph);
                    } else {
                        run_impl<double>(*s, q, p_begin, p_end, v_ptrs, mac_value, G, eps2, 0, nullptr, host_graph);
                    }
                } catch (...) {
                    s->want_done_event = false;
                    throw;
                }
                s->want_done_event = false;
                RK_HIP(hipEventSynchronize(s->ev1));
                return;
            }
        }
        if (need < (size_t(1) << 20)) {
            // Small results: device scratch + one copy per array (the only path that needs the scratch).
            if (s->d_out_bytes < need) {
                if (s->d_out) {
                    RK_HIP(hipDeviceSynchronize());
                    rk::pool_free(s->d_out);
                    s->d_out = nullptr;
                    s->d_out_bytes = 0;
                }
                s->d_out = rk::pool_alloc(need);
                s->d_out_bytes = need;
            }
            void *d_ptrs[4] = {};
            for (int k = 0; k < nres; ++k) {
                d_ptrs[k] = static_cast<unsigned char *>(s->d_out) + static_cast<size_t>(k) * count * fsz;
            }
            if (s->fp == RK_F32) {
                run_impl<float>(*s, q, p_begin, p_end, d_ptrs, mac_value, G, eps2, 0, nullptr, false);
            } else {
                run_impl<double>(*s, q, p_begin, p_end, d_ptrs, mac_value, G, eps2, 0, nullptr, false);
            }
            for (int k = 0; k < nres; ++k) {
                RK_HIP(hipMemcpy(dst[k], d_ptrs[k], count * fsz, hipMemcpyDeviceToHost));
            }
            return;
        }
        // Optional (RK_HOST_REGISTER=1; OFF by default): register the caller's pageable arrays for the duration of this blocking
        // call, let the kernels write into them, unregister. Measured at 4M fp32 (48 MB): 2.27 ms per call into arrays used
        // before against 2.86 through the staging buffer. It is NOT safe in a process where anything else pins host memory
        // that shares pages with the arrays: the HIP runtime keeps a cache of the ranges it pinned for pageable hipMemcpy
        // calls (sources read-only), a registration that overlaps one of those gets its mapping, and the traversal dies of
        // "Memory access fault by GPU ... Write access to a read-only page" or of a fault when the cached pin is evicted
        // (tools/stress_host_register.py: every run with the registration on aborts within seconds, none without;
        // profiles/r03/host_register_overlap.txt). Only for applications that never hand pageable memory to HIP copies.
        // With it on: always for arrays the previous call on this state wrote, otherwise up to 256 MB.
        {
            static const bool reg = [] {
                const char *e = std::getenv("RK_HOST_REGISTER");
                return e && std::atoi(e) != 0;
            }();
            constexpr size_t reg_max = size_t(256) << 20;
            bool seen = true;
            for (int k = 0; k < nres; ++k) {
                seen = seen && s->last_host_out[k] == dst[k];
            }
            seen = seen && s->last_host_bytes == count * fsz;
            for (int k = 0; k < 4; ++k) {
                s->last_host_out[k] = k < nres ? dst[k] : nullptr;
            }
            s->last_host_bytes = count * fsz;
            if (reg && (seen || need <= reg_max)) {
                // Arrays that share a page (slices of one allocation) are registered as one range.
                struct range {
                    unsigned char *b, *e;
                    void *dev;
                };
                std::vector<range> ranges;
                {
                    std::vector<std::pair<unsigned char *, unsigned char *>> v;
                    for (int k = 0; k < nres; ++k) {
                        v.emplace_back(dst[k], dst[k] + count * fsz);
                    }
                    std::sort(v.begin(), v.end());
                    for (const auto &r : v) {
                        if (!ranges.empty() && r.first <= ranges.back().e + 4096) {
                            ranges.back().e = std::max(ranges.back().e, r.second);
                        } else {
                            ranges.push_back(range{r.first, r.second, nullptr});
                        }
                    }
                }
                size_t done = 0;
                bool ok = true;
                for (; ok && done < ranges.size(); ++done) {
                    auto &r = ranges[done];
                    ok = hipHostRegister(r.b, static_cast<size_t>(r.e - r.b), hipHostRegisterDefault) == hipSuccess;
                    if (ok && hipHostGetDevicePointer(&r.dev, r.b, 0) != hipSuccess) {
                        (void)hipHostUnregister(r.b);
                        ok = false;
                    }
                    if (!ok) {
                        (void)hipGetLastError(); // e.g. part of the range is registered already: the staging path serves the call
                        break;
                    }
                }
                struct unreg {
                    std::vector<range> &r;
                    size_t n;
                    ~unreg()
                    {
                        for (size_t k = 0; k < n; ++k) {
                            (void)hipHostUnregister(r[k].b);
                        }
                    }
                } guard_{ranges, ok ? ranges.size() : done};
                if (ok) {
                    void *v_ptrs[4] = {};
                    for (int k = 0; k < nres; ++k) {
                        for (const auto &r : ranges) {
                            if (dst[k] >= r.b && dst[k] < r.e) {
                                v_ptrs[k] = static_cast<unsigned char *>(r.dev) + (dst[k] - r.b);
                            }
                        }
                    }
                    if (s->fp == RK_F32) {
                        run_impl<float>(*s, q, p_begin, p_end, v_ptrs, mac_value, G, eps2, 0, nullptr, false);
                    } else {
                        run_impl<double>(*s, q, p_begin, p_end, v_ptrs, mac_value, G, eps2, 0, nullptr, false);
                    }
                    RK_HIP(hipEventSynchronize(s->ev1));
                    return;
                }
            }
        }
        // Large results: the kernels write straight into a pinned staging buffer (host memory mapped into the device's
        // address space: posted PCIe writes that trickle out while the traversal computes -- 48 MB during a 2.3 ms kernel
        // at 4M particles), so nothing is left to transfer when the kernels end; host threads then move the staging
        // buffer into the caller's pageable arrays. Measured at 4M fp32 (ms per call, 2.3 ms of it the kernels): a plain
        // hipMemcpy into pageable memory afterwards 6.4, device -> pinned chunks + threaded delivery after the kernels
        // 4.2, the same with the traversal launched in 4 Morton chunks so that the copies overlap it 4.3 (four small
        // launches lose what the overlap wins), this 3.1 (profiles/r02/host_output_path.txt).
        if (s->h_stage_bytes < need) {
            if (s->h_stage) {
                RK_HIP(hipDeviceSynchronize());
                stage_give(phys(s->device), s->h_stage, s->h_stage_bytes);
                s->h_stage = nullptr;
                s->h_stage_bytes = 0;
            }
            size_t got = 0;
            s->h_stage = stage_take(phys(s->device), need, got);
            if (s->h_stage) {
                s->h_stage_bytes = got;
            } else {
                RK_HIP(hipHostMalloc(&s->h_stage, need, hipHostMallocDefault));
                s->h_stage_bytes = need;
            }
        }
        void *h_ptrs[4] = {};
        for (int k = 0; k < nres; ++k) {
            h_ptrs[k] = static_cast<unsigned char *>(s->h_stage) + static_cast<size_t>(k) * count * fsz;
        }
        const auto *stage = static_cast<const unsigned char *>(s->h_stage);
        static const int max_thr = [] {
            const char *e = std::getenv("RK_HOST_THREADS"); // delivery threads (default 8; memory-bound beyond that)
            const int v = e ? std::atoi(e) : 8;
            return v < 1 ? 1 : v;
        }();
        const size_t arr = count * fsz;
        // Elements [eb, ee) of every staged array -> the caller's arrays, in 2 MB pieces on the pool threads, with streaming
        // stores (no read-for-ownership of destination lines that are overwritten whole).
        auto deliver = [&](size_t eb, size_t ee) {
            const size_t piece = size_t(2) << 20, span = (ee - eb) * fsz;
            const size_t per = (span + piece - 1) / piece;
            const int n_items = static_cast<int>(per * static_cast<size_t>(nres));
            if (n_items <= 0) {
                return;
            }
            const int n_thr = std::max(1, std::min<int>({max_thr, n_items, static_cast<int>(std::thread::hardware_concurrency())}));
            delivery_pool::get().run(n_items, n_thr, [&](int item) {
                const size_t k = static_cast<size_t>(item) / per, off = eb * fsz + (static_cast<size_t>(item) % per) * piece;
                const size_t nb = std::min(piece, ee * fsz - off);
                stream_copy(dst[k] + off, stage + k * arr + off, nb);
            });
        };
        // (replayed from a hipGraph when the call recurs, like the pinned-output call above)
        constexpr bool staged_graph = true;
        auto run = [&](int64_t b, int64_t e, void *const *ptrs) {
            s->want_done_event = true;
            try {
                if (s->fp == RK_F32) {
                    run_impl<float>(*s, q, b, e, ptrs, mac_value, G, eps2, 0, nullptr, staged_graph);
                } else {
                    run_impl<double>(*s, q, b, e, ptrs, mac_value, G, eps2, 0, nullptr, staged_graph);
                }
            } catch (...) {
                s->want_done_event = false;
                throw;
            }
            s->want_done_event = false;
        };
        // Two parts (round 4): the first 85 % of the range is traversed first and DELIVERED by the host threads
        // while the second part is traversed; only the second part's delivery is left when the kernels end. The cut is a
        // critical-node boundary, the two parts are ordinary sub-range calls (their union equals the one-part result bit for
        // bit), and two launches cost about 0.1 ms more than one at 4M, where the delivery of 85 % of the results costs
        // 0.3-0.5: 2.65 -> 2.53-2.54 ms per call (fractions 0.7 / 0.8 / 0.85 / 0.9: 2.68 / 2.56 / 2.54 / 2.53-2.80), 2M 1.59 -> 1.39,
        // 4M accelerations + potentials 3.32 -> 2.79 (tools/archive/jobs_r04/r04_job39.sh). Results below
        // 16 MB: one part, delivered at the end.
        constexpr double split_frac = 0.85;
        int64_t cut = p_begin;
        if (split_frac > 0.0 && need >= (size_t(16) << 20)) {
            ensure_mirrors(*s);
            const auto target = p_begin + static_cast<int64_t>(split_frac * static_cast<double>(count));
            const auto it = std::lower_bound(s->crit_begin.begin(), s->crit_begin.end(), target);
            cut = it == s->crit_begin.end() ? p_end : *it;
        }
        if (cut > p_begin && cut < p_end) {
            if (!s->ev_mid) {
                RK_HIP(hipEventCreateWithFlags(&s->ev_mid, hipEventDisableTiming));
            }
            run(p_begin, cut, h_ptrs);
            RK_HIP(hipEventRecord(s->ev_mid, nullptr));
            void *h2[4] = {};
            for (int k = 0; k < nres; ++k) {
                h2[k] = static_cast<unsigned char *>(h_ptrs[k]) + static_cast<size_t>(cut - p_begin) * fsz;
            }
            s->keep_ev0 = true;
            try {
                run(cut, p_end, h2);
            } catch (...) {
                s->keep_ev0 = false;
                throw;
            }
            s->keep_ev0 = false;
            RK_HIP(hipEventSynchronize(s->ev_mid));
            deliver(0, static_cast<size_t>(cut - p_begin));
            RK_HIP(hipEventSynchronize(s->ev1));
            deliver(static_cast<size_t>(cut - p_begin), count);
        } else {
            run(p_begin, p_end, h_ptrs);
            RK_HIP(hipEventSynchronize(s->ev1));
            deliver(0, count);
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

int rk_state_export(const rk_state *s, int *count, void **ptrs, int64_t *bytes, int64_t meta[RK_META_WORDS])
{
    return guard([&] {
        if (!s || !count || !ptrs || !bytes || !meta) {
            throw rk::error(RK_EINVAL, "null argument");
        }
        {
            device_guard dg(s->device);
            ensure_mirrors(*const_cast<rk_state *>(s)); // the class-list buffer must be complete before it travels
        }
        // The RK_NBUF traversal buffers, then the permutation (uint32 per particle; 0 bytes if the state has none).
        *count = RK_NBUF + 1;
        for (int i = 0; i < RK_NBUF; ++i) {
            ptrs[i] = s->buf[i];
            bytes[i] = s->buf_bytes[i];
        }
        ptrs[RK_NBUF] = s->bld_perm;
        bytes[RK_NBUF] = s->bld_perm ? s->nparts * static_cast<int64_t>(sizeof(uint32_t)) : 0;
        std::fill(meta, meta + RK_META_WORDS, int64_t(0));
        meta[0] = state_layout_tag;
        meta[1] = s->fp;
        meta[2] = s->mac;
        meta[3] = s->nparts;
        meta[4] = s->tree_size;
        meta[5] = s->n_crit;
        meta[6] = static_cast<int64_t>(s->ncrit);
        meta[7] = s->n_internal;
        meta[24] = s->ndim;
        for (int i = 0; i <= RK_NBUF; ++i) {
            meta[8 + i] = bytes[i];
        }
        std::memcpy(&meta[25], &s->box_size, sizeof(double));
        meta[26] = s->box_deduced;
        meta[27] = static_cast<int64_t>(s->max_leaf_n);
    });
}

// A replica is made in three steps shared by rk_state_import, rk_state_clone, rk_state_clone_all and rk_state_broadcast:
// replica_shell() checks the meta block and allocates the state with empty buffers on `device`; the caller fills the
// buffers (device-to-device, peer copies, RCCL); replica_finish() derives the host mirrors from the critical-node buffer.
static state_ptr replica_shell(int device, int count, const int64_t *bytes, const int64_t meta[RK_META_WORDS])
{
    if (meta[0] != state_layout_tag || count != RK_NBUF + 1) {
        throw rk::error(RK_EINVAL, "unrecognised state layout");
    }
    check_common(static_cast<int>(meta[1]), static_cast<int>(meta[2]));
    check_ndim(static_cast<int>(meta[24]));
    check_device(device);
    // The meta block is trusted no further than the buffers it describes: every count must match the byte size
    // of its buffer, and the limits of rk_state_create apply.
    const int64_t nparts = meta[3], tree_size = meta[4], n_crit = meta[5], n_internal = meta[7];
    const int64_t fsz = meta[1] == RK_F32 ? 4 : 8;
    if (nparts < 0 || tree_size < 0 || n_crit < 0 || n_internal < 0 || n_crit > tree_size || n_internal > tree_size
        || static_cast<uint64_t>(nparts) >= 0xffffffffull || static_cast<uint64_t>(tree_size) >= rk::max_list_nodes
        || (nparts > 0 && (tree_size == 0 || n_crit == 0)) || meta[6] <= 0) {
        throw rk::error(RK_EINVAL, "inconsistent counts in the meta block of rk_state_import");
    }
    const int64_t rec_bytes = meta[1] == RK_F32 ? int64_t(sizeof(rk::node_rec<float>)) : int64_t(sizeof(rk::node_rec<double>));
    const int64_t expect[RK_NBUF] = {nparts * 4 * fsz, tree_size * 4 * fsz, tree_size * 2 * fsz, tree_size * 16,
                                     n_crit * 16,      n_internal * 8 * 4,  -1 /* class lists: checked below */,
                                     tree_size * rec_bytes, n_crit * 2 * 4 * fsz};
    for (int i = 0; i < RK_NBUF; ++i) {
        if (bytes[i] != meta[8 + i] || (expect[i] >= 0 && bytes[i] != expect[i])) {
            throw rk::error(RK_EINVAL, "buffer " + std::to_string(i) + " of rk_state_import has " + std::to_string(bytes[i])
                                           + " bytes, which does not match the meta block");
        }
    }
    if (bytes[RK_BUF_CLASS] != 2 * n_crit * 4) {
        throw rk::error(RK_EINVAL, "class-list buffer size mismatch in rk_state_import");
    }
    if (bytes[RK_NBUF] != meta[8 + RK_NBUF] || (bytes[RK_NBUF] != 0 && bytes[RK_NBUF] != nparts * 4)) {
        throw rk::error(RK_EINVAL, "permutation buffer size mismatch in rk_state_import");
    }
    if (static_cast<size_t>(bytes[RK_BUF_CRIT]) != static_cast<size_t>(n_crit) * sizeof(uint4)) {
        throw rk::error(RK_EINVAL, "critical node buffer size mismatch in rk_state_import");
    }
    device_guard dg(device);
    state_ptr s(new rk_state);
    s->fp = static_cast<int>(meta[1]);
    s->mac = static_cast<int>(meta[2]);
    s->device = device;
    s->nparts = nparts;
    s->tree_size = tree_size;
    s->n_crit = n_crit;
    s->ncrit = static_cast<uint64_t>(meta[6]);
    s->n_internal = n_internal;
    s->ndim = static_cast<int>(meta[24]);
    std::memcpy(&s->box_size, &meta[25], sizeof(double));
    s->box_deduced = meta[26] != 0;
    s->max_leaf_n = static_cast<uint64_t>(meta[27]);
    for (int i = 0; i < RK_NBUF; ++i) {
        s->buf_bytes[i] = bytes[i];
        if (bytes[i]) {
            s->buf[i] = rk::pool_alloc(static_cast<size_t>(bytes[i]));
        }
    }
    if (bytes[RK_NBUF]) {
        // With the permutation a replica serves RK_OUT_ORDERED like the state it was exported from.
        s->bld_perm = rk::pool_alloc(static_cast<size_t>(bytes[RK_NBUF]));
    }
    return s;
}

// Destination of exported buffer i in a replica (nullptr for an empty one).
static void *replica_buffer(rk_state &s, int i)
{
    return i < RK_NBUF ? s.buf[i] : s.bld_perm;
}

static void replica_finish(rk_state &s)
{
    device_guard dg(s.device);
    std::vector<uint4> crit(static_cast<size_t>(s.n_crit));
    if (!crit.empty()) {
        RK_HIP(hipMemcpy(crit.data(), s.buf[RK_BUF_CRIT], crit.size() * sizeof(uint4), hipMemcpyDeviceToHost));
    }
    build_host_mirrors(s, crit);
    ensure_call_resources_any(s);
    rk::replica_first_order(s); // (small trees: the first call of a replica runs in heavy-first order like its source's)
}

// rk_state_import (buffers already on `device`: src_device < 0) and rk_state_clone (buffers on src_device).
static int import_impl(rk_state **out, int device, int count, void *const *ptrs, const int64_t *bytes,
                       const int64_t meta[RK_META_WORDS], int src_device)
{
    return guard([&] {
        if (!out || !ptrs || !bytes || !meta) {
            throw rk::error(RK_EINVAL, "null argument");
        }
        *out = nullptr;
        state_ptr s = replica_shell(device, count, bytes, meta);
        device_guard dg(device);
        for (int i = 0; i <= RK_NBUF; ++i) {
            if (!bytes[i]) {
                continue;
            }
            if (!ptrs[i]) {
                throw rk::error(RK_EINVAL, "null buffer in rk_state_import");
            }
            // Device-to-device on one GPU, or a peer copy over xGMI between two GPUs of this process.
            if (src_device < 0 || phys(src_device) == phys(device)) {
                RK_HIP(hipMemcpy(replica_buffer(*s, i), ptrs[i], static_cast<size_t>(bytes[i]), hipMemcpyDeviceToDevice));
            } else {
                RK_HIP(hipMemcpyPeer(replica_buffer(*s, i), phys(device), ptrs[i], phys(src_device), static_cast<size_t>(bytes[i])));
            }
        }
        replica_finish(*s);
        *out = s.release();
    });
}

int rk_state_import(rk_state **out, int device, int count, void *const *ptrs, const int64_t *bytes,
                    const int64_t meta[RK_META_WORDS])
{
    return import_impl(out, device, count, ptrs, bytes, meta, -1);
}

// Export of a state whose device work has completed (what every replication entry starts from).
static int export_settled(const rk_state *src, int *count, void **ptrs, int64_t *bytes, int64_t *meta)
{
    const int rc = rk_state_export(src, count, ptrs, bytes, meta);
    if (rc != RK_OK) {
        return rc;
    }
    // Everything the source has enqueued (its build, an upload) must have landed before its buffers are read.
    return guard([&] {
        device_guard dg(src->device);
        RK_HIP(hipDeviceSynchronize());
    });
}

int rk_state_clone(rk_state **out, const rk_state *src, int device)
{
    if (!out || !src) {
        return guard([] { throw rk::error(RK_EINVAL, "null argument"); });
    }
    int count = 0;
    void *ptrs[RK_MAX_BUFFERS] = {};
    int64_t bytes[RK_MAX_BUFFERS] = {}, meta[RK_META_WORDS] = {};
    const int rc = export_settled(src, &count, ptrs, bytes, meta);
    if (rc != RK_OK) {
        return rc;
    }
    return import_impl(out, device, count, ptrs, bytes, meta, src->device);
}

// Replicas of `src` on n devices at once. The copies fan out as a doubling tree: in every round each device that holds the
// state sends it to one that does not, all transfers of a round in flight together (asynchronous peer copies on one stream
// per destination), so that n replicas take ceil(log2(n + 1)) rounds over as many xGMI links as there are senders instead
// of n copies leaving the source one after the other (the reference uploads tree and particles to every device from the
// host on every call, on one stream per device: src/rakau_cuda.cu:492-527).
int rk_state_clone_all(rk_state **outs, const rk_state *src, const int *devices, int n)
{
    if (!outs || !src || !devices || n < 0) {
        return guard([] { throw rk::error(RK_EINVAL, "null argument"); });
    }
    for (int i = 0; i < n; ++i) {
        outs[i] = nullptr;
    }
    int count = 0;
    void *ptrs[RK_MAX_BUFFERS] = {};
    int64_t bytes[RK_MAX_BUFFERS] = {}, meta[RK_META_WORDS] = {};
    const int rc = export_settled(src, &count, ptrs, bytes, meta);
    if (rc != RK_OK) {
        return rc;
    }
    std::vector<state_ptr> made(static_cast<size_t>(n));
    std::vector<hipStream_t> streams(static_cast<size_t>(n), nullptr);
    const int rc2 = guard([&] {
        for (int i = 0; i < n; ++i) {
            made[static_cast<size_t>(i)] = replica_shell(devices[i], count, bytes, meta);
            device_guard dg(devices[i]);
            RK_HIP(hipStreamCreateWithFlags(&streams[static_cast<size_t>(i)], hipStreamNonBlocking));
        }
        // holders: -1 = the source, i >= 0 = made[i] (complete).
        std::vector<int> holders{-1};
        int next = 0;
        static const bool trace = [] {
            const char *e = std::getenv("RK_CLONE_TRACE"); // prints the rounds (which device sends to which)
            return e && std::atoi(e) != 0;
        }();
        int round = 0;
        while (next < n) {
            const int first = next;
            const size_t n_holders = holders.size();
            for (size_t h = 0; h < n_holders && next < n; ++h, ++next) {
                const rk_state &from = holders[h] < 0 ? *src : *made[static_cast<size_t>(holders[h])];
                rk_state &to = *made[static_cast<size_t>(next)];
                device_guard dg(to.device);
                for (int b = 0; b <= RK_NBUF; ++b) {
                    if (!bytes[b]) {
                        continue;
                    }
                    const void *sp = holders[h] < 0 ? ptrs[b] : replica_buffer(const_cast<rk_state &>(from), b);
                    if (phys(from.device) == phys(to.device)) {
                        RK_HIP(hipMemcpyAsync(replica_buffer(to, b), sp, static_cast<size_t>(bytes[b]), hipMemcpyDeviceToDevice,
                                              streams[static_cast<size_t>(next)]));
                    } else {
                        RK_HIP(hipMemcpyPeerAsync(replica_buffer(to, b), phys(to.device), sp, phys(from.device),
                                                  static_cast<size_t>(bytes[b]), streams[static_cast<size_t>(next)]));
                    }
                }
                if (trace) {
                    std::fprintf(stderr, "rk_state_clone_all round %d: device %d -> device %d\n", round, from.device, to.device);
                }
            }
            for (int i = first; i < next; ++i) {
                device_guard dg(made[static_cast<size_t>(i)]->device);
                RK_HIP(hipStreamSynchronize(streams[static_cast<size_t>(i)]));
                holders.push_back(i);
            }
            ++round;
        }
        for (int i = 0; i < n; ++i) {
            replica_finish(*made[static_cast<size_t>(i)]);
        }
    });
    for (int i = 0; i < n; ++i) {
        if (streams[static_cast<size_t>(i)]) {
            int prev = 0;
            (void)hipGetDevice(&prev);
            (void)hipSetDevice(phys(devices[i]));
            (void)hipStreamDestroy(streams[static_cast<size_t>(i)]);
            (void)hipSetDevice(prev);
        }
    }
    if (rc2 != RK_OK) {
        return rc2;
    }
    for (int i = 0; i < n; ++i) {
        outs[i] = made[static_cast<size_t>(i)].release();
    }
    return RK_OK;
}

// ---- one process per GPU: the replicate step over RCCL (rccl(): the collectives library, bound at run time) ----

int rk_comm_unique_id(char id[RK_COMM_ID_BYTES])
{
    return guard([&] {
        if (!id) {
            throw rk::error(RK_EINVAL, "null argument");
        }
        rccl_api::unique_id u{};
        rccl_check(rccl().get_unique_id(&u), "ncclGetUniqueId");
        std::memcpy(id, u.internal, sizeof(u.internal));
    });
}

int rk_comm_init(void **comm, int n_ranks, const char id[RK_COMM_ID_BYTES], int rank, int device)
{
    return guard([&] {
        if (!comm || !id) {
            throw rk::error(RK_EINVAL, "null argument");
        }
        *comm = nullptr;
        check_device(device);
        device_guard dg(device);
        rccl_api::unique_id u{};
        std::memcpy(u.internal, id, sizeof(u.internal));
        rccl_check(rccl().comm_init_rank(comm, n_ranks, u, rank), "ncclCommInitRank");
    });
}

int rk_comm_destroy(void *comm)
{
    return guard([&] {
        if (comm) {
            rccl_check(rccl().comm_destroy(comm), "ncclCommDestroy");
        }
    });
}

// The replicate step of the one-process-per-GPU model: the state of rank `root` becomes a state on every rank's device.
// ncclBroadcast of the meta block, then of every exported buffer inside one group (RCCL pipelines them over the xGMI
// ring / tree it built for the communicator). No data-path collective is needed afterwards: every rank traverses its
// own Morton range of targets and nothing is reduced (north star).
int rk_state_broadcast(rk_state **state, int root, int rank, int device, void *comm, void *stream_)
{
    return guard([&] {
        if (!state || !comm || (rank == root && !*state)) {
            throw rk::error(RK_EINVAL, "null argument");
        }
        auto stream = static_cast<hipStream_t>(stream_);
        const rccl_api &api = rccl();
        check_device(device);
        device_guard dg(device);
        int count = RK_NBUF + 1;
        void *ptrs[RK_MAX_BUFFERS] = {};
        int64_t bytes[RK_MAX_BUFFERS] = {}, meta[RK_META_WORDS] = {};
        if (rank == root) {
            if ((*state)->device != device) {
                throw rk::error(RK_EINVAL, "the root's state does not live on the device of this rank");
            }
            const int rc = export_settled(*state, &count, ptrs, bytes, meta);
            if (rc != RK_OK) {
                throw rk::error(rc, rk_last_error());
            }
        }
        // Meta block first (through a small device buffer: RCCL moves device memory).
        struct dev_block {
            void *p = nullptr;
            ~dev_block()
            {
                rk::pool_free(p);
            }
        } dmeta;
        dmeta.p = rk::pool_alloc(sizeof(meta));
        if (rank == root) {
            RK_HIP(hipMemcpyAsync(dmeta.p, meta, sizeof(meta), hipMemcpyHostToDevice, stream));
        }
        rccl_check(api.broadcast(dmeta.p, dmeta.p, sizeof(meta), /* ncclUint8 */ 1, root, comm, stream), "ncclBroadcast(meta)");
        RK_HIP(hipMemcpyAsync(meta, dmeta.p, sizeof(meta), hipMemcpyDeviceToHost, stream));
        RK_HIP(hipStreamSynchronize(stream));
        state_ptr made;
        if (rank != root) {
            for (int i = 0; i <= RK_NBUF; ++i) {
                bytes[i] = meta[8 + i];
            }
            made = replica_shell(device, count, bytes, meta);
            for (int i = 0; i <= RK_NBUF; ++i) {
                ptrs[i] = replica_buffer(*made, i);
            }
        }
        rccl_check(api.group_start(), "ncclGroupStart");
        for (int i = 0; i <= RK_NBUF; ++i) {
            if (bytes[i]) {
                rccl_check(api.broadcast(ptrs[i], ptrs[i], static_cast<size_t>(bytes[i]), 1, root, comm, stream), "ncclBroadcast");
            }
        }
        rccl_check(api.group_end(), "ncclGroupEnd");
        RK_HIP(hipStreamSynchronize(stream));
        if (rank != root) {
            replica_finish(*made);
            *state = made.release();
        }
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
            case 4: { // table of the light-tail arrangement (FIRST_TAB_WORDS uint32; null unless the tree came with one)
                const bool have = s->first_tail_valid && s->first_tab;
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
