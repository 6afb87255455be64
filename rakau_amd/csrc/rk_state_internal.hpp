// Shared between the translation units of the C ABI's host side (rk_state.hip: states, their life cycle and creation;
// rk_launch.hip: launch plans, the graph cache and the launch sequence of a traversal; rk_host_out.hip: results delivered into
// host arrays; rk_replica.hip: export / import / clones / the RCCL broadcast). Everything here is internal to librakau_amd.so
// (hidden visibility).
#ifndef RK_STATE_INTERNAL_HPP
#define RK_STATE_INTERNAL_HPP

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

namespace rkst
{

extern thread_local std::string g_err;

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

// rk_state.hip
int physical_device_count();
int alias_devices();
int logical_device_count();
int phys(int device);
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

// Layout tag of rk_state_export / rk_state_import ("rk04"): bump it whenever the buffer list or the meta block changes.
constexpr int64_t state_layout_tag = 0x726b3034;
extern std::atomic<int> g_build_exact;
int user_nres(const rk_state &s, int q);
void *stage_take(int dev, size_t need, size_t &got);
void stage_give(int dev, void *p, size_t bytes);
void stage_trim();
void release_tree(rk_state *s);
void free_state(rk_state *s);
struct state_deleter {
    void operator()(rk_state *s) const
    {
        free_state(s);
    }
};
using state_ptr = std::unique_ptr<rk_state, state_deleter>;
void alloc_upload(rk_state &s, int which, const void *host, size_t bytes);
void build_host_mirrors(rk_state &s, const std::vector<uint4> &crit);
std::vector<uint32_t> concat_class_lists(const rk_state &s);
void ensure_mirrors(rk_state &s);
void check_common(int fp, int mac);
void check_ndim(int ndim);
void check_device(int device);

// rk_launch.hip
extern std::atomic<int> g_forked_execs;
void drop_graph_exec(rk_state &s);
void park_class_graphs(rk_state &s); // (a state that goes away hands its re-targetable class-kernel graphs to the next one)
std::vector<void *> take_retired_plan_buffers();
void ensure_call_resources_any(rk_state &s);
void check_call(const rk_state *s, int q, void *const *out, double mac_value, double G, double eps2);
template <typename F>
void run_impl(rk_state &s, int q, int64_t p_begin, int64_t p_end, void *const *d_out, double mac_value, double G, double eps2,
              int offset_output, hipStream_t stream, bool allow_graph = true);
extern template void run_impl<float>(rk_state &, int, int64_t, int64_t, void *const *, double, double, double, int, hipStream_t, bool);
extern template void run_impl<double>(rk_state &, int, int64_t, int64_t, void *const *, double, double, double, int, hipStream_t, bool);

} // namespace rkst

// rk_host_out.hip
extern "C" void host_blocks_trim();

using namespace rkst;

#endif
