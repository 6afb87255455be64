// Product-side half of rk_xcheck.hpp: the launchers of the cross-check kernels (variant 1, variant 4) forward to
// librakau_amd_xcheck.so, which is loaded from this library's own directory the first time one of them is needed.
#include <dlfcn.h>

#include <mutex>
#include <string>

#include "rk_xcheck.hpp"

namespace rk
{

namespace
{
const xcheck_vtable *load_xcheck()
{
    Dl_info info;
    if (!dladdr(reinterpret_cast<const void *>(&rk_set_kernel_variant), &info) || !info.dli_fname) {
        throw error(RK_ERUNTIME, "cannot locate librakau_amd.so to load the cross-check kernels next to it");
    }
    std::string path(info.dli_fname);
    const auto slash = path.find_last_of('/');
    path = (slash == std::string::npos ? std::string() : path.substr(0, slash + 1)) + "librakau_amd_xcheck.so";
    void *h = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
    if (!h) {
        const char *why = dlerror();
        throw error(RK_ERUNTIME, "the cross-check kernels (variant 1: scalar depth-first walk, variant 4: split traversal) are "
                                 "not part of librakau_amd.so; loading " + path + " failed: " + (why ? why : "?"));
    }
    using entry_t = const xcheck_vtable *(*)();
    auto entry = reinterpret_cast<entry_t>(dlsym(h, "rk_xcheck_entry"));
    const xcheck_vtable *vt = entry ? entry() : nullptr;
    if (!vt || vt->abi != xcheck_abi_tag()) {
        throw error(RK_ERUNTIME, path + " does not match this build of librakau_amd.so");
    }
    return vt;
}
void check(const xcheck_vtable &vt, int rc)
{
    if (rc != RK_OK) {
        throw error(rc, vt.last_error());
    }
}
} // namespace

const xcheck_vtable &xcheck()
{
    static std::mutex mtx;
    static const xcheck_vtable *vt = nullptr;
    std::lock_guard<std::mutex> lk(mtx);
    if (!vt) {
        vt = load_xcheck(); // a failed load is retried by the next call
    }
    return *vt;
}

template <>
void launch_traversal<float>(const rk_state &s, int q, const kparams<float> &p, const int64_t cb[n_classes],
                             const int64_t ce[n_classes], hipStream_t stream)
{
    check(xcheck(), xcheck().traversal_f(&s, q, &p, cb, ce, stream));
}
template <>
void launch_traversal<double>(const rk_state &s, int q, const kparams<double> &p, const int64_t cb[n_classes],
                              const int64_t ce[n_classes], hipStream_t stream)
{
    check(xcheck(), xcheck().traversal_d(&s, q, &p, cb, ce, stream));
}
template <>
void launch_block<float>(const rk_state &s, int q, const kparams<float> &p, const uint32_t *list, int64_t n, hipStream_t stream)
{
    check(xcheck(), xcheck().block_f(&s, q, &p, list, n, stream));
}
template <>
void launch_block<double>(const rk_state &s, int q, const kparams<double> &p, const uint32_t *list, int64_t n, hipStream_t stream)
{
    check(xcheck(), xcheck().block_d(&s, q, &p, list, n, stream));
}
template <>
void launch_lists<float>(const rk_state &s, const kparams<float> &p, int64_t g_begin, int64_t g_end, hipStream_t stream)
{
    check(xcheck(), xcheck().lists_f(&s, &p, g_begin, g_end, stream));
}
template <>
void launch_lists<double>(const rk_state &s, const kparams<double> &p, int64_t g_begin, int64_t g_end, hipStream_t stream)
{
    check(xcheck(), xcheck().lists_d(&s, &p, g_begin, g_end, stream));
}
template <>
void launch_dense<float>(const rk_state &s, int q, const kparams<float> &p, const int64_t cb[n_classes], const int64_t ce[n_classes],
                         hipStream_t const streams[n_list_R], unsigned class_mask, int what)
{
    check(xcheck(), xcheck().dense_f(&s, q, &p, cb, ce, streams, class_mask, what));
}
template <>
void launch_dense<double>(const rk_state &s, int q, const kparams<double> &p, const int64_t cb[n_classes],
                          const int64_t ce[n_classes], hipStream_t const streams[n_list_R], unsigned class_mask, int what)
{
    check(xcheck(), xcheck().dense_d(&s, q, &p, cb, ce, streams, class_mask, what));
}

// (rk_init does not touch the cross-check library: a process that never selects variant 1 or 4 never maps it.)
void touch_split() {}

} // namespace rk
