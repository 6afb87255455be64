// Binding between librakau_amd.so and librakau_amd_xcheck.so (the cross-check kernels: variant 1 = scalar depth-first walk,
// variant 4 = split traversal). The product library holds no code of either; rk_set_kernel_variant(state, 1 | 4) and
// RK_BIG_DFS=1 load the cross-check library from the product library's own directory on first use, and fail loudly if it
// is not there. One table of plain function pointers; errors come back as status codes (no C++ exception crosses the
// boundary between the two shared objects).
#ifndef RK_XCHECK_HPP
#define RK_XCHECK_HPP

#include "rk_common.hpp"

namespace rk
{

struct xcheck_vtable {
    uint64_t abi; // xcheck_abi_tag() of the build: both libraries must come from the same sources
    const char *(*last_error)();
    int (*traversal_f)(const rk_state *, int, const kparams<float> *, const int64_t *, const int64_t *, hipStream_t);
    int (*traversal_d)(const rk_state *, int, const kparams<double> *, const int64_t *, const int64_t *, hipStream_t);
    int (*block_f)(const rk_state *, int, const kparams<float> *, const uint32_t *, int64_t, hipStream_t);
    int (*block_d)(const rk_state *, int, const kparams<double> *, const uint32_t *, int64_t, hipStream_t);
    int (*lists_f)(const rk_state *, const kparams<float> *, int64_t, int64_t, hipStream_t);
    int (*lists_d)(const rk_state *, const kparams<double> *, int64_t, int64_t, hipStream_t);
    int (*dense_f)(const rk_state *, int, const kparams<float> *, const int64_t *, const int64_t *, const hipStream_t *, unsigned,
                   int);
    int (*dense_d)(const rk_state *, int, const kparams<double> *, const int64_t *, const int64_t *, const hipStream_t *, unsigned,
                   int);
    int (*touch)();
};

// Layout fingerprint of what crosses the boundary by reference.
constexpr uint64_t xcheck_abi_tag()
{
    return (static_cast<uint64_t>(sizeof(rk_state)) << 40) ^ (static_cast<uint64_t>(sizeof(kparams<float>)) << 20)
           ^ static_cast<uint64_t>(sizeof(kparams<double>)) ^ 0x7204ull;
}

// librakau_amd.so only: the table of the cross-check library, loaded on first use (throws rk::error if it is not there).
const xcheck_vtable &xcheck();

} // namespace rk

#endif
