// Device-side helpers shared by the traversal kernels.
#ifndef RK_DEVICE_HPP
#define RK_DEVICE_HPP

#include <type_traits>

#include "rk_common.hpp"

namespace rk
{

// ------------------------------------------------------------------------------------------------
// Arithmetic helpers.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float rk_fma(float a, float b, float c)
{
    return __builtin_fmaf(a, b, c);
}
__device__ __forceinline__ double rk_fma(double a, double b, double c)
{
    return __builtin_fma(a, b, c);
}
#ifndef RK_RSQ64_STEPS
#define RK_RSQ64_STEPS 1
#endif
// 1/sqrt(x): v_rsq_f32 (1 ulp) for fp32; for fp64 v_rsq_f64 refined by Newton steps.
__device__ __forceinline__ float rk_rsqrt(float x)
{
    return __builtin_amdgcn_rsqf(x);
}
__device__ __forceinline__ double rk_rsqrt(double x)
{
    double y = __builtin_amdgcn_rsq(x);
    // y <- y + y * (0.5 * (1 - x*y*y)): v_rsq_f64 delivers ~2^-26 relative accuracy, one step squares it to ~2 ulp,
    // a second one reaches rounding level (RK_RSQ64_STEPS).
    double e = rk_fma(-x * y, y, 1.0);
    y = rk_fma(y * 0.5, e, y);
#if RK_RSQ64_STEPS >= 2
    e = rk_fma(-x * y, y, 1.0);
    y = rk_fma(y * 0.5, e, y);
#endif
    return y;
}

__device__ __forceinline__ float rk_min(float a, float b)
{
    return __builtin_fminf(a, b);
}
__device__ __forceinline__ double rk_min(double a, double b)
{
    return __builtin_fmin(a, b);
}

__device__ __forceinline__ float rk_max3(float a, float b, float c)
{
    return __builtin_fmaxf(__builtin_fmaxf(a, b), c); // folds to v_max3_f32
}
__device__ __forceinline__ double rk_max3(double a, double b, double c)
{
    return __builtin_fmax(__builtin_fmax(a, b), c);
}

template <typename F>
__device__ __forceinline__ F mac_lhs(int mac, typename vt<F>::v2 mp, F mac_value)
{
    // tree.hpp:2632-2642 of the reference.
    if (mac == RK_MAC_BH) {
        return mp.x * mac_value;
    }
    const F t = rk_fma(mp.x, mac_value, mp.y);
    return t * t;
}

// One monopole interaction on a target: d = source - target, d2 = softened squared distance.
// Q == 0: acc[0..2] += d * m / r^3.  Q == 1: acc[0] -= m_tgt * m / r.  Q == 2: both (pot in acc[3]).
// Arithmetic of tree.hpp:2008-2068 / 2564-2589 of the reference with 1/sqrt in place of sqrt + divide.
// ND = 2 (quadtrees in the z = 0 plane): the z component is identically zero and is not accumulated.
template <typename F, int Q, int ND = 3>
__device__ __forceinline__ void interact(F (&acc)[nres_of(Q)], F dx, F dy, F dz, F d2, F m_src, F m_tgt)
{
#if RK_RSQ64_STEPS == 1
    if constexpr (sizeof(F) == 8) {
        // fp64: the Newton correction of v_rsq_f64 applied to the results directly. With y0 = rsq(d2) and
        // e = 1 - d2 y0^2 (|e| < 2^-25): 1/r = y0 (1 + e/2 + O(e^2)), 1/r^3 = y0^3 (1 + 3/2 e + O(e^2)). Accelerations
        // take six operations after the rsq instead of the seven of "refine y, then cube" (4-5 % of the 16M fp64 step),
        // potentials five either way; same order of error (dropped: 15/8 e^2 <= 8 ulp worst case against 9/8 e^2;
        // agreement with the CPU checker of the test suite unchanged: 3e-13). The same expressions serve Q = 0, 1 and 2, so
        // that accs_u(), pots_u() and accs_pots_u() keep agreeing bit for bit.
        const double y0 = __builtin_amdgcn_rsq(d2);
        const double s = y0 * y0;
        const double e = rk_fma(-d2, s, 1.0);
        const double my = m_src * y0;
        if constexpr (Q == 0 || Q == 2) {
            const double mr3 = (my * s) * rk_fma(1.5, e, 1.0);
            acc[0] = rk_fma(dx, mr3, acc[0]);
            acc[1] = rk_fma(dy, mr3, acc[1]);
            if constexpr (ND == 3) {
                acc[2] = rk_fma(dz, mr3, acc[2]);
            }
        }
        if constexpr (Q == 1 || Q == 2) {
            const double mr = my * rk_fma(0.5, e, 1.0);
            acc[Q == 1 ? 0 : 3] = rk_fma(-m_tgt, mr, acc[Q == 1 ? 0 : 3]);
        }
        return;
    }
#endif
    const F rinv = rk_rsqrt(d2);
    const F mr = m_src * rinv;
    if constexpr (Q == 0 || Q == 2) {
        const F mr3 = mr * (rinv * rinv);
        acc[0] = rk_fma(dx, mr3, acc[0]);
        acc[1] = rk_fma(dy, mr3, acc[1]);
        if constexpr (ND == 3) {
            acc[2] = rk_fma(dz, mr3, acc[2]);
        }
    }
    if constexpr (Q == 1) {
        acc[0] = rk_fma(-m_tgt, mr, acc[0]);
    }
    if constexpr (Q == 2) {
        acc[3] = rk_fma(-m_tgt, mr, acc[3]);
    }
}


// ------------------------------------------------------------------------------------------------
// Wave-level primitives (wave64).
// ------------------------------------------------------------------------------------------------
// Number of set bits of `mask` below the calling lane.
__device__ __forceinline__ unsigned wave_prefix_count(unsigned long long mask)
{
    return __builtin_amdgcn_mbcnt_hi(static_cast<unsigned>(mask >> 32),
                                     __builtin_amdgcn_mbcnt_lo(static_cast<unsigned>(mask), 0u));
}
// Orders LDS traffic between the lanes of one wavefront (no instruction besides the waits the compiler
// needs; DS operations of a wave execute in order).
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// Inclusive prefix sum over the 64 lanes (all of them active). Data-parallel-primitive form: a Hillis-Steele scan inside every
// row of 16 lanes (row_shr 1, 2, 4, 8; lanes without a source add 0), then the row totals are passed on (row_bcast:15 into
// rows 1 and 3, row_bcast:31 into rows 2 and 3): six v_add_u32 with DPP operands. Rounds 1-4 used six dependent __shfl_up,
// i.e. six ds_bpermute_b32 round trips through the LDS crossbar per scan (~400 cycles in the leaf-gathering path of every
// list-building wave) that also kept six address registers and six lane masks alive for the whole kernel.
__device__ __forceinline__ unsigned wave_incl_scan(unsigned v)
{
    const auto step = [](unsigned x, auto ctrl, auto rows) __attribute__((always_inline)) {
        return static_cast<unsigned>(__builtin_amdgcn_update_dpp(0, static_cast<int>(x), decltype(ctrl)::value, decltype(rows)::value,
                                                                 0xf, false));
    };
    using std::integral_constant;
    v += step(v, integral_constant<int, 0x111>{}, integral_constant<int, 0xf>{}); // row_shr:1
    v += step(v, integral_constant<int, 0x112>{}, integral_constant<int, 0xf>{}); // row_shr:2
    v += step(v, integral_constant<int, 0x114>{}, integral_constant<int, 0xf>{}); // row_shr:4
    v += step(v, integral_constant<int, 0x118>{}, integral_constant<int, 0xf>{}); // row_shr:8
    v += step(v, integral_constant<int, 0x142>{}, integral_constant<int, 0xa>{}); // row_bcast:15 -> rows 1, 3
    v += step(v, integral_constant<int, 0x143>{}, integral_constant<int, 0xc>{}); // row_bcast:31 -> rows 2, 3
    return v;
}

// OR of a value over the 64 lanes (all of them active), returned to every lane: the DPP network of wave_incl_scan().
__device__ __forceinline__ unsigned wave_reduce_or(unsigned v)
{
    const auto step = [](unsigned x, auto ctrl, auto rows) __attribute__((always_inline)) {
        return static_cast<unsigned>(__builtin_amdgcn_update_dpp(0, static_cast<int>(x), decltype(ctrl)::value, decltype(rows)::value,
                                                                 0xf, false));
    };
    using std::integral_constant;
    v |= step(v, integral_constant<int, 0x111>{}, integral_constant<int, 0xf>{});
    v |= step(v, integral_constant<int, 0x112>{}, integral_constant<int, 0xf>{});
    v |= step(v, integral_constant<int, 0x114>{}, integral_constant<int, 0xf>{});
    v |= step(v, integral_constant<int, 0x118>{}, integral_constant<int, 0xf>{});
    v |= step(v, integral_constant<int, 0x142>{}, integral_constant<int, 0xa>{});
    v |= step(v, integral_constant<int, 0x143>{}, integral_constant<int, 0xc>{});
    return static_cast<unsigned>(__builtin_amdgcn_readlane(static_cast<int>(v), 63));
}

} // namespace rk

#endif
