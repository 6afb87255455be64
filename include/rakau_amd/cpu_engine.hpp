// CPU traversal engine of rakau_amd::tree: the host share of a `split` acc/pot call.
//
// The reference runs the CPU share of kwargs::split on its TBB + xsimd engine while the accelerator works on the rest
// (include/rakau/tree.hpp:3047-3113 of the reference). This is the same engine re-designed around interaction lists:
// per critical node (the unit of work of tree.hpp:2871-3022) a depth-first walk with skip pointers (tree.hpp:2798-2849)
// takes the all-targets MAC decision of tree.hpp:2662-2672 on SIMD batches of targets, and instead of updating the
// result arrays once per visited node (tree.hpp:2477-2590, 2327-2471) it appends the accepted nodes and the particles
// of the opened leaves to a source list in traversal order. The list is then evaluated with the targets of a batch
// and their accumulators held in registers. Every target still receives its contributions in the reference's order
// with the reference's arithmetic, so the `exact` flavours reproduce the reference's scalar results bit for bit
// outside the critical node (inside it the SIMD flavours sum in a different order).
//
// Flavours (cpu_flavour): automatic = widest SIMD compiled in, with rsqrt + one Newton step for fp32 (the reference's
// AVX fast path, detail/simd.hpp:76-146); simd_exact = SIMD with sqrt + divide; scalar = one target at a time with
// the arithmetic and the summation order of the reference's RAKAU_DISABLE_SIMD branch everywhere.
#ifndef RAKAU_AMD_CPU_ENGINE_HPP
#define RAKAU_AMD_CPU_ENGINE_HPP

#include <algorithm>
#include <array>
#include <atomic>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <exception>
#include <fstream>
#include <mutex>
#include <thread>
#include <type_traits>
#include <vector>

#if defined(__AVX2__) && defined(__FMA__)
#include <immintrin.h>
#define RAKAU_AMD_CPU_AVX2 1
#if defined(__AVX512F__) && defined(__AVX512DQ__)
#define RAKAU_AMD_CPU_AVX512 1
#endif
#endif
// The engine's templates live in an inline namespace named after the instruction set they were compiled for, so that
// objects built with different -m flags (the AVX-512 flavour of librakau_amd is a separate shared object) never share
// a symbol.
#if defined(RAKAU_AMD_CPU_AVX512)
#define RAKAU_AMD_CPU_ISA isa_avx512
#elif defined(RAKAU_AMD_CPU_AVX2)
#define RAKAU_AMD_CPU_ISA isa_avx2
#else
#define RAKAU_AMD_CPU_ISA isa_generic
#endif
#if defined(__linux__)
#include <sched.h>
#endif

namespace rakau_amd
{
inline namespace detail
{

// Host threads a parallel section should use: hardware threads, narrowed by the affinity mask and the cgroup v2 CPU
// quota (a container on a 256-thread host may own 16 cores), overridable with RAKAU_AMD_NUM_THREADS.
inline unsigned usable_hw_threads()
{
    static const unsigned cached = [] {
        if (const char *e = std::getenv("RAKAU_AMD_NUM_THREADS")) {
            const long v = std::atol(e);
            if (v > 0) {
                return static_cast<unsigned>(std::min<long>(v, 1024));
            }
        }
        unsigned n = std::thread::hardware_concurrency();
        n = n ? n : 1u;
#if defined(__linux__)
        cpu_set_t set;
        if (sched_getaffinity(0, sizeof(set), &set) == 0) {
            const int c = CPU_COUNT(&set);
            if (c > 0) {
                n = std::min(n, static_cast<unsigned>(c));
            }
        }
        std::ifstream f("/sys/fs/cgroup/cpu.max");
        std::string quota;
        double period = 0;
        if (f >> quota >> period && quota != "max" && period > 0) {
            const double q = std::atof(quota.c_str()) / period;
            if (q >= 1.) {
                n = std::min(n, static_cast<unsigned>(std::ceil(q)));
            }
        }
#endif
        return std::max(1u, n);
    }();
    return cached;
}

enum class cpu_flavour : int { automatic = 0, scalar = 1, simd_exact = 2 };

namespace cpu
{
inline namespace RAKAU_AMD_CPU_ISA
{

// ---- batches of W targets --------------------------------------------------------------------------------------
template <typename F, int W>
struct batch;

template <typename F>
struct batch<F, 1> {
    static constexpr int size = 1;
    F v;
    static batch load(const F *p)
    {
        return {*p};
    }
    static batch set1(F x)
    {
        return {x};
    }
    void store(F *p) const
    {
        *p = v;
    }
    friend batch operator+(batch a, batch b)
    {
        return {a.v + b.v};
    }
    friend batch operator-(batch a, batch b)
    {
        return {a.v - b.v};
    }
    friend batch operator*(batch a, batch b)
    {
        return {a.v * b.v};
    }
    friend batch operator/(batch a, batch b)
    {
        return {a.v / b.v};
    }
    static batch fma(batch a, batch b, batch c)
    {
        return {std::fma(a.v, b.v, c.v)};
    }
    static batch fnma(batch a, batch b, batch c) // -(a * b) + c
    {
        return {std::fma(-a.v, b.v, c.v)};
    }
    static batch sqrt(batch a)
    {
        return {std::sqrt(a.v)};
    }
    static batch rsqrt(batch a)
    {
        return {F(1) / std::sqrt(a.v)};
    }
    static bool any_ge(batch a, batch b) // any lane with a >= b
    {
        return a.v >= b.v;
    }
    // Lanes whose index (first + lane) equals j get `other`.
    static batch select_index(batch a, batch other, std::size_t first, std::size_t j)
    {
        return first == j ? other : a;
    }
};

#if defined(RAKAU_AMD_CPU_AVX2)
template <>
struct batch<float, 8> {
    static constexpr int size = 8;
    __m256 v;
    static batch load(const float *p)
    {
        return {_mm256_loadu_ps(p)};
    }
    static batch set1(float x)
    {
        return {_mm256_set1_ps(x)};
    }
    void store(float *p) const
    {
        _mm256_storeu_ps(p, v);
    }
    friend batch operator+(batch a, batch b)
    {
        return {_mm256_add_ps(a.v, b.v)};
    }
    friend batch operator-(batch a, batch b)
    {
        return {_mm256_sub_ps(a.v, b.v)};
    }
    friend batch operator*(batch a, batch b)
    {
        return {_mm256_mul_ps(a.v, b.v)};
    }
    friend batch operator/(batch a, batch b)
    {
        return {_mm256_div_ps(a.v, b.v)};
    }
    static batch fma(batch a, batch b, batch c)
    {
        return {_mm256_fmadd_ps(a.v, b.v, c.v)};
    }
    static batch fnma(batch a, batch b, batch c)
    {
        return {_mm256_fnmadd_ps(a.v, b.v, c.v)};
    }
    static batch sqrt(batch a)
    {
        return {_mm256_sqrt_ps(a.v)};
    }
    // 1/sqrt: the hardware estimate (12 bits) refined by one Newton step, y <- y * (1.5 - 0.5 * x * y * y): ~22-23 good
    // bits, the numerics of the reference's fast path (detail/simd.hpp:76-146).
    static batch rsqrt(batch a)
    {
        const __m256 y = _mm256_rsqrt_ps(a.v);
        const __m256 hx = _mm256_mul_ps(a.v, _mm256_set1_ps(0.5f));
        const __m256 t = _mm256_fnmadd_ps(_mm256_mul_ps(hx, y), y, _mm256_set1_ps(1.5f));
        return {_mm256_mul_ps(y, t)};
    }
    static bool any_ge(batch a, batch b)
    {
        return _mm256_movemask_ps(_mm256_cmp_ps(a.v, b.v, _CMP_GE_OQ)) != 0;
    }
    static batch select_index(batch a, batch other, std::size_t first, std::size_t j)
    {
        if (j < first || j >= first + 8u) {
            return a;
        }
        const __m256i lane = _mm256_setr_epi32(0, 1, 2, 3, 4, 5, 6, 7);
        const __m256 m = _mm256_castsi256_ps(_mm256_cmpeq_epi32(lane, _mm256_set1_epi32(static_cast<int>(j - first))));
        return {_mm256_blendv_ps(a.v, other.v, m)};
    }
};

template <>
struct batch<double, 4> {
    static constexpr int size = 4;
    __m256d v;
    static batch load(const double *p)
    {
        return {_mm256_loadu_pd(p)};
    }
    static batch set1(double x)
    {
        return {_mm256_set1_pd(x)};
    }
    void store(double *p) const
    {
        _mm256_storeu_pd(p, v);
    }
    friend batch operator+(batch a, batch b)
    {
        return {_mm256_add_pd(a.v, b.v)};
    }
    friend batch operator-(batch a, batch b)
    {
        return {_mm256_sub_pd(a.v, b.v)};
    }
    friend batch operator*(batch a, batch b)
    {
        return {_mm256_mul_pd(a.v, b.v)};
    }
    friend batch operator/(batch a, batch b)
    {
        return {_mm256_div_pd(a.v, b.v)};
    }
    static batch fma(batch a, batch b, batch c)
    {
        return {_mm256_fmadd_pd(a.v, b.v, c.v)};
    }
    static batch fnma(batch a, batch b, batch c)
    {
        return {_mm256_fnmadd_pd(a.v, b.v, c.v)};
    }
    static batch sqrt(batch a)
    {
        return {_mm256_sqrt_pd(a.v)};
    }
    static batch rsqrt(batch a)
    {
        return {_mm256_div_pd(_mm256_set1_pd(1.), _mm256_sqrt_pd(a.v))};
    }
    static bool any_ge(batch a, batch b)
    {
        return _mm256_movemask_pd(_mm256_cmp_pd(a.v, b.v, _CMP_GE_OQ)) != 0;
    }
    static batch select_index(batch a, batch other, std::size_t first, std::size_t j)
    {
        if (j < first || j >= first + 4u) {
            return a;
        }
        const __m256i lane = _mm256_setr_epi64x(0, 1, 2, 3);
        const __m256d m
            = _mm256_castsi256_pd(_mm256_cmpeq_epi64(lane, _mm256_set1_epi64x(static_cast<long long>(j - first))));
        return {_mm256_blendv_pd(a.v, other.v, m)};
    }
};
#endif

#if defined(RAKAU_AMD_CPU_AVX512)
template <>
struct batch<float, 16> {
    static constexpr int size = 16;
    __m512 v;
    static batch load(const float *p)
    {
        return {_mm512_loadu_ps(p)};
    }
    static batch set1(float x)
    {
        return {_mm512_set1_ps(x)};
    }
    void store(float *p) const
    {
        _mm512_storeu_ps(p, v);
    }
    friend batch operator+(batch a, batch b)
    {
        return {_mm512_add_ps(a.v, b.v)};
    }
    friend batch operator-(batch a, batch b)
    {
        return {_mm512_sub_ps(a.v, b.v)};
    }
    friend batch operator*(batch a, batch b)
    {
        return {_mm512_mul_ps(a.v, b.v)};
    }
    friend batch operator/(batch a, batch b)
    {
        return {_mm512_div_ps(a.v, b.v)};
    }
    static batch fma(batch a, batch b, batch c)
    {
        return {_mm512_fmadd_ps(a.v, b.v, c.v)};
    }
    static batch fnma(batch a, batch b, batch c)
    {
        return {_mm512_fnmadd_ps(a.v, b.v, c.v)};
    }
    static batch sqrt(batch a)
    {
        return {_mm512_sqrt_ps(a.v)};
    }
    // vrsqrt14ps (14 bits) + one Newton step: the reference's AVX-512 fast path (detail/simd.hpp:110-146).
    static batch rsqrt(batch a)
    {
        const __m512 y = _mm512_rsqrt14_ps(a.v);
        const __m512 hx = _mm512_mul_ps(a.v, _mm512_set1_ps(0.5f));
        const __m512 t = _mm512_fnmadd_ps(_mm512_mul_ps(hx, y), y, _mm512_set1_ps(1.5f));
        return {_mm512_mul_ps(y, t)};
    }
    static bool any_ge(batch a, batch b)
    {
        return _mm512_cmp_ps_mask(a.v, b.v, _CMP_GE_OQ) != 0;
    }
    static batch select_index(batch a, batch other, std::size_t first, std::size_t j)
    {
        if (j < first || j >= first + 16u) {
            return a;
        }
        return {_mm512_mask_blend_ps(static_cast<__mmask16>(1u << (j - first)), a.v, other.v)};
    }
};

template <>
struct batch<double, 8> {
    static constexpr int size = 8;
    __m512d v;
    static batch load(const double *p)
    {
        return {_mm512_loadu_pd(p)};
    }
    static batch set1(double x)
    {
        return {_mm512_set1_pd(x)};
    }
    void store(double *p) const
    {
        _mm512_storeu_pd(p, v);
    }
    friend batch operator+(batch a, batch b)
    {
        return {_mm512_add_pd(a.v, b.v)};
    }
    friend batch operator-(batch a, batch b)
    {
        return {_mm512_sub_pd(a.v, b.v)};
    }
    friend batch operator*(batch a, batch b)
    {
        return {_mm512_mul_pd(a.v, b.v)};
    }
    friend batch operator/(batch a, batch b)
    {
        return {_mm512_div_pd(a.v, b.v)};
    }
    static batch fma(batch a, batch b, batch c)
    {
        return {_mm512_fmadd_pd(a.v, b.v, c.v)};
    }
    static batch fnma(batch a, batch b, batch c)
    {
        return {_mm512_fnmadd_pd(a.v, b.v, c.v)};
    }
    static batch sqrt(batch a)
    {
        return {_mm512_sqrt_pd(a.v)};
    }
    static batch rsqrt(batch a)
    {
        return {_mm512_div_pd(_mm512_set1_pd(1.), _mm512_sqrt_pd(a.v))};
    }
    static bool any_ge(batch a, batch b)
    {
        return _mm512_cmp_pd_mask(a.v, b.v, _CMP_GE_OQ) != 0;
    }
    static batch select_index(batch a, batch other, std::size_t first, std::size_t j)
    {
        if (j < first || j >= first + 8u) {
            return a;
        }
        return {_mm512_mask_blend_pd(static_cast<__mmask8>(1u << (j - first)), a.v, other.v)};
    }
};
#endif

// Widest batch compiled in for F.
template <typename F>
inline constexpr int native_width =
#if defined(RAKAU_AMD_CPU_AVX512)
    std::is_same_v<F, float> ? 16 : 8;
#elif defined(RAKAU_AMD_CPU_AVX2)
    std::is_same_v<F, float> ? 8 : 4;
#else
    1;
#endif

// ---- source list -----------------------------------------------------------------------------------------------
// Sources in traversal order. Runs of equal kind are evaluated by a loop specialised for the kind, because the two
// kinds round differently with softening, exactly as in the reference: an accepted node adds eps2 AFTER the
// unsoftened distance that its MAC test used (tree.hpp:2662-2700), a leaf particle starts the sum of squares from
// eps2 (tree.hpp:2432-2470).
enum : unsigned { kind_node = 0, kind_particle = 1, kind_self = 2 };

template <typename F, std::size_t NDim>
struct source_list {
    static constexpr std::size_t cap = 1024;
    F c[NDim + 1u][cap]; // coordinates, then mass
    std::size_t n = 0;
    struct run {
        unsigned kind;
        std::size_t begin, end;
    };
    std::vector<run> runs;
    void push(unsigned kind, const F *vals) // vals = NDim coordinates, mass
    {
        for (std::size_t j = 0; j <= NDim; ++j) {
            c[j][n] = vals[j];
        }
        if (runs.empty() || runs.back().kind != kind) {
            runs.push_back(run{kind, n, n + 1u});
        } else {
            runs.back().end = n + 1u;
        }
        ++n;
    }
    void clear()
    {
        n = 0;
        runs.clear();
    }
};

// One source against one batch of targets. RSQ: rsqrt-based arithmetic (fp32 fast path) instead of sqrt + divide.
template <unsigned Q, std::size_t NDim, unsigned Kind, bool RSQ, typename B, typename F>
inline void interact(const B (&t)[NDim + 1u], B (&acc)[NDim + 1u], const F *const (&src)[NDim + 1u], std::size_t s, F eps2,
                     std::size_t first_target, std::size_t self_target)
{
    B diff[NDim];
    B d2 = Kind == kind_node ? B::set1(F(0)) : B::set1(eps2);
    for (std::size_t j = 0; j < NDim; ++j) {
        diff[j] = B::set1(src[j][s]) - t[j];
        d2 = B::fma(diff[j], diff[j], d2);
    }
    if constexpr (Kind == kind_node) {
        d2 = d2 + B::set1(eps2);
    }
    B ms = B::set1(src[NDim][s]);
    if constexpr (Kind == kind_self) {
        // The group's own particles: lane == source is masked out (zero mass at a harmless distance).
        d2 = B::select_index(d2, B::set1(F(1)), first_target, self_target);
        ms = B::select_index(ms, B::set1(F(0)), first_target, self_target);
    }
    constexpr std::size_t pot_idx = Q == 1u ? 0u : NDim;
    if constexpr (RSQ) {
        const B rinv = B::rsqrt(d2), mr = ms * rinv;
        if constexpr (Q == 0u || Q == 2u) {
            const B mr3 = mr * (rinv * rinv);
            for (std::size_t j = 0; j < NDim; ++j) {
                acc[j] = B::fma(diff[j], mr3, acc[j]);
            }
        }
        if constexpr (Q == 1u || Q == 2u) {
            acc[pot_idx] = B::fnma(t[NDim], mr, acc[pot_idx]);
        }
    } else {
        // tree.hpp:2564-2589 / 2432-2470: dist = sqrt(dist2); m / (dist * dist2); m / dist.
        const B dist = B::sqrt(d2);
        if constexpr (Q == 0u || Q == 2u) {
            const B m_dist3 = ms / (dist * d2);
            for (std::size_t j = 0; j < NDim; ++j) {
                acc[j] = B::fma(diff[j], m_dist3, acc[j]);
            }
        }
        if constexpr (Q == 1u || Q == 2u) {
            acc[pot_idx] = B::fnma(t[NDim], ms / dist, acc[pot_idx]);
        }
    }
}

// Per-thread scratch: targets and results of one critical node (padded to whole batches) and the source list.
template <typename F, std::size_t NDim>
struct scratch {
    std::vector<F> tgt[NDim + 1u], res[NDim + 1u];
    source_list<F, NDim> list;
};

// Evaluate the list against all targets: accumulators of a batch stay in registers over the whole list.
template <unsigned Q, std::size_t NDim, bool RSQ, typename B, typename F>
inline void flush(scratch<F, NDim> &s, std::size_t padded, F eps2)
{
    constexpr std::size_t nres = Q == 0u ? NDim : (Q == 1u ? 1u : NDim + 1u);
    auto &L = s.list;
    if (!L.n) {
        return;
    }
    const F *src[NDim + 1u];
    for (std::size_t j = 0; j <= NDim; ++j) {
        src[j] = L.c[j];
    }
    const F *const(&csrc)[NDim + 1u] = src;
    for (std::size_t b = 0; b < padded; b += static_cast<std::size_t>(B::size)) {
        B t[NDim + 1u], acc[NDim + 1u];
        for (std::size_t j = 0; j <= NDim; ++j) {
            t[j] = B::load(s.tgt[j].data() + b);
        }
        for (std::size_t j = 0; j < nres; ++j) {
            acc[j] = B::load(s.res[j].data() + b);
        }
        for (const auto &r : L.runs) {
            if (r.kind == kind_node) {
                for (std::size_t i = r.begin; i < r.end; ++i) {
                    interact<Q, NDim, kind_node, RSQ>(t, acc, csrc, i, eps2, b, 0);
                }
            } else if (r.kind == kind_particle) {
                for (std::size_t i = r.begin; i < r.end; ++i) {
                    interact<Q, NDim, kind_particle, RSQ>(t, acc, csrc, i, eps2, b, 0);
                }
            } else {
                // The group's own particles in target order: source i of the run is target i - r.begin.
                for (std::size_t i = r.begin; i < r.end; ++i) {
                    interact<Q, NDim, kind_self, RSQ>(t, acc, csrc, i, eps2, b, i - r.begin);
                }
            }
        }
        for (std::size_t j = 0; j < nres; ++j) {
            acc[j].store(s.res[j].data() + b);
        }
    }
    L.clear();
}

// Interactions inside the critical node with the arithmetic AND the summation order of the reference's scalar branch
// (tree.hpp:2258-2320): symmetric pairs, the i2 > i1 part of a target summed apart and added at the end.
template <unsigned Q, std::size_t NDim, typename F>
inline void self_interactions_scalar(F eps2, std::size_t tgt_size, const F *const *p, F *const *res)
{
    constexpr std::size_t nres = Q == 0u ? NDim : (Q == 1u ? 1u : NDim + 1u);
    constexpr std::size_t pot_idx = Q == 1u ? 0u : NDim;
    const F *m_ptr = p[NDim];
    F diffs[NDim], pos1[NDim];
    for (std::size_t i1 = 0; i1 < tgt_size; ++i1) {
        for (std::size_t j = 0; j < NDim; ++j) {
            pos1[j] = p[j][i1];
        }
        const F m1 = m_ptr[i1];
        F a1[nres];
        for (std::size_t j = 0; j < nres; ++j) {
            a1[j] = F(0);
        }
        for (std::size_t i2 = i1 + 1u; i2 < tgt_size; ++i2) {
            F dist2(eps2);
            for (std::size_t j = 0; j < NDim; ++j) {
                diffs[j] = p[j][i2] - pos1[j];
                dist2 = std::fma(diffs[j], diffs[j], dist2);
            }
            const F dist = std::sqrt(dist2), m2 = m_ptr[i2];
            if constexpr (Q == 0u || Q == 2u) {
                const F dist3 = dist2 * dist, m2_dist3 = m2 / dist3, m1_dist3 = m1 / dist3;
                for (std::size_t j = 0; j < NDim; ++j) {
                    a1[j] = std::fma(m2_dist3, diffs[j], a1[j]);
                    res[j][i2] = std::fma(m1_dist3, -diffs[j], res[j][i2]);
                }
            }
            if constexpr (Q == 1u || Q == 2u) {
                const F mut_pot = m1 / dist * m2;
                a1[pot_idx] -= mut_pot;
                res[pot_idx][i2] -= mut_pot;
            }
        }
        for (std::size_t j = 0; j < nres; ++j) {
            res[j][i1] += a1[j];
        }
    }
}

// One critical node: walk, lists, self interactions, G, output (tree.hpp:2798-2849, 2871-3022).
template <unsigned Q, std::size_t NDim, bool BH, bool RSQ, typename B, typename F, typename Node, typename CNode>
inline void run_group(scratch<F, NDim> &s, const Node *tree, std::size_t tree_size, const CNode &cn,
                      const std::array<const F *, NDim + 1u> &parts, F *const *out, F mac_value, F G, F eps2)
{
    constexpr std::size_t nres = Q == 0u ? NDim : (Q == 1u ? 1u : NDim + 1u);
    constexpr std::size_t W = static_cast<std::size_t>(B::size);
    const std::size_t tgt_begin = static_cast<std::size_t>(cn.begin), T = static_cast<std::size_t>(cn.end) - tgt_begin;
    const std::size_t padded = (T + W - 1u) / W * W;
    // Targets, padded with copies of the last one: a copy takes the same MAC decisions as the original (the padding
    // rule of tree.hpp:2875-2973 serves the same purpose) and its results are never stored.
    for (std::size_t j = 0; j <= NDim; ++j) {
        s.tgt[j].resize(padded);
        std::copy(parts[j] + tgt_begin, parts[j] + tgt_begin + T, s.tgt[j].data());
        std::fill(s.tgt[j].data() + T, s.tgt[j].data() + padded, s.tgt[j][T - 1u]);
    }
    for (std::size_t j = 0; j < nres; ++j) {
        s.res[j].assign(padded, F(0));
    }
    auto &L = s.list;
    L.clear();
    const std::uint64_t tgt_code = static_cast<std::uint64_t>(cn.code);
    const unsigned tgt_level = (63u - static_cast<unsigned>(__builtin_clzll(tgt_code))) / static_cast<unsigned>(NDim);
    F vals[NDim + 1u];
    for (std::size_t src_idx = 0; src_idx < tree_size;) {
        const Node &src = tree[src_idx];
        const std::uint64_t src_code = static_cast<std::uint64_t>(src.code);
        const unsigned src_level = static_cast<unsigned>(src.level);
        const std::size_t n_children = static_cast<std::size_t>(src.n_children);
        // Ancestor-or-self of the target node (tree.hpp:2828-2838; the level test is explicit here).
        if (src_level <= tgt_level && (tgt_code >> ((tgt_level - src_level) * static_cast<unsigned>(NDim))) == src_code) {
            src_idx += 1u + (src_code == tgt_code ? n_children : 0u);
            continue;
        }
        // tree.hpp:2632-2642.
        F mac_lh;
        if constexpr (BH) {
            mac_lh = src.dim2 * mac_value;
        } else {
            const F t = std::fma(src.dim, mac_value, src.delta);
            mac_lh = t * t;
        }
        // MAC: every target must have mac_lh < dist2 (unsoftened), tree.hpp:2662-2672.
        bool accept = true;
        const B lh = B::set1(mac_lh);
        for (std::size_t b = 0; b < padded; b += W) {
            B d2 = B::set1(F(0));
            for (std::size_t j = 0; j < NDim; ++j) {
                const B diff = B::set1(src.props[j]) - B::load(s.tgt[j].data() + b);
                d2 = B::fma(diff, diff, d2);
            }
            if (B::any_ge(lh, d2)) {
                accept = false;
                break;
            }
        }
        if (accept) {
            if (L.n == L.cap) {
                flush<Q, NDim, RSQ, B>(s, padded, eps2);
            }
            L.push(kind_node, src.props);
            src_idx += n_children + 1u;
            continue;
        }
        if (!n_children) {
            // Opened leaf: all its particles become sources (tree.hpp:2327-2471).
            for (std::size_t i = static_cast<std::size_t>(src.begin); i < static_cast<std::size_t>(src.end); ++i) {
                if (L.n == L.cap) {
                    flush<Q, NDim, RSQ, B>(s, padded, eps2);
                }
                for (std::size_t j = 0; j <= NDim; ++j) {
                    vals[j] = parts[j][i];
                }
                L.push(kind_particle, vals);
            }
        }
        ++src_idx;
    }
    F *res[NDim + 1u] = {};
    for (std::size_t j = 0; j < nres; ++j) {
        res[j] = s.res[j].data();
    }
    flush<Q, NDim, RSQ, B>(s, padded, eps2);
    if (W == 1u || T > L.cap) {
        const F *p[NDim + 1u];
        for (std::size_t j = 0; j <= NDim; ++j) {
            p[j] = s.tgt[j].data();
        }
        self_interactions_scalar<Q, NDim>(eps2, T, p, res);
    } else {
        // The group's own particles as one more run of sources with the self pair masked (tree.hpp:2073-2321
        // evaluates the same pairs symmetrically).
        for (std::size_t i = 0; i < T; ++i) {
            for (std::size_t j = 0; j <= NDim; ++j) {
                vals[j] = s.tgt[j][i];
            }
            L.push(kind_self, vals);
        }
        flush<Q, NDim, RSQ, B>(s, padded, eps2);
    }
    // G is the last multiply (tree.hpp:2986-3002), then the results leave in Morton order (3004-3007).
    for (std::size_t j = 0; j < nres; ++j) {
        F *o = out[j] + tgt_begin;
        if (G != F(1)) {
            for (std::size_t k = 0; k < T; ++k) {
                o[k] = res[j][k] * G;
            }
        } else {
            std::copy(res[j], res[j] + T, o);
        }
    }
}

// Critical nodes [c_begin, c_end) on `nthreads` host threads (dynamic chunks: core groups cost several times more than
// halo groups). out[j] address element 0 of full-size Morton-order arrays.
template <unsigned Q, std::size_t NDim, bool BH, typename F, typename Node, typename CNode>
inline void run(const Node *tree, std::size_t tree_size, const CNode *crit, std::size_t c_begin, std::size_t c_end,
                const std::array<const F *, NDim + 1u> &parts, F *const *out, F mac_value, F G, F eps2,
                cpu_flavour flavour, unsigned nthreads)
{
    if (c_begin >= c_end) {
        return;
    }
    nthreads = nthreads ? nthreads : 1u; // resolved by the caller (tree.hpp: usable_hw_threads())
    nthreads = static_cast<unsigned>(std::min<std::size_t>(nthreads, (c_end - c_begin + 7u) / 8u));
    std::atomic<std::size_t> next(c_begin);
    std::exception_ptr ep;
    std::mutex mx;
    auto worker = [&]() {
        try {
            scratch<F, NDim> s;
            constexpr std::size_t chunk = 8;
            for (;;) {
                const std::size_t b = next.fetch_add(chunk);
                if (b >= c_end) {
                    break;
                }
                const std::size_t e = std::min(c_end, b + chunk);
                for (std::size_t ci = b; ci < e; ++ci) {
                    constexpr int NW = native_width<F>;
                    if (flavour == cpu_flavour::scalar || NW == 1) {
                        run_group<Q, NDim, BH, false, batch<F, 1>>(s, tree, tree_size, crit[ci], parts, out, mac_value, G,
                                                                   eps2);
                    } else if (flavour == cpu_flavour::simd_exact || !std::is_same_v<F, float>) {
                        run_group<Q, NDim, BH, false, batch<F, NW>>(s, tree, tree_size, crit[ci], parts, out, mac_value, G,
                                                                    eps2);
                    } else {
                        run_group<Q, NDim, BH, true, batch<F, NW>>(s, tree, tree_size, crit[ci], parts, out, mac_value, G,
                                                                   eps2);
                    }
                }
            }
        } catch (...) {
            std::lock_guard<std::mutex> lk(mx);
            if (!ep) {
                ep = std::current_exception();
            }
        }
    };
    if (nthreads <= 1u) {
        worker();
    } else {
        std::vector<std::thread> th;
        for (unsigned t = 1; t < nthreads; ++t) {
            th.emplace_back(worker);
        }
        worker();
        for (auto &t : th) {
            t.join();
        }
    }
    if (ep) {
        std::rethrow_exception(ep);
    }
}

} // namespace RAKAU_AMD_CPU_ISA
} // namespace cpu
} // namespace detail
} // namespace rakau_amd

#endif
