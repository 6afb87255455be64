// Type-erased entry into the CPU engine of include/rakau_amd/cpu_engine.hpp: maps an rk_cpu_job onto the template
// instantiation for its (ndim, F, UInt, MAC, Q). Included by the translation unit that is compiled for AVX-512
// (rk_cpu_engine.cpp -> librakau_amd_cpu512.so).
#ifndef RK_CPU_ENGINE_IMPL_HPP
#define RK_CPU_ENGINE_IMPL_HPP

#include <stdexcept>

#include "../../include/rakau_amd/tree.hpp"

namespace rk_cpu
{

using namespace rakau_amd;

template <std::size_t ND, typename F, typename UInt, mac M, unsigned Q>
void run_typed(const rk_cpu_job &j)
{
    using node_t = tree_node_t<ND, F, UInt, M>;
    using cnode_t = tree_cnode_t<F, UInt>;
    std::array<const F *, ND + 1u> parts;
    for (std::size_t k = 0; k <= ND; ++k) {
        parts[k] = static_cast<const F *>(j.parts[k]);
    }
    F *out[ND + 1u] = {};
    for (std::size_t k = 0; k < tree_nvecs_res<Q, ND>; ++k) {
        out[k] = static_cast<F *>(j.out[k]);
    }
    detail::cpu::run<Q, ND, M == mac::bh>(static_cast<const node_t *>(j.tree), static_cast<std::size_t>(j.tree_size),
                                          static_cast<const cnode_t *>(j.crit), static_cast<std::size_t>(j.c_begin),
                                          static_cast<std::size_t>(j.c_end), parts, out, static_cast<F>(j.mac_value),
                                          static_cast<F>(j.G), static_cast<F>(j.eps2), static_cast<cpu_flavour>(j.flavour),
                                          j.nthreads);
}

template <std::size_t ND, typename F, typename UInt, mac M>
void run_q(const rk_cpu_job &j)
{
    switch (j.q) {
        case 0: run_typed<ND, F, UInt, M, 0>(j); break;
        case 1: run_typed<ND, F, UInt, M, 1>(j); break;
        case 2: run_typed<ND, F, UInt, M, 2>(j); break;
        default: throw std::invalid_argument("q must be 0, 1 or 2");
    }
}

template <std::size_t ND, typename F, typename UInt>
void run_mac(const rk_cpu_job &j)
{
    if (j.mac == RK_MAC_BH) {
        run_q<ND, F, UInt, mac::bh>(j);
    } else if (j.mac == RK_MAC_BH_GEOM) {
        run_q<ND, F, UInt, mac::bh_geom>(j);
    } else {
        throw std::invalid_argument("invalid mac");
    }
}

template <std::size_t ND, typename F>
void run_bits(const rk_cpu_job &j)
{
    if (j.code_bits == 64) {
        run_mac<ND, F, std::uint64_t>(j);
    } else if (j.code_bits == 32) {
        run_mac<ND, F, std::uint32_t>(j);
    } else {
        throw std::invalid_argument("code_bits must be 32 or 64");
    }
}

inline void run_job(const rk_cpu_job &j)
{
    if (j.ndim != 2 && j.ndim != 3) {
        throw std::invalid_argument("ndim must be 2 or 3");
    }
    if (j.fp == RK_F32) {
        j.ndim == 3 ? run_bits<3, float>(j) : run_bits<2, float>(j);
    } else if (j.fp == RK_F64) {
        j.ndim == 3 ? run_bits<3, double>(j) : run_bits<2, double>(j);
    } else {
        throw std::invalid_argument("invalid fp");
    }
}

} // namespace rk_cpu

#endif
