// Kick-drift-kick leapfrog of a Plummer sphere with every array resident in HBM: the loop of the reference's
// benchmark/benchmark_leapfrog.cpp (G = M = 1, equal masses, velocities from the distribution function, clipped at
// 10 core radii, Athanassoula softening 0.45 * N^-0.73; per step: half kick, drift, tree rebuild, accelerations,
// half kick) written against the C ABI of librakau_amd.so only:
//
//   rk_state_build_nd(on_device)      first tree, from device-resident x, y, z, m in the caller's particle order
//   rk_state_rebuild_device           every step: Morton sort + tree on the GPU, buffers recycled
//   rk_acc_pot_device(RK_OUT_ORDERED) traversal; results land at the ORIGINAL particle index (accs_o semantics), so the
//                                     velocity arrays never follow last_perm() (benchmark_leapfrog.cpp:263-281, 372-381)
//
// The integrator kernels are the caller's own (below). Build: make -C examples   Run: examples/leapfrog [options]
//   --nparts N (1000000) --steps K (20) --warmup W (2) --timestep dt (1e-4) --mac_value theta (0.75)
//   --fp_type float|double --mac_type bh|bh_geom --track-integrals
//   --reorder K (8): every K steps the harness permutes its own arrays into the Morton order of the current tree
//                    (0 = never: the arrays keep their initial, random order)
// Prints one JSON line (steps/s, per-step split, optional energy drift).
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "../include/rakau_amd.h"

#define HIP_OK(e)                                                                                                      \
    do {                                                                                                               \
        const hipError_t e_ = (e);                                                                                     \
        if (e_ != hipSuccess) {                                                                                        \
            std::fprintf(stderr, "%s failed: %s\n", #e, hipGetErrorString(e_));                                        \
            std::exit(2);                                                                                              \
        }                                                                                                              \
    } while (0)
#define RK_OK_OR_DIE(e)                                                                                                \
    do {                                                                                                               \
        if ((e) != RK_OK) {                                                                                            \
            std::fprintf(stderr, "%s failed: %s\n", #e, rk_last_error());                                              \
            std::exit(3);                                                                                              \
        }                                                                                                              \
    } while (0)

// v += a * h ; x += v * dt (the kicked velocity is kept: benchmark_leapfrog.cpp:349-369).
template <typename F>
__global__ void k_kick_drift(F *x, F *y, F *z, F *vx, F *vy, F *vz, const F *ax, const F *ay, const F *az, F h, F dt,
                             unsigned n)
{
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const F kx = fma(ax[i], h, vx[i]), ky = fma(ay[i], h, vy[i]), kz = fma(az[i], h, vz[i]);
        vx[i] = kx, vy[i] = ky, vz[i] = kz;
        x[i] = fma(kx, dt, x[i]), y[i] = fma(ky, dt, y[i]), z[i] = fma(kz, dt, z[i]);
    }
}
template <typename F>
__global__ void k_kick(F *vx, F *vy, F *vz, const F *ax, const F *ay, const F *az, F h, unsigned n)
{
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        vx[i] = fma(ax[i], h, vx[i]), vy[i] = fma(ay[i], h, vy[i]), vz[i] = fma(az[i], h, vz[i]);
    }
}
// dst[i] = src[perm[i]] for seven arrays at once: puts the caller's arrays into the Morton order of the last tree, so
// that the next rebuild reads them (and the ordered outputs are written) almost sequentially -- the device-side
// counterpart of the reference keeping its particles in tree order (benchmark_leapfrog.cpp:263-281).
template <typename F>
__global__ void k_reorder(const unsigned *perm, unsigned n, const F *const *src, F *const *dst)
{
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const unsigned j = perm[i];
#pragma unroll
        for (int k = 0; k < 7; ++k) {
            dst[k][i] = src[k][j];
        }
    }
}
// Partial sums of the kinetic energy and of the mutual potential energies (double accumulation, fixed order).
template <typename F>
__global__ void k_energy(const F *vx, const F *vy, const F *vz, const F *m, const F *pot, unsigned n, double *partial)
{
    __shared__ double sk[256], sw[256];
    double k = 0., w = 0.;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const double v2 = double(vx[i]) * vx[i] + double(vy[i]) * vy[i] + double(vz[i]) * vz[i];
        k += 0.5 * double(m[i]) * v2;
        w += 0.5 * double(pot[i]);
    }
    sk[threadIdx.x] = k, sw[threadIdx.x] = w;
    __syncthreads();
    for (unsigned s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) {
            sk[threadIdx.x] += sk[threadIdx.x + s], sw[threadIdx.x] += sw[threadIdx.x + s];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        partial[2 * blockIdx.x] = sk[0], partial[2 * blockIdx.x + 1] = sw[0];
    }
}

struct options {
    unsigned long nparts = 1000000;
    int steps = 20, warmup = 2, reorder = 8;
    double timestep = 1e-4, theta = 0.75, a = 1.;
    bool f64 = false, geom = false, integrals = false;
};

// Plummer model with velocities (benchmark_leapfrog.cpp:50-104), clipped at 10 a (:190-214).
template <typename F>
static void plummer(const options &o, std::vector<F> (&pos)[3], std::vector<F> (&vel)[3])
{
    std::mt19937_64 rng(0);
    std::uniform_real_distribution<double> u01(0., 1.), u11(-1., 1.), uphi(0., 2. * M_PI), uq(0., 0.1);
    for (unsigned long i = 0; i < o.nparts; ++i) {
        const double r = o.a / std::sqrt(std::pow(u01(rng), -2. / 3.) - 1.);
        double th = std::acos(u11(rng)), ph = uphi(rng);
        const double x = r * std::sin(th) * std::cos(ph), y = r * std::sin(th) * std::sin(ph), z = r * std::cos(th);
        double q = 0., g = 0.1;
        while (g > q * q * std::pow(1. - q * q, 3.5)) {
            q = u01(rng), g = uq(rng);
        }
        const double v = q * std::sqrt(2. / o.a) * std::pow(1. + r * r / (o.a * o.a), -0.25);
        th = std::acos(u11(rng)), ph = uphi(rng);
        if (x * x + y * y + z * z < 100. * o.a * o.a) {
            pos[0].push_back(F(x)), pos[1].push_back(F(y)), pos[2].push_back(F(z));
            vel[0].push_back(F(v * std::sin(th) * std::cos(ph))), vel[1].push_back(F(v * std::sin(th) * std::sin(ph))),
                vel[2].push_back(F(v * std::cos(th)));
        }
    }
}

template <typename F>
static int run(const options &o)
{
    std::vector<F> hpos[3], hvel[3];
    plummer<F>(o, hpos, hvel);
    const auto n = static_cast<unsigned>(hpos[0].size());
    const double eps = 0.45 * std::pow(double(n), -0.73); // benchmark_leapfrog.cpp:221
    std::vector<F> hm(n, F(1) / F(n));
    const size_t bytes = size_t(n) * sizeof(F);
    F *pos[3], *vel[3], *mass, *out[4];
    for (int k = 0; k < 3; ++k) {
        HIP_OK(hipMalloc(&pos[k], bytes));
        HIP_OK(hipMalloc(&vel[k], bytes));
        HIP_OK(hipMemcpy(pos[k], hpos[k].data(), bytes, hipMemcpyHostToDevice));
        HIP_OK(hipMemcpy(vel[k], hvel[k].data(), bytes, hipMemcpyHostToDevice));
    }
    HIP_OK(hipMalloc(&mass, bytes));
    HIP_OK(hipMemcpy(mass, hm.data(), bytes, hipMemcpyHostToDevice));
    for (auto &p : out) {
        HIP_OK(hipMalloc(&p, bytes));
    }
    double *d_partial = nullptr;
    HIP_OK(hipMalloc(&d_partial, 2 * 256 * sizeof(double)));
    // Double buffers + pointer tables for the periodic reordering of {x, y, z, vx, vy, vz, m}.
    F *alt[7];
    for (auto &p : alt) {
        HIP_OK(hipMalloc(&p, bytes));
    }
    F **d_src = nullptr, **d_dst = nullptr;
    HIP_OK(hipMalloc(&d_src, 7 * sizeof(F *)));
    HIP_OK(hipMalloc(&d_dst, 7 * sizeof(F *)));

    const int q = o.integrals ? 2 : 0;
    const F th = F(o.theta);
    const double mac_value = o.geom ? double(F(1) / th) : double(F(1) / (th * th));
    const double eps2 = double(F(eps) * F(eps));
    const void *parts[4] = {pos[0], pos[1], pos[2], mass};
    void *outs[4] = {out[0], out[1], out[2], out[3]};
    int step_no = 0;
    rk_state *st = nullptr;
    RK_OK_OR_DIE(rk_state_build_nd(&st, 3, o.f64 ? RK_F64 : RK_F32, o.geom ? RK_MAC_BH_GEOM : RK_MAC_BH, 0, parts, 1, n,
                                   0., 16, 128));
    RK_OK_OR_DIE(rk_state_set_timing(st, 0)); // this loop never asks for rk_last_kernel_ms()
    auto accs = [&] { RK_OK_OR_DIE(rk_acc_pot_device(st, q, 0, n, outs, mac_value, 1., eps2, RK_OUT_ORDERED, nullptr)); };
    auto energy = [&](double &kin, double &pot) {
        hipLaunchKernelGGL((k_energy<F>), dim3(256), dim3(256), 0, nullptr, vel[0], vel[1], vel[2], mass, out[3], n, d_partial);
        double h[512];
        HIP_OK(hipMemcpy(h, d_partial, sizeof(h), hipMemcpyDeviceToHost));
        kin = pot = 0.;
        for (int b = 0; b < 256; ++b) {
            kin += h[2 * b], pot += h[2 * b + 1];
        }
    };
    accs();
    double k0 = 0., w0 = 0., k1 = 0., w1 = 0.;
    if (o.integrals) {
        energy(k0, w0);
    }
    const F h = F(o.timestep / 2.), dt = F(o.timestep);
    const dim3 grid((n + 255) / 256), block(256);
    double t_build = 0., t_trav = 0.;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto secs = [](auto a, auto b) { return std::chrono::duration<double>(b - a).count(); };
    auto step = [&](bool timed) {
        hipLaunchKernelGGL((k_kick_drift<F>), grid, block, 0, nullptr, pos[0], pos[1], pos[2], vel[0], vel[1], vel[2], out[0],
                           out[1], out[2], h, dt, n);
        HIP_OK(hipGetLastError());
        if (o.reorder > 0 && ++step_no % o.reorder == 0) {
            // The accelerations of the old order have just been consumed: move everything to tree order.
            void *perm = nullptr;
            int64_t pbytes = 0;
            RK_OK_OR_DIE(rk_state_device_ptr(st, 1, &perm, &pbytes));
            F *cur[7] = {pos[0], pos[1], pos[2], vel[0], vel[1], vel[2], mass};
            // Synchronous copies: the tables live in pageable stack arrays that change right after the launch.
            HIP_OK(hipMemcpy(d_src, cur, sizeof(cur), hipMemcpyHostToDevice));
            HIP_OK(hipMemcpy(d_dst, alt, sizeof(alt), hipMemcpyHostToDevice));
            hipLaunchKernelGGL((k_reorder<F>), grid, block, 0, nullptr, static_cast<const unsigned *>(perm), n, d_src, d_dst);
            HIP_OK(hipGetLastError());
            for (int k = 0; k < 3; ++k) {
                std::swap(pos[k], alt[k]);
                std::swap(vel[k], alt[3 + k]);
            }
            std::swap(mass, alt[6]);
            parts[0] = pos[0], parts[1] = pos[1], parts[2] = pos[2], parts[3] = mass;
        }
        if (timed) {
            HIP_OK(hipDeviceSynchronize());
        }
        const auto t0 = now();
        RK_OK_OR_DIE(rk_state_rebuild_device(st, parts, n, 0.));
        if (timed) {
            HIP_OK(hipDeviceSynchronize());
        }
        const auto t1 = now();
        accs();
        if (timed) {
            HIP_OK(hipDeviceSynchronize());
            t_build += secs(t0, t1), t_trav += secs(t1, now());
        }
        hipLaunchKernelGGL((k_kick<F>), grid, block, 0, nullptr, vel[0], vel[1], vel[2], out[0], out[1], out[2], h, n);
        HIP_OK(hipGetLastError());
    };
    for (int i = 0; i < o.warmup; ++i) {
        step(false);
    }
    HIP_OK(hipDeviceSynchronize());
    size_t free0 = 0, free1 = 0, total = 0;
    HIP_OK(hipMemGetInfo(&free0, &total));
    const auto t0 = now();
    for (int i = 0; i < o.steps; ++i) {
        step(true);
    }
    HIP_OK(hipDeviceSynchronize());
    const double wall = secs(t0, now());
    // The same number of steps again WITHOUT the synchronisations between the phases (which the loop above needs for its per-phase
    // times and a real integrator does not have): what a step costs when nothing but the rebuild's own look-ups stops the host.
    const auto f0 = now();
    for (int i = 0; i < o.steps; ++i) {
        step(false);
    }
    HIP_OK(hipDeviceSynchronize());
    const double wall_free = secs(f0, now());
    HIP_OK(hipMemGetInfo(&free1, &total));
    if (o.integrals) {
        energy(k1, w1);
    }
    int64_t info[8];
    RK_OK_OR_DIE(rk_state_info(st, info));
    std::printf("{\"metric\": \"leapfrog steps/s (KDK, tree rebuilt every step, all arrays resident in HBM; native harness)\", "
                "\"value\": %.3f, \"unit\": \"steps/s\", \"nparts\": %u, \"steps\": %d, \"ms_per_step\": %.4f, "
                "\"ms_rebuild\": %.4f, \"ms_traversal\": %.4f, \"ms_per_step_free_running\": %.4f, \"dtype\": \"%s\", \"theta\": %g, \"timestep\": %g, "
                "\"eps\": %.6g, \"reorder_every\": %d, \"tree_size\": %lld, \"n_crit\": %lld, \"device_mem_growth_mb\": %.1f",
                o.steps / wall, n, o.steps, 1e3 * wall / o.steps, 1e3 * t_build / o.steps, 1e3 * t_trav / o.steps,
                1e3 * wall_free / o.steps, o.f64 ? "f64" : "f32", o.theta, o.timestep, eps, o.reorder, static_cast<long long>(info[1]),
                static_cast<long long>(info[2]), (double(free0) - double(free1)) / 1048576.);
    if (o.integrals) {
        std::printf(", \"energy_start\": %.12g, \"energy_end\": %.12g, \"energy_rel_drift\": %.3e, \"virial_2K_over_W\": %.6f",
                    k0 + w0, k1 + w1, std::fabs((k1 + w1) - (k0 + w0)) / std::fabs(k0 + w0), -2. * k0 / w0);
    }
    std::printf("}\n");
    rk_state_destroy(st);
    for (int k = 0; k < 3; ++k) {
        HIP_OK(hipFree(pos[k]));
        HIP_OK(hipFree(vel[k]));
    }
    HIP_OK(hipFree(mass));
    HIP_OK(hipFree(d_partial));
    for (auto &p : alt) {
        HIP_OK(hipFree(p));
    }
    HIP_OK(hipFree(d_src));
    HIP_OK(hipFree(d_dst));
    for (auto &p : out) {
        HIP_OK(hipFree(p));
    }
    return 0;
}

int main(int argc, char **argv)
{
    options o;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto next = [&] { return i + 1 < argc ? argv[++i] : ""; };
        if (a == "--nparts") o.nparts = std::strtoul(next(), nullptr, 10);
        else if (a == "--steps") o.steps = std::atoi(next());
        else if (a == "--warmup") o.warmup = std::atoi(next());
        else if (a == "--reorder") o.reorder = std::atoi(next());
        else if (a == "--timestep") o.timestep = std::atof(next());
        else if (a == "--mac_value") o.theta = std::atof(next());
        else if (a == "--a") o.a = std::atof(next());
        else if (a == "--fp_type") o.f64 = std::string(next()) == "double";
        else if (a == "--mac_type") o.geom = std::string(next()) == "bh_geom";
        else if (a == "--track-integrals") o.integrals = true;
        else {
            std::fprintf(stderr, "unknown option %s\n", a.c_str());
            return 1;
        }
    }
    if (!rk_has_accelerator()) {
        std::fprintf(stderr, "no gfx950 accelerator available\n");
        return 4;
    }
    return o.f64 ? run<double>(o) : run<float>(o);
}
