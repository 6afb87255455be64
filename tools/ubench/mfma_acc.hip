// Microbenchmark behind the round-6 experiment "accumulation on the matrix pipe" (VERDICT r05 item 2).
//
// The dense phase's interaction body is 13 VALU operations per (target, source) pair: 3 sub, 3 fma (r^2), v_rsq_f32, 3 mul
// (m r^-1, r^-2, m r^-3), 3 fma (acc += d * m r^-3). The matrix pipe is idle meanwhile. v_mfma_f32_4x4x1_16B_f32 computes, for 16
// independent blocks of four lanes, D[i][j] += A[i] * B[j]: with the kernel's own lane mapping (lanes 4b .. 4b + 3 = four
// consecutive target slots of ONE source split) and
//     B[j] = r^-3 of lane j's own pair,   A[i] = component i of {m sx, m sy, m sz, m} of the block's source,
// lane (b, j) receives in its four D registers  sum_s r^-3 m {sx, sy, sz, 1}  for ITS target: the three accumulate fmas and the
// mass multiply leave the vector pipe (a_t = D.xyz - t * D.w once per node; coordinates relative to the node's centre). What the
// vector pipe keeps per pair: 3 sub, 3 fma, rsq, 2 mul = 9, plus per SOURCE (shared by the R targets of the lane) one more LDS read
// (ds_read_b32 of component lane & 3 of the record) and 2 operations (A = component * (lane & 3 == 3 ? 1 : m)).
//
// Modes:  0 / 1: R = 2 targets per lane, VALU body / MFMA accumulation;   2 / 3: R = 4 likewise.
// Layout as in list_node(): one wavefront per workgroup, a 128-source LDS tile per wave, TP = 32 target slots x NS = 2 source
// splits (chunked), broadcast ds_read_b128 per source. Prints ns per 64 pairs per SIMD at 1 .. 8 waves per SIMD, the shader clock
// held (s_memtime / s_memrealtime), and the largest relative difference between the two accumulations.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

typedef float v4f __attribute__((ext_vector_type(4)));
constexpr int SRC_CAP = 128;

struct wave_lds {
    float4 src[SRC_CAP];
    unsigned pad[768]; // the stack and queues of the real kernel: 5 KiB per wave in all
};

template <int R, bool MFMA>
__global__ void __launch_bounds__(64, 8) k(float *out, unsigned long long *clk, int tiles, float seed)
{
    __shared__ wave_lds L;
    const int lane = threadIdx.x;
    // Sources relative to the node's centre, a few node radii away (MAC-accepted nodes).
    for (int i = lane; i < SRC_CAP; i += 64) {
        L.src[i] = make_float4(3.f + 0.37f * i * seed, -2.f + 0.11f * i, 1.5f + 0.05f * i, 1.f + 1e-3f * i);
    }
    __syncthreads();
    constexpr int TP = 32, NS = 2;
    const int ts = lane % TP, sp = lane / TP;
    float tx[R], ty[R], tz[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int t = ts + r * TP;
        tx[r] = 0.01f * t * seed, ty[r] = -0.02f * t, tz[r] = 0.015f * t;
    }
    const float eps2 = 1e-6f * seed;
    const int full = SRC_CAP / NS;
    const float4 *p = L.src + sp * full;
    const float *pw = reinterpret_cast<const float *>(p) + (lane & 3);
    const bool is3 = (lane & 3) == 3;
    float acc[R][3];
    v4f D[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        acc[r][0] = acc[r][1] = acc[r][2] = 0.f;
        D[r] = v4f{0.f, 0.f, 0.f, 0.f};
    }
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), w0 = __builtin_amdgcn_s_memrealtime();
    for (int t = 0; t < tiles; ++t) {
#pragma unroll 2
        for (int it = 0; it < full; ++it) {
            const float4 s = p[it];
            if constexpr (MFMA) {
                const float comp = pw[4 * it];
                const float a_op = comp * (is3 ? 1.f : s.w);
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const float dx = s.x - tx[r], dy = s.y - ty[r], dz = s.z - tz[r];
                    const float r2 = __fmaf_rn(dz, dz, __fmaf_rn(dy, dy, __fmaf_rn(dx, dx, eps2)));
                    const float ri = __builtin_amdgcn_rsqf(r2);
                    const float ri3 = (ri * ri) * ri;
                    D[r] = __builtin_amdgcn_mfma_f32_4x4x1f32(a_op, ri3, D[r], 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const float dx = s.x - tx[r], dy = s.y - ty[r], dz = s.z - tz[r];
                    const float r2 = __fmaf_rn(dz, dz, __fmaf_rn(dy, dy, __fmaf_rn(dx, dx, eps2)));
                    const float ri = __builtin_amdgcn_rsqf(r2);
                    const float mr3 = (s.w * ri) * (ri * ri);
                    acc[r][0] = __fmaf_rn(dx, mr3, acc[r][0]);
                    acc[r][1] = __fmaf_rn(dy, mr3, acc[r][1]);
                    acc[r][2] = __fmaf_rn(dz, mr3, acc[r][2]);
                }
            }
        }
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), w1 = __builtin_amdgcn_s_memrealtime();
#pragma unroll
    for (int r = 0; r < R; ++r) {
        float ax, ay, az;
        if constexpr (MFMA) {
            ax = D[r][0] - tx[r] * D[r][3], ay = D[r][1] - ty[r] * D[r][3], az = D[r][2] - tz[r] * D[r][3];
        } else {
            ax = acc[r][0], ay = acc[r][1], az = acc[r][2];
        }
        float *o = out + (static_cast<size_t>(blockIdx.x) * 64 * R + r * 64 + lane) * 3;
        o[0] = ax, o[1] = ay, o[2] = az;
    }
    if (lane == 0) {
        clk[2 * blockIdx.x] = c1 - c0;
        clk[2 * blockIdx.x + 1] = w1 - w0;
    }
}

template <int R, bool MFMA>
int run(const char *name, int waves_per_simd, float *d_out, unsigned long long *d_clk, std::vector<float> *keep)
{
    const int tiles = 400, nblk = 256 * 4 * waves_per_simd;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k<R, MFMA>), dim3(nblk), dim3(64), 0, 0, d_out, d_clk, 10, 1.0f);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<R, MFMA>), dim3(nblk), dim3(64), 0, 0, d_out, d_clk, tiles, 1.0f);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> clk(2 * nblk);
    CHECK(hipMemcpy(clk.data(), d_clk, clk.size() * 8, hipMemcpyDeviceToHost));
    double cyc = 0, wall = 0;
    for (int b = 0; b < nblk; ++b) {
        cyc += clk[2 * b], wall += clk[2 * b + 1];
    }
    // 64 lanes x R pairs per source step; SRC_CAP / NS steps per tile
    const double steps = static_cast<double>(tiles) * (SRC_CAP / 2), pairs64 = steps * R;
    printf("%-34s waves/SIMD=%d  ms=%8.3f  ns per 64 pairs per SIMD=%6.2f  cycles per 64 pairs per wave=%7.2f  clock=%.2f GHz\n", name,
           waves_per_simd, ms, ms * 1e6 / (pairs64 * waves_per_simd), cyc / nblk / pairs64, cyc / (wall * 10.0));
    if (keep) {
        keep->resize(static_cast<size_t>(64) * R * 3);
        CHECK(hipMemcpy(keep->data(), d_out, keep->size() * 4, hipMemcpyDeviceToHost));
    }
    return 0;
}

int main()
{
    float *d_out;
    unsigned long long *d_clk;
    CHECK(hipMalloc(&d_out, static_cast<size_t>(256) * 4 * 8 * 64 * 4 * 3 * 4));
    CHECK(hipMalloc(&d_clk, static_cast<size_t>(256) * 4 * 8 * 16));
    for (int w : {1, 2, 4, 6, 8}) {
        std::vector<float> a, b, c, d;
        run<2, false>("R=2 VALU body (13 ops per pair)", w, d_out, d_clk, &a);
        run<2, true>("R=2 MFMA 4x4x1 accumulation", w, d_out, d_clk, &b);
        run<4, false>("R=4 VALU body", w, d_out, d_clk, &c);
        run<4, true>("R=4 MFMA 4x4x1 accumulation", w, d_out, d_clk, &d);
        if (w == 1) {
            auto cmp = [](const std::vector<float> &u, const std::vector<float> &v) {
                double worst = 0;
                for (size_t i = 0; i + 2 < u.size(); i += 3) {
                    const double n = std::sqrt(double(u[i]) * u[i] + double(u[i + 1]) * u[i + 1] + double(u[i + 2]) * u[i + 2]);
                    const double e = std::sqrt(std::pow(double(u[i]) - v[i], 2) + std::pow(double(u[i + 1]) - v[i + 1], 2)
                                               + std::pow(double(u[i + 2]) - v[i + 2], 2));
                    worst = std::max(worst, e / n);
                }
                return worst;
            };
            printf("max relative difference MFMA vs VALU accumulation: R=2 %.3g, R=4 %.3g\n", cmp(a, b), cmp(c, d));
        }
    }
    return 0;
}
