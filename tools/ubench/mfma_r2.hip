// Microbenchmark for a possible next step of the dense phase: let the matrix cores produce r^2 for node (monopole)
// interactions, r^2 = |s|^2 + |t|^2 - 2 s.t with coordinates relative to the group centre, as a 16x16x4 fp32 MFMA
// (A = {-2tx, -2ty, -2tz, 1} per target row, B = {sx, sy, sz, |s|^2} per source column, C = |t|^2), and keep on the
// vector ALU only w = rsq(r^2)^3 and the accumulation A_i += w * {m sx, m sy, m sz, m} (7 instructions per pair
// instead of 13; a_i = A_i.xyz - t_i * A_i.w at the end).
//   mode 0: today's body (3 sub, 3 fma, rsq, 3 mul, 3 fma), 1 target per lane, sources broadcast from LDS
//   mode 1..4: MFMA variant with RT = mode row tiles (16 * RT targets per wave)
// Prints pairs/s per chip extrapolated from one launch that fills the device.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

typedef float v4f __attribute__((ext_vector_type(4)));
#ifndef NSRC_N
#define NSRC_N 512
#endif
constexpr int NSRC = NSRC_N; // sources per tile set in LDS (per wave)

__global__ void __launch_bounds__(64) k_valu(float *out, int iters)
{
    __shared__ float4 src[NSRC];
    for (int i = threadIdx.x; i < NSRC; i += 64) src[i] = make_float4(i * 0.37f, i * 0.11f + 3.f, i * 0.05f - 1.f, 1.f + i * 1e-3f);
    __syncthreads();
    const float tx = threadIdx.x * 0.01f, ty = threadIdx.x * 0.02f, tz = threadIdx.x * 0.03f;
    float ax = 0, ay = 0, az = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll 4
        for (int j = 0; j < NSRC; ++j) {
            const float4 s = src[j];
            const float dx = s.x - tx, dy = s.y - ty, dz = s.z - tz;
            const float r2 = __fmaf_rn(dz, dz, __fmaf_rn(dy, dy, __fmaf_rn(dx, dx, 1e-3f)));
            const float ri = __builtin_amdgcn_rsqf(r2);
            const float mr3 = (s.w * ri) * (ri * ri);
            ax = __fmaf_rn(dx, mr3, ax), ay = __fmaf_rn(dy, mr3, ay), az = __fmaf_rn(dz, mr3, az);
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = ax + ay + az;
}

template <int RT>
__global__ void __launch_bounds__(64) k_mfma(float *out, int iters)
{
    __shared__ float4 src[NSRC];  // {sx, sy, sz, |s|^2}
    __shared__ float4 srcm[NSRC]; // {m sx, m sy, m sz, m}
    for (int i = threadIdx.x; i < NSRC; i += 64) {
        const float x = i * 0.37f, y = i * 0.11f + 3.f, z = i * 0.05f - 1.f, m = 1.f + i * 1e-3f;
        src[i] = make_float4(x, y, z, x * x + y * y + z * z + 1e-3f);
        srcm[i] = make_float4(m * x, m * y, m * z, m);
    }
    __syncthreads();
    const int lane = threadIdx.x, row = lane & 15, kk = lane >> 4;
    float a_op[RT];
    v4f t2[RT], acc[RT][4];
#pragma unroll
    for (int r = 0; r < RT; ++r) {
        const float tx = (row + 16 * r) * 0.01f, ty = (row + 16 * r) * 0.02f, tz = (row + 16 * r) * 0.03f;
        a_op[r] = kk == 0 ? -2.f * tx : (kk == 1 ? -2.f * ty : (kk == 2 ? -2.f * tz : 1.f));
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int tr = 4 * kk + i + 16 * r;
            t2[r][i] = (tr * 0.01f) * (tr * 0.01f) + (tr * 0.02f) * (tr * 0.02f) + (tr * 0.03f) * (tr * 0.03f);
            acc[r][i] = v4f{0, 0, 0, 0};
        }
    }
    const float *srcf = reinterpret_cast<const float *>(src);
    for (int it = 0; it < iters; ++it) {
#pragma unroll 2
        for (int j0 = 0; j0 < NSRC; j0 += 16) {
            const float b_op = srcf[(j0 + row) * 4 + kk];   // B[k][col]: component kk of source col = row
            const float4 sm = srcm[j0 + row];               // this lane's column source, mass-weighted
#pragma unroll
            for (int r = 0; r < RT; ++r) {
                const v4f r2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a_op[r], b_op, t2[r], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float ri = __builtin_amdgcn_rsqf(r2[i]);
                    const float w = (ri * ri) * ri;
                    acc[r][i][0] = __fmaf_rn(w, sm.x, acc[r][i][0]);
                    acc[r][i][1] = __fmaf_rn(w, sm.y, acc[r][i][1]);
                    acc[r][i][2] = __fmaf_rn(w, sm.z, acc[r][i][2]);
                    acc[r][i][3] = __fmaf_rn(w, sm.w, acc[r][i][3]);
                }
            }
        }
    }
    float s = 0;
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int i = 0; i < 4; ++i) s += acc[r][i][0] + acc[r][i][1] + acc[r][i][2] + acc[r][i][3];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}

template <typename K>
static int run(const char *name, K kern, double pairs_per_wave_iter, float *d_out, int blocks, int iters)
{
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), 0, 0, d_out, 2);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), 0, 0, d_out, iters);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double pairs = pairs_per_wave_iter * iters * blocks;
    printf("%-28s %8.3f ms  %.3e pairs/s  (%.2f ns per 64 pairs per SIMD)\n", name, ms, pairs / (ms * 1e-3),
           ms * 1e6 / (pairs / 64.0) * 1024.0);
    return 0;
}

int main()
{
    float *d_out;
    const int blocks = 256 * 4 * 7 * 4; // 4 rounds of 7 waves per SIMD
    CHECK(hipMalloc(&d_out, sizeof(float) * 64 * blocks));
    const int iters = 40;
    run("valu body (13 instr/pair)", k_valu, 64.0 * NSRC, d_out, blocks, iters);
    run("mfma r2, 16 targets/wave", k_mfma<1>, 16.0 * NSRC, d_out, blocks, iters);
    run("mfma r2, 32 targets/wave", k_mfma<2>, 32.0 * NSRC, d_out, blocks, iters);
    run("mfma r2, 48 targets/wave", k_mfma<3>, 48.0 * NSRC, d_out, blocks, iters);
    run("mfma r2, 64 targets/wave", k_mfma<4>, 64.0 * NSRC, d_out, blocks, iters);
    return 0;
}
