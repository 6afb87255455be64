// Microbenchmark: fp64 VALU issue rates on gfx950 behind the fp64 interaction body (v_fma_f64, v_rsq_f64, conversions),
// and two ways to get 1/sqrt(x) in fp64: v_rsq_f64 + one Newton step (the kernel's) vs v_rsq_f32 seed + two Newton steps.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__device__ __forceinline__ double rsq_a(double x)
{
    double y = __builtin_amdgcn_rsq(x);
    double e = __builtin_fma(-x * y, y, 1.0);
    return __builtin_fma(y * 0.5, e, y);
}
__device__ __forceinline__ double rsq_b(double x)
{
    double y = static_cast<double>(__builtin_amdgcn_rsqf(static_cast<float>(x)));
    double e = __builtin_fma(-x * y, y, 1.0);
    y = __builtin_fma(y * 0.5, e, y);
    e = __builtin_fma(-x * y, y, 1.0);
    return __builtin_fma(y * 0.5, e, y);
}

template <int MODE>
__global__ void __launch_bounds__(256) k(double *out, int iters, double seed)
{
    double a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const double c = seed * 0.5, d = seed * 0.25;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (MODE == 0) {
                asm volatile("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n"
                             "v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));
            } else if (MODE == 1) {
                asm volatile("v_rsq_f64 %0, %0\n v_rsq_f64 %1, %1\n v_rsq_f64 %2, %2\n v_rsq_f64 %3, %3\n"
                             "v_rsq_f64 %4, %4\n v_rsq_f64 %5, %5\n v_rsq_f64 %6, %6\n v_rsq_f64 %7, %7\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            } else if (MODE == 2) {
                asm volatile("v_add_f64 %0, %0, %8\n v_add_f64 %1, %1, %8\n v_add_f64 %2, %2, %8\n v_add_f64 %3, %3, %8\n"
                             "v_mul_f64 %4, %4, %9\n v_mul_f64 %5, %5, %9\n v_mul_f64 %6, %6, %9\n v_mul_f64 %7, %7, %9\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));
            } else if (MODE == 3) { // 8 x (rsq_f64 + 1 Newton step)
                a0 = rsq_a(a0 + c); a1 = rsq_a(a1 + c); a2 = rsq_a(a2 + c); a3 = rsq_a(a3 + c);
                a4 = rsq_a(a4 + c); a5 = rsq_a(a5 + c); a6 = rsq_a(a6 + c); a7 = rsq_a(a7 + c);
            } else if (MODE == 4) { // 8 x (cvt + rsq_f32 + cvt + 2 Newton steps)
                a0 = rsq_b(a0 + c); a1 = rsq_b(a1 + c); a2 = rsq_b(a2 + c); a3 = rsq_b(a3 + c);
                a4 = rsq_b(a4 + c); a5 = rsq_b(a5 + c); a6 = rsq_b(a6 + c); a7 = rsq_b(a7 + c);
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <int MODE>
int run(const char *name, int n_per_iter, double *d_out, int nblk_per_cu)
{
    const int iters = 2000, threads = 256, nblk = 256 * nblk_per_cu;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<MODE>, dim3(nblk), dim3(threads), 0, 0, d_out, 10, 1.0);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<MODE>, dim3(nblk), dim3(threads), 0, 0, d_out, iters, 1.0);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double waves_per_simd = nblk_per_cu, per_simd = double(iters) * n_per_iter * waves_per_simd;
    printf("%-44s waves/SIMD=%2d  ms=%8.3f  cycles@2.4GHz per wave-op per SIMD = %6.2f\n", name, nblk_per_cu, ms, ms * 1e6 / per_simd * 2.4);
    return 0;
}

int main()
{
    double *d_out;
    CHECK(hipMalloc(&d_out, size_t(256) * 8 * 256 * 8));
    for (int w : {2, 4, 8}) {
        run<0>("v_fma_f64", 32, d_out, w);
        run<1>("v_rsq_f64", 32, d_out, w);
        run<2>("v_add_f64 / v_mul_f64", 32, d_out, w);
        run<3>("rsqrt: v_rsq_f64 + 1 Newton (per result)", 32, d_out, w);
        run<4>("rsqrt: v_rsq_f32 seed + 2 Newton (per result)", 32, d_out, w);
    }
    return 0;
}
