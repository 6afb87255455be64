// Does s_setprio change how a SIMD shares its VALU issue slots between resident waves? 6 single-wave blocks per SIMD
// run the same dependent-FMA loop; one block in six raises its priority to 3. Prints the mean duration of the two kinds.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void __launch_bounds__(64) k(float *out, unsigned long long *t, int iters, int every)
{
    const bool hi = every > 0 && (blockIdx.x % every) == 0;
    if (hi) {
        __builtin_amdgcn_s_setprio(3);
    }
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            a0 = __fmaf_rn(a0, 1.0001f, 0.5f), a1 = __fmaf_rn(a1, 1.0001f, 0.5f);
            a2 = __fmaf_rn(a2, 1.0001f, 0.5f), a3 = __fmaf_rn(a3, 1.0001f, 0.5f);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3;
    if (threadIdx.x == 0) {
        t[2 * blockIdx.x] = t0, t[2 * blockIdx.x + 1] = t1;
    }
}

int main()
{
    const int blocks = 256 * 4 * 6;
    float *d_out;
    unsigned long long *d_t;
    CHECK(hipMalloc(&d_out, sizeof(float) * 64 * blocks));
    CHECK(hipMalloc(&d_t, 16 * blocks));
    std::vector<unsigned long long> h(2 * blocks);
    for (int every : {0, 6, 2}) {
        hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, d_out, d_t, 2000, every);
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(h.data(), d_t, 16 * blocks, hipMemcpyDeviceToHost));
        double s[2] = {0, 0}, e[2] = {0, 0};
        int n[2] = {0, 0};
        unsigned long long tmin = ~0ull;
        for (int b = 0; b < blocks; ++b) tmin = h[2 * b] < tmin ? h[2 * b] : tmin;
        for (int b = 0; b < blocks; ++b) {
            const int kind = every > 0 && b % every == 0;
            s[kind] += double(h[2 * b + 1] - h[2 * b]) * 0.01, e[kind] += double(h[2 * b + 1] - tmin) * 0.01, ++n[kind];
        }
        printf("one block in %d at priority 3: normal waves mean duration %.1f us (end %.1f), raised waves %.1f us (end %.1f)\n",
               every, s[0] / (n[0] ? n[0] : 1), e[0] / (n[0] ? n[0] : 1), s[1] / (n[1] ? n[1] : 1), e[1] / (n[1] ? n[1] : 1));
    }
    return 0;
}
