// Microbenchmark (round 5): what does it cost a stream of small kernels to record an event after every launch?
//   a: kernels only   b: + hipEventRecord of a hipEventDisableTiming event after every kernel   c: + a timing event
//   d: as b, and a second stream waits on the event before its own kernel (the pattern of a caller alternating two streams)
// Prints us per iteration (wall clock over N back-to-back iterations, one synchronisation at the end) for two kernel lengths.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__global__ void spin(float *out, int iters)
{
    float a = threadIdx.x;
    for (int i = 0; i < iters; ++i) a = __fmaf_rn(a, 1.0001f, 0.5f);
    if (a == 12345.f) out[0] = a;
}
int main()
{
    float *d; CHECK(hipMalloc(&d, 4096));
    hipStream_t s1, s2; CHECK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CHECK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    hipEvent_t eu, et; CHECK(hipEventCreateWithFlags(&eu, hipEventDisableTiming)); CHECK(hipEventCreate(&et));
    const int N = 3000;
    for (int iters : {2000, 40000}) {
        for (int mode = 0; mode < 4; ++mode) {
            for (int rep = 0; rep < 2; ++rep) {
                CHECK(hipDeviceSynchronize());
                auto t0 = std::chrono::steady_clock::now();
                for (int i = 0; i < N; ++i) {
                    hipLaunchKernelGGL(spin, dim3(256), dim3(64), 0, s1, d, iters);
                    if (mode == 1 || mode == 3) CHECK(hipEventRecord(eu, s1));
                    if (mode == 2) CHECK(hipEventRecord(et, s1));
                    if (mode == 3) { CHECK(hipStreamWaitEvent(s2, eu, 0)); hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s2, d, 10); }
                }
                CHECK(hipDeviceSynchronize());
                double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
                if (rep == 1) printf("kernel of %5d fma steps, mode %c: %7.2f us per iteration\n", iters, "abcd"[mode], us);
            }
        }
    }
    return 0;
}
