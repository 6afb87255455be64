// Microbenchmark (round 5): when do a kernel's stores to pinned host memory become visible to the host?
// A kernel of 2048 blocks writes 16 KiB per block to a hipHostMalloc'd buffer (every block: write its chunk, then spin ~1 us per
// block index so that the whole launch takes a couple of ms and chunks are written progressively); the host polls one word per chunk
// and records, for allocation flags default / coherent / non-coherent / write-combined, how many chunks it has seen at 25 / 50 / 75 %
// of the kernel's duration and whether any chunk changed AFTER the completion event (a late or repeated write).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__global__ void k(unsigned *out, unsigned tag, int spin)
{
    unsigned *chunk = out + (size_t)blockIdx.x * 4096;
    // a little work first, growing with the block index, so that blocks finish progressively
    float a = threadIdx.x;
    for (int i = 0; i < spin * (int)(blockIdx.x / 64 + 1); ++i) a = __fmaf_rn(a, 1.0001f, 0.5f);
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) chunk[i] = tag + (a == 12345.f);
}
int main()
{
    const int nblk = 2048; const size_t words = (size_t)nblk * 4096;
    const unsigned flags[4] = {hipHostMallocDefault, hipHostMallocCoherent, hipHostMallocNonCoherent, hipHostMallocWriteCombined};
    const char *names[4] = {"default", "coherent", "non-coherent", "write-combined"};
    for (int f = 0; f < 4; ++f) {
        unsigned *h = nullptr;
        if (hipHostMalloc((void **)&h, words * 4, flags[f]) != hipSuccess) { printf("%-15s allocation failed\n", names[f]); (void)hipGetLastError(); continue; }
        unsigned *d = nullptr; CHECK(hipHostGetDevicePointer((void **)&d, h, 0));
        hipEvent_t ev; CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        for (int rep = 0; rep < 3; ++rep) {
            const unsigned tag = 0x1000u * (rep + 1);
            std::memset(h, 0, words * 4);
            CHECK(hipDeviceSynchronize());
            auto t0 = std::chrono::steady_clock::now();
            hipLaunchKernelGGL(k, dim3(nblk), dim3(256), 0, 0, d, tag, 400);
            CHECK(hipEventRecord(ev, 0));
            std::vector<double> seen_at(nblk, -1.0);
            int seen = 0; double t_end = 0;
            for (;;) {
                const bool done = hipEventQuery(ev) == hipSuccess;
                const double t = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
                for (int b = 0; b < nblk; ++b) if (seen_at[b] < 0 && ((volatile unsigned *)h)[(size_t)b * 4096 + 4095] == tag) { seen_at[b] = t; ++seen; }
                if (done) { t_end = t; break; }
            }
            int q1 = 0, q2 = 0, q3 = 0, at_end = 0;
            for (int b = 0; b < nblk; ++b) { if (seen_at[b] >= 0) { ++at_end; if (seen_at[b] < 0.25 * t_end) ++q1; if (seen_at[b] < 0.5 * t_end) ++q2; if (seen_at[b] < 0.75 * t_end) ++q3; } }
            // wipe and watch for late writes
            std::memset(h, 0, words * 4);
            auto t1 = std::chrono::steady_clock::now(); int late = 0;
            while (std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count() < 3.0)
                for (int b = 0; b < nblk; ++b) if (((volatile unsigned *)h)[(size_t)b * 4096 + 4095] != 0) { ++late; ((volatile unsigned *)h)[(size_t)b * 4096 + 4095] = 0; }
            if (rep) printf("%-15s kernel %.0f us: chunks visible at 25/50/75/100 %% of it: %d / %d / %d / %d of %d; words rewritten after the event: %d\n", names[f], t_end, q1, q2, q3, at_end, nblk, late);
        }
        CHECK(hipHostFree(h));
    }
    return 0;
}
