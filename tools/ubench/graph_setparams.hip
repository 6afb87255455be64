// What does re-targeting a small executable graph cost per launch on this runtime? A manually built graph of four independent kernel
// nodes (the four class kernels of a call), launched into a stream behind a short "pre-pass" kernel:
//   direct  : pre-pass + four launches forked onto four streams with events (what a first call does today)
//   replay  : pre-pass + hipGraphLaunch of the unchanged executable
//   retarget: pre-pass + hipGraphExecKernelNodeSetParams on all four nodes (new grid size and arguments) + hipGraphLaunch
// Prints host microseconds per call (enqueue only) and wall microseconds per call with the device kept busy by kernels of ~50 us.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

struct params {
    float *out;
    int iters;
    float seed;
    int pad[13];
};

__global__ void k_work(const params P, const unsigned *list, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (static_cast<int>(blockIdx.x) >= n) {
        return;
    }
    float a = P.seed + i;
    for (int k = 0; k < P.iters; ++k) {
        a = __builtin_fmaf(a, 1.0001f, 0.5f);
    }
    P.out[i] = a + (list ? list[0] : 0u);
}

int main()
{
    float *d_out;
    unsigned *d_list;
    CHECK(hipMalloc(&d_out, 4096 * 256 * 4));
    CHECK(hipMalloc(&d_list, 1024));
    CHECK(hipMemset(d_list, 0, 1024));
    hipStream_t st, aux[3];
    CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    for (auto &a : aux) CHECK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
    hipEvent_t fork, join[3];
    CHECK(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
    for (auto &j : join) CHECK(hipEventCreateWithFlags(&j, hipEventDisableTiming));
    params P{d_out, 20000, 1.f, {}};
    int n = 1024;
    const unsigned *lp = d_list;
    // the graph
    hipGraph_t g;
    CHECK(hipGraphCreate(&g, 0));
    hipGraphNode_t node[4];
    hipKernelNodeParams kp[4];
    void *args[4][3];
    params Pn[4];
    int nn[4];
    const unsigned *ln[4];
    for (int c = 0; c < 4; ++c) {
        Pn[c] = P, nn[c] = n, ln[c] = lp;
        args[c][0] = &Pn[c], args[c][1] = &ln[c], args[c][2] = &nn[c];
        kp[c] = hipKernelNodeParams{};
        kp[c].func = reinterpret_cast<void *>(k_work);
        kp[c].gridDim = dim3(1024), kp[c].blockDim = dim3(256), kp[c].sharedMemBytes = 0, kp[c].kernelParams = args[c], kp[c].extra = nullptr;
        CHECK(hipGraphAddKernelNode(&node[c], g, nullptr, 0, &kp[c]));
    }
    hipGraphExec_t ex;
    CHECK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
    params Ppre{d_out, 2000, 1.f, {}};
    auto prepass = [&] { hipLaunchKernelGGL(k_work, dim3(64), dim3(256), 0, st, Ppre, lp, 64); };
    auto direct = [&] {
        prepass();
        (void)hipEventRecord(fork, st);
        for (auto &a : aux) (void)hipStreamWaitEvent(a, fork, 0);
        hipLaunchKernelGGL(k_work, dim3(1024), dim3(256), 0, aux[1], P, lp, n);
        hipLaunchKernelGGL(k_work, dim3(1024), dim3(256), 0, aux[2], P, lp, n);
        hipLaunchKernelGGL(k_work, dim3(1024), dim3(256), 0, st, P, lp, n);
        hipLaunchKernelGGL(k_work, dim3(1024), dim3(256), 0, aux[0], P, lp, n);
        for (int i = 0; i < 3; ++i) {
            (void)hipEventRecord(join[i], aux[i]);
            (void)hipStreamWaitEvent(st, join[i], 0);
        }
    };
    auto replay = [&] {
        prepass();
        (void)hipGraphLaunch(ex, st);
    };
    int flip = 0;
    auto retarget = [&] {
        prepass();
        flip ^= 1;
        for (int c = 0; c < 4; ++c) {
            nn[c] = 1024 - flip * (c + 1);
            Pn[c].seed = 1.f + flip;
            kp[c].gridDim = dim3(static_cast<unsigned>(nn[c]));
            (void)hipGraphExecKernelNodeSetParams(ex, node[c], &kp[c]);
        }
        (void)hipGraphLaunch(ex, st);
    };
    auto bench = [&](const char *name, auto fn) {
        for (int i = 0; i < 20; ++i) fn();
        (void)hipStreamSynchronize(st);
        const int reps = 400;
        double host = 0;
        const auto w0 = std::chrono::steady_clock::now();
        for (int i = 0; i < reps; ++i) {
            const auto t0 = std::chrono::steady_clock::now();
            fn();
            host += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            if (i % 8 == 7) (void)hipStreamSynchronize(st); // (a time-stepping loop synchronises often)
        }
        (void)hipStreamSynchronize(st);
        const double wall = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - w0).count();
        // one synchronised call: latency from the call to its completion
        double lat = 0;
        for (int i = 0; i < 50; ++i) {
            const auto t0 = std::chrono::steady_clock::now();
            fn();
            (void)hipStreamSynchronize(st);
            lat += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        }
        printf("%-10s host us per call %7.2f   wall us per call (queued) %8.2f   synchronised call %8.2f us\n", name, host / reps, wall / reps, lat / 50);
    };
    for (int rep = 0; rep < 2; ++rep) {
        bench("direct", direct);
        bench("replay", replay);
        bench("retarget", retarget);
    }
    // Is a launch that is still queued affected by re-targeting its executable? 16 re-targeted launches back to back, no
    // synchronisation in between, every one with its own output slot and seed; each kernel runs ~0.6 ms, so launches 2 .. 16 are
    // re-targeted while launch 1 is still running and the ones in between are queued.
    {
        const int calls = 16, blocks = 1024;
        float *d_slots;
        CHECK(hipMalloc(&d_slots, static_cast<size_t>(calls) * 4 * blocks * 256 * 4));
        CHECK(hipMemset(d_slots, 0, static_cast<size_t>(calls) * 4 * blocks * 256 * 4));
        CHECK(hipStreamSynchronize(st));
        for (int call = 0; call < calls; ++call) {
            for (int c = 0; c < 4; ++c) {
                Pn[c].out = d_slots + (static_cast<size_t>(call) * 4 + c) * blocks * 256;
                Pn[c].iters = 0; // out[i] = seed + i
                Pn[c].seed = 1000.f * (call + 1) + 100.f * c;
                nn[c] = blocks;
                kp[c].gridDim = dim3(blocks);
                CHECK(hipGraphExecKernelNodeSetParams(ex, node[c], &kp[c]));
            }
            // a long kernel in front of every launch keeps the queue full
            hipLaunchKernelGGL(k_work, dim3(1024), dim3(256), 0, st, P, lp, n);
            CHECK(hipGraphLaunch(ex, st));
        }
        CHECK(hipStreamSynchronize(st));
        std::vector<float> h(static_cast<size_t>(calls) * 4 * blocks * 256);
        CHECK(hipMemcpy(h.data(), d_slots, h.size() * 4, hipMemcpyDeviceToHost));
        long bad = 0;
        for (int call = 0; call < calls; ++call)
            for (int c = 0; c < 4; ++c)
                for (int i = 0; i < blocks * 256; i += 97) {
                    const float want = 1000.f * (call + 1) + 100.f * c + i;
                    bad += h[(static_cast<size_t>(call) * 4 + c) * blocks * 256 + i] != want;
                }
        printf("re-targeting with earlier launches still queued: %ld wrong values (0 = launches keep the arguments they were launched with)\n", bad);
    }
    hipError_t e = hipGetLastError();
    printf("last error: %s\n", hipGetErrorString(e));
    return 0;
}
