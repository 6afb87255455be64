// Microbenchmark 2: is the dense phase VALU-bound or LDS-bound? Interaction body with the source coming
// from (a) a broadcast ds_read_b128, (b) registers only, for 1/2/4 targets per lane; plus raw LDS read
// rates for broadcast and per-lane addresses.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int R, bool LDS>
__global__ void __launch_bounds__(256) k_inter(float *out, int iters, float seed)
{
    __shared__ float4 lds[1024];
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) lds[i] = make_float4(i, i + 1, i + 2, 1.f);
    __syncthreads();
    float tx[R], ty[R], tz[R], ax[R], ay[R], az[R];
    for (int r = 0; r < R; ++r) { tx[r] = seed + threadIdx.x + r; ty[r] = tx[r] * 0.5f; tz[r] = tx[r] * 0.25f; ax[r] = ay[r] = az[r] = 0.f; }
    const float eps2 = seed * 0.001f;
    float4 sreg = make_float4(seed, seed * 2, seed * 3, 1.f);
    for (int i = 0; i < iters; ++i) {
#pragma unroll 4
        for (int u = 0; u < 32; ++u) {
            float4 s;
            if (LDS) s = lds[(i * 32 + u) & 1023];
            else { s = sreg; sreg.x += 1.f; }
#pragma unroll
            for (int r = 0; r < R; ++r) {
                float dx = s.x - tx[r], dy = s.y - ty[r], dz = s.z - tz[r];
                float r2 = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, __builtin_fmaf(dx, dx, eps2)));
                float ri = __builtin_amdgcn_rsqf(r2);
                float mr = s.w * ri; float ri2 = ri * ri; float mr3 = mr * ri2;
                ax[r] = __builtin_fmaf(dx, mr3, ax[r]); ay[r] = __builtin_fmaf(dy, mr3, ay[r]); az[r] = __builtin_fmaf(dz, mr3, az[r]);
            }
        }
    }
    float acc = 0; for (int r = 0; r < R; ++r) acc += ax[r] + ay[r] + az[r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc + sreg.x;
}

template <int MODE>
__global__ void __launch_bounds__(256) k_lds(float *out, int iters)
{
    __shared__ float4 lds[2048];
    for (int i = threadIdx.x; i < 2048; i += blockDim.x) lds[i] = make_float4(i, i + 1, i + 2, 1.f);
    __syncthreads();
    float4 a = make_float4(0, 0, 0, 0);
    const int lane = threadIdx.x & 63;
    for (int i = 0; i < iters; ++i) {
#pragma unroll 8
        for (int u = 0; u < 32; ++u) {
            int idx;
            if (MODE == 0) idx = (i * 32 + u) & 1023;            // broadcast: all lanes same address
            else if (MODE == 1) idx = ((i * 32 + u) * 64 + lane) & 2047; // per-lane consecutive
            else idx = ((i * 32 + u) + (lane >> 4) * 8) & 1023;   // 4 distinct addresses (splits)
            float4 s = lds[idx];
            a.x += s.x; a.y += s.y; a.z += s.z; a.w += s.w;
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a.x + a.y + a.z + a.w;
}

template <typename K>
int timeit(const char *name, K kern, int blocks_per_cu, double work_per_wave, float *d_out, int iters)
{
    int nblk = 256 * blocks_per_cu;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    kern(nblk, 10);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    kern(nblk, iters);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    double waves_per_simd = blocks_per_cu; // 256-thread blocks: one wave per SIMD each
    double per_simd = work_per_wave * iters * waves_per_simd;
    printf("%-34s waves/SIMD=%d ms=%8.3f  ns per unit per SIMD = %7.3f (cyc@2.4GHz %6.2f)\n", name, blocks_per_cu, ms, ms * 1e6 / per_simd, ms * 1e6 / per_simd * 2.4);
    return 0;
}

int main()
{
    float *d_out; CHECK(hipMalloc(&d_out, 256 * 8 * 256 * 4 + 4096));
    const int iters = 3000;
    for (int w : {2, 4, 8}) {
        // unit = one source processed by one wave (R interactions per lane)
        timeit("inter R=1 LDS src (per source)", [&](int nb, int it) { hipLaunchKernelGGL((k_inter<1, true>), dim3(nb), dim3(256), 0, 0, d_out, it, 1.f); }, w, 32, d_out, iters);
        timeit("inter R=1 reg src (per source)", [&](int nb, int it) { hipLaunchKernelGGL((k_inter<1, false>), dim3(nb), dim3(256), 0, 0, d_out, it, 1.f); }, w, 32, d_out, iters);
        timeit("inter R=2 LDS src (per source)", [&](int nb, int it) { hipLaunchKernelGGL((k_inter<2, true>), dim3(nb), dim3(256), 0, 0, d_out, it, 1.f); }, w, 32, d_out, iters);
        timeit("inter R=2 reg src (per source)", [&](int nb, int it) { hipLaunchKernelGGL((k_inter<2, false>), dim3(nb), dim3(256), 0, 0, d_out, it, 1.f); }, w, 32, d_out, iters);
        timeit("inter R=4 LDS src (per source)", [&](int nb, int it) { hipLaunchKernelGGL((k_inter<4, true>), dim3(nb), dim3(256), 0, 0, d_out, it, 1.f); }, w, 32, d_out, iters);
        timeit("inter R=4 reg src (per source)", [&](int nb, int it) { hipLaunchKernelGGL((k_inter<4, false>), dim3(nb), dim3(256), 0, 0, d_out, it, 1.f); }, w, 32, d_out, iters);
        timeit("ds_read_b128 broadcast (per read)", [&](int nb, int it) { hipLaunchKernelGGL((k_lds<0>), dim3(nb), dim3(256), 0, 0, d_out, it); }, w, 32, d_out, iters);
        timeit("ds_read_b128 per-lane (per read)", [&](int nb, int it) { hipLaunchKernelGGL((k_lds<1>), dim3(nb), dim3(256), 0, 0, d_out, it); }, w, 32, d_out, iters);
        timeit("ds_read_b128 4 addrs (per read)", [&](int nb, int it) { hipLaunchKernelGGL((k_lds<2>), dim3(nb), dim3(256), 0, 0, d_out, it); }, w, 32, d_out, iters);
    }
    return 0;
}
