// Microbenchmark: how the one-target-per-lane interaction body gets its sources.
//   mode 0: one broadcast ds_read_b128 per source (what lk_eval_tile does today), 13 VALU per source;
//   mode 1: one ds_read_b128 per 64 sources (lane = source), the sources handed out with 4 v_readlane_b32 each and used as
//           SGPR operands, 17 VALU per source, same order of the sources;
//   mode 2: two targets per lane, one broadcast ds_read_b128 per source (26 VALU per source).
// Prints ns per source per SIMD at 1..8 waves per SIMD (lower is better; for mode 2 per source for BOTH targets).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__device__ __forceinline__ void body(float sx, float sy, float sz, float sm, float tx, float ty, float tz, float eps2, float &ax, float &ay, float &az)
{
    const float dx = sx - tx, dy = sy - ty, dz = sz - tz;
    const float r2 = __fmaf_rn(dz, dz, __fmaf_rn(dy, dy, __fmaf_rn(dx, dx, eps2)));
    const float ri = __builtin_amdgcn_rsqf(r2);
    const float mr = sm * ri, ri2 = ri * ri, mr3 = mr * ri2;
    ax = __fmaf_rn(dx, mr3, ax), ay = __fmaf_rn(dy, mr3, ay), az = __fmaf_rn(dz, mr3, az);
}

template <int MODE>
__global__ void __launch_bounds__(256) k(float *out, int tiles, float seed)
{
    __shared__ float4 lds[1024];
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) lds[i] = make_float4(i * 0.37f, i + 1.5f, i * 0.11f + 2, 1.f);
    __syncthreads();
    const float tx = seed + threadIdx.x, ty = tx * 0.5f, tz = tx * 0.25f, ux = tx + 0.3f, uy = ty + 0.1f, uz = tz + 0.7f, eps2 = seed * 1e-3f;
    float ax = 0, ay = 0, az = 0, bx = 0, by = 0, bz = 0;
    const int lane = threadIdx.x & 63;
    long long t0 = clock64();
    for (int t = 0; t < tiles; ++t) {
        const float4 *tile = lds + ((t * 64) & 1023);
        if (MODE == 0) {
#pragma unroll 4
            for (int u = 0; u < 64; ++u) {
                const float4 s = tile[u];
                body(s.x, s.y, s.z, s.w, tx, ty, tz, eps2, ax, ay, az);
            }
        } else if (MODE == 1) {
            const float4 mine = tile[lane];
#pragma unroll
            for (int u = 0; u < 64; ++u) {
                const float sx = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mine.x), u));
                const float sy = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mine.y), u));
                const float sz = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mine.z), u));
                const float sm = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mine.w), u));
                body(sx, sy, sz, sm, tx, ty, tz, eps2, ax, ay, az);
            }
        } else {
#pragma unroll 4
            for (int u = 0; u < 64; ++u) {
                const float4 s = tile[u];
                body(s.x, s.y, s.z, s.w, tx, ty, tz, eps2, ax, ay, az);
                body(s.x, s.y, s.z, s.w, ux, uy, uz, eps2, bx, by, bz);
            }
        }
    }
    long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = ax + ay + az + bx + by + bz;
    if (threadIdx.x == 0) {
        ((long long *)(out + gridDim.x * blockDim.x))[blockIdx.x] = t1 - t0;
    }
}

template <int MODE>
int run(const char *name, float *d_out, int nblk_per_cu)
{
    const int tiles = 2000, threads = 256, nblk = 256 * nblk_per_cu;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<MODE>, dim3(nblk), dim3(threads), 0, 0, d_out, 10, 1.0f);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<MODE>, dim3(nblk), dim3(threads), 0, 0, d_out, tiles, 1.0f);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double waves_per_simd = nblk_per_cu, sources = (double)tiles * 64;
    printf("%-46s waves/SIMD=%2.0f  ms=%8.3f  ns per source per wave=%7.2f  ns per source per SIMD=%6.2f\n", name, waves_per_simd, ms,
           ms * 1e6 / sources, ms * 1e6 / (sources * waves_per_simd));
    return 0;
}

int main()
{
    float *d_out;
    CHECK(hipMalloc(&d_out, (size_t)256 * 8 * 256 * 4 + 256 * 8 * 16 + 4096));
    for (int w : {1, 2, 3, 5, 8}) {
        run<0>("1 target, broadcast ds_read_b128 per source", d_out, w);
        run<1>("1 target, 64-source register tile + v_readlane", d_out, w);
        run<2>("2 targets, broadcast ds_read_b128 per source", d_out, w);
    }
    return 0;
}
