// Microbenchmark (round 5): the dense phase of the list kernel with scalar v_*_f32 against packed v_pk_*_f32 bodies, in the
// kernel's own shape: single-wave workgroups, a private 2 KiB LDS tile of 128 sources read with ds_read_b128 (two source splits
// per wave), R targets per lane, accumulators in registers across tiles. Prints ns per 64-lane interaction per SIMD for
// 1..7 wavefronts per SIMD (lower is better) and the sustained shader clock.
//   mode 0: R = 2 scalar (13 VALU per interaction)        mode 1: R = 2 packed over the two targets (12 pk + 2 rsq per two)
//   mode 2: R = 4 scalar                                   mode 3: R = 4 packed (two pairs)
//   mode 4: R = 1 scalar, 4 sources unrolled               mode 5: R = 1 packed over two consecutive SOURCES (tile stored as
//                                                                  pairs {x0,x1,y0,y1},{z0,z1,m0,m1})
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void body1(const float4 s, float tx, float ty, float tz, float eps2, float &ax, float &ay, float &az)
{
    const float dx = s.x - tx, dy = s.y - ty, dz = s.z - tz;
    const float r2 = __fmaf_rn(dz, dz, __fmaf_rn(dy, dy, __fmaf_rn(dx, dx, eps2)));
    const float ri = __builtin_amdgcn_rsqf(r2);
    const float mr = s.w * ri, mr3 = mr * (ri * ri);
    ax = __fmaf_rn(dx, mr3, ax), ay = __fmaf_rn(dy, mr3, ay), az = __fmaf_rn(dz, mr3, az);
}
__device__ __forceinline__ void body2(const float4 s, f2 tx, f2 ty, f2 tz, f2 eps2, f2 &ax, f2 &ay, f2 &az)
{
    const f2 sx = {s.x, s.x}, sy = {s.y, s.y}, sz = {s.z, s.z}, sm = {s.w, s.w};
    const f2 dx = sx - tx, dy = sy - ty, dz = sz - tz;
    const f2 r2 = __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dy, dy, __builtin_elementwise_fma(dx, dx, eps2)));
    const f2 ri = {__builtin_amdgcn_rsqf(r2.x), __builtin_amdgcn_rsqf(r2.y)};
    const f2 mr = sm * ri, mr3 = mr * (ri * ri);
    ax = __builtin_elementwise_fma(dx, mr3, ax), ay = __builtin_elementwise_fma(dy, mr3, ay), az = __builtin_elementwise_fma(dz, mr3, az);
}
// Two sources {x0,x1,y0,y1},{z0,z1,m0,m1} on one target.
__device__ __forceinline__ void body_s2(const float4 a, const float4 b, float tx, float ty, float tz, f2 eps2, f2 &ax, f2 &ay, f2 &az)
{
    const f2 sx = {a.x, a.y}, sy = {a.z, a.w}, sz = {b.x, b.y}, sm = {b.z, b.w};
    const f2 txx = {tx, tx}, tyy = {ty, ty}, tzz = {tz, tz};
    const f2 dx = sx - txx, dy = sy - tyy, dz = sz - tzz;
    const f2 r2 = __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dy, dy, __builtin_elementwise_fma(dx, dx, eps2)));
    const f2 ri = {__builtin_amdgcn_rsqf(r2.x), __builtin_amdgcn_rsqf(r2.y)};
    const f2 mr = sm * ri, mr3 = mr * (ri * ri);
    ax = __builtin_elementwise_fma(dx, mr3, ax), ay = __builtin_elementwise_fma(dy, mr3, ay), az = __builtin_elementwise_fma(dz, mr3, az);
}

template <int MODE>
__global__ void __launch_bounds__(64) k(float *out, int tiles, float seed, unsigned long long *stamps)
{
    __shared__ float4 tile[352]; // 5.5 KiB like the list kernel's per-wave LDS (128 used)
    const int lane = threadIdx.x;
    for (int i = lane; i < 128; i += 64) tile[i] = make_float4(i * 0.37f + seed, i + 1.5f, i * 0.11f + 2, 1.f + i * 1e-3f);
    __syncthreads();
    const int sp = lane >> 5; // two splits of 32 target slots
    const float eps2 = seed * 1e-3f;
    const f2 e2 = {eps2, eps2};
    float t[4][3];
    for (int r = 0; r < 4; ++r) { t[r][0] = seed + lane + 0.3f * r; t[r][1] = t[r][0] * 0.5f; t[r][2] = t[r][0] * 0.25f; }
    float a[4][3] = {};
    f2 pa[2][3] = {};
    const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime(), c0 = __builtin_amdgcn_s_memtime();
    for (int tl = 0; tl < tiles; ++tl) {
        const float4 *p = tile + sp * 64;
        if (MODE == 0) {
#pragma unroll 2
            for (int it = 0; it < 64; ++it) { const float4 s = p[it];
                body1(s, t[0][0], t[0][1], t[0][2], eps2, a[0][0], a[0][1], a[0][2]);
                body1(s, t[1][0], t[1][1], t[1][2], eps2, a[1][0], a[1][1], a[1][2]); }
        } else if (MODE == 1) {
            const f2 tx = {t[0][0], t[1][0]}, ty = {t[0][1], t[1][1]}, tz = {t[0][2], t[1][2]};
#pragma unroll 2
            for (int it = 0; it < 64; ++it) { const float4 s = p[it]; body2(s, tx, ty, tz, e2, pa[0][0], pa[0][1], pa[0][2]); }
        } else if (MODE == 2) {
#pragma unroll 1
            for (int it = 0; it < 64; ++it) { const float4 s = p[it];
#pragma unroll
                for (int r = 0; r < 4; ++r) body1(s, t[r][0], t[r][1], t[r][2], eps2, a[r][0], a[r][1], a[r][2]); }
        } else if (MODE == 3) {
            const f2 tx0 = {t[0][0], t[1][0]}, ty0 = {t[0][1], t[1][1]}, tz0 = {t[0][2], t[1][2]};
            const f2 tx1 = {t[2][0], t[3][0]}, ty1 = {t[2][1], t[3][1]}, tz1 = {t[2][2], t[3][2]};
#pragma unroll 1
            for (int it = 0; it < 64; ++it) { const float4 s = p[it];
                body2(s, tx0, ty0, tz0, e2, pa[0][0], pa[0][1], pa[0][2]);
                body2(s, tx1, ty1, tz1, e2, pa[1][0], pa[1][1], pa[1][2]); }
        } else if (MODE == 4) {
#pragma unroll 4
            for (int it = 0; it < 64; ++it) { const float4 s = p[it]; body1(s, t[0][0], t[0][1], t[0][2], eps2, a[0][0], a[0][1], a[0][2]); }
        } else {
#pragma unroll 2
            for (int it = 0; it < 32; ++it) { const float4 s0 = p[2 * it], s1 = p[2 * it + 1];
                body_s2(s0, s1, t[0][0], t[0][1], t[0][2], e2, pa[0][0], pa[0][1], pa[0][2]); }
        }
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), rt1 = __builtin_amdgcn_s_memrealtime();
    float r = 0;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 3; ++j) r += a[i][j];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 3; ++j) r += pa[i][j].x + pa[i][j].y;
    out[blockIdx.x * 64 + lane] = r;
    if (lane == 0) { stamps[2 * blockIdx.x] = c1 - c0; stamps[2 * blockIdx.x + 1] = rt1 - rt0; }
}

template <int MODE>
int run(const char *name, int R, float *d_out, unsigned long long *d_st, int w)
{
    const int tiles = 600, nblk = 256 * 4 * w;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<MODE>, dim3(nblk), dim3(64), 0, 0, d_out, 20, 1.0f, d_st);
    CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k<MODE>, dim3(nblk), dim3(64), 0, 0, d_out, tiles, 1.0f, d_st);
        CHECK(hipEventRecord(e1));
        CHECK(hipDeviceSynchronize());
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    std::vector<unsigned long long> st(2 * (size_t)nblk);
    CHECK(hipMemcpy(st.data(), d_st, st.size() * 8, hipMemcpyDeviceToHost));
    double ghz = 0; for (int b = 0; b < nblk; ++b) ghz += (double)st[2 * b] / (double)st[2 * b + 1] * 0.1; ghz /= nblk;
    // 64-lane interactions per wave: tiles * 64 iterations * R (both splits work on their half: 64 sources each).
    const double inter = (double)tiles * 64 * R, per_simd = inter * w;
    printf("%-34s waves/SIMD=%d  ms=%7.3f  ns per 64-lane interaction per SIMD=%6.2f  clock %.3f GHz  cycles %.1f\n", name, w, best,
           best * 1e6 / per_simd, ghz, best * 1e6 / per_simd * ghz);
    return 0;
}

int main()
{
    float *d_out; unsigned long long *d_st;
    CHECK(hipMalloc(&d_out, (size_t)256 * 4 * 8 * 64 * 4));
    CHECK(hipMalloc(&d_st, (size_t)256 * 4 * 8 * 16));
    for (int w : {1, 2, 3, 5, 7}) {
        run<0>("R=2 scalar", 2, d_out, d_st, w);
        run<1>("R=2 packed (targets)", 2, d_out, d_st, w);
        run<2>("R=4 scalar", 4, d_out, d_st, w);
        run<3>("R=4 packed (targets)", 4, d_out, d_st, w);
        run<4>("R=1 scalar", 1, d_out, d_st, w);
        run<5>("R=1 packed (two sources)", 1, d_out, d_st, w);
    }
    return 0;
}
