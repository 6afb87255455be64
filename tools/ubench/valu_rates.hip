// Microbenchmark: VALU issue rates on gfx950 that shape the traversal kernel design.
// Measures cycles per wave-instruction (per SIMD) for v_fma_f32, v_pk_fma_f32, v_rsq_f32,
// a full monopole interaction body, and LDS broadcast reads, at 1..8 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

typedef float float2v __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void __launch_bounds__(256) k(float *out, int iters, float seed)
{
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float2v p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};
    const float c = seed * 0.5f, d = seed * 0.25f;
    const float2v pc = {c, c}, pd = {d, d};
    __shared__ float4 lds[1024];
    if (MODE == 4 || MODE == 5) {
        for (int i = threadIdx.x; i < 1024; i += blockDim.x) lds[i] = make_float4(i, i + 1, i + 2, 1.f);
        __syncthreads();
    }
    // Shader clock (s_memtime) against the fixed 100 MHz counter (s_memrealtime): the clock the chip SUSTAINS while this
    // row runs is d(memtime) / d(memrealtime) * 100 MHz -- cycles per instruction are quoted at that clock, not at 2.4 GHz.
    const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime();
    long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) { // plain fma, 8 independent chains
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                             "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));
            }
        } else if (MODE == 1) { // packed fma
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                             "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pc), "v"(pd));
            }
        } else if (MODE == 2) { // rsq
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                asm volatile("v_rsq_f32 %0, %0\n v_rsq_f32 %1, %1\n v_rsq_f32 %2, %2\n v_rsq_f32 %3, %3\n"
                             "v_rsq_f32 %4, %4\n v_rsq_f32 %5, %5\n v_rsq_f32 %6, %6\n v_rsq_f32 %7, %7\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            }
        } else if (MODE == 3) { // 3 fma : 1 rsq interleaved (does the transcendental overlap with fma?)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                asm volatile("v_rsq_f32 %0, %0\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                             "v_rsq_f32 %4, %4\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));
            }
        } else if (MODE == 4) { // interaction body, source from LDS broadcast (ds_read_b128), 1 target/lane
#pragma unroll 8
            for (int u = 0; u < 32; ++u) {
                float4 s = lds[(i * 32 + u) & 1023];
                float dx = s.x - a0, dy = s.y - a1, dz = s.z - a2;
                float r2 = __fmaf_rn(dz, dz, __fmaf_rn(dy, dy, __fmaf_rn(dx, dx, c)));
                float ri = __builtin_amdgcn_rsqf(r2);
                float mr = s.w * ri;
                float ri2 = ri * ri;
                float mr3 = mr * ri2;
                a3 = __fmaf_rn(dx, mr3, a3); a4 = __fmaf_rn(dy, mr3, a4); a5 = __fmaf_rn(dz, mr3, a5);
            }
        } else if (MODE == 5) { // interaction body, 2 targets/lane
#pragma unroll 8
            for (int u = 0; u < 32; ++u) {
                float4 s = lds[(i * 32 + u) & 1023];
                {
                    float dx = s.x - a0, dy = s.y - a1, dz = s.z - a2;
                    float r2 = __fmaf_rn(dz, dz, __fmaf_rn(dy, dy, __fmaf_rn(dx, dx, c)));
                    float ri = __builtin_amdgcn_rsqf(r2);
                    float mr = s.w * ri; float ri2 = ri * ri; float mr3 = mr * ri2;
                    a3 = __fmaf_rn(dx, mr3, a3); a4 = __fmaf_rn(dy, mr3, a4); a5 = __fmaf_rn(dz, mr3, a5);
                }
                {
                    float dx = s.x - p0.x, dy = s.y - p0.y, dz = s.z - p1.x;
                    float r2 = __fmaf_rn(dz, dz, __fmaf_rn(dy, dy, __fmaf_rn(dx, dx, c)));
                    float ri = __builtin_amdgcn_rsqf(r2);
                    float mr = s.w * ri; float ri2 = ri * ri; float mr3 = mr * ri2;
                    a6 = __fmaf_rn(dx, mr3, a6); a7 = __fmaf_rn(dy, mr3, a7); p2.x = __fmaf_rn(dz, mr3, p2.x);
                }
            }
        } else if (MODE == 6) { // plain v_sub with SGPR operand + fma mix (MAC test body): 3 sub(s,v) + 3 fma + 1 min
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float sx = __builtin_amdgcn_readfirstlane(a7) + u;
                asm volatile("v_sub_f32 %0, %3, %0\n v_sub_f32 %1, %3, %1\n v_sub_f32 %2, %3, %2\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2) : "s"(sx));
                asm volatile("v_fma_f32 %0, %1, %1, %0\n v_fma_f32 %0, %2, %2, %0\n v_fma_f32 %0, %3, %3, %0\n v_min_f32 %4, %4, %0\n"
                             : "+v"(a3), "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a4));
            }
        }
    }
    long long t1 = clock64();
    const unsigned long long rt1 = __builtin_amdgcn_s_memrealtime();
    float r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y + p4.x + p5.x + p6.x + p7.x;
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if (threadIdx.x == 0) {
        ((long long *)(out + gridDim.x * blockDim.x))[2 * blockIdx.x] = t1 - t0;
        ((long long *)(out + gridDim.x * blockDim.x))[2 * blockIdx.x + 1] = static_cast<long long>(rt1 - rt0);
    }
}

template <int MODE>
int run(const char *name, int ninstr_per_iter, float *d_out, int nblk_per_cu, int threads)
{
    const int iters = 4000;
    int nblk = 256 * nblk_per_cu;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<MODE>, dim3(nblk), dim3(threads), 0, 0, d_out, 10, 1.0f);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<MODE>, dim3(nblk), dim3(threads), 0, 0, d_out, iters, 1.0f);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<long long> cyc(2 * (size_t)nblk);
    CHECK(hipMemcpy(cyc.data(), d_out + (size_t)nblk * threads, 2 * nblk * sizeof(long long), hipMemcpyDeviceToHost));
    double avg = 0, ghz = 0;
    for (int b = 0; b < nblk; ++b) {
        avg += cyc[2 * b];
        ghz += (double)cyc[2 * b] / (double)cyc[2 * b + 1] * 0.1; // shader cycles per 10 ns tick
    }
    avg /= nblk;
    ghz /= nblk;
    double waves_per_simd = (double)nblk_per_cu * threads / 64 / 4;
    double winstr = (double)iters * ninstr_per_iter; // wave-instrs per wave
    // wall-based: total wave-instr per SIMD / time
    double per_simd_instr = winstr * waves_per_simd;
    double ns_per = ms * 1e6 / per_simd_instr;
    printf("%-28s waves/SIMD=%4.1f  ms=%8.3f  ns/wave-instr/SIMD=%6.3f  sustained clock=%5.3f GHz  cyc/wave-instr/SIMD at that clock=%5.2f  "
           "clock64-cyc/instr/wave=%6.2f\n", name, waves_per_simd, ms, ns_per, ghz, ns_per * ghz, avg / winstr);
    return 0;
}

int main()
{
    float *d_out;
    CHECK(hipMalloc(&d_out, (size_t)256 * 8 * 256 * 4 + 256 * 8 * 16 + 4096));
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    printf("device %s CUs=%d clock=%d kHz\n", prop.name, prop.multiProcessorCount, prop.clockRate);
    for (int w : {1, 2, 4, 8}) { // blocks of 256 threads (1 wave per SIMD each) per CU
        run<0>("v_fma_f32", 32, d_out, w, 256);
        run<1>("v_pk_fma_f32", 32, d_out, w, 256);
        run<2>("v_rsq_f32", 32, d_out, w, 256);
        run<3>("3fma:1rsq mix", 32, d_out, w, 256);
        run<4>("interaction(13 valu) 1tgt", 32 * 13, d_out, w, 256);
        run<5>("interaction(26 valu) 2tgt", 32 * 26, d_out, w, 256);
        run<6>("mac body 3sub(s)+3fma+min", 28, d_out, w, 256);
    }
    return 0;
}
