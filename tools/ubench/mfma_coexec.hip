// Do fp32 MFMA instructions and fp32 VALU instructions of DIFFERENT waves on one SIMD execute side by side on gfx950?
// (The fp32 matrix peak equals the fp32 vector peak, 64 flop / clk / SIMD: shared multipliers would look exactly like that.)
// Workgroups of 512 threads: waves w and w + 4 share a SIMD. Three launches with 8 waves per SIMD:
//   all waves VALU (a loop of 8 independent v_fma_f32 chains),
//   all waves MFMA (a loop of 4 independent v_mfma_f32_4x4x1_16B_f32 / v_mfma_f32_32x32x2_f32 accumulators),
//   waves 0-3 of every workgroup VALU, waves 4-7 MFMA (each SIMD: four of each kind, same work per wave as above).
// Separate pipes: mixed ~ max(VALU, MFMA) / 2. One pipe: mixed ~ (VALU + MFMA) / 2.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));

template <int KIND> // 0: 4x4x1, 1: 32x32x2
__device__ __forceinline__ float mfma_loop(int iters, float seed)
{
    if constexpr (KIND == 0) {
        v4f d0{0, 0, 0, 0}, d1 = d0, d2 = d0, d3 = d0;
        float a = seed, b = seed * 0.5f;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                d0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, d0, 0, 0, 0);
                d1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, d1, 0, 0, 0);
                d2 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, d2, 0, 0, 0);
                d3 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, d3, 0, 0, 0);
            }
        }
        return d0[0] + d1[1] + d2[2] + d3[3];
    } else {
        v16f d0{}, d1{};
        float a = seed, b = seed * 0.5f;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                d0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, d0, 0, 0, 0);
                d1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, d1, 0, 0, 0);
            }
        }
        return d0[0] + d1[5];
    }
}

__device__ __forceinline__ float valu_loop(int iters, float seed)
{
    float a[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] = seed + k;
    const float m = 1.0001f, c = seed * 1e-3f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int k = 0; k < 8; ++k) a[k] = __builtin_fmaf(a[k], m, c);
        }
    }
    float s = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += a[k];
    return s;
}

// mode 0: all VALU; 1: all MFMA; 2: waves 0-3 VALU, 4-7 MFMA
template <int KIND>
__global__ void __launch_bounds__(512) k(float *out, int mode, int iters_valu, int iters_mfma, float seed)
{
    const int wave = threadIdx.x >> 6;
    const bool mfma = mode == 1 || (mode == 2 && wave >= 4);
    float r;
    if (mfma) {
        r = mfma_loop<KIND>(iters_mfma, seed);
    } else {
        r = valu_loop(iters_valu, seed);
    }
    out[blockIdx.x * 512 + threadIdx.x] = r;
}

template <int KIND>
int run(const char *name, float *d_out)
{
    const int nblk = 256 * 4; // 4 workgroups of 8 waves per CU: 8 waves per SIMD
    const int iv = 20000, im = KIND == 0 ? 10000 : 2500;
    float ms[3];
    for (int mode = 0; mode < 3; ++mode) {
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0));
        CHECK(hipEventCreate(&e1));
        hipLaunchKernelGGL((k<KIND>), dim3(nblk), dim3(512), 0, 0, d_out, mode, iv / 10, im / 10, 1.0f);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((k<KIND>), dim3(nblk), dim3(512), 0, 0, d_out, mode, iv, im, 1.0f);
        CHECK(hipEventRecord(e1));
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventElapsedTime(&ms[mode], e0, e1));
    }
    const double vi = double(iv) * 32, mi = double(im) * (KIND == 0 ? 16 : 4);
    printf("%s: all-VALU %.3f ms (%.2f cycles@2.4GHz per instr per SIMD), all-MFMA %.3f ms (%.2f per instr), mixed (half the waves each) %.3f ms;"
           " separate pipes would give %.3f, one pipe %.3f\n",
           name, ms[0], ms[0] * 1e-3 * 2.4e9 / (vi * 8), ms[1], ms[1] * 1e-3 * 2.4e9 / (mi * 8), ms[2], (ms[0] > ms[1] ? ms[0] : ms[1]) / 2,
           (ms[0] + ms[1]) / 2);
    return 0;
}

int main()
{
    float *d_out;
    CHECK(hipMalloc(&d_out, 256 * 4 * 512 * 4));
    for (int rep = 0; rep < 2; ++rep) {
        run<0>("v_mfma_f32_4x4x1_16B_f32", d_out);
        run<1>("v_mfma_f32_32x32x2_f32  ", d_out);
    }
    return 0;
}
