// One wavefront, one dependent chain of fused multiply-adds (the exact-mode node sums of the device tree build): cycles per
// link and the clock the chip runs at while nothing else is resident.
//   mode 0: bare chain from registers      mode 1: operands from LDS, next group read ahead (the kernel's loop)
//   mode 2: as 1, with 3 more wavefronts of the workgroup streaming global memory meanwhile
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int MODE>
__global__ void __launch_bounds__(256) k(float *out, const float4 *g, int groups, unsigned long long *t)
{
    __shared__ __attribute__((aligned(16))) float rows[2][2048];
    const int tid = threadIdx.x;
    for (int i = tid; i < 2048; i += 256) {
        rows[0][i] = 1.0f + i * 1e-7f, rows[1][i] = 1e-9f * i;
    }
    __syncthreads();
    if (tid >= 64) {
        if (MODE == 2) {
            float4 acc = make_float4(0, 0, 0, 0);
            for (int i = tid; i < groups * 8; i += 192) {
                const float4 v = g[i & 0xfffff];
                acc.x += v.x, acc.y += v.y, acc.z += v.z, acc.w += v.w;
            }
            out[tid] = acc.x + acc.y + acc.z + acc.w;
        }
        return;
    }
    float sum = 0.f;
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();
    if (MODE == 0) {
        const float a = rows[0][tid], b = rows[1][tid];
        for (int i = 0; i < groups; ++i) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(sum) : "v"(a), "v"(b));
            }
        }
    } else {
        float4 am0 = *(const float4 *)&rows[0][0], am1 = *(const float4 *)&rows[0][4], ac0 = *(const float4 *)&rows[1][0],
               ac1 = *(const float4 *)&rows[1][4], bm0, bm1, bc0, bc1;
        for (int i = 0; i < groups; i += 2) {
            const int j = ((i + 1) * 8) & 2047, j2 = ((i + 2) * 8) & 2047;
            bm0 = *(const float4 *)&rows[0][j], bm1 = *(const float4 *)&rows[0][j + 4], bc0 = *(const float4 *)&rows[1][j],
            bc1 = *(const float4 *)&rows[1][j + 4];
            __builtin_amdgcn_sched_barrier(0);
            sum = fmaf(am0.x, ac0.x, sum), sum = fmaf(am0.y, ac0.y, sum), sum = fmaf(am0.z, ac0.z, sum), sum = fmaf(am0.w, ac0.w, sum);
            sum = fmaf(am1.x, ac1.x, sum), sum = fmaf(am1.y, ac1.y, sum), sum = fmaf(am1.z, ac1.z, sum), sum = fmaf(am1.w, ac1.w, sum);
            __builtin_amdgcn_sched_barrier(0);
            am0 = *(const float4 *)&rows[0][j2], am1 = *(const float4 *)&rows[0][j2 + 4], ac0 = *(const float4 *)&rows[1][j2],
            ac1 = *(const float4 *)&rows[1][j2 + 4];
            __builtin_amdgcn_sched_barrier(0);
            sum = fmaf(bm0.x, bc0.x, sum), sum = fmaf(bm0.y, bc0.y, sum), sum = fmaf(bm0.z, bc0.z, sum), sum = fmaf(bm0.w, bc0.w, sum);
            sum = fmaf(bm1.x, bc1.x, sum), sum = fmaf(bm1.y, bc1.y, sum), sum = fmaf(bm1.z, bc1.z, sum), sum = fmaf(bm1.w, bc1.w, sum);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    out[tid] = sum;
    if (tid == 0) {
        t[0] = c1 - c0, t[1] = r1 - r0;
    }
}

template <int MODE>
int run(const char *name, float *out, const float4 *g, unsigned long long *t)
{
    const int groups = 1 << 19; // 4M links
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(256), 0, 0, out, g, groups, t);
        CHECK(hipDeviceSynchronize());
    }
    unsigned long long h[2];
    CHECK(hipMemcpy(h, t, sizeof(h), hipMemcpyDeviceToHost));
    const double ns = h[1] * 10.0, links = groups * 8.0;
    printf("%-44s %7.2f ms for 4M links  %5.2f ns/link  clock %5.3f GHz  %5.2f cycles/link\n", name, ns * 1e-6, ns / links,
           h[0] / ns, h[0] / links);
    return 0;
}

int main()
{
    float *out;
    float4 *g;
    unsigned long long *t;
    CHECK(hipMalloc(&out, 4096));
    CHECK(hipMalloc(&g, (1 << 20) * sizeof(float4)));
    CHECK(hipMemset(g, 0, (1 << 20) * sizeof(float4)));
    CHECK(hipMalloc(&t, 64));
    if (run<0>("bare chain (registers)", out, g, t)) return 1;
    if (run<1>("LDS operands, read ahead", out, g, t)) return 1;
    if (run<2>("LDS operands + 3 waves streaming HBM", out, g, t)) return 1;
    return 0;
}
