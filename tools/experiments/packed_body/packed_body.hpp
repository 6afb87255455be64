// Round 5 experiment, NOT part of the library: the fp32 dense phase with pairs of targets in 64-bit register pairs and
// v_pk_{add,mul,fma}_f32 (12 packed operations + 2 v_rsq_f32 per source and pair of targets instead of 26 scalar ones).
// These are the two pieces that sat in rakau_amd/csrc/rk_list_common.hpp behind RK_PK_BODY (commit 69c3dd2 has them in
// place; lk_regs<F, Q, R> is the scalar flavour that stayed). Bit-identical to the scalar bodies (same IEEE operations, same
// order) -- and not faster: see README.md next to this file.
#if RK_PK_BODY
// ------------------------------------------------------------------------------------------------
// Packed fp32 interaction body: two targets of a lane in one 64-bit register pair, every operation of lk_interact_src() /
// interact() as one v_pk_{add,mul,fma}_f32 on the pair (the source is broadcast into both halves by op_sel), v_rsq_f32 per
// half. A wave64 v_fma_f32 occupies 16 of a SIMD's 32 fp32 lanes for 4 cycles, a v_pk_fma_f32 all 32: the same arithmetic
// throughput when two wavefronts alternate, twice the rate whenever a wavefront finds the SIMD to itself (launch tails,
// small launches, the other waves parked on memory) -- and half the instruction issues either way. Every component is the
// IEEE operation the scalar body performs, in the same order: the bits do not change.
// ------------------------------------------------------------------------------------------------
typedef float rk_f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ rk_f2 rk_pk_fma(rk_f2 a, rk_f2 b, rk_f2 c)
{
    return __builtin_elementwise_fma(a, b, c);
}

// One source on one pair of targets. tq = {x, y, z, m} of the pair; tidx = list positions of the two targets (SELF only).
template <int Q, bool SELF, int ND>
__device__ __forceinline__ void lk_interact_pair(const float4 &s, int j, const rk_f2 (&tq)[4], rk_f2 (&acc)[nres_of(Q)], rk_f2 eps2,
                                                 int tidx0, int tidx1)
{
    const rk_f2 sx = {s.x, s.x}, sy = {s.y, s.y}, sz = {s.z, s.z};
    const rk_f2 ex = sx - tq[0], ey = sy - tq[1];
    rk_f2 ez = {0.f, 0.f};
    rk_f2 e2 = rk_pk_fma(ey, ey, rk_pk_fma(ex, ex, eps2));
    if constexpr (ND == 3) {
        ez = sz - tq[2];
        e2 = rk_pk_fma(ez, ez, e2);
    }
    rk_f2 ms = {s.w, s.w};
    if constexpr (SELF) {
        const bool self0 = (j == tidx0), self1 = (j == tidx1);
        e2.x = self0 ? 1.f : e2.x;
        e2.y = self1 ? 1.f : e2.y;
        ms.x = self0 ? 0.f : ms.x;
        ms.y = self1 ? 0.f : ms.y;
    }
    const rk_f2 rinv = {rk_rsqrt(e2.x), rk_rsqrt(e2.y)};
    rk_f2 mr;
    if constexpr (SELF) {
        mr = ms * rinv;
    } else {
        // {m, m} x {1/r0, 1/r1} with the mass taken from the HIGH half of the source's {z, m} register pair by op_sel. The
        // compiler has this form for additions and not for this multiplication (it copies m into a fresh pair first: one
        // v_mov_b32 per source, a sixth of what packing saves); the same IEEE multiplication either way.
        const rk_f2 zm = {s.z, s.w};
        asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]" : "=v"(mr) : "v"(zm), "v"(rinv));
    }
    if constexpr (Q == 0 || Q == 2) {
        const rk_f2 mr3 = mr * (rinv * rinv);
        acc[0] = rk_pk_fma(ex, mr3, acc[0]);
        acc[1] = rk_pk_fma(ey, mr3, acc[1]);
        if constexpr (ND == 3) {
            acc[2] = rk_pk_fma(ez, mr3, acc[2]);
        }
    }
    if constexpr (Q == 1) {
        acc[0] = rk_pk_fma(-tq[3], mr, acc[0]);
    }
    if constexpr (Q == 2) {
        acc[3] = rk_pk_fma(-tq[3], mr, acc[3]);
    }
}
#endif

#if RK_PK_BODY
template <int Q, int R>
struct lk_regs<float, Q, R, true> {
    using v4 = float4;
    static constexpr int NR = nres_of(Q), NP = R / 2;
    static constexpr bool packed = true, odd = (R % 2) != 0;
    rk_f2 tq[NP][4], pa[NP][NR];
    v4 tl;        // odd last target (R = 3)
    float al[NR]; // ... and its sums
    __device__ __forceinline__ void set_target(int r, const v4 &p)
    {
        if (odd && r == R - 1) {
            tl = p;
#pragma unroll
            for (int k = 0; k < NR; ++k) {
                al[k] = 0.f;
            }
            return;
        }
        const int q = r >> 1;
        if (r & 1) {
            tq[q][0].y = p.x, tq[q][1].y = p.y, tq[q][2].y = p.z, tq[q][3].y = p.w;
#pragma unroll
            for (int k = 0; k < NR; ++k) {
                pa[q][k].y = 0.f;
            }
        } else {
            tq[q][0].x = p.x, tq[q][1].x = p.y, tq[q][2].x = p.z, tq[q][3].x = p.w;
#pragma unroll
            for (int k = 0; k < NR; ++k) {
                pa[q][k].x = 0.f;
            }
        }
    }
    __device__ __forceinline__ float tcomp(int r, int c) const
    {
        if (odd && r == R - 1) {
            return c == 0 ? tl.x : (c == 1 ? tl.y : tl.z);
        }
        return (r & 1) ? tq[r >> 1][c].y : tq[r >> 1][c].x;
    }
    __device__ __forceinline__ float tx(int r) const { return tcomp(r, 0); }
    __device__ __forceinline__ float ty(int r) const { return tcomp(r, 1); }
    __device__ __forceinline__ float tz(int r) const { return tcomp(r, 2); }
    __device__ __forceinline__ float get(int r, int k) const
    {
        if (odd && r == R - 1) {
            return al[k];
        }
        return (r & 1) ? pa[r >> 1][k].y : pa[r >> 1][k].x;
    }
    __device__ __forceinline__ void set(int r, int k, float v)
    {
        if (odd && r == R - 1) {
            al[k] = v;
        } else if (r & 1) {
            pa[r >> 1][k].y = v;
        } else {
            pa[r >> 1][k].x = v;
        }
    }
    template <bool SELF, int ND>
    __device__ __forceinline__ void interact_all(const v4 &s, int j, float eps2, const int (&tidx)[R])
    {
        const rk_f2 eps2p = {eps2, eps2};
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            lk_interact_pair<Q, SELF, ND>(s, j, tq[p], pa[p], eps2p, tidx[2 * p], tidx[2 * p + 1]);
        }
        if constexpr (odd) {
            const float ex = s.x - tl.x, ey = s.y - tl.y, ez = ND == 3 ? s.z - tl.z : 0.f;
            float e2 = rk_fma(ey, ey, rk_fma(ex, ex, eps2));
            if constexpr (ND == 3) {
                e2 = rk_fma(ez, ez, e2);
            }
            float ms = s.w;
            if constexpr (SELF) {
                const bool self = (j == tidx[R - 1]);
                e2 = self ? 1.f : e2;
                ms = self ? 0.f : ms;
            }
            interact<float, Q, ND>(al, ex, ey, ez, e2, ms, tl.w);
        }
    }
};
#endif
