// The supergroups' common source lists evaluated ONCE per supergroup instead of once per member (round 4).
//
// k_super (rk_kernels_list.hip) leaves per supergroup S -- K consecutive critical nodes -- the list of sources every
// member accepts. Until round 3 each of the K member waves copied that list into its own LDS tile and evaluated it in its
// own lane mapping (R = 1 .. 4 targets per lane, 92 % of the lanes on the 4M tree): 39 % of all interactions of the 4M step.
// The particles of the K members are one contiguous Morton range, so k_common evaluates the list for them in chunks of 256
// targets, one independent wavefront per chunk: 4 targets per lane on full lanes (the 52-instruction body that shares one
// broadcast LDS read among four targets); the last, partial chunk of a supergroup uses the list kernel's generic mapping
// (TP target slots x NS source splits, R = 4). The per-target sums go to sup_part[k][particle]; list_node / pc_node start
// the accumulators of their split 0 from them and skip the common list.
// A target's sum is formed in an order that depends on the tree and the MAC value only (list order; in a partial chunk the
// splits' contiguous shares of every tile, added in split order at the end), so shard unions, the list and the producer /
// consumer kernels and repeated calls still agree bit for bit. The decisions, hence the interaction set, are unchanged.
//
//
// NOT the default (rk_set_common_eval / RK_COMMON=1 select it): measured on MI355X it loses at every size -- 4M fp32:
// 2.42 ms per step against 2.27 -- because the members already evaluate these lists at the chip's practical issue rate and
// their dense work is what hides the latency of their own list building; only 1 % of the VALU instructions go away
// (1.712e9 against 1.733e9). profiles/r04/common_eval_*.txt, DESIGN.md section 3.7. It stays as a tested alternative
// summation order (an independent check of the members' common-list handling) and as the record of the experiment.
// (First built fused with the pre-pass -- one workgroup per supergroup, wave 0 walking while the others wait for the tile:
// 862 us at 4M against 737 us that the members spend on the same lists.)
#define RK_UNR4 2 // two sources in flight per lane in this file's dense loop (4M: 2.42 against 2.49 ms per step with one)
#include "rk_list_common.hpp"

namespace rk
{

constexpr int SE_R = 4;        // targets per lane
#ifndef RK_SE_CHUNK
#define RK_SE_CHUNK 64 // targets per chunk (= per wavefront): 64 -> 16 target slots x 4 source splits
#endif
constexpr int SE_CHUNK = RK_SE_CHUNK;
static_assert(SE_CHUNK <= 64 * SE_R && SE_CHUNK % SE_R == 0);
#ifndef RK_SE_WPS
#define RK_SE_WPS 12 // wavefronts (single-wave workgroups) launched per supergroup; wave c serves chunks c, c + RK_SE_WPS, ...
#endif
#ifndef RK_SE_TILE
#define RK_SE_TILE 256 // sources per LDS tile
#endif
#ifndef RK_SE_W32
#define RK_SE_W32 6 // waves per SIMD the fp32 kernels are compiled for
#endif
#ifndef RK_SE_W64
#define RK_SE_W64 4
#endif

template <typename F>
struct se_lds {
    typename vt<F>::v4 tile[RK_SE_TILE];
    F red[64 * 4]; // split reduction, one target slot r at a time
};

template <typename F, int Q, int ND>
__global__ void __launch_bounds__(64, sizeof(F) == 4 ? RK_SE_W32 : RK_SE_W64)
    k_common(const kparams<F> P, uint32_t s_begin, uint32_t s_end)
{
    using v4 = typename vt<F>::v4;
    constexpr int NR = nres_of(Q);
    constexpr int CAP = RK_SE_TILE;
    __shared__ se_lds<F> L;
    const int lane = threadIdx.x;
    // One contiguous slice of the (supergroup, wave) pairs per XCD, as the member kernels have of the critical nodes.
    const unsigned blk = xcd_chunked_block(blockIdx.x, gridDim.x);
    const uint32_t S = __builtin_amdgcn_readfirstlane(s_begin + blk / unsigned(RK_SE_WPS));
    const int c0 = static_cast<int>(blk % unsigned(RK_SE_WPS));
    if (S >= s_end) {
        return;
    }
    const uint32_t K = P.super_k;
    const uint32_t g0 = S * K, g1 = (g0 + K < P.n_crit) ? g0 + K : P.n_crit;
    // The members' particles: one contiguous Morton range (the critical nodes cover [0, N) in order).
    const uint32_t pb = __builtin_amdgcn_readfirstlane(P.crit[g0].x), pe = __builtin_amdgcn_readfirstlane(P.crit[g1 - 1u].y);
    const int TS = static_cast<int>(pe - pb);
    const int nfull = TS / SE_CHUNK, rem = TS % SE_CHUNK, nchunks = nfull + (rem != 0 ? 1 : 0);
    if (c0 >= nchunks) {
        return;
    }
    const uint2 cnt = P.sup_cnt[S];
    // A supergroup whose pre-pass overflowed is redone from the root by every member: its sums are zero.
    const int n_common = (cnt.y >> 31) ? 0 : static_cast<int>(__builtin_amdgcn_readfirstlane(cnt.x));
    const v4 *common = P.sup_common + static_cast<size_t>(S) * SUP_CAPC;
    const F eps2 = P.eps2;
    const size_t stride = P.sup_part_stride;

    for (int c = c0; c < nchunks; c += RK_SE_WPS) {
        const uint32_t tb = pb + static_cast<uint32_t>(c * SE_CHUNK);
        const int T = c < nfull ? SE_CHUNK : rem;
        const int TP = (T + SE_R - 1) / SE_R;
        const int NS = 64 / TP;
        const int ts = lane % TP, sp_raw = lane / TP;
        const bool lane_on = sp_raw < NS;
        const int sp = lane_on ? sp_raw : 0;
        const bool owner = lane_on && sp_raw == 0;
        v4 tp[SE_R];
        int tidx[SE_R];
        F acc[SE_R][NR];
#pragma unroll
        for (int r = 0; r < SE_R; ++r) {
            tidx[r] = ts + r * TP;
            const bool valid = tidx[r] < T;
            tp[r] = P.part4[tb + (valid ? tidx[r] : 0)];
            if (!valid) {
                tidx[r] = -1;
            }
#pragma unroll
            for (int k = 0; k < NR; ++k) {
                acc[r][k] = F(0);
            }
        }
        for (int base = 0; base < n_common; base += CAP) {
            const int n = n_common - base < CAP ? n_common - base : CAP;
            for (int j = lane; j < n; j += 64) {
                L.tile[j] = common[base + j];
            }
            wave_sync();
            lk_eval_tile<F, Q, SE_R, false, ND>(L.tile, n, n / NS, sp, NS, true, lane_on, tp, acc, eps2, tidx);
            wave_sync();
        }
        if (NS > 1) {
            F *red = L.red;
#pragma unroll
            for (int r = 0; r < SE_R; ++r) {
                if (lane_on) {
#pragma unroll
                    for (int k = 0; k < NR; ++k) {
                        red[(sp_raw * TP + ts) * NR + k] = acc[r][k];
                    }
                }
                wave_sync();
                if (owner) {
#pragma unroll
                    for (int k = 0; k < NR; ++k) {
                        F sum = F(0);
                        for (int s = 0; s < NS; ++s) {
                            sum += red[(s * TP + ts) * NR + k];
                        }
                        acc[r][k] = sum;
                    }
                }
                wave_sync();
            }
        }
        if (owner) {
#pragma unroll
            for (int r = 0; r < SE_R; ++r) {
                if (tidx[r] >= 0) {
#pragma unroll
                    for (int k = 0; k < NR; ++k) {
                        P.sup_part[static_cast<size_t>(k) * stride + tb + static_cast<uint32_t>(tidx[r])] = acc[r][k];
                    }
                }
            }
        }
    }
}

template <typename F>
void launch_common(const rk_state &s, int q, const kparams<F> &p, int64_t s_begin, int64_t s_end, hipStream_t stream)
{
    const int64_t n = s_end - s_begin;
    if (n <= 0 || !p.super_k) {
        return;
    }
    const dim3 grid(static_cast<unsigned>(n * RK_SE_WPS)), block(64);
    const auto sb = static_cast<uint32_t>(s_begin), se = static_cast<uint32_t>(s_end);
    auto go = [&](auto Qt, auto Mt) {
        constexpr int Q = decltype(Qt)::value;
        (void)Mt; // the MAC was the pre-pass's business
        if (s.ndim == 3 || !RK_QUAD_BODY) {
            hipLaunchKernelGGL((k_common<F, Q, 3>), grid, block, 0, stream, p, sb, se);
        } else {
            hipLaunchKernelGGL((k_common<F, Q, 2>), grid, block, 0, stream, p, sb, se);
        }
    };
    using i0 = std::integral_constant<int, 0>;
    using i1 = std::integral_constant<int, 1>;
    using i2 = std::integral_constant<int, 2>;
    switch (q * 2 + s.mac) {
        case 0: go(i0{}, i0{}); break;
        case 1: go(i0{}, i1{}); break;
        case 2: go(i1{}, i0{}); break;
        case 3: go(i1{}, i1{}); break;
        case 4: go(i2{}, i0{}); break;
        case 5: go(i2{}, i1{}); break;
        default: throw error(RK_EINVAL, "invalid q / mac combination");
    }
    RK_HIP(hipGetLastError());
}
template void launch_common<float>(const rk_state &, int, const kparams<float> &, int64_t, int64_t, hipStream_t);
template void launch_common<double>(const rk_state &, int, const kparams<double> &, int64_t, int64_t, hipStream_t);

void touch_common()
{
    hipFuncAttributes attr{};
    RK_HIP(hipFuncGetAttributes(&attr, reinterpret_cast<const void *>(&k_common<float, 0, 3>)));
}

} // namespace rk
