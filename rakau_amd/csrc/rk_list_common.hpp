// Pieces shared by the interaction-list kernels (rk_kernels_list.hip: one wave per target group; rk_kernels_pc.hip:
// producer / consumer waves per target group): tuning knobs, per-wave LDS layout, block -> group mapping, the dense
// targets x sources evaluation of one LDS tile.
#ifndef RK_LIST_COMMON_HPP
#define RK_LIST_COMMON_HPP

#include "rk_common.hpp"
#include "rk_device.hpp"

// Tuning knobs (defaults chosen by measurement on MI355X).
#ifndef RK_UNR1
#define RK_UNR1 4 // sources in flight per lane in the dense loop, R = 1
#endif
#ifndef RK_UNR2
#define RK_UNR2 1 // R = 2 (round 5, at 8 waves per SIMD: 1M 0.593 -> 0.584 ms, 4M -0.2 %; rounds 1-4: 2)
#endif
#ifndef RK_UNR3
#define RK_UNR3 1 // R = 3
#endif
#ifndef RK_UNR4
#define RK_UNR4 1 // R = 4
#endif
#ifndef RK_CHUNKED_SPLITS
#define RK_CHUNKED_SPLITS 1 // dense phase: contiguous (1) or interleaved (0) assignment of tile sources to splits
#endif
#ifndef RK_QUAD_BODY
#define RK_QUAD_BODY 1 // quadtrees: interaction body without the z terms (0: the 3-D body on z = 0 data)
#endif
#ifndef RK_CARRY_REMAINDER
#define RK_CARRY_REMAINDER 1 // dense phase: carry the sources that do not fill a round of NS to the next tile instead of a masked step
                             // per tile. Round 2 (7 waves per SIMD) measured it slower, 2.30 vs 2.27 ms; on round 5's kernels it is
                             // ahead: 1M 0.590 -> 0.583 ms, 4M 2.177 -> 2.160 (tools/archive/jobs_r05/r05_job30.sh). The producer of the
                             // producer / consumer kernel hands over the same tiles (rk_kernels_pc.hip, publish()): same bits
#endif
#ifndef RK_EXACT_TRANSPOSED
#define RK_EXACT_TRANSPOSED 1 // exact MAC test with lane = target when only a few candidates are queued
#endif
#ifndef RK_LEAF_LANE_PARTICLE
#define RK_LEAF_LANE_PARTICLE 1 // leaf gathering: lane = particle (1) or lane = leaf with eight records in flight (0: rounds 1-4).
                                // 4M 2.235 -> 2.213 ms, 1M 0.612 -> 0.592, 0.5M 0.321 -> 0.317 (tools/archive/jobs_r05/r05_job6.sh)
#endif
#ifndef RK_MASK_IDLE
#define RK_MASK_IDLE 1 // dense phase: the lanes left over by TP x NS < 64 are switched off for the tile (1) or shadow split 0 (0:
                       // rounds 1-4). Same instruction count, 0.7 % less time at 4M, 1.5-2 % on 100k-350k launches
                       // (tools/archive/jobs_r05/r05_job4.sh): 6.8 % of the lanes no longer read LDS and burn issue power for nothing
#endif
#ifndef RK_WPB
#define RK_WPB 1 // wavefronts (= target groups) per workgroup. Measured 4 -> 2.58 ms, 2 -> 2.45, 1 -> 2.35 at 4M: a block keeps its
                 // LDS and wave slots until its slowest group ends, single-wave blocks free them at once (waves never sync)
#endif
#ifndef RK_W64
#define RK_W64 4 // fp64 kernels (all R): waves per SIMD they are compiled for (3: 63.3 ms, 4: 62.8, 5: 67.3 at 16M)
#endif
#ifndef RK_W64_R3
#define RK_W64_R3 4 // fp64, R = 3 class kernel
#endif
#ifndef RK_W64_ANY
#define RK_W64_ANY 4 // fp64: k_list_any and the kernel for oversized nodes
#endif
#ifndef RK_W64_R4
#define RK_W64_R4 3 // fp64, R = 4 class kernel: 165 VGPRs, no scratch (at 4 waves: 128 VGPRs + 80 bytes of scratch per lane). 16M fp64
                    // theta 0.5: 59.05 against 59.65 ms per step, same box, two rounds (tools/archive/jobs_r04/r04_job36.sh); the R = 3
                    // kernel at 3 waves as well: 60.0
#endif
#ifndef RK_W12
#define RK_W12 8 // waves per SIMD the R <= 2 kernels are compiled for. Round 5: 8 (64 VGPRs, 12 / 44 bytes of scratch, 5 KiB of LDS per
                 // wave) once the DPP prefix sum had freed six address registers and six lane masks: 4M 2.215 -> 2.195 ms, the
                 // seam's call 1717-1728 -> 1759-1767 Mparticles/s (tools/archive/jobs_r05/r05_job11.sh); rounds 1-4: 7 (8 spilled 50-60 B)
#endif
#ifndef RK_W3
#define RK_W3 6 // R = 3
#endif
#ifndef RK_W4
#define RK_W4 6 // R = 4 (round 5: 80 VGPRs, 36 bytes of scratch: 4M 2.189 -> 2.156 ms over four alternating rounds,
                // tools/archive/jobs_r05/r05_job19.sh; rounds 1-4: 5)
#endif
#ifndef RK_WANY
#define RK_WANY 5 // k_list_any (one launch over all classes), compiled for the registers of its largest R at 5 waves per SIMD (6: 1M
                  // 0.596 -> 0.576 ms but 350k 0.224 -> 0.236, shards of the 4M tree equal: tools/archive/jobs_r05/r05_job12.sh, _job18)
#endif
#ifndef RK_W5
#define RK_W5 4 // R = 5
#endif
#ifndef RK_W6
#define RK_W6 3 // R = 6
#endif


// RK_MAC_RUNTIME=1 instantiates the list and producer / consumer kernels once, with RK_MAC_RT: the criterion is then read
// from kparams::mac at run time (a wave-uniform select in mac_lhs(), twice per batch of 64 candidates). Same arithmetic per
// criterion, hence the same bits, and half the code objects (-4 MB of library) -- but NOT the default: the R = 1 and R = 3 list
// kernels then spill 8 more bytes per lane (28 / 32 instead of 20 / 24; they are compiled at the 72 / 80-register cliff), which
// leaves results-in-HBM calls unchanged and makes calls whose epilogue stores cross PCIe (host outputs) 1.5-2 % slower
// (4M: 2.36-2.39 against 2.32-2.33 ms, same box, tools/archive/jobs_r04/r04_job7.sh, r04_job11.sh).
#ifndef RK_MAC_RUNTIME
#define RK_MAC_RUNTIME 0
#endif

namespace rk
{

constexpr int RK_MAC_RT = -1;
// Template argument the launchers instantiate for a state's criterion.
constexpr int mac_targ(int mac)
{
    return RK_MAC_RUNTIME ? RK_MAC_RT : mac;
}

#ifdef RK_STAMPS
// Diagnostic build: wave-lifetime cycles per section (s_memtime), summed over waves into P.dbg[].
#define RK_STAMP_DECL unsigned long long st_t0 = __builtin_amdgcn_s_memtime(), st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define RK_STAMP(i)                                                                                                    \
    {                                                                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        const unsigned long long st_t1 = __builtin_amdgcn_s_memtime();                                                 \
        __builtin_amdgcn_s_waitcnt(0xC07F);                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        st_acc[i] += st_t1 - st_t0;                                                                                    \
        st_t0 = st_t1;                                                                                                 \
    }
#define RK_STAMP_FLUSH                                                                                                 \
    if (lane == 0 && P.dbg) {                                                                                          \
        for (int i_ = 0; i_ < 8; ++i_) atomicAdd(&P.dbg[i_], st_acc[i_]);                                              \
    }
#else
#define RK_STAMP_DECL
#define RK_STAMP(i)
#define RK_STAMP_FLUSH
#endif

#ifdef RK_COUNTS
// Diagnostic build: dynamic event counts per lane-mapping class (wave-uniform integers, added to P.dbg[(R - 1) * 32 + i] when the
// node is done): batches, candidates, exact tests, leaf gathering, tiles, dense-loop trips, remainder steps -- the multipliers of
// the per-section instruction counts in the instruction budget of DESIGN.md section 4 (tools/counts_probe.py prints them).
#define RK_COUNT_DECL long long cn_[32] = {};
#define RK_COUNT(i, v) cn_[i] += static_cast<long long>(v);
#define RK_COUNT_FLUSH                                                                                                 \
    if (lane == 0 && P.dbg) {                                                                                          \
        for (int i_ = 0; i_ < 32; ++i_) atomicAdd(&P.dbg[(R - 1) * 32 + i_], static_cast<unsigned long long>(cn_[i_])); \
    }
#else
#define RK_COUNT_DECL
#define RK_COUNT(i, v)
#define RK_COUNT_FLUSH
#endif

// Stack of pending sibling runs; an entry = (first child record << 3) | (number of children - 1) names up to
// 8 candidate nodes. Popping k entries can push at most 8k (every candidate opened).
#ifndef RK_STACK_CAP
#define RK_STACK_CAP 384 // (rounds 1-4: 512. 384 entries + tile + queues = 5 KiB per wave = 32 single-wave workgroups per CU; the 237
                         // entries above the depth-first reserve still take 8 runs per batch in every tree measured: same bits)
#endif
constexpr int LK_STACK_CAP = RK_STACK_CAP;
// Worst-case growth of the stack while descending depth-first from one entry: 7 pending entries per level.
constexpr int LK_DFS_RESERVE = 7 * 21;
#ifndef RK_LQ_CAP
#define RK_LQ_CAP 128
#endif
constexpr int LK_LQ_CAP = RK_LQ_CAP;
// Queue of candidates left undecided by the bounding-box / probe tests.
constexpr int LK_UQ_CAP = 128;
// Critical nodes too large for one wavefront (k_list<..., BIG>): wavefronts per workgroup, targets per chunk.
#ifndef RK_BIG_WPB
#define RK_BIG_WPB 4 // 8: 1.9 instead of 3.2 ms when only 512 such nodes exist, 2.86 instead of 2.44 s on the 256M tree of
                     // profiles/r02/big_runs.txt (tools/archive/jobs_r02/r02_job33.sh)
#endif
#ifndef RK_WBIG
#define RK_WBIG 7 // waves per SIMD the BIG kernels are compiled for (5: spill-free, equal or slower)
#endif
constexpr int LK_BIG_WPB = RK_BIG_WPB;
constexpr int LK_BIG_CHUNK = 128;

template <typename F>
struct lk_cfg {
    static constexpr int src_cap = 128; // sources per tile (2 KiB fp32, 4 KiB fp64)
};

// Per-wave LDS: 1.5 + 2 + 1 + 0.5 KiB = 5 KiB (fp32; 7 KiB fp64): 32 single-wave blocks per CU = 8 waves per SIMD.
template <typename F>
struct lk_wave_lds {
    uint32_t stack[LK_STACK_CAP];
    typename vt<F>::v4 src[lk_cfg<F>::src_cap];
    uint2 lq[LK_LQ_CAP];
    uint32_t uq[LK_UQ_CAP]; // records whose MAC test needs the exact all-targets loop
};

// Blocks are dealt round-robin to the 8 XCDs; give each XCD a contiguous slice of the (Morton-ordered)
// group list so that the groups resident on one XCD share most of their nodes in that XCD's L2.
// Bijective for any grid size. Placement only affects speed.
__device__ __forceinline__ unsigned xcd_chunked_block(unsigned b, unsigned nb)
{
    const unsigned q = nb >> 3, r = nb & 7u, xcd = b & 7u, pos = b >> 3;
    return (xcd < r ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q) + pos;
}

// mode 0: XCD x walks chunks x, x + 8, x + 16, ... of XCD_CHUNK consecutive blocks (L2 locality inside a
// chunk, work spread evenly over the XCDs whatever the spatial variation of the group costs);
// mode 1: one contiguous slice per XCD; mode 2: identity (hardware round-robin: block i on XCD i % 8; the light-tail launch
// plan of rk_launch.hip lays its list out for this); mode 3: identity too -- the light-tail arrangement made on the device for
// FIRST calls (rk_build.hip k_tail_sizes), whose per-XCD queues are not interleaved into one padded list but found through a table.
constexpr unsigned XCD_CHUNK = 16;
__device__ __forceinline__ unsigned xcd_map_block(unsigned b, unsigned nb, int mode)
{
    if (mode >= 2) {
        return b; // (3: the class kernels look their node up in the queue table themselves, k_list)
    }
    if (mode == 1) {
        return xcd_chunked_block(b, nb);
    }
    // Blocks beyond the last full round of 8 chunks keep their identity mapping.
    const unsigned span = 8u * XCD_CHUNK, full = nb - nb % span;
    if (b >= full) {
        return b;
    }
    const unsigned xcd = b & 7u, pos = b >> 3;
    return (pos / XCD_CHUNK * 8u + xcd) * XCD_CHUNK + pos % XCD_CHUNK;
}

// One-launch kernels, xcd_mode 4: the list is eight queues (one per XCD region) behind a table of their starts [x] and lengths
// [8 + x] -- the launch order of a first call made on the device with the tree (rk_build.hip bin_classes). Block i serves entry
// i / 8 of queue i % 8; returns false for a block past the end of its queue.
template <typename F>
__device__ __forceinline__ bool any_list_entry(const kparams<F> &P, unsigned blk, int n_list, unsigned &entry)
{
    if (P.xcd_mode == 4) {
        const uint32_t x = blk & 7u, pos = blk >> 3;
        if (pos >= P.first_tab[8u + x]) {
            return false;
        }
        entry = P.first_tab[x] + pos;
        return true;
    }
    entry = P.any_rev ? static_cast<unsigned>(n_list) - 1u - blk : blk;
    return true;
}

template <typename F, int Q, int R, bool SELF, int ND>
__device__ __forceinline__ void lk_interact_src(const typename vt<F>::v4 &s, int j, const typename vt<F>::v4 (&tp)[R],
                                                F (&acc)[R][nres_of(Q)], F eps2, const int (&tidx)[R])
{
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const F ex = s.x - tp[r].x, ey = s.y - tp[r].y, ez = ND == 3 ? s.z - tp[r].z : F(0);
        F e2 = rk_fma(ey, ey, rk_fma(ex, ex, eps2));
        if constexpr (ND == 3) {
            e2 = rk_fma(ez, ez, e2);
        }
        F ms = s.w;
        if constexpr (SELF) {
            const bool self = (j == tidx[r]);
            e2 = self ? F(1) : e2;
            ms = self ? F(0) : ms;
        }
        interact<F, Q, ND>(acc[r], ex, ey, ez, e2, ms, tp[r].w);
    }
}


// ------------------------------------------------------------------------------------------------
// The targets of one lane (R of them) and their accumulators, as the dense phase keeps them in registers for the whole
// critical node: R records and R x NR sums; all indices are compile-time constants after unrolling. (A second flavour with
// pairs of targets in 64-bit register pairs and v_pk_*_f32 arithmetic was built and measured in round 5 -- same bits, no
// gain: tools/experiments/packed_body/.)
// ------------------------------------------------------------------------------------------------
template <typename F, int Q, int R>
struct lk_regs {
    using v4 = typename vt<F>::v4;
    static constexpr int NR = nres_of(Q);
    v4 tp[R];
    F acc[R][NR];
    __device__ __forceinline__ void set_target(int r, const v4 &p)
    {
        tp[r] = p;
#pragma unroll
        for (int k = 0; k < NR; ++k) {
            acc[r][k] = F(0);
        }
    }
    __device__ __forceinline__ F tx(int r) const { return tp[r].x; }
    __device__ __forceinline__ F ty(int r) const { return tp[r].y; }
    __device__ __forceinline__ F tz(int r) const { return tp[r].z; }
    __device__ __forceinline__ F get(int r, int k) const { return acc[r][k]; }
    __device__ __forceinline__ void set(int r, int k, F v) { acc[r][k] = v; }
    template <bool SELF, int ND>
    __device__ __forceinline__ void interact_all(const v4 &s, int j, F eps2, const int (&tidx)[R])
    {
        lk_interact_src<F, Q, R, SELF, ND>(s, j, tp, acc, eps2, tidx);
    }
};


// Dense targets x sources evaluation of one LDS tile. The trip count of the main loop is uniform (full = n_src / ns,
// computed by the caller); the n_src - full * ns sources left over at the end of the tile are one masked step when
// `with_rem` is set, otherwise the caller carries them over to the next tile.
template <typename F, int Q, int R, bool SELF, int ND>
__device__ __forceinline__ void lk_eval_tile(const typename vt<F>::v4 *__restrict__ src, int n_src, int full, int sp,
                                             int ns, bool with_rem, bool lane_on, lk_regs<F, Q, R> &tg, F eps2,
                                             const int (&tidx)[R])
{
    using v4 = typename vt<F>::v4;
#ifdef RK_ABLATE_DENSE
    return; // diagnostic build: list building only
#endif
    // Keep about four interactions in flight per lane whatever R is.
    constexpr int UNR = R >= 4 ? RK_UNR4 : (R == 3 ? RK_UNR3 : (R == 2 ? RK_UNR2 : RK_UNR1));
    static_assert(R <= 6);
#ifdef RK_ABLATE_REM
    const int rem = 0; // diagnostic build: the masked remainder step of every tile is dropped
#else
    const int rem = with_rem ? n_src - full * ns : 0;
#endif
#if RK_CHUNKED_SPLITS
    // Split sp owns the contiguous sources [sp * full, (sp + 1) * full): consecutive iterations read consecutive
    // LDS slots (immediate offsets, no address arithmetic in the loop); the remainder sits at the end of the tile.
    const v4 *p = src + sp * full;
    int j = sp * full;
#if RK_MASK_IDLE
    if (lane_on) // lanes beyond the last split sit the tile out (EXEC off) instead of shadowing split 0
#endif
    {
#pragma unroll UNR
        for (int it = 0; it < full; ++it) {
            const v4 s = p[it];
            tg.template interact_all<SELF, ND>(s, j + it, eps2, tidx);
        }
    }
    if (lane_on && sp < rem) {
        const v4 s = src[ns * full + sp];
        tg.template interact_all<SELF, ND>(s, ns * full + sp, eps2, tidx);
    }
#else
    const v4 *p = src + sp;
    int j = sp;
#pragma unroll UNR
    for (int it = 0; it < full; ++it) {
        const v4 s = *p;
        tg.template interact_all<SELF, ND>(s, j, eps2, tidx);
        p += ns;
        j += ns;
    }
    if (lane_on && sp < rem) {
        const v4 s = *p;
        tg.template interact_all<SELF, ND>(s, j, eps2, tidx);
    }
#endif
}

} // namespace rk

#endif
