// Variant 4 of the traversal ("split"): list building and dense evaluation are two kernels, with the interaction
// lists of the critical nodes in HBM between them.
//
// The fused list kernel (rk_kernels_list.hip) alternates two very different kinds of work inside one wavefront: list
// building is a chain of dependent memory round trips (pop candidates, fetch their records, test, push children) with
// little arithmetic, the dense targets x sources phase is pure vector-ALU streaming. Fused, each waits for the other:
// a launch that does not fill the device ends on the serial chains of its heaviest nodes, and on a full device the
// registers and LDS of the one phase cost the other its occupancy. Here
//
//   * k_lists (one wavefront per critical node) runs the list building of the fused kernel unchanged -- same stack of
//     sibling runs, same box / probe / exact MAC tests, same supergroup pre-pass inputs, hence the reference's decisions
//     (include/rakau/tree.hpp:2662-2672, 2828-2838 of the reference) -- but emits 32-bit SOURCE INDICES instead of
//     sources: the depth-first index of every accepted node, the Morton index of every particle of every opened leaf.
//     They are staged in an LDS ring and written to HBM in whole 512-byte blocks (list segments of SL_SEG entries,
//     the first one at a fixed slot per node, further ones from a bump counter);
//   * k_dense (one wavefront per critical node, the lane mapping of the fused kernel: R targets per lane, NS source
//     splits) streams the supergroup's common sources and then the node's own list through one LDS tile: indices are
//     fetched two tiles ahead, the {x, y, z, m} records they name one tile ahead (gathers from the node / particle
//     arrays, which live in L2 / Infinity Cache), so that the wave only ever waits for LDS.
//
// The order in which a target receives its contributions is a function of the tree and the MAC value alone (list
// order, tiles of 128 sources cut at fixed positions, NS = f(number of targets)), never of the launch: shards and
// full-range calls give the same bits.
#include "rk_list_common.hpp"

#ifndef RK_SLA_W
#define RK_SLA_W 5 // k_lists: waves per SIMD it is compiled for (its 8 KiB of LDS per wave allow no more)
#endif
#ifndef RK_SLD_W12
#define RK_SLD_W12 6 // k_dense, R <= 2 (4 to 8 waves per SIMD measure the same; fewer leave registers for the pipeline)
#endif
#ifndef RK_SLD_W3
#define RK_SLD_W3 5
#endif
#ifndef RK_SLD_W4
#define RK_SLD_W4 4
#endif
#ifndef RK_SLD_W64
#define RK_SLD_W64 4 // fp64
#endif
// k_dense: sources per group of the software-pipelined tile loop (the LDS reads of group i + 1 are issued before the
// arithmetic of group i), per number of targets a lane holds.
#ifndef RK_SLD_EXP
#define RK_SLD_EXP 0 // timing experiments (wrong results): 1 = all splits read the same sources, 2 = tiles are not refilled
#endif
#ifndef RK_SLD_U1
#define RK_SLD_U1 4
#endif
#ifndef RK_SLD_U2
#define RK_SLD_U2 4
#endif
#ifndef RK_SLD_U3
#define RK_SLD_U3 2
#endif
#ifndef RK_SLD_U4
#define RK_SLD_U4 2
#endif

namespace rk
{

constexpr int SL_TILE = 128;

// Per-wave LDS of k_lists: frontier 2 + two staging rings 2 + 2 + leaf queue 1 + exact-test candidates 1 KiB (fp32).
constexpr int SL_FQ_CAP = 512;
template <typename F>
struct sl_lds {
    uint32_t fq[SL_FQ_CAP];   // frontier: runs of sibling nodes, (first record << 3) | (count - 1), first in first out
    uint32_t ring[2][SL_RING]; // staging of the node list [0] and the particle list [1]
    uint2 lq[LK_LQ_CAP];      // opened leaves, in the order they were opened
    typename vt<F>::v4 cand[64]; // {centre of mass, MAC threshold} of the candidates of the exact test
};

// ------------------------------------------------------------------------------------------------
// k_lists: interaction lists of one critical node per wavefront.
//
// A breadth-first walk: the candidates the supergroup pre-pass left to its members first, then the frontier queue
// (children of opened nodes) first in, first out. Every batch of up to 64 candidates is decided completely before the
// next one (box accept, probe open, and for what is left the exact all-targets test with lane = target), so the order in
// which nodes are accepted and leaves are opened is the breadth-first order of the candidates -- whatever the batch
// boundaries, which lets the records of the next batch be requested before the current one is classified. Two lists per
// node: accepted nodes (depth-first indices) and the particles of opened leaves, each in that order.
// The decisions are those of the reference (include/rakau/tree.hpp:2662-2672, 2828-2838 of the reference).
// ------------------------------------------------------------------------------------------------
template <typename F, int MAC>
__global__ void __launch_bounds__(64, RK_SLA_W) k_lists(const kparams<F> P, uint32_t g_begin, uint32_t g_end)
{
    using v4 = typename vt<F>::v4;
    using v2 = typename vt<F>::v2;
    __shared__ sl_lds<F> L;
    const int lane = threadIdx.x;
    // A contiguous slice of the (Morton-ordered) nodes per XCD: neighbours share tree nodes in that XCD's L2.
    const uint32_t g = __builtin_amdgcn_readfirstlane(g_begin + xcd_chunked_block(blockIdx.x, gridDim.x));
    if (g >= g_end) {
        return;
    }
    const uint4 c = P.crit[g];
    const uint32_t gb = c.x, ge = c.y, cnode = c.z;
    const int TG = static_cast<int>(ge - gb);
    if (TG > 64 * RK_MAX_R) {
        return; // too large for one wavefront: served by k_list<..., BIG>
    }
    // Targets for the lane = target exact test: lane l keeps targets l, l + 64, ... (the last one repeated).
    const int RT = (TG + 63) >> 6;
    v4 tp[RK_MAX_R];
#pragma unroll
    for (int r = 0; r < RK_MAX_R; ++r) {
        const int ti = lane + 64 * r;
        tp[r] = P.part4[gb + static_cast<uint32_t>(ti < TG ? ti : TG - 1)];
    }
    const F mac_value = P.mac_value;
    const v4 blo = P.crit_box[2u * g], bhi = P.crit_box[2u * g + 1u];
    const v4 pr0 = P.part4[gb], pr1 = P.part4[ge - 1u];

    RK_STAMP_DECL
    uint32_t fq_head = 0, fq_tail = 0; // frontier entries popped / pushed
    int n_lq = 0;
    // The two lists: entries written to HBM / appended to the ring, current segment.
    uint32_t head[2] = {0u, 0u}, tail[2] = {0u, 0u};
    uint32_t cur_seg[2] = {2u * (g - P.sl_g0), 2u * (g - P.sl_g0) + 1u};
    bool over = false;

    uint32_t sup_S = 0, sup_nresid = 0;
    bool from_root = true;
    if (P.super_k != 0u) {
        sup_S = g / P.super_k;
        const uint2 cnt = P.sup_cnt[sup_S];
        if ((cnt.y >> 31) == 0u) {
            from_root = false;
            sup_nresid = cnt.y;
        }
    }
    if (from_root) {
        const node_rec<F> *root = P.node_rec;
        const uint32_t r_nch = root->nch, r_a = root->a, r_b = root->b;
        if (cnode != 0u && r_nch != 0u) {
            if (lane == 0) {
                L.fq[0] = (r_a << 3) | (r_b - 1u);
            }
            fq_tail = 1;
        }
    }
    wave_sync();

    // Write the whole 128-entry blocks of ring w (everything when `final`) to list w of the node.
    auto flush_blocks = [&](int w, bool final) __attribute__((always_inline)) {
        while (!over && (tail[w] - head[w] >= static_cast<uint32_t>(SL_TILE) || (final && tail[w] > head[w]))) {
            const uint32_t n = tail[w] - head[w] < static_cast<uint32_t>(SL_TILE) ? tail[w] - head[w] : static_cast<uint32_t>(SL_TILE);
            if ((head[w] & (SL_SEG - 1u)) == 0u && head[w] != 0u) {
                // The current segment is full: take the next one from the pool and link it.
                if (head[w] >= P.sl_max_len) {
                    over = true; // a property of the node and the MAC value: the same in every launch
                    break;
                }
                uint32_t ns = 0u;
                if (lane == 0) {
                    ns = P.sl_nslot + __hip_atomic_fetch_add(&P.sl_ctl[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                ns = __builtin_amdgcn_readfirstlane(ns);
                if (ns >= P.sl_nseg) {
                    over = true; // pool exhausted (reported to the host, which grows the pool for the next call)
                    if (lane == 0) {
                        __hip_atomic_fetch_add(&P.sl_ctl[3], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    break;
                }
                if (lane == 0) {
                    P.sl_next[cur_seg[w]] = ns;
                }
                cur_seg[w] = ns;
            }
            uint32_t *dst = P.sl_idx + static_cast<size_t>(cur_seg[w]) * SL_SEG + (head[w] & (SL_SEG - 1u));
#pragma unroll
            for (uint32_t i = 0; i < 2u; ++i) {
                const uint32_t j = static_cast<uint32_t>(lane) + 64u * i;
                if (j < n) {
                    dst[j] = L.ring[w][(head[w] + j) & (SL_RING - 1u)];
                }
            }
            head[w] += n;
        }
    };

    // Expand the queued leaves into particle indices (list 1), oldest first.
    auto drain_leaves = [&]() __attribute__((always_inline)) {
        while (n_lq > 0 && !over) {
            flush_blocks(1, false); // fewer than 128 entries stay in the ring
            if (over) {
                break;
            }
            const uint32_t room = SL_RING - (tail[1] - head[1]);
            uint2 lf = make_uint2(0u, 0u);
            if (lane < n_lq) {
                lf = L.lq[lane];
            }
            const unsigned cnt = lane < n_lq ? lf.y - lf.x : 0u;
            const unsigned incl = wave_incl_scan(cnt);
            const bool fits = lane < n_lq && incl <= room;
            const int m = __builtin_popcountll(__builtin_amdgcn_ballot_w64(fits)); // leaves [0, m) fit (prefix property)
            if (m == 0) {
                // The first leaf is larger than the free part of the ring: take `room` of its particles.
                const uint32_t b0 = __builtin_amdgcn_readfirstlane(lf.x);
                for (uint32_t j = static_cast<uint32_t>(lane); j < room; j += 64u) {
                    L.ring[1][(tail[1] + j) & (SL_RING - 1u)] = b0 + j;
                }
                if (lane == 0) {
                    L.lq[0] = make_uint2(b0 + room, lf.y);
                }
                tail[1] += room;
                wave_sync();
                continue;
            }
            const unsigned mycnt = fits ? cnt : 0u;
            const uint32_t dst = tail[1] + (incl - cnt);
            for (unsigned j = 0; __builtin_amdgcn_ballot_w64(j < mycnt) != 0ull; ++j) {
                if (j < mycnt) {
                    L.ring[1][(dst + j) & (SL_RING - 1u)] = lf.x + j;
                }
            }
            tail[1] += static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(incl), m - 1));
            const int rest = n_lq - m;
            wave_sync();
            for (int j0 = 0; j0 < rest; j0 += 64) {
                const int j = j0 + lane;
                uint2 mv = make_uint2(0u, 0u);
                if (j < rest) {
                    mv = L.lq[j + m];
                }
                wave_sync();
                if (j < rest) {
                    L.lq[j] = mv;
                }
                wave_sync();
            }
            n_lq = rest;
        }
    };

    struct batch_t {
        bool active;
        v4 com;
        v2 mp;
        uint32_t node, nch, ra, rb, rec;
    };
    auto load_rec = [&](batch_t &bt) __attribute__((always_inline)) {
        const node_rec<F> *rec = P.node_rec + bt.rec;
        bt.com = rec->com;
        bt.mp = rec->mac;
        bt.node = rec->dfs;
        bt.nch = rec->nch;
        bt.ra = rec->a;
        bt.rb = rec->b;
    };
    // Next batch of the frontier (requests the records; false: nothing to take). `pending` = runs that batches already
    // taken may still push. Normally the OLDEST 8 runs; when the queue is about to overflow (tiny opening angles), the
    // newest ones, fewer at a time: depth first, whose growth LK_DFS_RESERVE bounds.
    auto pop_and_load = [&](batch_t &bt, int pending) __attribute__((always_inline)) -> bool {
        const int size = static_cast<int>(fq_tail - fq_head);
        if (size == 0) {
            return false;
        }
        int k = size < 8 ? size : 8;
        const int room = (SL_FQ_CAP - LK_DFS_RESERVE - pending - size) / 7;
        const int e_idx = lane >> 3, e_sub = lane & 7;
        uint32_t entry = 0u;
        if (room >= k) {
            if (e_idx < k) {
                entry = L.fq[(fq_head + static_cast<uint32_t>(e_idx)) & (SL_FQ_CAP - 1)];
            }
            fq_head += static_cast<uint32_t>(k);
        } else {
            if (pending != 0) {
                return false; // settle what is in flight first
            }
            k = room >= 1 ? room : 1;
            if (e_idx < k) {
                entry = L.fq[(fq_tail - 1u - static_cast<uint32_t>(e_idx)) & (SL_FQ_CAP - 1)];
            }
            fq_tail -= static_cast<uint32_t>(k);
        }
        bt.active = e_idx < k && static_cast<uint32_t>(e_sub) <= (entry & 7u);
        bt.rec = bt.active ? (entry >> 3) + static_cast<uint32_t>(e_sub) : 0u;
        load_rec(bt);
        return true;
    };
    uint32_t sup_rpos = 0;
    auto resid_load = [&](batch_t &bt, int pending) __attribute__((always_inline)) -> bool {
        if (sup_rpos >= sup_nresid) {
            return false;
        }
        // Every candidate may push a run: leave the frontier room for that (otherwise take from the frontier first).
        if (SL_FQ_CAP - LK_DFS_RESERVE - pending - static_cast<int>(fq_tail - fq_head) < 64) {
            return false;
        }
        const uint32_t left = sup_nresid - sup_rpos, k = left < 64u ? left : 64u;
        bt.active = static_cast<uint32_t>(lane) < k;
        bt.rec = bt.active ? P.sup_resid[static_cast<size_t>(sup_S) * SUP_CAPR + sup_rpos + static_cast<uint32_t>(lane)] : 0u;
        sup_rpos += k;
        load_rec(bt);
        return true;
    };

    // Decide a batch completely and route it: accepted nodes to list 0, opened leaves to the leaf queue, the children of
    // opened internal nodes to the back of the frontier.
    auto process = [&](const batch_t &bt) __attribute__((always_inline)) {
        const v4 com = bt.com;
        // Ancestor-or-self of the target node, on the depth-first index interval of the subtree.
        const bool anc = bt.active && bt.node <= cnode && cnode <= bt.node + bt.nch;
        const bool self = anc && bt.node == cnode;
        const bool test = bt.active && !anc;
        const F mac_lh = mac_lhs<F>(MAC, bt.mp, mac_value);
#ifdef RK_STAMPS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        st_acc[6] += 1;
#endif
        RK_STAMP(0)
        // Accept if the squared distance from the centre of mass to the node's bounding box exceeds the threshold by a
        // margin (then every target passes); open if one of two probe targets fails; the rest: exact test.
        const F bx = rk_max3(blo.x - com.x, com.x - bhi.x, F(0)), by = rk_max3(blo.y - com.y, com.y - bhi.y, F(0)),
                bz = rk_max3(blo.z - com.z, com.z - bhi.z, F(0));
        const F dbox2 = rk_fma(bz, bz, rk_fma(by, by, bx * bx));
        const bool box_accept = dbox2 > mac_lh * F(1.00001);
        const F p0x = com.x - pr0.x, p0y = com.y - pr0.y, p0z = com.z - pr0.z;
        const F p1x = com.x - pr1.x, p1y = com.y - pr1.y, p1z = com.z - pr1.z;
        const F d2p0 = rk_fma(p0z, p0z, rk_fma(p0y, p0y, p0x * p0x)), d2p1 = rk_fma(p1z, p1z, rk_fma(p1y, p1y, p1x * p1x));
        const bool probe_open = mac_lh >= rk_min(d2p0, d2p1);
        const bool undecided = test && !box_accept && !probe_open;
        RK_STAMP(1)
        bool fail = false; // exact test outcome of the undecided candidates
        const unsigned long long m_und = __builtin_amdgcn_ballot_w64(undecided);
        if (m_und != 0ull) {
            const int k = __builtin_popcountll(m_und);
            if (k * (7 * RT + 3) < TG * 7) {
                // lane = target: the candidates go through LDS, two per step.
                if (undecided) {
                    v4 cd;
                    cd.x = com.x, cd.y = com.y, cd.z = com.z, cd.w = mac_lh;
                    L.cand[wave_prefix_count(m_und)] = cd;
                }
                wave_sync();
                unsigned long long fail_mask = 0ull;
                for (int ci = 0; ci < k; ci += 2) {
                    const v4 c0 = L.cand[ci], c1 = L.cand[ci + 1 < k ? ci + 1 : ci];
                    bool f0 = false, f1 = false;
#pragma unroll
                    for (int r = 0; r < RK_MAX_R; ++r) {
                        if (r < RT) {
                            const F ax = c0.x - tp[r].x, ay = c0.y - tp[r].y, az = c0.z - tp[r].z;
                            f0 |= c0.w >= rk_fma(az, az, rk_fma(ay, ay, ax * ax));
                            const F bx2 = c1.x - tp[r].x, by2 = c1.y - tp[r].y, bz2 = c1.z - tp[r].z;
                            f1 |= c1.w >= rk_fma(bz2, bz2, rk_fma(by2, by2, bx2 * bx2));
                        }
                    }
                    if (__builtin_amdgcn_ballot_w64(f0) != 0ull) {
                        fail_mask |= 1ull << ci;
                    }
                    if (__builtin_amdgcn_ballot_w64(f1) != 0ull) {
                        fail_mask |= 1ull << (ci + 1 < k ? ci + 1 : ci);
                    }
                }
                fail = undecided && ((fail_mask >> wave_prefix_count(m_und)) & 1ull) != 0ull;
                wave_sync();
            } else {
                // lane = candidate; the targets arrive through scalar loads as SGPR operands.
                F mind2 = std::numeric_limits<F>::infinity();
                for (int t = 0; t < TG; t += 4) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int ti = (t + u < TG) ? t + u : TG - 1;
                        const v4 tg = P.part4[gb + static_cast<uint32_t>(ti)];
                        const F dx = com.x - tg.x, dy = com.y - tg.y, dz = com.z - tg.z;
                        mind2 = rk_min(mind2, rk_fma(dz, dz, rk_fma(dy, dy, dx * dx)));
                    }
                }
                fail = undecided && mac_lh >= mind2;
            }
        }
        RK_STAMP(3)
        const bool accept = test && (box_accept || (undecided && !fail));
        const bool open = (test && !box_accept && (probe_open || fail)) || (anc && !self);
        const bool leaf = open && bt.nch == 0u;
        const bool expand = open && bt.nch != 0u;
        const unsigned long long m_acc = __builtin_amdgcn_ballot_w64(accept);
        if (accept) {
            L.ring[0][(tail[0] + wave_prefix_count(m_acc)) & (SL_RING - 1u)] = bt.node;
        }
        tail[0] += static_cast<uint32_t>(__builtin_popcountll(m_acc));
        const unsigned long long m_leaf = __builtin_amdgcn_ballot_w64(leaf);
        if (leaf) {
            L.lq[n_lq + static_cast<int>(wave_prefix_count(m_leaf))] = make_uint2(bt.ra, bt.rb);
        }
        n_lq += __builtin_popcountll(m_leaf);
        const unsigned long long m_exp = __builtin_amdgcn_ballot_w64(expand);
        if (expand) {
            L.fq[(fq_tail + wave_prefix_count(m_exp)) & (SL_FQ_CAP - 1)] = (bt.ra << 3) | (bt.rb - 1u);
        }
        fq_tail += static_cast<uint32_t>(__builtin_popcountll(m_exp));
        wave_sync();
        RK_STAMP(2)
    };
    // Room for the worst-case output of one batch (64 nodes, 64 leaves).
    auto settle = [&]() __attribute__((always_inline)) {
        RK_STAMP(7)
        if (n_lq + 64 > LK_LQ_CAP) {
            drain_leaves();
        }
        RK_STAMP(4)
        flush_blocks(0, false);
        RK_STAMP(5)
    };

    // Two batches in flight: the records of the next one are requested before the current one is classified.
    batch_t A, B;
    bool haveA = resid_load(A, 0);
    if (!haveA) {
        haveA = pop_and_load(A, 0);
    }
    while (haveA && !over) {
        bool haveB = resid_load(B, 64);
        if (!haveB) {
            haveB = pop_and_load(B, 64);
        }
        process(A);
        settle();
        if (over) {
            break;
        }
        if (!haveB) {
            haveB = resid_load(B, 0) || pop_and_load(B, 0); // A may have pushed what B can take now
        }
        if (!haveB) {
            break;
        }
        bool haveA2 = resid_load(A, 64);
        if (!haveA2) {
            haveA2 = pop_and_load(A, 64);
        }
        process(B);
        settle();
        if (!haveA2) {
            haveA2 = resid_load(A, 0) || pop_and_load(A, 0);
        }
        haveA = haveA2;
    }
    RK_STAMP(7)
    drain_leaves();
    RK_STAMP(4)
    flush_blocks(0, true);
    flush_blocks(1, true);
    RK_STAMP(5)

    if (!over && P.sl_parts_mode) {
        // The call evaluates one part per wavefront: reserve the node's partial-sum slots.
        uint32_t n1 = 0u;
        if (!from_root) {
            n1 = P.sup_cnt[sup_S].x;
        }
        const uint32_t ntt = (n1 + SL_TILE - 1) / SL_TILE + (tail[0] + SL_TILE - 1) / SL_TILE + (tail[1] + SL_TILE - 1) / SL_TILE
                             + (static_cast<uint32_t>(TG) + SL_TILE - 1) / SL_TILE;
        const uint32_t n_parts = (ntt + 3u) / 4u;
        uint32_t pb = 0u;
        if (lane == 0) {
            pb = __hip_atomic_fetch_add(&P.sl_ctl[4], n_parts, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        pb = __builtin_amdgcn_readfirstlane(pb);
        if (pb + n_parts > P.sl_npart) {
            over = true;
            if (lane == 0) {
                __hip_atomic_fetch_add(&P.sl_ctl[3], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        } else if (lane == 0) {
            P.sl_pbase[g] = pb;
        }
    }
    if (lane == 0) {
        if (over) {
            P.sl_cnt[2u * g] = SL_OVER;
            const uint32_t slot = __hip_atomic_fetch_add(&P.sl_ctl[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            P.sl_fb[slot] = g;
        } else {
            P.sl_cnt[2u * g] = tail[0];
            P.sl_cnt[2u * g + 1u] = tail[1];
            __hip_atomic_fetch_add(&P.sl_ctl[2], tail[0] + tail[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    RK_STAMP(7)
    RK_STAMP_FLUSH
}

// ------------------------------------------------------------------------------------------------
// k_dense: the interactions of one critical node, sources streamed from the lists.
//
// The sources of a node form a sequence of TILES of (at most) 128: the supergroup's common sources, the node's own
// lists (nodes, then particles), the node's own particles (self pair masked). Four consecutive tiles are a PART. Every
// lane (target slot x source split, the fused kernel's mapping) accumulates a part from zero, and adds the parts up in
// order; the splits are summed in order at the very end. That order is a function of the node and the MAC value only,
// and it can be walked by one wavefront (PARTS = false: all parts one after the other, the sums in registers) or by one
// wavefront per part (PARTS = true: calls over few critical nodes, whose heaviest nodes would otherwise keep single
// wavefronts busy long after the rest of the device has drained; the per-lane sums of each part go to a scratch array
// and k_combine adds them up in the same order). Same bits either way.
// ------------------------------------------------------------------------------------------------
constexpr int SL_PART_TILES = 4;
constexpr int SL_PARTS_PER_NODE = 4; // PARTS launches: wavefronts per node (part k, k + 4, ... each)
constexpr int SL_SLOT = 64 * 4 * 4;  // values per partial-sum slot: 64 lanes x at most 4 targets x 4 results

// Dense targets x sources evaluation of one LDS tile for k_dense: the arithmetic and the assignment of sources to splits
// of lk_eval_tile() (split sp owns the contiguous sources [sp * full, (sp + 1) * full), the n_src - ns * full left over
// are one masked step), with the LDS reads software-pipelined: the sources of group i + 1 are requested before group i
// is evaluated, so that a wave never waits for LDS with nothing to issue.
template <typename F, int Q, int R, bool SELF, int ND>
__device__ __forceinline__ void sl_eval_tile(const typename vt<F>::v4 *__restrict__ src, int n_src, int full, int sp, int ns,
                                             bool lane_on, const typename vt<F>::v4 (&tp)[R], F (&acc)[R][nres_of(Q)], F eps2,
                                             const int (&tidx)[R])
{
    using v4 = typename vt<F>::v4;
    constexpr int U = R >= 4 ? RK_SLD_U4 : (R == 3 ? RK_SLD_U3 : (R == 2 ? RK_SLD_U2 : RK_SLD_U1));
    const int rem = n_src - full * ns;
#if RK_SLD_EXP == 1
    const v4 *p = src; // timing experiment: every split reads the same sources (pure broadcast, no bank conflicts)
#else
    const v4 *p = src + sp * full;
#endif
    const int j0 = sp * full;
    const int ng = full / U;
    if (ng > 0) {
        // Two register sets A / B, no copies: B is requested before A is evaluated and the other way round. The scheduling
        // barriers keep the compiler from moving the requests down to their first use.
        v4 a[U], b[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            a[u] = p[u];
        }
        int g = 0;
        for (; g + 1 < ng; g += 2) {
            const v4 *qb = p + (g + 1) * U;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                b[u] = qb[u];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                lk_interact_src<F, Q, R, SELF, ND>(a[u], j0 + g * U + u, tp, acc, eps2, tidx);
            }
            __builtin_amdgcn_sched_barrier(0);
            // (the last pair re-reads group g + 1: in bounds, the values are dropped)
            const v4 *qa = p + (g + 2 < ng ? g + 2 : g + 1) * U;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                a[u] = qa[u];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                lk_interact_src<F, Q, R, SELF, ND>(b[u], j0 + (g + 1) * U + u, tp, acc, eps2, tidx);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (g < ng) {
            // An odd number of groups: the last one is in A.
#pragma unroll
            for (int u = 0; u < U; ++u) {
                lk_interact_src<F, Q, R, SELF, ND>(a[u], j0 + g * U + u, tp, acc, eps2, tidx);
            }
        }
    }
    for (int it = ng * U; it < full; ++it) {
        const v4 s = p[it];
        lk_interact_src<F, Q, R, SELF, ND>(s, j0 + it, tp, acc, eps2, tidx);
    }
    if (lane_on && sp < rem) {
        const v4 s = src[ns * full + sp];
        lk_interact_src<F, Q, R, SELF, ND>(s, ns * full + sp, tp, acc, eps2, tidx);
    }
}

// Geometry of a node's work, the same in k_dense and k_combine.
template <typename F>
struct sl_node_geom {
    uint32_t tb, te, n1, n2, n3;
    int T, nt1, nt2, nt3, nts, ntt, n_parts;
    const typename vt<F>::v4 *common;
};
template <typename F>
__device__ __forceinline__ bool sl_geom(const kparams<F> &P, uint32_t g, sl_node_geom<F> &o)
{
    const uint32_t n2w = __builtin_amdgcn_readfirstlane(P.sl_cnt[2u * g]);
    if (n2w & SL_OVER) {
        return false; // the lists were not completed: the fused kernel serves this node
    }
    const uint4 c = P.crit[g];
    o.tb = c.x, o.te = c.y;
    o.T = static_cast<int>(o.te - o.tb);
    o.n1 = 0u;
    o.common = nullptr;
    if (P.super_k != 0u) {
        const uint32_t S = g / P.super_k;
        const uint2 cnt = P.sup_cnt[S];
        if ((cnt.y >> 31) == 0u) {
            o.n1 = cnt.x;
            o.common = P.sup_common + static_cast<size_t>(S) * SUP_CAPC;
        }
    }
    o.n2 = n2w;
    o.n3 = __builtin_amdgcn_readfirstlane(P.sl_cnt[2u * g + 1u]);
    o.nt1 = static_cast<int>((o.n1 + SL_TILE - 1) / SL_TILE);
    o.nt2 = static_cast<int>((o.n2 + SL_TILE - 1) / SL_TILE);
    o.nt3 = static_cast<int>((o.n3 + SL_TILE - 1) / SL_TILE);
    o.nts = (o.T + SL_TILE - 1) / SL_TILE;
    o.ntt = o.nt1 + o.nt2 + o.nt3 + o.nts;
    o.n_parts = (o.ntt + SL_PART_TILES - 1) / SL_PART_TILES;
    return true;
}

// Sum of the source splits in a fixed order, times G, store. `acc` holds every lane's total.
template <typename F, int Q, int R, int ND>
__device__ __forceinline__ void sl_epilogue(const kparams<F> &P, typename vt<F>::v4 *tile, F (&acc)[R][nres_of(Q)], uint32_t tb,
                                            const int (&tidx)[R], int TP, int NS, int ts, int sp_raw, bool lane_on)
{
    constexpr int NR = nres_of(Q);
    const F G = P.G;
    if (NS > 1) {
        F *red = reinterpret_cast<F *>(tile);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if (lane_on) {
#pragma unroll
                for (int k = 0; k < NR; ++k) {
                    red[(sp_raw * TP + ts) * NR + k] = acc[r][k];
                }
            }
            wave_sync();
            if (lane_on && sp_raw == 0) {
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
    if (lane_on && sp_raw == 0) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if (tidx[r] >= 0) {
                const uint32_t o = out_index(P, tb + static_cast<uint32_t>(tidx[r]));
#pragma unroll
                for (int k = 0; k < NR; ++k) {
                    if (ND == 3 || Q == 1 || k != 2) { // a quadtree has no z acceleration (and no array for it)
                        P.out[k][o] = acc[r][k] * G;
                    }
                }
            }
        }
    }
}

template <typename F, int Q, int R, int ND, bool PARTS>
__global__ void __launch_bounds__(64, sizeof(F) == 4 ? (R <= 2 ? RK_SLD_W12 : (R == 3 ? RK_SLD_W3 : RK_SLD_W4)) : RK_SLD_W64)
    k_dense(const kparams<F> P, const uint32_t *__restrict__ list, int n_list)
{
    using v4 = typename vt<F>::v4;
    constexpr int NR = nres_of(Q);
    static_assert(SL_TILE * 4 >= 64 * 4, "reduction scratch does not fit the tile");
    __shared__ v4 tile[SL_TILE];
    const int lane = threadIdx.x;
    const unsigned blk = PARTS ? blockIdx.x / unsigned(SL_PARTS_PER_NODE) : xcd_map_block(blockIdx.x, gridDim.x, P.xcd_mode);
    if (static_cast<int>(blk) >= n_list) {
        return;
    }
    const uint32_t g = __builtin_amdgcn_readfirstlane(list[blk]);
    if (g == RK_PLAN_PAD_VALUE) {
        return;
    }
    sl_node_geom<F> N;
    if (!sl_geom(P, g, N)) {
        return;
    }
#ifdef RK_TRACE
    const unsigned long long tr_t0 = __builtin_amdgcn_s_memrealtime(), tr_c0 = __builtin_amdgcn_s_memtime();
#endif
    const int first_part = PARTS ? static_cast<int>(blockIdx.x % unsigned(SL_PARTS_PER_NODE)) : 0;
    if (PARTS && first_part >= N.n_parts) {
        return;
    }
    const uint32_t tb = N.tb;
    const int T = N.T;
    const int TP = (T + R - 1) / R;
    const int NS = 64 / TP;
    const int ts = lane % TP, sp_raw = lane / TP;
    const bool lane_on = sp_raw < NS;
    const int sp = lane_on ? sp_raw : 0;
    v4 tp[R];
    int tidx[R];
    F acc[R][NR], tot[R][NR];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        tidx[r] = ts + r * TP;
        const bool valid = tidx[r] < T;
        tp[r] = P.part4[tb + (valid ? tidx[r] : 0)];
        if (!valid) {
            tidx[r] = -1;
        }
#pragma unroll
        for (int k = 0; k < NR; ++k) {
            acc[r][k] = F(0);
            tot[r][k] = F(0);
        }
    }
    const F eps2 = P.eps2;
    const int inv_ns = (65536 + NS - 1) / NS; // n / NS for n <= 128 without a division per tile
    static_assert(SL_TILE * 64 < 65536);
    const uint32_t n1 = N.n1, n2 = N.n2, n3 = N.n3;
    const int nt1 = N.nt1, nt12 = N.nt1 + N.nt2, nt123 = nt12 + N.nt3, ntt = N.ntt;
    const v4 *common = N.common;
    constexpr uint32_t TPS = SL_SEG / SL_TILE; // tiles per segment

    // Segment of the list tile whose indices are fetched next (tiles are visited in increasing order).
    int seg_list = -1;
    uint32_t seg_no = 0, seg_id = 0;
    uint32_t ix[2] = {0u, 0u};
    v4 pv[2];
    // Indices of tile u (list tiles only: the node list, then the particle list).
    auto stage1 = [&](int u) __attribute__((always_inline)) {
        if (u >= nt1 && u < nt123) {
            const int w = u < nt12 ? 0 : 1;
            const uint32_t t2 = static_cast<uint32_t>(u - (w ? nt12 : nt1)), nw = w ? n3 : n2;
            if (seg_list != w) {
                seg_list = w;
                seg_no = 0;
                seg_id = 2u * (g - P.sl_g0) + static_cast<uint32_t>(w);
            }
            while (seg_no < t2 / TPS) {
                seg_id = __builtin_amdgcn_readfirstlane(P.sl_next[seg_id]);
                ++seg_no;
            }
            const uint32_t *src = P.sl_idx + static_cast<size_t>(seg_id) * SL_SEG + (t2 % TPS) * SL_TILE;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const uint32_t j = static_cast<uint32_t>(lane) + 64u * i;
                ix[i] = t2 * SL_TILE + j < nw ? src[j] : 0u;
            }
        }
    };
    // Records of tile u.
    auto stage2 = [&](int u) __attribute__((always_inline)) {
        if (u < nt1) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const uint32_t j = static_cast<uint32_t>(u) * SL_TILE + static_cast<uint32_t>(lane) + 64u * i;
                pv[i] = common[j < n1 ? j : 0u];
            }
        } else if (u < nt123) {
            const v4 *base = u < nt12 ? P.node_com : P.part4;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                pv[i] = base[ix[i]];
            }
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int j = (u - nt123) * SL_TILE + lane + 64 * i;
                pv[i] = P.part4[tb + static_cast<uint32_t>(j < T ? j : 0)];
            }
        }
    };
    // The tiles [u0, u1), software-pipelined: indices two tiles ahead, records one tile ahead.
    auto run_tiles = [&](int u0, int u1) __attribute__((always_inline)) {
        stage1(u0);
        stage2(u0);
        if (u0 + 1 < u1) {
            stage1(u0 + 1);
        }
        for (int u = u0; u < u1; ++u) {
#if RK_SLD_EXP == 2
            if (u == u0) {
                tile[lane] = pv[0];
                tile[lane + 64] = pv[1];
            }
            wave_sync();
#else
            tile[lane] = pv[0];
            tile[lane + 64] = pv[1];
            wave_sync();
            if (u + 1 < u1) {
                stage2(u + 1);
            }
            if (u + 2 < u1) {
                stage1(u + 2);
            }
#endif
            if (u < nt123) {
                const uint32_t left = u < nt1 ? n1 - static_cast<uint32_t>(u) * SL_TILE
                                              : (u < nt12 ? n2 - static_cast<uint32_t>(u - nt1) * SL_TILE
                                                          : n3 - static_cast<uint32_t>(u - nt12) * SL_TILE);
                const int n = left < static_cast<uint32_t>(SL_TILE) ? static_cast<int>(left) : SL_TILE;
                sl_eval_tile<F, Q, R, false, ND>(tile, n, (n * inv_ns) >> 16, sp, NS, lane_on, tp, acc, eps2, tidx);
            } else {
                // The node's own particles: the self pair is masked.
                const int b0 = (u - nt123) * SL_TILE;
                const int n = (T - b0) < SL_TILE ? (T - b0) : SL_TILE;
                int tloc[R];
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    tloc[r] = tidx[r] < 0 ? -1 : tidx[r] - b0;
                }
                sl_eval_tile<F, Q, R, true, ND>(tile, n, (n * inv_ns) >> 16, sp, NS, lane_on, tp, acc, eps2, tloc);
            }
            wave_sync();
            if (!PARTS && ((u + 1) % SL_PART_TILES == 0 || u + 1 == u1)) {
                // End of a part: add it to the lane's total.
#pragma unroll
                for (int r = 0; r < R; ++r) {
#pragma unroll
                    for (int k = 0; k < NR; ++k) {
                        tot[r][k] += acc[r][k];
                        acc[r][k] = F(0);
                    }
                }
            }
        }
    };
    if constexpr (!PARTS) {
        run_tiles(0, ntt);
        sl_epilogue<F, Q, R, ND>(P, tile, tot, tb, tidx, TP, NS, ts, sp_raw, lane_on);
#ifdef RK_TRACE
        if (lane == 0 && P.dbg) {
            const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20);
            P.dbg[4u * g] = tr_t0;
            P.dbg[4u * g + 1u] = __builtin_amdgcn_s_memrealtime();
            P.dbg[4u * g + 2u] = (static_cast<unsigned long long>(xcc) << 32) | hw;
            // {R, T, shader cycles of this wave}: cycles / duration = the clock the chip held meanwhile.
            P.dbg[4u * g + 3u] = (static_cast<unsigned long long>(R) << 56) | (static_cast<unsigned long long>(T) << 32)
                                 | ((__builtin_amdgcn_s_memtime() - tr_c0) & 0xffffffffull);
        }
#endif
    } else {
        // Parts first_part, first_part + SL_PARTS_PER_NODE, ...: per-lane sums to the node's partial-sum slots.
        F *slots = static_cast<F *>(P.sl_part) + static_cast<size_t>(P.sl_pbase[g]) * SL_SLOT;
        for (int part = first_part; part < N.n_parts; part += SL_PARTS_PER_NODE) {
            const int u0 = part * SL_PART_TILES, u1 = u0 + SL_PART_TILES < ntt ? u0 + SL_PART_TILES : ntt;
            run_tiles(u0, u1);
            F *dst = slots + static_cast<size_t>(part) * SL_SLOT;
#pragma unroll
            for (int r = 0; r < R; ++r) {
#pragma unroll
                for (int k = 0; k < NR; ++k) {
                    dst[(r * NR + k) * 64 + lane] = acc[r][k];
                    acc[r][k] = F(0);
                }
            }
        }
    }
}

// PARTS launches: the per-lane partial sums of a node's parts, added up in order (what one k_dense<..., false> wavefront
// does in registers), then the common epilogue.
template <typename F, int Q, int R, int ND>
__global__ void __launch_bounds__(64) k_combine(const kparams<F> P, const uint32_t *__restrict__ list, int n_list)
{
    using v4 = typename vt<F>::v4;
    constexpr int NR = nres_of(Q);
    __shared__ v4 tile[SL_TILE];
    const int lane = threadIdx.x;
    if (static_cast<int>(blockIdx.x) >= n_list) {
        return;
    }
    const uint32_t g = __builtin_amdgcn_readfirstlane(list[blockIdx.x]);
    if (g == RK_PLAN_PAD_VALUE) {
        return;
    }
    sl_node_geom<F> N;
    if (!sl_geom(P, g, N)) {
        return;
    }
    const int T = N.T;
    const int TP = (T + R - 1) / R;
    const int NS = 64 / TP;
    const int ts = lane % TP, sp_raw = lane / TP;
    const bool lane_on = sp_raw < NS;
    int tidx[R];
    F tot[R][NR];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        tidx[r] = ts + r * TP < T ? ts + r * TP : -1;
#pragma unroll
        for (int k = 0; k < NR; ++k) {
            tot[r][k] = F(0);
        }
    }
    const F *slots = static_cast<const F *>(P.sl_part) + static_cast<size_t>(P.sl_pbase[g]) * SL_SLOT;
    for (int part = 0; part < N.n_parts; ++part) {
        const F *src = slots + static_cast<size_t>(part) * SL_SLOT;
#pragma unroll
        for (int r = 0; r < R; ++r) {
#pragma unroll
            for (int k = 0; k < NR; ++k) {
                tot[r][k] += src[(r * NR + k) * 64 + lane];
            }
        }
    }
    sl_epilogue<F, Q, R, ND>(P, tile, tot, N.tb, tidx, TP, NS, ts, sp_raw, lane_on);
}

// ------------------------------------------------------------------------------------------------
// Launch.
// ------------------------------------------------------------------------------------------------
template <typename F>
void launch_lists(const rk_state &s, const kparams<F> &p, int64_t g_begin, int64_t g_end, hipStream_t stream)
{
    const int64_t n = g_end - g_begin;
    if (n <= 0) {
        return;
    }
    const dim3 grid(static_cast<unsigned>(n)), block(64);
    if (s.mac == RK_MAC_BH) {
        hipLaunchKernelGGL((k_lists<F, 0>), grid, block, 0, stream, p, static_cast<uint32_t>(g_begin),
                           static_cast<uint32_t>(g_end));
    } else {
        hipLaunchKernelGGL((k_lists<F, 1>), grid, block, 0, stream, p, static_cast<uint32_t>(g_begin),
                           static_cast<uint32_t>(g_end));
    }
    RK_HIP(hipGetLastError());
}
template void launch_lists<float>(const rk_state &, const kparams<float> &, int64_t, int64_t, hipStream_t);
template void launch_lists<double>(const rk_state &, const kparams<double> &, int64_t, int64_t, hipStream_t);

template <typename F, int Q>
static void launch_dense_q(const rk_state &s, const kparams<F> &p, const int64_t cb[n_classes], const int64_t ce[n_classes],
                           hipStream_t const streams[n_list_R], unsigned class_mask, int what)
{
    const auto *lists = s.cur_lists;
    auto go = [&](auto Rtag, int c) {
        constexpr int R = decltype(Rtag)::value;
        const int64_t n = ce[c] - cb[c];
        if (n <= 0 || !((class_mask >> c) & 1u)) {
            return;
        }
        const dim3 block(64);
        const uint32_t *l = lists + s.cur_off[c] + cb[c];
        const int cnt = static_cast<int>(n);
        auto both = [&](auto NDt) {
            constexpr int ND = decltype(NDt)::value;
            if (what == 0) {
                hipLaunchKernelGGL((k_dense<F, Q, R, ND, false>), dim3(static_cast<unsigned>(n)), block, 0, streams[c], p, l, cnt);
            } else if (what == 1) {
                hipLaunchKernelGGL((k_dense<F, Q, R, ND, true>), dim3(static_cast<unsigned>(n * SL_PARTS_PER_NODE)), block, 0,
                                   streams[c], p, l, cnt);
            } else {
                hipLaunchKernelGGL((k_combine<F, Q, R, ND>), dim3(static_cast<unsigned>(n)), block, 0, streams[c], p, l, cnt);
            }
        };
        if (s.ndim == 3 || !RK_QUAD_BODY) {
            both(std::integral_constant<int, 3>{});
        } else {
            both(std::integral_constant<int, 2>{});
        }
    };
    static_assert(RK_MAX_R == 4, "the dense kernel is instantiated for R = 1..4");
    go(std::integral_constant<int, 2>{}, 1);
    go(std::integral_constant<int, 3>{}, 2);
    go(std::integral_constant<int, 4>{}, 3);
    go(std::integral_constant<int, 1>{}, 0);
}

template <typename F>
void launch_dense(const rk_state &s, int q, const kparams<F> &p, const int64_t cb[n_classes], const int64_t ce[n_classes],
                  hipStream_t const streams[n_list_R], unsigned class_mask, int what)
{
    for (int c = RK_MAX_R; c < big_class; ++c) {
        if (ce[c] != cb[c]) {
            throw error(RK_ERUNTIME, "internal error: target group in a lane-mapping class beyond RK_MAX_R");
        }
    }
    switch (q) {
        case 0: launch_dense_q<F, 0>(s, p, cb, ce, streams, class_mask, what); break;
        case 1: launch_dense_q<F, 1>(s, p, cb, ce, streams, class_mask, what); break;
        case 2: launch_dense_q<F, 2>(s, p, cb, ce, streams, class_mask, what); break;
        default: throw error(RK_EINVAL, "invalid q");
    }
    RK_HIP(hipGetLastError());
}
template void launch_dense<float>(const rk_state &, int, const kparams<float> &, const int64_t[n_classes],
                                  const int64_t[n_classes], hipStream_t const[n_list_R], unsigned, int);
template void launch_dense<double>(const rk_state &, int, const kparams<double> &, const int64_t[n_classes],
                                   const int64_t[n_classes], hipStream_t const[n_list_R], unsigned, int);

// Makes the runtime load this translation unit's code object now (rk_init) instead of at the first launch.
void touch_split()
{
    hipFuncAttributes attr{};
    RK_HIP(hipFuncGetAttributes(&attr, reinterpret_cast<const void *>(&k_lists<float, 0>)));
}

} // namespace rk
