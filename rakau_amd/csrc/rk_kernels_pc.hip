// Variant 3 of the traversal: producer / consumer wavefronts per target group.
//
// The list kernel (rk_kernels_list.hip) alternates list building and dense evaluation inside ONE wavefront per critical
// node. That is the best use of the wave slots when the device is full; a launch with few critical nodes, however, ends
// with the longest serial chains -- the dense phase of the largest groups, R targets per lane times all their sources --
// running alone on their SIMDs. Here a workgroup of 1 + min(R, 2) wavefronts serves one critical node of lane-mapping class R
// (rounds 2-3: 1 + R; see RK_PC_NCONS below for why smaller workgroups are faster):
//
//   * the PRODUCER wave builds the interaction list exactly as the list kernel does (same stack of sibling runs, same
//     box / probe / exact MAC tests, same leaf gathering, same supergroup pre-pass inputs) into a double-buffered LDS tile;
//   * the CONSUMER waves evaluate every published tile, each for one or two of the R target slots of every lane of the list
//     kernel's mapping (slot r = targets r * TP .. r * TP + TP - 1; two slots: the two-target body), while the producer fills
//     the other buffer. One workgroup barrier per tile: the producer arrives when tile k is complete, the consumers when
//     they have finished tile k - 1.
//
// The sequence of tiles, their contents, the number of source splits NS and the order in which every target receives its
// contributions are those of the list kernel, so the results are BIT-IDENTICAL to it (tests assert this) -- which kernel
// serves a call is therefore a pure scheduling decision (rk_launch.hip picks this one for calls over few critical nodes).
// What it buys there: the critical path of a group is max(list building, dense / consumers) instead of list building + dense.
//
// The MAC decisions are those of the reference's CPU engine (include/rakau/tree.hpp:2662-2672 of the reference).
#include "rk_list_common.hpp"

#ifndef RK_PC_W
#define RK_PC_W 5 // fp32: waves per SIMD the kernels are compiled for (96 VGPRs: at most 24 bytes of spills, in the producer)
#endif
#ifndef RK_PC_W64
#define RK_PC_W64 4 // fp64
#endif

#ifndef RK_PC_NB
#define RK_PC_NB 2 // tile buffers per workgroup: 2 = hand-off at a workgroup barrier per tile; 4 / 8 = ring with counters in LDS
#endif
static_assert(RK_PC_NB == 2 || RK_PC_NB == 4 || RK_PC_NB == 8);
#ifndef RK_PC_NCONS
// Consumer wavefronts per workgroup at most (a node of class R uses min(R, RK_PC_NCONS); with fewer consumers than target
// slots a consumer takes two). A launch that cannot fill the device is bound by how many NODES are resident, and workgroups
// are admitted whole: five-wave workgroups (one consumer per slot) were resident 3.1 per CU, three-wave workgroups ~6.
// k_pc_any, kernel ms, 4 -> 3 -> 2 -> 1 consumers: 100k particles 0.137 / 0.112 / 0.107 / 0.137, 150k 0.198 / 0.157 / 0.145 /
// 0.155, 250k 0.31 / 0.23 / 0.217 / 0.198, 350k 0.43 / 0.32 / 0.29 / 0.26 (k_list_any 0.245); fp64 100k 0.248 / - / 0.191 / 0.206
// (profiles/r04/pc_consumers_per_workgroup.txt). Same bits whatever the value.
#define RK_PC_NCONS 2
#endif
static_assert(RK_PC_NCONS >= 1 && RK_PC_NCONS <= 4);

namespace rk
{

constexpr uint32_t PC_LAST = 0x80000000u;

// Ring hand-off (RK_PC_NB > 2): counters in LDS, written by lane 0 of one wavefront and polled by the others. DS operations
// of a wavefront execute in order and the LDS serves one instruction at a time, so a counter stored after a tile's stores
// is seen after them. A wait that outlives any node by orders of magnitude (0.5 s) aborts the kernel instead of hanging.
__device__ __forceinline__ void pc_wait_ge(const uint32_t *p, uint32_t v)
{
    unsigned spins = 0;
    while (__hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < v) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (1u << 23)) {
            __builtin_trap();
        }
    }
}
__device__ __forceinline__ void pc_signal(uint32_t *p, uint32_t v, int lane)
{
    wave_sync();
    if (lane == 0) {
        __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}

// Workgroup LDS: producer-private stack and queues, two tiles of src_cap sources, the published counts.
// fp32: 2 + 1 + 0.5 + 4 KiB = 7.5 KiB per group.
template <typename F>
struct pc_lds {
    uint32_t stack[LK_STACK_CAP];
    uint2 lq[LK_LQ_CAP];
    uint32_t uq[LK_UQ_CAP];
    uint32_t tile_n[RK_PC_NB];
    uint32_t sync[8]; // ring: [0] tiles published, [1 + c] tiles finished by consumer c
    typename vt<F>::v4 tile[RK_PC_NB][lk_cfg<F>::src_cap];
};

// Everything the workgroup (1 + R wavefronts; `role` 0 = producer, 1 + c = consumer c) does for critical node g. Shared by
// the per-class kernels k_pc<..., R> and by k_pc_any, which picks R per node at run time (its surplus wavefronts leave at
// once): the same code, hence the same bits, whichever kernel runs it.
template <typename F, int Q, int MAC, int R, int ND>
__device__ __forceinline__ void pc_node(const kparams<F> &P, pc_lds<F> &L, const uint32_t g, const int role, const int lane)
{
    using v4 = typename vt<F>::v4;
    using v2 = typename vt<F>::v2;
    constexpr int NR = nres_of(Q);
    constexpr int SRC_CAP = lk_cfg<F>::src_cap;
    constexpr int TILE_CAP = SRC_CAP;
    static_assert(R >= 1 && R <= 4);
    // The consumers' split-reduction scratch (64 * NR values each) lives in the two tiles.
    static_assert(RK_PC_NB * lk_cfg<F>::src_cap * 4 * sizeof(F) >= size_t(R) * 64 * 4 * sizeof(F), "reduction scratch does not fit");
    const uint4 c = P.crit[g];
    const uint32_t tb = c.x, te = c.y, cnode = c.z;
    const int T = static_cast<int>(te - tb);
#ifdef RK_TRACE
    const unsigned long long tr_t0 = __builtin_amdgcn_s_memrealtime();
#endif

    // Lane mapping of the list kernel's dense phase: TP target slots, NS source splits, R targets per slot.
    const int TP = (T + R - 1) / R;
    const int NS = 64 / TP;
    const int ts = lane % TP, sp_raw = lane / TP;
    const bool lane_on = sp_raw < NS;
    const int sp = lane_on ? sp_raw : 0; // idle lanes shadow split 0; their results are never stored
    const F eps2 = P.eps2;
    constexpr int NB = RK_PC_NB;
    if constexpr (NB > 2) {
        if (role == 0 && lane < 8) {
            L.sync[lane] = 0u;
        }
        __syncthreads();
    }

    // Consumers per workgroup: min(R, RK_PC_NCONS). With fewer consumers than target slots a consumer takes several slots
    // (CR targets per lane, the multi-target body of the list kernel): every target still receives the sources of its split
    // in list order, so the bits do not depend on how the slots are dealt out.
    constexpr int NC = R < RK_PC_NCONS ? R : RK_PC_NCONS;
    if (role != 0) {
        // =====================================================================================================
        // Consumer `cons`: dense evaluation of the published tiles for target slots r0 .. r0 + CR - 1 (targets r * TP + ts).
        // =====================================================================================================
        const int cons = role - 1;
        auto consume = [&](auto CRtag, const int r0) __attribute__((always_inline)) {
            constexpr int CR = decltype(CRtag)::value;
            lk_regs<F, Q, CR> tg; // this consumer's targets and their sums
            int tidx[CR];
#pragma unroll
            for (int j = 0; j < CR; ++j) {
                tidx[j] = ts + (r0 + j) * TP;
                const bool valid = tidx[j] < T;
                tg.set_target(j, P.part4[tb + (valid ? tidx[j] : 0)]);
                if (!valid) {
                    tidx[j] = -1;
                }
            }
            // n / NS for n <= SRC_CAP without a division per tile: exact for n * NS < 2^16.
            const int inv_ns = (65536 + NS - 1) / NS;
            static_assert(lk_cfg<F>::src_cap * 64 < 65536);
            int buf = 0;
            if constexpr (NB == 2) {
                for (;;) {
                    __syncthreads(); // tile `buf` is published; the producer now owns the other buffer
                    const uint32_t word = L.tile_n[buf];
                    const int n = static_cast<int>(word & ~PC_LAST);
                    if (n > 0) {
                        lk_eval_tile<F, Q, CR, false, ND>(L.tile[buf], n, (n * inv_ns) >> 16, sp, NS, true, lane_on, tg, eps2, tidx);
                    }
                    if (word & PC_LAST) {
                        break;
                    }
                    buf ^= 1;
                }
            } else {
                // Ring: tile i sits in buffer i % NB once the producer has counted it; this consumer counts it when done.
                for (uint32_t it = 0;;) {
                    pc_wait_ge(&L.sync[0], it + 1u);
                    buf = static_cast<int>(it & static_cast<uint32_t>(NB - 1));
                    const uint32_t word = L.tile_n[buf];
                    const int n = static_cast<int>(word & ~PC_LAST);
                    if (n > 0) {
                        lk_eval_tile<F, Q, CR, false, ND>(L.tile[buf], n, (n * inv_ns) >> 16, sp, NS, true, lane_on, tg, eps2, tidx);
                    }
                    ++it;
                    pc_signal(&L.sync[1 + cons], it, lane);
                    if (word & PC_LAST) {
                        break;
                    }
                }
                if constexpr (NC > 1) {
                    __syncthreads(); // every consumer has finished every tile (the producer has left or is leaving)
                }
            }
            // ---- interactions inside the group: its own particles as sources, self pair masked ----
            // The producer has published its last tile and left: the buffer it would fill next is free. Consumer 0 loads the
            // group's particles, every consumer evaluates them for its own targets.
            {
                v4 *self = L.tile[NB == 2 ? (buf ^ 1) : 0];
                for (int b0 = 0; b0 < T; b0 += SRC_CAP) {
                    const int n = (T - b0) < SRC_CAP ? (T - b0) : SRC_CAP;
                    if (cons == 0) {
                        for (int j = lane; j < n; j += 64) {
                            self[j] = P.part4[tb + static_cast<uint32_t>(b0 + j)];
                        }
                    }
                    if constexpr (NC > 1) {
                        __syncthreads();
                    } else {
                        wave_sync();
                    }
                    int tloc[CR];
#pragma unroll
                    for (int j = 0; j < CR; ++j) {
                        tloc[j] = tidx[j] < 0 ? -1 : tidx[j] - b0;
                    }
                    lk_eval_tile<F, Q, CR, true, ND>(self, n, (n * inv_ns) >> 16, sp, NS, true, lane_on, tg, eps2, tloc);
                    if constexpr (NC > 1) {
                        __syncthreads();
                    } else {
                        wave_sync();
                    }
                }
            }
            // ---- sum the source splits in a fixed order (scratch: one region of 64 * NR values per target slot) ----
            if (NS > 1) {
                F *red = reinterpret_cast<F *>(&L.tile[0][0]) + r0 * 64 * NR;
                if (lane_on) {
#pragma unroll
                    for (int j = 0; j < CR; ++j) {
#pragma unroll
                        for (int k = 0; k < NR; ++k) {
                            red[j * 64 * NR + (sp_raw * TP + ts) * NR + k] = tg.get(j, k);
                        }
                    }
                }
                wave_sync();
                if (lane_on && sp_raw == 0) {
#pragma unroll
                    for (int j = 0; j < CR; ++j) {
#pragma unroll
                        for (int k = 0; k < NR; ++k) {
                            F sum = F(0);
                            for (int s = 0; s < NS; ++s) {
                                sum += red[j * 64 * NR + (s * TP + ts) * NR + k];
                            }
                            tg.set(j, k, sum);
                        }
                    }
                }
            }
            // ---- scale by G, write out ----
            if (lane_on && sp_raw == 0) {
                const F G = P.G;
                // (nothing but stores once the positions are known: see the list kernel's epilogue)
                auto store_all = [&](auto pos) __attribute__((always_inline)) {
#pragma unroll
                    for (int j = 0; j < CR; ++j) {
                        if (tidx[j] >= 0) {
                            const uint32_t o = pos(j);
#pragma unroll
                            for (int k = 0; k < NR; ++k) {
                                if (ND == 3 || Q == 1 || k != 2) { // a quadtree has no z acceleration (and no array for it)
                                    P.out[k][o] = tg.get(j, k) * G;
                                }
                            }
                        }
                    }
                };
                if (P.perm) {
                    uint32_t o[CR];
#pragma unroll
                    for (int j = 0; j < CR; ++j) {
                        o[j] = P.perm[tb + static_cast<uint32_t>(tidx[j] >= 0 ? tidx[j] : 0)];
                    }
                    __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0)
                    store_all([&](int j) { return o[j]; });
                } else {
                    store_all([&](int j) { return tb + static_cast<uint32_t>(tidx[j]) - P.out_sub; });
                }
            }
        };
        // Slots dealt out as evenly as possible: the first R % NC consumers take one more.
        constexpr int CR_LO = R / NC, N_HI = R % NC;
        if (N_HI != 0 && cons < N_HI) {
            consume(std::integral_constant<int, CR_LO + (N_HI != 0 ? 1 : 0)>{}, cons * (CR_LO + 1));
        } else {
            consume(std::integral_constant<int, CR_LO>{}, N_HI * (CR_LO + 1) + (cons - N_HI) * CR_LO);
        }
#ifdef RK_TRACE
        if (cons == 0 && lane == 0 && P.dbg) {
            const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20);
            P.dbg[4u * g] = tr_t0;
            P.dbg[4u * g + 1u] = __builtin_amdgcn_s_memrealtime();
            P.dbg[4u * g + 2u] = (static_cast<unsigned long long>(xcc) << 32) | hw;
            P.dbg[4u * g + 3u] = (static_cast<unsigned long long>(R) << 32) | static_cast<unsigned>(T);
        }
#endif
        return;
    }

    // The producer keeps all R targets of the list kernel's lane mapping for the lane = target exact MAC test.
    v4 tp[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int ti = ts + r * TP;
        tp[r] = P.part4[tb + (ti < T ? ti : 0)];
    }
    // =========================================================================================================
    // Producer: list building (the code of the list kernel, with flush() replaced by publish()).
    // =========================================================================================================
    const F mac_value = P.mac_value;
    // Bounding box of the group's particles and two probe targets (first and last): wave-uniform.
    const v4 blo = P.crit_box[2u * g], bhi = P.crit_box[2u * g + 1u];
    const v4 pr0 = P.part4[tb], pr1 = P.part4[te - 1u];
    int size = 0, n_src = 0, n_lq = 0, n_uq = 0, cur = 0;
    v4 *src = L.tile[0];
    // Supergroup pre-pass results for this group's supergroup (if enabled and not overflowed).
    uint32_t sup_S = 0, sup_ncommon = 0, sup_nresid = 0, sup_rpos = 0;
    bool from_root = true;
    if (P.super_k != 0u) {
        sup_S = g / P.super_k;
        const uint2 cnt = P.sup_cnt[sup_S];
        if ((cnt.y >> 31) == 0u) {
            from_root = false;
            sup_ncommon = cnt.x;
            sup_nresid = cnt.y;
        }
    }
    if (from_root) {
        // The root is an ancestor of every group (or the group itself): start from its children.
        const node_rec<F> *root = P.node_rec;
        const uint32_t r_nch = root->nch, r_a = root->a, r_b = root->b;
        if (cnode != 0u && r_nch != 0u) {
            if (lane == 0) {
                L.stack[0] = (r_a << 3) | (r_b - 1u);
            }
            size = 1;
        }
    }
    wave_sync();

    // Hand the current tile to the consumers and take the other buffer (which they have finished with once they
    // arrive at this barrier).
    [[maybe_unused]] uint32_t n_pub = 0;
    // `whole`: the consumers evaluate everything in the tile (masked remainder step); otherwise -- as in the list kernel's
    // flush(false) with RK_CARRY_REMAINDER -- only whole rounds of NS sources are handed over and the fewer than NS left over
    // open the next tile: the same tile contents as the list kernel's, hence the same bits.
    const int inv_ns_p = (65536 + NS - 1) / NS;
    auto publish = [&](bool last, bool whole) __attribute__((always_inline)) {
        int n_out = n_src, left = 0;
        if (RK_CARRY_REMAINDER && !whole && !last) {
            n_out = ((n_src * inv_ns_p) >> 16) * NS;
            left = n_src - n_out;
        }
        const int from = cur;
        if (lane == 0) {
            L.tile_n[cur] = static_cast<uint32_t>(n_out) | (last ? PC_LAST : 0u);
        }
        if constexpr (NB == 2) {
            __syncthreads();
            cur ^= 1;
        } else {
            ++n_pub;
            pc_signal(&L.sync[0], n_pub, lane);
            cur = (cur + 1) & (NB - 1);
            if (!last && n_pub >= static_cast<uint32_t>(NB)) {
                // The buffer of tile n_pub held tile n_pub - NB: every consumer must have counted it.
                for (int c = 0; c < NC; ++c) {
                    pc_wait_ge(&L.sync[1 + c], n_pub - static_cast<uint32_t>(NB) + 1u);
                }
            }
        }
        src = L.tile[cur];
        if (left > 0) {
            // (the buffer just handed over is only read by the consumers; the one taken is free: see the waits above)
            if (lane < left) {
                src[lane] = L.tile[from][n_out + lane];
            }
            wave_sync();
        }
        n_src = left;
    };
    // The list kernel evaluates a tile as soon as another batch (up to 64 sources) might not fit; so does this one
    // (identical tile boundaries).
    auto flush = [&](bool whole) __attribute__((always_inline)) {
        if (n_src > 0) {
            publish(false, whole);
        }
    };

    // Gather the particles of the queued leaves into the source tile, publishing the tile when it fills.
    auto drain_leaves = [&]() __attribute__((always_inline)) {
        while (n_lq > 0) {
            const int free_slots = TILE_CAP - n_src;
            uint2 lf = make_uint2(0u, 0u);
            if (lane < n_lq) {
                lf = L.lq[lane];
            }
            const unsigned cnt = lane < n_lq ? lf.y - lf.x : 0u;
            const unsigned incl = wave_incl_scan(cnt);
            const bool fits = lane < n_lq && incl <= static_cast<unsigned>(free_slots);
            const unsigned long long m_fit = __builtin_amdgcn_ballot_w64(fits);
            const int m = __builtin_popcountll(m_fit); // leaves [0, m) fit (prefix property)
            if (m == 0) {
                if (n_src > 0) {
                    flush(true); // everything, so that the tile really is empty afterwards
                    continue;
                }
                // A single leaf larger than the whole tile: take TILE_CAP of its particles.
                const uint32_t b0 = __builtin_amdgcn_readfirstlane(lf.x);
                for (int j = lane; j < TILE_CAP; j += 64) {
                    src[j] = P.part4[b0 + static_cast<uint32_t>(j)];
                }
                if (lane == 0) {
                    L.lq[0] = make_uint2(b0 + static_cast<uint32_t>(TILE_CAP), lf.y);
                }
                n_src = TILE_CAP;
                wave_sync();
                flush(true);
                continue;
            }
            // Lane l copies the particles of leaf l, eight loads in flight at a time.
            const unsigned mycnt = fits ? cnt : 0u;
            const int dst = n_src + static_cast<int>(incl - cnt);
            // Unconditional loads (index clamped into the leaf; particle 0 for idle lanes), all issued before the first
            // store. Written out by hand: as an array the eight records end up in scratch memory in this kernel.
            const uint32_t lbase = mycnt ? lf.x : 0u, llast = mycnt ? mycnt - 1u : 0u;
            for (unsigned j0 = 0; __builtin_amdgcn_ballot_w64(j0 < mycnt) != 0ull; j0 += 8u) {
#define RK_PC_LD(i) const v4 t##i = P.part4[lbase + (j0 + i##u < llast ? j0 + i##u : llast)];
                RK_PC_LD(0) RK_PC_LD(1) RK_PC_LD(2) RK_PC_LD(3) RK_PC_LD(4) RK_PC_LD(5) RK_PC_LD(6) RK_PC_LD(7)
#undef RK_PC_LD
#define RK_PC_ST(i)                                                                                                    \
    if (j0 + i##u < mycnt) {                                                                                           \
        src[dst + static_cast<int>(j0 + i##u)] = t##i;                                                                 \
    }
                RK_PC_ST(0) RK_PC_ST(1) RK_PC_ST(2) RK_PC_ST(3) RK_PC_ST(4) RK_PC_ST(5) RK_PC_ST(6) RK_PC_ST(7)
#undef RK_PC_ST
            }
            n_src += __builtin_amdgcn_readlane(static_cast<int>(incl), m - 1);
            // Drop the consumed leaves from the queue (move the rest down, 64 entries at a time).
            const int tail = n_lq - m;
            wave_sync();
            for (int j0 = 0; j0 < tail; j0 += 64) {
                const int j = j0 + lane;
                uint2 mv = make_uint2(0u, 0u);
                if (j < tail) {
                    mv = L.lq[j + m];
                }
                wave_sync();
                if (j < tail) {
                    L.lq[j] = mv;
                }
                wave_sync();
            }
            n_lq = tail;
            if (n_src + 64 > TILE_CAP) {
                flush(false);
            }
        }
    };

    // A batch of up to 64 candidate nodes held in registers (lane = candidate).
    struct batch_t {
        bool active;
        v4 com;
        v2 mp;
        uint32_t node, nch, ra, rb, rec;
    };
    auto load_rec = [&](batch_t &bt) __attribute__((always_inline)) {
        // Everything about the candidate in three independent 16-byte loads (record 0 for idle lanes).
        const node_rec<F> *rec = P.node_rec + bt.rec;
        bt.com = rec->com;
        bt.mp = rec->mac;
        bt.node = rec->dfs;
        bt.nch = rec->nch;
        bt.ra = rec->a;
        bt.rb = rec->b;
    };
    // Pop up to 8 sibling runs and issue the loads of their records. Returns the number of entries popped.
    // `pending` = number of entries a batch already in flight may still push (no batch is prefetched at present: two
    // batches in flight were measured on trees of 30k-1M particles and changed nothing, tools/archive/jobs_r02/r02_job10.sh).
    auto pop_and_load = [&](batch_t &bt, int pending) __attribute__((always_inline)) -> int {
        if (size == 0) {
            return 0;
        }
        int k = size < 8 ? size : 8;
        // Keep the stack within bounds even if every candidate is opened (8 pushes per popped entry);
        // otherwise fall back to one entry at a time (depth-first), whose growth is bounded by LK_DFS_RESERVE -- only when
        // nothing else is in flight.
        const int room = (LK_STACK_CAP - LK_DFS_RESERVE - pending - n_uq - size) / 7;
        if (room < k) {
            if (room >= 1) {
                k = room;
            } else if (pending == 0) {
                k = 1;
            } else {
                return 0;
            }
        }
        const int e_idx = lane >> 3, e_sub = lane & 7;
        uint32_t entry = 0u;
        if (e_idx < k) {
            entry = L.stack[size - 1 - e_idx];
        }
        size -= k;
        bt.active = e_idx < k && static_cast<uint32_t>(e_sub) <= (entry & 7u);
        bt.rec = bt.active ? (entry >> 3) + static_cast<uint32_t>(e_sub) : 0u;
        load_rec(bt);
        return k;
    };

    // Route classified candidates: accepted nodes go to the source tile, opened leaves to the leaf queue,
    // opened internal nodes push their run of children, undecided ones go to the exact-test queue.
    auto route = [&](bool accept, bool open, bool undecided, const batch_t &bt) __attribute__((always_inline)) {
        const bool leaf = open && bt.nch == 0u;
        const bool expand = open && bt.nch != 0u;
        const unsigned long long m_acc = __builtin_amdgcn_ballot_w64(accept);
        if (accept) {
            src[n_src + static_cast<int>(wave_prefix_count(m_acc))] = bt.com;
        }
        n_src += __builtin_popcountll(m_acc);
        const unsigned long long m_leaf = __builtin_amdgcn_ballot_w64(leaf);
        if (leaf) {
            L.lq[n_lq + static_cast<int>(wave_prefix_count(m_leaf))] = make_uint2(bt.ra, bt.rb);
        }
        n_lq += __builtin_popcountll(m_leaf);
        const unsigned long long m_exp = __builtin_amdgcn_ballot_w64(expand);
        if (expand) {
            L.stack[size + static_cast<int>(wave_prefix_count(m_exp))] = (bt.ra << 3) | (bt.rb - 1u);
        }
        size += __builtin_popcountll(m_exp);
        const unsigned long long m_und = __builtin_amdgcn_ballot_w64(undecided);
        if (undecided) {
            L.uq[n_uq + static_cast<int>(wave_prefix_count(m_und))] = bt.rec;
        }
        n_uq += __builtin_popcountll(m_und);
        wave_sync();
    };

    // First-stage MAC test of one batch: accept if the squared distance from the centre of mass to the group's
    // bounding box exceeds mac_lh by a margin; open if one of two probe targets already violates the criterion; the rest
    // is queued for the exact all-targets test (see rk_kernels_list.hip), so every decision equals the reference's.
    auto process = [&](const batch_t &bt) __attribute__((always_inline)) {
        const v4 com = bt.com;
        // Ancestor-or-self of the target group (tree.hpp:2828-2838 of the reference) on the depth-first
        // index interval of the subtree.
        const bool anc = bt.active && bt.node <= cnode && cnode <= bt.node + bt.nch;
        const bool self = anc && bt.node == cnode;
        const bool test = bt.active && !anc;
        const F mac_lh = mac_lhs<F>(MAC == RK_MAC_RT ? P.mac : MAC, bt.mp, mac_value);
        const F bx = rk_max3(blo.x - com.x, com.x - bhi.x, F(0)), by = rk_max3(blo.y - com.y, com.y - bhi.y, F(0)),
                bz = rk_max3(blo.z - com.z, com.z - bhi.z, F(0));
        const F dbox2 = rk_fma(bz, bz, rk_fma(by, by, bx * bx));
        const bool box_accept = dbox2 > mac_lh * F(1.00001);
        const F p0x = com.x - pr0.x, p0y = com.y - pr0.y, p0z = com.z - pr0.z;
        const F p1x = com.x - pr1.x, p1y = com.y - pr1.y, p1z = com.z - pr1.z;
        const F d2p0 = rk_fma(p0z, p0z, rk_fma(p0y, p0y, p0x * p0x)), d2p1 = rk_fma(p1z, p1z, rk_fma(p1y, p1y, p1x * p1x));
        const bool probe_open = mac_lh >= rk_min(d2p0, d2p1);
        const bool accept = test && box_accept;
        const bool open = (test && !box_accept && probe_open) || (anc && !self);
        const bool undecided = test && !box_accept && !probe_open;
        route(accept, open, undecided, bt);
    };

    // Exact MAC test (all targets) of up to 64 queued candidates.
    auto process_exact = [&]() __attribute__((always_inline)) {
        const int k = n_uq < 64 ? n_uq : 64;
        batch_t bt;
        bt.active = lane < k;
        bt.rec = bt.active ? L.uq[n_uq - 1 - lane] : 0u;
        n_uq -= k;
        load_rec(bt);
        const v4 com = bt.com;
        const F mac_lh = mac_lhs<F>(MAC == RK_MAC_RT ? P.mac : MAC, bt.mp, mac_value);
        bool fail;
        if (k * (7 * R + 3) < T * 7) {
            // Few candidates: lane = target (every lane keeps R targets of the group in registers; unused slots repeat
            // target 0). The main loop keeps 64 free slots behind n_src in the tile; the candidates are staged there.
            v4 cd;
            cd.x = com.x, cd.y = com.y, cd.z = com.z, cd.w = mac_lh;
            if (bt.active) {
                src[n_src + lane] = cd;
            }
            wave_sync();
            unsigned long long fail_mask = 0ull;
            for (int ci = 0; ci < k; ++ci) {
                const v4 cand = src[n_src + ci];
                bool f = false;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const F dx = cand.x - tp[r].x, dy = cand.y - tp[r].y, dz = cand.z - tp[r].z;
                    const F d2 = rk_fma(dz, dz, rk_fma(dy, dy, dx * dx));
                    f |= cand.w >= d2;
                }
                if (__builtin_amdgcn_ballot_w64(f) != 0ull) {
                    fail_mask |= 1ull << ci;
                }
            }
            fail = ((fail_mask >> lane) & 1ull) != 0ull;
            wave_sync();
        } else {
            // min over the targets of the unsoftened squared distance to the node's centre of mass. The target
            // coordinates are wave-uniform: they arrive through scalar loads as SGPR operands.
            F mind2 = std::numeric_limits<F>::infinity();
            for (int t = 0; t < T; t += 4) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int ti = (t + u < T) ? t + u : T - 1;
                    const v4 tg = P.part4[tb + static_cast<uint32_t>(ti)];
                    const F dx = com.x - tg.x, dy = com.y - tg.y, dz = com.z - tg.z;
                    const F d2 = rk_fma(dz, dz, rk_fma(dy, dy, dx * dx));
                    mind2 = rk_min(mind2, d2);
                }
            }
            fail = mac_lh >= mind2;
        }
        route(bt.active && !fail, bt.active && fail, false, bt);
    };

    // ---- sources accepted for the whole supergroup: stream them from the pre-pass list through the tile ----
    {
        const v4 *common = P.sup_common + static_cast<size_t>(sup_S) * SUP_CAPC;
        for (uint32_t base = 0; base < sup_ncommon;) {
            const uint32_t room = static_cast<uint32_t>(TILE_CAP - n_src), left = sup_ncommon - base;
            const uint32_t take = left < room ? left : room;
            for (uint32_t j = lane; j < take; j += 64u) {
                src[n_src + static_cast<int>(j)] = common[base + j];
            }
            n_src += static_cast<int>(take);
            base += take;
            wave_sync();
            if (n_src == TILE_CAP) {
                flush(false);
            }
        }
    }
    // Candidates the pre-pass left to the member groups: taken 64 at a time whenever the stack runs empty.
    auto resid_load = [&](batch_t &bt) __attribute__((always_inline)) -> int {
        if (sup_rpos >= sup_nresid) {
            return 0;
        }
        const uint32_t left = sup_nresid - sup_rpos, k = left < 64u ? left : 64u;
        bt.active = static_cast<uint32_t>(lane) < k;
        bt.rec = bt.active ? P.sup_resid[static_cast<size_t>(sup_S) * SUP_CAPR + sup_rpos + static_cast<uint32_t>(lane)] : 0u;
        sup_rpos += k;
        load_rec(bt);
        return 1;
    };
    // ---- list building ----
    auto fetch = [&](batch_t &bt, int pending) __attribute__((always_inline)) -> bool {
        if (pop_and_load(bt, pending) != 0) {
            return true;
        }
        return resid_load(bt) != 0;
    };
    {
        bool done = false;
        for (;;) {
            // Room for the worst-case output of one pass (64 sources, 64 leaves); everything is settled at the end.
            if (n_lq + 64 > LK_LQ_CAP || done) {
                drain_leaves();
            }
            if (done) {
                break;
            }
            if (n_src + 64 > TILE_CAP) {
                flush(false);
            }
            if (n_uq >= 64) {
                process_exact();
                continue;
            }
            // Near the stack bound the descent is one entry at a time, and LK_DFS_RESERVE only bounds a STRICT
            // depth-first descent: settle the parked candidates first.
            if (n_uq > 0 && LK_STACK_CAP - LK_DFS_RESERVE - n_uq - size < 7) {
                process_exact();
                continue;
            }
            batch_t A;
            if (!fetch(A, 0)) {
                // Stack and residual list exhausted: settle the undecided candidates (they may open new runs).
                if (n_uq > 0) {
                    process_exact();
                } else {
                    done = true;
                }
                continue;
            }
            process(A);
        }
    }
    publish(true, true);
}

template <typename F, int Q, int MAC, int R, int ND>
__global__ void __launch_bounds__(64 * (1 + (R < RK_PC_NCONS ? R : RK_PC_NCONS)), (sizeof(F) == 4 ? RK_PC_W : RK_PC_W64))
    k_pc(const kparams<F> P, const uint32_t *__restrict__ list, int n_list)
{
    __shared__ pc_lds<F> L;
    const int role = threadIdx.x >> 6; // 0 = producer, 1 + c = consumer c
    const int lane = threadIdx.x & 63;
    const unsigned blk = xcd_map_block(blockIdx.x, gridDim.x, P.xcd_mode);
    if (static_cast<int>(blk) >= n_list) {
        return;
    }
    const uint32_t g = __builtin_amdgcn_readfirstlane(list[blk]);
    if (g == RK_PLAN_PAD_VALUE) {
        return; // padding of a launch plan
    }
    pc_node<F, Q, MAC, R, ND>(P, L, g, role, lane);
}

// One launch for the critical nodes of ALL lane-mapping classes (calls over very few critical nodes): workgroups of five
// wavefronts, of which a node of class R uses 1 + R; the others return before the first barrier (a barrier waits only for
// the wavefronts of the workgroup that are still alive).
template <typename F, int Q, int MAC, int ND>
__global__ void __launch_bounds__(64 * (1 + RK_PC_NCONS), (sizeof(F) == 4 ? RK_PC_W : RK_PC_W64))
    k_pc_any(const kparams<F> P, const uint32_t *__restrict__ list, int n_list)
{
    __shared__ pc_lds<F> L;
    const int role = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63;
    const unsigned blk = xcd_map_block(blockIdx.x, gridDim.x, P.xcd_mode);
    if (static_cast<int>(blk) >= n_list) {
        return;
    }
    unsigned entry = 0u;
    if (!any_list_entry(P, blk, n_list, entry)) {
        return;
    }
    const uint32_t g = __builtin_amdgcn_readfirstlane(list[entry]);
    if (g == RK_PLAN_PAD_VALUE) {
        return;
    }
    const int cls = class2_of_compute(static_cast<int64_t>(__builtin_amdgcn_readfirstlane(P.crit[g].w)));
    if (cls < 0 || cls >= RK_MAX_R || role > (cls + 1 < RK_PC_NCONS ? cls + 1 : RK_PC_NCONS)) {
        return;
    }
    switch (cls) {
        case 0: pc_node<F, Q, MAC, 1, ND>(P, L, g, role, lane); break;
        case 1: pc_node<F, Q, MAC, 2, ND>(P, L, g, role, lane); break;
        case 2: pc_node<F, Q, MAC, 3, ND>(P, L, g, role, lane); break;
        default: pc_node<F, Q, MAC, 4, ND>(P, L, g, role, lane); break;
    }
}

// ------------------------------------------------------------------------------------------------
// Launch.
// ------------------------------------------------------------------------------------------------
template <typename F, int Q, int MAC>
static void launch_pc_qm(const rk_state &s, const kparams<F> &p, const int64_t cb[n_classes], const int64_t ce[n_classes],
                         hipStream_t const streams[n_list_R], unsigned class_mask)
{
    const auto *lists = s.cur_lists; // class lists of the state, or the launch plan of this call
    auto go = [&](auto Rtag, int c) {
        constexpr int R = decltype(Rtag)::value;
        const int64_t n = ce[c] - cb[c];
        if (n <= 0 || !((class_mask >> c) & 1u)) {
            return;
        }
        const auto grid = static_cast<unsigned>(n);
        if (s.ndim == 3 || !RK_QUAD_BODY) {
            hipLaunchKernelGGL((k_pc<F, Q, MAC, R, 3>), dim3(grid), dim3(64 * (1 + (R < RK_PC_NCONS ? R : RK_PC_NCONS))), 0, streams[c], p,
                               lists + s.cur_off[c] + cb[c], static_cast<int>(n));
        } else {
            hipLaunchKernelGGL((k_pc<F, Q, MAC, R, 2>), dim3(grid), dim3(64 * (1 + (R < RK_PC_NCONS ? R : RK_PC_NCONS))), 0, streams[c], p,
                               lists + s.cur_off[c] + cb[c], static_cast<int>(n));
        }
    };
    static_assert(RK_MAX_R == 4, "the producer / consumer kernel is instantiated for R = 1..4");
    // The classes with the longest serial chains first.
    go(std::integral_constant<int, 4>{}, 3);
    go(std::integral_constant<int, 2>{}, 1);
    go(std::integral_constant<int, 3>{}, 2);
    go(std::integral_constant<int, 1>{}, 0);
}

template <typename F>
void launch_pc_any(const rk_state &s, int q, const kparams<F> &p, const uint32_t *list, int64_t n, hipStream_t stream)
{
    if (n <= 0) {
        return;
    }
    const dim3 grid(static_cast<unsigned>(n)), block(64 * (1 + RK_PC_NCONS));
    const int cnt = static_cast<int>(n);
    auto go = [&](auto Qt, auto Mt) {
        constexpr int Q = decltype(Qt)::value, M = mac_targ(decltype(Mt)::value);
        if (s.ndim == 3 || !RK_QUAD_BODY) {
            hipLaunchKernelGGL((k_pc_any<F, Q, M, 3>), grid, block, 0, stream, p, list, cnt);
        } else {
            hipLaunchKernelGGL((k_pc_any<F, Q, M, 2>), grid, block, 0, stream, p, list, cnt);
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
template void launch_pc_any<float>(const rk_state &, int, const kparams<float> &, const uint32_t *, int64_t, hipStream_t);
template void launch_pc_any<double>(const rk_state &, int, const kparams<double> &, const uint32_t *, int64_t, hipStream_t);

template <typename F>
void launch_pc(const rk_state &s, int q, const kparams<F> &p, const int64_t cb[n_classes], const int64_t ce[n_classes],
               hipStream_t const streams[n_list_R], unsigned class_mask)
{
    for (int c = RK_MAX_R; c < big_class; ++c) {
        if (ce[c] != cb[c]) {
            throw error(RK_ERUNTIME, "internal error: target group in a lane-mapping class beyond RK_MAX_R");
        }
    }
    switch (q * 2 + s.mac) {
        case 0: launch_pc_qm<F, 0, mac_targ(0)>(s, p, cb, ce, streams, class_mask); break;
        case 1: launch_pc_qm<F, 0, mac_targ(1)>(s, p, cb, ce, streams, class_mask); break;
        case 2: launch_pc_qm<F, 1, mac_targ(0)>(s, p, cb, ce, streams, class_mask); break;
        case 3: launch_pc_qm<F, 1, mac_targ(1)>(s, p, cb, ce, streams, class_mask); break;
        case 4: launch_pc_qm<F, 2, mac_targ(0)>(s, p, cb, ce, streams, class_mask); break;
        case 5: launch_pc_qm<F, 2, mac_targ(1)>(s, p, cb, ce, streams, class_mask); break;
        default: throw error(RK_EINVAL, "invalid q / mac combination");
    }
    RK_HIP(hipGetLastError());
}

template void launch_pc<float>(const rk_state &, int, const kparams<float> &, const int64_t[n_classes],
                               const int64_t[n_classes], hipStream_t const[n_list_R], unsigned);
template void launch_pc<double>(const rk_state &, int, const kparams<double> &, const int64_t[n_classes],
                                const int64_t[n_classes], hipStream_t const[n_list_R], unsigned);

// Makes the runtime load this translation unit's code object now (rk_init) instead of at the first launch.
void touch_pc()
{
    hipFuncAttributes attr{};
    RK_HIP(hipFuncGetAttributes(&attr, reinterpret_cast<const void *>(&k_pc<float, 0, mac_targ(0), 2, 3>)));
}

} // namespace rk
