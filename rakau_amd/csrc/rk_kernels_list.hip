// Variant 2 of the traversal: per-wavefront interaction lists in LDS.
//
// One wavefront serves one target group (critical node). It alternates between two lane mappings:
//
//  (1) list building, lane = candidate node. Up to 64 candidate nodes are popped from a per-wave LDS
//      stack (runs of siblings), their records are fetched with 64 independent loads (memory-level parallelism
//      instead of the dependent scalar chain of variant 1), and every lane tests ITS node: first against the
//      group's bounding box and two probe targets, and only if that is inconclusive against ALL targets of
//      the group (target coordinates arrive through scalar loads, i.e. as SGPR operands). Accepted nodes
//      are compacted (ballot + prefix popcount) into an LDS tile of sources {x, y, z, m}; rejected
//      internal nodes push their children; rejected leaves queue their particle range, which is then
//      gathered into the same tile.
//  (2) dense evaluation, lane = (target slot, source split). When the tile is full it is consumed by a
//      dense targets x sources loop: each lane keeps R targets in registers and walks ITS contiguous share of
//      the tile (NS = number of source splits that fit in 64 lanes next to the target slots), reading
//      sources with broadcast ds_read_b128. Accumulators live in registers across tiles; the splits are
//      summed in a fixed order at the end (deterministic).
//
// The MAC decisions are those of the reference's CPU engine (all particles of the critical node must
// pass, include/rakau/tree.hpp:2662-2672 of the reference), hence the interaction set is identical;
// only the summation order differs.
#include "rk_list_common.hpp"

namespace rk
{

// Everything one wavefront does for critical node g (BIG: for the chunks wib, wib + LK_BIG_WPB, ... of its targets). Shared
// by the per-class kernels k_list<..., R> and by k_list_any, which picks R per node at run time: the same code, hence the
// same bits, whichever kernel runs it.
template <typename F, int Q, int MAC, int R, int ND, bool BIG>
__device__ __forceinline__ void list_node(const kparams<F> &P, lk_wave_lds<F> &L, const uint32_t g, const int lane, const int wib)
{
    using v4 = typename vt<F>::v4;
    using v2 = typename vt<F>::v2;
    constexpr int NR = nres_of(Q);
    constexpr int SRC_CAP = lk_cfg<F>::src_cap;
    static_assert(sizeof(lk_wave_lds<F>) >= 64 * 4 * sizeof(F), "reduction scratch does not fit");
    const uint4 c = P.crit[g];
    const uint32_t gb = c.x, ge = c.y, cnode = c.z;
    const int TG = static_cast<int>(ge - gb);
    // The targets [tb, te) of this wavefront: the whole critical node, or one chunk of an oversized one.
    auto run_targets = [&](const uint32_t tb, const uint32_t te) __attribute__((always_inline)) {
        const int T = static_cast<int>(te - tb);

        // Lane mapping of the dense phase: TP target slots, NS source splits.
        const int TP = (T + R - 1) / R;
        const int NS = 64 / TP;
        const int ts = lane % TP, sp_raw = lane / TP;
        const bool lane_on = sp_raw < NS;
        const int sp = lane_on ? sp_raw : 0; // idle lanes shadow split 0; their results are never stored

        lk_regs<F, Q, R> tg; // the lane's R targets and their sums
        int tidx[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            tidx[r] = ts + r * TP;
            const bool valid = tidx[r] < T;
            tg.set_target(r, P.part4[tb + (valid ? tidx[r] : 0)]);
            if (!valid) {
                tidx[r] = -1;
            }
        }

        const F mac_value = P.mac_value, eps2 = P.eps2;
        // Bounding box of the group's particles and two probe targets (first and last): wave-uniform.
        const v4 blo = P.crit_box[2u * g], bhi = P.crit_box[2u * g + 1u];
        const v4 pr0 = P.part4[gb], pr1 = P.part4[ge - 1u];
        RK_STAMP_DECL
#ifdef RK_TRACE
        // Diagnostic build: wall-clock interval (100 MHz counter) and placement of every wave, for occupancy timelines.
        const unsigned long long tr_t0 = __builtin_amdgcn_s_memrealtime(), tr_c0 = __builtin_amdgcn_s_memtime();
#endif
        int size = 0, n_src = 0, n_lq = 0, n_uq = 0;
        RK_COUNT_DECL
        RK_COUNT(0, 1) RK_COUNT(20, NS) RK_COUNT(21, TP * NS) RK_COUNT(22, T * NS) RK_COUNT(26, T)
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

        // n / NS for n <= SRC_CAP without a division per tile: exact for n * NS < 2^16.
        const int inv_ns = (65536 + NS - 1) / NS;
        static_assert(SRC_CAP * 64 < 65536);
        // Evaluate the tile. Unless `final`, only whole rounds of NS sources are consumed and the (< NS) sources left
        // over move to the front of the tile, so that no masked remainder step is paid per tile.
        auto flush = [&](bool final) __attribute__((always_inline)) {
            if (n_src > 0) {
                RK_STAMP(7)
                const int full = (n_src * inv_ns) >> 16;
                RK_COUNT(12, 1) RK_COUNT(13, full) RK_COUNT(14, (final || !RK_CARRY_REMAINDER) && n_src - full * NS > 0 ? 1 : 0) RK_COUNT(15, n_src)
                lk_eval_tile<F, Q, R, false, ND>(L.src, n_src, full, sp, NS, final || !RK_CARRY_REMAINDER, lane_on, tg, eps2, tidx);
#if RK_CARRY_REMAINDER
                const int left = final ? 0 : n_src - full * NS;
                v4 keep;
                if (lane < left) {
                    keep = L.src[full * NS + lane];
                }
                wave_sync();
                if (lane < left) {
                    L.src[lane] = keep;
                }
                n_src = left;
#else
                n_src = 0;
#endif
                wave_sync();
                RK_STAMP(4)
            }
        };

        // Gather the particles of the queued leaves into the source tile, evaluating the tile when it fills.
        auto drain_leaves = [&]() __attribute__((always_inline)) {
#ifdef RK_ABLATE_LEAVES
            n_lq = 0; // diagnostic build: opened leaves are dropped
#endif
            while (n_lq > 0) {
                RK_STAMP(7)
                const int free_slots = SRC_CAP - n_src;
                RK_COUNT(8, 1)
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
                    // A single leaf larger than the whole tile: take SRC_CAP of its particles.
                    const uint32_t b0 = __builtin_amdgcn_readfirstlane(lf.x);
                    for (int j = lane; j < SRC_CAP; j += 64) {
                        L.src[j] = P.part4[b0 + static_cast<uint32_t>(j)];
                    }
                    if (lane == 0) {
                        L.lq[0] = make_uint2(b0 + static_cast<uint32_t>(SRC_CAP), lf.y);
                    }
                    n_src = SRC_CAP;
                    wave_sync();
                    flush(true);
                    continue;
                }
                const unsigned mycnt = fits ? cnt : 0u;
                const int dst = n_src + static_cast<int>(incl - cnt);
                const int total = __builtin_amdgcn_readlane(static_cast<int>(incl), m - 1);
                RK_COUNT(10, m)
#if RK_LEAF_LANE_PARTICLE
                // Lane = particle. First every fitting leaf writes the Morton indices of its particles into the first word of
                // the tile slots they will occupy (lane = leaf, one 4-byte LDS store per particle), then lane i fetches the
                // record slot i names and stores it over the index: one record in flight per lane and 64 per wavefront, whatever
                // the sizes of the leaves -- four registers where eight records per LEAF used to be held (32: the register
                // peak of list building). Same records in the same slots as before.
                {
                    constexpr int W = static_cast<int>(sizeof(v4) / sizeof(uint32_t));
                    uint32_t *slot_word = reinterpret_cast<uint32_t *>(&L.src[0]);
                    for (unsigned k = 0; __builtin_amdgcn_ballot_w64(k < mycnt) != 0ull; ++k) {
                        if (k < mycnt) {
                            slot_word[(dst + static_cast<int>(k)) * W] = lf.x + k;
                        }
                    }
                    wave_sync();
                    RK_COUNT(9, (total + 63) / 64)
                    for (int i0 = 0; i0 < total; i0 += 64) {
                        const int i = i0 + lane;
                        if (i < total) {
                            const uint32_t pi = slot_word[(n_src + i) * W];
                            L.src[n_src + i] = P.part4[pi];
                        }
                    }
                }
#else
                // Lane l copies the particles of leaf l, eight loads in flight at a time.
                for (unsigned j0 = 0; __builtin_amdgcn_ballot_w64(j0 < mycnt) != 0ull; j0 += 8u) {
                    RK_COUNT(9, 1)
                    v4 tmp[8];
                    // Unconditional loads (index clamped into the leaf; particle 0 for idle lanes).
                    const uint32_t lbase = mycnt ? lf.x : 0u, llast = mycnt ? mycnt - 1u : 0u;
#pragma unroll
                    for (unsigned u = 0; u < 8u; ++u) {
                        const uint32_t jj = j0 + u < llast ? j0 + u : llast;
                        tmp[u] = P.part4[lbase + jj];
                    }
#pragma unroll
                    for (unsigned u = 0; u < 8u; ++u) {
                        if (j0 + u < mycnt) {
                            L.src[dst + static_cast<int>(j0 + u)] = tmp[u];
                        }
                    }
                }
#endif
                n_src += total;
                RK_COUNT(11, total)
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
                RK_STAMP(3)
                if (n_src + 64 > SRC_CAP) {
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
        // Pop up to 8 sibling runs and issue the loads of their records. `pending` = number of entries that
        // batches already in flight may still push. Returns the number of entries popped.
        auto pop_and_load = [&](batch_t &bt, int pending, bool allow_dfs) __attribute__((always_inline)) -> int {
            if (size == 0) {
                return 0;
            }
            int k = size < 8 ? size : 8;
            // Keep the stack within bounds even if every candidate is opened (8 pushes per popped entry);
            // otherwise fall back to one entry at a time (depth-first), whose growth is bounded by
            // LK_DFS_RESERVE -- only when nothing else is in flight.
            const int room = (LK_STACK_CAP - LK_DFS_RESERVE - pending - n_uq - size) / 7;
            if (room < k) {
                if (room >= 1) {
                    k = room;
                } else if (allow_dfs) {
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
            // Everything about the candidate in three independent 16-byte loads (record 0 for idle lanes).
            bt.rec = bt.active ? (entry >> 3) + static_cast<uint32_t>(e_sub) : 0u;
            const node_rec<F> *rec = P.node_rec + bt.rec;
            bt.com = rec->com;
            bt.mp = rec->mac;
            bt.node = rec->dfs;
            bt.nch = rec->nch;
            bt.ra = rec->a;
            bt.rb = rec->b;
            return k;
        };

        // Route classified candidates: accepted nodes go to the source tile, opened leaves to the leaf queue,
        // opened internal nodes push their run of children, undecided ones go to the exact-test queue.
        auto route = [&](bool accept, bool open, bool undecided, const batch_t &bt) __attribute__((always_inline)) {
            const bool leaf = open && bt.nch == 0u;
            const bool expand = open && bt.nch != 0u;
            const unsigned long long m_acc = __builtin_amdgcn_ballot_w64(accept);
            if (accept) {
                L.src[n_src + static_cast<int>(wave_prefix_count(m_acc))] = bt.com;
            }
            n_src += __builtin_popcountll(m_acc);
            RK_COUNT(23, __builtin_popcountll(m_acc))
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
            RK_COUNT(24, __builtin_popcountll(m_exp))
            const unsigned long long m_und = __builtin_amdgcn_ballot_w64(undecided);
            if (undecided) {
                L.uq[n_uq + static_cast<int>(wave_prefix_count(m_und))] = bt.rec;
            }
            n_uq += __builtin_popcountll(m_und);
            RK_COUNT(7, __builtin_popcountll(m_und))
            wave_sync();
        };

        // First-stage MAC test of one batch. The reference's criterion is "every target t of the group has
        // d2(t) > mac_lh" with d2(t) the squared distance from target t to the node's centre of mass
        // (tree.hpp:2662-2672). Two cheap tests reproduce that decision for most candidates:
        //  * accept if the squared distance from the centre of mass to the group's bounding box exceeds mac_lh
        //    by a margin (1e-5 relative, far above the ~1e-6 rounding of either quantity): every d2(t) is larger;
        //  * open if one of two probe targets already violates the criterion (same formula as the exact test).
        // The rest is queued for the exact all-targets loop, so every decision equals the reference's.
        auto process = [&](const batch_t &bt) __attribute__((always_inline)) {
            const v4 com = bt.com;
            // Ancestor-or-self of the target group (tree.hpp:2828-2838 of the reference) on the depth-first
            // index interval of the subtree.
            const bool anc = bt.active && bt.node <= cnode && cnode <= bt.node + bt.nch;
            const bool self = anc && bt.node == cnode;
            const bool test = bt.active && !anc;
            RK_COUNT(1, 1) RK_COUNT(2, __builtin_popcountll(__builtin_amdgcn_ballot_w64(bt.active)))
            const F mac_lh = mac_lhs<F>(MAC == RK_MAC_RT ? P.mac : MAC, bt.mp, mac_value);
#ifdef RK_STAMPS
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            st_acc[6] += 1;
#endif
            RK_STAMP(0)
            const F bx = rk_max3(blo.x - com.x, com.x - bhi.x, F(0)), by = rk_max3(blo.y - com.y, com.y - bhi.y, F(0)),
                    bz = rk_max3(blo.z - com.z, com.z - bhi.z, F(0));
            const F dbox2 = rk_fma(bz, bz, rk_fma(by, by, bx * bx));
            const bool box_accept = dbox2 > mac_lh * F(1.00001);
            const F p0x = com.x - pr0.x, p0y = com.y - pr0.y, p0z = com.z - pr0.z;
            const F p1x = com.x - pr1.x, p1y = com.y - pr1.y, p1z = com.z - pr1.z;
            const F d2p0 = rk_fma(p0z, p0z, rk_fma(p0y, p0y, p0x * p0x)), d2p1 = rk_fma(p1z, p1z, rk_fma(p1y, p1y, p1x * p1x));
            const bool probe_open = mac_lh >= rk_min(d2p0, d2p1);
            RK_STAMP(1)
#ifdef RK_ABLATE_EXACT
            const bool accept = test && (box_accept || !probe_open); // diagnostic build: no exact all-targets test
            const bool open = (test && !box_accept && probe_open) || (anc && !self);
            const bool undecided = false;
#else
            const bool accept = test && box_accept;
            const bool open = (test && !box_accept && probe_open) || (anc && !self);
            const bool undecided = test && !box_accept && !probe_open;
#endif
            route(accept, open, undecided, bt);
            RK_STAMP(2)
        };

        // Exact MAC test (all targets) of up to 64 queued candidates.
        auto process_exact = [&]() __attribute__((always_inline)) {
            const int k = n_uq < 64 ? n_uq : 64;
            batch_t bt;
            bt.active = lane < k;
            bt.rec = bt.active ? L.uq[n_uq - 1 - lane] : 0u;
            n_uq -= k;
            const node_rec<F> *rec = P.node_rec + bt.rec;
            bt.com = rec->com;
            bt.mp = rec->mac;
            bt.node = rec->dfs;
            bt.nch = rec->nch;
            bt.ra = rec->a;
            bt.rb = rec->b;
            const v4 com = bt.com;
            const F mac_lh = mac_lhs<F>(MAC == RK_MAC_RT ? P.mac : MAC, bt.mp, mac_value);
            bool fail;
#if RK_EXACT_TRANSPOSED
            if (!BIG && k * (7 * R + 3) < T * 7) {
                RK_COUNT(5, 1) RK_COUNT(6, k)
                // Few candidates: lane = target. Every lane already keeps R targets of the group in registers (unused
                // slots repeat target 0), so a candidate costs one broadcast LDS read and 7 R + 3 instructions instead of
                // a share of the 7 T of the loop below. Same formula, same operands: same decision.
                // The main loop keeps 64 free slots behind n_src in the source tile; the candidates are staged there.
                v4 cd;
                cd.x = com.x, cd.y = com.y, cd.z = com.z, cd.w = mac_lh;
                if (bt.active) {
                    L.src[n_src + lane] = cd;
                }
                wave_sync();
                unsigned long long fail_mask = 0ull;
                for (int c = 0; c < k; ++c) {
                    const v4 cand = L.src[n_src + c];
                    bool f = false;
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const F dx = cand.x - tg.tx(r), dy = cand.y - tg.ty(r), dz = cand.z - tg.tz(r);
                        const F d2 = rk_fma(dz, dz, rk_fma(dy, dy, dx * dx));
                        f |= cand.w >= d2;
                    }
                    if (__builtin_amdgcn_ballot_w64(f) != 0ull) {
                        fail_mask |= 1ull << c;
                    }
                }
                fail = ((fail_mask >> lane) & 1ull) != 0ull;
                wave_sync();
            } else
#endif
            {
                // min over the targets of the unsoftened squared distance to the node's centre of mass. The target
                // coordinates are wave-uniform: they arrive through scalar loads as SGPR operands.
                F mind2 = std::numeric_limits<F>::infinity();
                RK_COUNT(3, 1) RK_COUNT(4, TG) RK_COUNT(25, k)
                for (int t = 0; t < TG; t += 4) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int ti = (t + u < TG) ? t + u : TG - 1;
                        const v4 tg = P.part4[gb + static_cast<uint32_t>(ti)];
                        const F dx = com.x - tg.x, dy = com.y - tg.y, dz = com.z - tg.z;
                        const F d2 = rk_fma(dz, dz, rk_fma(dy, dy, dx * dx));
                        mind2 = rk_min(mind2, d2);
                    }
                }
                fail = mac_lh >= mind2;
            }
            route(bt.active && !fail, bt.active && fail, false, bt);
            RK_STAMP(3)
        };

        // ---- list building ----
        // Sources accepted for the whole supergroup: stream them from the pre-pass list through the tile.
        {
            const v4 *common = P.sup_common + static_cast<size_t>(sup_S) * SUP_CAPC;
#ifdef RK_ABLATE_COMMON
            sup_ncommon = 0; // diagnostic build: the supergroup's common sources are dropped
#endif
            for (uint32_t base = 0; base < sup_ncommon;) {
                const uint32_t room = static_cast<uint32_t>(SRC_CAP - n_src), left = sup_ncommon - base;
                const uint32_t take = left < room ? left : room;
                RK_COUNT(16, (take + 63u) / 64u) RK_COUNT(17, take)
                for (uint32_t j = lane; j < take; j += 64u) {
                    L.src[n_src + static_cast<int>(j)] = common[base + j];
                }
                n_src += static_cast<int>(take);
                base += take;
                wave_sync();
                if (n_src == SRC_CAP) {
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
            const node_rec<F> *rec = P.node_rec + bt.rec;
            bt.com = rec->com;
            bt.mp = rec->mac;
            bt.node = rec->dfs;
            bt.nch = rec->nch;
            bt.ra = rec->a;
            bt.rb = rec->b;
            return 1;
        };
        auto next_batch = [&](batch_t &bt) __attribute__((always_inline)) -> int {
            const int k = pop_and_load(bt, 0, true);
            return k ? k : resid_load(bt);
        };
        // ---- list building ----
        // Queues are settled BEFORE the next batch of records is fetched, so that no candidate registers are live
        // across the dense phase (register pressure decides the occupancy of this kernel).
        bool done = false;
#ifdef RK_ABLATE_LIST
        size = 0, sup_nresid = 0; // diagnostic build: no list building at all (prologue, own particles, reduction, epilogue remain)
#endif
        for (;;) {
            // Room for the worst-case output of one pass (64 sources, 64 leaves); everything is settled at the end.
            if (n_lq + 64 > LK_LQ_CAP || done) {
                drain_leaves();
            }
            if (n_src + 64 > SRC_CAP || done) {
                flush(done);
            }
            if (done) {
                break;
            }
            if (n_uq >= 64) {
                process_exact();
                continue;
            }
            // Near the stack bound pop_and_load() descends one entry at a time, and LK_DFS_RESERVE (7 pending entries per
            // level) only bounds a STRICT depth-first descent: settle the parked candidates first, so that everything an
            // entry can push is on the stack before the next entry is popped.
            if (n_uq > 0 && LK_STACK_CAP - LK_DFS_RESERVE - n_uq - size < 7) {
                process_exact();
                continue;
            }
            RK_STAMP(7)
            batch_t A;
            if (next_batch(A) == 0) {
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

        RK_STAMP(7)
        // ---- interactions inside the group: its own particles as sources, self-pair masked ----
        for (int b0 = 0; b0 < TG; b0 += SRC_CAP) {
            const int n = (TG - b0) < SRC_CAP ? (TG - b0) : SRC_CAP;
            for (int j = lane; j < n; j += 64) {
                L.src[j] = P.part4[gb + static_cast<uint32_t>(b0 + j)];
            }
            wave_sync();
            int tloc[R];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                tloc[r] = tidx[r] < 0 ? -1 : static_cast<int>(tb - gb) + tidx[r] - b0;
            }
            RK_COUNT(18, 1) RK_COUNT(19, (n * inv_ns) >> 16) RK_COUNT(27, n - ((n * inv_ns) >> 16) * NS > 0 ? 1 : 0)
            lk_eval_tile<F, Q, R, true, ND>(L.src, n, (n * inv_ns) >> 16, sp, NS, true, lane_on, tg, eps2, tloc);
            wave_sync();
        }

        RK_STAMP(5)
        // ---- sum the source splits in a fixed order, scale by G, write out ----
        const F G = P.G;
        if (NS > 1) {
            // One target slot r at a time: the scratch then needs 64 * NR values (<= 2 KiB), well inside
            // this wave's LDS region for every F, Q and R.
            F *red = reinterpret_cast<F *>(&L);
#pragma unroll
            for (int r = 0; r < R; ++r) {
                if (lane_on) {
#pragma unroll
                    for (int k = 0; k < NR; ++k) {
                        red[(sp_raw * TP + ts) * NR + k] = tg.get(r, k);
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
                        tg.set(r, k, sum);
                    }
                }
                wave_sync();
            }
        }
        if (lane_on && sp_raw == 0) {
            // Nothing but stores once the output positions are known. (With the look-up of the ordered form inside the loop,
            // rounds 1-4, every target's stores waited -- s_waitcnt vmcnt(0) -- for the stores of the target before: a round trip
            // to the caller's HOST array per target when the results cross PCIe, 0.1 ms of the 4M seam call.)
            auto store_all = [&](auto pos) __attribute__((always_inline)) {
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    if (tidx[r] >= 0) {
                        const uint32_t o = pos(r);
#pragma unroll
                        for (int k = 0; k < NR; ++k) {
                            if (ND == 3 || Q == 1 || k != 2) { // a quadtree has no z acceleration (and no array for it)
                                P.out[k][o] = tg.get(r, k) * G;
                            }
                        }
                    }
                }
            };
            if (P.perm) {
                // Original-order output: all look-ups through the permutation first, one wait, then the stores.
                uint32_t o[R];
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    o[r] = P.perm[tb + static_cast<uint32_t>(tidx[r] >= 0 ? tidx[r] : 0)];
                }
                __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0)
                store_all([&](int r) { return o[r]; });
            } else {
                store_all([&](int r) { return tb + static_cast<uint32_t>(tidx[r]) - P.out_sub; });
            }
        }
        RK_STAMP(7)
        RK_STAMP_FLUSH
        RK_COUNT_FLUSH
#ifdef RK_TRACE
        if (!BIG && lane == 0 && P.dbg) {
            const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20);
            P.dbg[4u * g] = tr_t0;
            P.dbg[4u * g + 1u] = __builtin_amdgcn_s_memrealtime();
            P.dbg[4u * g + 2u] = (static_cast<unsigned long long>(xcc) << 32) | hw;
            // {R, T, shader cycles of this wave}: cycles / duration = the clock the chip held meanwhile.
            P.dbg[4u * g + 3u] = (static_cast<unsigned long long>(R) << 56) | (static_cast<unsigned long long>(T) << 32)
                                 | ((__builtin_amdgcn_s_memtime() - tr_c0) & 0xffffffffull);
        }
#endif
    }; // run_targets
    if constexpr (BIG) {
        const int n_chunks = (TG + LK_BIG_CHUNK - 1) / LK_BIG_CHUNK;
        const int chunk_len = (TG + n_chunks - 1) / n_chunks;
        for (int chunk = wib; chunk < n_chunks; chunk += LK_BIG_WPB) {
            const uint32_t tb = gb + static_cast<uint32_t>(chunk * chunk_len);
            run_targets(tb, tb + static_cast<uint32_t>(chunk_len) < ge ? tb + static_cast<uint32_t>(chunk_len) : ge);
            wave_sync(); // the next chunk reuses this wave's LDS region
        }
    } else {
        run_targets(gb, ge);
    }
}

template <typename F, int Q, int MAC, int R, int ND, bool BIG = false>
__global__ void __launch_bounds__(64 * (BIG ? LK_BIG_WPB : RK_WPB), BIG ? (sizeof(F) == 4 ? RK_WBIG : RK_W64_ANY) : (sizeof(F) == 4 ? (R <= 2 ? RK_W12 : (R == 3 ? RK_W3 : (R == 4 ? RK_W4 : (R == 5 ? RK_W5 : RK_W6)))) : (R >= 4 ? RK_W64_R4 : (R == 3 ? RK_W64_R3 : RK_W64)))) k_list(const kparams<F> P, const uint32_t *__restrict__ list, int n_list, const uint32_t *__restrict__ n_list_dev = nullptr)
{
    __shared__ lk_wave_lds<F> s_lds[BIG ? LK_BIG_WPB : RK_WPB];

    const int wib = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63;
    // BIG: one workgroup per critical node too large for one wavefront (more than 64 * RK_MAX_R particles: ncrit or
    // max_leaf_n above 256, or a tree that ran out of levels). Its targets are cut into chunks of at most
    // LK_BIG_CHUNK, every wavefront of the workgroup serves chunks wib, wib + LK_BIG_WPB, ... exactly like a group of
    // its own -- except that every MAC decision is taken for ALL particles of the critical node (bounding box, probes
    // and the exact test range over the node, not the chunk), so the interaction set is the reference's.
    const unsigned blk = BIG ? blockIdx.x : xcd_map_block(blockIdx.x, gridDim.x, P.xcd_mode);
    const int wave = __builtin_amdgcn_readfirstlane(BIG ? static_cast<int>(blk) : static_cast<int>(blk * unsigned(RK_WPB)) + wib);
    if (BIG && n_list_dev) {
        // The fallback list of the split traversal: its length is only known on the device.
        n_list = static_cast<int>(__builtin_amdgcn_readfirstlane(*n_list_dev));
    }
    if (wave >= n_list) {
        return;
    }
    uint32_t entry = static_cast<uint32_t>(wave);
    if (!BIG && RK_WPB == 1 && P.xcd_mode == 3) {
        // First call on a large tree: block i serves entry i / 8 of the queue of (this kernel's class, XCD region i % 8) -- the
        // blocks are dealt round-robin to the XCDs, so a region's queue runs on one XCD in every class kernel.
        const uint32_t x = blk & 7u, pos = blk >> 3;
        const uint32_t *tab = P.first_tab + (R - 1) * 16;
        if (pos >= tab[8u + x]) {
            return; // past the end of this region's queue
        }
        entry = tab[x] + pos;
    }
    const uint32_t g = __builtin_amdgcn_readfirstlane(list[entry]);
    if (g == RK_PLAN_PAD_VALUE) {
        return; // padding of a launch plan
    }
    list_node<F, Q, MAC, R, ND, BIG>(P, s_lds[wib], g, lane, wib);
}

// One launch for the critical nodes of ALL lane-mapping classes (calls over few critical nodes, whose four class kernels
// would otherwise start 25-45 us apart and leave the device to drain over the class launched last): every wavefront
// picks the R of its node and runs the same list_node<..., R> the class kernels run. Compiled for the registers of the
// largest R; the launch is too small to fill the device anyway.
template <typename F, int Q, int MAC, int ND>
__global__ void __launch_bounds__(64, sizeof(F) == 4 ? RK_WANY : RK_W64_ANY) k_list_any(const kparams<F> P, const uint32_t *__restrict__ list, int n_list)
{
    __shared__ lk_wave_lds<F> s_lds;
    const int lane = threadIdx.x;
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
    static_assert(RK_MAX_R == 4, "k_list_any dispatches R = 1..4");
    switch (class2_of_compute(static_cast<int64_t>(__builtin_amdgcn_readfirstlane(P.crit[g].w)))) {
        case 0: list_node<F, Q, MAC, 1, ND, false>(P, s_lds, g, lane, 0); break;
        case 1: list_node<F, Q, MAC, 2, ND, false>(P, s_lds, g, lane, 0); break;
        case 2: list_node<F, Q, MAC, 3, ND, false>(P, s_lds, g, lane, 0); break;
        case 3: list_node<F, Q, MAC, 4, ND, false>(P, s_lds, g, lane, 0); break;
        default: break; // oversized nodes are not on this list
    }
}

// ------------------------------------------------------------------------------------------------
// Supergroup pre-pass. K consecutive target groups (a "supergroup" S) visit almost the same upper part of the
// tree. One wavefront per S walks it once with tests that are valid for EVERY member group:
//   * a node whose squared distance to the union of the members' bounding boxes exceeds mac_lh (with the 1e-5
//     margin) passes the reference's criterion for every particle of every member  -> common source list;
//   * a node whose FARTHEST box corner is still within mac_lh fails it for every particle -> opened here;
//   * common strict ancestors of all members are opened here;
//   * everything else (including nodes that contain some member) is left to the members: residual list.
// Each member then streams the common list through its dense phase and starts its own list building from the
// residual list. The per-group decisions, hence the interaction lists, are unchanged.
// ------------------------------------------------------------------------------------------------
constexpr int SUP_STACK_CAP = 512;

template <typename F, int MAC>
__global__ void __launch_bounds__(256) k_super(const kparams<F> P, uint32_t s_begin, uint32_t s_end)
{
    using v4 = typename vt<F>::v4;
    using v2 = typename vt<F>::v2;
    __shared__ uint32_t s_stack[4][SUP_STACK_CAP];
    const int wib = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63;
    const uint32_t S = __builtin_amdgcn_readfirstlane(s_begin + blockIdx.x * 4u + static_cast<uint32_t>(wib));
    if (S >= s_end) {
        return;
    }
    uint32_t *stack = s_stack[wib];
    const uint32_t K = P.super_k;
    const uint32_t g0 = S * K, g1 = (g0 + K < P.n_crit) ? g0 + K : P.n_crit;
    // Union of the members' boxes (wave-uniform after the reduction) and the depth-first span of the members.
    F lo[3], hi[3];
    {
        const uint32_t gl = g0 + static_cast<uint32_t>(lane) < g1 ? g0 + static_cast<uint32_t>(lane) : g1 - 1u;
        const v4 a = P.crit_box[2u * gl], b = P.crit_box[2u * gl + 1u];
        lo[0] = a.x, lo[1] = a.y, lo[2] = a.z, hi[0] = b.x, hi[1] = b.y, hi[2] = b.z;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                lo[k] = rk_min(lo[k], __shfl_xor(lo[k], d, 64));
                hi[k] = -rk_min(-hi[k], -__shfl_xor(hi[k], d, 64));
            }
        }
    }
    const uint32_t c_first = P.crit[g0].z, c_last = P.crit[g1 - 1u].z;
    const uint32_t c_last_end = c_last + P.node_topo[c_last].x;
    const F mac_value = P.mac_value;
    v4 *common = P.sup_common + static_cast<size_t>(S) * SUP_CAPC;
    uint32_t *resid = P.sup_resid + static_cast<size_t>(S) * SUP_CAPR;
    uint32_t n_common = 0, n_resid = 0;
    bool overflow = false;
    int size = 0;
    {
        const node_rec<F> *root = P.node_rec;
        const uint32_t r_nch = root->nch, r_a = root->a, r_b = root->b;
        // c_first == 0: the root itself is the (only) target group; nothing to traverse.
        if (c_first != 0u && r_nch != 0u) {
            if (lane == 0) {
                stack[0] = (r_a << 3) | (r_b - 1u);
            }
            size = 1;
        }
    }
    wave_sync();
    while (size > 0) {
        if (n_common + 64u > SUP_CAPC || n_resid + 64u > SUP_CAPR) {
            overflow = true;
            break;
        }
        int k = size < 8 ? size : 8;
        const int room = (SUP_STACK_CAP - LK_DFS_RESERVE - size) / 7;
        if (room < k) {
            k = room > 1 ? room : 1;
        }
        const int e_idx = lane >> 3, e_sub = lane & 7;
        uint32_t entry = 0u;
        if (e_idx < k) {
            entry = stack[size - 1 - e_idx];
        }
        size -= k;
        const bool active = e_idx < k && static_cast<uint32_t>(e_sub) <= (entry & 7u);
        const uint32_t recidx = active ? (entry >> 3) + static_cast<uint32_t>(e_sub) : 0u;
        const node_rec<F> *rec = P.node_rec + recidx;
        const v4 com = rec->com;
        const v2 mp = rec->mac;
        const uint32_t dfs = rec->dfs, nch = rec->nch, ra = rec->a, rb = rec->b;
        const F mac_lh = mac_lhs<F>(MAC == RK_MAC_RT ? P.mac : MAC, mp, mac_value);
        // Relation of the node's subtree [dfs, dfs + nch] to the members' span [c_first, c_last_end].
        const bool touches = dfs <= c_last_end && dfs + nch >= c_first;
        const bool common_anc = dfs < c_first && dfs + nch >= c_last_end;
        const F nx = rk_max3(lo[0] - com.x, com.x - hi[0], F(0)), ny = rk_max3(lo[1] - com.y, com.y - hi[1], F(0)),
                nz = rk_max3(lo[2] - com.z, com.z - hi[2], F(0));
        const F dnear2 = rk_fma(nz, nz, rk_fma(ny, ny, nx * nx));
        const F fx = -rk_min(-(com.x - lo[0]), -(hi[0] - com.x)), fy = -rk_min(-(com.y - lo[1]), -(hi[1] - com.y)),
                fz = -rk_min(-(com.z - lo[2]), -(hi[2] - com.z)); // distance to the farthest face per axis
        const F dfar2 = rk_fma(fz, fz, rk_fma(fy, fy, fx * fx));
        const bool acc_all = active && !touches && dnear2 > mac_lh * F(1.00001);
        const bool open_all = active && ((touches && common_anc) || (!touches && !acc_all && dfar2 * F(1.00001) <= mac_lh))
                              && nch != 0u;
        const bool to_resid = active && !acc_all && !open_all;
        const unsigned long long m_acc = __builtin_amdgcn_ballot_w64(acc_all);
        if (acc_all) {
            common[n_common + wave_prefix_count(m_acc)] = com;
        }
        n_common += static_cast<uint32_t>(__builtin_popcountll(m_acc));
        const unsigned long long m_res = __builtin_amdgcn_ballot_w64(to_resid);
        if (to_resid) {
            resid[n_resid + wave_prefix_count(m_res)] = recidx;
        }
        n_resid += static_cast<uint32_t>(__builtin_popcountll(m_res));
        const unsigned long long m_exp = __builtin_amdgcn_ballot_w64(open_all);
        if (open_all) {
            stack[size + static_cast<int>(wave_prefix_count(m_exp))] = (ra << 3) | (rb - 1u);
        }
        size += __builtin_popcountll(m_exp);
        wave_sync();
    }
    if (lane == 0) {
        P.sup_cnt[S] = make_uint2(n_common, n_resid | (overflow ? 0x80000000u : 0u));
    }
}

template <typename F>
void launch_super(const rk_state &s, const kparams<F> &p, int64_t s_begin, int64_t s_end, hipStream_t stream)
{
    const int64_t n = s_end - s_begin;
    if (n <= 0 || !p.super_k) {
        return;
    }
    const auto grid = static_cast<unsigned>((n + 3) / 4);
    if (RK_MAC_RUNTIME || s.mac == RK_MAC_BH) {
        hipLaunchKernelGGL((k_super<F, mac_targ(0)>), dim3(grid), dim3(256), 0, stream, p, static_cast<uint32_t>(s_begin),
                           static_cast<uint32_t>(s_end));
    } else {
        hipLaunchKernelGGL((k_super<F, mac_targ(1)>), dim3(grid), dim3(256), 0, stream, p, static_cast<uint32_t>(s_begin),
                           static_cast<uint32_t>(s_end));
    }
    RK_HIP(hipGetLastError());
}
template void launch_super<float>(const rk_state &, const kparams<float> &, int64_t, int64_t, hipStream_t);
#ifndef RK_SLIM
template void launch_super<double>(const rk_state &, const kparams<double> &, int64_t, int64_t, hipStream_t);
#endif

// ------------------------------------------------------------------------------------------------
// Launch.
// ------------------------------------------------------------------------------------------------
template <typename F, int Q, int MAC>
static void launch_list_qm(const rk_state &s, const kparams<F> &p, const int64_t cb[n_classes],
                           const int64_t ce[n_classes], hipStream_t const streams[n_list_R], unsigned class_mask)
{
    const auto *lists = s.cur_lists; // class lists of the state, or the launch plan of this call
    auto go = [&](auto Rtag, int c) {
        constexpr int R = decltype(Rtag)::value;
        const int64_t n = ce[c] - cb[c];
        if (n <= 0 || !((class_mask >> c) & 1u)) {
            return;
        }
        const auto grid = static_cast<unsigned>((n + RK_WPB - 1) / RK_WPB);
        if (s.ndim == 3 || !RK_QUAD_BODY) {
            hipLaunchKernelGGL((k_list<F, Q, MAC, R, 3>), dim3(grid), dim3(64 * RK_WPB), 0, streams[c], p,
                               lists + s.cur_off[c] + cb[c], static_cast<int>(n), nullptr);
        } else if constexpr (R <= RK_MAX_R) {
            // Quadtrees: the same kernel without the z terms of the interaction (10 instead of 13 operations).
            hipLaunchKernelGGL((k_list<F, Q, MAC, R, 2>), dim3(grid), dim3(64 * RK_WPB), 0, streams[c], p,
                               lists + s.cur_off[c] + cb[c], static_cast<int>(n), nullptr);
        } else {
            throw error(RK_ERUNTIME, "internal error: quadtree group in a lane-mapping class beyond RK_MAX_R");
        }
    };
#if RK_MAX_R >= 5
    go(std::integral_constant<int, 5>{}, 4); // (R = 5, 6: measured slower, only built when RK_MAX_R asks for them)
#endif
    // Order in which the class kernels are handed to the runtime. Replayed from a graph they share one queue, and the kernel
    // handed over first is served first as slots free up: the R = 3 kernel, which ends last, must lead (round 5, 4M device-resident
    // kernel ms over two boxes: 3-1-2-4 (rounds 1-4) 2.15-2.17, 3-4-1-2 2.14-2.16, 3-2-4-1 2.15-2.16, 3-4-2-1 / 3-1-4-2 2.17;
    // with R = 1 or R = 4 first -- 1-2-3-4, 4-3-2-1, 4-3-1-2, 1-3-4-2 -- 2.21-2.24: tools/archive/jobs_r05/r05_job27.sh, _job28).
    go(std::integral_constant<int, 3>{}, 2);
    go(std::integral_constant<int, 4>{}, 3);
    go(std::integral_constant<int, 1>{}, 0);
    go(std::integral_constant<int, 2>{}, 1);
#if RK_MAX_R >= 6
    go(std::integral_constant<int, 6>{}, 5);
#endif
    for (int c = RK_MAX_R; c < big_class; ++c) {
        if (ce[c] != cb[c] && ((class_mask >> c) & 1u)) {
            throw error(RK_ERUNTIME, "internal error: target group in a lane-mapping class beyond RK_MAX_R");
        }
    }
}

template <typename F>
void launch_list(const rk_state &s, int q, const kparams<F> &p, const int64_t cb[n_classes],
                 const int64_t ce[n_classes], hipStream_t const streams[n_list_R], unsigned class_mask)
{
    switch (q * 2 + s.mac) {
        case 0: launch_list_qm<F, 0, mac_targ(0)>(s, p, cb, ce, streams, class_mask); break;
#ifndef RK_SLIM // (RK_SLIM: device-code inspection builds with one instantiation per kernel; never linked)
        case 1: launch_list_qm<F, 0, mac_targ(1)>(s, p, cb, ce, streams, class_mask); break;
        case 2: launch_list_qm<F, 1, mac_targ(0)>(s, p, cb, ce, streams, class_mask); break;
        case 3: launch_list_qm<F, 1, mac_targ(1)>(s, p, cb, ce, streams, class_mask); break;
        case 4: launch_list_qm<F, 2, mac_targ(0)>(s, p, cb, ce, streams, class_mask); break;
        case 5: launch_list_qm<F, 2, mac_targ(1)>(s, p, cb, ce, streams, class_mask); break;
#endif
        default: throw error(RK_EINVAL, "invalid q / mac combination");
    }
    RK_HIP(hipGetLastError());
}

template <typename F>
const void *list_kernel_symbol(const rk_state &s, int q, int c)
{
    auto pick = [&](auto Qt, auto Mt) -> const void * {
        constexpr int Q = decltype(Qt)::value, M = mac_targ(decltype(Mt)::value);
        auto of_nd = [&](auto NDt) -> const void * {
            constexpr int ND = decltype(NDt)::value;
            switch (c) {
                case 0: return reinterpret_cast<const void *>(&k_list<F, Q, M, 1, ND>);
                case 1: return reinterpret_cast<const void *>(&k_list<F, Q, M, 2, ND>);
                case 2: return reinterpret_cast<const void *>(&k_list<F, Q, M, 3, ND>);
                case 3: return reinterpret_cast<const void *>(&k_list<F, Q, M, 4, ND>);
                default: return nullptr;
            }
        };
        if (s.ndim == 3 || !RK_QUAD_BODY) {
            return of_nd(std::integral_constant<int, 3>{});
        }
        return of_nd(std::integral_constant<int, 2>{});
    };
    using i0 = std::integral_constant<int, 0>;
    using i1 = std::integral_constant<int, 1>;
    using i2 = std::integral_constant<int, 2>;
    switch (q * 2 + s.mac) {
        case 0: return pick(i0{}, i0{});
#ifndef RK_SLIM
        case 1: return pick(i0{}, i1{});
        case 2: return pick(i1{}, i0{});
        case 3: return pick(i1{}, i1{});
        case 4: return pick(i2{}, i0{});
        case 5: return pick(i2{}, i1{});
#endif
        default: return nullptr;
    }
}
template const void *list_kernel_symbol<float>(const rk_state &, int, int);
#ifndef RK_SLIM
template const void *list_kernel_symbol<double>(const rk_state &, int, int);
#endif

// One launch over a list of critical nodes of any lane-mapping class (k_list_any).
template <typename F>
void launch_list_any(const rk_state &s, int q, const kparams<F> &p, const uint32_t *list, int64_t n, hipStream_t stream)
{
    if (n <= 0) {
        return;
    }
    const dim3 grid(static_cast<unsigned>(n)), block(64);
    const int cnt = static_cast<int>(n);
    auto go = [&](auto Qt, auto Mt) {
        constexpr int Q = decltype(Qt)::value, M = mac_targ(decltype(Mt)::value);
        if (s.ndim == 3 || !RK_QUAD_BODY) {
            hipLaunchKernelGGL((k_list_any<F, Q, M, 3>), grid, block, 0, stream, p, list, cnt);
        } else {
            hipLaunchKernelGGL((k_list_any<F, Q, M, 2>), grid, block, 0, stream, p, list, cnt);
        }
    };
    using i0 = std::integral_constant<int, 0>;
    using i1 = std::integral_constant<int, 1>;
    using i2 = std::integral_constant<int, 2>;
    switch (q * 2 + s.mac) {
        case 0: go(i0{}, i0{}); break;
#ifndef RK_SLIM
        case 1: go(i0{}, i1{}); break;
        case 2: go(i1{}, i0{}); break;
        case 3: go(i1{}, i1{}); break;
        case 4: go(i2{}, i0{}); break;
        case 5: go(i2{}, i1{}); break;
#endif
        default: throw error(RK_EINVAL, "invalid q / mac combination");
    }
    RK_HIP(hipGetLastError());
}
template void launch_list_any<float>(const rk_state &, int, const kparams<float> &, const uint32_t *, int64_t, hipStream_t);
#ifndef RK_SLIM
template void launch_list_any<double>(const rk_state &, int, const kparams<double> &, const uint32_t *, int64_t, hipStream_t);
#endif

// Critical nodes of more than 64 * RK_MAX_R particles: k_list<..., BIG> (one workgroup per node).
template <typename F>
void launch_list_big(const rk_state &s, int q, const kparams<F> &p, const uint32_t *list, int64_t n, hipStream_t stream,
                     const uint32_t *n_dev)
{
    if (n <= 0) {
        return;
    }
    const dim3 grid(static_cast<unsigned>(n)), block(64 * LK_BIG_WPB);
    const int cnt = static_cast<int>(n);
    auto go = [&](auto Qt, auto Mt) {
        constexpr int Q = decltype(Qt)::value, M = mac_targ(decltype(Mt)::value);
        if (s.ndim == 3 || !RK_QUAD_BODY) {
            hipLaunchKernelGGL((k_list<F, Q, M, 2, 3, true>), grid, block, 0, stream, p, list, cnt, n_dev);
        } else {
            hipLaunchKernelGGL((k_list<F, Q, M, 2, 2, true>), grid, block, 0, stream, p, list, cnt, n_dev);
        }
    };
    using i0 = std::integral_constant<int, 0>;
    using i1 = std::integral_constant<int, 1>;
    using i2 = std::integral_constant<int, 2>;
    switch (q * 2 + s.mac) {
        case 0: go(i0{}, i0{}); break;
#ifndef RK_SLIM
        case 1: go(i0{}, i1{}); break;
        case 2: go(i1{}, i0{}); break;
        case 3: go(i1{}, i1{}); break;
        case 4: go(i2{}, i0{}); break;
        case 5: go(i2{}, i1{}); break;
#endif
        default: throw error(RK_EINVAL, "invalid q / mac combination");
    }
    RK_HIP(hipGetLastError());
}
template void launch_list_big<float>(const rk_state &, int, const kparams<float> &, const uint32_t *, int64_t, hipStream_t,
                                     const uint32_t *);
#ifndef RK_SLIM
template void launch_list_big<double>(const rk_state &, int, const kparams<double> &, const uint32_t *, int64_t, hipStream_t,
                                      const uint32_t *);
#endif

template void launch_list<float>(const rk_state &, int, const kparams<float> &, const int64_t[n_classes],
                                 const int64_t[n_classes], hipStream_t const[n_list_R], unsigned);
#ifndef RK_SLIM
template void launch_list<double>(const rk_state &, int, const kparams<double> &, const int64_t[n_classes],
                                  const int64_t[n_classes], hipStream_t const[n_list_R], unsigned);
#endif

// Makes the runtime load this translation unit's code object now (rk_init) instead of at the first launch.
void touch_list()
{
    hipFuncAttributes attr{};
    RK_HIP(hipFuncGetAttributes(&attr, reinterpret_cast<const void *>(&k_super<float, mac_targ(0)>)));
}

} // namespace rk
