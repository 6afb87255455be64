// Shared declarations of the rakau_amd engine (host + device).
#ifndef RK_COMMON_HPP
#define RK_COMMON_HPP

#include <hip/hip_runtime.h>

#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/rakau_amd.h"

namespace rk
{

// Vector types per floating-point type.
template <typename F>
struct vt;
template <>
struct vt<float> {
    using v4 = float4;
    using v2 = float2;
};
template <>
struct vt<double> {
    using v4 = double4;
    using v2 = double2;
};

// Number of result vectors for Q (tree_fwd.hpp:129-137 of the reference).
__host__ __device__ constexpr int nres_of(int q)
{
    return q == 0 ? 3 : (q == 1 ? 1 : 4);
}

// Target groups (critical nodes) are binned by the number of targets each lane of a wave holds:
// class c holds groups with size <= 64 * R(c); the last class is served by the block-per-group kernel.
constexpr uint32_t RK_PLAN_PAD_VALUE = 0xffffffffu; // launch-plan list entry without a critical node (skipped)
constexpr int n_classes = 7;  // classes 0..5 are served by wave kernels, the last one by the block-per-group kernel
constexpr int big_class = n_classes - 1;
// Beyond FIRST_ORDER_MAX and up to this many critical nodes a tree gets the LIGHT-TAIL arrangement of a first call instead (per
// class and per XCD region: the nodes in Morton order, the lightest quarter of the class at the end; rk_build.hip k_tail_sizes).
// The same limit holds for the light-tail plans of repeated calls: beyond it the plain Morton slices per XCD win (rk_launch.hip).
constexpr unsigned FIRST_TAIL_MAX = 250000;
// Table of the light-tail arrangement (uint32, device memory): per wave-kernel class c (R = c + 1) and XCD region x the start
// [c * 16 + x] and the length [c * 16 + 8 + x] of its queue inside first_order; [64 + c] = size from which a node of class c is bulk.
constexpr unsigned FIRST_TAB_WORDS = 72;
constexpr unsigned FIRST_ORDER_MAX = 49152; // critical nodes up to which a tree gets its first-call launch order on the device (round 4: 32768)
constexpr int n_list_R = 6;   // variant 2: class c keeps R = c + 1 targets per lane
__host__ __device__ constexpr int class_R(int c)
{
    return c == 0 ? 1 : (c == 1 ? 2 : (c == 2 ? 4 : 8));
}
// Largest number of targets per lane used by the list kernel. Measured on MI355X (4M Plummer): 4 -> 2.58 ms, 5 -> 2.85 ms,
// 6 -> 2.79 ms: beyond 4 the registers cost more occupancy than the better lane packing returns.
#ifndef RK_MAX_R
#define RK_MAX_R 4
#endif
#ifndef RK_MIN_R
#define RK_MIN_R 1
#endif
#ifndef RK_TIE_HIGH
#define RK_TIE_HIGH 0
#endif
// Relative cost per lane-iteration of the classes R = 1..4 in the class choice (1: the choice minimises idle lanes only).
#ifndef RK_CLASS_W1
#define RK_CLASS_W1 1.0
#endif
#ifndef RK_CLASS_W2
#define RK_CLASS_W2 1.0
#endif
#ifndef RK_CLASS_W3
#define RK_CLASS_W3 1.0
#endif
#ifndef RK_CLASS_W4
#define RK_CLASS_W4 1.0
#endif
// Targets per lane of the variant-2 (list kernel) classes.
__host__ __device__ constexpr int class2_R(int c)
{
    return c + 1;
}
inline int class_of(int64_t size)
{
    if (size <= 64) return 0;
    if (size <= 128) return 1;
    if (size <= 256) return 2;
    if (size <= 512) return 3;
    return big_class;
}
// Variant 2 (LDS interaction lists): each lane holds R targets, TP = ceil(T / R) target slots and
// NS = floor(64 / TP) source splits share the wave. The dense phase costs R / NS lane-iterations per
// source; pick the R that minimises it (ties: fewer registers).
__host__ __device__ inline int class2_of_compute(int64_t size)
{
    if (size > 64 * RK_MAX_R) return big_class;
    int best = -1;
    double best_cost = 0.;
    for (int c = RK_MIN_R - 1; c < RK_MAX_R; ++c) { // R = RK_MIN_R .. RK_MAX_R
        const int64_t R = class2_R(c), TP = (size + R - 1) / R;
        if (TP > 64) continue;
        // Cost of one source for all targets of the node, in lane-iterations (R / NS), times the relative cost of a
        // lane-iteration with R targets per lane: every source is one broadcast LDS read, which R targets share.
        constexpr double w[6] = {RK_CLASS_W1, RK_CLASS_W2, RK_CLASS_W3, RK_CLASS_W4, 1.0, 1.0};
        const double cost = static_cast<double>(R) / static_cast<double>(64 / TP) * w[c];
#if RK_TIE_HIGH
        if (best < 0 || cost < best_cost + 1e-12) { // ties go to the larger R
#else
        if (best < 0 || cost < best_cost - 1e-12) {
#endif
            best = c;
            best_cost = cost;
        }
    }
    return best;
}

inline int class2_of(int64_t size)
{
    struct table {
        signed char v[64 * RK_MAX_R + 1];
        table()
        {
            for (int i = 0; i <= 64 * RK_MAX_R; ++i) {
                v[i] = static_cast<signed char>(class2_of_compute(i ? i : 1));
            }
        }
    };
    static const table t;
    return size > 64 * RK_MAX_R ? big_class : t.v[size];
}

// Node records of the list kernel, stored in SIBLING order: the children of a node occupy consecutive
// records, so that one stack entry (first child, number of children) names up to 8 candidate nodes and a
// wave fetches them with coalesced loads. `dfs` / `nch` keep the depth-first index and descendant count of
// the reference's layout (needed for the ancestor test and nothing else). Leaves store their particle
// range where internal nodes store the location of their children.
template <typename F>
struct node_rec {
    typename vt<F>::v4 com; // COM x, y, z, mass
    typename vt<F>::v2 mac; // {dim2, 0} (bh) or {dim, delta} (bh_geom)
    uint32_t dfs;           // index of the node in the depth-first array
    uint32_t nch;           // number of descendants (0 for a leaf)
    uint32_t a, b;          // internal node: a = record index of the first child, b = number of children
                            // leaf: a = first particle, b = one past the last particle
    uint32_t pad[2];
};
static_assert(sizeof(node_rec<float>) == 48 && sizeof(node_rec<double>) == 96, "unexpected node record size");
// A stack entry packs (first child record << 3) | (number of children - 1).
constexpr uint32_t max_list_nodes = 1u << 29;
// Capacities of the per-supergroup lists written by the pre-pass kernel.
constexpr uint32_t SUP_CAPC = 1536, SUP_CAPR = 512;

// Split traversal (variant 4: k_lists writes the interaction lists of every critical node to HBM, k_dense evaluates
// them). Two lists per node: the depth-first indices of the accepted nodes, and the Morton indices of the particles of
// the opened leaves. A list is a chain of segments of SL_SEG 32-bit entries. Segments 2 s and 2 s + 1, s < (number of
// critical nodes of the call), are the first segments of the two lists of node sl_g0 + s; further segments are taken
// from a bump counter (sl_ctl[0]) and linked through sl_next[].
constexpr uint32_t SL_SEG = 512;          // entries per segment: 4 tiles of 128 sources
constexpr uint32_t SL_OVER = 0x80000000u; // sl_cnt[2 g]: the lists were not completed (pool exhausted or longer than
                                          // sl_max_len): the node is served by the fused list kernel instead
constexpr uint32_t SL_RING = 512;         // entries of the LDS ring in which k_lists stages a list

// Kernel parameter block (passed by value).
template <typename F>
struct kparams {
    const typename vt<F>::v4 *part4;    // particles, Morton order: x, y, z, m
    const typename vt<F>::v4 *node_com; // nodes, depth-first order: COM x, y, z, mass
    const typename vt<F>::v2 *node_mac; // {dim2, 0} (bh) or {dim, delta} (bh_geom)
    const uint4 *node_topo;             // {n_children, begin, end, child-table slot}
    const uint4 *crit;                  // target groups: {begin, end, node index, size}
    const uint32_t *child_tab;          // 8 child node indices per internal node (0 = none)
    const node_rec<F> *node_rec;        // sibling-ordered records (list kernel)
    const typename vt<F>::v4 *crit_box; // per target group: {min x,y,z, -}, {max x,y,z, -} of its particles
    uint32_t n_nodes;
    F mac_value, eps2, G;
    F *out[4];
    uint32_t out_sub; // value subtracted from the particle index when writing (compact output)
    const uint32_t *perm; // non-null: original-order output, results of Morton particle i go to out[perm[i]]
    unsigned long long *dbg; // diagnostic builds only (RK_STAMPS): per-section cycle totals
    int mac;                 // RK_MAC_BH | RK_MAC_BH_GEOM (the list kernels read it at run time: one code object for both)
    int xcd_mode;            // block -> group-list mapping (see xcd_map_block); 3: the queues of first_tab per (class, XCD region), class
                             // kernels; 4: its eight queues per XCD region, one-launch kernels
    const uint32_t *first_tab; // xcd_mode 3 / 4: queue starts and lengths, see rk_state::first_tab
    int any_rev;             // one-launch kernels: block i serves list entry n - 1 - i (class lists read backwards: R = 4 first)
    // Supergroup pre-pass (k_super): K consecutive target groups share the upper part of list building.
    // super_k == 0 disables it. Per supergroup S: sup_common[S * SUP_CAPC ...] = sources {x, y, z, m} accepted for every
    // member group; sup_resid[S * SUP_CAPR ...] = node records every member still has to test itself;
    // sup_cnt[S] = {number of common sources, number of residual records | overflow flag in bit 31}.
    uint32_t super_k, n_crit;
    typename vt<F>::v4 *sup_common;
    uint32_t *sup_resid;
    uint2 *sup_cnt;
    // Split traversal (variant 4), see SL_SEG above.
    uint32_t *sl_idx;    // list segments
    uint32_t *sl_next;   // sl_next[s] = segment that continues segment s
    uint32_t *sl_cnt;    // per critical node g: [2 g] entries of its node list | SL_OVER, [2 g + 1] of its particle list
    uint32_t *sl_ctl;    // [0] segments taken from the pool, [1] nodes on the fallback list, [2] entries written, [3] nodes
                         // that found a pool exhausted, [4] partial-sum slots taken (8 words, zeroed before every call)
    uint32_t *sl_fb;     // fallback list: nodes whose list was not completed
    uint32_t sl_g0;      // first critical node of the call (first segments of node g: 2 (g - sl_g0) and that + 1)
    uint32_t sl_nslot;   // fixed first segments (two per critical node of the call); pool segments follow
    uint32_t sl_nseg;    // segments in all
    uint32_t sl_max_len; // longest list k_lists writes (a property of the call's parameters, not of the launch)
    // Calls over few critical nodes: one wavefront per part of a node (sl_parts_mode); the per-lane sums of every part go
    // to sl_part (slots of 64 x 16 values, sl_pbase[g] = first slot of node g, taken from the counter sl_ctl[4]).
    int sl_parts_mode;
    uint32_t sl_npart;   // slots in sl_part
    uint32_t *sl_pbase;
    void *sl_part;
};

// Index of the output element of Morton particle i.
template <typename F>
__device__ __forceinline__ uint32_t out_index(const kparams<F> &P, uint32_t i)
{
    return P.perm ? P.perm[i] : i - P.out_sub;
}

struct error : std::runtime_error {
    int code;
    error(int c, const std::string &m) : std::runtime_error(m), code(c) {}
};

// Device memory for states and build temporaries comes from a per-device cache of freed blocks (rk_pool.hip):
// rebuilding a tree every time step would otherwise pay hipMalloc/hipFree (implicit device syncs, ~0.1-1 ms each)
// dozens of times. Blocks are handed back to the driver by pool_trim() (rk_pool_trim()). RK_POOL=0 disables caching.
// Reuse is safe because every producer/consumer of a recycled block is ordered on the null stream or fenced by a
// device synchronisation in pool_release_sync().
void *pool_alloc(size_t bytes);
void pool_free(void *p) noexcept;
void pool_trim() noexcept;

#define RK_HIP(expr)                                                                                                   \
    do {                                                                                                               \
        hipError_t e_ = (expr);                                                                                        \
        if (e_ != hipSuccess) {                                                                                        \
            throw ::rk::error(e_ == hipErrorOutOfMemory ? RK_ENOMEM : RK_ERUNTIME,                                     \
                              std::string("HIP call failed: " #expr ": ") + hipGetErrorString(e_));                    \
        }                                                                                                              \
    } while (0)

} // namespace rk

// Device buffer indices in rk_state::buf (also the export order).
enum { RK_BUF_PART4 = 0, RK_BUF_NODE_COM, RK_BUF_NODE_MAC, RK_BUF_NODE_TOPO, RK_BUF_CRIT, RK_BUF_CHILD, RK_BUF_CLASS, RK_BUF_NODE_REC, RK_BUF_CRIT_BOX, RK_NBUF };

struct rk_state {
    int fp = 0, mac = 0, device = 0;
    int ndim = 3; // 3 = octree, 2 = quadtree (particles carry z = 0 and the kernels' z output goes to z_scratch)
    void *z_scratch = nullptr;
    int64_t nparts = 0, tree_size = 0, n_crit = 0, max_group = 0, n_internal = 0;
    uint64_t ncrit = 0;
    void *buf[RK_NBUF] = {};
    int64_t buf_bytes[RK_NBUF] = {};
    // Host mirrors used to map a particle range onto groups.
    // Host mirrors of the critical-node ranges and class lists. States built on the device bin their groups on the
    // device too and only fetch the per-class counts; the mirrors are then filled on first use (sub-range calls,
    // variant 1, export, rk_state_crit_ranges): mirrors_valid says which.
    bool mirrors_valid = true;
    int64_t class2_count[rk::n_classes] = {};
    std::vector<int64_t> crit_begin, crit_end;
    std::vector<uint32_t> class_list[rk::n_classes]; // ascending group ids per class (variant 1 binning)
    int64_t class_off[rk::n_classes + 1] = {};       // offsets into the concatenated device list
    std::vector<uint32_t> class2_list[rk::n_classes]; // variant 2 binning (best targets-per-lane R)
    int64_t class2_off[rk::n_classes + 1] = {};
    // Output scratch for rk_acc_pot (host outputs).
    void *d_out = nullptr;
    size_t d_out_bytes = 0;
    // Host-output path of rk_acc_pot(): pinned staging buffer the kernels write into directly (host memory mapped into
    // the device's address space), delivered to the caller's pageable arrays by host threads.
    void *h_stage = nullptr;
    size_t h_stage_bytes = 0;
    // Output arrays of the previous rk_acc_pot() call (pageable arrays seen before are registered whatever their size).
    const void *last_host_out[4] = {};
    size_t last_host_bytes = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // Side streams (and fork/join events) that let the per-class kernels of one call overlap.
    hipStream_t aux_stream[rk::n_list_R] = {};
    hipEvent_t ev_fork = nullptr, ev_join[rk::n_list_R] = {};
    // hipGraph of the launch sequence of the last call (replayed when a call repeats it).
    struct graph_key {
        int q, offset_output, super_k, variant;
        int with_super, pad; // whether the captured sequence starts with the supergroup pre-pass
        int64_t p_begin, p_end;
        double mac_value, G, eps2;
        void *out[4];
        const void *perm; // permutation buffer the ordered epilogue reads (null: Morton-order output)
    };
    // Executable graphs of the launch sequences this state has been asked for more than once, most recently used last
    // (at most RK_GRAPH_CACHE = 8): a caller that alternates among a few signatures -- accs_u then pots_u, several ranges --
    // keeps replaying all of them. Every entry holds the launch plan its sequence was captured with (the kernels of a planned
    // call read the plan's list buffer, so that buffer lives as long as the graph).
    struct graph_entry {
        graph_key key;
        hipGraphExec_t exec;
        bool forked; // the captured sequence has parallel branches (such executables are never destroyed, see rk_launch.hip)
    };
    std::vector<graph_entry> gcache;
    // Executable graphs of the class kernels alone (no pre-pass, no events: four independent kernel nodes), RE-TARGETED on every use
    // (hipGraphExecKernelNodeSetParams: new grid sizes and arguments, microseconds): what a call that is the first of its kind --
    // every traversal of a time-stepping loop -- launches its class kernels through instead of forking them onto side streams
    // (rk_launch.hip launch_classes_retargeted). func[] identifies the kernels (precision, Q, criterion, dimension, R).
    struct class_graph {
        int pdev = -1;
        unsigned mask = 0u;
        const void *func[4] = {};
        hipGraph_t graph = nullptr;
        hipGraphExec_t exec = nullptr;
        hipGraphNode_t node[4] = {};
    };
    std::vector<class_graph> class_graphs;
    std::vector<graph_key> seen_keys; // signatures of the last calls (a graph is captured when one recurs)
    uint64_t graph_stats[4] = {};     // replays, captures (instantiated or re-targeted), direct launches, re-targeted executables
    hipStream_t cap_stream = nullptr;
    bool timed = false;  // the last call recorded ev0 / ev1
    bool timing = true;  // rk_state_set_timing
    int variant = 0;
    // Group lists the class kernels of the current call read: the state's lists (ascending critical nodes per class,
    // RK_BUF_CLASS + class2_off) or a launch plan.
    const uint32_t *cur_lists = nullptr;
    int64_t cur_off[rk::n_classes + 1] = {};
    // Launch plan of a repeated small call: the critical nodes of the range, per class, reordered so that the most
    // expensive supergroups are dispatched first (a launch with only a few rounds of waves otherwise ends in a long,
    // badly occupied tail of expensive groups that started late). Results do not depend on the dispatch order.
    struct launch_plan {
        int64_t p_begin = -1, p_end = -1;
        double mac_value = 0.;
        void *d_lists = nullptr;
        std::shared_ptr<void> hold; // owns d_lists (shared with the cached graphs captured on this plan; never rewritten while shared)
        int64_t alloc = 0; // entries allocated
        int64_t off[rk::n_classes + 1] = {};
        // Heavy-first plans also carry merged lists for the one-launch kernels (k_pc_any / k_list_any): the wave-kernel
        // classes together, and without the R = 2 class (which then keeps its own producer / consumer launch).
        int64_t off_all = 0, n_all = 0, off_oth = 0, n_oth = 0;
        int64_t off_reg = 0, n_reg = 0; // all classes as eight heavy-first queues, one per XCD region, interleaved and padded: what
                                        // the one-launch kernels run over (off_all: the same nodes as one sorted list)
    } plan;
    std::vector<launch_plan> gcache_plan; // gcache_plan[i]: the plan gcache[i] was captured with (d_lists null: none)
    std::vector<launch_plan> plans;       // the last few plans built (most recent last): a caller alternating among ranges --
                                          // the two parts of a staged host-output call -- does not rebuild them
    hipEvent_t ev_mid = nullptr;          // end of the first part of a two-part host-output call
    hipEvent_t ev_arr[4] = {};            // ordered host outputs: result array k has arrived in the staging buffer
    bool want_done_event = false;         // blocking host-output call: ev1 is recorded whatever the timing setting
    bool keep_ev0 = false;                // second part of such a call: the timing start event stays where the first part put it
    std::vector<uint64_t> work_cache; // launch-plan weight of every critical node (its size; empty: not computed)
    // Scratch of the supergroup pre-pass (allocated on first use).
    void *sup_common = nullptr, *sup_resid = nullptr, *sup_cnt = nullptr;
    int64_t sup_alloc = 0; // number of supergroups the scratch was sized for
    // What the scratch currently holds: the pre-pass output depends on the tree and the MAC value only, so a call that
    // repeats (or follows: accs_u then pots_u) another one on the same tree reuses it instead of running k_super again.
    double sup_mac = 0.;
    int64_t sup_b = 0, sup_e = 0; // supergroups [sup_b, sup_e) are valid for sup_mac (empty: nothing cached)
    hipEvent_t sup_ev = nullptr;  // recorded after the last k_super
    // Calls on different streams. The state's scratch (pre-pass lists, launch plans, the cross-check variant's list pool) belongs to
    // one call at a time; calls on ONE stream are ordered by the stream. A call that arrives on another stream than the previous one
    // waits -- on the device, hipStreamWaitEvent -- for the event the previous call recorded behind its last launch: ev1 while the
    // timing events are on (the default), ev_done once the state has seen a stream change with them off. last_stream is only ever
    // COMPARED (the caller may have destroyed that stream since); last_done names the event that covers the previous call, or null.
    hipStream_t last_stream = nullptr;
    bool has_last_stream = false, multi_stream = false;
    hipEvent_t ev_done = nullptr, last_done = nullptr;
    int super_k = -1;      // members per supergroup of the pre-pass; -1 = not initialised yet (16)
    // Split traversal (variant 4): list pool and control words, sized per call (grown, never shrunk, until the state goes).
    void *sl_idx = nullptr, *sl_next = nullptr, *sl_cnt = nullptr, *sl_ctl = nullptr, *sl_fb = nullptr;
    // Launch order of the first call on a small tree (rk_build.hip k_first_order): the nodes of the wave kernels' classes by
    // decreasing size, made on the device with the tree. Valid for the tree it was made with only.
    void *first_order = nullptr;
    int64_t first_order_cap = 0; // entries allocated
    bool first_order_valid = false;
    // ... or, for trees of FIRST_ORDER_MAX .. FIRST_TAIL_MAX critical nodes, the light-tail arrangement: first_order holds the queues
    // of every (class, XCD region), first_tab their starts and lengths, first_grid[c] = 8 x the longest queue of class c (the grid of
    // its class kernel: block i serves entry i / 8 of the queue of region i % 8, blocks past the end of their queue exit).
    bool first_tail_valid = false;
    void *first_tab = nullptr;
    uint32_t first_grid[4] = {};
    int64_t sl_nseg = 0, sl_ncnt = 0;   // segments / per-node counters allocated
    void *sl_pbase = nullptr, *sl_part = nullptr;
    int64_t sl_npart = 0, sl_part_hint = 0; // partial-sum slots allocated / asked for by the last reports
    int64_t sl_extra_hint = 0;          // pool segments (beyond the fixed first ones) the last reports ask for
    uint32_t *sl_host = nullptr;        // pinned mirror of sl_ctl, filled by the report copy of a call
    hipEvent_t sl_rep_ev = nullptr;     // recorded behind that copy
    bool sl_used = false;
    struct sl_key {                     // (range, MAC value) of a call
        int64_t p_begin = -1, p_end = -1;
        double mac_value = 0.;
        bool operator==(const sl_key &o) const
        {
            return p_begin == o.p_begin && p_end == o.p_end && mac_value == o.mac_value;
        }
    };
    sl_key sl_rep_key, sl_clean_key;    // call whose report is in flight / call known to need no fallback launch
    bool sl_rep_pending = false, sl_clean_valid = false;
    // ... and the form that call ran in (one wave per node / per part, pool sizes): a clean report only vouches for calls in
    // the same form with pools at least as large.
    int sl_rep_mode = 0, sl_clean_mode = 0;
    int64_t sl_rep_npart = 0, sl_rep_nseg = 0, sl_clean_npart = 0, sl_clean_nseg = 0;
    std::vector<sl_key> plan_keys;      // (range, MAC value) of the last calls: a launch plan is built when one recurs
    // Filled by the device-side tree build (rk_state_build); null for states created from a host tree.
    void *bld_codes = nullptr;     // uint64 sorted Morton codes [nparts]
    void *bld_perm = nullptr;      // uint32 original index of the particle at Morton position i [nparts]
    void *bld_node_code = nullptr; // uint64 nodal codes in depth-first order [tree_size]
    int bld_max_level = -1;        // deepest leaf level of the last device build of this state (-1: none yet): the next rebuild sorts
                                   // only the code bits of that many levels + 1 and orders the leaves' insides itself (rk_build.hip)
    double box_size = 0.;
    bool box_deduced = false;
    uint64_t max_leaf_n = 0;
};

namespace rk
{
// Implemented in rk_kernels.hip.
template <typename F>
void launch_traversal(const rk_state &s, int q, const kparams<F> &p, const int64_t cls_begin[n_classes],
                      const int64_t cls_end[n_classes], hipStream_t stream);
template <typename F>
void launch_list(const rk_state &s, int q, const kparams<F> &p, const int64_t cls_begin[n_classes],
                 const int64_t cls_end[n_classes], hipStream_t const streams[n_list_R], unsigned class_mask = ~0u);
// The class kernel launch_list() would launch for lane-mapping class c of this state and Q (for kernel nodes of explicit graphs).
template <typename F>
const void *list_kernel_symbol(const rk_state &s, int q, int c);
// One launch over critical nodes of any lane-mapping class (small calls): the list kernel, the producer / consumer kernel.
template <typename F>
void launch_list_any(const rk_state &s, int q, const kparams<F> &p, const uint32_t *list, int64_t n, hipStream_t stream);
template <typename F>
void launch_pc_any(const rk_state &s, int q, const kparams<F> &p, const uint32_t *list, int64_t n, hipStream_t stream);
// n_dev (optional): the number of list entries lives in device memory (at most n).
template <typename F>
void launch_list_big(const rk_state &s, int q, const kparams<F> &p, const uint32_t *list, int64_t n, hipStream_t stream,
                     const uint32_t *n_dev = nullptr);
template <typename F>
void launch_pc(const rk_state &s, int q, const kparams<F> &p, const int64_t cls_begin[n_classes],
               const int64_t cls_end[n_classes], hipStream_t const streams[n_list_R], unsigned class_mask = ~0u);
template <typename F>
void launch_super(const rk_state &s, const kparams<F> &p, int64_t s_begin, int64_t s_end, hipStream_t stream);
// Split traversal (rk_kernels_split.hip): list building for the critical nodes [g_begin, g_end), dense evaluation per class.
template <typename F>
void launch_lists(const rk_state &s, const kparams<F> &p, int64_t g_begin, int64_t g_end, hipStream_t stream);
template <typename F>
// what: 0 = one wavefront per node, 1 = one wavefront per part of a node (partial sums to sl_part), 2 = k_combine.
void launch_dense(const rk_state &s, int q, const kparams<F> &p, const int64_t cls_begin[n_classes],
                  const int64_t cls_end[n_classes], hipStream_t const streams[n_list_R], unsigned class_mask, int what);
template <typename F>
void launch_block(const rk_state &s, int q, const kparams<F> &p, const uint32_t *list, int64_t n, hipStream_t stream);
template <typename F, int ND>
void build_device(rk_state &s, const void *const parts[4], bool parts_on_device, int64_t nparts, double box_size,
                  uint64_t max_leaf_n, std::string &bad_coord_msg);
// rk_state_create: the device buffers of a state derived ON THE DEVICE from a host-built tree (rk_build.hip).
template <typename F, int ND>
void convert_device(rk_state &s, const void *const parts[4], int64_t nparts, const void *tree, int64_t tree_size,
                    int64_t node_stride);
// Launch order of the first call for a replica whose critical nodes are on its device already (rk_build.hip).
void replica_first_order(rk_state &s);
// Device builder: sum node properties in the reference's serial association (rk_set_build_exact / RK_BUILD_EXACT).
bool exact_node_sums();
void touch_kernels();
void touch_list();
void touch_pc();
void touch_split();
void touch_build();
template <typename F>
void launch_census(const rk_state &s, const kparams<F> &p, int64_t g_begin, int64_t g_end,
                   unsigned long long *d_counts, unsigned long long *d_per_group, hipStream_t stream);
} // namespace rk

#endif
