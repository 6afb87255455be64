/*
 * rakau_amd C ABI -- the drop-in boundary of the MI355X-native Barnes-Hut traversal engine.
 *
 * Every entry point replaces one symbol of the accelerator seam of bluescarni/rakau
 * (paths relative to the reference repository):
 *
 *   rk_min_size()            <- rocm_min_size()             include/rakau/detail/rocm_fwd.hpp:22
 *                               cuda_min_size()             include/rakau/detail/cuda_fwd.hpp:23
 *   rk_has_accelerator()     <- rocm_has_accelerator()      include/rakau/detail/rocm_fwd.hpp:24
 *   rk_device_count()        <- cuda_device_count()         include/rakau/detail/cuda_fwd.hpp:29
 *   rk_state_create()        <- rocm_state<..>::rocm_state  include/rakau/detail/rocm_fwd.hpp:29-30
 *                               (called from tree::rocm_init_state(), include/rakau/tree.hpp:1495-1508)
 *   rk_state_destroy()       <- rocm_state<..>::~rocm_state include/rakau/detail/rocm_fwd.hpp:38
 *                               (tree::rocm_reset_state(), include/rakau/tree.hpp:1511-1519)
 *   rk_acc_pot()             <- rocm_state<..>::acc_pot<Q>  include/rakau/detail/rocm_fwd.hpp:41-42
 *                               (called from tree::acc_pot_impl(), include/rakau/tree.hpp:3078-3094)
 *                               cuda_acc_pot_impl<Q,..>     include/rakau/detail/cuda_fwd.hpp:25-29 (one rk_acc_pot per
 *                               device share on replicas made by rk_state_clone_all: integration/rakau_amd_cuda_bridge.cpp)
 *
 * The two reference-side bindings are real translation units: integration/rakau_amd_bridge.cpp (ROCm seam) and
 * integration/rakau_amd_cuda_bridge.cpp (CUDA seam, the reference's multi-GPU dispatch of tree.hpp:3131-3257).
 *
 *   rk_state_create_nd()     <- the same seam for NDim = 2 (quadtrees), src/rakau_rocm.cpp:333-345
 *
 * The remaining entry points have no counterpart in the reference's seam; they serve the callers either side of it
 *  and the measurements:
 *   device-resident outputs      rk_acc_pot_device (+ RK_OUT_ORDERED = the accs_o/pots_o scatter, tree.hpp:3320-3330)
 *   multi-GPU replication        rk_state_clone / rk_state_clone_all (one process, peer copies over xGMI),
 *                                rk_state_broadcast + rk_comm_* (one process per GPU: RCCL broadcast inside the library),
 *                                rk_state_export / rk_state_import / rk_device_memcpy (RCCL broadcast of the exported
 *                                buffers: the replacement of the reference's multi-GPU split, src/rakau_cuda.cu:410-527),
 *                                rk_state_crit_ranges, rk_group_work (where to cut)
 *   tree construction on the GPU rk_state_build / _build_device / _build_nd / rk_state_rebuild_device (the constructor
 *                                and update_particles_u of tree.hpp:1330-1487, 3678-3765), rk_state_download,
 *                                rk_state_tree_info, rk_state_device_ptr, rk_state_set_perm, rk_set_build_exact,
 *                                rk_pool_trim
 *   CPU share of kwargs::split   rk_cpu_engine_run (AVX-512 flavour of the header's CPU engine, chosen at run time)
 *   warm-up                      rk_init
 *   diagnostics                  rk_state_info, rk_state_ndim, rk_last_kernel_ms, rk_state_set_timing, rk_count_interactions,
 *                                rk_set_kernel_variant
 *
 * Plain C: pointers and sizes only. All functions are blocking and may be called from any thread;
 * at most one call may be in flight per state (same contract as the reference, tree.hpp:3071-3113).
 * Functions returning int return RK_OK or an error code; rk_last_error() then holds the message
 * (thread-local). The C++17 header include/rakau_amd/tree.hpp maps the codes back onto the
 * exception types the reference throws.
 */
#ifndef RAKAU_AMD_H
#define RAKAU_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RK_EXPORT __attribute__((visibility("default")))

/* Floating-point type of the tree (template parameter F of rakau::tree). */
enum { RK_F32 = 0, RK_F64 = 1 };
/* Multipole acceptance criterion (rakau::mac, include/rakau/detail/tree_fwd.hpp:46). */
enum { RK_MAC_BH = 0, RK_MAC_BH_GEOM = 1 };
/* Output addressing flags of rk_acc_pot / rk_acc_pot_device. */
enum { RK_OUT_COMPACT = 0, RK_OUT_OFFSET = 1, RK_OUT_ORDERED = 2 };
/* Status codes; the C++ header rethrows them as the exception types listed. */
enum {
    RK_OK = 0,
    RK_EINVAL = 1,    /* std::invalid_argument */
    RK_EDOMAIN = 2,   /* std::domain_error     */
    RK_EOVERFLOW = 3, /* std::overflow_error   */
    RK_ERUNTIME = 4,  /* std::runtime_error (HIP failure, missing device) */
    RK_ENOMEM = 5     /* std::bad_alloc        */
};

typedef struct rk_state rk_state;

/* Message of the last failing call on this thread ("" if none). */
RK_EXPORT const char *rk_last_error(void);

/* Smallest particle range worth offloading (rocm_min_size() == 64 in the reference). */
RK_EXPORT unsigned rk_min_size(void);
/* 1 if at least one gfx950 device is visible. Never throws. */
RK_EXPORT int rk_has_accelerator(void);
/* Number of visible HIP devices (0 on failure). Test knob: RK_ALIAS_DEVICES=<n> makes the library report n LOGICAL
 * devices mapped round-robin onto the physical ones, so that multi-device host logic can run on a 1-GPU box. */
RK_EXPORT int rk_device_count(void);
/* Optional: pay the one-off costs of the first call on `device` now (HIP runtime and device context, the code objects of all
 * kernel families, the device-memory cache) -- 0.1-0.3 s on a warm box, seconds on a box whose libraries are not yet in the
 * page cache -- so that the first rk_state_create / rk_acc_pot of a latency-sensitive caller do not. */
RK_EXPORT int rk_init(int device);

/*
 * Build the device-resident state for a constructed tree (3-D, 64-bit codes).
 *
 *  fp, mac     RK_F32|RK_F64, RK_MAC_BH|RK_MAC_BH_GEOM: the F and MAC template parameters.
 *  device      HIP device ordinal.
 *  parts       {x, y, z, m}: host arrays of nparts values of type F in Morton order
 *              (tree::p_its_u(), tree.hpp:3638-3641).
 *  codes       sorted Morton codes (tree::m_codes). Not needed by this engine; may be NULL.
 *  nparts      number of particles.
 *  tree        host array of tree_size records laid out as rakau::tree_node_t<3, F, uint64_t, MAC>
 *              (include/rakau/detail/tree_fwd.hpp:77-116): uint64 begin, end, n_children, code, level;
 *              F props[4] (COM x,y,z, mass); then F dim2 (bh) or F dim, delta (bh_geom).
 *              Every field the reference fills must be valid, `code` included: the low NDim bits of a node's code (its
 *              octant inside the parent, tree.hpp:1046-1060) give the sibling order of the engine's node records and
 *              two children of one node must differ in them. Records with codes left at 0 are refused with
 *              "inconsistent tree: not every node is reachable from the root" (RK_CREATE_ON_HOST=1, the host-side
 *              conversion kept as a cross-check, derives the order from the array positions and ignores the codes).
 *  node_stride sizeof of one record in bytes (64/80 for bh fp32/fp64, 64/88 for bh_geom).
 *  ncrit       tree::m_ncrit. Critical nodes (the target groups, tree.hpp:794-807) are re-derived
 *              from the node array: first node on each root->leaf path with
 *              (end - begin <= ncrit || n_children == 0).
 *
 * The state owns device copies; host arrays may be released after the call returns.
 */
RK_EXPORT int rk_state_create(rk_state **out, int fp, int mac, int device, const void *const parts[4],
                              const uint64_t *codes, int64_t nparts, const void *tree, int64_t tree_size,
                              int64_t node_stride, uint64_t ncrit);

/* The same for an ndim-dimensional tree (ndim = 2: quadtree, 3: octree; the NDim template parameter of rocm_state,
 * rocm_fwd.hpp:26): parts = the ndim coordinate arrays followed by the masses, tree = tree_node_t<ndim, F, uint64_t, MAC>
 * records (props[ndim + 1]). Every call that lists outputs then takes ndim accelerations, 1 potential, or ndim + 1
 * arrays (tree_nvecs_res, tree_fwd.hpp). Quadtrees run on the same kernels in the z = 0 plane. */
RK_EXPORT int rk_state_create_nd(rk_state **out, int ndim, int fp, int mac, int device, const void *const *parts,
                                 const uint64_t *codes, int64_t nparts, const void *tree, int64_t tree_size,
                                 int64_t node_stride, uint64_t ncrit);

/* Number of dimensions of a state's tree (0 for a null state). */
RK_EXPORT int rk_state_ndim(const rk_state *s);

RK_EXPORT void rk_state_destroy(rk_state *s);

/* info[0..7] = nparts, tree_size, n_crit, max group size, fp, mac, device, ncrit. */
RK_EXPORT int rk_state_info(const rk_state *s, int64_t info[8]);

/*
 * Copy the particle ranges of the critical nodes (the target groups): begin_end[2*i], begin_end[2*i+1].
 * Used to snap a split to a group boundary (tree.hpp:3053-3063). n_crit entries (see rk_state_info).
 */
RK_EXPORT int rk_state_crit_ranges(const rk_state *s, int64_t *begin_end);

/*
 * Accelerations and/or potentials for the particles [p_begin, p_end) (Morton order).
 *
 *  q              0: accelerations (3 outputs), 1: potentials (1), 2: both (4: ax, ay, az, pot).
 *  p_begin,p_end  must coincide with critical-node boundaries (p_end == nparts qualifies).
 *  out            q-dependent number of host arrays of type F.
 *  mac_value      theta^-2 (bh) or theta^-1 (bh_geom), as computed at tree.hpp:3303-3312.
 *  G              gravitational constant, applied as the final multiply (tree.hpp:2986-3002).
 *  eps2           square of the softening length (tree.hpp:3268-3281).
 *  offset_output  bit 0 (RK_OUT_OFFSET): out[j] addresses element 0 of a full-size array and results are written
 *                 at out[j] + p_begin; clear: out[j] is a compact array of p_end - p_begin values
 *                 (same meaning as in src/rakau_rocm.cpp:114-116).
 *                 bit 1 (RK_OUT_ORDERED): original-order output -- the result of the particle at Morton position i
 *                 goes to out[j][perm[i]] (the accs_o/pots_o scatter of tree.hpp:3320-3330 done in the kernel
 *                 epilogue); out[j] are full-size arrays. rk_acc_pot_device: any range. rk_acc_pot (host arrays):
 *                 the whole range [0, nparts) only -- the kernels scatter into a buffer in HBM, the ordered arrays then
 *                 travel to the host in one piece each. Needs the permutation (rk_state_set_perm, or a device-built state).
 */
RK_EXPORT int rk_acc_pot(rk_state *s, int q, int64_t p_begin, int64_t p_end, void *const *out, double mac_value,
                         double G, double eps2, int offset_output);

/*
 * Pinned (page-locked, device-visible) host memory for output arrays. rk_acc_pot() recognises output arrays that live
 * in such memory -- from rk_host_alloc(), hipHostMalloc() or hipHostRegister() -- and lets the kernels write the
 * results straight into them: no staging buffer, no copy after the kernels (4M fp32 accelerations: 2.4 ms per call
 * instead of 3.1 into pageable arrays). The reference's seam takes raw `F *` outputs (detail/rocm_fwd.hpp:38-40), so
 * this is opt-in by where the caller allocates; `rakau_amd::pinned_allocator` (tree.hpp) wraps it for the
 * `std::vector<F, Allocator>` overloads of accs_u()/pots_u()/accs_pots_u() (tree.hpp:3406-3497 of the reference).
 * bytes == 0 yields a null pointer; rk_host_free(NULL) is a no-op. Freed blocks of 1 MiB and more are parked (at most eight,
 * 512 MiB in all) and handed out again to requests they fit within a factor of two -- pinning costs milliseconds per 16 MiB --;
 * rk_pool_trim() returns them to the system.
 */
RK_EXPORT int rk_host_alloc(void **ptr, int64_t bytes);
RK_EXPORT int rk_host_free(void *ptr);

/*
 * Same, but out[j] are DEVICE pointers (on the state's device) and the kernels are enqueued on
 * `hip_stream` (a hipStream_t; NULL = default stream) without synchronising: results are ready
 * when the stream reaches this point. No host transfer takes place.
 */
RK_EXPORT int rk_acc_pot_device(rk_state *s, int q, int64_t p_begin, int64_t p_end, void *const *d_out,
                                double mac_value, double G, double eps2, int offset_output, void *hip_stream);

/*
 * Timing of the kernels enqueued by the last rk_acc_pot/rk_acc_pot_device call on this state,
 * measured with HIP events on the stream the kernels ran on. Blocks until they have finished.
 * ms[0] = elapsed milliseconds, first launch start -> last launch end.
 */
RK_EXPORT int rk_last_kernel_ms(rk_state *s, float *ms);

/*
 * Kernel timing of rk_acc_pot_device() calls on or off (default: on). Each of the two timing events is a barrier packet
 * between consecutive calls on a stream; a caller that issues calls back to back and never asks for rk_last_kernel_ms()
 * saves them (4M particles: 2.27 -> 2.25 ms per call, 500k: 0.38 -> 0.36, 100k: 0.19 -> 0.17). rk_last_kernel_ms() fails while timing is off.
 */
RK_EXPORT int rk_state_set_timing(rk_state *s, int on);

/*
 * Replication across GPUs. rk_state_export() lists the device buffers that make up a state
 * (count <= RK_MAX_BUFFERS; ptrs/bytes are filled). A peer process allocates buffers of the same sizes
 * on its own GPU, receives the contents (e.g. torch.distributed.broadcast over RCCL/xGMI) and calls
 * rk_state_import() with the same meta block to obtain an equivalent state. Buffers passed to import are
 * copied device-to-device; the caller keeps ownership of them. The last buffer of the list is the permutation
 * (0 bytes for a state that has none): replicas of a state that can do RK_OUT_ORDERED can do it too. import
 * checks every buffer size against the meta block and applies the limits of rk_state_create.
 */
#define RK_MAX_BUFFERS 16
#define RK_META_WORDS 32
RK_EXPORT int rk_state_export(const rk_state *s, int *count, void **ptrs, int64_t *bytes, int64_t meta[RK_META_WORDS]);
RK_EXPORT int rk_state_import(rk_state **out, int device, int count, void *const *ptrs, const int64_t *bytes,
                              const int64_t meta[RK_META_WORDS]);
/* Replica of `src` on another device of THIS process (the reference's multi-GPU host code drives all devices from one
 * process, src/rakau_cuda.cu:410-527): every buffer travels device to device -- a peer copy over xGMI between two
 * GPUs -- instead of being converted and uploaded from host memory again. The C++ header replicates the state of
 * device 0 this way for kwargs::split = {cpu, dev0, dev1, ...}. */
RK_EXPORT int rk_state_clone(rk_state **out, const rk_state *src, int device);
/* Replicas of `src` on the n devices devices[0..n) of this process at once, outs[i] on devices[i]. The copies fan out as a
 * doubling tree (every device that holds the state sends it on, all transfers of a round concurrently: asynchronous peer
 * copies over xGMI on one stream per destination), ceil(log2(n + 1)) rounds instead of n copies out of the source one after
 * the other. Replaces the per-call, per-device upload of src/rakau_cuda.cu:492-527. All or nothing. */
RK_EXPORT int rk_state_clone_all(rk_state **outs, const rk_state *src, const int *devices, int n);

/* One process per GPU: the replicate step over RCCL (xGMI), inside the library. librccl.so.1 is bound at run time (the copy
 * the process has already mapped, e.g. PyTorch's, otherwise the system's); without it these calls fail with RK_ERUNTIME.
 *   rk_comm_unique_id : ncclGetUniqueId on one rank; ship the RK_COMM_ID_BYTES bytes to the others by any means.
 *   rk_comm_init      : ncclCommInitRank for this rank on `device` (collective over all ranks).
 *   rk_state_broadcast: *state of rank `root` (resident on its `device`) is replicated: on every other rank *state is
 *                       created on that rank's `device`. Meta block and buffers travel with ncclBroadcast on `stream`
 *                       (a hipStream_t, may be NULL); blocking. No collective is needed on the data path afterwards.
 * The caller may pass any ncclComm_t of its own as `comm` instead of one made by rk_comm_init. */
#define RK_COMM_ID_BYTES 128
RK_EXPORT int rk_comm_unique_id(char id[RK_COMM_ID_BYTES]);
RK_EXPORT int rk_comm_init(void **comm, int n_ranks, const char id[RK_COMM_ID_BYTES], int rank, int device);
RK_EXPORT int rk_comm_destroy(void *comm);
RK_EXPORT int rk_state_broadcast(rk_state **state, int root, int rank, int device, void *comm, void *stream);

/*
 * Device-side tree construction: what rakau::tree's constructor does on the
 * host -- box-size deduction, discretisation + Morton encoding, indirect sort, permutation, node build, node
 * properties, critical nodes (include/rakau/tree.hpp:1330-1487, 932-1111, 1116-1237) -- executed on `device`;
 * the result is a traversal state like rk_state_create()'s, without any host tree.
 *  parts        {x, y, z, m}: HOST arrays of nparts values of type F in the caller's ORIGINAL order.
 *  box_size     0 = deduce from the data (2 * max|coord| * 1.05), otherwise the domain size. (A caller that must
 *               distinguish an EXPLICIT zero box -- an error in the reference -- does so before the call, as
 *               include/rakau_amd/tree.hpp does.)
 *  max_leaf_n, ncrit   tree parameters (rakau defaults 16 and 128).
 * Errors and messages follow the reference's constructor (invalid_argument for coordinates outside the box...).
 * Node centres of mass are aggregated child -> parent, so they agree with the host builder to rounding.
 */
RK_EXPORT int rk_state_build(rk_state **out, int fp, int mac, int device, const void *const parts[4], int64_t nparts,
                             double box_size, uint64_t max_leaf_n, uint64_t ncrit);

/* Same, with x, y, z, m already resident on `device` (DEVICE pointers): no host transfer at all. */
RK_EXPORT int rk_state_build_device(rk_state **out, int fp, int mac, int device, const void *const d_parts[4],
                                    int64_t nparts, double box_size, uint64_t max_leaf_n, uint64_t ncrit);

/* ndim-dimensional variant of rk_state_build / rk_state_build_device: parts = ndim coordinate arrays + masses, host
 * (on_device = 0) or device (on_device != 0) pointers. rk_state_rebuild_device() keeps a state's ndim. */
RK_EXPORT int rk_state_build_nd(rk_state **out, int ndim, int fp, int mac, int device, const void *const *parts,
                                int on_device, int64_t nparts, double box_size, uint64_t max_leaf_n, uint64_t ncrit);

/* Rebuild the tree of an existing state in place from new device-resident particles (tree::update_particles_u(),
 * tree.hpp:3560-3640 of the reference: the particles moved, sort again and rebuild). fp, mac, max_leaf_n and ncrit
 * are kept; streams, events and scratch are reused and device memory is recycled through the library's block cache,
 * so a time-stepping loop rebuilds without touching the driver allocator. box_size 0 = deduce again. After a
 * failure the state is empty (nparts 0) but valid. */
RK_EXPORT int rk_state_rebuild_device(rk_state *s, const void *const d_parts[4], int64_t nparts, double box_size);

/* Device builder, process-wide: on != 0 makes rk_state_build* / rk_state_rebuild_device sum the node properties in the
 * reference's association (tree.hpp:1162-1168: serially in particle order). The node records -- hence every MAC
 * decision and the interaction census -- are then bit-identical to the host builders', at the price
 * of serial chains (about 12 ms more at 4M particles). Off (default): children are aggregated into their parents
 * (fast; properties equal to rounding). The environment variable RK_BUILD_EXACT=1 sets the default. */
RK_EXPORT void rk_set_build_exact(int on);

/* Return the blocks cached by the library's device allocator to the driver (all devices). */
RK_EXPORT void rk_pool_trim(void);

/* Give a state created from a host tree the permutation tree::perm() (host array of nparts uint64), which the
 * original-order output mode needs. States built on the device have it already. */
RK_EXPORT int rk_state_set_perm(rk_state *s, const uint64_t *perm);

/* Device address and size of a resident array: 0 = particles {x, y, z, m} in Morton order, 1 = perm (uint32),
 * 2 = sorted Morton codes (uint64). For callers that keep their own data on the GPU (gather / scatter).
 * 3 (diagnostic) = the launch order of the first call on the tree, critical-node indices as uint32: eight per-region queues, nodes by
 * decreasing size inside, for trees of at most 49152 critical nodes; the per-class / per-region queues of the light-tail arrangement
 * up to 250000 (0 bytes beyond);
 * 4 (diagnostic) = the table of those queues (72 uint32. Small trees: start [x] and length [8 + x] of the queue of region x. Large
 * ones: per class c and region x the start [16 c + x] and length [16 c + 8 + x], [64 + c] = the size from which a node of class c
 * counts as bulk; 0 bytes for trees without such an order). */
RK_EXPORT int rk_state_device_ptr(const rk_state *s, int what, void **ptr, int64_t *bytes);

/* *box_size = domain size; info[0..3] = box deduced, max_leaf_n, built on device (0/1), number of internal nodes. */
RK_EXPORT int rk_state_tree_info(const rk_state *s, double *box_size, int64_t info[4]);

/*
 * Copy a piece of the resident tree to the host. what: 0..3 = x, y, z, masses in Morton order (p_its_u);
 * 4 = sorted Morton codes (uint64); 5 = perm() as uint64 (original index of the particle at Morton position i);
 * 6 = nodes() in the reference's record layout (stride 64/80 B for bh fp32/fp64, 64/88 for bh_geom);
 * 7 = critical nodes as {code, begin, end} uint64 triples; 8 = the particles as {x, y, z, m} records
 * (4 * nparts values, Morton order; z = 0 for quadtrees, whose selector 2 is invalid and whose node records have
 * props[3]). 4, 6, 7 need a state built on the device; 5 also works after rk_state_set_perm().
 */
RK_EXPORT int rk_state_download(const rk_state *s, int what, void *dst);

/* Device-to-device copy on `device` (plumbing for rk_state_export/import users that stage through their
 * own device buffers). */
RK_EXPORT int rk_device_memcpy(void *dst, const void *src, int64_t bytes, int device);

/*
 * Census of the traversal for the particles [p_begin, p_end): counts[0] = target x node MAC evaluations,
 * counts[1] = target x accepted-node (monopole) interactions, counts[2] = target x particle interactions
 * with opened leaves, counts[3] = ordered particle pairs inside the critical nodes. The sum of the last
 * three is the number of particle-level interactions an acc/pot call evaluates (the reference's CPU engine
 * evaluates exactly the same set); bench.py derives the algorithmic flop count from it.
 */
RK_EXPORT int rk_count_interactions(rk_state *s, int64_t p_begin, int64_t p_end, double mac_value,
                                    uint64_t counts[4]);

/* work[g] = number of particle-level interactions (node + leaf + in-group) that critical node g performs at this MAC
 * value, g = 0 .. n_crit-1 (host array). The traversal cost of a Morton range is proportional to the sum over its
 * critical nodes: callers use it to cut shards / `split` fractions of equal work rather than equal particle counts
 * (the reference leaves the split vector to the user, tree.hpp:2853-2935). */
RK_EXPORT int rk_group_work(rk_state *s, double mac_value, uint64_t *work);

/*
 * CPU engine hand-off (used by include/rakau_amd/tree.hpp for the CPU share of kwargs::split, tree.hpp:3047-3113 of the
 * reference): the header's CPU engine is a template compiled with the CALLER's instruction-set flags. On a CPU with
 * AVX-512 the library can run the same engine compiled for AVX-512 (a separate shared object next to librakau_amd.so,
 * loaded on first use). rk_cpu_engine_run() returns RK_OK if it did the job, a negative value if no wider flavour is
 * available (the caller then runs its own instantiation), a positive status code on error. A NULL job probes:
 * RK_OK if the AVX-512 flavour is available.
 * tree / crit: arrays of rakau::tree_node_t<ndim, F, UInt, MAC> / tree_cnode_t<F, UInt> (code_bits = 32 or 64);
 * critical nodes [c_begin, c_end); parts: ndim coordinate arrays + masses (Morton order); out: full-size Morton-order
 * result arrays; flavour: 0 = automatic, 2 = SIMD with sqrt + divide; nthreads > 0.
 */
typedef struct rk_cpu_job {
    int q, ndim, fp, code_bits, mac, flavour;
    unsigned nthreads;
    const void *tree;
    uint64_t tree_size;
    const void *crit;
    uint64_t c_begin, c_end;
    const void *parts[4];
    void *out[4];
    double mac_value, G, eps2;
} rk_cpu_job;
RK_EXPORT int rk_cpu_engine_run(const rk_cpu_job *job);

/* Select the traversal kernel: 0 = automatic (default: the producer / consumer kernel for calls over few critical nodes,
 * the list kernel otherwise), 1 = wave-per-group scalar DFS, 2 = LDS interaction-list kernel (one wave per critical
 * node), 3 = producer / consumer waves per critical node, 4 = split traversal (list building and dense evaluation as two
 * kernels, the lists in HBM; slower, kept as a cross-check). 2 and 3 give bit-identical results (same interaction lists,
 * same summation order); 1 sums in the CPU engine's order, 4 in a breadth-first order of its own that does not depend on
 * the launch either. Variants 1 and 4 are not part of librakau_amd.so: selecting one loads librakau_amd_xcheck.so from the
 * same directory (RK_ERUNTIME if it is not there). For tests and benchmarks. */
RK_EXPORT int rk_set_kernel_variant(rk_state *s, int variant);

/* Diagnostics of the hipGraph replay of repeated rk_acc_pot_device() calls: stats[0] = calls replayed from a cached executable
 * graph, [1] = calls that captured their launch sequence (a signature seen before), [2] = calls launched directly, [3] = captures
 * that re-targeted a parked executable (hipGraphExecUpdate) instead of instantiating a new one, [4] = executable graphs the state
 * holds now (at most RK_GRAPH_CACHE = 8, least recently used evicted), [5] = executables with parallel branches alive in the
 * process (never destroyed; at most RK_GRAPH_FORKED_MAX = 64). */
RK_EXPORT int rk_state_graph_stats(const rk_state *s, int64_t stats[6]);

#ifdef __cplusplus
}
#endif

#endif
