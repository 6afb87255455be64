/*
 * C ABI over the host side of rakau_amd::octree<F, MAC> (include/rakau_amd/tree.hpp), for hosts that
 * cannot include a C++17 header (the Python test/benchmark harness binds it with ctypes).
 * Each function forwards to the member of the same name; the reference members they mirror are cited
 * in include/rakau_amd/tree.hpp. Status codes and rk_last_error() as in include/rakau_amd.h.
 */
#ifndef RAKAU_AMD_TREE_H
#define RAKAU_AMD_TREE_H

#include "rakau_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rk_tree rk_tree;

/* octree<F, MAC>{x_coords, y_coords, z_coords, masses, nparts, [box_size], max_leaf_n, ncrit}.
 * box_size == 0 -> deduced from the data. flags: bit 0 = build (and rebuild) the tree on the GPU
 * (kwargs::device_build, rk_state_build). */
RK_EXPORT int rk_tree_create(rk_tree **out, int fp, int mac, const void *x, const void *y, const void *z,
                             const void *m, int64_t nparts, double box_size, uint64_t max_leaf_n, uint64_t ncrit,
                             int flags);
/* The same for ndim = 2 (quadtree<F, MAC>) or 3: src = the ndim coordinate arrays followed by the masses. Output
 * lists of the other entries then have ndim / 1 / ndim + 1 arrays, and the z arguments are ignored for quadtrees.
 * flags bit 0: build on the device; bit 1: 32-bit Morton codes (tree<ndim, F, std::uint32_t, MAC>: 10 / 15 bits per
 * coordinate; codes and the code / level fields of the node records are then 32 bits wide; host builder only). */
RK_EXPORT int rk_tree_create_nd(rk_tree **out, int ndim, int fp, int mac, const void *const *src, int64_t nparts,
                                double box_size, uint64_t max_leaf_n, uint64_t ncrit, int flags);
RK_EXPORT void rk_tree_destroy(rk_tree *t);
/* info[0..7] = nparts, number of nodes, number of critical nodes, max_leaf_n, ncrit, box_size_deduced,
 * sizeof(node record), ndim; *box_size = box_size(). */
RK_EXPORT int rk_tree_info(const rk_tree *t, int64_t info[8], double *box_size);
/* Copy out one array. what: 0..3 = x, y, z, masses in Morton order (p_its_u); 4 = codes (c_it_u);
 * 5 = perm(); 6 = last_perm(); 7 = inv_perm(); 8 = critical nodes as {code, begin, end} uint64 triples. */
RK_EXPORT int rk_tree_get(const rk_tree *t, int what, void *dst);
/* View of nodes(): pointer to the first record, record count and record size in bytes. */
RK_EXPORT int rk_tree_nodes(const rk_tree *t, const void **ptr, int64_t *count, int64_t *stride);
/* Device-resident state on device 0 (created on first use; owned by the tree). */
RK_EXPORT int rk_tree_state(const rk_tree *t, rk_state **state);
/* accs_{u,o} (q=0), pots_{u,o} (q=1), accs_pots_{u,o} (q=2) with kwargs G, eps, split. out: 3/1/4 host
 * arrays of nparts values. theta is the un-transformed MAC value. */
RK_EXPORT int rk_tree_acc_pot(const rk_tree *t, int q, int ordered, void *const *out, double theta, double G,
                              double eps, const double *split, int n_split);
/* The CPU engine of the header alone (the engine that runs the host share of kwargs::split, tree.hpp:3047-3113 of the
 * reference; tree::cpu_acc_pot_u): Morton-order results for the whole tree. flavour: 0 = automatic (widest SIMD, fp32
 * rsqrt + Newton step), 1 = scalar (the arithmetic and summation order of the reference's scalar branch), 2 = SIMD with
 * sqrt + divide. nthreads 0 = all usable host threads. Needs no GPU. */
RK_EXPORT int rk_tree_cpu_acc_pot(const rk_tree *t, int q, void *const *out, double theta, double G, double eps,
                                  int flavour, unsigned nthreads);
/* exact_{acc,pot,acc_pot}_{u,o}(idx, G, eps): out receives 3/1/4 values. */
RK_EXPORT int rk_tree_exact(const rk_tree *t, int q, int ordered, int64_t idx, double G, double eps, void *out);
/* update_particles_u with a functor that overwrites the Morton-ordered x, y, z, masses with the given
 * arrays (a null pointer leaves that array unchanged). */
RK_EXPORT int rk_tree_update_particles(rk_tree *t, const void *x, const void *y, const void *z, const void *m);

#ifdef __cplusplus
}
#endif

#endif
