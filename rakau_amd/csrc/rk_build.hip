// Device-side tree construction: everything the reference does in
// construct_impl() / build_tree() / compute_node_properties() (include/rakau/tree.hpp:1330-1487, 932-1111,
// 1116-1237 of the reference) runs on the GPU and leaves the traversal state resident in HBM:
//
//   max |coord| -> box size            (tree.hpp:1279-1319)
//   discretise + Morton encode         (tree.hpp:381-429, 1441-1450)
//   stable radix sort of (code, index) (tree.hpp:1267-1274; rocPRIM's onesweep passes under this file's launch sequence from
//                                       2^20 particles, the library's merge sort below: sort_codes())
//   permute particles, perm            (tree.hpp:1461-1482)
//   node topology in depth-first order (tree.hpp:723-833)
//   node mass / centre of mass / size  (tree.hpp:1116-1237)
//   critical nodes                     (tree.hpp:801-807)
//   + the kernel-side structures of rk_state_create() (sibling-ordered records, group boxes, child table).
//
// Topology without a level loop: with sorted codes c[0..n), let ldiv(i) be the first level at which c[i] leaves
// the cell of c[i-1] and leaf(i) the level of the leaf that holds particle i. Particle i is the first particle of
// exactly the nodes at levels ldiv(i)..leaf(i); sorting nodes by (first particle, level) IS the depth-first order,
// so an exclusive scan of max(0, leaf(i) - ldiv(i) + 1) gives every node its depth-first index directly; every node then finds
// its own end (one thread per node).
//
// Node properties: every node takes the sums over its particles from a summation pyramid over the particles (aligned runs of
// 2^l particles, two or three launches; no level passes). The reference sums every node's particles serially; its own SIMD
// flavour (tree.hpp:1134-1161) already associates differently, so centres of mass agree to rounding, not bit for bit. Exact
// mode (rk_set_build_exact) reproduces the reference's serial association instead.
//
// The host looks at the build three times (node count; critical / internal node counts; class sizes) through asynchronous copies
// of a control block; a dependent launch costs ~5 us on this device whatever it holds, so the build is written for few launches
// (36 at 100k particles, ten of them the library's merge sort).
#include "rk_common.hpp"
#include "rk_device.hpp"

#include <hipcub/hipcub.hpp>

#include <cmath>
#include <cstdio>
#include <cstring>

namespace rk
{
namespace bld
{

// Geometry of the Morton codes of an ND-dimensional tree: CB bits per coordinate (tree_fwd.hpp:141-150 of the
// reference: 21 for octrees, 31 for quadtrees), ND bits per level.
template <int ND>
struct geo {
    static_assert(ND == 2 || ND == 3);
    static constexpr unsigned CB = ND == 3 ? 21u : 31u;
    static constexpr unsigned DB = static_cast<unsigned>(ND);
    static constexpr unsigned DMASK = (1u << DB) - 1u;
    static constexpr unsigned SLACK = 64u - CB * DB; // unused high bits of a particle code
};

__host__ __device__ inline uint64_t spread3(uint64_t v)
{
    v &= 0x1fffffULL;
    v = (v | (v << 32)) & 0x1f00000000ffffULL;
    v = (v | (v << 16)) & 0x1f0000ff0000ffULL;
    v = (v | (v << 8)) & 0x100f00f00f00f00fULL;
    v = (v | (v << 4)) & 0x10c30c30c30c30c3ULL;
    v = (v | (v << 2)) & 0x1249249249249249ULL;
    return v;
}
__host__ __device__ inline uint64_t compact3(uint64_t v)
{
    v &= 0x1249249249249249ULL;
    v = (v ^ (v >> 2)) & 0x10c30c30c30c30c3ULL;
    v = (v ^ (v >> 4)) & 0x100f00f00f00f00fULL;
    v = (v ^ (v >> 8)) & 0x1f0000ff0000ffULL;
    v = (v ^ (v >> 16)) & 0x1f00000000ffffULL;
    v = (v ^ (v >> 32)) & 0x1fffffULL;
    return v;
}

__host__ __device__ inline uint64_t spread2(uint64_t v)
{
    v &= 0xffffffffULL;
    v = (v | (v << 16)) & 0x0000ffff0000ffffULL;
    v = (v | (v << 8)) & 0x00ff00ff00ff00ffULL;
    v = (v | (v << 4)) & 0x0f0f0f0f0f0f0f0fULL;
    v = (v | (v << 2)) & 0x3333333333333333ULL;
    v = (v | (v << 1)) & 0x5555555555555555ULL;
    return v;
}
__host__ __device__ inline uint64_t compact2(uint64_t v)
{
    v &= 0x5555555555555555ULL;
    v = (v ^ (v >> 1)) & 0x3333333333333333ULL;
    v = (v ^ (v >> 2)) & 0x0f0f0f0f0f0f0f0fULL;
    v = (v ^ (v >> 4)) & 0x00ff00ff00ff00ffULL;
    v = (v ^ (v >> 8)) & 0x0000ffff0000ffffULL;
    v = (v ^ (v >> 16)) & 0xffffffffULL;
    return v;
}
template <int ND>
__host__ __device__ inline uint64_t morton_coord(uint64_t code, unsigned j)
{
    if constexpr (ND == 3) {
        return compact3(code >> j);
    } else {
        return compact2(code >> j);
    }
}

__device__ inline float d_fma(float a, float b, float c)
{
    return __builtin_fmaf(a, b, c);
}
__device__ inline double d_fma(double a, double b, double c)
{
    return __builtin_fma(a, b, c);
}

// Control block of one build: everything the host needs to know, fetched with one copy per synchronisation point.
struct ctrl_block {
    unsigned long long maxbits; // bit pattern of max |coordinate| (non-negative IEEE values order like integers)
    double box;                 // domain size actually used
    unsigned err;               // bit 0: non-finite coordinate while deducing the box, 1: non-finite COM,
                                // 2: non-finite node dimension, 3: non-finite deduced box
    unsigned bad_inv;           // ~(index of the first particle that cannot be discretised), 0 = none
    unsigned n_nonroot;         // number of nodes besides the root
    unsigned n_crit, n_int, n_children; // critical nodes, internal nodes, sum of child counts
    unsigned max_group;         // particles in the largest critical node
    unsigned class2_count[8];   // critical nodes per lane-mapping class (list kernel binning)
    unsigned max_level;         // deepest leaf level of the tree (the next rebuild sorts the code bits of that many levels + 1 only)
    unsigned pad[4];
    unsigned first_grid[4];     // light-tail arrangement of a first call: 8 x the longest per-region queue of each wave-kernel class
};
enum { ERR_COORD = 1u, ERR_COM = 2u, ERR_DIM = 4u, ERR_BOX = 8u };

// ---- box size -------------------------------------------------------------------------------------------
template <typename F>
__global__ void __launch_bounds__(1024) k_maxabs(const F *x, const F *y, const F *z, uint32_t n, ctrl_block *ctrl)
{
    unsigned long long *out_bits = &ctrl->maxbits;
    unsigned *err = &ctrl->err;
    F mx = F(0);
    bool bad = false;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const F a = fabs(x[i]), b = fabs(y[i]), c = z ? fabs(z[i]) : F(0);
        bad |= !(isfinite(a) && isfinite(b) && isfinite(c));
        mx = fmax(mx, fmax(a, fmax(b, c)));
    }
    // Non-negative IEEE values order like their bit patterns.
    unsigned long long bits;
    if constexpr (sizeof(F) == 4) {
        bits = __float_as_uint(static_cast<float>(mx));
    } else {
        bits = static_cast<unsigned long long>(__double_as_longlong(static_cast<double>(mx)));
    }
    // One atomic per block (same-address 64-bit atomics serialise at ~10 ns each).
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long other = __shfl_xor(bits, o, 64);
        bits = other > bits ? other : bits;
    }
    __shared__ unsigned long long s_bits[16];
    __shared__ int s_bad[16];
    const unsigned w = threadIdx.x >> 6;
    const bool wave_bad = __ballot(bad) != 0ull;
    if ((threadIdx.x & 63u) == 0u) {
        s_bits[w] = bits;
        s_bad[w] = wave_bad;
    }
    __syncthreads();
    if (threadIdx.x == 0u) {
        for (unsigned k = 1; k < blockDim.x / 64u; ++k) {
            bits = s_bits[k] > bits ? s_bits[k] : bits;
            s_bad[0] |= s_bad[k];
        }
        if (s_bad[0]) {
            atomicOr(err, static_cast<unsigned>(ERR_COORD));
        }
        atomicMax(out_bits, bits);
    }
}

// ---- discretise + encode ----------------------------------------------------------------------------------
template <typename F, int ND>
__global__ void k_encode(const F *x, const F *y, const F *z, uint32_t n, ctrl_block *ctrl, F box_in, uint64_t *codes,
                         uint32_t *idx)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) {
        return;
    }
    // Domain size: given, or 2 * max|coord| + 5% (tree.hpp:1306-1312 of the reference) -- every thread derives it from the
    // reduction's result (three operations), thread 0 records it for the host: no one-thread kernel in between.
    F box = box_in;
    if (box_in == F(0)) {
        F mx;
        if constexpr (sizeof(F) == 4) {
            mx = __uint_as_float(static_cast<unsigned>(ctrl->maxbits));
        } else {
            mx = __longlong_as_double(static_cast<long long>(ctrl->maxbits));
        }
        const F b = mx * F(2);
        box = d_fma(b, F(1) / F(20), b);
    }
    if (i == 0u) {
        if (!isfinite(box)) {
            atomicOr(&ctrl->err, static_cast<unsigned>(ERR_BOX));
        }
        ctrl->box = static_cast<double>(box);
    }
    const F inv_box = F(1) / box;
    constexpr F factor = F(1u << geo<ND>::CB);
    uint64_t d[3] = {};
    const F v[3] = {x[i], y[i], ND == 3 ? z[i] : F(0)};
    bool bad = false;
#pragma unroll
    for (int k = 0; k < ND; ++k) {
        F tmp = d_fma(v[k], inv_box, F(0.5));
        tmp *= factor;
        if (!isfinite(tmp) || tmp < F(0) || tmp >= factor) {
            bad = true;
            tmp = F(0);
        }
        d[k] = static_cast<uint64_t>(tmp);
    }
    if (bad) {
        atomicMax(&ctrl->bad_inv, ~i);
    }
    if constexpr (ND == 3) {
        codes[i] = spread3(d[0]) | (spread3(d[1]) << 1) | (spread3(d[2]) << 2);
    } else {
        codes[i] = spread2(d[0]) | (spread2(d[1]) << 1);
    }
    idx[i] = i;
}

template <typename F>
__global__ void k_permute(const F *x, const F *y, const F *z, const F *m, const uint32_t *order, uint32_t n,
                          typename vt<F>::v4 *part4)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) {
        return;
    }
    const uint32_t s = order[i];
    typename vt<F>::v4 p;
    p.x = x[s], p.y = y[s], p.z = z ? z[s] : F(0), p.w = m[s];
    part4[i] = p;
}

// ---- topology ---------------------------------------------------------------------------------------------
// Range [lo, hi) of the particles sharing the level-`lvl` cell of particle i, searched inside the parent range.
template <int ND>
__device__ inline void narrow(const uint64_t *codes, uint32_t i, unsigned lvl, uint32_t &lo, uint32_t &hi)
{
    const unsigned shift = geo<ND>::DB * (geo<ND>::CB - lvl);
    const uint64_t p = codes[i] >> shift;
    uint32_t a = lo, b = i; // first j in [lo, i] with prefix == p
    while (a < b) {
        const uint32_t mid = a + (b - a) / 2u;
        if ((codes[mid] >> shift) < p) {
            a = mid + 1u;
        } else {
            b = mid;
        }
    }
    lo = a;
    a = i + 1u;
    b = hi; // first j in (i, hi] with prefix > p
    while (a < b) {
        const uint32_t mid = a + (b - a) / 2u;
        if ((codes[mid] >> shift) <= p) {
            a = mid + 1u;
        } else {
            b = mid;
        }
    }
    hi = a;
}

// Number of leading ND-bit digits (levels) two codes share: 0..CB. Octree codes use bits 0..62, quadtree codes 0..61.
template <int ND>
__device__ inline unsigned common_levels(uint64_t a, uint64_t b)
{
    const uint64_t xr = a ^ b;
    return xr ? (static_cast<unsigned>(__clzll(static_cast<long long>(xr))) - geo<ND>::SLACK) / geo<ND>::DB : geo<ND>::CB;
}

// Depth of the leaf holding each particle, by search (any max_leaf_n): descend while the cell holds too many.
template <int ND>
__global__ void k_leaf_levels_search(const uint64_t *codes, uint32_t n, uint32_t max_leaf_n, uint8_t *leaf)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) {
        return;
    }
    uint32_t lo = 0, hi = n;
    unsigned lvl = 0;
    while (hi - lo > max_leaf_n && lvl < geo<ND>::CB) {
        ++lvl;
        narrow<ND>(codes, i, lvl, lo, hi);
    }
    leaf[i] = static_cast<uint8_t>(lvl);
}

// Same result without searching, for small max_leaf_n = m: the level-L cell of particle i holds more than m
// particles iff some window of m + 1 consecutive (sorted) particles containing i shares its first L digits, so
//   leaf(i) = min(CBITS, 1 + max_{j in [i-m, i], j+m < n} common_levels(c[j], c[j+m]))      (0 without windows).
// win[j] = common_levels(c[j], c[j+m]) + 1 for a valid window, 0 otherwise. A block computes the windows its 256 particles look
// at (the m in front of it included) into LDS and takes the maxima from there. (Rounds 2-5: one launch for win[], one for the
// maxima, 11 + 21 us at 4M particles against 14.)
// The same launch takes the node counts of k_node_counts (each particle's own leaf level and its two neighbouring codes are all they
// need) and puts the particles into tree order (k_permute: nothing to do with the levels, but one launch less).
// block_max[block] = deepest leaf level among the block's particles (for the host: the next rebuild's partial sort). order == null:
// the particles are put into tree order elsewhere (k_local_sort).
template <typename F, int ND>
__global__ void __launch_bounds__(256) k_leaf_levels_windows(const uint64_t *codes, uint32_t n, uint32_t m, uint8_t *leaf, uint8_t *ldiv,
                                                             uint32_t *cnt, const F *px, const F *py, const F *pz, const F *pm,
                                                             const uint32_t *order, typename vt<F>::v4 *part4, uint8_t *block_max)
{
    __shared__ uint8_t s_win[256 + 64];
    __shared__ unsigned s_any;
    const uint32_t base = blockIdx.x * 256u;
    if (threadIdx.x == 0u) {
        s_any = 1u;
    }
    if (order && base + threadIdx.x < n) {
        const uint32_t src = order[base + threadIdx.x];
        typename vt<F>::v4 q;
        q.x = px[src], q.y = py[src], q.z = pz ? pz[src] : F(0), q.w = pm[src];
        part4[base + threadIdx.x] = q;
    }
    for (uint32_t t = threadIdx.x; t < 256u + m; t += 256u) {
        // window j = base - m + t (none in front of particle 0)
        const bool have = base + t >= m;
        const uint32_t j = base + t - m;
        s_win[t] = (have && m < n && j < n - m) ? static_cast<uint8_t>(common_levels<ND>(codes[j], codes[j + m]) + 1u) : uint8_t(0);
    }
    __syncthreads();
    const uint32_t i = base + threadIdx.x;
    unsigned lvl = 0u;
    if (i < n) {
        unsigned best = 0;
        for (uint32_t t = threadIdx.x; t <= threadIdx.x + m; ++t) { // windows i - m .. i
            best = max(best, static_cast<unsigned>(s_win[t]));
        }
        lvl = min(best, geo<ND>::CB);
        leaf[i] = static_cast<uint8_t>(lvl);
        const unsigned dv = i > 0 ? common_levels<ND>(codes[i - 1], codes[i]) + 1u : 1u; // identical codes never start a node
        ldiv[i] = static_cast<uint8_t>(dv);
        cnt[i] = dv <= lvl ? lvl - dv + 1u : 0u;
        if (i == 0u) {
            cnt[n] = 0u; // the scan runs over n + 1 values so that off[n] is the total
        }
    }
    // Levels are below 32: the OR of (1 << level) over the block carries the maximum (a DPP reduction per wavefront, one LDS atomic each).
    const unsigned any = wave_reduce_or(1u << (lvl & 31u));
    if ((threadIdx.x & 63u) == 0u) {
        atomicOr(&s_any, any);
    }
    __syncthreads();
    if (threadIdx.x == 0u) {
        block_max[blockIdx.x] = static_cast<uint8_t>(31u - static_cast<unsigned>(__clz(static_cast<int>(s_any))));
    }
}

// Completes a sort that looked at the code bits of the first `L` levels only (sort_codes with begin_bit > 0), given that no leaf is
// deeper than L: particles of different leaves are in their final relative order already (their codes differ inside those bits), the
// particles of one leaf are contiguous but in the order the stable partial sort left them in -- by original index. Every particle
// finds its leaf's range by walking to both sides while the codes share the leaf's level (a leaf above the deepest level holds at
// most max_leaf_n particles), counts the members in front of it by (code, index) and moves its code, index and record there.
// A leaf of the deepest level holds identical codes only: nothing to do. (If a leaf IS deeper than L the result is garbage
// inside the arrays' bounds; the host sees the deepest level with the first look-up and builds again with a full sort.)
template <typename F, int ND>
__global__ void __launch_bounds__(256) k_local_sort(const uint64_t *codes, const uint32_t *order, const uint8_t *leaf, uint32_t n, uint32_t span,
                                                    const F *px, const F *py, const F *pz, const F *pm, uint64_t *codes_out,
                                                    uint32_t *order_out, typename vt<F>::v4 *part4)
{
    // The block's codes and indices with `span` (<= 64) neighbours on either side, in LDS: the walks read them from there.
    __shared__ uint64_t s_code[256 + 128];
    __shared__ uint32_t s_ord[256 + 128];
    const uint32_t base = blockIdx.x * 256u;
    for (uint32_t t = threadIdx.x; t < 256u + 2u * span; t += 256u) {
        const bool have = base + t >= span && base + t - span < n;
        const uint32_t j = base + t - span;
        s_code[t] = have ? codes[j] : ~0ull; // (no level in common with any code: bit 63 is never set)
        s_ord[t] = have ? order[j] : 0u;
    }
    __syncthreads();
    const uint32_t i = base + threadIdx.x;
    if (i >= n) {
        return;
    }
    const uint32_t me = threadIdx.x + span;
    const uint64_t c = s_code[me];
    const uint32_t o = s_ord[me];
    const unsigned lf = leaf[i];
    uint32_t dest = i;
    if (lf < geo<ND>::CB) {
        // Same leaf <=> the first lf levels of the codes agree <=> nothing of (cj ^ c) is left above the bits of the levels below lf.
        // (The filler ~0 has bit 63 set, no code has: it never matches.)
        const unsigned shift = geo<ND>::DB * (geo<ND>::CB - lf);
        uint32_t before = 0, first = i;
        for (uint32_t k = 1; k <= span; ++k) {
            const uint64_t cj = s_code[me - k];
            if (((cj ^ c) >> shift) != 0ull) {
                break;
            }
            first = i - k;
            before += (cj < c || (cj == c && s_ord[me - k] < o)) ? 1u : 0u;
        }
        for (uint32_t k = 1; k <= span; ++k) {
            const uint64_t cj = s_code[me + k];
            if (((cj ^ c) >> shift) != 0ull) {
                break;
            }
            before += (cj < c || (cj == c && s_ord[me + k] < o)) ? 1u : 0u;
        }
        dest = first + before;
    }
    codes_out[dest] = c;
    order_out[dest] = o;
    typename vt<F>::v4 q;
    q.x = px[o], q.y = py[o], q.z = pz ? pz[o] : F(0), q.w = pm[o];
    part4[dest] = q;
}

// ldiv[i] = first level at which c[i] leaves the cell of c[i-1]; cnt[i] = number of nodes whose first particle is i.
template <int ND>
__global__ void k_node_counts(const uint64_t *codes, uint32_t n, const uint8_t *leaf, uint8_t *ldiv, uint32_t *cnt)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        // Identical codes never start a node.
        const unsigned dv = i > 0 ? common_levels<ND>(codes[i - 1], codes[i]) + 1u : 1u;
        const unsigned lvl = leaf[i];
        ldiv[i] = static_cast<uint8_t>(dv);
        cnt[i] = dv <= lvl ? lvl - dv + 1u : 0u;
        if (i == 0u) {
            cnt[n] = 0u; // the scan runs over n + 1 values so that off[n] is the total
        }
    }
}
// (Rounds 2-5 also reduced the deepest leaf level here, for the host to skip the empty level passes of the node sums: first with one
// atomicMax per block on one word of the control block -- 15 600 same-address atomics, 150 us at 4M particles --, then with an
// agent-scope load of that word per block, 62 us against 14 for the kernel's own work. The level passes are gone; the deepest level
// -- now the hint for the next rebuild's partial sort -- is taken by the leaf-level kernel, one plain store per block.)

// One block: the node count (total of the scan) and the deepest leaf level (maximum over the leaf-level kernel's blocks; block_max ==
// null: not known, the deepest possible) for the host.
__global__ void __launch_bounds__(256) k_pack_nodes(ctrl_block *ctrl, const uint32_t *off_n, const uint8_t *block_max, uint32_t n_blocks,
                                                    unsigned deepest)
{
    __shared__ unsigned s_any;
    if (threadIdx.x == 0u) {
        s_any = 1u;
        ctrl->n_nonroot = *off_n;
    }
    __syncthreads();
    unsigned any = 1u;
    if (block_max) {
        // Sixteen bytes per load (the array comes from the block cache, 16-byte aligned).
        const uint32_t n16 = n_blocks / 16u;
        const auto *v = reinterpret_cast<const uint4 *>(block_max);
        for (uint32_t g = threadIdx.x; g < n16; g += blockDim.x) {
            const uint4 q = v[g];
            const unsigned w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                any |= (1u << (w[j] & 31u)) | (1u << ((w[j] >> 8) & 31u)) | (1u << ((w[j] >> 16) & 31u)) | (1u << ((w[j] >> 24) & 31u));
            }
        }
        for (uint32_t b = n16 * 16u + threadIdx.x; b < n_blocks; b += blockDim.x) {
            any |= 1u << (block_max[b] & 31u);
        }
    } else {
        any = 1u << (deepest & 31u);
    }
    atomicOr(&s_any, any);
    __syncthreads();
    if (threadIdx.x == 0u) {
        ctrl->max_level = 31u - static_cast<unsigned>(__clz(static_cast<int>(s_any)));
    }
}

// ---- emit the nodes: one thread per NODE ----
// off[] = exclusive scan of cnt[] (off[n] = number of non-root nodes). The nodes whose first particle is i are nested (levels
// ldiv(i)..leaf(i)) and consecutive in depth-first order from 1 + off[i]. k_node_starts writes, for every node, its first particle
// (the thread of a first particle fills the 1..12 slots of its nest -- no searching); k_emit_per_node then finds the end of that one
// node by galloping from its first particle. Rounds 2-5 gave the whole nest to the thread of the first particle, which found the ends
// one after the other, each search starting where the previous one ended -- fewer probes in all, but the first particle of a large
// cell walks up a dozen levels and its wavefront waits for that lane: 128 us at 4M particles against 9 + 35 (rebuild 0.945 -> 0.853 ms,
// 1M 0.462 -> 0.425; tools/archive/jobs_r05/r05_job47.sh).
__global__ void k_node_starts(uint32_t n, const uint8_t *leaf, const uint8_t *ldiv, const uint32_t *off, uint32_t *start_of)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) {
        return;
    }
    const unsigned lf = leaf[i], dv = ldiv[i];
    if (dv > lf) {
        return;
    }
    const uint32_t o = off[i];
    for (unsigned l = dv; l <= lf; ++l) {
        start_of[o + (l - dv)] = i;
    }
}
template <int ND>
__global__ void k_emit_per_node(const uint64_t *codes, uint32_t n, const uint8_t *ldiv, const uint32_t *off, const uint32_t *start_of,
                                uint32_t n_nodes, uint4 *topo, uint64_t *ncode, uint32_t *parent)
{
    const uint32_t d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= n_nodes) {
        return;
    }
    if (d == 0u) {
        topo[0] = make_uint4(off[n], 0u, n, 0u);
        ncode[0] = 1ull;
        parent[0] = 0xffffffffu;
        return;
    }
    const uint32_t i = start_of[d - 1u];
    const unsigned dv = ldiv[i];
    const unsigned lvl = dv + (d - 1u - off[i]);
    const uint64_t ci = codes[i];
    const unsigned shift = geo<ND>::DB * (geo<ND>::CB - lvl);
    const uint64_t p = ci >> shift;
    // Smallest j > i with j == n or a different level-lvl prefix. Both phases are 8-ary: the gallop grows its stride by 8, the search
    // issues seven independent probes per step -- a third of the dependent steps of the binary forms for 2.3 x the loads.
    uint32_t a = i + 1u, b, step = 1u;
    for (;;) {
        b = a + (step - 1u);
        if (b >= n || b < a) {
            b = n;
            break;
        }
        if ((codes[b] >> shift) != p) {
            break;
        }
        a = b + 1u;
        step = step > (1u << 28) ? step : step << 3;
    }
    while (b - a >= 8u) {
        const uint32_t w = (b - a) >> 3;
        uint64_t c[7];
#pragma unroll
        for (uint32_t k = 0; k < 7u; ++k) {
            c[k] = codes[a + w * (k + 1u)];
        }
        uint32_t na = a, nb = b;
        bool found = false;
#pragma unroll
        for (uint32_t k = 0; k < 7u; ++k) {
            const bool same = (c[k] >> shift) == p;
            if (!found) {
                if (same) {
                    na = a + w * (k + 1u) + 1u;
                } else {
                    nb = a + w * (k + 1u);
                    found = true;
                }
            }
        }
        a = na, b = nb;
    }
    {
        uint32_t run = 0;
        bool open = true;
#pragma unroll
        for (uint32_t k = 0; k < 7u; ++k) {
            const uint32_t j = a + k;
            const bool same = j < b && (codes[j < n ? j : n - 1u] >> shift) == p;
            open = open && same;
            run += open ? 1u : 0u;
        }
        a += run;
    }
    const uint32_t hi = a;
    const uint32_t next = 1u + off[hi]; // depth-first index of the first node starting at or after hi
    topo[d] = make_uint4(next - d - 1u, i, hi, 0u);
    ncode[d] = (1ull << (geo<ND>::DB * lvl)) | p;
}

template <int ND>
__device__ inline unsigned level_of(uint64_t code)
{
    return (63u - static_cast<unsigned>(__clzll(static_cast<long long>(code)))) / geo<ND>::DB;
}

// parent[] of every non-root node and the child-octant mask of every node, both written by the parent (children of k: k + 1,
// then skipping subtrees). (Rounds 2-5 had every child atomicOr its octant into the parent's mask in k_flags and a k_popc pass
// count the bits: 1.4M atomics on 0.35M words and one launch more, 42 + 6 us at 4M particles.)
template <typename F, int ND, bool LEAF_SUMS>
__global__ void k_parents(const uint4 *topo, const uint64_t *ncode, uint32_t n_nodes, uint32_t *parent, uint32_t *mask,
                          const typename vt<F>::v4 *part4, typename vt<F>::v4 *sums)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_nodes) {
        return;
    }
    if (k == 0u) {
        mask[n_nodes] = 0u;
    }
    const uint4 t = topo[k];
    if (LEAF_SUMS && t.x == 0u) {
        // A leaf has no children to walk; its thread sums the leaf's particles instead, serially in particle order (tree.hpp:1162-1168
        // of the reference) -- one launch for the two.
        F mt = F(0), sx = F(0), sy = F(0), sz = F(0);
        for (uint32_t i = t.y; i < t.z; ++i) {
            const typename vt<F>::v4 q = part4[i];
            mt += q.w;
            sx = d_fma(q.w, q.x, sx);
            sy = d_fma(q.w, q.y, sy);
            sz = d_fma(q.w, q.z, sz);
        }
        typename vt<F>::v4 r;
        r.x = sx, r.y = sy, r.z = sz, r.w = mt;
        sums[k] = r;
    }
    const uint32_t last = k + t.x;
    uint32_t m = 0;
    for (uint32_t c = k + 1u; c <= last; c += topo[c].x + 1u) {
        parent[c] = k;
        m |= 1u << (static_cast<unsigned>(ncode[c]) & geo<ND>::DMASK);
    }
    mask[k] = m;
}

// ---- node properties --------------------------------------------------------------------------------------
// (Rounds 2-5 aggregated children into parents with one launch per tree level -- 9 launches at 100k particles, 11 at 4M, ~5 us each
// however little a level holds; the summation pyramid further down replaced them. Variants of the level passes built and measured
// before that, none kept -- tools/archive/jobs_r05/r05_job37.sh, r05_job38.sh: (1) leaf sums and all levels in ONE launch,
// the last child to deliver its sum adds up the parent (atomic counters): on eight XCDs every release / acquire pair is an L2
// write-back + invalidate, 2.1 ms instead of 0.07 at 4M particles; (2) two or three levels per launch, the upper ones recomputing
// the sums of their internal children instead of reading them: the walk over a node's children is a chain of dependent loads
// (c += topo[c].x + 1), nested it is 72 deep -- rebuild +0.03 ms at 100k, +0.07 at 4M for two levels, +0.13 / +0.35 for three.
// (3) the top five levels in ONE launch of one workgroup (the internal nodes found from the root through the child table into LDS
// queues, then summed level by level between barriers): 24 us at 100k particles and 52 us at 4M for the five ~5 us passes it replaced --
// a lone workgroup pays every dependent load in full (tools/archive/jobs_r05/r05_job50.sh, r05_job51.sh).)

// ---- node sums in the reference's association (exact mode) ----
// The reference sums a node's particles serially in particle order (tree.hpp:1162-1168). Nodes that start at the same
// particle are nested -- a node, its first child, that one's first child, ... down to a leaf; consecutive in depth-first
// order -- and the serial sum of each is a PREFIX of the serial sum of the outermost one. So one chain per distinct first
// particle yields them all: it starts from the leaf's sum, walks on from the leaf's last particle to the last particle of
// the outermost node ("head": a node that is not the first child of its parent), one fused multiply-add per particle and
// component, in order, and drops a sum at the end of every node of the nest. Chains of different heads share nothing, so
// all of them run at once; the build takes as long as the longest, the root's N links (10.5 cycles per dependent
// v_fma_f32 of a lone wavefront, 12-14.5 when fed from LDS: tools/ubench/chain_rate.hip), instead of the sum of the longest
// chain of every tree level (round 2: 58 ms at 4M -- in a centrally concentrated system eleven levels each own nodes whose
// bulk sits in their LAST child, 3 ms of chain per level on top of the root's 21). Identical bits to the host builders.
constexpr uint32_t EXACT_WAVE_MIN = 1024; // links from which a chain gets a workgroup of its own
constexpr uint32_t EXACT_NO_ROOT = 0xffffffffu;

__device__ inline bool exact_is_head(const uint4 *topo, uint32_t k)
{
    return topo[k].x != 0u && (k == 0u || topo[k - 1u].y != topo[k].y);
}

// Short chains: one thread per head. Long ones are listed for k_exact_chains_wave (the root's in slot 0: it is the longest
// and must be dispatched first).
template <typename F>
__global__ void k_exact_chains(const uint4 *topo, uint32_t n_nodes, const typename vt<F>::v4 *part4, typename vt<F>::v4 *sums,
                               uint32_t *list, uint32_t *count)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_nodes || !exact_is_head(topo, k)) {
        return;
    }
    uint32_t leaf = k + 1u;
    while (topo[leaf].x != 0u) {
        ++leaf;
    }
    uint32_t i = topo[leaf].z;
    if (topo[k].z - i >= EXACT_WAVE_MIN) {
        list[k == 0u ? 0u : atomicAdd(count, 1u)] = k;
        return;
    }
    typename vt<F>::v4 s = sums[leaf];
    for (uint32_t j = leaf; j-- > k;) {
        const uint32_t end = topo[j].z;
        // Eight particles are loaded together: one wait for memory per eight links.
        for (; i + 8u <= end; i += 8u) {
            typename vt<F>::v4 p[8];
#pragma unroll
            for (uint32_t u = 0; u < 8u; ++u) {
                p[u] = part4[i + u];
            }
#pragma unroll
            for (uint32_t u = 0; u < 8u; ++u) {
                s.w += p[u].w;
                s.x = d_fma(p[u].w, p[u].x, s.x);
                s.y = d_fma(p[u].w, p[u].y, s.y);
                s.z = d_fma(p[u].w, p[u].z, s.z);
            }
        }
        for (; i < end; ++i) {
            const typename vt<F>::v4 p = part4[i];
            s.w += p.w;
            s.x = d_fma(p.w, p.x, s.x);
            s.y = d_fma(p.w, p.y, s.y);
            s.z = d_fma(p.w, p.z, s.z);
        }
        sums[j] = s;
    }
}

// One workgroup (4 wavefronts) per listed head. The chain itself is serial: lanes 0-3 of wave 0 carry the four sums
// {m x, m y, m z, m}, one dependent fused multiply-add per particle and lane (the mass sum is fma(m, 1, sum) = m + sum
// exactly). Everything else feeds it: waves 1-3 load the NEXT super-chunk of 2048 particles (coalesced) into registers
// while wave 0 consumes the current one from LDS (component-major rows, one 16-byte LDS read per four operands), then the
// registers go to the other LDS buffer. The loads of 2048 particles are in flight during the microseconds the chain needs
// for 2048 particles, so the chain never waits for memory.
template <typename F>
__global__ void __launch_bounds__(256) k_exact_chains_wave(const uint4 *topo, const uint32_t *list, const uint32_t *count,
                                                           const typename vt<F>::v4 *part4, typename vt<F>::v4 *sums)
{
    using v4 = typename vt<F>::v4;
    constexpr uint32_t SC = 2048; // particles per super-chunk
    constexpr uint32_t CG = sizeof(F) == 4 ? 8 : 4, CL = 4 * CG; // a burst of the chain: CG four-operand LDS reads per row, CL links
    __shared__ __attribute__((aligned(32))) F s_tile[2][4][SC + CL]; // per buffer: rows x, y, z, m (+ CL: read-ahead past the end)
    __shared__ __attribute__((aligned(32))) F s_ones[8];
    const uint32_t e = blockIdx.x;
    if (e >= *count) {
        return;
    }
    const uint32_t k = list[e];
    if (k == EXACT_NO_ROOT) {
        return;
    }
    const int tid = static_cast<int>(threadIdx.x), lane = tid & 63;
    if (tid < 8) {
        s_ones[tid] = F(1);
    }
    uint32_t leaf = k + 1u;
    while (topo[leaf].x != 0u) {
        ++leaf;
    }
    const int comp = lane & 3;
    F sum = reinterpret_cast<const F *>(&sums[leaf])[comp];
    // Wavefronts 1-3 (192 threads) move the particles; wavefront 0 has no memory loads of its own in flight, so nothing but
    // the chain's own LDS reads ever makes it wait.
    constexpr uint32_t NF = 192u, PER = (SC + NF - 1u) / NF; // 11 particles per fetching thread
    const uint32_t ft = static_cast<uint32_t>(tid) - 64u;
    v4 nxt[PER];
    uint32_t start = topo[leaf].z;
    for (uint32_t j = leaf; j-- > k;) {
        const uint32_t end = topo[j].z;
        if (end > start) {
            auto fetch = [&](uint32_t base) {
                if (tid >= 64) {
#pragma unroll
                    for (uint32_t u = 0; u < PER; ++u) {
                        const uint32_t jj = u * NF + ft, i = base + jj;
                        if (jj < SC) {
                            nxt[u] = part4[i < end ? i : end - 1u];
                        }
                    }
                }
            };
            auto stash = [&](int buf) {
                if (tid >= 64) {
#pragma unroll
                    for (uint32_t u = 0; u < PER; ++u) {
                        const uint32_t jj = u * NF + ft;
                        if (jj < SC) {
                            s_tile[buf][0][jj] = nxt[u].x, s_tile[buf][1][jj] = nxt[u].y, s_tile[buf][2][jj] = nxt[u].z,
                            s_tile[buf][3][jj] = nxt[u].w;
                        }
                    }
                }
            };
            fetch(start);
            stash(0);
            __syncthreads();
            int buf = 0;
            for (uint32_t base = start; base < end; base += SC, buf ^= 1) {
                const bool more = base + SC < end;
                if (more) {
                    fetch(base + SC); // in flight while wave 0 walks the chain
                }
                if (tid < 64) {
                    const uint32_t cnt = end - base < SC ? end - base : SC;
                    const F *row_m = s_tile[buf][3];
                    // The mass chain multiplies by one: its "coordinate" row is a row of ones (re-read, never advanced).
                    const F *row_c = comp == 3 ? s_ones : s_tile[buf][comp];
                    const uint32_t cstep = comp == 3 ? 0u : 1u;
                    // The operands of the NEXT burst are read from LDS into a second register set before the dependent
                    // multiply-adds of the current one are issued (two sets, swapped by unrolling; the scheduling barriers
                    // keep the compiler from sinking the reads back to their first use). A lone wavefront starts a dependent
                    // v_fma_f32 every 10.5 cycles, but only while the multiply-adds follow each other directly
                    // (tools/ubench/chain_rate.hip): interleaving the reads and address updates with them measured slower
                    // twice (sched_group_barrier {fma, read} x 4: the root's chain at 4M takes 32 ms; hand-written assembly:
                    // 30 ms; reads in a block of their own between bursts of 8: 22 ms), so the bursts are long: 32 links.
                    auto ld = [&](uint32_t i, v4 (&m)[CG], v4 (&c)[CG]) { // (rows are padded: i <= SC stays inside)
#pragma unroll
                        for (uint32_t g = 0; g < CG; ++g) {
                            m[g] = *reinterpret_cast<const v4 *>(row_m + i + 4u * g);
                            c[g] = *reinterpret_cast<const v4 *>(row_c + cstep * (i + 4u * g));
                        }
                    };
                    auto chain = [&](const v4 (&m)[CG], const v4 (&c)[CG]) {
#pragma unroll
                        for (uint32_t g = 0; g < CG; ++g) {
                            sum = d_fma(m[g].x, c[g].x, sum);
                            sum = d_fma(m[g].y, c[g].y, sum);
                            sum = d_fma(m[g].z, c[g].z, sum);
                            sum = d_fma(m[g].w, c[g].w, sum);
                        }
                    };
                    v4 am[CG], ac[CG], bm[CG], bc[CG];
                    const uint32_t cntb = cnt / CL * CL;
                    uint32_t i = 0;
                    if (cntb) {
                        ld(0u, am, ac);
                    }
                    for (; i + 2u * CL <= cntb; i += 2u * CL) {
                        ld(i + CL, bm, bc);
                        __builtin_amdgcn_sched_barrier(0);
                        chain(am, ac);
                        __builtin_amdgcn_sched_barrier(0);
                        ld(i + 2u * CL, am, ac);
                        __builtin_amdgcn_sched_barrier(0);
                        chain(bm, bc);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if (i < cntb) { // one more burst (already in the A set)
                        chain(am, ac);
                        i += CL;
                    }
                    for (; i < cnt; ++i) { // fewer than a burst left in this super-chunk
                        sum = d_fma(row_m[i], comp == 3 ? F(1) : row_c[i], sum);
                    }
                }
                if (more) {
                    stash(buf ^ 1); // the other buffer was consumed one iteration ago
                }
                __syncthreads();
            }
        }
        if (tid < 4) {
            reinterpret_cast<F *>(&sums[j])[tid] = sum;
        }
        start = end;
    }
}

// ---- node sums without level passes: a summation pyramid over the particles (default association, round 5) ----
// Level 0 is the particles' (m x, m y, m z, m) in tree order, level l holds the sums of aligned runs of 2^l of them: P[l][i] =
// P[l-1][2i] + P[l-1][2i+1]. The sum of a node -- of ANY range [a, b) of particles -- is then the sum of the O(log(b - a)) aligned
// runs the range decomposes into, taken in a fixed order (left ends by ascending level, then the right ends): no node needs another
// node's result, so every node takes its sum in k_finalize itself and the 9 (100k particles) to 11 (4M) level passes of
// k_up_sums -- ~5 us of launch each, however little they hold -- become the two or three launches that build the pyramid (nine
// levels per launch through LDS). Pairwise association: rounding errors grow with log n. (The exact mode keeps the reference's
// serial association and its chains.)
struct pyr_desc {
    uint32_t off[32]; // first entry of level l (1-based) in the pyramid array
    uint32_t cnt[32]; // entries of level l; cnt[0] = particles
    uint32_t levels;  // highest level
};
template <typename F>
__device__ inline typename vt<F>::v4 pyr_entry(const typename vt<F>::v4 *part4, const typename vt<F>::v4 *pyr, const pyr_desc &d,
                                               unsigned lvl, uint32_t i)
{
    if (lvl == 0u) {
        typename vt<F>::v4 q = part4[i];
        q.x *= q.w, q.y *= q.w, q.z *= q.w;
        return q;
    }
    return pyr[d.off[lvl] + i];
}
// Levels base + 1 .. base + 9 from level base: a block takes 512 entries of it.
template <typename F>
__global__ void __launch_bounds__(256) k_pyr_pass(const typename vt<F>::v4 *part4, typename vt<F>::v4 *pyr, const pyr_desc d, unsigned base)
{
    using v4 = typename vt<F>::v4;
    __shared__ v4 sm[256];
    const uint32_t t = threadIdx.x, in0 = blockIdx.x * 512u + 2u * t, n_in = d.cnt[base];
    v4 v;
    v.x = v.y = v.z = v.w = F(0);
    if (in0 < n_in) {
        v = pyr_entry<F>(part4, pyr, d, base, in0);
        if (in0 + 1u < n_in) {
            const v4 u = pyr_entry<F>(part4, pyr, d, base, in0 + 1u);
            v.x += u.x, v.y += u.y, v.z += u.z, v.w += u.w;
        }
    }
    uint32_t width = 256u; // entries of the current level this block holds
    for (unsigned l = base + 1u;; ++l) {
        const uint32_t idx = blockIdx.x * width + t;
        if (t < width && l <= d.levels && idx < d.cnt[l]) {
            pyr[d.off[l] + idx] = v;
        }
        if (width == 1u || l >= d.levels) {
            break;
        }
        sm[t] = v;
        __syncthreads();
        width >>= 1;
        if (t < width) {
            const v4 a = sm[2u * t], b = sm[2u * t + 1u];
            v.x = a.x + b.x, v.y = a.y + b.y, v.z = a.z + b.z, v.w = a.w + b.w;
        }
        __syncthreads();
    }
}
// Sum over the particles [a, b).
template <typename F>
__device__ inline typename vt<F>::v4 range_sum(const typename vt<F>::v4 *part4, const typename vt<F>::v4 *pyr, const pyr_desc &d,
                                               uint32_t a, uint32_t b)
{
    using v4 = typename vt<F>::v4;
    v4 left, right;
    left.x = left.y = left.z = left.w = F(0);
    right = left;
    for (unsigned l = 0; a < b; ++l) {
        if (a & 1u) {
            const v4 e = pyr_entry<F>(part4, pyr, d, l, a);
            left.x += e.x, left.y += e.y, left.z += e.z, left.w += e.w;
            ++a;
        }
        if (a < b && (b & 1u)) {
            --b;
            const v4 e = pyr_entry<F>(part4, pyr, d, l, b);
            right.x = e.x + right.x, right.y = e.y + right.y, right.z = e.z + right.z, right.w = e.w + right.w;
        }
        a >>= 1, b >>= 1;
    }
    left.x += right.x, left.y += right.y, left.z += right.z, left.w += right.w;
    return left;
}

template <typename F, int ND>
__device__ inline void finalize_node(uint32_t k, const uint64_t *ncode, const typename vt<F>::v4 s, F box, int mac,
                                     typename vt<F>::v4 *node_com, typename vt<F>::v2 *node_mac, ctrl_block *ctrl)
{
    const uint64_t code = ncode[k];
    const unsigned lvl = level_of<ND>(code);
    // Geometric centre (tree.hpp:452-482 of the reference).
    constexpr unsigned DB = geo<ND>::DB, CB = geo<ND>::CB;
    const uint64_t first_cell = (code - (1ull << (DB * lvl))) << (DB * (CB - lvl));
    const F node_dim = box / static_cast<F>(1ull << lvl);
    const F half_dim = node_dim * F(0.5), cell = box * (F(1) / static_cast<F>(1ull << CB));
    F ctr[3] = {F(0), F(0), F(0)}; // a quadtree lives in the z = 0 plane
#pragma unroll
    for (int j = 0; j < ND; ++j) {
        ctr[j] = d_fma(static_cast<F>(morton_coord<ND>(first_cell, static_cast<unsigned>(j))), cell,
                       half_dim - box * F(0.5));
    }
    F com[3];
    if (s.w == F(0)) {
        com[0] = ctr[0], com[1] = ctr[1], com[2] = ctr[2];
    } else {
        const F inv = F(1) / s.w;
        com[0] = s.x * inv, com[1] = s.y * inv, com[2] = s.z * inv;
    }
    if (!(isfinite(com[0]) && isfinite(com[1]) && isfinite(com[2]) && isfinite(s.w))) {
        atomicOr(&ctrl->err, static_cast<unsigned>(ERR_COM));
    }
    typename vt<F>::v4 c;
    c.x = com[0], c.y = com[1], c.z = com[2], c.w = s.w;
    node_com[k] = c;
    typename vt<F>::v2 mp;
    if (mac == RK_MAC_BH) {
        mp.x = node_dim * node_dim;
        mp.y = F(0);
    } else {
        mp.x = node_dim;
        F d2 = (com[0] - ctr[0]) * (com[0] - ctr[0]);
        d2 = d_fma(com[1] - ctr[1], com[1] - ctr[1], d2);
        d2 = d_fma(com[2] - ctr[2], com[2] - ctr[2], d2);
        mp.y = sqrt(d2);
    }
    if (!(isfinite(mp.x) && isfinite(mp.y))) {
        atomicOr(&ctrl->err, static_cast<unsigned>(ERR_DIM));
    }
    node_mac[k] = mp;
}

// ---- critical nodes, child masks, records -----------------------------------------------------------------
// Three counters scanned together: critical nodes, internal nodes, children.
struct tri {
    uint32_t a, b, c;
};
struct tri_sum {
    __host__ __device__ tri operator()(const tri &x, const tri &y) const
    {
        return tri{x.a + y.a, x.b + y.b, x.c + y.c};
    }
};

// MASKS_DONE: the child masks are complete (k_parents of the device build); the child count is taken here, no k_popc pass.
template <int ND, bool MASKS_DONE>
__device__ inline void flags_node(uint32_t k, const uint4 *topo, const uint64_t *ncode, const uint32_t *parent, uint32_t n_nodes,
                                  uint32_t ncrit_clamped, tri *flags, uint32_t *mask)
{
    auto cand = [&](uint32_t j) { return (topo[j].z - topo[j].y) <= ncrit_clamped || topo[j].x == 0u; };
    const bool c = cand(k);
    // A parent lies before its children in depth-first order. An index that does not (the sentinel convert_device() leaves in
    // nodes that no parent's walk reached -- a malformed host tree) is never dereferenced: such a node counts as a child of
    // nobody, the child count then misses it and the conversion is refused ("not every node is reachable from the root").
    const uint32_t par = k != 0u ? parent[k] : 0u;
    const bool orphan = k != 0u && par >= k;
    flags[k].a = (c && (k == 0u || orphan || !cand(par))) ? 1u : 0u;
    flags[k].b = topo[k].x != 0u ? 1u : 0u;
    if constexpr (MASKS_DONE) {
        flags[k].c = static_cast<uint32_t>(__popc(mask[k]));
        if (k == 0u) {
            flags[n_nodes] = tri{0u, 0u, 0u};
        }
    } else {
        if (k == 0u) {
            flags[n_nodes] = tri{0u, 0u, 0u};
        } else if (!orphan) {
            atomicOr(&mask[par], 1u << (static_cast<unsigned>(ncode[k]) & geo<ND>::DMASK));
        }
    }
}

template <int ND, bool MASKS_DONE = false>
__global__ void k_flags(const uint4 *topo, const uint64_t *ncode, const uint32_t *parent, uint32_t n_nodes,
                        uint32_t ncrit_clamped, tri *flags, uint32_t *mask)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n_nodes) {
        flags_node<ND, MASKS_DONE>(k, topo, ncode, parent, n_nodes, ncrit_clamped, flags, mask);
    }
}
// Device build: centre of mass / size of every node. PYR: the node takes its sum from the pyramid (sums = the pyramid array),
// otherwise from sums[k]. (Its own launch, behind the flags + scan whose counts the host waits for: it runs while the host looks.)
template <typename F, int ND, bool PYR>
__global__ void k_finalize(const uint4 *topo, const uint64_t *ncode, uint32_t n_nodes, const typename vt<F>::v4 *sums,
                           const typename vt<F>::v4 *part4, const pyr_desc d, F box, int mac, typename vt<F>::v4 *node_com,
                           typename vt<F>::v2 *node_mac, ctrl_block *ctrl)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n_nodes) {
        typename vt<F>::v4 sk;
        if constexpr (PYR) {
            sk = range_sum<F>(part4, sums, d, topo[k].y, topo[k].z);
        } else {
            sk = sums[k];
        }
        finalize_node<F, ND>(k, ncode, sk, box, mac, node_com, node_mac, ctrl);
    }
}

__global__ void k_popc(const uint32_t *mask, uint32_t n_nodes, tri *flags)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n_nodes) {
        flags[k].c = static_cast<uint32_t>(__popc(mask[k]));
    }
}

__global__ void k_pack_counts(ctrl_block *ctrl, const tri *total)
{
    ctrl->n_crit = total->a;
    ctrl->n_int = total->b;
    ctrl->n_children = total->c;
}

// (Also clears the child table k_records fills afterwards -- a memset's launch less: thread k takes words [k zper, (k + 1) zper).)
template <typename F>
__global__ void k_crit(const uint4 *topo, const tri *flags, const tri *offs, uint32_t n_nodes, uint4 *crit, uint32_t *child_tab,
                       uint32_t tab_words, uint32_t zper)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_nodes) {
        return;
    }
    for (size_t w = static_cast<size_t>(k) * zper, e = min(w + zper, static_cast<size_t>(tab_words)); w < e; ++w) {
        child_tab[w] = 0u;
    }
    if (!flags[k].a) {
        return;
    }
    const uint32_t g = offs[k].a, b = topo[k].y, e = topo[k].z;
    crit[g] = make_uint4(b, e, k, e - b);
}

// Tight bounding box of every critical node's particles: one wavefront per node (lanes stride over its particles, minima and
// maxima folded across the lanes; both are exact, so the boxes are those of a serial loop). Rounds 2-4 had one THREAD walk the up
// to ncrit particles of its node inside k_crit: 75 us at 4M particles against 8.
template <typename F>
__global__ void __launch_bounds__(256) k_crit_boxes(const uint4 *crit, uint32_t n_crit, const typename vt<F>::v4 *part4,
                                                    typename vt<F>::v4 *boxes)
{
    const uint32_t g = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (g >= n_crit) {
        return;
    }
    const uint32_t lane = threadIdx.x & 63u, b = crit[g].x, e = crit[g].y;
    typename vt<F>::v4 lo = part4[b], hi = lo; // (every lane starts from the first particle: idle lanes are neutral)
    for (uint32_t i = b + lane; i < e; i += 64u) {
        const typename vt<F>::v4 p = part4[i];
        lo.x = fmin(lo.x, p.x), lo.y = fmin(lo.y, p.y), lo.z = fmin(lo.z, p.z);
        hi.x = fmax(hi.x, p.x), hi.y = fmax(hi.y, p.y), hi.z = fmax(hi.z, p.z);
    }
    // Minima / maxima over the 64 lanes: the DPP network of wave_incl_scan() (row_shr 1, 2, 4, 8 inside the rows of 16, then
    // row_bcast 15 / 31 across them; lanes without a source keep their own value -- neutral for min and max), total in lane 63.
    // (Rounds 2-5: six xor-shuffles per component, 36 ds_bpermute round trips per node.)
    const auto fold = [](F v, auto op) __attribute__((always_inline)) {
        const auto step = [&](F x, auto ctrl, auto rows) __attribute__((always_inline)) {
            if constexpr (sizeof(F) == 4) {
                const int r = __builtin_amdgcn_update_dpp(__float_as_int(x), __float_as_int(x), decltype(ctrl)::value,
                                                          decltype(rows)::value, 0xf, false);
                return op(x, __int_as_float(r));
            } else {
                const long long b = __double_as_longlong(x);
                const int l = static_cast<int>(b), h = static_cast<int>(b >> 32);
                const int rl = __builtin_amdgcn_update_dpp(l, l, decltype(ctrl)::value, decltype(rows)::value, 0xf, false);
                const int rh = __builtin_amdgcn_update_dpp(h, h, decltype(ctrl)::value, decltype(rows)::value, 0xf, false);
                return op(x, __longlong_as_double((static_cast<long long>(rh) << 32) | static_cast<unsigned>(rl)));
            }
        };
        using std::integral_constant;
        v = step(v, integral_constant<int, 0x111>{}, integral_constant<int, 0xf>{});
        v = step(v, integral_constant<int, 0x112>{}, integral_constant<int, 0xf>{});
        v = step(v, integral_constant<int, 0x114>{}, integral_constant<int, 0xf>{});
        v = step(v, integral_constant<int, 0x118>{}, integral_constant<int, 0xf>{});
        v = step(v, integral_constant<int, 0x142>{}, integral_constant<int, 0xa>{});
        v = step(v, integral_constant<int, 0x143>{}, integral_constant<int, 0xc>{});
        return v;
    };
    const auto mn = [](F a, F b) { return fmin(a, b); };
    const auto mx = [](F a, F b) { return fmax(a, b); };
    lo.x = fold(lo.x, mn), lo.y = fold(lo.y, mn), lo.z = fold(lo.z, mn);
    hi.x = fold(hi.x, mx), hi.y = fold(hi.y, mx), hi.z = fold(hi.z, mx);
    if (lane == 63u) {
        lo.w = hi.w = F(0);
        boxes[2u * g] = lo;
        boxes[2u * g + 1u] = hi;
    }
}

template <typename F, int ND>
__global__ void k_records(uint4 *topo, const uint64_t *ncode, const uint32_t *parent, const uint32_t *mask,
                          const tri *offs, uint32_t n_nodes,
                          const typename vt<F>::v4 *node_com, const typename vt<F>::v2 *node_mac, node_rec<F> *recs,
                          uint32_t *child_tab)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_nodes) {
        return;
    }
    uint32_t rec = 0;
    if (k != 0u) {
        const uint32_t p = parent[k];
        const unsigned digit = static_cast<unsigned>(ncode[k]) & geo<ND>::DMASK;
        const uint32_t rank = static_cast<uint32_t>(__popc(mask[p] & ((1u << digit) - 1u)));
        rec = 1u + offs[p].c + rank; // children of p occupy records [1 + offs[p].c, ...)
        child_tab[static_cast<size_t>(offs[p].b) * 8u + rank] = k;
    }
    const uint4 t = topo[k];
    node_rec<F> r;
    r.com = node_com[k];
    r.mac = node_mac[k];
    r.dfs = k;
    r.nch = t.x;
    r.pad[0] = r.pad[1] = 0u;
    if (t.x != 0u) {
        r.a = 1u + offs[k].c;
        r.b = static_cast<uint32_t>(__popc(mask[k]));
        topo[k].w = offs[k].b;
    } else {
        r.a = t.y;
        r.b = t.z;
        topo[k].w = 0xffffffffu;
    }
    recs[rec] = r;
}

// ---- lane-mapping classes of the critical nodes (rk_common.hpp: class2_of), binned on the device -------------
// Stable multi-bin partition: per-block histograms, a serial scan per bin over the blocks, then a scatter that
// ranks every group inside its block with ballots. Group ids stay ascending inside each class (locality).
constexpr unsigned NBIN = 8;
static_assert(NBIN >= static_cast<unsigned>(n_classes));

// Launch order of the FIRST call on a small tree (at most FIRST_ORDER_MAX critical nodes): what the heavy-first launch plan of
// rk_launch.hip build_plan gives repeated calls, made here so that a time-stepping loop, whose every traversal is a first call, has
// it too (100k particles, one-launch producer / consumer kernel: 0.144 ms over the class lists read backwards, 0.101 over a sorted
// list) -- EIGHT QUEUES, one per XCD region (eighths of the particle range; the critical nodes tile it, so a node's region follows
// from its first particle), the critical nodes of the wave kernels' classes inside a queue by decreasing size, ties in Morton order.
// Keys of a stable partition over eight bits: (region, size in steps of eight: 0 = the largest a wavefront serves, the sizes below
// sixteen together); oversized nodes (their own kernel) get the last key and fall off the end of the list. (Rounds 4-5 and the
// first half of round 6: one list, sizes in steps of two.)
constexpr unsigned FIRST_ORDER_KEY_BITS = 8;
static_assert(64 * RK_MAX_R == 256, "first_key(): 31 size buckets of eight");
// The three kernels below also make the first-call launch order of small trees when they are handed key_hist / first_order: the same
// stable partition over the 256 keys of first_key() -- per-block histograms, a scan per key over the blocks, a scatter that ranks
// every node among the equal keys of its block (wavefront: the lanes with the same key from eight ballots; block: counts per wave
// in LDS). Rounds 4-5 sorted (key, index) pairs with the library afterwards: 4-8 launches of ~5 us each on a 100k-particle tree.
constexpr unsigned NKEY = 1u << FIRST_ORDER_KEY_BITS;
__host__ __device__ inline uint32_t first_key(uint32_t size, uint32_t begin, uint32_t nparts)
{
    if (size == 0u || size > 64u * RK_MAX_R) {
        return NKEY - 1u;
    }
    const uint64_t x8 = static_cast<uint64_t>(begin) * 8u / nparts;
    const uint32_t bucket = (64u * RK_MAX_R - size) >> 3;
    return static_cast<uint32_t>(x8 < 7u ? x8 : 7u) * 32u + (bucket < 30u ? bucket : 30u);
}

// Trees of FIRST_ORDER_MAX .. FIRST_TAIL_MAX critical nodes get the LIGHT-TAIL arrangement for their first call instead -- what the
// host makes for repeated calls of that size (rk_launch.hip build_plan, arrange_light_tail): per wave-kernel class the nodes in Morton
// order with the lightest quarter of the class moved to the end, so that the device drains over short waves, and one spatial region
// of the tree per XCD, the same regions in every class kernel, so that neighbouring nodes share an L2 whatever their class. Same
// three partition kernels, other key: (class, region, bulk | light). Regions are cut at equal weight = particle count, and the
// critical nodes tile the particle range, so the region of a node follows from its first particle alone. The size from which a
// node of class c counts as bulk (the value at the first quartile of the sizes of the class, estimated on a sample: k_tail_thr)
// is computed first.
__device__ inline uint32_t tail_key(uint32_t size, uint32_t c, uint32_t begin, uint32_t nparts, const uint32_t (&thr)[RK_MAX_R])
{
    if (size == 0u || size > 64u * RK_MAX_R) {
        return NKEY - 1u;
    }
    const uint64_t x8 = static_cast<uint64_t>(begin) * 8u / nparts;
    return c * 16u + static_cast<uint32_t>(x8 < 7u ? x8 : 7u) * 2u + (size >= thr[c] ? 0u : 1u);
}
static_assert(RK_MAX_R * 16 <= NKEY - 1);
// thr[c] = the size at position floor(n_c / 4) of the ascending sizes of class c among a SAMPLE of the critical nodes -- every
// tail_stride(n_crit)-th one, at most 8192: one block takes their sizes into an LDS histogram (no global atomics: thousands of nodes
// share every common size, and same-address atomics cost the 4M rebuild 0.45 ms when every node made one), one thread per size
// looks its class up, a prefix sum per class finds the quartile. The result goes behind the queue table, first_tab[64 + c].
__host__ __device__ inline uint32_t tail_stride(uint32_t n_crit)
{
    return n_crit / 8192u > 1u ? n_crit / 8192u : 1u;
}
__global__ void __launch_bounds__(1024) k_tail_thr(const uint4 *crit, uint32_t n_crit, uint32_t *first_tab)
{
    constexpr unsigned NS = 64u * RK_MAX_R + 1u; // sizes 0 .. 256
    __shared__ uint32_t cnt[NS];
    __shared__ uint8_t cls[NS];
    __shared__ uint32_t pre[2][NS];
    for (unsigned sz = threadIdx.x; sz < NS; sz += blockDim.x) {
        cnt[sz] = 0u;
        cls[sz] = static_cast<uint8_t>(class2_of_compute(sz ? sz : 1u));
    }
    __syncthreads();
    // The sample: nodes 0, stride, 2 stride, ... -- the first 8 192 of them, eight per thread, all loads of a thread in flight
    // together -- into the LDS histogram. (The first form of this kernel, a strided loop with one load in flight per thread and a
    // serial scan per class, took 20 us of a 4M rebuild: tools/jobs_r06/r06_job13.sh.)
    const uint32_t stride = tail_stride(n_crit);
    uint32_t mine[8];
#pragma unroll
    for (unsigned k = 0; k < 8u; ++k) {
        const uint64_t g = (static_cast<uint64_t>(k) * blockDim.x + threadIdx.x) * stride;
        mine[k] = g < n_crit ? crit[g].w : 0u;
    }
#pragma unroll
    for (unsigned k = 0; k < 8u; ++k) {
        if (mine[k] >= 1u && mine[k] < NS) {
            atomicAdd(&cnt[mine[k]], 1u);
        }
    }
    __syncthreads();
    // Per class: inclusive prefix sums of the class's counts over the sizes (Hillis-Steele in LDS), then the size at which the
    // prefix first exceeds floor(n_c / 4).
    for (unsigned c = 0; c < static_cast<unsigned>(RK_MAX_R); ++c) {
        unsigned cur = 0u;
        for (unsigned sz = threadIdx.x; sz < NS; sz += blockDim.x) {
            pre[0][sz] = cls[sz] == c ? cnt[sz] : 0u;
        }
        __syncthreads();
        for (unsigned d = 1u; d < NS; d <<= 1) {
            for (unsigned sz = threadIdx.x; sz < NS; sz += blockDim.x) {
                pre[cur ^ 1u][sz] = pre[cur][sz] + (sz >= d ? pre[cur][sz - d] : 0u);
            }
            cur ^= 1u;
            __syncthreads();
        }
        const uint32_t n_c = pre[cur][NS - 1u];
        const uint32_t k = n_c ? (n_c / 4u < n_c - 1u ? n_c / 4u : n_c - 1u) : 0u;
        for (unsigned sz = threadIdx.x; sz < NS; sz += blockDim.x) {
            if (sz >= 1u && n_c && pre[cur][sz] > k && pre[cur][sz - 1u] <= k) {
                first_tab[64u + c] = sz;
            }
        }
        if (threadIdx.x == 0u && n_c == 0u) {
            first_tab[64u + c] = 0u;
        }
        __syncthreads();
    }
}

// tail != 0: the keys of the light-tail arrangement instead of those of the heavy-first order. nparts: the number of particles.
__global__ void __launch_bounds__(256) k_bin_count(const uint4 *crit, uint32_t n_crit, uint32_t *block_hist, ctrl_block *ctrl,
                                                   uint32_t *key_hist, uint32_t nparts, uint32_t tail, const uint32_t *first_tab)
{
    __shared__ uint32_t h[NBIN];
    __shared__ uint32_t hk[NKEY];
    __shared__ uint32_t mx;
    if (threadIdx.x < NBIN) {
        h[threadIdx.x] = 0u;
    }
    hk[threadIdx.x] = 0u;
    static_assert(NKEY == 256u, "one key counter per thread of the block");
    if (threadIdx.x == 0u) {
        mx = 0u;
    }
    uint32_t thr[RK_MAX_R] = {};
    if (tail) {
#pragma unroll
        for (int k = 0; k < RK_MAX_R; ++k) {
            thr[k] = first_tab[64 + k]; // (k_tail_thr)
        }
    }
    __syncthreads();
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g < n_crit) {
        const uint4 c4 = crit[g];
        const uint32_t size = c4.w;
        const auto c = static_cast<uint32_t>(class2_of_compute(size));
        atomicAdd(&h[c], 1u);
        atomicMax(&mx, size);
        if (key_hist) {
            atomicAdd(&hk[tail ? tail_key(size, c, c4.x, nparts, thr) : first_key(size, c4.x, nparts)], 1u);
        }
    }
    __syncthreads();
    if (threadIdx.x < NBIN) {
        block_hist[blockIdx.x * NBIN + threadIdx.x] = h[threadIdx.x];
    }
    if (key_hist) {
        key_hist[blockIdx.x * NKEY + threadIdx.x] = hk[threadIdx.x];
    }
    if (threadIdx.x == 0u) {
        atomicMax(&ctrl->max_group, mx);
    }
}

// Exclusive scan of the per-block counts of one bin (blockIdx.x) over the blocks: 256 threads take contiguous runs of
// blocks each, their run totals are scanned in LDS. (One thread per bin walking all blocks took 57 us at 4M particles and
// grew with the particle count: 5 % of a tree rebuild.) Blocks NBIN .. NBIN + NKEY - 1 do the same for the keys of the first-call
// order (stride NKEY, totals to key_total).
__global__ void __launch_bounds__(256) k_bin_scan(uint32_t *block_hist, uint32_t n_blocks, ctrl_block *ctrl, uint32_t *key_hist,
                                                  uint32_t *key_total)
{
    __shared__ uint32_t part[256];
    const bool keys = blockIdx.x >= NBIN;
    const uint32_t c = keys ? blockIdx.x - NBIN : blockIdx.x, t = threadIdx.x, stride = keys ? NKEY : NBIN;
    uint32_t *hist = keys ? key_hist : block_hist;
    const uint32_t per = (n_blocks + 255u) / 256u;
    const uint32_t b0 = t * per < n_blocks ? t * per : n_blocks, b1 = b0 + per < n_blocks ? b0 + per : n_blocks;
    uint32_t sum = 0u;
    for (uint32_t b = b0; b < b1; ++b) {
        sum += hist[b * stride + c];
    }
    part[t] = sum;
    __syncthreads();
    // Hillis-Steele inclusive scan of the 256 run totals.
    for (uint32_t d = 1u; d < 256u; d <<= 1) {
        const uint32_t v = t >= d ? part[t - d] : 0u;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    uint32_t run = part[t] - sum; // exclusive prefix of this thread's run
    for (uint32_t b = b0; b < b1; ++b) {
        const uint32_t v = hist[b * stride + c];
        hist[b * stride + c] = run;
        run += v;
    }
    if (t == 255u) {
        if (keys) {
            key_total[c] = part[255];
        } else {
            ctrl->class2_count[c] = part[255];
        }
    }
}

__global__ void __launch_bounds__(256) k_bin_scatter(const uint4 *crit, uint32_t n_crit, const uint32_t *block_base,
                                                     ctrl_block *ctrl, uint32_t *lists, const uint32_t *key_base,
                                                     const uint32_t *key_total, uint32_t *first_order, uint32_t nparts, uint32_t tail,
                                                     uint32_t *first_tab)
{
    __shared__ uint32_t wave_cnt[4][NBIN];
    __shared__ uint32_t wave_key[4][NKEY];
    __shared__ uint32_t key_off[NKEY];
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    const bool valid = g < n_crit;
    const uint4 c4 = valid ? crit[g] : make_uint4(0u, 0u, 0u, 0u);
    const uint32_t size = c4.w;
    uint32_t thr[RK_MAX_R] = {};
    if (tail) {
#pragma unroll
        for (int k = 0; k < RK_MAX_R; ++k) {
            thr[k] = first_tab[64 + k]; // (k_tail_thr)
        }
    }
    const unsigned c = valid ? static_cast<unsigned>(class2_of_compute(size)) : NBIN;
    uint32_t rank = 0u;
#pragma unroll
    for (unsigned b = 0; b < NBIN; ++b) {
        const unsigned long long m = __ballot(c == b);
        if (c == b) {
            rank = static_cast<uint32_t>(__popcll(m & ((1ull << lane) - 1ull)));
        }
        if (lane == 0u) {
            wave_cnt[w][b] = static_cast<uint32_t>(__popcll(m));
        }
    }
    uint32_t key = 0u, krank = 0u;
    if (first_order) {
        // Exclusive scan of the key totals (every block for itself: 256 values) and the counts per wavefront and key.
        key_off[threadIdx.x] = key_total[threadIdx.x];
#pragma unroll
        for (unsigned k = 0; k < 4u; ++k) {
            wave_key[k][threadIdx.x] = 0u;
        }
        __syncthreads();
        for (uint32_t d = 1u; d < NKEY; d <<= 1) {
            const uint32_t v = threadIdx.x >= d ? key_off[threadIdx.x - d] : 0u;
            __syncthreads();
            key_off[threadIdx.x] += v;
            __syncthreads();
        }
        key = tail ? tail_key(size, c, c4.x, nparts, thr) : first_key(size, c4.x, nparts);
        if (!tail && blockIdx.x == 0u) {
            // Heavy-first order: the table of the eight queues (starts [x], lengths [8 + x]) the one-launch kernels read
            // (rk_list_common.hpp any_list_entry()) and the grid of their launch, 8 x the longest queue.
            if (threadIdx.x < 8u) {
                const unsigned k0 = threadIdx.x * 32u;
                const uint32_t start = key_off[k0] - key_total[k0];
                first_tab[threadIdx.x] = start;
                first_tab[8u + threadIdx.x] = key_off[k0 + 30u] - start;
            }
            if (threadIdx.x == 0u) {
                uint32_t longest = 0u;
                for (unsigned x = 0; x < 8u; ++x) {
                    longest = max(longest, key_off[x * 32u + 30u] - (key_off[x * 32u] - key_total[x * 32u]));
                }
                ctrl->first_grid[0] = 8u * longest;
                ctrl->first_grid[1] = ctrl->first_grid[2] = ctrl->first_grid[3] = 0u;
            }
        }
        if (tail && blockIdx.x == 0u) {
            // The table the class kernels read (rk_common.hpp FIRST_TAB_WORDS) and the grids of their launches.
            if (threadIdx.x < RK_MAX_R * 8u) {
                const unsigned c = threadIdx.x >> 3, x = threadIdx.x & 7u, k0 = c * 16u + x * 2u;
                first_tab[c * 16u + x] = key_off[k0] - key_total[k0];
                first_tab[c * 16u + 8u + x] = key_total[k0] + key_total[k0 + 1u];
            }
            if (threadIdx.x < RK_MAX_R) {
                uint32_t longest = 0u;
                for (unsigned x = 0; x < 8u; ++x) {
                    const unsigned k0 = threadIdx.x * 16u + x * 2u;
                    longest = max(longest, key_total[k0] + key_total[k0 + 1u]);
                }
                ctrl->first_grid[threadIdx.x] = 8u * longest;
            }
        }
        unsigned long long same = __ballot(valid);
#pragma unroll
        for (unsigned b = 0; b < FIRST_ORDER_KEY_BITS; ++b) {
            const bool bit = (key >> b) & 1u;
            const unsigned long long m = __ballot(valid && bit);
            same &= bit ? m : ~m;
        }
        if (valid) {
            krank = static_cast<uint32_t>(__popcll(same & ((1ull << lane) - 1ull)));
            if (krank == 0u) {
                wave_key[w][key] = static_cast<uint32_t>(__popcll(same));
            }
        }
    }
    __syncthreads();
    if (!valid) {
        return;
    }
    if (lists) {
        uint32_t pos = block_base[blockIdx.x * NBIN + c] + rank;
        for (unsigned k = 0; k < w; ++k) {
            pos += wave_cnt[k][c];
        }
        for (unsigned b = 0; b < c; ++b) {
            pos += ctrl->class2_count[b];
        }
        lists[pos] = g;
    }
    if (first_order) {
        // (inclusive scan - own total = exclusive offset of the key)
        uint32_t kpos = key_off[key] - key_total[key] + key_base[blockIdx.x * NKEY + key] + krank;
        for (unsigned k = 0; k < w; ++k) {
            kpos += wave_key[k][key];
        }
        first_order[kpos] = g;
    }
}

struct dev_free {
    void operator()(void *p) const
    {
        if (p) {
            pool_free(p);
        }
    }
};
template <typename T>
using dptr = std::unique_ptr<T, dev_free>;
template <typename T>
dptr<T> dalloc(size_t count)
{
    return dptr<T>(static_cast<T *>(pool_alloc(std::max<size_t>(count, 1) * sizeof(T))));
}

inline unsigned nblk(size_t n, unsigned bs = 256)
{
    return static_cast<unsigned>((n + bs - 1) / bs);
}

// Exclusive prefix sum of n values into out[0..n] (out[n] = total).
void exclusive_scan(const uint32_t *in, uint32_t *out, size_t n, hipStream_t st)
{
    size_t tb = 0;
    RK_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tb, in, out, static_cast<int>(n + 1), st));
    auto tmp = dalloc<unsigned char>(tb);
    // Scan n + 1 elements with a zero appended: requires in[n] readable; callers over-allocate by one and zero it.
    RK_HIP(hipcub::DeviceScan::ExclusiveSum(tmp.get(), tb, in, out, static_cast<int>(n + 1), st));
    // No synchronisation: the scratch goes back to the block cache and is only ever reused on this stream.
}

void exclusive_scan(const tri *in, tri *out, size_t n, hipStream_t st)
{
    size_t tb = 0;
    RK_HIP(hipcub::DeviceScan::ExclusiveScan(nullptr, tb, in, out, tri_sum{}, tri{0u, 0u, 0u}, static_cast<int>(n + 1), st));
    auto tmp = dalloc<unsigned char>(tb);
    RK_HIP(hipcub::DeviceScan::ExclusiveScan(tmp.get(), tb, in, out, tri_sum{}, tri{0u, 0u, 0u}, static_cast<int>(n + 1), st));
}

// ---- Morton sort ------------------------------------------------------------------------------------------
// The onesweep launch sequence below calls functions of rocPRIM's PRIVATE namespace (rocprim::detail::onesweep_histograms,
// onesweep_scan_histograms, onesweep_iteration, onesweep_lookback_state, block_id_wrapper). It was written against -- and its
// look-back protocol (zero = empty state, offsets_in / offsets_out semantics) checked against -- rocPRIM 4.2.0 (ROCm 7.2.0) only:
// with any other version of the library, or with -DRK_SORT_LIBRARY_ONLY, it is compiled out, every sort goes through the public
// hipcub::DeviceRadixSort::SortPairs() and rebuilds sort full keys (sort_partial_ok() is false). RK_SORT_MIN=-1 selects the same
// path at run time; tests/test_gpu_leapfrog.py compares the two bit for bit.
#if !defined(RK_SORT_LIBRARY_ONLY) && defined(ROCPRIM_VERSION) && ROCPRIM_VERSION == 400200
#define RK_ONESWEEP_INTERNALS 1
#else
#define RK_ONESWEEP_INTERNALS 0
#endif
#if RK_ONESWEEP_INTERNALS
// Stable LSD radix sort of (63-bit code, index) pairs: rocPRIM's onesweep DEVICE functions (histograms of every digit place in one
// pass over the keys, then one decoupled-look-back scatter pass per 8-bit digit) under this file's own launch sequence.
// hipcub::DeviceRadixSort::SortPairs() runs the same device code but resets its look-back states and its ordered-block counter with
// two memsets in front of EVERY pass plus one for the histograms -- 17 fill kernels of ~5.5 us, a fifth of the sort at 4M
// particles (profiles/r05/rebuild_timeline_*.txt) -- and below 2^20 items it switches to a merge sort of 1 + 2 log2(n / 4096)
// launches (21 at 1M). Here the states of all eight passes are laid out side by side and cleared by ONE memset, the passes
// ping-pong between the two caller-owned buffer pairs (no third copy of the pairs in a scratch allocation), and every size
// takes the onesweep path. Same result: both are stable sorts of the same keys.
namespace sortk
{
constexpr unsigned END_BIT = 63;
using bid_t = rocprim::detail::block_id_wrapper<unsigned int, true>; // blocks take their index from a counter, in arrival order
using lookback_t = rocprim::detail::onesweep_lookback_state;
static_assert(sizeof(lookback_t) == sizeof(uint32_t));

template <unsigned BS, unsigned IPT, unsigned RB>
__global__ void __launch_bounds__(BS) k_sort_hist(const uint64_t *keys, uint32_t *counts, uint32_t n, uint32_t full_blocks, unsigned begin_bit)
{
    rocprim::detail::onesweep_histograms<BS, IPT, RB, false>(keys, counts, n, full_blocks, rocprim::identity_decomposer{}, begin_bit, END_BIT);
}
template <unsigned BS, unsigned RB>
__global__ void __launch_bounds__(BS) k_sort_scan(uint32_t *counts)
{
    rocprim::detail::onesweep_scan_histograms<BS, RB>(counts);
}
template <unsigned BS, unsigned IPT, unsigned RB>
__global__ void __launch_bounds__(BS) k_sort_pass(const uint64_t *kin, uint64_t *kout, const uint32_t *vin, uint32_t *vout, uint32_t n,
                                                  uint32_t *offs_in, uint32_t *offs_out, lookback_t *lb, unsigned bit, unsigned bits,
                                                  unsigned full_blocks, bid_t bid)
{
    rocprim::detail::onesweep_iteration<BS, IPT, RB, false, rocprim::block_radix_rank_algorithm::match>(
        kin, kout, vin, vout, n, offs_in, offs_out, lb, rocprim::identity_decomposer{}, bit, bits, full_blocks, bid);
}

// State for n items in blocks of BS * IPT and digits of RB bits: digit counts of every place, a scratch row the last block of a pass
// writes, one block counter per pass, the look-back states of every pass (2^RB words per block and pass). Bits [begin_bit, END_BIT)
// of the keys count; returns true when the number of passes is odd, i.e. the result is in (kb, vb).
template <unsigned BS, unsigned IPT, unsigned RB>
bool onesweep(uint64_t *ka, uint32_t *va, uint64_t *kb, uint32_t *vb, uint32_t n, unsigned begin_bit, hipStream_t st)
{
    constexpr unsigned IPB = BS * IPT, RADIX = 1u << RB;
    const unsigned places = (END_BIT - begin_bit + RB - 1u) / RB;
    const unsigned blocks = (n + IPB - 1u) / IPB, full_blocks = n % IPB == 0u ? blocks : blocks - 1u;
    const size_t words = static_cast<size_t>(places) * RADIX + RADIX + 16 + static_cast<size_t>(places) * RADIX * blocks;
    auto state = dalloc<uint32_t>(words);
    RK_HIP(hipMemsetAsync(state.get(), 0, words * sizeof(uint32_t), st));
    uint32_t *counts = state.get(), *scratch_row = counts + places * RADIX, *bids = scratch_row + RADIX;
    auto *lb = reinterpret_cast<lookback_t *>(bids + 16);
    hipLaunchKernelGGL((k_sort_hist<BS, IPT, RB>), dim3(blocks), dim3(BS), 0, st, ka, counts, n, full_blocks, begin_bit);
    hipLaunchKernelGGL((k_sort_scan<BS, RB>), dim3(places), dim3(BS), 0, st, counts);
    for (unsigned p = 0; p < places; ++p) {
        const unsigned bit = begin_bit + p * RB, bits = std::min(RB, END_BIT - bit);
        const bool fwd = p % 2u == 0u;
        hipLaunchKernelGGL((k_sort_pass<BS, IPT, RB>), dim3(blocks), dim3(BS), 0, st, fwd ? ka : kb, fwd ? kb : ka, fwd ? va : vb,
                           fwd ? vb : va, n, counts + p * RADIX, scratch_row, lb + static_cast<size_t>(p) * RADIX * blocks, bit, bits,
                           full_blocks, bid_t::create(bids + p));
    }
    RK_HIP(hipGetLastError());
    // (The state goes back to the block cache and is only ever reused on this stream.)
    return places % 2u == 1u;
}
} // namespace sortk
#else
namespace sortk
{
constexpr unsigned END_BIT = 63;
}
#endif

// Sorts the n pairs of (ka, va) by bits [begin_bit, 63) of the keys, stably; (kb, vb) is scratch of the same size. Returns true when the
// result is in (kb, vb) instead of (ka, va). begin_bit > 0 (the caller completes the order itself, k_local_sort) always takes the
// onesweep passes.
long sort_onesweep_min()
{
#if !RK_ONESWEEP_INTERNALS
    return -1; // (not the rocPRIM this file's onesweep sequence was written against: the public library call for every size)
#endif
    static const long knob_min = [] {
        const char *e = std::getenv("RK_SORT_MIN"); // items from which the onesweep sequence replaces the library call (-1: never)
        return e ? std::atol(e) : (1l << 20);
    }();
    return knob_min;
}
bool sort_codes(uint64_t *ka, uint32_t *va, uint64_t *kb, uint32_t *vb, uint32_t n, unsigned begin_bit, hipStream_t st)
{
    using namespace sortk;
    if (n < 2u) {
        return false;
    }
    const long knob_min = sort_onesweep_min();
    // (The merge sort itself with tiles of 2048 / 4096 items instead of its 1024 -- fewer merge passes, no copy launches when their number
    // is even -- moves the rebuild by -14...+9 us between 30k and 1M items, inside the box-to-box noise: tools/archive/jobs_r05/r05_job58.sh.)
    // Below 2^20 items the library's merge sort wins (launches of a dozen blocks of 8192 items leave the device empty: 100k +0.11 ms,
    // 1M +0.055 ms with the onesweep passes; 2M -0.045, 4M -0.057: tools/archive/jobs_r05/r05_job37.sh).
    if (begin_bit == 0u && (knob_min < 0 || static_cast<long>(n) < knob_min || n > (1u << 28))) {
        size_t tb = 0;
        RK_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, tb, ka, kb, va, vb, static_cast<int>(n), 0, static_cast<int>(END_BIT), st));
        auto tmp = dalloc<unsigned char>(tb);
        RK_HIP(hipcub::DeviceRadixSort::SortPairs(tmp.get(), tb, ka, kb, va, vb, static_cast<int>(n), 0, static_cast<int>(END_BIT), st));
        return true;
    }
#if !RK_ONESWEEP_INTERNALS
    throw error(RK_ERUNTIME, "internal error: a partial-key sort was asked for in a build without the onesweep launch sequence");
#else
    // rocPRIM's gfx942 / gfx950 block shape for 8 + 4 byte pairs. Smaller blocks for mid-size sorts (256 x 8, 256 x 16, 512 x 8,
    // 256 x 4 from 100k to 2M items) all lose to the merge sort and to this shape: tools/archive/jobs_r05/r05_job40.sh.
    // Digits of 8 bits (the library's) or 9: whichever takes fewer passes over the bits that count (63 bits: 8 against 7 passes;
    // 36 bits, a rebuild of a tree eleven levels deep: 5 against 4). A 9-bit pass costs what an 8-bit one does (4M rebuild: 0.706 ->
    // 0.661 ms partial, 0.787 -> 0.757 full), a 10-bit one 19 us more at 4M -- the runs a block writes per bin get too short
    // (tools/archive/jobs_r05/r05_job68.sh, r05_job69.sh).
    const unsigned bits = END_BIT - begin_bit;
    const bool nine = (bits + 8u) / 9u < (bits + 7u) / 8u;
    return nine ? onesweep<512, 16, 9>(ka, va, kb, vb, n, begin_bit, st) : onesweep<512, 16, 8>(ka, va, kb, vb, n, begin_bit, st);
#endif
}
// Whether a rebuild of n particles may sort a partial key (the onesweep passes look at a sub-range of the bits; the library call
// does not): from half the size at which full sorts go to the onesweep passes -- four 9-bit passes + k_local_sort against the merge
// sort: 350k equal, 600k -6 %, 1M -13 % of the rebuild (tools/archive/jobs_r05/r05_job70.sh).
bool sort_partial_ok(uint32_t n)
{
    const long knob_min = sort_onesweep_min();
    return n >= 2u && knob_min >= 0 && static_cast<long>(n) >= knob_min / 2 && n <= (1u << 28);
}

} // namespace bld

static bool first_order_enabled()
{
    static const bool on = [] {
        const char *e = std::getenv("RK_FIRST_ORDER"); // 0: no launch order for first calls is made with the tree (they read the class lists)
        return !(e && std::atoi(e) == 0);
    }();
    return on;
}

static bool want_first_tail(uint32_t n_crit, uint32_t nparts)
{
    return first_order_enabled() && n_crit > FIRST_ORDER_MAX && n_crit <= FIRST_TAIL_MAX && nparts > 0u && RK_MAX_R == 4;
}

// The lane-mapping classes of the critical nodes (second half of RK_BUF_CLASS; `lists` null: a replica has them already) and, with
// them, the launch order of the FIRST call on the tree: heavy-first up to FIRST_ORDER_MAX critical nodes (first_key()), the light-tail
// arrangement up to FIRST_TAIL_MAX (tail_key(); its grids arrive in ctrl->first_grid with the caller's next look-up of the control
// block: take_first_grid()). Returns the temporaries the kernels work on; the caller keeps them until it has synchronised.
struct bin_tmp {
    bld::dptr<uint32_t> hist, key_hist;
};
static bin_tmp bin_classes(rk_state &s, const uint4 *crit, uint32_t n_crit, uint32_t nparts, uint32_t *lists, bld::ctrl_block *ctrl,
                           hipStream_t st)
{
    using namespace bld;
    bin_tmp t;
    const unsigned nb = nblk(n_crit);
    t.hist = dalloc<uint32_t>(static_cast<size_t>(nb) * NBIN);
    const bool want_first = first_order_enabled() && n_crit <= FIRST_ORDER_MAX && nparts > 0u;
    const bool want_tail = want_first_tail(n_crit, nparts);
    uint32_t *key_total = nullptr, *first_order = nullptr;
    if (want_first || want_tail) {
        const int64_t need = std::max<int64_t>(FIRST_ORDER_MAX, n_crit);
        if (!s.first_order || s.first_order_cap < need) {
            // (the caller has synchronised the device since the last traversal of the old tree: release_tree())
            pool_free(s.first_order);
            s.first_order = nullptr;
            s.first_order = pool_alloc(static_cast<size_t>(need) * sizeof(uint32_t));
            s.first_order_cap = need;
        }
        first_order = static_cast<uint32_t *>(s.first_order);
        t.key_hist = dalloc<uint32_t>(static_cast<size_t>(nb) * NKEY + NKEY);
        key_total = t.key_hist.get() + static_cast<size_t>(nb) * NKEY;
    }
    if ((want_first || want_tail) && !s.first_tab) {
        s.first_tab = pool_alloc(FIRST_TAB_WORDS * sizeof(uint32_t));
    }
    if (want_tail) {
        hipLaunchKernelGGL(k_tail_thr, dim3(1), dim3(1024), 0, st, crit, n_crit, static_cast<uint32_t *>(s.first_tab));
    }
    const uint32_t tail = want_tail ? 1u : 0u;
    hipLaunchKernelGGL(k_bin_count, dim3(nb), dim3(256), 0, st, crit, n_crit, t.hist.get(), ctrl, t.key_hist.get(), nparts, tail,
                       static_cast<const uint32_t *>(s.first_tab));
    hipLaunchKernelGGL(k_bin_scan, dim3(NBIN + (first_order ? NKEY : 0u)), dim3(256), 0, st, t.hist.get(), nb, ctrl, t.key_hist.get(),
                       key_total);
    hipLaunchKernelGGL(k_bin_scatter, dim3(nb), dim3(256), 0, st, crit, n_crit, t.hist.get(), ctrl, lists, t.key_hist.get(), key_total,
                       first_order, nparts, tail, static_cast<uint32_t *>(s.first_tab));
    s.first_order_valid = want_first;
    s.first_tail_valid = want_tail;
    return t;
}
static void take_first_grid(rk_state &s, const bld::ctrl_block &hc)
{
    for (int c = 0; c < 4; ++c) {
        s.first_grid[c] = (s.first_tail_valid || s.first_order_valid) ? hc.first_grid[c] : 0u;
    }
}

// The same for a replica (rk_state_clone / import / broadcast: the critical nodes and their class lists are on its device already).
void replica_first_order(rk_state &s)
{
    using namespace bld;
    s.first_order_valid = s.first_tail_valid = false;
    if (first_order_enabled() && s.n_crit > 0 && s.n_crit <= static_cast<int64_t>(FIRST_TAIL_MAX) && s.buf[RK_BUF_CRIT]) {
        auto ctrl = dalloc<ctrl_block>(1);
        RK_HIP(hipMemsetAsync(ctrl.get(), 0, sizeof(ctrl_block), nullptr));
        const bin_tmp tmp = bin_classes(s, static_cast<const uint4 *>(s.buf[RK_BUF_CRIT]), static_cast<uint32_t>(s.n_crit),
                                        static_cast<uint32_t>(s.nparts), nullptr, ctrl.get(), nullptr);
        ctrl_block hc{};
        RK_HIP(hipMemcpy(&hc, ctrl.get(), sizeof(ctrl_block), hipMemcpyDeviceToHost)); // (also: the temporaries may go back to the pool)
        take_first_grid(s, hc);
    }
}

// The host's view of a build's control block: the block is copied into pinned host memory by an asynchronous copy behind the kernel
// that completes it, an event marks the copy, and the host waits for the event only when it needs the numbers -- after it has
// handed the device the work that does not depend on them (a blocking hipMemcpy cost ~15 us of idle device per look-up, three per
// build). One set of buffers and events per calling thread and device, made on first use, given back when the thread ends
// (a thread's destructors run before the runtime's own at process exit; errors from a runtime that is already gone are ignored).
struct lookup_slots {
    int device = -1;
    bld::ctrl_block *host = nullptr;
    hipEvent_t ev[3] = {};
};
struct lookup_slots_owner {
    std::vector<lookup_slots> all;
    ~lookup_slots_owner()
    {
        for (auto &l : all) {
            for (auto &e : l.ev) {
                if (e) {
                    (void)hipEventDestroy(e);
                }
            }
            if (l.host) {
                (void)hipHostFree(l.host);
            }
        }
    }
};
static lookup_slots &thread_lookup_slots()
{
    static thread_local lookup_slots_owner owner;
    auto &all = owner.all;
    int dev = 0;
    RK_HIP(hipGetDevice(&dev));
    for (auto &l : all) {
        if (l.device == dev) {
            return l;
        }
    }
    lookup_slots l;
    l.device = dev;
    void *p = nullptr;
    RK_HIP(hipHostMalloc(&p, 3 * sizeof(bld::ctrl_block), hipHostMallocDefault));
    l.host = static_cast<bld::ctrl_block *>(p);
    for (auto &e : l.ev) {
        RK_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    all.push_back(l);
    return all.back();
}

// Builds the tree and fills `s` (buffers, sizes). Host inputs in the caller's original order.
template <typename F, int ND>
void build_device(rk_state &s, const void *const parts[4], bool parts_on_device, int64_t nparts, double box_size_in,
                  uint64_t max_leaf_n, std::string &bad_coord_msg)
{
    constexpr unsigned CBITS = bld::geo<ND>::CB;
    using namespace bld;
    using v4 = typename vt<F>::v4;
    using v2 = typename vt<F>::v2;
    hipStream_t st = nullptr;
    const auto n = static_cast<uint32_t>(nparts);
    const size_t fb = static_cast<size_t>(n) * sizeof(F);

    // Inputs: copied from the host, or used in place when they already live on this device.
    // parts = ND coordinate arrays, then the masses; in[] is always {x, y, z or null, m}.
    dptr<F> own[4];
    const F *in[4] = {};
    for (int k = 0; k < 4; ++k) {
        const int src = k < ND ? k : (k == 3 ? ND : -1);
        if (src < 0) {
            continue;
        }
        if (parts_on_device) {
            in[k] = static_cast<const F *>(parts[src]);
        } else {
            own[k] = dalloc<F>(n);
            RK_HIP(hipMemcpyAsync(own[k].get(), parts[src], fb, hipMemcpyHostToDevice, st));
            in[k] = own[k].get();
        }
    }
    struct view {
        const F *p;
        const F *get() const
        {
            return p;
        }
    };
    const view dx{in[0]}, dy{in[1]}, dz{in[2]}, dm{in[3]};

    auto ctrl = dalloc<ctrl_block>(1);
    ctrl_block hc{};
    RK_HIP(hipMemsetAsync(ctrl.get(), 0, sizeof(ctrl_block), st));
    // (Round 5, measured and not kept: the last kernel in front of a look-up storing the block + a sequence number to pinned host
    // memory that the host polls, instead of a copy -- rebuild +0.025 ms at 100k, +0.045 at 4M, tools/archive/jobs_r05/r05_job39.sh.)
    lookup_slots &lk = thread_lookup_slots();
    const auto lookup_begin = [&](int slot) {
        RK_HIP(hipMemcpyAsync(lk.host + slot, ctrl.get(), sizeof(ctrl_block), hipMemcpyDeviceToHost, st));
        RK_HIP(hipEventRecord(lk.ev[slot], st));
    };
    const auto lookup_end = [&](int slot) {
        RK_HIP(hipEventSynchronize(lk.ev[slot]));
        std::memcpy(&hc, lk.host + slot, sizeof(hc));
    };

    // ---- box size, encode, sort, permute, leaf levels, node counts: no host round trip ----
    s.box_deduced = box_size_in == 0.;
    const auto mln = static_cast<uint32_t>(std::min<uint64_t>(max_leaf_n, 0xffffffffu));
    void *p4 = pool_alloc(std::max<size_t>(n, 1) * sizeof(v4));
    s.buf[RK_BUF_PART4] = p4;
    s.buf_bytes[RK_BUF_PART4] = static_cast<int64_t>(n * sizeof(v4));
    // A REBUILD sorts only the code bits of the levels the tree can be expected to use -- the deepest leaf level of the state's
    // previous build + 1 -- and completes the order inside the leaves itself (k_local_sort): five onesweep passes instead of eight
    // for a tree twelve levels deep. The first look-up brings the deepest level actually found; if it is deeper than assumed the
    // front of the build runs again with the full sort.
    static const int partial_bias = [] {
        const char *e = std::getenv("RK_SORT_PARTIAL"); // levels added to the previous build's depth (default 1); -100: full sort always
        return e ? std::atoi(e) : 1;
    }();
    unsigned sort_levels = CBITS;
    if (s.bld_max_level >= 0 && partial_bias > -100 && mln <= 64u && sort_partial_ok(n)) {
        sort_levels = static_cast<unsigned>(std::clamp(s.bld_max_level + partial_bias, 1, static_cast<int>(CBITS)));
    }
    dptr<uint64_t> keys_a;
    dptr<uint32_t> vals_a;
    dptr<uint8_t> leaf, ldiv;
    dptr<uint32_t> cnt, off;
    // Node sums: from a summation pyramid over the particles, or -- exact mode -- in an array, in the reference's serial association.
    const bool pyramid = !exact_node_sums();
    pyr_desc pd{};
    pd.cnt[0] = n;
    size_t pyr_entries = 0;
    while (pd.cnt[pd.levels] > 1u && pd.levels < 31u) {
        const unsigned l = ++pd.levels;
        pd.cnt[l] = (pd.cnt[l - 1u] + 1u) / 2u;
        pd.off[l] = static_cast<uint32_t>(pyr_entries);
        pyr_entries += pd.cnt[l];
    }
    dptr<v4> sums;
    for (;;) {
        const bool partial = sort_levels < CBITS;
        const unsigned begin_bit = partial ? bld::geo<ND>::DB * (CBITS - sort_levels) : 0u;
        if (s.box_deduced) {
            // (At most one block of 1024 threads per CU: the blocks end with one atomic each on ONE word, ~10 ns apiece -- 1024 blocks
            // of 256 threads spent half of the kernel's 20 us at 4M particles in them.)
            hipLaunchKernelGGL((k_maxabs<F>), dim3(std::min(nblk(n, 1024), 256u)), dim3(1024), 0, st, dx.get(), dy.get(), dz.get(), n,
                               ctrl.get());
        }
        // (ka, va): the codes and indices, sorted in place; (kb, vb): the other half of the sort's ping-pong.
        keys_a = dalloc<uint64_t>(n);
        vals_a = dalloc<uint32_t>(n);
        auto keys_b = dalloc<uint64_t>(n);
        auto vals_b = dalloc<uint32_t>(n);
        hipLaunchKernelGGL((k_encode<F, ND>), dim3(nblk(n)), dim3(256), 0, st, dx.get(), dy.get(), dz.get(), n, ctrl.get(),
                           static_cast<F>(box_size_in), keys_a.get(), vals_a.get());
        if (sort_codes(keys_a.get(), vals_a.get(), keys_b.get(), vals_b.get(), n, begin_bit, st)) {
            keys_a.swap(keys_b);
            vals_a.swap(vals_b);
        }
        if (mln > 64u) {
            hipLaunchKernelGGL((k_permute<F>), dim3(nblk(n)), dim3(256), 0, st, dx.get(), dy.get(), dz.get(), dm.get(), vals_a.get(), n,
                               static_cast<v4 *>(p4));
        }
        leaf = dalloc<uint8_t>(n), ldiv = dalloc<uint8_t>(n);
        cnt = dalloc<uint32_t>(static_cast<size_t>(n) + 1), off = dalloc<uint32_t>(static_cast<size_t>(n) + 1);
        dptr<uint8_t> block_max;
        if (mln <= 64u) {
            block_max = dalloc<uint8_t>(nblk(n));
            hipLaunchKernelGGL((k_leaf_levels_windows<F, ND>), dim3(nblk(n)), dim3(256), 0, st, keys_a.get(), n, mln, leaf.get(), ldiv.get(),
                               cnt.get(), dx.get(), dy.get(), dz.get(), dm.get(), partial ? nullptr : vals_a.get(), static_cast<v4 *>(p4),
                               block_max.get());
        } else {
            hipLaunchKernelGGL(k_leaf_levels_search<ND>, dim3(nblk(n)), dim3(256), 0, st, keys_a.get(), n, mln, leaf.get());
            hipLaunchKernelGGL(k_node_counts<ND>, dim3(nblk(n)), dim3(256), 0, st, keys_a.get(), n, leaf.get(), ldiv.get(), cnt.get());
        }
        exclusive_scan(cnt.get(), off.get(), n, st);
        hipLaunchKernelGGL(k_pack_nodes, dim3(1), dim3(256), 0, st, ctrl.get(), off.get() + n, block_max.get(), nblk(n), CBITS);

        // ---- first look-up: input errors (in the reference's order), box, node count, deepest level. While the host waits the
        // device completes a partial sort and builds the summation pyramid over the particles (node sums): neither needs anything the
        // host is waiting for ----
        lookup_begin(0);
        if (partial) {
            hipLaunchKernelGGL((k_local_sort<F, ND>), dim3(nblk(n)), dim3(256), 0, st, keys_a.get(), vals_a.get(), leaf.get(), n, mln,
                               dx.get(), dy.get(), dz.get(), dm.get(), keys_b.get(), vals_b.get(), static_cast<v4 *>(p4));
            keys_a.swap(keys_b);
            vals_a.swap(vals_b);
        }
        keys_b.reset();
        vals_b.reset();
        if (pyramid) {
            sums = dalloc<v4>(std::max<size_t>(pyr_entries, 1));
            for (unsigned base = 0; base < pd.levels; base += 9u) {
                hipLaunchKernelGGL((k_pyr_pass<F>), dim3((pd.cnt[base] + 511u) / 512u), dim3(256), 0, st, static_cast<const v4 *>(p4),
                                   sums.get(), pd, base);
            }
        }
        lookup_end(0);
        static const bool sort_trace = [] {
            const char *e = std::getenv("RK_SORT_TRACE"); // 1: one line per build on stderr (tests/test_gpu_leapfrog.py)
            return e && std::atoi(e) != 0;
        }();
        if (sort_trace) {
            std::fprintf(stderr, "rk_build: n %u sorted levels %u of %u (bits from %u), deepest leaf level %u%s\n", n, sort_levels, CBITS,
                         begin_bit, hc.max_level, partial && hc.max_level > sort_levels ? " -> again with all bits" : "");
        }
        if (partial && hc.max_level > sort_levels && !(hc.err & (ERR_COORD | ERR_BOX)) && hc.bad_inv == 0u) {
            // A leaf deeper than the sorted bits reach: everything above was computed on an order that is not the full one. Again, with
            // all bits (the control block starts from zero like the first time).
            sort_levels = CBITS;
            RK_HIP(hipMemsetAsync(ctrl.get(), 0, sizeof(ctrl_block), st));
            continue;
        }
        break;
    }
    s.bld_codes = keys_a.release();
    s.bld_perm = vals_a.release();
    const auto *codes = static_cast<const uint64_t *>(s.bld_codes);
    s.bld_max_level = static_cast<int>(hc.max_level);
    if (hc.err & ERR_COORD) {
        throw error(RK_EINVAL, "While trying to automatically determine the domain size, a non-finite coordinate "
                               "was encountered");
    }
    const F box = static_cast<F>(hc.box);
    if (hc.err & ERR_BOX) {
        throw error(RK_EINVAL, "The automatic deduction of the domain size produced the non-finite value "
                                   + std::to_string(box));
    }
    s.box_size = hc.box;
    if (hc.bad_inv != 0u) {
        // Rebuild the reference's message (tree.hpp:398-413) for the first offending coordinate.
        const unsigned first_bad = ~hc.bad_inv;
        const F inv_box = F(1) / box;
        for (int k = 0; k < ND; ++k) {
            F xv;
            RK_HIP(hipMemcpy(&xv, in[k] + first_bad, sizeof(F), hipMemcpyDeviceToHost));
            F tmp = std::fma(xv, inv_box, F(1) / F(2));
            tmp *= F(1u << CBITS);
            if (!std::isfinite(tmp)) {
                bad_coord_msg = "While trying to discretise the input coordinate " + std::to_string(xv)
                                + " in a box of size " + std::to_string(F(1) / inv_box) + ", the non-finite value "
                                + std::to_string(tmp) + " was generated";
                break;
            }
            if (tmp < F(0) || tmp >= F(1u << CBITS)) {
                bad_coord_msg = "The discretisation of the input coordinate " + std::to_string(xv)
                                + " in a box of size " + std::to_string(F(1) / inv_box)
                                + " produced the floating-point value " + std::to_string(tmp)
                                + ", which is outside the allowed bounds";
                break;
            }
        }
        throw error(RK_EINVAL, bad_coord_msg);
    }
    for (auto &o : own) {
        o.reset();
    }
    const size_t nn = static_cast<size_t>(hc.n_nonroot) + 1;
    if (nn >= max_list_nodes) {
        throw error(RK_EOVERFLOW, "The number of tree nodes (" + std::to_string(nn)
                                      + ") exceeds the 2^29 limit of the traversal kernel's node references");
    }
    s.tree_size = static_cast<int64_t>(nn);
    auto alloc_buf = [&](int which, size_t bytes) {
        s.buf[which] = pool_alloc(std::max<size_t>(bytes, 16));
        s.buf_bytes[which] = static_cast<int64_t>(bytes);
        return s.buf[which];
    };
    auto *topo = static_cast<uint4 *>(alloc_buf(RK_BUF_NODE_TOPO, nn * sizeof(uint4)));
    auto *node_com = static_cast<v4 *>(alloc_buf(RK_BUF_NODE_COM, nn * sizeof(v4)));
    auto *node_mac = static_cast<v2 *>(alloc_buf(RK_BUF_NODE_MAC, nn * sizeof(v2)));
    auto *recs = static_cast<node_rec<F> *>(alloc_buf(RK_BUF_NODE_REC, nn * sizeof(node_rec<F>)));
    s.bld_node_code = pool_alloc(nn * sizeof(uint64_t));
    auto *ncode = static_cast<uint64_t *>(s.bld_node_code);
    auto parent = dalloc<uint32_t>(nn);
    auto mask = dalloc<uint32_t>(nn + 1);
    {
        auto start_of = dalloc<uint32_t>(nn);
        hipLaunchKernelGGL(k_node_starts, dim3(nblk(n)), dim3(256), 0, st, n, leaf.get(), ldiv.get(), off.get(), start_of.get());
        hipLaunchKernelGGL(k_emit_per_node<ND>, dim3(nblk(nn)), dim3(256), 0, st, codes, n, ldiv.get(), off.get(), start_of.get(),
                           static_cast<uint32_t>(nn), topo, ncode, parent.get());
    }
    if (pyramid) {
        hipLaunchKernelGGL((k_parents<F, ND, false>), dim3(nblk(nn)), dim3(256), 0, st, topo, ncode, static_cast<uint32_t>(nn), parent.get(),
                           mask.get(), static_cast<const v4 *>(p4), sums.get());
    } else {
        sums = dalloc<v4>(nn);
        hipLaunchKernelGGL((k_parents<F, ND, true>), dim3(nblk(nn)), dim3(256), 0, st, topo, ncode, static_cast<uint32_t>(nn), parent.get(),
                           mask.get(), static_cast<const v4 *>(p4), sums.get());
    }
    leaf.reset(), ldiv.reset(), cnt.reset(), off.reset();

    if (exact_node_sums()) {
        // The reference's association (bit-identical node properties): one serial chain per distinct first particle, all
        // at once; the root's N links set the time (~25 ms at 4M particles). (The leaves have their sums from k_parents.)
        const auto max_big = static_cast<unsigned>(static_cast<size_t>(n) / EXACT_WAVE_MIN * (CBITS + 1u) + 2u);
        auto big = dalloc<uint32_t>(static_cast<size_t>(max_big) + 1u);
        // Slot 0 is the root's; the counter starts behind it.
        RK_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(big.get()), static_cast<int>(EXACT_NO_ROOT), 1, st));
        RK_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(big.get() + max_big), 1, 1, st));
        hipLaunchKernelGGL((k_exact_chains<F>), dim3(nblk(nn)), dim3(256), 0, st, topo, static_cast<uint32_t>(nn),
                           static_cast<const v4 *>(p4), sums.get(), big.get(), big.get() + max_big);
        // (Chains of one tree level are disjoint in particles: at most n / EXACT_WAVE_MIN long ones per level exist.)
        hipLaunchKernelGGL((k_exact_chains_wave<F>), dim3(std::min<unsigned>(max_big, static_cast<unsigned>(nn))), dim3(256), 0, st,
                           topo, big.get(), big.get() + max_big, static_cast<const v4 *>(p4), sums.get());
    }
    // ---- critical nodes, child counts: one scan of three counters; the centres of mass while the host looks at the counts ----
    const auto ncrit_c = static_cast<uint32_t>(std::min<uint64_t>(s.ncrit, 0xffffffffu));
    auto flags = dalloc<tri>(nn + 1), offs = dalloc<tri>(nn + 1);
    hipLaunchKernelGGL((k_flags<ND, true>), dim3(nblk(nn)), dim3(256), 0, st, topo, ncode, parent.get(), static_cast<uint32_t>(nn), ncrit_c,
                       flags.get(), mask.get());
    exclusive_scan(flags.get(), offs.get(), nn, st);
    hipLaunchKernelGGL(k_pack_counts, dim3(1), dim3(1), 0, st, ctrl.get(), offs.get() + nn);
    lookup_begin(1);
    if (pyramid) {
        hipLaunchKernelGGL((k_finalize<F, ND, true>), dim3(nblk(nn)), dim3(256), 0, st, topo, ncode, static_cast<uint32_t>(nn), sums.get(),
                           static_cast<const v4 *>(p4), pd, box, s.mac, node_com, node_mac, ctrl.get());
    } else {
        hipLaunchKernelGGL((k_finalize<F, ND, false>), dim3(nblk(nn)), dim3(256), 0, st, topo, ncode, static_cast<uint32_t>(nn), sums.get(),
                           static_cast<const v4 *>(p4), pd, box, s.mac, node_com, node_mac, ctrl.get());
    }
    sums.reset();

    // ---- second look-up: counts (the node-property errors come with the third: k_finalize may still be running) ----
    lookup_end(1);
    const uint32_t n_crit = hc.n_crit, n_int = hc.n_int;
    if (static_cast<size_t>(hc.n_children) + 1 != nn) {
        throw error(RK_ERUNTIME, "internal error: inconsistent node count in the device tree build");
    }
    s.n_internal = n_int;
    auto *crit = static_cast<uint4 *>(alloc_buf(RK_BUF_CRIT, static_cast<size_t>(n_crit) * sizeof(uint4)));
    auto *boxes = static_cast<v4 *>(alloc_buf(RK_BUF_CRIT_BOX, static_cast<size_t>(n_crit) * 2 * sizeof(v4)));
    auto *child_tab = static_cast<uint32_t *>(alloc_buf(RK_BUF_CHILD, static_cast<size_t>(n_int) * 8 * sizeof(uint32_t)));
    {
        const auto tab_words = static_cast<uint32_t>(static_cast<size_t>(n_int) * 8u);
        const auto zper = static_cast<uint32_t>((static_cast<size_t>(tab_words) + nn - 1u) / nn);
        hipLaunchKernelGGL((k_crit<F>), dim3(nblk(nn)), dim3(256), 0, st, topo, flags.get(), offs.get(), static_cast<uint32_t>(nn), crit,
                           child_tab, tab_words, zper);
    }
    hipLaunchKernelGGL((k_crit_boxes<F>), dim3((n_crit + 3u) / 4u), dim3(256), 0, st, crit, n_crit, static_cast<const v4 *>(p4), boxes);
    hipLaunchKernelGGL((k_records<F, ND>), dim3(nblk(nn)), dim3(256), 0, st, topo, ncode, parent.get(), mask.get(), offs.get(),
                       static_cast<uint32_t>(nn), node_com, node_mac, recs, child_tab);

    // ---- group lists of the list kernel (second half of RK_BUF_CLASS; the first half, the cross-check kernel's
    // binning, is filled with the host mirrors on demand) ----
    auto *lists = static_cast<uint32_t *>(alloc_buf(RK_BUF_CLASS, static_cast<size_t>(n_crit) * 2 * sizeof(uint32_t)));
    // (with the launch order of the first call: see bin_classes())
    const bin_tmp bin_scratch = bin_classes(s, crit, n_crit, n, lists + n_crit, ctrl.get(), st);

    // ---- third look-up: node-property errors, class sizes (also the final synchronisation) ----
    lookup_begin(2);
    lookup_end(2);
    RK_HIP(hipGetLastError());
    if (hc.err & ERR_COM) {
        throw error(RK_EINVAL, "The computation of the centre of mass of a node produced a non-finite value");
    }
    if (hc.err & ERR_DIM) {
        throw error(RK_EINVAL, "The computation of the dimension of a node produced a non-finite value");
    }
    s.n_crit = n_crit;
    s.max_group = hc.max_group;
    take_first_grid(s, hc);
    s.mirrors_valid = false;
    s.crit_begin.clear();
    s.crit_end.clear();
    s.class_off[0] = 0;
    s.class2_off[0] = n_crit;
    for (int c = 0; c < n_classes; ++c) {
        s.class_list[c].clear();
        s.class2_list[c].clear();
        s.class_off[c + 1] = 0;
        s.class2_count[c] = hc.class2_count[c];
        s.class2_off[c + 1] = s.class2_off[c] + s.class2_count[c];
    }
}

template void build_device<float, 3>(rk_state &, const void *const[4], bool, int64_t, double, uint64_t, std::string &);
template void build_device<double, 3>(rk_state &, const void *const[4], bool, int64_t, double, uint64_t, std::string &);
template void build_device<float, 2>(rk_state &, const void *const[4], bool, int64_t, double, uint64_t, std::string &);
template void build_device<double, 2>(rk_state &, const void *const[4], bool, int64_t, double, uint64_t, std::string &);

// ---- a tree built on the HOST, converted on the device (rk_state_create) --------------------------------------
namespace bld
{

template <typename F>
__global__ void k_interleave(const F *x, const F *y, const F *z, const F *m, uint32_t n, typename vt<F>::v4 *part4)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        typename vt<F>::v4 p;
        p.x = x[i], p.y = y[i], p.z = z ? z[i] : F(0), p.w = m[i];
        part4[i] = p;
    }
}

// One record of rakau::tree_node_t<ND, F, uint64_t, MAC> (tree_fwd.hpp:77-116 of the reference: uint64 begin, end,
// n_children, code, level; F props[ND + 1]; F dim2 | F dim, delta) -> the arrays the rest of the build works on. A record
// that cannot be right (empty or out-of-range particle range, more descendants than nodes follow it) is reported through
// ctrl->pad[0] (~index of the first one) and neutralised, so that the kernels behind this one stay inside their arrays.
template <typename F, int ND>
__global__ void k_from_aos(const unsigned char *aos, uint32_t stride, uint32_t n_nodes, uint32_t nparts, int mac, uint4 *topo,
                           uint64_t *ncode, typename vt<F>::v4 *com, typename vt<F>::v2 *macp, uint32_t *parent, ctrl_block *ctrl)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_nodes) {
        return;
    }
    // k_parents_checked writes parent[c] only for the nodes some parent's walk reaches; the array comes from the block cache
    // uninitialised. Every node starts as "claimed by nobody" (k_flags treats any index >= k as that).
    parent[k] = 0xffffffffu;
    const unsigned char *rec = aos + static_cast<size_t>(k) * stride;
    const auto *hdr = reinterpret_cast<const uint64_t *>(rec);
    const auto *fp = reinterpret_cast<const F *>(rec + 5 * sizeof(uint64_t));
    uint64_t begin = hdr[0], end = hdr[1], nch = hdr[2];
    if (begin >= end || end > nparts || nch > static_cast<uint64_t>(n_nodes - 1u - k)) {
        atomicMax(&ctrl->pad[0], ~k);
        begin = 0, end = 1, nch = 0;
    }
    topo[k] = make_uint4(static_cast<uint32_t>(nch), static_cast<uint32_t>(begin), static_cast<uint32_t>(end), 0u);
    ncode[k] = hdr[3];
    typename vt<F>::v4 c;
    c.x = fp[0], c.y = fp[1], c.z = ND == 3 ? fp[2] : F(0), c.w = fp[ND];
    com[k] = c;
    typename vt<F>::v2 mp;
    mp.x = fp[ND + 1], mp.y = mac == RK_MAC_BH ? F(0) : fp[ND + 2];
    macp[k] = mp;
}

// k_parents with the host builder's consistency check: the children of k must tile (k, k + n_children(k)].
__global__ void k_parents_checked(const uint4 *topo, uint32_t n_nodes, uint32_t *parent, uint32_t *mask, ctrl_block *ctrl)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_nodes) {
        return;
    }
    mask[k] = 0u;
    if (k == 0u) {
        mask[n_nodes] = 0u;
    }
    const uint32_t last = k + topo[k].x;
    uint32_t c = k + 1u, cnt = 0u;
    for (; c <= last; c += topo[c].x + 1u) {
        parent[c] = k;
        ++cnt;
    }
    if (c != last + 1u || cnt > (1u << 3)) {
        atomicMax(&ctrl->pad[1], ~k);
    }
}

// The critical nodes must tile [0, nparts) in order.
__global__ void k_check_tiling(const uint4 *crit, uint32_t n_crit, uint32_t nparts, ctrl_block *ctrl)
{
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_crit) {
        return;
    }
    const uint32_t expect = g ? crit[g - 1u].y : 0u;
    if (crit[g].x != expect || (g + 1u == n_crit && crit[g].y != nparts)) {
        atomicOr(&ctrl->pad[2], 1u);
    }
}

} // namespace bld

// What rk_state_create() does with a tree built on the host: the particle arrays and the node records (the reference's own
// AoS layout) are uploaded AS THEY ARE and everything the kernels need is derived from them on the device -- {x, y, z, m}
// records, depth-first SoA copies, parents and child masks, critical nodes and their boxes, the sibling-ordered records, the
// lane-mapping classes -- with the kernels the device builder uses behind its own node sums. Round 3 did all of that in
// host loops (73-130 ms at 4M, of which the uploads of the derived arrays were a third); a reference tree that hands its
// arrays over after every update_particles() pays this on every time step. Same buffers, same contents: states made either
// way give the same bits (tests/test_gpu_state_create.py; RK_CREATE_ON_HOST=1 selects the host loops).
template <typename F, int ND>
void convert_device(rk_state &s, const void *const parts[4], int64_t nparts, const void *tree, int64_t tree_size,
                    int64_t node_stride)
{
    using namespace bld;
    using v4 = typename vt<F>::v4;
    using v2 = typename vt<F>::v2;
    hipStream_t st = nullptr;
    const auto n = static_cast<uint32_t>(nparts);
    const size_t nn = static_cast<size_t>(tree_size);
    auto alloc_buf = [&](int which, size_t bytes) {
        s.buf[which] = pool_alloc(std::max<size_t>(bytes, 16));
        s.buf_bytes[which] = static_cast<int64_t>(bytes);
        return s.buf[which];
    };
    auto ctrl = dalloc<ctrl_block>(1);
    RK_HIP(hipMemsetAsync(ctrl.get(), 0, sizeof(ctrl_block), st));
    ctrl_block hc{};
    auto fetch_ctrl = [&] { RK_HIP(hipMemcpy(&hc, ctrl.get(), sizeof(hc), hipMemcpyDeviceToHost)); };

    // ---- uploads: the caller's arrays, unconverted ----
    auto *p4 = static_cast<v4 *>(alloc_buf(RK_BUF_PART4, static_cast<size_t>(n) * sizeof(v4)));
    {
        dptr<F> in[4];
        const F *d[4] = {};
        for (int k = 0; k < 4; ++k) {
            const int src = k < ND ? k : (k == 3 ? ND : -1);
            if (src < 0) {
                continue;
            }
            in[k] = dalloc<F>(n);
            RK_HIP(hipMemcpyAsync(in[k].get(), parts[src], static_cast<size_t>(n) * sizeof(F), hipMemcpyHostToDevice, st));
            d[k] = in[k].get();
        }
        hipLaunchKernelGGL((k_interleave<F>), dim3(nblk(n)), dim3(256), 0, st, d[0], d[1], d[2], d[3], n, p4);
        RK_HIP(hipStreamSynchronize(st)); // the staging arrays go back to the block cache
    }
    auto aos = dalloc<unsigned char>(nn * static_cast<size_t>(node_stride));
    RK_HIP(hipMemcpyAsync(aos.get(), tree, nn * static_cast<size_t>(node_stride), hipMemcpyHostToDevice, st));
    auto *topo = static_cast<uint4 *>(alloc_buf(RK_BUF_NODE_TOPO, nn * sizeof(uint4)));
    auto *node_com = static_cast<v4 *>(alloc_buf(RK_BUF_NODE_COM, nn * sizeof(v4)));
    auto *node_mac = static_cast<v2 *>(alloc_buf(RK_BUF_NODE_MAC, nn * sizeof(v2)));
    auto *recs = static_cast<node_rec<F> *>(alloc_buf(RK_BUF_NODE_REC, nn * sizeof(node_rec<F>)));
    auto ncode = dalloc<uint64_t>(nn);
    auto parent = dalloc<uint32_t>(nn);
    auto mask = dalloc<uint32_t>(nn + 1);
    hipLaunchKernelGGL((k_from_aos<F, ND>), dim3(nblk(nn)), dim3(256), 0, st, aos.get(), static_cast<uint32_t>(node_stride),
                       static_cast<uint32_t>(nn), n, s.mac, topo, ncode.get(), node_com, node_mac, parent.get(), ctrl.get());
    hipLaunchKernelGGL(k_parents_checked, dim3(nblk(nn)), dim3(256), 0, st, topo, static_cast<uint32_t>(nn), parent.get(),
                       mask.get(), ctrl.get());

    // ---- critical nodes, child masks: one scan of three counters ----
    const auto ncrit_c = static_cast<uint32_t>(std::min<uint64_t>(s.ncrit, 0xffffffffu));
    auto flags = dalloc<tri>(nn + 1), offs = dalloc<tri>(nn + 1);
    hipLaunchKernelGGL(k_flags<ND>, dim3(nblk(nn)), dim3(256), 0, st, topo, ncode.get(), parent.get(), static_cast<uint32_t>(nn),
                       ncrit_c, flags.get(), mask.get());
    hipLaunchKernelGGL(k_popc, dim3(nblk(nn)), dim3(256), 0, st, mask.get(), static_cast<uint32_t>(nn), flags.get());
    exclusive_scan(flags.get(), offs.get(), nn, st);
    hipLaunchKernelGGL(k_pack_counts, dim3(1), dim3(1), 0, st, ctrl.get(), offs.get() + nn);
    fetch_ctrl();
    aos.reset();
    if (hc.pad[0]) {
        throw error(RK_EINVAL, "inconsistent tree node at index " + std::to_string(~hc.pad[0]));
    }
    if (hc.pad[1]) {
        throw error(RK_EINVAL, "inconsistent children counts below tree node " + std::to_string(~hc.pad[1]));
    }
    if (static_cast<size_t>(hc.n_children) + 1 != nn) {
        // (two children of one node in the same octant, or a node no parent claims)
        throw error(RK_EINVAL, "inconsistent tree: not every node is reachable from the root");
    }
    const uint32_t n_crit = hc.n_crit, n_int = hc.n_int;
    s.n_internal = n_int;
    auto *crit = static_cast<uint4 *>(alloc_buf(RK_BUF_CRIT, static_cast<size_t>(n_crit) * sizeof(uint4)));
    auto *boxes = static_cast<v4 *>(alloc_buf(RK_BUF_CRIT_BOX, static_cast<size_t>(n_crit) * 2 * sizeof(v4)));
    auto *child_tab = static_cast<uint32_t *>(alloc_buf(RK_BUF_CHILD, static_cast<size_t>(n_int) * 8 * sizeof(uint32_t)));
    {
        const auto tab_words = static_cast<uint32_t>(static_cast<size_t>(n_int) * 8u);
        const auto zper = static_cast<uint32_t>((static_cast<size_t>(tab_words) + nn - 1u) / nn);
        hipLaunchKernelGGL((k_crit<F>), dim3(nblk(nn)), dim3(256), 0, st, topo, flags.get(), offs.get(), static_cast<uint32_t>(nn), crit,
                           child_tab, tab_words, zper);
    }
    hipLaunchKernelGGL((k_crit_boxes<F>), dim3((n_crit + 3u) / 4u), dim3(256), 0, st, crit, n_crit, static_cast<const v4 *>(p4), boxes);
    hipLaunchKernelGGL(k_check_tiling, dim3(nblk(n_crit)), dim3(256), 0, st, crit, n_crit, n, ctrl.get());
    hipLaunchKernelGGL((k_records<F, ND>), dim3(nblk(nn)), dim3(256), 0, st, topo, ncode.get(), parent.get(), mask.get(), offs.get(),
                       static_cast<uint32_t>(nn), node_com, node_mac, recs, child_tab);
    // ---- lane-mapping classes (second half of RK_BUF_CLASS; the first half is filled with the host mirrors on demand) ----
    auto *lists = static_cast<uint32_t *>(alloc_buf(RK_BUF_CLASS, static_cast<size_t>(n_crit) * 2 * sizeof(uint32_t)));
    // (with the launch order of the first call: see bin_classes())
    const bin_tmp bin_scratch = bin_classes(s, crit, n_crit, n, lists + n_crit, ctrl.get(), st);
    fetch_ctrl();
    RK_HIP(hipGetLastError());
    if (hc.pad[2]) {
        throw error(RK_EINVAL, "the critical nodes derived from the tree do not tile the particle range");
    }
    s.n_crit = n_crit;
    s.max_group = hc.max_group;
    take_first_grid(s, hc);
    s.mirrors_valid = false;
    s.crit_begin.clear();
    s.crit_end.clear();
    s.class_off[0] = 0;
    s.class2_off[0] = n_crit;
    for (int c = 0; c < n_classes; ++c) {
        s.class_list[c].clear();
        s.class2_list[c].clear();
        s.class_off[c + 1] = 0;
        s.class2_count[c] = hc.class2_count[c];
        s.class2_off[c + 1] = s.class2_off[c] + s.class2_count[c];
    }
}
template void convert_device<float, 3>(rk_state &, const void *const[4], int64_t, const void *, int64_t, int64_t);
template void convert_device<double, 3>(rk_state &, const void *const[4], int64_t, const void *, int64_t, int64_t);
template void convert_device<float, 2>(rk_state &, const void *const[4], int64_t, const void *, int64_t, int64_t);
template void convert_device<double, 2>(rk_state &, const void *const[4], int64_t, const void *, int64_t, int64_t);

// Makes the runtime load this translation unit's code object now (rk_init) instead of at the first build.
void touch_build()
{
    hipFuncAttributes attr{};
    RK_HIP(hipFuncGetAttributes(&attr, reinterpret_cast<const void *>(&bld::k_pack_nodes)));
}

} // namespace rk
