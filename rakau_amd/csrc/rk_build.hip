// Device-side tree construction: everything the reference does in
// construct_impl() / build_tree() / compute_node_properties() (include/rakau/tree.hpp:1330-1487, 932-1111,
// 1116-1237 of the reference) runs on the GPU and leaves the traversal state resident in HBM:
//
//   max |coord| -> box size            (tree.hpp:1279-1319)
//   discretise + Morton encode         (tree.hpp:381-429, 1441-1450)
//   stable radix sort of (code, index) (tree.hpp:1267-1274; hipCUB DeviceRadixSort)
//   permute particles, perm            (tree.hpp:1461-1482)
//   node topology in depth-first order (tree.hpp:723-833)
//   node mass / centre of mass / size  (tree.hpp:1116-1237)
//   critical nodes                     (tree.hpp:801-807)
//   + the kernel-side structures of rk_state_create() (sibling-ordered records, group boxes, child table).
//
// Topology without a level loop: with sorted codes c[0..n), let ldiv(i) be the first level at which c[i] leaves
// the cell of c[i-1] and leaf(i) the level of the leaf that holds particle i. Particle i is the first particle of
// exactly the nodes at levels ldiv(i)..leaf(i); sorting nodes by (first particle, level) IS the depth-first order,
// so an exclusive scan of max(0, leaf(i) - ldiv(i) + 1) gives every node its depth-first index directly.
//
// Node properties are aggregated bottom-up (children -> parent, in child order; leaves summed serially in
// particle order like the reference). The reference sums every node's particles serially; its own SIMD flavour
// (tree.hpp:1134-1161) already associates differently, so centres of mass agree to rounding, not bit for bit.
#include "rk_common.hpp"

#include <hipcub/hipcub.hpp>

#include <cmath>
#include <cstring>

namespace rk
{
namespace bld
{

constexpr unsigned CBITS = 21;

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

__device__ inline float d_fma(float a, float b, float c)
{
    return __builtin_fmaf(a, b, c);
}
__device__ inline double d_fma(double a, double b, double c)
{
    return __builtin_fma(a, b, c);
}

// ---- box size -------------------------------------------------------------------------------------------
template <typename F>
__global__ void k_maxabs(const F *x, const F *y, const F *z, uint32_t n, unsigned long long *out_bits, int *err)
{
    F mx = F(0);
    bool bad = false;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const F a = fabs(x[i]), b = fabs(y[i]), c = fabs(z[i]);
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
    // One atomic per wavefront.
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long other = __shfl_xor(bits, o, 64);
        bits = other > bits ? other : bits;
    }
    const bool any_bad = __ballot(bad) != 0ull;
    if ((threadIdx.x & 63u) == 0u) {
        if (any_bad) {
            atomicOr(err, 1);
        }
        atomicMax(out_bits, bits);
    }
}

// ---- discretise + encode ----------------------------------------------------------------------------------
template <typename F>
__global__ void k_encode(const F *x, const F *y, const F *z, uint32_t n, F inv_box, uint64_t *codes, uint32_t *idx,
                         unsigned *first_bad)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) {
        return;
    }
    constexpr F factor = F(1u << CBITS);
    uint64_t d[3];
    const F v[3] = {x[i], y[i], z[i]};
    bool bad = false;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        F tmp = d_fma(v[k], inv_box, F(0.5));
        tmp *= factor;
        if (!isfinite(tmp) || tmp < F(0) || tmp >= factor) {
            bad = true;
            tmp = F(0);
        }
        d[k] = static_cast<uint64_t>(tmp);
    }
    if (bad) {
        atomicMin(first_bad, i);
    }
    codes[i] = spread3(d[0]) | (spread3(d[1]) << 1) | (spread3(d[2]) << 2);
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
    p.x = x[s], p.y = y[s], p.z = z[s], p.w = m[s];
    part4[i] = p;
}

// ---- topology ---------------------------------------------------------------------------------------------
// Range [lo, hi) of the particles sharing the level-`lvl` cell of particle i, searched inside the parent range.
__device__ inline void narrow(const uint64_t *codes, uint32_t i, unsigned lvl, uint32_t &lo, uint32_t &hi)
{
    const unsigned shift = 3u * (CBITS - lvl);
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

// Number of leading 3-bit digits (levels) two codes share: 0..CBITS. Codes use bits 0..62.
__device__ inline unsigned common_levels(uint64_t a, uint64_t b)
{
    const uint64_t xr = a ^ b;
    return xr ? (static_cast<unsigned>(__clzll(static_cast<long long>(xr))) - 1u) / 3u : CBITS;
}

// Depth of the leaf holding each particle, by search (any max_leaf_n): descend while the cell holds too many.
__global__ void k_leaf_levels_search(const uint64_t *codes, uint32_t n, uint32_t max_leaf_n, uint8_t *leaf)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) {
        return;
    }
    uint32_t lo = 0, hi = n;
    unsigned lvl = 0;
    while (hi - lo > max_leaf_n && lvl < CBITS) {
        ++lvl;
        narrow(codes, i, lvl, lo, hi);
    }
    leaf[i] = static_cast<uint8_t>(lvl);
}

// Same result without searching, for small max_leaf_n = m: the level-L cell of particle i holds more than m
// particles iff some window of m + 1 consecutive (sorted) particles containing i shares its first L digits, so
//   leaf(i) = min(CBITS, 1 + max_{j in [i-m, i], j+m < n} common_levels(c[j], c[j+m]))      (0 without windows).
// win[j] = common_levels(c[j], c[j+m]) + 1 for a valid window, 0 otherwise.
__global__ void k_windows(const uint64_t *codes, uint32_t n, uint32_t m, uint8_t *win)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) {
        return;
    }
    win[j] = (m < n && j < n - m) ? static_cast<uint8_t>(common_levels(codes[j], codes[j + m]) + 1u) : uint8_t(0);
}
__global__ void k_leaf_levels_windows(const uint8_t *win, uint32_t n, uint32_t m, uint8_t *leaf)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) {
        return;
    }
    unsigned best = 0;
    for (uint32_t j = i >= m ? i - m : 0u; j <= i; ++j) {
        best = max(best, static_cast<unsigned>(win[j]));
    }
    leaf[i] = static_cast<uint8_t>(min(best, CBITS));
}

// ldiv[i] = first level at which c[i] leaves the cell of c[i-1]; cnt[i] = number of nodes whose first particle is i.
__global__ void k_node_counts(const uint64_t *codes, uint32_t n, const uint8_t *leaf, uint8_t *ldiv, uint32_t *cnt)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) {
        return;
    }
    // Identical codes never start a node.
    const unsigned dv = i > 0 ? common_levels(codes[i - 1], codes[i]) + 1u : 1u;
    const unsigned lvl = leaf[i];
    ldiv[i] = static_cast<uint8_t>(dv);
    cnt[i] = dv <= lvl ? lvl - dv + 1u : 0u;
}

// Emit the nodes whose first particle is i. off[] = exclusive scan of cnt[] (off[n] = number of non-root nodes).
// The nodes starting at i are nested (levels ldiv(i)..leaf(i)); their ends are found deepest first by galloping
// from the end of the child, so a leaf of a dozen particles costs a handful of probes.
__global__ void k_emit_nodes(const uint64_t *codes, uint32_t n, const uint8_t *leaf, const uint8_t *ldiv,
                             const uint32_t *off, uint4 *topo, uint64_t *ncode, uint32_t *parent)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) {
        return;
    }
    const unsigned lf = leaf[i], dv = ldiv[i];
    if (i == 0) {
        topo[0] = make_uint4(off[n], 0u, n, 0u);
        ncode[0] = 1ull;
        parent[0] = 0xffffffffu;
    }
    if (dv > lf) {
        return;
    }
    const uint64_t ci = codes[i];
    const uint32_t base_dfs = 1u + off[i];
    uint32_t hi = i + 1u; // every particle in [i, hi) is known to lie in the current node
    for (unsigned lvl = lf; lvl >= dv; --lvl) {
        const unsigned shift = 3u * (CBITS - lvl);
        const uint64_t p = ci >> shift;
        // Smallest j >= hi with j == n or a different level-lvl prefix.
        uint32_t a = hi, b, step = 1u;
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
            step <<= 1;
        }
        while (a < b) {
            const uint32_t mid = a + (b - a) / 2u;
            if ((codes[mid] >> shift) == p) {
                a = mid + 1u;
            } else {
                b = mid;
            }
        }
        hi = a;
        const uint32_t dfs = base_dfs + (lvl - dv);
        const uint32_t next = 1u + off[hi]; // depth-first index of the first node starting at or after hi
        topo[dfs] = make_uint4(next - dfs - 1u, i, hi, 0u);
        ncode[dfs] = (1ull << (3u * lvl)) | p;
    }
}

// parent[] of every non-root node, written by the parent (children of k: k + 1, then skipping subtrees).
__global__ void k_parents(const uint4 *topo, uint32_t n_nodes, uint32_t *parent)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_nodes) {
        return;
    }
    const uint32_t last = k + topo[k].x;
    for (uint32_t c = k + 1u; c <= last; c += topo[c].x + 1u) {
        parent[c] = k;
    }
}

// ---- node properties --------------------------------------------------------------------------------------
__device__ inline unsigned level_of(uint64_t code)
{
    return (63u - static_cast<unsigned>(__clzll(static_cast<long long>(code)))) / 3u;
}

template <typename F>
__global__ void k_leaf_sums(const uint4 *topo, uint32_t n_nodes, const typename vt<F>::v4 *part4,
                            typename vt<F>::v4 *sums)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_nodes || topo[k].x != 0u) {
        return;
    }
    // Serial summation in particle order (tree.hpp:1162-1168 of the reference).
    F mt = F(0), sx = F(0), sy = F(0), sz = F(0);
    for (uint32_t i = topo[k].y; i < topo[k].z; ++i) {
        const typename vt<F>::v4 p = part4[i];
        mt += p.w;
        sx = d_fma(p.w, p.x, sx);
        sy = d_fma(p.w, p.y, sy);
        sz = d_fma(p.w, p.z, sz);
    }
    typename vt<F>::v4 s;
    s.x = sx, s.y = sy, s.z = sz, s.w = mt;
    sums[k] = s;
}

template <typename F>
__global__ void k_up_sums(const uint4 *topo, const uint64_t *ncode, uint32_t n_nodes, unsigned lvl,
                          typename vt<F>::v4 *sums)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_nodes || topo[k].x == 0u || level_of(ncode[k]) != lvl) {
        return;
    }
    F mt = F(0), sx = F(0), sy = F(0), sz = F(0);
    const uint32_t last = k + topo[k].x;
    for (uint32_t c = k + 1u; c <= last; c += topo[c].x + 1u) {
        const typename vt<F>::v4 s = sums[c];
        mt += s.w;
        sx += s.x;
        sy += s.y;
        sz += s.z;
    }
    typename vt<F>::v4 s;
    s.x = sx, s.y = sy, s.z = sz, s.w = mt;
    sums[k] = s;
}

template <typename F>
__global__ void k_finalize(const uint4 *topo, const uint64_t *ncode, uint32_t n_nodes, const typename vt<F>::v4 *sums,
                           F box, int mac, typename vt<F>::v4 *node_com, typename vt<F>::v2 *node_mac, int *err)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_nodes) {
        return;
    }
    const uint64_t code = ncode[k];
    const unsigned lvl = level_of(code);
    const typename vt<F>::v4 s = sums[k];
    // Geometric centre (tree.hpp:452-482 of the reference).
    const uint64_t first_cell = (code - (1ull << (3u * lvl))) << (3u * (CBITS - lvl));
    const F node_dim = box / static_cast<F>(1ull << lvl);
    const F half_dim = node_dim * F(0.5), cell = box * (F(1) / static_cast<F>(1ull << CBITS));
    F ctr[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        ctr[j] = d_fma(static_cast<F>(compact3(first_cell >> j)), cell, half_dim - box * F(0.5));
    }
    F com[3];
    if (s.w == F(0)) {
        com[0] = ctr[0], com[1] = ctr[1], com[2] = ctr[2];
    } else {
        const F inv = F(1) / s.w;
        com[0] = s.x * inv, com[1] = s.y * inv, com[2] = s.z * inv;
    }
    if (!(isfinite(com[0]) && isfinite(com[1]) && isfinite(com[2]) && isfinite(s.w))) {
        atomicOr(err, 2);
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
        atomicOr(err, 4);
    }
    node_mac[k] = mp;
}

// ---- critical nodes, child masks, records -----------------------------------------------------------------
__global__ void k_flags(const uint4 *topo, const uint64_t *ncode, const uint32_t *parent, uint32_t n_nodes,
                        uint32_t ncrit_clamped, uint32_t *is_crit, uint32_t *is_internal, uint32_t *mask)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_nodes) {
        return;
    }
    auto cand = [&](uint32_t j) { return (topo[j].z - topo[j].y) <= ncrit_clamped || topo[j].x == 0u; };
    const bool c = cand(k);
    is_crit[k] = (c && (k == 0u || !cand(parent[k]))) ? 1u : 0u;
    is_internal[k] = topo[k].x != 0u ? 1u : 0u;
    if (k != 0u) {
        atomicOr(&mask[parent[k]], 1u << static_cast<unsigned>(ncode[k] & 7ull));
    }
}

__global__ void k_popc(const uint32_t *mask, uint32_t n, uint32_t *out)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) {
        out[k] = static_cast<uint32_t>(__popc(mask[k]));
    }
}

template <typename F>
__global__ void k_crit(const uint4 *topo, const uint32_t *is_crit, const uint32_t *crit_off, uint32_t n_nodes,
                       const typename vt<F>::v4 *part4, uint4 *crit, typename vt<F>::v4 *boxes)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_nodes || !is_crit[k]) {
        return;
    }
    const uint32_t g = crit_off[k], b = topo[k].y, e = topo[k].z;
    crit[g] = make_uint4(b, e, k, e - b);
    typename vt<F>::v4 lo = part4[b], hi = lo;
    for (uint32_t i = b + 1u; i < e; ++i) {
        const typename vt<F>::v4 p = part4[i];
        lo.x = fmin(lo.x, p.x), lo.y = fmin(lo.y, p.y), lo.z = fmin(lo.z, p.z);
        hi.x = fmax(hi.x, p.x), hi.y = fmax(hi.y, p.y), hi.z = fmax(hi.z, p.z);
    }
    lo.w = hi.w = F(0);
    boxes[2u * g] = lo;
    boxes[2u * g + 1u] = hi;
}

template <typename F>
__global__ void k_records(uint4 *topo, const uint64_t *ncode, const uint32_t *parent, const uint32_t *mask,
                          const uint32_t *child_off, const uint32_t *slot_off, uint32_t n_nodes,
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
        const unsigned digit = static_cast<unsigned>(ncode[k] & 7ull);
        const uint32_t rank = static_cast<uint32_t>(__popc(mask[p] & ((1u << digit) - 1u)));
        rec = 1u + child_off[p] + rank; // children of p occupy records [1 + child_off[p], ...)
        child_tab[static_cast<size_t>(slot_off[p]) * 8u + rank] = k;
    }
    const uint4 t = topo[k];
    node_rec<F> r;
    r.com = node_com[k];
    r.mac = node_mac[k];
    r.dfs = k;
    r.nch = t.x;
    r.pad[0] = r.pad[1] = 0u;
    if (t.x != 0u) {
        r.a = 1u + child_off[k];
        r.b = static_cast<uint32_t>(__popc(mask[k]));
        topo[k].w = slot_off[k];
    } else {
        r.a = t.y;
        r.b = t.z;
        topo[k].w = 0xffffffffu;
    }
    recs[rec] = r;
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
    RK_HIP(hipMemsetAsync(out + n, 0, sizeof(uint32_t), st));
    size_t tb = 0;
    RK_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tb, in, out, static_cast<int>(n + 1), st));
    auto tmp = dalloc<unsigned char>(tb);
    // Scan n + 1 elements with a zero appended: requires in[n] readable; callers over-allocate by one and zero it.
    RK_HIP(hipcub::DeviceScan::ExclusiveSum(tmp.get(), tb, in, out, static_cast<int>(n + 1), st));
    RK_HIP(hipStreamSynchronize(st));
}

} // namespace bld

// Builds the tree and fills `s` (buffers, sizes). Host inputs in the caller's original order.
template <typename F>
void build_device(rk_state &s, const void *const parts[4], bool parts_on_device, int64_t nparts, double box_size_in,
                  uint64_t max_leaf_n, std::string &bad_coord_msg)
{
    using namespace bld;
    using v4 = typename vt<F>::v4;
    using v2 = typename vt<F>::v2;
    hipStream_t st = nullptr;
    const auto n = static_cast<uint32_t>(nparts);
    const size_t fb = static_cast<size_t>(n) * sizeof(F);

    // Inputs: copied from the host, or used in place when they already live on this device.
    dptr<F> own[4];
    const F *in[4];
    for (int k = 0; k < 4; ++k) {
        if (parts_on_device) {
            in[k] = static_cast<const F *>(parts[k]);
        } else {
            own[k] = dalloc<F>(n);
            RK_HIP(hipMemcpyAsync(own[k].get(), parts[k], fb, hipMemcpyHostToDevice, st));
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

    auto d_err = dalloc<int>(1);
    auto d_bits = dalloc<unsigned long long>(1);
    auto d_bad = dalloc<unsigned>(1);
    RK_HIP(hipMemsetAsync(d_err.get(), 0, sizeof(int), st));
    RK_HIP(hipMemsetAsync(d_bits.get(), 0, sizeof(unsigned long long), st));
    RK_HIP(hipMemsetAsync(d_bad.get(), 0xff, sizeof(unsigned), st));

    // ---- box size ----
    F box = static_cast<F>(box_size_in);
    s.box_deduced = box_size_in == 0.;
    if (s.box_deduced) {
        hipLaunchKernelGGL((k_maxabs<F>), dim3(std::min(nblk(n), 2048u)), dim3(256), 0, st, dx.get(), dy.get(), dz.get(), n,
                           d_bits.get(), d_err.get());
        unsigned long long bits = 0;
        int err = 0;
        RK_HIP(hipMemcpy(&bits, d_bits.get(), sizeof(bits), hipMemcpyDeviceToHost));
        RK_HIP(hipMemcpy(&err, d_err.get(), sizeof(err), hipMemcpyDeviceToHost));
        if (err) {
            throw error(RK_EINVAL, "While trying to automatically determine the domain size, a non-finite coordinate "
                                   "was encountered");
        }
        F mx;
        if constexpr (sizeof(F) == 4) {
            const auto b32 = static_cast<uint32_t>(bits);
            std::memcpy(&mx, &b32, 4);
        } else {
            std::memcpy(&mx, &bits, 8);
        }
        F b = mx * F(2);
        b = std::fma(b, F(1) / F(20), b); // 5% slack, tree.hpp:1310-1312
        if (!std::isfinite(b)) {
            throw error(RK_EINVAL, "The automatic deduction of the domain size produced the non-finite value "
                                       + std::to_string(b));
        }
        box = b;
    }
    s.box_size = static_cast<double>(box);
    const F inv_box = F(1) / box;

    // ---- encode + sort ----
    auto keys_in = dalloc<uint64_t>(n), keys_out = dalloc<uint64_t>(n);
    auto vals_in = dalloc<uint32_t>(n), vals_out = dalloc<uint32_t>(n);
    hipLaunchKernelGGL((k_encode<F>), dim3(nblk(n)), dim3(256), 0, st, dx.get(), dy.get(), dz.get(), n, inv_box,
                       keys_in.get(), vals_in.get(), d_bad.get());
    unsigned first_bad = 0xffffffffu;
    RK_HIP(hipMemcpy(&first_bad, d_bad.get(), sizeof(first_bad), hipMemcpyDeviceToHost));
    if (first_bad != 0xffffffffu) {
        // Rebuild the reference's message (tree.hpp:398-413) for the first offending coordinate.
        for (int k = 0; k < 3; ++k) {
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
    {
        size_t tb = 0;
        RK_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, tb, keys_in.get(), keys_out.get(), vals_in.get(),
                                                  vals_out.get(), static_cast<int>(n), 0, 63, st));
        auto tmp = dalloc<unsigned char>(tb);
        RK_HIP(hipcub::DeviceRadixSort::SortPairs(tmp.get(), tb, keys_in.get(), keys_out.get(), vals_in.get(),
                                                  vals_out.get(), static_cast<int>(n), 0, 63, st));
        RK_HIP(hipStreamSynchronize(st));
    }
    keys_in.reset();
    vals_in.reset();

    // ---- particles in Morton order ----
    void *p4 = pool_alloc(std::max<size_t>(n, 1) * sizeof(v4));
    s.buf[RK_BUF_PART4] = p4;
    s.buf_bytes[RK_BUF_PART4] = static_cast<int64_t>(n * sizeof(v4));
    hipLaunchKernelGGL((k_permute<F>), dim3(nblk(n)), dim3(256), 0, st, dx.get(), dy.get(), dz.get(), dm.get(),
                       vals_out.get(), n, static_cast<v4 *>(p4));
    RK_HIP(hipStreamSynchronize(st));
    for (auto &o : own) {
        o.reset();
    }
    s.bld_codes = keys_out.release();
    s.bld_perm = vals_out.release();
    const auto *codes = static_cast<const uint64_t *>(s.bld_codes);

    // ---- topology ----
    const auto mln = static_cast<uint32_t>(std::min<uint64_t>(max_leaf_n, 0xffffffffu));
    auto leaf = dalloc<uint8_t>(n), ldiv = dalloc<uint8_t>(n);
    auto cnt = dalloc<uint32_t>(static_cast<size_t>(n) + 1), off = dalloc<uint32_t>(static_cast<size_t>(n) + 1);
    RK_HIP(hipMemsetAsync(cnt.get() + n, 0, sizeof(uint32_t), st));
    if (mln <= 64u) {
        // ldiv doubles as the window scratch until k_node_counts fills it.
        hipLaunchKernelGGL(k_windows, dim3(nblk(n)), dim3(256), 0, st, codes, n, mln, ldiv.get());
        hipLaunchKernelGGL(k_leaf_levels_windows, dim3(nblk(n)), dim3(256), 0, st, ldiv.get(), n, mln, leaf.get());
    } else {
        hipLaunchKernelGGL(k_leaf_levels_search, dim3(nblk(n)), dim3(256), 0, st, codes, n, mln, leaf.get());
    }
    hipLaunchKernelGGL(k_node_counts, dim3(nblk(n)), dim3(256), 0, st, codes, n, leaf.get(), ldiv.get(), cnt.get());
    exclusive_scan(cnt.get(), off.get(), n, st);
    uint32_t n_nonroot = 0;
    RK_HIP(hipMemcpy(&n_nonroot, off.get() + n, sizeof(uint32_t), hipMemcpyDeviceToHost));
    const size_t nn = static_cast<size_t>(n_nonroot) + 1;
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
    hipLaunchKernelGGL(k_emit_nodes, dim3(nblk(n)), dim3(256), 0, st, codes, n, leaf.get(), ldiv.get(), off.get(), topo,
                       ncode, parent.get());
    hipLaunchKernelGGL(k_parents, dim3(nblk(nn)), dim3(256), 0, st, topo, static_cast<uint32_t>(nn), parent.get());
    leaf.reset(), ldiv.reset(), cnt.reset(), off.reset();

    // ---- node properties ----
    auto sums = dalloc<v4>(nn);
    hipLaunchKernelGGL((k_leaf_sums<F>), dim3(nblk(nn)), dim3(256), 0, st, topo, static_cast<uint32_t>(nn),
                       static_cast<const v4 *>(p4), sums.get());
    for (int lvl = static_cast<int>(CBITS) - 1; lvl >= 0; --lvl) {
        hipLaunchKernelGGL((k_up_sums<F>), dim3(nblk(nn)), dim3(256), 0, st, topo, ncode, static_cast<uint32_t>(nn),
                           static_cast<unsigned>(lvl), sums.get());
    }
    RK_HIP(hipMemsetAsync(d_err.get(), 0, sizeof(int), st));
    hipLaunchKernelGGL((k_finalize<F>), dim3(nblk(nn)), dim3(256), 0, st, topo, ncode, static_cast<uint32_t>(nn), sums.get(),
                       box, s.mac, node_com, node_mac, d_err.get());
    {
        int err = 0;
        RK_HIP(hipMemcpy(&err, d_err.get(), sizeof(err), hipMemcpyDeviceToHost));
        if (err & 2) {
            throw error(RK_EINVAL, "The computation of the centre of mass of a node produced a non-finite value");
        }
        if (err & 4) {
            throw error(RK_EINVAL, "The computation of the dimension of a node produced a non-finite value");
        }
    }
    sums.reset();

    // ---- critical nodes, child masks ----
    const auto ncrit_c = static_cast<uint32_t>(std::min<uint64_t>(s.ncrit, 0xffffffffu));
    auto is_crit = dalloc<uint32_t>(nn + 1), is_int = dalloc<uint32_t>(nn + 1), mask = dalloc<uint32_t>(nn + 1),
         nchild = dalloc<uint32_t>(nn + 1);
    auto crit_off = dalloc<uint32_t>(nn + 1), slot_off = dalloc<uint32_t>(nn + 1), child_off = dalloc<uint32_t>(nn + 1);
    RK_HIP(hipMemsetAsync(mask.get(), 0, (nn + 1) * sizeof(uint32_t), st));
    RK_HIP(hipMemsetAsync(is_crit.get() + nn, 0, sizeof(uint32_t), st));
    RK_HIP(hipMemsetAsync(is_int.get() + nn, 0, sizeof(uint32_t), st));
    hipLaunchKernelGGL(k_flags, dim3(nblk(nn)), dim3(256), 0, st, topo, ncode, parent.get(), static_cast<uint32_t>(nn), ncrit_c,
                       is_crit.get(), is_int.get(), mask.get());
    hipLaunchKernelGGL(k_popc, dim3(nblk(nn + 1)), dim3(256), 0, st, mask.get(), static_cast<uint32_t>(nn + 1), nchild.get());
    exclusive_scan(is_crit.get(), crit_off.get(), nn, st);
    exclusive_scan(is_int.get(), slot_off.get(), nn, st);
    exclusive_scan(nchild.get(), child_off.get(), nn, st);
    uint32_t n_crit = 0, n_int = 0, n_children_total = 0;
    RK_HIP(hipMemcpy(&n_crit, crit_off.get() + nn, sizeof(uint32_t), hipMemcpyDeviceToHost));
    RK_HIP(hipMemcpy(&n_int, slot_off.get() + nn, sizeof(uint32_t), hipMemcpyDeviceToHost));
    RK_HIP(hipMemcpy(&n_children_total, child_off.get() + nn, sizeof(uint32_t), hipMemcpyDeviceToHost));
    if (static_cast<size_t>(n_children_total) + 1 != nn) {
        throw error(RK_ERUNTIME, "internal error: inconsistent node count in the device tree build");
    }
    s.n_internal = n_int;
    auto *crit = static_cast<uint4 *>(alloc_buf(RK_BUF_CRIT, static_cast<size_t>(n_crit) * sizeof(uint4)));
    auto *boxes = static_cast<v4 *>(alloc_buf(RK_BUF_CRIT_BOX, static_cast<size_t>(n_crit) * 2 * sizeof(v4)));
    auto *child_tab = static_cast<uint32_t *>(alloc_buf(RK_BUF_CHILD, static_cast<size_t>(n_int) * 8 * sizeof(uint32_t)));
    RK_HIP(hipMemsetAsync(child_tab, 0, std::max<size_t>(static_cast<size_t>(n_int) * 8 * sizeof(uint32_t), 16), st));
    hipLaunchKernelGGL((k_crit<F>), dim3(nblk(nn)), dim3(256), 0, st, topo, is_crit.get(), crit_off.get(),
                       static_cast<uint32_t>(nn), static_cast<const v4 *>(p4), crit, boxes);
    hipLaunchKernelGGL((k_records<F>), dim3(nblk(nn)), dim3(256), 0, st, topo, ncode, parent.get(), mask.get(), child_off.get(),
                       slot_off.get(), static_cast<uint32_t>(nn), node_com, node_mac, recs, child_tab);
    RK_HIP(hipStreamSynchronize(st));
    RK_HIP(hipGetLastError());
}

template void build_device<float>(rk_state &, const void *const[4], bool, int64_t, double, uint64_t, std::string &);
template void build_device<double>(rk_state &, const void *const[4], bool, int64_t, double, uint64_t, std::string &);

} // namespace rk
