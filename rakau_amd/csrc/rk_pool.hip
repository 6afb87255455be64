// Caching device allocator for state buffers and tree-build temporaries (see rk_common.hpp).
#include "rk_common.hpp"

#include <cstdlib>
#include <map>
#include <mutex>
#include <unordered_map>

namespace rk
{
namespace
{

struct pool {
    std::mutex mtx;
    // device -> (rounded size -> cached blocks)
    std::map<int, std::multimap<size_t, void *>> free_blocks;
    std::unordered_map<void *, std::pair<int, size_t>> live; // block -> (device, rounded size)
    std::map<int, size_t> cached_bytes;                      // device -> bytes sitting in free_blocks
};

// Upper bound of the bytes kept per device (RK_POOL_MAX_MB, default 16 GiB): beyond it freed blocks go back to the
// driver, so that a long run whose buffer sizes drift from bin to bin cannot accumulate dead bins without limit.
size_t pool_limit()
{
    static const size_t v = [] {
        const char *e = std::getenv("RK_POOL_MAX_MB");
        const long long mb = e ? std::atoll(e) : 16384;
        return static_cast<size_t>(mb < 0 ? 0 : mb) << 20;
    }();
    return v;
}

pool &the_pool()
{
    static pool *p = new pool; // leaked on purpose: no HIP calls during static destruction
    return *p;
}

bool pool_enabled()
{
    static const bool on = [] {
        const char *e = std::getenv("RK_POOL");
        return !(e && std::atoi(e) == 0);
    }();
    return on;
}

// Sizes are rounded up to 4 significant bits (<= 6.25% slack) with a 512-byte floor, so that the buffers of a tree
// rebuilt with slightly different node counts land in the same bins.
size_t round_size(size_t b)
{
    if (b <= 512) {
        return 512;
    }
    const int hi = 63 - __builtin_clzll(static_cast<unsigned long long>(b));
    const size_t step = size_t(1) << (hi > 4 ? hi - 4 : 0);
    return (b + step - 1) & ~(step - 1);
}

} // namespace

// RK_POOL_POISON=<byte>: fill every block handed out with that byte (diagnostic for reads of uninitialised memory).
int poison_byte()
{
    static const int v = [] {
        const char *e = std::getenv("RK_POOL_POISON");
        return e ? (std::atoi(e) & 0xff) : -1;
    }();
    return v;
}

void *pool_alloc_raw(size_t bytes);

void *pool_alloc(size_t bytes)
{
    void *p = pool_alloc_raw(bytes);
    if (poison_byte() >= 0) {
        RK_HIP(hipMemset(p, poison_byte(), bytes ? bytes : 1));
    }
    return p;
}

void *pool_alloc_raw(size_t bytes)
{
    void *p = nullptr;
    if (!pool_enabled()) {
        RK_HIP(hipMalloc(&p, bytes ? bytes : 1));
        return p;
    }
    int dev = 0;
    RK_HIP(hipGetDevice(&dev));
    const size_t rs = round_size(bytes);
    auto &P = the_pool();
    {
        std::lock_guard<std::mutex> lk(P.mtx);
        auto &fb = P.free_blocks[dev];
        auto it = fb.find(rs);
        if (it != fb.end()) {
            p = it->second;
            fb.erase(it);
            P.cached_bytes[dev] -= rs;
            P.live.emplace(p, std::make_pair(dev, rs));
            return p;
        }
    }
    hipError_t e = hipMalloc(&p, rs);
    if (e == hipErrorOutOfMemory) {
        (void)hipGetLastError();
        pool_trim();
        e = hipMalloc(&p, rs);
    }
    RK_HIP(e);
    std::lock_guard<std::mutex> lk(P.mtx);
    P.live.emplace(p, std::make_pair(dev, rs));
    return p;
}

void pool_free(void *p) noexcept
{
    if (!p) {
        return;
    }
    auto &P = the_pool();
    {
        std::lock_guard<std::mutex> lk(P.mtx);
        auto it = P.live.find(p);
        if (it != P.live.end()) {
            const int dev = it->second.first;
            const size_t rs = it->second.second;
            P.live.erase(it);
            if (P.cached_bytes[dev] + rs <= pool_limit()) {
                P.free_blocks[dev].emplace(rs, p);
                P.cached_bytes[dev] += rs;
                return;
            }
            // Over the limit: hand the block back to the driver (hipFree waits for the device, so this is safe
            // whatever is still in flight).
        }
    }
    (void)hipFree(p); // not ours (pool disabled), or over the cache limit
}

void pool_trim() noexcept
{
    auto &P = the_pool();
    std::map<int, std::multimap<size_t, void *>> blocks;
    {
        std::lock_guard<std::mutex> lk(P.mtx);
        blocks.swap(P.free_blocks);
        P.cached_bytes.clear();
    }
    int prev = 0;
    (void)hipGetDevice(&prev);
    for (auto &d : blocks) {
        (void)hipSetDevice(d.first);
        for (auto &b : d.second) {
            (void)hipFree(b.second);
        }
    }
    (void)hipSetDevice(prev);
}

} // namespace rk
