// rk_acc_pot(): the seam's own signature -- results delivered into the caller's HOST arrays (pinned: written by the kernels;
// pageable: staged and delivered by parked host threads) -- rk_acc_pot_device(), and the pinned host blocks of rk_host_alloc().
#include "rk_state_internal.hpp"

extern "C" {

namespace
{

// Device-side address of [p, p + bytes) if the whole range is host memory the device can write to (hipHostMalloc /
// hipHostRegister: rk_host_alloc(), a pinned torch tensor, a user's registered vector), nullptr for pageable memory.
void *device_view_of_host_range(void *p, size_t bytes)
{
    if (!p || !bytes) {
        return nullptr;
    }
    hipPointerAttribute_t a0{}, a1{};
    if (hipPointerGetAttributes(&a0, p) != hipSuccess
        || hipPointerGetAttributes(&a1, static_cast<unsigned char *>(p) + bytes - 1u) != hipSuccess) {
        (void)hipGetLastError(); // pageable memory is reported as an error: clear it
        return nullptr;
    }
    if (a0.type != hipMemoryTypeHost || a1.type != hipMemoryTypeHost || !a0.devicePointer || !a1.devicePointer) {
        return nullptr;
    }
    // One registration: the device view is contiguous over the range.
    if (static_cast<unsigned char *>(a1.devicePointer) - static_cast<unsigned char *>(a0.devicePointer)
        != static_cast<ptrdiff_t>(bytes - 1u)) {
        return nullptr;
    }
    return a0.devicePointer;
}

// memcpy with non-temporal stores for the 16-byte aligned body of the destination (movntdq on the host): the delivery of a
// staged result overwrites whole cache lines that nobody reads soon.
inline void stream_copy(unsigned char *d, const unsigned char *src, size_t n)
{
    typedef long long v2di __attribute__((vector_size(16)));
    typedef long long v2di_u __attribute__((vector_size(16), aligned(1)));
    if (n < 256) {
        std::memcpy(d, src, n);
        return;
    }
    const size_t head = (16 - (reinterpret_cast<uintptr_t>(d) & 15)) & 15;
    std::memcpy(d, src, head);
    d += head;
    src += head;
    n -= head;
    const size_t body = n & ~size_t(63);
    for (size_t i = 0; i < body; i += 64) {
        const v2di a = *reinterpret_cast<const v2di_u *>(src + i), b = *reinterpret_cast<const v2di_u *>(src + i + 16),
                   c = *reinterpret_cast<const v2di_u *>(src + i + 32), e = *reinterpret_cast<const v2di_u *>(src + i + 48);
        __builtin_nontemporal_store(a, reinterpret_cast<v2di *>(d + i));
        __builtin_nontemporal_store(b, reinterpret_cast<v2di *>(d + i + 16));
        __builtin_nontemporal_store(c, reinterpret_cast<v2di *>(d + i + 32));
        __builtin_nontemporal_store(e, reinterpret_cast<v2di *>(d + i + 48));
    }
    std::memcpy(d + body, src + body, n - body);
}

// The host threads that deliver staged results: created once (a thread costs ~20 us to start and to join, seven of them
// per call were 5% of a 4M-particle call), parked on a condition variable between calls. Several callers (the device
// threads of a multi-device split) may post jobs at the same time; every caller also works on its own job.
class delivery_pool
{
    struct job {
        const std::function<void(int)> *fn;
        int n_items, max_helpers;
        std::atomic<int> next{0}, done{0}, helpers{0};
    };
    std::mutex m_;
    std::condition_variable cv_;
    std::vector<std::thread> thr_;
    std::deque<std::shared_ptr<job>> jobs_;
    bool stop_ = false;

    static void work(job &j)
    {
        for (;;) {
            const int item = j.next.fetch_add(1, std::memory_order_relaxed);
            if (item >= j.n_items) {
                return;
            }
            (*j.fn)(item);
            j.done.fetch_add(1, std::memory_order_release);
        }
    }
    void loop()
    {
        for (;;) {
            std::shared_ptr<job> j;
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&] {
                    while (!jobs_.empty() && jobs_.front()->next.load(std::memory_order_relaxed) >= jobs_.front()->n_items) {
                        jobs_.pop_front();
                    }
                    return stop_ || !jobs_.empty();
                });
                if (stop_) {
                    return;
                }
                j = jobs_.front();
                if (j->helpers.fetch_add(1) >= j->max_helpers) { // enough hands on this one: look at the next, or sleep
                    j.reset();
                    for (auto &o : jobs_) {
                        if (o->next.load(std::memory_order_relaxed) < o->n_items && o->helpers.fetch_add(1) < o->max_helpers) {
                            j = o;
                            break;
                        }
                    }
                    if (!j) {
                        cv_.wait_for(lk, std::chrono::microseconds(200));
                        continue;
                    }
                }
            }
            work(*j);
        }
    }

public:
    static delivery_pool &get()
    {
        static delivery_pool p;
        return p;
    }
    ~delivery_pool()
    {
        {
            std::lock_guard<std::mutex> lk(m_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto &t : thr_) {
            t.join();
        }
    }
    // fn(item) for item in [0, n_items), on the caller's thread and up to n_thr - 1 pool threads; returns when all are done.
    void run(int n_items, int n_thr, const std::function<void(int)> &fn)
    {
        auto j = std::make_shared<job>();
        j->fn = &fn;
        j->n_items = n_items;
        j->max_helpers = n_thr - 1;
        if (n_thr > 1) {
            {
                std::lock_guard<std::mutex> lk(m_);
                while (static_cast<int>(thr_.size()) < n_thr - 1) {
                    thr_.emplace_back([this] { loop(); });
                }
                jobs_.push_back(j);
            }
            cv_.notify_all();
        }
        work(*j);
        while (j->done.load(std::memory_order_acquire) < n_items) {
            std::this_thread::yield();
        }
        // Workers that still hold the job only look at its counters (the shared_ptr keeps them alive); fn is not called again.
    }
};

} // namespace

// rk_host_alloc() / rk_host_free(): pinned, device-visible host memory. Blocks of 1 MiB and more that are freed are parked (at most
// eight of them, 512 MiB in all) and handed out again to requests they fit within a factor of two: pinning costs milliseconds per
// 16 MiB, and the C++ header's staged overloads give their buffers back after every call on a large tree (tree.hpp,
// stage_buffers::trim) instead of keeping them per thread for the life of the process. rk_pool_trim() releases what is parked.
namespace
{
std::mutex g_host_mtx;
std::unordered_map<void *, size_t> g_host_live; // blocks handed out by rk_host_alloc -> their size
struct parked_host {
    void *p;
    size_t bytes;
};
std::vector<parked_host> g_host_parked;
size_t g_host_parked_bytes = 0;
} // namespace

void host_blocks_trim()
{
    std::vector<parked_host> v;
    {
        std::lock_guard<std::mutex> lk(g_host_mtx);
        v.swap(g_host_parked);
        g_host_parked_bytes = 0;
    }
    for (auto &e : v) {
        (void)hipHostFree(e.p);
    }
}

int rk_host_alloc(void **ptr, int64_t bytes)
{
    return guard([&] {
        if (!ptr || bytes < 0) {
            throw rk::error(RK_EINVAL, "rk_host_alloc: null pointer or negative size");
        }
        *ptr = nullptr;
        if (!bytes) {
            return;
        }
        const auto need = static_cast<size_t>(bytes);
        {
            std::lock_guard<std::mutex> lk(g_host_mtx);
            size_t best = g_host_parked.size();
            for (size_t i = 0; i < g_host_parked.size(); ++i) {
                if (g_host_parked[i].bytes >= need && g_host_parked[i].bytes <= 2 * need
                    && (best == g_host_parked.size() || g_host_parked[i].bytes < g_host_parked[best].bytes)) {
                    best = i;
                }
            }
            if (best != g_host_parked.size()) {
                *ptr = g_host_parked[best].p;
                g_host_live.emplace(*ptr, g_host_parked[best].bytes);
                g_host_parked_bytes -= g_host_parked[best].bytes;
                g_host_parked.erase(g_host_parked.begin() + static_cast<std::ptrdiff_t>(best));
                return;
            }
        }
        RK_HIP(hipHostMalloc(ptr, need, hipHostMallocPortable));
        std::lock_guard<std::mutex> lk(g_host_mtx);
        g_host_live.emplace(*ptr, need);
    });
}

int rk_host_free(void *ptr)
{
    return guard([&] {
        if (!ptr) {
            return;
        }
        void *drop = ptr;
        {
            std::lock_guard<std::mutex> lk(g_host_mtx);
            const auto it = g_host_live.find(ptr);
            if (it != g_host_live.end()) {
                const size_t bytes = it->second;
                g_host_live.erase(it);
                if (bytes >= (size_t(1) << 20) && g_host_parked.size() < 8 && g_host_parked_bytes + bytes <= (size_t(512) << 20)) {
                    g_host_parked.push_back(parked_host{ptr, bytes});
                    g_host_parked_bytes += bytes;
                    drop = nullptr;
                }
            }
        }
        if (drop) {
            RK_HIP(hipHostFree(drop));
        }
    });
}

int rk_acc_pot_device(rk_state *s, int q, int64_t p_begin, int64_t p_end, void *const *d_out, double mac_value,
                      double G, double eps2, int offset_output, void *hip_stream)
{
    return guard([&] {
        check_call(s, q, d_out, mac_value, G, eps2);
        device_guard dg(s->device);
        auto stream = static_cast<hipStream_t>(hip_stream);
        if (s->fp == RK_F32) {
            run_impl<float>(*s, q, p_begin, p_end, d_out, mac_value, G, eps2, offset_output, stream);
        } else {
            run_impl<double>(*s, q, p_begin, p_end, d_out, mac_value, G, eps2, offset_output, stream);
        }
    });
}

int rk_acc_pot(rk_state *s, int q, int64_t p_begin, int64_t p_end, void *const *out, double mac_value, double G,
               double eps2, int offset_output)
{
    return guard([&] {
        check_call(s, q, out, mac_value, G, eps2);
        if (p_begin < 0 || p_end < p_begin || p_end > s->nparts) {
            throw rk::error(RK_EINVAL, "invalid particle range");
        }
        if (offset_output & ~(RK_OUT_OFFSET | RK_OUT_ORDERED)) {
            throw rk::error(RK_EINVAL, "rk_acc_pot(): invalid output flags");
        }
        if ((offset_output & RK_OUT_ORDERED) && (p_begin != 0 || p_end != s->nparts)) {
            throw rk::error(RK_EINVAL, "rk_acc_pot() with RK_OUT_ORDERED (original-order host outputs) takes the whole range "
                                       "[0, nparts): the results are scattered all over the output arrays");
        }
        const size_t fsz = s->fp == RK_F32 ? sizeof(float) : sizeof(double);
        const auto count = static_cast<size_t>(p_end - p_begin);
        const int nres = user_nres(*s, q);
        if (!count) {
            return;
        }
        device_guard dg(s->device);
        const size_t need = count * fsz * static_cast<size_t>(nres);
        if (offset_output & RK_OUT_ORDERED) {
            // accs_o / pots_o for host arrays (tree.hpp:3320-3330): the kernels scatter the results through perm into a buffer
            // in HBM (where random 4-byte stores cost nothing: the RK_OUT_ORDERED epilogue of rk_acc_pot_device), the ordered
            // arrays then travel in one piece each -- straight into pinned arrays, through the staging buffer and the host
            // threads otherwise, array k being delivered while array k + 1 is still on its way. A host-side scatter of
            // 3 x 4M values through a random permutation costs 6 ms on eight threads (30 ms on one); this, 1.3 ms.
            if (s->d_out_bytes < need) {
                if (s->d_out) {
                    RK_HIP(hipDeviceSynchronize());
                    rk::pool_free(s->d_out);
                    s->d_out = nullptr;
                    s->d_out_bytes = 0;
                }
                s->d_out = rk::pool_alloc(need);
                s->d_out_bytes = need;
            }
            const size_t arr = count * fsz;
            void *d_ptrs[4] = {};
            unsigned char *dst[4] = {};
            for (int k = 0; k < nres; ++k) {
                d_ptrs[k] = static_cast<unsigned char *>(s->d_out) + static_cast<size_t>(k) * arr;
                dst[k] = static_cast<unsigned char *>(out[k]);
            }
            if (s->fp == RK_F32) {
                run_impl<float>(*s, q, p_begin, p_end, d_ptrs, mac_value, G, eps2, RK_OUT_OFFSET | RK_OUT_ORDERED, nullptr, false);
            } else {
                run_impl<double>(*s, q, p_begin, p_end, d_ptrs, mac_value, G, eps2, RK_OUT_OFFSET | RK_OUT_ORDERED, nullptr, false);
            }
            bool pinned = true;
            for (int k = 0; pinned && k < nres; ++k) {
                pinned = device_view_of_host_range(dst[k], arr) != nullptr;
            }
            if (pinned || need < (size_t(1) << 20)) {
                for (int k = 0; k < nres; ++k) {
                    RK_HIP(hipMemcpyAsync(dst[k], d_ptrs[k], arr, hipMemcpyDeviceToHost, nullptr));
                }
                RK_HIP(hipStreamSynchronize(nullptr));
                return;
            }
            if (s->h_stage_bytes < need) {
                if (s->h_stage) {
                    RK_HIP(hipDeviceSynchronize());
                    stage_give(phys(s->device), s->h_stage, s->h_stage_bytes);
                    s->h_stage = nullptr;
                    s->h_stage_bytes = 0;
                }
                size_t got = 0;
                s->h_stage = stage_take(phys(s->device), need, got);
                if (s->h_stage) {
                    s->h_stage_bytes = got;
                } else {
                    RK_HIP(hipHostMalloc(&s->h_stage, need, hipHostMallocDefault));
                    s->h_stage_bytes = need;
                }
            }
            auto *stage = static_cast<unsigned char *>(s->h_stage);
            for (int k = 0; k < nres; ++k) {
                if (!s->ev_arr[k]) {
                    RK_HIP(hipEventCreateWithFlags(&s->ev_arr[k], hipEventDisableTiming));
                }
                RK_HIP(hipMemcpyAsync(stage + static_cast<size_t>(k) * arr, d_ptrs[k], arr, hipMemcpyDeviceToHost, nullptr));
                RK_HIP(hipEventRecord(s->ev_arr[k], nullptr));
            }
            static const int max_thr_o = [] {
                const char *e = std::getenv("RK_HOST_THREADS");
                const int v = e ? std::atoi(e) : 8;
                return v < 1 ? 1 : v;
            }();
            const size_t piece = size_t(2) << 20;
            for (int k = 0; k < nres; ++k) {
                RK_HIP(hipEventSynchronize(s->ev_arr[k]));
                const int n_items = static_cast<int>((arr + piece - 1) / piece);
                const int n_thr = std::max(1, std::min<int>({max_thr_o, n_items, static_cast<int>(std::thread::hardware_concurrency())}));
                delivery_pool::get().run(n_items, n_thr, [&](int item) {
                    const size_t off = static_cast<size_t>(item) * piece;
                    stream_copy(dst[k] + off, stage + static_cast<size_t>(k) * arr + off, std::min(piece, arr - off));
                });
            }
            return;
        }
        // No hipGraph capture on this path: the callers of the host entry point drive several devices from several host
        // threads (kwargs::split), and a capture in one thread makes legacy-stream operations of the others fail
        // (hipErrorStreamCaptureImplicit). The 35 us a replay saves vanish next to the transfer of the results.
        unsigned char *dst[4] = {};
        for (int k = 0; k < nres; ++k) {
            dst[k] = static_cast<unsigned char *>(out[k]) + (offset_output ? static_cast<size_t>(p_begin) * fsz : 0);
        }
        // Output arrays in pinned host memory (rk_host_alloc(), rakau_amd::pinned_allocator, hipHostRegister): the kernels
        // write the results where the caller wants them, nothing is staged or copied.
        {
            void *v_ptrs[4] = {};
            bool all = true;
            for (int k = 0; all && k < nres; ++k) {
                v_ptrs[k] = device_view_of_host_range(dst[k], count * fsz);
                all = v_ptrs[k] != nullptr;
            }
            if (all) {
                // Repeated calls are replayed from a hipGraph like device-output calls (round 5). The four class kernels of a large
                // call then sit in ONE queue and share the device as the replayed device-output step does; launched directly on four
                // streams, the R <= 2 kernels (8 waves per SIMD, short waves) take most of the slots first and end at 1.3 ms of a
                // 2.2 ms step, which leaves the R = 3 / 4 kernels to run among themselves at 6 waves per SIMD and the R = 3 kernel
                // alone for the last 0.2 ms (tools/seam_timeline.py): 4M 2.303 -> 2.234 ms per call, 1737 -> 1790 Mparticles/s
                // (tools/archive/jobs_r05/r05_job24.sh; round 4 measured the replay 0.01 ms SLOWER: the kernels were 4 % slower then and
                // better balanced at 7/7/6/5 waves per SIMD).
                constexpr bool host_graph = true;
                s->want_done_event = true;
                try {
                    if (s->fp == RK_F32) {
                        run_impl<float>(*s, q, p_begin, p_end, v_ptrs, mac_value, G, eps2, 0, nullptr, host_graph);
                    } else {
                        run_impl<double>(*s, q, p_begin, p_end, v_ptrs, mac_value, G, eps2, 0, nullptr, host_graph);
                    }
                } catch (...) {
                    s->want_done_event = false;
                    throw;
                }
                s->want_done_event = false;
                RK_HIP(hipEventSynchronize(s->ev1));
                return;
            }
        }
        if (need < (size_t(1) << 20)) {
            // Small results: device scratch + one copy per array (the only path that needs the scratch).
            if (s->d_out_bytes < need) {
                if (s->d_out) {
                    RK_HIP(hipDeviceSynchronize());
                    rk::pool_free(s->d_out);
                    s->d_out = nullptr;
                    s->d_out_bytes = 0;
                }
                s->d_out = rk::pool_alloc(need);
                s->d_out_bytes = need;
            }
            void *d_ptrs[4] = {};
            for (int k = 0; k < nres; ++k) {
                d_ptrs[k] = static_cast<unsigned char *>(s->d_out) + static_cast<size_t>(k) * count * fsz;
            }
            if (s->fp == RK_F32) {
                run_impl<float>(*s, q, p_begin, p_end, d_ptrs, mac_value, G, eps2, 0, nullptr, false);
            } else {
                run_impl<double>(*s, q, p_begin, p_end, d_ptrs, mac_value, G, eps2, 0, nullptr, false);
            }
            for (int k = 0; k < nres; ++k) {
                RK_HIP(hipMemcpy(dst[k], d_ptrs[k], count * fsz, hipMemcpyDeviceToHost));
            }
            return;
        }
        // Optional (RK_HOST_REGISTER=1; OFF by default): register the caller's pageable arrays for the duration of this blocking
        // call, let the kernels write into them, unregister. Measured at 4M fp32 (48 MB): 2.27 ms per call into arrays used
        // before against 2.86 through the staging buffer. It is NOT safe in a process where anything else pins host memory
        // that shares pages with the arrays: the HIP runtime keeps a cache of the ranges it pinned for pageable hipMemcpy
        // calls (sources read-only), a registration that overlaps one of those gets its mapping, and the traversal dies of
        // "Memory access fault by GPU ... Write access to a read-only page" or of a fault when the cached pin is evicted
        // (tools/stress_host_register.py: every run with the registration on aborts within seconds, none without;
        // profiles/r03/host_register_overlap.txt). Only for applications that never hand pageable memory to HIP copies.
        // With it on: always for arrays the previous call on this state wrote, otherwise up to 256 MB.
        {
            static const bool reg = [] {
                const char *e = std::getenv("RK_HOST_REGISTER");
                return e && std::atoi(e) != 0;
            }();
            constexpr size_t reg_max = size_t(256) << 20;
            bool seen = true;
            for (int k = 0; k < nres; ++k) {
                seen = seen && s->last_host_out[k] == dst[k];
            }
            seen = seen && s->last_host_bytes == count * fsz;
            for (int k = 0; k < 4; ++k) {
                s->last_host_out[k] = k < nres ? dst[k] : nullptr;
            }
            s->last_host_bytes = count * fsz;
            if (reg && (seen || need <= reg_max)) {
                // Arrays that share a page (slices of one allocation) are registered as one range.
                struct range {
                    unsigned char *b, *e;
                    void *dev;
                };
                std::vector<range> ranges;
                {
                    std::vector<std::pair<unsigned char *, unsigned char *>> v;
                    for (int k = 0; k < nres; ++k) {
                        v.emplace_back(dst[k], dst[k] + count * fsz);
                    }
                    std::sort(v.begin(), v.end());
                    for (const auto &r : v) {
                        if (!ranges.empty() && r.first <= ranges.back().e + 4096) {
                            ranges.back().e = std::max(ranges.back().e, r.second);
                        } else {
                            ranges.push_back(range{r.first, r.second, nullptr});
                        }
                    }
                }
                size_t done = 0;
                bool ok = true;
                for (; ok && done < ranges.size(); ++done) {
                    auto &r = ranges[done];
                    ok = hipHostRegister(r.b, static_cast<size_t>(r.e - r.b), hipHostRegisterDefault) == hipSuccess;
                    if (ok && hipHostGetDevicePointer(&r.dev, r.b, 0) != hipSuccess) {
                        (void)hipHostUnregister(r.b);
                        ok = false;
                    }
                    if (!ok) {
                        (void)hipGetLastError(); // e.g. part of the range is registered already: the staging path serves the call
                        break;
                    }
                }
                struct unreg {
                    std::vector<range> &r;
                    size_t n;
                    ~unreg()
                    {
                        for (size_t k = 0; k < n; ++k) {
                            (void)hipHostUnregister(r[k].b);
                        }
                    }
                } guard_{ranges, ok ? ranges.size() : done};
                if (ok) {
                    void *v_ptrs[4] = {};
                    for (int k = 0; k < nres; ++k) {
                        for (const auto &r : ranges) {
                            if (dst[k] >= r.b && dst[k] < r.e) {
                                v_ptrs[k] = static_cast<unsigned char *>(r.dev) + (dst[k] - r.b);
                            }
                        }
                    }
                    if (s->fp == RK_F32) {
                        run_impl<float>(*s, q, p_begin, p_end, v_ptrs, mac_value, G, eps2, 0, nullptr, false);
                    } else {
                        run_impl<double>(*s, q, p_begin, p_end, v_ptrs, mac_value, G, eps2, 0, nullptr, false);
                    }
                    RK_HIP(hipEventSynchronize(s->ev1));
                    return;
                }
            }
        }
        // Large results: the kernels write straight into a pinned staging buffer (host memory mapped into the device's
        // address space: posted PCIe writes that trickle out while the traversal computes -- 48 MB during a 2.3 ms kernel
        // at 4M particles), so nothing is left to transfer when the kernels end; host threads then move the staging
        // buffer into the caller's pageable arrays. Measured at 4M fp32 (ms per call, 2.3 ms of it the kernels): a plain
        // hipMemcpy into pageable memory afterwards 6.4, device -> pinned chunks + threaded delivery after the kernels
        // 4.2, the same with the traversal launched in 4 Morton chunks so that the copies overlap it 4.3 (four small
        // launches lose what the overlap wins), this 3.1 (profiles/r02/host_output_path.txt).
        if (s->h_stage_bytes < need) {
            if (s->h_stage) {
                RK_HIP(hipDeviceSynchronize());
                stage_give(phys(s->device), s->h_stage, s->h_stage_bytes);
                s->h_stage = nullptr;
                s->h_stage_bytes = 0;
            }
            size_t got = 0;
            s->h_stage = stage_take(phys(s->device), need, got);
            if (s->h_stage) {
                s->h_stage_bytes = got;
            } else {
                RK_HIP(hipHostMalloc(&s->h_stage, need, hipHostMallocDefault));
                s->h_stage_bytes = need;
            }
        }
        void *h_ptrs[4] = {};
        for (int k = 0; k < nres; ++k) {
            h_ptrs[k] = static_cast<unsigned char *>(s->h_stage) + static_cast<size_t>(k) * count * fsz;
        }
        const auto *stage = static_cast<const unsigned char *>(s->h_stage);
        static const int max_thr = [] {
            const char *e = std::getenv("RK_HOST_THREADS"); // delivery threads (default 8; memory-bound beyond that)
            const int v = e ? std::atoi(e) : 8;
            return v < 1 ? 1 : v;
        }();
        const size_t arr = count * fsz;
        // Elements [eb, ee) of every staged array -> the caller's arrays, in 2 MB pieces on the pool threads, with streaming
        // stores (no read-for-ownership of destination lines that are overwritten whole).
        auto deliver = [&](size_t eb, size_t ee) {
            const size_t piece = size_t(2) << 20, span = (ee - eb) * fsz;
            const size_t per = (span + piece - 1) / piece;
            const int n_items = static_cast<int>(per * static_cast<size_t>(nres));
            if (n_items <= 0) {
                return;
            }
            const int n_thr = std::max(1, std::min<int>({max_thr, n_items, static_cast<int>(std::thread::hardware_concurrency())}));
            delivery_pool::get().run(n_items, n_thr, [&](int item) {
                const size_t k = static_cast<size_t>(item) / per, off = eb * fsz + (static_cast<size_t>(item) % per) * piece;
                const size_t nb = std::min(piece, ee * fsz - off);
                stream_copy(dst[k] + off, stage + k * arr + off, nb);
            });
        };
        // (replayed from a hipGraph when the call recurs, like the pinned-output call above)
        constexpr bool staged_graph = true;
        auto run = [&](int64_t b, int64_t e, void *const *ptrs) {
            s->want_done_event = true;
            try {
                if (s->fp == RK_F32) {
                    run_impl<float>(*s, q, b, e, ptrs, mac_value, G, eps2, 0, nullptr, staged_graph);
                } else {
                    run_impl<double>(*s, q, b, e, ptrs, mac_value, G, eps2, 0, nullptr, staged_graph);
                }
            } catch (...) {
                s->want_done_event = false;
                throw;
            }
            s->want_done_event = false;
        };
        // Two parts (round 4): the first 85 % of the range is traversed first and DELIVERED by the host threads
        // while the second part is traversed; only the second part's delivery is left when the kernels end. The cut is a
        // critical-node boundary, the two parts are ordinary sub-range calls (their union equals the one-part result bit for
        // bit), and two launches cost about 0.1 ms more than one at 4M, where the delivery of 85 % of the results costs
        // 0.3-0.5: 2.65 -> 2.53-2.54 ms per call (fractions 0.7 / 0.8 / 0.85 / 0.9: 2.68 / 2.56 / 2.54 / 2.53-2.80), 2M 1.59 -> 1.39,
        // 4M accelerations + potentials 3.32 -> 2.79 (tools/archive/jobs_r04/r04_job39.sh). Results below
        // 16 MB: one part, delivered at the end.
        constexpr double split_frac = 0.85;
        int64_t cut = p_begin;
        if (split_frac > 0.0 && need >= (size_t(16) << 20)) {
            ensure_mirrors(*s);
            const auto target = p_begin + static_cast<int64_t>(split_frac * static_cast<double>(count));
            const auto it = std::lower_bound(s->crit_begin.begin(), s->crit_begin.end(), target);
            cut = it == s->crit_begin.end() ? p_end : *it;
        }
        if (cut > p_begin && cut < p_end) {
            if (!s->ev_mid) {
                RK_HIP(hipEventCreateWithFlags(&s->ev_mid, hipEventDisableTiming));
            }
            run(p_begin, cut, h_ptrs);
            RK_HIP(hipEventRecord(s->ev_mid, nullptr));
            void *h2[4] = {};
            for (int k = 0; k < nres; ++k) {
                h2[k] = static_cast<unsigned char *>(h_ptrs[k]) + static_cast<size_t>(cut - p_begin) * fsz;
            }
            s->keep_ev0 = true;
            try {
                run(cut, p_end, h2);
            } catch (...) {
                s->keep_ev0 = false;
                throw;
            }
            s->keep_ev0 = false;
            RK_HIP(hipEventSynchronize(s->ev_mid));
            deliver(0, static_cast<size_t>(cut - p_begin));
            RK_HIP(hipEventSynchronize(s->ev1));
            deliver(static_cast<size_t>(cut - p_begin), count);
        } else {
            run(p_begin, p_end, h_ptrs);
            RK_HIP(hipEventSynchronize(s->ev1));
            deliver(0, count);
        }
    });
}

} // extern "C"
