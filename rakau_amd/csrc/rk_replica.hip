// Replication of a state: export / import of its buffers, clones on other devices of the process (peer copies over xGMI fanning
// out as a doubling tree), and -- one process per GPU -- the broadcast over RCCL, bound at run time.
#include "rk_state_internal.hpp"

// ---- one process per GPU: the replicate step over RCCL (the collectives library is bound at run time) ----
namespace
{
struct rccl_api {
    using result_t = int;
    struct unique_id {
        char internal[128];
    };
    result_t (*get_unique_id)(unique_id *) = nullptr;
    result_t (*comm_init_rank)(void **, int, unique_id, int) = nullptr;
    result_t (*comm_destroy)(void *) = nullptr;
    result_t (*broadcast)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    result_t (*group_start)() = nullptr;
    result_t (*group_end)() = nullptr;
    const char *(*error_string)(result_t) = nullptr;
    bool ok = false;
    std::string why;
};

const rccl_api &rccl()
{
    static const rccl_api api = [] {
        rccl_api a;
        // The copy already mapped into the process (PyTorch-ROCm brings its own under the same soname), else the system's.
        void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
        if (!h) {
            h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
        }
        if (!h) {
            h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
        }
        if (!h) {
            a.why = std::string("cannot load librccl.so.1: ") + dlerror();
            return a;
        }
        auto sym = [&](const char *name) {
            void *p = dlsym(h, name);
            if (!p && a.why.empty()) {
                a.why = std::string("librccl.so.1 lacks ") + name;
            }
            return p;
        };
        a.get_unique_id = reinterpret_cast<decltype(a.get_unique_id)>(sym("ncclGetUniqueId"));
        a.comm_init_rank = reinterpret_cast<decltype(a.comm_init_rank)>(sym("ncclCommInitRank"));
        a.comm_destroy = reinterpret_cast<decltype(a.comm_destroy)>(sym("ncclCommDestroy"));
        a.broadcast = reinterpret_cast<decltype(a.broadcast)>(sym("ncclBroadcast"));
        a.group_start = reinterpret_cast<decltype(a.group_start)>(sym("ncclGroupStart"));
        a.group_end = reinterpret_cast<decltype(a.group_end)>(sym("ncclGroupEnd"));
        a.error_string = reinterpret_cast<decltype(a.error_string)>(sym("ncclGetErrorString"));
        a.ok = a.why.empty();
        return a;
    }();
    if (!api.ok) {
        throw rk::error(RK_ERUNTIME, "RCCL is not available: " + api.why);
    }
    return api;
}

void rccl_check(int r, const char *what)
{
    if (r != 0) {
        throw rk::error(RK_ERUNTIME, std::string("RCCL call failed: ") + what + ": " + rccl().error_string(r));
    }
}
} // namespace

extern "C" {

int rk_state_export(const rk_state *s, int *count, void **ptrs, int64_t *bytes, int64_t meta[RK_META_WORDS])
{
    return guard([&] {
        if (!s || !count || !ptrs || !bytes || !meta) {
            throw rk::error(RK_EINVAL, "null argument");
        }
        {
            device_guard dg(s->device);
            ensure_mirrors(*const_cast<rk_state *>(s)); // the class-list buffer must be complete before it travels
        }
        // The RK_NBUF traversal buffers, then the permutation (uint32 per particle; 0 bytes if the state has none).
        *count = RK_NBUF + 1;
        for (int i = 0; i < RK_NBUF; ++i) {
            ptrs[i] = s->buf[i];
            bytes[i] = s->buf_bytes[i];
        }
        ptrs[RK_NBUF] = s->bld_perm;
        bytes[RK_NBUF] = s->bld_perm ? s->nparts * static_cast<int64_t>(sizeof(uint32_t)) : 0;
        std::fill(meta, meta + RK_META_WORDS, int64_t(0));
        meta[0] = state_layout_tag;
        meta[1] = s->fp;
        meta[2] = s->mac;
        meta[3] = s->nparts;
        meta[4] = s->tree_size;
        meta[5] = s->n_crit;
        meta[6] = static_cast<int64_t>(s->ncrit);
        meta[7] = s->n_internal;
        meta[24] = s->ndim;
        for (int i = 0; i <= RK_NBUF; ++i) {
            meta[8 + i] = bytes[i];
        }
        std::memcpy(&meta[25], &s->box_size, sizeof(double));
        meta[26] = s->box_deduced;
        meta[27] = static_cast<int64_t>(s->max_leaf_n);
    });
}

// A replica is made in three steps shared by rk_state_import, rk_state_clone, rk_state_clone_all and rk_state_broadcast:
// replica_shell() checks the meta block and allocates the state with empty buffers on `device`; the caller fills the
// buffers (device-to-device, peer copies, RCCL); replica_finish() derives the host mirrors from the critical-node buffer.
static state_ptr replica_shell(int device, int count, const int64_t *bytes, const int64_t meta[RK_META_WORDS])
{
    if (meta[0] != state_layout_tag || count != RK_NBUF + 1) {
        throw rk::error(RK_EINVAL, "unrecognised state layout");
    }
    check_common(static_cast<int>(meta[1]), static_cast<int>(meta[2]));
    check_ndim(static_cast<int>(meta[24]));
    check_device(device);
    // The meta block is trusted no further than the buffers it describes: every count must match the byte size
    // of its buffer, and the limits of rk_state_create apply.
    const int64_t nparts = meta[3], tree_size = meta[4], n_crit = meta[5], n_internal = meta[7];
    const int64_t fsz = meta[1] == RK_F32 ? 4 : 8;
    if (nparts < 0 || tree_size < 0 || n_crit < 0 || n_internal < 0 || n_crit > tree_size || n_internal > tree_size
        || static_cast<uint64_t>(nparts) >= 0xffffffffull || static_cast<uint64_t>(tree_size) >= rk::max_list_nodes
        || (nparts > 0 && (tree_size == 0 || n_crit == 0)) || meta[6] <= 0) {
        throw rk::error(RK_EINVAL, "inconsistent counts in the meta block of rk_state_import");
    }
    const int64_t rec_bytes = meta[1] == RK_F32 ? int64_t(sizeof(rk::node_rec<float>)) : int64_t(sizeof(rk::node_rec<double>));
    const int64_t expect[RK_NBUF] = {nparts * 4 * fsz, tree_size * 4 * fsz, tree_size * 2 * fsz, tree_size * 16,
                                     n_crit * 16,      n_internal * 8 * 4,  -1 /* class lists: checked below */,
                                     tree_size * rec_bytes, n_crit * 2 * 4 * fsz};
    for (int i = 0; i < RK_NBUF; ++i) {
        if (bytes[i] != meta[8 + i] || (expect[i] >= 0 && bytes[i] != expect[i])) {
            throw rk::error(RK_EINVAL, "buffer " + std::to_string(i) + " of rk_state_import has " + std::to_string(bytes[i])
                                           + " bytes, which does not match the meta block");
        }
    }
    if (bytes[RK_BUF_CLASS] != 2 * n_crit * 4) {
        throw rk::error(RK_EINVAL, "class-list buffer size mismatch in rk_state_import");
    }
    if (bytes[RK_NBUF] != meta[8 + RK_NBUF] || (bytes[RK_NBUF] != 0 && bytes[RK_NBUF] != nparts * 4)) {
        throw rk::error(RK_EINVAL, "permutation buffer size mismatch in rk_state_import");
    }
    if (static_cast<size_t>(bytes[RK_BUF_CRIT]) != static_cast<size_t>(n_crit) * sizeof(uint4)) {
        throw rk::error(RK_EINVAL, "critical node buffer size mismatch in rk_state_import");
    }
    device_guard dg(device);
    state_ptr s(new rk_state);
    s->fp = static_cast<int>(meta[1]);
    s->mac = static_cast<int>(meta[2]);
    s->device = device;
    s->nparts = nparts;
    s->tree_size = tree_size;
    s->n_crit = n_crit;
    s->ncrit = static_cast<uint64_t>(meta[6]);
    s->n_internal = n_internal;
    s->ndim = static_cast<int>(meta[24]);
    std::memcpy(&s->box_size, &meta[25], sizeof(double));
    s->box_deduced = meta[26] != 0;
    s->max_leaf_n = static_cast<uint64_t>(meta[27]);
    for (int i = 0; i < RK_NBUF; ++i) {
        s->buf_bytes[i] = bytes[i];
        if (bytes[i]) {
            s->buf[i] = rk::pool_alloc(static_cast<size_t>(bytes[i]));
        }
    }
    if (bytes[RK_NBUF]) {
        // With the permutation a replica serves RK_OUT_ORDERED like the state it was exported from.
        s->bld_perm = rk::pool_alloc(static_cast<size_t>(bytes[RK_NBUF]));
    }
    return s;
}

// Destination of exported buffer i in a replica (nullptr for an empty one).
static void *replica_buffer(rk_state &s, int i)
{
    return i < RK_NBUF ? s.buf[i] : s.bld_perm;
}

static void replica_finish(rk_state &s)
{
    device_guard dg(s.device);
    std::vector<uint4> crit(static_cast<size_t>(s.n_crit));
    if (!crit.empty()) {
        RK_HIP(hipMemcpy(crit.data(), s.buf[RK_BUF_CRIT], crit.size() * sizeof(uint4), hipMemcpyDeviceToHost));
    }
    build_host_mirrors(s, crit);
    ensure_call_resources_any(s);
    rk::replica_first_order(s); // (small trees: the first call of a replica runs in heavy-first order like its source's)
}

// rk_state_import (buffers already on `device`: src_device < 0) and rk_state_clone (buffers on src_device).
static int import_impl(rk_state **out, int device, int count, void *const *ptrs, const int64_t *bytes,
                       const int64_t meta[RK_META_WORDS], int src_device)
{
    return guard([&] {
        if (!out || !ptrs || !bytes || !meta) {
            throw rk::error(RK_EINVAL, "null argument");
        }
        *out = nullptr;
        state_ptr s = replica_shell(device, count, bytes, meta);
        device_guard dg(device);
        for (int i = 0; i <= RK_NBUF; ++i) {
            if (!bytes[i]) {
                continue;
            }
            if (!ptrs[i]) {
                throw rk::error(RK_EINVAL, "null buffer in rk_state_import");
            }
            // Device-to-device on one GPU, or a peer copy over xGMI between two GPUs of this process.
            if (src_device < 0 || phys(src_device) == phys(device)) {
                RK_HIP(hipMemcpy(replica_buffer(*s, i), ptrs[i], static_cast<size_t>(bytes[i]), hipMemcpyDeviceToDevice));
            } else {
                RK_HIP(hipMemcpyPeer(replica_buffer(*s, i), phys(device), ptrs[i], phys(src_device), static_cast<size_t>(bytes[i])));
            }
        }
        replica_finish(*s);
        *out = s.release();
    });
}

int rk_state_import(rk_state **out, int device, int count, void *const *ptrs, const int64_t *bytes,
                    const int64_t meta[RK_META_WORDS])
{
    return import_impl(out, device, count, ptrs, bytes, meta, -1);
}

// Export of a state whose device work has completed (what every replication entry starts from).
static int export_settled(const rk_state *src, int *count, void **ptrs, int64_t *bytes, int64_t *meta)
{
    const int rc = rk_state_export(src, count, ptrs, bytes, meta);
    if (rc != RK_OK) {
        return rc;
    }
    // Everything the source has enqueued (its build, an upload) must have landed before its buffers are read.
    return guard([&] {
        device_guard dg(src->device);
        RK_HIP(hipDeviceSynchronize());
    });
}

int rk_state_clone(rk_state **out, const rk_state *src, int device)
{
    if (!out || !src) {
        return guard([] { throw rk::error(RK_EINVAL, "null argument"); });
    }
    int count = 0;
    void *ptrs[RK_MAX_BUFFERS] = {};
    int64_t bytes[RK_MAX_BUFFERS] = {}, meta[RK_META_WORDS] = {};
    const int rc = export_settled(src, &count, ptrs, bytes, meta);
    if (rc != RK_OK) {
        return rc;
    }
    return import_impl(out, device, count, ptrs, bytes, meta, src->device);
}

// Replicas of `src` on n devices at once. The copies fan out as a doubling tree: in every round each device that holds the
// state sends it to one that does not, all transfers of a round in flight together (asynchronous peer copies on one stream
// per destination), so that n replicas take ceil(log2(n + 1)) rounds over as many xGMI links as there are senders instead
// of n copies leaving the source one after the other (the reference uploads tree and particles to every device from the
// host on every call, on one stream per device: src/rakau_cuda.cu:492-527).
int rk_state_clone_all(rk_state **outs, const rk_state *src, const int *devices, int n)
{
    if (!outs || !src || !devices || n < 0) {
        return guard([] { throw rk::error(RK_EINVAL, "null argument"); });
    }
    for (int i = 0; i < n; ++i) {
        outs[i] = nullptr;
    }
    int count = 0;
    void *ptrs[RK_MAX_BUFFERS] = {};
    int64_t bytes[RK_MAX_BUFFERS] = {}, meta[RK_META_WORDS] = {};
    const int rc = export_settled(src, &count, ptrs, bytes, meta);
    if (rc != RK_OK) {
        return rc;
    }
    std::vector<state_ptr> made(static_cast<size_t>(n));
    std::vector<hipStream_t> streams(static_cast<size_t>(n), nullptr);
    const int rc2 = guard([&] {
        for (int i = 0; i < n; ++i) {
            made[static_cast<size_t>(i)] = replica_shell(devices[i], count, bytes, meta);
            device_guard dg(devices[i]);
            RK_HIP(hipStreamCreateWithFlags(&streams[static_cast<size_t>(i)], hipStreamNonBlocking));
        }
        // holders: -1 = the source, i >= 0 = made[i] (complete).
        std::vector<int> holders{-1};
        int next = 0;
        static const bool trace = [] {
            const char *e = std::getenv("RK_CLONE_TRACE"); // prints the rounds (which device sends to which)
            return e && std::atoi(e) != 0;
        }();
        int round = 0;
        while (next < n) {
            const int first = next;
            const size_t n_holders = holders.size();
            for (size_t h = 0; h < n_holders && next < n; ++h, ++next) {
                const rk_state &from = holders[h] < 0 ? *src : *made[static_cast<size_t>(holders[h])];
                rk_state &to = *made[static_cast<size_t>(next)];
                device_guard dg(to.device);
                for (int b = 0; b <= RK_NBUF; ++b) {
                    if (!bytes[b]) {
                        continue;
                    }
                    const void *sp = holders[h] < 0 ? ptrs[b] : replica_buffer(const_cast<rk_state &>(from), b);
                    if (phys(from.device) == phys(to.device)) {
                        RK_HIP(hipMemcpyAsync(replica_buffer(to, b), sp, static_cast<size_t>(bytes[b]), hipMemcpyDeviceToDevice,
                                              streams[static_cast<size_t>(next)]));
                    } else {
                        RK_HIP(hipMemcpyPeerAsync(replica_buffer(to, b), phys(to.device), sp, phys(from.device),
                                                  static_cast<size_t>(bytes[b]), streams[static_cast<size_t>(next)]));
                    }
                }
                if (trace) {
                    std::fprintf(stderr, "rk_state_clone_all round %d: device %d -> device %d\n", round, from.device, to.device);
                }
            }
            for (int i = first; i < next; ++i) {
                device_guard dg(made[static_cast<size_t>(i)]->device);
                RK_HIP(hipStreamSynchronize(streams[static_cast<size_t>(i)]));
                holders.push_back(i);
            }
            ++round;
        }
        for (int i = 0; i < n; ++i) {
            replica_finish(*made[static_cast<size_t>(i)]);
        }
    });
    for (int i = 0; i < n; ++i) {
        if (streams[static_cast<size_t>(i)]) {
            int prev = 0;
            (void)hipGetDevice(&prev);
            (void)hipSetDevice(phys(devices[i]));
            (void)hipStreamDestroy(streams[static_cast<size_t>(i)]);
            (void)hipSetDevice(prev);
        }
    }
    if (rc2 != RK_OK) {
        return rc2;
    }
    for (int i = 0; i < n; ++i) {
        outs[i] = made[static_cast<size_t>(i)].release();
    }
    return RK_OK;
}

// ---- one process per GPU: the replicate step over RCCL (rccl(): the collectives library, bound at run time) ----

int rk_comm_unique_id(char id[RK_COMM_ID_BYTES])
{
    return guard([&] {
        if (!id) {
            throw rk::error(RK_EINVAL, "null argument");
        }
        rccl_api::unique_id u{};
        rccl_check(rccl().get_unique_id(&u), "ncclGetUniqueId");
        std::memcpy(id, u.internal, sizeof(u.internal));
    });
}

int rk_comm_init(void **comm, int n_ranks, const char id[RK_COMM_ID_BYTES], int rank, int device)
{
    return guard([&] {
        if (!comm || !id) {
            throw rk::error(RK_EINVAL, "null argument");
        }
        *comm = nullptr;
        check_device(device);
        device_guard dg(device);
        rccl_api::unique_id u{};
        std::memcpy(u.internal, id, sizeof(u.internal));
        rccl_check(rccl().comm_init_rank(comm, n_ranks, u, rank), "ncclCommInitRank");
    });
}

int rk_comm_destroy(void *comm)
{
    return guard([&] {
        if (comm) {
            rccl_check(rccl().comm_destroy(comm), "ncclCommDestroy");
        }
    });
}

// The replicate step of the one-process-per-GPU model: the state of rank `root` becomes a state on every rank's device.
// ncclBroadcast of the meta block, then of every exported buffer inside one group (RCCL pipelines them over the xGMI
// ring / tree it built for the communicator). No data-path collective is needed afterwards: every rank traverses its
// own Morton range of targets and nothing is reduced (north star).
int rk_state_broadcast(rk_state **state, int root, int rank, int device, void *comm, void *stream_)
{
    return guard([&] {
        if (!state || !comm || (rank == root && !*state)) {
            throw rk::error(RK_EINVAL, "null argument");
        }
        auto stream = static_cast<hipStream_t>(stream_);
        const rccl_api &api = rccl();
        check_device(device);
        device_guard dg(device);
        int count = RK_NBUF + 1;
        void *ptrs[RK_MAX_BUFFERS] = {};
        int64_t bytes[RK_MAX_BUFFERS] = {}, meta[RK_META_WORDS] = {};
        if (rank == root) {
            if ((*state)->device != device) {
                throw rk::error(RK_EINVAL, "the root's state does not live on the device of this rank");
            }
            const int rc = export_settled(*state, &count, ptrs, bytes, meta);
            if (rc != RK_OK) {
                throw rk::error(rc, rk_last_error());
            }
        }
        // Meta block first (through a small device buffer: RCCL moves device memory).
        struct dev_block {
            void *p = nullptr;
            ~dev_block()
            {
                rk::pool_free(p);
            }
        } dmeta;
        dmeta.p = rk::pool_alloc(sizeof(meta));
        if (rank == root) {
            RK_HIP(hipMemcpyAsync(dmeta.p, meta, sizeof(meta), hipMemcpyHostToDevice, stream));
        }
        rccl_check(api.broadcast(dmeta.p, dmeta.p, sizeof(meta), /* ncclUint8 */ 1, root, comm, stream), "ncclBroadcast(meta)");
        RK_HIP(hipMemcpyAsync(meta, dmeta.p, sizeof(meta), hipMemcpyDeviceToHost, stream));
        RK_HIP(hipStreamSynchronize(stream));
        state_ptr made;
        if (rank != root) {
            for (int i = 0; i <= RK_NBUF; ++i) {
                bytes[i] = meta[8 + i];
            }
            made = replica_shell(device, count, bytes, meta);
            for (int i = 0; i <= RK_NBUF; ++i) {
                ptrs[i] = replica_buffer(*made, i);
            }
        }
        rccl_check(api.group_start(), "ncclGroupStart");
        for (int i = 0; i <= RK_NBUF; ++i) {
            if (bytes[i]) {
                rccl_check(api.broadcast(ptrs[i], ptrs[i], static_cast<size_t>(bytes[i]), 1, root, comm, stream), "ncclBroadcast");
            }
        }
        rccl_check(api.group_end(), "ncclGroupEnd");
        RK_HIP(hipStreamSynchronize(stream));
        if (rank != root) {
            replica_finish(*made);
            *state = made.release();
        }
    });
}

} // extern "C"
