// Peer-mapped memory regions: the one-sided data plane of the asynchronous parameter-server mode on one node.
//
// Every rank (one process per GPU) allocates ONE region in its HBM, publishes its hipIpc handle and maps the regions
// of all other ranks.  A worker then *puts* its gradient shard straight into the owner's inbox slot, and an owner
// puts fresh parameters straight into the worker's landing zone: one device-to-device copy over xGMI per message,
// issued on a stream of the sending process, with no rendezvous between the two processes - which is what an
// asynchronous parameter server needs (the reference moves the same tensors through gRPC RecvTensor calls,
// /root/reference/example.py:128-129,136-139).
//
// Ordering and visibility use only what the HIP runtime guarantees for copies: a put is complete and visible to
// every device once the stream it was issued on has been synchronised (dtf_mbox_sync); the sender then tells the
// receiver through the control plane (TCP), and the receiver launches its kernels after that message.  No flags in
// device memory, no polling kernels.
#include "common.h"

#include <cstring>

namespace {

constexpr int MBOX_MAX_RANKS = 64;

struct Mailbox {
    int rank = 0, world = 0;
    size_t bytes = 0;                      // this rank's region
    size_t size[MBOX_MAX_RANKS] = {};      // every rank's region (equal unless dtf_mbox_set_peer_bytes says otherwise)
    void* local = nullptr;
    void* peer[MBOX_MAX_RANKS] = {};       // peer[rank] == local
    bool opened[MBOX_MAX_RANKS] = {};
};

struct PeerTable {
    char* ptr[MBOX_MAX_RANKS];
};

// One launch moves the shards of ALL peers: blockIdx.y is the peer, so the loads (gather) or stores (scatter) of a
// launch are in flight on every xGMI link of this GPU at once - the links are point-to-point, and a chain of
// per-peer copies on one stream would use them one at a time.  16 bytes per lane, grid-stride inside a shard.
// GATHER: dst[j*n .. (j+1)*n) = peer j's region at `off`;  SCATTER: peer j's region at `off` = src[j*n .. (j+1)*n).
template <bool GATHER>
__global__ __launch_bounds__(256) void peer_shards_kernel(PeerTable peers, long long off_bytes, char* local, long long n16,
                                                          long long shard_bytes) {
    const int j = blockIdx.y;
    f32x4* remote = reinterpret_cast<f32x4*>(peers.ptr[j] + off_bytes);
    f32x4* mine = reinterpret_cast<f32x4*>(local + (long long)j * shard_bytes);
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) {
        if (GATHER) mine[i] = __builtin_nontemporal_load(remote + i);
        else __builtin_nontemporal_store(mine[i], remote + i);
    }
    if (!GATHER) __threadfence_system();        // the stores are on their way to the peers before the launch ends
}

int launch_peer_shards(Mailbox* m, bool gather, long long off, void* local, long long shard_bytes, hipStream_t stream) {
    if (shard_bytes == 0) return 0;
    if (off < 0 || shard_bytes < 0 || (shard_bytes & 15) || (off & 15) || (reinterpret_cast<uintptr_t>(local) & 15)) return -3;
    PeerTable t;
    for (int r = 0; r < m->world; ++r) {
        if (!m->peer[r]) return -2;
        if ((size_t)off + (size_t)shard_bytes > m->size[r]) return -3;
        t.ptr[r] = static_cast<char*>(m->peer[r]);
    }
    const long long n16 = shard_bytes / 16;
    long long bx = (n16 + 255) / 256;
    const long long cap = 2048 / m->world > 1 ? 2048 / m->world : 1;     // ~8 workgroups per CU over all peers
    if (bx > cap) bx = cap;
    if (bx < 1) bx = 1;
    const dim3 grid((unsigned)bx, (unsigned)m->world);
    if (gather)
        hipLaunchKernelGGL(peer_shards_kernel<true>, grid, dim3(256), 0, stream, t, off, static_cast<char*>(local), n16, shard_bytes);
    else
        hipLaunchKernelGGL(peer_shards_kernel<false>, grid, dim3(256), 0, stream, t, off, static_cast<char*>(local), n16, shard_bytes);
    DTF_CHECK_LAUNCH();
    return 0;
}

// A list of ranges in ONE launch: range r connects `n16` 16-byte units at byte `local_off` of a local buffer with byte
// `remote_base + remote_off` of rank `peer`'s region.  What a worker of a cluster whose ps tasks own whole variables
// needs every step: the flat ranges of every owner's variables (round-robin placement interleaves them), pushed into
// the owners' inbox slots or pulled from their variable planes, all owners and all ranges at once instead of one
// runtime copy each.
struct RangeDesc {
    int peer, pad;
    long long remote_off, local_off, n16;
};

struct RangeTable {
    RangeDesc* dev = nullptr;
    int count = 0;
    long long max_n16 = 0;
    long long local_end = 0;                    // bytes of the local buffer the ranges reach
    long long remote_end[MBOX_MAX_RANKS] = {};  // bytes of each peer's region the ranges reach (before remote_base)
};

template <bool GATHER>
__global__ __launch_bounds__(256) void peer_ranges_kernel(PeerTable peers, const RangeDesc* __restrict__ ranges, char* local,
                                                          long long remote_base) {
    const RangeDesc r = ranges[blockIdx.y];
    f32x4* remote = reinterpret_cast<f32x4*>(peers.ptr[r.peer] + remote_base + r.remote_off);
    f32x4* mine = reinterpret_cast<f32x4*>(local + r.local_off);
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < r.n16; i += stride) {
        if (GATHER) mine[i] = __builtin_nontemporal_load(remote + i);
        else __builtin_nontemporal_store(mine[i], remote + i);
    }
    if (!GATHER) __threadfence_system();
}

}  // namespace

// Validates `count` ranges (host array of {int peer, int pad, long long remote_off, local_off, n16}; offsets in bytes,
// multiples of 16) against the regions' sizes and uploads them.  Null if any range is malformed.
DTF_API void* dtf_mbox_ranges_create(void* h, const void* host_descs, int count) {
    auto* m = static_cast<Mailbox*>(h);
    if (count < 1 || count > 65535 || !host_descs) return nullptr;
    const RangeDesc* d = static_cast<const RangeDesc*>(host_descs);
    auto* t = new RangeTable();
    t->count = count;
    for (int i = 0; i < count; ++i) {
        const RangeDesc& r = d[i];
        if (r.peer < 0 || r.peer >= m->world || !m->peer[r.peer] || r.n16 < 0 || r.remote_off < 0 || r.local_off < 0 ||
            (r.remote_off & 15) || (r.local_off & 15) || (size_t)(r.remote_off + r.n16 * 16) > m->size[r.peer]) {
            delete t;
            return nullptr;
        }
        if (r.n16 > t->max_n16) t->max_n16 = r.n16;
        if (r.local_off + r.n16 * 16 > t->local_end) t->local_end = r.local_off + r.n16 * 16;
        if (r.remote_off + r.n16 * 16 > t->remote_end[r.peer]) t->remote_end[r.peer] = r.remote_off + r.n16 * 16;
    }
    if (hipMalloc(&t->dev, sizeof(RangeDesc) * (size_t)count) != hipSuccess ||
        hipMemcpy(t->dev, d, sizeof(RangeDesc) * (size_t)count, hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipGetLastError();
        if (t->dev) (void)hipFree(t->dev);
        delete t;
        return nullptr;
    }
    return t;
}

DTF_API void dtf_mbox_ranges_destroy(void* table) {
    auto* t = static_cast<RangeTable*>(table);
    if (!t) return;
    if (t->dev) (void)hipFree(t->dev);
    delete t;
}

// Runs the table: gather != 0 copies remote -> local, else local -> remote.  `remote_base` (bytes, multiple of 16)
// is added to every range's remote offset (which inbox slot; 0 for the variable plane); `local` holds at least
// `local_bytes` bytes.
DTF_API int dtf_mbox_ranges_run(void* h, void* table, int gather, long long remote_base, void* local, long long local_bytes,
                                hipStream_t stream) {
    auto* m = static_cast<Mailbox*>(h);
    auto* t = static_cast<RangeTable*>(table);
    if (!t || !t->dev) return -2;
    if (remote_base < 0 || (remote_base & 15) || (reinterpret_cast<uintptr_t>(local) & 15) || t->local_end > local_bytes) return -3;
    PeerTable pt;
    for (int r = 0; r < m->world; ++r) {
        if (t->remote_end[r] && (!m->peer[r] || (size_t)(remote_base + t->remote_end[r]) > m->size[r])) return -3;
        pt.ptr[r] = static_cast<char*>(m->peer[r]);
    }
    if (t->max_n16 == 0) return 0;
    long long bx = (t->max_n16 + 255) / 256;
    const long long cap = 2048 / t->count > 1 ? 2048 / t->count : 1;
    if (bx > cap) bx = cap;
    const dim3 grid((unsigned)bx, (unsigned)t->count);
    if (gather)
        hipLaunchKernelGGL(peer_ranges_kernel<true>, grid, dim3(256), 0, stream, pt, t->dev, static_cast<char*>(local), remote_base);
    else
        hipLaunchKernelGGL(peer_ranges_kernel<false>, grid, dim3(256), 0, stream, pt, t->dev, static_cast<char*>(local), remote_base);
    DTF_CHECK_LAUNCH();
    return 0;
}

// dst[j * shard_bytes ...] = region of rank j at byte offset `src_off`, for every rank j, in ONE launch on `stream`.
DTF_API int dtf_mbox_gather(void* h, long long src_off, void* dst, long long shard_bytes, hipStream_t stream) {
    return launch_peer_shards(static_cast<Mailbox*>(h), true, src_off, dst, shard_bytes, stream);
}

// region of rank j at byte offset `dst_off` = src[j * shard_bytes ...], for every rank j, in ONE launch on `stream`.
DTF_API int dtf_mbox_scatter(void* h, long long dst_off, const void* src, long long shard_bytes, hipStream_t stream) {
    return launch_peer_shards(static_cast<Mailbox*>(h), false, dst_off, const_cast<void*>(src), shard_bytes, stream);
}

// Allocates and zeroes this rank's region on the current device.  Null on bad arguments or allocation failure.
DTF_API void* dtf_mbox_create(int rank, int world, long long bytes) {
    if (world < 1 || world > MBOX_MAX_RANKS || rank < 0 || rank >= world || bytes <= 0) return nullptr;
    auto* m = new Mailbox();
    m->rank = rank; m->world = world;
    m->bytes = ((size_t)bytes + 255) / 256 * 256;
    if (hipMalloc(&m->local, m->bytes) != hipSuccess) { (void)hipGetLastError(); delete m; return nullptr; }
    if (hipMemset(m->local, 0, m->bytes) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
        (void)hipFree(m->local);
        delete m;
        return nullptr;
    }
    m->peer[rank] = m->local;
    for (int r = 0; r < world; ++r) m->size[r] = m->bytes;
    return m;
}

// Regions need not be equally sized (parameter-server tasks hold slots and inboxes, their workers hold nothing): tells
// this rank how large rank `r`'s region is, for the bounds checks of put / get / gather / scatter.
DTF_API int dtf_mbox_set_peer_bytes(void* h, int r, long long bytes) {
    auto* m = static_cast<Mailbox*>(h);
    if (r < 0 || r >= m->world || bytes < 0) return -2;
    if (r == m->rank) return (size_t)bytes <= m->bytes ? 0 : -3;
    m->size[r] = (size_t)bytes;
    return 0;
}

DTF_API int dtf_mbox_handle_bytes() { return (int)sizeof(hipIpcMemHandle_t); }

DTF_API long long dtf_mbox_bytes(void* h) { return (long long)static_cast<Mailbox*>(h)->bytes; }

// Device pointer of the local region (the Python side wraps it as a tensor).
DTF_API void* dtf_mbox_local_ptr(void* h) { return static_cast<Mailbox*>(h)->local; }

// Rank r's region as this process's kernels address it (null until it is mapped): kernels that read a peer's
// variables or write into its inbox directly take their pointers from here.
DTF_API void* dtf_mbox_peer_ptr(void* h, int r) {
    auto* m = static_cast<Mailbox*>(h);
    return (r >= 0 && r < m->world) ? m->peer[r] : nullptr;
}

DTF_API int dtf_mbox_local_handle(void* h, void* out_handle) {
    auto* m = static_cast<Mailbox*>(h);
    hipIpcMemHandle_t ipc;
    hipError_t e = hipIpcGetMemHandle(&ipc, m->local);
    if (e != hipSuccess) return (int)e;
    std::memcpy(out_handle, &ipc, sizeof ipc);
    return 0;
}

// Maps the region of rank `r` from its handle (ranks on the same GPU map it just the same).
DTF_API int dtf_mbox_open_peer(void* h, int r, const void* handle) {
    auto* m = static_cast<Mailbox*>(h);
    if (r < 0 || r >= m->world) return -2;
    if (r == m->rank || m->peer[r]) return 0;
    hipIpcMemHandle_t ipc;
    std::memcpy(&ipc, handle, sizeof ipc);
    void* ptr = nullptr;
    hipError_t e = hipIpcOpenMemHandle(&ptr, ipc, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) return (int)e;
    m->peer[r] = ptr;
    m->opened[r] = true;
    return 0;
}

// Rank `r` has been restarted and allocated a new region: drop the mapping of the old one and map the new handle.
DTF_API int dtf_mbox_reopen_peer(void* h, int r, const void* handle) {
    auto* m = static_cast<Mailbox*>(h);
    if (r < 0 || r >= m->world || r == m->rank) return -2;
    if (m->opened[r]) {
        (void)hipDeviceSynchronize();
        (void)hipIpcCloseMemHandle(m->peer[r]);
        m->opened[r] = false;
    }
    m->peer[r] = nullptr;
    return dtf_mbox_open_peer(h, r, handle);
}

// Copies `bytes` from local device memory `src` into rank `peer`'s region at byte offset `dst_off`, on `stream`.
DTF_API int dtf_mbox_put(void* h, int peer, long long dst_off, const void* src, long long bytes, hipStream_t stream) {
    auto* m = static_cast<Mailbox*>(h);
    if (peer < 0 || peer >= m->world || !m->peer[peer]) return -2;
    if (bytes == 0) return 0;
    if (dst_off < 0 || bytes < 0 || (size_t)dst_off + (size_t)bytes > m->size[peer]) return -3;
    return (int)hipMemcpyAsync(static_cast<char*>(m->peer[peer]) + dst_off, src, (size_t)bytes, hipMemcpyDefault, stream);
}

// Copies `bytes` from rank `peer`'s region at byte offset `src_off` into local device memory `dst`, on `stream`.
DTF_API int dtf_mbox_get(void* h, int peer, long long src_off, void* dst, long long bytes, hipStream_t stream) {
    auto* m = static_cast<Mailbox*>(h);
    if (peer < 0 || peer >= m->world || !m->peer[peer]) return -2;
    if (bytes == 0) return 0;
    if (src_off < 0 || bytes < 0 || (size_t)src_off + (size_t)bytes > m->size[peer]) return -3;
    return (int)hipMemcpyAsync(dst, static_cast<const char*>(m->peer[peer]) + src_off, (size_t)bytes, hipMemcpyDefault, stream);
}

// Blocks the calling host thread until everything issued on `stream` so far - puts included - has completed.
DTF_API int dtf_mbox_sync(hipStream_t stream) { return (int)hipStreamSynchronize(stream); }

DTF_API void dtf_mbox_destroy(void* h) {
    auto* m = static_cast<Mailbox*>(h);
    (void)hipDeviceSynchronize();
    for (int r = 0; r < m->world; ++r)
        if (m->opened[r]) (void)hipIpcCloseMemHandle(m->peer[r]);
    if (m->local) (void)hipFree(m->local);
    delete m;
}
