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
    size_t bytes = 0;
    void* local = nullptr;
    void* peer[MBOX_MAX_RANKS] = {};       // peer[rank] == local
    bool opened[MBOX_MAX_RANKS] = {};
};

}  // namespace

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
    return m;
}

DTF_API int dtf_mbox_handle_bytes() { return (int)sizeof(hipIpcMemHandle_t); }

DTF_API long long dtf_mbox_bytes(void* h) { return (long long)static_cast<Mailbox*>(h)->bytes; }

// Device pointer of the local region (the Python side wraps it as a tensor).
DTF_API void* dtf_mbox_local_ptr(void* h) { return static_cast<Mailbox*>(h)->local; }

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

// Copies `bytes` from local device memory `src` into rank `peer`'s region at byte offset `dst_off`, on `stream`.
DTF_API int dtf_mbox_put(void* h, int peer, long long dst_off, const void* src, long long bytes, hipStream_t stream) {
    auto* m = static_cast<Mailbox*>(h);
    if (peer < 0 || peer >= m->world || !m->peer[peer]) return -2;
    if (bytes == 0) return 0;
    if (dst_off < 0 || bytes < 0 || (size_t)dst_off + (size_t)bytes > m->bytes) return -3;     // regions are equally sized
    return (int)hipMemcpyAsync(static_cast<char*>(m->peer[peer]) + dst_off, src, (size_t)bytes, hipMemcpyDefault, stream);
}

// Copies `bytes` from rank `peer`'s region at byte offset `src_off` into local device memory `dst`, on `stream`.
DTF_API int dtf_mbox_get(void* h, int peer, long long src_off, void* dst, long long bytes, hipStream_t stream) {
    auto* m = static_cast<Mailbox*>(h);
    if (peer < 0 || peer >= m->world || !m->peer[peer]) return -2;
    if (bytes == 0) return 0;
    if (src_off < 0 || bytes < 0 || (size_t)src_off + (size_t)bytes > m->bytes) return -3;
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
