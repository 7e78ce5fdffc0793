// One-shot all-reduce for small messages over peer pointers (xGMI loads), for the ranks of ONE node.
//
// A ring all-reduce of a 100-500 KB gradient is pure latency: 2(N-1) dependent hops plus the launch of the
// collective.  On a fully connected xGMI node every GPU can instead read every peer's buffer directly:
//
//   publish kernel   copy the rank's contribution into its own exchange buffer (double-buffered by call parity),
//                    make it visible system-wide, then store the call number into slot [rank] of every peer's
//                    flag array (one remote 4-byte store per peer);
//   reduce kernel    every workgroup waits until all N slots of the local flag array have reached the call number,
//                    then out[i] = sum over r = 0..N-1 of peer_r[i], read straight from the peers' memory in rank
//                    order - all 7 links of the GPU carry a share, one hop, and every rank adds in the same order, so
//                    the replicas stay bit-identical.
//
// The exchange buffers are the only memory shared between processes (hipIpc handles, opened once).  Double buffering
// makes one flag round per call sufficient: a rank can only overwrite buffer p of call c+2 after every peer has
// published call c+1, which each peer does after its own reduce of call c has completed in stream order.
// The wait is bounded (30 s by default, dtf_p2p_set_timeout_ms): ranks of a healthy job are routinely seconds apart
// (the chief writing a checkpoint, first-step allocations), so the bound only has to beat "a peer is gone".  A rank
// whose peers never show up sets an error word AND fills its output with NaN, so that a missed reduction can never
// pass for a gradient; the strategy polls the error word and ends the rank.
#include "common.h"

#include <cstring>

namespace {

constexpr int MAX_RANKS = 16;
constexpr long long DEFAULT_SPIN_TICKS = 30LL * 100 * 1000 * 1000;   // 30 s of the 100 MHz wall clock

struct Peers {
    float* data[MAX_RANKS];        // exchange buffers: [2][capacity] floats each
    unsigned* flags[MAX_RANKS];    // flag arrays: [MAX_RANKS] words each
};

struct Comm {
    int rank = 0, world = 0;
    long long capacity = 0;        // floats per half
    void* local = nullptr;         // one allocation: [2 * capacity floats | MAX_RANKS flag words | error word]
    size_t local_bytes = 0;
    void* opened[MAX_RANKS] = {};
    Peers peers = {};
    unsigned call = 0;
    bool fine_grained = false;
    long long spin_ticks = DEFAULT_SPIN_TICKS;
};

__device__ __forceinline__ void store_system(unsigned* p, unsigned v) {
    __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ unsigned load_system(const unsigned* p) {
    return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// grid-stride copy into the rank's own half, then one thread of the LAST workgroup to finish raises the flags
__global__ __launch_bounds__(256) void publish_kernel(const float* __restrict__ src, float* __restrict__ mine, long long n,
                                                      Peers peers, int rank, int world, unsigned call,
                                                      unsigned* __restrict__ done_counter) {
    const long long n4 = n >> 2;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x)
        reinterpret_cast<f32x4*>(mine)[i] = reinterpret_cast<const f32x4*>(src)[i];
    if (blockIdx.x == 0)
        for (long long i = (n4 << 2) + threadIdx.x; i < n; i += blockDim.x) mine[i] = src[i];
    __threadfence_system();            // this thread's stores are visible to the peers before the flag can be
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned t = atomicAdd(done_counter, 1u);
        if (t == gridDim.x - 1) {      // every workgroup has fenced its part
            *done_counter = 0u;
            for (int r = 0; r < world; ++r) store_system(peers.flags[r] + rank, call);
        }
    }
}

__global__ __launch_bounds__(256) void reduce_kernel(float* __restrict__ out, long long n, Peers peers, int rank, int world,
                                                     unsigned call, int half, long long capacity,
                                                     unsigned* __restrict__ error_word, long long spin_ticks) {
    __shared__ int ok;
    if (threadIdx.x == 0) {
        const unsigned* mine = peers.flags[rank];
        const long long t0 = (long long)wall_clock64();
        int good = 1;
        for (int r = 0; r < world; ++r) {
            // call numbers only grow; a wrapped counter is handled by the signed distance
            while ((int)(load_system(mine + r) - call) < 0) {
                if ((long long)wall_clock64() - t0 > spin_ticks) { good = 0; break; }
                __builtin_amdgcn_s_sleep(8);
            }
            if (!good) break;
        }
        if (!good) store_system(error_word, call ? call : 1u);
        ok = good;
    }
    __syncthreads();
    const long long n4 = n >> 2;
    if (!ok) {
        // a peer never published: poison the output instead of leaving the local gradient in place
        const float nan = __builtin_nanf("");
        const f32x4 nan4 = {nan, nan, nan, nan};
        for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x)
            reinterpret_cast<f32x4*>(out)[i] = nan4;
        if (blockIdx.x == 0)
            for (long long i = (n4 << 2) + threadIdx.x; i < n; i += blockDim.x) out[i] = nan;
        return;
    }
    const long long off = (long long)half * capacity;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int r = 0; r < world; ++r) {
            // peer memory: bypass this GPU's caches (the line may have been read during an earlier call)
            const f32x4* p = reinterpret_cast<const f32x4*>(peers.data[r] + off) + i;
            acc += __builtin_nontemporal_load(p);
        }
        reinterpret_cast<f32x4*>(out)[i] = acc;
    }
    if (blockIdx.x == 0) {
        for (long long i = (n4 << 2) + threadIdx.x; i < n; i += blockDim.x) {
            float acc = 0.f;
            for (int r = 0; r < world; ++r) acc += __builtin_nontemporal_load(peers.data[r] + off + i);
            out[i] = acc;
        }
    }
}

inline unsigned* flags_of(void* base, long long capacity) {
    return reinterpret_cast<unsigned*>(static_cast<float*>(base) + 2 * capacity);
}

}  // namespace

// Allocates this rank's exchange buffer.  Returns null when the arguments are unusable or the allocation fails.
DTF_API void* dtf_p2p_create(int rank, int world, long long capacity_floats) {
    if (world < 1 || world > MAX_RANKS || rank < 0 || rank >= world || capacity_floats <= 0) return nullptr;
    auto* c = new Comm();
    c->rank = rank; c->world = world;
    c->capacity = (capacity_floats + 3) / 4 * 4;
    c->local_bytes = (size_t)(2 * c->capacity) * sizeof(float) + (MAX_RANKS + 4) * sizeof(unsigned);
    // fine-grained memory: peers' flag stores and this GPU's polling loads are coherent without cache maintenance
    if (hipExtMallocWithFlags(&c->local, c->local_bytes, hipDeviceMallocFinegrained) == hipSuccess) {
        c->fine_grained = true;
    } else {
        (void)hipGetLastError();
        if (hipMalloc(&c->local, c->local_bytes) != hipSuccess) { delete c; return nullptr; }
    }
    if (hipMemset(c->local, 0, c->local_bytes) != hipSuccess) { (void)hipFree(c->local); delete c; return nullptr; }
    (void)hipDeviceSynchronize();
    c->peers.data[rank] = static_cast<float*>(c->local);
    c->peers.flags[rank] = flags_of(c->local, c->capacity);
    return c;
}

DTF_API int dtf_p2p_handle_bytes() { return (int)sizeof(hipIpcMemHandle_t); }

DTF_API int dtf_p2p_is_fine_grained(void* h) { return static_cast<Comm*>(h)->fine_grained ? 1 : 0; }

// The handle other processes open to reach this rank's buffer.
DTF_API int dtf_p2p_local_handle(void* h, void* out_handle) {
    auto* c = static_cast<Comm*>(h);
    hipIpcMemHandle_t ipc;
    hipError_t e = hipIpcGetMemHandle(&ipc, c->local);
    if (e != hipSuccess) return (int)e;
    std::memcpy(out_handle, &ipc, sizeof ipc);
    return 0;
}

// `handles`: world consecutive handles in rank order (this rank's own entry is ignored).
DTF_API int dtf_p2p_open_peers(void* h, const void* handles) {
    auto* c = static_cast<Comm*>(h);
    const char* p = static_cast<const char*>(handles);
    for (int r = 0; r < c->world; ++r) {
        if (r == c->rank) continue;
        hipIpcMemHandle_t ipc;
        std::memcpy(&ipc, p + (size_t)r * sizeof ipc, sizeof ipc);
        void* ptr = nullptr;
        hipError_t e = hipIpcOpenMemHandle(&ptr, ipc, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) return (int)e;
        c->opened[r] = ptr;
        c->peers.data[r] = static_cast<float*>(ptr);
        c->peers.flags[r] = flags_of(ptr, c->capacity);
    }
    return 0;
}

// inout[0..n) <- sum over the ranks, on `stream`.  Every rank must call with the same n, in the same order.
DTF_API int dtf_p2p_allreduce(void* h, float* inout, long long n, hipStream_t stream) {
    auto* c = static_cast<Comm*>(h);
    if (n <= 0) return 0;
    if (n > c->capacity) return -2;
    for (int r = 0; r < c->world; ++r)
        if (!c->peers.data[r]) return -3;                   // peers not opened
    if ((reinterpret_cast<uintptr_t>(inout) & 15) != 0) return -4;
    const unsigned call = ++c->call;
    const int half = (int)(call & 1u);
    unsigned* flags = c->peers.flags[c->rank];
    unsigned* done_counter = flags + MAX_RANKS;
    unsigned* error_word = flags + MAX_RANKS + 1;
    float* mine = c->peers.data[c->rank] + (long long)half * c->capacity;
    long long blocks = (n / 4 + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 512) blocks = 512;
    hipLaunchKernelGGL(publish_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, inout, mine, n, c->peers, c->rank, c->world,
                       call, done_counter);
    DTF_CHECK_LAUNCH();
    hipLaunchKernelGGL(reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, inout, n, c->peers, c->rank, c->world, call,
                       half, c->capacity, error_word, c->spin_ticks);
    DTF_CHECK_LAUNCH();
    return 0;
}

// How long a reduce waits for its peers before it gives up (milliseconds; the default is 30 s).
DTF_API void dtf_p2p_set_timeout_ms(void* h, long long ms) {
    if (ms > 0) static_cast<Comm*>(h)->spin_ticks = ms * 100 * 1000;       // 100 MHz wall clock
}

// Non-zero after a reduce gave up waiting for a peer (the value is the call number); reading it synchronises the
// device, so callers poll it rarely.
DTF_API int dtf_p2p_error(void* h) {
    auto* c = static_cast<Comm*>(h);
    unsigned v = 0;
    if (hipMemcpy(&v, flags_of(c->local, c->capacity) + MAX_RANKS + 1, sizeof v, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return (int)v;
}

DTF_API void dtf_p2p_destroy(void* h) {
    auto* c = static_cast<Comm*>(h);
    (void)hipDeviceSynchronize();
    for (int r = 0; r < c->world; ++r)
        if (c->opened[r]) (void)hipIpcCloseMemHandle(c->opened[r]);
    if (c->local) (void)hipFree(c->local);
    delete c;
}
