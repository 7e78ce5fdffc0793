// Owner pump: the native service loop of a parameter-server owner, and the worker-side doorbell.
//
// In the reference a parameter-server task applies every gradient a worker sends as it arrives, lock-free, and the
// workers never wait for one another (/root/reference/example.py:136-139,168-170).  Here an owner is a thread of this
// file plus one HIP stream: it watches the doorbell words of its workers, launches the fused apply-gradients kernel on
// the gradient shard a worker has put into its inbox, and acknowledges through the same words.  Nothing on this path
// crosses a socket or waits for a stream on the host:
//
//   worker w, step t                                      owner j
//   ----------------                                      -------
//   put grad shard -> inbox[w][t % D]   (stream copy)
//   behind it on that stream (kernel / host function):    pump thread sees lane[w].door > consumed[w]
//       lane[w].pull_version[t % D] = version pulled        apply kernel on the owner's stream (var, m, v in its HBM)
//       lane[w].door = t + 1                                behind it on that stream (kernel / host function):
//                                                               lane[w].ack_version[t % D] = version after the apply
//   (before reusing inbox[w][(t+D) % D]:                        ctl.version, ctl.global_step, lane[w].ack = t + 1
//    wait until lane[w].ack > t)
//
// Where the words live.  On GPUs whose memory the CPU can address (large BAR: every MI355X) each rank keeps a replica
// of all control blocks at the end of its peer region, in ITS OWN device memory: a CPU reads words only from the
// replica of its own process (the CPU cannot address memory mapped from another process's hipIpc handle, and host
// memory cannot be registered for GPU access everywhere), and every word is written by a one-workgroup kernel
// that stores it into all replicas through the peer mappings - queued right behind the copies / the apply kernel on
// the same stream (stream order: the data is complete when the word changes; system-scope release stores).  No host
// callback sits between a gradient's arrival and its apply, or between the apply and its acknowledgement.
// Otherwise (CPUs; DTF_DOORBELL_KERNELS=0) the blocks are POSIX shared memory and the words are written by host code:
// directly on a CPU, from stream host functions on a GPU.  The gradient and parameter data itself stays in device
// memory (peer-mapped regions, see peer_mailbox.hip).
//
// On a CPU (the plumbing configuration and the tests) the same thread applies the update itself and the doorbell is
// a plain store.
#include "common.h"

#include <atomic>
#include <chrono>
#include <cmath>
#include <cstring>
#include <thread>
#include <vector>

extern "C" int dtf_adam_apply(float* var, float* m, float* v, const float* g, long long n, float lr, float b1, float b2,
                              float eps, float grad_scale, const long long* step_ptr, long long* advance_step,
                              float* beta_powers, unsigned int* ticket, hipStream_t stream);
extern "C" int dtf_rmsprop_apply(float* var, float* ms, float* mom, const float* g, long long n, float lr, float rho,
                                 float momentum, float eps, int eps_inside, float grad_scale, long long* advance_step,
                                 unsigned int* ticket, hipStream_t stream);
extern "C" int dtf_adadelta_apply(float* var, float* acc, float* acc_update, const float* g, long long n, float lr, float rho,
                                  float eps, float grad_scale, long long* advance_step, unsigned int* ticket,
                                  hipStream_t stream);
extern "C" int dtf_adagrad_apply(float* var, float* acc, const float* g, long long n, float lr, float eps, int eps_inside,
                                 float grad_scale, long long* advance_step, unsigned int* ticket, hipStream_t stream);
extern "C" int dtf_sgd_apply(float* var, float* mom, const float* g, long long n, float lr, float momentum,
                             float grad_scale, long long* advance_step, unsigned int* ticket, hipStream_t stream);

namespace {

constexpr int PUMP_MAX_DEPTH = 4;
constexpr int PUMP_HIST = 64;

// One per (owner, worker).  128 bytes: the worker-written and the owner-written halves sit in different cache lines.
struct Lane {
    std::atomic<uint64_t> door;                 // gradients of this worker complete in the owner's inbox   [worker]
    std::atomic<uint64_t> clock;                // the worker's clock for the staleness bound               [worker]
    uint64_t pull_version[PUMP_MAX_DEPTH];      // owner version the gradient in slot s was computed from   [worker]
    uint64_t pad0[2];
    std::atomic<uint64_t> ack;                  // gradients of this worker applied                         [owner]
    uint64_t ack_version[PUMP_MAX_DEPTH];       // owner version right after the apply of slot s            [owner]
    uint64_t pad1[3];
};
static_assert(sizeof(Lane) == 128, "lane layout is shared with the Python side");

struct Ctl {
    std::atomic<uint64_t> version;              // updates applied by this owner
    std::atomic<uint64_t> global_step;          // owner 0: the cluster-wide step (step at initialisation + version)
    std::atomic<uint64_t> gs_base;              // global step when the owner was (re-)initialised
    uint64_t pad[13];
    Lane lane[1];                               // [workers]
};
static_assert(offsetof(Ctl, lane) == 128, "control block layout is shared with the Python side");

struct Config {
    int on_gpu, owner, workers, depth;
    int optimizer;                              // 0 Adam, 1 SGD (momentum slot optional), 2 RMSProp, 3 Adagrad, 4 Adadelta
    int pad;
    long long n;                                // elements of this owner's shard
    long long slot_stride;                      // elements between inbox slots
    float *var, *m, *v, *inbox;
    long long* step;                            // device (GPU) or host (CPU) int64: updates applied, for bias correction
    float* beta_powers;
    unsigned int* ticket;
    float lr, b1, b2, eps, momentum;
    int pad2;
    void* ctl;                                  // this owner's control block as this process's CPU addresses it
    // replica mode (rep_world > 0): the control areas of all ranks as this GPU addresses them, and the position of this
    // owner's block inside an area (in 64-bit words); acknowledgements are stored into every replica by a kernel
    void* rep_base[64];
    int rep_world;
    int pad3;
    long long block_word;
};

struct Pump;

struct AckRecord {
    Pump* pump;
    int worker, slot;
    uint64_t seq, version, pulled;
};

struct Pump {
    Config cfg;
    Ctl* ctl;
    hipStream_t stream = nullptr;
    std::thread thread;
    std::atomic<int> stop{0}, pause_req{0}, paused{0}, error{0};
    std::vector<uint64_t> consumed;
    uint64_t enqueued_version = 0;              // version after everything launched so far
    std::atomic<uint64_t> hist[PUMP_HIST];
    std::vector<AckRecord> ring;                // records of acknowledgements in flight
    size_t ring_pos = 0;
};

// Word indices inside a lane / the header, in 64-bit units (the layout of Lane / Ctl above).
constexpr int W_DOOR = 0, W_CLOCK = 1, W_PULLV = 2, W_ACK = 8, W_ACKV = 9;
constexpr int W_VERSION = 0, W_GSTEP = 1;

__device__ __forceinline__ void word_store(unsigned long long* p, unsigned long long v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void word_release(unsigned long long* p, unsigned long long v) {
    __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

struct Replicas {
    unsigned long long* base[64];       // the control area of every rank, as this GPU addresses it
    int world;
};

// Owner: the apply queued before this launch on the same stream is complete.  Thread r writes replica r (addresses
// come from the thread index: vector stores), the acknowledgement itself last.
__global__ __launch_bounds__(64) void ack_kernel(Replicas reps, long long block_word, int worker, int slot, unsigned long long seq,
                                                 unsigned long long version, unsigned long long gstep, int owner0) {
    const int r = threadIdx.x;
    if (r < reps.world) {
        unsigned long long* ctl = reps.base[r] + block_word;
        unsigned long long* lane = ctl + 16 + 16 * worker;
        word_store(lane + W_ACKV + slot, version);
        word_store(ctl + W_VERSION, version);
        if (owner0) {
            // owner 0 also publishes the shared step and the worker's clock (its completed pushes): the owners are
            // the ranks that always hold a current mapping of every replica, a restarted worker's included
            word_store(ctl + W_GSTEP, gstep);
            word_store(lane + W_CLOCK, seq);
        }
        __threadfence_system();
        word_release(lane + W_ACK, seq);
    }
}

struct RingArgs {
    long long lane_word[64];            // owner i: position of this worker's lane in owner i's block, inside a control area
    unsigned long long pulled[64];
    int count, slot;
    unsigned long long seq, clock;
};

// Worker: everything queued before this launch on the same stream (the gradient shards in the owners' inboxes) is
// complete.  Thread i rings owner i, in every replica it is given (the owners' and the worker's own).
__global__ __launch_bounds__(64) void ring_kernel(Replicas reps, RingArgs a) {
    const int i = threadIdx.x;
    if (i < a.count) {
        // (the worker's clock is published by owner 0 when it acknowledges this gradient: see ack_kernel)
        for (int r = 0; r < reps.world; ++r) word_store(reps.base[r] + a.lane_word[i] + W_PULLV + a.slot, a.pulled[i]);
        __threadfence_system();
        for (int r = 0; r < reps.world; ++r) word_release(reps.base[r] + a.lane_word[i] + W_DOOR, a.seq);
    }
}

// One word of the control area, the same value into every replica (the rare updates host code makes: a finished
// worker's clock, a re-based step, a restarted worker's lane).
__global__ __launch_bounds__(64) void word_kernel(Replicas reps, long long word, unsigned long long value) {
    const int r = threadIdx.x;
    if (r < reps.world) word_release(reps.base[r] + word, value);
}

bool fill_replicas(Replicas& reps, void* const* bases, int world) {
    if (!bases || world < 1 || world > 64) return false;
    reps.world = world;
    for (int r = 0; r < world; ++r) {
        if (!bases[r]) return false;
        reps.base[r] = static_cast<unsigned long long*>(bases[r]);
    }
    return true;
}

void acknowledge(const AckRecord& r) {
    Pump* p = r.pump;
    Lane& lane = p->ctl->lane[r.worker];
    lane.ack_version[r.slot] = r.version;
    p->ctl->version.store(r.version, std::memory_order_release);
    if (p->cfg.owner == 0)
        p->ctl->global_step.store(p->ctl->gs_base.load(std::memory_order_relaxed) + r.version, std::memory_order_release);
    const uint64_t before = r.version - 1;
    uint64_t st = before > r.pulled ? before - r.pulled : 0;
    if (st >= (uint64_t)PUMP_HIST) st = PUMP_HIST - 1;
    p->hist[st].fetch_add(1, std::memory_order_relaxed);
    lane.ack.store(r.seq, std::memory_order_release);
}

void ack_host_fn(void* data) { acknowledge(*static_cast<AckRecord*>(data)); }

// The TensorFlow ApplyAdam formula (epsilon outside the bias correction), as optim.hip computes it.
void cpu_apply(const Config& c, const float* g) {
    const long long n = c.n;
    if (c.optimizer == 0) {
        const float t = (float)(*c.step + 1);
        const float lr_t = c.lr * std::sqrt(1.0f - std::pow(c.b2, t)) / (1.0f - std::pow(c.b1, t));
        const float omb1 = 1.0f - c.b1, omb2 = 1.0f - c.b2;
        for (long long i = 0; i < n; ++i) {
            const float gi = g[i];
            const float mi = c.b1 * c.m[i] + omb1 * gi;
            const float vi = c.b2 * c.v[i] + omb2 * gi * gi;
            c.m[i] = mi;
            c.v[i] = vi;
            c.var[i] -= lr_t * mi / (std::sqrt(vi) + c.eps);
        }
        if (c.beta_powers) {
            c.beta_powers[0] *= c.b1;
            c.beta_powers[1] *= c.b2;
        }
    } else if (c.optimizer == 1) {
        for (long long i = 0; i < n; ++i) {
            float gi = g[i];
            if (c.m && c.momentum > 0.0f) {
                gi = c.momentum * c.m[i] + gi;
                c.m[i] = gi;
            }
            c.var[i] -= c.lr * gi;
        }
    } else if (c.optimizer == 4) {
        // Adadelta: m = accumulated squared gradients, v = accumulated squared updates, b1 = rho
        for (long long i = 0; i < n; ++i) {
            const float gi = g[i];
            const float a = c.b1 * c.m[i] + (1.0f - c.b1) * gi * gi;
            const float u = gi * std::sqrt(c.v[i] + c.eps) / std::sqrt(a + c.eps);
            c.m[i] = a;
            c.v[i] = c.b1 * c.v[i] + (1.0f - c.b1) * u * u;
            c.var[i] -= c.lr * u;
        }
    } else {
        // RMSProp (2: m = mean square, v = momentum accumulator or null, b1 = rho) and Adagrad (3: m = accumulator);
        // b2 != 0: epsilon inside the root (TensorFlow's kernels), else outside (Keras' formulas) - as optim.hip
        const bool inside = c.b2 != 0.0f;
        for (long long i = 0; i < n; ++i) {
            const float gi = g[i];
            const float a = c.optimizer == 2 ? c.b1 * c.m[i] + (1.0f - c.b1) * gi * gi : c.m[i] + gi * gi;
            c.m[i] = a;
            float u = c.lr * gi / (inside ? std::sqrt(a + c.eps) : std::sqrt(a) + c.eps);
            if (c.optimizer == 2 && c.v && c.momentum > 0.0f) {
                u = c.momentum * c.v[i] + u;
                c.v[i] = u;
            }
            c.var[i] -= u;
        }
    }
    *c.step += 1;
}

int gpu_apply(Pump* p, const float* g) {
    const Config& c = p->cfg;
    if (c.n <= 0) return 0;
    if (c.optimizer == 0)
        return dtf_adam_apply(c.var, c.m, c.v, g, c.n, c.lr, c.b1, c.b2, c.eps, 1.0f, c.step, c.step, c.beta_powers,
                              c.ticket, p->stream);
    if (c.optimizer == 1)
        return dtf_sgd_apply(c.var, c.momentum > 0.0f ? c.m : nullptr, g, c.n, c.lr, c.momentum, 1.0f, c.step, c.ticket,
                             p->stream);
    if (c.optimizer == 2)
        return dtf_rmsprop_apply(c.var, c.m, c.momentum > 0.0f ? c.v : nullptr, g, c.n, c.lr, c.b1, c.momentum, c.eps,
                                 c.b2 != 0.0f ? 1 : 0, 1.0f, c.step, c.ticket, p->stream);
    if (c.optimizer == 4)
        return dtf_adadelta_apply(c.var, c.m, c.v, g, c.n, c.lr, c.b1, c.eps, 1.0f, c.step, c.ticket, p->stream);
    return dtf_adagrad_apply(c.var, c.m, g, c.n, c.lr, c.eps, c.b2 != 0.0f ? 1 : 0, 1.0f, c.step, c.ticket, p->stream);
}

void pump_loop(Pump* p, int device) {
    const Config& c = p->cfg;
    if (c.on_gpu && hipSetDevice(device) != hipSuccess) {
        p->error.store(-10);
        return;
    }
    int idle = 0, first = 0;
    while (!p->stop.load(std::memory_order_acquire)) {
        if (p->pause_req.load(std::memory_order_acquire)) {
            if (c.on_gpu && hipStreamSynchronize(p->stream) != hipSuccess) p->error.store(-11);
            p->paused.store(1, std::memory_order_release);
            while (p->pause_req.load(std::memory_order_acquire) && !p->stop.load(std::memory_order_acquire))
                std::this_thread::sleep_for(std::chrono::microseconds(100));
            p->paused.store(0, std::memory_order_release);
            continue;
        }
        bool any = false;
        for (int k = 0; k < c.workers; ++k) {
            const int w = (first + k) % c.workers;          // rotate the starting lane: no worker is always served last
            Lane& lane = p->ctl->lane[w];
            const uint64_t door = lane.door.load(std::memory_order_acquire);
            if (door <= p->consumed[w]) continue;
            const uint64_t seq = p->consumed[w];
            const int slot = (int)(seq % (uint64_t)c.depth);
            const float* g = c.inbox + ((long long)w * c.depth + slot) * c.slot_stride;
            AckRecord rec{p, w, slot, seq + 1, ++p->enqueued_version, lane.pull_version[slot]};
            if (c.on_gpu && c.rep_world > 0) {
                // acknowledged by a kernel behind the apply; the applies of this owner are serial, so the version this
                // one will leave behind - and with it the staleness of the gradient - is known now
                int rc = gpu_apply(p, g);
                if (rc == 0) {
                    const uint64_t before = rec.version - 1;
                    uint64_t st = before > rec.pulled ? before - rec.pulled : 0;
                    if (st >= (uint64_t)PUMP_HIST) st = PUMP_HIST - 1;
                    p->hist[st].fetch_add(1, std::memory_order_relaxed);
                    Replicas reps;
                    (void)fill_replicas(reps, c.rep_base, c.rep_world);
                    hipLaunchKernelGGL(ack_kernel, dim3(1), dim3(64), 0, p->stream, reps, c.block_word, w,
                                       slot, (unsigned long long)rec.seq, (unsigned long long)rec.version,
                                       (unsigned long long)(p->ctl->gs_base.load(std::memory_order_relaxed) + rec.version),
                                       c.owner == 0 ? 1 : 0);
                    rc = (int)hipGetLastError();
                }
                if (rc != 0) {
                    p->error.store(rc);
                    return;
                }
            } else if (c.on_gpu) {
                int rc = gpu_apply(p, g);
                AckRecord* slot_rec = &p->ring[p->ring_pos];
                p->ring_pos = (p->ring_pos + 1) % p->ring.size();
                *slot_rec = rec;
                if (rc == 0) rc = (int)hipLaunchHostFunc(p->stream, ack_host_fn, slot_rec);
                if (rc != 0) {
                    p->error.store(rc);
                    return;
                }
            } else {
                if (c.n > 0) cpu_apply(c, g);
                acknowledge(rec);
            }
            p->consumed[w] = seq + 1;
            any = true;
        }
        first = (first + 1) % (c.workers > 0 ? c.workers : 1);
        if (any) {
            idle = 0;
        } else if (++idle < 200) {
            __builtin_ia32_pause();
        } else {
            std::this_thread::sleep_for(std::chrono::microseconds(idle < 2000 ? 20 : 200));
        }
    }
    if (c.on_gpu) (void)hipStreamSynchronize(p->stream);
}

struct RingRecord {
    int count, slot;
    uint64_t seq, clock;
    Lane* lanes[64];
    uint64_t pulled[64];
};

void ring_now(const RingRecord& r) {
    for (int i = 0; i < r.count; ++i) {
        Lane* lane = r.lanes[i];
        lane->pull_version[r.slot] = r.pulled[i];
        if (i == 0) lane->clock.store(r.clock, std::memory_order_release);   // lanes[0] belongs to owner 0's block
        lane->door.store(r.seq, std::memory_order_release);
    }
}

void ring_host_fn(void* data) {
    RingRecord* r = static_cast<RingRecord*>(data);
    ring_now(*r);
    delete r;
}

}  // namespace

DTF_API int dtf_pump_config_bytes() { return (int)sizeof(Config); }
DTF_API int dtf_pump_lane_bytes() { return (int)sizeof(Lane); }
DTF_API long long dtf_pump_ctl_bytes(int workers) { return (long long)(offsetof(Ctl, lane) + sizeof(Lane) * (size_t)workers); }

// Starts the owner's service thread (and, on a GPU, its stream) for the shard described by `cfg`.  The control block
// must have been zeroed by the caller; `consumed` starts at the doorbell values found there.
DTF_API void* dtf_pump_start(const void* cfg_bytes) {
    Config cfg;
    std::memcpy(&cfg, cfg_bytes, sizeof cfg);
    if (cfg.workers < 1 || cfg.workers > 64 || cfg.depth < 1 || cfg.depth > PUMP_MAX_DEPTH || !cfg.ctl || cfg.n < 0)
        return nullptr;
    if (cfg.n > 0 && (!cfg.var || !cfg.inbox || !cfg.step)) return nullptr;
    auto* p = new Pump();
    p->cfg = cfg;
    p->ctl = static_cast<Ctl*>(cfg.ctl);
    p->consumed.resize(cfg.workers);
    for (int w = 0; w < cfg.workers; ++w) p->consumed[w] = p->ctl->lane[w].ack.load();
    p->enqueued_version = p->ctl->version.load();
    for (auto& h : p->hist) h.store(0);
    // every worker has at most `depth` gradients outstanding, so workers * depth records are never all in flight
    p->ring.resize((size_t)cfg.workers * cfg.depth + 8);
    int device = 0;
    if (cfg.on_gpu) {
        if (hipGetDevice(&device) != hipSuccess) { delete p; return nullptr; }
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        if (hipStreamCreateWithPriority(&p->stream, hipStreamNonBlocking, hi) != hipSuccess) {
            (void)hipGetLastError();
            delete p;
            return nullptr;
        }
    }
    p->thread = std::thread(pump_loop, p, device);
    return p;
}

// Stops applying (everything already launched completes) / resumes.  While paused the owner's shard is stable.
DTF_API int dtf_pump_pause(void* h) {
    auto* p = static_cast<Pump*>(h);
    p->pause_req.store(1, std::memory_order_release);
    for (int i = 0; i < 600000 && !p->paused.load(std::memory_order_acquire); ++i) {
        if (p->error.load()) return p->error.load();
        std::this_thread::sleep_for(std::chrono::microseconds(100));
    }
    return p->paused.load() ? 0 : -12;
}

DTF_API int dtf_pump_resume(void* h, int resync) {
    auto* p = static_cast<Pump*>(h);
    if (resync) {
        // the control block was reset while paused (owner re-initialised): start over from what it says now
        for (int w = 0; w < p->cfg.workers; ++w) p->consumed[w] = p->ctl->lane[w].ack.load();
        p->enqueued_version = p->ctl->version.load();
    }
    p->pause_req.store(0, std::memory_order_release);
    for (int i = 0; i < 600000 && p->paused.load(std::memory_order_acquire); ++i)
        std::this_thread::sleep_for(std::chrono::microseconds(100));
    return 0;
}

// Replica mode, pump paused: rank r's control area is now at `base` (a restarted worker's new region).
DTF_API int dtf_pump_set_replica(void* h, int r, void* base) {
    auto* p = static_cast<Pump*>(h);
    if (r < 0 || r >= p->cfg.rep_world || !base || !p->paused.load()) return -2;
    p->cfg.rep_base[r] = base;
    return 0;
}

DTF_API int dtf_pump_error(void* h) { return static_cast<Pump*>(h)->error.load(); }

DTF_API void* dtf_pump_stream(void* h) { return static_cast<Pump*>(h)->stream; }

DTF_API void dtf_pump_histogram(void* h, long long* out) {
    auto* p = static_cast<Pump*>(h);
    for (int i = 0; i < PUMP_HIST; ++i) out[i] = (long long)p->hist[i].load();
}

DTF_API int dtf_pump_stop(void* h) {
    auto* p = static_cast<Pump*>(h);
    p->stop.store(1, std::memory_order_release);
    if (p->thread.joinable()) p->thread.join();
    const int err = p->error.load();
    if (p->stream) (void)hipStreamDestroy(p->stream);
    delete p;
    return err;
}

// Worker side: after everything queued on `stream` so far (the puts of this step's gradient shards), tell `count`
// owners that gradient number `seq` (1-based) of this worker is complete in slot `slot` of their inboxes; `pulled[i]`
// is the owner version the gradient was computed from; `clock` is published in owner 0's block for the staleness
// bound.  Two forms: `replicas` != null (GPU, replica mode) - the control areas of all `world` ranks as this GPU
// addresses them, `lane_words[i]` the position of this worker's lane in owner i's block inside an area: a kernel on
// `stream` stores the words into every replica.  Otherwise `lanes[i]` is the lane in owner i's shared-memory block:
// a host function on `stream` writes it (stream == null on a CPU: the stores happen now).
DTF_API int dtf_doorbell_ring(hipStream_t stream, int on_gpu, void* const* lanes, void* const* replicas, int world,
                              const long long* lane_words, const unsigned long long* pulled, int count, int slot,
                              unsigned long long seq, unsigned long long clock) {
    if (count < 1 || count > 64 || slot < 0 || slot >= PUMP_MAX_DEPTH) return -2;
    if (on_gpu && replicas) {
        Replicas reps;
        if (!fill_replicas(reps, replicas, world) || !lane_words) return -2;
        RingArgs a;
        a.count = count; a.slot = slot; a.seq = seq; a.clock = clock;
        for (int i = 0; i < count; ++i) {
            a.lane_word[i] = lane_words[i];
            a.pulled[i] = pulled[i];
        }
        hipLaunchKernelGGL(ring_kernel, dim3(1), dim3(64), 0, stream, reps, a);
        return (int)hipGetLastError();
    }
    if (!lanes) return -2;
    auto* r = new RingRecord();
    r->count = count;
    r->slot = slot;
    r->seq = seq;
    r->clock = clock;
    for (int i = 0; i < count; ++i) {
        r->lanes[i] = static_cast<Lane*>(lanes[i]);
        r->pulled[i] = pulled[i];
    }
    if (!on_gpu) {
        ring_now(*r);
        delete r;
        return 0;
    }
    const hipError_t e = hipLaunchHostFunc(stream, ring_host_fn, r);
    if (e != hipSuccess) {
        delete r;
        return (int)e;
    }
    return 0;
}

// Replica mode: store `value` into 64-bit word `word` of every rank's control area, on `stream` (null stream allowed).
DTF_API int dtf_words_broadcast(void* const* replicas, int world, long long word, unsigned long long value, hipStream_t stream) {
    Replicas reps;
    if (!fill_replicas(reps, replicas, world) || word < 0) return -2;
    hipLaunchKernelGGL(word_kernel, dim3(1), dim3(64), 0, stream, reps, word, value);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    return (int)e;
}

// 1 when this process's CPU can address the current device's memory (large BAR): what the replica mode needs.
DTF_API int dtf_device_large_bar() {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return prop.isLargeBar ? 1 : 0;
}

// ---------------------------------------------------------------------------------------------------------------
// A worker's link to its owners: the communication stream and the two events of the push choreography, so that one
// native call does what the step needs - order the copies behind the backward pass, move the gradient into the
// owners' inboxes, ring the doorbells, and let the compute stream reuse the gradient buffer once it has been read.
extern "C" int dtf_mbox_ranges_run(void* h, void* table, int gather, long long remote_base, void* local, long long local_bytes,
                                   hipStream_t stream);
extern "C" int dtf_mbox_scatter(void* h, long long dst_off, const void* src, long long shard_bytes, hipStream_t stream);

namespace {
struct Link {
    hipStream_t comm = nullptr;
    hipEvent_t ev_grad = nullptr, ev_copied = nullptr;
};
}  // namespace

DTF_API void* dtf_link_create() {
    auto* l = new Link();
    if (hipStreamCreateWithFlags(&l->comm, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&l->ev_grad, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&l->ev_copied, hipEventDisableTiming) != hipSuccess) {
        (void)hipGetLastError();
        if (l->ev_grad) (void)hipEventDestroy(l->ev_grad);
        if (l->comm) (void)hipStreamDestroy(l->comm);
        delete l;
        return nullptr;
    }
    return l;
}

DTF_API void dtf_link_destroy(void* h) {
    auto* l = static_cast<Link*>(h);
    if (!l) return;
    (void)hipStreamSynchronize(l->comm);
    (void)hipEventDestroy(l->ev_grad);
    (void)hipEventDestroy(l->ev_copied);
    (void)hipStreamDestroy(l->comm);
    delete l;
}

DTF_API void* dtf_link_stream(void* h) { return static_cast<Link*>(h)->comm; }

// One training step's push.  `table` != null: the gradient's flat ranges go to `remote_base` of the owners' regions
// (ps tasks as owners); else shard j of `grad` (shard_bytes each) goes to offset `remote_base` of rank j's region
// (owners colocated with the workers).  Offsets in bytes.  Everything is queued; nothing waits on the host.
DTF_API int dtf_link_push(void* link, void* mbox, void* table, hipStream_t cur, const void* grad, long long grad_bytes,
                          long long remote_base, long long shard_bytes, void* const* lanes, void* const* replicas, int world,
                          const long long* lane_words, const unsigned long long* pulled, int count, int slot,
                          unsigned long long seq, unsigned long long clock) {
    auto* l = static_cast<Link*>(link);
    if (!l || !mbox || !grad) return -2;
    hipError_t e = hipEventRecord(l->ev_grad, cur);
    if (e == hipSuccess) e = hipStreamWaitEvent(l->comm, l->ev_grad, 0);
    if (e != hipSuccess) return (int)e;
    int rc = table ? dtf_mbox_ranges_run(mbox, table, 0, remote_base, const_cast<void*>(grad), grad_bytes, l->comm)
                   : dtf_mbox_scatter(mbox, remote_base, grad, shard_bytes, l->comm);
    if (rc) return rc;
    e = hipEventRecord(l->ev_copied, l->comm);
    if (e != hipSuccess) return (int)e;
    rc = dtf_doorbell_ring(l->comm, 1, lanes, replicas, world, lane_words, pulled, count, slot, seq, clock);
    if (rc) return rc;
    return (int)hipStreamWaitEvent(cur, l->ev_copied, 0);     // the next backward pass may overwrite the gradient buffer
}
